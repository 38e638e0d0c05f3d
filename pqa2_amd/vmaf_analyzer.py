"""VMAFAnalyzer -- drop-in for the reference's app/vmaf_analyzer.py:18 `class VMAFAnalyzer(QObject)`.

Same constructor state (app/vmaf_analyzer.py:25-39), setters (:44-137), `terminate_analysis` (:139),
`get_video_metadata` (:162), `analyze_videos(reference_path, distorted_path, model, duration)` (:242)
and result dict (:919-932), same four signals (:20-23) and the same files on disk
(`<test>_<ts>_vmaf.json`, `_psnr.txt`, `_ssim.txt`, :304-311).  What changed is what happens where
the reference spawns `ffmpeg -lavfi libvmaf=...` (:446) and `psnr=`/`ssim=` (:1037,:1067): the
frame pairs go through the MI355X HIP kernels behind include/pqa_vmaf.h instead, all three metrics in
ONE pass over the decoded frames (the reference decodes both files three times).

With PyQt5 importable the class is a real QObject with pyqtSignals, so app/ui/tabs/analysis_tab.py
(:585-640) can use it unchanged; otherwise a small signal shim with the same connect/emit surface is
used (plain callables, as SURVEY.md 8(b) prescribes).
"""
from __future__ import annotations

import json
import logging
import os
import subprocess
import sys
import threading
import time
from datetime import datetime

logger = logging.getLogger(__name__)
_METADATA_CACHE = {}   # (abspath, size, mtime_ns) -> metadata dict of get_video_metadata

try:  # pragma: no cover - PyQt5 is not installed in the build container
    from PyQt5.QtCore import QObject, pyqtSignal
    _HAVE_QT = True
except Exception:
    _HAVE_QT = False

    class _BoundSignal:
        def __init__(self):
            self._slots = []
            self._lock = threading.Lock()

        def connect(self, slot):
            with self._lock:
                self._slots.append(slot)

        def disconnect(self, slot=None):
            with self._lock:
                if slot is None:
                    self._slots.clear()
                else:
                    self._slots.remove(slot)

        def emit(self, *args):
            with self._lock:
                slots = list(self._slots)
            for s in slots:
                s(*args)

    class pyqtSignal:  # noqa: N801 - mirrors the Qt name
        """Class-level declaration that hands every instance its own connect/emit object."""

        def __init__(self, *types):
            self._name = None

        def __set_name__(self, owner, name):
            self._name = "_sig_" + name

        def __get__(self, obj, objtype=None):
            if obj is None:
                return self
            sig = obj.__dict__.get(self._name)
            if sig is None:
                sig = obj.__dict__[self._name] = _BoundSignal()
            return sig

    class QObject:  # noqa: D401
        def __init__(self, *a, **k):
            pass


# Child jobs (torchrun agent + one worker per GPU) run in their own session so that a stop reaches the whole group; the
# price is that a Ctrl-C / SIGTERM aimed at THIS process no longer reaches them.  Every live child is therefore registered
# here: an atexit hook and, where nothing else has claimed them, SIGTERM / SIGINT handlers stop the groups before this
# process goes (the workers also ask the kernel for SIGTERM on parent death, pqa2_amd/score.py).
_LIVE_CHILDREN = set()
_HOOKS_INSTALLED = False


def _stop_live_children():
    for proc in list(_LIVE_CHILDREN):
        try:
            VMAFAnalyzer._stop_child(proc, grace=1.0)
        except Exception:
            pass
        _LIVE_CHILDREN.discard(proc)


def _install_child_hooks():
    global _HOOKS_INSTALLED
    if _HOOKS_INSTALLED:
        return
    _HOOKS_INSTALLED = True
    import atexit
    import signal
    import threading
    atexit.register(_stop_live_children)
    if threading.current_thread() is not threading.main_thread():
        return                                  # signal handlers can only be set from the main thread
    for sig in (signal.SIGTERM, signal.SIGINT):
        try:
            prev = signal.getsignal(sig)
            if prev not in (signal.SIG_DFL, signal.default_int_handler):
                continue                        # the application has its own handler: leave it alone (atexit still runs)

            def handler(signum, frame, _prev=prev):
                _stop_live_children()
                signal.signal(signum, _prev)    # then what would have happened anyway
                os.kill(os.getpid(), signum)
            signal.signal(sig, handler)
        except Exception:
            pass


class VMAFAnalyzer(QObject):
    """VMAF analyzer for measuring video quality with signals for UI integration."""
    analysis_progress = pyqtSignal(int)   # 0-100%
    analysis_complete = pyqtSignal(dict)  # results dict
    error_occurred = pyqtSignal(str)
    status_update = pyqtSignal(str)

    def __init__(self):
        super().__init__()
        self.output_directory = None
        self.test_name = None
        self._process_lock = threading.Lock()
        self._current_process = None
        self._terminate_requested = False
        self.threads = 4                      # kept for interface parity; the GPU path ignores it
        self.pool_method = "mean"
        self.enable_motion_score = False
        self.enable_temporal_features = False
        self.feature_subsample = 1
        self.psnr_enabled = True
        self.ssim_enabled = True
        # extensions (defaults reproduce the reference's behaviour)
        self.device = 0                       # HIP device ordinal for single-process runs
        self.gpus = 1                         # >1: frame-sharded child job, one process per GPU
        self.max_batch = 0                    # 0: the library sizes launches itself
        self.child_backend = "nccl"           # collective backend of the child job: nccl (= RCCL) | gloo
        self.child_share_device = False       # True: every rank of the child job uses device 0 (one-GPU rehearsal)
        self.fixed_point = 0                  # PQA_FIXED_* mask (1 VIF, 2 motion): libvmaf's integer arithmetic, slower
        self.last_fps = 0.0
        self._engine_factory = None           # tests inject a stand-in; product code leaves it None

    # ---- option plumbing (app/vmaf_analyzer.py:44-137) ------------------------------------------
    def set_options_from_manager(self, options_manager):
        if not options_manager:
            logger.warning("No options manager provided, using default settings")
            return
        try:
            s = options_manager.get_setting("vmaf")
            self.threads = s.get("threads", 4)
            self.feature_subsample = s.get("feature_subsample", 1)
            self.pool_method = s.get("pool_method", "mean")
            self.enable_motion_score = s.get("enable_motion_score", False)
            self.enable_temporal_features = s.get("enable_temporal_features", False)
            self.psnr_enabled = s.get("psnr_enabled", True)
            self.ssim_enabled = s.get("ssim_enabled", True)
            logger.info(f"VMAF options set from manager: threads={self.threads}, "
                        f"feature_subsample={self.feature_subsample}, pool={self.pool_method}")
        except Exception as e:
            logger.error(f"Error setting VMAF options from manager: {e}")

    set_options_manager = set_options_from_manager

    def set_output_directory(self, output_dir):
        self.output_directory = output_dir
        logger.info(f"Set output directory to: {self.output_directory}")

    def set_test_name(self, test_name):
        self.test_name = test_name
        logger.info(f"Set test name to: {test_name}")

    def set_advanced_options(self, pool_method="mean", enable_motion_score=False, enable_temporal_features=False,
                             feature_subsample=1, psnr_enabled=True, ssim_enabled=True):
        self.pool_method = pool_method
        self.enable_motion_score = enable_motion_score
        self.enable_temporal_features = enable_temporal_features
        self.feature_subsample = feature_subsample
        self.psnr_enabled = psnr_enabled
        self.ssim_enabled = ssim_enabled

    def terminate_analysis(self):
        """Terminate a running analysis (legal from another thread, like the reference's)."""
        self._terminate_requested = True
        proc = self._current_process
        if proc is not None:
            try:
                self._stop_child(proc)
            except Exception as e:
                logger.error(f"Error terminating VMAF process: {e}")

    @staticmethod
    def _stop_child(proc, grace=3.0):
        """The child job is a torchrun agent PLUS one worker per GPU (the reference had a single ffmpeg child).  It is
        started as its own session, so SIGTERM goes to the whole process group, and after `grace` seconds SIGKILL does:
        no worker is left behind holding a GPU context or blocked in the RCCL gather."""
        import signal
        if proc.poll() is not None:
            return
        try:
            pgid = os.getpgid(proc.pid)
        except ProcessLookupError:
            return
        own_group = pgid == proc.pid      # only signal a group this analyzer created (start_new_session=True)
        try:
            os.killpg(pgid, signal.SIGTERM) if own_group else proc.terminate()
        except ProcessLookupError:
            return
        try:
            proc.wait(timeout=grace)
        except subprocess.TimeoutExpired:
            pass
        try:
            os.killpg(pgid, signal.SIGKILL) if own_group else proc.kill()   # stragglers of the group, if any
        except ProcessLookupError:
            pass

    # ---- metadata (replaces the ffprobe call, app/vmaf_analyzer.py:162-240) -------------------------
    def get_video_metadata(self, video_path, ffprobe_exe=None):
        try:
            from .yuvio import open_video
            # one analysis asks four times for the same two files (app/vmaf_analyzer.py:320-321, :919-920): the answer is
            # kept per (path, size, mtime) -- opening a clip maps it and finds its frames
            st = os.stat(video_path)
            key = (os.path.abspath(video_path), st.st_size, st.st_mtime_ns)
            hit = _METADATA_CACHE.get(key)
            if hit is not None:
                logger.info(f"Video metadata extracted: {hit['width']}x{hit['height']} @ {hit['frame_rate']}fps")
                return dict(hit, path=video_path)
            info = open_video(video_path).info
            fps = info.fps
            md = {
                "path": video_path,
                "duration": (info.n_frames / fps) if fps else 0.0,
                "frame_rate": fps,
                "width": info.width,
                "height": info.height,
                "pix_fmt": info.pix_fmt,
                "codec_name": "rawvideo",
                "bit_rate": int(info.frame_bytes * 8 * fps) if fps else 0,
                "nb_frames": info.n_frames,
            }
            logger.info(f"Video metadata extracted: {md['width']}x{md['height']} @ {md['frame_rate']}fps")
            if len(_METADATA_CACHE) > 64:
                _METADATA_CACHE.clear()
            _METADATA_CACHE[key] = dict(md)
            return md
        except Exception as e:
            logger.error(f"Error extracting video metadata: {e}")
            return None

    # ---- the entry point (app/vmaf_analyzer.py:242-616) ---------------------------------------------
    def analyze_videos(self, reference_path, distorted_path, model="vmaf_v0.6.1", duration=None):
        """Score `distorted_path` against `reference_path`.  Returns the results dict or None;
        never raises (errors go to `error_occurred`, as in the reference).  `duration` is accepted and
        unused, exactly like the reference's."""
        with self._process_lock:
            try:
                self._terminate_requested = False
                self.status_update.emit(f"Analyzing videos with model: {model}")
                logger.info(f"Starting VMAF analysis with model: {model}")
                if not os.path.exists(reference_path):
                    return self._fail(f"Reference video not found: {reference_path}")
                if not os.path.exists(distorted_path):
                    return self._fail(f"Distorted video not found: {distorted_path}")

                output_dir = self.output_directory or os.path.dirname(reference_path)
                timestamp = datetime.now().strftime("%Y%m%d_%H%M%S")
                test_name = self.test_name or "Test"
                parent_dir = os.path.dirname(reference_path)
                if test_name and test_name in parent_dir:
                    test_dir = parent_dir
                else:
                    test_dir = os.path.join(output_dir, f"{test_name}_{timestamp}")
                    os.makedirs(test_dir, exist_ok=True)
                json_path = os.path.join(test_dir, f"{test_name}_{timestamp}_vmaf.json")
                psnr_path = os.path.join(test_dir, f"{test_name}_{timestamp}_psnr.txt")
                ssim_path = os.path.join(test_dir, f"{test_name}_{timestamp}_ssim.txt")

                self.get_video_metadata(reference_path)
                dist_meta = self.get_video_metadata(distorted_path)
                total_frames = 0
                if dist_meta:
                    if dist_meta.get("nb_frames", 0) > 0:
                        total_frames = dist_meta["nb_frames"]
                    elif dist_meta.get("frame_rate", 0) > 0 and dist_meta.get("duration", 0) > 0:
                        total_frames = int(dist_meta["frame_rate"] * dist_meta["duration"])
                if model is None:
                    model = "vmaf_v0.6.1"

                self.analysis_progress.emit(0)
                self.status_update.emit("Starting VMAF analysis...")
                ok = (self._run_child_job if self.gpus > 1 else self._run_in_process)(
                    reference_path, distorted_path, model, json_path,
                    psnr_path if self.psnr_enabled else None, ssim_path if self.ssim_enabled else None,
                    total_frames)
                if not ok:
                    return None
                if self.psnr_enabled or self.ssim_enabled:
                    # one pass already produced these; the status lines keep the reference's sequence
                    self.status_update.emit("VMAF completed, running PSNR/SSIM analysis...")
                return self._parse_vmaf_results(json_path, psnr_path if self.psnr_enabled else None,
                                                ssim_path if self.ssim_enabled else None,
                                                distorted_path, reference_path)
            except Exception as e:
                import traceback
                logger.error(traceback.format_exc())
                return self._fail(f"Error in VMAF analysis: {e}")

    analyze = analyze_videos  # BASELINE.json's name for the entry point

    def _fail(self, msg):
        logger.error(msg)
        self.error_occurred.emit(msg)
        return None

    def _emit_progress(self, done, total, state):
        if total <= 0:
            return
        progress = min(95, int(done / total * 100))
        now = time.time()
        if now - state["t"] > 0.5:
            state["t"] = now
            self.analysis_progress.emit(progress)
            self.status_update.emit(f"Processing frame {done}/{total} ({progress}%)")

    def _run_in_process(self, ref, dis, model, json_path, psnr_path, ssim_path, total_frames):
        from . import _native as N
        from . import report
        from .pipeline import score_files
        state = {"t": time.time()}
        try:
            res = score_files(ref, dis, model, psnr=bool(psnr_path), ssim=bool(ssim_path),
                              n_subsample=max(1, int(self.feature_subsample or 1)), device=self.device,
                              max_batch=self.max_batch, engine_factory=self._engine_factory,
                              fixed_point=self.fixed_point,
                              progress=lambda d, t: self._emit_progress(d, total_frames or t, state),
                              cancelled=lambda: self._terminate_requested)
        except N.PqaCancelled:
            self._fail("VMAF analysis was terminated by user")
            return False
        except Exception as e:
            self._fail(f"Error running VMAF analysis: {e}")
            return False
        if self._terminate_requested:
            self._fail("VMAF analysis was terminated by user")
            return False
        self.last_fps = res["fps"]
        log = report.build_vmaf_log(res["metrics"], res["fps"], res["frame_indices"], {"model": res["model_name"]})
        report.write_vmaf_json(json_path, log)
        if psnr_path and res["psnr_lines"] is not None:
            self.status_update.emit("Running PSNR analysis...")
            with open(psnr_path, "w") as f:
                f.write("\n".join(res["psnr_lines"]) + "\n")
        if ssim_path and res["ssim_lines"] is not None:
            self.status_update.emit("Running SSIM analysis...")
            with open(ssim_path, "w") as f:
                f.write("\n".join(res["ssim_lines"]) + "\n")
        return True

    def _run_child_job(self, ref, dis, model, json_path, psnr_path, ssim_path, total_frames):
        """Frame-sharded run: one process per GPU under torch.distributed.run, driven like the
        reference drives its ffmpeg child (stderr `frame=` lines, terminate -> kill, return code)."""
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={self.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "pqa2_amd.score",
               ref, dis, "--model", model, "--json", json_path,
               "--n-subsample", str(max(1, int(self.feature_subsample or 1))), "--batch", str(self.max_batch)]
        if self.fixed_point:
            cmd += ["--fixed-point", str(int(self.fixed_point))]
        if self.child_backend != "nccl":
            cmd += ["--backend", self.child_backend]
        if self.child_share_device:
            cmd += ["--share-device"]
        if psnr_path:
            cmd += ["--psnr-log", psnr_path]
        if ssim_path:
            cmd += ["--ssim-log", ssim_path]
        env = os.environ.copy()
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        state = {"t": time.time()}
        stderr_lines = []
        try:
            _install_child_hooks()
            self._current_process = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                                                     bufsize=1, env=env, start_new_session=True)
            _LIVE_CHILDREN.add(self._current_process)
            for line in iter(self._current_process.stderr.readline, ""):
                if self._terminate_requested:
                    break
                stderr_lines.append(line)
                if "frame=" in line:
                    tok = line.split("frame=")[1].split()
                    if tok and tok[0].isdigit():
                        self._emit_progress(int(tok[0]), total_frames, state)
            if self._terminate_requested:
                self.terminate_analysis()
            returncode = self._current_process.wait(timeout=60)
        except Exception as e:
            self._fail(f"Error running VMAF analysis: {e}")
            return False
        finally:
            proc, self._current_process = self._current_process, None
            if proc is not None and proc.poll() is None:
                self._stop_child(proc, grace=1.0)
            _LIVE_CHILDREN.discard(proc)
        if self._terminate_requested:
            self._fail("VMAF analysis was terminated by user")
            return False
        if returncode != 0:
            self._fail(f"VMAF analysis failed with return code {returncode}: {''.join(stderr_lines[-20:])}")
            return False
        return True

    # ---- result assembly (app/vmaf_analyzer.py:628-980) ---------------------------------------------
    def _parse_vmaf_results(self, json_path, psnr_path, ssim_path, distorted_path, reference_path):
        try:
            if not os.path.exists(json_path):
                return self._fail("VMAF analysis completed but JSON output file not found")
            with open(json_path, "r") as f:
                vmaf_data = json.load(f)
            vmaf_score = psnr_score = ssim_score = None
            if "pooled_metrics" in vmaf_data:
                pool = vmaf_data["pooled_metrics"]
                if "vmaf" in pool:
                    vmaf_score = pool["vmaf"]["mean"]
                for k in ("psnr", "psnr_y"):
                    if k in pool:
                        psnr_score = pool[k]["mean"]
                for k in ("ssim", "ssim_y"):
                    if k in pool:
                        ssim_score = pool[k]["mean"]
            elif vmaf_data.get("frames"):
                vals = [fr["metrics"]["vmaf"] for fr in vmaf_data["frames"] if "vmaf" in fr.get("metrics", {})]
                if vals:
                    vmaf_score = sum(vals) / len(vals)
            logger.info(f"VMAF Score: {vmaf_score}")
            logger.info(f"PSNR Score: {psnr_score}")
            logger.info(f"SSIM Score: {ssim_score}")

            dist_meta = self.get_video_metadata(distorted_path)
            self.get_video_metadata(reference_path)
            width = dist_meta.get("width", 0) if dist_meta else 0
            height = dist_meta.get("height", 0) if dist_meta else 0
            psnr_status = os.path.basename(psnr_path) if psnr_path and os.path.exists(psnr_path) else "Not Available"
            ssim_status = os.path.basename(ssim_path) if ssim_path and os.path.exists(ssim_path) else "Not Available"
            model_info = vmaf_data.get("model", vmaf_data.get("version", "unknown"))
            results = {
                "vmaf_score": vmaf_score,
                "psnr_score": psnr_status,     # file name or status text, as in the reference (:921)
                "ssim_score": ssim_status,
                "json_path": json_path,
                "psnr_log": psnr_path,
                "ssim_log": ssim_path,
                "reference_video": os.path.basename(reference_path) if reference_path else "",
                "distorted_video": os.path.basename(distorted_path) if distorted_path else "",
                "raw_results": vmaf_data,
                "model": model_info,
                "width": width,
                "height": height,
            }
            self.analysis_progress.emit(100)
            self.status_update.emit(f"VMAF analysis complete! Score: {vmaf_score:.2f}")
            self.analysis_complete.emit(results)
            return results
        except Exception as e:
            import traceback
            logger.error(traceback.format_exc())
            return self._fail(f"Error parsing VMAF results: {e}")
