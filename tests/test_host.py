"""Host-side logic on CPU: model files, SVM, pooling, libvmaf-format JSON, FFmpeg stats files, Y4M I/O,
and the VMAFAnalyzer boundary (signals, files, result dict, errors, cancel) -- with the oracle-backed
stand-in engine injected where the HIP engine would be."""
import json
import os
import sys
import threading

import numpy as np
import pytest

from oracle.oracle import svr_predict_py
from pqa2_amd import model as M
from pqa2_amd import report, synth, yuvio
from tests.fake_engine import OracleEngine


def _write_pair(tmp_path, w=96, h=64, n=4, bpc=8):
    refs, diss = synth.make_clip(w, h, n, bpc, chroma=True)
    info = synth.clip_info(w, h, bpc)
    rp, dp = str(tmp_path / "ref.y4m"), str(tmp_path / "dist.y4m")
    yuvio.write_y4m(rp, refs, info)
    yuvio.write_y4m(dp, diss, info)
    return rp, dp, refs, diss


def test_models_load_unchanged_and_predict_matches_loop_restatement():
    rng = np.random.default_rng(1)
    for name in ("vmaf_v0.6.1", "vmaf_4k_v0.6.1", "vmaf_float_v0.6.1", "vmaf_v0.6.1neg"):
        m = M.load_model(name)
        with open(m.path) as f:
            md = json.load(f)["model_dict"]
        X = np.column_stack([rng.uniform(0.5, 1, 5), rng.uniform(0, 10, 5)] + [rng.uniform(0.2, 1, 5) for _ in range(4)])
        got = m.main.predict(X)
        want = [svr_predict_py(md, list(x)) for x in X]
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)
    assert abs(M.load_model("vmaf_v0.6.1").main.predict(np.array([[1, 0, 1, 1, 1, 1.0]]))[0] - 97.428043) < 1e-6
    neg = M.load_model("vmaf_v0.6.1neg")
    assert neg.vif_enhn_gain_limit == 1.0 and neg.adm_enhn_gain_limit == 1.0
    assert M.load_model("vmaf_v0.6.1").main.metric_keys[0] == "integer_adm2"
    assert M.load_model("vmaf_float_v0.6.1").main.metric_keys[0] == "adm2"
    # VIF padding follows the extractor the model names: integer_vif.c (reflect-101) vs vif_tools.c
    assert [M.load_model(n).vif_border for n in ("vmaf_v0.6.1", "vmaf_4k_v0.6.1", "vmaf_float_v0.6.1")] == [1, 1, 0]
    b = M.load_model("vmaf_b_v0.6.3")
    assert len(b.models) == 21
    with pytest.raises(FileNotFoundError):
        M.load_model("vmaf_v9")


def test_pooling_and_log_schema(tmp_path):
    v = np.array([80.0, 90.0, 100.0])
    p = M.pool(v)
    assert p["mean"] == 90.0 and p["min"] == 80.0 and p["max"] == 100.0
    assert abs(p["harmonic_mean"] - (1.0 / np.mean(1.0 / (v + 1.0)) - 1.0)) < 1e-12
    metrics = {"integer_motion2": np.array([0.0, 1.5, 1.5]), "vmaf": v}
    log = report.build_vmaf_log(metrics, 123.456, None, {"model": "vmaf_v0.6.1"})
    path = str(tmp_path / "x_vmaf.json")
    report.write_vmaf_json(path, log)
    text = open(path).read()
    assert '"vmaf": 80.000000' in text and '"frameNum": 0' in text          # libvmaf's %.6f form
    back = json.load(open(path))
    assert list(back.keys())[:1] == ["version"] and back["frames"][1]["metrics"]["integer_motion2"] == 1.5
    assert back["pooled_metrics"]["vmaf"]["mean"] == 90.0 and back["aggregate_metrics"] == {}


def test_ffmpeg_stats_file_lines():
    sse = np.array([[1000, 10, 0]], np.uint64)
    sizes = [(8, 8), (4, 4), (4, 4)]
    line = report.psnr_stats_lines(sse, sizes, 8)[0]
    mse_y, mse_u = 1000 / 64, 10 / 16
    mse_avg = (mse_y * 64 + mse_u * 16) / 96
    assert line.startswith(f"n:1 mse_avg:{mse_avg:.2f} mse_y:{mse_y:.2f} mse_u:{mse_u:.2f} mse_v:0.00 ")
    assert "psnr_v:inf" in line and f"psnr_y:{10 * np.log10(255 ** 2 / mse_y):.2f}" in line
    s = report.ssim_stats_lines(np.array([[0.9, 0.8, 1.0]]), sizes)[0]
    allv = (0.9 * 64 + 0.8 * 16 + 1.0 * 16) / 96
    assert s == f"n:1 Y:0.900000 U:0.800000 V:1.000000 All:{allv:f} ({-10 * np.log10(1 - allv):f})"


def test_y4m_roundtrip_and_raw(tmp_path):
    for bpc in (8, 10):
        rp, dp, refs, diss = _write_pair(tmp_path, 50, 34, 3, bpc)
        rd = yuvio.open_video(rp)
        assert len(rd) == 3 and (rd.info.width, rd.info.height, rd.info.bit_depth) == (50, 34, bpc)
        for i in range(3):
            for p in range(3):
                np.testing.assert_array_equal(rd.frame(i)[p], refs[i][p])
    raw = tmp_path / "clip_50x34.yuv"
    with open(raw, "wb") as f:
        for fr in refs[:2]:
            for p in fr:
                f.write(np.ascontiguousarray(p).tobytes())
    rr = yuvio.open_video(str(raw), bit_depth=10)
    assert len(rr) == 2
    np.testing.assert_array_equal(rr.frame(1)[0], refs[1][0])


def _analyzer(tmp_path):
    from pqa2_amd.vmaf_analyzer import VMAFAnalyzer
    a = VMAFAnalyzer()
    a._engine_factory = OracleEngine
    a.set_output_directory(str(tmp_path / "out"))
    os.makedirs(tmp_path / "out", exist_ok=True)
    a.set_test_name("T")
    ev = {"progress": [], "status": [], "errors": [], "complete": []}
    a.analysis_progress.connect(ev["progress"].append)
    a.status_update.connect(ev["status"].append)
    a.error_occurred.connect(ev["errors"].append)
    a.analysis_complete.connect(ev["complete"].append)
    return a, ev


def test_compressed_input_is_decoded_once_by_a_system_ffmpeg(tmp_path, monkeypatch):
    """Anything that is not .y4m/.yuv goes through `ffmpeg -f yuv4mpegpipe` (decode only, as the reference's
    ffmpeg child decodes, app/vmaf_analyzer.py:411-419).  No ffmpeg exists here, so a stand-in script that
    copies a prepared Y4M plays its part; the decode is cached per clip and the temp file removed on close."""
    refs, _ = synth.make_clip(64, 48, 2, 8)
    src = str(tmp_path / "a.y4m")
    yuvio.write_y4m(src, refs, synth.clip_info(64, 48))
    mp4 = str(tmp_path / "a.mp4")
    with open(mp4, "wb") as f:
        f.write(b"\x00\x00\x00\x18ftypmp42")
    with pytest.raises(RuntimeError, match="no ffmpeg on PATH"):
        monkeypatch.setenv("PATH", str(tmp_path / "empty"))
        yuvio.open_video(mp4)
    bindir = tmp_path / "bin"
    bindir.mkdir()
    fake = bindir / "ffmpeg"
    fake.write_text(f'#!/bin/sh\nfor a; do last="$a"; done\ncp {src} "$last"\n')
    fake.chmod(0o755)
    monkeypatch.setenv("PATH", str(bindir) + os.pathsep + "/usr/bin" + os.pathsep + "/bin")
    r1 = yuvio.open_video(mp4)
    assert yuvio.open_video(mp4) is r1 and len(r1) == 2 and np.array_equal(r1.frame(1)[0], refs[1][0])
    tmp = r1._tempfile
    assert os.path.exists(tmp)
    r1.close()
    assert not os.path.exists(tmp)
    assert yuvio.open_video(mp4) is not r1
    yuvio._drop_decodes()


def test_analyzer_boundary_contract(tmp_path, oracle32):
    rp, dp, refs, diss = _write_pair(tmp_path)
    a, ev = _analyzer(tmp_path)
    res = a.analyze_videos(rp, dp, "vmaf_v0.6.1", duration=5)
    assert ev["errors"] == [] and res is not None and ev["complete"] == [res]
    # result dict keys = the reference's final dict (app/vmaf_analyzer.py:919-932)
    assert set(res) == {"distorted_video", "height", "json_path", "model", "psnr_log", "psnr_score", "raw_results",
                        "reference_video", "ssim_log", "ssim_score", "vmaf_score", "width"}
    assert (res["width"], res["height"]) == (96, 64)
    assert res["reference_video"] == "ref.y4m" and res["distorted_video"] == "dist.y4m"
    assert os.path.basename(res["json_path"]).startswith("T_") and res["json_path"].endswith("_vmaf.json")
    assert res["psnr_score"] == os.path.basename(res["psnr_log"]) and res["ssim_score"].endswith("_ssim.txt")
    assert ev["progress"][0] == 0 and ev["progress"][-1] == 100
    assert ev["status"][0] == "Analyzing videos with model: vmaf_v0.6.1" and ev["status"][-1].startswith("VMAF analysis complete! Score: ")
    raw = res["raw_results"]
    assert res["vmaf_score"] == raw["pooled_metrics"]["vmaf"]["mean"]
    fr = raw["frames"]
    assert [f["frameNum"] for f in fr] == [0, 1, 2, 3]
    for k in ("integer_adm2", "integer_motion2", "integer_vif_scale0", "integer_vif_scale3", "vmaf", "psnr_y", "ssim"):
        assert k in fr[0]["metrics"]
    # scores are what the oracle + model give for these files
    rec = np.zeros((4, 24)); rec[:, :17] = oracle32.clip_features([r[0] for r in refs], [d[0] for d in diss], 8, vif_border101=True)
    want = M.score_frames(M.load_model("vmaf_v0.6.1"), M.metrics_from_records(rec, 96, 64, "integer_"))["vmaf"]
    np.testing.assert_allclose([f["metrics"]["vmaf"] for f in fr], want, atol=1e-6)
    # stats files hold one FFmpeg-format line per frame
    lines = open(res["psnr_log"]).read().strip().split("\n")
    assert len(lines) == 4 and lines[0].startswith("n:1 mse_avg:") and "psnr_y:" in lines[3]
    assert open(res["ssim_log"]).read().startswith("n:1 Y:")


def test_analyzer_errors_and_options(tmp_path):
    rp, dp, _, _ = _write_pair(tmp_path)
    a, ev = _analyzer(tmp_path)
    assert a.analyze_videos(str(tmp_path / "missing.y4m"), dp) is None
    assert ev["errors"] == [f"Reference video not found: {tmp_path / 'missing.y4m'}"]
    assert a.analyze(rp, str(tmp_path / "nope.y4m")) is None and "Distorted video not found" in ev["errors"][-1]
    assert a.analyze_videos(rp, dp, "vmaf_does_not_exist") is None and "unknown VMAF model" in ev["errors"][-1]
    # size mismatch is an error signal, not an exception
    refs, _ = synth.make_clip(64, 48, 2, 8, chroma=True)
    other = str(tmp_path / "small.y4m")
    yuvio.write_y4m(other, refs, synth.clip_info(64, 48))
    assert a.analyze_videos(rp, other) is None and "96x64" in ev["errors"][-1]

    class OM:
        def get_setting(self, k):
            return {"threads": 8, "feature_subsample": 2, "pool_method": "harmonic_mean", "psnr_enabled": False,
                    "ssim_enabled": False}
    a.set_options_from_manager(OM())
    assert (a.threads, a.feature_subsample, a.pool_method, a.psnr_enabled, a.ssim_enabled) == (8, 2, "harmonic_mean", False, False)
    res = a.analyze_videos(rp, dp)
    assert res["psnr_score"] == "Not Available" and res["psnr_log"] is None
    assert [f["frameNum"] for f in res["raw_results"]["frames"]] == [0, 2]       # n_subsample=2
    assert "psnr_y" not in res["raw_results"]["frames"][0]["metrics"]


def test_analyzer_terminate_from_another_thread(tmp_path):
    rp, dp, _, _ = _write_pair(tmp_path, n=6)
    a, ev = _analyzer(tmp_path)

    class SlowEngine(OracleEngine):
        def submit(self, *args):
            if len(self.rec) == 1:
                threading.Thread(target=a.terminate_analysis).start()
                import time
                time.sleep(0.2)
            return super().submit(*args)
    a._engine_factory = SlowEngine
    assert a.analyze_videos(rp, dp) is None
    assert ev["errors"] == ["VMAF analysis was terminated by user"] and ev["complete"] == []
    a._engine_factory = OracleEngine
    assert a.analyze_videos(rp, dp) is not None      # a new analysis re-arms, like the reference (:254)


def test_bookend_rules_on_exact_stats():
    from pqa2_amd import bookend
    rng = np.random.default_rng(3)
    frames = [np.full((40, 50), 250, np.uint8), rng.integers(0, 256, (40, 50), dtype=np.uint8),
              np.clip(rng.normal(235, 30, (40, 50)), 0, 255).astype(np.uint8)]
    thr = 200
    stats = np.array([[int(f.sum()), int((f.astype(np.int64) ** 2).sum()), int((f > thr).sum())] for f in frames], np.uint64)
    mean, std, ratio = bookend.brightness_from_stats(stats, 2000)
    np.testing.assert_allclose(mean, [f.mean() for f in frames], rtol=1e-14)
    np.testing.assert_allclose(std, [f.std() for f in frames], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(ratio, [(f > thr).mean() for f in frames])
    # the reference's rules, evaluated frame by frame the way its cv2 loop does
    def ref_initial(f, wt, st, idx):
        b, sd = np.mean(f), np.std(f)
        return b > wt if idx < 2 else (b > wt and sd < st)

    def ref_refined(f, t, st):
        b, sd = np.mean(f), np.std(f)
        if sd < st * 1.2:
            return b > t * 0.95
        if b > t:
            return True
        return b > t * 0.9 and np.sum(f > t) / f.size > 0.7
    for idx in (0, 2):
        assert bookend.is_white_initial(mean, std, 200, 30, idx).tolist() == [ref_initial(f, 200, 30, idx) for f in frames]
    for t in (200, 220, 240):
        st2 = np.array([[int(f.sum()), int((f.astype(np.int64) ** 2).sum()), int((f > t).sum())] for f in frames], np.uint64)
        m2, s2, r2 = bookend.brightness_from_stats(st2, 2000)
        assert bookend.is_white_refined(m2, s2, r2, t, 30).tolist() == [bool(ref_refined(f, t, 30)) for f in frames]
    assert bookend.starts_with_bookend(ratio) and not bookend.starts_with_bookend(ratio[1:2])


def test_bootstrap_collection_and_pooling_keys():
    """vmaf_b_v0.6.3: 21 nu-SVR models; entry 0 scores `vmaf`, entries 1..20 give the bagging statistics."""
    m = M.load_model("vmaf_b_v0.6.3")
    rng = np.random.default_rng(2)
    n = 6
    rec = np.zeros((n, 24))
    rec[:, 0:4] = rng.uniform(0.5, 1.0, (n, 4)); rec[:, 4:8] = 1.0
    rec[:, 8:12] = rng.uniform(0.8, 1.0, (n, 4)); rec[:, 12:16] = 1.0
    rec[:, 16] = rng.uniform(0, 8, n); rec[0, 16] = 0
    out = M.score_frames(m, M.metrics_from_records(rec, 1920, 1080, "integer_"))
    for k in ("vmaf", "vmaf_bagging", "vmaf_stddev", "vmaf_ci_p95_lo", "vmaf_ci_p95_hi"):
        assert k in out and out[k].shape == (n,)
    assert np.all(out["vmaf_ci_p95_lo"] <= out["vmaf_bagging"]) and np.all(out["vmaf_bagging"] <= out["vmaf_ci_p95_hi"])
    assert np.all(out["vmaf_stddev"] > 0) and np.all(np.abs(out["vmaf"] - out["vmaf_bagging"]) < 5.0)
    single = M.load_model("vmaf_b_v0.6.3").models[0].predict(
        np.column_stack([out["integer_adm2"], out["integer_motion2"]] + [out[f"integer_vif_scale{s}"] for s in range(4)]))
    np.testing.assert_array_equal(single, out["vmaf"])
    log = report.build_vmaf_log(out, 30.0)
    assert set(log["pooled_metrics"]["vmaf_bagging"]) == {"min", "max", "mean", "harmonic_mean"}


def test_committed_kernel_counters_match_the_kernel_source():
    """bench.py's `roofline.traffic` / `roofline.valu` come from committed rocprofv3 --pmc passes (profiles/
    kernel_counters.json) and are dropped when the dominant kernel's source has changed since.  This guards the repo
    state: after editing csrc/vif.hip (or pqa_device.h / kernels.h) re-run `tools/profile_round.sh <tag> 2160p` on the
    GPU box and copy gpurun_out/prof_<tag>/* into profiles/."""
    import json
    import os
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, "profiles", "kernel_counters.json")))
    for wl, b_alg in (("2160p", 2 * 3840 * 2160), ("2160p10", 4 * 3840 * 2160)):   # BASELINE configs 3 and 5
        e = d[wl]["vif_stat_s0"]
        assert e["src_hash"] == bench.kernel_source_hash(), f"profiles/kernel_counters.json[{wl}] is stale: re-run tools/profile_round.sh"
        assert e["hbm_bytes_per_frame"] > b_alg and e["valu_wave_insts_per_frame"] > 0
        assert os.path.exists(os.path.join(root, e["traffic_source"].split(" ")[0]))
        adm = d[wl]["adm_s0"]   # bench.py's `roofline_adm` (the memory-side half of the frame since round 4)
        assert adm["src_hash"] == bench.kernel_source_hash("adm"), f"profiles/kernel_counters.json[{wl}].adm_s0 is stale: re-run tools/profile_round.sh"
        assert adm["hbm_bytes_per_frame"] > b_alg


_FAKE_FFMPEG = r'''#!/usr/bin/env python3
import json, re, sys
a = sys.argv[1:]
if "-filters" in a:
    print("Filters:\n ... lenscorrection      V->V       Rectify the image.\n%s ... libplacebo         N->V       x\n" % LIBVMAF_LINE)
    sys.exit(0)
if "-version" in a:
    print("ffmpeg version 7.1.1-fake Copyright (c) the FFmpeg developers")
    sys.exit(0)
opt = a[a.index("-lavfi") + 1]
assert a[a.index("-i") + 1].endswith("dis.y4m") and "model=version=vmaf_v0.6.1" in opt and "n_subsample=1" in opt, a
log = re.search(r"log_path=([^:]+)", opt).group(1)
json.dump({"frames": [{"frameNum": i, "metrics": {"vmaf": 80.0 + i}} for i in range(5)]}, open(log, "w"))
'''


def _fake_ffmpeg(tmp_path, monkeypatch, with_libvmaf):
    import stat
    d = tmp_path / "bin"
    d.mkdir()
    line = " ... libvmaf            VV->V      Calculate the VMAF between two video streams.\\n" if with_libvmaf else ""
    f = d / "ffmpeg"
    f.write_text(_FAKE_FFMPEG.replace("LIBVMAF_LINE", repr(line)))
    f.chmod(f.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(d) + os.pathsep + os.environ.get("PATH", ""))


def test_bench_probes_for_libvmaf_and_runs_the_reference_command_when_it_is_there(tmp_path, monkeypatch):
    """BASELINE.md section 3 / SURVEY 8(d): bench.py looks for an ffmpeg with the libvmaf filter instead of assuming there is
    none, and when it finds one it runs the reference's argv (app/vmaf_analyzer.py:411-419: distorted first, model=version=,
    n_threads, n_subsample) and diffs the log's per-frame vmaf against ours.  A stand-in `ffmpeg` on PATH plays the binary."""
    import torch
    import bench
    _fake_ffmpeg(tmp_path, monkeypatch, with_libvmaf=True)
    probe = bench.probe_libvmaf()
    assert probe["ffmpeg_has_libvmaf"] and probe["ffmpeg"].endswith("ffmpeg") and "7.1.1" in probe["ffmpeg_version"]
    w, h, n = 32, 16, 5
    ref = [torch.zeros((n, h, w), dtype=torch.uint8)]
    dis = [torch.ones((n, h, w), dtype=torch.uint8)]
    ours = np.array([80.0, 81.0, 82.004, 83.0, 84.0])
    leg = bench._libvmaf_leg(probe, ref, dis, 0, 8, w, h, "vmaf_v0.6.1", ours, n)
    assert "error" not in leg, leg
    t4 = leg["n_threads_4"]
    assert t4["returncode"] == 0 and t4["fps"] > 0 and abs(t4["max_abs_vmaf_diff_per_frame"] - 0.004) < 1e-9
    assert abs(t4["pooled_abs_vmaf_diff"] - 0.0008) < 1e-9


def test_bench_probe_says_no_when_the_filter_is_missing(tmp_path, monkeypatch):
    import bench
    _fake_ffmpeg(tmp_path, monkeypatch, with_libvmaf=False)
    probe = bench.probe_libvmaf()
    assert probe["ffmpeg"] and not probe["ffmpeg_has_libvmaf"]


def test_registered_child_jobs_are_stopped_when_the_process_goes(tmp_path):
    """the torchrun child runs in its own session (so that a stop reaches its whole process group), which also
    takes it out of reach of a Ctrl-C / SIGTERM aimed at the analyzer's process.  Live children are therefore registered
    and an atexit / signal hook stops their groups; here the hook body is driven directly on a stand-in child."""
    import subprocess
    import sys
    from pqa2_amd import vmaf_analyzer as va
    va._install_child_hooks()
    va._install_child_hooks()                      # idempotent
    child = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(60)"], start_new_session=True)
    va._LIVE_CHILDREN.add(child)
    va._stop_live_children()
    assert child.poll() is not None and child not in va._LIVE_CHILDREN


def _march_table():
    import ctypes as C
    from pqa2_amd import _native as N
    lib = N.load()
    need = -lib.pqa_debug_vif_march_table(None, 0)
    buf = np.zeros(need, np.uint16)
    n = lib.pqa_debug_vif_march_table(buf.ctypes.data, need)
    assert n == 24, n                                              # 0 would mean: a tap does not split exactly into f16 pieces
    return buf.view(np.float16).reshape(n, 64, 8).astype(np.float64)   # [fragment][lane][element], exact in f64


def _as_B(frag):
    """A fragment as the B operand of v_mfma_f32_16x16x32_f16: lane l supplies B[k = 8 (l >> 4) + j][n = l & 15]."""
    B = np.zeros((32, 16))
    for l in range(64):
        B[8 * (l >> 4):8 * (l >> 4) + 8, l & 15] = frag[l]
    return B


def _gauss(n):
    d = np.arange(n) - n // 2
    v = np.exp(-0.5 * d * d / (n / 5.0) ** 2)
    return (v / v.sum()).astype(np.float32).astype(np.float64)


def test_march_tap_tables_replayed_in_numpy_are_the_separable_gaussian():
    """csrc/vif_march.hip runs both passes of the 17-tap filter as banded matrices on the matrix cores; the second pass takes
    its operand straight from the first pass's accumulator registers, which only works because the K order of its tap matrix
    is DEFINED by that register layout.  Replay the whole dataflow on the CPU with the very table the kernel loads
    (pqa_debug_vif_march_table) -- operand and accumulator lane layouts of v_mfma_f32_16x16x32_f16 spelled out -- and compare
    with a direct convolution: both passes, both block orders (F_V / F_W), the pieces -- all three of a pass-1 tap sum to the
    f32 tap exactly, the two the kernel multiplies (PQA_MARCH_TAP_PIECES = 2 since round 4) and the two of a pass-2 tap to 22
    bits of it -- and the next scale's 9-tap even-row / even-column planes for ref and dis."""
    T = _march_table()
    F_HI, F_LO, F_DR, F_DD, F_V, F_VD, F_W, F_WD, F_L9, F_L8 = 0, 3, 6, 9, 12, 14, 16, 18, 20, 22
    c17, c9 = _gauss(17), _gauss(9)
    rng = np.random.default_rng(5)

    # ---- pass 1: D1[row m][slot n] = sum_k X[m][k] * B[k][n], X = 16 rows x 32 window columns (output column n at window n + 8)
    X = rng.integers(0, 256, (16, 32)).astype(np.float64)
    want = np.stack([[np.dot(c17, X[m, n:n + 17]) for n in range(16)] for m in range(16)])
    lo = sum(_as_B(T[F_LO + p]) for p in range(3))                     # c * 2^11, three exact pieces
    hi = sum(_as_B(T[F_HI + p]) for p in range(3))                     # c * 2^19
    l9 = sum(_as_B(T[F_L9 + p]) for p in range(2))                     # c * 2^9, two pieces (22 bits of the tap)
    assert np.array_equal(X @ lo, want * 2048.0) and np.array_equal(X @ hi, want * 524288.0)
    assert np.abs(X @ l9 / 512.0 - want).max() < 2e-6 * np.abs(want).max()
    l8 = sum(_as_B(T[F_L8 + p]) for p in range(2))                     # c * 2^8, two pieces: the 12-bit low digits
    assert np.abs(X @ l8 / 256.0 - want).max() < 2e-6 * np.abs(want).max()
    # what the kernel multiplies: the first two pieces -- the SAME 22 bits of the tap in both tables (pieces scale with the
    # power of two), a filter whose taps differ from the f32 ones by less than 2^-22 each
    lo2 = sum(_as_B(T[F_LO + p]) for p in range(2))
    hi2 = sum(_as_B(T[F_HI + p]) for p in range(2))
    assert np.array_equal(hi2, lo2 * 256.0)
    assert np.abs(lo2 - lo).max() <= 2.0 ** -22 * np.abs(lo).max() and np.abs(X @ lo2 / 2048.0 - want).max() < 3e-7 * np.abs(want).max()
    # next scale's input: slots 0..7 = even columns of the plane filtered through F_DR, 8..15 = of the one through F_DD
    dr = sum(_as_B(T[F_DR + p]) for p in range(3))
    dd = sum(_as_B(T[F_DD + p]) for p in range(3))
    dr2, dd2 = sum(_as_B(T[F_DR + p]) for p in range(2)), sum(_as_B(T[F_DD + p]) for p in range(2))
    assert np.abs(dr2 - dr).max() <= 2.0 ** -22 * np.abs(dr).max() and np.abs(dd2 - dd).max() <= 2.0 ** -22 * np.abs(dd).max()
    Y = rng.integers(0, 256, (16, 32)).astype(np.float64)
    got = X @ dr + Y @ dd
    for e in range(8):
        assert np.allclose(got[:, e], [np.dot(c9, X[m, 2 * e + 4:2 * e + 13]) * 262144.0 for m in range(16)], rtol=0, atol=1e-6)
        assert np.allclose(got[:, 8 + e], [np.dot(c9, Y[m, 2 * e + 4:2 * e + 13]) * 262144.0 for m in range(16)], rtol=0, atol=1e-6)

    # ---- pass 2: the accumulator of pass 1 leaves lane l = (n = l & 15, kq = l >> 4) with rows 4 kq + i of column n.  The A
    # operand of the next MFMA takes from that same lane row m = l & 15 (the COLUMN) and K elements 8 kq + j: j < 4 from the
    # block in dwords {0, 1} of the operand vector, j >= 4 from the block in dwords {2, 3}.
    P, Cb = rng.standard_normal((16, 16)) * 50, rng.standard_normal((16, 16)) * 50    # previous / current block [row][col]

    def operand(first, second):                    # -> A[col][k]
        A = np.zeros((16, 32))
        for col in range(16):
            for kq in range(4):
                A[col, 8 * kq:8 * kq + 4] = first[4 * kq:4 * kq + 4, col]
                A[col, 8 * kq + 4:8 * kq + 8] = second[4 * kq:4 * kq + 4, col]
        return A
    W = np.vstack([P, Cb])                                              # the 32-row window; output row n at window row n + 8
    want2 = np.stack([[np.dot(c17, W[n:n + 17, col]) for n in range(16)] for col in range(16)]) * 256.0
    v = _as_B(T[F_V]) + _as_B(T[F_V + 1])                               # older block first
    w_ = _as_B(T[F_W]) + _as_B(T[F_W + 1])                              # newer block first
    assert np.abs(operand(P, Cb) @ v - want2).max() < 3e-7 * np.abs(want2).max()      # two f16 pieces: 22 bits of each tap
    assert np.abs(operand(Cb, P) @ w_ - want2).max() < 3e-7 * np.abs(want2).max()
    vd = _as_B(T[F_VD]) + _as_B(T[F_VD + 1])
    wd = _as_B(T[F_WD]) + _as_B(T[F_WD + 1])
    want9 = np.stack([[np.dot(c9, W[2 * e + 4:2 * e + 13, col]) for e in range(8)] for col in range(16)]) * 256.0
    for got9 in (operand(P, Cb) @ vd, operand(Cb, P) @ wd):
        assert np.abs(got9[:, :8] - want9).max() < 3e-7 * np.abs(want9).max() and np.all(got9[:, 8:] == 0.0)


# ---- pqa2_amd.score: a worker must die with its parent, and only then --------------------------------------
def test_worker_parent_check_does_not_mistake_a_live_pid1_parent_for_a_dead_one():
    """`getppid() == 1` is not 'the parent is gone': a container entrypoint or an init-less shell IS pid 1.  The check
    compares the parent now with the parent at start-up instead; a sub-reaper's pid (not 1) counts as re-parented too."""
    import signal
    from pqa2_amd import score
    sent = []
    kill = lambda pid, sig: sent.append((pid, sig))
    score._die_with_parent(1, getppid=lambda: 1, kill=kill)            # live parent really is pid 1: nothing happens
    assert sent == []
    score._die_with_parent(4321, getppid=lambda: 4321, kill=kill)      # ordinary live parent
    assert sent == []
    score._die_with_parent(4321, getppid=lambda: 1, kill=kill)         # orphaned to init before the request took effect
    score._die_with_parent(4321, getppid=lambda: 777, kill=kill)       # orphaned to a sub-reaper
    assert sent == [(os.getpid(), signal.SIGTERM)] * 2


_REAPER = r'''
import ctypes, os, subprocess, sys, time
ctypes.CDLL(None).prctl(36, 1, 0, 0, 0)          # PR_SET_CHILD_SUBREAPER: orphans below come to this process, not to pid 1
root, orphan, flag = sys.argv[1], sys.argv[2] == "orphan", sys.argv[3]
# the worker notes its parent, says so (flag file), waits until that parent is gone (orphan case) and only then asks
worker = ("import os, sys, time; sys.path.insert(0, %r); from pqa2_amd import score; exp = score._expected_parent(); "
          "open(%r, 'w').close(); t = time.time(); "
          "[time.sleep(0.02) for _ in iter(lambda: %r and os.getppid() == exp and time.time() - t < 20, False)]; "
          "score._die_with_parent(exp); time.sleep(0.5); print('survived', flush=True)") % (root, flag, orphan)
if orphan:   # a middle process starts the worker and exits once the worker has noted it: the worker is re-parented to this reaper
    mid = subprocess.Popen([sys.executable, "-c", "import os, subprocess, sys, time; subprocess.Popen([sys.executable, '-c', sys.argv[1]])\n"
                            "while not os.path.exists(sys.argv[2]): time.sleep(0.02)", worker, flag])
    mid.wait()
    pid, status = os.wait()                      # the re-parented worker
    print("worker signal", os.WTERMSIG(status) if os.WIFSIGNALED(status) else 0, flush=True)
else:        # the worker's parent (this process) stays: it must survive
    w = subprocess.run([sys.executable, "-c", worker], capture_output=True, text=True)
    print("worker rc", w.returncode, w.stdout.strip(), flush=True)
'''


@pytest.mark.skipif(not sys.platform.startswith("linux"), reason="prctl")
def test_worker_dies_when_reparented_to_a_subreaper_and_survives_a_live_parent(tmp_path):
    """The real thing, no mocks: under a sub-reaper an orphan's parent is the reaper (pid != 1) -- the old `ppid == 1`
    test missed it; and a worker whose parent is alive must not kill itself."""
    import signal
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "PQA_PARENT_PID"}
    r = subprocess.run([sys.executable, "-c", _REAPER, root, "orphan", str(tmp_path / "noted1")], capture_output=True, text=True, timeout=60, env=env)
    assert f"worker signal {int(signal.SIGTERM)}" in r.stdout, (r.stdout, r.stderr[-500:])
    r = subprocess.run([sys.executable, "-c", _REAPER, root, "alive", str(tmp_path / "noted2")], capture_output=True, text=True, timeout=60, env=env)
    assert "worker rc 0 survived" in r.stdout, (r.stdout, r.stderr[-500:])


# ---- round 4 host trims: the Y4M frame index, run strides, the metadata cache -------------------------------------------
def test_y4m_frame_index_fast_path_equals_the_scan_and_falls_back_on_frame_parameters(tmp_path):
    """A plain Y4M file (every frame header the bare "FRAME\\n") is indexed from three headers; a file whose FRAME headers
    carry parameters of varying length must still be scanned header by header.  Same offsets as a byte-level scan either way,
    and run_stride() -- what pqa_submit_fd_run is fed -- is only given where the frames really are equally spaced."""
    w, h, n = 32, 18, 7
    info = synth.clip_info(w, h, 8)
    refs, _ = synth.make_clip(w, h, n, 8, chroma=True)
    plain = str(tmp_path / "plain.y4m")
    yuvio.write_y4m(plain, refs, info)
    raw = open(plain, "rb").read()
    head_end = raw.index(b"\n") + 1
    fb = info.frame_bytes

    def scan(buf):
        offs, off = [], buf.index(b"\n") + 1
        while off + 6 <= len(buf) and buf[off:off + 5] == b"FRAME":
            start = buf.index(b"\n", off) + 1
            if start + fb > len(buf):
                break
            offs.append(start)
            off = start + fb
        return offs

    rd = yuvio.open_video(plain)
    assert len(rd) == n and [rd.plane_offsets(i)[0] for i in range(n)] == scan(raw)
    assert rd.run_stride(0, n) == fb + 6 and rd.run_stride(2, 1) == 0
    # the same frames with a parameter on every other FRAME header (legal Y4M: "FRAME Ip\n")
    odd = bytearray(raw[:head_end])
    for i in range(n):
        odd += (b"FRAME Ip\n" if i % 2 else b"FRAME\n") + raw[scan(raw)[i]:scan(raw)[i] + fb]
    path = str(tmp_path / "params.y4m")
    open(path, "wb").write(bytes(odd))
    rd2 = yuvio.open_video(path)
    assert len(rd2) == n and [rd2.plane_offsets(i)[0] for i in range(n)] == scan(bytes(odd))
    assert rd2.run_stride(0, 3) is None and rd2.run_stride(0, 1) == 0
    for i in range(n):
        assert np.array_equal(np.asarray(rd2.frame(i)[0]), refs[i][0])
    # a truncated plain file: the last, incomplete frame is not counted (fast path and scan agree)
    cut = str(tmp_path / "cut.y4m")
    open(cut, "wb").write(raw[:scan(raw)[n - 1] + fb // 2])
    assert len(yuvio.open_video(cut)) == n - 1


def test_metadata_cache_follows_the_file(tmp_path):
    """get_video_metadata keeps its answer per (path, size, mtime): a rewritten file must be looked at again."""
    import os
    import time
    from pqa2_amd.vmaf_analyzer import VMAFAnalyzer
    an = VMAFAnalyzer()
    p = str(tmp_path / "clip.y4m")
    refs, _ = synth.make_clip(32, 18, 5, 8, chroma=True)
    yuvio.write_y4m(p, refs, synth.clip_info(32, 18, 8))
    a = an.get_video_metadata(p)
    assert a["nb_frames"] == 5 and an.get_video_metadata(p) == a
    refs2, _ = synth.make_clip(48, 26, 3, 8, chroma=True)
    yuvio.write_y4m(p, refs2, synth.clip_info(48, 26, 8))
    os.utime(p, ns=(time.time_ns(), time.time_ns() + 5_000_000_000))
    b = an.get_video_metadata(p)
    assert (b["width"], b["height"], b["nb_frames"]) == (48, 26, 3)
    assert an.get_video_metadata(str(tmp_path / "missing.y4m")) is None
