"""FeatureEngine: thin Python owner of one pqa_ctx (one per GPU).

Frames go in as numpy planes (host path, `submit`) or as a torch CUDA tensor that already holds a
whole clip in HBM (`submit_resident`); per-frame feature records come back as a [n, 24] float64
array.  All arithmetic happens in the HIP kernels behind the C ABI (include/pqa_vmaf.h).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _native as N


_PARKED = {}   # at most one entry: configuration bytes -> (library, context) parked by FeatureEngine.release()


def _take_parked(key):
    hit = _PARKED.pop(key, None)
    return hit[1] if hit else None


def clear_parked():
    """Destroy the context FeatureEngine.release() parked (also runs at interpreter exit)."""
    while _PARKED:
        _, (lib, ctx) = _PARKED.popitem()
        try:
            lib.pqa_destroy(ctx)
        except Exception:
            pass


def _park(lib, key, ctx):
    clear_parked()            # one per process: whatever was parked before goes
    if not _PARKED_HOOK:
        import atexit
        atexit.register(clear_parked)
        _PARKED_HOOK.append(True)
    _PARKED[key] = (lib, C.c_void_p(ctx.value))


_PARKED_HOOK = []


class FeatureEngine:
    def __init__(self, width: int, height: int, bit_depth: int = 8, n_planes: int = 1,
                 chroma_shift=(1, 1), features: int = N.FEAT_VMAF, device: int = 0, max_batch: int = 0,
                 result_capacity: int = 16384, n_subsample: int = 1,
                 vif_enhn_gain_limit: float = 100.0, adm_enhn_gain_limit: float = 100.0,
                 vif_border: int = N.VIF_BORDER_FLOAT, fixed_point: int = 0, reuse: bool = False):
        self.lib = N.load()
        cfg = N.PqaConfig()
        self.lib.pqa_config_init(C.byref(cfg), width, height)
        cfg.device = device
        cfg.bit_depth = bit_depth
        cfg.n_planes = n_planes
        cfg.chroma_hshift, cfg.chroma_vshift = chroma_shift
        cfg.features = features
        cfg.max_batch = max_batch
        cfg.result_capacity = result_capacity
        cfg.n_subsample = n_subsample
        cfg.vif_enhn_gain_limit = vif_enhn_gain_limit
        cfg.adm_enhn_gain_limit = adm_enhn_gain_limit
        cfg.vif_border = vif_border
        cfg.fixed_point = int(fixed_point)
        self.cfg = cfg
        self.width, self.height, self.bit_depth, self.n_planes = width, height, bit_depth, n_planes
        self.dtype = np.uint8 if bit_depth <= 8 else np.dtype("<u2")
        self._ctx = C.c_void_p()
        self.luma_gray = N.GRAY_LUMA   # mirror of the context's sticky gray mode (pqa_set_luma_gray)
        self._key = bytes(cfg)         # every field of the configuration: what a parked context must match
        parked = _take_parked(self._key) if reuse else None
        if parked is not None:         # a context release() parked: same configuration, back to its initial state
            self._ctx = parked
            if self.lib.pqa_reset(self._ctx) == N.PQA_OK and self.lib.pqa_set_luma_gray(self._ctx, N.GRAY_LUMA) == N.PQA_OK:
                return
            self.lib.pqa_destroy(self._ctx)
            self._ctx = C.c_void_p()
        rc = self.lib.pqa_create(C.byref(cfg), C.byref(self._ctx))
        if rc != N.PQA_OK:
            raise N.PqaError(rc, (self.lib.pqa_last_error(None) or b"").decode())

    # -- lifetime ----------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self.lib.pqa_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def release(self):
        """Like close(), but a healthy context is PARKED for the next FeatureEngine(..., reuse=True) of the very same
        configuration instead of being destroyed (the reference's caller makes a fresh analyzer per run,
        app/ui/tabs/analysis_tab.py:588: creating and destroying a 2160p context costs ~7 ms of a 180 ms analysis).  One
        context per process is kept; it is destroyed when another one is parked, by clear_parked(), or at interpreter exit.
        PQA_CONTEXT_CACHE=0 turns parking off.  Call only after a successful run: a context that reported an error is closed."""
        if getattr(self, "_ctx", None) is None or not self._ctx.value:
            return
        if os.environ.get("PQA_CONTEXT_CACHE", "1") == "0":
            return self.close()
        _park(self.lib, self._key, self._ctx)
        self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int):
        if rc == N.PQA_OK:
            return
        msg = (self.lib.pqa_last_error(self._ctx) or b"").decode()
        raise (N.PqaCancelled if rc == N.PQA_ECANCELLED else N.PqaError)(rc, msg)

    # -- host path ---------------------------------------------------------------------------
    def submit(self, index: int, ref_planes, dis_planes):
        """ref_planes / dis_planes: sequences of 2-D numpy arrays (Y[,U,V]) of this engine's dtype."""
        P, S = C.c_void_p * 3, C.c_int64 * 3
        rp, rs, dp, ds = P(), S(), P(), S()
        keep = []
        for p in range(self.n_planes):
            for arr, pp, ss in ((ref_planes[p], rp, rs), (dis_planes[p], dp, ds)):
                a = np.asarray(arr)
                if a.dtype != self.dtype or a.strides[1] != a.itemsize:
                    a = np.ascontiguousarray(a, dtype=self.dtype)
                keep.append(a)
                pp[p] = a.ctypes.data
                ss[p] = a.strides[0]
        self._check(self.lib.pqa_submit(self._ctx, index, C.byref(rp), C.byref(rs), C.byref(dp), C.byref(ds)))

    def submit_file(self, index: int, ref_fd: int, ref_offsets, dis_fd: int, dis_offsets):
        """A frame pair lying in two files as packed planes (pqa_submit_fd): *_offsets = byte offset of each plane."""
        S = C.c_int64 * 3
        ro, do = S(), S()
        for p in range(self.n_planes):
            ro[p], do[p] = int(ref_offsets[p]), int(dis_offsets[p])
        self._check(self.lib.pqa_submit_fd(self._ctx, index, int(ref_fd), C.byref(ro), int(dis_fd), C.byref(do)))

    def submit_file_run(self, first_index: int, n_frames: int, ref_fd: int, ref_offsets, ref_frame_stride: int,
                        dis_fd: int, dis_offsets, dis_frame_stride: int):
        """n_frames consecutive frame pairs of two files (pqa_submit_fd_run): frame first_index + k has plane p at
        *_offsets[p] + k * *_frame_stride.  The library pipelines reading and upload inside the call."""
        ro = (C.c_int64 * 3)(*[int(x) for x in list(ref_offsets) + [0] * (3 - len(ref_offsets))])
        do = (C.c_int64 * 3)(*[int(x) for x in list(dis_offsets) + [0] * (3 - len(dis_offsets))])
        self._check(self.lib.pqa_submit_fd_run(self._ctx, int(first_index), int(n_frames), int(ref_fd), C.byref(ro),
                                               int(ref_frame_stride), int(dis_fd), C.byref(do), int(dis_frame_stride)))

    def set_motion_halo(self, prev_ref_luma: np.ndarray | None):
        if prev_ref_luma is None:
            self._check(self.lib.pqa_set_motion_halo(self._ctx, None, 0))
            return
        a = np.ascontiguousarray(prev_ref_luma, dtype=self.dtype)
        self._check(self.lib.pqa_set_motion_halo(self._ctx, a.ctypes.data, a.strides[0]))

    # -- device-resident path ------------------------------------------------------------------
    def submit_resident(self, first_index: int, n_frames: int, ref_ptrs, dis_ptrs, row_pitch, frame_pitch,
                        prev_ref_luma_ptr: int = 0, prev_row_pitch: int = 0):
        """ref_ptrs/dis_ptrs: device addresses of frame 0's planes; pitches in bytes per plane."""
        r, d = N.PqaDeviceClip(), N.PqaDeviceClip()
        for p in range(self.n_planes):
            r.plane[p], d.plane[p] = ref_ptrs[p], dis_ptrs[p]
            r.row_pitch[p] = d.row_pitch[p] = row_pitch[p]
            r.frame_pitch[p] = d.frame_pitch[p] = frame_pitch[p]
        self._check(self.lib.pqa_submit_device(self._ctx, first_index, n_frames, C.byref(r), C.byref(d),
                                               prev_ref_luma_ptr or None, prev_row_pitch))

    @staticmethod
    def surface_clip(fmt: int, luma_ptr: int, luma_row_pitch: int, luma_frame_pitch: int, chroma_ptr: int = 0,
                     chroma_row_pitch: int = 0, chroma_frame_pitch: int = 0) -> "N.PqaSurfaceClip":
        """A clip of decoder surfaces in device memory (NV12 / P010 / P012; pitches in bytes)."""
        s = N.PqaSurfaceClip()
        s.struct_size = C.sizeof(N.PqaSurfaceClip)
        s.format = fmt
        s.luma, s.chroma = luma_ptr or None, chroma_ptr or None
        s.luma_row_pitch, s.luma_frame_pitch = luma_row_pitch, luma_frame_pitch
        s.chroma_row_pitch, s.chroma_frame_pitch = chroma_row_pitch, chroma_frame_pitch
        return s

    def submit_surfaces(self, first_index: int, n_frames: int, ref: "N.PqaSurfaceClip", dis: "N.PqaSurfaceClip",
                        prev_ref: "N.PqaSurfaceClip | None" = None):
        """Frames as a hardware decoder leaves them (pqa_submit_surfaces): NV12 luma is scored in place, interleaved
        chroma is split and 16-bit samples are shifted down on the device."""
        self._check(self.lib.pqa_submit_surfaces(self._ctx, first_index, n_frames, C.byref(ref), C.byref(dis),
                                                 C.byref(prev_ref) if prev_ref is not None else None))

    def set_luma_gray(self, mode: int):
        """N.GRAY_LUMA: statistics of the luma samples; N.GRAY_BT601_FULL: of the limited -> full range gray the
        reference's cv2 path sees (8-bit units for every bit depth; thresholds in those units)."""
        self._check(self.lib.pqa_set_luma_gray(self._ctx, int(mode)))
        self.luma_gray = int(mode)

    def luma_stats_resident(self, luma_ptr: int, row_pitch: int, frame_pitch: int, n_frames: int,
                            threshold: int) -> np.ndarray:
        """[n,3] uint64 {sum, sum of squares, count(sample > threshold)} per frame of a clip in HBM."""
        out = np.zeros((n_frames, 3), np.uint64)
        self._check(self.lib.pqa_luma_stats_device(self._ctx, luma_ptr, row_pitch, frame_pitch, n_frames,
                                                   int(threshold), out.ctypes.data))
        return out

    def luma_stats(self, luma_frames, threshold: int) -> np.ndarray:
        """[n,3] uint64 {sum, sum of squares, count(sample > threshold)} for luma planes in HOST memory (a list of 2-D
        arrays of this engine's dtype with one common row stride; they are packed and uploaded by the library)."""
        n = len(luma_frames)
        out = np.zeros((n, 3), np.uint64)
        if n == 0:
            return out
        keep, ptrs = [], (C.c_void_p * n)()
        stride = None
        for i, f in enumerate(luma_frames):
            a = np.asarray(f)
            if a.dtype != self.dtype or a.strides[1] != a.itemsize or (stride is not None and a.strides[0] != stride):
                a = np.ascontiguousarray(a, dtype=self.dtype)
            if stride is None:
                stride = a.strides[0]
            if a.strides[0] != stride:   # the first frame had an odd stride: normalise everything
                return self.luma_stats([np.ascontiguousarray(x, dtype=self.dtype) for x in luma_frames], threshold)
            if a.shape != (self.height, self.width):
                raise ValueError(f"luma frame {i} is {a.shape}, engine is {(self.height, self.width)}")
            keep.append(a)
            ptrs[i] = a.ctypes.data
        self._check(self.lib.pqa_luma_stats(self._ctx, ptrs, stride, n, int(threshold), out.ctypes.data))
        return out

    # -- results -----------------------------------------------------------------------------
    def collect(self, first_index: int, count: int) -> np.ndarray:
        out = np.zeros((count, N.RECORD_DOUBLES), np.float64)
        self._check(self.lib.pqa_collect(self._ctx, first_index, count, out.ctypes.data))
        return out

    def flush(self):
        self._check(self.lib.pqa_flush(self._ctx))

    def sync(self):
        self._check(self.lib.pqa_sync(self._ctx))

    def cancel(self):
        self.lib.pqa_cancel(self._ctx)

    def reset(self):
        self._check(self.lib.pqa_reset(self._ctx))

    # -- measurement ---------------------------------------------------------------------------
    def profile_enable(self, on=True):
        """True: time every kernel; False: stop; an iterable of kernel ids: time only those (event records
        between kernels are not free, so the bench times just the dominant kernel inside its timed region)."""
        if on is True:
            code = 1
        elif not on:
            code = 0
        else:
            mask = 0
            for k in on:
                mask |= 1 << int(k)
            code = mask << 1
        self._check(self.lib.pqa_profile_enable(self._ctx, code))

    def profile_read(self) -> dict:
        out = {}
        for k in range(N.PROF_KERNELS):
            ms, n, fr = C.c_double(), C.c_uint64(), C.c_uint64()
            self._check(self.lib.pqa_profile_read(self._ctx, k, C.byref(ms), C.byref(n), C.byref(fr)))
            out[self.lib.pqa_profile_kernel_name(k).decode()] = {"ms": ms.value, "launches": n.value, "frames": fr.value}
        return out


def sse_from_records(rec: np.ndarray) -> np.ndarray:
    """[n,3] uint64 SSE (Y,U,V) bit-cast out of the record slots."""
    return np.ascontiguousarray(rec[:, N.REC_SSE:N.REC_SSE + 3]).view(np.uint64)
