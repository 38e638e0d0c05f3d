"""ctypes front-end of the fixed-point oracle (oracle/vmaf_int_oracle.c).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see the header of vmaf_int_oracle.c): a from-memory restatement of libvmaf's
integer_vif / integer_adm / integer_motion, the extractors `model=version=vmaf_v0.6.1`
(app/vmaf_analyzer.py:377) selects.  Used to measure how far the float extractors the HIP kernels
implement sit from the fixed-point ones; never imported by pqa2_amd/.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import oracle as _o

_HERE = os.path.dirname(os.path.abspath(__file__))
N_FEAT = _o.N_FEAT


class IntOracle:
    def __init__(self):
        _o.build()
        if not os.path.exists(os.path.join(_HERE, "liboracle_int.so")):
            _o.build(force=True)
        self.lib = L = C.CDLL(os.path.join(_HERE, "liboracle_int.so"))
        L.orc_int_motion_blur.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_int_motion_sad.restype = C.c_uint64
        L.orc_int_motion_sad.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.orc_int_motion_score.restype = C.c_double
        L.orc_int_motion_score.argtypes = [C.c_uint64, C.c_int, C.c_int]
        for fn in (L.orc_int_vif, L.orc_int_adm):
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
        L.orc_int_init.restype = None
        L.orc_int_init()
        L.orc_int_log2_entry.restype = C.c_uint16
        L.orc_int_log2_entry.argtypes = [C.c_int]

    def motion_blur(self, ref_y: np.ndarray, bpc: int = 8) -> np.ndarray:
        ref_y = np.ascontiguousarray(ref_y)
        h, w = ref_y.shape
        out = np.empty((h, w), np.uint16)
        self.lib.orc_int_motion_blur(ref_y.ctypes.data, ref_y.strides[0], bpc, w, h, out.ctypes.data)
        return out

    def motion_sad(self, a: np.ndarray, b: np.ndarray) -> int:
        h, w = a.shape
        return int(self.lib.orc_int_motion_sad(a.ctypes.data, b.ctypes.data, w, h))

    def motion_score(self, sad: int, w: int, h: int) -> float:
        return float(self.lib.orc_int_motion_score(sad, w, h))

    def _numden(self, fn, ref_y, dis_y, bpc, gain_limit):
        ref_y = np.ascontiguousarray(ref_y)
        dis_y = np.ascontiguousarray(dis_y)
        assert ref_y.shape == dis_y.shape and ref_y.dtype == dis_y.dtype
        assert ref_y.dtype == (np.uint8 if bpc <= 8 else np.uint16) and ref_y.strides == dis_y.strides
        h, w = ref_y.shape
        out = np.zeros(8, np.float64)
        if fn(ref_y.ctypes.data, dis_y.ctypes.data, ref_y.strides[0], bpc, w, h, gain_limit, out.ctypes.data) != 0:
            raise MemoryError("oracle allocation failed")
        return out

    def vif(self, ref_y, dis_y, bpc: int = 8, gain_limit: float = 100.0) -> np.ndarray:
        return self._numden(self.lib.orc_int_vif, ref_y, dis_y, bpc, gain_limit)

    def adm(self, ref_y, dis_y, bpc: int = 8, gain_limit: float = 100.0) -> np.ndarray:
        return self._numden(self.lib.orc_int_adm, ref_y, dis_y, bpc, gain_limit)

    def clip_features(self, ref_frames, dis_frames, bpc: int = 8, vif_gain_limit: float = 100.0,
                      adm_gain_limit: float = 100.0) -> np.ndarray:
        """[n, 17] raw records in the float oracle's layout (vif num/den, adm num/den, motion)."""
        out = []
        prev = None
        for r, d in zip(ref_frames, dis_frames):
            f = np.zeros(N_FEAT)
            f[0:8] = self.vif(r, d, bpc, vif_gain_limit)
            f[8:16] = self.adm(r, d, bpc, adm_gain_limit)
            blur = self.motion_blur(r, bpc)
            if prev is not None:
                h, w = blur.shape
                f[16] = self.motion_score(self.motion_sad(prev, blur), w, h)
            prev = blur
            out.append(f)
        return np.stack(out) if out else np.zeros((0, N_FEAT))

    def clip_features_mt(self, ref_frames, dis_frames, bpc: int = 8, threads: int = 4, vif_gain_limit: float = 100.0,
                         adm_gain_limit: float = 100.0) -> np.ndarray:
        """clip_features on a thread pool (ctypes releases the GIL): frame i blurs reference frame i-1 itself."""
        from concurrent.futures import ThreadPoolExecutor

        def one(i):
            f = np.zeros(N_FEAT)
            f[0:8] = self.vif(ref_frames[i], dis_frames[i], bpc, vif_gain_limit)
            f[8:16] = self.adm(ref_frames[i], dis_frames[i], bpc, adm_gain_limit)
            if i > 0:
                a, b = self.motion_blur(ref_frames[i - 1], bpc), self.motion_blur(ref_frames[i], bpc)
                f[16] = self.motion_score(self.motion_sad(a, b), b.shape[1], b.shape[0])
            return f

        with ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
            out = list(ex.map(one, range(len(ref_frames))))
        return np.stack(out) if out else np.zeros((0, N_FEAT))
