"""ctypes front-end of the CPU oracle (oracle/vmaf_oracle.c).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED against libvmaf (see the header of vmaf_oracle.c): the reference's hot path
is `ffmpeg -lavfi libvmaf` (app/vmaf_analyzer.py:406-419,446) and `psnr=`/`ssim=`
(app/vmaf_analyzer.py:1027-1034,1057-1064); none of those binaries exist offline.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

N_FEAT = 17  # vif num[4], vif den[4], adm num[4], adm den[4], motion


def build(force: bool = False) -> None:
    """Compile liboracle_f32.so / liboracle_f64.so with gcc (a few seconds)."""
    need = force or any(
        not os.path.exists(os.path.join(_HERE, n)) or
        os.path.getmtime(os.path.join(_HERE, n)) < os.path.getmtime(os.path.join(_HERE, "vmaf_oracle.c"))
        for n in ("liboracle_f32.so", "liboracle_f64.so"))
    if need:
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)


class Oracle:
    """One precision flavour of the oracle: 'f32' (libvmaf's arithmetic type) or 'f64'."""

    def __init__(self, precision: str = "f32"):
        assert precision in ("f32", "f64")
        build()
        self.precision = precision
        self.real = np.float32 if precision == "f32" else np.float64
        self.lib = C.CDLL(os.path.join(_HERE, f"liboracle_{precision}.so"))
        L = self.lib
        L.orc_real_size.restype = C.c_int
        assert L.orc_real_size() == np.dtype(self.real).itemsize
        L.orc_frame_features2.restype = C.c_int
        L.orc_frame_features2.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_sse_plane.restype = C.c_uint64
        L.orc_sse_plane.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_ssim_plane.restype = C.c_double
        L.orc_ssim_plane.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_picture_copy.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_vif.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]
        L.orc_adm.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]
        L.orc_motion_blur.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_motion_sad.restype = C.c_double
        L.orc_motion_sad.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.orc_gaussian_taps.argtypes = [C.c_int, C.c_void_p]
        L.orc_adm_rfactors.argtypes = [C.c_void_p]

    # -- small accessors ------------------------------------------------------------------
    def gaussian_taps(self, n: int) -> np.ndarray:
        out = np.zeros(n, np.float64)
        self.lib.orc_gaussian_taps(n, out.ctypes.data)
        return out

    def adm_rfactors(self) -> np.ndarray:
        out = np.zeros(8, np.float64)
        self.lib.orc_adm_rfactors(out.ctypes.data)
        return out.reshape(4, 2)

    def picture_copy(self, plane: np.ndarray, bpc: int) -> np.ndarray:
        plane = np.ascontiguousarray(plane)
        h, w = plane.shape
        out = np.empty((h, w), self.real)
        self.lib.orc_picture_copy(plane.ctypes.data, plane.strides[0], bpc, w, h, out.ctypes.data)
        return out

    # -- per-frame features ---------------------------------------------------------------
    def frame_features(self, ref_y: np.ndarray, dis_y: np.ndarray, bpc: int = 8,
                       prev_blur: np.ndarray | None = None,
                       vif_gain_limit: float = 100.0, adm_gain_limit: float = 100.0, vif_border101: bool = False):
        """Returns (feat[17] float64, blur plane) for one frame pair (luma planes, uint8/uint16).
        vif_border101: pad VIF the way integer_vif.c does (reflect-101 on both edges) instead of vif_tools.c."""
        ref_y = np.ascontiguousarray(ref_y)
        dis_y = np.ascontiguousarray(dis_y)
        assert ref_y.shape == dis_y.shape and ref_y.dtype == dis_y.dtype
        assert ref_y.dtype == (np.uint8 if bpc <= 8 else np.uint16)
        h, w = ref_y.shape
        feat = np.zeros(N_FEAT, np.float64)
        blur = np.empty((h, w), self.real)
        rc = self.lib.orc_frame_features2(
            ref_y.ctypes.data, dis_y.ctypes.data, ref_y.strides[0], bpc, w, h,
            vif_gain_limit, adm_gain_limit,
            prev_blur.ctypes.data if prev_blur is not None else None,
            blur.ctypes.data, feat.ctypes.data, 1 if vif_border101 else 0)
        if rc != 0:
            raise MemoryError("oracle allocation failed")
        return feat, blur

    def clip_features(self, ref_frames, dis_frames, bpc: int = 8, prev_ref: np.ndarray | None = None,
                      vif_gain_limit: float = 100.0, adm_gain_limit: float = 100.0,
                      vif_border101: bool = False) -> np.ndarray:
        """[n, 17] raw feature records for a clip; motion of frame 0 uses prev_ref (halo) if given."""
        prev_blur = None
        if prev_ref is not None:
            _, prev_blur = self._blur_only(prev_ref, bpc)
        out = []
        for r, d in zip(ref_frames, dis_frames):
            f, prev_blur = self.frame_features(r, d, bpc, prev_blur, vif_gain_limit, adm_gain_limit, vif_border101)
            out.append(f)
        return np.stack(out) if out else np.zeros((0, N_FEAT))

    def clip_features_mt(self, ref_frames, dis_frames, bpc: int = 8, threads: int = 4,
                         vif_gain_limit: float = 100.0, adm_gain_limit: float = 100.0,
                         vif_border101: bool = False) -> np.ndarray:
        """clip_features on a thread pool (ctypes releases the GIL): frame i is independent once it blurs
        reference frame i-1 itself.  Same numbers as clip_features."""
        from concurrent.futures import ThreadPoolExecutor
        n = len(ref_frames)

        def one(i):
            prev_blur = self._blur_only(ref_frames[i - 1], bpc)[1] if i > 0 else None
            return self.frame_features(ref_frames[i], dis_frames[i], bpc, prev_blur, vif_gain_limit, adm_gain_limit,
                                       vif_border101)[0]

        with ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
            out = list(ex.map(one, range(n)))
        return np.stack(out) if out else np.zeros((0, N_FEAT))

    def _blur_only(self, ref_y: np.ndarray, bpc: int):
        ref = self.picture_copy(ref_y, bpc)
        blur = np.empty_like(ref)
        h, w = ref.shape
        self.lib.orc_motion_blur(ref.ctypes.data, w, h, blur.ctypes.data)
        return ref, blur

    # -- FFmpeg psnr / ssim side features ---------------------------------------------------
    def sse_plane(self, a: np.ndarray, b: np.ndarray, bpc: int = 8) -> int:
        a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
        h, w = a.shape
        return int(self.lib.orc_sse_plane(a.ctypes.data, a.strides[0], b.ctypes.data, b.strides[0], bpc, w, h))

    def ssim_plane(self, main: np.ndarray, ref: np.ndarray, bpc: int = 8) -> float:
        main = np.ascontiguousarray(main); ref = np.ascontiguousarray(ref)
        h, w = main.shape
        return float(self.lib.orc_ssim_plane(main.ctypes.data, main.strides[0], ref.ctypes.data,
                                             ref.strides[0], bpc, w, h))


# ---------------------------------------------------------------------------------------------
# feature records -> VMAF features (libvmaf float_vif.c / float_adm.c / float_motion.c epilogues)
# ---------------------------------------------------------------------------------------------
def finish_features(rec: np.ndarray, w: int, h: int) -> dict:
    """rec: [n,17] raw records -> dict of per-frame arrays adm2, adm_scale0..3, motion, motion2,
    vif_scale0..3 (float-extractor key names)."""
    rec = np.asarray(rec, np.float64)
    n = rec.shape[0]
    out = {}
    for s in range(4):
        out[f"vif_scale{s}"] = rec[:, s] / rec[:, 4 + s]
        out[f"adm_scale{s}"] = rec[:, 8 + s] / rec[:, 12 + s]
    numden_limit = 1e-10 * (w * h) / (1920.0 * 1080.0)
    num = rec[:, 8:12].sum(1)
    den = rec[:, 12:16].sum(1)
    num = np.where(num < numden_limit, 0.0, num)
    den = np.where(den < numden_limit, 0.0, den)
    out["adm2"] = np.where(den == 0.0, 1.0, num / np.where(den == 0.0, 1.0, den))
    motion = rec[:, 16].copy()
    out["motion"] = motion
    m2 = motion.copy()
    if n > 1:
        m2[:-1] = np.minimum(motion[:-1], motion[1:])
    out["motion2"] = m2  # frame 0: motion_0 = 0 so min(motion_0, motion_1) = 0 as libvmaf emits
    return out


# ---------------------------------------------------------------------------------------------
# nu-SVR predict, restated with plain Python loops (libsvm svm_predict + libvmaf predict.c).
# Pinned: (adm2, motion2, vif0..3) = (1,0,1,1,1,1) -> 97.428043 for vmaf_v0.6.1.
# ---------------------------------------------------------------------------------------------
def parse_libsvm_model(text: str):
    lines = text.strip().split("\n")
    hdr = {}
    i = 0
    while lines[i].strip() != "SV":
        k, *v = lines[i].split()
        hdr[k] = v
        i += 1
    svs = []
    for ln in lines[i + 1:]:
        parts = ln.split()
        if not parts:
            continue
        coef = float(parts[0])
        vec = {}
        for p in parts[1:]:
            idx, val = p.split(":")
            vec[int(idx)] = float(val)
        svs.append((coef, vec))
    return {"gamma": float(hdr["gamma"][0]), "rho": float(hdr["rho"][0]), "svs": svs}


def svr_predict_py(model_dict: dict, feats: list[float]) -> float:
    """feats in model feature_names order -> VMAF score (denormalised, clipped, no transform)."""
    svm = parse_libsvm_model(model_dict["model"])
    slopes, intercepts = model_dict["slopes"], model_dict["intercepts"]
    x = [slopes[k + 1] * feats[k] + intercepts[k + 1] for k in range(len(feats))]
    acc = 0.0
    for coef, vec in svm["svs"]:
        d2 = 0.0
        for k in range(len(x)):
            d = x[k] - vec.get(k + 1, 0.0)
            d2 += d * d
        acc += coef * math.exp(-svm["gamma"] * d2)
    y = acc - svm["rho"]
    y = (y - intercepts[0]) / slopes[0]
    lo, hi = model_dict.get("score_clip", [None, None])
    if lo is not None:
        y = min(max(y, lo), hi)
    return y
