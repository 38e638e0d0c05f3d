"""Independent float64 numpy restatement of the VMAF feature definitions.  TEST INFRASTRUCTURE.

Written separately from oracle/vmaf_oracle.c (vectorised, padding-based instead of index
mirroring, whole-plane instead of scanline) so that a transcription slip in either shows up as
a disagreement.  Follows SURVEY.md Appendix C.2-C.4 and section 8(a) rows a4-VIF/ADM/motion
(the arithmetic behind the reference's `libvmaf=` call site, app/vmaf_analyzer.py:373-419).
PARITY UNPINNED against libvmaf itself -- see oracle/vmaf_oracle.c.
"""
from __future__ import annotations

import numpy as np


def _pad_axis(x: np.ndarray, lo: int, hi: int, axis: int) -> np.ndarray:
    """libvmaf border rule: index -i -> i (no edge repeat), index n-1+i -> n-i (edge repeated)."""
    x = np.moveaxis(x, axis, 0)
    n = x.shape[0]
    idx = np.arange(-lo, n + hi)
    idx = np.where(idx < 0, -idx, idx)
    idx = np.where(idx >= n, 2 * n - idx - 1, idx)
    # fold again for planes smaller than the radius
    for _ in range(8):
        idx = np.where(idx < 0, -idx, idx)
        idx = np.where(idx >= n, 2 * n - idx - 1, idx)
    return np.moveaxis(x[idx], 0, axis)


def gaussian_taps(n: int) -> np.ndarray:
    k = np.arange(n) - n // 2
    t = np.exp(-0.5 * (k / (n / 5.0)) ** 2)
    return (t / t.sum()).astype(np.float32).astype(np.float64)


def _corr_axis(x: np.ndarray, taps: np.ndarray, axis: int, step: int = 1, first: int | None = None,
               n_out: int | None = None) -> np.ndarray:
    """correlate along axis; output sample o reads inputs first + o*step + [0..len(taps))."""
    r = len(taps) // 2
    first = -r if first is None else first
    n = x.shape[axis]
    n_out = n if n_out is None else n_out
    lo = max(0, -first)
    hi = max(0, first + (n_out - 1) * step + len(taps) - n)
    p = _pad_axis(x, lo, hi, axis)
    p = np.moveaxis(p, axis, 0)
    base = first + lo
    out = 0.0
    for k, c in enumerate(taps):
        out = out + c * p[base + k: base + k + (n_out - 1) * step + 1: step]
    return np.moveaxis(out, 0, axis)


def _sep(x: np.ndarray, taps: np.ndarray) -> np.ndarray:
    return _corr_axis(_corr_axis(x, taps, 0), taps, 1)


def _sep101(x: np.ndarray, taps: np.ndarray) -> np.ndarray:
    """Same separable filter with integer_vif.c's padding: reflect-101 on all four edges (numpy's 'reflect')."""
    r = len(taps) // 2
    p = np.pad(x, r, mode="reflect")
    v = sum(c * p[k: k + x.shape[0], :] for k, c in enumerate(taps))
    return sum(c * v[:, k: k + x.shape[1]] for k, c in enumerate(taps))


def picture_copy(plane: np.ndarray, bpc: int) -> np.ndarray:
    x = plane.astype(np.float64)
    if bpc > 8:
        x = x / float(1 << (bpc - 8))
    return x - 128.0


def vif(ref: np.ndarray, dis: np.ndarray, gain_limit: float = 100.0, border101: bool = False):
    """(num[4], den[4]) from picture_copy'd planes."""
    _sep = _sep101 if border101 else globals()["_sep"]
    num, den = np.zeros(4), np.zeros(4)
    for s, n in enumerate((17, 9, 5, 3)):
        f = gaussian_taps(n)
        if s > 0:
            ref = _sep(ref, f)[::2, ::2][: ref.shape[0] // 2, : ref.shape[1] // 2]
            dis = _sep(dis, f)[::2, ::2][: dis.shape[0] // 2, : dis.shape[1] // 2]
        mu1, mu2 = _sep(ref, f), _sep(dis, f)
        # squares / products are formed before the vertical pass
        xx = _sep(ref * ref, f) - mu1 * mu1
        yy = _sep(dis * dis, f) - mu2 * mu2
        xy = _sep(ref * dis, f) - mu1 * mu2
        s1 = np.maximum(xx, 0.0)
        s2 = np.maximum(yy, 0.0)
        eps = 1e-10
        g = xy / (s1 + eps)
        sv = s2 - g * xy
        m = s1 < eps
        g = np.where(m, 0.0, g); sv = np.where(m, s2, sv); s1 = np.where(m, 0.0, s1)
        m = s2 < eps
        g = np.where(m, 0.0, g); sv = np.where(m, 0.0, sv)
        m = g < 0
        sv = np.where(m, s2, sv); g = np.where(m, 0.0, g)
        sv = np.maximum(sv, eps)
        g = np.minimum(g, gain_limit)
        nv = np.log2(1.0 + g * g * s1 / (sv + 2.0))
        dv = np.log2(1.0 + s1 / 2.0)
        nv = np.where(xy < 0, 0.0, nv)
        low = s1 < 2.0
        nv = np.where(low, 1.0 - s2 * (4.0 / (255.0 * 255.0)), nv)
        dv = np.where(low, 1.0, dv)
        num[s], den[s] = nv.sum(), dv.sum()
    return num, den


_LO = np.array([0.482962913144690, 0.836516303737469, 0.224143868041857, -0.129409522550921],
               np.float32).astype(np.float64)
_HI = np.array([-0.129409522550921, -0.224143868041857, 0.836516303737469, -0.482962913144690],
               np.float32).astype(np.float64)

_AMP = [(0.67234, 0.72709), (0.41317, 0.49428), (0.22727, 0.28688), (0.11792, 0.15214)]


def rfactors(scale: int):
    r = 3.0 * 1080 * np.pi / 180.0
    out = []
    for g, amp in ((1.0, _AMP[scale][0]), (0.534, _AMP[scale][1])):
        t = np.log10(2.0 ** (scale + 1) * 0.401 * g / r)
        q = 2.0 * 0.495 * 10.0 ** (0.466 * t * t) / amp
        out.append(1.0 / q)
    return out[0], out[1]


def _dwt2(x: np.ndarray):
    h, w = x.shape
    oh, ow = (h + 1) // 2, (w + 1) // 2
    lo_v = _corr_axis(x, _LO, 0, step=2, first=-1, n_out=oh)
    hi_v = _corr_axis(x, _HI, 0, step=2, first=-1, n_out=oh)
    a = _corr_axis(lo_v, _LO, 1, step=2, first=-1, n_out=ow)
    v = _corr_axis(lo_v, _HI, 1, step=2, first=-1, n_out=ow)
    hh = _corr_axis(hi_v, _LO, 1, step=2, first=-1, n_out=ow)
    d = _corr_axis(hi_v, _HI, 1, step=2, first=-1, n_out=ow)
    return a, hh, v, d


def adm(ref: np.ndarray, dis: np.ndarray, gain_limit: float = 100.0):
    """(num[4], den[4]) from picture_copy'd planes."""
    num, den = np.zeros(4), np.zeros(4)
    cos2 = np.cos(np.pi / 180.0) ** 2
    for s in range(4):
        ra, rh, rv, rd = _dwt2(ref)
        da, dh, dv, dd = _dwt2(dis)
        h, w = ra.shape
        dp = rh * dh + rv * dv
        ok = (dp >= 0) & (dp * dp >= cos2 * (rh * rh + rv * rv) * (dh * dh + dv * dv))
        rf_hv, rf_d = rfactors(s)
        left, top = int(w * 0.1 - 0.5), int(h * 0.1 - 0.5)
        right, bottom = w - left, h - top
        area_term = np.cbrt((bottom - top) * (right - left) / 32.0)
        thr = 0.0
        R = []
        for o, t, rf in ((rh, dh, rf_hv), (rv, dv, rf_hv), (rd, dd, rf_d)):
            k = np.clip(t / (o + 1e-30), 0.0, 1.0)
            rst = k * o
            lim = np.where(rst > 0, np.minimum(rst * gain_limit, t),
                           np.where(rst < 0, np.maximum(rst * gain_limit, t), rst))
            rst = np.where(ok, lim, rst)
            a_ = t - rst
            F = np.abs(a_ * rf) / 30.0
            P = _pad_axis(_pad_axis(F, 1, 1, 0), 1, 1, 1)
            box = sum(P[i:i + h, j:j + w] for i in range(3) for j in range(3))
            thr = thr + box + F
            R.append(np.abs(rst * rf))
            den[s] += np.cbrt((np.abs(o * rf)[top:bottom, left:right] ** 3).sum()) + area_term
        for Rt in R:
            x = np.maximum(Rt - thr, 0.0)[top:bottom, left:right]
            num[s] += np.cbrt((x ** 3).sum()) + area_term
        ref, dis = ra, da
    return num, den


def motion_blur(ref: np.ndarray) -> np.ndarray:
    return _sep(ref, gaussian_taps(5))


def frame_features(ref_y, dis_y, bpc=8, prev_blur=None, vif_gain_limit=100.0, adm_gain_limit=100.0):
    r, d = picture_copy(ref_y, bpc), picture_copy(dis_y, bpc)
    vn, vd = vif(r, d, vif_gain_limit)
    an, ad = adm(r, d, adm_gain_limit)
    blur = motion_blur(r)
    mot = 0.0 if prev_blur is None else float(np.abs(blur - prev_blur).mean())
    return np.concatenate([vn, vd, an, ad, [mot]]), blur


def sse_plane(a: np.ndarray, b: np.ndarray) -> int:
    d = a.astype(np.int64) - b.astype(np.int64)
    return int((d * d).sum())


def ssim_plane(main: np.ndarray, ref: np.ndarray, bpc: int = 8) -> float:
    """FFmpeg vf_ssim.c definition in float64 (window ratio not rounded to float32)."""
    h, w = main.shape
    bh, bw = h >> 2, w >> 2
    if bh < 2 or bw < 2:
        return 0.0
    a = main[: bh * 4, : bw * 4].astype(np.int64).reshape(bh, 4, bw, 4)
    b = ref[: bh * 4, : bw * 4].astype(np.int64).reshape(bh, 4, bw, 4)
    s1, s2 = a.sum((1, 3)), b.sum((1, 3))
    ss = (a * a).sum((1, 3)) + (b * b).sum((1, 3))
    s12 = (a * b).sum((1, 3))
    win = lambda x: x[:-1, :-1] + x[:-1, 1:] + x[1:, :-1] + x[1:, 1:]
    s1, s2, ss, s12 = win(s1), win(s2), win(ss), win(s12)
    mx = (1 << bpc) - 1
    c1 = int(.01 * .01 * mx * mx * 64 + .5)
    c2 = int(.03 * .03 * mx * mx * 64 * 63 + .5)
    vars_ = ss * 64 - s1 * s1 - s2 * s2
    covar = s12 * 64 - s1 * s2
    val = ((2 * s1 * s2 + c1).astype(np.float64) * (2 * covar + c2).astype(np.float64)
           / ((s1 * s1 + s2 * s2 + c1).astype(np.float64) * (vars_ + c2).astype(np.float64)))
    return float(val.mean())
