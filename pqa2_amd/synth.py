"""Deterministic synthetic clips for tests and benchmarks (SURVEY.md 8(d) 'Synthetic inputs').

Reference clip: moving sinusoids (non-zero motion) + band-limited texture; distorted clip:
blur mix + noise + 8x8 block quantisation, which lands VMAF in the 60-90 range and exercises
both VIF branches and ADM masking.  Pure numpy; cheap enough to regenerate on the GPU box.
"""
from __future__ import annotations

import numpy as np

from .yuvio import VideoInfo

REF_SEED = 20250418
DIST_SEED = 20250419


def _blur_sep(x: np.ndarray, taps: np.ndarray) -> np.ndarray:
    r = len(taps) // 2
    p = np.pad(x, ((r, r), (0, 0)), mode="reflect")
    y = sum(taps[k] * p[k:k + x.shape[0]] for k in range(len(taps)))
    p = np.pad(y, ((0, 0), (r, r)), mode="reflect")
    return sum(taps[k] * p[:, k:k + x.shape[1]] for k in range(len(taps)))


def _gauss(sigma: float) -> np.ndarray:
    r = max(1, int(3 * sigma))
    t = np.exp(-0.5 * (np.arange(-r, r + 1) / sigma) ** 2)
    return (t / t.sum()).astype(np.float32)


def ref_luma(w: int, h: int, t: int, seed: int = REF_SEED) -> np.ndarray:
    """float32 luma in 8-bit units for frame t."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    u, v = xx / w, yy / h
    s = (np.sin(2 * np.pi * (3.0 * u + 0.011 * t)) * np.cos(2 * np.pi * (2.0 * v - 0.007 * t))
         + 0.6 * np.sin(2 * np.pi * (7.0 * u + 5.0 * v + 0.017 * t))
         + 0.4 * np.cos(2 * np.pi * (13.0 * u - 11.0 * v - 0.013 * t))
         + 0.3 * np.sin(2 * np.pi * (29.0 * v + 0.023 * t)))
    rng = np.random.Generator(np.random.PCG64(seed + t))
    tex = _blur_sep(rng.standard_normal((h, w), dtype=np.float32), _gauss(1.5))
    tex /= max(1e-6, float(tex.std()))
    return np.clip(128.0 + 26.0 * s + 25.0 * tex, 16.0, 235.0).astype(np.float32)


def dist_luma(ref: np.ndarray, t: int, seed: int = DIST_SEED, strength: float = 1.0) -> np.ndarray:
    h, w = ref.shape
    box = np.full(5, 0.2, np.float32)
    blurred = _blur_sep(ref, box)
    rng = np.random.Generator(np.random.PCG64(seed + t))
    x = (1 - 0.5 * strength) * ref + 0.5 * strength * blurred + 3.0 * strength * rng.standard_normal((h, w), dtype=np.float32)
    # 8x8 block quantisation of the block mean offset (emulates coarse DC quantisation)
    hb, wb = h // 8 * 8, w // 8 * 8
    if hb and wb:
        blk = x[:hb, :wb].reshape(hb // 8, 8, wb // 8, 8)
        m = blk.mean(axis=(1, 3), keepdims=True)
        q = 6.0 * strength
        blk += (np.round(m / q) * q - m) if q > 0 else 0
        x[:hb, :wb] = blk.reshape(hb, wb)
    return np.clip(x, 0.0, 255.0)


def _chroma(w: int, h: int, t: int, phase: float) -> np.ndarray:
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    return 128.0 + 20.0 * np.sin(2 * np.pi * (1.5 * xx / max(1, w) + 1.0 * yy / max(1, h) + 0.005 * t + phase))


def _quant(x: np.ndarray, bit_depth: int, rng=None) -> np.ndarray:
    if bit_depth <= 8:
        return np.clip(np.rint(x), 0, 255).astype(np.uint8)
    scale = 1 << (bit_depth - 8)
    y = x * scale
    if rng is not None:
        y = y + rng.integers(0, scale, size=x.shape)  # extra LSB noise
    return np.clip(np.rint(y), 0, (1 << bit_depth) - 1).astype(np.uint16)


def make_pair(w: int, h: int, t: int, bit_depth: int = 8, chroma: bool = True, strength: float = 1.0):
    """([Yr,Ur,Vr], [Yd,Ud,Vd]) for frame t.  4:2:0."""
    r = ref_luma(w, h, t)
    d = dist_luma(r, t, strength=strength)
    rng = np.random.Generator(np.random.PCG64(7919 + t)) if bit_depth > 8 else None
    ref = [_quant(r, bit_depth, rng)]
    dis = [_quant(d, bit_depth, rng)]
    if chroma:
        cw, ch = (w + 1) // 2, (h + 1) // 2
        for ph in (0.0, 0.37):
            c = _chroma(cw, ch, t, ph)
            crng = np.random.Generator(np.random.PCG64(104729 + t + int(ph * 100)))
            ref.append(_quant(c, bit_depth))
            dis.append(_quant(c + 1.5 * strength * crng.standard_normal(c.shape, dtype=np.float32), bit_depth))
    return ref, dis


def make_clip(w: int, h: int, n: int, bit_depth: int = 8, chroma: bool = True, strength: float = 1.0, t0: int = 0):
    refs, diss = [], []
    for t in range(t0, t0 + n):
        r, d = make_pair(w, h, t, bit_depth, chroma, strength)
        refs.append(r)
        diss.append(d)
    return refs, diss


def clip_info(w: int, h: int, bit_depth: int = 8, chroma: bool = True, fps: int = 30) -> VideoInfo:
    tag = ("mono" if not chroma else "420jpeg") if bit_depth <= 8 else (f"mono{bit_depth}" if not chroma else f"420p{bit_depth}")
    return VideoInfo(w, h, bit_depth, 1, 1, not chroma, fps, 1, 0, tag)
