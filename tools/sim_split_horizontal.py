#!/usr/bin/env python3
"""Would VIF scale 0's HORIZONTAL pass survive the matrix cores?  (numpy simulation, no GPU)

The vertical pass runs on the f16 MFMA pipe exactly because its inputs are small integers.  The horizontal pass works on
the f32 results of the vertical one; to go through f16 MFMA they would have to be split into f16 pieces.  Three pieces
reproduce f32 but cost more VALU than the packed FMAs they replace.  TWO pieces (22 bits) and two
tap pieces (second pieces error-diffused so that the taps still sum to the f32 taps' sum) would cut the scale-0 kernel by
an estimated 13-18 % -- at the price of a 4x coarser intermediate.  This script measures that price on the feature level:
VIF scale-0 numerator / denominator from (a) an all-f32 evaluation in libvmaf's order, (b) the split evaluation, both
against an f64 evaluation of the same formulas.
usage: sim_split_horizontal.py [--size 1920x1080] [--frames 3]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pqa2_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="1920x1080")
ap.add_argument("--frames", type=int, default=3)
ap.add_argument("--flat", action="store_true", help="bright low-contrast content (worst case for E[x^2] - mu^2)")
a = ap.parse_args()
w, h = map(int, a.size.split("x"))


def taps17():
    s = 17 / 5.0
    d = np.arange(17) - 8
    v = np.exp(-0.5 * d * d / (s * s))
    return (v / v.sum()).astype(np.float32)


C = taps17()


def pad(x, r, axis):   # vif_tools.c border: -i -> i, n-1+i -> n-i
    n = x.shape[axis]
    idx = np.arange(-r, n + r)
    idx = np.where(idx < 0, -idx, idx)
    idx = np.where(idx >= n, 2 * n - idx - 1, idx)
    return np.take(x, idx, axis=axis)


def conv(x, taps, axis, dt):
    xp = pad(x, 8, axis).astype(dt)
    out = np.zeros(x.shape, dt)
    n = x.shape[axis]
    for k in range(17):   # tap by tap, libvmaf's order
        sl = [slice(None)] * 2
        sl[axis] = slice(k, k + n)
        out = (out + dt(taps[k]) * xp[tuple(sl)]).astype(dt)
    return out


def split2(v):   # f32 -> two f16 pieces, round to nearest each
    hi = v.astype(np.float16)
    lo = (v - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def tap_pieces():   # c * 2^11 as two f16 pieces; second pieces adjusted so the pieces sum to the f32 taps' sum
    c = C.astype(np.float64) * 2048.0
    p1 = c.astype(np.float16).astype(np.float64)
    p2 = (c - p1).astype(np.float16).astype(np.float64)
    err = (c - p1 - p2)           # what two pieces lose per tap
    # push the total loss into the centre tap's second piece (it has the coarsest ulp headroom there anyway)
    p2[8] = np.float64(np.float16(p2[8] + err.sum()))
    return p1 / 2048.0, p2 / 2048.0, float((c - p1 - p2).sum() / 2048.0)


def conv_split(T, axis):
    hi, lo = split2(T)
    p1, p2, _ = tap_pieces()
    hp, lp = pad(hi.astype(np.float64), 8, axis), pad(lo.astype(np.float64), 8, axis)
    n = T.shape[axis]
    # exact f16 x f16 products (f64 here), accumulated in f32 one product at a time: a pessimistic model of the MFMA's
    # f32 accumulator (which adds several products per step before rounding)
    acc = np.zeros(T.shape, np.float32)
    for k in range(17):
        sl = [slice(None)] * 2
        sl[axis] = slice(k, k + n)
        for prod in (p1[k] * hp[tuple(sl)], p2[k] * hp[tuple(sl)], p1[k] * lp[tuple(sl)]):
            acc = (acc.astype(np.float64) + prod).astype(np.float32)
    return acc


def vif_stat(mu1, mu2, xx, yy, xy, dt):
    s1 = np.maximum(xx - mu1 * mu1, 0); s2 = np.maximum(yy - mu2 * mu2, 0); s12 = xy - mu1 * mu2
    eps, nsq = dt(1e-10), dt(2.0)
    g = s12 / (s1 + eps)
    sv = s2 - g * s12
    g = np.where(s1 < eps, 0, g); sv = np.where(s1 < eps, s2, sv); s1 = np.where(s1 < eps, 0, s1)
    g = np.where(s2 < eps, 0, g); sv = np.where(s2 < eps, 0, sv)
    sv = np.where(g < 0, s2, sv); g = np.maximum(g, 0)
    sv = np.maximum(sv, eps)
    g = np.minimum(g, dt(100.0))
    num = np.log2(1 + g * g * s1 / (sv + nsq)); den = np.log2(1 + s1 / nsq)
    num = np.where(s12 < 0, 0, num)
    low = s1 < nsq
    num = np.where(low, 1 - s2 * dt(4.0 / (255 * 255)), num); den = np.where(low, 1, den)
    return float(num.astype(np.float64).sum()), float(den.astype(np.float64).sum())


def scale0(r, d, mode):
    dt = np.float64 if mode == "f64" else np.float32
    r = r.astype(dt) - dt(128); d = d.astype(dt) - dt(128)
    sig = [r, d, r * r, d * d, r * d]
    V = [conv(s, C, 0, dt) for s in sig]                       # vertical
    if mode == "split":
        H = [conv_split(v, 1) for v in V]
    else:
        H = [conv(v, C, 1, dt) for v in V]
    return vif_stat(*H, dt)


refs, diss = synth.make_clip(w, h, a.frames, 8, chroma=False)
if a.flat:   # low-contrast variant: most windows near the sigma_nsq = 2 branch point and deep in the cancellation regime
    rng = np.random.default_rng(7)
    for i in range(a.frames):
        base = 200.0 + 6.0 * (refs[i][0].astype(np.float64) - 128.0) / 128.0
        refs[i][0][:] = np.clip(np.rint(base + rng.normal(0, 1.2, base.shape)), 0, 255).astype(np.uint8)
        diss[i][0][:] = np.clip(np.rint(base + rng.normal(0, 1.6, base.shape)), 0, 255).astype(np.uint8)
print(f"{a.size}, {a.frames} frames; tap pieces' sum error after diffusion: {tap_pieces()[2]:.2e}")
worst = {"f32": 0.0, "split": 0.0}
for i in range(a.frames):
    t = scale0(refs[i][0], diss[i][0], "f64")
    for mode in ("f32", "split"):
        g = scale0(refs[i][0], diss[i][0], mode)
        e = max(abs(g[0] - t[0]) / t[0], abs(g[1] - t[1]) / t[1])
        worst[mode] = max(worst[mode], e)
        print(f"frame {i} {mode:5s}: num rel err {abs(g[0] - t[0]) / t[0]:.2e}  den rel err {abs(g[1] - t[1]) / t[1]:.2e}")
print(f"worst relative feature error vs f64: all-f32 {worst['f32']:.2e}, two-piece horizontal pass {worst['split']:.2e}")
