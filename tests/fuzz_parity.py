#!/usr/bin/env python3
"""Long randomized parity run (not collected by pytest): random geometries, bit depths, content kinds, gain limits
and VIF borders (plus random chroma subsampling for the exact PSNR SSE and the SSIM) through the HIP library, against the
oracles.  f32 kernels: relative tolerance max(5e-5, 8 x the f32
oracle's own distance from f64) -- smooth content on tiny planes makes sigma = E[x^2] - mu^2 cancel to ~1e-3 per pixel in
ANY f32 evaluation order, libvmaf's included, so the bar scales with what f32 itself can hold; fixed-point kernels:
bit-equality.   usage: python tests/fuzz_parity.py [seconds] [seed] [hd]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle.int_oracle import IntOracle
from oracle.oracle import Oracle
from pqa2_amd import _native as N
from pqa2_amd import model as M
from pqa2_amd.engine import FeatureEngine, sse_from_records

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
hd_only = len(sys.argv) > 3 and sys.argv[3] == "hd"   # only frames of the sizes the 0.01 VMAF target is stated for
rng = np.random.default_rng(seed)
o64, o32, into = Oracle("f64"), Oracle("f32"), IntOracle()
t0 = time.time()
last = t0
n_cases = 0
worst = 0.0
ratio = 0.0
flips = 0
worst_dv = 0.0
worst_dv_f32 = 0.0   # |dVMAF| against the f32 oracle ALONE (reported, not a bar: the bar is on the nearer of the two oracles)
worst_dv_f64 = 0.0   # ... and against the f64 oracle alone
oracles_apart = 0.0  # ... and the two oracles' own |dVMAF|, for scale
worst_dv_small = 0.0
worst_small_tag = None
over_bar = []   # cases beyond the hard bars (5e-3 on a feature, 0.01 VMAF from 500k pixels): listed at the end, exit code 1
mdl = M.load_model("vmaf_v0.6.1")


def vmaf(rec17, w, h):
    full = np.zeros((rec17.shape[0], 24))
    full[:, :17] = rec17
    return M.score_frames(mdl, M.metrics_from_records(full, w, h))["vmaf"]


while time.time() - t0 < budget:
    if hd_only or rng.integers(0, 12) == 0:   # now and then a frame of the sizes the 0.01 VMAF target is stated for
        w, h = int(rng.integers(900, 1930)), int(rng.integers(540, 1090))
    else:
        w, h = int(rng.integers(16, 700)), int(rng.integers(16, 400))
    if rng.integers(0, 3) == 0:   # multiples of 4 (from 128 up) take the ADM pyramid kernel: make sure they come up
        w, h = max(16, w & ~3), max(16, h & ~3)
    bpc = int(rng.choice([8, 8, 8, 10, 12]))
    peak = (1 << bpc) - 1
    kind = int(rng.integers(0, 5))
    gain = float(rng.choice([100.0, 1.0, 1.5]))
    border = int(rng.integers(0, 2))
    n = 2
    frames = []
    for t in range(n):
        if kind == 0:
            y = rng.integers(0, peak + 1, (h, w))
        elif kind == 1:
            y = np.repeat(np.repeat(rng.integers(0, peak + 1, (-(-h // 16), -(-w // 16))), 16, 0), 16, 1)[:h, :w] + t
        elif kind == 2:
            y = (np.add.outer(np.arange(h) * 2, np.arange(w)) + 7 * t) % (peak + 1)
        elif kind == 3:
            y = ((np.add.outer(np.arange(h), np.arange(w)) + t) % 2) * peak
        else:   # smooth field + mild noise (natural-ish)
            yy, xx = np.mgrid[0:h, 0:w]
            y = peak * (0.5 + 0.3 * np.sin(xx / 17.0 + t) * np.cos(yy / 11.0)) + rng.normal(0, peak / 64, (h, w))
        frames.append(np.clip(y, 0, peak).astype(np.uint8 if bpc == 8 else np.uint16))
    amp = int(rng.integers(1, 12)) * (1 << (bpc - 8))
    dis = [np.clip(f.astype(np.int32) + rng.integers(-amp, amp + 1, f.shape), 0, peak).astype(f.dtype) for f in frames]
    tag = (w, h, bpc, kind, gain, border)
    exp = o64.clip_features(frames, dis, bpc, vif_gain_limit=gain, adm_gain_limit=gain, vif_border101=bool(border))
    exp32 = o32.clip_features(frames, dis, bpc, vif_gain_limit=gain, adm_gain_limit=gain, vif_border101=bool(border))
    # chroma planes for the FFmpeg psnr / ssim side features: random subsampling, own noise
    hs, vs = [(1, 1), (1, 0), (0, 0)][int(rng.integers(0, 3))]
    cw, ch = -(-w >> hs), -(-h >> vs)
    dt = np.uint8 if bpc == 8 else np.uint16
    cref = [[rng.integers(0, peak + 1, (ch, cw)).astype(dt) for _ in range(2)] for _ in range(n)]
    cdis = [[np.clip(p.astype(np.int32) + rng.integers(-amp, amp + 1, p.shape), 0, peak).astype(dt) for p in fr] for fr in cref]
    with FeatureEngine(w, h, bit_depth=bpc, n_planes=3, chroma_shift=(hs, vs), features=N.FEAT_ALL,
                       vif_enhn_gain_limit=gain, adm_enhn_gain_limit=gain, vif_border=border,
                       max_batch=int(rng.integers(1, 4)) + 7) as eng:
        for i in range(n):
            eng.submit(i, [frames[i]] + cref[i], [dis[i]] + cdis[i])
        rec = eng.collect(0, n)
    got = rec[:, :17]
    sse = sse_from_records(rec)
    for i in range(n):
        for p, (a_, b_) in enumerate(zip([dis[i]] + cdis[i], [frames[i]] + cref[i])):
            assert int(sse[i, p]) == o32.sse_plane(a_, b_, bpc), ("sse", tag, i, p)
            if min(a_.shape) >= 8:
                assert abs(rec[i, N.REC_SSIM + p] - o32.ssim_plane(a_, b_, bpc)) < 1e-9, ("ssim", tag, i, p)
    assert np.all(np.isfinite(got)), ("non-finite", tag)
    # distance to the nearer of the two oracles: where f32 and f64 disagree (a branch decided at a 1e-8 margin), agreeing
    # with libvmaf's own arithmetic type is as right as agreeing with the f64 truth
    rel = np.minimum(np.abs(got[:, :16] - exp[:, :16]), np.abs(got[:, :16] - exp32[:, :16])) / np.maximum(np.abs(exp[:, :16]), 1e-9)
    rel32 = np.abs(exp32[:, :16] - exp[:, :16]) / np.maximum(np.abs(exp[:, :16]), 1e-9)
    tol = max(5e-5, 8.0 * float(rel32.max()))
    ratio = max(ratio, float(rel.max()) / max(float(rel32.max()), 1.25e-5))
    if not rel.max() < tol:
        # A single f32 threshold flip (ADM's 1-degree angle test or a VIF branch decided at a 1e-8 margin) moves a
        # feature of a tiny band by ~1e-4: a coin toss in libvmaf's own f32 too.  Log it, keep the inputs, and hold
        # the hard bars instead: 2e-3 on any feature and the north-star 0.01 on the VMAF score.
        flips += 1
        os.makedirs("gpurun_out", exist_ok=True)
        np.savez_compressed(f"gpurun_out/fuzz_flip_{flips}.npz", ref=np.stack(frames), dis=np.stack(dis), tag=np.array(tag),
                            got=got, exp=exp, exp32=exp32)
        print(f"threshold-flip suspect {tag}: rel {rel.max():.2e} at {np.unravel_index(rel.argmax(), rel.shape)} "
              f"(f32 oracle {rel32.max():.2e})", flush=True)
    # One flipped pixel / coefficient at a deep scale is worth 1/(pixels there) of a feature, and the deep ADM scales
    # carry most of adm2: on a 71 x 16 band a single angle-test flip (margin 1e-7, verified by replaying the flip in
    # numpy) moved VMAF by 0.038 -- in libvmaf's own f32 as much as here.  So: 5e-3 on any feature at any size, and the
    # north-star 0.01 VMAF from 500k pixels up (the sizes it is stated for: 1080p, 2160p); smaller frames are reported.
    px = w * h
    # Ramps and checkerboards (kinds 2, 3) are degenerate for ADM: a whole row of coefficients shares one structure and
    # sits exactly ON a decision (k = t/o, the angle test), so the f32 evaluation order decides for all of them at once
    # -- the f32 oracle itself jumps by 2e-2 against f64 when such a clip is cropped by one row.  They stay in the run
    # for finiteness and for the fixed-point bit-equality; their f32 bar is the size of such a jump.
    feat_bar = 5e-2 if kind in (2, 3) else 5e-3
    if not rel.max() < feat_bar:   # recorded, the run goes on: a long run should end with the full statistics
        over_bar.append(("f32 parity", tag, float(rel.max()), float(rel32.max()), tuple(int(x) for x in np.unravel_index(rel.argmax(), rel.shape))))
        print("OVER THE BAR:", over_bar[-1], flush=True)
    dv = np.minimum(np.abs(vmaf(got, w, h) - vmaf(exp, w, h)), np.abs(vmaf(got, w, h) - vmaf(exp32, w, h))).max()
    if px >= 500_000 and kind not in (2, 3):
        worst_dv = max(worst_dv, float(dv))
        worst_dv_f32 = max(worst_dv_f32, float(np.abs(vmaf(got, w, h) - vmaf(exp32, w, h)).max()))
        worst_dv_f64 = max(worst_dv_f64, float(np.abs(vmaf(got, w, h) - vmaf(exp, w, h)).max()))
        oracles_apart = max(oracles_apart, float(np.abs(vmaf(exp32, w, h) - vmaf(exp, w, h)).max()))
        if not dv < 0.01:
            over_bar.append(("vmaf", tag, float(dv)))
            print("OVER THE BAR:", over_bar[-1], flush=True)
    elif kind not in (2, 3) and float(dv) > worst_dv_small:
        worst_dv_small, worst_small_tag = float(dv), tag
    assert abs(got[1, 16] - exp[1, 16]) < 2e-5 + 5e-6 * exp[1, 16], ("motion", tag, got[1, 16], exp[1, 16])
    worst = max(worst, float(rel.max()))
    want = np.zeros((n, 17))
    for i in range(n):
        want[i, 0:8] = into.vif(frames[i], dis[i], bpc, gain)
        want[i, 8:16] = into.adm(frames[i], dis[i], bpc, gain)
    want[1, 16] = into.motion_score(into.motion_sad(into.motion_blur(frames[0], bpc), into.motion_blur(frames[1], bpc)), w, h)
    with FeatureEngine(w, h, bit_depth=bpc, vif_enhn_gain_limit=gain, adm_enhn_gain_limit=gain,
                       fixed_point=N.FIXED_ALL) as eng:
        for i in range(n):
            eng.submit(i, [frames[i]], [dis[i]])
        fx = eng.collect(0, n)[:, :17]
    bad = np.argwhere(fx.view(np.uint64) != want.view(np.uint64))
    assert bad.size == 0, ("fixed-point", tag, bad[:4].tolist(), fx[tuple(bad[0])], want[tuple(bad[0])])
    n_cases += 1
    if time.time() - last > 30:
        last = time.time()
        print(f"{n_cases} cases ok, worst f32 rel err {worst:.2e} (worst gpu/f32-oracle error ratio {ratio:.1f}), last {tag}", flush=True)
print(f"fuzz {'ok' if not over_bar else 'done'}: {n_cases} cases in {time.time() - t0:.0f} s, worst f32 rel err {worst:.2e}, worst ratio {ratio:.1f}, "
      f"{flips} threshold-flip suspects, worst |dVMAF| {worst_dv:.4f} on frames >= 500k pixels (bar 0.01), {worst_dv_small:.4f} below (non-degenerate content; {worst_small_tag}); on those frames vs the f32 oracle alone {worst_dv_f32:.4f}, vs the f64 oracle alone {worst_dv_f64:.4f}, the two oracles apart {oracles_apart:.4f}; fixed-point mode bit-exact in every case")
if over_bar:
    print(f"{len(over_bar)} case(s) over a hard bar:")
    for c in over_bar:
        print("  ", c)
    sys.exit(1)
