#!/usr/bin/env python3
"""What does taking frames as decoder surfaces cost?  The same device-resident 4:2:0 clip scored (VMAF + PSNR + SSIM on
all planes) from planar planes (pqa_submit_device) and from NV12 / P010 surfaces (pqa_submit_surfaces: chroma split,
16-bit samples shifted down, on the device), and VMAF alone from an NV12 luma plane in place.
usage: surface_bench.py [--size 3840x2160] [--frames 64] [--bits 8]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pqa2_amd import _native as N, synth_torch
from pqa2_amd.engine import FeatureEngine

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="3840x2160")
ap.add_argument("--frames", type=int, default=64)
ap.add_argument("--bits", type=int, default=8)
a = ap.parse_args()
w, h = map(int, a.size.split("x"))
n, bpc = a.frames, a.bits
es = 1 if bpc <= 8 else 2
sh = 0 if bpc <= 8 else 16 - bpc
clip = synth_torch.make_clip_cuda(w, h, n, bpc, chroma=True)
R, D = clip["ref"], clip["dis"]
cw, ch = R[1].shape[2], R[1].shape[1]

def surf(P):
    L = (P[0] << sh) if sh else P[0]
    C = torch.empty((n, ch, 2 * cw), dtype=P[1].dtype, device="cuda")
    C[:, :, 0::2] = (P[1] << sh) if sh else P[1]
    C[:, :, 1::2] = (P[2] << sh) if sh else P[2]
    return L.contiguous(), C
RL, RC = surf(R); DL, DC = surf(D)
torch.cuda.synchronize()
fmt = N.SURFACE_NV12 if bpc <= 8 else N.SURFACE_P01X

def timed(fn, eng, rounds=5):
    best = 1e9
    for _ in range(rounds + 1):
        eng.lib.pqa_reset(eng._ctx)
        t = time.perf_counter(); fn(eng); eng.lib.pqa_sync(eng._ctx)
        best = min(best, time.perf_counter() - t)
    return round(n / best, 1)

out = {"size": a.size, "bits": bpc, "frames": n}
kw = dict(bit_depth=bpc, n_planes=3, features=N.FEAT_ALL, max_batch=16)
with FeatureEngine(w, h, **kw) as eng:
    rp = [w * es, cw * es, cw * es]; fp = [w * h * es, cw * ch * es, cw * ch * es]
    out["planar_all_planes_fps"] = timed(lambda e: e.submit_resident(0, n, [p.data_ptr() for p in R], [p.data_ptr() for p in D], rp, fp), eng)
    mk = lambda L, C: FeatureEngine.surface_clip(fmt, L.data_ptr(), w * es, w * h * es, C.data_ptr(), 2 * cw * es, 2 * cw * ch * es)
    out["surfaces_all_planes_fps"] = timed(lambda e: e.submit_surfaces(0, n, mk(RL, RC), mk(DL, DC)), eng)
with FeatureEngine(w, h, bit_depth=bpc, max_batch=16) as eng:
    out["planar_vmaf_only_fps"] = timed(lambda e: e.submit_resident(0, n, [R[0].data_ptr()], [D[0].data_ptr()], [w * es], [w * h * es]), eng)
    s = lambda L: FeatureEngine.surface_clip(fmt, L.data_ptr(), w * es, w * h * es)
    out["surfaces_vmaf_only_fps"] = timed(lambda e: e.submit_surfaces(0, n, s(RL), s(DL)), eng)
print(json.dumps(out))
