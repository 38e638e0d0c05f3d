#!/usr/bin/env python3
"""Interleaved A/B timing of two builds of libpqa_vmaf.so in ONE process on ONE device
(cdna_hip_programming.md rule 24: perf deltas come from interleaved rounds in one process; boxes differ by
several percent).  usage: ab_bench.py A.so B.so [--size 3840x2160] [--frames 96] [--rounds 7]
A library may carry environment switches that pqa_create reads once: path/to/lib.so@PQA_MULTI_STREAM=3,PQA_X=1 (set around
that library's pqa_create only), so two modes of ONE build can be timed against each other."""
import argparse, ctypes as C, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pqa2_amd import _native as N, synth_torch

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--size", default="3840x2160")
ap.add_argument("--frames", type=int, default=96)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--features", type=int, default=7, help="PQA_FEAT_* mask: 1 VIF, 2 ADM, 4 motion (isolate one chain)")
ap.add_argument("--bits", type=int, default=8)
ap.add_argument("--fixed", type=int, default=0, help="pqa_config.fixed_point mask")
a = ap.parse_args()
w, h = map(int, a.size.split("x"))
clip = synth_torch.make_clip_cuda(w, h, a.frames, a.bits)
R, D = clip["ref"][0], clip["dis"][0]
torch.cuda.synchronize()

def bind(spec):
    path, _, envs = spec.partition("@")
    kv = dict(e.split("=", 1) for e in envs.split(",") if e)
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update(kv)
    try:
        return _bind(path)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v

def _bind(path):
    lib = C.CDLL(os.path.abspath(path))
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.pqa_config_init.argtypes = [C.POINTER(N.PqaConfig), C.c_uint32, C.c_uint32]; lib.pqa_config_init.restype = None
    lib.pqa_create.argtypes = [C.POINTER(N.PqaConfig), C.POINTER(vp)]
    lib.pqa_destroy.argtypes = [vp]; lib.pqa_destroy.restype = None
    lib.pqa_submit_device.argtypes = [vp, i64, i32, C.POINTER(N.PqaDeviceClip), C.POINTER(N.PqaDeviceClip), vp, i64]
    lib.pqa_sync.argtypes = [vp]; lib.pqa_reset.argtypes = [vp]
    cfg = N.PqaConfig(); lib.pqa_config_init(C.byref(cfg), w, h); cfg.max_batch = a.batch; cfg.features = a.features; cfg.bit_depth = a.bits; cfg.fixed_point = a.fixed
    ctx = vp(); assert lib.pqa_create(C.byref(cfg), C.byref(ctx)) == 0
    return lib, ctx

ctxs = [bind(p) for p in a.libs]
r, d = N.PqaDeviceClip(), N.PqaDeviceClip()
r.plane[0], d.plane[0] = R.data_ptr(), D.data_ptr()
es = 1 if a.bits <= 8 else 2
r.row_pitch[0] = d.row_pitch[0] = w * es
r.frame_pitch[0] = d.frame_pitch[0] = w * h * es
def run(lib, ctx):
    lib.pqa_reset(ctx)
    t = time.perf_counter()
    assert lib.pqa_submit_device(ctx, 0, a.frames, C.byref(r), C.byref(d), None, 0) == 0
    assert lib.pqa_sync(ctx) == 0
    return time.perf_counter() - t
times = [[] for _ in ctxs]
for rnd in range(a.rounds + 1):
    for i, (lib, ctx) in enumerate(ctxs):
        dt = run(lib, ctx)
        if rnd:
            times[i].append(dt)
for p, t in zip(a.libs, times):
    print(f"{os.path.basename(p.partition('@')[0]) + ('@' + p.partition('@')[2] if '@' in p else ''):44s} median {a.frames / statistics.median(t):9.1f} fps   best {a.frames / min(t):9.1f} fps   ({len(t)} rounds)")
