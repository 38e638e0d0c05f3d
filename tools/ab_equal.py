#!/usr/bin/env python3
"""Are two builds of libpqa_vmaf.so BIT-identical on the same clips?  (for changes that must not move a single result:
work skipped because its contribution is zero by definition, address arithmetic, launch shape)
usage: ab_equal.py A.so B.so [--features 7] [--fixed 7]      exit code 1 when any record differs."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pqa2_amd import _native as N, synth_torch

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs=2)
ap.add_argument("--features", type=int, default=7)
ap.add_argument("--fixed", type=int, default=0, help="pqa_config.fixed_point mask (7 = every extractor in fixed point)")
a = ap.parse_args()
CASES = [(3840, 2160, 8, 6), (1920, 1080, 8, 6), (1039, 913, 8, 4), (200, 120, 8, 4), (64, 48, 8, 3), (3840, 2160, 10, 4),
         (1280, 720, 10, 4), (720, 486, 12, 3)]


def run(spec, w, h, bits, n, R, D):
    # lib.so@PQA_X=1,PQA_Y=2: switches pqa_create reads once, set around this library's context only
    path, _, envs = spec.partition("@")
    kv = dict(e.split("=", 1) for e in envs.split(",") if e)
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update(kv)
    try:
        return _run(path, w, h, bits, n, R, D)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


def _run(path, w, h, bits, n, R, D):
    lib = C.CDLL(os.path.abspath(path))
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.pqa_config_init.argtypes = [C.POINTER(N.PqaConfig), C.c_uint32, C.c_uint32]; lib.pqa_config_init.restype = None
    lib.pqa_create.argtypes = [C.POINTER(N.PqaConfig), C.POINTER(vp)]
    lib.pqa_destroy.argtypes = [vp]; lib.pqa_destroy.restype = None
    lib.pqa_submit_device.argtypes = [vp, i64, i32, C.POINTER(N.PqaDeviceClip), C.POINTER(N.PqaDeviceClip), vp, i64]
    lib.pqa_collect.argtypes = [vp, i64, i32, C.POINTER(C.c_double)]
    cfg = N.PqaConfig(); lib.pqa_config_init(C.byref(cfg), w, h)
    cfg.max_batch = 4; cfg.features = a.features; cfg.bit_depth = bits; cfg.n_planes = 1; cfg.fixed_point = a.fixed
    ctx = vp(); assert lib.pqa_create(C.byref(cfg), C.byref(ctx)) == 0
    r, d = N.PqaDeviceClip(), N.PqaDeviceClip()
    es = 1 if bits <= 8 else 2
    r.plane[0], d.plane[0] = R.data_ptr(), D.data_ptr()
    r.row_pitch[0] = d.row_pitch[0] = w * es
    r.frame_pitch[0] = d.frame_pitch[0] = w * h * es
    assert lib.pqa_submit_device(ctx, 0, n, C.byref(r), C.byref(d), None, 0) == 0
    rec = np.zeros((n, 24))
    assert lib.pqa_collect(ctx, 0, n, rec.ctypes.data_as(C.POINTER(C.c_double))) == 0
    lib.pqa_destroy(ctx)
    return rec


bad = 0
for w, h, bits, n in CASES:
    clip = synth_torch.make_clip_cuda(w, h, n, bits)
    R, D = clip["ref"][0], clip["dis"][0]
    torch.cuda.synchronize()
    ra, rb = run(a.libs[0], w, h, bits, n, R, D), run(a.libs[1], w, h, bits, n, R, D)
    same = np.array_equal(ra.view(np.uint64), rb.view(np.uint64))
    worst = float(np.max(np.abs(ra - rb) / np.maximum(np.abs(ra), 1e-30)))
    print(f"{w}x{h} {bits}-bit, {n} frames: {'bit-identical' if same else 'DIFFERENT (max rel %.3e)' % worst}")
    bad += not same
sys.exit(1 if bad else 0)
