#!/usr/bin/env python3
"""How far are two builds of libpqa_vmaf.so from the f64 oracle (and how far is the f32 oracle itself)?  For changes that
trade arithmetic for speed: a build may move WITHIN the f32 oracle's own distance from f64, not beyond it.
usage: python tests/ab_vs_oracle.py A.so B.so      (test tooling like fuzz_parity.py: it calls the oracle; not collected by pytest)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle.oracle import Oracle
from pqa2_amd import _native as N, synth

CASES = [(3840, 2160, 8, 2, "natural"), (1920, 1080, 8, 3, "natural"), (1920, 1080, 8, 2, "flat"), (1280, 720, 10, 3, "natural"),
         (1920, 1080, 10, 2, "dark"), (1920, 1080, 12, 2, "natural"), (1280, 720, 12, 2, "dark"), (640, 360, 8, 4, "natural"),
         (200, 120, 8, 4, "natural")]


def clip(w, h, bpc, n, kind):
    refs, diss = synth.make_clip(w, h, n, bpc, chroma=False)
    if kind == "flat":      # nearly flat content far from mid-grey: the worst case for E[x^2] - mu^2
        rng = np.random.default_rng(7)
        refs = [[(np.full((h, w), 236, np.uint8) - (rng.random((h, w)) < 0.02)).astype(np.uint8)] for _ in range(n)]
        diss = [[(r[0] - (rng.random((h, w)) < 0.03)).astype(np.uint8)] for r in refs]
    if kind == "dark":
        refs = [[(r[0] // 8 + 64).astype(r[0].dtype)] for r in refs]
        diss = [[(d[0] // 8 + 64).astype(d[0].dtype)] for d in diss]
    return refs, diss


def run(path, w, h, bpc, refs, diss):
    lib = C.CDLL(os.path.abspath(path))
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.pqa_config_init.argtypes = [C.POINTER(N.PqaConfig), C.c_uint32, C.c_uint32]; lib.pqa_config_init.restype = None
    lib.pqa_create.argtypes = [C.POINTER(N.PqaConfig), C.POINTER(vp)]
    lib.pqa_destroy.argtypes = [vp]; lib.pqa_destroy.restype = None
    lib.pqa_submit.argtypes = [vp, i64, C.POINTER(vp * 3), C.POINTER(i64 * 3), C.POINTER(vp * 3), C.POINTER(i64 * 3)]
    lib.pqa_collect.argtypes = [vp, i64, i32, C.POINTER(C.c_double)]
    cfg = N.PqaConfig(); lib.pqa_config_init(C.byref(cfg), w, h)
    cfg.features = 1; cfg.bit_depth = bpc; cfg.n_planes = 1
    ctx = vp(); assert lib.pqa_create(C.byref(cfg), C.byref(ctx)) == 0
    n = len(refs)
    for i in range(n):
        r, d = np.ascontiguousarray(refs[i][0]), np.ascontiguousarray(diss[i][0])
        rp, dp = (vp * 3)(r.ctypes.data, None, None), (vp * 3)(d.ctypes.data, None, None)
        rs, ds = (i64 * 3)(r.strides[0], 0, 0), (i64 * 3)(d.strides[0], 0, 0)
        assert lib.pqa_submit(ctx, i, C.byref(rp), C.byref(rs), C.byref(dp), C.byref(ds)) == 0
    out = np.zeros((n, N.RECORD_DOUBLES))
    assert lib.pqa_collect(ctx, 0, n, out.ctypes.data_as(C.POINTER(C.c_double))) == 0
    lib.pqa_destroy(ctx)
    return out[:, :8]


o64, o32 = Oracle("f64"), Oracle("f32")
for w, h, bpc, n, kind in CASES:
    refs, diss = clip(w, h, bpc, n, kind)
    ry, dy = [r[0] for r in refs], [d[0] for d in diss]
    e64 = o64.clip_features_mt(ry, dy, bpc, threads=8)[:, :8]
    e32 = o32.clip_features_mt(ry, dy, bpc, threads=8)[:, :8]
    rel = lambda a: np.abs(a - e64) / np.maximum(np.abs(e64), 1e-30)
    line = f"{w}x{h} {bpc}-bit {kind:8s} f32 oracle vs f64: " + " ".join(f"{x:.1e}" for x in rel(e32).max(0)[:2]) + " |"
    for p in sys.argv[1:3]:
        g = run(p, w, h, bpc, refs, diss)
        r64, r32 = rel(g).max(0), (np.abs(g - e32) / np.maximum(np.abs(e32), 1e-30)).max(0)
        line += f" {os.path.basename(os.path.dirname(os.path.dirname(os.path.dirname(p)))) or 'work'}: vs f64 s0 {max(r64[0], r64[4]):.1e} s1 {max(r64[1], r64[5]):.1e}, vs f32 s0 {max(r32[0], r32[4]):.1e} s1 {max(r32[1], r32[5]):.1e} |"
    print(line, flush=True)
