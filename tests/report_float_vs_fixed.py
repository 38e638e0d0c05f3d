#!/usr/bin/env python3
"""Prints the table in DESIGN.md "float vs fixed-point": how far the float extractors (what the HIP kernels
compute, oracle/vmaf_oracle.c) sit from the fixed-point restatement (oracle/vmaf_int_oracle.c) of the
extractors the default models name, with vif_tools.c's border and with integer_vif.c's padding.
CPU only; not collected by pytest.   usage: python tests/report_float_vs_fixed.py [WxH:frames ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle.int_oracle import IntOracle
from oracle.oracle import Oracle, finish_features
from pqa2_amd import model as M
from pqa2_amd import synth

cases = sys.argv[1:] or ["352x288:6", "640x360:6", "1920x1080:3", "3840x2160:2"]
mdl = {False: M.load_model("vmaf_v0.6.1"), True: M.load_model("vmaf_4k_v0.6.1")}
into, flo = IntOracle(), Oracle("f32")


def vmaf(rec, w, h):
    full = np.zeros((rec.shape[0], 24))
    full[:, :17] = rec
    return M.score_frames(mdl[w >= 3840], M.metrics_from_records(full, w, h))["vmaf"]


print("| size | VIF border | max rel vif_scale0..3 | max rel adm2 | max rel motion | max abs dVMAF | mean dVMAF |")
print("|---|---|---|---|---|---|---|")
for c in cases:
    wh, n = c.split(":")
    w, h = map(int, wh.split("x"))
    refs, diss = synth.make_clip(w, h, int(n), 8, chroma=False)
    R, D = [r[0] for r in refs], [d[0] for d in diss]
    fi = into.clip_features(R, D, 8)
    a = finish_features(fi, w, h)
    for b101 in (False, True):
        ff = flo.clip_features_mt(R, D, 8, threads=4, vif_border101=b101)
        b = finish_features(ff, w, h)
        rel = lambda k: np.max(np.abs(a[k] - b[k]) / np.maximum(np.abs(b[k]), 1e-9))
        dv = vmaf(ff, w, h) - vmaf(fi, w, h)
        print(f"| {w}x{h} | {'integer_vif.c (101)' if b101 else 'vif_tools.c'} | "
              f"{' '.join('%.1e' % rel(f'vif_scale{s}') for s in range(4))} | {rel('adm2'):.1e} | "
              f"{np.max(np.abs(a['motion'][1:] - b['motion'][1:]) / b['motion'][1:]):.1e} | {np.max(np.abs(dv)):.4f} | {np.mean(dv):+.4f} |")
