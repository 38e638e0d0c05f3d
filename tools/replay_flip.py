#!/usr/bin/env python3
"""Replay a case saved by tests/fuzz_parity.py (gpurun_out/fuzz_flip_*.npz: frames, tag, oracle outputs) through the
library -- PQA_LIB_PATH selects another build -- and print every feature's distance from the f64 / f32 oracle values
stored in the file and the VMAF difference.  usage: replay_flip.py case.npz [case2.npz ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pqa2_amd import _native as N, model as M
from pqa2_amd.engine import FeatureEngine

mdl = M.load_model("vmaf_v0.6.1")
def vmaf(rec17, w, h):
    full = np.zeros((rec17.shape[0], 24)); full[:, :17] = rec17
    return M.score_frames(mdl, M.metrics_from_records(full, w, h))["vmaf"]

print("library:", N.LIB_PATH)
for f in sys.argv[1:]:
    d = np.load(f)
    w, h, bpc, kind, gain, border = d["tag"]
    w, h, bpc, border = int(w), int(h), int(bpc), int(border)
    ref, dis = d["ref"], d["dis"]
    for mfma in ("1", "0"):
        os.environ["PQA_VIF_MFMA"] = mfma
        with FeatureEngine(w, h, bit_depth=bpc, vif_enhn_gain_limit=float(gain), adm_enhn_gain_limit=float(gain), vif_border=border) as eng:
            for i in range(ref.shape[0]):
                eng.submit(i, [ref[i]], [dis[i]])
            got = eng.collect(0, ref.shape[0])[:, :17]
        r64 = np.abs(got[:, :16] - d["exp"][:, :16]) / np.maximum(np.abs(d["exp"][:, :16]), 1e-9)
        r32 = np.abs(got[:, :16] - d["exp32"][:, :16]) / np.maximum(np.abs(d["exp32"][:, :16]), 1e-9)
        o = np.abs(d["exp32"][:, :16] - d["exp"][:, :16]) / np.maximum(np.abs(d["exp"][:, :16]), 1e-9)
        dv64 = np.abs(vmaf(got, w, h) - vmaf(d["exp"], w, h)).max(); dv32 = np.abs(vmaf(got, w, h) - vmaf(d["exp32"], w, h)).max()
        print(f"{os.path.basename(f)} {w}x{h} {bpc}-bit kind {int(kind)} gain {gain} border {border} PQA_VIF_MFMA={mfma}: "
              f"max rel vs f64 {r64.max():.2e} at {np.unravel_index(r64.argmax(), r64.shape)}, vs f32 {r32.max():.2e}, oracle f32-vs-f64 {o.max():.2e}; "
              f"|dVMAF| vs f64 {dv64:.5f} vs f32 {dv32:.5f}")
        print("   per-feature rel vs f64 (frame 0):", " ".join(f"{x:.1e}" for x in r64[0]))
