#!/usr/bin/env python3
"""One-command pinning of this repo's restatements against a real libvmaf run.

Nothing in this repo can run libvmaf (no ffmpeg / vmaf binary in the image, none on the GPU box), so parity with
libvmaf is UNPINNED.  Whoever has a box with a libvmaf-enabled ffmpeg can close that in two
commands, using the clips committed under tests/golden/clips/:

    ffmpeg -i tests/golden/clips/c352x288_8_dist.y4m -i tests/golden/clips/c352x288_8_ref.y4m \\
           -lavfi "libvmaf=log_fmt=json:log_path=lv.json:model=version=vmaf_v0.6.1:n_threads=4" -f null -
    python tools/compare_libvmaf_log.py lv.json tests/golden/clips/c352x288_8_ref.y4m tests/golden/clips/c352x288_8_dist.y4m

(the same filter line the reference builds, app/vmaf_analyzer.py:373-419: input 0 = distorted, input 1 = reference).
For the float extractors use model=version=vmaf_float_v0.6.1 (metric keys without the integer_ prefix); a `vmaf`
CLI JSON (--json, --feature float_vif ...) has the same per-frame schema and works as well.

For every per-frame metric of the log that this repo computes, the tool prints the largest absolute difference
against (a) the f32 restatement oracle/vmaf_oracle.c with the border rule the key implies, (b) the fixed-point
restatement oracle/vmaf_int_oracle.c for integer_* keys, (c) with --gpu, the HIP kernels (f32 path, and the
fixed-point mode for integer_* keys), and names the VERIFY items of the restatements each mismatch implicates.
Exit code 0 when every compared metric is inside its bar (the log prints %.6f: bar 2e-6 for exact restatements,
--tol for the f32 ones), 1 otherwise.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# metric family -> what a mismatch implicates (file:line of the VERIFY item in the restatements)
IMPLICATES = {
    "vif_scale0": ["float: border rule of vif_tools.c convolution (oracle/vmaf_oracle.c:47-57 mirror), 17-tap table "
                   "(:gaussian taps), vif_statistic_s branch order (:154-191)",
                   "integer: reflect-101 padding (oracle/vmaf_int_oracle.c:37 [VERIFY]), MAX(sigma2_sq,0) (:187 [VERIFY]), "
                   "Q formats / log2 LUT of the scale-0 path (:11 header)"],
    "vif_scale1": ["decimation phase: filter with THIS scale's kernel then keep even samples (oracle/vmaf_oracle.c:209-218)",
                   "integer: shifts of the deeper-scale path, rounding constants (oracle/vmaf_int_oracle.c:11)"],
    "vif_scale2": ["as vif_scale1"], "vif_scale3": ["as vif_scale1"],
    "adm2": ["sum-of-scales epilogue and the numden limit (pqa2_amd/model.py metrics_from_records)"],
    "adm_scale0": ["db2 taps / mirror (oracle/vmaf_oracle.c:241-342), Watson CSF factors, decouple cos^2(1deg) test, "
                   "3x3 CM with the extra centre weight (:397-418), crop (int)(w*0.1-0.5) (:347-353)",
                   "integer: range normalisation (oracle/vmaf_int_oracle.c:337 [VERIFY]), literal rfactor constants "
                   "(:533 [VERIFY]), div_lookup table, cube-accumulator shifts"],
    "adm_scale1": ["as adm_scale0 (int32 band path at scales 1-3 for integer_*)"],
    "adm_scale2": ["as adm_scale1"], "adm_scale3": ["as adm_scale1"],
    "motion": ["5-tap blur table and border; SAD normalisation (oracle/vmaf_oracle.c motion); integer: Q8 rounding "
               "after each pass (oracle/vmaf_int_oracle.c motion)"],
    "motion2": ["min(motion_i, motion_{i+1}) rule and frame-0 / last-frame handling (pqa2_amd/model.py:195-200)"],
    "vmaf": ["SVM predict / score clip (pqa2_amd/model.py:40-80) if every feature matches; otherwise a consequence"],
}
FAMILIES = ["vif_scale0", "vif_scale1", "vif_scale2", "vif_scale3", "adm2", "adm_scale0", "adm_scale1", "adm_scale2",
            "adm_scale3", "motion", "motion2"]


def load_log(path):
    with open(path) as f:
        d = json.load(f)
    frames = sorted(d["frames"], key=lambda fr: fr["frameNum"])
    keys = sorted({k for fr in frames for k in fr["metrics"]})
    cols = {k: np.array([fr["metrics"].get(k, np.nan) for fr in frames], np.float64) for k in keys}
    return [fr["frameNum"] for fr in frames], cols, d


def our_columns(ref_path, dis_path, model_name, integer_keys, use_gpu, frame_nums):
    """dict name -> {metric key -> per-frame array} for each restatement / kernel path that applies."""
    from oracle.oracle import Oracle
    from pqa2_amd import model as M
    from pqa2_amd.yuvio import open_video
    rd, dd = open_video(ref_path), open_video(dis_path)
    info = rd.info
    n = min(len(rd), len(dd))
    refs = [np.asarray(rd.frame(i)[0]) for i in range(n)]
    diss = [np.asarray(dd.frame(i)[0]) for i in range(n)]
    mdl = M.load_model(model_name)
    prefix = "integer_" if integer_keys else ""
    out = {}

    def cols_from(rec17, tag):
        rec = np.zeros((n, 24))
        rec[:, :17] = rec17
        m = M.score_frames(mdl, M.metrics_from_records(rec, info.width, info.height, prefix))
        out[tag] = {k: np.asarray(v)[frame_nums] for k, v in m.items()}

    orc = Oracle("f32")
    cols_from(orc.clip_features(refs, diss, info.bit_depth, vif_gain_limit=mdl.vif_enhn_gain_limit,
                                adm_gain_limit=mdl.adm_enhn_gain_limit, vif_border101=integer_keys),
              "f32 restatement (oracle/vmaf_oracle.c, %s border)" % ("integer_vif.c" if integer_keys else "vif_tools.c"))
    if integer_keys:
        from oracle.int_oracle import IntOracle
        cols_from(IntOracle().clip_features(refs, diss, info.bit_depth, mdl.vif_enhn_gain_limit, mdl.adm_enhn_gain_limit),
                  "fixed-point restatement (oracle/vmaf_int_oracle.c)")
    if use_gpu:
        from pqa2_amd import _native as N
        from pqa2_amd.engine import FeatureEngine
        for tag, fx in (("HIP f32 kernels", 0),) + ((("HIP fixed-point kernels", N.FIXED_ALL),) if integer_keys else ()):
            with FeatureEngine(info.width, info.height, bit_depth=info.bit_depth, vif_border=int(integer_keys),
                               vif_enhn_gain_limit=mdl.vif_enhn_gain_limit, adm_enhn_gain_limit=mdl.adm_enhn_gain_limit,
                               fixed_point=fx) as eng:
                for i in range(n):
                    eng.submit(i, [refs[i]], [diss[i]])
                cols_from(eng.collect(0, n)[:, :17], tag)
    return out, n


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("libvmaf_json")
    ap.add_argument("reference")
    ap.add_argument("distorted")
    ap.add_argument("--model", default=None, help="default: vmaf_v0.6.1 for integer_* logs, vmaf_float_v0.6.1 otherwise")
    ap.add_argument("--gpu", action="store_true", help="also run the HIP kernels (needs an MI355X)")
    ap.add_argument("--tol", type=float, default=2e-5, help="bar for the f32 paths on a feature (default 2e-5)")
    ap.add_argument("--vmaf-tol", type=float, default=0.01, help="bar on the vmaf score (north_star: 0.01)")
    a = ap.parse_args(argv)

    frame_nums, log, raw = load_log(a.libvmaf_json)
    integer_keys = any(k.startswith("integer_") for k in log)
    model = a.model or ("vmaf_v0.6.1" if integer_keys else "vmaf_float_v0.6.1")
    print(f"libvmaf log: version {raw.get('version', '?')}, {len(frame_nums)} frames, "
          f"{'integer_*' if integer_keys else 'float'} feature keys; model {model}")
    ours, n = our_columns(a.reference, a.distorted, model, integer_keys, a.gpu, frame_nums)
    if max(frame_nums, default=-1) >= n:
        print(f"error: the log has frame {max(frame_nums)} but the clips hold {n} frames", file=sys.stderr)
        return 2
    prefix = "integer_" if integer_keys else ""
    bad = False
    for tag, cols in ours.items():
        exact = tag.startswith("fixed-point") or tag.startswith("HIP fixed")
        print(f"\n== {tag} ==")
        print(f"{'metric':28s} {'max |ours - libvmaf|':>22s} {'at frame':>9s}   bar     verdict")
        for fam in FAMILIES + ["vmaf"]:
            key = fam if fam == "vmaf" else prefix + fam
            if key not in log or key not in cols:
                continue
            d = np.abs(cols[key] - log[key])
            j = int(np.nanargmax(d)) if d.size else 0
            bar = a.vmaf_tol if fam == "vmaf" else (2e-6 if exact else a.tol)
            ok = bool(np.nanmax(d) <= bar) if d.size else True
            print(f"{key:28s} {np.nanmax(d) if d.size else 0.0:22.3e} {frame_nums[j] if d.size else 0:9d}   {bar:.0e}   "
                  f"{'ok' if ok else 'MISMATCH'}")
            if not ok:
                bad = True
                for line in IMPLICATES.get(fam, []):
                    if ("integer:" in line) and not integer_keys:
                        continue
                    print(f"{'':28s}   -> check: {line}")
    missing = [prefix + f for f in FAMILIES if prefix + f not in log]
    if missing:
        print(f"\nnot in the log (not compared): {', '.join(missing)}")
    print("\nRESULT:", "MISMATCH -- see the implicated VERIFY items above" if bad else
          "every compared metric inside its bar: this log pins the restatement(s) listed above for these clips")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
