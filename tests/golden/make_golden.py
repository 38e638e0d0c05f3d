#!/usr/bin/env python3
"""Generates tests/golden/*.  The reference (yoseph007/PQA2) holds no golden vectors for this path
(no tests, no media, arithmetic in an external ffmpeg+libvmaf binary), so these fixtures are produced
by the f64 oracle (oracle/vmaf_oracle.c, cross-checked by oracle/np_restatement.py) on seeded
synthetic frames: regression anchors, NOT libvmaf outputs ("parity unpinned").

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.int_oracle import IntOracle  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
from pqa2_amd import model as M, synth  # noqa: E402

CASES = [("c64x48_8", 64, 48, 8, 3), ("c176x144_8", 176, 144, 8, 3), ("c321x241_8", 321, 241, 8, 2),
         ("c200x120_10", 200, 120, 10, 2), ("c352x288_8", 352, 288, 8, 3)]
# clips also committed as .y4m under clips/: the inputs of tools/compare_libvmaf_log.py (run ffmpeg+libvmaf on them
# wherever one exists, diff its JSON log against every restatement here)
Y4M_CASES = ("c64x48_8", "c352x288_8", "c200x120_10")


def main():
    o64 = Oracle("f64")
    into = IntOracle()
    out = {"note": "f64 oracle on pqa2_amd.synth.make_clip(w,h,n,bpc,chroma=True); records = vif num[4], vif den[4], "
                   "adm num[4], adm den[4], motion; sse/ssim per plane (FFmpeg psnr/ssim definitions); "
                   "records_fixed_point = the same layout from oracle/vmaf_int_oracle.c (integer arithmetic: exact)",
           "cases": {}}
    mdl = M.load_model("vmaf_v0.6.1")
    for name, w, h, bpc, n in CASES:
        refs, diss = synth.make_clip(w, h, n, bpc, chroma=True)
        sha = hashlib.sha256()
        for fr in refs + diss:
            for p in fr:
                sha.update(np.ascontiguousarray(p).tobytes())
        rec = o64.clip_features([r[0] for r in refs], [d[0] for d in diss], bpc)
        full = np.zeros((n, 24)); full[:, :17] = rec
        vm = M.score_frames(mdl, M.metrics_from_records(full, w, h, "integer_"))["vmaf"]
        sse = [[o64.sse_plane(diss[i][p], refs[i][p], bpc) for p in range(3)] for i in range(n)]
        ssim = [[o64.ssim_plane(diss[i][p], refs[i][p], bpc) for p in range(3)] for i in range(n)]
        rec_fx = into.clip_features([r[0] for r in refs], [d[0] for d in diss], bpc)
        out["cases"][name] = {"w": w, "h": h, "bpc": bpc, "n": n, "input_sha256": sha.hexdigest(),
                              "records": rec.tolist(), "vmaf_v0.6.1": vm.tolist(), "sse": sse, "ssim": ssim,
                              "records_fixed_point": rec_fx.tolist()}
        if name in Y4M_CASES:
            from pqa2_amd import yuvio
            os.makedirs(os.path.join(HERE, "clips"), exist_ok=True)
            info = synth.clip_info(w, h, bpc)
            yuvio.write_y4m(os.path.join(HERE, "clips", f"{name}_ref.y4m"), refs, info)
            yuvio.write_y4m(os.path.join(HERE, "clips", f"{name}_dist.y4m"), diss, info)
        if name == "c64x48_8":  # ship the actual bytes of the smallest case
            np.savez_compressed(os.path.join(HERE, "c64x48_8_frames.npz"),
                                **{f"ref{i}_{p}": refs[i][p] for i in range(n) for p in range(3)},
                                **{f"dis{i}_{p}": diss[i][p] for i in range(n) for p in range(3)})
    with open(os.path.join(HERE, "golden_features.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", os.path.join(HERE, "golden_features.json"))


if __name__ == "__main__":
    main()
