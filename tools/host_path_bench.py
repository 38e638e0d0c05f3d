#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer path (pqa_submit): frames start in pageable host memory,
are packed into pinned staging by the library, uploaded on the copy stream and scored.  This is NOT
bench.py's `value` (that one starts with the clip resident in HBM); DESIGN.md quotes both."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="3840x2160")
ap.add_argument("--frames", type=int, default=96)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--chroma", action="store_true")
a = ap.parse_args()
w, h = map(int, a.size.split("x"))
import torch
from pqa2_amd import _native as N, synth_torch
from pqa2_amd.engine import FeatureEngine
clip = synth_torch.make_clip_cuda(w, h, a.frames, 8, chroma=a.chroma)
ref = [t.cpu().numpy() for t in clip["ref"]]
dis = [t.cpu().numpy() for t in clip["dis"]]
del clip; torch.cuda.empty_cache()
npl = 3 if a.chroma else 1
feats = N.FEAT_VMAF | ((N.FEAT_PSNR | N.FEAT_SSIM) if a.chroma else 0)
with FeatureEngine(w, h, n_planes=npl, features=feats, max_batch=a.batch) as eng:
    for rep in range(2):
        eng.reset()
        t0 = time.perf_counter()
        for i in range(a.frames):
            eng.submit(i, [p[i] for p in ref], [p[i] for p in dis])
        rec = eng.collect(0, a.frames)
        dt = time.perf_counter() - t0
bytes_per_frame = sum(p[0].nbytes for p in ref) * 2
print(json.dumps({"path": "host buffers -> pinned staging -> H2D -> kernels", "size": a.size, "frames": a.frames,
                  "planes": npl, "fps": round(a.frames / dt, 1), "host_to_device_GBps": round(a.frames * bytes_per_frame / dt / 1e9, 2)}))
