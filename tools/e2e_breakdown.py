#!/usr/bin/env python3
"""Where does a first analyze_videos() go?  Wall-clock stages of pipeline.score_files() on two Y4M files, first and second
analysis of the process.  usage: e2e_breakdown.py [--size 3840x2160] [--frames 300] [--dir DIR]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pqa2_amd import synth, synth_torch, yuvio, _native as N, model as M, pipeline
from pqa2_amd.engine import FeatureEngine

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="3840x2160"); ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--dir", default="/tmp/pqa_e2e_bd"); ap.add_argument("--chunk", type=int, default=50)
a = ap.parse_args()
w, h = map(int, a.size.split("x"))
os.makedirs(a.dir, exist_ok=True)
info = synth.clip_info(w, h, 8)
paths = {s: os.path.join(a.dir, f"{s}_{w}x{h}.y4m") for s in ("ref", "dis")}
def frames_of(side):
    for c0 in range(0, a.frames, a.chunk):
        n = min(a.chunk, a.frames - c0)
        clip = synth_torch.make_clip_cuda(w, h, n, 8, chroma=True, t0=c0)
        planes = [t.cpu().numpy() for t in clip[side]]
        for i in range(n):
            yield [p[i] for p in planes]
for s in ("ref", "dis"):
    if not os.path.exists(paths[s]):
        yuvio.write_y4m(paths[s], frames_of(s), info)
torch.cuda.empty_cache(); torch.cuda.synchronize()

for rep in range(2):
    T = {}; t0 = time.perf_counter()
    def lap(name, _t=[t0]):
        now = time.perf_counter(); T[name] = round(1e3 * (now - _t[0]), 2); _t[0] = now
    rr, dr = yuvio.open_video(paths["ref"]), yuvio.open_video(paths["dis"]); lap("open_clips")
    mdl = M.load_model("vmaf_v0.6.1"); lap("load_model")
    eng = FeatureEngine(w, h, bit_depth=8, n_planes=3, chroma_shift=(1, 1), features=N.FEAT_VMAF | N.FEAT_PSNR | N.FEAT_SSIM,
                        result_capacity=max(a.frames, 16), vif_border=mdl.vif_border); lap("create_context")
    eng.submit_file(0, rr.fileno(), rr.plane_offsets(0), dr.fileno(), dr.plane_offsets(0)); lap("first_submit (staging)")
    for i in range(1, a.frames):
        eng.submit_file(i, rr.fileno(), rr.plane_offsets(i), dr.fileno(), dr.plane_offsets(i))
    lap("other_submits")
    rec = eng.collect(0, a.frames); lap("collect")
    eng.close(); lap("close")
    res = pipeline.finish_records(rec, mdl, rr.info, psnr=True, ssim=True, n_planes=3); lap("host_epilogue")
    T["total_ms"] = round(1e3 * (time.perf_counter() - t0), 2)
    print(json.dumps({"analysis": rep, "size": a.size, "frames": a.frames, **T}), flush=True)
