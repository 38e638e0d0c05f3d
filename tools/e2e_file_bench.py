#!/usr/bin/env python3
"""End-to-end through the drop-in boundary: two Y4M files on disk -> VMAFAnalyzer.analyze_videos() ->
libvmaf-format JSON + psnr/ssim stats files (the reference's config C1/C2 flow).
usage: e2e_file_bench.py [--size 1920x1080] [--frames 300] [--dir /tmp/pqa_e2e]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pqa2_amd import synth, synth_torch, yuvio
from pqa2_amd.vmaf_analyzer import VMAFAnalyzer

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="1920x1080")
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--dir", default="/tmp/pqa_e2e")
ap.add_argument("--model", default="vmaf_v0.6.1")
a = ap.parse_args()
w, h = map(int, a.size.split("x"))
os.makedirs(a.dir, exist_ok=True)
clip = synth_torch.make_clip_cuda(w, h, a.frames, 8, chroma=True)
info = synth.clip_info(w, h, 8)
paths = {}
for side in ("ref", "dis"):
    planes = [t.cpu().numpy() for t in clip[side]]
    paths[side] = os.path.join(a.dir, f"{side}_{w}x{h}.y4m")
    yuvio.write_y4m(paths[side], ([p[i] for p in planes] for i in range(a.frames)), info)
del clip
torch.cuda.empty_cache()
an = VMAFAnalyzer()
an.set_output_directory(a.dir)
an.set_test_name("e2e")
errs = []
an.error_occurred.connect(errs.append)
for rep in range(2):   # second pass: files in the page cache
    t0 = time.perf_counter()
    res = an.analyze_videos(paths["ref"], paths["dis"], a.model)
    dt = time.perf_counter() - t0
    assert res is not None, errs
    print(json.dumps({"pass": rep, "size": a.size, "frames": a.frames, "seconds": round(dt, 3),
                      "fps_end_to_end": round(a.frames / dt, 1), "vmaf": round(res["vmaf_score"], 4),
                      "files": [os.path.basename(res["json_path"]), res["psnr_score"], res["ssim_score"]]}))
