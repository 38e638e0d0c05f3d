#!/usr/bin/env python3
"""End-to-end through the drop-in boundary: two Y4M files on disk -> VMAFAnalyzer.analyze_videos() ->
libvmaf-format JSON + psnr/ssim stats files (the reference's config C1/C2 flow).
Prints one JSON line per analysis -- pass 0 is the FIRST analysis of the process (it pins the staging buffers and
creates the context: what a one-shot caller pays), pass 1 the second (staging parked in the library, files in the page
cache) -- and one line with the measured host-to-device copy rate and the frames/s ceiling it implies for this frame size.
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
ap.add_argument("--chunk", type=int, default=50, help="frames generated and written at a time (bounds host memory)")
a = ap.parse_args()
w, h = map(int, a.size.split("x"))
os.makedirs(a.dir, exist_ok=True)
info = synth.clip_info(w, h, 8)
paths = {side: os.path.join(a.dir, f"{side}_{w}x{h}.y4m") for side in ("ref", "dis")}


def frames_of(side):   # chunk by chunk: a 300-frame 2160p 4:2:0 clip is 3.7 GB per file
    for c0 in range(0, a.frames, a.chunk):
        n = min(a.chunk, a.frames - c0)
        clip = synth_torch.make_clip_cuda(w, h, n, 8, chroma=True, t0=c0)
        planes = [t.cpu().numpy() for t in clip[side]]
        del clip
        for i in range(n):
            yield [p[i] for p in planes]


for side in ("ref", "dis"):
    yuvio.write_y4m(paths[side], frames_of(side), info)
torch.cuda.empty_cache()

# host -> device copy rate of this box (pinned source, 256 MiB, best of 3): the ceiling of any host-frame path
src = torch.empty(256 << 20, dtype=torch.uint8).pin_memory()
dst = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
best = 0.0
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    best = max(best, src.numel() / (time.perf_counter() - t0) / 1e9)
del src, dst
pair_bytes = 2 * info.frame_bytes
print(json.dumps({"h2d_GBps": round(best, 2), "bytes_per_frame_pair": pair_bytes,
                  "pcie_ceiling_fps": round(best * 1e9 / pair_bytes, 1)}), flush=True)

an = VMAFAnalyzer()
an.set_output_directory(a.dir)
an.set_test_name("e2e")
errs = []
an.error_occurred.connect(errs.append)
for rep in range(2):
    t0 = time.perf_counter()
    res = an.analyze_videos(paths["ref"], paths["dis"], a.model)
    dt = time.perf_counter() - t0
    assert res is not None, errs
    print(json.dumps({"pass": rep, "what": "first analysis of the process" if rep == 0 else "second analysis (staging parked, page cache warm)",
                      "size": a.size, "frames": a.frames, "seconds": round(dt, 3),
                      "fps_end_to_end": round(a.frames / dt, 1), "vmaf": round(res["vmaf_score"], 4),
                      "files": [os.path.basename(res["json_path"]), res["psnr_score"], res["ssim_score"]]}), flush=True)
