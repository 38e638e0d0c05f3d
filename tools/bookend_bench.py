#!/usr/bin/env python3
"""Measurement of the bookend row: (a) the luma-statistics kernel on a clip resident in HBM --
pure streaming, priced against the HBM roofline; (b) pqa2_amd.bookend.detect() end to end on a Y4M file (host frames ->
pinned staging -> kernel), frames/s of the whole detector.  Prints one JSON line.
usage: bookend_bench.py [--size 1920x1080] [--frames 300]"""
import argparse, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pqa2_amd import bookend, synth, yuvio
from pqa2_amd.engine import FeatureEngine

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="1920x1080")
ap.add_argument("--frames", type=int, default=300)
a = ap.parse_args()
w, h = map(int, a.size.split("x"))
n = a.frames
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
clip = torch.randint(16, 200, (n, h, w), dtype=torch.uint8, device="cuda", generator=gen)
clip[5:14] = 250; clip[n - 20:n - 8] = 250          # two white sections
torch.cuda.synchronize()
out = {"size": a.size, "frames": n}
with FeatureEngine(w, h, max_batch=64) as eng:
    eng.luma_stats_resident(clip.data_ptr(), w, w * h, n, 200)     # warm-up
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        st = eng.luma_stats_resident(clip.data_ptr(), w, w * h, n, 200)
    dt = (time.perf_counter() - t0) / reps
    out["resident"] = {"frames_per_s": round(n / dt, 1), "GB_per_s": round(n * w * h / dt / 1e9, 1), "hbm_frac": round(n * w * h / dt / 8e12, 4),
                       "what": "pqa_luma_stats_device: sum, sum of squares, count > threshold per frame; one read of the luma plane "
                               "(includes one D2H of 24 B per frame and a sync per 64-frame chunk)"}
    d = tempfile.mkdtemp(prefix="pqa_bk_")
    path = os.path.join(d, "clip.y4m")
    host = clip.cpu().numpy()
    yuvio.write_y4m(path, ([host[i]] for i in range(n)), synth.clip_info(w, h, 8, chroma=False, fps=30))
    rd = yuvio.open_video(path)
    bookend.detect(rd, eng)                                         # warm-up (page cache, staging buffers)
    t0 = time.perf_counter()
    found = bookend.detect(rd, eng)
    dt = time.perf_counter() - t0
    out["detect"] = {"clip_frames_per_s": round(n / dt, 1), "seconds": round(dt, 4),
                     "bookends": [(b["start_frame"], b["end_frame"]) for b in found],
                     "what": "bookend.detect on a Y4M file: brightness sampling + coarse scan + frame-accurate scan of the candidate "
                             "regions (the reference seeks and decodes the same frames with cv2, bookend_alignment.py:755-1133)"}
    t0 = time.perf_counter()
    cpu = bookend.detect(rd, stats_fn=bookend.numpy_stats_fn(rd, "auto"))
    out["detect_numpy"] = {"clip_frames_per_s": round(n / (time.perf_counter() - t0), 1), "same_result": cpu == found}
    os.remove(path); os.rmdir(d)
print(json.dumps(out))
