#!/usr/bin/env python3
"""Pin row a5 (FFmpeg `psnr` / `ssim` filters) against real FFmpeg output, the companion of compare_libvmaf_log.py.

The reference runs (app/vmaf_analyzer.py:1027-1045 and :1057-1075, input 0 = distorted, input 1 = reference):
    ffmpeg -i dist.y4m -i ref.y4m -lavfi "psnr=stats_file=psnr.txt" -f null -
    ffmpeg -i dist.y4m -i ref.y4m -lavfi "ssim=stats_file=ssim.txt" -f null -
Give this tool those stats files and the two clips:
    python tools/compare_ffmpeg_stats.py --psnr psnr.txt --ssim ssim.txt ref.y4m dist.y4m [--gpu]
It regenerates both files from the CPU restatement (oracle/vmaf_oracle.c: exact integer SSE, x264-style SSIM) -- and with
--gpu from the HIP kernels -- and diffs them line by line: PSNR lines must be IDENTICAL text (they are printed from exact
integers with %.2f), SSIM fields may differ in the last printed digit (%f of a float ratio).  Exit code 0 / 1."""
from __future__ import annotations

import argparse
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def our_lines(ref_path, dis_path, use_gpu):
    from oracle.oracle import Oracle
    from pqa2_amd import report
    from pqa2_amd.yuvio import open_video
    rd, dd = open_video(ref_path), open_video(dis_path)
    info = rd.info
    n = min(len(rd), len(dd))
    planes = 1 if info.mono else 3
    sizes = [(info.width, info.height)] + ([(info.chroma_w, info.chroma_h)] * 2 if planes == 3 else [])
    out = {}
    orc = Oracle("f32")
    sse = np.zeros((n, planes), np.uint64)
    ssim = np.zeros((n, planes))
    for i in range(n):
        r, d = rd.frame(i), dd.frame(i)
        for p in range(planes):
            sse[i, p] = orc.sse_plane(d[p], r[p], info.bit_depth)
            ssim[i, p] = orc.ssim_plane(d[p], r[p], info.bit_depth)
    out["CPU restatement (oracle/vmaf_oracle.c)"] = (report.psnr_stats_lines(sse, sizes, info.bit_depth),
                                                      report.ssim_stats_lines(ssim, sizes))
    if use_gpu:
        from pqa2_amd import _native as N
        from pqa2_amd.engine import FeatureEngine, sse_from_records
        with FeatureEngine(info.width, info.height, bit_depth=info.bit_depth, n_planes=planes,
                           chroma_shift=(info.hshift, info.vshift), features=N.FEAT_PSNR | N.FEAT_SSIM) as eng:
            for i in range(n):
                eng.submit(i, rd.frame(i)[:planes], dd.frame(i)[:planes])
            rec = eng.collect(0, n)
        out["HIP kernels (csrc/psnr_ssim.hip)"] = (report.psnr_stats_lines(sse_from_records(rec)[:, :planes], sizes, info.bit_depth),
                                                   report.ssim_stats_lines(rec[:, N.REC_SSIM:N.REC_SSIM + planes], sizes))
    return out


def numbers(line):
    return [float(x) if x not in ("inf", "-inf", "nan") else float(x) for x in re.findall(r"[:(]\s*(-?inf|nan|-?\d+\.?\d*(?:e[-+]?\d+)?)", line)]


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("reference")
    ap.add_argument("distorted")
    ap.add_argument("--psnr")
    ap.add_argument("--ssim")
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--ssim-tol", type=float, default=2e-6, help="largest allowed difference of a printed SSIM field")
    a = ap.parse_args(argv)
    if not a.psnr and not a.ssim:
        ap.error("give --psnr and / or --ssim")
    bad = False
    for tag, (pl, sl) in our_lines(a.reference, a.distorted, a.gpu).items():
        print(f"== {tag} ==")
        if a.psnr:
            theirs = [l.rstrip("\n") for l in open(a.psnr) if l.strip()]
            ok = len(theirs) == len(pl)
            first = None
            for i, (x, y) in enumerate(zip(theirs, pl)):
                if x.strip() != y.strip():
                    ok = False
                    first = first if first is not None else i
            print(f"psnr stats_file: {len(theirs)} lines, {'IDENTICAL text' if ok else 'MISMATCH'}")
            if not ok:
                bad = True
                if first is not None:
                    print(f"   first difference at line {first + 1}:\n      ffmpeg: {theirs[first]}\n      ours  : {pl[first]}")
                print("   -> check: per-plane SSE (exact integers), mse_avg weighting by plane area, %.2f rounding (pqa2_amd/report.py)")
        if a.ssim:
            theirs = [l.rstrip("\n") for l in open(a.ssim) if l.strip()]
            ok = len(theirs) == len(sl)
            worst = 0.0
            for x, y in zip(theirs, sl):
                nx, ny = numbers(x), numbers(y)
                if len(nx) != len(ny):
                    ok = False
                    continue
                for u, v in zip(nx[:-1], ny[:-1]):      # the dB figure in parentheses amplifies the last digit: skipped
                    if np.isfinite(u) and np.isfinite(v):
                        worst = max(worst, abs(u - v))
                    elif u != v:
                        ok = False
            ok = ok and worst <= a.ssim_tol
            print(f"ssim stats_file: {len(theirs)} lines, largest field difference {worst:.2e} (bar {a.ssim_tol:.0e}): {'ok' if ok else 'MISMATCH'}")
            if not ok:
                bad = True
                print("   -> check: 4x4 block sums, 8x8 windows on a 4-pixel grid, constants c1 = 416 / c2 = 235963 scaled by bit depth, "
                      "plane-area weighting of All (oracle/vmaf_oracle.c:524-532)")
    print("RESULT:", "MISMATCH" if bad else "FFmpeg's stats files and ours agree: row a5 is pinned for these clips")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
