"""On-disk artefacts the reference reads back after the three ffmpeg passes:
the libvmaf `log_fmt=json` log (app/vmaf_analyzer.py:374-375, parsed at :638-690 and by
app/ui/tabs/results_tab.py:3000-3028, app/report_generator.py:288-311) and FFmpeg's psnr/ssim
`stats_file` text logs (app/vmaf_analyzer.py:1032,1062).
"""
from __future__ import annotations

import math

import numpy as np

from .model import pool

ENGINE_VERSION = "pqa2_amd-0.2.0"


def _f6(x: float) -> str:
    x = float(x)
    if math.isnan(x):
        return "nan"
    if math.isinf(x):
        return "inf" if x > 0 else "-inf"
    return f"{x:.6f}"


def build_vmaf_log(metrics: dict, fps: float, frame_indices=None, extra_top: dict | None = None) -> dict:
    """dict with libvmaf's JSON schema: version, fps, frames[{frameNum, metrics}], pooled_metrics,
    aggregate_metrics.  Values are rounded to 6 decimals like libvmaf's %.6f writer."""
    keys = list(metrics.keys())
    n = len(next(iter(metrics.values()))) if metrics else 0
    idx = list(range(n)) if frame_indices is None else list(frame_indices)
    frames = []
    for j in range(n):
        frames.append({"frameNum": int(idx[j]),
                       "metrics": {k: float(_f6(metrics[k][j])) for k in keys}})
    pooled = {k: {a: float(_f6(b)) for a, b in pool(metrics[k]).items()} for k in keys}
    log = {"version": ENGINE_VERSION, "fps": float(f"{fps:.2f}"), "frames": frames,
           "pooled_metrics": pooled, "aggregate_metrics": {}}
    if extra_top:
        log.update(extra_top)
    return log


def write_vmaf_json(path: str, log: dict) -> None:
    """Hand-rolled writer so numbers keep libvmaf's fixed 6-decimal form (json.dump would drop zeros)."""
    def num(v):
        return _f6(v) if isinstance(v, float) else str(v)
    with open(path, "w") as f:
        f.write("{\n")
        f.write(f'  "version": "{log["version"]}",\n')
        for k, v in log.items():
            if k in ("version", "fps", "frames", "pooled_metrics", "aggregate_metrics"):
                continue
            f.write(f'  "{k}": "{v}",\n' if isinstance(v, str) else f'  "{k}": {num(v)},\n')
        f.write(f'  "fps": {log["fps"]:.2f},\n')
        f.write('  "frames": [')
        for i, fr in enumerate(log["frames"]):
            f.write("\n    {\n")
            f.write(f'      "frameNum": {fr["frameNum"]},\n      "metrics": {{\n')
            items = list(fr["metrics"].items())
            for j, (k, v) in enumerate(items):
                f.write(f'        "{k}": {_f6(v)}{"," if j + 1 < len(items) else ""}\n')
            f.write("      }\n    }" + ("," if i + 1 < len(log["frames"]) else ""))
        f.write("\n  ],\n")
        f.write('  "pooled_metrics": {')
        pitems = list(log["pooled_metrics"].items())
        for i, (k, d) in enumerate(pitems):
            f.write(f'\n    "{k}": {{\n')
            ditems = list(d.items())
            for j, (a, b) in enumerate(ditems):
                f.write(f'      "{a}": {_f6(b)}{"," if j + 1 < len(ditems) else ""}\n')
            f.write("    }" + ("," if i + 1 < len(pitems) else ""))
        f.write("\n  },\n")
        f.write('  "aggregate_metrics": {\n  }\n}\n')


def _psnr(mse: float, peak: float) -> float:
    return float("inf") if mse <= 0 else 10.0 * math.log10(peak * peak / mse)


def _fmt2(x: float) -> str:
    return "inf" if math.isinf(x) else f"{x:0.2f}"


def psnr_stats_lines(sse: np.ndarray, plane_sizes, bit_depth: int, comps="yuv"):
    """FFmpeg vf_psnr.c stats_file lines from exact per-plane SSE.  sse: [n, planes] uint64."""
    peak = float((1 << bit_depth) - 1)
    sizes = np.array([w * h for (w, h) in plane_sizes], np.float64)
    weights = sizes / sizes.sum()
    lines = []
    for i in range(sse.shape[0]):
        comp_mse = [float(sse[i, p]) / sizes[p] for p in range(len(sizes))]
        mse = float(sum(comp_mse[p] * weights[p] for p in range(len(sizes))))
        parts = [f"n:{i + 1}", f"mse_avg:{mse:0.2f}"]
        parts += [f"mse_{comps[p]}:{comp_mse[p]:0.2f}" for p in range(len(sizes))]
        parts.append(f"psnr_avg:{_fmt2(_psnr(mse, peak))}")
        parts += [f"psnr_{comps[p]}:{_fmt2(_psnr(comp_mse[p], peak))}" for p in range(len(sizes))]
        lines.append(" ".join(parts) + " ")
    return lines


def psnr_values(sse: np.ndarray, plane_sizes, bit_depth: int):
    """per-frame psnr_y (dB; inf when identical) and the plane-weighted average."""
    peak = float((1 << bit_depth) - 1)
    sizes = np.array([w * h for (w, h) in plane_sizes], np.float64)
    mse_p = sse.astype(np.float64) / sizes[None, :]
    mse_avg = (mse_p * (sizes / sizes.sum())[None, :]).sum(1)
    with np.errstate(divide="ignore"):
        psnr_p = np.where(mse_p > 0, 10.0 * np.log10(peak * peak / np.where(mse_p > 0, mse_p, 1.0)), np.inf)
        psnr_avg = np.where(mse_avg > 0, 10.0 * np.log10(peak * peak / np.where(mse_avg > 0, mse_avg, 1.0)), np.inf)
    return psnr_p, psnr_avg


def ssim_all(ssim: np.ndarray, plane_sizes) -> np.ndarray:
    sizes = np.array([w * h for (w, h) in plane_sizes], np.float64)
    return (ssim * (sizes / sizes.sum())[None, :]).sum(1)


def ssim_stats_lines(ssim: np.ndarray, plane_sizes, comps="YUV"):
    """FFmpeg vf_ssim.c stats_file lines.  ssim: [n, planes]."""
    allv = ssim_all(ssim, plane_sizes)
    lines = []
    for i in range(ssim.shape[0]):
        parts = [f"n:{i + 1}"] + [f"{comps[p]}:{ssim[i, p]:f}" for p in range(ssim.shape[1])]
        db = float("inf") if allv[i] >= 1.0 else -10.0 * math.log10(1.0 - allv[i])
        parts.append(f"All:{allv[i]:f} ({'inf' if math.isinf(db) else f'{db:f}'})")
        lines.append(" ".join(parts))
    return lines
