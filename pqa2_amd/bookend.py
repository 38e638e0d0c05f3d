"""White "bookend" frame detection from exact per-frame luma reductions.

The reference brackets captures with white frames and finds them by looping over decoded frames with
cv2/numpy: `np.mean(gray)`, `np.std(gray)`, `np.sum(gray > threshold) / gray.size`
(app/bookend_alignment.py:796-800, 902-913, 998-1020) and a "first 30 frames >= 85 % above 200" check
(app/reference_analyzer.py:112-151).  Here the three reductions come from one streaming HIP kernel
(`pqa_luma_stats_device`: sum, sum of squares, count above threshold -- exact integers) and the
reference's decision rules are applied to them on the host.  Gray is taken to be the luma plane.
"""
from __future__ import annotations

import numpy as np


def brightness_from_stats(stats: np.ndarray, n_pixels: int):
    """stats [n,3] uint64 {sum, sumsq, count_above} -> (mean, std, white_ratio) float64 arrays.
    std is numpy's population std: sqrt(E[x^2] - E[x]^2), evaluated exactly from the integer sums."""
    s = stats[:, 0].astype(np.float64)
    q = stats[:, 1].astype(np.float64)
    n = float(n_pixels)
    mean = s / n
    # (n*q - s^2) is an exact integer below 2^64 for planes up to 8K x 8K 8-bit; do it in Python ints
    var = np.array([(int(n_pixels) * int(qq) - int(ss) * int(ss)) for ss, qq in zip(stats[:, 0], stats[:, 1])],
                   dtype=np.float64) / (n * n)
    return mean, np.sqrt(np.maximum(var, 0.0)), stats[:, 2].astype(np.float64) / n


def is_white_initial(mean, std, whiteness_threshold: float, std_dev_threshold: float, threshold_idx: int):
    """Coarse pass (app/bookend_alignment.py:907-913)."""
    mean, std = np.asarray(mean), np.asarray(std)
    if threshold_idx < 2:
        return mean > whiteness_threshold
    return (mean > whiteness_threshold) & (std < std_dev_threshold)


def is_white_refined(mean, std, white_ratio, threshold: float, std_dev_threshold: float):
    """Frame-accurate pass (app/bookend_alignment.py:1003-1019); white_ratio = fraction of pixels > threshold."""
    mean, std, ratio = np.asarray(mean), np.asarray(std), np.asarray(white_ratio)
    uniform = std < std_dev_threshold * 1.2
    return np.where(uniform, mean > threshold * 0.95,
                    (mean > threshold) | ((mean > threshold * 0.9) & (ratio > 0.7)))


def starts_with_bookend(white_ratio_at_200, max_frames: int = 30) -> bool:
    """app/reference_analyzer.py:124-144: any of the first 30 frames with > 85 % of pixels above 200."""
    r = np.asarray(white_ratio_at_200)[:max_frames]
    return bool(np.any(r > 0.85))
