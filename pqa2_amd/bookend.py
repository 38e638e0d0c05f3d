"""White "bookend" frame detection from exact per-frame luma reductions.

The reference brackets captures with white frames and finds them by looping over decoded frames with
cv2/numpy: `np.mean(gray)`, `np.std(gray)`, `np.sum(gray > threshold) / gray.size`
(app/bookend_alignment.py:796-800, 902-913, 998-1020) and a "first 30 frames >= 85 % above 200" check
(app/reference_analyzer.py:112-151).  Here the three reductions come from one streaming HIP kernel
(`pqa_luma_stats_device`: sum, sum of squares, count above threshold -- exact integers) and the
reference's decision rules are applied to them on the host.

What "gray" is (`gray=` of detect()).  The reference's gray is cv2.cvtColor(frame, COLOR_BGR2GRAY) of a frame that
cv2.VideoCapture has already converted to full-range BGR: for a limited-range (TV) clip, Y = 235 arrives as 255 and
Y = 190 as about 203, and the absolute constants of the rules (180 / 200 / 220 / 230 / 240) are in those units.
  "bt601_full": gray = clamp(round((Y - 16 s) * 255 / (219 s)), 0, 255), s = 2^(bpc - 8) -- the expansion, applied PER
                SAMPLE inside the reduction kernel (pqa_set_luma_gray), so clipping and rounding are the per-pixel ones
                and the statistics are exact integers of the 8-bit gray;
  "luma":       the luma samples as they are (scaled to 8-bit units) -- right for full-range clips;
  "auto":       "luma" when the container says full range (Y4M XCOLORRANGE=FULL), else "bt601_full".
Still parity-unpinned against cv2 itself (absent here): swscale's and cvtColor's own fixed-point roundings can move a
gray value by one step; BT.601 on both legs is what makes the chroma terms cancel (swscale's default for untagged input).
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


# ---- the detector: coarse scan -> regions -> frame-accurate scan (app/bookend_alignment.py:755-1133) -----------------
def expand_bt601_full(y: np.ndarray, bit_depth: int) -> np.ndarray:
    """Limited-range luma samples -> the 8-bit full-range gray of the module docstring, in exact integer arithmetic
    (floor(x + 0.5) rounding, saturated to 0..255): what csrc/luma_stats.hip applies per sample."""
    s = 1 << (bit_depth - 8)
    v = (y.astype(np.int64) - 16 * s) * 510 + 219 * s
    return np.clip(np.floor_divide(v, 438 * s), 0, 255).astype(np.uint64)


def resolve_gray(gray: str, info) -> str:
    if gray not in ("auto", "luma", "bt601_full"):
        raise ValueError(f"gray must be 'auto', 'luma' or 'bt601_full', not {gray!r}")
    if gray == "auto":
        return "luma" if getattr(info, "color_range", None) == "full" else "bt601_full"
    return gray


def numpy_stats_fn(reader, gray: str = "luma"):
    """CPU stand-in for the GPU reduction, for tests: the same three exact integers from numpy."""
    bpc = reader.info.bit_depth
    mode = resolve_gray(gray, reader.info)

    def fn(indices, threshold):
        out = np.zeros((len(indices), 3), np.uint64)
        for k, i in enumerate(indices):
            y = np.asarray(reader.frame(int(i))[0])
            y = expand_bt601_full(y, bpc) if mode == "bt601_full" else y.astype(np.uint64)
            out[k] = (y.sum(), (y * y).sum(), (y > np.uint64(threshold)).sum())
        return out
    return fn


def engine_stats_fn(reader, engine, chunk: int = 64, gray: str = "luma"):
    """Frames of `reader` -> FeatureEngine.luma_stats (pqa_luma_stats: pack, upload, one streaming HIP reduction)."""
    from . import _native as N
    mode = N.GRAY_BT601_FULL if resolve_gray(gray, reader.info) == "bt601_full" else N.GRAY_LUMA

    def fn(indices, threshold):
        # the gray mode is sticky context state: leave the caller's engine as it was found (a later luma_stats with a
        # luma-unit threshold would otherwise silently count the mapped 8-bit gray)
        prev = getattr(engine, "luma_gray", N.GRAY_LUMA)
        engine.set_luma_gray(mode)
        try:
            parts = []
            for a in range(0, len(indices), chunk):
                frames = [reader.frame(int(i))[0] for i in indices[a:a + chunk]]
                parts.append(engine.luma_stats(frames, threshold))
        finally:
            engine.set_luma_gray(prev)
        return np.concatenate(parts) if parts else np.zeros((0, 3), np.uint64)
    return fn


def detect(reader, engine=None, *, frame_sampling_rate: float = 5, adaptive_brightness: bool = True,
           white_threshold: float = 230, fallback_to_full_video: bool = True, stats_fn=None, gray: str = "auto"):
    """White bookend sections of a clip, the reference's `_detect_white_bookends` (app/bookend_alignment.py:755-1133)
    on exact GPU reductions: returns the list of dicts {start_frame, end_frame, start_time, end_time, frame_count,
    brightness, std_dev[, is_fallback]} sorted by start_frame (the caller pairs first/last like the reference does).

    `reader`: a pqa2_amd.yuvio reader (len, .frame(i) -> planes, .info).  `engine`: a FeatureEngine of the clip's
    geometry (its pqa_luma_stats does the reductions); or pass `stats_fn(indices, int_threshold) -> [n,3] uint64`.
    `gray`: "auto" | "luma" | "bt601_full" (module docstring).  With a caller-supplied `stats_fn` the function must
    already deliver statistics of that gray (numpy_stats_fn(reader, gray) / engine_stats_fn(reader, engine, gray=...)).
    In "luma" mode deeper samples are scaled to 8-bit units (the count threshold up); in "bt601_full" mode the kernel's
    gray is 8-bit for every depth.
    Structural differences from the reference that change no decision: the per-frame (mean, std) of the coarse pass are
    computed once and reused for all three thresholds (the reference re-decodes the clip per threshold); frames are
    fetched by index instead of cv2 seeks.  Against cv2's own gray the result is parity-unpinned (module docstring)."""
    info = reader.info
    fps = float(info.fps)
    frame_count = len(reader)
    if frame_count <= 0 or fps <= 0:
        return None
    duration = frame_count / fps
    n_pixels = info.width * info.height
    mode = resolve_gray(gray, info)
    scale = 1.0 if mode == "bt601_full" else float(1 << (info.bit_depth - 8))
    if stats_fn is None:
        if engine is None:
            raise ValueError("detect() needs a FeatureEngine (the reductions run on the GPU) or a stats_fn")
        stats_fn = engine_stats_fn(reader, engine, gray=mode)

    def brightness(indices, threshold8=255.0):
        thr = int(np.floor(max(0.0, float(threshold8)) * scale))   # gray > t  <=>  integer sample > floor(t * scale)
        st = stats_fn(list(indices), thr)
        mean, std, ratio = brightness_from_stats(st, n_pixels)
        return mean / scale, std / scale, ratio

    # 1. brightness samples across the clip (:775-805)
    sample_interval = max(1, int(fps / frame_sampling_rate))
    sample_idx = list(range(0, frame_count, sample_interval))
    s_mean, s_std, _ = brightness(sample_idx)
    if len(s_mean) == 0:
        return None
    avg_brightness, std_brightness = float(np.mean(s_mean)), float(np.std(s_mean))
    max_brightness, avg_std_dev = float(np.max(s_mean)), float(np.mean(s_std))

    # 2. thresholds (:818-852)
    if adaptive_brightness:
        dynamic = max(avg_brightness + 2.0 * std_brightness, max_brightness * 0.85, 180)
        if max_brightness > 240:
            dynamic = max(dynamic, 220)
        elif max_brightness < 200:
            dynamic = max(avg_brightness + 1.5 * std_brightness, 160)
        thresholds = [dynamic, dynamic * 0.9, max(avg_brightness + 20, 160)]
    else:
        thresholds = [white_threshold, white_threshold * 0.9, white_threshold * 0.8]

    min_white_frames = max(3, int(0.1 * fps)) if fps > 25 else 3          # :871-874
    initial_sample_rate = max(3, int(fps // 8))                           # :879
    std_dev_threshold = min(45, avg_std_dev * 1.8)                        # :882

    # 3. coarse pass (:885-935): one set of reductions, three threshold sweeps over it
    coarse_idx = list(range(0, frame_count, initial_sample_rate))
    c_mean, c_std, _ = brightness(coarse_idx)
    regions = []
    for t_idx, thr in enumerate(thresholds):
        white = is_white_initial(c_mean, c_std, thr, std_dev_threshold, t_idx)
        potential, cur = [], None
        frame_idx = 0
        for k, frame_idx in enumerate(coarse_idx):
            if white[k]:
                if cur is None:
                    cur = {"start_frame": max(0, frame_idx - initial_sample_rate), "brightness": float(c_mean[k])}
            elif cur is not None:
                cur["end_frame"] = min(frame_count - 1, frame_idx + initial_sample_rate)
                potential.append(cur)
                cur = None
        if cur is not None:
            cur["end_frame"] = min(frame_count - 1, frame_idx + initial_sample_rate)
            potential.append(cur)
        for r in potential:
            regions.append((max(0, r["start_frame"] - initial_sample_rate),
                            min(frame_count - 1, r["end_frame"] + initial_sample_rate), thr))
    if not regions:
        regions = [(0, frame_count - 1, thresholds[-1])]                  # :938-940
    if len(regions) > 1:                                                  # merge overlaps (:943-957)
        regions.sort()
        merged = []
        cs, ce, ct = regions[0]
        for s, e, t in regions[1:]:
            if s <= ce:
                ce, ct = max(ce, e), min(ct, t)
            else:
                merged.append((cs, ce, ct))
                cs, ce, ct = s, e, t
        merged.append((cs, ce, ct))
        regions = merged

    # 4. frame-accurate pass (:964-1063)
    all_bookends = []
    for start, end, thr in regions:
        if end - start < min_white_frames:
            continue
        idx = list(range(start, end + 1))
        mean, std, ratio = brightness(idx, thr)
        white = is_white_refined(mean, std, ratio, thr, std_dev_threshold)
        run, cur = 0, None
        for k, f in enumerate(idx):
            if white[k]:
                run += 1
                if cur is None:
                    cur = {"start_frame": f, "start_time": f / fps, "frame_count": 1,
                           "brightness": float(mean[k]), "std_dev": float(std[k])}
            elif cur is not None:
                cur["end_frame"], cur["end_time"], cur["frame_count"] = f - 1, (f - 1) / fps, run
                if run >= min_white_frames:
                    all_bookends.append(cur)
                cur, run = None, 0
        if cur is not None and run >= min_white_frames:
            cur["end_frame"], cur["end_time"], cur["frame_count"] = end, end / fps, run
            all_bookends.append(cur)

    # 5. de-duplicate, sort, fall back (:1066-1128)
    unique = []
    for b in all_bookends:
        dup = False
        for e in unique:
            if b["start_frame"] <= e["end_frame"] and b["end_frame"] >= e["start_frame"]:
                if b["frame_count"] > e["frame_count"] or b["brightness"] > e["brightness"]:
                    unique.remove(e)
                    unique.append(b)
                dup = True
                break
        if not dup:
            unique.append(b)
    bookends = sorted(unique, key=lambda x: x["start_frame"])
    if len(bookends) < 2 and fallback_to_full_video:
        bookends = [
            {"start_frame": 0, "end_frame": min(5, frame_count - 1), "start_time": 0, "end_time": min(5, frame_count - 1) / fps,
             "frame_count": min(5, frame_count), "brightness": 0, "std_dev": 0, "is_fallback": True},
            {"start_frame": max(0, frame_count - 5), "end_frame": frame_count - 1, "start_time": max(0, frame_count - 5) / fps,
             "end_time": duration, "frame_count": min(5, frame_count), "brightness": 0, "std_dev": 0, "is_fallback": True},
        ]
    return bookends


def content_span(bookends, fps: float):
    """First/last bookend pairing of the aligner (app/bookend_alignment.py:324-345): the content lies between the end of
    the first white section and the start of the last one, with a 1.5-frame buffer on both sides.  Returns
    (content_start_time, content_end_time) in seconds, or None when the timing is invalid."""
    if not bookends or len(bookends) < 2 or fps <= 0:
        return None
    first, last = bookends[0], bookends[-1]
    buf = 1.5 / fps
    a, b = first["end_time"] + buf, last["start_time"] - buf
    return (a, b) if b > a else None
