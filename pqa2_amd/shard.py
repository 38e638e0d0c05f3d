"""Frame-axis sharding across the GPUs of one node (one process per GPU, torch.distributed).

The path partitions by frames: VIF, ADM, PSNR, SSIM are per-frame; motion needs the previous
frame's reference luma -- the ONE-FRAME HALO each rank loads in front of its chunk; motion2 needs the
next frame's motion scalar, which falls out on the host once all records are gathered.  The only
exchange step is one all-gather of [frames_per_rank, 24] float64 records (RCCL over xGMI with the
nccl backend; ~192 B/frame, latency-bound, so a single direct all-gather -- no ring of per-frame
messages).  The reference has no counterpart: it runs one ffmpeg child with libvmaf's n_threads=4
(app/vmaf_analyzer.py:378).
"""
from __future__ import annotations

import numpy as np

RECORD_DOUBLES = 24


def shard_bounds(n_frames: int, world_size: int, rank: int):
    """Contiguous chunk [a, b) of rank `rank`; the first n_frames % world_size ranks get one extra."""
    base, rem = divmod(n_frames, world_size)
    a = rank * base + min(rank, rem)
    b = a + base + (1 if rank < rem else 0)
    return a, b


def all_bounds(n_frames: int, world_size: int):
    return [shard_bounds(n_frames, world_size, r) for r in range(world_size)]


def gather_records(local: np.ndarray, n_frames: int, world_size: int, rank: int, device=None,
                   force_collective: bool = False):
    """All-gather the per-rank record tiles into the full [n_frames, 24] array (on every rank).

    One collective: equal-sized tiles of ceil(n/world) rows (short ranks pad with zeros).  float64
    payload carries the uint64 SSE slots bit-exactly (no arithmetic touches them).
    `force_collective`: run the collective even at world_size 1 (a one-rank process group) -- how the RCCL leg is
    exercised on a one-GPU box."""
    if world_size == 1 and not force_collective:
        return np.asarray(local, np.float64).reshape(-1, RECORD_DOUBLES)
    import torch
    import torch.distributed as dist
    rows = -(-n_frames // world_size)
    tile = np.zeros((rows, RECORD_DOUBLES), np.float64)
    a, b = shard_bounds(n_frames, world_size, rank)
    assert local.shape[0] == b - a
    tile[: b - a] = local
    # bit-preserving transport: ship the 8-byte slots as int64 (NaN payloads / uint64 SSE survive)
    t = torch.from_numpy(tile.view(np.int64))
    if device is not None:
        t = t.to(device)
    out = torch.empty((world_size * rows, RECORD_DOUBLES), dtype=torch.int64, device=t.device)
    dist.all_gather_into_tensor(out, t)
    full = out.cpu().numpy().view(np.float64).reshape(world_size, rows, RECORD_DOUBLES)
    parts = []
    for r in range(world_size):
        ra, rb = shard_bounds(n_frames, world_size, r)
        parts.append(full[r, : rb - ra])
    return np.concatenate(parts, 0)


def score_sharded(n_frames: int, world_size: int, rank: int, feature_fn, device=None):
    """Run `feature_fn(first, last, halo_index)` on this rank's chunk and gather all records.

    feature_fn must return the [last-first, 24] records of frames [first, last); halo_index is
    first-1 (the reference frame to load in front of the chunk) or None for the rank that owns frame 0."""
    a, b = shard_bounds(n_frames, world_size, rank)
    local = feature_fn(a, b, a - 1 if a > 0 else None) if b > a else np.zeros((0, RECORD_DOUBLES))
    return gather_records(np.asarray(local, np.float64), n_frames, world_size, rank, device)


def gather_vector(local: np.ndarray, n_frames: int, world_size: int, rank: int, device=None) -> np.ndarray:
    """All-gather one float64 per frame (e.g. the per-frame scores each rank predicted for its own chunk)."""
    if world_size == 1:
        return np.asarray(local, np.float64)
    import torch
    import torch.distributed as dist
    rows = -(-n_frames // world_size)
    tile = np.zeros(rows, np.float64)
    a, b = shard_bounds(n_frames, world_size, rank)
    tile[: b - a] = local
    t = torch.from_numpy(tile)
    if device is not None:
        t = t.to(device)
    out = torch.empty(world_size * rows, dtype=torch.float64, device=t.device)
    dist.all_gather_into_tensor(out, t)
    full = out.cpu().numpy().reshape(world_size, rows)
    return np.concatenate([full[r, : shard_bounds(n_frames, world_size, r)[1] - shard_bounds(n_frames, world_size, r)[0]]
                           for r in range(world_size)])
