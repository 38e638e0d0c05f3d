"""File pair -> per-frame metrics.  The part of the reference that lived inside the three ffmpeg
children (app/vmaf_analyzer.py:446, :1037, :1067): read both clips, run the extractors on every
frame pair, gather, apply the model.  Works as a single process (world_size 1) or as one rank of a
frame-sharded torch.distributed job (one process per GPU)."""
from __future__ import annotations

import os
import time

import numpy as np

from . import _native as N
from . import model as M
from . import report, shard
from .yuvio import open_video


class ScoreResult(dict):
    """keys: metrics (ordered dict name -> per-frame array), frame_indices, records, info, fps,
    psnr_lines, ssim_lines, model_name"""


def _cap(v: np.ndarray, cap: float) -> np.ndarray:
    return np.minimum(v, cap)


def score_files(reference_path: str, distorted_path: str, model: str | None = "vmaf_v0.6.1", *,
                psnr: bool = True, ssim: bool = True, n_subsample: int = 1, device: int = 0,
                rank: int = 0, world_size: int = 1, gather_device=None, max_batch: int = 0,
                progress=None, cancelled=None, engine_factory=None, raw_kwargs=None,
                fixed_point: int = 0) -> ScoreResult | None:
    """Returns the ScoreResult on rank 0 (None on other ranks).  `progress(frames_done, frames_total)`
    is called as frames are submitted; `cancelled()` is polled between frames (True -> PqaCancelled).
    `fixed_point`: PQA_FIXED_* mask -- extractors to run in libvmaf's fixed-point arithmetic (include/pqa_vmaf.h)."""
    from .engine import FeatureEngine
    raw_kwargs = raw_kwargs or {}
    ref_rd = open_video(reference_path, **raw_kwargs)
    dis_rd = open_video(distorted_path, **raw_kwargs)
    ri, di = ref_rd.info, dis_rd.info
    if (ri.width, ri.height) != (di.width, di.height):
        raise ValueError(f"reference is {ri.width}x{ri.height} but distorted is {di.width}x{di.height}")
    if ri.bit_depth != di.bit_depth or (ri.hshift, ri.vshift, ri.mono) != (di.hshift, di.vshift, di.mono):
        raise ValueError("reference and distorted clips differ in pixel format")
    n = min(len(ref_rd), len(dis_rd))
    if n <= 0:
        raise ValueError("no frames to analyse")
    mdl = M.load_model(model)
    side = (psnr or ssim)
    n_planes = 1 if (ri.mono or not side) else 3
    feats = N.FEAT_VMAF | (N.FEAT_PSNR if psnr else 0) | (N.FEAT_SSIM if ssim else 0)
    a, b = shard.shard_bounds(n, world_size, rank)
    t_start = time.perf_counter()

    make = engine_factory or (lambda *aa, **kw: FeatureEngine(*aa, reuse=True, **kw))   # a parked context of this configuration
    eng = make(ri.width, ri.height, bit_depth=ri.bit_depth, n_planes=n_planes,
               chroma_shift=(ri.hshift, ri.vshift), features=feats, device=device, max_batch=max_batch,
               result_capacity=max(b - a, 16), n_subsample=n_subsample,
               vif_enhn_gain_limit=mdl.vif_enhn_gain_limit, adm_enhn_gain_limit=mdl.adm_enhn_gain_limit,
               vif_border=mdl.vif_border, **({"fixed_point": int(fixed_point)} if fixed_point else {}))
    try:
        if a > 0:
            eng.set_motion_halo(ref_rd.frame(a - 1)[0])   # one-frame halo in front of this rank's chunk
        # both clips are files of packed planes (.y4m): the library reads them straight into its pinned staging (pqa_submit_fd:
        # one copy, no page faults) instead of copying frames out of the readers' mappings
        by_fd = all(hasattr(r, "fileno") and hasattr(r, "plane_offsets") for r in (ref_rd, dis_rd)) and hasattr(eng, "submit_file")
        # ... and a run of frames that lie equally spaced in both files goes down in ONE call (pqa_submit_fd_run: reading frame
        # k + 1 overlaps the upload of frame k inside the library); 8 = a staging half, and the grain of progress / cancel
        by_run = (by_fd and hasattr(eng, "submit_file_run") and all(hasattr(r, "run_stride") for r in (ref_rd, dis_rd))
                  and os.environ.get("PQA_FD_RUN", "1") != "0")   # PQA_FD_RUN=0: frame by frame (A/B partner)
        i = a
        while i < b:
            if cancelled is not None and cancelled():
                eng.cancel()
                raise N.PqaCancelled(N.PQA_ECANCELLED, "cancelled")
            m = 1
            if by_run:
                m = min(8, b - i)
                rs, ds = ref_rd.run_stride(i, m), dis_rd.run_stride(i, m)
                if rs is None or ds is None:
                    m = 1
            if m > 1:
                eng.submit_file_run(i, m, ref_rd.fileno(), ref_rd.plane_offsets(i)[:n_planes], rs,
                                    dis_rd.fileno(), dis_rd.plane_offsets(i)[:n_planes], ds)
            elif by_fd:
                eng.submit_file(i, ref_rd.fileno(), ref_rd.plane_offsets(i)[:n_planes], dis_rd.fileno(), dis_rd.plane_offsets(i)[:n_planes])
            else:
                eng.submit(i, ref_rd.frame(i)[:n_planes], dis_rd.frame(i)[:n_planes])
            i += m
            if progress is not None:
                progress(i - a, b - a)
        local = eng.collect(a, b - a) if b > a else np.zeros((0, N.RECORD_DOUBLES))
    except BaseException:
        eng.close()
        raise
    (eng.release if hasattr(eng, "release") else eng.close)()   # healthy: parked for the next analysis of this geometry
    rec = shard.gather_records(local, n, world_size, rank, gather_device)
    if rank != 0:
        return None
    elapsed = time.perf_counter() - t_start
    return finish_records(rec, mdl, ri, psnr=psnr, ssim=ssim, n_subsample=n_subsample, n_planes=n_planes,
                          fps=n / elapsed if elapsed > 0 else 0.0)


def finish_records(rec: np.ndarray, mdl: M.VmafModel, info, *, psnr: bool, ssim: bool, n_subsample: int = 1,
                   n_planes: int = 1, fps: float = 0.0) -> ScoreResult:
    """Host epilogue: records -> libvmaf-named metric columns (+ vmaf), stats-file lines."""
    n = rec.shape[0]
    prefix = "integer_" if mdl.is_integer else ""
    metrics = M.metrics_from_records(rec, info.width, info.height, prefix)
    plane_sizes = [(info.width, info.height)] + ([(info.chroma_w, info.chroma_h)] * 2 if n_planes == 3 else [])
    psnr_lines = ssim_lines = None
    if psnr:
        from .engine import sse_from_records
        sse = sse_from_records(rec)[:, :n_planes]
        psnr_lines = report.psnr_stats_lines(sse, plane_sizes, info.bit_depth)
        pp, _ = report.psnr_values(sse, plane_sizes, info.bit_depth)
        cap = 6.0 * info.bit_depth + 12.0          # libvmaf's psnr feature caps at 60 dB (8-bit) / 72 dB (10-bit)
        for p, name in enumerate(("psnr_y", "psnr_cb", "psnr_cr")[:n_planes]):
            metrics[name] = _cap(pp[:, p], cap)
    if ssim:
        sv = rec[:, N.REC_SSIM:N.REC_SSIM + n_planes]
        ssim_lines = report.ssim_stats_lines(sv, plane_sizes)
        metrics["ssim"] = report.ssim_all(sv, plane_sizes)
    scored = M.score_frames(mdl, metrics)
    idx = np.arange(n)
    if n_subsample > 1:
        keep = idx % n_subsample == 0
        scored = {k: np.asarray(v)[keep] for k, v in scored.items()}
        idx = idx[keep]
    return ScoreResult(metrics=scored, frame_indices=idx, records=rec, info=info, fps=fps,
                       psnr_lines=psnr_lines, ssim_lines=ssim_lines, model_name=mdl.name)
