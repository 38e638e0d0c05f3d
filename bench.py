#!/usr/bin/env python3
"""bench.py -- VMAF frames/s of the MI355X-native scoring path (BASELINE.json metric).

A "step" is one pass of the hot path over one synthetic clip that is already resident in HBM:
VIF + ADM + motion kernels for every frame -> per-frame records -> (N > 1: one RCCL all-gather of
the records) -> SVM + pooling on the host of rank 0.  Weak scaling: every rank owns `--frames`
consecutive frames of one long clip (rank r holds frames [r*F, (r+1)*F) plus the one-frame motion
halo in front), so the job scores N*F frames per step.

    python bench.py                      # N=1, 2160p vmaf_4k_v0.6.1, 300 frames
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel,
HIP-event timed on the stream it runs on) and `cpu_baseline` (the CPU oracle = scalar port of the
libvmaf float extractors, timed on a bounded sample of the same frames; N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# the pool's host driver supports dmabuf IPC only: without this RCCL (and any CUDA-tensor sharing across the ranks of an N > 1
# job) fails with hipIpcGetMemHandle: invalid argument.  The boxes export it; a launcher that scrubs the environment must not matter.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (width, height, bit_depth, model, chroma+psnr+ssim)
    "2160p": (3840, 2160, 8, "vmaf_4k_v0.6.1", False),
    "1080p": (1920, 1080, 8, "vmaf_v0.6.1", False),
    "2160p10": (3840, 2160, 10, "vmaf_v0.6.1neg", True),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
_T0 = time.perf_counter()
_STAGES = os.environ.get("PQA_BENCH_STAGES", "") == "1"


def _stage(msg: str):
    """Flushed progress marker on stderr (PQA_BENCH_STAGES=1 / --stages): when a profiler pass stalls, the last marker
    says where (tools/profile_round.sh keeps every pass's stderr)."""
    if _STAGES:
        print(f"[bench +{time.perf_counter() - _T0:7.2f}s] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="2160p", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=300, help="frames per rank")
    ap.add_argument("--batch", type=int, default=0, help="frames per kernel launch (0: the library's auto size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-frames", type=int, default=0, help="0: sized for ~12 s of CPU work")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0: min(16, host cores) -- the 1-GPU share of the box")
    ap.add_argument("--no-events", action="store_true", help="skip per-kernel HIP-event timing")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="finish every clip (collect, score, pool) before the next one is queued (A/B partner of the default)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --share-device rehearses the N>1 path on a one-GPU box")
    ap.add_argument("--share-device", action="store_true", help="every rank uses cuda:0 (rehearsal only)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the extra 1080p measurement the default 2160p run appends as `other_configs`")
    ap.add_argument("--no-e2e", action="store_true",
                    help="skip the end-to-end leg (Y4M files -> VMAFAnalyzer.analyze_videos -> JSON + stats files)")
    ap.add_argument("--cpu-all-cores", action="store_true", help=argparse.SUPPRESS)  # kept for scripts; the all-core leg is on by default
    ap.add_argument("--fixed-point", type=int, default=0,
                    help="PQA_FIXED_* mask (1 VIF, 2 motion): measure libvmaf's fixed-point arithmetic instead of the "
                         "default f32 path")
    ap.add_argument("--stages", action="store_true", help="flushed stage markers on stderr (also PQA_BENCH_STAGES=1)")
    args = ap.parse_args()
    global _STAGES
    _STAGES = _STAGES or args.stages

    _stage("importing torch")
    import torch
    import torch.distributed as dist
    from pqa2_amd import _native as N
    from pqa2_amd import model as M
    from pqa2_amd import shard, synth_torch
    from pqa2_amd.engine import FeatureEngine

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible (the scoring path has no CPU fallback)", file=sys.stderr)
        sys.exit(3)
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    gather_dev = dev if args.backend == "nccl" else None
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    w, h, bpc, model_name, side = WORKLOADS[args.workload]
    if args.batch <= 0:   # same rule as pqa_create: about 1.5 GiB of luma per launch, 8..256 frames
        args.batch = int(max(8, min(256, (1536 << 20) // (2 * w * h * (1 if bpc <= 8 else 2)))))
    F = args.frames
    total = F * world
    a = rank * F
    feats = N.FEAT_VMAF | ((N.FEAT_PSNR | N.FEAT_SSIM) if side else 0)
    n_planes = 3 if side else 1
    model = M.load_model(model_name)

    # ---- synthetic clip straight into HBM (this rank's chunk + one halo frame in front) ----------
    halo = 1 if a > 0 else 0
    _stage(f"device {local_rank} selected; generating the synthetic clip ({F + (1 if a > 0 else 0)} frames {w}x{h} {bpc}-bit, chroma={side})")
    clip = synth_torch.make_clip_cuda(w, h, F + halo, bpc, device=dev, chroma=side, t0=a - halo,
                                      progress=_stage if _STAGES else None)
    es = 1 if bpc <= 8 else 2
    ref_t, dis_t = clip["ref"], clip["dis"]
    ref_ptrs = [t[halo:].data_ptr() for t in ref_t]
    dis_ptrs = [t[halo:].data_ptr() for t in dis_t]
    row_pitch = [t.shape[2] * es for t in ref_t]
    frame_pitch = [t.shape[1] * t.shape[2] * es for t in ref_t]
    halo_ptr = ref_t[0][0].data_ptr() if halo else 0
    torch.cuda.synchronize()
    _stage("clip generated and synchronised; creating the context")

    eng = FeatureEngine(w, h, bit_depth=bpc, n_planes=n_planes, features=feats, device=local_rank,
                        max_batch=args.batch, result_capacity=max(2 * F, 1024),   # two clips in flight (see submit())
                        vif_enhn_gain_limit=model.vif_enhn_gain_limit,
                        adm_enhn_gain_limit=model.adm_enhn_gain_limit, vif_border=model.vif_border,
                        fixed_point=args.fixed_point)
    prefix = "integer_" if model.is_integer else ""
    result = {}
    _stage(f"context created (batch {args.batch})")

    n_launch = -(-F // args.batch)
    per_launch = -(-F // n_launch)

    # A step = one clip: submit() queues its launches, finish() collects its records launch by launch, scores and pools.
    # Consecutive clips alternate between two index ranges of the record ring, so clip k + 1 can be queued BEHIND clip k's
    # kernels before the host has scored clip k's last launch -- what a job that scores clip after clip does; without it the
    # GPU idles ~0.3 ms per 300-frame clip while the host scores the last launch and queues the next clip.  Every clip
    # starts its own motion chain (pqa_set_motion_halo(NULL): motion of its first frame is 0) or takes the rank's halo frame.
    def submit(k: int) -> int:
        base = a + (k & 1) * F
        if not halo_ptr:
            eng.set_motion_halo(None)
        eng.submit_resident(base, F, ref_ptrs, dis_ptrs, row_pitch, frame_pitch, halo_ptr, row_pitch[0])
        return base

    def finish(base: int):
        # records come back batch by batch: pqa_collect waits for ITS batch only (per-batch completion events), so the
        # host's share (feature epilogues + SVM) of batch k runs while the GPU works on batches k+1..  motion2 of a
        # frame needs its successor's motion: the last frame of what has arrived is scored with the next batch.
        rec = np.empty((F, N.RECORD_DOUBLES))
        vm_local = np.empty(F)
        scored = 0
        clip_end = a + F == total            # this rank holds the clip's last frame (its motion2 is its own motion)
        for b0 in range(0, F, per_launch):   # the library cuts a run into equal launches (pqa_submit_device)
            n = min(per_launch, F - b0)
            rec[b0:b0 + n] = eng.collect(base + b0, n)
            e = b0 + n
            upto = e if (e == F and clip_end) else e - 1
            if upto > scored:
                m = M.metrics_from_records(rec[scored:e], w, h, prefix)
                vm_local[scored:upto] = M.score_frames(model, {k: v[:upto - scored] for k, v in m.items()})["vmaf"]
                scored = upto
        if world == 1:
            full, vmaf = rec, vm_local
        else:
            # every rank has scored its own frames batch by batch, under its own kernels, exactly as at N = 1; what is left
            # is the LAST frame of its chunk, whose motion2 needs the first motion of the next rank's chunk: one all-gather
            # of the records (192 B per frame; rank 0 reports them), one frame through the SVM, and a second 8-byte-per-frame
            # gather of the scores.  No serial host stage grows with N.
            full = shard.gather_records(rec, total, world, rank, gather_dev)
            if scored < F:
                m = M.metrics_from_records(full[a + scored:a + F + 1], w, h, prefix)
                vm_local[scored:F] = M.score_frames(model, {k: v[:F - scored] for k, v in m.items()})["vmaf"]
            vmaf = shard.gather_vector(vm_local, total, world, rank, gather_dev)
        if rank == 0:
            result["records"] = full
            result["vmaf"] = vmaf
            result["pooled"] = M.pool(vmaf)

    n_steps = [0]

    def step():                       # one clip on its own: submitted, collected, scored
        n_steps[0] += 1
        finish(submit(n_steps[0]))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step()
        _stage(f"warm-up step {i + 1}/{args.warmup} done")
    if not args.no_events:
        eng.profile_enable([0])       # HIP events around the dominant kernel only (vif_stat_s0) while timing
    barrier()
    t0 = time.perf_counter()
    if args.no_pipeline:
        for i in range(args.steps):
            step()
            _stage(f"timed step {i + 1}/{args.steps} submitted and collected")
    else:                             # exactly K clips, every launch and every score inside the timed region
        k0 = n_steps[0] + 1
        pending = submit(k0)
        for i in range(args.steps):
            nxt = submit(k0 + i + 1) if i + 1 < args.steps else None
            finish(pending)
            pending = nxt
            _stage(f"timed step {i + 1}/{args.steps} collected" + ("" if nxt is None else "; the next one is queued"))
        n_steps[0] += args.steps
    barrier()
    elapsed = time.perf_counter() - t0
    _stage("timed region closed")
    prof = eng.profile_read() if not args.no_events else {}
    _stage("kernel events read")
    breakdown = {}
    if not args.no_events:
        eng.profile_enable(True)      # one extra, untimed pass with every kernel timed, for the breakdown
        step()
        breakdown = eng.profile_read()
        eng.profile_enable(False)
        _stage("breakdown pass done")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=gather_dev if gather_dev is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        fps = total * args.steps / elapsed
        b_alg = 2 * w * h * es  # every luma sample of the pair read once
        out = {
            "metric": "VMAF frames/sec (vif+adm+motion2 -> SVM), device-resident synthetic YUV pairs",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32" if not args.fixed_point else f"f32 + fixed-point extractors (mask {args.fixed_point})"), "data": "synthetic",
            "dtype_note": "every filter and statistic accumulates in f32.  VIF scale 0 (vif_march.hip) runs both passes on the f16 matrix "
                          "cores: the HORIZONTAL pass multiplies exact integer digit planes by two-piece f16 taps (22 bits of each f32 "
                          "tap; products exact in f32, f32 accumulators); the vertical pass feeds hi/lo f16 splits of those f32 results (about 22 "
                          "bits, the lo x lo term dropped) against two-piece taps, f32 accumulation -- an f32-grade evaluation in a "
                          "different rounding order, 8e-8 from f64 on the scale-0 numerator; fuzz bound of the whole path "
                          "|dVMAF| <= 0.0031 from 500 k pixels up (profiles/r03n_fuzz_summary.txt)",
            "config": {"workload": f"{args.workload} {w}x{h} {bpc}-bit, {model_name}, {F} frames/GPU"
                                   f"{' + PSNR/SSIM all planes' if side else ''}"
                                   f"{f' [--fixed-point {args.fixed_point}: libvmaf integer arithmetic]' if args.fixed_point else ''}",
                       "frames_per_gpu": F, "frames_total": total, "batch": args.batch, "frames_per_launch": per_launch,
                       "clips_pipelined": not args.no_pipeline,
                       "parallelism": f"frame-shard x{world}, 1-frame motion halo, all-gather of records + of scores"},
            "pooled_vmaf_mean": round(result["pooled"]["mean"], 6),
        }
        k = prof.get("vif_stat_s0")
        if k and k["launches"]:
            avg_ms = k["ms"] / k["launches"]
            frames_per_launch = k["frames"] / k["launches"]
            achieved = b_alg * frames_per_launch / (avg_ms * 1e-3) / 1e9
            cnt = {} if args.fixed_point & 1 else _kernel_counters(args.workload, "vif_stat_s0")
            per_frame = cnt.get("hbm_bytes_per_frame")
            traffic = int(per_frame * frames_per_launch) if per_frame else None
            if args.fixed_point & 1:
                kname = f"vif_fixed_kernel<{'u8' if bpc == 8 else 'u16'},17,240,9>"
            elif bpc <= 10:   # the march kernel: both filter passes on the f16 matrix cores (csrc/vif_march.hip)
                kname = (f"vif_s0_march_kernel<{'u8' if bpc == 8 else 'u16'}> (17-tap horizontal pass exact on f16 MFMA, vertical pass "
                         f"on two-piece f16 splits)")
            else:
                kname = "vif_stat_kernel<u16,17,240,9>"
            out["roofline"] = {"bound": "hbm", "bound_is_measured_limiter": False,
                               "bound_note": "north_star names the HBM roofline, so `frac` is priced against it; the kernel's measured "
                                             "limiter is vector + matrix instruction issue (`measured_limiter`, `valu`, `mfma`)",
                               "kernel": kname + " (VIF scale 0 + fused decimation to scale 1)",
                               "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(achieved / HBM_PEAK_GBS, 5),
                               "traffic": traffic,
                               "traffic_source": cnt.get("traffic_source", "no PMC pass committed for this workload / kernel source"),
                               "alg_bytes_per_launch": int(b_alg * frames_per_launch),
                               "avg_launch_ms": round(avg_ms, 4), "launches": k["launches"],
                               # SURVEY 8(d)'s primary formula for the WHOLE path: frames/s x B_alg / HBM peak
                               "pipeline_frac": round(fps / world * b_alg / (HBM_PEAK_GBS * 1e9), 5),
                               # what actually limits the kernel: FP32 issue, not bytes
                               "measured_limiter": "instruction issue: VALU (4 clk per wave64 instruction) + MFMA (8 issue clk of 16 each) on "
                                                   "one port per SIMD, 80 % busy at 3 waves per SIMD (profiles/*_sq_counters.txt); not HBM: see `valu`, `mfma`"}
            # matrix-core share of the same launches (v_mfma_f32_16x16x32_f16 = 16384 FLOP each; dense f16 peak 2.5 PFLOP/s,
            # MI355X_MICROARCH.md).  March kernel: per 16 x 16 block the pass-1 MFMAs (repeated once per segment) + 18 in
            # pass 2; the counts and the segment rule are asked from the library (pqa_debug_vif_march_shape)
            mfma_wave_insts, mfma_shape = 0, None
            if bpc <= 10 and not args.fixed_point:
                import ctypes as _C
                shp = (_C.c_int32 * 6)()
                if N.load().pqa_debug_vif_march_shape(w, h, _C.byref(shp)) == 0:   # the library's own segment rule and MFMA counts
                    n_cb, rbk, seg, n_seg, m1, m2 = list(shp)
                    mfma_shape = {"stripes": n_cb, "row_blocks": rbk, "blocks_per_segment": seg, "segments": n_seg,
                                  "pass1_mfma_per_block": m1, "pass2_mfma_per_block": m2}
                    mfma_wave_insts = n_cb * (m1 * (rbk + n_seg) + m2 * rbk)
            if cnt.get("valu_insts_per_wave"):
                # issue floor = the clocks the VALU / matrix issue port of a SIMD is held per frame: SQ_INSTS_VALU counts the
                # MFMAs too; an MFMA 16x16x32 holds the port 8 clocks (of the 16 it occupies the matrix pipe), every other
                # counted instruction is priced at 4 (what v_pk_fma_f32, converts, selects, permutes cost; plain v_fma / v_mul /
                # v_add issue in 2-3: tools/ubench/valu_ops.hip -- so this is an upper estimate of the port time, by a little)
                clk = cnt.get("shader_clock_ghz", 2.0)
                wave_insts = cnt.get("valu_wave_insts_per_frame") or cnt["valu_insts_per_wave"] * cnt["waves_per_frame"]
                mfma_n = min(mfma_wave_insts, wave_insts)
                floor_us = ((wave_insts - mfma_n) * 4.0 + mfma_n * 8.0) / (1024 * clk * 1e3)
                meas_us = 1e3 * avg_ms / frames_per_launch
                out["roofline"]["valu"] = {
                    "valu_insts_per_wave": cnt["valu_insts_per_wave"], "waves_per_frame": cnt["waves_per_frame"],
                    "valu_wave_insts_per_frame": int(wave_insts), "mfma_wave_insts_per_frame": int(mfma_n),
                    "shader_clock_ghz": clk, "issue_floor_us_per_frame": round(floor_us, 2),
                    "measured_us_per_frame": round(meas_us, 2), "frac": round(floor_us / meas_us, 4),
                    "peak_note": "issue-port clocks per frame / (1024 SIMDs x shader clock): MFMAs (counted by SQ_INSTS_VALU) at 8 "
                                 "clocks each, everything else at 4; a count-based model of the port, not a cycle trace",
                    "source": cnt.get("valu_source")}
            if mfma_wave_insts:
                mf = mfma_wave_insts * 16384.0
                note = (f"{mfma_shape['pass1_mfma_per_block'] + mfma_shape['pass2_mfma_per_block']} MFMAs per 16 x 16 output block "
                        f"({mfma_shape['pass1_mfma_per_block']} first-pass + {mfma_shape['pass2_mfma_per_block']} second-pass; "
                        f"{mfma_shape['blocks_per_segment']}-block segments repeat one first-pass block each); a Toeplitz band uses 17 of "
                        "the 32 K slots, so the USEFUL share of these FLOP is about half")
                out["roofline"]["mfma"] = {"flop_per_frame": mf, "achieved_tflops": round(mf * frames_per_launch / (avg_ms * 1e-3) / 1e12, 1),
                                           "peak_tflops": 2500.0, "frac": round(mf * frames_per_launch / (avg_ms * 1e-3) / 2.5e15, 4),
                                           "note": note + "; the matrix pipe shares its issue port with the VALU: neither fraction can "
                                                          "approach 1 alone"}
            out["kernel_ms_per_frame"] = {name: round(v["ms"] / max(1, v["frames"]), 5)
                                          for name, v in breakdown.items() if v["launches"]}
            out["kernel_ms_note"] = "from one extra untimed pass with every kernel event-timed (ms per frame)"
            # the second-largest launch since round 4: ADM scales 0 + 1 in one kernel (csrc/adm_pyramid.hip), which -- unlike VIF
            # scale 0 -- sits on the memory side.  Same accounting: algorithmic bytes (the luma pair once) over its event time.
            ka = breakdown.get("adm_scale_s0")
            if ka and ka["launches"] and not (args.fixed_point & 4):
                a_ms = ka["ms"] / ka["launches"]
                a_fpl = ka["frames"] / ka["launches"]
                a_ach = b_alg * a_fpl / (a_ms * 1e-3) / 1e9
                fused = "adm_scale_s1" not in breakdown or not breakdown["adm_scale_s1"]["launches"]
                ca = _kernel_counters(args.workload, "adm_s0")
                out["roofline_adm"] = {"bound": "hbm", "kernel": ("adm_pyramid_kernel (ADM scales 0 + 1 in one launch)" if fused
                                                                 else "adm_march_kernel (ADM scale 0)"),
                                       "achieved": round(a_ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(a_ach / HBM_PEAK_GBS, 5), "avg_launch_ms": round(a_ms, 4),
                                       "traffic": int(ca["hbm_bytes_per_frame"] * a_fpl) if ca.get("hbm_bytes_per_frame") else None,
                                       "traffic_source": ca.get("traffic_source"),
                                       "timed": "breakdown pass (one stream, every kernel event-timed), not the timed region"}

        if world == 1 and not args.no_cpu_baseline and not args.fixed_point:  # the CPU leg times the float restatement
            threads = args.cpu_threads or min(16, os.cpu_count() or 1)
            n_sample = args.cpu_sample_frames or max(2, min(F, threads * max(1, int(12.0 / (w * h * 2.0e-7)))))
            out["cpu_baseline"] = _cpu_baseline(ref_t, dis_t, halo, bpc, w, h, n_sample, threads,
                                                result["records"], model, prefix)
        if world == 1 and args.workload == "2160p" and not args.fixed_point and not args.no_other_configs:
            out["other_configs"] = _other_configs(args)
        if world == 1 and args.workload == "2160p" and not args.fixed_point and not args.no_e2e:
            out["e2e"] = _e2e_leg()
            out["bookend"] = _bookend_leg()
        probe = probe_libvmaf()
        if probe["ffmpeg_has_libvmaf"] and world == 1 and not args.no_cpu_baseline:
            n_lv = min(F, 60 if w * h > 1920 * 1080 else 150)
            out["libvmaf_side_by_side"] = {"probe": probe, **_libvmaf_leg(probe, ref_t, dis_t, halo, bpc, w, h, model_name,
                                                                           result["vmaf"], n_lv)}
        else:
            out["libvmaf_side_by_side"] = {"probe": probe,
                                           "result": "not run: no ffmpeg with the libvmaf filter on this box (probed PATH and "
                                                     "`ffmpeg -filters`; a `vmaf` CLI alone is reported but not driven); "
                                                     "cpu_baseline.kind = port, parity against libvmaf stays unpinned"
                                                     if not probe["ffmpeg_has_libvmaf"] else "not run (N > 1 or --no-cpu-baseline)"}
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


def _other_configs(args):
    """BASELINE.json names 1080p (configs[1]) and 2160p 10-bit neg + PSNR + SSIM (configs[4]) next to the 4K headline:
    measure them too, each in a child process with the same step definition (its own context and clip), and carry the
    essentials of their lines along.  Never part of `value`."""
    import subprocess
    res = {}
    for wl in ("1080p", "2160p10"):
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", wl, "--steps", str(max(3, args.steps)),
               "--warmup", str(args.warmup), "--no-cpu-baseline", "--no-other-configs", "--no-e2e"]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            rf = d.get("roofline", {})
            res[wl] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
                       "workload": d["config"]["workload"], "roofline_frac": rf.get("frac"),
                       "pipeline_frac": rf.get("pipeline_frac"), "kernel": rf.get("kernel"),
                       # the committed rocprofv3 --pmc figures of that workload's dominant kernel (profiles/kernel_counters.json),
                       # dropped by the child when the kernel source has changed since they were taken
                       "traffic": rf.get("traffic"), "traffic_source": rf.get("traffic_source"),
                       "valu": {k: rf["valu"][k] for k in ("valu_insts_per_wave", "waves_per_frame", "issue_floor_us_per_frame",
                                                           "measured_us_per_frame", "frac", "source")} if "valu" in rf else None}
        except Exception as e:  # the headline must not depend on this
            res[wl] = {"error": str(e)[:200]}
    return res


def _e2e_leg():
    """SURVEY 8(d) end-to-end leg: two 4:2:0 Y4M files on disk -> VMAFAnalyzer.analyze_videos() (host planes -> pinned
    staging -> H2D overlapped with the kernels -> SVM -> libvmaf-format JSON + psnr/ssim stats files), wall clock, in a
    child process per size (tools/e2e_file_bench.py), 300 frames as BASELINE.json's configs.  Both analyses of the
    process are reported -- the FIRST (context creation + pinning the staging buffers: what a one-shot caller gets) and the
    second (staging parked in the library) -- next to the host-to-device copy ceiling measured in the same process.
    PCIe- and memcpy-bound; never part of `value`."""
    import shutil
    import subprocess
    import tempfile
    out = {}
    for size, frames in (("1920x1080", 300), ("3840x2160", 300)):
        d = tempfile.mkdtemp(prefix="pqa_e2e_")
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e_file_bench.py"), "--size", size,
                                "--frames", str(frames), "--dir", d], capture_output=True, text=True, timeout=600)
            lines = [json.loads(x) for x in r.stdout.strip().splitlines() if x.startswith("{")]
            passes = {x["pass"]: x for x in lines if "pass" in x}
            link = next((x for x in lines if "h2d_GBps" in x), {})
            out[size] = {"frames": frames,
                         "first_analysis_fps": passes[0]["fps_end_to_end"], "first_analysis_seconds": passes[0]["seconds"],
                         "second_analysis_fps": passes[1]["fps_end_to_end"], "second_analysis_seconds": passes[1]["seconds"],
                         "h2d_GBps": link.get("h2d_GBps"), "bytes_per_frame_pair": link.get("bytes_per_frame_pair"),
                         "pcie_ceiling_fps": link.get("pcie_ceiling_fps"),
                         "what": "2 Y4M files (written just before: page cache) -> analyze_videos -> JSON + psnr.txt + ssim.txt, "
                                 "all planes, vmaf_v0.6.1; context creation and file writing included; first = cold pinned "
                                 "staging, second = staging parked by the first; ceiling = bytes per frame pair / measured "
                                 "pinned H2D rate"}
        except Exception as e:
            out[size] = {"error": (str(e) or "failed")[:200]}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return out


def probe_libvmaf():
    """BASELINE.md section 3 step 1 / SURVEY 8(d): is there an ffmpeg with the libvmaf filter, or a `vmaf` CLI, on this
    box?  No installation, no download: a lookup on PATH and `ffmpeg -hide_banner -filters`."""
    import shutil
    import subprocess
    info = {"ffmpeg": shutil.which("ffmpeg"), "vmaf_cli": shutil.which("vmaf"), "ffmpeg_has_libvmaf": False,
            "ffmpeg_version": None}
    if info["ffmpeg"]:
        try:
            r = subprocess.run([info["ffmpeg"], "-hide_banner", "-filters"], capture_output=True, text=True, timeout=30)
            info["ffmpeg_has_libvmaf"] = any(len(l.split()) > 1 and l.split()[1] == "libvmaf" for l in r.stdout.splitlines())
            v = subprocess.run([info["ffmpeg"], "-version"], capture_output=True, text=True, timeout=30)
            info["ffmpeg_version"] = (v.stdout.splitlines() or [None])[0]
        except Exception as e:
            info["probe_error"] = str(e)[:200]
    return info


def _libvmaf_leg(probe, ref_t, dis_t, halo, bpc, w, h, model_name, gpu_vmaf, n_frames):
    """The reference's own command (app/vmaf_analyzer.py:411-419) on the first `n_frames` frames of the bench clip, with
    n_threads = 4 (the reference's fixed default, :32) and = all usable cores: frames / wall clock, and the per-frame
    `vmaf` of its JSON log against ours.  Only runs when probe_libvmaf() found the filter."""
    import shutil
    import subprocess
    import tempfile
    from pqa2_amd import synth, yuvio
    d = tempfile.mkdtemp(prefix="pqa_libvmaf_")
    out = {"frames": n_frames, "ffmpeg_version": probe.get("ffmpeg_version"), "host_cores": os.cpu_count()}
    try:
        info = synth.clip_info(w, h, bpc, chroma=False)
        paths = {}
        for side, t in (("ref", ref_t), ("dis", dis_t)):
            paths[side] = os.path.join(d, f"{side}.y4m")
            frames = ([t[0][halo + i].cpu().numpy().view(np.uint16 if bpc > 8 else np.uint8)] for i in range(n_frames))
            yuvio.write_y4m(paths[side], frames, info)
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 4)
        for threads in sorted({4, avail}):
            log = os.path.join(d, f"vmaf_t{threads}.json")
            opt = f"libvmaf=log_path={log}:log_fmt=json:model=version={model_name}:n_threads={threads}:n_subsample=1"
            t0 = time.perf_counter()
            r = subprocess.run([probe["ffmpeg"], "-hide_banner", "-loglevel", "info", "-i", paths["dis"], "-i", paths["ref"],
                                "-lavfi", opt, "-f", "null", "-"], capture_output=True, text=True, timeout=1200)
            dt = time.perf_counter() - t0
            rec = {"fps": round(n_frames / dt, 3), "seconds": round(dt, 2), "returncode": r.returncode}
            try:
                with open(log) as f:
                    lv = json.load(f)
                theirs = np.array([fr["metrics"]["vmaf"] for fr in lv["frames"]][:n_frames])
                rec["max_abs_vmaf_diff_per_frame"] = float(np.abs(theirs - gpu_vmaf[:len(theirs)]).max())
                rec["pooled_abs_vmaf_diff"] = float(abs(theirs.mean() - gpu_vmaf[:len(theirs)].mean()))
            except Exception as e:
                rec["log_error"] = str(e)[:200]
            out[f"n_threads_{threads}"] = rec
    except Exception as e:
        out["error"] = str(e)[:300]
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out


def _bookend_leg():
    """The step in front of the scoring path: the luma-statistics kernel on a resident 1080p clip
    against the HBM roofline, and bookend.detect() end to end on a Y4M file (tools/bookend_bench.py, child process)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bookend_bench.py"), "--size", "1920x1080",
                            "--frames", "300"], capture_output=True, text=True, timeout=300)
        d = json.loads([x for x in r.stdout.strip().splitlines() if x.startswith("{")][-1])
        return {"luma_stats_resident_GBps": d["resident"]["GB_per_s"], "luma_stats_hbm_frac": d["resident"]["hbm_frac"],
                "detect_clip_frames_per_s": d["detect"]["clip_frames_per_s"],
                "detect_numpy_clip_frames_per_s": d["detect_numpy"]["clip_frames_per_s"],
                "same_bookends_as_numpy": d["detect_numpy"]["same_result"],
                "what": "1920x1080, 300 frames, two white sections; resident = pqa_luma_stats_device, detect = "
                        "pqa2_amd.bookend.detect on a Y4M file through pqa_luma_stats"}
    except Exception as e:
        return {"error": (str(e) or "failed")[:200]}


def kernel_source_hash(group: str = "vif") -> str:
    """sha256 over the sources of a kernel whose committed counters bench.py quotes: counters measured for another version
    of it are stale.  "vif" (the dominant kernel, VIF scale 0): the VIF kernel files, the shared device helpers and the VIF
    section of kernels.h (tile geometry, launcher contracts) -- not the declarations of unrelated kernels in that header.
    "adm": the ADM march / pyramid kernels and their shared chain."""
    import hashlib
    hsh = hashlib.sha256()
    d = os.path.join(ROOT, "pqa2_amd", "csrc")
    # the scale-0 kernels live in vif.hip / vif_march.hip; march_common.h holds the march kernel's splits and MFMA wrappers
    files = {"vif": ("vif.hip", "vif_march.hip", "march_common.h", "pqa_device.h"),
             "adm": ("adm_pyramid.hip", "adm_march.hip", "adm_chain.h", "adm.hip", "pqa_device.h")}[group]
    for f in files:
        if os.path.exists(os.path.join(d, f)):
            with open(os.path.join(d, f), "rb") as fh:
                hsh.update(fh.read())
    with open(os.path.join(d, "kernels.h"), "rb") as fh:
        k = fh.read()
    marks = {"vif": (b"// ---- VIF", b"// ---- ADM"), "adm": (b"// ---- ADM", b"// ---- motion")}[group]
    a, b = k.find(marks[0]), k.find(marks[1])
    hsh.update(k[a:b] if 0 <= a < b else k)
    return hsh.hexdigest()[:16]


def _kernel_counters(workload: str, kernel: str) -> dict:
    """Per-frame PMC figures of a kernel from the committed rocprofv3 passes (bench.py cannot collect PMC counters
    itself; tools/profile_round.sh + tools/summarize_rocprof.py write profiles/kernel_counters.json).  Figures measured
    on a DIFFERENT version of the kernel source are dropped, and the line says so."""
    p = os.path.join(ROOT, "profiles", "kernel_counters.json")
    try:
        with open(p) as f:
            d = json.load(f).get(workload, {}).get(kernel, {})
    except Exception:
        return {}
    if not d:
        return {}
    now = kernel_source_hash("adm" if kernel.startswith("adm") else "vif")
    if d.get("src_hash") != now:
        return {"traffic_source": f"{d.get('traffic_source', p)} is for kernel source {d.get('src_hash')}, the kernel has "
                                  f"changed since (now {now}): dropped as stale"}
    return d


def _cgroup_cpu_max():
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(p) as f:
                return f.read().strip()
        except Exception:
            pass
    return None


def _cpu_baseline(ref_t, dis_t, halo, bpc, w, h, n_sample, threads, gpu_records, model, prefix):
    """Time the CPU oracle (scalar C port of the libvmaf float extractors, one frame per thread) on the
    first `n_sample` frames of the very same clip, and report how far the GPU records are from it."""
    from oracle.oracle import Oracle
    from pqa2_amd import model as M
    orc = Oracle("f32")
    refs = [ref_t[0][halo + i].cpu().numpy() for i in range(n_sample)]
    diss = [dis_t[0][halo + i].cpu().numpy() for i in range(n_sample)]
    if bpc > 8:
        refs = [r.view(np.uint16) for r in refs]
        diss = [d.view(np.uint16) for d in diss]
    t0 = time.perf_counter()
    exp = orc.clip_features_mt(refs, diss, bpc, threads, vif_gain_limit=model.vif_enhn_gain_limit,
                               adm_gain_limit=model.adm_enhn_gain_limit, vif_border101=bool(model.vif_border))
    dt = time.perf_counter() - t0
    got = gpu_records[:n_sample, :17]
    rel = np.abs(got[:, :16] - exp[:, :16]) / np.maximum(np.abs(exp[:, :16]), 1e-12)
    rec = np.zeros((n_sample, 24))
    rec[:, :17] = exp
    v_cpu = M.score_frames(model, M.metrics_from_records(rec, w, h, prefix))["vmaf"]
    v_gpu = M.score_frames(model, M.metrics_from_records(gpu_records[:n_sample], w, h, prefix))["vmaf"]
    # the same frames through the fixed-point restatement: the arithmetic libvmaf's DEFAULT models run on a CPU
    # (integer_vif / integer_adm / integer_motion), again one frame per thread
    from oracle.int_oracle import IntOracle
    into = IntOracle()
    n_fx = min(n_sample, 2 * threads)
    t0 = time.perf_counter()
    into.clip_features_mt(refs[:n_fx], diss[:n_fx], bpc, threads, model.vif_enhn_gain_limit, model.adm_enhn_gain_limit)
    dt_fx = time.perf_counter() - t0
    # the same restatement on every core this process may use ("all cores, labelled")
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or threads)
    quota = _cgroup_cpu_max()
    try:   # a container's CPU quota caps what "all cores" means: 256 visible cores under a 16-CPU quota are 16 cores
        q, per = quota.split()[:2]
        if q != "max":
            avail = max(1, min(avail, int(int(q) / int(per))))
    except Exception:
        pass
    all_core = None
    if avail > threads:
        n_all = min(ref_t[0].shape[0] - halo, avail)
        more_r = refs + [ref_t[0][halo + i].cpu().numpy() for i in range(n_sample, n_all)]
        more_d = diss + [dis_t[0][halo + i].cpu().numpy() for i in range(n_sample, n_all)]
        if bpc > 8:
            more_r = [r.view(np.uint16) for r in more_r]
            more_d = [d.view(np.uint16) for d in more_d]
        t0 = time.perf_counter()
        orc.clip_features_mt(more_r[:n_all], more_d[:n_all], bpc, avail, vif_gain_limit=model.vif_enhn_gain_limit,
                             adm_gain_limit=model.adm_enhn_gain_limit, vif_border101=bool(model.vif_border))
        dt_all = time.perf_counter() - t0
        all_core = {"value": round(n_all / dt_all, 4), "cores": avail, "cgroup_cpu_max": _cgroup_cpu_max(),
                    "sample": f"first {n_all} frames, one frame per thread on all {avail} schedulable cores, {dt_all:.1f} s"}
        del more_r, more_d
    return {"value": round(n_sample / dt, 4), "unit": "frames/s", "cores": threads, "kind": "port",
            "all_cores": all_core if all_core else f"{avail} usable cores (affinity {len(os.sched_getaffinity(0))}, cgroup cpu.max "
                                                   f"'{quota}'): the {threads}-thread figure IS the all-core figure of this box share",
            "fixed_point_port_value": round(n_fx / dt_fx, 4),
            "fixed_point_port_sample": f"first {n_fx} frames, oracle/vmaf_int_oracle.c, {dt_fx:.1f} s on {threads} threads",
            "sample": f"first {n_sample} frames of the same clip, oracle/vmaf_oracle.c f32 (VIF+ADM+motion), "
                      f"{dt:.1f} s on {threads} threads (one frame each) of {os.cpu_count()} host cores; "
                      f"ffmpeg/libvmaf not present on this box",
            "gpu_vs_oracle_max_rel_feature_err": float(rel.max()),
            "gpu_vs_oracle_max_abs_vmaf_err": float(np.abs(v_cpu - v_gpu).max())}


if __name__ == "__main__":
    main()
