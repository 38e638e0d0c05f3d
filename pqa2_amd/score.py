"""Command-line / torchrun entry of the scoring path:

    python -m pqa2_amd.score REF DIS --model vmaf_v0.6.1 --json out.json [--psnr-log p.txt --ssim-log s.txt]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           -m pqa2_amd.score REF DIS --json out.json          # frame-sharded, one process per GPU

Progress goes to stderr as `frame= N` lines -- the same shape the reference parses from its ffmpeg
child (app/vmaf_analyzer.py:475-492), so VMAFAnalyzer drives a multi-GPU job exactly the way the
reference drives ffmpeg."""
from __future__ import annotations

import argparse
import os
import sys
import time


def _expected_parent() -> int:
    """The process a worker must not outlive: PQA_PARENT_PID when the launcher exported it (a launcher that starts this
    module DIRECTLY may do so to close the window between fork and this line; never set it under torchrun, whose agent --
    not the analyzer -- is the workers' parent), else the parent as it is now.  Read at the very top of main()."""
    try:
        return int(os.environ.get("PQA_PARENT_PID", "")) or os.getppid()
    except ValueError:
        return os.getppid()


def _die_with_parent(expected: int, getppid=os.getppid, kill=os.kill) -> None:
    """A worker must not outlive the job that started it (a rank blocked in the record gather holds its GPU context for
    ever): ask the kernel for SIGTERM when the parent -- torchrun's agent, or whoever started a one-rank job -- goes away.
    Linux only (prctl PR_SET_PDEATHSIG); elsewhere a no-op.

    The parent may have died BEFORE the request took effect; then this process has already been re-parented and no signal
    will ever come.  That case is recognised by comparing the parent now with `expected` (the parent at start-up) -- not by
    `getppid() == 1`: a container's entrypoint or an init-less shell IS pid 1 and perfectly alive, and under a sub-reaper
    (systemd --user, `docker run --init`, a test harness) an orphan's parent is the reaper, not 1.

    Caveat of PR_SET_PDEATHSIG: it fires when the THREAD that forked this process exits, not the parent process.  A
    launcher that spawns this module from a short-lived thread (a QThread that returns while the job runs) would end the
    job early; spawn from a thread that outlives the child.  Under torchrun the workers' parent is the agent's main thread."""
    try:
        import ctypes
        import signal
        libc = ctypes.CDLL(None, use_errno=True)
        libc.prctl(1, int(signal.SIGTERM), 0, 0, 0)   # PR_SET_PDEATHSIG = 1
        if getppid() != expected:                      # re-parented already: the parent went away before we asked
            kill(os.getpid(), signal.SIGTERM)
    except Exception:
        pass


def main(argv=None) -> int:
    parent = _expected_parent()    # before anything slow: imports, argument parsing
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool's driver: RCCL needs it (before torch loads)
    ap = argparse.ArgumentParser(prog="pqa2_amd.score")
    ap.add_argument("reference")
    ap.add_argument("distorted")
    ap.add_argument("--model", default="vmaf_v0.6.1")
    ap.add_argument("--json", required=True)
    ap.add_argument("--psnr-log")
    ap.add_argument("--ssim-log")
    ap.add_argument("--n-subsample", type=int, default=1)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--fixed-point", type=int, default=0,
                    help="PQA_FIXED_* mask (1 VIF, 2 motion): extractors run in libvmaf's fixed-point arithmetic")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend of the record gather; gloo + --share-device rehearses N ranks on one GPU")
    ap.add_argument("--share-device", action="store_true", help="every rank uses device 0 (rehearsal on a one-GPU box)")
    a = ap.parse_args(argv)

    from . import report
    from .pipeline import score_files
    _die_with_parent(parent)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    gather_device = None
    if a.share_device:
        local_rank = 0
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if a.backend == "nccl":      # RCCL: the records travel GPU to GPU over xGMI
            gather_device = torch.device("cuda", local_rank)
            dist.init_process_group("nccl", device_id=gather_device)
        else:                        # gloo: host tensors (several ranks may then share one GPU)
            dist.init_process_group("gloo")
    last = [0.0]

    def progress(done, total):
        now = time.time()
        if rank == 0 and (now - last[0] > 0.25 or done == total):
            last[0] = now
            print(f"frame= {done * world} fps=0 q=0.0 size=N/A", file=sys.stderr, flush=True)

    try:
        res = score_files(a.reference, a.distorted, a.model, psnr=bool(a.psnr_log), ssim=bool(a.ssim_log),
                          n_subsample=a.n_subsample, device=local_rank, rank=rank, world_size=world,
                          gather_device=gather_device, max_batch=a.batch, progress=progress, fixed_point=a.fixed_point)
    except Exception as e:  # one line on stderr, non-zero exit: what the caller's returncode check expects
        print(f"pqa2_amd.score: error: {e}", file=sys.stderr, flush=True)
        return 1
    finally:
        if world > 1:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()
    if rank == 0:
        log = report.build_vmaf_log(res["metrics"], res["fps"], res["frame_indices"], {"model": res["model_name"]})
        report.write_vmaf_json(a.json, log)
        if a.psnr_log and res["psnr_lines"] is not None:
            with open(a.psnr_log, "w") as f:
                f.write("\n".join(res["psnr_lines"]) + "\n")
        if a.ssim_log and res["ssim_lines"] is not None:
            with open(a.ssim_log, "w") as f:
                f.write("\n".join(res["ssim_lines"]) + "\n")
        print(f"VMAF score: {log['pooled_metrics']['vmaf']['mean']:.6f}", file=sys.stderr, flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
