"""world_size-2 (and 3) run of the frame-sharding path on CPU with the gloo backend: contiguous
chunks, one-frame motion halo, one all-gather of records; result must equal the single-process run
bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from pqa2_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_and_balance():
    for n in (1, 7, 300, 3000, 5):
        for world in (1, 2, 3, 8):
            b = shard.all_bounds(n, world)
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [y - x for x, y in b]
            assert max(sizes) - min(sizes) <= 1
    assert shard.all_bounds(3000, 8)[1] == (375, 750)


def _worker(rank, world, port, rp, dp, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pqa2_amd.pipeline import score_files
    from tests.fake_engine import OracleEngine
    res = score_files(rp, dp, "vmaf_v0.6.1", psnr=True, ssim=True, rank=rank, world_size=world,
                      engine_factory=OracleEngine)
    if rank == 0:
        np.save(out_path, res["records"])
    else:
        assert res is None
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_equals_single_process(tmp_path, world):
    from pqa2_amd import synth, yuvio
    from pqa2_amd.pipeline import score_files
    from tests.fake_engine import OracleEngine
    w, h, n = 80, 48, 7
    refs, diss = synth.make_clip(w, h, n, 8, chroma=True)
    rp, dp = str(tmp_path / "r.y4m"), str(tmp_path / "d.y4m")
    yuvio.write_y4m(rp, refs, synth.clip_info(w, h))
    yuvio.write_y4m(dp, diss, synth.clip_info(w, h))
    single = score_files(rp, dp, "vmaf_v0.6.1", psnr=True, ssim=True, engine_factory=OracleEngine)["records"]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rec.npy")
    mp.spawn(_worker, args=(world, port, rp, dp, out), nprocs=world, join=True)
    got = np.load(out)
    assert got.shape == single.shape
    assert np.array_equal(got.view(np.uint64), single.view(np.uint64))   # incl. motion across the shard seams
    assert np.all(got[1:, 16] > 0) and got[0, 16] == 0


def _vec_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 11
    a, b = shard.shard_bounds(n, world, rank)
    full = shard.gather_vector(np.arange(a, b, dtype=np.float64) * 1.5, n, world, rank)
    if rank == 1:
        np.save(out_path, full)
    dist.destroy_process_group()


def test_gather_vector_two_ranks(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "v.npy")
    mp.spawn(_vec_worker, args=(2, port, out), nprocs=2, join=True)
    assert np.array_equal(np.load(out), np.arange(11) * 1.5)


def test_synthetic_frames_do_not_depend_on_how_the_clip_is_sharded():
    """bench.py gives every rank `make_clip_cuda(..., t0=first frame of its chunk)`: frame t must be the same bytes whichever
    chunk generates it (luma and chroma, 8 and 10 bit), or an N-rank job scores a different clip than the one-rank job."""
    import torch
    from pqa2_amd import synth_torch
    for bpc in (8, 10):
        whole = synth_torch.make_clip_cuda(96, 64, 6, bpc, device="cpu", chroma=True)
        part = synth_torch.make_clip_cuda(96, 64, 3, bpc, device="cpu", chroma=True, t0=3)
        for side in ("ref", "dis"):
            for p in range(3):
                assert torch.equal(whole[side][p][3:], part[side][p]), (bpc, side, p)
        assert not torch.equal(whole["dis"][0][0], whole["dis"][0][1])
