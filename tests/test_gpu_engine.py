"""GPU end-to-end: VMAFAnalyzer.analyze_videos through the real HIP engine; score parity with the
oracle within north_star's tolerance (|dVMAF| <= 0.01); full-size (1080p / 2160p) checks through
size-independent properties."""
import json
import os

import numpy as np
import pytest

from pqa2_amd import model as M
from pqa2_amd import synth, yuvio

pytestmark = pytest.mark.gpu
VMAF_TOL = 0.01   # BASELINE.json north_star: per-frame and pooled scores within +-0.01 VMAF


def _pair(tmp_path, w, h, n, bpc=8):
    refs, diss = synth.make_clip(w, h, n, bpc, chroma=True)
    info = synth.clip_info(w, h, bpc)
    rp, dp = str(tmp_path / "ref.y4m"), str(tmp_path / "dist.y4m")
    yuvio.write_y4m(rp, refs, info)
    yuvio.write_y4m(dp, diss, info)
    return rp, dp, refs, diss


@pytest.mark.parametrize("w,h,bpc,model", [(352, 288, 8, "vmaf_v0.6.1"), (321, 241, 8, "vmaf_4k_v0.6.1"),
                                            (320, 180, 10, "vmaf_v0.6.1neg")])
def test_analyzer_scores_match_oracle(tmp_path, oracle32, w, h, bpc, model):
    from pqa2_amd.vmaf_analyzer import VMAFAnalyzer
    n = 5
    rp, dp, refs, diss = _pair(tmp_path, w, h, n, bpc)
    a = VMAFAnalyzer()
    a.set_output_directory(str(tmp_path))
    a.set_test_name("gpu")
    errors = []
    a.error_occurred.connect(errors.append)
    res = a.analyze_videos(rp, dp, model)
    assert errors == [] and res is not None
    mdl = M.load_model(model)
    rec = np.zeros((n, 24))
    rec[:, :17] = oracle32.clip_features([r[0] for r in refs], [d[0] for d in diss], bpc,
                                         vif_gain_limit=mdl.vif_enhn_gain_limit, adm_gain_limit=mdl.adm_enhn_gain_limit,
                                         vif_border101=bool(mdl.vif_border))
    want = M.score_frames(mdl, M.metrics_from_records(rec, w, h, "integer_"))
    got = res["raw_results"]["frames"]
    dv = max(abs(got[i]["metrics"]["vmaf"] - want["vmaf"][i]) for i in range(n))
    assert dv <= VMAF_TOL, dv
    assert abs(res["vmaf_score"] - float(np.mean(want["vmaf"]))) <= VMAF_TOL
    for k in ("integer_adm2", "integer_motion2", "integer_vif_scale0", "integer_vif_scale1", "integer_vif_scale2", "integer_vif_scale3"):
        d = max(abs(got[i]["metrics"][k] - want[k][i]) for i in range(n))
        assert d < 2e-5, (k, d)
    # PSNR stats: exact integer SSE -> identical 2-decimal text to the oracle-derived line
    from pqa2_amd import report
    sse = np.array([[oracle32.sse_plane(diss[i][p], refs[i][p], bpc) for p in range(3)] for i in range(n)], np.uint64)
    sizes = [(w, h), ((w + 1) // 2, (h + 1) // 2), ((w + 1) // 2, (h + 1) // 2)]
    assert open(res["psnr_log"]).read().split("\n")[:-1] == report.psnr_stats_lines(sse, sizes, bpc)


@pytest.mark.parametrize("w,h", [(1920, 1080), (3840, 2160), (4098, 2162), (7680, 4320)])
def test_full_size_properties(w, h):
    """BASELINE sizes, checked through properties that need no oracle pass:
    identical -> vif/adm == 1, SSE == 0; static -> motion == 0; +c offset -> SSE == c^2*W*H and a
    bounded, monotone VMAF drop; host path == device-resident path bit for bit."""
    import torch
    from pqa2_amd import _native as N
    from pqa2_amd import synth_torch
    from pqa2_amd.engine import FeatureEngine, sse_from_records
    clip = synth_torch.make_clip_cuda(w, h, 3, 8, device="cuda")
    R, D = clip["ref"][0], clip["dis"][0]
    torch.cuda.synchronize()
    with FeatureEngine(w, h, features=N.FEAT_VMAF | N.FEAT_PSNR | N.FEAT_SSIM, max_batch=2) as eng:
        eng.submit_resident(0, 3, [R.data_ptr()], [R.data_ptr()], [w], [w * h])          # identical
        ident = eng.collect(0, 3)
        eng.reset()
        eng.submit_resident(0, 3, [R.data_ptr()], [D.data_ptr()], [w], [w * h])          # real pair
        pair = eng.collect(0, 3)
        Rs = R[:1].expand(3, h, w).contiguous()
        eng.reset()
        torch.cuda.synchronize()   # the context has its own stream: torch's kernels must be done
        eng.submit_resident(0, 3, [Rs.data_ptr()], [Rs.data_ptr()], [w], [w * h])        # static
        static = eng.collect(0, 3)
        c = 3
        Ro = (R.to(torch.int16).clamp(0, 255 - c) + c).to(torch.uint8)
        Rc = R.to(torch.int16).clamp(0, 255 - c).to(torch.uint8)
        eng.reset()
        torch.cuda.synchronize()
        eng.submit_resident(0, 3, [Rc.data_ptr()], [Ro.data_ptr()], [w], [w * h])        # constant offset
        off = eng.collect(0, 3)
    # vif: flat regions take the low-variance branch (num = 1 - sigma2^2 * 4/255^2, den = 1), so identical
    # frames give 1 - O(1e-6), in libvmaf too; adm is exactly num == den
    np.testing.assert_allclose(ident[:, 0:4] / ident[:, 4:8], 1.0, atol=2e-5)
    assert np.all(ident[:, 0:4] <= ident[:, 4:8])
    np.testing.assert_allclose(ident[:, 8:12] / ident[:, 12:16], 1.0, atol=1e-6)
    assert np.all(sse_from_records(ident)[:, 0] == 0) and np.allclose(ident[:, 17], 1.0)
    assert np.all(static[:, 16] == 0.0) and pair[0, 16] == 0.0 and np.all(pair[1:, 16] > 0)
    assert np.all(sse_from_records(off)[:, 0] == c * c * w * h)
    # host path over the same bytes is bit-identical (fixed-order reductions)
    with FeatureEngine(w, h, features=N.FEAT_VMAF | N.FEAT_PSNR | N.FEAT_SSIM, max_batch=3) as eng:
        Rh, Dh = R.cpu().numpy(), D.cpu().numpy()
        for i in range(3):
            eng.submit(i, [Rh[i]], [Dh[i]])
        host = eng.collect(0, 3)
    assert np.array_equal(host.view(np.uint64), pair.view(np.uint64))
    mdl = M.load_model("vmaf_v0.6.1")
    v_pair = M.score_frames(mdl, M.metrics_from_records(pair, w, h, "integer_"))["vmaf"]
    v_id = M.score_frames(mdl, M.metrics_from_records(ident, w, h, "integer_"))["vmaf"]
    assert np.all(v_pair < v_id) and np.all(v_pair > 20)


def test_one_oracle_frame_at_1080p(oracle32):
    """One full-size frame against the oracle (about 1 s of CPU)."""
    from pqa2_amd.engine import FeatureEngine
    w, h = 1920, 1080
    refs, diss = synth.make_clip(w, h, 2, 8, chroma=False)
    exp = oracle32.clip_features([r[0] for r in refs], [d[0] for d in diss], 8)
    with FeatureEngine(w, h) as eng:
        for i in range(2):
            eng.submit(i, refs[i], diss[i])
        got = eng.collect(0, 2)[:, :17]
    rel = np.abs(got[:, :16] - exp[:, :16]) / np.abs(exp[:, :16])
    # motion: the f32 oracle sums 2M pixels in float like libvmaf (~1e-6 relative); the kernel sums in double
    assert rel.max() < 5e-5 and abs(got[1, 16] - exp[1, 16]) < 5e-6 * exp[1, 16] + 2e-5


def test_fixed_point_vif_at_1080p_and_through_the_analyzer(tmp_path):
    """fixed_point at a BASELINE size: bit-equal to the fixed-point restatement on full frames (edge tiles, all four
    scales, several launches), and reachable through VMAFAnalyzer (`fixed_point`)."""
    from oracle.int_oracle import IntOracle
    from pqa2_amd.engine import FeatureEngine
    from pqa2_amd.vmaf_analyzer import VMAFAnalyzer
    into = IntOracle()
    w, h = 1920, 1080
    refs, diss = synth.make_clip(w, h, 3, 8, chroma=False)
    want = np.stack([into.vif(refs[i][0], diss[i][0]) for i in range(3)])
    with FeatureEngine(w, h, fixed_point=7, max_batch=2) as eng:
        for i in range(3):
            eng.submit(i, refs[i], diss[i])
        got = eng.collect(0, 3)
    assert np.array_equal(got[:, :8].view(np.uint64), want.view(np.uint64))
    want_adm = np.stack([into.adm(refs[i][0], diss[i][0]) for i in range(3)])
    assert np.array_equal(got[:, 8:16].view(np.uint64), want_adm.view(np.uint64))
    blur = [into.motion_blur(refs[i][0]) for i in range(3)]
    assert got[:, 16].tolist() == [0.0] + [into.motion_score(into.motion_sad(blur[i - 1], blur[i]), w, h) for i in (1, 2)]
    # analyzer: same files scored in both arithmetic modes differ by the quantisation residual only
    rp, dp, r2, d2 = _pair(tmp_path, 352, 288, 4, 8)
    scores = {}
    for fixed in (False, True):
        a = VMAFAnalyzer()
        a.set_output_directory(str(tmp_path))
        a.set_test_name("fx" if fixed else "fl")
        a.fixed_point = 7 if fixed else 0
        res = a.analyze_videos(rp, dp, "vmaf_v0.6.1")
        assert res is not None
        scores[fixed] = res
    fx_frames = scores[True]["raw_results"]["frames"]
    want_small = [into.vif(r2[i][0], d2[i][0]) for i in range(4)]
    for i in range(4):
        for s_ in range(4):
            assert abs(fx_frames[i]["metrics"][f"integer_vif_scale{s_}"] - want_small[i][s_] / want_small[i][4 + s_]) < 1e-6
    assert 0 < abs(scores[True]["vmaf_score"] - scores[False]["vmaf_score"]) < 0.01


def test_score_cli_single_and_torchrun(tmp_path):
    """python -m pqa2_amd.score, plain and under torch.distributed.run (1 rank on this 1-GPU box)."""
    import subprocess
    import sys
    rp, dp, refs, diss = _pair(tmp_path, 256, 144, 6)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    j1, j2 = str(tmp_path / "a.json"), str(tmp_path / "b.json")
    r = subprocess.run([sys.executable, "-m", "pqa2_amd.score", rp, dp, "--json", j1, "--psnr-log", str(tmp_path / "p.txt")],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "frame=" in r.stderr and "VMAF score:" in r.stderr
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
                        "127.0.0.1", "--master-port", "29533", "-m", "pqa2_amd.score", rp, dp, "--json", j2],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    a, b = json.load(open(j1)), json.load(open(j2))
    assert [f["metrics"]["vmaf"] for f in a["frames"]] == [f["metrics"]["vmaf"] for f in b["frames"]]
    assert len(open(tmp_path / "p.txt").read().strip().split("\n")) == 6
    # unreadable input -> non-zero exit and a one-line error, like an ffmpeg child
    r = subprocess.run([sys.executable, "-m", "pqa2_amd.score", rp, str(tmp_path / "missing.y4m"), "--json", j1],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "error" in r.stderr


def test_ragged_clip_lengths_use_the_shorter(tmp_path):
    from pqa2_amd.pipeline import score_files
    refs, diss = synth.make_clip(128, 96, 5, 8, chroma=True)
    info = synth.clip_info(128, 96)
    rp, dp = str(tmp_path / "r.y4m"), str(tmp_path / "d.y4m")
    yuvio.write_y4m(rp, refs, info)
    yuvio.write_y4m(dp, diss[:3], info)
    res = score_files(rp, dp, "vmaf_v0.6.1")
    assert res["records"].shape[0] == 3 and list(res["frame_indices"]) == [0, 1, 2]
    yuvio.write_y4m(dp, diss[:1], info)          # a single frame: no predecessor, no successor
    one = score_files(rp, dp, "vmaf_v0.6.1")
    assert one["records"].shape[0] == 1 and one["records"][0, 16] == 0.0
    assert np.all(np.isfinite(one["records"][:, :17]))
    assert np.array_equal(one["records"][0, :16], res["records"][0, :16])   # frame 0 scores as it does inside a clip
    yuvio.write_y4m(dp, [], info)
    with pytest.raises(ValueError):
        score_files(rp, dp, "vmaf_v0.6.1")


def test_analyzer_child_job_drives_torchrun_like_ffmpeg(tmp_path):
    """The gpus > 1 path of VMAFAnalyzer (a torch.distributed.run child, progress from stderr `frame=` lines),
    exercised here with one rank because the box has one GPU."""
    from pqa2_amd import VMAFAnalyzer
    rp, dp, refs, diss = _pair(tmp_path, 256, 144, 8)
    a = VMAFAnalyzer()
    a.set_output_directory(str(tmp_path))
    a.set_test_name("child")
    a.gpus = 1
    ev = {"p": [], "e": []}
    a.analysis_progress.connect(ev["p"].append)
    a.error_occurred.connect(ev["e"].append)
    in_proc = a.analyze_videos(rp, dp, "vmaf_v0.6.1")
    assert in_proc is not None and ev["e"] == []
    jp = str(tmp_path / "child_vmaf.json")
    ok = a._run_child_job(rp, dp, "vmaf_v0.6.1", jp, str(tmp_path / "c_psnr.txt"), str(tmp_path / "c_ssim.txt"), 8)
    assert ok and ev["e"] == []
    child = json.load(open(jp))
    assert [f["metrics"] for f in child["frames"]] == [f["metrics"] for f in in_proc["raw_results"]["frames"]]
    assert open(tmp_path / "c_psnr.txt").read() == open(in_proc["psnr_log"]).read()
    # a failing child surfaces as an error signal carrying the return code, like a failing ffmpeg
    assert not a._run_child_job(rp, str(tmp_path / "nope.y4m"), "vmaf_v0.6.1", jp, None, None, 8)
    assert "return code" in ev["e"][-1]


@pytest.mark.gpu
def test_frames_read_straight_from_files_match_frames_from_memory(tmp_path):
    """pqa_submit_fd: the library preads packed planes out of two files into its pinned staging (what analyze_videos now does
    for .y4m / .yuv inputs).  Same records, bit for bit, as the same frames handed over as memory planes -- at a geometry
    whose staging rows are padded (odd width: row-by-row reads) and at one where a plane is a single read; a truncated file is
    PQA_EINVAL, not a crash, and the context stays usable."""
    import os
    from pqa2_amd import _native as N
    from pqa2_amd import synth, yuvio
    from pqa2_amd.engine import FeatureEngine
    for w, h, bpc in ((322, 182, 8), (640, 360, 10)):
        n = 5
        refs, diss = synth.make_clip(w, h, n, bpc, chroma=True)
        info = synth.clip_info(w, h, bpc)
        rp, dp = str(tmp_path / f"r_{w}.y4m"), str(tmp_path / f"d_{w}.y4m")
        yuvio.write_y4m(rp, refs, info)
        yuvio.write_y4m(dp, diss, info)
        rr, dr = yuvio.open_video(rp), yuvio.open_video(dp)
        with FeatureEngine(w, h, bit_depth=bpc, n_planes=3, features=N.FEAT_ALL, max_batch=2) as eng:
            for i in range(n):
                eng.submit(i, refs[i], diss[i])
            mem = eng.collect(0, n)
        with FeatureEngine(w, h, bit_depth=bpc, n_planes=3, features=N.FEAT_ALL, max_batch=2) as eng:
            for i in range(n):
                eng.submit_file(i, rr.fileno(), rr.plane_offsets(i), dr.fileno(), dr.plane_offsets(i))
            fil = eng.collect(0, n)
            assert np.array_equal(mem.view(np.uint64), fil.view(np.uint64))
            # a file that ends inside the frame
            short = str(tmp_path / "short.y4m")
            with open(rp, "rb") as f, open(short, "wb") as g:
                g.write(f.read(rr.plane_offsets(0)[0] + 1000))
            fd = os.open(short, os.O_RDONLY)
            try:
                with pytest.raises(N.PqaError) as e:
                    eng.submit_file(n, fd, rr.plane_offsets(0), dr.fileno(), dr.plane_offsets(0))
                assert e.value.code == N.PQA_EINVAL and "short read" in str(e.value)
            finally:
                os.close(fd)
            eng.submit_file(n, rr.fileno(), rr.plane_offsets(1), dr.fileno(), dr.plane_offsets(1))   # still usable
            assert eng.collect(n, 1).shape == (1, N.RECORD_DOUBLES)


@pytest.mark.gpu
def test_a_run_of_frames_from_files_in_one_call_matches_frame_by_frame(tmp_path):
    """pqa_submit_fd_run: n consecutive frame pairs of two files in one call (the library overlaps reading frame k + 1 with
    the upload of frame k).  Bit-identical to frame-by-frame pqa_submit_fd whatever the run length and however the runs
    fall on the staging halves (max_batch 2 and 8: halves of 2 and 8 frames; runs of 1, 3, 5, all; a run that starts
    mid-half), with the pack threads on (4:2:0 1080p slots are > 4 MB) and off (small frames); a file that ends inside a
    run is PQA_EINVAL, nothing of the half it hit is submitted, and the context stays usable; analyze-like chunking
    through pipeline.score_files gives the same scores as before."""
    import os
    from pqa2_amd import _native as N
    from pqa2_amd import pipeline, synth, yuvio
    from pqa2_amd.engine import FeatureEngine
    for w, h, bpc, n in ((322, 182, 8, 11), (1920, 1080, 8, 11), (640, 360, 10, 7)):
        refs, diss = synth.make_clip(w, h, n, bpc, chroma=True)
        info = synth.clip_info(w, h, bpc)
        rp, dp = str(tmp_path / f"r_{w}.y4m"), str(tmp_path / f"d_{w}.y4m")
        yuvio.write_y4m(rp, refs, info)
        yuvio.write_y4m(dp, diss, info)
        rr, dr = yuvio.open_video(rp), yuvio.open_video(dp)
        rs, ds = rr.run_stride(0, n), dr.run_stride(0, n)
        assert rs == info.frame_bytes + 6 and ds == rs
        with FeatureEngine(w, h, bit_depth=bpc, n_planes=3, features=N.FEAT_ALL, max_batch=2) as eng:
            for i in range(n):
                eng.submit_file(i, rr.fileno(), rr.plane_offsets(i), dr.fileno(), dr.plane_offsets(i))
            one = eng.collect(0, n)
        for mb, runs in ((2, (1, 3, 5, n)), (8, (3, n))):
            for run in runs:
                with FeatureEngine(w, h, bit_depth=bpc, n_planes=3, features=N.FEAT_ALL, max_batch=mb) as eng:
                    i = 0
                    if run == 3:   # start mid-half: one single frame first
                        eng.submit_file(0, rr.fileno(), rr.plane_offsets(0), dr.fileno(), dr.plane_offsets(0))
                        i = 1
                    while i < n:
                        m = min(run, n - i)
                        eng.submit_file_run(i, m, rr.fileno(), rr.plane_offsets(i), rs, dr.fileno(), dr.plane_offsets(i), ds)
                        i += m
                    got = eng.collect(0, n)
                assert np.array_equal(one.view(np.uint64), got.view(np.uint64)), (w, mb, run)
        # a reference file that ends inside frame 4: the run 2..6 fails, frames 0, 1 stay collectable, the context usable
        short = str(tmp_path / f"short_{w}.y4m")
        with open(rp, "rb") as f, open(short, "wb") as g:
            g.write(f.read(rr.plane_offsets(4)[0] + 1000))
        fd = os.open(short, os.O_RDONLY)
        try:
            with FeatureEngine(w, h, bit_depth=bpc, n_planes=3, features=N.FEAT_ALL, max_batch=8) as eng:
                eng.submit_file_run(0, 2, fd, rr.plane_offsets(0), rs, dr.fileno(), dr.plane_offsets(0), ds)
                with pytest.raises(N.PqaError) as e:
                    eng.submit_file_run(2, 5, fd, rr.plane_offsets(2), rs, dr.fileno(), dr.plane_offsets(2), ds)
                assert e.value.code == N.PQA_EINVAL and "short read" in str(e.value)
                assert np.array_equal(eng.collect(0, 2).view(np.uint64), one[:2].view(np.uint64))
                with pytest.raises(N.PqaError) as e:
                    eng.collect(2, 1)                      # never submitted
                assert e.value.code == N.PQA_ESTATE
                eng.submit_file_run(2, n - 2, rr.fileno(), rr.plane_offsets(2), rs, dr.fileno(), dr.plane_offsets(2), ds)
                assert np.array_equal(eng.collect(2, n - 2).view(np.uint64), one[2:].view(np.uint64))
        finally:
            os.close(fd)
        if w == 322:   # the caller of the run entry point: score_files in runs of eight against frame by frame (PQA_FD_RUN=0)
            res = pipeline.score_files(rp, dp, "vmaf_v0.6.1")
            os.environ["PQA_FD_RUN"] = "0"
            try:
                res1 = pipeline.score_files(rp, dp, "vmaf_v0.6.1")
            finally:
                os.environ.pop("PQA_FD_RUN")
            assert np.array_equal(res["records"].view(np.uint64), res1["records"].view(np.uint64))


@pytest.mark.gpu
def test_a_parked_context_is_reused_and_scores_the_same(tmp_path):
    """pipeline.score_files() parks its context (FeatureEngine.release) and the next analysis of the same configuration takes
    it back through pqa_reset instead of creating one: same records bit for bit, whatever ran in between -- a different clip
    of the same geometry (motion continuity must not leak: the first frame's motion is 0 again), a bookend pass that switched
    the gray mode, a different geometry (the parked context is replaced, never two), PQA_CONTEXT_CACHE=0 (nothing parked)."""
    import os
    from pqa2_amd import _native as N
    from pqa2_amd import engine as E
    from pqa2_amd import pipeline, synth, yuvio
    E.clear_parked()

    def clip(tag, w, h, n, seed):
        refs, diss = synth.make_clip(w, h, n, 8, chroma=True)
        info = synth.clip_info(w, h, 8)
        rp, dp = str(tmp_path / f"r_{tag}.y4m"), str(tmp_path / f"d_{tag}.y4m")
        yuvio.write_y4m(rp, refs if seed == 0 else refs[::-1], info)
        yuvio.write_y4m(dp, diss if seed == 0 else diss[::-1], info)
        return rp, dp

    a = clip("a", 320, 180, 9, 0)
    b = clip("b", 320, 180, 9, 1)      # same geometry, other content (the frames in reverse order)
    c = clip("c", 256, 144, 5, 0)
    first = pipeline.score_files(*a, "vmaf_v0.6.1")["records"]
    assert len(E._PARKED) == 1
    other = pipeline.score_files(*b, "vmaf_v0.6.1")["records"]
    assert len(E._PARKED) == 1 and other[0, N.REC_MOTION] == 0.0 and not np.array_equal(other, first)
    again = pipeline.score_files(*a, "vmaf_v0.6.1")["records"]
    assert np.array_equal(first.view(np.uint64), again.view(np.uint64))
    key_a = next(iter(E._PARKED))
    pipeline.score_files(*c, "vmaf_v0.6.1")
    assert len(E._PARKED) == 1 and next(iter(E._PARKED)) != key_a
    os.environ["PQA_CONTEXT_CACHE"] = "0"
    try:
        E.clear_parked()
        cold = pipeline.score_files(*a, "vmaf_v0.6.1")["records"]
        assert len(E._PARKED) == 0
    finally:
        os.environ.pop("PQA_CONTEXT_CACHE")
    assert np.array_equal(first.view(np.uint64), cold.view(np.uint64))
    E.clear_parked()
