"""tools/compare_libvmaf_log.py: the one-command pinning tool.  No libvmaf exists here, so the test feeds it a log made
from this repo's own restatement (must report a match) and a doctored copy (must name the mismatch and the VERIFY items
it implicates).  The committed .y4m clips must be the exact bytes the golden fixtures were computed from."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLIPS = os.path.join(ROOT, "tests", "golden", "clips")


def _log_from_restatement(tmp_path, name, w, h, integer=True):
    from oracle.int_oracle import IntOracle
    from oracle.oracle import Oracle
    from pqa2_amd import model as M, report
    from pqa2_amd.yuvio import open_video
    rd, dd = open_video(os.path.join(CLIPS, f"{name}_ref.y4m")), open_video(os.path.join(CLIPS, f"{name}_dist.y4m"))
    refs = [np.asarray(rd.frame(i)[0]) for i in range(len(rd))]
    diss = [np.asarray(dd.frame(i)[0]) for i in range(len(dd))]
    rec = np.zeros((len(refs), 24))
    if integer:
        rec[:, :17] = IntOracle().clip_features(refs, diss, 8)
        mdl, prefix = M.load_model("vmaf_v0.6.1"), "integer_"
    else:
        rec[:, :17] = Oracle("f32").clip_features(refs, diss, 8)
        mdl, prefix = M.load_model("vmaf_float_v0.6.1"), ""
    m = M.score_frames(mdl, M.metrics_from_records(rec, w, h, prefix))
    p = str(tmp_path / f"{name}_{'int' if integer else 'float'}.json")
    report.write_vmaf_json(p, report.build_vmaf_log(m, 30.0))
    return p


def _run(args):
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "compare_libvmaf_log.py")] + args,
                          capture_output=True, text=True, timeout=300, env=dict(os.environ, PYTHONPATH=ROOT))


def test_committed_clips_are_the_golden_inputs():
    from pqa2_amd.yuvio import open_video
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_features.json")))["cases"]
    for name in ("c64x48_8", "c352x288_8", "c200x120_10"):
        rd, dd = open_video(os.path.join(CLIPS, f"{name}_ref.y4m")), open_video(os.path.join(CLIPS, f"{name}_dist.y4m"))
        assert len(rd) == len(dd) == g[name]["n"] and (rd.info.width, rd.info.height) == (g[name]["w"], g[name]["h"])
        sha = hashlib.sha256()
        for r in (rd, dd):
            for i in range(len(r)):
                for p in r.frame(i):
                    sha.update(np.ascontiguousarray(p).tobytes())
        assert sha.hexdigest() == g[name]["input_sha256"]


def test_tool_reports_match_and_mismatch(tmp_path):
    ref, dis = os.path.join(CLIPS, "c64x48_8_ref.y4m"), os.path.join(CLIPS, "c64x48_8_dist.y4m")
    log = _log_from_restatement(tmp_path, "c64x48_8", 64, 48, integer=True)
    r = _run([log, ref, dis, "--tol", "5e-2"])      # f32-vs-fixed differs by design at this size: wide bar for (a)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "fixed-point restatement" in r.stdout and "every compared metric inside its bar" in r.stdout
    d = json.load(open(log))
    d["frames"][1]["metrics"]["integer_adm_scale0"] += 1e-3
    bad = str(tmp_path / "bad.json")
    json.dump(d, open(bad, "w"))
    r = _run([bad, ref, dis, "--tol", "5e-2"])
    assert r.returncode == 1 and "MISMATCH" in r.stdout and "vmaf_int_oracle.c:337" in r.stdout
    # float keys: compared against the f32 restatement with the vif_tools.c border only
    flog = _log_from_restatement(tmp_path, "c64x48_8", 64, 48, integer=False)
    r = _run([flog, ref, dis])
    assert r.returncode == 0 and "vif_tools.c border" in r.stdout and "fixed-point restatement" not in r.stdout


def test_ffmpeg_stats_tool_on_its_own_output(tmp_path):
    """tools/compare_ffmpeg_stats.py (row a5): fed stats files written by this repository's own writers it must report a
    match; a changed digit in a PSNR line and a moved SSIM field must both be flagged."""
    from oracle.oracle import Oracle
    from pqa2_amd import report
    from pqa2_amd.yuvio import open_video
    ref, dis = os.path.join(CLIPS, "c200x120_10_ref.y4m"), os.path.join(CLIPS, "c200x120_10_dist.y4m")
    rd, dd = open_video(ref), open_video(dis)
    info, n = rd.info, len(rd)
    assert info.bit_depth == 10
    sizes = [(info.width, info.height), (info.chroma_w, info.chroma_h), (info.chroma_w, info.chroma_h)]
    orc = Oracle("f32")
    sse = np.array([[orc.sse_plane(dd.frame(i)[p], rd.frame(i)[p], 10) for p in range(3)] for i in range(n)], np.uint64)
    ssim = np.array([[orc.ssim_plane(dd.frame(i)[p], rd.frame(i)[p], 10) for p in range(3)] for i in range(n)])
    pp, sp = str(tmp_path / "psnr.txt"), str(tmp_path / "ssim.txt")
    open(pp, "w").write("\n".join(report.psnr_stats_lines(sse, sizes, 10)) + "\n")
    open(sp, "w").write("\n".join(report.ssim_stats_lines(ssim, sizes)) + "\n")
    tool = os.path.join(ROOT, "tools", "compare_ffmpeg_stats.py")
    run = lambda args: subprocess.run([sys.executable, tool] + args, capture_output=True, text=True, timeout=300,
                                      env=dict(os.environ, PYTHONPATH=ROOT))
    r = run(["--psnr", pp, "--ssim", sp, ref, dis])
    assert r.returncode == 0 and "IDENTICAL text" in r.stdout and "row a5 is pinned" in r.stdout, r.stdout + r.stderr
    lines = open(pp).read().split("\n")
    lines[0] = lines[0].replace("mse_avg:", "mse_avg:1", 1)
    open(pp, "w").write("\n".join(lines))
    ssim[1, 0] += 1e-4
    open(sp, "w").write("\n".join(report.ssim_stats_lines(ssim, sizes)) + "\n")
    r = run(["--psnr", pp, "--ssim", sp, ref, dis])
    assert r.returncode == 1 and r.stdout.count("MISMATCH") >= 3
