"""ThreadSanitizer and AddressSanitizer + UBSan over the library's own host code (SURVEY section 5 "race detection /
sanitizers").  The device-free parts of csrc/pqa_api.hip -- the pack pool that preads frames into staging
(pqa2_amd/csrc/host_pack.h) and the record-ring bookkeeping (host_ring.h) -- are headers of their own; tests/host_harness.cpp
drives them on the CPU through the call sequences the GPU tests use (memory and file sources, runs of frames with in-order
hand-over, a truncated file, a failing upload callback, a cancel flag from another thread, the PQA_ESTATE sequences, ring
wrap).  GPU AddressSanitizer is not available on this pool, so this is where the host code gets its sanitizer coverage.
Memory / race safety only -- nothing here is a parity claim."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_harness.cpp")


def _build(tmp_path, san):
    exe = str(tmp_path / f"host_harness_{san.replace(',', '_')}")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", f"-fsanitize={san}", "-fno-omit-frame-pointer", "-pthread",
                        "-Wall", "-Wextra", "-o", exe, SRC], capture_output=True, text=True, timeout=300)
    if r.returncode != 0 and ("cannot find" in r.stderr or "unrecognized" in r.stderr):
        pytest.skip(f"g++ has no -fsanitize={san} runtime here: {r.stderr[-200:]}")
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
@pytest.mark.parametrize("san,selftest,marker", [("thread", "race", "data race"),
                                                 ("address,undefined", "overflow", "heap-buffer-overflow")])
def test_host_code_is_clean_under(tmp_path, san, selftest, marker):
    exe = _build(tmp_path, san)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66", ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1")
    # the sanitizer must be alive: it has to flag a deliberate defect of its kind ...
    bad = subprocess.run([exe, "--selftest", selftest], capture_output=True, text=True, timeout=120, env=env)
    assert bad.returncode != 0 and marker in bad.stderr, (bad.returncode, bad.stderr[-500:])
    # ... and find nothing in the real code
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "host harness ok" in r.stdout, (r.returncode, r.stdout[-300:], r.stderr[-3000:])
    assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
