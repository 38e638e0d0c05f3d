"""The C-ABI library loads on a GPU-less host and exports every symbol include/pqa_vmaf.h declares.
No compute calls here: without a device pqa_create must fail loudly (there is no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "pqa_vmaf.h")).read()
    return sorted(set(re.findall(r"PQA_API\s+[\w\s\*]+?\b(pqa_\w+)\s*\(", src)))


def test_header_and_binding_agree():
    from pqa2_amd import _native as N
    assert _declared() == sorted(N.EXPORTS)
    src = open(os.path.join(ROOT, "include", "pqa_vmaf.h")).read()
    assert int(re.search(r"PQA_RECORD_DOUBLES\s*=\s*(\d+)", src).group(1)) == N.RECORD_DOUBLES
    for name, val in (("PQA_REC_MOTION", N.REC_MOTION), ("PQA_REC_SSIM", N.REC_SSIM), ("PQA_REC_SSE", N.REC_SSE)):
        assert int(re.search(name + r"\s*=\s*(\d+)", src).group(1)) == val


def test_library_exports_every_declared_symbol():
    from pqa2_amd import _native as N
    if not os.path.exists(N.LIB_PATH):
        N.build()
    lib = N.load()
    for sym in _declared():
        assert hasattr(lib, sym), sym
    assert b"gfx950" in lib.pqa_version()
    assert lib.pqa_record_doubles() == 24
    cfg = N.PqaConfig()
    lib.pqa_config_init(C.byref(cfg), 1920, 1080)
    assert (cfg.struct_size, cfg.width, cfg.height, cfg.bit_depth, cfg.features) == (C.sizeof(N.PqaConfig), 1920, 1080, 8, N.FEAT_VMAF)
    assert cfg.vif_enhn_gain_limit == 100.0 and cfg.adm_enhn_gain_limit == 100.0
    assert lib.pqa_profile_kernel_name(0) == b"vif_stat_s0"


def test_create_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from pqa2_amd import _native as N
    from pqa2_amd.engine import FeatureEngine
    with pytest.raises(N.PqaError) as e:
        FeatureEngine(64, 64)
    assert e.value.code == N.PQA_EDEVICE and "no CPU fallback" in str(e.value)


def test_bad_arguments_do_not_crash():
    from pqa2_amd import _native as N
    lib = N.load()
    assert lib.pqa_create(None, None) == N.PQA_EINVAL
    cfg = N.PqaConfig()
    lib.pqa_config_init(C.byref(cfg), 8, 8)
    ctx = C.c_void_p()
    assert lib.pqa_create(C.byref(cfg), C.byref(ctx)) == N.PQA_EINVAL and b"frame size" in lib.pqa_last_error(None)
    lib.pqa_destroy(None)
    assert lib.pqa_cancel(None) == N.PQA_EINVAL


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/pqa_vmaf.h must be consumable by a C compiler (the boundary is a C ABI, not C++): compile a C99
    program against it with gcc, link it to the library, run it on this GPU-less host.  It checks the struct
    layout the Python binding assumes and that creation fails with PQA_EDEVICE/EINVAL instead of crashing."""
    import shutil
    import subprocess
    from pqa2_amd import _native as N
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    if not os.path.exists(N.LIB_PATH):
        N.build()
    src = tmp_path / "abi.c"
    src.write_text(r'''
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include "pqa_vmaf.h"
int main(void) {
  pqa_config cfg;
  pqa_ctx* ctx = NULL;
  int rc;
  pqa_config_init(&cfg, 1920, 1080);
  printf("%zu %zu %zu %zu %d %u\n", sizeof(pqa_config), offsetof(pqa_config, vif_enhn_gain_limit),
         offsetof(pqa_config, vif_border), offsetof(pqa_config, fixed_point), PQA_RECORD_DOUBLES, cfg.struct_size);
  cfg.fixed_point = PQA_FIXED_ALL;
  cfg.vif_border = PQA_VIF_BORDER_INTEGER;
  rc = pqa_create(&cfg, &ctx);
  printf("%d %s\n", rc, pqa_last_error(NULL));
  if (ctx) pqa_destroy(ctx);
  return strstr(pqa_version(), "gfx950") ? 0 : 1;
}
''')
    exe = tmp_path / "abi"
    libdir = os.path.dirname(N.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src),
                    "-o", str(exe), "-L", libdir, "-lpqa_vmaf", f"-Wl,-rpath,{libdir}"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    first, second = r.stdout.strip().split("\n")
    size, off_gain, off_border, off_fixed, rec, struct_size = map(int, first.split())
    assert size == C.sizeof(N.PqaConfig) == struct_size and rec == 24
    assert off_gain == N.PqaConfig.vif_enhn_gain_limit.offset
    assert off_border == N.PqaConfig.vif_border.offset and off_fixed == N.PqaConfig.fixed_point.offset
    import torch
    if not torch.cuda.is_available():
        assert second.startswith(str(N.PQA_EDEVICE)) and "no CPU fallback" in second


def test_integration_doc_binding_matches_the_header():
    """INTEGRATION.md shows the ctypes stub a PQA2 maintainer would add; its pqa_config field list must be the
    header's (and so the binding's), in order -- a drifted doc would corrupt memory in pqa_config_init."""
    from pqa2_amd import _native as N
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = doc[doc.index("class PqaConfig(C.Structure):"):doc.index("lib = C.CDLL")]
    doc_fields = re.findall(r'\("(\w+)",\s*C\.c_(\w+)\)', block)
    want = [(n, t.__name__.replace("c_", "")) for n, t in N.PqaConfig._fields_]
    assert [(n, {"uint": "uint32", "int": "int32"}.get(t, t)) for n, t in doc_fields] == \
           [(n, {"uint": "uint32", "int": "int32"}.get(t, t)) for n, t in want]
    hdr = open(os.path.join(ROOT, "include", "pqa_vmaf.h")).read()
    struct = hdr[hdr.index("typedef struct pqa_config {"):hdr.index("} pqa_config;")]
    hdr_fields = []
    for decl in re.findall(r"^\s*(?:uint32_t|int32_t|double)\s+([\w\s,]+);", struct, re.M):
        hdr_fields += [f.strip() for f in decl.split(",")]
    assert hdr_fields == [n for n, _ in N.PqaConfig._fields_]


def test_march_kernels_fit_three_waves_per_simd_without_scratch(tmp_path):
    """vif_s0_march_kernel keeps a block's loads in flight while the previous block is computed.  A register spill in it is
    not just slow: the reload's s_waitcnt vmcnt(0) also waits for the prefetch and serialises the march (round 3 lost 10 %
    that way to two spare dwords in the prefetch registers).  hipcc cross-compiles here: check the compiler's own figures."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "pqa2_amd", "csrc", "vif_march.hip")
    out = tmp_path / "march.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", str(out)],
                   check=True, capture_output=True)
    text = out.read_text()
    kernels = re.findall(r"^; Function info:.*?^; Occupancy: (\d+)", text, re.S | re.M)
    scratch = [int(x) for x in re.findall(r"^; ScratchSize: (\d+)", text, re.M)]
    vgprs = [int(x) for x in re.findall(r"^; TotalNumVgprs: (\d+)", text, re.M)]
    assert len(scratch) >= 3 and all(s == 0 for s in scratch), scratch          # the 8-, 10- and 12-bit instances
    assert all(v <= 168 for v in vgprs), vgprs                                   # 512 / 3 waves, 8-register granules
    assert "Folded Reload" not in text and "Folded Spill" not in text


def test_round4_march_kernels_keep_their_occupancy_without_scratch(tmp_path):
    """adm_pyramid_kernel carries two scales' state per lane; its per-thread double sums were moved to LDS precisely so that
    it fits three waves per SIMD (<= 168 VGPRs) without spilling -- with them in registers the compiler spilled 36 dwords at
    three waves and two waves measured 5 % slower.  adm_march_kernel and motion_march_kernel hide their memory latency with
    five-plus waves per SIMD (<= 96 VGPRs) and must never touch scratch (the f32 EDGE path once did: register shuffling
    through scratch_store_dwordx3).  Built with the flags build.sh uses for these files."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    for name, max_vgprs, n_inst in (("adm_pyramid", 168, 2), ("adm_march", 96, 3), ("motion_march", 96, 2)):
        src = os.path.join(ROOT, "pqa2_amd", "csrc", name + ".hip")
        out = tmp_path / (name + ".s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-S", "--cuda-device-only", src,
                        "-o", str(out)], check=True, capture_output=True)
        text = out.read_text()
        scratch = [int(x) for x in re.findall(r"^; ScratchSize: (\d+)", text, re.M)]
        vgprs = [int(x) for x in re.findall(r"^; TotalNumVgprs: (\d+)", text, re.M)]
        assert len(scratch) >= n_inst and all(s == 0 for s in scratch), (name, scratch)
        assert all(v <= max_vgprs for v in vgprs), (name, vgprs)
        assert "Folded Reload" not in text and "Folded Spill" not in text, name
