"""ctypes binding of libpqa_vmaf.so (include/pqa_vmaf.h).  No fallback: if the HIP extension is
missing or fails to load, importing callers get a loud ImportError/RuntimeError."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# PQA_LIB_PATH: load another build of the SAME library (tools/build_variant.sh) -- A/B parity and timing runs only
LIB_PATH = os.environ.get("PQA_LIB_PATH") or os.path.join(_HERE, "csrc", "libpqa_vmaf.so")

PQA_OK, PQA_EINVAL, PQA_EDEVICE, PQA_ENOMEM, PQA_ECANCELLED, PQA_ESTATE = 0, -1, -2, -3, -4, -5
FEAT_VIF, FEAT_ADM, FEAT_MOTION, FEAT_PSNR, FEAT_SSIM = 1, 2, 4, 8, 16
FEAT_VMAF = FEAT_VIF | FEAT_ADM | FEAT_MOTION
FEAT_ALL = FEAT_VMAF | FEAT_PSNR | FEAT_SSIM
VIF_BORDER_FLOAT, VIF_BORDER_INTEGER = 0, 1  # pqa_config.vif_border (include/pqa_vmaf.h)
FIXED_VIF, FIXED_MOTION, FIXED_ADM, FIXED_ALL = 1, 2, 4, 7   # pqa_config.fixed_point mask
REC_VIF_NUM, REC_VIF_DEN, REC_ADM_NUM, REC_ADM_DEN, REC_MOTION, REC_SSIM, REC_SSE = 0, 4, 8, 12, 16, 17, 20
RECORD_DOUBLES = 24
PROF_KERNELS = 15
GRAY_LUMA, GRAY_BT601_FULL = 0, 1   # pqa_set_luma_gray

# every symbol include/pqa_vmaf.h declares
EXPORTS = [
    "pqa_version", "pqa_record_doubles", "pqa_config_init", "pqa_create", "pqa_destroy", "pqa_set_stream",
    "pqa_submit", "pqa_submit_fd", "pqa_submit_fd_run", "pqa_submit_device", "pqa_submit_surfaces", "pqa_set_motion_halo", "pqa_flush", "pqa_collect", "pqa_sync",
    "pqa_cancel", "pqa_reset", "pqa_last_error", "pqa_luma_stats_device", "pqa_luma_stats", "pqa_set_luma_gray",
    "pqa_profile_enable",
    "pqa_profile_read", "pqa_profile_kernel_name", "pqa_debug_vif_march_table", "pqa_debug_vif_march_shape",
]


class PqaConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("device", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32),
        ("bit_depth", C.c_uint32), ("n_planes", C.c_uint32), ("chroma_hshift", C.c_uint32),
        ("chroma_vshift", C.c_uint32), ("features", C.c_uint32), ("max_batch", C.c_uint32),
        ("result_capacity", C.c_uint32), ("n_subsample", C.c_uint32),
        ("vif_enhn_gain_limit", C.c_double), ("adm_enhn_gain_limit", C.c_double),
        ("vif_border", C.c_uint32), ("fixed_point", C.c_uint32),
    ]


class PqaDeviceClip(C.Structure):
    _fields_ = [("plane", C.c_void_p * 3), ("row_pitch", C.c_int64 * 3), ("frame_pitch", C.c_int64 * 3)]


SURFACE_NV12, SURFACE_P01X = 1, 2


class PqaSurfaceClip(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("format", C.c_uint32), ("luma", C.c_void_p), ("chroma", C.c_void_p),
                ("luma_row_pitch", C.c_int64), ("luma_frame_pitch", C.c_int64),
                ("chroma_row_pitch", C.c_int64), ("chroma_frame_pitch", C.c_int64)]


class PqaError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"pqa error {code}: {msg}")
        self.code = code


class PqaCancelled(PqaError):
    pass


def build(force: bool = False) -> str:
    """Compile libpqa_vmaf.so for gfx950 in-tree with hipcc (cross-compiles without a GPU)."""
    script = os.path.join(_HERE, "csrc", "build.sh")
    if force:
        for f in os.listdir(os.path.join(_HERE, "csrc")):
            if f.endswith(".o"):
                os.remove(os.path.join(_HERE, "csrc", f))
    r = subprocess.run(["bash", script], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc build failed:\n{r.stdout}\n{r.stderr}")
    return LIB_PATH


_lib = None


def load():
    """Load the HIP extension.  Raises ImportError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the MI355X HIP extension is not built (run `python -c 'import "
            f"__graft_entry__ as g; g.build()'` or pqa2_amd/csrc/build.sh).  There is no CPU fallback.")
    # torch wheels bundle their own libamdhip64.so (same SONAME).  Import torch FIRST so that the one HIP
    # runtime in this process is torch's: device pointers and streams are then shared between torch
    # (memory, RCCL) and the kernels.  Loading ours first would leave torch unable to see the GPU.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch-less consumers use the system ROCm runtime
        pass
    lib = C.CDLL(LIB_PATH)
    missing = [s for s in EXPORTS if not hasattr(lib, s)]
    if missing:
        raise ImportError(f"{LIB_PATH} lacks symbols {missing}")
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    lib.pqa_version.restype = C.c_char_p
    lib.pqa_record_doubles.restype = C.c_int
    lib.pqa_config_init.argtypes = [C.POINTER(PqaConfig), C.c_uint32, C.c_uint32]
    lib.pqa_config_init.restype = None
    lib.pqa_create.argtypes = [C.POINTER(PqaConfig), C.POINTER(vp)]
    lib.pqa_destroy.argtypes = [vp]
    lib.pqa_destroy.restype = None
    lib.pqa_set_stream.argtypes = [vp, vp]
    lib.pqa_submit.argtypes = [vp, i64, C.POINTER(vp * 3), C.POINTER(i64 * 3), C.POINTER(vp * 3), C.POINTER(i64 * 3)]
    lib.pqa_submit_fd.argtypes = [vp, i64, C.c_int, C.POINTER(i64 * 3), C.c_int, C.POINTER(i64 * 3)]
    lib.pqa_submit_fd_run.argtypes = [vp, i64, i32, C.c_int, C.POINTER(i64 * 3), i64, C.c_int, C.POINTER(i64 * 3), i64]
    lib.pqa_submit_device.argtypes = [vp, i64, i32, C.POINTER(PqaDeviceClip), C.POINTER(PqaDeviceClip), vp, i64]
    lib.pqa_submit_surfaces.argtypes = [vp, i64, i32, C.POINTER(PqaSurfaceClip), C.POINTER(PqaSurfaceClip),
                                        C.POINTER(PqaSurfaceClip)]
    lib.pqa_set_motion_halo.argtypes = [vp, vp, i64]
    lib.pqa_flush.argtypes = [vp]
    lib.pqa_collect.argtypes = [vp, i64, i32, vp]
    lib.pqa_sync.argtypes = [vp]
    lib.pqa_cancel.argtypes = [vp]
    lib.pqa_luma_stats_device.argtypes = [vp, vp, i64, i64, i32, C.c_uint32, vp]
    lib.pqa_luma_stats.argtypes = [vp, C.POINTER(vp), i64, i32, C.c_uint32, vp]
    lib.pqa_set_luma_gray.argtypes = [vp, C.c_uint32]
    lib.pqa_reset.argtypes = [vp]
    lib.pqa_last_error.argtypes = [vp]
    lib.pqa_last_error.restype = C.c_char_p
    lib.pqa_profile_enable.argtypes = [vp, C.c_int]
    lib.pqa_profile_read.argtypes = [vp, C.c_int, C.POINTER(dbl), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.pqa_debug_vif_march_table.argtypes = [vp, i32]
    lib.pqa_debug_vif_march_shape.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(i32 * 6)]
    lib.pqa_profile_kernel_name.argtypes = [C.c_int]
    lib.pqa_profile_kernel_name.restype = C.c_char_p
    assert lib.pqa_record_doubles() == RECORD_DOUBLES
    _lib = lib
    return lib
