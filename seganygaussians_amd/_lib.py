"""ctypes binding of the C-ABI library (include/mi_rast.h).  Fails loudly when the HIP extension is
missing: there is NO CPU / PyTorch fallback in the product path."""
from __future__ import annotations

import ctypes as C
import os

from .build import LIB_PATH

RESIZE_FN = C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_void_p)

MI_GEOM_FIELDS = ["depths", "means2D", "conic_opacity", "cov3D", "rgb", "clamped", "tiles_touched",
                  "depth_key", "index_rec", "cull_counter", "band_bits", "bwd_pack"]
MI_IMG_FIELDS = ["final_T", "n_contrib", "ranges", "tile_consumed", "tile_count", "tile_cursor", "num_rendered", "tile_nsurv"]
MI_BIN_FIELDS = ["entries", "scratch", "blend_list"]
MI_RAST_FULL_LISTS, MI_RAST_F32_BLEND, MI_RAST_NO_CULL, MI_RAST_FAST_EXP, MI_RAST_VERIFY_LISTS, MI_RAST_TILE_FWD = 1, 2, 4, 8, 16, 32   # `flags` of mi_rast_forward (include/mi_rast.h)
MI_RAST_PREZERO_BWD = 64   # forward + the one backward of that forward (include/mi_rast.h)
MI_RAST_EXACT_EXP = 128    # forward blend with expf for every pair instead of the hybrid form (include/mi_rast.h)
MI_RAST_EQUAL_RUNS = 256   # A/B aid: XCD runs of equal tile counts in both blend kernels instead of equal modelled work (include/mi_rast.h)
MI_RAST_BWD_FEATURES_ONLY = 512   # mi_rast_backward: dL_dcolor alone (extension, include/mi_rast.h)
MI_STAGES = ["preprocess", "tile_scan", "emit", "tile_sort", "blend_fwd", "blend_bwd", "geom_bwd"]

EXPORTS = [
    "mi_rast_forward", "mi_rast_forward_reuse", "mi_rast_last_longest_run", "mi_rast_fingerprint", "mi_rast_features_only_supported", "mi_rast_backward", "mi_rast_mark_visible", "mi_rast_mask_forward",
    "mi_rast_mask_backward", "mi_rast_last_error", "mi_rast_version", "mi_rast_supported_channels",
    "mi_rast_get_higher_msb", "mi_rast_geometry_layout", "mi_rast_image_layout", "mi_rast_binning_layout",
    "mi_rast_profile_enable", "mi_rast_profile_read",
    "mi_knn_smooth_forward", "mi_knn_smooth_backward",  # include/mi_knn_smooth.h
    "mi_knn_workspace_bytes", "mi_knn_build", "mi_knn_query", "mi_knn_mean_dist2",  # include/mi_knn.h
    "mi_contrastive_forward", "mi_contrastive_backward",  # include/mi_contrastive.h
]

_lib = None


class MiRastError(RuntimeError):
    pass


def load():
    """Returns the loaded library; raises MiRastError if libmi_rast.so is absent or unloadable."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own HIP runtime; load it first so that this library binds to the same one (loading
    # /opt/rocm's copy first leaves the process with a runtime that sees no device once torch initialises its own)
    import torch  # noqa: F401
    lib_path = os.environ.get("MI_RAST_LIB") or LIB_PATH   # MI_RAST_LIB: e.g. the profiling build (build.py), for tools/
    if not os.path.exists(lib_path):
        raise MiRastError(
            f"HIP extension {lib_path} is missing. Build it with `python -m seganygaussians_amd.build` "
            "(or __graft_entry__.build()). There is no CPU fallback.")
    try:
        L = C.CDLL(lib_path)
    except OSError as e:  # e.g. libamdhip64 missing
        raise MiRastError(f"cannot load {lib_path}: {e}") from e
    vp, i, f = C.c_void_p, C.c_int, C.c_float
    L.mi_rast_forward.restype = i
    L.mi_rast_forward.argtypes = [RESIZE_FN, vp, RESIZE_FN, vp, RESIZE_FN, vp, i, i, i, i, vp, i, i,
                                  vp, vp, vp, vp, vp, f, vp, vp, vp, vp, vp, f, f, i,
                                  vp, vp, vp, vp, vp, i, i, vp, vp, vp, C.POINTER(i)]
    L.mi_rast_forward_reuse.restype = i
    L.mi_rast_forward_reuse.argtypes = [i, i, i, vp, i, i, vp, vp, vp, vp, vp, vp, i, vp, vp, vp, vp, i, vp, vp, vp]
    L.mi_rast_fingerprint.restype = i
    L.mi_rast_fingerprint.argtypes = [i, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), vp]
    L.mi_rast_last_longest_run.restype = i
    L.mi_rast_last_longest_run.argtypes = []
    L.mi_rast_features_only_supported.restype = i
    L.mi_rast_features_only_supported.argtypes = [i]
    L.mi_rast_backward.restype = i
    L.mi_rast_backward.argtypes = [i, i, i, i, i, vp, i, i, vp, vp, vp, vp, f, vp, vp, vp, vp, vp, f, f,
                                   vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, vp]
    L.mi_rast_mark_visible.restype = i
    L.mi_rast_mark_visible.argtypes = [i, vp, vp, vp, vp, vp]
    L.mi_rast_mask_forward.restype = i
    L.mi_rast_mask_forward.argtypes = [RESIZE_FN, vp, RESIZE_FN, vp, RESIZE_FN, vp, i, i, i, vp, vp, vp, vp, f,
                                       vp, vp, vp, vp, f, f, i, vp, vp, i, i, vp, C.POINTER(i)]
    L.mi_rast_mask_backward.restype = i
    L.mi_rast_mask_backward.argtypes = [i, i, i, i, vp, vp, vp, vp, vp, i, i, vp]
    L.mi_rast_last_error.restype = C.c_char_p
    L.mi_rast_version.restype = C.c_char_p
    L.mi_rast_supported_channels.restype = i
    L.mi_rast_supported_channels.argtypes = [C.POINTER(i), i]
    L.mi_rast_get_higher_msb.restype = C.c_uint32
    L.mi_rast_get_higher_msb.argtypes = [C.c_uint32]
    for name, n in (("mi_rast_geometry_layout", 1), ("mi_rast_binning_layout", 1)):
        fn = getattr(L, name)
        fn.restype = C.c_size_t
        fn.argtypes = [i, C.POINTER(C.c_size_t)]
    L.mi_rast_image_layout.restype = C.c_size_t
    L.mi_rast_image_layout.argtypes = [i, i, C.POINTER(C.c_size_t)]
    L.mi_rast_profile_enable.restype = i
    L.mi_rast_profile_enable.argtypes = [i]
    L.mi_rast_profile_read.restype = i
    L.mi_rast_profile_read.argtypes = [C.POINTER(f)]
    u32 = C.c_uint32
    L.mi_knn_smooth_forward.restype = i
    L.mi_knn_smooth_forward.argtypes = [i, i, i, vp, u32, vp, vp, i, vp]
    L.mi_knn_smooth_backward.restype = i
    L.mi_knn_smooth_backward.argtypes = [i, i, i, vp, vp, vp, u32, vp, vp, vp, vp, i, vp]
    L.mi_knn_workspace_bytes.restype = C.c_size_t
    L.mi_knn_workspace_bytes.argtypes = [i]
    L.mi_knn_build.restype = i
    L.mi_knn_build.argtypes = [i, vp, vp, C.c_size_t, vp]
    L.mi_knn_query.restype = i
    L.mi_knn_query.argtypes = [i, vp, i, vp, i, i, vp, vp, vp]
    L.mi_knn_mean_dist2.restype = i
    L.mi_knn_mean_dist2.argtypes = [i, vp, vp, C.c_size_t, vp, vp]
    L.mi_contrastive_forward.restype = i
    L.mi_contrastive_forward.argtypes = [i, i, i, vp, i, i, i, vp, i, vp, vp, vp, vp, vp, vp, vp]
    L.mi_contrastive_backward.restype = i
    L.mi_contrastive_backward.argtypes = [i, i, i, vp, i, i, i, vp, i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    _lib = L
    return L


def last_error() -> str:
    return load().mi_rast_last_error().decode("utf-8", "replace")


def geometry_layout(P: int):
    off = (C.c_size_t * len(MI_GEOM_FIELDS))()
    total = load().mi_rast_geometry_layout(int(P), off)
    return int(total), dict(zip(MI_GEOM_FIELDS, [int(o) for o in off]))


def image_layout(W: int, H: int):
    off = (C.c_size_t * len(MI_IMG_FIELDS))()
    total = load().mi_rast_image_layout(int(W), int(H), off)
    return int(total), dict(zip(MI_IMG_FIELDS, [int(o) for o in off]))


def binning_layout(R: int):
    off = (C.c_size_t * len(MI_BIN_FIELDS))()
    total = load().mi_rast_binning_layout(int(R), off)
    return int(total), dict(zip(MI_BIN_FIELDS, [int(o) for o in off]))


def profile_enable(on: bool) -> None:
    if load().mi_rast_profile_enable(1 if on else 0) != 0:
        raise MiRastError(last_error())


def profile_read() -> dict:
    ms = (C.c_float * len(MI_STAGES))()
    if load().mi_rast_profile_read(ms) != 0:
        raise MiRastError(last_error())
    return dict(zip(MI_STAGES, [float(x) for x in ms]))
