"""Builds the HIP C-ABI library in-tree (seganygaussians_amd/libmi_rast.so) with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the resulting
.so is git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmi_rast.so")
PROF_LIB_PATH = os.path.join(_HERE, "libmi_rast_prof.so")
SRC_DIR = os.path.join(_HERE, "csrc")
SOURCES = ["mi_rast.hip", "common.h", "cull.h", "geometry.h", "binning.h", "knn_smooth.h", "knn.h", "blend_fwd.h", "blend_fwd_split.h", "blend_fwd_wave.h", "blend_fwd_x3.h", "blend_bwd.h", "blend_bwd_shared.h", "blend_bwd_wave.h", "blend_bwd_feat.h", "contrastive.h"]
HEADER = os.path.join(os.path.dirname(_HERE), "include", "mi_rast.h")
HEADERS = [os.path.join(os.path.dirname(_HERE), "include", h) for h in ("mi_rast.h", "mi_knn.h", "mi_knn_smooth.h", "mi_contrastive.h")]

# -ffp-contract=off is part of the numeric contract: the geometry path that feeds the
# integer tile/sort results must round every binary32 op separately, like the oracle.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics",
               "-fPIC", "-shared", "-Wno-unused-result", "-fno-slp-vectorize"]


def source_hash(extra_flags=()) -> str:
    """Stamp of what a library is built from: the kernel sources, the C-ABI headers and the compiler flags.  Compiled into the
    library (mi_rast_version()) so that measurements taken with one build -- profiles/traffic_*.json, alu_*.json -- are never
    reported next to timings of another (bench.py prints null instead)."""
    import hashlib
    h = hashlib.sha256()
    for path in sorted([os.path.join(SRC_DIR, f) for f in os.listdir(SRC_DIR) if f.endswith((".h", ".hip"))] + HEADERS):
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    h.update(" ".join(list(HIPCC_FLAGS) + list(extra_flags)).encode())
    return h.hexdigest()[:12]


def _hash_flag(extra_flags=()):
    return ['-DMI_RAST_SRC_HASH="' + source_hash(extra_flags) + '"']


def find_hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def is_stale(lib_path: str = None) -> bool:
    lib_path = lib_path or LIB_PATH
    if not os.path.exists(lib_path):
        return True
    t = os.path.getmtime(lib_path)
    deps = [os.path.join(SRC_DIR, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB_PATH
    cmd = [find_hipcc()] + HIPCC_FLAGS + _hash_flag() + ["-o", LIB_PATH + ".tmp", os.path.join(SRC_DIR, "mi_rast.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


def build_profiling_library(verbose: bool = False) -> str:
    """The same sources with -DMI_RAST_PROFILING: run-time ablation masks (MI_RAST_ABLATE / MI_RAST_ABLATE_FWD), in-kernel
    cycle counters, the VALU comparison kernels.  For tools/ only; the product library carries none of it.  Select it with
    MI_RAST_LIB=<path> (seganygaussians_amd/_lib.py)."""
    cmd = [find_hipcc()] + HIPCC_FLAGS + ["-DMI_RAST_PROFILING"] + _hash_flag(["-DMI_RAST_PROFILING"]) + ["-o", PROF_LIB_PATH + ".tmp",
                                          os.path.join(SRC_DIR, "mi_rast.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(PROF_LIB_PATH + ".tmp", PROF_LIB_PATH)
    return PROF_LIB_PATH


if __name__ == "__main__":
    import sys
    if "--profiling" in sys.argv:
        print(build_profiling_library(verbose=True))
    else:
        print(build_library(force=True, verbose=True))
