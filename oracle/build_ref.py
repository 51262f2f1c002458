"""TEST INFRASTRUCTURE ONLY: builds oracle/_ref/ -- the REFERENCE's own rasterizer, translated for the test box.

The reference (Jumpat/SegAnyGAussians) is CUDA.  Its core (`cuda_rasterizer/{forward,backward,rasterizer_impl}.cu`,
SURVEY.md section 8c "optional stronger oracle") contains no NVIDIA-only construct beyond spelling: this recipe runs
ROCm's `hipify-perl` over the sources WHERE THEY LIE under /root/reference into a temporary directory (the same
translation `torch.utils.cpp_extension.CUDAExtension` applies under ROCm), patches four spelling issues hipify-perl does
not handle, and compiles them with hipcc for gfx950 together with oracle/ref_shim.cpp (our C-ABI wrapper).  Only the shared
objects stay, in oracle/_ref/ (git-ignored, shipped to the GPU box by gpurun like our own .so).  No reference source is
copied into the repository; the temporary directory is removed.

This is NOT the product and never linked into it: the product is seganygaussians_amd/ (hand-written gfx950 kernels).
_ref exists so that (1) the CPU oracle is PINNED against the reference implementation itself and (2) the product path is
compared with the reference directly, at sizes the CPU oracle would take too long for.

Variants (one .so each; the reference fixes NUM_CHANNELS at compile time, CF/cuda_rasterizer/config_contrastive_f.h):
    cf32    CF/    NUM_CHANNELS 32   diff_gaussian_rasterization_contrastive_f  (the headline path)
    cf64    CF/    NUM_CHANNELS 64   what a user builds for 64-D features (cfg5)
    cf16 / cf128  CF/  NUM_CHANNELS 16 / 128   (parity of the product's channel blocks: any multiple of 16)
    cf8 / cf40 / cf100  CF/  NUM_CHANNELS 8 / 40 / 100   (widths that end in a partial 16-channel block)
    base3   BASE/  NUM_CHANNELS 3    diff_gaussian_rasterization
    depth3  DEPTH/ NUM_CHANNELS 3    diff_gaussian_rasterization_depth (mask + depth + mask-only pair)
    knn     simple-knn/simple_knn.cu  (distCUDA2's core, SimpleKNN::knn)
Floating-point mode: `-ffp-contract=off` ("strict": every binary32 operation rounded separately, the numeric contract of
DESIGN.md section 2, which makes the integer path comparable bit for bit) and, for cf32, additionally hipcc's default
contraction ("fast": what an out-of-the-box hipified build computes; nvcc contracts too, in its own places) so that the
tests can report how many radii / list entries an FMA decision moves.
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("SAGA_REFERENCE_ROOT", "/root/reference")
OUT_DIR = os.path.join(_HERE, "_ref")
SHIM = os.path.join(_HERE, "ref_shim.cpp")
KNN_SHIM = os.path.join(_HERE, "ref_knn_shim.cpp")
HIPCC = "/opt/rocm/bin/hipcc"
HIPIFY = "/opt/rocm/bin/hipify-perl"

SUB = {"cf": "submodules/diff-gaussian-rasterization_contrastive_f",
       "base": "submodules/diff-gaussian-rasterization",
       "depth": "submodules/diff-gaussian-rasterization-depth"}
# name -> (source tree, NUM_CHANNELS, shim defines, contraction)
VARIANTS = {
    "cf32": ("cf", 32, [], "off"),
    "cf32_fast": ("cf", 32, [], "fast"),
    "cf64": ("cf", 64, [], "off"),
    "cf16": ("cf", 16, [], "off"),      # the narrowest / widest feature the product's channel blocks cover in the parity tests
    "cf128": ("cf", 128, [], "off"),
    "cf8": ("cf", 8, [], "off"),        # widths that are no multiple of 16: the product's last channel block is partial
    "cf40": ("cf", 40, [], "off"),
    "cf100": ("cf", 100, [], "off"),
    "base3": ("base", 3, ["-DREF_BASE"], "off"),
    "depth3": ("depth", 3, ["-DREF_DEPTH"], "off"),
}


def lib_path(variant: str) -> str:
    return os.path.join(OUT_DIR, f"libsaga_ref_{variant}.so")


def reference_present() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, SUB["cf"], "cuda_rasterizer"))


def _hipify(src: str, dst: str) -> None:
    out = subprocess.run([HIPIFY, src], check=True, capture_output=True, text=True).stdout
    # what hipify-perl leaves behind (spelling only):
    out = re.sub(r"^#include <cooperative_groups/reduce.h>\s*$", "", out, flags=re.M)       # unused header, absent in HIP
    out = re.sub(r'^#include ""\s*$', "", out, flags=re.M)                                   # was device_launch_parameters.h
    out = re.sub(r"^#include <cub/device/device_radix_sort.cuh>\s*$", "", out, flags=re.M)   # covered by hipcub.hpp
    out = out.replace("<< <", "<<<").replace(">> >", ">>>")                                   # launch chevrons written with a blank
    out = out.replace("__trap()", "__builtin_trap()")
    with open(dst, "w") as f:
        f.write(out)


def _newest(paths) -> float:
    return max(os.path.getmtime(p) for p in paths)


def build_variant(name: str, force: bool = False, verbose: bool = False) -> str:
    tree, channels, defines, contract = VARIANTS[name]
    src_dir = os.path.join(REF_ROOT, SUB[tree], "cuda_rasterizer")
    glm = os.path.join(REF_ROOT, SUB[tree], "third_party", "glm")
    out = lib_path(name)
    srcs = [os.path.join(src_dir, f) for f in sorted(os.listdir(src_dir))]
    if not force and os.path.exists(out) and os.path.getmtime(out) >= _newest(srcs + [SHIM]):
        return out
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix=f"saga_ref_{name}_")
    try:
        for s in srcs:
            _hipify(s, os.path.join(tmp, os.path.basename(s)))
        # the channel count is a compile-time constant of the reference
        for cfg in ("config.h", "config_contrastive_f.h"):
            p = os.path.join(tmp, cfg)
            if os.path.exists(p):
                txt = open(p).read()
                txt = re.sub(r"#define NUM_CHANNELS \d+", f"#define NUM_CHANNELS {channels}", txt)
                open(p, "w").write(txt)
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
               f"-ffp-contract={contract}", "-Wno-unused-value", "-Wno-unused-result", "-I" + tmp, "-I" + glm, "-x", "hip",
               os.path.join(tmp, "forward.cu"), os.path.join(tmp, "backward.cu"), os.path.join(tmp, "rasterizer_impl.cu"),
               SHIM, "-Wl,-Bsymbolic", "-o", out + ".tmp"] + defines
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL,
                              stderr=None if verbose else subprocess.DEVNULL)
        os.replace(out + ".tmp", out)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def build_knn(force: bool = False, verbose: bool = False) -> str:
    src = os.path.join(REF_ROOT, "submodules/simple-knn/simple_knn.cu")
    hdr = os.path.join(REF_ROOT, "submodules/simple-knn/simple_knn.h")
    out = lib_path("knn")
    if not force and os.path.exists(out) and os.path.getmtime(out) >= _newest([src, hdr, KNN_SHIM]):
        return out
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="saga_ref_knn_")
    try:
        _hipify(src, os.path.join(tmp, "simple_knn.cu"))
        _hipify(hdr, os.path.join(tmp, "simple_knn.h"))
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
               "-ffp-contract=off", "-Wno-unused-value", "-Wno-unused-result", "-include", "float.h",  # FLT_MAX: the
               # reference relies on a transitive include of older CUDA toolkits
               "-I" + tmp, "-x", "hip", os.path.join(tmp, "simple_knn.cu"), KNN_SHIM, "-Wl,-Bsymbolic", "-o", out + ".tmp"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL,
                              stderr=None if verbose else subprocess.DEVNULL)
        os.replace(out + ".tmp", out)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


# The reference's Python CALLERS of the rasterizer -- north_star: "render.py / train_contrastive_feature.py call it unchanged" --
# byte-compiled (py_compile) from where they lie, for tests/test_zz_reference_callers.py: the test box has no /root/reference,
# and reference sources are never copied into this repository; the .pyc files in oracle/_ref/pyref/ are build outputs like the
# shared objects above (git-ignored, shipped by gpurun).
PYREF_MODULES = {
    "gaussian_renderer": "gaussian_renderer/__init__.py",       # render, render_mask, render_with_depth, render_contrastive_feature
    "scene.gaussian_model": "scene/gaussian_model.py",          # GaussianModel (activations, get_* properties)
    "scene.gaussian_model_ff": "scene/gaussian_model_ff.py",    # FeatureGaussianModel (+ KNN feature smoothing)
    "scene.cameras": "scene/cameras.py",                        # Camera (matrix conventions)
    "utils.sh_utils": "utils/sh_utils.py",
    "utils.general_utils": "utils/general_utils.py",
    "utils.graphics_utils": "utils/graphics_utils.py",
    "utils.system_utils": "utils/system_utils.py",
    # the training / rendering scripts north_star names, with the scene loader they go through (tests/test_zz_reference_training.py)
    "utils.camera_utils": "utils/camera_utils.py",
    "scene.colmap_loader": "scene/colmap_loader.py",
    "scene.dataset_readers": "scene/dataset_readers.py",
    "scene": "scene/__init__.py",                              # Scene
    "arguments": "arguments/__init__.py",                      # ModelParams / OptimizationParams / PipelineParams
    "train_contrastive_feature": "train_contrastive_feature.py",
    "render": "render.py",
}
PYREF_DIR = os.path.join(OUT_DIR, "pyref")


def pyref_path(module: str) -> str:
    return os.path.join(PYREF_DIR, module + ".pyc")


def build_pyref(force: bool = False):
    import py_compile
    os.makedirs(PYREF_DIR, exist_ok=True)
    out = []
    for mod, rel in PYREF_MODULES.items():
        src, dst = os.path.join(REF_ROOT, rel), pyref_path(mod)
        if force or not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src):
            py_compile.compile(src, cfile=dst, dfile=f"<reference>/{rel}", doraise=True)
        out.append(dst)
    return out


def build_all(force: bool = False, verbose: bool = False):
    """Builds every variant when /root/reference is present (the build container); on the GPU box the prebuilt
    files in oracle/_ref/ are used as they are."""
    if not reference_present():
        return [p for p in [lib_path(v) for v in list(VARIANTS) + ["knn"]] + [pyref_path(m) for m in PYREF_MODULES]
                if os.path.exists(p)]
    from concurrent.futures import ThreadPoolExecutor          # one hipcc per variant, about a minute each
    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        futs = [ex.submit(build_variant, v, force, verbose) for v in VARIANTS] + [ex.submit(build_knn, force, verbose)]
        return [f.result() for f in futs] + build_pyref(force)


if __name__ == "__main__":
    for p in build_all(force="--force" in sys.argv, verbose="-v" in sys.argv):
        print(p)
