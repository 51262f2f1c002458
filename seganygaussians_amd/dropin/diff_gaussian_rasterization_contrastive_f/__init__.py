"""Drop-in for submodules/diff-gaussian-rasterization_contrastive_f.  The channel count is a compile-time macro in the
reference (NUM_CHANNELS = 32, cuda_rasterizer/config_contrastive_f.h:15: 64-D features mean editing it and rebuilding); here it is
a run-time argument of the C-ABI, so by default the rasterizer takes it from the call -- the width of `colors_precomp` (any
multiple of 16 up to 256; 32 and 64 run in one pass of the blend kernels).  SAGA_FEATURE_CHANNELS=<n> before import pins it, like
the reference's build does."""
import os

from seganygaussians_amd.rasterizer import GaussianRasterizationSettings, cpu_deep_copy_tuple, make_auto_rasterizer, make_rasterizer

if os.environ.get("SAGA_FEATURE_CHANNELS"):
    NUM_CHANNELS = int(os.environ["SAGA_FEATURE_CHANNELS"])
    _RasterizeGaussians, rasterize_gaussians, GaussianRasterizer = make_rasterizer(NUM_CHANNELS)
else:
    NUM_CHANNELS = 32          # the reference's default; the call decides
    rasterize_gaussians, GaussianRasterizer = make_auto_rasterizer(NUM_CHANNELS)
    _RasterizeGaussians = make_rasterizer(NUM_CHANNELS)[0]   # (the autograd Function of the default width, for introspection)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "cpu_deep_copy_tuple"]
