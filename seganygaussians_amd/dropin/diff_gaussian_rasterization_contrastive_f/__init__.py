"""Drop-in for submodules/diff-gaussian-rasterization_contrastive_f (NUM_CHANNELS = 32,
config_contrastive_f.h:15).  The channel count is a compile-time macro in the reference; set
SAGA_FEATURE_CHANNELS=64 before import for the 64-D build (BASELINE config 5)."""
import os

from seganygaussians_amd.rasterizer import GaussianRasterizationSettings, cpu_deep_copy_tuple, make_rasterizer

NUM_CHANNELS = int(os.environ.get("SAGA_FEATURE_CHANNELS", "32"))
_RasterizeGaussians, rasterize_gaussians, GaussianRasterizer = make_rasterizer(NUM_CHANNELS)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "cpu_deep_copy_tuple"]
