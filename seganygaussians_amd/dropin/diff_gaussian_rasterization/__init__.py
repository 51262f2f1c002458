"""Drop-in for submodules/diff-gaussian-rasterization (NUM_CHANNELS = 3, config.h:15): same public
names as diff_gaussian_rasterization/__init__.py of the reference, served by the MI355X C-ABI library."""
from seganygaussians_amd.rasterizer import GaussianRasterizationSettings, cpu_deep_copy_tuple, make_rasterizer

NUM_CHANNELS = 3
_RasterizeGaussians, rasterize_gaussians, GaussianRasterizer = make_rasterizer(NUM_CHANNELS)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "cpu_deep_copy_tuple"]
