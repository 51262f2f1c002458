"""Drop-in for submodules/diff-gaussian-rasterization-depth (RGB + per-Gaussian mask + depth, plus the
mask-only render pair): same public names as diff_gaussian_rasterization_depth/__init__.py."""
from seganygaussians_amd.rasterizer import GaussianRasterizationSettings, cpu_deep_copy_tuple, make_depth_rasterizer

NUM_CHANNELS = 3
(_RasterizeGaussians, _RasterizeMaskGaussians, rasterize_gaussians, rasterize_mask_gaussians,
 GaussianRasterizer) = make_depth_rasterizer()

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "rasterize_mask_gaussians",
           "cpu_deep_copy_tuple"]
