"""MI355X-native differentiable Gaussian-splatting (feature) rasterizer -- the one hot path of
Jumpat/SegAnyGAussians, behind the reference's Python extension API.

    from seganygaussians_amd import install_dropin
    install_dropin()          # makes diff_gaussian_rasterization{,_contrastive_f,_depth} importable
    from diff_gaussian_rasterization_contrastive_f import GaussianRasterizationSettings, GaussianRasterizer
"""
import os
import sys

__version__ = "0.1.0"

DROPIN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dropin")


def install_dropin() -> str:
    """Puts the drop-in packages (same import names as the reference's pip-installed submodules,
    environment.yml:18-21) at the FRONT of sys.path."""
    if DROPIN_DIR not in sys.path:
        sys.path.insert(0, DROPIN_DIR)
    return DROPIN_DIR
