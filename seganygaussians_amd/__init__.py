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


def install_dropin(fuse_smoothing: bool = False) -> str:
    """Puts the drop-in packages (same import names as the reference's pip-installed submodules,
    environment.yml:18-21) at the FRONT of sys.path.

    fuse_smoothing=True (opt-in) additionally rebinds `FeatureGaussianModel.get_smoothed_point_features`
    (scene/gaussian_model_ff.py:338-364) to the fused HIP gather kernels (knn_smooth.py, include/mi_knn_smooth.h): at once
    if `scene.gaussian_model_ff` is already imported, otherwise right after it is imported.  Same signature, same column
    draw from the CPU generator, same values and gradients; the reference's own PyTorch expression is no longer executed."""
    if DROPIN_DIR not in sys.path:
        sys.path.insert(0, DROPIN_DIR)
    if fuse_smoothing:
        mod = sys.modules.get(_FF_MODULE)
        if mod is not None and hasattr(mod, "FeatureGaussianModel"):
            patch_feature_model(mod.FeatureGaussianModel)
        elif not any(isinstance(f, _PatchOnImport) for f in sys.meta_path):
            sys.meta_path.insert(0, _PatchOnImport())
    return DROPIN_DIR


_FF_MODULE = "scene.gaussian_model_ff"


def patch_feature_model(cls) -> None:
    """Rebinds cls.get_smoothed_point_features (the reference's FeatureGaussianModel) to the fused HIP path; idempotent.
    The original stays reachable as cls._reference_get_smoothed_point_features."""
    if getattr(cls, "_mi_fused_smoothing", False):
        return
    from .knn_smooth import fused_get_smoothed_point_features
    cls._reference_get_smoothed_point_features = cls.get_smoothed_point_features
    cls.get_smoothed_point_features = fused_get_smoothed_point_features
    cls._mi_fused_smoothing = True


class _PatchOnImport:
    """sys.meta_path finder: lets the normal machinery find scene.gaussian_model_ff, then patches the class once the
    module has been executed."""

    def find_spec(self, name, path=None, target=None):
        if name != _FF_MODULE:
            return None
        import importlib.util
        sys.meta_path.remove(self)
        try:
            spec = importlib.util.find_spec(name)
        finally:
            sys.meta_path.insert(0, self)
        if spec is None or spec.loader is None or not hasattr(spec.loader, "exec_module"):
            return spec
        inner = spec.loader.exec_module

        def exec_module(module, _inner=inner):
            _inner(module)
            if hasattr(module, "FeatureGaussianModel"):
                patch_feature_model(module.FeatureGaussianModel)
            if self in sys.meta_path:
                sys.meta_path.remove(self)

        spec.loader.exec_module = exec_module
        return spec
