"""Loads the REFERENCE's own Python modules (byte-compiled where they lie by oracle/build_ref.py -> oracle/_ref/pyref/*.pyc)
under their own names, on top of the drop-in packages: diff_gaussian_rasterization{,_depth,_contrastive_f}, simple_knn._C and
pytorch3d.ops are OURS (seganygaussians_amd/dropin); `plyfile` is tests/plyfile_shim.py and `torchvision.utils.save_image` a
PIL writer (both packages are absent from this image).  TEST INFRASTRUCTURE -- this executes code compiled from the untrusted
reference tree, on the GPU box too; nothing under seganygaussians_amd/ imports it."""
import importlib.machinery
import importlib.util
import os
import sys
import types

import seganygaussians_amd
from oracle import build_ref

MODULES = ("utils.system_utils", "utils.general_utils", "utils.graphics_utils", "utils.sh_utils", "scene.cameras",
           "scene.gaussian_model", "scene.gaussian_model_ff", "scene.colmap_loader", "scene.dataset_readers", "utils.camera_utils",
           "arguments", "scene", "gaussian_renderer", "train_contrastive_feature", "render")
_TOUCHED = ("plyfile", "torchvision", "torchvision.utils", "sklearn", "sklearn.preprocessing", "utils", "scene", "arguments", "gaussian_renderer",
            "train_contrastive_feature", "render")


def _exec_pyc(name, module=None):
    path = build_ref.pyref_path(name)
    assert os.path.exists(path), f"{path} missing: run python oracle/build_ref.py in the build container"
    loader = importlib.machinery.SourcelessFileLoader(name, path)
    if module is None:
        module = importlib.util.module_from_spec(importlib.util.spec_from_loader(name, loader))
        sys.modules[name] = module
    exec(loader.get_code(name), module.__dict__)
    return module


def _save_image(tensor, path, **_):
    """torchvision.utils.save_image for a single (C, H, W) image in [0, 1]."""
    import numpy as np
    from PIL import Image
    a = (tensor.detach().clamp(0, 1).cpu().numpy() * 255 + 0.5).astype(np.uint8)
    a = a[0] if a.shape[0] == 1 else np.transpose(a[:3], (1, 2, 0))
    Image.fromarray(a).save(path)


class _QuantileTransformer:
    """sklearn.preprocessing.QuantileTransformer(output_distribution='uniform') as train_contrastive_feature.py:41-62 uses it
    (fit on a column of mask scales, transform to [0, 1]) -- for boxes without scikit-learn (the GPU test image has none)."""

    def __init__(self, output_distribution="uniform", n_quantiles=1000):
        assert output_distribution == "uniform"
        self.n_quantiles = n_quantiles

    def fit(self, X):
        import numpy as np
        X = np.asarray(X, np.float64).reshape(-1)
        n = min(self.n_quantiles, X.size)
        self.references_ = np.linspace(0, 1, n)
        self.quantiles_ = np.nanpercentile(X, self.references_ * 100)
        return self

    def transform(self, X):
        import numpy as np
        X = np.asarray(X, np.float64)
        q, r = self.quantiles_, self.references_
        out = 0.5 * (np.interp(X.reshape(-1), q, r) - np.interp(-X.reshape(-1), -q[::-1], -r[::-1]))
        return out.reshape(X.shape)


class ReferenceEnv:
    """Context manager: `with ReferenceEnv() as ref: ref.mod['train_contrastive_feature'].training(...)`."""

    def __enter__(self):
        seganygaussians_amd.install_dropin()
        self._saved = {k: sys.modules.get(k) for k in _TOUCHED + tuple(m for m in MODULES)}
        from tests import plyfile_shim
        sys.modules["plyfile"] = plyfile_shim
        tv = types.ModuleType("torchvision")
        tv.utils = types.ModuleType("torchvision.utils")
        tv.utils.save_image = _save_image
        sys.modules["torchvision"], sys.modules["torchvision.utils"] = tv, tv.utils
        try:
            import sklearn.preprocessing  # noqa: F401
        except ImportError:
            sk = types.ModuleType("sklearn")
            sk.preprocessing = types.ModuleType("sklearn.preprocessing")
            sk.preprocessing.QuantileTransformer = _QuantileTransformer
            sys.modules["sklearn"], sys.modules["sklearn.preprocessing"] = sk, sk.preprocessing
        for pkg in ("utils", "scene"):   # packages: `utils` has no __init__ in the reference, scene/__init__.py is executed last
            m = types.ModuleType(pkg)
            m.__path__ = []
            sys.modules[pkg] = m
        self.mod = {}
        for name in MODULES:
            if name == "scene":
                self.mod[name] = _exec_pyc(name, sys.modules["scene"])
            else:
                self.mod[name] = _exec_pyc(name)
            parent, _, child = name.rpartition(".")
            if parent:
                setattr(sys.modules[parent], child, self.mod[name])
        return self

    def __exit__(self, *exc):
        for k, v in self._saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        return False
