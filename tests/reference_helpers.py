"""The reference's OWN Python helpers, imported from /root/reference (build container only; the GPU box has no copy):

    utils/sh_utils.py:57-112        eval_sh
    utils/general_utils.py:78-110   build_rotation, build_scaling_rotation
    utils/graphics_utils.py:22-98   geom_transform_points, getWorld2View2, getProjectionMatrix

They pin tests/dense_ref.py (which restates them so that it can run anywhere) and generate the golden fixtures
(tests/golden/make_golden.py).  `general_utils` allocates on device "cuda" in fp32; the wrappers below run the same code
on the CPU in fp64 by redirecting torch.zeros for the duration of the call -- the arithmetic is the reference's.
"""
from __future__ import annotations

import contextlib
import importlib.util
import os

import torch

REF_ROOT = os.environ.get("SAGA_REFERENCE_ROOT", "/root/reference")


def present() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "utils", "sh_utils.py"))


def _load(name):
    path = os.path.join(REF_ROOT, "utils", name + ".py")
    spec = importlib.util.spec_from_file_location("saga_reference_utils_" + name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@contextlib.contextmanager
def _cpu_fp64_zeros():
    orig = torch.zeros

    def zeros(*a, **k):
        k.pop("device", None)
        k["dtype"] = torch.float64
        return orig(*a, **k)

    torch.zeros = zeros
    try:
        yield
    finally:
        torch.zeros = orig


class ReferenceHelpers:
    def __init__(self):
        self.sh_utils = _load("sh_utils")
        self.general_utils = _load("general_utils")
        self.graphics_utils = _load("graphics_utils")

    def eval_sh(self, deg, sh_pk3, dirs):
        """sh_pk3: (P, K, 3) as the rasterizer takes it; the reference's eval_sh wants (P, 3, K)
        (gaussian_renderer/__init__.py:69: shs_view = features.transpose(1, 2))."""
        return self.sh_utils.eval_sh(deg, sh_pk3.transpose(1, 2), dirs)

    def build_rotation(self, q):
        with _cpu_fp64_zeros():
            return self.general_utils.build_rotation(q)

    def build_scaling_rotation(self, s, q):
        with _cpu_fp64_zeros():
            return self.general_utils.build_scaling_rotation(s, q)

    def geom_transform_points(self, pts, m):
        return self.graphics_utils.geom_transform_points(pts, m)
