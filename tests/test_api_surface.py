"""The drop-in packages expose the reference's Python API: names, NamedTuple field order, argument
validation messages (CF/diff_gaussian_rasterization_contrastive_f/__init__.py:156-219,
DEPTH/diff_gaussian_rasterization_depth/__init__.py:294-391).  CPU-only: nothing is rasterized."""
import inspect

import pytest
import torch

import seganygaussians_amd

seganygaussians_amd.install_dropin()

FIELDS = ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
          "sh_degree", "campos", "prefiltered", "debug")


@pytest.mark.parametrize("pkg", ["diff_gaussian_rasterization", "diff_gaussian_rasterization_contrastive_f",
                                 "diff_gaussian_rasterization_depth"])
def test_package_surface(pkg):
    mod = __import__(pkg)
    S, Rz = mod.GaussianRasterizationSettings, mod.GaussianRasterizer
    assert S._fields == FIELDS
    s = S(image_height=4, image_width=5, tanfovx=1.0, tanfovy=1.0, bg=torch.zeros(3), scale_modifier=1.0,
          viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=0, campos=torch.zeros(3), prefiltered=False,
          debug=False)
    r = Rz(raster_settings=s)
    assert isinstance(r, torch.nn.Module) and r.raster_settings is s
    params = list(inspect.signature(r.forward).parameters)
    if pkg.endswith("_depth"):
        assert params == ["means3D", "means2D", "opacities", "mask", "shs", "colors_precomp", "scales", "rotations",
                          "cov3D_precomp"]
        assert list(inspect.signature(r.forward_mask).parameters) == ["means3D", "means2D", "opacities", "mask",
                                                                       "scales", "rotations", "cov3D_precomp"]
    else:
        assert params == ["means3D", "means2D", "opacities", "shs", "colors_precomp", "scales", "rotations",
                          "cov3D_precomp"]
    assert hasattr(r, "markVisible") and hasattr(mod, "rasterize_gaussians") and hasattr(mod, "_RasterizeGaussians")
    m = torch.zeros(2, 3)
    kw = dict(mask=torch.ones(2)) if pkg.endswith("_depth") else {}
    with pytest.raises(Exception, match="Please provide excatly one of either SHs or precomputed colors!"):
        r(means3D=m, means2D=m, opacities=m[:, :1], scales=m, rotations=torch.zeros(2, 4), **kw)
    with pytest.raises(Exception, match="Please provide excatly one of either SHs or precomputed colors!"):
        r(means3D=m, means2D=m, opacities=m[:, :1], shs=torch.zeros(2, 1, 3), colors_precomp=m, scales=m,
          rotations=torch.zeros(2, 4), **kw)
    with pytest.raises(Exception, match="Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!"):
        r(means3D=m, means2D=m, opacities=m[:, :1], colors_precomp=m, scales=m, **kw)
    with pytest.raises(Exception, match="Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!"):
        r(means3D=m, means2D=m, opacities=m[:, :1], colors_precomp=m, scales=m, rotations=torch.zeros(2, 4),
          cov3D_precomp=torch.zeros(2, 6), **kw)


def test_channel_counts():
    import diff_gaussian_rasterization as b
    import diff_gaussian_rasterization_contrastive_f as cf
    import diff_gaussian_rasterization_depth as d
    assert (b.NUM_CHANNELS, cf.NUM_CHANNELS, d.NUM_CHANNELS) == (3, 32, 3)


def test_knn_dropin_names_and_no_cpu_fallback():
    """`simple_knn._C.distCUDA2` and `pytorch3d.ops.knn_points` resolve to the HIP search (values: tests/test_knn.py, GPU);
    like the rasterizer they have no CPU path."""
    from simple_knn._C import distCUDA2
    import pytorch3d.ops
    from seganygaussians_amd import knn
    assert distCUDA2 is knn.distCUDA2 and pytorch3d.ops.knn_points is knn.knn_points
    pts = torch.tensor([[0.0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3], [5, 5, 5]])
    with pytest.raises(RuntimeError, match="GPU tensor"):
        distCUDA2(pts)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        pytorch3d.ops.knn_points(pts[None], pts[None], K=3)
