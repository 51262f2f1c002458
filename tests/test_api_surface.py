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


def test_prezero_follows_the_callers_grad_mode(monkeypatch):
    """The forward asks for the backward's accumulators to be zero-filled (prezero) only when a backward can follow: an input
    requires grad AND the caller's grad mode is on.  `ctx.needs_input_grad` alone says requires_grad whatever the mode, so
    render.py's `torch.no_grad()` forwards of a model whose parameters require grad would fill 128 MB per view for nothing."""
    import diff_gaussian_rasterization_contrastive_f as cf
    from seganygaussians_amd import rasterizer as R

    seen = []

    class Stop(Exception):
        pass

    def fake_native(*a, prezero=False, **k):
        seen.append(bool(prezero))
        raise Stop()

    monkeypatch.setattr(R, "rasterize_gaussians_native", fake_native)
    s = cf.GaussianRasterizationSettings(image_height=4, image_width=5, tanfovx=1.0, tanfovy=1.0, bg=torch.zeros(32),
                                         scale_modifier=1.0, viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=0,
                                         campos=torch.zeros(3), prefiltered=False, debug=False)
    r = cf.GaussianRasterizer(raster_settings=s)
    m = torch.zeros(2, 3)
    feats = torch.zeros(2, 32, requires_grad=True)
    kw = dict(means3D=m, means2D=m, opacities=m[:, :1], colors_precomp=feats, scales=m, rotations=torch.zeros(2, 4))
    with pytest.raises(Stop):
        r(**kw)
    with torch.no_grad(), pytest.raises(Stop):
        r(**kw)
    with pytest.raises(Stop):
        r(**dict(kw, colors_precomp=feats.detach()))
    assert seen == [True, False, False]


def test_channel_width_mismatches_are_refused():
    """The width is a run-time argument here (the reference compiles NUM_CHANNELS in and indexes bg / colors unchecked): a
    background with fewer entries than channels, or colours of another width than the call says, raise before anything runs."""
    from seganygaussians_amd import rasterizer as R
    m = torch.zeros(2, 3)
    args = lambda ch, bg, col: (ch, False, bg, m, col, m[:, :1], None, m, torch.zeros(2, 4), 1.0, torch.empty(0), torch.eye(4),
                                torch.eye(4), 1.0, 1.0, 4, 5, torch.empty(0), 0, torch.zeros(3), False, False)
    with pytest.raises(RuntimeError, match="bg must hold one value per channel"):
        R.rasterize_gaussians_native(*args(64, torch.zeros(32), torch.zeros(2, 64)))
    with pytest.raises(RuntimeError, match=r"colors_precomp must have dimensions \(num_points, 32\)"):
        R.rasterize_gaussians_native(*args(32, torch.zeros(32), torch.zeros(2, 64)))
