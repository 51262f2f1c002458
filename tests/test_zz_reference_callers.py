"""The reference's own Python callers, UNCHANGED, on top of the drop-in packages (north_star: "gaussian_renderer/__init__.render,
GaussianRasterizer, GaussianRasterizationSettings so train_contrastive_feature.py and render.py call it unchanged").

oracle/build_ref.py byte-compiles the reference modules where they lie (`oracle/_ref/pyref/*.pyc`: gaussian_renderer,
scene.gaussian_model, scene.gaussian_model_ff, scene.cameras, utils.*); this test loads them under their own names next to
`install_dropin()` -- diff_gaussian_rasterization{,_depth,_contrastive_f}, simple_knn._C and pytorch3d.ops are OUR packages,
`plyfile` (absent here, only needed for load/save) is an empty stub -- builds the reference's GaussianModel /
FeatureGaussianModel / Camera objects from a synthetic scene, calls

    render, render_mask                     gaussian_renderer/__init__.py:18,108   (diff_gaussian_rasterization)
    render_with_depth                       :194                                   (diff_gaussian_rasterization_depth)
    render_contrastive_feature              :300  (smooth_type None / 'traditional', norm_point_features)

and compares every output (and the gradients that reach the model's parameters) with the CPU oracle fed the same
activated tensors.  File name sorts last: a missing oracle/_ref is a failure here, which must not hide other tests.
"""
import importlib.machinery
import importlib.util
import math
import os
import sys
import types

import numpy as np
import pytest
import torch

import seganygaussians_amd
from oracle import build_ref
from oracle import saga_oracle as so
from seganygaussians_amd import scenes
from tests import helpers as hp

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _load_pyc(name):
    path = build_ref.pyref_path(name)
    assert os.path.exists(path), f"{path} missing: run python oracle/build_ref.py in the build container"
    loader = importlib.machinery.SourcelessFileLoader(name, path)
    spec = importlib.util.spec_from_loader(name, loader)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def ref():
    """The reference's modules, imported under their own names on top of our drop-in packages."""
    seganygaussians_amd.install_dropin()
    saved = {k: sys.modules.get(k) for k in ("plyfile", "utils", "scene", "gaussian_renderer")}
    ply = types.ModuleType("plyfile")           # only load_ply / save_ply use it
    ply.PlyData = ply.PlyElement = type("Unavailable", (), {})
    sys.modules["plyfile"] = ply
    for pkg in ("utils", "scene"):               # namespace stubs: the reference's scene/__init__.py pulls in the dataset readers
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    mods = {}
    for name in ("utils.system_utils", "utils.general_utils", "utils.graphics_utils", "utils.sh_utils", "scene.cameras",
                 "scene.gaussian_model", "scene.gaussian_model_ff", "gaussian_renderer"):
        mods[name] = _load_pyc(name)
        parent, _, child = name.rpartition(".")
        if parent:
            setattr(sys.modules[parent], child, mods[name])
    import diff_gaussian_rasterization, diff_gaussian_rasterization_contrastive_f, diff_gaussian_rasterization_depth
    gr = mods["gaussian_renderer"]
    # the reference module really is bound to OUR packages
    assert gr.GaussianRasterizer is diff_gaussian_rasterization.GaussianRasterizer
    assert gr.GaussianRasterizerDepth is diff_gaussian_rasterization_depth.GaussianRasterizer
    assert gr.GaussianRasterizerContrastiveF is diff_gaussian_rasterization_contrastive_f.GaussianRasterizer
    yield types.SimpleNamespace(gr=gr, GaussianModel=mods["scene.gaussian_model"].GaussianModel,
                                FeatureGaussianModel=mods["scene.gaussian_model_ff"].FeatureGaussianModel,
                                Camera=mods["scene.cameras"].Camera, general_utils=mods["utils.general_utils"])
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


def _scene(P, W, H, C, seed):
    focal = 0.9 * W
    sc = scenes.make_scene(P, W, H, focal, C, math.log(0.05), 0.6, seed=seed, with_shs=(C == 3))
    return sc, focal


def _camera(ref, W, H, focal, orbit=True):
    cam = scenes.orbit_camera(W, H, focal, 0.15, 0.05) if orbit else scenes.look_at_camera(W, H, focal)
    fovx, fovy = 2 * math.atan(W / (2 * focal)), 2 * math.atan(H / (2 * focal))
    c = ref.Camera(colmap_id=0, R=cam.R, T=cam.T, FoVx=fovx, FoVy=fovy, image=torch.zeros(3, H, W), gt_alpha_mask=None,
                   image_name="synthetic", uid=0)
    return c


def _param(a):
    return torch.nn.Parameter(torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(DEV))


def _fill_geometry(pc, sc, ref):
    inv_sig = ref.general_utils.inverse_sigmoid
    pc._xyz = _param(sc.means3D)
    pc._scaling = _param(np.log(sc.scales))
    pc._rotation = _param(sc.rotations)
    pc._opacity = _param(inv_sig(torch.as_tensor(sc.opacities)).numpy())


def _oracle_inputs(cam, pc, C, W, H, bg, *, shs=None, sh_degree=0, colors=None, mask=None):
    n = lambda t: t.detach().cpu().numpy()
    return so.Inputs(means3D=n(pc.get_xyz), opacities=n(pc.get_opacity), viewmatrix=n(cam.world_view_transform),
                     projmatrix=n(cam.full_proj_transform), campos=n(cam.camera_center), bg=n(bg), image_width=W, image_height=H,
                     tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), channels=C, sh_degree=sh_degree,
                     shs=None if shs is None else n(shs), colors_precomp=None if colors is None else n(colors),
                     scales=n(pc.get_scaling), rotations=n(pc.get_rotation), mask=None if mask is None else n(mask))


PIPE = types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)


def test_render_and_render_mask_rgb(ref):
    P, W, H = 6000, 208, 144
    sc, focal = _scene(P, W, H, 3, seed=61)
    pc = ref.GaussianModel(3)
    _fill_geometry(pc, sc, ref)
    pc._features_dc = _param(sc.shs[:, :1])
    pc._features_rest = _param(sc.shs[:, 1:])
    pc.active_sh_degree = 3
    pc._mask = torch.rand(P, device=DEV)
    cam = _camera(ref, W, H, focal)
    bg = torch.tensor([0.1, 0.5, 0.9], device=DEV)
    out = ref.gr.render(cam, pc, PIPE, bg)
    inp = _oracle_inputs(cam, pc, 3, W, H, bg, shs=pc.get_features, sh_degree=3)
    fwd = so.forward(inp)
    hp.assert_close("render", out["render"].detach().cpu().numpy(), fwd.color, flip_frac=hp.FLIP_FRAC)
    np.testing.assert_array_equal(out["radii"].cpu().numpy(), fwd.radii)
    assert torch.equal(out["visibility_filter"], out["radii"] > 0)
    # training step of train_scene.py: loss.backward() reaches the model's raw parameters through the activations
    dL = scenes.make_grad_image(3, H, W, seed=2)
    (out["render"] * torch.as_tensor(dL).to(DEV)).sum().backward()
    bwd = so.backward(inp, fwd, dL)
    hp.assert_close("viewspace_points.grad", out["viewspace_points"].grad.cpu().numpy(), bwd.dL_dmeans2D, flip_frac=hp.GRAD_FLIP_FRAC)
    hp.assert_close("_xyz.grad", pc._xyz.grad.cpu().numpy(), bwd.dL_dmeans3D, flip_frac=hp.GRAD_FLIP_FRAC)
    hp.assert_close("_features_dc.grad", pc._features_dc.grad.cpu().numpy(), bwd.dL_dsh[:, :1], flip_frac=hp.GRAD_FLIP_FRAC)
    hp.assert_close("_features_rest.grad", pc._features_rest.grad.cpu().numpy(), bwd.dL_dsh[:, 1:], flip_frac=hp.GRAD_FLIP_FRAC)
    want_dscaling = bwd.dL_dscales * sc.scales            # d exp(s) = exp(s)
    hp.assert_close("_scaling.grad", pc._scaling.grad.cpu().numpy(), want_dscaling, flip_frac=hp.GRAD_FLIP_FRAC)
    # override_color + convert_SHs_python (render.py's paths)
    pipe2 = types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=True)
    out2 = ref.gr.render(cam, pc, pipe2, bg)
    hp.assert_close("render (SH in python)", out2["render"].detach().cpu().numpy(), fwd.color, rtol=2e-4, flip_frac=1e-4)
    # render_mask: the per-Gaussian mask broadcast to three channels through the plain RGB rasterizer
    m = ref.gr.render_mask(cam, pc, PIPE, torch.zeros(3, device=DEV))
    inp_m = _oracle_inputs(cam, pc, 3, W, H, torch.zeros(3), colors=pc.get_mask[:, None].repeat(1, 3))
    hp.assert_close("render_mask", m["mask"].detach().cpu().numpy(), so.forward(inp_m).color, flip_frac=hp.FLIP_FRAC)


def test_render_with_depth(ref):
    P, W, H = 5000, 176, 128
    sc, focal = _scene(P, W, H, 3, seed=62)
    pc = ref.GaussianModel(3)
    _fill_geometry(pc, sc, ref)
    pc._features_dc = _param(sc.shs[:, :1])
    pc._features_rest = _param(sc.shs[:, 1:])
    pc.active_sh_degree = 2
    pc._mask = torch.rand(P, device=DEV)
    cam = _camera(ref, W, H, focal, orbit=False)
    bg = torch.tensor([0.3, 0.2, 0.1], device=DEV)
    out = ref.gr.render_with_depth(cam, pc, PIPE, bg)
    inp = _oracle_inputs(cam, pc, 3, W, H, bg, shs=pc.get_features, sh_degree=2, mask=pc.get_mask)
    fwd = so.forward(inp)
    hp.assert_close("render", out["render"].detach().cpu().numpy(), fwd.color, flip_frac=hp.FLIP_FRAC)
    hp.assert_close("mask", out["mask"].detach().cpu().numpy(), fwd.mask, flip_frac=hp.FLIP_FRAC)
    hp.assert_close("depth", out["depth"].detach().cpu().numpy(), fwd.depth, flip_frac=hp.FLIP_FRAC)
    np.testing.assert_array_equal(out["radii"].cpu().numpy(), fwd.radii)


def test_render_contrastive_feature_and_smoothing(ref):
    P, W, H = 8000, 256, 160
    sc, focal = _scene(P, W, H, 32, seed=63)
    pc = ref.FeatureGaussianModel(32)
    _fill_geometry(pc, sc, ref)
    pc._point_features = _param(sc.features * np.linspace(0.5, 2.0, P, dtype=np.float32)[:, None])
    cam = _camera(ref, W, H, focal)
    fh, fw = cam.feature_height, cam.feature_width     # scene/cameras.py: the feature image has its own size
    bg = torch.zeros(32, device=DEV)                  # train_contrastive_feature.py:98
    # 1. plain features, normalised in the renderer (train_contrastive_feature.py:231)
    out = ref.gr.render_contrastive_feature(cam, pc, PIPE, bg, norm_point_features=True)
    f = pc.get_point_features
    fn = (f / (f.norm(dim=1, keepdim=True) + 1e-9)).detach()
    inp = _oracle_inputs(cam, pc, 32, fw, fh, bg, colors=fn)
    fwd = so.forward(inp)
    assert out["render"].shape == (32, fh, fw)
    hp.assert_close("feature render", out["render"].detach().cpu().numpy(), fwd.color, flip_frac=hp.FLIP_FRAC)
    dL = scenes.make_grad_image(32, fh, fw, seed=3)
    (out["render"] * torch.as_tensor(dL).to(DEV)).sum().backward()
    bwd = so.backward(inp, fwd, dL)
    # chain rule of the renderer's normalisation, in torch, from the oracle's dL/dcolors
    f2 = f.detach().clone().requires_grad_(True)
    (f2 / (f2.norm(dim=1, keepdim=True) + 1e-9)).backward(torch.as_tensor(bwd.dL_dcolors).to(DEV))
    hp.assert_close("_point_features.grad", pc._point_features.grad.cpu().numpy(), f2.grad.cpu().numpy(), rtol=2e-4,
                    flip_frac=hp.GRAD_FLIP_FRAC)
    # 2. 'traditional' smoothing: pytorch3d.ops.knn_points (ours) builds the neighbour map inside the reference's model
    torch.manual_seed(5)
    out_s = ref.gr.render_contrastive_feature(cam, pc, PIPE, bg, norm_point_features=True, smooth_type="traditional", smooth_K=16)
    nmap = pc.feature_smooth_map["m"]
    assert nmap.shape == (P, 16) and torch.equal(nmap[:, 0], torch.arange(P, device=DEV))
    torch.manual_seed(5)
    sm = pc.get_smoothed_point_features(K=16, dropout=0.5).detach()   # same CPU-generator draw as inside the call above
    sm = sm / (sm.norm(dim=1, keepdim=True) + 1e-9)
    fwd_s = so.forward(_oracle_inputs(cam, pc, 32, fw, fh, bg, colors=sm))
    hp.assert_close("smoothed feature render", out_s["render"].detach().cpu().numpy(), fwd_s.color, flip_frac=hp.FLIP_FRAC)
    # the neighbour map is exact
    d = torch.cdist(pc.get_xyz.double(), pc.get_xyz.double())
    want = d.topk(16, dim=1, largest=False).indices
    assert (nmap.sort(1).values == want.sort(1).values).all(1).float().mean() > 0.999


def test_fused_smoothing_opt_in_on_the_reference_model(ref):
    """install_dropin(fuse_smoothing=True) rebinds the reference's FeatureGaussianModel.get_smoothed_point_features to the fused
    HIP kernels.  Same CPU-generator column draw, so: same returned features, same `_point_features.grad`, same render through
    render_contrastive_feature(smooth_type='traditional') as the reference's own method (kept as _reference_get_...)."""
    import sys
    P, W, H = 8000, 256, 160
    sc, focal = _scene(P, W, H, 32, seed=64)
    pc = ref.FeatureGaussianModel(32)
    _fill_geometry(pc, sc, ref)
    pc._point_features = _param(sc.features * np.linspace(0.5, 2.0, P, dtype=np.float32)[:, None])
    cam = _camera(ref, W, H, focal)
    bg = torch.zeros(32, device=DEV)
    cls = ref.FeatureGaussianModel
    assert sys.modules["scene.gaussian_model_ff"].FeatureGaussianModel is cls
    seganygaussians_amd.install_dropin(fuse_smoothing=True)
    try:
        assert cls._mi_fused_smoothing and cls.get_smoothed_point_features is not cls._reference_get_smoothed_point_features
        up = torch.randn(P, 32, generator=torch.Generator().manual_seed(1)).to(DEV)
        torch.manual_seed(11)
        want = cls._reference_get_smoothed_point_features(pc, K=16, dropout=0.5)
        want.backward(up)
        gw = pc._point_features.grad.clone()
        pc._point_features.grad = None
        torch.manual_seed(11)
        got = pc.get_smoothed_point_features(K=16, dropout=0.5)
        got.backward(up)
        gg = pc._point_features.grad.clone()
        pc._point_features.grad = None
        torch.testing.assert_close(got, want, rtol=0, atol=2e-6 * float(want.abs().max()))
        torch.testing.assert_close(gg, gw, rtol=0, atol=2e-5 * float(gw.abs().max()))
        assert pc._mi_neighbour_map[0] is pc.feature_smooth_map["m"]
        # all K columns (dropout outside (0, 1)), and K <= 1 returns the raw features
        torch.testing.assert_close(pc.get_smoothed_point_features(K=16, dropout=-1),
                                   cls._reference_get_smoothed_point_features(pc, K=16, dropout=-1), rtol=0, atol=2e-6)
        assert pc.get_smoothed_point_features(K=1) is pc._point_features
        # through the reference's renderer
        torch.manual_seed(12)
        out_f = ref.gr.render_contrastive_feature(cam, pc, PIPE, bg, norm_point_features=True, smooth_type="traditional", smooth_K=16)
        cls.get_smoothed_point_features, fused = cls._reference_get_smoothed_point_features, cls.get_smoothed_point_features
        try:
            torch.manual_seed(12)
            out_r = ref.gr.render_contrastive_feature(cam, pc, PIPE, bg, norm_point_features=True, smooth_type="traditional", smooth_K=16)
        finally:
            cls.get_smoothed_point_features = fused
        hp.assert_close("render with fused smoothing", out_f["render"].detach().cpu().numpy(), out_r["render"].detach().cpu().numpy(),
                        flip_frac=hp.FLIP_FRAC)
    finally:
        cls.get_smoothed_point_features = cls._reference_get_smoothed_point_features
        cls._mi_fused_smoothing = False
