"""The features-only backward (EXTENSION, include/mi_rast.h: MI_RAST_BWD_FEATURES_ONLY; seganygaussians_amd/rasterizer.py; kernel:
csrc/blend_bwd_feat.h, one wave per half tile, channel blocks of 64 / 32 / 16):
dL_dcolors_precomp alone -- automatic when autograd asks for nothing else, opt-in (enable_features_only_backward /
MI_RAST_FEATURES_ONLY_BACKWARD=1) for callers whose other inputs require grad without anybody reading those gradients
(SAGA's contrastive feature training, scene/gaussian_model_ff.py:154-162).  Checked against the CPU oracle's dL_dcolors
(backward.cu:399-559 restated) and against the default backward of the same call."""
import numpy as np
import pytest
import torch

import seganygaussians_amd
from seganygaussians_amd import _lib, scenes
from seganygaussians_amd import rasterizer as R
from tests import helpers as hp

seganygaussians_amd.install_dropin()


def test_switch_and_applicability_rules():
    """Host logic only (no GPU): who gets the features-only backward."""
    prev = R.enable_features_only_backward(False)
    try:
        class _T:   # stands in for a tensor: numel() is all the rule looks at
            def __init__(self, n):
                self.n = n

            def numel(self):
                return self.n

        L = _lib.load()
        assert L.mi_rast_features_only_supported(32) == 1 and L.mi_rast_features_only_supported(64) == 1
        assert L.mi_rast_features_only_supported(48) == 1 and L.mi_rast_features_only_supported(16) == 1
        assert L.mi_rast_features_only_supported(3) == 0 and L.mi_rast_features_only_supported(40) == 0
        assert L.mi_rast_features_only_supported(0) == 0 and L.mi_rast_features_only_supported(272) == 0
        only_colors = (False, False, False, True, False, False, False, False, False)
        everything = (True, True, False, True, True, True, True, False, False)
        feats = _T(32 * 10)
        # automatic: autograd wants the colour gradient alone
        assert R._features_only_applies(32, feats, only_colors, False)
        assert not R._features_only_applies(32, feats, everything, False)
        # never: debug, SH colours (no colors_precomp), widths the kernel form does not serve, no colour gradient wanted
        assert not R._features_only_applies(32, feats, only_colors, True)
        assert not R._features_only_applies(32, _T(0), only_colors, False)
        assert not R._features_only_applies(3, _T(30), only_colors, False)
        assert not R._features_only_applies(32, feats, (True,) + (False,) * 8, False)
        # opt-in: everything requires grad, the switch decides
        assert R.enable_features_only_backward(True) is False
        assert R.features_only_backward_enabled()
        assert R._features_only_applies(32, feats, everything, False)
        assert not R._features_only_applies(32, feats, everything, True)
        assert not R._features_only_applies(3, _T(30), everything, False)
    finally:
        R.enable_features_only_backward(prev)


def _leaves(inp, dev, geometry_grad):
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)
    means3D, opac = t(inp.means3D).requires_grad_(geometry_grad), t(inp.opacities).requires_grad_(geometry_grad)
    scales, rots = t(inp.scales).requires_grad_(geometry_grad), t(inp.rotations).requires_grad_(geometry_grad)
    feats = t(inp.colors_precomp).requires_grad_(True)
    means2D = torch.zeros_like(means3D, requires_grad=geometry_grad)
    return means3D, means2D, feats, opac, scales, rots


def _render_and_backward(C, inp, dev, geometry_grad, dL):
    from diff_gaussian_rasterization_contrastive_f import GaussianRasterizationSettings
    _, _, GaussianRasterizer = R.make_rasterizer(C)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)
    settings = GaussianRasterizationSettings(
        image_height=inp.image_height, image_width=inp.image_width, tanfovx=inp.tanfovx, tanfovy=inp.tanfovy, bg=t(inp.bg),
        scale_modifier=inp.scale_modifier, viewmatrix=t(inp.viewmatrix), projmatrix=t(inp.projmatrix), sh_degree=0,
        campos=t(inp.campos), prefiltered=False, debug=False)
    means3D, means2D, feats, opac, scales, rots = _leaves(inp, dev, geometry_grad)
    color, _radii = GaussianRasterizer(settings)(means3D=means3D, means2D=means2D, shs=None, colors_precomp=feats, opacities=opac,
                                                 scales=scales, rotations=rots, cov3D_precomp=None)
    torch.autograd.backward(color, grad_tensors=t(dL))
    torch.cuda.synchronize()
    return dict(means3D=means3D, means2D=means2D, feats=feats, opac=opac, scales=scales, rots=rots, color=color.detach())


@pytest.mark.gpu
@pytest.mark.parametrize("C,W,H,bg", [(32, 208, 144, None), (32, 200, 136, "random"), (64, 176, 112, None), (48, 160, 96, "random"),
                                      (16, 96, 80, None), (128, 120, 72, None), (80, 104, 88, "random")])
def test_opt_in_matches_the_default_backward_and_the_oracle(C, W, H, bg):
    """Every input requires grad (the reference's feature training); with the switch on only colors_precomp receives one --
    the one the oracle and the default backward compute."""
    from oracle import saga_oracle as so
    dev = torch.device("cuda:0")
    inp = hp.make_inputs(7000, W, H, C, seed=77 + C, bg=bg, camera="orbit")
    dL = scenes.make_grad_image(C, H, W, seed=5)
    full = _render_and_backward(C, inp, dev, True, dL)
    prev = R.enable_features_only_backward(True)
    try:
        lean = _render_and_backward(C, inp, dev, True, dL)
    finally:
        R.enable_features_only_backward(prev)
    assert torch.equal(lean["color"], full["color"])
    for k in ("means3D", "means2D", "opac", "scales", "rots"):
        assert lean[k].grad is None, k
        assert full[k].grad is not None, k
    fwd = so.forward(inp)
    bwd = so.backward(inp, fwd, dL)
    got, ref = lean["feats"].grad.cpu().numpy(), full["feats"].grad.cpu().numpy()
    hp.assert_close("features-only dL_dcolors vs oracle", got, bwd.dL_dcolors, flip_frac=hp.GRAD_FLIP_FRAC)
    # against the default backward of the same call: the same alpha / T / contraction, only the order of the atomic sums differs
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 2e-5 * scale, (np.abs(got - ref).max(), scale)
    assert np.array_equal(got == 0, ref == 0)   # the same rows are touched


@pytest.mark.gpu
def test_automatic_when_only_the_features_require_grad():
    """Frozen geometry (requires_grad False everywhere but the features): autograd asks for one gradient, one is computed --
    no switch involved -- and it is the default backward's."""
    dev = torch.device("cuda:0")
    C, W, H = 32, 224, 160
    inp = hp.make_inputs(9000, W, H, C, seed=123, camera="orbit")
    dL = scenes.make_grad_image(C, H, W, seed=9)
    assert not R.features_only_backward_enabled()
    calls = []
    orig = R.rasterize_gaussians_backward_native

    def spy(*a, **k):
        calls.append(bool(k.get("features_only")))
        return orig(*a, **k)

    R.rasterize_gaussians_backward_native = spy
    try:
        full = _render_and_backward(C, inp, dev, True, dL)
        lean = _render_and_backward(C, inp, dev, False, dL)
    finally:
        R.rasterize_gaussians_backward_native = orig
    assert calls == [False, True]
    got, ref = lean["feats"].grad.cpu().numpy(), full["feats"].grad.cpu().numpy()
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()
    assert lean["means3D"].grad is None


@pytest.mark.gpu
def test_c_abi_refuses_what_the_form_does_not_serve():
    """MI_RAST_BWD_FEATURES_ONLY with RGB, or without colors_precomp: MI_RAST_ERR_INVALID and a message, nothing launched."""
    inp = hp.make_inputs(2000, 96, 64, 3, seed=3)
    g = hp.GpuRun(inp)
    g.forward(full_lists=False)
    L = _lib.load()
    rc = L.mi_rast_backward(g.P, 0, 0, 3, int(g.num_rendered), None, inp.image_width, inp.image_height, None, None, None, None, 1.0,
                            None, None, g.view.data_ptr(), g.proj.data_ptr(), g.campos.data_ptr(), float(inp.tanfovx),
                            float(inp.tanfovy), None, g.geom.data_ptr(), g.binning.data_ptr(), g.img.data_ptr(), None, None,
                            None, None, None, None, None, None, None, None, None, None, 0, _lib.MI_RAST_BWD_FEATURES_ONLY, None)
    assert rc == 1
    assert b"MI_RAST_BWD_FEATURES_ONLY" in L.mi_rast_last_error()


@pytest.mark.gpu
def test_bench_reports_both_opt_ins_as_separately_labelled_objects():
    """bench.py --features-only-grad --frozen-geometry: the headline fields are there as always, the two opt-ins appear as objects of
    their own (never in `value`), the features-only gradient equals the default backward's, the other leaves stay without gradients."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "2", "--points", "150000", "--no-cpu-baseline",
                          "--settle", "0", "--dist-blocks", "0", "--sustained-seconds", "0", "--features-only-grad", "--frozen-geometry",
                          "--views", "3", "--frozen-passes", "2"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["value"] > 0 and line["config"]["counters"]["P"] == 150000 and "geom_bwd" in line["config"]["stages_ms"]
    fo, fg = line["features_only_backward"], line["frozen_geometry"]
    assert fo["other_gradients_unset"] is True and fo["views_per_s"] > 0 and "geom_bwd" not in fo["stages_ms"]
    assert fo["dL_dfeatures_max_abs_diff_vs_default"] <= 2e-5 * fo["dL_dfeatures_max_abs"]
    assert fo["views_per_s_with_frozen_geometry_all_hits"] > 0
    assert fg["hits"] >= 3 and fg["views_per_s"] > 0
