"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs.

Bar: bit-exact on the integer tile/sort path; <= 1e-4 relative on the float image
and on every gradient.  Run on a real MI355X:  python -m pytest tests -m gpu -x -q
"""
import math
import os

import numpy as np
import pytest

from oracle import saga_oracle as so
from seganygaussians_amd import scenes
from tests import helpers as hp

pytestmark = pytest.mark.gpu


def _fwd_bwd(inp, grad_seed=1, check_int=True, use_mask_grad=False, skip=()):
    gpu = hp.GpuRun(inp).forward()
    fwd = so.forward(inp)
    assert fwd.rc == 0
    if check_int:
        hp.compare_integer_path(gpu, fwd)
    rep = hp.compare_float_forward(gpu, fwd)
    W, H, C = inp.image_width, inp.image_height, inp.channels
    dL = scenes.make_grad_image(C, H, W, seed=grad_seed)
    dLm = None
    if inp.mask is not None:
        dLm = (np.random.default_rng(grad_seed + 7).normal(0, 1, (1, H, W)) / (W * H)).astype(np.float32)
    grads = gpu.backward(dL, dLm)
    bwd = so.backward(inp, fwd, dL, None if dLm is None else dLm[0])
    rep.update(hp.compare_gradients(grads, bwd, skip=skip))
    # the product default lists only the overlaps that pass the cull: same blend lists, same results
    hp.compare_lean_with_full(inp, gpu, dL, dLm, grads)
    return rep, gpu, fwd


def test_cfg1_rgb_precomp():
    """BASELINE config 1: 10k Gaussians, 256x256, RGB (precomputed colours)."""
    rep, gpu, fwd = _fwd_bwd(hp.inputs_from_config("cfg1"))
    assert fwd.num_rendered > 40_000


def test_cfg1_sh_degree3_random_bg():
    inp = hp.inputs_from_config("cfg1", with_shs=True)
    inp.bg = np.array([0.2, 0.7, 0.4], np.float32)
    _fwd_bwd(inp)


@pytest.mark.parametrize("deg", [0, 1, 2])
def test_sh_lower_degrees(deg):
    inp = hp.make_inputs(3000, 160, 120, 3, seed=20 + deg, with_shs=True, sh_degree=deg, camera="orbit")
    _fwd_bwd(inp)


def test_features32_dense():
    """Reduced BASELINE config 3: 32-D features, dense overlap (long per-tile lists, early termination)."""
    inp = hp.make_inputs(60_000, 640, 360, 32, seed=3, focal=480.0, log_scale=math.log(0.03), log_scale_std=0.8)
    rep, gpu, fwd = _fwd_bwd(inp)
    c = fwd.state.counters()
    assert c["E"] < c["R"], "scene should exercise early termination"


@pytest.mark.parametrize("C", [32, 64])
def test_features_forward_x3_matches_f32_mfma(C, request):
    """The 32 / 64-channel forward (blend_fwd_wave.h: one wave per quadrant) accumulates on the bf16 matrix pipe with
    exactly split operands; MI_RAST_TILE_FWD selects the tile-batched kernel with the same accumulation (blend_fwd_x3.h, round 2's
    default), MI_RAST_F32_BLEND the f32-MFMA kernel (a bit-exact fmaf chain).  With expf for every pair (MI_RAST_EXACT_EXP; the
    other two kernels have no other form): same lists, same alpha / T / n_contrib and the same per-tile walk counters
    (bit-identical), and images that differ by rounding of the f32 accumulation only (a few ulp; same distance from the
    fp64-accumulating oracle).  The product default (hybrid exp, csrc/common.h) is judged against that in
    test_hybrid_exp_takes_the_decisions_of_expf.  (The two comparison kernels live in the profiling build: helpers.py.)"""
    if not hp.rerun_with_profiling_library(request):
        return
    inp = hp.make_inputs(60_000, 640, 360, C, seed=3, focal=480.0, log_scale=math.log(0.03), log_scale_std=0.8)
    x3 = hp.GpuRun(inp).forward(exact_exp=True)
    f32 = hp.GpuRun(inp).forward(f32_blend=True)
    a, b = x3.color.cpu().numpy().astype(np.float64), f32.color.cpu().numpy().astype(np.float64)
    assert not np.array_equal(a, b), "expected two different kernels (is the switch still wired?)"
    scale = np.abs(b).max()
    assert np.abs(a - b).max() <= 2e-6 * scale, (np.abs(a - b).max(), scale)  # a few ulp of the accumulated sums
    # against the fp64-accumulating oracle both sit at the f32 rounding level (measured RMS: 2.1e-8 vs 0.7e-8 at an image
    # scale of 0.5 -- the MFMA rounds its 16-product partial sums differently from sixteen chained fmaf)
    ref = so.forward(inp).color.astype(np.float64)
    rms_a, rms_b = np.sqrt(((a - ref) ** 2).mean()), np.sqrt(((b - ref) ** 2).mean())
    assert rms_a <= 1e-7 * scale and rms_b <= 1e-7 * scale, (rms_a, rms_b, scale)
    ia, ib = x3.img_fields(), f32.img_fields()
    np.testing.assert_array_equal(ia["final_T"].view(np.uint32), ib["final_T"].view(np.uint32))
    np.testing.assert_array_equal(ia["n_contrib"], ib["n_contrib"])
    tile = hp.GpuRun(inp).forward(tile_fwd=True)
    c = tile.color.cpu().numpy().astype(np.float64)
    assert np.abs(c - a).max() <= 2e-6 * scale and np.sqrt(((c - ref) ** 2).mean()) <= 1e-7 * scale
    ic = tile.img_fields()
    np.testing.assert_array_equal(ia["final_T"].view(np.uint32), ic["final_T"].view(np.uint32))
    np.testing.assert_array_equal(ia["n_contrib"], ic["n_contrib"])
    for k in ("tile_consumed", "tile_nsurv"):   # the wave kernel gathers them with atomicMax over the four quadrant waves
        np.testing.assert_array_equal(ia[k], ic[k], err_msg=k)
        np.testing.assert_array_equal(ia[k], ib[k], err_msg=k)


@pytest.mark.parametrize("case", ["features32", "features64", "rgb", "depth", "faint"])
def test_hybrid_exp_takes_the_decisions_of_expf(case):
    """Product default of the wave-per-quadrant forward (csrc/common.h "HYBRID evaluation"): v_exp_f32(x log2e) away from the
    alpha >= 1/255 cut, the device library's expf for every pair of entries in which some pixel comes within 4e-6 (relative) of
    it.  Against MI_RAST_EXACT_EXP (expf everywhere = what a build of the reference's kernels computes): the image agrees to
    3e-6 of its scale (measured: 1.3e-6), final_T to 1.2e-5 relative (1e-6 per factor), and n_contrib -- the last entry each pixel blended -- on all but a handful of
    pixels (T < 1e-4 stop decisions that sit within an ulp or two; no tolerance for the 1/255 decisions: a flipped one moves
    n_contrib on its pixel AND the image there by ~0.4 %, which the bound on the image would catch)."""
    if case == "features32":
        inp = hp.make_inputs(60_000, 640, 360, 32, seed=3, focal=480.0, log_scale=math.log(0.03), log_scale_std=0.8)
    elif case == "features64":
        inp = hp.make_inputs(20_000, 320, 208, 64, seed=5, log_scale=math.log(0.04))
    elif case == "rgb":
        inp = hp.inputs_from_config("cfg1", with_shs=True)
    elif case == "depth":
        inp = hp.make_inputs(20_000, 480, 272, 3, seed=6, with_shs=True, sh_degree=3, use_mask=True, bg="random")
    else:   # opacities scaled down: most pairs sit near the cut
        inp = hp.make_inputs(30_000, 320, 208, 32, seed=9, log_scale=math.log(0.05))
        inp.opacities = (np.asarray(inp.opacities) * 0.02).astype(np.float32)
    hy = hp.GpuRun(inp).forward(full_lists=False)
    ex = hp.GpuRun(inp).forward(full_lists=False, exact_exp=True)
    a, b = hy.color.cpu().numpy().astype(np.float64), ex.color.cpu().numpy().astype(np.float64)
    assert not np.array_equal(a, b), "expected two different evaluations (is the flag still wired?)"
    scale = np.abs(b).max()
    ih, ie = hy.img_fields(), ex.img_fields()
    flipped = (ih["n_contrib"] != ie["n_contrib"]).reshape(inp.image_height, inp.image_width)
    assert np.abs(a - b)[:, ~flipped].max() <= 3e-6 * scale, (np.abs(a - b)[:, ~flipped].max(), scale)
    assert np.abs(a - b).max() <= 1e-3 * scale    # (one entry more or less at T ~ 1e-4)
    # (a pixel whose T < 1e-4 stop fell the other way differs by one blended entry: counted with n_contrib below)
    same = ih["n_contrib"] == ie["n_contrib"]
    # final_T is a product of (1 - alpha) factors, each alpha within ~1e-6 (relative) of the expf form's: the relative difference of the
    # product is bounded by 1e-6 * sum alpha / (1 - alpha) <~ 1e-6 * ln(1 / T) -- 9e-6 where T has come down to 1e-4 (measured: 3.03e-6
    # on one pixel of 230 399, 1.3e-6 typical)
    np.testing.assert_allclose(ih["final_T"][same], ie["final_T"][same], rtol=1.2e-5, atol=0)
    nc = float((ih["n_contrib"] != ie["n_contrib"]).mean())
    print(f"hybrid vs expf ({case}): image max diff {np.abs(a - b).max() / scale:.1e} of scale, n_contrib differs on {nc:.1e} of the pixels")
    assert nc <= 2e-5, nc
    for k in ("out_mask", "out_depth"):
        if getattr(hy, k) is not None:
            x, y = getattr(hy, k).cpu().numpy().astype(np.float64), getattr(ex, k).cpu().numpy().astype(np.float64)
            assert np.abs(x - y)[:, ~flipped].max() <= 3e-6 * np.abs(y).max(), k
    # the backward (expf) re-takes the forward's decisions: gradients of the two forwards agree far inside the 1e-4 contract
    dL = scenes.make_grad_image(inp.channels, inp.image_height, inp.image_width, seed=1)
    dLm = None if inp.mask is None else (np.random.default_rng(8).normal(0, 1, (1, inp.image_height, inp.image_width))
                                         / (inp.image_width * inp.image_height)).astype(np.float32)
    gh, ge = hy.backward(dL, dLm), ex.backward(dL, dLm)
    for k, want in ge.items():
        hp.assert_close(k + " (hybrid vs expf forward)", gh[k], want, rtol=2e-4, flip_frac=max(hp.GRAD_FLIP_FRAC, 1.5 / max(1, want.size)))


def test_features32_odd_size_random_bg():
    """Image size not a multiple of 16: edge tiles with pixels outside the image (forward.cu:288-290,378)."""
    inp = hp.make_inputs(8_000, 203, 117, 32, seed=4, bg="random", camera="orbit")
    _fwd_bwd(inp)


def test_features64():
    """BASELINE config 5 channel count (64-D), reduced size."""
    inp = hp.make_inputs(20_000, 320, 208, 64, seed=5, log_scale=math.log(0.04))
    _fwd_bwd(inp)


@pytest.mark.parametrize("C", [16, 48, 80, 96, 112, 128, 256, 1, 2, 5, 8, 15, 17, 40, 50, 100, 255])
def test_feature_widths_in_channel_blocks(C):
    """Any width up to 256 (the reference: any compile-time NUM_CHANNELS): blended in channel blocks of 64 / 32 / 16, e.g.
    112 = 64 + 32 + 16, the last block partial when the width is no multiple of 16 (17 = 16 + 1 of 16; 50 = 32 + 16 + 2 of 16, with
    rows that are not 16-byte aligned; 255 = 3 x 64 + 32 + 16 + 15 of 16) -- image, every gradient (the geometry gradients are sums
    over the blocks), lean == full, against the oracle; random background so that the background term of dL/dalpha is split over
    the blocks as well.  (3 channels with precomputed colours take the RGB kernels: test_rgb_* above.)"""
    _fwd_bwd(hp.make_inputs(6000, 208, 144, C, seed=50 + C, log_scale=math.log(0.05), bg="random", camera="orbit"))


def test_unsupported_channel_counts_fail_loudly():
    import torch
    from seganygaussians_amd import rasterizer as R
    inp = hp.make_inputs(10, 32, 32, 3, seed=1)
    g = hp.GpuRun(inp)
    for C in (257, 272):   # (0 channels: refused by the glue's shape checks before the library sees it)
        with pytest.raises(RuntimeError, match="unsupported channel count"):
            R.rasterize_gaussians_native(C, False, torch.zeros(C, device="cuda"), g.means3D, torch.zeros(10, C, device="cuda"), g.opac,
                                         None, g.scales, g.rots, 1.0, g.cov, g.view, g.proj, inp.tanfovx, inp.tanfovy, 32, 32,
                                         g.shs, 0, g.campos, False, False)
    # the comparison kernels of earlier rounds are no part of the product library
    if os.environ.get("MI_RAST_LIB") != hp.PROF_LIB:
        inp32 = hp.make_inputs(10, 32, 32, 32, seed=1)
        for kw in (dict(tile_fwd=True), dict(f32_blend=True)):
            with pytest.raises(RuntimeError, match="comparison kernels of the profiling build"):
                hp.GpuRun(inp32).forward(**kw)


def test_depth_variant_with_mask():
    """BASELINE config 2 shape (RGB + mask + depth), reduced size, SH colours."""
    inp = hp.make_inputs(20_000, 480, 272, 3, seed=6, with_shs=True, sh_degree=3, use_mask=True, bg="random")
    _fwd_bwd(inp)


def test_cov3d_precomp():
    inp = hp.make_inputs(5_000, 256, 192, 3, seed=7, use_cov=True)
    _fwd_bwd(inp)


def test_scale_modifier():
    inp = hp.make_inputs(5_000, 256, 192, 32, seed=8, scale_modifier=1.6)
    _fwd_bwd(inp)


def test_mask_only_pair():
    """mask-only render pair (DEPTH forward_mask)."""
    import torch
    from seganygaussians_amd import rasterizer as R
    inp = hp.make_inputs(10_000, 320, 240, 3, seed=9, use_mask=True)
    g = hp.GpuRun(inp)
    nr, out_mask, radii, geom, binning, img = R.rasterize_mask_gaussians_native(
        g.means3D, g.opac, g.mask, g.scales, g.rots, 1.0, g.cov, g.view, g.proj, inp.tanfovx, inp.tanfovy,
        inp.image_height, inp.image_width, False, False)
    fwd = so.mask_forward(inp)
    assert nr == fwd.num_rendered
    np.testing.assert_array_equal(radii.cpu().numpy(), fwd.radii)
    hp.assert_close("mask", out_mask.cpu().numpy(), fwd.mask, flip_frac=hp.FLIP_FRAC)
    dLm = np.random.default_rng(3).normal(0, 1, (1, inp.image_height, inp.image_width)).astype(np.float32)
    gm = R.rasterize_mask_gaussians_backward_native(g.means3D, torch.as_tensor(dLm).cuda(), geom, nr, binning, img, False)
    want = so.mask_backward(inp, fwd, dLm[0])
    hp.assert_close("dL_dmask", gm.cpu().numpy().reshape(-1), want, flip_frac=hp.GRAD_FLIP_FRAC)


def test_mark_visible():
    from seganygaussians_amd import rasterizer as R
    inp = hp.make_inputs(5000, 64, 64, 3, seed=11, z_range=(-3.0, 5.0))
    g = hp.GpuRun(inp)
    got = R.mark_visible_native(g.means3D, g.view, g.proj).cpu().numpy()
    want = so.mark_visible(inp.means3D, inp.viewmatrix, inp.projmatrix)
    np.testing.assert_array_equal(got, want)
    assert 0 < want.sum() < len(want)


def test_edge_cases_empty_and_culled():
    import torch
    from seganygaussians_amd import rasterizer as R
    # P == 0: image all zeros (NOT background), num_rendered 0 (rasterize_points.cu:80-114)
    e = torch.empty(0, device="cuda")
    bg = torch.tensor([0.3, 0.4, 0.5], device="cuda")
    eye = torch.eye(4, device="cuda")
    nr, color, radii, *_ = R.rasterize_gaussians_native(3, False, bg, torch.empty(0, 3, device="cuda"), e, e, None, e, e,
                                                       1.0, e, eye, eye, 1.0, 1.0, 32, 48, e, 0, torch.zeros(3, device="cuda"),
                                                       False, False)
    assert nr == 0 and color.shape == (3, 32, 48) and float(color.abs().max()) == 0.0 and radii.numel() == 0
    # every Gaussian behind the camera: num_rendered == 0, every pixel == bg (rasterizer_impl.cu:310-317, forward.cu:383)
    inp = hp.make_inputs(500, 40, 24, 3, seed=12, z_range=(-5.0, -1.0), bg="random")
    gpu = hp.GpuRun(inp).forward()
    fwd = so.forward(inp)
    assert gpu.num_rendered == 0 == fwd.num_rendered
    np.testing.assert_array_equal(gpu.color.cpu().numpy(), fwd.color)
    np.testing.assert_array_equal(gpu.color.cpu().numpy(), np.broadcast_to(inp.bg[:, None, None], (3, 24, 40)))
    grads = gpu.backward(scenes.make_grad_image(3, 24, 40))
    assert all(float(np.abs(v).max()) == 0.0 for v in grads.values() if v.size)
    # single Gaussian exactly on a pixel centre
    inp = hp.make_inputs(1, 32, 32, 3, seed=13)
    inp.means3D = np.array([[0.0, 0.0, 4.0]], np.float32)
    _fwd_bwd(inp)


def test_non_rgb_without_colors_errors():
    import torch
    from seganygaussians_amd import rasterizer as R
    inp = hp.make_inputs(10, 32, 32, 3, seed=1, with_shs=True, sh_degree=1)
    g = hp.GpuRun(inp)
    with pytest.raises(RuntimeError, match="For non-RGB, provide precomputed Gaussian colors!"):
        R.rasterize_gaussians_native(32, False, torch.zeros(32, device="cuda"), g.means3D, torch.empty(0), g.opac, None,
                                     g.scales, g.rots, 1.0, g.cov, g.view, g.proj, inp.tanfovx, inp.tanfovy, 32, 32,
                                     g.shs, 1, g.campos, False, False)
    with pytest.raises(RuntimeError, match="means3D must have dimensions"):
        R.rasterize_gaussians_native(3, False, g.bg, g.means3D.reshape(-1), g.colors, g.opac, None, g.scales, g.rots, 1.0,
                                     g.cov, g.view, g.proj, inp.tanfovx, inp.tanfovy, 32, 32, g.shs, 1, g.campos, False,
                                     False)


def test_full_size_cfg3_properties():
    """BASELINE config 3 at FULL size (1M Gaussians, 1080p, 32-D) through size-independent properties:
    sortedness + stability of the key list, range consistency, sum(tiles_touched) == R, linearity of the
    render in the features, and the closed-form identity sum_ch-weighted gradient == <render, dL>."""
    import torch
    inp = hp.inputs_from_config("cfg3")
    gpu = hp.GpuRun(inp).forward()
    R = gpu.num_rendered
    g, b, im = gpu.geom_fields(), gpu.bin_fields(), gpu.img_fields()
    tt = g["tiles_touched"].astype(np.int64)
    assert int(tt.sum()) == R
    keys, vals = gpu.sorted_keys(), b["point_list"]
    assert np.all(keys[1:] >= keys[:-1]), "keys ascending"
    same = keys[1:] == keys[:-1]
    assert np.all(vals[1:][same] > vals[:-1][same]), "stable: equal keys keep Gaussian-index order"
    # every Gaussian appears exactly tiles_touched times (checksum of checksums over the whole list)
    np.testing.assert_array_equal(np.bincount(vals, minlength=len(tt)), tt)
    tiles = (keys >> np.uint64(32)).astype(np.int64)
    ranges = im["ranges"].reshape(-1, 2).astype(np.int64)
    counts = np.bincount(tiles, minlength=len(ranges))
    np.testing.assert_array_equal(ranges[:, 1] - ranges[:, 0], counts)
    # n_contrib is a 1-based position inside the pixel's own tile list
    tile_of_pix = (np.arange(inp.image_height)[:, None] // 16) * ((inp.image_width + 15) // 16) + np.arange(inp.image_width)[None, :] // 16
    assert np.all(im["n_contrib"].reshape(inp.image_height, inp.image_width) <= counts[tile_of_pix])
    # linearity in the features (bg = 0): render(2*f1 - 0.5*f2) == 2*render(f1) - 0.5*render(f2)
    c1 = gpu.color.clone()
    f1 = gpu.colors
    f2 = torch.roll(f1, 1, dims=1).contiguous()
    gpu.colors = f2
    c2 = gpu.forward().color.clone()
    gpu.colors = (2.0 * f1 - 0.5 * f2).contiguous()
    c3 = gpu.forward().color
    ref = 2.0 * c1 - 0.5 * c2
    err = float((c3 - ref).abs().max())
    assert err <= 2e-5 * max(1.0, float(ref.abs().max())), err
    # Euler identity: render is linear in features => sum(f * dL/df) == <render, dL_dout>  (bg = 0)
    gpu.colors = f1
    gpu.forward()
    dL = scenes.make_grad_image(32, inp.image_height, inp.image_width, seed=1)
    grads = gpu.backward(dL)
    lhs = float((f1.cpu().double().numpy() * grads["dL_dcolors"].astype(np.float64)).sum())
    rhs = float((gpu.color.cpu().double().numpy() * dL.astype(np.float64)).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(abs(rhs), 1e-9) + 1e-9, (lhs, rhs)


def _max_tile_list(gpu):
    r = gpu.img_fields()["ranges"].reshape(-1, 2).astype(np.int64)
    return int((r[:, 1] - r[:, 0]).max())


def test_depth_ties_dense_layers():
    """Thousands of Gaussians at EXACTLY the same depth (front camera: view z == world z): the (depth, index) order
    must come out of the bucketed depth ordering bit-exactly -- one bucket of 1500 pairs (LDS radix path), then
    buckets of 6000 and 12000 (> 2048: HBM ping-pong path) next to a 1-ulp neighbour layer and a far layer."""
    for n in (1500, 6000):
        inp = hp.make_inputs(n, 256, 160, 3, seed=11, log_scale=math.log(0.01), log_scale_std=0.3)
        inp.means3D = np.ascontiguousarray(inp.means3D, np.float32)
        inp.means3D[:, 2] = 4.0
        _fwd_bwd(inp)
    inp = hp.make_inputs(20000, 256, 160, 3, seed=12, log_scale=math.log(0.008), log_scale_std=0.3)
    m = np.ascontiguousarray(inp.means3D, np.float32)
    m[:12000, 2] = 3.0
    m[12000:16000, 2] = np.nextafter(np.float32(3.0), np.float32(4.0))
    m[16000:, 2] = 5.0
    rng = np.random.default_rng(5)
    inp.means3D = m[rng.permutation(20000)]          # layers interleaved in index order
    _fwd_bwd(inp)


def test_long_tile_lists_all_sort_classes():
    """Tile lists beyond 2048, 6144 and 12288 entries: the three size classes of the per-tile sort, including the one
    that ping-pongs through HBM."""
    inp = hp.make_inputs(90_000, 96, 64, 3, seed=13, focal=40.0, log_scale=math.log(0.25), log_scale_std=0.4,
                         z_range=(2.0, 9.0))
    rep, gpu, fwd = _fwd_bwd(inp)
    assert _max_tile_list(gpu) > 12288, _max_tile_list(gpu)
    inp = hp.make_inputs(20_000, 96, 64, 3, seed=14, focal=40.0, log_scale=math.log(0.2), log_scale_std=0.4,
                         z_range=(2.0, 9.0))
    rep, gpu, fwd = _fwd_bwd(inp)
    assert 2048 < _max_tile_list(gpu) <= 12288, _max_tile_list(gpu)


@pytest.mark.parametrize("case", ["rgb", "features32", "depth"])
def test_cull_is_exactly_conservative(case, request):
    """MI_RAST_NO_CULL hands EVERY overlap of the reference's tile lists to the blend kernels (all four quadrant bits).  The
    exact-conservative cull may only ever drop pairs that cannot reach alpha >= 1/255 anywhere in their quadrant, so the
    image, final_T, n_contrib, mask and depth must be BIT-IDENTICAL with and without it (a wrongly culled pair at
    alpha ~ 1/255 would move a pixel by ~0.4 %: no tolerance could tell that from a threshold flip), the integer path is
    untouched, and the gradients agree up to the order of the atomic sums."""
    if case == "features32" and not hp.rerun_with_profiling_library(request):   # (uses the f32-chain forward: profiling build)
        return
    if case == "rgb":
        inp = hp.inputs_from_config("cfg1", with_shs=True)
    elif case == "features32":
        inp = hp.make_inputs(60_000, 640, 360, 32, seed=3, focal=480.0, log_scale=math.log(0.03), log_scale_std=0.8)
    else:
        inp = hp.make_inputs(20_000, 480, 272, 3, seed=6, with_shs=True, sh_degree=3, use_mask=True, bg="random")
    # 32 channels: the f32 FMA-chain forward adds the pairs one by one in list order, so pairs with weight 0 change no bit; the
    # default bf16x3 forward sums 16 pairs per matrix instruction, and extra zero-weight pairs regroup its partial sums (ulps)
    f32 = True if inp.channels == 32 else None
    # (expf for every pair: the hybrid form falls back to expf per PAIR of queued entries, and without the cull the pairs differ)
    on = hp.GpuRun(inp).forward(f32_blend=f32, exact_exp=True)
    off = hp.GpuRun(inp).forward(no_cull=True, f32_blend=f32, exact_exp=True)
    assert off.num_rendered == on.num_rendered
    if f32:
        a, b = hp.GpuRun(inp).forward().color, hp.GpuRun(inp).forward(no_cull=True).color
        assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max())
    ids_on, _, cnt_on = on.blend_lists()
    ids_off, qm_off, cnt_off = off.blend_lists()
    assert cnt_off.sum() == off.num_rendered > cnt_on.sum() and np.all(qm_off == 15), "the cull should have been off"
    ion, ioff = on.img_fields(), off.img_fields()
    np.testing.assert_array_equal(ioff["ranges"], ion["ranges"])
    np.testing.assert_array_equal(off.bin_fields()["point_list"], on.bin_fields()["point_list"])
    for name, a, b in (("color", off.color, on.color), ("mask", off.out_mask, on.out_mask), ("depth", off.out_depth, on.out_depth)):
        if a is not None:
            np.testing.assert_array_equal(a.cpu().numpy().view(np.uint32), b.cpu().numpy().view(np.uint32), err_msg=name)
    np.testing.assert_array_equal(ioff["final_T"].view(np.uint32), ion["final_T"].view(np.uint32))
    np.testing.assert_array_equal(ioff["n_contrib"], ion["n_contrib"])
    W, H, C = inp.image_width, inp.image_height, inp.channels
    dL = scenes.make_grad_image(C, H, W, seed=1)
    dLm = None if inp.mask is None else (np.random.default_rng(8).normal(0, 1, (1, H, W)) / (W * H)).astype(np.float32)
    g_on, g_off = on.backward(dL, dLm), off.backward(dL, dLm)
    for k, want in g_on.items():
        hp.assert_close(k + " (cull off vs on)", g_off[k], want, rtol=2e-4, flip_frac=max(hp.GRAD_FLIP_FRAC, 1.5 / max(1, want.size)))


@pytest.mark.parametrize("C,kw", [(3, {}), (3, {"use_mask": True}), (32, {}), (48, {}), (64, {}), (16, {}), (40, {}),
                                  (32, {"P": 60_000, "W": 64, "H": 48})])
def test_forward_prefills_the_backward_accumulators(C, kw):
    """include/mi_rast.h dL_dcolor_next / MI_RAST_PREZERO_BWD: the forward's blend kernel leaves the backward's accumulators --
    the (P, channels) dL_dcolor buffer and the packed field gradients + work-queue counters in the geometry buffer -- zero-filled
    (what torch::zeros does in CF/rasterize_points.cu:153-159), and the backward then skips its fills.  Checked with NaN /
    0xFF left in the allocator's free blocks, for the wave-per-quadrant kernels (RGB, RGB + mask + depth, 32, 64, 48 = 32 + 16:
    the first block's launch takes the fill), for kernels that do not take it (16 channels: fill commands), for a width
    whose (P, C) block the partial last block shares (40 = 32 + 8 of 16), with an odd P (the (P, 3) buffer is not a whole number of 16-byte units), and with more zeros than image (60 000
    Gaussians on 64 x 48 pixels: the kernel takes the fill only while it at most doubles its own stores).  Gradients equal those of a plain run."""
    import torch
    from seganygaussians_amd import _lib
    kw = dict(kw)
    tile_fwd = kw.pop("tile_fwd", None)
    P, W, H = kw.pop("P", 7001), kw.pop("W", 200), kw.pop("H", 136)
    inp = hp.make_inputs(P, W, H, C, seed=41, camera="orbit", bg="random", **kw)
    dL = scenes.make_grad_image(C, H, W, seed=5)
    dLm = None if inp.mask is None else (np.random.default_rng(9).normal(0, 1, (1, H, W)) / (W * H)).astype(np.float32)
    plain = hp.GpuRun(inp).forward(full_lists=False, tile_fwd=tile_fwd)
    g_plain = plain.backward(dL, dLm)
    geom_bytes, off = _lib.geometry_layout(P)
    dev = plain.dev
    junk = [torch.full((P, C), float("nan"), device=dev), torch.full((int(plain.geom.numel()),), 0xFF, dtype=torch.uint8, device=dev)]
    del plain, junk
    run = hp.GpuRun(inp).forward(full_lists=False, tile_fwd=tile_fwd, prezero="always")   # ("always": also when P > H W)
    pre = run.geom.mi_prezero
    assert run.geom.mi_pack_zeroed is True
    torch.cuda.synchronize()
    assert tuple(pre.shape) == (P, C) and pre.dtype == torch.float32
    assert int(torch.count_nonzero(pre.view(torch.int32))) == 0, "dL_dcolor_next not zero-filled"
    pack = run._view(run.geom, off["bwd_pack"], 8 * P + 16 * 8 * 16, np.uint32)
    assert not pack.any(), "bwd_pack / queue counters not zero-filled"
    g = run.backward(dL, dLm, prezeroed=pre)
    for k, want in g_plain.items():
        hp.assert_close(k + " (prezeroed vs plain)", g[k], want, rtol=2e-4, flip_frac=max(hp.GRAD_FLIP_FRAC, 1.5 / max(1, want.size)))
    hidden = run.radii.cpu().numpy() == 0
    assert hidden.any() and not g["dL_dcolors"][hidden].any() and np.isfinite(g["dL_dcolors"]).all()
    # a second backward on the same buffers must fill for itself (the autograd Functions hand the tensor over once)
    g2 = run.backward(dL, dLm)
    for k, want in g_plain.items():
        hp.assert_close(k + " (second backward)", g2[k], want, rtol=2e-4, flip_frac=max(hp.GRAD_FLIP_FRAC, 1.5 / max(1, want.size)))


def test_4k_image_walked_in_bands():
    """3840 x 2160 = 32 400 tiles: more than one launch of the count / emit passes has LDS counters for (29 632), so the
    host walks the image in two bands of tile rows.  Same bit-exact integer path, same tolerances."""
    inp = hp.make_inputs(40_000, 3840, 2160, 3, seed=21, focal=3000.0, log_scale=math.log(0.06), log_scale_std=0.7, bg="random")
    rep, gpu, fwd = _fwd_bwd(inp)
    assert fwd.num_rendered > 1_000_000
    ranges = gpu.img_fields()["ranges"].reshape(-1, 2).astype(np.int64)
    assert len(ranges) == 240 * 135 and (ranges[30000:, 1] - ranges[30000:, 0]).sum() > 0, "the second band must hold overlaps"


def test_5k_image_range_scan_in_segments():
    """5120 x 2880 = 57 600 tiles: more than the range scan's workgroup holds in LDS (40 896), so tile_ranges_kernel walks the
    tile totals in two segments with the running total carried over (and the count / emit passes walk three bands).  Same
    bit-exact integer path, same tolerances; the reference has no image-size limit (rasterizer_impl.cu:198-336)."""
    inp = hp.make_inputs(30_000, 5120, 2880, 3, seed=22, focal=4000.0, log_scale=math.log(0.06), log_scale_std=0.7, bg="random")
    rep, gpu, fwd = _fwd_bwd(inp)
    assert fwd.num_rendered > 1_000_000
    ranges = gpu.img_fields()["ranges"].reshape(-1, 2).astype(np.int64)
    assert len(ranges) == 320 * 180 and (ranges[40896:, 1] - ranges[40896:, 0]).sum() > 0, "the second segment must hold overlaps"
    # what remains refused: more than 1023 tiles across or 2047 down (the span records pack tile coordinates in 10 + 11 bits)
    from seganygaussians_amd import rasterizer as R
    g = hp.GpuRun(hp.make_inputs(100, 64, 64, 3, seed=1))
    with pytest.raises(RuntimeError, match="image too large"):
        R.rasterize_gaussians_native(3, False, g.bg, g.means3D, g.colors, g.opac, None, g.scales, g.rots, 1.0, g.cov, g.view,
                                     g.proj, 1.0, 1.0, 16, 16400, g.shs, 0, g.campos, False, False)


def test_lean_expf_is_the_device_expf(tmp_path):
    """gauss_exp<true> (csrc/common.h) is the device library's expf restated with a clamp instead of its two range selects:
    tools/expf_check.hip compares the two on ALL 2^32 f32 inputs -- bit-identical on [-103.28, 0], and 0 or the smallest
    denormal where expf underflows to 0 (the blend kernels' >= 1/255 test cannot tell those apart)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "expf_check")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize",
                    "-Wno-unused-result", "-o", exe, os.path.join(root, "tools", "expf_check.hip")], check=True, timeout=600)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0 and r.stdout.startswith("mismatches 0 0 "), r.stdout + r.stderr
