"""The pin: the REFERENCE's own rasterizer core (oracle/_ref, built by oracle/build_ref.py from the sources under
/root/reference -- test-only translation, see that file) run on the MI355X next to the CPU oracle and the HIP product path.

  1. oracle  vs reference : pins oracle/saga_rast_oracle.c (and with it every other parity test and golden fixture);
  2. product vs reference : the product path against the reference itself, incl. the BASELINE configs at FULL size
                            (cfg3 1M/1080p/32-D fwd+bwd, cfg2 1M/1080p SH RGB+mask+depth, cfg5 5M/1600x1063/64-D).

Bar: integer tile/sort path (radii, tiles_touched, depth/means2D bits, the sorted 64-bit key list, point_list, ranges,
num_rendered) bit-exact; image / final_T / every gradient within 1e-4 (tests/helpers.py).  The reference build used for the
bit-exact comparisons is compiled with -ffp-contract=off (the numeric contract); what the default
contraction of an out-of-the-box hipify build moves is measured in test_reference_contraction_sensitivity.

The file name sorts last on purpose: a missing oracle/_ref is a FAILURE here (not a skip), and `pytest -x` must not let
that hide the rest of the suite.
"""
import math

import numpy as np
import pytest

from oracle import saga_oracle as so
from oracle import saga_ref as sr
from seganygaussians_amd import scenes
from tests import helpers as hp

pytestmark = pytest.mark.gpu


def _dL(inp, seed=1):
    W, H, C = inp.image_width, inp.image_height, inp.channels
    dL = scenes.make_grad_image(C, H, W, seed=seed)
    dLm = None
    if inp.mask is not None:
        dLm = (np.random.default_rng(seed + 7).normal(0, 1, (1, H, W)) / (W * H)).astype(np.float32)
    return dL, dLm


def _pin_oracle(inp, variant=None):
    """oracle vs reference on one input: forward (integer path bit-exact) + every gradient."""
    ref = sr.RefRun(inp, variant)
    rf = ref.forward()
    of = so.forward(inp)
    assert of.rc == 0
    rep = hp.compare_forward_outs(of, rf, "oracle vs ref:")
    dL, dLm = _dL(inp)
    rb = ref.backward(dL, dLm)
    ob = so.backward(inp, of, dL, None if dLm is None else dLm[0])
    rep.update(hp.compare_gradients(hp.grads_as_dict(ob), rb))
    return rep, ref, rf, rb


# The product's forward evaluates exp() in the hybrid form of csrc/common.h: every alpha >= 1/255 decision is expf's, i.e. the
# reference build's; alpha itself differs by <= 1e-6 relative, so T does too, and a T < 1e-4 stop decision that sits within an
# ulp or two can fall the other way: n_contrib may differ from the reference's on a few pixels in a million.  With
# MI_RAST_EXACT_EXP it is equal on every pixel (test_full_size_cfg3_product_vs_ref_and_oracle checks both).
NC_MISMATCH_HYBRID = 5e-6


def _product_vs_ref(inp, variant=None, lean=True, fast_exp=None):
    """product (full-list mode for the integer path) vs reference; then the product default (lean lists) vs full."""
    ref = sr.RefRun(inp, variant)
    rf = ref.forward()
    gpu = hp.GpuRun(inp).forward(fast_exp=fast_exp)
    hp.compare_integer_path(gpu, rf)
    rep = hp.compare_float_forward(gpu, rf)
    dL, dLm = _dL(inp)
    grads = gpu.backward(dL, dLm)
    rb = ref.backward(dL, dLm)
    rep.update(hp.compare_gradients(grads, rb))
    rep["stats"] = hp.error_stats(grads, rb)
    if lean:
        hp.compare_lean_with_full(inp, gpu, dL, dLm, grads, fast_exp=fast_exp)
    return rep, gpu, ref, rf, rb, grads


def test_reference_libraries_present():
    for v in ("cf32", "cf32_fast", "cf64", "cf16", "cf128", "cf8", "cf40", "cf100", "base3", "depth3", "knn"):
        assert sr.available(v), f"oracle/_ref/libsaga_ref_{v}.so missing: run python oracle/build_ref.py in the build container"
    assert sr.lib("cf32").saga_ref_channels() == 32 and sr.lib("cf64").saga_ref_channels() == 64
    assert sr.lib("base3").saga_ref_channels() == 3 and sr.lib("depth3").saga_ref_is_depth() == 1


# ---------------------------------------------------------------------------------------------------------------------
# 1. the oracle is pinned against the reference
# ---------------------------------------------------------------------------------------------------------------------

def test_pin_cfg1_rgb_precomp():
    _pin_oracle(hp.inputs_from_config("cfg1"))


def test_pin_cfg1_sh_degree3_random_bg():
    inp = hp.inputs_from_config("cfg1", with_shs=True)
    inp.bg = np.array([0.2, 0.7, 0.4], np.float32)
    _pin_oracle(inp)


@pytest.mark.parametrize("deg", [0, 1, 2])
def test_pin_sh_lower_degrees(deg):
    _pin_oracle(hp.make_inputs(3000, 160, 120, 3, seed=20 + deg, with_shs=True, sh_degree=deg, camera="orbit"))


def test_pin_features32_dense():
    _pin_oracle(hp.make_inputs(60_000, 640, 360, 32, seed=3, focal=480.0, log_scale=math.log(0.03), log_scale_std=0.8))


def test_pin_features32_odd_size_random_bg():
    _pin_oracle(hp.make_inputs(8_000, 203, 117, 32, seed=4, bg="random", camera="orbit"))


def test_pin_features64():
    _pin_oracle(hp.make_inputs(20_000, 320, 208, 64, seed=5, log_scale=math.log(0.04)))


@pytest.mark.parametrize("C", [16, 128])
def test_pin_features16_and_128(C):
    """The reference compiles one NUM_CHANNELS into its kernels (CF/cuda_rasterizer/config_contrastive_f.h:15); builds with 16
    and 128 pin the oracle at the ends of the range the product covers with channel blocks."""
    _pin_oracle(hp.make_inputs(12_000, 320, 208, C, seed=30 + C, log_scale=math.log(0.04), bg="random"))


def test_pin_depth_variant_with_mask():
    _pin_oracle(hp.make_inputs(20_000, 480, 272, 3, seed=6, with_shs=True, sh_degree=3, use_mask=True, bg="random"))


def test_pin_cov3d_precomp_and_scale_modifier():
    _pin_oracle(hp.make_inputs(5_000, 256, 192, 3, seed=7, use_cov=True))
    _pin_oracle(hp.make_inputs(5_000, 256, 192, 32, seed=8, scale_modifier=1.6))


def test_pin_depth_ties_and_long_lists():
    inp = hp.make_inputs(6000, 256, 160, 3, seed=11, log_scale=math.log(0.01), log_scale_std=0.3)
    inp.means3D = np.ascontiguousarray(inp.means3D, np.float32)
    inp.means3D[:, 2] = 4.0
    _pin_oracle(inp)
    _pin_oracle(hp.make_inputs(20_000, 96, 64, 3, seed=14, focal=40.0, log_scale=math.log(0.2), log_scale_std=0.4,
                               z_range=(2.0, 9.0)))


def test_pin_mask_only_pair_and_mark_visible():
    inp = hp.make_inputs(10_000, 320, 240, 3, seed=9, use_mask=True)
    ref = sr.RefRun(inp, "depth3")
    rf = ref.mask_forward()
    of = so.mask_forward(inp)
    assert rf.num_rendered == of.num_rendered
    np.testing.assert_array_equal(rf.radii, of.radii)
    hp.assert_close("mask (oracle vs ref)", of.mask, rf.mask, flip_frac=hp.FLIP_FRAC)
    dLm = np.random.default_rng(3).normal(0, 1, (1, inp.image_height, inp.image_width)).astype(np.float32)
    hp.assert_close("dL_dmask (oracle vs ref)", so.mask_backward(inp, of, dLm[0]), ref.mask_backward(dLm),
                    flip_frac=hp.GRAD_FLIP_FRAC)
    inp = hp.make_inputs(5000, 64, 64, 3, seed=11, z_range=(-3.0, 5.0))
    np.testing.assert_array_equal(so.mark_visible(inp.means3D, inp.viewmatrix, inp.projmatrix), sr.mark_visible(inp))


def test_pin_all_culled_and_single():
    inp = hp.make_inputs(500, 40, 24, 3, seed=12, z_range=(-5.0, -1.0), bg="random")
    rf = sr.RefRun(inp).forward()
    of = so.forward(inp)
    assert rf.num_rendered == 0 == of.num_rendered
    np.testing.assert_array_equal(rf.color, of.color)
    inp = hp.make_inputs(1, 32, 32, 3, seed=13)
    inp.means3D = np.array([[0.0, 0.0, 4.0]], np.float32)
    _pin_oracle(inp)


def test_pin_oracle_at_full_cfg3():
    """The oracle against the reference at the BENCHMARKED size (1M Gaussians, 1080p, 32-D, fwd+bwd)."""
    rep, ref, rf, rb = _pin_oracle(hp.inputs_from_config("cfg3"))
    assert rf.num_rendered > 10_000_000


def test_reference_contraction_sensitivity():
    """What hipcc's default FMA contraction moves in the reference itself (strict vs fast build of the SAME sources):
    reported, and bounded -- a handful of radii / tile counts out of a million, nothing else."""
    inp = hp.inputs_from_config("cfg3", P=200_000)
    a = sr.RefRun(inp, "cf32").forward()
    b = sr.RefRun(inp, "cf32_fast").forward()
    d_radii = int((a.radii != b.radii).sum())
    d_tiles = int((a.state.field(so.F_TILES_TOUCHED) != b.state.field(so.F_TILES_TOUCHED)).sum())
    print(f"contraction moves {d_radii} radii, {d_tiles} tile counts of {len(a.radii)}; R {a.num_rendered} vs {b.num_rendered}")
    assert d_radii <= 2e-3 * len(a.radii)
    hp.assert_close("color strict vs fast", b.color, a.color, flip_frac=1e-3)


# ---------------------------------------------------------------------------------------------------------------------
# 2. the product against the reference
# ---------------------------------------------------------------------------------------------------------------------

def test_product_vs_ref_cfg1():
    _product_vs_ref(hp.inputs_from_config("cfg1"))
    inp = hp.inputs_from_config("cfg1", with_shs=True)
    inp.bg = np.array([0.2, 0.7, 0.4], np.float32)
    _product_vs_ref(inp)


def test_product_vs_ref_reduced_cfg3_cfg2_cfg5():
    _product_vs_ref(hp.inputs_from_config("cfg3", P=100_000))
    _product_vs_ref(hp.make_inputs(20_000, 480, 272, 3, seed=6, with_shs=True, sh_degree=3, use_mask=True, bg="random"))
    _product_vs_ref(hp.make_inputs(20_000, 320, 208, 64, seed=5, log_scale=math.log(0.04)))


@pytest.mark.parametrize("C", [16, 128, 8, 40, 100])
def test_product_vs_ref_channel_blocks(C):
    """Feature widths other than 32 / 64: the product blends them in channel blocks of 64 / 32 / 16 (mi_rast.hip:
    channels_supported), the last one partial when the width is no multiple of 16 (8 = 8 of 16, 40 = 32 + 8 of 16,
    100 = 64 + 32 + 4 of 16) -- against reference builds with those NUM_CHANNELS, fwd + bwd, lean == full."""
    _product_vs_ref(hp.make_inputs(12_000, 320, 208, C, seed=30 + C, log_scale=math.log(0.04), bg="random"))
    _product_vs_ref(hp.make_inputs(4_000, 203, 117, C, seed=40 + C, camera="orbit"))


# The norm-wise error of dL_dcov3D / dL_dscales / dL_drotations hangs on a handful of cancellation-prone rows of the (shared,
# binary32) per-Gaussian geometry backward: a single Gaussian can carry it (cfg5, round 4: one row 243 tolerances off = the
# product's 3.0e-5 against the reference's 1.3e-5, both sides' rows-outside-tolerance shares equal), and WHICH row that is differs
# between the implementations and, through the order of the f32 atomics, from run to run (the REFERENCE against the exact-pairs
# oracle in round 3: cov3D 8.1e-6 .. 8.2e-5, scales 1.2e-5 .. 6.7e-5, rotations 3.3e-5 .. 1.3e-4).  These three are therefore
# judged on DRAWS taken inside the test -- the backward of the reference and of the product each run SWING_DRAWS times on the
# same forward -- and on the TRIMMED norm-wise error (helpers.trimmed_norm_error: the 32 worst rows of either side left out of
# the numerator), compared like with like: the product's MEDIAN within 2x the reference's MEDIAN and its LARGEST draw within 2x the
# reference's largest.  The untrimmed norm keeps a gross bound (8x the reference's largest draw), and the rows left out are still
# counted by the rows-outside-tolerance share (2x the reference's).  Round 3 bounded the three by constants copied from earlier runs instead.  Every other tensor is stable (measured
# ratios 0.9 .. 1.1) and keeps the 4x-of-this-run bound on the plain norm.
SWING = ("dL_dcov3D", "dL_dscales", "dL_drotations")
SWING_DRAWS = 3


def _norm_bound(k, ref_norm, factor=4.0):
    return factor * ref_norm + 1e-7


def _swing_draws(gpu, ref, dL, dLm, ob, mine, theirs):
    """{tensor: {statistic: (product's median, reference's max, product's max, reference's median)}} of the (trimmed) norm-wise error against `ob` over SWING_DRAWS
    backward runs each (the draws already in `mine` / `theirs` count as the first)."""
    stats = ("norm_trim", "norm")
    prod = {k: {st: [mine[k][st]] for st in stats} for k in SWING if k in mine}
    refd = {k: {st: [theirs[k][st]] for st in stats} for k in prod}
    for _ in range(SWING_DRAWS - 1):
        g = hp.error_stats(gpu.backward(dL, dLm), ob, names=SWING)
        r = hp.error_stats(hp.grads_as_dict(ref.backward(dL, dLm)), ob, names=SWING)
        for k in prod:
            for st in stats:
                prod[k][st].append(g[k][st])
                refd[k][st].append(r[k][st])
    return {k: {st: (float(np.median(prod[k][st])), float(max(refd[k][st])), float(max(prod[k][st])), float(np.median(refd[k][st])))
                for st in stats} for k in prod}


def _judge_against_reference(what, mine, theirs, swing):
    bad = []
    for k, s in mine.items():
        r = theirs[k]
        print(f"{what} {k}: norm {s['norm']:.2e} (trimmed {s['norm_trim']:.2e}) rows outside {s['row_frac']:.2e} worst {s['row_worst']:.1f}"
              f" | reference's own: norm {r['norm']:.2e} (trimmed {r['norm_trim']:.2e}) rows {r['row_frac']:.2e} worst {r['row_worst']:.1f}")
        assert not s["zero_rows_touched"]
        if k in swing:
            (med_t, ref_t, max_t, ref_med_t), (med, ref_max, _, _) = swing[k]["norm_trim"], swing[k]["norm"]
            print(f"{what} {k}: over {SWING_DRAWS} draws each, trimmed norm: product median {med_t:.2e} / max {max_t:.2e}, reference median "
                  f"{ref_med_t:.2e} / max {ref_t:.2e}; plain norm: product median {med:.2e}, reference max {ref_max:.2e}")
            # the trimmed norm is the stable statistic (round 5, five full runs: product / reference between 0.8 and 1.1 on every case): it is
            # compared like with like -- median with median, largest draw with largest draw; the plain norm, which one row carries and which
            # swings tenfold between draws of either side, keeps its gross bound
            norm_ok = med_t <= 2 * ref_med_t + 1e-7 and max_t <= 2 * ref_t + 1e-7 and med <= 8 * ref_max + 1e-7
        else:
            norm_ok = s["norm"] <= _norm_bound(k, r["norm"])
        if not (norm_ok and s["row_frac"] <= 2 * r["row_frac"] + 5e-5):
            bad.append((k, s, r, swing.get(k)))
    assert not bad, bad


def _check_stats(stats, ref_stats, what, row_floor=1e-3, norm_floor=1e-4):
    """Norm-wise and per-row errors of the product (vs the fp64-accumulating oracle or vs the reference) next to the
    reference's own f32-atomic noise against the same yardstick."""
    for k, s in stats.items():
        r = ref_stats.get(k) if ref_stats else None
        print(f"{what} {k}: norm {s['norm']:.2e} rows outside {s['row_frac']:.2e} worst {s['row_worst']:.1f}"
              + (f" | reference's own: norm {r['norm']:.2e} rows {r['row_frac']:.2e} worst {r['row_worst']:.1f}" if r else ""))
        assert not s["zero_rows_touched"], f"{what} {k}: a Gaussian the reference leaves at exactly 0 got a gradient"
        assert s["norm"] <= max(norm_floor, _norm_bound(k, r["norm"] if r else 0, 3.0)), (what, k, s, r)
        assert s["row_frac"] <= max(row_floor, 3 * (r["row_frac"] if r else 0)), (what, k, s, r)


_CFG3_ORACLE = {}


def _cfg3_oracle(inp, dL):
    """The oracle on full cfg3 with binary64 per-pair values AND sums (so.backward(exact_pairs=True): decisions stay binary32):
    the yardstick shares neither implementation's per-pair rounding.  Computed once for the tests below."""
    if "b" not in _CFG3_ORACLE:
        of = so.forward(inp)
        _CFG3_ORACLE["f"], _CFG3_ORACLE["b"] = of, so.backward(inp, of, dL, exact_pairs=True)
    return _CFG3_ORACLE["f"], _CFG3_ORACLE["b"]


def _cfg3_stats(fast_exp):
    inp = hp.inputs_from_config("cfg3")
    rep, gpu, ref, rf, rb, grads = _product_vs_ref(inp, fast_exp=fast_exp)
    dL, _ = _dL(inp)
    of, ob = _cfg3_oracle(inp, dL)
    mine = hp.error_stats(grads, ob)
    theirs = hp.error_stats(hp.grads_as_dict(rb), ob)
    img_norm = hp.norm_error(gpu.color.cpu().numpy(), of.color)
    ref_norm = hp.norm_error(rf.color, of.color)
    print(f"cfg3 image norm-wise error against the fp64 oracle: product {img_norm:.2e}, reference {ref_norm:.2e}")
    nc_mismatch = float((gpu.img_fields()["n_contrib"] != rf.state.field(so.F_N_CONTRIB)).mean())
    print(f"cfg3: n_contrib differs from the reference on {nc_mismatch:.2e} of the pixels")
    swing = {} if fast_exp else _swing_draws(gpu, ref, dL, None, ob, mine, theirs)
    return mine, theirs, img_norm, ref_norm, nc_mismatch, swing


def test_full_size_cfg3_product_vs_ref_and_oracle():
    """BASELINE config 3 at FULL size, product default: product vs reference (integer path bit-exact in full-list mode;
    image, final_T, all gradients; lean == full), with norm-wise and per-row (median floor) error statistics measured
    against the oracle for BOTH the product and the reference.  The yardstick evaluates every per-pair value AND every sum in
    binary64 (so.backward(exact_pairs=True)); the default oracle mode rounds per-pair values like the reference's kernel, which
    hides that part of the reference's error (~2.4e-7 norm-wise) and counts it against everybody else.  The product takes
    every alpha >= 1/255 decision with the device library's expf like the reference's kernels (FEAT/forward.cu:343,
    backward.cu:483; forward: the hybrid form of csrc/common.h), so the same pairs blend as in the reference; the per-pixel
    contributor counts agree on all but a few pixels in a million (T < 1e-4 stops within an ulp or two; EQUAL on every pixel
    under MI_RAST_EXACT_EXP, checked at the end), and the product's gradient errors are those of the reference itself (measured ratios 0.9 .. 1.1; the figures of dL_dcov3D / scales /
    rotations hang on a few cancellation-prone rows and move by 2x from run to run in BOTH implementations, the order of f32
    atomics not being deterministic) -- asserted at 2x (rows outside tolerance) / 4x (norm-wise; the three swinging tensors on draws, see SWING)."""
    mine, theirs, img_norm, ref_norm, nc_mismatch, swing = _cfg3_stats(fast_exp=None)
    # 4x norm-wise of this run's reference draw (the three swinging tensors: product's median of SWING_DRAWS draws within 2x the
    # reference's largest, see SWING above), 2x on the row count
    _judge_against_reference("cfg3", mine, theirs, swing)
    assert nc_mismatch <= NC_MISMATCH_HYBRID
    assert img_norm <= max(1e-6, 3 * ref_norm)
    # expf for every pair: every decision falls as in the reference build -- the per-pixel contributor counts are EQUAL everywhere
    inp = hp.inputs_from_config("cfg3")
    ex = hp.GpuRun(inp).forward(exact_exp=True)
    rf = sr.RefRun(inp, None).forward()
    assert float((ex.img_fields()["n_contrib"] != rf.state.field(so.F_N_CONTRIB)).mean()) == 0.0


def test_full_size_cfg3_fast_exp_mode():
    """MI_RAST_FAST_EXP (include/mi_rast.h): v_exp_f32(x * log2e), ~5 ulp.  A handful of pairs within a few ulp of the
    alpha >= 1/255 cut change side; still inside the contract (1e-4 element-wise, checked by _product_vs_ref) and within
    3x the reference's own norm-wise / per-row noise (floors 1e-4 / 1e-3)."""
    mine, theirs, img_norm, ref_norm, nc_mismatch, _ = _cfg3_stats(fast_exp=True)
    _check_stats(mine, theirs, "cfg3 fast-exp product-vs-oracle")
    assert nc_mismatch < 1e-4
    assert img_norm <= max(1e-5, 3 * ref_norm)


def test_full_size_cfg2_forward_vs_ref():
    """BASELINE config 2 at FULL size: 1M Gaussians, 1080p, SH degree 3 RGB + mask + depth, forward (DEPTH package)."""
    inp = hp.inputs_from_config("cfg2", with_shs=True, use_mask=True)
    ref = sr.RefRun(inp, "depth3")
    rf = ref.forward()
    gpu = hp.GpuRun(inp).forward()
    hp.compare_integer_path(gpu, rf)
    hp.compare_float_forward(gpu, rf)
    lean = hp.compare_lean_with_full(inp, gpu)
    for name, a, b in (("color", gpu.color.cpu().numpy(), rf.color), ("mask", gpu.out_mask.cpu().numpy(), rf.mask),
                       ("depth", gpu.out_depth.cpu().numpy(), rf.depth)):
        n = hp.norm_error(a, b)
        print(f"cfg2 {name} norm-wise error vs reference {n:.2e}")
        assert n <= 2e-5, (name, n)


def _assert_like_reference(inp, what):
    """Product vs reference (integer path bit-exact in full-list mode, floats within 1e-4, lean == full), then product AND
    reference each against the fp64-accumulating oracle on the same input: the product's norm-wise error must stay within 4x
    and its rows outside tolerance within 2x of the reference's own f32-atomic noise; n_contrib equal on every pixel."""
    rep, gpu, ref, rf, rb, grads = _product_vs_ref(inp)
    dL, dLm = _dL(inp)
    of = so.forward(inp)
    ob = so.backward(inp, of, dL, exact_pairs=True)   # binary64 per-pair values: a yardstick that shares nobody's rounding
    mine = hp.error_stats(grads, ob)
    theirs = hp.error_stats(hp.grads_as_dict(rb), ob)
    nc_mismatch = float((gpu.img_fields()["n_contrib"] != rf.state.field(so.F_N_CONTRIB)).mean())
    print(f"{what}: R = {rf.num_rendered}, n_contrib differs from the reference on {nc_mismatch:.2e} of the pixels")
    _judge_against_reference(what, mine, theirs, _swing_draws(gpu, ref, dL, dLm, ob, mine, theirs))
    assert nc_mismatch <= NC_MISMATCH_HYBRID
    return rf


@pytest.mark.parametrize("rank", [3, 7])
def test_full_size_cfg4_orbit_poses_product_vs_ref_and_oracle(rank):
    """BASELINE config 4 = config 3's Gaussians seen from 8 orbit poses, one per rank (bench.py:136-137).  Ranks 3 and 7 at
    FULL size, judged like the front view of config 3."""
    rf = _assert_like_reference(hp.inputs_from_config("cfg3", camera=hp.rank_camera(rank)), f"cfg4 rank {rank}")
    assert rf.num_rendered > 5_000_000


def test_full_size_cfg5_product_vs_ref_and_oracle():
    """BASELINE config 5 at FULL size: 5M Gaussians, 1600x1063, 64-D features, fwd+bwd, judged like config 3 (the fp64 oracle
    runs on exactly this input; no loosened floors)."""
    rf = _assert_like_reference(hp.inputs_from_config("cfg5"), "cfg5")
    assert rf.num_rendered > 25_000_000


def test_dist_cuda2_vs_reference_simple_knn():
    """distCUDA2 of the product (include/mi_knn.h) against the reference's own SimpleKNN::knn (oracle/_ref knn build,
    submodules/simple-knn/simple_knn.cu:185-218): both are exact searches with the same unfused distance expression."""
    import torch
    from seganygaussians_amd import knn
    rng = np.random.default_rng(0)
    c = rng.normal(0, 3, (30, 3))
    pts = (c[rng.integers(0, 30, 200_000)] + rng.normal(0, 0.2, (200_000, 3))).astype(np.float32)
    want = sr.knn_mean_dist2(pts)
    got = knn.distCUDA2(torch.as_tensor(pts).cuda()).cpu().numpy()
    assert np.allclose(got, want, rtol=1e-6, atol=0)
    assert (got != want).mean() < 1e-3, (got != want).mean()   # ties between the 3rd and 4th neighbour only
