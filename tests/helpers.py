"""Shared helpers for the parity tests: build one set of seeded inputs, run it through the CPU oracle
and through the HIP C-ABI path, and compare.

Tolerances (BASELINE.md section 3 / north_star): bit-exact on the integer tile/sort path; <= 1e-4
relative (with an absolute floor scaled to the tensor's max-abs) on the float image and gradients.
alpha >= 1/255 and T < 1e-4 are DISCONTINUITIES of the reference function itself: a 1-ulp difference
in exp() (libm vs v_exp_f32 -- CUDA's own expf differs from both) can flip a pixel-Gaussian pair in
or out.  Such flips are allowed for a tiny fraction of elements (FLIP_FRAC) and reported.
"""
from __future__ import annotations

import math
import os

import numpy as np

from oracle import saga_oracle as so
from seganygaussians_amd import scenes

RTOL = 1e-4
FLIP_FRAC = 2e-5      # fraction of elements allowed outside tolerance because of threshold flips


def make_inputs(P, W, H, C, seed=0, *, focal=None, log_scale=None, log_scale_std=0.6, with_shs=False, sh_degree=0,
                use_cov=False, use_mask=False, bg=None, scale_modifier=1.0, camera="front", z_range=(1.5, 12.0)):
    focal = focal or 0.85 * W
    log_scale = math.log(0.05) if log_scale is None else log_scale
    sc = scenes.make_scene(P, W, H, focal, C, log_scale, log_scale_std, seed=seed, with_shs=with_shs, z_range=z_range)
    if camera == "front":
        cam = scenes.look_at_camera(W, H, focal)
    elif camera == "orbit":
        cam = scenes.orbit_camera(W, H, focal, 0.2, 0.07)
    else:  # ("orbit", angle, tilt): e.g. the pose bench.py gives rank r of BASELINE config 4, (0.05 r, 0.02 r)
        cam = scenes.orbit_camera(W, H, focal, float(camera[1]), float(camera[2]))
    rng = np.random.default_rng(seed + 1000)
    if bg is None:
        bg = np.zeros(C, np.float32)
    elif isinstance(bg, str) and bg == "random":
        bg = rng.uniform(0, 1, C).astype(np.float32)
    cov = None
    if use_cov:
        # world covariance in fp64 from scale/rotation, handed over as cov3D_precomp
        q = sc.rotations.astype(np.float64)
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                      2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                      2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
        S = sc.scales.astype(np.float64)[:, None, :] * scale_modifier
        L = R * S
        Sig = L @ L.transpose(0, 2, 1)
        cov = np.stack([Sig[:, 0, 0], Sig[:, 0, 1], Sig[:, 0, 2], Sig[:, 1, 1], Sig[:, 1, 2], Sig[:, 2, 2]],
                       1).astype(np.float32)
    mask = rng.uniform(0, 1, P).astype(np.float32) if use_mask else None
    return so.Inputs(means3D=sc.means3D, opacities=sc.opacities, viewmatrix=cam.viewmatrix, projmatrix=cam.projmatrix,
                     campos=cam.campos, bg=bg, image_width=W, image_height=H, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                     channels=C, scale_modifier=scale_modifier, sh_degree=sh_degree,
                     shs=sc.shs if with_shs else None, colors_precomp=None if with_shs else sc.features,
                     scales=None if use_cov else sc.scales, rotations=None if use_cov else sc.rotations,
                     cov3D_precomp=cov, mask=mask)


def inputs_from_config(name, P=None, seed=0, with_shs=False, use_mask=False, camera="front"):
    c = scenes.CONFIGS[name]
    inp = make_inputs(c["P"] if P is None else P, c["W"], c["H"], c["C"], seed, focal=c["focal"],
                      log_scale=c["ls_mean"], log_scale_std=c["ls_std"], with_shs=with_shs,
                      sh_degree=3 if with_shs else 0, use_mask=use_mask, camera=camera)
    return inp


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF_LIB = os.path.join(ROOT, "seganygaussians_amd", "libmi_rast_prof.so")


def rerun_with_profiling_library(request) -> bool:
    """The comparison kernels of earlier rounds (MI_RAST_TILE_FWD, MI_RAST_F32_BLEND: tile-batched and f32-chain forwards) are
    compiled into the PROFILING build only (libmi_rast_prof.so, built by __graft_entry__.build(); the product library refuses
    the flags).  A test that compares the product kernels with them calls this first: in the normal run it re-runs itself in a
    subprocess with MI_RAST_LIB pointing at the profiling build (which holds the product kernels as well, compiled from the
    same sources) and returns False -- nothing left to do --; inside that subprocess it returns True."""
    import subprocess
    import sys
    if os.environ.get("MI_RAST_LIB") == PROF_LIB:
        return True
    assert os.path.exists(PROF_LIB), f"{PROF_LIB} missing: python -m seganygaussians_amd.build --profiling (or __graft_entry__.build())"
    out = subprocess.run([sys.executable, "-m", "pytest", request.node.nodeid, "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"],
                         env=dict(os.environ, MI_RAST_LIB=PROF_LIB), capture_output=True, text=True, timeout=1800, cwd=ROOT)
    assert out.returncode == 0, "under the profiling library:\n" + out.stdout[-4000:] + out.stderr[-2000:]
    return False


def rank_camera(rank):
    """The pose bench.py gives rank `rank` of BASELINE config 4 (8 orbit poses over the same Gaussians; rank 0 = front view)."""
    return "front" if rank == 0 else ("orbit", 0.05 * rank, 0.02 * rank)


# ---------------------------------------------------------------------------------------------------
# GPU (C-ABI) runner
# ---------------------------------------------------------------------------------------------------

class GpuRun:
    """Runs the HIP path through seganygaussians_amd.rasterizer's native entry points (the same functions
    the drop-in packages call) and keeps everything needed for comparisons."""

    def __init__(self, inp: so.Inputs, device="cuda:0"):
        import torch
        from seganygaussians_amd import rasterizer as R
        self.torch, self.R, self.inp, self.dev = torch, R, inp, torch.device(device)
        t = lambda a, shape=None: (torch.empty(0) if a is None else
                                   torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(self.dev))
        P = np.asarray(inp.means3D).reshape(-1, 3).shape[0]
        self.P = P
        self.means3D = t(np.asarray(inp.means3D).reshape(-1, 3))
        self.opac = t(np.asarray(inp.opacities).reshape(-1, 1))
        self.shs = t(None if inp.shs is None else np.asarray(inp.shs).reshape(P, -1, 3))
        self.colors = t(inp.colors_precomp)
        self.scales, self.rots, self.cov = t(inp.scales), t(inp.rotations), t(inp.cov3D_precomp)
        self.view, self.proj, self.campos, self.bg = t(inp.viewmatrix), t(inp.projmatrix), t(inp.campos), t(inp.bg)
        self.mask = t(inp.mask)
        self.with_mask = inp.mask is not None

    def forward(self, debug=False, full_lists=True, f32_blend=None, no_cull=None, fast_exp=None, verify_lists=None, tile_fwd=None,
                prezero=False, exact_exp=None):
        """full_lists=True materialises the reference's point_list / full-list positions (what the bit-exact
        comparisons with the oracle read); False is the product default ("lean" lists, include/mi_rast.h)."""
        i = self.inp
        with self.R.forward_flags(full_lists=bool(full_lists), f32_blend=f32_blend, no_cull=no_cull, fast_exp=fast_exp,
                                  verify_lists=verify_lists, tile_fwd=tile_fwd, exact_exp=exact_exp):
            res = self.R.rasterize_gaussians_native(
                i.channels, self.with_mask, self.bg, self.means3D, self.colors, self.opac, self.mask, self.scales,
                self.rots, i.scale_modifier, self.cov, self.view, self.proj, i.tanfovx, i.tanfovy, i.image_height,
                i.image_width, self.shs, i.sh_degree, self.campos, i.prefiltered, debug, prezero=prezero)
        self.full_lists = bool(full_lists or debug)
        if self.with_mask:
            (self.num_rendered, self.color, self.out_mask, self.out_depth, self.radii, self.geom, self.binning,
             self.img) = res
        else:
            self.num_rendered, self.color, self.radii, self.geom, self.binning, self.img = res
            self.out_mask = self.out_depth = None
        return self

    def backward(self, dL_dout_color, dL_dout_mask=None, debug=False, prezeroed=None, features_only=False):
        """features_only (include/mi_rast.h: MI_RAST_BWD_FEATURES_ONLY): returns {"dL_dcolors": ...} alone."""
        i, torch = self.inp, self.torch
        g = torch.as_tensor(np.ascontiguousarray(dL_dout_color, np.float32)).to(self.dev)
        gm = None
        if self.with_mask:
            gm = torch.as_tensor(np.ascontiguousarray(
                np.zeros((1, i.image_height, i.image_width), np.float32) if dL_dout_mask is None else dL_dout_mask,
                np.float32)).to(self.dev)
        res = self.R.rasterize_gaussians_backward_native(
            i.channels, self.with_mask, self.bg, self.means3D, self.radii, self.colors, self.scales, self.rots,
            i.scale_modifier, self.cov, self.view, self.proj, i.tanfovx, i.tanfovy, g, gm, self.shs, i.sh_degree,
            self.campos, self.geom, self.num_rendered, self.binning, self.img, debug, prezeroed=prezeroed, features_only=features_only)
        if features_only:
            assert all(v is None for k_, v in enumerate(res) if k_ != 1)
            return {"dL_dcolors": res[1].detach().cpu().numpy()}
        names = (["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmask", "dL_dmeans3D", "dL_dcov3D", "dL_dsh",
                  "dL_dscales", "dL_drotations"] if self.with_mask else
                 ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
                  "dL_drotations"])
        self.grads = {k: v.detach().cpu().numpy() for k, v in zip(names, res)}
        return self.grads

    # ---- views into the opaque buffers (private layout published by mi_rast_*_layout) ----
    def _view(self, buf, off, count, dtype):
        raw = buf.detach().cpu().numpy()
        nbytes = count * np.dtype(dtype).itemsize
        return raw[off:off + nbytes].view(dtype).copy()

    def geom_fields(self):
        from seganygaussians_amd import _lib
        _, off = _lib.geometry_layout(self.P)
        P = self.P
        return dict(depths=self._view(self.geom, off["depths"], P, np.float32),
                    means2D=self._view(self.geom, off["means2D"], 2 * P, np.float32),
                    conic_opacity=self._view(self.geom, off["conic_opacity"], 4 * P, np.float32),
                    cov3D=self._view(self.geom, off["cov3D"], 6 * P, np.float32),
                    rgb=self._view(self.geom, off["rgb"], 3 * P, np.float32),
                    clamped=self._view(self.geom, off["clamped"], 3 * P, np.uint8),
                    tiles_touched=self._view(self.geom, off["tiles_touched"], P, np.uint32),
                    depth_key=self._view(self.geom, off["depth_key"], P, np.uint32))

    def bin_fields(self):
        """point_list: the low 28 bits of the blend list -- with full lists the reference's sorted id list, bit for bit."""
        from seganygaussians_amd import _lib
        R = self.num_rendered
        _, off = _lib.binning_layout(R)
        bl = self._view(self.binning, off["blend_list"], R, np.uint32)
        return dict(point_list=bl & np.uint32(0x0FFFFFFF), blend_list=bl)

    def blend_lists(self):
        """The per-tile blend lists as the blend kernels walk them: concatenated Gaussian ids and quadrant masks in list order
        (entries without a quadrant bit -- the culled overlaps a full list carries -- left out: no kernel stops at them), plus the
        per-tile counts of what is left.  Positions depend on the list mode and are not returned."""
        R = self.num_rendered
        ranges = self.img_fields()["ranges"].reshape(-1, 2).astype(np.int64)
        if R == 0:
            return np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(len(ranges), np.int64)
        bl = self.bin_fields()["blend_list"]
        lens = ranges[:, 1] - ranges[:, 0]
        sel = np.concatenate([np.arange(a, b) for a, b in ranges if b > a]) if lens.sum() else np.zeros(0, np.int64)
        tiles = np.repeat(np.arange(len(ranges)), lens)
        e = bl[sel]
        keep = (e >> np.uint32(28)) != 0
        counts = np.bincount(tiles[keep], minlength=len(ranges)).astype(np.int64)
        return (e[keep] & np.uint32(0x0FFFFFFF)).copy(), (e[keep] >> np.uint32(28)).copy(), counts

    def sorted_keys(self):
        """The reference's sorted 64-bit key list (tile << 32 | depth bits), which our pipeline never
        materialises, reconstructed from ranges + point_list + depth bits."""
        im, pl = self.img_fields(), self.bin_fields()["point_list"]
        ranges = im["ranges"].reshape(-1, 2).astype(np.int64)
        counts = ranges[:, 1] - ranges[:, 0]
        tiles = np.repeat(np.arange(len(ranges), dtype=np.uint64), counts)
        order = np.argsort(np.repeat(ranges[:, 0], counts), kind="stable")   # list positions ascend with tile id
        assert np.array_equal(order, np.arange(len(order)))
        dbits = self.geom_fields()["depths"].view(np.uint32)[pl].astype(np.uint64)
        return (tiles << np.uint64(32)) | dbits

    def img_fields(self):
        from seganygaussians_amd import _lib
        i = self.inp
        W, H = i.image_width, i.image_height
        _, off = _lib.image_layout(W, H)
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        return dict(final_T=self._view(self.img, off["final_T"], W * H, np.float32),
                    n_contrib=self._view(self.img, off["n_contrib"], W * H, np.uint32),
                    ranges=self._view(self.img, off["ranges"], 2 * tiles, np.uint32),
                    tile_consumed=self._view(self.img, off["tile_consumed"], tiles, np.uint32),
                    tile_nsurv=self._view(self.img, off["tile_nsurv"], tiles, np.uint32))


# ---------------------------------------------------------------------------------------------------
# comparisons
# ---------------------------------------------------------------------------------------------------

def close_report(got, want, rtol=RTOL, floor=None):
    """|got-want| <= rtol*|want| + rtol*max|want| ; returns (fraction outside, max err, scale)."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    if want.size == 0:
        return 0.0, 0.0, 0.0
    scale = float(np.abs(want).max())
    tol = rtol * np.abs(want) + rtol * max(scale, 1e-30) if floor is None else rtol * np.abs(want) + floor
    err = np.abs(got - want)
    bad = ~(err <= tol)
    return float(bad.mean()), float(err.max()), scale


def assert_close(name, got, want, rtol=RTOL, flip_frac=0.0):
    frac, emax, scale = close_report(got, want, rtol)
    assert frac <= flip_frac, f"{name}: fraction outside tol {frac:.3e} (allowed {flip_frac:.1e}); max err {emax:.3e}, scale {scale:.3e}"
    return frac


def compare_integer_path(gpu: GpuRun, fwd: so.ForwardOut):
    """Bit-exact: radii, tiles_touched, num_rendered, sorted (key,value) list, tile ranges."""
    st = fwd.state
    np.testing.assert_array_equal(gpu.radii.cpu().numpy(), fwd.radii, err_msg="radii")
    assert gpu.num_rendered == fwd.num_rendered, (gpu.num_rendered, fwd.num_rendered)
    g = gpu.geom_fields()
    np.testing.assert_array_equal(g["tiles_touched"], st.field(so.F_TILES_TOUCHED), err_msg="tiles_touched")
    vis = fwd.radii > 0
    # depths / means2D feed the integer path (key bits, rects): bit-exact where the Gaussian is visible
    np.testing.assert_array_equal(g["depths"].view(np.uint32)[vis], st.field(so.F_DEPTHS).view(np.uint32)[vis],
                                  err_msg="depth bits")
    np.testing.assert_array_equal(g["means2D"].view(np.uint32).reshape(-1, 2)[vis],
                                  st.field(so.F_MEANS2D).view(np.uint32).reshape(-1, 2)[vis], err_msg="means2D bits")
    b = gpu.bin_fields()
    np.testing.assert_array_equal(b["point_list"], st.field(so.F_POINT_LIST), err_msg="point_list")
    im = gpu.img_fields()
    np.testing.assert_array_equal(im["ranges"], st.field(so.F_RANGES), err_msg="tile ranges")
    if gpu.num_rendered:
        np.testing.assert_array_equal(gpu.sorted_keys(), st.field(so.F_KEYS_SORTED), err_msg="sorted (tile|depth) keys")


def compare_lean_with_full(inp, full: GpuRun, dL=None, dLm=None, full_grads=None, fast_exp=None):
    """The product default ("lean" lists) against the full-list run of the same input: identical blend lists
    (ids, quadrant masks, order), bit-identical images / final_T / radii; gradients equal up to the order of the
    atomic float sums.  The lean run is made with MI_RAST_VERIFY_LISTS: every list slot the count pass reserved must have been
    written by the emit pass (the two passes take their float decisions in two template instances of one kernel)."""
    lean = GpuRun(inp).forward(full_lists=False, fast_exp=fast_exp, verify_lists=True)
    assert lean.num_rendered == full.num_rendered
    np.testing.assert_array_equal(lean.radii.cpu().numpy(), full.radii.cpu().numpy(), err_msg="radii (lean)")
    ids_l, qm_l, cnt_l = lean.blend_lists()
    ids_f, qm_f, cnt_f = full.blend_lists()
    np.testing.assert_array_equal(cnt_l, cnt_f, err_msg="blend list lengths (lean vs full)")
    np.testing.assert_array_equal(ids_l, ids_f, err_msg="blend list ids (lean vs full)")
    np.testing.assert_array_equal(qm_l, qm_f, err_msg="blend list quadrant masks (lean vs full)")
    np.testing.assert_array_equal(lean.color.cpu().numpy().view(np.uint32), full.color.cpu().numpy().view(np.uint32),
                                  err_msg="out_color bits (lean vs full)")
    np.testing.assert_array_equal(lean.img_fields()["final_T"].view(np.uint32), full.img_fields()["final_T"].view(np.uint32),
                                  err_msg="final_T bits (lean vs full)")
    if full.with_mask:
        np.testing.assert_array_equal(lean.out_mask.cpu().numpy(), full.out_mask.cpu().numpy(), err_msg="out_mask (lean)")
        np.testing.assert_array_equal(lean.out_depth.cpu().numpy(), full.out_depth.cpu().numpy(), err_msg="out_depth (lean)")
    if dL is not None:
        g = lean.backward(dL, dLm)
        for k, want in full_grads.items():
            # same sums in a different atomic order: float noise only (one element of a small tensor may land outside)
            assert_close(k + " (lean vs full)", g[k], want, rtol=2e-4, flip_frac=max(GRAD_FLIP_FRAC, 1.5 / max(1, np.asarray(want).size)))
    return lean


def compare_float_forward(gpu: GpuRun, fwd: so.ForwardOut, flip_frac=FLIP_FRAC, image_state=True):
    """image_state=False leaves final_T / n_contrib out: where a pixel STOPS (T < 1e-4) is decided on the v_exp_f32 alphas in the
    product's default exp mode (include/mi_rast.h: MI_RAST_EXACT_EXP), so those two fields are compared under exact_exp."""
    rep = {}
    rep["color"] = assert_close("out_color", gpu.color.cpu().numpy(), fwd.color, flip_frac=flip_frac)
    im = gpu.img_fields()
    if image_state:
        rep["final_T"] = assert_close("final_T", im["final_T"], fwd.state.field(so.F_FINAL_T), flip_frac=flip_frac)
        nc_g, nc_o = im["n_contrib"], fwd.state.field(so.F_N_CONTRIB)
        rep["n_contrib_mismatch"] = float((nc_g != nc_o).mean())
        assert rep["n_contrib_mismatch"] <= max(flip_frac, 1e-4), rep
    if gpu.with_mask:
        rep["mask"] = assert_close("out_mask", gpu.out_mask.cpu().numpy(), fwd.mask, flip_frac=flip_frac)
        rep["depth"] = assert_close("out_depth", gpu.out_depth.cpu().numpy(), fwd.depth, flip_frac=flip_frac)
    # conic/opacity, cov3D: float path of preprocess (should in fact be bit-identical: same op order)
    g = gpu.geom_fields()
    vis = fwd.radii > 0
    co_g = g["conic_opacity"].reshape(-1, 4)[vis]
    co_o = fwd.state.field(so.F_CONIC_OPACITY).reshape(-1, 4)[vis]
    rep["conic_bits_equal"] = bool(np.array_equal(co_g.view(np.uint32), co_o.view(np.uint32)))
    assert_close("conic_opacity", co_g, co_o)
    return rep


GRAD_FLIP_FRAC = 2e-4


def compare_gradients(grads: dict, bwd: so.BackwardOut, rtol=RTOL, flip_frac=GRAD_FLIP_FRAC, skip=()):
    rep = {}
    for k, got in grads.items():
        want = getattr(bwd, k)
        if want is None or k in skip:
            continue
        want = np.asarray(want).reshape(got.shape)
        rep[k] = assert_close(k, got, want, rtol=rtol, flip_frac=flip_frac)
    return rep


# ---------------------------------------------------------------------------------------------------
# stricter error statistics (the max-norm floor of assert_close lets small rows hide)
# ---------------------------------------------------------------------------------------------------

def norm_error(got, want) -> float:
    """||got - want||_2 / ||want||_2 in fp64."""
    got = np.asarray(got, np.float64).ravel()
    want = np.asarray(want, np.float64).ravel()
    d = float(np.linalg.norm(want))
    return float(np.linalg.norm(got - want)) / d if d > 0 else float(np.linalg.norm(got))


def row_error_report(got, want, rtol=RTOL):
    """Per-row relative error with a floor at the MEDIAN row magnitude (not the max): row r is bad when
    ||got_r - want_r||_inf > rtol * (||want_r||_inf + median_r' ||want_r'||_inf over non-zero rows).
    Returns (fraction of bad rows among non-zero rows, worst ratio err / allowed, median row magnitude,
    whether a row that is exactly zero in `want` is non-zero in `got`)."""
    want = np.asarray(want, np.float64)
    got = np.asarray(got, np.float64).reshape(want.shape)
    if want.ndim == 1:
        want, got = want[:, None], got[:, None]
    want = want.reshape(want.shape[0], -1)
    got = got.reshape(want.shape)
    mag = np.abs(want).max(axis=1)
    nz = mag > 0
    if not nz.any():
        return 0.0, 0.0, 0.0, bool((np.abs(got) != 0).any())
    med = float(np.median(mag[nz]))
    err = np.abs(got - want).max(axis=1)
    allowed = rtol * (mag + med)
    # rows the reference leaves at exactly zero must be exactly zero here too
    bad = (err > allowed) & nz
    zero_rows_touched = bool((err[~nz] != 0).any())
    ratio = float((err[nz] / allowed[nz]).max())
    return float(bad.sum() / nz.sum()), ratio, med, zero_rows_touched


def error_stats(got: dict, want, names=None) -> dict:
    """{name: {norm, row_frac, row_worst}} for every gradient tensor (want: BackwardOut or dict)."""
    out = {}
    for k, g in got.items():
        w = want[k] if isinstance(want, dict) else getattr(want, k)
        if w is None or g is None or (names and k not in names):
            continue
        w = np.asarray(w).reshape(np.asarray(g).shape)
        if w.size == 0:
            continue
        frac, worst, med, zt = row_error_report(g, w)
        out[k] = dict(norm=norm_error(g, w), norm_trim=trimmed_norm_error(g, w), row_frac=frac, row_worst=worst, zero_rows_touched=zt)
    return out


TRIM_ROWS = 32


def trimmed_norm_error(got, want, trim=TRIM_ROWS) -> float:
    """||got - want||_2 / ||want||_2 with the `trim` rows of largest squared error left out of the numerator: the norm-wise error
    of everything but a handful of rows.  The plain norm-wise error of dL_dcov3D / dL_dscales / dL_drotations at 1-5 M Gaussians is
    routinely ONE ill-conditioned Gaussian's (cfg5, round 4: a single row 243 tolerances off carried the product's 3.0e-5 while
    its rows-outside-tolerance share equalled the reference's) -- a statistic of which row drew the short straw, in either
    implementation.  The rows left out are still judged: by the rows-outside-tolerance share and by the untrimmed norm's own bound."""
    w = np.asarray(want, np.float64)
    w = w.reshape(w.shape[0], -1)
    g = np.asarray(got, np.float64).reshape(w.shape)
    e2 = ((g - w) ** 2).sum(axis=1)
    if e2.size > trim:
        e2 = np.partition(e2, e2.size - trim)[: e2.size - trim]
    den = float(np.sqrt((w ** 2).sum()))
    return float(np.sqrt(e2.sum())) / den if den > 0 else 0.0


def grads_as_dict(b) -> dict:
    return {k: v for k, v in b.__dict__.items() if v is not None} if not isinstance(b, dict) else b


def compare_forward_outs(a: so.ForwardOut, b: so.ForwardOut, what="", flip_frac=FLIP_FRAC):
    """Two ForwardOut bundles (CPU oracle and/or oracle/_ref runs): integer path bit-exact, float path <= 1e-4."""
    sa, sb = a.state, b.state
    np.testing.assert_array_equal(a.radii, b.radii, err_msg=what + " radii")
    assert a.num_rendered == b.num_rendered, (what, a.num_rendered, b.num_rendered)
    np.testing.assert_array_equal(sa.field(so.F_TILES_TOUCHED), sb.field(so.F_TILES_TOUCHED), err_msg=what + " tiles_touched")
    vis = a.radii > 0
    np.testing.assert_array_equal(sa.field(so.F_DEPTHS).view(np.uint32)[vis], sb.field(so.F_DEPTHS).view(np.uint32)[vis],
                                  err_msg=what + " depth bits")
    np.testing.assert_array_equal(sa.field(so.F_MEANS2D).view(np.uint32).reshape(-1, 2)[vis],
                                  sb.field(so.F_MEANS2D).view(np.uint32).reshape(-1, 2)[vis], err_msg=what + " means2D bits")
    np.testing.assert_array_equal(sa.field(so.F_KEYS_SORTED), sb.field(so.F_KEYS_SORTED), err_msg=what + " sorted keys")
    np.testing.assert_array_equal(sa.field(so.F_POINT_LIST), sb.field(so.F_POINT_LIST), err_msg=what + " point_list")
    T = len(sb.field(so.F_RANGES)) // 2
    np.testing.assert_array_equal(sa.field(so.F_RANGES)[:2 * T], sb.field(so.F_RANGES), err_msg=what + " ranges")
    rep = {}
    rep["color"] = assert_close(what + " out_color", a.color, b.color, flip_frac=flip_frac)
    rep["final_T"] = assert_close(what + " final_T", sa.field(so.F_FINAL_T), sb.field(so.F_FINAL_T), flip_frac=flip_frac)
    rep["n_contrib_mismatch"] = float((sa.field(so.F_N_CONTRIB) != sb.field(so.F_N_CONTRIB)).mean())
    assert rep["n_contrib_mismatch"] <= max(flip_frac, 1e-4), (what, rep)
    if a.mask is not None and b.mask is not None:
        rep["mask"] = assert_close(what + " out_mask", a.mask, b.mask, flip_frac=flip_frac)
    if a.depth is not None and b.depth is not None:
        rep["depth"] = assert_close(what + " out_depth", a.depth, b.depth, flip_frac=flip_frac)
    co_a = sa.field(so.F_CONIC_OPACITY).reshape(-1, 4)[vis]
    co_b = sb.field(so.F_CONIC_OPACITY).reshape(-1, 4)[vis]
    rep["conic_bits_equal"] = bool(np.array_equal(co_a.view(np.uint32), co_b.view(np.uint32)))
    assert_close(what + " conic_opacity", co_a, co_b)
    return rep
