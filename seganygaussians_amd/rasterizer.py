"""Host-side mirror of the reference's Python extension API, on top of the C-ABI (include/mi_rast.h).

This file plays the role of BOTH the reference's torch glue (CF/rasterize_points.cu:35-216: tensor
allocation, shape check, P==0 short circuit, M = sh.size(1)) and its Python package
(CF/diff_gaussian_rasterization_contrastive_f/__init__.py: GaussianRasterizationSettings,
GaussianRasterizer, _RasterizeGaussians), parameterised by channel count and variant so that ONE
implementation serves the three reference packages:

    diff_gaussian_rasterization               C = 3                  (BASE/)
    diff_gaussian_rasterization_contrastive_f C = 32 (or 64)         (CF/)
    diff_gaussian_rasterization_depth         C = 3 + mask + depth   (DEPTH/)

Same names, argument order, return order and error messages as the reference.  PyTorch is used for
device memory, streams and autograd plumbing only; all compute is in libmi_rast.so.  There is no
CPU fallback: missing library or non-GPU tensors raise.
"""
from __future__ import annotations

import contextlib
import os
import ctypes as C
import threading
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _lib


def cpu_deep_copy_tuple(input_tuple):
    copied_tensors = [item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple]
    return tuple(copied_tensors)


class GaussianRasterizationSettings(NamedTuple):
    # field order == CF/diff_gaussian_rasterization_contrastive_f/__init__.py:156-168
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


# --------------------------------------------------------------------------------------------------
# C-ABI call helpers (the "_C" layer)
# --------------------------------------------------------------------------------------------------

def _dev_ptr(t, name, device=None, dtype=torch.float32):
    """Empty tensor -> NULL (reference convention, CF/.../__init__.py:196-206 + `!= nullptr` branches)."""
    if t is None or t.numel() == 0:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor (got {t.device}); the MI355X rasterizer has no CPU path")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype} (got {t.dtype})")
    if device is not None and t.device != device:
        raise RuntimeError(f"{name} is on {t.device}, expected {device}")
    return t.data_ptr()


def _contig(t):
    return t if (t is None or t.numel() == 0) else t.contiguous()


class _Resizer:
    """Replacement for resizeFunctional (CF/rasterize_points.cu:27-33): a growable torch uint8 buffer
    handed to the library as a C callback."""

    def __init__(self, device):
        self.device = device
        self.tensor = torch.empty(0, dtype=torch.uint8, device=device)
        self.cb = _lib.RESIZE_FN(self._resize)

    def _resize(self, nbytes, _user):
        try:
            self.tensor = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
            return self.tensor.data_ptr()
        except Exception:  # allocation failure -> NULL -> MI_RAST_ERR_ALLOC
            return None


def _check(rc):
    if rc != 0:
        raise RuntimeError(_lib.last_error())


def _stream_ptr(device):
    return torch.cuda.current_stream(device).cuda_stream


# Per-call options of the native forward (include/mi_rast.h: `flags`, `features_ready_event`).  The C library keeps no
# state between calls; what the reference API has no argument for is held HERE, per host thread, and handed over with
# each call.
class _CallOptions(threading.local):
    def __init__(self):
        self.flags = 0
        self.features_ready = None   # torch.cuda.Event, kept alive until the forward that consumes it has returned
        self.grad_mode = True        # torch.is_grad_enabled() of the CALLER (autograd switches it off inside Function.forward)


_opts = _CallOptions()


@contextlib.contextmanager
def forward_flags(full_lists=None, f32_blend=None, no_cull=None, fast_exp=None, verify_lists=None, tile_fwd=None, exact_exp=None,
                  equal_runs=None):
    """Within the block, forwards of this thread run with the given modes: `full_lists` materialises the reference's
    point_list / full-list positions (parity tests), `f32_blend` / `tile_fwd` select the f32 FMA-chain / tile-batched forwards for 32/64
    channels (comparison kernels of the PROFILING build only -- MI_RAST_LIB=libmi_rast_prof.so; the product library refuses them with
    MI_RAST_ERR_INVALID), `no_cull` switches the exact-conservative cull off (testing aid), `fast_exp` evaluates exp() as
    v_exp_f32(x * log2e) instead of the device library's expf the reference's kernels call (+2 % views/s, ~5 ulp: a few
    alpha >= 1/255 decisions differ from the reference's); `verify_lists` (debugging aid) checks that the count and emit passes of the
    lean lists agree slot by slot; `exact_exp` makes the forward blend call expf for every pair (product default: the hybrid form of
    csrc/common.h -- same decisions, alpha to 1e-6): alpha / T / n_contrib / final_T are then bit-identical to a build of the
    reference's kernels; `equal_runs` (A/B aid) keeps the blend kernels' XCD runs at equal tile counts instead of equal modelled work.  The flags of a
    forward are remembered with its buffers and handed to its backward."""
    prev = _opts.flags
    for bit, v in ((_lib.MI_RAST_FULL_LISTS, full_lists), (_lib.MI_RAST_F32_BLEND, f32_blend), (_lib.MI_RAST_NO_CULL, no_cull),
                   (_lib.MI_RAST_FAST_EXP, fast_exp), (_lib.MI_RAST_VERIFY_LISTS, verify_lists),
                   (_lib.MI_RAST_TILE_FWD, tile_fwd), (_lib.MI_RAST_EXACT_EXP, exact_exp), (_lib.MI_RAST_EQUAL_RUNS, equal_runs)):
        if v is not None:
            _opts.flags = (_opts.flags | bit) if v else (_opts.flags & ~bit)
    try:
        yield
    finally:
        _opts.flags = prev


# --------------------------------------------------------------------------------------------------
# frozen-geometry reuse (opt-in; an extension: the reference recomputes everything per call)
# --------------------------------------------------------------------------------------------------
class GeometryCache:
    """Per-camera cache of what the geometry-only stages of a forward produce (preprocess, binning, per-tile sort): the geometry
    buffer, the blend lists, the tile ranges + XCD run boundaries, radii and num_rendered.  A later forward of the SAME geometry
    from the SAME camera runs the blend stage alone (include/mi_rast.h: mi_rast_forward_reuse).  Meant for SAGA's contrastive
    feature training, which optimises the feature rows only (scene/gaussian_model_ff.py:154-162) and revisits each of ~200 cameras
    ~50 times (train_contrastive_feature.py:231).

    The key is CONTENT: 64-bit fingerprints (mi_rast_fingerprint) of means3D, opacities, scales, rotations, cov3D_precomp, shs (when
    they colour the Gaussians), view / projection matrix and camera position, next to the scalar settings.  The reference's
    renderer passes activation OUTPUTS (gaussian_renderer/__init__.py:337-348: pc.get_opacity, get_scaling, get_rotation), new
    tensors per call, so storage identity alone would never hit -- and a freed tensor's address can be handed to another one.  A
    fingerprint is memoised per tensor OBJECT (weak reference) and version counter: a parameter or camera tensor that is passed
    again unchanged costs nothing, any in-place change (`means3D.add_(...)`, an optimizer step on a geometry tensor) bumps
    `_version`, gets a new fingerprint and misses.  Tensors seen for the first time are fingerprinted by one kernel, behind which
    the calling thread waits for the stream.

    Bytes kept per view: the geometry buffer (139 bytes per Gaussian), 4 bytes per blend-list entry, 8 bytes per tile, radii (4 bytes
    per Gaussian).  Least recently used views are dropped beyond `max_bytes`.

    The cached buffers are SHARED between the forwards of a view: two backward passes of the same cached view must not run at the
    same time on different streams (they would share the packed-gradient scratch of the geometry buffer); one after the other --
    what a training loop does -- is fine.  `debug=True` forwards and the full-list / verify modes of the tests are never cached."""

    def __init__(self, max_bytes=64 << 30):
        import collections
        self.max_bytes = int(max_bytes)
        self.enabled = True
        self.entries = collections.OrderedDict()
        self.bytes = 0
        self.hits = self.misses = 0
        self._memo = {}          # id(tensor) -> (weakref, _version, fingerprint)
        self.lock = threading.Lock()

    def stats(self):
        n = self.hits + self.misses
        return {"hits": self.hits, "misses": self.misses, "hit_rate": (self.hits / n) if n else 0.0, "views": len(self.entries),
                "bytes_cached": self.bytes}

    def clear(self):
        with self.lock:
            self.entries.clear()
            self._memo.clear()
            self.bytes = 0
            self.hits = self.misses = 0

    def fingerprints(self, tensors, dev):
        """One 64-bit content fingerprint per tensor (None for an absent one); memoised per (tensor object, version)."""
        import weakref
        out = [None] * len(tensors)
        todo = []
        for k, t in enumerate(tensors):
            if t is None or t.numel() == 0:
                continue
            m = self._memo.get(id(t))
            if m is not None and m[0]() is t and m[1] == t._version:
                out[k] = m[2]
            else:
                todo.append(k)
        for g0 in range(0, len(todo), 8):
            grp = todo[g0:g0 + 8]
            n = len(grp)
            ptrs = (C.c_void_p * n)(*[tensors[k].data_ptr() for k in grp])
            sizes = (C.c_size_t * n)(*[tensors[k].numel() * tensors[k].element_size() for k in grp])
            res = (C.c_uint64 * n)()
            with torch.cuda.device(dev):
                _check(_lib.load().mi_rast_fingerprint(n, ptrs, sizes, res, _stream_ptr(dev)))
            for k, v in zip(grp, res):
                t = tensors[k]
                out[k] = (int(v), tuple(t.shape), str(t.dtype))
                self._memo[id(t)] = (weakref.ref(t), t._version, out[k])
        if len(self._memo) > 4096:   # forget tensors that are gone
            self._memo = {i: m for i, m in self._memo.items() if m[0]() is not None}
        return out

    def lookup(self, key):
        with self.lock:
            e = self.entries.get(key)
            if e is not None:
                self.entries.move_to_end(key)
                self.hits += 1
            else:
                self.misses += 1
            return e

    def insert(self, key, entry):
        with self.lock:
            old = self.entries.pop(key, None)
            if old is not None:
                self.bytes -= old["bytes"]
            self.entries[key] = entry
            self.bytes += entry["bytes"]
            while self.bytes > self.max_bytes and len(self.entries) > 1:
                _, dropped = self.entries.popitem(last=False)
                self.bytes -= dropped["bytes"]


_geometry_cache = None


def enable_geometry_cache(max_bytes=64 << 30):
    """Switches the frozen-geometry reuse on for every forward of this process (GeometryCache); returns the cache (`.stats()`).
    Also switched on by MI_RAST_GEOMETRY_CACHE=<GiB> (or 1: 64 GiB) in the environment, for unchanged reference scripts."""
    global _geometry_cache
    if _geometry_cache is None:
        _geometry_cache = GeometryCache(max_bytes)
    else:
        _geometry_cache.max_bytes = int(max_bytes)
    _geometry_cache.enabled = True
    return _geometry_cache


def disable_geometry_cache(drop=False):
    """Forwards recompute everything again; the cached views are kept for a later enable_geometry_cache() unless `drop`."""
    global _geometry_cache
    if _geometry_cache is not None:
        _geometry_cache.enabled = False
        if drop:
            _geometry_cache = None


def geometry_cache():
    return _geometry_cache


if os.environ.get("MI_RAST_GEOMETRY_CACHE", "") not in ("", "0"):
    try:
        _gib = float(os.environ["MI_RAST_GEOMETRY_CACHE"])
    except ValueError:
        _gib = 1.0
    enable_geometry_cache(int((64 if _gib == 1.0 else _gib) * (1 << 30)))


# ---- features-only backward (EXTENSION, include/mi_rast.h: MI_RAST_BWD_FEATURES_ONLY) ------------------------------------------
# SAGA's contrastive feature training optimises `_point_features` alone (scene/gaussian_model_ff.py:154-162), but every parameter of
# its model requires grad and the renderer creates `screenspace_points` with requires_grad=True (gaussian_renderer/__init__.py:308):
# autograd asks the rasterizer for all eight gradients, the reference computes them, and nobody reads seven of them.
#   * AUTOMATIC, always on: when autograd itself says that only colors_precomp needs a gradient (ctx.needs_input_grad), the backward
#     computes that one alone -- plain autograd semantics, nothing to opt into.
#   * OPT-IN, for unchanged reference scripts: enable_features_only_backward() / MI_RAST_FEATURES_ONLY_BACKWARD=1 makes every backward
#     of the feature rasterizer return dL_dcolors_precomp and None for the rest (means2D.grad, xyz.grad, ... stay unset).
# Either way the result equals the default backward's dL_dcolors up to the order of the atomic sums.  Never used by the headline
# benchmark: the reference's step computes all eight gradients, and so does bench.py's.
_features_only_backward = os.environ.get("MI_RAST_FEATURES_ONLY_BACKWARD", "") not in ("", "0")


def enable_features_only_backward(on=True):
    """Opt-in: backward passes through the rasterizer produce the gradient of colors_precomp only (see above); returns the previous
    setting.  Applies to calls with precomputed colours / features of a width that is a multiple of 16; others run the full backward."""
    global _features_only_backward
    prev, _features_only_backward = _features_only_backward, bool(on)
    return prev


def features_only_backward_enabled():
    return _features_only_backward


def _features_only_applies(channels, colors_precomp, needs_input_grad, debug):
    """needs_input_grad: (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, ...) of the Function."""
    if debug or colors_precomp is None or colors_precomp.numel() == 0 or not needs_input_grad[3]:
        return False
    if not _lib.load().mi_rast_features_only_supported(int(channels)):
        return False
    others = [needs_input_grad[k] for k in (0, 1, 2, 4, 5, 6, 7)]
    return _features_only_backward or not any(others)


_NOCACHE_FLAGS = _lib.MI_RAST_FULL_LISTS | _lib.MI_RAST_NO_CULL | _lib.MI_RAST_VERIFY_LISTS | _lib.MI_RAST_TILE_FWD | _lib.MI_RAST_F32_BLEND


def rasterize_gaussians_native(channels, with_mask_depth, background, means3D, colors, opacity, mask, scales,
                               rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy,
                               image_height, image_width, sh, degree, campos, prefiltered, debug, prezero=False):
    """RasterizeGaussiansCUDA (CF/rasterize_points.cu:35-115; DEPTH/rasterize_points.cu:35-130).

    prezero (a backward will follow: the autograd Functions set it when an input requires grad): the forward also leaves the
    backward's accumulators zero-filled -- a (P, channels) dL_dcolors tensor allocated here and the packed field gradients
    inside the geometry buffer -- stored by the blend kernel beside its own work (include/mi_rast.h: dL_dcolor_next,
    MI_RAST_PREZERO_BWD) instead of by two fill passes in front of the backward.  The tensor is left on the returned geometry
    buffer as `.mi_prezero` (with `.mi_pack_zeroed = True`); hand both to ONE rasterize_gaussians_backward_native call
    (`prezeroed=`, `pack_zeroed=`).  The dL_dcolors tensor is only made here while it is no larger than the image (P <= H W:
    the kernel takes the fill while it at most doubles its own stores; a larger one is cheapest as the backward's own
    torch.zeros, as before); prezero="always" makes it regardless (tests: the library then uses a fill command)."""
    ready = _opts.features_ready          # one-shot: consumed by THIS forward whatever happens below (P == 0, an exception)
    _opts.features_ready = None
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    _check_widths(channels, background, colors, means3D.size(0))
    L = _lib.load()
    P, H, W = means3D.size(0), int(image_height), int(image_width)
    dev = means3D.device
    if not means3D.is_cuda:
        raise RuntimeError("means3D must be a GPU tensor; the MI355X rasterizer has no CPU path")
    radii = torch.zeros(P, dtype=torch.int32, device=dev) if P == 0 else torch.empty(P, dtype=torch.int32, device=dev)
    geom, binning, img = _Resizer(dev), _Resizer(dev), _Resizer(dev)
    rendered = 0
    if P != 0:
        # the kernels write every element, so no zero-fill pass (reference: torch::full(0.0), :68)
        out_color = torch.empty((channels, H, W), dtype=torch.float32, device=dev)
        out_mask = torch.empty((1, H, W), dtype=torch.float32, device=dev) if with_mask_depth else None
        out_depth = torch.empty((1, H, W), dtype=torch.float32, device=dev) if with_mask_depth else None
        M = sh.size(1) if sh.numel() != 0 else 0
        t = [_contig(x) for x in (background, means3D, sh, colors, opacity, scales, rotations, cov3D_precomp,
                                   viewmatrix, projmatrix, campos, mask)]
        bg_c, m3_c, sh_c, col_c, op_c, sc_c, rot_c, cov_c, vm_c, pm_c, cp_c, mk_c = t
        n = C.c_int(0)
        grad_colors = None
        if prezero and (P <= H * W or prezero == "always"):
            grad_colors = torch.empty((P, channels), dtype=torch.float32, device=dev)
        if with_mask_depth and (mk_c is None or mk_c.numel() != P or not mk_c.is_cuda or mk_c.dtype != torch.float32):
            # the DEPTH package always passes a mask (DEPTH/.../__init__.py:323); without one out_mask / out_depth
            # would be left unwritten
            raise RuntimeError("mask must hold one float32 per Gaussian on the GPU (diff_gaussian_rasterization_depth)")
        # frozen-geometry reuse (opt-in, GeometryCache): same geometry + camera as an earlier forward -> the blend stage alone
        cache, ckey = _geometry_cache, None
        if cache is not None and cache.enabled and not debug and not (int(_opts.flags) & _NOCACHE_FLAGS) and not prefiltered:
            colours_from_sh = col_c is None or col_c.numel() == 0
            fps = cache.fingerprints([m3_c, op_c, sc_c, rot_c, cov_c, sh_c if colours_from_sh else None, vm_c, pm_c, cp_c], dev)
            ckey = (dev.index, P, H, W, float(tan_fovx), float(tan_fovy), float(scale_modifier), int(degree), int(M) if colours_from_sh else -1,
                    int(_opts.flags), tuple(fps))
            hit = cache.lookup(ckey)
            if hit is not None:
                img_t = torch.empty(hit["img_bytes"], dtype=torch.uint8, device=dev)
                with torch.cuda.device(dev):
                    rc = L.mi_rast_forward_reuse(
                        P, int(channels), int(hit["num_rendered"]), _dev_ptr(bg_c, "bg", dev), W, H, _dev_ptr(col_c, "colors_precomp", dev),
                        hit["geom"].data_ptr(), hit["blend_list"].data_ptr(), hit["ranges"].data_ptr(), hit["words"].data_ptr(),
                        img_t.data_ptr(), int(hit["longest_run"]), _dev_ptr(mk_c, "mask", dev) if with_mask_depth else None,
                        out_color.data_ptr(), out_mask.data_ptr() if with_mask_depth else None,
                        out_depth.data_ptr() if with_mask_depth else None, int(_opts.flags),
                        None if ready is None else C.c_void_p(ready.cuda_event),
                        None if grad_colors is None else grad_colors.data_ptr(), _stream_ptr(dev))
                del ready
                _check(rc)
                geom_t = hit["geom"].view(-1)        # a new tensor object over the shared storage: the notes below are per forward
                geom_t.mi_flags = int(_opts.flags)
                if grad_colors is not None:
                    geom_t.mi_prezero = grad_colors
                    geom_t.mi_pack_zeroed = False    # (the packed-gradient scratch is shared between the forwards of a cached view: its backward fills it)
                res = (hit["num_rendered"], out_color) + ((out_mask, out_depth) if with_mask_depth else ()) + \
                      (hit["radii"], geom_t, hit["blend_list"], img_t)
                return res
        with torch.cuda.device(dev):
            rc = L.mi_rast_forward(
                geom.cb, None, binning.cb, None, img.cb, None, P, int(degree), int(M), int(channels),
                _dev_ptr(bg_c, "bg", dev), W, H, _dev_ptr(m3_c, "means3D", dev), _dev_ptr(sh_c, "sh", dev),
                _dev_ptr(col_c, "colors_precomp", dev), _dev_ptr(op_c, "opacities", dev),
                _dev_ptr(sc_c, "scales", dev), float(scale_modifier), _dev_ptr(rot_c, "rotations", dev),
                _dev_ptr(cov_c, "cov3D_precomp", dev), _dev_ptr(vm_c, "viewmatrix", dev),
                _dev_ptr(pm_c, "projmatrix", dev), _dev_ptr(cp_c, "campos", dev), float(tan_fovx), float(tan_fovy),
                int(bool(prefiltered)), _dev_ptr(mk_c, "mask", dev) if with_mask_depth else None,
                out_color.data_ptr(), out_mask.data_ptr() if with_mask_depth else None,
                out_depth.data_ptr() if with_mask_depth else None, radii.data_ptr(), int(bool(debug)),
                int(_opts.flags) | (_lib.MI_RAST_PREZERO_BWD if prezero else 0),
                None if ready is None else C.c_void_p(ready.cuda_event),
                None if grad_colors is None else grad_colors.data_ptr(), _stream_ptr(dev), C.byref(n))
        del ready
        geom.tensor.mi_flags = int(_opts.flags)   # the backward re-takes the forward's decisions: it needs the same flags
        _check(rc)
        if prezero:
            geom.tensor.mi_pack_zeroed = True
            if grad_colors is not None:
                geom.tensor.mi_prezero = grad_colors
        rendered = n.value
        if ckey is not None:
            # first visit of this (geometry, camera): keep what the geometry-only stages produced.  The blend list (first field of
            # the binning buffer, include/mi_rast.h) is copied at its real length, tile ranges and the 16 words behind the R partial
            # sums likewise; the geometry buffer is kept as it is (this forward's backward shares it).
            _, ioff = _lib.image_layout(W, H)
            tiles = ((W + 15) // 16) * ((H + 15) // 16)
            words = img.tensor[ioff["num_rendered"] + 8192:ioff["num_rendered"] + 8192 + 64].clone()
            n_list = int(words.view(torch.int32)[0].item()) if rendered > 0 else 0   # entries the lists hold (lean: <= num_rendered)
            entry = {"geom": geom.tensor, "num_rendered": rendered, "radii": radii, "img_bytes": int(img.tensor.numel()),
                     "blend_list": binning.tensor[:max(4 * n_list, 4)].clone(), "ranges": img.tensor[ioff["ranges"]:ioff["ranges"] + 8 * tiles].clone(),
                     "words": words, "longest_run": int(L.mi_rast_last_longest_run())}
            entry["bytes"] = sum(int(entry[k].numel()) * entry[k].element_size() for k in ("geom", "blend_list", "ranges", "words", "radii"))
            cache.insert(ckey, entry)
    else:
        out_color = torch.zeros((channels, H, W), dtype=torch.float32, device=dev)
        out_mask = torch.zeros((1, H, W), dtype=torch.float32, device=dev) if with_mask_depth else None
        out_depth = torch.zeros((1, H, W), dtype=torch.float32, device=dev) if with_mask_depth else None
    if with_mask_depth:
        return rendered, out_color, out_mask, out_depth, radii, geom.tensor, binning.tensor, img.tensor
    return rendered, out_color, radii, geom.tensor, binning.tensor, img.tensor


def _check_widths(channels, background, colors, P):
    """The reference compiles NUM_CHANNELS into its kernels, which then index `bg_color[ch]` and `colors[id * C + ch]` for every
    ch < C without a check (forward.cu:343-374, backward.cu:446-535).  Here the width is a run-time argument (the contrastive_f
    drop-in reads it off `colors_precomp`), so a background left over from another width, or a colour tensor of another width
    than the call says, is caught instead of read out of bounds."""
    if background is None or background.numel() < channels:   # (a longer one is harmless: the kernels read bg[0 .. channels))
        raise RuntimeError(f"bg must hold one value per channel: {0 if background is None else background.numel()} given, "
                           f"{channels} channels rendered")
    if colors is not None and colors.numel() != 0 and P != 0 and (colors.ndimension() != 2 or colors.size(0) != P
                                                                  or colors.size(1) != channels):
        raise RuntimeError(f"colors_precomp must have dimensions (num_points, {channels}); got {tuple(colors.shape)}")


def set_features_ready_event(event) -> None:
    """The next forward of this host thread makes its stream wait for `event` (a recorded torch.cuda.Event, or None to
    cancel) right before its blend stage; everything before it -- preprocess, binning, per-tile sort -- only
    reads the geometry and runs ahead.  For training loops that optimise colors_precomp alone.  The event is passed to that
    one mi_rast_forward call as an argument (include/mi_rast.h: features_ready_event) and a reference is held until it
    returns; an error in that forward drops it as well."""
    _opts.features_ready = event


def _flags_of(geomBuffer, flags):
    """The MI_RAST_* flags a backward must run with are those of the forward that filled `geomBuffer`: given explicitly
    (the autograd Functions keep them in ctx), else the note the native forward left on its tensor, else none."""
    return flags if flags is not None else getattr(geomBuffer, "mi_flags", 0)


def rasterize_gaussians_backward_native(channels, with_mask_depth, background, means3D, radii, colors, scales,
                                        rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx,
                                        tan_fovy, dL_dout_color, dL_dout_mask, sh, degree, campos, geomBuffer, R,
                                        binningBuffer, imageBuffer, debug, flags=None, prezeroed=None, pack_zeroed=None,
                                        features_only=False):
    """RasterizeGaussiansBackwardCUDA (CF/rasterize_points.cu:117-196; DEPTH/rasterize_points.cu).

    prezeroed: the zero-filled (P, channels) tensor the forward of THESE buffers produced with prezero=True, not used by any
    backward before: it becomes dL_dcolors.  pack_zeroed: that forward also left the packed field gradients in its geometry
    buffer zeroed (None: as `prezeroed`): their fill is skipped.
    features_only (extension, include/mi_rast.h: MI_RAST_BWD_FEATURES_ONLY): dL_dcolors alone is computed and returned in its place
    of the tuple, None everywhere else."""
    L = _lib.load()
    P = means3D.size(0)
    H, W = dL_dout_color.size(1), dL_dout_color.size(2)
    M = sh.size(1) if sh.numel() != 0 else 0
    dev = means3D.device
    _check_widths(channels, background, colors, P)
    if dL_dout_color.ndimension() != 3 or dL_dout_color.size(0) != channels:
        raise RuntimeError(f"dL_dout_color must have dimensions ({channels}, H, W); got {tuple(dL_dout_color.shape)}")
    o = dict(device=dev, dtype=torch.float32)
    # The reference allocates ten zero tensors (rasterize_points.cu:151-159).  Same tensors here: dL_dcolors and dL_dsh,
    # which the kernels accumulate into, zero-filled; the others carved from ONE uninitialised block that
    # mi_rast_backward writes in full (zeros for Gaussians that were not rendered, include/mi_rast.h) -- 96 B per Gaussian
    # of fill traffic and nine fill launches less.  dL_dcolors -- the one gradient SAGA's feature training keeps
    # (`_point_features.grad`) -- has its own storage, so holding it does not pin the geometry gradients.
    # `debug` poisons the block with NaN first: a row the library failed to write would surface in the gradients.
    shapes = [("dL_dmeans3D", (P, 3)), ("dL_dmeans2D", (P, 3)), ("dL_dcolors", (P, channels)), ("dL_dconic", (P, 2, 2)),
              ("dL_dopacity", (P, 1)), ("dL_dcov3D", (P, 6)), ("dL_dsh", (P, M, 3)), ("dL_dscales", (P, 3)),
              ("dL_drotations", (P, 4))]
    if with_mask_depth:
        shapes.append(("dL_dmask", (P, 1)))   # DEPTH/rasterize_points.cu:167: torch::zeros({P, 1})
    shapes = [x for x in shapes if x[0] not in ("dL_dcolors", "dL_dsh")]
    if features_only:
        if with_mask_depth or colors.numel() == 0:
            raise RuntimeError("features_only backward needs precomputed colours and the plain (not DEPTH) rasterizer")
        if pack_zeroed is None:
            pack_zeroed = prezeroed is not None
        if prezeroed is not None and (tuple(prezeroed.shape) != (P, channels) or prezeroed.device != dev):
            prezeroed = None
        dL_dcolors = prezeroed if prezeroed is not None else torch.zeros((P, channels), **o)
        if P != 0:
            t = [_contig(x) for x in (background, means3D, colors, viewmatrix, projmatrix, campos, dL_dout_color, radii)]
            bg_c, m3_c, col_c, vm_c, pm_c, cp_c, dpix_c, radii_c = t
            with torch.cuda.device(dev):
                rc = L.mi_rast_backward(
                    P, int(degree), 0, int(channels), int(R), _dev_ptr(bg_c, "bg", dev), W, H,
                    _dev_ptr(m3_c, "means3D", dev), None, _dev_ptr(col_c, "colors_precomp", dev), None, float(scale_modifier), None,
                    None, _dev_ptr(vm_c, "viewmatrix", dev), _dev_ptr(pm_c, "projmatrix", dev), _dev_ptr(cp_c, "campos", dev),
                    float(tan_fovx), float(tan_fovy), _dev_ptr(radii_c, "radii", dev, torch.int32), geomBuffer.data_ptr(),
                    binningBuffer.data_ptr(), imageBuffer.data_ptr(), _dev_ptr(dpix_c, "dL_dout_color", dev), None,
                    None, None, None, dL_dcolors.data_ptr(), None, None, None, None, None, None, 0,
                    int(_flags_of(geomBuffer, flags)) | _lib.MI_RAST_BWD_FEATURES_ONLY | (_lib.MI_RAST_PREZERO_BWD if pack_zeroed else 0),
                    _stream_ptr(dev))
            _check(rc)
        return None, dL_dcolors, None, None, None, None, None, None
    sizes = []
    for _n, shp in shapes:
        n = 1
        for d in shp:
            n *= d
        sizes.append((n + 3) // 4 * 4)  # keep every tensor 16-byte aligned
    flat = torch.empty(sum(sizes), **o)
    if debug:
        flat.fill_(float("nan"))
    if pack_zeroed is None:
        pack_zeroed = prezeroed is not None
    if prezeroed is not None and (tuple(prezeroed.shape) != (P, channels) or prezeroed.device != dev or debug):
        prezeroed = None
    g = {"dL_dcolors": prezeroed if prezeroed is not None else torch.zeros((P, channels), **o), "dL_dsh": torch.zeros((P, M, 3), **o)}
    off = 0
    for (name, shp), n in zip(shapes, sizes):
        cnt = 1
        for d in shp:
            cnt *= d
        g[name] = flat[off:off + cnt].view(shp)
        off += n
    dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dconic = g["dL_dmeans3D"], g["dL_dmeans2D"], g["dL_dcolors"], g["dL_dconic"]
    dL_dopacity, dL_dcov3D, dL_dsh, dL_dscales = g["dL_dopacity"], g["dL_dcov3D"], g["dL_dsh"], g["dL_dscales"]
    dL_drotations = g["dL_drotations"]
    dL_dmask = g.get("dL_dmask")
    if P != 0:
        t = [_contig(x) for x in (background, means3D, sh, colors, scales, rotations, cov3D_precomp, viewmatrix,
                                   projmatrix, campos, dL_dout_color, dL_dout_mask, radii)]
        bg_c, m3_c, sh_c, col_c, sc_c, rot_c, cov_c, vm_c, pm_c, cp_c, dpix_c, dmask_c, radii_c = t
        with torch.cuda.device(dev):
            rc = L.mi_rast_backward(
                P, int(degree), int(M), int(channels), int(R), _dev_ptr(bg_c, "bg", dev), W, H,
                _dev_ptr(m3_c, "means3D", dev), _dev_ptr(sh_c, "sh", dev), _dev_ptr(col_c, "colors_precomp", dev),
                _dev_ptr(sc_c, "scales", dev), float(scale_modifier), _dev_ptr(rot_c, "rotations", dev),
                _dev_ptr(cov_c, "cov3D_precomp", dev), _dev_ptr(vm_c, "viewmatrix", dev),
                _dev_ptr(pm_c, "projmatrix", dev), _dev_ptr(cp_c, "campos", dev), float(tan_fovx), float(tan_fovy),
                _dev_ptr(radii_c, "radii", dev, torch.int32), geomBuffer.data_ptr(), binningBuffer.data_ptr(),
                imageBuffer.data_ptr(), _dev_ptr(dpix_c, "dL_dout_color", dev),
                _dev_ptr(dmask_c, "dL_dout_mask", dev) if with_mask_depth else None,
                dL_dmeans2D.data_ptr(), dL_dconic.data_ptr(), dL_dopacity.data_ptr(), dL_dcolors.data_ptr(),
                dL_dmask.data_ptr() if with_mask_depth else None, dL_dmeans3D.data_ptr(), dL_dcov3D.data_ptr(),
                dL_dsh.data_ptr() if M > 0 else None, dL_dscales.data_ptr(), dL_drotations.data_ptr(),
                int(bool(debug)), int(_flags_of(geomBuffer, flags)) | (_lib.MI_RAST_PREZERO_BWD if pack_zeroed else 0),
                _stream_ptr(dev))
        _check(rc)
    if with_mask_depth:
        return (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmask, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales,
                dL_drotations)
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


def mark_visible_native(means3D, viewmatrix, projmatrix):
    """markVisible (CF/rasterize_points.cu:198-216)."""
    L = _lib.load()
    P = means3D.size(0)
    dev = means3D.device
    present = torch.zeros(P, dtype=torch.bool, device=dev)
    if P != 0:
        m3_c, vm_c, pm_c = _contig(means3D), _contig(viewmatrix), _contig(projmatrix)
        with torch.cuda.device(dev):
            rc = L.mi_rast_mark_visible(P, _dev_ptr(m3_c, "means3D", dev), _dev_ptr(vm_c, "viewmatrix", dev),
                                        _dev_ptr(pm_c, "projmatrix", dev), present.data_ptr(), _stream_ptr(dev))
        _check(rc)
    return present


def rasterize_mask_gaussians_native(means3D, opacity, mask, scales, rotations, scale_modifier, cov3D_precomp,
                                    viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width,
                                    prefiltered, debug):
    """RasterizeMaskGaussiansCUDA (DEPTH/rasterize_points.cu, mask-only forward)."""
    _opts.features_ready = None           # a features-ready event is for the next FEATURE forward of this thread: never left armed
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    L = _lib.load()
    P, H, W = means3D.size(0), int(image_height), int(image_width)
    dev = means3D.device
    radii = torch.zeros(P, dtype=torch.int32, device=dev)
    geom, binning, img = _Resizer(dev), _Resizer(dev), _Resizer(dev)
    rendered = 0
    if P != 0:
        out_mask = torch.empty((1, H, W), dtype=torch.float32, device=dev)
        t = [_contig(x) for x in (means3D, opacity, mask, scales, rotations, cov3D_precomp, viewmatrix, projmatrix)]
        m3_c, op_c, mk_c, sc_c, rot_c, cov_c, vm_c, pm_c = t
        n = C.c_int(0)
        with torch.cuda.device(dev):
            rc = L.mi_rast_mask_forward(
                geom.cb, None, binning.cb, None, img.cb, None, P, W, H, _dev_ptr(m3_c, "means3D", dev),
                _dev_ptr(op_c, "opacities", dev), _dev_ptr(mk_c, "mask", dev), _dev_ptr(sc_c, "scales", dev),
                float(scale_modifier), _dev_ptr(rot_c, "rotations", dev), _dev_ptr(cov_c, "cov3D_precomp", dev),
                _dev_ptr(vm_c, "viewmatrix", dev), _dev_ptr(pm_c, "projmatrix", dev), float(tan_fovx),
                float(tan_fovy), int(bool(prefiltered)), out_mask.data_ptr(), radii.data_ptr(), int(bool(debug)),
                int(_opts.flags), _stream_ptr(dev), C.byref(n))
        _check(rc)
        rendered = n.value
        geom.tensor.mi_flags = int(_opts.flags)
    else:
        out_mask = torch.zeros((1, H, W), dtype=torch.float32, device=dev)
    return rendered, out_mask, radii, geom.tensor, binning.tensor, img.tensor


def rasterize_mask_gaussians_backward_native(means3D, dL_dout_mask, geomBuffer, R, binningBuffer, imageBuffer, debug, flags=None):
    L = _lib.load()
    P = means3D.size(0)
    H, W = dL_dout_mask.size(-2), dL_dout_mask.size(-1)
    dev = means3D.device
    dL_dmask = torch.zeros((P, 1), dtype=torch.float32, device=dev)   # DEPTH/rasterize_points.cu:349
    if P != 0:
        d_c = _contig(dL_dout_mask)
        with torch.cuda.device(dev):
            rc = L.mi_rast_mask_backward(P, int(R), W, H, geomBuffer.data_ptr(), binningBuffer.data_ptr(),
                                         imageBuffer.data_ptr(), _dev_ptr(d_c, "dL_dout_mask", dev),
                                         dL_dmask.data_ptr(), int(bool(debug)), int(_flags_of(geomBuffer, flags)),
                                         _stream_ptr(dev))
        _check(rc)
    return dL_dmask


# --------------------------------------------------------------------------------------------------
# autograd + nn.Module layer, generated per (channels, variant)
# --------------------------------------------------------------------------------------------------

_MSG_COLORS = 'Please provide excatly one of either SHs or precomputed colors!'
_MSG_COV = 'Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!'


def _make_plain(channels):
    """BASE / CF packages: CF/diff_gaussian_rasterization_contrastive_f/__init__.py:21-220."""

    class _RasterizeGaussians(torch.autograd.Function):
        @staticmethod
        def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                    raster_settings):
            rs = raster_settings
            args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                    rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh,
                    rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)

            # a backward will follow: its accumulators are zero-filled by the forward's blend kernel (rasterize_gaussians_native)
            # (needs_input_grad is requires_grad of the inputs whatever the grad mode: render.py's no_grad forwards of a model whose
            # parameters require grad would zero 128 MB per view for a backward that never comes)
            prezero = (_opts.grad_mode and bool(any(ctx.needs_input_grad)) and not rs.debug
                       and not os.environ.get("MI_RAST_NO_PREZERO"))   # (env: A/B aid)

            def call():
                (bg, m3, col, op, sc, rot, smod, cov, vm, pm, tx, ty, ih, iw, sh_, deg, cp, pre, dbg) = args
                return rasterize_gaussians_native(channels, False, bg, m3, col, op, None, sc, rot, smod, cov, vm, pm,
                                                  tx, ty, ih, iw, sh_, deg, cp, pre, dbg, prezero=prezero)

            if rs.debug:
                cpu_args = cpu_deep_copy_tuple(args)  # Copy them before they can be corrupted
                try:
                    num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = call()
                except Exception as ex:
                    torch.save(cpu_args, "snapshot_fw.dump")
                    print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                    raise ex
            else:
                num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = call()
            ctx.raster_settings = rs
            ctx.num_rendered = num_rendered
            ctx.mi_flags = getattr(geomBuffer, "mi_flags", 0)
            ctx.mi_prezero = geomBuffer.__dict__.pop("mi_prezero", None)   # one-shot: the first backward takes them
            ctx.mi_pack_zeroed = bool(geomBuffer.__dict__.pop("mi_pack_zeroed", False))
            ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                                  binningBuffer, imgBuffer)
            ctx.mark_non_differentiable(radii)
            # autograd would otherwise hand backward a zero-filled int32 (P,) "gradient" for radii on every call (a 4-MB fill)
            ctx.set_materialize_grads(False)
            ctx.out_shape = tuple(color.shape)
            return color, radii

        @staticmethod
        def backward(ctx, grad_out_color, _):
            num_rendered = ctx.num_rendered
            rs = ctx.raster_settings
            if grad_out_color is None:   # the image was not used: what autograd would have materialised
                grad_out_color = torch.zeros(ctx.out_shape, dtype=torch.float32, device=ctx.saved_tensors[1].device)
            (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer,
             imgBuffer) = ctx.saved_tensors
            args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                    rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, sh, rs.sh_degree, rs.campos,
                    geomBuffer, num_rendered, binningBuffer, imgBuffer, rs.debug)

            prezeroed, ctx.mi_prezero = ctx.mi_prezero, None   # (a second backward through a retained graph fills for itself)
            pack_zeroed, ctx.mi_pack_zeroed = ctx.mi_pack_zeroed, False

            feat_only = _features_only_applies(channels, colors_precomp, ctx.needs_input_grad, rs.debug)

            def call():
                (bg, m3, rad, col, sc, rot, smod, cov, vm, pm, tx, ty, gout, sh_, deg, cp, gb, nr, bb, ib, dbg) = args
                return rasterize_gaussians_backward_native(channels, False, bg, m3, rad, col, sc, rot, smod, cov, vm,
                                                           pm, tx, ty, gout, None, sh_, deg, cp, gb, nr, bb, ib, dbg,
                                                           flags=ctx.mi_flags, prezeroed=prezeroed, pack_zeroed=pack_zeroed,
                                                           features_only=feat_only)

            if rs.debug:
                cpu_args = cpu_deep_copy_tuple(args)
                try:
                    res = call()
                except Exception as ex:
                    torch.save(cpu_args, "snapshot_bw.dump")
                    print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                    raise ex
            else:
                res = call()
            (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh,
             grad_scales, grad_rotations) = res
            return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales,
                    grad_rotations, grad_cov3Ds_precomp, None)

    def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                            raster_settings):
        # grad mode as the CALLER sees it (inside Function.forward it is always off): decides whether the forward leaves the backward's
        # buffers zeroed (prezero).  Every entry point that reaches .apply must set it; it is put back afterwards so that a caller
        # reaching .apply by another path never sees the value of somebody else's call (the backward fills for itself when in doubt).
        _opts.grad_mode = torch.is_grad_enabled()
        try:
            return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                             cov3Ds_precomp, raster_settings)
        finally:
            _opts.grad_mode = True

    class GaussianRasterizer(nn.Module):
        def __init__(self, raster_settings):
            super().__init__()
            self.raster_settings = raster_settings

        def markVisible(self, positions):
            with torch.no_grad():
                rs = self.raster_settings
                visible = mark_visible_native(positions, rs.viewmatrix, rs.projmatrix)
            return visible

        def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                    cov3D_precomp=None):
            raster_settings = self.raster_settings
            if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
                raise Exception(_MSG_COLORS)
            if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                    ((scales is not None or rotations is not None) and cov3D_precomp is not None):
                raise Exception(_MSG_COV)
            if shs is None:
                shs = torch.Tensor([])
            if colors_precomp is None:
                colors_precomp = torch.Tensor([])
            if scales is None:
                scales = torch.Tensor([])
            if rotations is None:
                rotations = torch.Tensor([])
            if cov3D_precomp is None:
                cov3D_precomp = torch.Tensor([])
            return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                       cov3D_precomp, raster_settings)

    return _RasterizeGaussians, rasterize_gaussians, GaussianRasterizer


def _make_depth():
    """DEPTH package: DEPTH/diff_gaussian_rasterization_depth/__init__.py:21-391."""
    channels = 3

    class _RasterizeGaussians(torch.autograd.Function):
        @staticmethod
        def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, mask, scales, rotations, cov3Ds_precomp,
                    raster_settings):
            rs = raster_settings
            args = (rs.bg, means3D, colors_precomp, opacities, mask, scales, rotations, rs.scale_modifier,
                    cov3Ds_precomp, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height,
                    rs.image_width, sh, rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)

            def call():
                (bg, m3, col, op, mk, sc, rot, smod, cov, vm, pm, tx, ty, ih, iw, sh_, deg, cp, pre, dbg) = args
                return rasterize_gaussians_native(channels, True, bg, m3, col, op, mk, sc, rot, smod, cov, vm, pm, tx,
                                                  ty, ih, iw, sh_, deg, cp, pre, dbg)

            if rs.debug:
                cpu_args = cpu_deep_copy_tuple(args)
                try:
                    res = call()
                except Exception as ex:
                    torch.save(cpu_args, "snapshot_fw.dump")
                    print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                    raise ex
            else:
                res = call()
            num_rendered, color, out_mask, depth, radii, geomBuffer, binningBuffer, imgBuffer = res
            ctx.mask_shape = mask.shape
            ctx.raster_settings = rs
            ctx.num_rendered = num_rendered
            ctx.mi_flags = getattr(geomBuffer, "mi_flags", 0)
            ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                                  binningBuffer, imgBuffer)
            ctx.mark_non_differentiable(radii)
            ctx.set_materialize_grads(False)   # see the plain variant: no zero-filled "gradients" for radii / unused outputs
            ctx.out_shapes = (tuple(color.shape), tuple(out_mask.shape))
            return color, out_mask, depth, radii

        @staticmethod
        def backward(ctx, grad_out_color, grad_out_mask, grad_out_depth, _):
            num_rendered = ctx.num_rendered
            rs = ctx.raster_settings
            dev_ = ctx.saved_tensors[1].device
            if grad_out_color is None:
                grad_out_color = torch.zeros(ctx.out_shapes[0], dtype=torch.float32, device=dev_)
            if grad_out_mask is None:
                grad_out_mask = torch.zeros(ctx.out_shapes[1], dtype=torch.float32, device=dev_)
            (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer,
             imgBuffer) = ctx.saved_tensors
            args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                    rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, grad_out_mask, sh,
                    rs.sh_degree, rs.campos, geomBuffer, num_rendered, binningBuffer, imgBuffer, rs.debug)

            def call():
                (bg, m3, rad, col, sc, rot, smod, cov, vm, pm, tx, ty, gout, gmask, sh_, deg, cp, gb, nr, bb, ib,
                 dbg) = args
                return rasterize_gaussians_backward_native(channels, True, bg, m3, rad, col, sc, rot, smod, cov, vm,
                                                           pm, tx, ty, gout, gmask, sh_, deg, cp, gb, nr, bb, ib, dbg,
                                                           flags=ctx.mi_flags)

            if rs.debug:
                cpu_args = cpu_deep_copy_tuple(args)
                try:
                    res = call()
                except Exception as ex:
                    torch.save(cpu_args, "snapshot_bw.dump")
                    print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                    raise ex
            else:
                res = call()
            (grad_means2D, grad_colors_precomp, grad_opacities, grad_mask, grad_means3D, grad_cov3Ds_precomp, grad_sh,
             grad_scales, grad_rotations) = res
            # the reference hands back (P,1) (DEPTH/rasterize_points.cu:167), which autograd accepts only for a (P,1)
            # mask; same P values here, shaped like the mask that came in, so that (P,) works as well
            grad_mask = grad_mask.view(ctx.mask_shape)
            return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_mask, grad_scales,
                    grad_rotations, grad_cov3Ds_precomp, None)

    class _RasterizeMaskGaussians(torch.autograd.Function):
        # DEPTH/diff_gaussian_rasterization_depth/__init__.py:185-292
        @staticmethod
        def forward(ctx, means3D, means2D, opacities, mask, scales, rotations, cov3Ds_precomp, raster_settings):
            rs = raster_settings
            num_rendered, out_mask, radii, geomBuffer, binningBuffer, imgBuffer = rasterize_mask_gaussians_native(
                means3D, opacities, mask, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
                rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, rs.prefiltered, rs.debug)
            ctx.raster_settings = rs
            ctx.num_rendered = num_rendered
            ctx.mask_shape = mask.shape
            ctx.mi_flags = getattr(geomBuffer, "mi_flags", 0)
            ctx.save_for_backward(means3D, means2D, opacities, scales, rotations, cov3Ds_precomp, radii, geomBuffer,
                                  binningBuffer, imgBuffer)
            ctx.mark_non_differentiable(radii)
            ctx.set_materialize_grads(False)
            ctx.out_shape = tuple(out_mask.shape)
            return out_mask, radii

        @staticmethod
        def backward(ctx, grad_out_mask, _):
            rs = ctx.raster_settings
            if grad_out_mask is None:
                grad_out_mask = torch.zeros(ctx.out_shape, dtype=torch.float32, device=ctx.saved_tensors[0].device)
            (means3D, means2D, opacities, scales, rotations, cov3Ds_precomp, radii, geomBuffer, binningBuffer,
             imgBuffer) = ctx.saved_tensors
            grad_mask = rasterize_mask_gaussians_backward_native(means3D, grad_out_mask, geomBuffer, ctx.num_rendered,
                                                                 binningBuffer, imgBuffer, rs.debug,
                                                                 flags=ctx.mi_flags).view(ctx.mask_shape)
            # only the mask receives a real gradient; the reference hands ZEROS (not None) to every other input
            # (DEPTH/.../__init__.py:278-290) -- kept, but shaped like the inputs so autograd accepts them.
            z = [torch.zeros_like(t) if need else None for t, need in
                 zip((means3D, means2D, opacities), ctx.needs_input_grad[0:3])]
            z2 = [torch.zeros_like(t) if (need and t.numel()) else None for t, need in
                  zip((scales, rotations, cov3Ds_precomp), ctx.needs_input_grad[4:7])]
            return z[0], z[1], z[2], grad_mask, z2[0], z2[1], z2[2], None

    def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, mask, scales, rotations, cov3Ds_precomp,
                            raster_settings):
        return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, mask, scales, rotations,
                                         cov3Ds_precomp, raster_settings)

    def rasterize_mask_gaussians(means3D, means2D, opacities, mask, scales, rotations, cov3Ds_precomp,
                                 raster_settings):
        return _RasterizeMaskGaussians.apply(means3D, means2D, opacities, mask, scales, rotations, cov3Ds_precomp,
                                             raster_settings)

    class GaussianRasterizer(nn.Module):
        def __init__(self, raster_settings):
            super().__init__()
            self.raster_settings = raster_settings

        def markVisible(self, positions):
            with torch.no_grad():
                rs = self.raster_settings
                visible = mark_visible_native(positions, rs.viewmatrix, rs.projmatrix)
            return visible

        def forward(self, means3D, means2D, opacities, mask, shs=None, colors_precomp=None, scales=None,
                    rotations=None, cov3D_precomp=None):
            raster_settings = self.raster_settings
            if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
                raise Exception(_MSG_COLORS)
            if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                    ((scales is not None or rotations is not None) and cov3D_precomp is not None):
                raise Exception(_MSG_COV)
            if shs is None:
                shs = torch.Tensor([])
            if colors_precomp is None:
                colors_precomp = torch.Tensor([])
            if scales is None:
                scales = torch.Tensor([])
            if rotations is None:
                rotations = torch.Tensor([])
            if cov3D_precomp is None:
                cov3D_precomp = torch.Tensor([])
            return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, mask, scales, rotations,
                                       cov3D_precomp, raster_settings)

        def forward_mask(self, means3D, means2D, opacities, mask, scales=None, rotations=None, cov3D_precomp=None):
            raster_settings = self.raster_settings
            if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                    ((scales is not None or rotations is not None) and cov3D_precomp is not None):
                raise Exception(_MSG_COV)
            if scales is None:
                scales = torch.Tensor([])
            if rotations is None:
                rotations = torch.Tensor([])
            if cov3D_precomp is None:
                cov3D_precomp = torch.Tensor([])
            return rasterize_mask_gaussians(means3D, means2D, opacities, mask, scales, rotations, cov3D_precomp,
                                            raster_settings)

    return _RasterizeGaussians, _RasterizeMaskGaussians, rasterize_gaussians, rasterize_mask_gaussians, \
        GaussianRasterizer


_PLAIN_CACHE = {}


def make_auto_rasterizer(default_channels: int = 32):
    """The feature rasterizer with the channel count taken from the call: NUM_CHANNELS is a compile-time constant of the
    reference's package (CF/cuda_rasterizer/config_contrastive_f.h:15 = 32; a user who trains 64-D features edits the header and
    rebuilds), here a run-time argument of the C-ABI -- so the drop-in reads it off `colors_precomp` (or the background
    colour when SH colours are given), falling back to `default_channels`.  Returns (rasterize_gaussians, GaussianRasterizer)
    with the reference's signatures."""

    def _channels(colors_precomp, raster_settings):
        if colors_precomp is not None and torch.is_tensor(colors_precomp) and colors_precomp.dim() == 2 and colors_precomp.size(1) > 0:
            return int(colors_precomp.size(1))
        bg = getattr(raster_settings, "bg", None)
        if torch.is_tensor(bg) and bg.numel() > 0:
            return int(bg.numel())
        return int(default_channels)

    def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
        return make_rasterizer(_channels(colors_precomp, raster_settings))[1](means3D, means2D, sh, colors_precomp, opacities, scales,
                                                                             rotations, cov3Ds_precomp, raster_settings)

    class GaussianRasterizer(nn.Module):
        def __init__(self, raster_settings):
            super().__init__()
            self.raster_settings = raster_settings

        def markVisible(self, positions):
            with torch.no_grad():
                rs = self.raster_settings
                return mark_visible_native(positions, rs.viewmatrix, rs.projmatrix)

        def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
            impl = make_rasterizer(_channels(colors_precomp, self.raster_settings))[2](self.raster_settings)
            return impl(means3D=means3D, means2D=means2D, opacities=opacities, shs=shs, colors_precomp=colors_precomp, scales=scales,
                        rotations=rotations, cov3D_precomp=cov3D_precomp)

    return rasterize_gaussians, GaussianRasterizer


def make_rasterizer(channels: int):
    """(autograd Function, functional wrapper, nn.Module) for the plain C-channel rasterizer."""
    if channels not in _PLAIN_CACHE:
        _PLAIN_CACHE[channels] = _make_plain(channels)
    return _PLAIN_CACHE[channels]


_DEPTH = None


def make_depth_rasterizer():
    global _DEPTH
    if _DEPTH is None:
        _DEPTH = _make_depth()
    return _DEPTH
