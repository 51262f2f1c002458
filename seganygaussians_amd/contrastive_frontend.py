"""Contrastive-loss front end without the x10 feature tensor.

The reference (train_contrastive_feature.py:237-253) resizes the rendered (C, h, w) feature image to the SAM-mask
resolution, repeats it once per sampled scale, multiplies by the scale gates -- a (N_scales, C, H, W) tensor,
2.6 GB at 10 x 32 x 1080p -- and only then keeps the ~1000 sampled rays:

    rendered  = F.interpolate(rendered[None], (H, W), mode='bilinear')[0]
    with_scale = rendered[None].repeat(N, 1, 1, 1) * gates[:, :, None, None]
    sampled   = with_scale[:, :, sampled_ray].permute(0, 2, 1)
    out       = F.normalize(sampled, dim=-1, p=2)

Every step is per pixel, so the rays can be taken FIRST: bilinear-sample the C channels at the S sampled pixels
(four taps each), scale by the gates, normalise.  Same values (same interpolation weights as
F.interpolate(..., align_corners=False)), same autograd graph semantics, O(S * C) memory instead of O(N * C * H * W).
The dense part of the loss -- the feature-norm regulariser on the un-resized render (:234-235) -- is untouched, so
the rasterizer backward still sees a dense gradient image."""
from __future__ import annotations

import torch


def _src_index(n_out: int, n_in: int, device, dtype):
    """Source coordinate, lower tap, upper tap and upper weight for every output index: the area_pixel_compute_*
    rule of bilinear F.interpolate with align_corners=False (scale = n_in / n_out, clamp at 0)."""
    scale = float(n_in) / float(n_out)
    dst = torch.arange(n_out, device=device, dtype=dtype)
    src = torch.clamp((dst + 0.5) * scale - 0.5, min=0.0)
    i0 = src.floor().long().clamp(max=n_in - 1)
    i1 = torch.clamp(i0 + 1, max=n_in - 1)
    lam = src - i0.to(dtype)
    return i0, i1, lam


def sample_scale_conditioned_features(rendered_features: torch.Tensor, out_hw, sampled_ray: torch.Tensor,
                                      gates: torch.Tensor) -> torch.Tensor:
    """rendered_features (C, h, w); out_hw = (H, W) of the SAM masks; sampled_ray bool (H, W); gates (N, C).
    Returns the (N, S, C) L2-normalised scale-conditioned features of train_contrastive_feature.py:237-253,
    S = sampled_ray.sum(), rays in row-major order (the order boolean-mask indexing produces)."""
    C, h, w = rendered_features.shape
    H, W = int(out_hw[0]), int(out_hw[1])
    if sampled_ray.shape != (H, W):
        raise ValueError(f"sampled_ray has shape {tuple(sampled_ray.shape)}, expected {(H, W)}")
    dev, dt = rendered_features.device, rendered_features.dtype
    ys, xs = torch.nonzero(sampled_ray, as_tuple=True)               # row-major == mask-indexing order
    y0, y1, ly = _src_index(H, h, dev, dt)
    x0, x1, lx = _src_index(W, w, dev, dt)
    y0, y1, ly = y0[ys], y1[ys], ly[ys]
    x0, x1, lx = x0[xs], x1[xs], lx[xs]
    f = rendered_features
    top = f[:, y0, x0] * (1 - lx) + f[:, y0, x1] * lx                # (C, S)
    bot = f[:, y1, x0] * (1 - lx) + f[:, y1, x1] * lx
    rays = (top * (1 - ly) + bot * ly).transpose(0, 1)               # (S, C)
    scaled = rays.unsqueeze(0) * gates.unsqueeze(1)                  # (N, S, C)
    return torch.nn.functional.normalize(scaled, dim=-1, p=2)


# ---------------------------------------------------------------------------------------------------------------------
# The HIP path (include/mi_contrastive.h, csrc/contrastive.h): regulariser + rays, forward and backward
# ---------------------------------------------------------------------------------------------------------------------

def _ptr(t):
    return None if t is None or t.numel() == 0 else t.data_ptr()


class _ContrastiveFrontEnd(torch.autograd.Function):
    """(rendered (C,h,w), gates (N,C)) -> (out (N,S,C), rendered_feature_norm ()) through mi_contrastive_forward / _backward."""

    @staticmethod
    def forward(ctx, rendered, gates, ray_yx, H, W):
        from . import _lib
        if not rendered.is_cuda:
            raise RuntimeError("contrastive_front_end needs GPU tensors: the front end has no CPU fallback "
                               "(sample_scale_conditioned_features is the PyTorch restatement the tests compare with)")
        L = _lib.load()
        ctx.in_dtypes = (rendered.dtype, gates.dtype)
        dev = rendered.device
        rendered = rendered.contiguous().float()
        # everything the kernels dereference lives on `rendered`'s device (contrastive_front_end moves what is elsewhere; a CPU
        # index tensor would hand the kernels a host pointer)
        if gates.device != dev or ray_yx.device != dev:
            raise RuntimeError(f"gates ({gates.device}) and ray coordinates ({ray_yx.device}) must be on the device of the "
                               f"rendered features ({dev})")
        gates_c = gates.contiguous().float()
        ray_yx = ray_yx.to(torch.int32).contiguous()
        C, h, w = rendered.shape
        N, S = gates_c.shape[0], ray_yx.shape[0]
        out = torch.empty((N, S, C), device=dev, dtype=torch.float32)
        ray_feat = torch.empty((S, C), device=dev, dtype=torch.float32)
        inv_len = torch.empty((N, S), device=dev, dtype=torch.float32)
        inv_norm = torch.empty((h * w,), device=dev, dtype=torch.float32)
        norm_sum = torch.zeros((64 * 16,), device=dev, dtype=torch.float64)   # MI_CONTRASTIVE_NORM_SLOTS partial sums, one per 128-byte line
        with torch.cuda.device(dev):   # the launch pairs the stream with the CURRENT device
            stream = torch.cuda.current_stream(dev).cuda_stream
            rc = L.mi_contrastive_forward(C, h, w, rendered.data_ptr(), int(H), int(W), S, _ptr(ray_yx), N, gates_c.data_ptr(),
                                          _ptr(out), _ptr(ray_feat), _ptr(inv_len), inv_norm.data_ptr(), norm_sum.data_ptr(), stream)
        if rc != 0:
            raise RuntimeError(_lib.last_error())
        ctx.save_for_backward(rendered, gates_c, ray_yx, out, ray_feat, inv_len, inv_norm)
        ctx.dims = (C, h, w, int(H), int(W), S, N)
        return out, (norm_sum.sum() / float(h * w)).float()

    @staticmethod
    def backward(ctx, d_out, d_norm):
        from . import _lib
        L = _lib.load()
        rendered, gates_c, ray_yx, out, ray_feat, inv_len, inv_norm = ctx.saved_tensors
        C, h, w, H, W, S, N = ctx.dims
        dev = rendered.device
        d_rendered = torch.empty_like(rendered)
        d_gates = torch.zeros_like(gates_c)
        d_out_c = None if d_out is None else d_out.contiguous().float()
        if d_out_c is None and S > 0:
            d_out_c = torch.zeros((N, S, C), device=dev, dtype=torch.float32)
        g = None if d_norm is None else d_norm.reshape(1).contiguous().float()
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            rc = L.mi_contrastive_backward(C, h, w, rendered.data_ptr(), H, W, S, _ptr(ray_yx), N, gates_c.data_ptr(), _ptr(out),
                                           _ptr(ray_feat), _ptr(inv_len), inv_norm.data_ptr(), _ptr(d_out_c), _ptr(g),
                                           d_rendered.data_ptr(), d_gates.data_ptr(), stream)
        if rc != 0:
            raise RuntimeError(_lib.last_error())
        # gradients in the dtype (and, for the gates, on the device) the inputs came in
        return d_rendered.to(ctx.in_dtypes[0]), d_gates.to(ctx.in_dtypes[1]), None, None, None


def contrastive_front_end(rendered_features: torch.Tensor, out_hw, sampled_ray: torch.Tensor, gates: torch.Tensor):
    """train_contrastive_feature.py:234-254 in one forward launch (and two backward launches) on the MI355X:

        rendered_feature_norm = rendered_features.norm(dim=0, p=2).mean()
        scale_conditioned     = normalize((interpolate(rendered_features, out_hw)[None] * gates[:, :, None, None])
                                          [:, :, sampled_ray].permute(0, 2, 1), dim=-1)

    rendered_features (C, h, w) fp32 GPU tensor; out_hw = (H, W) of the SAM masks; sampled_ray bool (H, W) -- or an int (S, 2)
    tensor of (y, x) ray coordinates in row-major order; gates (N, C).  Returns (scale_conditioned (N, S, C),
    rendered_feature_norm scalar); both differentiable w.r.t. rendered_features and gates."""
    H, W = int(out_hw[0]), int(out_hw[1])
    dev = rendered_features.device
    gates = gates.to(dev)               # (differentiable moves: the gradient returns to where the tensor came from)
    sampled_ray = sampled_ray.to(dev)
    if sampled_ray.dtype == torch.bool:
        if sampled_ray.shape != (H, W):
            raise ValueError(f"sampled_ray has shape {tuple(sampled_ray.shape)}, expected {(H, W)}")
        ray_yx = torch.nonzero(sampled_ray).to(torch.int32).contiguous()     # row-major == mask-indexing order
    else:
        ray_yx = sampled_ray.to(torch.int32).contiguous()
    return _ContrastiveFrontEnd.apply(rendered_features, gates, ray_yx, H, W)
