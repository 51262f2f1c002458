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
