"""Fused KNN feature smoothing: host side of include/mi_knn_smooth.h.

Mirrors FeatureGaussianModel.get_smoothed_point_features (scene/gaussian_model_ff.py:338-364) plus the renderer's
re-normalisation of its result (gaussian_renderer/__init__.py:362-363):

    normed = F.normalize(features, dim=-1, p=2)
    cols   = torch.randperm(K)[:int(K * dropout)]              # all K columns when dropout is outside (0, 1)
    ret    = normed[knn_idx[:, cols], :].mean(dim=1)
    ret    = ret / (ret.norm(dim=1, keepdim=True) + 1e-9)      # renderer, norm_point_features=True

as ONE gather kernel forward and two gather kernels backward (no (P, k, C) tensor, no index_put atomics).
The neighbour map itself is built once and cached by the reference (pytorch3d.ops.knn_points, absent here);
`NeighbourMap` caches the map and its inverse lists (`NeighbourMap.from_points` builds it with the exact HIP KNN of
seganygaussians_amd/knn.py); `knn_points_bruteforce` is the exhaustive checker the tests use.
There is no CPU path: CPU tensors raise."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from .rasterizer import _check, _dev_ptr, _stream_ptr


FUSED_MAX_K = 32   # neighbours per row the fused kernels handle (include/mi_knn_smooth.h)


class NeighbourMap:
    """knn_idx [P, K] (int32 on the GPU) plus the inverse lists the backward walks, built once per map
    (the reference caches `feature_smooth_map` the same way, gaussian_model_ff.py:345-352)."""

    def __init__(self, knn_idx: torch.Tensor):
        if knn_idx.dim() != 2 or knn_idx.size(1) < 1 or knn_idx.size(1) > FUSED_MAX_K:
            raise ValueError("knn_idx must have shape (P, K) with 1 <= K <= 32")
        if not knn_idx.is_cuda:
            raise RuntimeError("knn_idx must be on the GPU: the fused smoothing has no CPU path")
        P, K = knn_idx.shape
        self.P, self.K = int(P), int(K)
        self.idx = knn_idx.to(torch.int32).contiguous()
        flat = self.idx.reshape(-1).to(torch.int64)
        if P and (int(flat.min()) < 0 or int(flat.max()) >= P):
            raise ValueError("knn_idx holds indices outside [0, P)")
        order = torch.argsort(flat, stable=True)                       # entries grouped by the referenced Gaussian
        counts = torch.bincount(flat, minlength=P)
        off = torch.zeros(P + 1, dtype=torch.int64, device=knn_idx.device)
        off[1:] = torch.cumsum(counts, 0)
        self.inv_offsets = off.to(torch.int32).contiguous()
        self.inv_entries = (((order // K) << 5) | (order % K)).to(torch.int32).contiguous()  # uint32 bit pattern


    @classmethod
    def from_points(cls, xyz: torch.Tensor, K: int = 16) -> "NeighbourMap":
        """The map SAGA builds once per scene: `pytorch3d.ops.knn_points(xyz[None], xyz[None], K=K).idx.squeeze()`
        (gaussian_model_ff.py:345-352), here by the exact HIP search (seganygaussians_amd/knn.py)."""
        from .knn import KnnIndex
        idx, _ = KnnIndex(xyz).query(None, K)
        return cls(idx)


def _mask_of(K: int, cols) -> int:
    m = 0
    for c in (cols.tolist() if torch.is_tensor(cols) else cols):
        c = int(c)
        if not 0 <= c < K:
            raise ValueError(f"neighbour column {c} outside [0, {K})")
        m |= 1 << c
    return m


class _KnnSmooth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, nmap: NeighbourMap, sel_mask: int, normalize_out: bool):
        L = _lib.load()
        if not features.is_cuda:
            raise RuntimeError("features must be on the GPU: the fused smoothing has no CPU path")
        P, C = features.shape
        if P != nmap.P:
            raise ValueError(f"features has {P} rows, the neighbour map {nmap.P}")
        dev = features.device
        f = features.contiguous().float()
        out = torch.empty_like(f)
        with torch.cuda.device(dev):
            rc = L.mi_knn_smooth_forward(P, C, nmap.K, nmap.idx.data_ptr(), sel_mask, _dev_ptr(f, "features", dev),
                                         out.data_ptr(), int(bool(normalize_out)), _stream_ptr(dev))
        _check(rc)
        ctx.save_for_backward(f)
        ctx.nmap, ctx.sel_mask, ctx.normalize_out = nmap, sel_mask, bool(normalize_out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        L = _lib.load()
        (f,) = ctx.saved_tensors
        nmap = ctx.nmap
        P, C = f.shape
        dev = f.device
        g = grad_out.contiguous().float()
        dmean = torch.empty_like(f)
        dF = torch.empty_like(f)
        with torch.cuda.device(dev):
            rc = L.mi_knn_smooth_backward(P, C, nmap.K, nmap.idx.data_ptr(), nmap.inv_offsets.data_ptr(),
                                          nmap.inv_entries.data_ptr(), ctx.sel_mask, f.data_ptr(),
                                          _dev_ptr(g, "grad", dev), dmean.data_ptr(), dF.data_ptr(),
                                          int(ctx.normalize_out), _stream_ptr(dev))
        _check(rc)
        return dF, None, None, None


def smooth_point_features(features: torch.Tensor, nmap: NeighbourMap, cols=None, normalize_out: bool = True):
    """mean over the neighbour columns `cols` (default: all K) of the L2-normalised rows, optionally re-normalised."""
    mask = _mask_of(nmap.K, range(nmap.K) if cols is None else cols)
    return _KnnSmooth.apply(features, nmap, mask, normalize_out)


def get_smoothed_point_features(features: torch.Tensor, nmap: NeighbourMap, K: int = 16, dropout: float = 0.5,
                                normalize_out: bool = False, generator: Optional[torch.Generator] = None):
    """Drop-in for FeatureGaussianModel.get_smoothed_point_features (gaussian_model_ff.py:338-364): same column
    draw (`torch.randperm(K)[:int(K*dropout)]` on the CPU generator), same result.  normalize_out=True also applies
    the renderer's `x / (|x| + 1e-9)` (gaussian_renderer/__init__.py:362-363) inside the same kernel."""
    if K <= 1:
        return features
    assert dropout < 0 or int(K * dropout) >= 1
    if nmap.K != K:
        raise ValueError(f"neighbour map was built for K={nmap.K}, asked for K={K}")
    cols = torch.randperm(K, generator=generator)[: int(K * dropout)] if 0 < dropout < 1 else None
    return smooth_point_features(features, nmap, cols, normalize_out)


def fused_get_smoothed_point_features(self, K=16, dropout=0.5):
    """What `install_dropin(fuse_smoothing=True)` binds to the reference's FeatureGaussianModel.get_smoothed_point_features
    (scene/gaussian_model_ff.py:338-364): same signature and state (`self.feature_smooth_map = {"K", "m"}`, built with
    pytorch3d.ops.knn_points = the HIP KNN drop-in), same `torch.randperm(K)[:int(K*dropout)]` draw from the CPU generator, the
    normalise -> gather -> mean and its backward in the fused kernels instead of PyTorch's (P, k, C) gather and index_put."""
    if K <= 1:
        return self._point_features
    # The fused kernels keep a row's K neighbours in registers (K <= 32) and the HIP search returns K real neighbours (K <= P);
    # the reference expression takes any smooth_K and pytorch3d pads: outside those limits it is the reference's own method
    # that runs.
    if K > FUSED_MAX_K or K > self._point_features.shape[0]:
        ref = getattr(type(self), "_reference_get_smoothed_point_features", None)
        if ref is None:
            raise ValueError(f"fused feature smoothing supports K <= {FUSED_MAX_K} and K <= number of points; got K={K}")
        return ref(self, K, dropout)
    assert dropout < 0 or int(K * dropout) >= 1
    with torch.no_grad():
        if self.feature_smooth_map is None or self.feature_smooth_map["K"] != K:
            import pytorch3d.ops
            xyz = self.get_xyz
            nearest_k_idx = pytorch3d.ops.knn_points(xyz.unsqueeze(0), xyz.unsqueeze(0), K=K).idx.squeeze()
            self.feature_smooth_map = {"K": K, "m": nearest_k_idx}
        m = self.feature_smooth_map["m"]
        cached = getattr(self, "_mi_neighbour_map", None)
        if cached is None or cached[0] is not m:           # inverse lists: once per neighbour map, like the map itself
            cached = (m, NeighbourMap(m))
            self._mi_neighbour_map = cached
    cols = torch.randperm(K)[: int(K * dropout)] if 0 < dropout < 1 else None
    return smooth_point_features(self._point_features, cached[1], cols, normalize_out=False)


def knn_points_bruteforce(xyz: torch.Tensor, K: int, chunk: int = 4096) -> torch.Tensor:
    """K nearest neighbours (self included, nearest first) by chunked exhaustive search: what
    pytorch3d.ops.knn_points(xyz[None], xyz[None], K=K).idx.squeeze() returns.  O(P^2): the CHECKER of the HIP search
    (seganygaussians_amd/knn.py) in the tests; NeighbourMap.from_points uses the HIP search."""
    P = xyz.size(0)
    out = torch.empty((P, K), dtype=torch.int64, device=xyz.device)
    for s in range(0, P, chunk):
        d = torch.cdist(xyz[s:s + chunk], xyz)
        out[s:s + chunk] = d.topk(K, dim=1, largest=False).indices
    return out
