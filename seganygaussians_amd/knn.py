"""Exact K-nearest-neighbour search on the GPU (include/mi_knn.h, csrc/knn.h): the two KNN entry points SAGA uses at
the edges of the rasterizer hot path, with their reference call signatures.

    knn_points(p1, p2, K=...)   pytorch3d.ops.knn_points as scene/gaussian_model_ff.py:326,347,380 calls it
                                (batch of one; returns a namedtuple with .dists, .idx, .knn like pytorch3d's)
    distCUDA2(points)           simple_knn._C.distCUDA2 (scene/gaussian_model.py:20, create_from_pcd)

PyTorch supplies device memory and the stream; the search runs in libmi_rast.so.  No CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional

import torch

from . import _lib


class _KNN(NamedTuple):   # pytorch3d.ops.knn._KNN
    dists: torch.Tensor
    idx: torch.Tensor
    knn: Optional[torch.Tensor]


_SUPPORTED_K = (1, 3, 4, 8, 16, 32)


def _check(rc):
    if rc != 0:
        raise RuntimeError(_lib.last_error())


def _prep(x: torch.Tensor, name: str) -> torch.Tensor:
    if not x.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor (got {x.device}); the MI355X KNN has no CPU path")
    return x.detach().to(torch.float32).contiguous()


class KnnIndex:
    """The index over one set of reference points (Morton-sorted copy + two levels of bounding boxes)."""

    def __init__(self, ref: torch.Tensor):
        L = _lib.load()
        ref = _prep(ref, "reference points")
        if ref.dim() != 2 or ref.size(1) != 3 or ref.size(0) == 0:
            raise RuntimeError("reference points must have dimensions (num_points > 0, 3)")
        self.M, self.device = ref.size(0), ref.device
        nbytes = int(L.mi_knn_workspace_bytes(self.M))
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=ref.device)
        with torch.cuda.device(ref.device):
            _check(L.mi_knn_build(self.M, ref.data_ptr(), self.workspace.data_ptr(), nbytes,
                                  torch.cuda.current_stream(ref.device).cuda_stream))
        self._ref = ref   # not needed by the index; kept so that query(None) can report shapes

    def query(self, query: Optional[torch.Tensor], K: int, exclude_self: bool = False):
        """(idx int64 [rows, K], dist2 float32 [rows, K]); query=None: the references themselves, row i = reference i."""
        L = _lib.load()
        kt = next((k for k in _SUPPORTED_K if k >= K), None)
        if K < 1 or kt is None:
            raise RuntimeError(f"K must be between 1 and {_SUPPORTED_K[-1]}")
        q = None if query is None else _prep(query, "query points")
        if q is not None and (q.dim() != 2 or q.size(1) != 3):
            raise RuntimeError(f"query points must have shape (N, 3), got {tuple(q.shape)}")
        if K > self.M - (1 if (q is None and exclude_self) else 0):
            # fewer candidates than K: the kernels would pad with index -1 / distance 3.4e38, and `features[idx]` with -1 silently
            # picks the LAST row under torch indexing -- refuse instead
            raise RuntimeError(f"K = {K} neighbours requested from {self.M} reference points")
        rows = self.M if q is None else q.size(0)
        idx = torch.empty((rows, kt), dtype=torch.int64, device=self.device)
        d2 = torch.empty((rows, kt), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _check(L.mi_knn_query(rows, None if q is None else q.data_ptr(), self.M, self.workspace.data_ptr(), kt,
                                  int(bool(exclude_self)), idx.data_ptr(), d2.data_ptr(),
                                  torch.cuda.current_stream(self.device).cuda_stream))
        return (idx[:, :K].contiguous(), d2[:, :K].contiguous()) if kt != K else (idx, d2)


def knn_points(p1: torch.Tensor, p2: torch.Tensor, lengths1=None, lengths2=None, norm: int = 2, K: int = 1, version: int = -1,
               return_nn: bool = False, return_sorted: bool = True) -> _KNN:
    """pytorch3d.ops.knn_points for the calls SAGA makes: p1 (1, N, 3), p2 (1, M, 3), squared L2 distances, sorted
    ascending.  When p1 and p2 are the same tensor the queries are taken to be the references (a point's nearest
    neighbour is itself, distance 0 -- what pytorch3d returns as well)."""
    if norm != 2 or lengths1 is not None or lengths2 is not None:
        raise NotImplementedError("knn_points: only norm=2 without per-cloud lengths (what SAGA uses)")
    if p1.dim() != 3 or p2.dim() != 3 or p1.size(0) != 1 or p2.size(0) != 1 or p1.size(2) != 3 or p2.size(2) != 3:
        raise NotImplementedError("knn_points: expected one cloud per call, shapes (1, N, 3) and (1, M, 3)")
    index = KnnIndex(p2[0])
    same = p1.data_ptr() == p2.data_ptr() and p1.shape == p2.shape
    idx, d2 = index.query(None if same else p1[0], K)
    nn = p2[0][idx.clamp_min(0)][None] if return_nn else None
    return _KNN(dists=d2[None], idx=idx[None], knn=nn)


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """simple_knn._C.distCUDA2 (submodules/simple-knn/spatial.cu:16-25): mean squared distance of every point to its
    three nearest other points."""
    L = _lib.load()
    pts = _prep(points, "points")
    P = pts.size(0)
    out = torch.zeros(P, dtype=torch.float32, device=pts.device)
    if P == 0:
        return out
    nbytes = int(L.mi_knn_workspace_bytes(P))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=pts.device)
    with torch.cuda.device(pts.device):
        _check(L.mi_knn_mean_dist2(P, pts.data_ptr(), ws.data_ptr(), nbytes, out.data_ptr(),
                                   torch.cuda.current_stream(pts.device).cuda_stream))
    return out
