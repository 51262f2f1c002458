"""GPU tests of the exact HIP KNN (include/mi_knn.h) against exhaustive search: the pytorch3d.ops.knn_points calls SAGA
makes (scene/gaussian_model_ff.py:326,347,380) and simple_knn's distCUDA2 (scene/gaussian_model.py:20)."""
import numpy as np
import pytest
import torch

import seganygaussians_amd
from seganygaussians_amd import knn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cloud(n, seed, clustered=True):
    g = torch.Generator().manual_seed(seed)
    if not clustered:
        return torch.rand(n, 3, generator=g).to(DEV) * 4 - 2
    # surfaces + clusters + a few far outliers: what a trained scene's means look like (very uneven density)
    c = torch.randn(40, 3, generator=g) * 3
    k = torch.randint(0, 40, (n,), generator=g)
    p = c[k] + torch.randn(n, 3, generator=g) * (0.02 + 0.3 * torch.rand(40, generator=g)[k, None])
    p[: n // 4, 2] = 0.25 * p[: n // 4, 0]            # a plane
    p[-20:] = torch.randn(20, 3, generator=g) * 60     # outliers stretch the bounding box
    return p.to(DEV)


def _brute(q, ref, K, exclude_self=False):
    d = torch.cdist(q.double(), ref.double()).pow(2)
    if exclude_self:
        d.fill_diagonal_(float("inf"))
    v, i = d.topk(K, dim=1, largest=False)
    return i, v


def _d2_f32(q, ref, idx):
    """the kernel's own arithmetic: d.x*d.x + d.y*d.y + d.z*d.z in fp32, unfused"""
    d = ref[idx] - q[:, None, :]
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]


def _check(q, ref, K, idx, d2, exclude_self=False):
    bi, bv = _brute(q, ref, K, exclude_self)
    assert idx.shape == (q.size(0), K) and d2.shape == (q.size(0), K)
    assert torch.all(d2[:, 1:] >= d2[:, :-1]), "ascending"
    # the reported distances are the fp32 distances of the reported neighbours ...
    assert torch.equal(d2, _d2_f32(q, ref, idx))
    # ... and they are the K smallest: equal to exhaustive search up to fp32 rounding of a distance
    assert torch.allclose(d2.double(), bv, rtol=2e-6, atol=1e-12)
    # same neighbour sets wherever the K-th and (K+1)-th distances are not tied
    same = (idx.sort(1).values == bi.sort(1).values).all(1)
    assert same.float().mean() > 0.999, float(same.float().mean())
    if exclude_self:
        assert not (idx == torch.arange(q.size(0), device=idx.device)[:, None]).any()


@pytest.mark.parametrize("clustered", [False, True])
@pytest.mark.parametrize("K", [16, 4])
def test_knn_points_self(clustered, K):
    xyz = _cloud(20_000, 1, clustered)
    r = knn.knn_points(xyz.unsqueeze(0), xyz.unsqueeze(0), K=K)
    assert r.idx.shape == (1, 20_000, K) and r.dists.shape == (1, 20_000, K) and r.idx.dtype == torch.int64
    _check(xyz, xyz, K, r.idx[0], r.dists[0])
    assert torch.equal(r.idx[0][:, 0], torch.arange(20_000, device=DEV)) and float(r.dists[0][:, 0].abs().max()) == 0.0
    # the reference's own expression (gaussian_model_ff.py:345-352): .idx.squeeze()
    assert r.idx.squeeze().shape == (20_000, K)


def test_knn_points_subset_references_and_free_queries():
    """get_multi_resolution_smoothed_point_features (gaussian_model_ff.py:376-384): references = a random subset."""
    xyz = _cloud(15_000, 2)
    pm = torch.rand(15_000, generator=torch.Generator().manual_seed(3)) < 0.5
    ref = xyz[pm.to(DEV)]
    r = knn.knn_points(xyz.unsqueeze(0), ref.unsqueeze(0), K=4)
    _check(xyz, ref, 4, r.idx[0], r.dists[0])
    q = _cloud(3_000, 4, clustered=False) * 3          # queries partly outside the references' bounding box
    r = knn.knn_points(q.unsqueeze(0), ref.unsqueeze(0), K=8, return_nn=True)
    _check(q, ref, 8, r.idx[0], r.dists[0])
    assert torch.equal(r.knn[0], ref[r.idx[0]])


@pytest.mark.parametrize("K", [1, 3, 5, 32])
def test_knn_other_k_and_duplicates(K):
    xyz = _cloud(5_000, 5)
    xyz[100:140] = xyz[100]                           # 40 coincident points: ties at distance 0, broken by index
    idx, d2 = knn.KnnIndex(xyz).query(None, K)
    _check(xyz, xyz, K, idx, d2)
    if K >= 3:
        assert torch.equal(idx[100, :3], torch.tensor([100, 101, 102], device=DEV))


def test_knn_fewer_references_than_k():
    """K > M is refused: the kernels would pad with index -1, which `features[idx]` silently maps to the LAST row."""
    xyz = _cloud(3, 6, clustered=False)
    with pytest.raises(RuntimeError, match="neighbours requested from 3 reference points"):
        knn.KnnIndex(xyz).query(None, 4)
    with pytest.raises(RuntimeError, match="neighbours requested"):
        knn.KnnIndex(xyz).query(None, 3, exclude_self=True)
    idx, d2 = knn.KnnIndex(xyz).query(None, 3)
    assert torch.equal(idx.sort(1).values, torch.arange(3, device=DEV).expand(3, 3))
    with pytest.raises(RuntimeError, match="shape"):
        knn.KnnIndex(xyz).query(torch.zeros(5, 2, device=DEV), 1)


def test_dist_cuda2_matches_exhaustive_search():
    xyz = _cloud(30_000, 7)
    got = knn.distCUDA2(xyz)
    bi, bv = _brute(xyz, xyz, 3, exclude_self=True)
    want = _d2_f32(xyz, xyz, bi).sort(1).values
    s3 = (want[:, 0] + want[:, 1]) + want[:, 2]
    want = torch.div(s3, torch.full_like(s3, 3.0))      # a tensor divisor: true division (a scalar one becomes x * (1/3))
    assert torch.allclose(got, want, rtol=1e-6, atol=0)
    assert float((got != want).float().mean()) < 1e-3      # bit-identical except where the 3rd / 4th neighbours tie
    # through the drop-in import name the reference modules use
    seganygaussians_amd.install_dropin()
    from simple_knn._C import distCUDA2
    assert torch.equal(distCUDA2(xyz), got)


def test_pytorch3d_dropin_name_and_full_size_timing():
    """`pytorch3d.ops.knn_points` resolves to the HIP search; 1M points, K = 16 (the neighbour map of a real scene)."""
    seganygaussians_amd.install_dropin()
    import pytorch3d.ops
    xyz = _cloud(1_000_000, 8)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    idx = pytorch3d.ops.knn_points(xyz.unsqueeze(0), xyz.unsqueeze(0), K=16).idx.squeeze()
    e1.record()
    torch.cuda.synchronize()
    print(f"knn_points 1M x 1M, K=16: {e0.elapsed_time(e1):.1f} ms")
    e0.record()
    d = knn.distCUDA2(xyz)
    e1.record()
    torch.cuda.synchronize()
    print(f"distCUDA2 1M: {e0.elapsed_time(e1):.1f} ms")
    assert idx.shape == (1_000_000, 16) and torch.equal(idx[:, 0], torch.arange(1_000_000, device=DEV))
    sel = torch.arange(0, 1_000_000, 997, device=DEV)
    bi, bv = _brute(xyz[sel], xyz, 16)
    assert torch.allclose(_d2_f32(xyz[sel], xyz, idx[sel]).double(), bv, rtol=2e-6, atol=1e-12)
    assert bool(torch.isfinite(d).all()) and float(d.min()) >= 0


def test_neighbour_map_from_points_matches_bruteforce_map():
    from seganygaussians_amd import knn_smooth as ks
    xyz = _cloud(12_000, 9)
    nmap = ks.NeighbourMap.from_points(xyz, K=16)
    want, _ = _brute(xyz, xyz, 16)                 # fp64 exhaustive search (ks.knn_points_bruteforce's fp32 cdist is too coarse)
    same = (nmap.idx.long().sort(1).values == want.sort(1).values).all(1)
    assert same.float().mean() > 0.999
