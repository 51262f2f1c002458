"""CPU check of the rank slices of the lean count / emit passes (seganygaussians_amd/csrc/binning.h: slice_rank_range), restated in
Python: for any per-bucket work and bucket sizes the slices partition the visible ranks [0, V) in order, every boundary is the
end of a non-empty bucket (or rank 0), empty inputs give empty slices, and on a work profile as skewed as the depth axis of the
benchmark scenes (near Gaussians cover hundreds of tiles, far ones one) no slice carries much more than the mean work."""
import numpy as np


def slice_rank_range(work, counts, s, nslices):
    """work[b], counts[b] per depth bucket; returns ranks [begin, end) of slice s (integer arithmetic of the kernel)."""
    starts = np.concatenate([[0], np.cumsum(counts)])
    ranges_y = np.where(counts > 0, starts[1:], 0)          # tile_ranges_kernel leaves {0, 0} for an empty bucket
    wpre = np.concatenate([[0], np.cumsum(work)])[:-1]      # W(b): work in front of bucket b
    total = int(work.sum())
    out = []
    for k in (s, s + 1):
        need = (k * total + nslices - 1) // nslices
        b = int((wpre < need).sum())
        out.append(int(ranges_y[b - 1]) if b else 0)
    return out


def _check(work, counts, nslices):
    V = int(counts.sum())
    prev_end = 0
    per_slice = []
    starts = np.concatenate([[0], np.cumsum(counts)])
    per_rank_bucket = np.repeat(np.arange(len(counts)), counts)
    for s in range(nslices):
        b, e = slice_rank_range(work, counts, s, nslices)
        assert b == prev_end and e >= b, (s, b, e, prev_end)
        assert b == 0 or b in set(starts[1:][counts > 0]), "boundary inside a bucket"
        prev_end = e
        per_slice.append(work[np.unique(per_rank_bucket[b:e])].sum() if e > b else 0)
    assert prev_end == V
    return np.array(per_slice, np.float64)


def test_slices_partition_the_visible_ranks():
    rng = np.random.default_rng(0)
    for nb, nslices in ((64, 4), (1024, 256), (16384, 256), (16384, 7), (300, 1)):
        counts = rng.integers(0, 4, nb) * (rng.random(nb) < 0.6)
        work = np.where(counts > 0, counts * rng.integers(9, 500, nb), 0)
        _check(work, counts, nslices)
    _check(np.zeros(128, np.int64), np.zeros(128, np.int64), 16)                      # nothing visible
    one = np.zeros(128, np.int64); cnt = np.zeros(128, np.int64); one[77] = 12345; cnt[77] = 1000
    ps = _check(one, cnt, 16)                                                         # a single bucket: one slice gets it all
    assert (ps > 0).sum() == 1


def test_slices_carry_equal_work_on_a_skewed_depth_axis():
    rng = np.random.default_rng(1)
    nb, nslices = 16384, 256
    counts = rng.poisson(60, nb)
    z = np.linspace(1.5, 12.0, nb)                       # tiles per Gaussian ~ 1 / z^2: the near buckets are the heavy ones
    tiles = np.maximum(1, (600.0 / z ** 2) * rng.lognormal(0, 0.5, nb)).astype(np.int64)
    work = counts * (tiles + 8)
    ps = _check(work, counts, nslices)
    assert ps.max() <= 1.3 * ps.mean(), (ps.max(), ps.mean())   # (boundaries are bucket boundaries: a near bucket is a tenth of a slice here)
