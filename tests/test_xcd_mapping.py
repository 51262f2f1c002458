"""CPU check of the blend kernels' workgroup -> (tile, quadrant) assignments (seganygaussians_amd/csrc/common.h: xcd_run_start,
xcd_max_run, xcd_static_len_max, xcd_queued_tiles_max, xcd_clamp_runs, xcd_grab_runs; binning.h: run_bounds_from_walks_kernel;
blend_bwd_wave.h: the end of the kernel; blend_fwd_wave.h: fwd_wave_item / fwd_runs_longest), restated in Python.

Backward: the row-major tile sequence is cut into eight contiguous runs -- equal tile counts (what every forward's range scan
leaves) or equal sums of (tile_nsurv + XCD_TILE_WEIGHT) (what the scan behind a forward that will be differentiated leaves),
clamped to at most twice the equal share.  For ANY such boundaries the statically assigned ids and the queued items together
cover every (tile, quadrant) exactly once, and the grid the stateless host launches (sized for the longest run the clamp allows)
has an id for every static item and at least one taker per queued item.
Forward: 8 m runs dealt to the XCDs round robin; every (tile, quadrant) has exactly one id in the grid."""
import numpy as np

QUEUE_DIV, MAX_RUN_FACTOR, TILE_WEIGHT = 2, 2, 128   # common.h


def run_start(x, n):
    return (x * n) >> 3


def max_run(n):
    return min(n, MAX_RUN_FACTOR * ((n + 7) >> 3))


def static_len_max(n):
    return max_run(n) - max_run(n) // QUEUE_DIV


def queued_tiles_max(n):
    return n // QUEUE_DIV


def clamp_runs(b, n):
    """common.h: xcd_clamp_runs."""
    b = list(b)
    mr = max_run(n)
    b[0], b[8] = 0, n
    for k in range(1, 8):
        prev, need = b[k - 1], (8 - k) * mr
        lo, hi = max(prev, n - need if n > need else 0), min(n, prev + mr)
        b[k] = min(max(b[k], lo), hi)
    return b


def bounds_from_walks(nsurv):
    """binning.h: run_bounds_from_walks_kernel."""
    n = len(nsurv)
    w = np.concatenate([[0], np.cumsum(np.asarray(nsurv, np.int64) + TILE_WEIGHT)])
    total = int(w[-1])
    b = [0] * 9
    for k in range(1, 8):
        b[k] = int(np.searchsorted(8 * w[:n], k * total, side="left"))   # first i with 8 W(i) >= k W_total
    return clamp_runs(b, n)


def bounds_from_model(counts, cap, fix):
    """binning.h: tile_ranges_kernel with run_cap > 0: runs of equal sum(min(list length, cap) + fix)."""
    return bounds_from_walks(np.minimum(np.asarray(counts, np.int64), cap) + fix - TILE_WEIGHT)


def assignment(n, b):
    """-> (items taken by static ids, items in the queues, grid size, takers) for run boundaries b[0..8]."""
    nstatic = 32 * static_len_max(n)
    static = []
    for wg in range(nstatic):
        x, jj = wg & 7, wg >> 3
        start, length = b[x], b[x + 1] - b[x]
        if (jj >> 2) < length - length // QUEUE_DIV:
            static.append(4 * start + jj)
    queued = []
    for x in range(8):
        start, length = b[x], b[x + 1] - b[x]
        q = length // QUEUE_DIV
        queued += [(start + length - q) * 4 + got for got in range(4 * q)]
    return static, queued, nstatic + 4 * queued_tiles_max(n), 4 * queued_tiles_max(n)


def _check(n, b):
    assert b[0] == 0 and b[8] == n and all(b1 >= b0 for b0, b1 in zip(b, b[1:])), b
    assert max(b1 - b0 for b0, b1 in zip(b, b[1:])) <= max_run(n), (n, b)
    static, queued, grid, takers = assignment(n, b)
    items = np.sort(np.array(static + queued, np.int64))
    assert np.array_equal(items, np.arange(4 * n)), (n, b)
    assert takers >= len(queued) and grid >= len(static) + len(queued)


def test_every_quadrant_exactly_once_for_any_runs():
    rng = np.random.default_rng(0)
    for n in list(range(1, 70)) + [120 * 68, 100 * 67, 16 * 16, 256 * 135, 40896, 8191, 8193]:
        _check(n, [run_start(x, n) for x in range(9)])                       # equal tile counts
        for _ in range(4):                                                   # any proposal survives the clamp as a valid partition
            _check(n, clamp_runs(sorted(rng.integers(0, n + 1, 9).tolist()), n))
        _check(n, clamp_runs([0] * 9, n))
        _check(n, clamp_runs([n] * 9, n))


def test_runs_of_equal_walked_weight():
    rng = np.random.default_rng(1)
    n = 120 * 68
    # a scene whose upper half is walked five times as deep as its lower half: the runs follow the weight, not the tile count
    nsurv = np.where(np.arange(n) < n // 2, rng.integers(400, 700, n), rng.integers(60, 160, n))
    b = bounds_from_walks(nsurv)
    _check(n, b)
    w = np.add.reduceat(nsurv + TILE_WEIGHT, b[:-1])
    assert w.max() <= 1.05 * w.mean(), w
    eq = np.add.reduceat(nsurv + TILE_WEIGHT, [run_start(x, n) for x in range(8)])
    assert eq.max() > 1.4 * eq.mean()                                        # what equal counts would have handed the busiest XCD
    _check(n, bounds_from_walks(np.zeros(n, np.int64)))                      # nothing walked: equal counts
    assert bounds_from_walks(np.zeros(n, np.int64)) == [run_start(x, n) for x in range(9)]
    # everything in one corner: the clamp keeps every run within twice the equal share
    corner = np.zeros(n, np.int64)
    corner[:200] = 100000
    _check(n, bounds_from_walks(corner))
    for m in (1, 3, 9, 17):
        _check(m, bounds_from_walks(rng.integers(0, 50, m)))
    # the range scan's MODEL of the same (before any walk is known): a long list costs what its cap says
    counts = np.where(np.arange(n) < n // 2, rng.integers(1500, 3000, n), rng.integers(20, 200, n))
    bm = bounds_from_model(counts, 256, 64)
    _check(n, bm)
    wm = np.add.reduceat(np.minimum(counts, 256) + 64, bm[:-1])
    assert wm.max() <= 1.05 * wm.mean(), wm


def fwd_item(wg, n, m):
    """blend_fwd_wave.h: fwd_wave_item."""
    x, jj = wg & 7, wg >> 3
    tl, quad = jj >> 2, jj & 3
    for k in range(m):
        i = x + 8 * k
        start = (i * n) // (8 * m)
        length = ((i + 1) * n) // (8 * m) - start
        if tl < length:
            return start + tl, quad
        tl -= length
    return None


def fwd_longest(n, m):
    return max(sum(((x + 8 * k + 1) * n) // (8 * m) - ((x + 8 * k) * n) // (8 * m) for k in range(m)) for x in range(8))


def test_forward_runs_cover_every_quadrant_once():
    for n in list(range(1, 40)) + [8160, 6700, 255]:
        for m in (1, 2, 4, 16):
            grid = 32 * fwd_longest(n, m)
            got = sorted(4 * it[0] + it[1] for it in (fwd_item(wg, n, m) for wg in range(grid)) if it is not None)
            assert got == list(range(4 * n)), (n, m)
