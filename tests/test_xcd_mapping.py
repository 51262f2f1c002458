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


def fwd_item_runs(wg, b):
    """blend_fwd_wave.h: fwd_wave_item with m == 0 -- the PRODUCT path: one run per XCD, boundaries from the range scan."""
    x, jj = wg & 7, wg >> 3
    tl, quad = jj >> 2, jj & 3
    start, length = b[x], b[x + 1] - b[x]
    return (start + tl, quad) if tl < length else None


def longest_of(b, n):
    """mi_rast.hip: longest_of -- what sizes the forward's grid."""
    longest = max(max(b1 - b0, 0) for b0, b1 in zip(b, b[1:]))
    return min(max(longest, (n + 7) >> 3), max_run(n))


def piece_search_bounds(counts, cap, fix):
    """binning.h: tile_ranges_kernel with run_cap > 0, as the kernel does it: 1024 threads own contiguous pieces of `per` tiles, a
    boundary k lies in the ONE piece whose weight prefix brackets k W / 8, and that piece's thread walks its tiles."""
    n = len(counts)
    w = np.minimum(np.asarray(counts, np.int64), cap) + fix
    per = (n + 1023) // 1024
    starts = [min(n, t * per) for t in range(1024)]
    ends = [min(n, s + per) for s in starts]
    piece_w = [int(w[s:e].sum()) for s, e in zip(starts, ends)]
    incl = np.cumsum(piece_w)
    total = int(incl[-1])
    b = [run_start(x, n) for x in range(9)]
    for t in range(1024):
        w_lo, w_hi = 8 * (int(incl[t]) - piece_w[t]), 8 * int(incl[t])
        for k in range(1, 8):
            want = k * total
            if w_lo < want <= w_hi:
                acc, i = w_lo, starts[t]
                while i < ends[t]:
                    acc += 8 * int(w[i])
                    if acc >= want:
                        break
                    i += 1
                b[k] = min(i + 1, ends[t])
    return clamp_runs(b, n)


def test_product_forward_mapping_and_piece_search():
    """The paths the product takes: m == 0 with the range scan's boundaries, and the piece-granular boundary search of
    tile_ranges_kernel against the plain searchsorted restatement (bounds_from_model)."""
    rng = np.random.default_rng(2)
    for n in (1, 7, 8, 255, 6700, 8160, 34680):
        for law in range(3):
            counts = (rng.integers(0, 50, n) if law == 0 else
                      np.where(np.arange(n) < n // 3, rng.integers(800, 3000, n), rng.integers(0, 120, n)) if law == 1 else
                      np.zeros(n, np.int64))
            b = piece_search_bounds(counts, 768, 128)
            _check(n, b)
            # first tile i with 8 W(i) >= k W_total, W(i) = weight in front of tile i + 1 (the kernel counts a tile in when its own
            # weight reaches the mark): the same boundaries as the prefix search up to that convention
            w = np.minimum(counts, 768) + 128
            cw = np.cumsum(w)
            want = [0] + [int(np.searchsorted(8 * cw, k * int(cw[-1]), side="left")) + 1 for k in range(1, 8)] + [n]
            assert b == clamp_runs(want, n), (n, law, b, want)
            grid = 32 * longest_of(b, n)
            got = sorted(4 * it[0] + it[1] for it in (fwd_item_runs(wg, b) for wg in range(grid)) if it is not None)
            assert got == list(range(4 * n)), (n, law)


# ---- bands of the lean count / emit passes (binning.h: band_geometry, band_plan) ----------------------------------------------------
MAX_BANDS, SPAN_MAX_HEAD_WORDS = 24, 40 * 1024 - 16 * 64 * 18 - 64


def band_geometry(gx, gy, max_head=SPAN_MAX_HEAD_WORDS):
    band_h = max(1, (gy + 15) // 16)
    while band_h > 1 and (band_h + 1) * (gx + 2) > max_head:
        band_h -= 1
    nbands = (gy + band_h - 1) // band_h
    return band_h, nbands, nbands <= MAX_BANDS and (band_h + 1) * (gx + 2) <= max_head


def band_plan(cnt, nwg):
    nbands, total = len(cnt), int(sum(cnt))
    avail = nwg - nbands if nwg > nbands else 0
    n = [1 + (c * avail // total if total else 0) for c in cnt]
    first = [0]
    for v in n:
        first.append(first[-1] + v)
    return first


def test_band_geometry_and_plan():
    for gx, gy in [(120, 68), (100, 67), (16, 16), (1, 1), (3, 40), (256, 160), (480, 270), (1023, 300)]:
        band_h, nbands, ok = band_geometry(gx, gy)
        assert ok and 1 <= nbands <= MAX_BANDS and band_h * nbands >= gy > band_h * (nbands - 1), (gx, gy, band_h, nbands)
    assert not band_geometry(1023, 2047)[2]          # beyond 24 bands of what fits the LDS: refused by the host
    rng = np.random.default_rng(3)
    for nbands in (1, 2, 14, 24):
        for nwg in (nbands, nbands + 1, 40, 256):
            for _ in range(20):
                cnt = rng.integers(0, 100000, nbands) * (rng.random(nbands) < 0.8)
                first = band_plan(cnt.tolist(), nwg)
                sizes = np.diff(first)
                assert first[0] == 0 and first[-1] <= nwg and sizes.min() >= 1, (cnt, nwg, first)
                if cnt.sum() and nwg >= 8 * nbands:      # workgroups follow the load: no band more than ~2x over its share
                    share = cnt / cnt.sum() * (nwg - nbands)
                    assert np.all(sizes <= share + 1 + 1e-9) and np.all(sizes >= np.floor(share)), (cnt, sizes)
