"""CPU check of the backward blend's workgroup -> (tile, quadrant) assignment (seganygaussians_amd/csrc/common.h: xcd_run_start,
xcd_static_len, xcd_queued_tiles, xcd_grab; blend_bwd_wave.h: the end of the kernel), restated in Python: for any tile count
the statically assigned ids and the queued items together cover every (tile, quadrant) exactly once, every XCD's tiles form one
contiguous run, and the grid the host launches has exactly one workgroup per queued item behind the static ids."""
import numpy as np

QUEUE_DIV = 2   # common.h: XCD_QUEUE_DIV


def run_start(x, n):
    return (x * n) >> 3


def static_len(n):
    longest = (n + 7) >> 3
    return longest - longest // QUEUE_DIV


def queued_tiles(n):
    return sum((run_start(x + 1, n) - run_start(x, n)) // QUEUE_DIV for x in range(8))


def assignment(n):
    """-> (items taken by static ids, items in the queues, grid size)"""
    nstatic = 32 * static_len(n)
    static = []
    for b in range(nstatic):
        x, jj = b & 7, b >> 3
        start, length = run_start(x, n), run_start(x + 1, n) - run_start(x, n)
        if (jj >> 2) < length - length // QUEUE_DIV:
            static.append(4 * start + jj)
    queued = []
    for x in range(8):
        start, length = run_start(x, n), run_start(x + 1, n) - run_start(x, n)
        q = length // QUEUE_DIV
        queued += [(start + length - q) * 4 + got for got in range(4 * q)]
    return static, queued, nstatic + 4 * queued_tiles(n)


def test_every_quadrant_exactly_once():
    for n in list(range(1, 70)) + [120 * 68, 100 * 67, 16 * 16, 256 * 135, 40896, 8191, 8193]:
        static, queued, grid = assignment(n)
        items = np.sort(np.array(static + queued, np.int64))
        assert np.array_equal(items, np.arange(4 * n)), n
        assert grid == 32 * static_len(n) + len(queued), n
        assert len(queued) == 4 * queued_tiles(n)
        # the quadrants of a tile are consecutive ids of one XCD (same id modulo 8), the tiles of an XCD one contiguous run
        for b in range(0, 32 * static_len(n), 97):
            x, jj = b & 7, b >> 3
            start, length = run_start(x, n), run_start(x + 1, n) - run_start(x, n)
            if (jj >> 2) < length - length // QUEUE_DIV:
                assert start <= (4 * start + jj) >> 2 < start + length


def test_runs_partition_the_tiles():
    for n in (1, 7, 8, 9, 8160, 6700, 40896):
        bounds = [run_start(x, n) for x in range(9)]
        assert bounds[0] == 0 and bounds[8] == n and all(b1 >= b0 for b0, b1 in zip(bounds, bounds[1:]))
        assert max(b1 - b0 for b0, b1 in zip(bounds, bounds[1:])) <= (n + 7) // 8
