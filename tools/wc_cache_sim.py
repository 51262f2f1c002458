"""Prices a per-CU software write-combining cache for the backward blend's gradient rows BEFORE anything is built:
how many of the (quadrant, record) rows the backward walks would find their Gaussian's row already in an LDS slot of their CU?

Model (run on the GPU box: it needs one forward of the configuration for the lists and walk lengths):
  * the kernel as it would have to be restructured: ONE persistent workgroup per CU, 12 waves (3 per SIMD), a wave = one 8x8 quadrant;
    the 32 CUs of an XCD share that XCD's contiguous run of tiles (run bounds of the range scan), either as 32 contiguous sub-runs
    ("contiguous": best case for locality) or dealt tile by tile ("dealt": what a queue gives);
  * a CU works on 3 tiles (12 quadrants) at a time, its waves advance 16 rows per turn, round robin, back to front like the kernel;
  * cache: direct-mapped by Gaussian id, S slots per CU (a slot = 32 feature + 8 geometry floats = 160 bytes: S = 256 is 40 KB,
    S = 1024 the whole LDS); hit: ds_add into the slot, no global request; miss: the slot's row leaves with global atomics (3 segment
    requests, what every row costs today) and the slot is taken.  Everything is flushed at the end.
Prints rows / misses (= the factor by which the 64-byte segment requests shrink) per S and assignment.
    python tools/wc_cache_sim.py [cfg3|cfg3s|cfg5] """
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seganygaussians_amd import _lib  # noqa: E402
from tests import helpers as hp  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
inp = hp.inputs_from_config(cfg)
g = hp.GpuRun(inp).forward(full_lists=False)
W, H = inp.image_width, inp.image_height
tx, ty = (W + 15) // 16, (H + 15) // 16
nt = tx * ty
R = g.num_rendered
_, boff = _lib.binning_layout(R)
_, ioff = _lib.image_layout(W, H)
im = g.img_fields()
ranges = im["ranges"].reshape(-1, 2).astype(np.int64)
bl = g._view(g.binning, boff["blend_list"], R, np.uint32)
nsurv = im["tile_nsurv"].astype(np.int64)
nc = im["n_contrib"].reshape(H, W).astype(np.int64)
pad = np.zeros((ty * 16, tx * 16), np.int64)
pad[:H, :W] = nc
Lq = pad.reshape(ty, 2, 8, tx, 2, 8).max(axis=(2, 5))          # [ty, qy, tx, qx]: entries the quadrant's backward walks
nr_off = ioff["num_rendered"] + 4 * (64 * 32 + 4)
bounds = g._view(g.img, nr_off, 9, np.uint32).astype(np.int64)   # the blend kernels' XCD runs (binning.h: NR_RUN_BOUNDS)

# rows of every (tile, quadrant): Gaussian ids back to front
quad_rows = [None] * (4 * nt)
total_rows = 0
for t in range(nt):
    a, n = ranges[t, 0], nsurv[t]
    y, x = divmod(t, tx)
    e = bl[a:a + n]
    ids, mask = (e & np.uint32(0x0FFFFFFF)).astype(np.int64), (e >> np.uint32(28)).astype(np.int64)
    pos = np.arange(n)
    for q in range(4):
        sel = (((mask >> q) & 1) == 1) & (pos < Lq[y, q >> 1, x, q & 1])
        r = ids[sel][::-1]
        quad_rows[4 * t + q] = r
        total_rows += len(r)


def simulate(S, contiguous):
    misses = 0
    for xcd in range(8):
        tiles = np.arange(bounds[xcd], bounds[xcd + 1])
        for cu in range(32):
            mine = tiles[(len(tiles) * cu) // 32:(len(tiles) * (cu + 1)) // 32] if contiguous else tiles[cu::32]
            tag = np.full(S, -1, np.int64)
            for g0 in range(0, len(mine), 3):
                lists = [quad_rows[4 * t + q] for t in mine[g0:g0 + 3] for q in range(4)]
                turn = 0
                live = True
                while live:
                    live = False
                    for r in lists:
                        c = r[16 * turn:16 * turn + 16]
                        if len(c) == 0:
                            continue
                        live = True
                        for i in c.tolist():
                            s = i % S
                            if tag[s] != i:
                                tag[s] = i
                                misses += 1
                    turn += 1
    return misses


print(f"{cfg}: {total_rows} (quadrant, record) rows walked by the backward = {3 * total_rows} segment requests today; run bounds {bounds.tolist()}")
distinct_pairs = sum(len(np.unique(np.concatenate([quad_rows[4 * t + q] for q in range(4)]))) for t in range(nt) if nsurv[t])
print(f"  merging the four quadrants of a tile exactly (no cache): rows / distinct (tile, record) pairs = {total_rows / max(distinct_pairs, 1):.2f}")
for contiguous in (True, False):
    for S in (128, 256, 512, 1024):
        m = simulate(S, contiguous)
        print(f"  {'contiguous sub-runs' if contiguous else 'dealt tile by tile  '}  S = {S:4d} slots ({S * 160 // 1024:3d} KB): misses {m}, "
              f"requests shrink by {total_rows / m:.2f}x")
