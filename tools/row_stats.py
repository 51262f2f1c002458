"""How many (quadrant, record) rows the backward walks on cfg3, and how many distinct (tile, record) pairs they are
(run on the GPU box): decides whether a tile-level reduction in front of the gradient atomics could pay."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as hp
from seganygaussians_amd import _lib
inp = hp.inputs_from_config("cfg3")
g = hp.GpuRun(inp).forward(full_lists=False)
W, H = inp.image_width, inp.image_height
tx, ty = (W + 15) // 16, (H + 15) // 16
R = g.num_rendered
_, off = _lib.binning_layout(R)
_, ioff = _lib.image_layout(W, H)
im = g.img_fields()
ranges = im["ranges"].reshape(-1, 2).astype(np.int64)
# the records the blend kernels stage: index_rec[id] of every blend-list entry (id | quadrant mask << 28), pm = position << 4 | mask
_, goff = _lib.geometry_layout(g.P)
bl = g._view(g.binning, off["blend_list"], R, np.uint32)
rec = g._view(g.geom, goff["index_rec"], 8 * g.P, np.uint32).reshape(g.P, 8)[bl & np.uint32(0x0FFFFFFF)]
pos_in_tile = np.arange(R, dtype=np.int64) - np.repeat(ranges[:, 0], ranges[:, 1] - ranges[:, 0]) if R else np.zeros(0, np.int64)
rec[:, 3] = (pos_in_tile.astype(np.uint32) << np.uint32(4)) | (bl >> np.uint32(28))
nsurv = g._view(g.img, ioff["tile_nsurv"], tx * ty, np.uint32).astype(np.int64)
nc = im["n_contrib"].reshape(H, W).astype(np.int64)
pad = np.zeros((ty * 16, tx * 16), np.int64); pad[:H, :W] = nc
Lq = pad.reshape(ty, 2, 8, tx, 2, 8).max(axis=(2, 5))          # [ty, qy, tx, qx]
rows = pairs = recs = 0
live_rows = live_lanes = lanes = 0   # rows with at least one blending pixel; blending (pixel, row) pairs; all of them
multi = np.zeros(5, np.int64)
recf = rec.view(np.float32)
ncp = pad.reshape(ty, 16, tx, 16)
yy, xx = np.mgrid[0:16, 0:16]
for t in range(tx * ty):
    a, n = ranges[t, 0], nsurv[t]
    if n == 0: continue
    pm = rec[a:a + n, 3].astype(np.int64)
    pos, mask = pm >> 4, pm & 15
    y, x = divmod(t, tx)
    hit = np.zeros(n, np.int64)
    for q in range(4):
        lt = Lq[y, q >> 1, x, q & 1]
        hit += (((mask >> q) & 1) == 1) & (pos < lt)
    rows += hit.sum(); pairs += (hit > 0).sum(); recs += n
    # which of those rows have a pixel that really blends (alpha >= 1/255, before the pixel's last contributor)?
    r = recf[a:a + n]
    dx = r[:, 0, None, None] - (x * 16 + xx)[None].astype(np.float32)
    dy = r[:, 1, None, None] - (y * 16 + yy)[None].astype(np.float32)
    power = (np.float32(-0.5) * r[:, 4, None, None] * dx * dx + np.float32(-0.5) * r[:, 6, None, None] * dy * dy) - r[:, 5, None, None] * dx * dy
    tt = r[:, 7, None, None] * np.exp(np.minimum(power, 0))
    blends = (power <= 0) & (tt >= np.float32(1 / 255)) & (pos[:, None, None] < ncp[y, :, x, :][None])
    bq = blends.reshape(n, 2, 8, 2, 8)
    for q in range(4):
        lt = Lq[y, q >> 1, x, q & 1]
        walked = (((mask >> q) & 1) == 1) & (pos < lt)
        cnt = bq[:, q >> 1, :, q & 1, :].sum(axis=(1, 2))
        live_rows += int(((cnt > 0) & walked).sum()); live_lanes += int(cnt[walked].sum()); lanes += 64 * int(walked.sum())
    multi += np.bincount(hit, minlength=5)[:5]
print(f"rows with at least one blending pixel: {live_rows} of {rows} ({live_rows / max(rows, 1):.3f}); blending lanes {live_lanes} of {lanes} "
      f"({live_lanes / max(lanes, 1):.3f})")
print(f"records walked {recs}, (quadrant, record) rows {rows}, distinct (tile, record) pairs {pairs}; records by number of quadrants 0..4: {multi.tolist()}")
