"""CPU property test of the lean-list rect shrink (seganygaussians_amd/csrc/cull.h: shrink_rect).

The emit / count passes of the default ("lean") list mode walk only the part of a Gaussian's reference tile rect that
`shrink_rect` leaves.  That is only correct if every tile the whole-tile cull `tile_may_blend` keeps lies inside the
shrunk rect.  Both functions are restated here in float32 numpy, operation by operation, and the inclusion is checked
on random conics -- isotropic to needle-like, opaque to just above 1/255 -- over every tile of their reference rect.
(On the GPU the same property is checked end to end: lean and full runs must produce identical blend lists.)"""
import numpy as np

F = np.float32


def _edge_min(A, B, C, rcpC, dxe, lo, hi):
    t = np.minimum(hi, np.maximum(lo, -B * dxe * rcpC))
    return F(0.5) * (A * dxe * dxe + C * t * t) + B * dxe * t


def tile_may_blend(x, y, A, B, C, o, tpx, tpy):
    """cull.h: tile_may_blend, vectorised over tiles (tpx, tpy arrays); scalars float32."""
    if not (o >= F(1.0 / 255.0)):
        return np.zeros(tpx.shape, bool)
    if not (A > 0 and C > 0 and A * C - B * B > 0):
        return np.ones(tpx.shape, bool)
    dxl, dxh = x - (tpx + F(15.0)), x - tpx
    dyl, dyh = y - (tpy + F(15.0)), y - tpy
    inside = (dxl <= 0) & (dxh >= 0) & (dyl <= 0) & (dyh >= 0)
    tau = np.log(F(255.0) * o).astype(F)
    rcpA, rcpC = F(1.0) / A, F(1.0) / C
    e0 = _edge_min(A, B, C, rcpC, dxl, dyl, dyh)
    e1 = _edge_min(A, B, C, rcpC, dxh, dyl, dyh)
    e2 = _edge_min(C, B, A, rcpA, dyl, dxl, dxh)
    e3 = _edge_min(C, B, A, rcpA, dyh, dxl, dxh)
    qmin = np.minimum(np.minimum(e0, e1), np.minimum(e2, e3))
    mx = np.maximum(np.abs(dxl), np.abs(dxh))
    my = np.maximum(np.abs(dyl), np.abs(dyh))
    mag = F(0.5) * (A * mx * mx + C * my * my) + np.abs(B) * mx * my
    return inside | ~(qmin > tau + F(1e-5) * mag + F(1e-4))


def get_rect(x, y, rad, gx, gy):
    """common.h: getRect (truncating float -> int conversions, clamped to the grid)."""
    c = lambda v, g: int(min(g, max(0, int(np.trunc(v)))))
    return (c((x - F(rad)) / F(16), gx), c((y - F(rad)) / F(16), gy),
            c((x + F(rad) + F(15)) / F(16), gx), c((y + F(rad) + F(15)) / F(16), gy))


def shrink_rect(x, y, A, B, C, o, rad, rect):
    """cull.h: shrink_rect; returns the shrunk (x0, y0, x1, y1), possibly empty."""
    x0, y0, x1, y1 = rect
    if not (o >= F(1.0 / 255.0)):
        return (x0, y0, x0, y0)
    det = A * C - B * B
    if not (A > 0 and C > 0 and det > 0):
        return rect
    m = F(rad) + F(16.0)
    mag_max = (F(0.5) * (A + C) + np.abs(B)) * m * m
    tau_big = np.log(F(255.0) * o).astype(F) + F(3e-5) * mag_max + F(2e-4)
    s = F(2.0) * tau_big * (F(1.0) / det)
    ex = np.sqrt(s * C).astype(F) * F(1.001) + F(0.01)
    ey = np.sqrt(s * A).astype(F) * F(1.001) + F(0.01)
    if not (ex < F(1e9) and ey < F(1e9)):
        return rect
    lx, hx = np.floor((x - ex - F(15)) * F(1 / 16)), np.floor((x + ex) * F(1 / 16)) + F(1)
    ly, hy = np.floor((y - ey - F(15)) * F(1 / 16)), np.floor((y + ey) * F(1 / 16)) + F(1)
    fx0, fx1 = max(F(x0), lx), min(F(x1), hx)
    fy0, fy1 = max(F(y0), ly), min(F(y1), hy)
    if not (fx1 > fx0 and fy1 > fy0):
        return (x0, y0, x0, y0)
    return (int(fx0), int(fy0), int(fx1), int(fy1))


def _random_conic(rng):
    """cov2D = R diag(s1^2, s2^2) R^T + 0.3 I (forward.cu:104-110), conic = its inverse, radius = ceil(3 sqrt(lambda_max))."""
    s1 = F(np.exp(rng.uniform(np.log(0.2), np.log(80.0))))
    s2 = F(s1 * np.exp(rng.uniform(np.log(0.01), 0.0)))   # aspect ratios up to 100
    th = rng.uniform(0, np.pi)
    c, s = F(np.cos(th)), F(np.sin(th))
    a = c * c * s1 * s1 + s * s * s2 * s2 + F(0.3)
    b = c * s * (s1 * s1 - s2 * s2)
    d = s * s * s1 * s1 + c * c * s2 * s2 + F(0.3)
    det = a * d - b * b
    A, B, C = d / det, -b / det, a / det
    mid = F(0.5) * (a + d)
    lam = mid + np.sqrt(max(F(0.1), mid * mid - det)).astype(F)
    rad = int(np.ceil(F(3.0) * np.sqrt(lam)))
    return F(A), F(B), F(C), rad


def test_every_tile_the_cull_keeps_is_inside_the_shrunk_rect():
    rng = np.random.default_rng(7)
    gx, gy = 120, 68
    kept_total = shrunk_total = full_total = 0
    for k in range(4000):
        A, B, C, rad = _random_conic(rng)
        x, y = F(rng.uniform(-60, 16 * gx + 60)), F(rng.uniform(-60, 16 * gy + 60))
        o = F(np.exp(rng.uniform(np.log(1.0 / 255.0), 0.0))) if k % 7 else F(rng.choice([0.0039, 0.00393, 0.004, 1.0, 0.99]))
        rect = get_rect(x, y, rad, gx, gy)
        x0, y0, x1, y1 = rect
        if x1 <= x0 or y1 <= y0:
            continue
        tx, ty = np.meshgrid(np.arange(x0, x1), np.arange(y0, y1))
        keep = tile_may_blend(x, y, A, B, C, o, (tx * 16).astype(F), (ty * 16).astype(F))
        sx0, sy0, sx1, sy1 = shrink_rect(x, y, A, B, C, o, rad, rect)
        inside = (tx >= sx0) & (tx < sx1) & (ty >= sy0) & (ty < sy1)
        assert not np.any(keep & ~inside), (k, float(A), float(B), float(C), float(o), rad, rect, (sx0, sy0, sx1, sy1))
        kept_total += int(keep.sum())
        shrunk_total += max(0, sx1 - sx0) * max(0, sy1 - sy0)
        full_total += (x1 - x0) * (y1 - y0)
    assert kept_total > 10000 and kept_total <= shrunk_total <= full_total
    # the shrink is worth something on this mix: at least a third of the reference rect tiles are never walked
    assert shrunk_total < 0.67 * full_total, (kept_total, shrunk_total, full_total)


def test_degenerate_inputs_keep_the_rect():
    rect = (3, 4, 9, 11)
    # not positive definite: no culling anywhere, the rect stays
    assert shrink_rect(F(100), F(100), F(1.0), F(2.0), F(1.0), F(0.5), 40, rect) == rect
    # opacity below 1/255 (or NaN): tile_may_blend is false everywhere, the rect is emptied
    for o in (F(0.003), F(np.nan)):
        r = shrink_rect(F(100), F(100), F(0.01), F(0.0), F(0.01), o, 40, rect)
        assert r[2] <= r[0] or r[3] <= r[1]
    # nearly singular conic: extents overflow to huge values, the rect stays
    assert shrink_rect(F(100), F(100), F(1e-30), F(0.0), F(1e-30), F(0.9), 40, rect) == rect
