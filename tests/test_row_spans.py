"""CPU property test of the row spans behind the lean lists (seganygaussians_amd/csrc/cull.h: shrink_rect, span_prepare,
band_columns; binning.h: bin_spans_kernel).

The product default lists a (Gaussian, tile) overlap only where the closed-form column interval of one of the tile's two
8-pixel bands says a pixel can reach alpha >= 1/255, and marks the quadrants the intervals cover.  That is only correct if
NO pixel that passes the blend kernels' own float test (`power <= 0` and `opacity * exp(power) >= 1/255`, forward.cu:339-347)
lies in a quadrant the spans leave out.  The functions are restated here in float32 numpy, operation by operation, and the
property is checked pixel by pixel on random conics -- isotropic to needle-like (aspect 100), opaque to just above 1/255,
means inside and outside the image.  (On the GPU: tests/test_gpu_parity.py::test_cull_is_exactly_conservative.)"""
import numpy as np

F = np.float32


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


def span_prepare(A, B, C, o, rad):
    """cull.h: span_prepare."""
    det = A * C - B * B
    cull = bool(A > 0 and C > 0 and det > 0)
    m = F(rad) + F(16.0)
    mag_max = (F(0.5) * (A + C) + np.abs(B)) * m * m
    with np.errstate(all="ignore"):
        tau_s = np.log(F(255.0) * o).astype(F) + F(3e-5) * mag_max + F(2e-4)
        s = F(2.0) * tau_s * (F(1.0) / det)
        ex = np.sqrt(s * C).astype(F)
        ey = np.sqrt(s * A).astype(F) * F(1.001) + F(0.01)
        if not (ex < F(1e9) and ey < F(1e9)):
            cull = False
        return dict(B=B, rcpA=F(1.0) / A, twotauA=F(2.0) * tau_s * A, det=det, ey=ey, ystar=B * ex * (F(1.0) / C), cull=cull)


def band_columns(p, x, y, Y, clo, chi):
    """cull.h: band_columns -> [lo, hi) in 8-pixel columns."""
    if not p["cull"]:
        return clo, chi
    dl, dh = max(y - (F(Y) + F(7.0)), -p["ey"]), min(y - F(Y), p["ey"])
    if not (dl <= dh):
        return clo, clo
    dyr = min(dh, max(dl, -p["ystar"]))
    dyl = min(dh, max(dl, p["ystar"]))
    Dr = max(p["twotauA"] - p["det"] * dyr * dyr, F(0))
    Dl = max(p["twotauA"] - p["det"] * dyl * dyl, F(0))
    dxmax = (np.sqrt(Dr).astype(F) - p["B"] * dyr) * p["rcpA"]
    dxmin = (-np.sqrt(Dl).astype(F) - p["B"] * dyl) * p["rcpA"]
    dxmax = dxmax + (F(1e-3) * np.abs(dxmax) + F(0.01))
    dxmin = dxmin - (F(1e-3) * np.abs(dxmin) + F(0.01))
    flo = np.ceil((x - dxmax - F(7.0)) * F(0.125))
    fhi = np.floor((x - dxmin) * F(0.125)) + F(1.0)
    lo, hi = max(flo, F(clo)), min(fhi, F(chi))
    if not (hi > lo):
        return clo, clo
    return int(lo), int(hi)


def quadrant_masks(x, y, A, B, C, o, rad, gx, gy):
    """What bin_spans_kernel lists for one Gaussian: {(tx, ty): 4-bit quadrant mask}."""
    rect = get_rect(x, y, rad, gx, gy)
    x0, y0, x1, y1 = shrink_rect(x, y, A, B, C, o, rad, rect)
    out = {}
    if x1 <= x0 or y1 <= y0:
        return out, rect
    pre = span_prepare(A, B, C, o, rad)
    for ty in range(y0, y1):
        lo0, hi0 = band_columns(pre, x, y, 16 * ty, 2 * x0, 2 * x1)
        lo1, hi1 = band_columns(pre, x, y, 16 * ty + 8, 2 * x0, 2 * x1)
        if not (hi0 > lo0 or hi1 > lo1):
            continue
        los = [l for l, h in ((lo0, hi0), (lo1, hi1)) if h > l]
        his = [h for l, h in ((lo0, hi0), (lo1, hi1)) if h > l]
        for tx in range(min(los) >> 1, (max(his) + 1) >> 1):
            c = 2 * tx
            m = (int(lo0 <= c < hi0) | int(lo0 <= c + 1 < hi0) << 1 | int(lo1 <= c < hi1) << 2 | int(lo1 <= c + 1 < hi1) << 3)
            if m:
                out[(tx, ty)] = m
    return out, rect


def blending_pixels(x, y, A, B, C, o, rect):
    """Pixels of the reference rect's tiles that pass the kernels' own test, float32 like gauss_power / gauss_exp (common.h)."""
    x0, y0, x1, y1 = rect
    px, py = np.meshgrid(np.arange(16 * x0, 16 * x1, dtype=F), np.arange(16 * y0, 16 * y1, dtype=F))
    dx, dy = x - px, y - py
    ha, nb, hc = F(-0.5) * A, -B, F(-0.5) * C
    power = (ha * dx * dx + hc * dy * dy) + nb * dx * dy
    with np.errstate(all="ignore"):
        t = o * np.exp(np.minimum(power, F(0)).astype(np.float64)).astype(F)
    ok = (power <= 0) & (t >= F(1.0 / 255.0))
    return px[ok].astype(int), py[ok].astype(int)


def test_no_blending_pixel_outside_the_spans():
    rng = np.random.default_rng(11)
    gx, gy = 40, 30
    listed = quads = blending_quads = rect_tiles = 0
    for k in range(2500):
        A, B, C, rad = _random_conic(rng)
        if rad > 400:
            continue
        x, y = F(rng.uniform(-40, 16 * gx + 40)), F(rng.uniform(-40, 16 * gy + 40))
        o = F(np.exp(rng.uniform(np.log(1.0 / 255.0), 0.0))) if k % 7 else F(rng.choice([0.0039, 0.00393, 0.004, 1.0, 0.99]))
        masks, rect = quadrant_masks(x, y, A, B, C, o, rad, gx, gy)
        if rect[2] <= rect[0] or rect[3] <= rect[1]:
            continue
        px, py = blending_pixels(x, y, A, B, C, o, rect)
        need = {}
        for u, v in zip(px.tolist(), py.tolist()):
            key = (u >> 4, v >> 4)
            need[key] = need.get(key, 0) | (1 << (((v >> 3) & 1) * 2 + ((u >> 3) & 1)))
        for key, m in need.items():
            have = masks.get(key, 0)
            assert (m & ~have) == 0, (k, key, bin(m), bin(have), float(A), float(B), float(C), float(o), rad, float(x), float(y))
        listed += len(masks)
        quads += sum(bin(m).count("1") for m in masks.values())
        blending_quads += sum(bin(m).count("1") for m in need.values())
        rect_tiles += (rect[2] - rect[0]) * (rect[3] - rect[1])
    assert blending_quads > 20000
    # tightness on this mix: the spans list few quadrants that no pixel blends into, and far fewer tiles than the rects hold
    assert quads <= 1.25 * blending_quads, (quads, blending_quads)
    assert listed < 0.5 * rect_tiles, (listed, rect_tiles)


def test_degenerate_conics_are_not_culled():
    gx, gy = 20, 20
    # not positive definite: every tile of the (reference) rect with all four quadrants
    masks, rect = quadrant_masks(F(100), F(100), F(1.0), F(2.0), F(1.0), F(0.5), 40, gx, gy)
    assert len(masks) == (rect[2] - rect[0]) * (rect[3] - rect[1]) and set(masks.values()) == {15}
    # nearly singular: extents overflow, same
    masks, rect = quadrant_masks(F(100), F(100), F(1e-30), F(0.0), F(1e-30), F(0.9), 40, gx, gy)
    assert len(masks) == (rect[2] - rect[0]) * (rect[3] - rect[1]) and set(masks.values()) == {15}
    # opacity below 1/255 or NaN: nothing is listed (alpha <= opacity < 1/255 everywhere)
    for o in (F(0.003), F(np.nan)):
        masks, _ = quadrant_masks(F(100), F(100), F(0.01), F(0.0), F(0.01), o, 40, gx, gy)
        assert not masks


def test_degenerate_inputs_keep_the_rect():
    rect = (3, 4, 9, 11)
    # not positive definite: no culling anywhere, the rect stays
    assert shrink_rect(F(100), F(100), F(1.0), F(2.0), F(1.0), F(0.5), 40, rect) == rect
    # opacity below 1/255 (or NaN): alpha <= opacity < 1/255 everywhere, the rect is emptied
    for o in (F(0.003), F(np.nan)):
        r = shrink_rect(F(100), F(100), F(0.01), F(0.0), F(0.01), o, 40, rect)
        assert r[2] <= r[0] or r[3] <= r[1]
    # nearly singular conic: extents overflow to huge values, the rect stays
    assert shrink_rect(F(100), F(100), F(1e-30), F(0.0), F(1e-30), F(0.9), 40, rect) == rect
