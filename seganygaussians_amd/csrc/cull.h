// cull.h -- exact-conservative culling of tile-list entries, per 8x8-pixel quadrant.
//
// The tile lists (bit-exact contract) are built from the 3-sigma bounding SQUARE of every Gaussian
// (CF/cuda_rasterizer/forward.cu:232-240), which is very loose for anisotropic or faint Gaussians: on the
// BASELINE config-3 scene only 1/3 of the consumed list entries blend into ANY pixel of their tile and only
// 1/4 of the (8x8-quadrant, entry) pairs have a blending pixel.  The reference evaluates all of them for
// every pixel.  Here an entry is kept for a quadrant only if some pixel of it can reach alpha >= 1/255
// (forward.cu:346-347), i.e. only if the ellipse  q(d) = 1/2 (A dx^2 + C dy^2) + B dx dy <= ln(255 * opacity) + margin
// meets the quadrant.  The margin (relative to the magnitude of the terms, >> float rounding of both this bound and
// the kernel's own `power`) keeps the test a strict superset, so the per-pixel arithmetic -- and therefore the result --
// is unchanged; culled entries are simply never visited.  Survivors keep their list order, so front-to-back order and
// contributor indices are preserved.
//
// The decision is taken per 8-pixel BAND of rows, in closed form (band_columns below): the binning pass of the lean lists
// (binning.h: bin_spans_kernel) never looks at a tile that has no quadrant to keep.
#pragma once

#include "common.h"

namespace mirast {

// Square root of the cull's closed forms: v_sqrt_f32 (1 ulp) instead of the correctly rounded sqrtf (a dozen instructions more; six
// of them per (Gaussian, row) item were a third of the count / emit passes' arithmetic).  Every extent computed here carries 0.1 %
// + 0.01 px of slack, and the count pass, the emit pass and the full-list masks all call these same functions: one more ulp moves
// no decision relative to another.
__device__ __forceinline__ float cull_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// Lean lists: the part of a Gaussian's reference tile rect [rmin, rmax) that the ellipse { q <= tau_big } can meet at all:
// its axis-aligned bounding box |dx| <= sqrt(2 tau_big C / det), |dy| <= sqrt(2 tau_big A / det).  tau_big = ln(255 o) plus
// the margin: 3e-5 of mag_max, the largest magnitude the terms of `power` reach anywhere in the rect (radius + 15 px), + 2e-4.  The 3-sigma SQUARE of the reference is about twice that area on the benchmark scene.
// Slack (3x the margin, 0.1 % + 0.01 px on the extents) dwarfs the float rounding of both sides: a strict superset.
__device__ __forceinline__ void shrink_rect(float2 xy, float4 co, int rad, uint2& rmin, uint2& rmax)
{
    const float A = co.x, B = co.y, C = co.z, o = co.w;
    if (!(o >= (1.0f / 255.0f))) {  // alpha <= opacity < 1/255 everywhere (also drops NaN opacity)
        rmax = rmin;
        return;
    }
    const float det = A * C - B * B;
    if (!(A > 0.f && C > 0.f && det > 0.f)) return;  // not positive definite: no culling
    const float m = (float)rad + 16.0f;
    const float mag_max = (0.5f * (A + C) + fabsf(B)) * m * m;
    const float tau_big = __logf(255.0f * o) + 3e-5f * mag_max + 2e-4f;
    const float s = 2.0f * tau_big * __builtin_amdgcn_rcpf(det);
    const float ex = cull_sqrt(s * C) * 1.001f + 0.01f, ey = cull_sqrt(s * A) * 1.001f + 0.01f;
    if (!(ex < 1e9f && ey < 1e9f)) return;  // overflow / NaN: keep the rect
    // tile column tx holds pixels 16 tx .. 16 tx + 15: it meets [x - ex, x + ex] iff tx >= (x - ex - 15) / 16 and tx <= (x + ex) / 16
    const float lx = floorf((xy.x - ex - 15.0f) * (1.0f / 16.0f)), hx = floorf((xy.x + ex) * (1.0f / 16.0f)) + 1.0f;
    const float ly = floorf((xy.y - ey - 15.0f) * (1.0f / 16.0f)), hy = floorf((xy.y + ey) * (1.0f / 16.0f)) + 1.0f;
    const float fx0 = fmaxf((float)rmin.x, lx), fx1 = fminf((float)rmax.x, hx);
    const float fy0 = fmaxf((float)rmin.y, ly), fy1 = fminf((float)rmax.y, hy);
    if (!(fx1 > fx0 && fy1 > fy0)) {
        rmax = rmin;
        return;
    }
    rmin = make_uint2((uint32_t)fx0, (uint32_t)fy0);
    rmax = make_uint2((uint32_t)fx1, (uint32_t)fy1);
}

// ---- row spans: which 8-pixel columns of an 8-pixel band of rows a Gaussian can blend into --------------------------------
// The set { q <= tau } is a convex ellipse, so inside one band of rows [Y, Y + 7] the pixels it can reach form ONE interval of
// columns: from the leftmost to the rightmost point of (ellipse intersected with the band).  Both are closed forms: for fixed
// dy the ellipse spans dx in [r-(dy), r+(dy)], r+-(dy) = (-B dy +- sqrt(2 tau A - det dy^2)) / A; r+ is concave with its
// maximum at the ellipse's rightmost point dy* = -B ex / C (ex = sqrt(2 tau C / det)), so over the band it peaks at dy*
// clamped into the band; r- mirrored.  One evaluation per (Gaussian, band) replaces the per-tile and per-quadrant box tests
// above: the quadrant mask of a tile is read off the column intervals of its two bands (binning.h: bin_spans_kernel).
// Conservative like shrink_rect: tau carries the same margin (3e-5 of the largest magnitude the terms of `power` reach
// anywhere in the rect, + 2e-4), the extents 0.1 % + 0.01 px of slack -- far above the float rounding of this closed form
// and of the kernels' own `power`; tests/test_row_spans.py checks the superset property pixel by pixel on the CPU and
// tests/test_gpu_parity.py::test_cull_is_exactly_conservative that images do not change.
struct SpanPre {
    float B, rcpA, twotauA, det, ey, ystar;
    bool cull;  // false: not positive definite or out of float range -> no culling (every column of the rect, every row)
};

__device__ __forceinline__ SpanPre span_prepare(float4 co, int rad)
{
    const float A = co.x, B = co.y, C = co.z, o = co.w;
    SpanPre p;
    const float det = A * C - B * B;
    p.cull = A > 0.f && C > 0.f && det > 0.f;
    const float m = (float)rad + 16.0f;
    const float mag_max = (0.5f * (A + C) + fabsf(B)) * m * m;
    const float tau_s = __logf(255.0f * o) + 3e-5f * mag_max + 2e-4f;
    const float s = 2.0f * tau_s * __builtin_amdgcn_rcpf(det);
    const float ex = cull_sqrt(s * C), ey = cull_sqrt(s * A) * 1.001f + 0.01f;
    if (!(ex < 1e9f && ey < 1e9f)) p.cull = false;  // overflow / NaN
    p.B = B;
    p.rcpA = __builtin_amdgcn_rcpf(A);
    p.twotauA = 2.0f * tau_s * A;
    p.det = det;
    p.ey = ey;
    p.ystar = B * ex * __builtin_amdgcn_rcpf(C);
    return p;
}

// Columns [lo, hi) (8 pixels each, clipped to [clo, chi)) of the band of pixel rows [Y, Y + 7]; lo == hi: none.
__device__ __forceinline__ void band_columns(const SpanPre& p, float2 xy, float Y, int clo, int chi, int& lo, int& hi)
{
    if (!p.cull) {
        lo = clo;
        hi = chi;
        return;
    }
    // d = mean - pixel: dy runs over [y - (Y + 7), y - Y], cut to the ellipse's own extent
    const float dl = fmaxf(xy.y - (Y + 7.0f), -p.ey), dh = fminf(xy.y - Y, p.ey);
    if (!(dl <= dh)) {
        lo = hi = clo;
        return;
    }
    const float dyr = fminf(dh, fmaxf(dl, -p.ystar)), dyl = fminf(dh, fmaxf(dl, p.ystar));
    const float Dr = fmaxf(p.twotauA - p.det * dyr * dyr, 0.f), Dl = fmaxf(p.twotauA - p.det * dyl * dyl, 0.f);
    float dxmax = (cull_sqrt(Dr) - p.B * dyr) * p.rcpA, dxmin = (-cull_sqrt(Dl) - p.B * dyl) * p.rcpA;
    dxmax += 1e-3f * fabsf(dxmax) + 0.01f;
    dxmin -= 1e-3f * fabsf(dxmin) + 0.01f;
    // pixel x = mean.x - dx in [x - dxmax, x - dxmin]; column c holds pixels 8c .. 8c + 7
    const float flo = ceilf((xy.x - dxmax - 7.0f) * 0.125f), fhi = floorf((xy.x - dxmin) * 0.125f) + 1.0f;
    const float l = fmaxf(flo, (float)clo), h = fminf(fhi, (float)chi);
    if (!(h > l)) {  // also NaN
        lo = hi = clo;
        return;
    }
    lo = (int)l;
    hi = (int)h;
}

// Quadrant mask of ONE tile (tx, ty) of the reference rect: bit q (= qy*2 + qx) set <=> the 8x8 pixel quadrant q may receive
// alpha >= 1/255 from this Gaussian.  The same decision as bin_spans_kernel takes from the same functions (the shrunk rect
// clips rows and columns, the two bands of the tile row give the column intervals), evaluated per tile: what the full-list
// mode (MI_RAST_FULL_LISTS, parity tests) uses, so that lean and full lists hold identical masks.
__device__ __forceinline__ uint32_t span_tile_mask(float2 xy, float4 co, int rad, uint32_t tx, uint32_t ty, uint32_t gx, uint32_t gy)
{
    uint2 rmin, rmax;
    getRect(xy.x, xy.y, rad, rmin, rmax, gx, gy);
    shrink_rect(xy, co, rad, rmin, rmax);
    if (!(tx >= rmin.x && tx < rmax.x && ty >= rmin.y && ty < rmax.y)) return 0u;
    const SpanPre pre = span_prepare(co, rad);
    int lo0, hi0, lo1, hi1;
    band_columns(pre, xy, (float)(ty * TILE_Y), (int)(2u * rmin.x), (int)(2u * rmax.x), lo0, hi0);
    band_columns(pre, xy, (float)(ty * TILE_Y + 8u), (int)(2u * rmin.x), (int)(2u * rmax.x), lo1, hi1);
    const uint32_t c = 2u * tx, n0 = (uint32_t)(hi0 - lo0), n1 = (uint32_t)(hi1 - lo1);
    return (uint32_t)(c - (uint32_t)lo0 < n0) | ((uint32_t)(c + 1u - (uint32_t)lo0 < n0) << 1) |
           ((uint32_t)(c - (uint32_t)lo1 < n1) << 2) | ((uint32_t)(c + 1u - (uint32_t)lo1 < n1) << 3);
}

}  // namespace mirast
