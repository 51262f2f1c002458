// cull.h -- exact-conservative per-quadrant culling of tile-list entries + order-preserving compaction.
//
// The tile lists (bit-exact contract) are built from the 3-sigma bounding SQUARE of every Gaussian
// (CF/cuda_rasterizer/forward.cu:232-240), which is very loose for anisotropic or faint Gaussians: on the
// BASELINE config-3 scene only 1/3 of the consumed list entries blend into ANY pixel of their tile and only
// 1/4 of the (8x8-quadrant, entry) pairs have a blending pixel.  The reference evaluates all of them for
// every pixel.  Here, while a batch of 256 entries is staged, the staging thread of each entry evaluates in
// closed form the MINIMUM over each 8x8 quadrant of  q(d) = 1/2 (A dx^2 + C dy^2) + B dx dy  (convex
// quadratic over a box: 0 if the mean is inside, else the best of the four clamped edge minima) and keeps
// the entry for that quadrant only if  min q <= ln(255 * opacity) + margin,  i.e. only if some pixel can
// reach alpha >= 1/255 (forward.cu:346-347).  The margin (relative to the magnitude of the terms, >> float
// rounding of both this bound and the kernel's own `power`) keeps the test a strict superset, so the
// per-pixel arithmetic -- and therefore the result -- is unchanged; culled entries are simply never visited.
// Survivors are compacted in list order, so front-to-back order and contributor indices are preserved.
#pragma once

#include "common.h"

namespace mirast {

// min over dy in [lo,hi] of  1/2 C dy^2 + (B dxe) dy + 1/2 A dxe^2   (C > 0)
__device__ __forceinline__ float edge_min(float A, float B, float C, float rcpC, float dxe, float lo, float hi)
{
    const float t = fminf(hi, fmaxf(lo, -B * dxe * rcpC));
    return 0.5f * (A * dxe * dxe + C * t * t) + B * dxe * t;
}

// Whole-tile part of quadrant_mask below (same arithmetic, same decision): false exactly when quadrant_mask returns 0
// through one of its first exits, i.e. when no pixel of the 16x16 tile can receive alpha >= 1/255 from this Gaussian.
__device__ __forceinline__ bool tile_may_blend(float2 xy, float4 co, float tile_px, float tile_py)
{
    const float A = co.x, B = co.y, C = co.z, o = co.w;
    if (!(o >= (1.0f / 255.0f))) return false;
    if (!(A > 0.f && C > 0.f && A * C - B * B > 0.f)) return true;
    const float dxl = xy.x - (tile_px + 15.0f), dxh = xy.x - tile_px;
    const float dyl = xy.y - (tile_py + 15.0f), dyh = xy.y - tile_py;
    if (dxl <= 0.f && dxh >= 0.f && dyl <= 0.f && dyh >= 0.f) return true;
    const float tau = __logf(255.0f * o);
    const float rcpA = __builtin_amdgcn_rcpf(A), rcpC = __builtin_amdgcn_rcpf(C);
    const float e0 = edge_min(A, B, C, rcpC, dxl, dyl, dyh);
    const float e1 = edge_min(A, B, C, rcpC, dxh, dyl, dyh);
    const float e2 = edge_min(C, B, A, rcpA, dyl, dxl, dxh);
    const float e3 = edge_min(C, B, A, rcpA, dyh, dxl, dxh);
    const float qmin = fminf(fminf(e0, e1), fminf(e2, e3));
    const float mx = fmaxf(fabsf(dxl), fabsf(dxh)), my = fmaxf(fabsf(dyl), fabsf(dyh));
    const float mag = 0.5f * (A * mx * mx + C * my * my) + fabsf(B) * mx * my;
    return !(qmin > tau + 1e-5f * mag + 1e-4f);
}

// Lean lists: the part of a Gaussian's reference tile rect [rmin, rmax) in which tile_may_blend can return true at all.
// A tile is kept only if the minimum of q over it is <= tau + (its margin); every margin of the rect is below
// 1e-5 mag_max + 1e-4 with mag_max taken at the largest |d| the rect reaches (radius + 15 px), so every kept tile
// meets the ellipse { q <= tau_big }, hence its axis-aligned bounding box |dx| <= sqrt(2 tau_big C / det),
// |dy| <= sqrt(2 tau_big A / det).  The 3-sigma SQUARE of the reference is about twice that area on the benchmark scene.
// Slack (3x the margin, 0.1 % + 0.01 px on the extents) dwarfs the float rounding of both sides: a strict superset.
__device__ __forceinline__ void shrink_rect(float2 xy, float4 co, int rad, uint2& rmin, uint2& rmax)
{
    const float A = co.x, B = co.y, C = co.z, o = co.w;
    if (!(o >= (1.0f / 255.0f))) {  // tile_may_blend is false everywhere
        rmax = rmin;
        return;
    }
    const float det = A * C - B * B;
    if (!(A > 0.f && C > 0.f && det > 0.f)) return;  // not positive definite: no culling
    const float m = (float)rad + 16.0f;
    const float mag_max = (0.5f * (A + C) + fabsf(B)) * m * m;
    const float tau_big = __logf(255.0f * o) + 3e-5f * mag_max + 2e-4f;
    const float s = 2.0f * tau_big * __builtin_amdgcn_rcpf(det);
    const float ex = sqrtf(s * C) * 1.001f + 0.01f, ey = sqrtf(s * A) * 1.001f + 0.01f;
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

// Bit q (= qy*2 + qx) set  <=>  the 8x8 pixel quadrant q of the tile at (tile_px, tile_py) may receive
// alpha >= 1/255 from this Gaussian.  xy: pixel-space mean; co: conic (A,B,C) + opacity.
__device__ __forceinline__ uint32_t quadrant_mask(float2 xy, float4 co, float tile_px, float tile_py)
{
    const float A = co.x, B = co.y, C = co.z, o = co.w;
    if (!(o >= (1.0f / 255.0f))) return 0u;  // alpha <= opacity < 1/255 everywhere (also drops NaN opacity)
    if (!(A > 0.f && C > 0.f && A * C - B * B > 0.f)) return 0xFu;  // not positive definite: no culling
    const float tau = __logf(255.0f * o);
    const float rcpA = __builtin_amdgcn_rcpf(A), rcpC = __builtin_amdgcn_rcpf(C);
    {
        // whole-tile reject first: two thirds of the list entries of the benchmark scene stop here
        const float dxl = xy.x - (tile_px + 15.0f), dxh = xy.x - tile_px;
        const float dyl = xy.y - (tile_py + 15.0f), dyh = xy.y - tile_py;
        if (!(dxl <= 0.f && dxh >= 0.f && dyl <= 0.f && dyh >= 0.f)) {
            const float e0 = edge_min(A, B, C, rcpC, dxl, dyl, dyh);
            const float e1 = edge_min(A, B, C, rcpC, dxh, dyl, dyh);
            const float e2 = edge_min(C, B, A, rcpA, dyl, dxl, dxh);
            const float e3 = edge_min(C, B, A, rcpA, dyh, dxl, dxh);
            const float qmin = fminf(fminf(e0, e1), fminf(e2, e3));
            const float mx = fmaxf(fabsf(dxl), fabsf(dxh)), my = fmaxf(fabsf(dyl), fabsf(dyh));
            const float mag = 0.5f * (A * mx * mx + C * my * my) + fabsf(B) * mx * my;
            if (qmin > tau + 1e-5f * mag + 1e-4f) return 0u;
        }
    }
    uint32_t mask = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float x_lo = tile_px + (float)((q & 1) * 8), y_lo = tile_py + (float)((q >> 1) * 8);
        // d = mean - pixel, pixel in [x_lo, x_lo+7] x [y_lo, y_lo+7]
        const float dxl = xy.x - (x_lo + 7.0f), dxh = xy.x - x_lo;
        const float dyl = xy.y - (y_lo + 7.0f), dyh = xy.y - y_lo;
        float qmin;
        if (dxl <= 0.f && dxh >= 0.f && dyl <= 0.f && dyh >= 0.f) {
            qmin = 0.f;
        } else {
            const float e0 = edge_min(A, B, C, rcpC, dxl, dyl, dyh);
            const float e1 = edge_min(A, B, C, rcpC, dxh, dyl, dyh);
            const float e2 = edge_min(C, B, A, rcpA, dyl, dxl, dxh);
            const float e3 = edge_min(C, B, A, rcpA, dyh, dxl, dxh);
            qmin = fminf(fminf(e0, e1), fminf(e2, e3));
        }
        const float mx = fmaxf(fabsf(dxl), fabsf(dxh)), my = fmaxf(fabsf(dyl), fabsf(dyh));
        const float mag = 0.5f * (A * mx * mx + C * my * my) + fabsf(B) * mx * my;
        if (!(qmin > tau + 1e-5f * mag + 1e-4f)) mask |= 1u << q;  // NaN -> keep
    }
    return mask;
}

// Order-preserving compaction of one batch: returns this thread's slot (or -1) and the survivor count.
// s_wcount: LDS uint32[NW] (NW waves in the workgroup).  Contains one workgroup barrier.
template <int NW = 4>
__device__ __forceinline__ int compact_slot(bool survive, int wave, uint32_t* s_wcount, int& total)
{
    const uint64_t b = ballot64(survive);
    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
    if ((threadIdx.x & 63) == 0) s_wcount[wave] = (uint32_t)__builtin_popcountll(b);
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const uint32_t c = s_wcount[w];
        woff += w < wave ? c : 0u;
        tot += c;
    }
    total = (int)tot;
    return survive ? (int)(woff + below) : -1;
}

}  // namespace mirast
