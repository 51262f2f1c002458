// blend_fwd.h -- per-tile front-to-back alpha compositing (forward).
//
// Restates renderCUDA<C> forward (CF/cuda_rasterizer/forward.cu:264-385); with EXTRA==2 the DEPTH
// variant (DEPTH/cuda_rasterizer/forward.cu:308-309,363-365,384-385: mask + depth accumulators,
// no background term); with C==0, EXTRA==1 the mask-only render (DEPTH/.../forward.cu:390-498).
//
// gfx950 mapping: one 256-thread workgroup per 16x16 tile (the tile size is part of the integer
// contract), four wave64s each owning an 8x8 pixel quadrant so that "no lane of this wave is
// touched by this Gaussian" is a frequent, wave-uniform (scalar-branch) skip.  The per-tile list is
// staged 256 entries at a time through LDS: geometry (xy, conic+opacity) AND the C feature floats of
// every staged Gaussian, with coalesced 16-B loads (8 lanes cover one Gaussian's 128 B at C=32), so
// the inner loop reads features as LDS broadcasts instead of the reference's per-pixel global loads.
// Early termination is per wave (ballot) and per workgroup (__syncthreads_and).
#pragma once

#include "common.h"

namespace mirast {

constexpr int BATCH = 256;

template <int CE>
struct FeatStage {
    // Feature rows padded to a multiple of 4 floats so that a row is read with ds_read_b128.
    static constexpr int ROW = (CE + 3) & ~3;
};

template <int C, int EXTRA>
__global__ void __launch_bounds__(256) blend_fwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H,
    const float2* __restrict__ points_xy_image, const float* __restrict__ features,
    const float4* __restrict__ conic_opacity, const float* __restrict__ mask, const float* __restrict__ depths,
    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_consumed,
    const float* __restrict__ bg_color,
    float* __restrict__ out_color, float* __restrict__ out_mask, float* __restrict__ out_depth)
{
    constexpr int CE = C + EXTRA;            // accumulated values per pixel
    constexpr int ROW = FeatStage<CE>::ROW;  // LDS floats per staged Gaussian
    constexpr bool VEC_STAGE = (EXTRA == 0) && (C % 4 == 0) && (C >= 4);

    __shared__ int s_id[BATCH];
    __shared__ float2 s_xy[BATCH];
    __shared__ float4 s_co[BATCH];
    __shared__ float4 s_feat4[BATCH * ROW / 4];
    __shared__ int s_consumed;
    float* s_feat = reinterpret_cast<float*>(s_feat4);

    const int tid = threadIdx.x;
    if (tid == 0) s_consumed = 0;
    const int wave = tid >> 6, lane = tid & 63;
    const uint32_t horizontal_blocks = (W + TILE_X - 1) / TILE_X;
    const uint32_t tile = blockIdx.y * horizontal_blocks + blockIdx.x;
    // 8x8 quadrant per wave
    const uint32_t px = blockIdx.x * TILE_X + (wave & 1) * 8 + (lane & 7);
    const uint32_t py = blockIdx.y * TILE_Y + (wave >> 1) * 8 + (lane >> 3);
    const uint32_t pix_id = W * py + px;
    const float pixfx = (float)px, pixfy = (float)py;
    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    bool done = !inside;

    const uint2 range = ranges[tile];
    int toDo = range.y - range.x;
    const int rounds = (toDo + BATCH - 1) / BATCH;

    float T = 1.0f;
    uint32_t last_contributor = 0;
    int consumed = 0;  // wave-uniform: list entries this wave walked
    float acc[CE > 0 ? CE : 1];
#pragma unroll
    for (int ch = 0; ch < CE; ch++) acc[ch] = 0.f;

    for (int i = 0; i < rounds; i++, toDo -= BATCH) {
        // whole workgroup finished? (also the barrier that protects LDS reuse)
        if (__syncthreads_and(done)) break;

        const int progress = i * BATCH + tid;
        if (range.x + progress < range.y) {
            const int coll_id = point_list[range.x + progress];
            s_id[tid] = coll_id;
            s_xy[tid] = points_xy_image[coll_id];
            s_co[tid] = conic_opacity[coll_id];
            if constexpr (!VEC_STAGE) {
#pragma unroll
                for (int ch = 0; ch < C; ch++) s_feat[tid * ROW + ch] = features[(size_t)coll_id * C + ch];
                if constexpr (EXTRA >= 1) s_feat[tid * ROW + C] = mask[coll_id];
                if constexpr (EXTRA >= 2) s_feat[tid * ROW + C + 1] = depths[coll_id];
            }
        }
        __syncthreads();
        const int nb = toDo < BATCH ? toDo : BATCH;
        if constexpr (VEC_STAGE) {
            constexpr int F4 = C / 4;  // float4s per Gaussian
#pragma unroll
            for (int k = 0; k < F4; k++) {
                const int q = tid + BATCH * k;
                const int g = q / F4, part = q % F4;
                if (g < nb) {
                    const float4 v = reinterpret_cast<const float4*>(features + (size_t)s_id[g] * C)[part];
                    s_feat4[g * F4 + part] = v;
                }
            }
            __syncthreads();
        }

        for (int j = 0; j < nb; j++) {
            if (ballot64(!done) == 0) break;  // this wave has nothing left to do
            consumed = i * BATCH + j + 1;
            const float2 xy = s_xy[j];
            const float4 con_o = s_co[j];
            const float dx = xy.x - pixfx, dy = xy.y - pixfy;
            const float power = -0.5f * (con_o.x * dx * dx + con_o.z * dy * dy) - con_o.y * dx * dy;
            const float alpha = fminf(0.99f, con_o.w * __expf(power));
            const bool ok = !done && power <= 0.0f && alpha >= (1.0f / 255.0f);
            const float test_T = T * (1 - alpha);
            const bool stop = ok && test_T < 0.0001f;
            done = done || stop;
            const bool blend = ok && !stop;
            if (ballot64(blend) == 0) continue;  // wave-uniform: nobody in this 8x8 quadrant is touched
            const float w = blend ? alpha * T : 0.f;
#pragma unroll
            for (int ch = 0; ch < CE; ch++) acc[ch] = fmaf(s_feat[j * ROW + ch], w, acc[ch]);
            T = blend ? test_T : T;
            last_contributor = blend ? (uint32_t)(i * BATCH + j + 1) : last_contributor;
        }
    }

    if (lane == 0) atomicMax(&s_consumed, consumed);
    __syncthreads();
    if (tid == 0) tile_consumed[tile] = (uint32_t)s_consumed;
    if (inside) {
        final_T[pix_id] = T;
        n_contrib[pix_id] = last_contributor;
        const size_t HW = (size_t)H * W;
#pragma unroll
        for (int ch = 0; ch < C; ch++) out_color[ch * HW + pix_id] = acc[ch] + T * bg_color[ch];
        if constexpr (EXTRA >= 1) out_mask[pix_id] = acc[C];
        if constexpr (EXTRA >= 2) out_depth[pix_id] = acc[C + 1];
    }
}

}  // namespace mirast
