// contrastive.h -- the contrastive-loss front end of SAGA's feature training (include/mi_contrastive.h; SURVEY.md 8(f) row 3;
// reference: train_contrastive_feature.py:234-254).  Two kinds of work in one launch:
//   * dense: every pixel's L2 norm over the C feature planes (the regulariser rendered_features.norm(dim=0).mean()) -- a
//     streaming pass over the (C, h, w) image, 4 pixels per thread, one 16-byte load per plane and thread, all C loads of a
//     thread independent (HBM-bound: 4 C bytes per pixel);
//   * rays: one wave per sampled ray -- four bilinear taps per channel (the weights of F.interpolate(mode='bilinear',
//     align_corners=False): area_pixel_compute_source_index), times the N scale gates, L2-normalised over the channels.
// The backward streams the image once more (reads f and 1/||f||, writes the whole gradient image), then a second launch adds
// the rays' tap gradients with float atomics (4 S C of them) and reduces the gate gradients per workgroup before its atomics.
#pragma once

#include "../../include/mi_contrastive.h"
#include "common.h"

namespace mirast {

constexpr int CT_THREADS = 256;
#ifndef CT_UNROLL
#define CT_UNROLL 8      // channel planes whose loads are in flight per thread of the streaming passes
#endif
constexpr int CT_NORM_SLOTS = MI_CONTRASTIVE_NORM_SLOTS;
constexpr int CT_MAX_CPL = 4;       // channels per lane of a ray's wave: C <= 256   // partial sums of the regulariser: same-address atomics serialise at ~22 ns each, and there are
                                    // ~2000 workgroups at 1080p -- 64 slots, 128 bytes apart, summed by the caller

// Source index and upper weight of output index `dst` for bilinear interpolation with align_corners = False
// (ATen/native/UpSample.h area_pixel_compute_source_index: scale = in / out, src = max(0, (dst + 0.5) scale - 0.5)).
__device__ __forceinline__ void bilinear_tap(int dst, int n_out, int n_in, int& i0, int& i1, float& lam)
{
    const float scale = (float)n_in / (float)n_out;
    const float src = fmaxf((dst + 0.5f) * scale - 0.5f, 0.0f);
    i0 = min((int)src, n_in - 1);
    i1 = min(i0 + 1, n_in - 1);
    lam = src - (float)i0;
}

template <int VEC>
__global__ void __launch_bounds__(CT_THREADS) contrastive_fwd_kernel(
    int C, int h, int w, const float* __restrict__ rendered, int H, int W, int S, const int* __restrict__ ray_yx, int N,
    const float* __restrict__ gates, float* __restrict__ out, float* __restrict__ ray_feat, float* __restrict__ inv_len,
    float* __restrict__ inv_norm, double* __restrict__ norm_sum, uint32_t dense_blocks)
{
    const size_t HW = (size_t)h * w;
    const uint32_t ray_blocks = gridDim.x - dense_blocks;
    // The ray workgroups come FIRST in the grid: each is a chain of dependent gathers and wave reductions (microseconds of latency,
    // no bandwidth) that hides under the streaming workgroups behind it -- at the end of the grid it would be the kernel's tail.
    if (blockIdx.x >= ray_blocks) {
        // ---- dense: VEC consecutive pixels per thread
        __shared__ float s_part[CT_THREADS / 64];
        const uint32_t db = blockIdx.x - ray_blocks;
        const size_t p0 = ((size_t)db * CT_THREADS + threadIdx.x) * VEC;
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; k++) acc[k] = 0.f;
        if (p0 < HW) {
            if constexpr (VEC == 4) {
#pragma unroll CT_UNROLL
                for (int c = 0; c < C; c++) {
                    const float4 v = *reinterpret_cast<const float4*>(rendered + (size_t)c * HW + p0);
                    acc[0] = fmaf(v.x, v.x, acc[0]);
                    acc[1] = fmaf(v.y, v.y, acc[1]);
                    acc[2] = fmaf(v.z, v.z, acc[2]);
                    acc[3] = fmaf(v.w, v.w, acc[3]);
                }
            } else {
#pragma unroll CT_UNROLL
                for (int c = 0; c < C; c++) {
                    const float v = rendered[(size_t)c * HW + p0];
                    acc[0] = fmaf(v, v, acc[0]);
                }
            }
        }
        float local = 0.f;
        float inv[VEC];
#pragma unroll
        for (int k = 0; k < VEC; k++) {
            const float n = sqrtf(acc[k]);
            local += n;
            inv[k] = n > 0.f ? 1.0f / n : 0.f;
        }
        if (p0 < HW) {
            if constexpr (VEC == 4) *reinterpret_cast<float4*>(inv_norm + p0) = make_float4(inv[0], inv[1], inv[2], inv[3]);
            else inv_norm[p0] = inv[0];
        }
        local = wave_sum(local);
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = local;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < CT_THREADS / 64; k++) t += (double)s_part[k];
            atomicAdd(&norm_sum[(db % CT_NORM_SLOTS) * 16], t);
        }
        return;
    }
    // ---- rays: one wave per ray; lane l holds channels l, l + 64, ... (at most CT_MAX_CPL of them) in registers
    const int s = (int)blockIdx.x * (CT_THREADS / 64) + (int)(threadIdx.x >> 6);
    if (s >= S) return;
    const int lane = threadIdx.x & 63;
    int y0, y1, x0, x1;
    float ly, lx;
    bilinear_tap(ray_yx[2 * s], H, h, y0, y1, ly);
    bilinear_tap(ray_yx[2 * s + 1], W, w, x0, x1, lx);
    float ray[CT_MAX_CPL];
#pragma unroll
    for (int k = 0; k < CT_MAX_CPL; k++) {
        const int c = lane + 64 * k;
        ray[k] = 0.f;
        if (c < C) {
            const float* f = rendered + (size_t)c * HW;
            const float top = f[(size_t)y0 * w + x0] * (1.f - lx) + f[(size_t)y0 * w + x1] * lx;
            const float bot = f[(size_t)y1 * w + x0] * (1.f - lx) + f[(size_t)y1 * w + x1] * lx;
            ray[k] = top * (1.f - ly) + bot * ly;
            ray_feat[(size_t)s * C + c] = ray[k];
        }
    }
    for (int n = 0; n < N; n++) {
        float v[CT_MAX_CPL], ss = 0.f;
#pragma unroll
        for (int k = 0; k < CT_MAX_CPL; k++) {
            const int c = lane + 64 * k;
            v[k] = c < C ? ray[k] * gates[(size_t)n * C + c] : 0.f;
            ss = fmaf(v[k], v[k], ss);
        }
        ss = wave_sum(ss);
        const float il = 1.0f / fmaxf(sqrtf(ss), 1e-12f);   // F.normalize: x / max(||x||, eps), eps = 1e-12
        if (lane == 0) inv_len[(size_t)n * S + s] = il;
#pragma unroll
        for (int k = 0; k < CT_MAX_CPL; k++) {
            const int c = lane + 64 * k;
            if (c < C) out[((size_t)n * S + s) * C + c] = v[k] * il;
        }
    }
}

template <int VEC>
__global__ void __launch_bounds__(CT_THREADS) contrastive_bwd_dense_kernel(
    int C, int h, int w, const float* __restrict__ rendered, const float* __restrict__ inv_norm,
    const float* __restrict__ g_norm, float* __restrict__ dL_drendered)
{
    const size_t HW = (size_t)h * w;
    const size_t p0 = ((size_t)blockIdx.x * CT_THREADS + threadIdx.x) * VEC;
    if (p0 >= HW) return;
    const float g = g_norm ? g_norm[0] / (float)HW : 0.f;   // d mean / d norm_p = 1 / (h w)
    if constexpr (VEC == 4) {
        float4 k = *reinterpret_cast<const float4*>(inv_norm + p0);
        k.x *= g, k.y *= g, k.z *= g, k.w *= g;
#pragma unroll CT_UNROLL
        for (int c = 0; c < C; c++) {
            const float4 v = *reinterpret_cast<const float4*>(rendered + (size_t)c * HW + p0);
            *reinterpret_cast<float4*>(dL_drendered + (size_t)c * HW + p0) = make_float4(v.x * k.x, v.y * k.y, v.z * k.z, v.w * k.w);
        }
    } else {
        const float k = inv_norm[p0] * g;
#pragma unroll CT_UNROLL
        for (int c = 0; c < C; c++) dL_drendered[(size_t)c * HW + p0] = rendered[(size_t)c * HW + p0] * k;
    }
}

// One wave per ray; a workgroup's four rays reduce their gate gradients in LDS before the atomics.  The gates are walked in
// batches of CT_GB: all of a batch's loads are issued before the first reduction (one memory latency per batch, not per gate).
constexpr int CT_GB = 5;
__global__ void __launch_bounds__(CT_THREADS) contrastive_bwd_rays_kernel(
    int C, int h, int w, int H, int W, int S, const int* __restrict__ ray_yx, int N, const float* __restrict__ gates,
    const float* __restrict__ out, const float* __restrict__ ray_feat, const float* __restrict__ inv_len,
    const float* __restrict__ dL_dout, float* __restrict__ dL_drendered, float* __restrict__ dL_dgates)
{
    extern __shared__ float s_dg[];   // [waves][N * C]
    const size_t HW = (size_t)h * w;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = (int)blockIdx.x * (CT_THREADS / 64) + wv;
    float* my_dg = s_dg + (size_t)wv * N * C;
    for (int i = lane; i < N * C; i += 64) my_dg[i] = 0.f;
    if (s < S) {
        int y0, y1, x0, x1;
        float ly, lx;
        bilinear_tap(ray_yx[2 * s], H, h, y0, y1, ly);
        bilinear_tap(ray_yx[2 * s + 1], W, w, x0, x1, lx);
        float ray[CT_MAX_CPL], dray[CT_MAX_CPL];
#pragma unroll
        for (int k = 0; k < CT_MAX_CPL; k++) {
            const int c = lane + 64 * k;
            ray[k] = c < C ? ray_feat[(size_t)s * C + c] : 0.f;
            dray[k] = 0.f;
        }
        for (int n0 = 0; n0 < N; n0 += CT_GB) {
            float g[CT_GB][CT_MAX_CPL], o[CT_GB][CT_MAX_CPL], gt[CT_GB][CT_MAX_CPL], il[CT_GB], dot[CT_GB];
#pragma unroll
            for (int b = 0; b < CT_GB; b++) {
                const int n = min(n0 + b, N - 1);
                il[b] = inv_len[(size_t)n * S + s];
#pragma unroll
                for (int k = 0; k < CT_MAX_CPL; k++) {
                    const int c = lane + 64 * k;
                    const size_t i = ((size_t)n * S + s) * C + c;
                    g[b][k] = c < C ? dL_dout[i] : 0.f;
                    o[b][k] = c < C ? out[i] : 0.f;
                    gt[b][k] = c < C ? gates[(size_t)n * C + c] : 0.f;
                }
            }
#pragma unroll
            for (int b = 0; b < CT_GB; b++) {
                float d = 0.f;
#pragma unroll
                for (int k = 0; k < CT_MAX_CPL; k++) d = fmaf(g[b][k], o[b][k], d);
                dot[b] = wave_sum(d);   // d normalize: (g - o <g, o>) / len over the channels of (n, s)
            }
#pragma unroll
            for (int b = 0; b < CT_GB; b++) {
                if (n0 + b >= N) break;
#pragma unroll
                for (int k = 0; k < CT_MAX_CPL; k++) {
                    const int c = lane + 64 * k;
                    if (c < C) {
                        const float dsc = (g[b][k] - o[b][k] * dot[b]) * il[b];   // d (ray * gate)
                        dray[k] = fmaf(dsc, gt[b][k], dray[k]);
                        my_dg[(n0 + b) * C + c] = dsc * ray[k];
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < CT_MAX_CPL; k++) {
            const int c = lane + 64 * k;
            if (c < C) {
                float* gp = dL_drendered + (size_t)c * HW;
                atomicAdd(&gp[(size_t)y0 * w + x0], dray[k] * (1.f - ly) * (1.f - lx));
                atomicAdd(&gp[(size_t)y0 * w + x1], dray[k] * (1.f - ly) * lx);
                atomicAdd(&gp[(size_t)y1 * w + x0], dray[k] * ly * (1.f - lx));
                atomicAdd(&gp[(size_t)y1 * w + x1], dray[k] * ly * lx);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N * C; i += CT_THREADS) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < CT_THREADS / 64; k++) t += s_dg[(size_t)k * N * C + i];
        if (t != 0.f) atomicAdd(&dL_dgates[i], t);
    }
}

}  // namespace mirast
