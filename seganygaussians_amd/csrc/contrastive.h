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

#include "common.h"

namespace mirast {

constexpr int CT_THREADS = 256;

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
    if (blockIdx.x < dense_blocks) {
        // ---- dense: VEC consecutive pixels per thread
        __shared__ float s_part[CT_THREADS / 64];
        const size_t p0 = ((size_t)blockIdx.x * CT_THREADS + threadIdx.x) * VEC;
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; k++) acc[k] = 0.f;
        if (p0 < HW) {
            if constexpr (VEC == 4) {
#pragma unroll 8
                for (int c = 0; c < C; c++) {
                    const float4 v = *reinterpret_cast<const float4*>(rendered + (size_t)c * HW + p0);
                    acc[0] = fmaf(v.x, v.x, acc[0]);
                    acc[1] = fmaf(v.y, v.y, acc[1]);
                    acc[2] = fmaf(v.z, v.z, acc[2]);
                    acc[3] = fmaf(v.w, v.w, acc[3]);
                }
            } else {
#pragma unroll 8
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
            atomicAdd(norm_sum, t);
        }
        return;
    }
    // ---- rays: one wave per ray
    const int s = (int)(blockIdx.x - dense_blocks) * (CT_THREADS / 64) + (int)(threadIdx.x >> 6);
    if (s >= S) return;
    const int lane = threadIdx.x & 63;
    int y0, y1, x0, x1;
    float ly, lx;
    bilinear_tap(ray_yx[2 * s], H, h, y0, y1, ly);
    bilinear_tap(ray_yx[2 * s + 1], W, w, x0, x1, lx);
    for (int c0 = 0; c0 < C; c0 += 64) {   // (C <= 64: one trip)
        const int c = c0 + lane;
        if (c < C) {
            const float* f = rendered + (size_t)c * HW;
            const float top = f[(size_t)y0 * w + x0] * (1.f - lx) + f[(size_t)y0 * w + x1] * lx;
            const float bot = f[(size_t)y1 * w + x0] * (1.f - lx) + f[(size_t)y1 * w + x1] * lx;
            ray_feat[(size_t)s * C + c] = top * (1.f - ly) + bot * ly;
        }
    }
    for (int n = 0; n < N; n++) {
        float ss = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float v = ray_feat[(size_t)s * C + c] * gates[(size_t)n * C + c];   // (own store above: same lane, same address)
            ss = fmaf(v, v, ss);
        }
        ss = wave_sum(ss);
        const float il = 1.0f / fmaxf(sqrtf(ss), 1e-12f);   // F.normalize: x / max(||x||, eps), eps = 1e-12
        if (lane == 0) inv_len[(size_t)n * S + s] = il;
        for (int c = lane; c < C; c += 64)
            out[((size_t)n * S + s) * C + c] = ray_feat[(size_t)s * C + c] * gates[(size_t)n * C + c] * il;
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
#pragma unroll 8
        for (int c = 0; c < C; c++) {
            const float4 v = *reinterpret_cast<const float4*>(rendered + (size_t)c * HW + p0);
            *reinterpret_cast<float4*>(dL_drendered + (size_t)c * HW + p0) = make_float4(v.x * k.x, v.y * k.y, v.z * k.z, v.w * k.w);
        }
    } else {
        const float k = inv_norm[p0] * g;
#pragma unroll 8
        for (int c = 0; c < C; c++) dL_drendered[(size_t)c * HW + p0] = rendered[(size_t)c * HW + p0] * k;
    }
}

// One wave per ray; a workgroup's four rays reduce their gate gradients in LDS before the atomics.
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
        for (int c0 = 0; c0 < C; c0 += 64) {
            const int c = c0 + lane;
            float dray = 0.f;
            for (int n = 0; n < N; n++) {
                // d normalize: (g - o <g, o>) / len over the channels of (n, s)
                float dot = 0.f;
                for (int cc = lane; cc < C; cc += 64) {
                    const size_t i = ((size_t)n * S + s) * C + cc;
                    dot = fmaf(dL_dout[i], out[i], dot);
                }
                dot = wave_sum(dot);
                if (c < C) {
                    const size_t i = ((size_t)n * S + s) * C + c;
                    const float dsc = (dL_dout[i] - out[i] * dot) * inv_len[(size_t)n * S + s];   // d (ray * gate)
                    dray = fmaf(dsc, gates[(size_t)n * C + c], dray);
                    my_dg[n * C + c] = dsc * ray_feat[(size_t)s * C + c];
                }
            }
            if (c < C) {
                float* g = dL_drendered + (size_t)c * HW;
                atomicAdd(&g[(size_t)y0 * w + x0], dray * (1.f - ly) * (1.f - lx));
                atomicAdd(&g[(size_t)y0 * w + x1], dray * (1.f - ly) * lx);
                atomicAdd(&g[(size_t)y1 * w + x0], dray * ly * (1.f - lx));
                atomicAdd(&g[(size_t)y1 * w + x1], dray * ly * lx);
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
