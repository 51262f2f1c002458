// knn_smooth.h -- fused KNN feature smoothing, forward and backward.
//
// Restates FeatureGaussianModel.get_smoothed_point_features (scene/gaussian_model_ff.py:338-364) together with the
// re-normalisation the renderer applies to its result (gaussian_renderer/__init__.py:362-363):
//     n_j    = F_j / max(|F_j|, 1e-12)                        (torch.nn.functional.normalize, p = 2)
//     m_i    = mean over the selected neighbour columns s of  n_{idx[i][s]}
//     out_i  = m_i / (|m_i| + 1e-9)                           (only when normalize_out)
// The reference materialises n (P x C), the gathered (P x k x C) tensor and m; its backward is an index_put with
// accumulation.  Here the forward is one gather pass (k rows of 4C bytes per Gaussian, norms recomputed on the fly),
// and the backward is two gather passes and no atomics: pass A recomputes m_i and writes dL/dm_i; pass B walks the
// INVERSE neighbour lists (built once per neighbour map, like the map itself) and turns the sum of the dL/dm rows
// that reference Gaussian j into dL/dF_j.  Layout: C/4 lanes per feature row (float4 each), 64/(C/4) rows per wave.
#pragma once

#include "common.h"

namespace mirast {

template <int LPR>
__device__ __forceinline__ float row_sum(float v)
{
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// m_i for the row this lane group owns (before the mean's 1/k), as float4 per lane
template <int C>
__device__ __forceinline__ float4 gather_mean(const float4* __restrict__ F4, const int* __restrict__ idx_row, int K,
                                              uint32_t sel_mask, int part)
{
    constexpr int LPR = C / 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < K; s++) {
        if (!((sel_mask >> s) & 1u)) continue;
        const int j = idx_row[s];
        const float4 v = F4[(size_t)j * LPR + part];
        const float ss = row_sum<LPR>(dot4(v, v));
        const float rn = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
        acc.x += v.x * rn;
        acc.y += v.y * rn;
        acc.z += v.z * rn;
        acc.w += v.w * rn;
    }
    return acc;
}

template <int C>
__global__ void __launch_bounds__(256) knn_smooth_fwd_kernel(int P, int K, const int* __restrict__ knn_idx,
                                                             uint32_t sel_mask, float inv_k, const float* __restrict__ F,
                                                             float* __restrict__ out, int normalize_out)
{
    constexpr int LPR = C / 4;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int row = gid / LPR, part = gid % LPR;
    if (row >= P) return;  // whole lane groups leave together (256 % LPR == 0)
    float4 m = gather_mean<C>(reinterpret_cast<const float4*>(F), knn_idx + (size_t)row * K, K, sel_mask, part);
    m.x *= inv_k;
    m.y *= inv_k;
    m.z *= inv_k;
    m.w *= inv_k;
    if (normalize_out) {
        const float r = sqrtf(row_sum<LPR>(dot4(m, m)));
        const float s = 1.0f / (r + 1e-9f);
        m.x *= s;
        m.y *= s;
        m.z *= s;
        m.w *= s;
    }
    reinterpret_cast<float4*>(out)[(size_t)row * LPR + part] = m;
}

// Backward pass A: dL/dm_i (already scaled by 1/k, i.e. the gradient every selected neighbour's n receives).
template <int C>
__global__ void __launch_bounds__(256) knn_smooth_bwd_mean_kernel(int P, int K, const int* __restrict__ knn_idx,
                                                                  uint32_t sel_mask, float inv_k,
                                                                  const float* __restrict__ F,
                                                                  const float* __restrict__ dL_dout,
                                                                  float* __restrict__ dmean, int normalize_out)
{
    constexpr int LPR = C / 4;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int row = gid / LPR, part = gid % LPR;
    if (row >= P) return;
    float4 g = reinterpret_cast<const float4*>(dL_dout)[(size_t)row * LPR + part];
    if (normalize_out) {
        float4 m = gather_mean<C>(reinterpret_cast<const float4*>(F), knn_idx + (size_t)row * K, K, sel_mask, part);
        m.x *= inv_k;
        m.y *= inv_k;
        m.z *= inv_k;
        m.w *= inv_k;
        // out = m / (r + eps):  dm = g / (r + eps) - m (m . g) / (r (r + eps)^2)
        const float r = sqrtf(row_sum<LPR>(dot4(m, m)));
        const float mg = row_sum<LPR>(dot4(m, g));
        const float a = 1.0f / (r + 1e-9f);
        const float b = r > 0.f ? mg * a * a / r : 0.f;
        g.x = g.x * a - m.x * b;
        g.y = g.y * a - m.y * b;
        g.z = g.z * a - m.z * b;
        g.w = g.w * a - m.w * b;
    }
    g.x *= inv_k;
    g.y *= inv_k;
    g.z *= inv_k;
    g.w *= inv_k;
    reinterpret_cast<float4*>(dmean)[(size_t)row * LPR + part] = g;
}

// Backward pass B: dL/dF_j from the dL/dm rows of the Gaussians that selected j as a neighbour.
// inv_offsets[P+1], inv_entries[P*K]: entry = (i << 5) | column, grouped by the referenced Gaussian j.
template <int C>
__global__ void __launch_bounds__(256) knn_smooth_bwd_feat_kernel(int P, const int* __restrict__ inv_offsets,
                                                                  const uint32_t* __restrict__ inv_entries,
                                                                  uint32_t sel_mask, const float* __restrict__ F,
                                                                  const float* __restrict__ dmean,
                                                                  float* __restrict__ dL_dF)
{
    constexpr int LPR = C / 4;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int row = gid / LPR, part = gid % LPR;
    if (row >= P) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const int e0 = inv_offsets[row], e1 = inv_offsets[row + 1];
    for (int e = e0; e < e1; e++) {
        const uint32_t ent = inv_entries[e];
        if (!((sel_mask >> (ent & 31u)) & 1u)) continue;
        const float4 d = reinterpret_cast<const float4*>(dmean)[(size_t)(ent >> 5) * LPR + part];
        s.x += d.x;
        s.y += d.y;
        s.z += d.z;
        s.w += d.w;
    }
    // n = F / max(|F|, eps):  dF = (s - n (n . s)) / |F|   (|F| >= eps);   dF = s / eps   (|F| < eps: n = F / eps)
    const float4 f = reinterpret_cast<const float4*>(F)[(size_t)row * LPR + part];
    const float nrm = sqrtf(row_sum<LPR>(dot4(f, f)));
    float4 o;
    if (nrm >= 1e-12f) {
        const float rn = 1.0f / nrm;
        const float4 n = make_float4(f.x * rn, f.y * rn, f.z * rn, f.w * rn);
        const float ns = row_sum<LPR>(dot4(n, s));
        o = make_float4((s.x - n.x * ns) * rn, (s.y - n.y * ns) * rn, (s.z - n.z * ns) * rn, (s.w - n.w * ns) * rn);
    } else {
        o = make_float4(s.x * 1e12f, s.y * 1e12f, s.z * 1e12f, s.w * 1e12f);
    }
    reinterpret_cast<float4*>(dL_dF)[(size_t)row * LPR + part] = o;
}

}  // namespace mirast
