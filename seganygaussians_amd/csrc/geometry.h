// geometry.h -- per-Gaussian kernels: forward preprocess, frustum marking, and the fused
// backward of the geometry path (cov2D backward + projection + SH + cov3D backward).
//
// One thread per Gaussian, 256-thread workgroups, streaming: ~100 B read / ~60 B written per
// Gaussian, HBM-bound.  AoS float3 inputs are read as three dword loads whose wave footprint is one
// contiguous 768-B span, so every fetched line is fully used.
#pragma once

#include "binning.h"
#include "common.h"

namespace mirast {

// CF/cuda_rasterizer/auxiliary.h:21-39
__device__ constexpr float SH_C0 = 0.28209479177387814f;
__device__ constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                       -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                       0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                       -0.5900435899266435f};

// CF/cuda_rasterizer/forward.cu:121-155 (quaternion is NOT re-normalised, :130)
__device__ __forceinline__ void computeCov3D(const float3 scale, float mod, const float4 rot, float* cov3D)
{
    Mat3 S = {{{1.0f, 0, 0}, {0, 1.0f, 0}, {0, 0, 1.0f}}};
    S.m[0][0] = mod * scale.x;
    S.m[1][1] = mod * scale.y;
    S.m[2][2] = mod * scale.z;
    float r = rot.x, x = rot.y, y = rot.z, z = rot.w;
    Mat3 R = {{{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
               {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
               {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}}};
    Mat3 M = m3mul(S, R);
    Mat3 Sigma = m3mul(m3transpose(M), M);
    cov3D[0] = Sigma.m[0][0];
    cov3D[1] = Sigma.m[0][1];
    cov3D[2] = Sigma.m[0][2];
    cov3D[3] = Sigma.m[1][1];
    cov3D[4] = Sigma.m[1][2];
    cov3D[5] = Sigma.m[2][2];
}

struct Cov2DCtx {
    float3 t;
    float txtz, tytz;
    Mat3 T, Vrk, W, cov;
};

// Shared by the forward EWA projection (CF/cuda_rasterizer/forward.cu:77-116) and its backward
// (CF/cuda_rasterizer/backward.cu:161-196).
__device__ __forceinline__ void cov2d_common(const float3& mean, const ViewParams& vp, const float* cov3D, Cov2DCtx& c)
{
    float3 t = transformPoint4x3(mean, vp.view);
    const float limx = 1.3f * vp.tan_fovx;
    const float limy = 1.3f * vp.tan_fovy;
    c.txtz = t.x / t.z;
    c.tytz = t.y / t.z;
    t.x = fminf(limx, fmaxf(-limx, c.txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, c.tytz)) * t.z;
    Mat3 J = {{{vp.focal_x / t.z, 0.0f, -(vp.focal_x * t.x) / (t.z * t.z)},
               {0.0f, vp.focal_y / t.z, -(vp.focal_y * t.y) / (t.z * t.z)},
               {0, 0, 0}}};
    c.W = {{{vp.view[0], vp.view[4], vp.view[8]}, {vp.view[1], vp.view[5], vp.view[9]}, {vp.view[2], vp.view[6], vp.view[10]}}};
    c.T = m3mul(c.W, J);
    c.Vrk = {{{cov3D[0], cov3D[1], cov3D[2]}, {cov3D[1], cov3D[3], cov3D[4]}, {cov3D[2], cov3D[4], cov3D[5]}}};
    c.cov = m3mul(m3mul(m3transpose(c.T), m3transpose(c.Vrk)), c.T);
    c.cov.m[0][0] += 0.3f;
    c.cov.m[1][1] += 0.3f;
    c.t = t;
}

// CF/cuda_rasterizer/forward.cu:23-74
// `sh`: the Gaussian's (max_coeffs, 3) coefficient row -- in the forward pass a row of the workgroup's LDS copy (below).
__device__ __forceinline__ float3 computeColorFromSH(int idx, int deg, const float3 pos, const ViewParams& vp,
                                                     const float* sh, uint8_t* clamped)
{
    float3 dir = make_float3(pos.x - vp.campos[0], pos.y - vp.campos[1], pos.z - vp.campos[2]);
    float len = sqrtf(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
    dir.x = dir.x / len;
    dir.y = dir.y / len;
    dir.z = dir.z / len;
    const float x = dir.x, y = dir.y, z = dir.z;
    float res[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
#define SH(k) sh[(k) * 3 + c]
        float result = SH_C0 * SH(0);
        if (deg > 0) {
            result = result - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z;
                float xy = x * y, yz = y * z, xz = x * z;
                result = result + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) + SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) +
                         SH_C2[3] * xz * SH(7) + SH_C2[4] * (xx - yy) * SH(8);
                if (deg > 2) {
                    result = result + SH_C3[0] * y * (3.0f * xx - yy) * SH(9) + SH_C3[1] * xy * z * SH(10) +
                             SH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
                             SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                             SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) + SH_C3[5] * z * (xx - yy) * SH(14) +
                             SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
                }
            }
        }
#undef SH
        result += 0.5f;
        clamped[3 * idx + c] = (result < 0);
        res[c] = fmaxf(result, 0.0f);
    }
    return make_float3(res[0], res[1], res[2]);
}

// Forward preprocess: CF/cuda_rasterizer/forward.cu:159-259 (+ in_frustum, auxiliary.h:139-164).
// `culled_prefiltered` is incremented when prefiltered is set and a point is culled (the reference
// printf+__trap()s the whole context there; we report an error instead).
// gfx950 additions (binning.h): writes the 32-bit depth sort key (0xFFFFFFFF for culled Gaussians) and accumulates R = sum of tiles_touched of the visible Gaussians.
__global__ void __launch_bounds__(256) preprocess_fwd_kernel(
    int P, int D, int M, const float* __restrict__ orig_points, const float* __restrict__ scales,
    const float* __restrict__ rotations, const float* __restrict__ opacities, const float* __restrict__ shs,
    uint8_t* __restrict__ clamped, const float* __restrict__ cov3D_precomp, int colors_given, ViewParams vp,
    int* __restrict__ radii, float2* __restrict__ points_xy_image, float* __restrict__ depths,
    float* __restrict__ cov3Ds, float* __restrict__ rgb, float4* __restrict__ conic_opacity,
    uint32_t* __restrict__ tiles_touched, uint32_t* __restrict__ depth_key, BlendRec* __restrict__ index_rec,
    int* __restrict__ r_slots, int prefiltered, int* __restrict__ culled_prefiltered, unsigned long long* __restrict__ band_bits,
    uint32_t band_h, uint32_t nbands, uint32_t* __restrict__ tile_total, uint32_t ntiles)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ uint32_t s_band[MAX_BANDS];   // Gaussians of this workgroup per band of tile rows (binning.h: lean count / emit passes)
    if (threadIdx.x < MAX_BANDS) s_band[threadIdx.x] = 0u;
    if ((uint32_t)idx < ntiles) tile_total[idx] = 0u;   // per-tile entry counts: the count pass adds to them (binning.h); P < tiles: host memset
    // SH colours (forward.cu:23-74 reads 3 M floats per Gaussian, 192 bytes at degree 3): a thread walking its own row makes
    // every load instruction of the wave touch 64 rows 192 bytes apart.  The workgroup's rows are one contiguous block of
    // memory: staged into LDS with coalesced 16-byte loads (culled Gaussians included: 25 % more bytes, all of them streamed),
    // read back per thread at a row stride padded to 3 M + 4 floats (b128 reads of 64 rows then cover all banks evenly).
    extern __shared__ float s_sh[];
    const int sh_row = 3 * M + 4;
    if (!colors_given) {
        const int rows = min((int)blockDim.x, P - (int)(blockIdx.x * blockDim.x));
        const size_t base = (size_t)blockIdx.x * blockDim.x * 3 * M;
        const int nfl = rows * 3 * M;
        if ((3 * M) % 4 == 0 && (reinterpret_cast<uintptr_t>(shs) & 15) == 0) {
            for (int i = 4 * (int)threadIdx.x; i < nfl; i += 4 * (int)blockDim.x) {
                const float4 v = *reinterpret_cast<const float4*>(shs + base + i);
                *reinterpret_cast<float4*>(&s_sh[(i / (3 * M)) * sh_row + i % (3 * M)]) = v;
            }
        } else {
            for (int i = (int)threadIdx.x; i < nfl; i += (int)blockDim.x) s_sh[(i / (3 * M)) * sh_row + i % (3 * M)] = shs[base + i];
        }
        __syncthreads();
    }
    int my_radii = 0;
    uint32_t my_tiles = 0;
    uint32_t my_key = 0xFFFFFFFFu;
    uint2 rect_min = make_uint2(0, 0), rect_max = make_uint2(0, 0);
    if (idx < P) do {
        const float3 p_orig = make_float3(orig_points[3 * idx], orig_points[3 * idx + 1], orig_points[3 * idx + 2]);
        const float3 p_view = transformPoint4x3(p_orig, vp.view);
        if (p_view.z <= 0.2f) {  // auxiliary.h:154
            if (prefiltered) atomicAdd(culled_prefiltered, 1);
            break;
        }
        const float4 p_hom = transformPoint4x4(p_orig, vp.proj);
        const float p_w = 1.0f / (p_hom.w + 0.0000001f);
        const float3 p_proj = make_float3(p_hom.x * p_w, p_hom.y * p_w, p_hom.z * p_w);

        float cov3D[6];
        if (cov3D_precomp != nullptr) {
#pragma unroll
            for (int i = 0; i < 6; i++) cov3D[i] = cov3D_precomp[(size_t)idx * 6 + i];
        } else {
            const float3 s = make_float3(scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]);
            const float4 q = reinterpret_cast<const float4*>(rotations)[idx];
            computeCov3D(s, vp.scale_modifier, q, cov3D);
#pragma unroll
            for (int i = 0; i < 6; i++) cov3Ds[(size_t)idx * 6 + i] = cov3D[i];
        }
        Cov2DCtx c;
        cov2d_common(p_orig, vp, cov3D, c);
        const float3 cov = make_float3(c.cov.m[0][0], c.cov.m[0][1], c.cov.m[1][1]);
        const float det = (cov.x * cov.z - cov.y * cov.y);
        if (det == 0.0f) break;
        const float det_inv = 1.f / det;
        const float3 conic = make_float3(cov.z * det_inv, -cov.y * det_inv, cov.x * det_inv);
        const float mid = 0.5f * (cov.x + cov.z);
        const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        const float2 point_image = make_float2(ndc2Pix(p_proj.x, vp.W), ndc2Pix(p_proj.y, vp.H));
        uint2 rmin, rmax;
        getRect(point_image.x, point_image.y, f2i(my_radius), rmin, rmax, vp.grid_x, vp.grid_y);
        if ((rmax.x - rmin.x) * (rmax.y - rmin.y) == 0) break;

        if (!colors_given) {
            const float3 col = computeColorFromSH(idx, D, p_orig, vp, s_sh + threadIdx.x * sh_row, clamped);
            rgb[idx * 3 + 0] = col.x;
            rgb[idx * 3 + 1] = col.y;
            rgb[idx * 3 + 2] = col.z;
        }
        depths[idx] = p_view.z;
        my_radii = f2i(my_radius);
        points_xy_image[idx] = point_image;
        conic_opacity[idx] = make_float4(conic.x, conic.y, conic.z, opacities[idx]);
        my_tiles = (rmax.y - rmin.y) * (rmax.x - rmin.x);
        // A Gaussian whose radius converts to <= 0 is "invisible" for every later stage (radii > 0 tests,
        // rasterizer_impl.cu:84, backward.cu:156,369) although tiles_touched is nonzero; keep that.
        if (my_radii > 0) {
            // duplicateWithKeys recomputes the rect from the stored int radius (rasterizer_impl.cu:91)
            getRect(point_image.x, point_image.y, my_radii, rect_min, rect_max, vp.grid_x, vp.grid_y);
            my_key = __float_as_uint(p_view.z);
            // everything the binning stages need of this Gaussian in one 32-byte record (depth_sort.h gathers it once)
            BlendRec rec;
            rec.xy = point_image;
            rec.id = my_key;   // the depth bits ride in the record (binning.h: BlendRec)
            rec.pm = (uint32_t)my_radii;
            rec.co = make_float4(conic.x, conic.y, conic.z, opacities[idx]);
            index_rec[idx] = rec;
        }
    } while (0);
    // bands of band_h tile rows the reference rect reaches (a superset of the rows the lean lists keep): one bit per band for the
    // count / emit passes, which walk the image band by band, and this workgroup's count per band for their work split
    uint32_t my_bands = 0u;
    if (my_radii > 0 && rect_max.y > rect_min.y && rect_max.x > rect_min.x) {
        const uint32_t b0 = rect_min.y / band_h, b1 = (rect_max.y - 1u) / band_h;
        my_bands = (b1 >= 31u ? 0xFFFFFFFFu : ((2u << b1) - 1u)) & ~((1u << b0) - 1u);
    }
    if (idx < P) {
        radii[idx] = my_radii;
        tiles_touched[idx] = my_tiles;
        depth_key[idx] = my_key;
    }
    // band_bits[b][w]: which of the 64 Gaussians of wave w (= 64-index chunk w of the view) reach band b -- one ballot per band, lane b
    // stores band b's word: the count / emit workgroups of a band sift 64 Gaussians per word instead of reading a mask per Gaussian
    {
        unsigned long long mine = 0ull;
        for (uint32_t b = 0; b < nbands; b++) {
            const unsigned long long bal = ballot64(((my_bands >> b) & 1u) != 0u);
            if ((uint32_t)(threadIdx.x & 63) == b) mine = bal;
        }
        const uint32_t chunk = (uint32_t)idx >> 6, nchunks = ((uint32_t)P + 63u) >> 6;
        if ((uint32_t)(threadIdx.x & 63) < nbands && chunk < nchunks) band_bits[(size_t)(threadIdx.x & 63) * nchunks + chunk] = mine;
    }
    __syncthreads();   // (s_band zeroed; the SH staging barrier above is conditional)
    for (uint32_t m = my_bands; m != 0u; m &= m - 1u) atomicAdd(&s_band[__builtin_ctz(m)], 1u);
    __syncthreads();
    if (threadIdx.x < nbands && s_band[threadIdx.x] != 0u)
        atomicAdd(&r_slots[(blockIdx.x % R_SLOTS) * R_SLOT_STRIDE + R_SLOT_BANDS + threadIdx.x], (int)s_band[threadIdx.x]);
    // R = sum of tiles_touched over the Gaussians that later stages treat as visible (what InclusiveSum's last
    // element is in the reference, rasterizer_impl.cu:277-281), and the range of their depth keys (depth_sort.h).
    // Three atomics per wave, spread over R_SLOTS lines: same-line L2 atomics serialise at ~22 ns each.
    // Line layout: [0] R partial sum, [1] max of ~key (= ~min key), [2] max key; all start at 0.
    {
        uint32_t sum = (rect_max.x - rect_min.x) * (rect_max.y - rect_min.y);
        uint32_t inv_min = my_key == 0xFFFFFFFFu ? 0u : ~my_key, mx = my_key == 0xFFFFFFFFu ? 0u : my_key;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            sum += (uint32_t)__shfl_xor((int)sum, o, 64);
            inv_min = max(inv_min, (uint32_t)__shfl_xor((int)inv_min, o, 64));
            mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
        }
        if ((threadIdx.x & 63) == 0 && sum) {
            int* line = &r_slots[(blockIdx.x % R_SLOTS) * R_SLOT_STRIDE];
            atomicAdd(&line[0], (int)sum);
            atomicMax(reinterpret_cast<uint32_t*>(&line[1]), inv_min);
            atomicMax(reinterpret_cast<uint32_t*>(&line[2]), mx);
        }
    }
}

// CF/cuda_rasterizer/rasterizer_impl.cu:54-66
__global__ void __launch_bounds__(256) check_frustum_kernel(int P, const float* __restrict__ orig_points, ViewParams vp,
                                                            uint8_t* __restrict__ present)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    const float3 p_orig = make_float3(orig_points[3 * idx], orig_points[3 * idx + 1], orig_points[3 * idx + 2]);
    const float3 p_view = transformPoint4x3(p_orig, vp.view);
    present[idx] = (p_view.z <= 0.2f) ? 0 : 1;
}

// ------------------------------------------------------------------------------------------------
// Backward of the geometry path, fused into ONE streaming kernel (the reference launches
// computeCov2DCUDA then preprocessCUDA, CF/cuda_rasterizer/backward.cu:144-274, 346-396, 561-624;
// both are per-Gaussian with no cross-thread dependency, so fusing saves one full pass over
// dL_dmean3D / dL_dcov3D).
// ------------------------------------------------------------------------------------------------

// CF/cuda_rasterizer/backward.cu:20-139.  Adds into dL_dmean; writes dL_dsh.
__device__ __forceinline__ void sh_backward(int idx, int deg, int max_coeffs, const float3 pos, const ViewParams& vp,
                                            const float* shs, const uint8_t* clamped, const float* dL_dcolor,
                                            float3& dL_dmean_acc, float* dL_dshs)
{
    const float3 dir_orig = make_float3(pos.x - vp.campos[0], pos.y - vp.campos[1], pos.z - vp.campos[2]);
    const float len = sqrtf(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
    const float x = dir_orig.x / len, y = dir_orig.y / len, z = dir_orig.z / len;
    const float* sh = shs + (size_t)idx * max_coeffs * 3;
    float dL_dRGB[3] = {dL_dcolor[3 * idx + 0], dL_dcolor[3 * idx + 1], dL_dcolor[3 * idx + 2]};
    dL_dRGB[0] *= clamped[3 * idx + 0] ? 0 : 1;
    dL_dRGB[1] *= clamped[3 * idx + 1] ? 0 : 1;
    dL_dRGB[2] *= clamped[3 * idx + 2] ? 0 : 1;
    float dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
    float* dL_dsh = dL_dshs + (size_t)idx * max_coeffs * 3;
#define SH(k) sh[(k) * 3 + c]
#define DSH(k, v)               \
    _Pragma("unroll") for (int c = 0; c < 3; c++) dL_dsh[(k) * 3 + c] = (v) * dL_dRGB[c]
    const float dRGBdsh0 = SH_C0;
    DSH(0, dRGBdsh0);
    if (deg > 0) {
        const float dRGBdsh1 = -SH_C1 * y;
        const float dRGBdsh2 = SH_C1 * z;
        const float dRGBdsh3 = -SH_C1 * x;
        DSH(1, dRGBdsh1);
        DSH(2, dRGBdsh2);
        DSH(3, dRGBdsh3);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            dRGBdx[c] = -SH_C1 * SH(3);
            dRGBdy[c] = -SH_C1 * SH(1);
            dRGBdz[c] = SH_C1 * SH(2);
        }
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z;
            const float xy = x * y, yz = y * z, xz = x * z;
            const float dRGBdsh4 = SH_C2[0] * xy;
            const float dRGBdsh5 = SH_C2[1] * yz;
            const float dRGBdsh6 = SH_C2[2] * (2.f * zz - xx - yy);
            const float dRGBdsh7 = SH_C2[3] * xz;
            const float dRGBdsh8 = SH_C2[4] * (xx - yy);
            DSH(4, dRGBdsh4);
            DSH(5, dRGBdsh5);
            DSH(6, dRGBdsh6);
            DSH(7, dRGBdsh7);
            DSH(8, dRGBdsh8);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                dRGBdx[c] += SH_C2[0] * y * SH(4) + SH_C2[2] * 2.f * -x * SH(6) + SH_C2[3] * z * SH(7) + SH_C2[4] * 2.f * x * SH(8);
                dRGBdy[c] += SH_C2[0] * x * SH(4) + SH_C2[1] * z * SH(5) + SH_C2[2] * 2.f * -y * SH(6) + SH_C2[4] * 2.f * -y * SH(8);
                dRGBdz[c] += SH_C2[1] * y * SH(5) + SH_C2[2] * 2.f * 2.f * z * SH(6) + SH_C2[3] * x * SH(7);
            }
            if (deg > 2) {
                const float dRGBdsh9 = SH_C3[0] * y * (3.f * xx - yy);
                const float dRGBdsh10 = SH_C3[1] * xy * z;
                const float dRGBdsh11 = SH_C3[2] * y * (4.f * zz - xx - yy);
                const float dRGBdsh12 = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                const float dRGBdsh13 = SH_C3[4] * x * (4.f * zz - xx - yy);
                const float dRGBdsh14 = SH_C3[5] * z * (xx - yy);
                const float dRGBdsh15 = SH_C3[6] * x * (xx - 3.f * yy);
                DSH(9, dRGBdsh9);
                DSH(10, dRGBdsh10);
                DSH(11, dRGBdsh11);
                DSH(12, dRGBdsh12);
                DSH(13, dRGBdsh13);
                DSH(14, dRGBdsh14);
                DSH(15, dRGBdsh15);
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    dRGBdx[c] += (SH_C3[0] * SH(9) * 3.f * 2.f * xy + SH_C3[1] * SH(10) * yz + SH_C3[2] * SH(11) * -2.f * xy +
                                  SH_C3[3] * SH(12) * -3.f * 2.f * xz + SH_C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) +
                                  SH_C3[5] * SH(14) * 2.f * xz + SH_C3[6] * SH(15) * 3.f * (xx - yy));
                    dRGBdy[c] += (SH_C3[0] * SH(9) * 3.f * (xx - yy) + SH_C3[1] * SH(10) * xz +
                                  SH_C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * SH(12) * -3.f * 2.f * yz +
                                  SH_C3[4] * SH(13) * -2.f * xy + SH_C3[5] * SH(14) * -2.f * yz +
                                  SH_C3[6] * SH(15) * -3.f * 2.f * xy);
                    dRGBdz[c] += (SH_C3[1] * SH(10) * xy + SH_C3[2] * SH(11) * 4.f * 2.f * yz +
                                  SH_C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * SH(13) * 4.f * 2.f * xz +
                                  SH_C3[5] * SH(14) * (xx - yy));
                }
            }
        }
    }
#undef SH
#undef DSH
    const float3 dL_ddir = make_float3(dRGBdx[0] * dL_dRGB[0] + dRGBdx[1] * dL_dRGB[1] + dRGBdx[2] * dL_dRGB[2],
                                       dRGBdy[0] * dL_dRGB[0] + dRGBdy[1] * dL_dRGB[1] + dRGBdy[2] * dL_dRGB[2],
                                       dRGBdz[0] * dL_dRGB[0] + dRGBdz[1] * dL_dRGB[1] + dRGBdz[2] * dL_dRGB[2]);
    const float3 dm = dnormvdv(dir_orig, dL_ddir);
    dL_dmean_acc.x += dm.x;
    dL_dmean_acc.y += dm.y;
    dL_dmean_acc.z += dm.z;
}

// CF/cuda_rasterizer/backward.cu:278-341
__device__ __forceinline__ void cov3d_backward(const float3 scale, float mod, const float4 rot, const float* dL_dcov3D,
                                               float3& dL_dscale, float4& dL_drot)
{
    const float r = rot.x, x = rot.y, y = rot.z, z = rot.w;
    const Mat3 R = {{{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                     {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                     {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}}};
    Mat3 S = {{{1.0f, 0, 0}, {0, 1.0f, 0}, {0, 0, 1.0f}}};
    const float3 s = make_float3(mod * scale.x, mod * scale.y, mod * scale.z);
    S.m[0][0] = s.x;
    S.m[1][1] = s.y;
    S.m[2][2] = s.z;
    const Mat3 M = m3mul(S, R);
    const float* d = dL_dcov3D;
    const Mat3 dL_dSigma = {{{d[0], 0.5f * d[1], 0.5f * d[2]}, {0.5f * d[1], d[3], 0.5f * d[4]}, {0.5f * d[2], 0.5f * d[4], d[5]}}};
    Mat3 M2;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int rr = 0; rr < 3; rr++) M2.m[c][rr] = 2.0f * M.m[c][rr];
    const Mat3 dL_dM = m3mul(M2, dL_dSigma);
    const Mat3 Rt = m3transpose(R);
    Mat3 dL_dMt = m3transpose(dL_dM);
    dL_dscale.x = Rt.m[0][0] * dL_dMt.m[0][0] + Rt.m[0][1] * dL_dMt.m[0][1] + Rt.m[0][2] * dL_dMt.m[0][2];
    dL_dscale.y = Rt.m[1][0] * dL_dMt.m[1][0] + Rt.m[1][1] * dL_dMt.m[1][1] + Rt.m[1][2] * dL_dMt.m[1][2];
    dL_dscale.z = Rt.m[2][0] * dL_dMt.m[2][0] + Rt.m[2][1] * dL_dMt.m[2][1] + Rt.m[2][2] * dL_dMt.m[2][2];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        dL_dMt.m[0][k] *= s.x;
        dL_dMt.m[1][k] *= s.y;
        dL_dMt.m[2][k] *= s.z;
    }
    dL_drot.x = 2 * z * (dL_dMt.m[0][1] - dL_dMt.m[1][0]) + 2 * y * (dL_dMt.m[2][0] - dL_dMt.m[0][2]) + 2 * x * (dL_dMt.m[1][2] - dL_dMt.m[2][1]);
    dL_drot.y = 2 * y * (dL_dMt.m[1][0] + dL_dMt.m[0][1]) + 2 * z * (dL_dMt.m[2][0] + dL_dMt.m[0][2]) + 2 * r * (dL_dMt.m[1][2] - dL_dMt.m[2][1]) - 4 * x * (dL_dMt.m[2][2] + dL_dMt.m[1][1]);
    dL_drot.z = 2 * x * (dL_dMt.m[1][0] + dL_dMt.m[0][1]) + 2 * r * (dL_dMt.m[2][0] - dL_dMt.m[0][2]) + 2 * z * (dL_dMt.m[1][2] + dL_dMt.m[2][1]) - 4 * y * (dL_dMt.m[2][2] + dL_dMt.m[0][0]);
    dL_drot.w = 2 * r * (dL_dMt.m[0][1] - dL_dMt.m[1][0]) + 2 * x * (dL_dMt.m[2][0] + dL_dMt.m[0][2]) + 2 * y * (dL_dMt.m[1][2] + dL_dMt.m[2][1]) - 4 * z * (dL_dMt.m[1][1] + dL_dMt.m[0][0]);
}

__global__ void __launch_bounds__(256) geometry_bwd_kernel(
    int P, int D, int M, const float* __restrict__ means, const int* __restrict__ radii, const float* __restrict__ shs,
    const uint8_t* __restrict__ clamped, const float* __restrict__ scales, const float* __restrict__ rotations,
    const float* __restrict__ cov3Ds, ViewParams vp, const float* __restrict__ gpack /*[P,8] from blend_bwd*/,
    float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconics, float* __restrict__ dL_dopacity,
    float* __restrict__ dL_dmask, float* __restrict__ dL_dmeans, const float* __restrict__ dL_dcolor,
    float* __restrict__ dL_dcov, float* __restrict__ dL_dsh, float* __restrict__ dL_dscale, float* __restrict__ dL_drot)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    if (!(radii[idx] > 0)) {
        // Not rendered: every gradient of this Gaussian is zero.  Written here, so that the caller need not zero-fill these
        // outputs (include/mi_rast.h; the reference's glue zero-fills all of them, rasterize_points.cu:151-159: 96 B per
        // Gaussian of HBM writes that four fifths of the rows overwrite).  dL_dcolor and dL_dsh are accumulated into and stay
        // the caller's to clear.
        dL_dmean2D[3 * idx + 0] = dL_dmean2D[3 * idx + 1] = dL_dmean2D[3 * idx + 2] = 0.f;
        reinterpret_cast<float4*>(dL_dconics)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
        dL_dopacity[idx] = 0.f;
        if (dL_dmask) dL_dmask[idx] = 0.f;
#pragma unroll
        for (int i = 0; i < 6; i++) dL_dcov[(size_t)idx * 6 + i] = 0.f;
        dL_dmeans[3 * idx + 0] = dL_dmeans[3 * idx + 1] = dL_dmeans[3 * idx + 2] = 0.f;
        dL_dscale[3 * idx + 0] = dL_dscale[3 * idx + 1] = dL_dscale[3 * idx + 2] = 0.f;
        reinterpret_cast<float4*>(dL_drot)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }

    // unpack the per-Gaussian record written by the blend backward: {mean2D.x, mean2D.y, conic.x, conic.y,
    // conic.w, opacity, mask, -} -> the reference's dL_dmean2D (P,3), dL_dconic (P,2,2), dL_dopacity, dL_dmask
    const float4 g0 = reinterpret_cast<const float4*>(gpack)[2 * idx];
    const float4 g1 = reinterpret_cast<const float4*>(gpack)[2 * idx + 1];
    dL_dmean2D[3 * idx + 0] = g0.x;
    dL_dmean2D[3 * idx + 1] = g0.y;
    dL_dmean2D[3 * idx + 2] = 0.f;  // never written by the reference either (backward.cu:541-542): stays the glue's zero
    reinterpret_cast<float4*>(dL_dconics)[idx] = make_float4(g0.z, g0.w, 0.f, g1.x);
    dL_dopacity[idx] = g1.y;
    if (dL_dmask) dL_dmask[idx] = g1.z;

    // ---- computeCov2DCUDA, backward.cu:144-274
    const float3 mean = make_float3(means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]);
    const float3 dL_dconic = make_float3(g0.z, g0.w, g1.x);
    float cov3D[6];
#pragma unroll
    for (int i = 0; i < 6; i++) cov3D[i] = cov3Ds[(size_t)idx * 6 + i];
    Cov2DCtx c;
    cov2d_common(mean, vp, cov3D, c);
    const float h_x = vp.focal_x, h_y = vp.focal_y;
    const float limx = 1.3f * vp.tan_fovx;
    const float limy = 1.3f * vp.tan_fovy;
    const float x_grad_mul = c.txtz < -limx || c.txtz > limx ? 0 : 1;
    const float y_grad_mul = c.tytz < -limy || c.tytz > limy ? 0 : 1;
    const Mat3& T = c.T;
    const Mat3& Vrk = c.Vrk;
    const Mat3& Wm = c.W;
    const float3 t = c.t;
    const float a = c.cov.m[0][0];
    const float b = c.cov.m[0][1];
    const float cc = c.cov.m[1][1];
    const float denom = a * cc - b * b;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float dcov[6];
    if (denom2inv != 0) {
        dL_da = denom2inv * (-cc * cc * dL_dconic.x + 2 * b * cc * dL_dconic.y + (denom - a * cc) * dL_dconic.z);
        dL_dc = denom2inv * (-a * a * dL_dconic.z + 2 * a * b * dL_dconic.y + (denom - a * cc) * dL_dconic.x);
        dL_db = denom2inv * 2 * (b * cc * dL_dconic.x - (denom + 2 * b * b) * dL_dconic.y + a * b * dL_dconic.z);
        dcov[0] = (T.m[0][0] * T.m[0][0] * dL_da + T.m[0][0] * T.m[1][0] * dL_db + T.m[1][0] * T.m[1][0] * dL_dc);
        dcov[3] = (T.m[0][1] * T.m[0][1] * dL_da + T.m[0][1] * T.m[1][1] * dL_db + T.m[1][1] * T.m[1][1] * dL_dc);
        dcov[5] = (T.m[0][2] * T.m[0][2] * dL_da + T.m[0][2] * T.m[1][2] * dL_db + T.m[1][2] * T.m[1][2] * dL_dc);
        dcov[1] = 2 * T.m[0][0] * T.m[0][1] * dL_da + (T.m[0][0] * T.m[1][1] + T.m[0][1] * T.m[1][0]) * dL_db + 2 * T.m[1][0] * T.m[1][1] * dL_dc;
        dcov[2] = 2 * T.m[0][0] * T.m[0][2] * dL_da + (T.m[0][0] * T.m[1][2] + T.m[0][2] * T.m[1][0]) * dL_db + 2 * T.m[1][0] * T.m[1][2] * dL_dc;
        dcov[4] = 2 * T.m[0][2] * T.m[0][1] * dL_da + (T.m[0][1] * T.m[1][2] + T.m[0][2] * T.m[1][1]) * dL_db + 2 * T.m[1][1] * T.m[1][2] * dL_dc;
    } else {
#pragma unroll
        for (int i = 0; i < 6; i++) dcov[i] = 0;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) dL_dcov[(size_t)idx * 6 + i] = dcov[i];

    const float dL_dT00 = 2 * (T.m[0][0] * Vrk.m[0][0] + T.m[0][1] * Vrk.m[0][1] + T.m[0][2] * Vrk.m[0][2]) * dL_da +
                          (T.m[1][0] * Vrk.m[0][0] + T.m[1][1] * Vrk.m[0][1] + T.m[1][2] * Vrk.m[0][2]) * dL_db;
    const float dL_dT01 = 2 * (T.m[0][0] * Vrk.m[1][0] + T.m[0][1] * Vrk.m[1][1] + T.m[0][2] * Vrk.m[1][2]) * dL_da +
                          (T.m[1][0] * Vrk.m[1][0] + T.m[1][1] * Vrk.m[1][1] + T.m[1][2] * Vrk.m[1][2]) * dL_db;
    const float dL_dT02 = 2 * (T.m[0][0] * Vrk.m[2][0] + T.m[0][1] * Vrk.m[2][1] + T.m[0][2] * Vrk.m[2][2]) * dL_da +
                          (T.m[1][0] * Vrk.m[2][0] + T.m[1][1] * Vrk.m[2][1] + T.m[1][2] * Vrk.m[2][2]) * dL_db;
    const float dL_dT10 = 2 * (T.m[1][0] * Vrk.m[0][0] + T.m[1][1] * Vrk.m[0][1] + T.m[1][2] * Vrk.m[0][2]) * dL_dc +
                          (T.m[0][0] * Vrk.m[0][0] + T.m[0][1] * Vrk.m[0][1] + T.m[0][2] * Vrk.m[0][2]) * dL_db;
    const float dL_dT11 = 2 * (T.m[1][0] * Vrk.m[1][0] + T.m[1][1] * Vrk.m[1][1] + T.m[1][2] * Vrk.m[1][2]) * dL_dc +
                          (T.m[0][0] * Vrk.m[1][0] + T.m[0][1] * Vrk.m[1][1] + T.m[0][2] * Vrk.m[1][2]) * dL_db;
    const float dL_dT12 = 2 * (T.m[1][0] * Vrk.m[2][0] + T.m[1][1] * Vrk.m[2][1] + T.m[1][2] * Vrk.m[2][2]) * dL_dc +
                          (T.m[0][0] * Vrk.m[2][0] + T.m[0][1] * Vrk.m[2][1] + T.m[0][2] * Vrk.m[2][2]) * dL_db;
    const float dL_dJ00 = Wm.m[0][0] * dL_dT00 + Wm.m[0][1] * dL_dT01 + Wm.m[0][2] * dL_dT02;
    const float dL_dJ02 = Wm.m[2][0] * dL_dT00 + Wm.m[2][1] * dL_dT01 + Wm.m[2][2] * dL_dT02;
    const float dL_dJ11 = Wm.m[1][0] * dL_dT10 + Wm.m[1][1] * dL_dT11 + Wm.m[1][2] * dL_dT12;
    const float dL_dJ12 = Wm.m[2][0] * dL_dT10 + Wm.m[2][1] * dL_dT11 + Wm.m[2][2] * dL_dT12;
    const float tz = 1.f / t.z;
    const float tz2 = tz * tz;
    const float tz3 = tz2 * tz;
    const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
    const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
    const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 + (2 * h_y * t.y) * tz3 * dL_dJ12;
    float3 dL_dmean_acc = transformVec4x3Transpose(make_float3(dL_dtx, dL_dty, dL_dtz), vp.view);  // assignment, :273

    // ---- preprocessCUDA backward, backward.cu:346-396
    const float* proj = vp.proj;
    const float4 m_hom = transformPoint4x4(mean, proj);
    const float m_w = 1.0f / (m_hom.w + 0.0000001f);
    const float mul1 = (proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12]) * m_w * m_w;
    const float mul2 = (proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13]) * m_w * m_w;
    const float gx = g0.x, gy = g0.y;
    float3 dm;
    dm.x = (proj[0] * m_w - proj[3] * mul1) * gx + (proj[1] * m_w - proj[3] * mul2) * gy;
    dm.y = (proj[4] * m_w - proj[7] * mul1) * gx + (proj[5] * m_w - proj[7] * mul2) * gy;
    dm.z = (proj[8] * m_w - proj[11] * mul1) * gx + (proj[9] * m_w - proj[11] * mul2) * gy;
    dL_dmean_acc.x += dm.x;
    dL_dmean_acc.y += dm.y;
    dL_dmean_acc.z += dm.z;
    if (shs) sh_backward(idx, D, M, mean, vp, shs, clamped, dL_dcolor, dL_dmean_acc, dL_dsh);
    dL_dmeans[3 * idx + 0] = dL_dmean_acc.x;
    dL_dmeans[3 * idx + 1] = dL_dmean_acc.y;
    dL_dmeans[3 * idx + 2] = dL_dmean_acc.z;
    if (scales) {
        const float3 s = make_float3(scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]);
        const float4 q = reinterpret_cast<const float4*>(rotations)[idx];
        float3 ds;
        float4 dq;
        cov3d_backward(s, vp.scale_modifier, q, dcov, ds, dq);
        dL_dscale[3 * idx + 0] = ds.x;
        dL_dscale[3 * idx + 1] = ds.y;
        dL_dscale[3 * idx + 2] = ds.z;
        reinterpret_cast<float4*>(dL_drot)[idx] = dq;
    } else {  // cov3D_precomp given: no scale / rotation gradient (backward.cu:390-394)
        dL_dscale[3 * idx + 0] = dL_dscale[3 * idx + 1] = dL_dscale[3 * idx + 2] = 0.f;
        reinterpret_cast<float4*>(dL_drot)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// mask-only pair: dL_dmask[i] = packed field 6
__global__ void __launch_bounds__(256) unpack_mask_kernel(int P, const float* __restrict__ gpack, float* __restrict__ dL_dmask)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < P) dL_dmask[idx] = gpack[(size_t)idx * 8 + 6];
}

}  // namespace mirast
