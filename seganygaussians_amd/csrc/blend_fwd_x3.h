// blend_fwd_x3.h -- forward blend for C = 32 / 64 with the accumulation on the bf16 matrix pipe at f32 accuracy ("bf16x3").
//
// Same algorithm, staging and per-pixel arithmetic as blend_fwd.h (renderCUDA<C> forward, CF/cuda_rasterizer/
// forward.cu:264-385): alpha, T, the 1/255 and 1e-4 tests, n_contrib and final_T are computed by the same f32 VALU
// code and are bit-identical.  What differs is how  acc[pixel][ch] += w[pixel] * f[g][ch]  is evaluated.
// blend_fwd.h issues one v_mfma_f32_32x32x1_2b_f32 per (wave, Gaussian): 64 cycles on the f32 ALUs, which a SIMD
// cannot use for VALU work of its other waves meanwhile (tools/overlap_probe.hip) -- half of that kernel's time.
// The bf16 matrix pipe is 16x faster per FMA and runs beside the VALU (tools/bf16_probe.hip: a VALU wave loses 7 %
// next to a wave issuing v_mfma_f32_32x32x16_bf16 back to back, and 2x next to an f32-MFMA wave).  So:
//   * every f32 operand is split EXACTLY into three bf16 terms, x = hi + mid + lo (8 + 8 + 8 significant bits, each
//     the round-to-nearest bf16 of the remainder): the feature rows once, while they are staged; w per Gaussian;
//   * 16 Gaussians of a quadrant's list form one K = 16 group; the product is the sum of the six partial products
//     hi.hi + hi.mid + mid.hi + hi.lo + lo.hi + mid.mid, each exact in f32 inside the MFMA and accumulated in f32.
//     The three dropped terms are below 2^-26 of |w f|.  The image is NOT bit-identical to blend_fwd.h (the MFMA rounds
//     its 16-product partial sums differently from sixteen chained fmaf) but sits at the same f32 rounding level:
//     max difference between the two kernels 7e-7 of the image scale, RMS error against the fp64-accumulating oracle
//     2.1e-8 vs 0.7e-8 (tests/test_gpu_parity.py::test_features_forward_x3_matches_f32_mfma).
//   * per group: 12 MFMA x 32 cycles for 64 px x 32 ch x 16 Gaussians = 24 cycles per (wave, Gaussian) instead of 64.
// Lane = pixel computes w for all 16 Gaussians of a group, but the MFMA wants lanes 0..31 to hold k = 0..7 and lanes
// 32..63 k = 8..15 of a 32-pixel block: lane l and lane l ^ 32 exchange half of their packed terms (v_permlane32_swap).
// B (features) is gathered per lane from the staged rows: lane (channel n, k half) reads 8 bf16 per term.
#pragma once

#include "blend_fwd.h"
#include "blend_fwd_split.h"

namespace mirast {

constexpr int XB = 128;    // blend-list records per batch

// C = 64: two 32-channel accumulator pairs (64 VGPRs), 24 MFMA per group, rows of 384 bytes: 2 workgroups per CU.
template <int C, bool XEXP = false, bool STRIDED = false>
__global__ void __launch_bounds__(256, C == 32 ? 4 : 2) blend_fwd_x3_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ blend_list, const BlendRec* __restrict__ index_rec,
    int W, int H, const float* __restrict__ features, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
    uint32_t* __restrict__ tile_consumed, uint32_t* __restrict__ tile_nsurv, const float* __restrict__ bg_color,
    float* __restrict__ out_color, int cstride_arg /* STRIDED: floats between feature rows (blend_fwd.h) */)
{
    const int cstride = STRIDED ? cstride_arg : C;
    static_assert(C == 32 || C == 64, "32-channel accumulator blocks");
    constexpr int F4 = C / 4, NCB = C / 32, XROW = 6 * C, PLANE = 2 * C;  // float4s per row; channel blocks; row / plane bytes
    __shared__ XRec s_rec[XB + 1];   // [XB] = padding record (opacity 0: never blends)
    __shared__ uint32_t s_id[XB];
    __shared__ uint4 s_feat4[(XB + 1) * XROW / 16];    // [XB] = all-zero row
    __shared__ uint64_t s_bits[4][XB / 64];
    __shared__ uint32_t s_list[4][XB + XG];            // per quadrant: BYTE OFFSETS of its records in s_rec, in list order, padded
    __shared__ int s_consumed, s_walked;
    char* const featb = reinterpret_cast<char*>(s_feat4);
    const char* const rec_bytes = reinterpret_cast<const char*>(s_rec);

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    if (tid == 0) {
        s_consumed = 0;
        s_walked = 0;
        s_rec[XB].q0 = make_float4(0.f, 0.f, -0.5f, 0.f);
        s_rec[XB].q1 = make_float4(-0.5f, 0.f, 0.f, 0.f);
    }
    if (tid < XROW / 16) s_feat4[XB * (XROW / 16) + tid] = make_uint4(0u, 0u, 0u, 0u);
    const uint32_t horizontal_blocks = (W + TILE_X - 1) / TILE_X;
    const uint32_t tile = blockIdx.y * horizontal_blocks + blockIdx.x;
    const uint32_t px = blockIdx.x * TILE_X + (wave & 1) * 8 + (lane & 7);
    const uint32_t py = blockIdx.y * TILE_Y + (wave >> 1) * 8 + (lane >> 3);
    const uint32_t pix_id = W * py + px;
    const float pixfx = (float)px, pixfy = (float)py;
    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    bool done = !inside;

    const uint2 range = ranges[tile];
    const int list_len = (int)(range.y - range.x);
    const int ns_total = (int)(range.y - range.x);
    const uint32_t* lst = blend_list + range.x;

    float T = 1.0f;
    uint32_t last_contributor = 0;
    int consumed = (ballot64(!done) != 0) ? list_len : 0;  // see blend_fwd.h
    int walked = 0;
    v16f acc0[NCB], acc1[NCB];  // pixels 0..31 / 32..63 of the quadrant x channels 32 cb + (lane & 31)
#pragma unroll
    for (int cb = 0; cb < NCB; cb++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            acc0[cb][r] = 0.f;
            acc1[cb][r] = 0.f;
        }
    const bool lower = lane < 32;
    const int chan2 = 2 * (lane & 31);  // byte offset of this lane's channel inside a bf16 plane

    BlendRec cur;
    if (tid < XB && tid < ns_total) cur = list_record(lst, index_rec, tid);

    for (int b0 = 0; b0 < ns_total; b0 += XB) {
        if (__syncthreads_and(done)) break;  // whole workgroup finished? (also the barrier that protects LDS reuse)
        const int nb = min(XB, ns_total - b0);
        // ---- A: records -> LDS; next batch's records -> registers
        if (tid < nb) {
            s_rec[tid].q0 = make_float4(cur.xy.x, cur.xy.y, -0.5f * cur.co.x, -cur.co.y);  // conic pre-scaled: see blend_fwd.h
            s_rec[tid].q1 = make_float4(-0.5f * cur.co.z, cur.co.w, __uint_as_float((cur.pm >> 4) + 1u), __uint_as_float(cur.pm));
            s_id[tid] = cur.id;
        }
        if (wave < XB / 64) {
            const uint32_t pmv = tid < nb ? cur.pm : 0u;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint64_t b = ballot64((pmv >> q) & 1u);
                if (lane == 0) s_bits[q][wave] = b;
            }
        }
        __syncthreads();
        // ---- B: feature rows, gathered by the ids just staged, split into their three bf16 terms
        {
            constexpr int NK = XB * F4 / BATCH;
            float4 v[NK];
#pragma unroll
            for (int k = 0; k < NK; k++) {
                const int q = tid + BATCH * k;
                const int g = q / F4, part = q % F4;
                v[k] = reinterpret_cast<const float4*>(features + (size_t)s_id[g < nb ? g : 0] * cstride)[part];
            }
            // pins every loaded value in registers here: hipcc otherwise sinks each load into the guarded store below
#pragma unroll
            for (int k = 0; k < NK; k++) asm volatile("" : "+v"(v[k].x), "+v"(v[k].y), "+v"(v[k].z), "+v"(v[k].w));
            if (tid < XB && b0 + XB + tid < ns_total) cur = list_record(lst, index_rec, b0 + XB + tid);  // after the gather: vmcnt retires in order
#pragma unroll
            for (int k = 0; k < NK; k++) {
                const int q = tid + BATCH * k;
                const int g = q / F4, part = q % F4;
                if (g < nb) {
                    uint2 hi, mid, lo;
                    split3_bf16x2(v[k].x, v[k].y, hi.x, mid.x, lo.x);
                    split3_bf16x2(v[k].z, v[k].w, hi.y, mid.y, lo.y);
                    uint2* row = reinterpret_cast<uint2*>(featb + g * XROW);
                    row[part] = hi;
                    row[F4 + part] = mid;
                    row[2 * F4 + part] = lo;
                }
            }
        }
        __syncthreads();

        // ---- C: blend, 16 list entries of this quadrant per group
        if (ballot64(!done) != 0) {
            walked = b0 + nb;
            int cnt = 0;
#pragma unroll
            for (int h = 0; h < XB / 64; h++) {
                const uint64_t b = s_bits[wave][h];
                const bool set = (b >> lane) & 1ull;
                const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
                if (set) s_list[wave][cnt + below] = (uint32_t)((64 * h + lane) * sizeof(XRec));
                cnt += __builtin_popcountll(b);
            }
            cnt = __builtin_amdgcn_readfirstlane(cnt);
            if (lane < XG) s_list[wave][cnt + lane] = (uint32_t)(XB * sizeof(XRec));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            uint64_t live = ballot64(!done);  // lanes still blending (wave-uniform copy of !done)
            bool finished = false;
            for (int j0 = 0; j0 < cnt && !finished; j0 += XG) {
                // -- alpha, T and w of the 16 entries (same f32 arithmetic as blend_fwd.h), w split and packed in pairs
                uint32_t wp[3][XG / 2];
                int fin_j = -1;
#pragma unroll
                for (int i = 0; i < XG / 2; i++) {
                    if (j0 + 2 * i >= cnt || live == 0) {  // padding pair of the batch's last group, or every pixel is done (wave-uniform): w = 0
                        wp[0][i] = wp[1][i] = wp[2][i] = 0u;
                        continue;
                    }
                    float w2[2];
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const uint32_t off = s_list[wave][j0 + 2 * i + h];
                        const float4 p0 = *reinterpret_cast<const float4*>(rec_bytes + off);
                        const float4 p1 = *reinterpret_cast<const float4*>(rec_bytes + off + 16);
                        const float dx = p0.x - pixfx, dy = p0.y - pixfy;
                        const float power = gauss_power(p0.z, p0.w, p1.x, dx, dy);
                        const float t = p1.y * gauss_exp<XEXP>(power);
                        const float alpha = fminf(0.99f, t);
                        const bool ok = !done && power <= 0.0f && t >= (1.0f / 255.0f);
                        const float test_T = T * (1 - alpha);
                        // forward.cu:358-362: done once a contributor would push T below 1e-4 (that one is not blended).
                        // The lane mask of the compare is taken directly (a ballot of the bool costs two more VALU).
                        const float tt = ok ? test_T : 1.0f;
                        live &= ~__builtin_amdgcn_fcmpf(tt, 0.0001f, 4 /* FCMP_OLT */);
                        const bool stop = tt < 0.0001f;
                        done = done || stop;
                        const bool blend = ok && !stop;
                        w2[h] = blend ? alpha * T : 0.f;
                        T = blend ? test_T : T;
                        last_contributor = blend ? __float_as_uint(p1.z) : last_contributor;
                        fin_j = (live == 0 && fin_j < 0) ? 2 * i + h : fin_j;  // first entry after which nobody is left
                    }
                    split3_bf16x2(w2[0], w2[1], wp[0][i], wp[1][i], wp[2][i]);
                    __builtin_amdgcn_sched_barrier(0);  // keeps the record reads of later pairs from piling up in registers
                }
                if (fin_j >= 0) {
                    const uint32_t of = __builtin_amdgcn_readfirstlane(s_list[wave][j0 + fin_j]);
                    consumed = (int)__builtin_amdgcn_readfirstlane(__float_as_uint(s_rec[of / sizeof(XRec)].q1.z));
                    walked = b0 + (int)(of / sizeof(XRec)) + 1;
                    finished = true;
                }
                // -- A operands: block 0 (pixels 0..31): lanes < 32 hold their k = 0..7, lanes >= 32 the k = 8..15 of lane - 32;
                //    block 1 (pixels 32..63): lanes < 32 hold the k = 0..7 of lane + 32, lanes >= 32 their k = 8..15
                // v_permlane32_swap x, y exchanges the upper half of x with the lower half of y (tools/bf16_probe.hip; it
                // needs wait states after a VALU write of its operands, which hipcc does not insert around inline asm)
                v4u A0[3], A1[3];
                asm volatile("s_nop 3");
#pragma unroll
                for (int p = 0; p < 3; p++) {
#pragma unroll
                    for (int d = 0; d < 4; d++) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(wp[p][d]), "+v"(wp[p][4 + d]));
                    A0[p] = (v4u){wp[p][0], wp[p][1], wp[p][2], wp[p][3]};
                    A1[p] = (v4u){wp[p][4], wp[p][5], wp[p][6], wp[p][7]};
                }
                // -- B operands: lane (channel, k half) gathers the three terms of its 8 entries' feature value, per 32-channel
                //    block; six partial products per pixel block, smallest first.  Inline asm with a tied accumulator (see
                //    blend_fwd.h); s_nop 1 covers VALU-written operands.
                const uint4* lp = reinterpret_cast<const uint4*>(&s_list[wave][j0 + (lower ? 0 : 8)]);
                const uint4 ka = lp[0], kb = lp[1];
                const uint32_t ks[8] = {ka.x, ka.y, ka.z, ka.w, kb.x, kb.y, kb.z, kb.w};
#define X3_MFMA(ACC, AP, BP) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(AP), "v"(BP))
#pragma unroll
                for (int cb = 0; cb < NCB; cb++) {
                    v4u B[3];
                    {
                        uint32_t bb[3][4];
#pragma unroll
                        for (int d = 0; d < 4; d++) {
                            const char* r0 = featb + ks[2 * d] * (XROW / (int)sizeof(XRec)) + chan2 + 64 * cb;
                            const char* r1 = featb + ks[2 * d + 1] * (XROW / (int)sizeof(XRec)) + chan2 + 64 * cb;
#pragma unroll
                            for (int p = 0; p < 3; p++)
                                bb[p][d] = (uint32_t)*reinterpret_cast<const uint16_t*>(r0 + PLANE * p) |
                                           ((uint32_t)*reinterpret_cast<const uint16_t*>(r1 + PLANE * p) << 16);
                        }
#pragma unroll
                        for (int p = 0; p < 3; p++) B[p] = (v4u){bb[p][0], bb[p][1], bb[p][2], bb[p][3]};
                    }
                    X3_MFMA(acc0[cb], A0[1], B[1]);
                    X3_MFMA(acc1[cb], A1[1], B[1]);
                    X3_MFMA(acc0[cb], A0[2], B[0]);
                    X3_MFMA(acc1[cb], A1[2], B[0]);
                    X3_MFMA(acc0[cb], A0[0], B[2]);
                    X3_MFMA(acc1[cb], A1[0], B[2]);
                    X3_MFMA(acc0[cb], A0[1], B[0]);
                    X3_MFMA(acc1[cb], A1[1], B[0]);
                    X3_MFMA(acc0[cb], A0[0], B[1]);
                    X3_MFMA(acc1[cb], A1[0], B[1]);
                    X3_MFMA(acc0[cb], A0[0], B[0]);
                    X3_MFMA(acc1[cb], A1[0], B[0]);
                }
#undef X3_MFMA
            }
        }
    }

    if (lane == 0) {
        atomicMax(&s_consumed, consumed);
        atomicMax(&s_walked, walked);
    }
    __syncthreads();
    if (tid == 0) {
        tile_consumed[tile] = (uint32_t)s_consumed;
        tile_nsurv[tile] = (uint32_t)s_walked;
    }
    const size_t HW = (size_t)H * W;
    if (inside) {
        final_T[pix_id] = T;
        n_contrib[pix_id] = last_contributor;
    }
    // accumulator tiles: lane l holds channel (l & 31); register r of block b <-> pixel 32 b + (r & 3) + 8 (r >> 2) + 4 (l >> 5)
    // (tools/mfma_probe.hip).  Transposed through LDS, 8 channels at a time per wave, then stored pixel-major.
    static_assert((XB + 1) * XROW >= 4 * 8 * 65 * 4, "feature buffer too small for the epilogue transpose");
    float* tp = reinterpret_cast<float*>(s_feat4) + wave * 8 * 65;
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");  // MFMA result -> VALU read
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
#pragma unroll
        for (int part = 0; part < 4; part++) {
            const int chl = lane & 31;
            if ((chl >> 3) == part) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int L = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    tp[(chl & 7) * 65 + L] = acc0[cb][r];
                    tp[(chl & 7) * 65 + 32 + L] = acc1[cb][r];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (inside) {
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    const int ch = 32 * cb + 8 * part + c;
                    out_color[ch * HW + pix_id] = tp[c * 65 + lane] + T * bg_color[ch];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }
    }
}

}  // namespace mirast
