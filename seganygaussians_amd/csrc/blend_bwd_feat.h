// blend_bwd_feat.h -- features-only backward blend (mi_rast.h: MI_RAST_BWD_FEATURES_ONLY): one wave per HALF TILE (16 x 8 pixels, TWO
// pixels per lane), no workgroup barriers.
//
// dL/dfeature[g] = sum over the pairs of alpha T dL/dpixel (renderCUDA<C> backward, CF/cuda_rasterizer/backward.cu:487-501) needs alpha
// and T of every pair and nothing else: no feature rows, no S = F dL^T contraction, no dL/dalpha recurrence, no moments, no packed-field
// atomics.  The first version of this form was blend_bwd_wave.h with those parts compiled out (one wave per 8 x 8 quadrant: 76 VGPRs,
// 5.9 KB of LDS).  Measured there (profiles/r06_bwd_atomics.md): the form is bound by the RATE of the atomic units and by nothing
// else -- 0.25 ms of ALU work behind 0.41 ms of atomics on cfg3, two 64-byte requests per (quadrant, record) row, and a record reaches
// 2.95 of a tile's four quadrants.  Registers and LDS are what the full kernel lacks for merging rows in front of the atomics; this form
// has both to spare (cfg3 0.41 -> 0.30 ms, cfg5 0.70 -> 0.50).  Here a wave owns the two quadrants of a tile's upper or lower half: lane l blends pixel (l & 7, l >> 3) of the left
// quadrant AND of the right one, the chunk's rows are the records that reach EITHER quadrant (a record outside one of them evaluates
// to alpha < 1/255 there: the lists are exact-conservative, cull.h), dF = W_left^T dL_left + W_right^T dL_right accumulates in ONE set
// of matrix accumulators, and one row of atomics leaves per (half tile, record): ~1.8 rows per (tile, record) instead of 2.95.
// The walk, the queue ring, the chunk pipeline and the XCD runs are blend_bwd_wave.h's (see there for the reasons behind each barrier,
// fence and opaque lane copy).
#pragma once

#include <type_traits>

#include "blend_bwd_shared.h"
#include "blend_bwd_wave.h"
#include "blend_fwd.h"
#include "common.h"

namespace mirast {

// C: channels of the block (16, 32 or 64: 126 / 126 / 220 VGPRs, three / three / two waves per SIMD), all of them in memory; STRIDED: rows of `dL_dcolors` are `cstride_arg` floats apart (one channel
// block of a wider feature), otherwise C.
template <int C, bool XEXP = true, bool STRIDED = false>
__global__ void __launch_bounds__(64, C == 64 ? 2 : 3) blend_bwd_feat_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ blend_list, const BlendRec* __restrict__ index_rec,
    const uint32_t* __restrict__ tile_nsurv, int W, int H, uint32_t horizontal_blocks, uint32_t ntiles,
    const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
    float* __restrict__ dL_dcolors, uint32_t* __restrict__ queue_ctr /* eight zeroed work-queue counters (common.h: xcd_grab) */,
    int cstride_arg, const uint32_t* __restrict__ run_bounds /* [9]: the XCDs' runs of tiles, left in the image buffer by the forward */)
{
    static_assert(C == 16 || C == 32 || C == 64, "16-, 32- and 64-channel blocks");
    constexpr int QCAP = 128;
    constexpr int NB = C / 16;   // 16-channel blocks of the dF contraction
    constexpr int SROW = 17;     // gradient-image staging row (floats; 16 channels at a time)
    const int cstride = STRIDED ? cstride_arg : C;
    static_assert(2 * CHK * WROW >= 64 * SROW, "gradient-image staging must fit in the w rows");

    __shared__ BwdPar s_par[CHK];                   // the chunk's 16 records
    __shared__ float4 s_w4[2 * CHK * WROW / 4];     // w rows of the left | right quadrant   (prologue: gradient-image staging)
    __shared__ uint2 s_queue[QCAP];                 // {walk index, entry = Gaussian id | quadrant mask << 28}

    // One (tile, half) item: half 0 = quadrants 0, 1 (upper 8 pixel rows), half 1 = quadrants 2, 3.
    auto half_tile = [&](const uint32_t tile, const uint32_t half) __attribute__((always_inline)) {
    const uint32_t tile_x = tile % horizontal_blocks, tile_y = tile / horizontal_blocks;
    const int lane = threadIdx.x & 63;
    const uint32_t hx0 = tile_x * TILE_X, hy0 = tile_y * TILE_Y + half * 8;
    const uint32_t pxa = hx0 + (lane & 7), pxb = pxa + 8, py = hy0 + (lane >> 3);
    const float pixfxa = (float)pxa, pixfxb = (float)pxb, pixfy = (float)py;
    const bool inside_a = pxa < (uint32_t)W && py < (uint32_t)H, inside_b = pxb < (uint32_t)W && py < (uint32_t)H;
    const size_t HW = (size_t)H * W;
    const size_t pix_a = inside_a ? (size_t)W * py + pxa : 0, pix_b = inside_b ? (size_t)W * py + pxb : 0;

    const uint2 range = ranges[tile];
    const int NS_tile = (int)tile_nsurv[tile];
    // per-pixel state and the gradient image of both pixels, requested at once (all loads unconditional: pix_* are valid pixels)
    const int nc_a = (int)n_contrib[pix_a], nc_b = (int)n_contrib[pix_b];
    const float Tr_a = final_Ts[pix_a], Tr_b = final_Ts[pix_b];
    // (64 channels: the right pixel's image is requested when the left one's has been staged -- 2 x 64 values do not fit the registers)
    constexpr bool TWO_PHASE = C == 64;
    float dLa[C], dLb[TWO_PHASE ? 1 : C];
#pragma unroll
    for (int ch = 0; ch < C; ch++) {
        dLa[ch] = dL_dpixels[(size_t)ch * HW + pix_a];
        if constexpr (!TWO_PHASE) dLb[ch] = dL_dpixels[(size_t)ch * HW + pix_b];
    }
    __builtin_amdgcn_sched_barrier(0);
    const int last_a = inside_a ? nc_a : 0, last_b = inside_b ? nc_b : 0;
    int wave_Lt = max(last_a, last_b);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) wave_Lt = max(wave_Lt, __shfl_xor(wave_Lt, o, 64));
    wave_Lt = __builtin_amdgcn_readfirstlane(wave_Lt);
    const int NS = min(NS_tile, wave_Lt);   // 0: nothing blended into this half tile -- the wave leaves after the staging
    const uint32_t* lst = NS > 0 ? blend_list + range.x : blend_list;
    uint32_t scan_reg = lst[max(0, NS - 1 - min(lane, NS - 1))];
    __builtin_amdgcn_sched_barrier(0);
    float Ta = inside_a ? Tr_a : 0.f, Tb = inside_b ? Tr_b : 0.f;

    // ---- gradient image: B of the dF contraction, dLT[nb][s] = dL[pixel 16 kq + s][channel 16 nb + n16], per quadrant
    const int n16 = lane & 15, kq = lane >> 4;
    float dLTa[NB][16], dLTb[NB][16];
    {
        float* stage = reinterpret_cast<float*>(s_w4);
#pragma unroll
        for (int q = 0; q < 2; q++) {
            if constexpr (TWO_PHASE) {
                if (q == 1) {
#pragma unroll
                    for (int ch = 0; ch < C; ch++) dLa[ch] = dL_dpixels[(size_t)ch * HW + pix_b];
                }
            }
#pragma unroll
            for (int h = 0; h < NB; h++) {
#pragma unroll
                for (int c = 0; c < 16; c++)
                    stage[lane * SROW + c] = (q == 0 || TWO_PHASE) ? ((q == 0 ? inside_a : inside_b) ? dLa[16 * h + c] : 0.f) : (inside_b ? dLb[TWO_PHASE ? 0 : 16 * h + c] : 0.f);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
                for (int s = 0; s < 16; s++) {
                    const float v = stage[(16 * kq + s) * SROW + n16];
                    if (q == 0) dLTa[h][s] = v;
                    else dLTb[h][s] = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            }
        }
    }
    if (NS == 0) return;
    const int last4a = last_a << 4, last4b = last_b << 4;  // compared with (position << 4 | mask)

    float* const my_wa = reinterpret_cast<float*>(s_w4);
    float* const my_wb = my_wa + CHK * WROW;
    const char* const par_bytes = reinterpret_cast<const char*>(s_par);

    // ---- the queue of this half tile's records: entries whose mask names one of its two quadrants
    int scanned = 0, qh = 0, qt = 0;
    auto consume_scan = [&]() {
        const int j = scanned + lane;
        const bool cand = j < NS && ((scan_reg >> (ID_BITS + 2 * half)) & 3u) != 0;
        const uint64_t bal = ballot64(cand);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (cand) s_queue[(qt + (int)below) & (QCAP - 1)] = make_uint2((uint32_t)j, scan_reg);
        qt += __builtin_popcountll(bal);
        scanned += 64;
        scan_reg = lst[NS - 1 - min(scanned + lane, NS - 1)];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    };
    // record quarter (lane & 3) of row (lane >> 2) of the next chunk (rows >= n repeat row n - 1 and become padding when staged)
    uint2 curq;
    auto request_rows = [&](int n) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int l = threadIdx.x & 63;
        asm volatile("" : "+v"(l));
        const int rq = min(l >> 2, n - 1), qq = l & 3;
        const uint32_t gq = s_queue[(qh + rq) & (QCAP - 1)].y & ID_MASK;
        curq = reinterpret_cast<const uint2*>(index_rec + gq)[qq];
    };

    while (qt - qh < CHK && scanned < NS) consume_scan();
    int nrows = min(CHK, qt - qh);
    if (nrows == 0) return;
    request_rows(nrows);

    int nnext = 0;
    auto do_chunk = [&](auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;   // all 16 rows are real: unconditional atomics
        // ---- 1. the chunk's rows: registers -> LDS (blend_bwd_wave.h step 1; padding rows: never valid, opacity 1)
        {
            const int rq = lane >> 2, qq = lane & 3;
            float2 v = make_float2(__uint_as_float(curq.x), __uint_as_float(curq.y));
            const uint2 qe = s_queue[(qh + (FULL ? rq : min(rq, nrows - 1))) & (QCAP - 1)];   // {walk index j, id | mask << 28}
            if (qq == 1) v = make_float2(__uint_as_float(((uint32_t)(NS - 1 - (int)qe.x) << 4) | (qe.y >> ID_BITS)), __uint_as_float(qe.y & ID_MASK));
            if (qq == 2) v = make_float2(-0.5f * v.x, -v.y);
            if (qq == 3) v = make_float2(-0.5f * v.x, v.y);
            if (!FULL) {
                if (rq >= nrows) {
                    v = make_float2(0.f, 0.f);
                    if (qq == 1) v = make_float2(__int_as_float(0x7ffffff0), __uint_as_float(0u));
                    if (qq == 2) v = make_float2(-0.5f, 0.f);
                    if (qq == 3) v = make_float2(-0.5f, 1.f);
                }
            }
            const int dst = qq == 0 ? 0 : (qq == 1 ? 24 : (qq == 2 ? 8 : 16));
            *reinterpret_cast<float2*>(reinterpret_cast<char*>(&s_par[rq]) + dst) = v;
        }
        qh += nrows;
        // ---- 2. keep the queue ahead of the chunks, request the next chunk's rows
        if (scanned < NS && qt - qh <= QCAP - 64) consume_scan();
        while (qt - qh < CHK && scanned < NS) consume_scan();
        nnext = min(CHK, qt - qh);
        if (nnext > 0) request_rows(nnext);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

        // ---- 4. alpha, T and w = alpha T of both pixels, back to front, four rows per reciprocal (backward.cu:470-487; the
        // arithmetic of blend_bwd_wave.h's row_group4).  A row that does not blend into a pixel: alpha = 0, T stays, w = 0.
#pragma unroll
        for (int r0 = 0; r0 < CHK; r0 += 4) {
            float ala[4], oma[4], alb[4], omb[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int rr = r0 + k;
                const float4 p0 = *reinterpret_cast<const float4*>(par_bytes + rr * (int)sizeof(BwdPar));
                const float4 p1 = *reinterpret_cast<const float4*>(par_bytes + rr * (int)sizeof(BwdPar) + 16);
                const float dy = p0.y - pixfy;
                const float dxa = p0.x - pixfxa, dxb = p0.x - pixfxb;
                const float pwa = gauss_power(p0.z, p0.w, p1.x, dxa, dy), pwb = gauss_power(p0.z, p0.w, p1.x, dxb, dy);
                const float Ga = gauss_exp<XEXP>(pwa), Gb = gauss_exp<XEXP>(pwb);
                const int pm = __float_as_int(p1.z);
                const float ta = (pm < last4a && pwa <= 0.0f) ? p1.y * Ga : 0.f;
                const float tb = (pm < last4b && pwb <= 0.0f) ? p1.y * Gb : 0.f;
                ala[k] = alpha_clamp(ta >= ALPHA_CUT ? ta : 0.f);
                alb[k] = alpha_clamp(tb >= ALPHA_CUT ? tb : 0.f);
                oma[k] = 1.f - ala[k];
                omb[k] = 1.f - alb[k];
            }
            float Tka[4], Tkb[4];
            Tka[3] = Ta * rcp_refined((oma[0] * oma[1]) * (oma[2] * oma[3]));
            Tkb[3] = Tb * rcp_refined((omb[0] * omb[1]) * (omb[2] * omb[3]));
            Tka[2] = Tka[3] * oma[3];
            Tkb[2] = Tkb[3] * omb[3];
            Tka[1] = Tka[2] * oma[2];
            Tkb[1] = Tkb[2] * omb[2];
            Tka[0] = Tka[1] * oma[1];
            Tkb[0] = Tkb[1] * omb[1];
            Ta = Tka[3];
            Tb = Tkb[3];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                my_wa[(r0 + k) * WROW + lane] = ala[k] * Tka[k];
                my_wb[(r0 + k) * WROW + lane] = alb[k] * Tkb[k];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

        // ---- 5. dF = W_a^T . dL_a + W_b^T . dL_b   (A rows from LDS, lane (m = n16, kq) reads pixels 16 kq .. 16 kq + 15)
        v4f facc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; nb++) facc[nb] = (v4f){0.f, 0.f, 0.f, 0.f};
        {
            const float4* wra = reinterpret_cast<const float4*>(my_wa + n16 * WROW + 16 * kq);
            const float4* wrb = reinterpret_cast<const float4*>(my_wb + n16 * WROW + 16 * kq);
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) {
                const float4 wva = wra[s4], wvb = wrb[s4];
                const float wa[4] = {wva.x, wva.y, wva.z, wva.w};
                const float wb[4] = {wvb.x, wvb.y, wvb.z, wvb.w};
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int s = 4 * s4 + t;
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) {
                        facc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t], dLTa[nb][s], facc[nb], 0, 0, 0);
                        facc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[t], dLTb[nb][s], facc[nb], 0, 0, 0);
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the operand reads of the w rows are done: the next chunk may write them
        // ---- 6. one row of atomics per (half tile, record): lane l holds channel n16 of rows 4 kq + r
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = 4 * kq + r;
            const uint32_t gid = __float_as_uint(*reinterpret_cast<const float*>(par_bytes + row * (int)sizeof(BwdPar) + 28));
#pragma unroll
            for (int nb = 0; nb < NB; nb++) {
                if (!FULL && row >= nrows) continue;
                atomicAdd(&dL_dcolors[(size_t)gid * cstride + 16 * nb + n16], facc[nb][r]);
            }
        }
    };
    if (nrows == CHK) {
        do_chunk(std::true_type{});
        while (nnext == CHK) {
            nrows = nnext;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            do_chunk(std::true_type{});
        }
        nrows = nnext;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    if (nrows != 0) do_chunk(std::false_type{});
    };

    {
        // workgroup -> (tile, half): the XCD runs and the half-static / half-queued hand-out of blend_bwd_wave.h with TWO items per tile
        // (id = 8 (2 j + half) + x: XCD x, j-th tile of its run)
        const uint32_t b = blockIdx.x, nstatic = 16u * xcd_static_len_max(ntiles);
        uint32_t item;
        if (b < nstatic) {
            const uint32_t x = b & 7u, jj = b >> 3;
            const uint32_t start = run_bounds[x], len = run_bounds[x + 1u] - start;
            item = (jj >> 1) < len - len / XCD_QUEUE_DIV ? 2u * start + jj : 0xFFFFFFFFu;
        } else {
            item = xcd_grab_runs(queue_ctr, xcd_load_runs(run_bounds), 2u);
        }
        if (item != 0xFFFFFFFFu) half_tile(item >> 1, item & 1u);
    }
}

}  // namespace mirast
