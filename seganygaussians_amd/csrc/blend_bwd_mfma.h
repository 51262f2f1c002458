// blend_bwd_mfma.h -- per-tile back-to-front gradient pass for C = 32 / 64 feature channels (and RGB padded to 16) with f32 MFMA.
//
// Same mathematics as blend_bwd.h (which restates renderCUDA<C> backward, CF/cuda_rasterizer/backward.cu:399-559,
// and remains the path for the mask-only pair and comparisons); same tolerance-checked results.  What changes is
// where the work runs.  Per valid (wave, Gaussian) pair the VALU version spends ~250 instructions and ~250 LDS
// cycles, almost all of them in three contractions:
//     S[g][p]   = sum_ch  f[g][ch]  * dL[p][ch]        (feeds dL/dalpha through the scalar recurrence)
//     dF[g][ch] = sum_p   w[g][p]   * dL[p][ch]        (the colour/feature gradient)
//     M[g][j]   = sum_p   u[g][p]   * Phi[p][j]        (six pixel moments 1, x, y, x^2, xy, y^2 of
//                                                        u = dL/dG * G, from which dL/dmean2D, dL/dconic and
//                                                        dL/dopacity of the Gaussian follow in closed form)
// A wave (8x8 pixels) therefore processes its records in chunks of 16 rows:
//   1. S for 16 rows x 64 pixels: 32 v_mfma_f32_16x16x4_f32 (A = feature rows read from LDS as MFMA operands,
//      B = dL held in registers), then a transpose through LDS so that lane = pixel again;
//   2. the short scalar recurrences per pixel (T, R, dL/dalpha: ~28 VALU per row), which leave w and u of the
//      16 rows in LDS, transposed;
//   3. dF (32 MFMA) and M (16 MFMA) with A = w / u rows from LDS and B = dL (registers) / Phi (generated on the fly);
//   4. per row ONE 128-byte line of float atomics for dF and ONE 32-byte packed record {mean2D.xy, conic.xyw,
//      opacity} -- rows without any contributing pixel are skipped.
// f32 MFMA is an exact fmaf chain (no reduced precision).  Moments are taken about the quadrant centre, so
// |coordinate| <= 3.5 and recentring to the Gaussian mean is benign (all terms of a sum share their sign pattern
// with the direct evaluation).
// Staging (records, feature rows) is per batch of RB2 records; for C = 32 it is pipelined over the batches (BwdCfg::PIPE).
#pragma once

#include "blend_bwd_shared.h"
#include "blend_fwd.h"
#include "common.h"

namespace mirast {

// Per channel count: blend-list records per batch (LDS budget of 3 workgroups per CU at C = 32, 2 at C = 64),
// padded feature row (floats; conflict-free 16-lane b128 operand reads), waves per SIMD the registers allow.
template <int C>
struct BwdCfg {
    static constexpr int RB2 = C == 32 ? 64 : 128;  // C == 16 is the RGB kernel (3 real channels padded to one MFMA block)
    static constexpr int FROW = C + 4;
    static constexpr int NBITS = (RB2 + 63) / 64;
    static constexpr int FEAT4 = (RB2 + 1) * FROW / 4;  // + one all-zero row for the list padding
    static constexpr int POOL4 = FEAT4 + 2 * 4 * CHK * WROW / 4;
    static constexpr int WAVES = C == 64 ? 2 : 3;
    // C = 32: the feature rows of the NEXT batch are requested while this batch is processed (ids fetched a batch earlier
    // still) and wait in registers, so no batch stops for a dependent global access.  What pays for the 10 registers:
    // records are fetched as 8-byte quarters by all 256 threads (2 registers instead of 8), S is read row by row.
    static constexpr bool PIPE = C == 32;
    static_assert(POOL4 * 4 >= 4 * 64 * DLROW, "gradient-image staging must fit in the aliased buffers");
    static_assert(RB2 + FROW / 4 <= 256, "staging roles are assigned by thread index");
};

// VALU issue is the limiter of this kernel: a wave issues one VALU instruction per ~8 cycles, a SIMD reaches
// 4.25 / 3.5 / 2.7 cycles per instruction with 2 / 3 / 4 resident waves, and an f32 MFMA occupies the same ALUs
// for its full 32 cycles (tools/valu_rate_probe.hip, tools/interleave_probe.hip, tools/overlap_probe.hip).  Hence:
// few instructions per (row, pixel), row parameters fetched by LDS broadcast reads (not VALU), no selects where
// arithmetic with alpha = 0 does the same, and the register budget of 3 waves per SIMD.
// C: channels as the MFMA tiling sees them (16, 32, 64); CR: channels in memory (CR == C, or 3 for RGB padded to C = 16).
// MASKGRAD (RGB only): the DEPTH variant's dL_dmask (DEPTH/cuda_rasterizer/backward.cu:457,516) rides as channel CR of
// the dF contraction only -- the mask's image gradient does not enter dL/dalpha -- and lands in field 6 of gpack.
template <int C, int CR = C, bool MASKGRAD = false>
__global__ void __launch_bounds__(256, BwdCfg<C>::WAVES) blend_bwd_mfma_kernel(
    const uint2* __restrict__ ranges, const BlendRec* __restrict__ blend_rec, const uint32_t* __restrict__ tile_nsurv,
    int W, int H, const float* __restrict__ bg_color, const float* __restrict__ colors,
    const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
    const float* __restrict__ dL_dout_mask,
    float* __restrict__ gpack /*[P,8] packed field gradients*/, float* __restrict__ dL_dcolors,
    int ablate /* timing experiments only; 0 in production */)
{
    constexpr int RB2 = BwdCfg<C>::RB2, FROW = BwdCfg<C>::FROW, NBITS = BwdCfg<C>::NBITS, FEAT4 = BwdCfg<C>::FEAT4;
    constexpr int POOL4 = BwdCfg<C>::POOL4;
    constexpr int CPL = C / 4;   // channels per lane in the S contraction: lane (n16, kq) holds channels CPL*kq .. +CPL-1
    constexpr int NB = C / 16;   // 16-channel blocks of the dF contraction
    constexpr int F4 = C / 4;    // float4s per feature row
    __shared__ BwdPar s_par[RB2 + 1];                // [RB2] = padding record (never valid)
    __shared__ float4 s_pool[POOL4];                 // feature rows | w rows | u rows  (prologue: gradient-image staging)
    __shared__ uint64_t s_bits[4][NBITS];
    __shared__ uint32_t s_list[4][RB2 + CHK];        // per quadrant: BYTE OFFSETS of its records in s_par, back to front
    __shared__ float4 s_mom4[4][CHK * 8 / 4];
    __shared__ int s_Lt[4];
    float4* const s_feat4 = s_pool;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const uint32_t horizontal_blocks = (W + TILE_X - 1) / TILE_X;
    const uint32_t tile = blockIdx.y * horizontal_blocks + blockIdx.x;
    const uint32_t qx0 = blockIdx.x * TILE_X + (wave & 1) * 8, qy0 = blockIdx.y * TILE_Y + (wave >> 1) * 8;
    const uint32_t px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const uint32_t pix_id = W * py + px;
    const float pixfx = (float)px, pixfy = (float)py;
    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    const size_t HW = (size_t)H * W;

    long long tk[6] = {0, 0, 0, 0, 0, 0};
    int n_chunks = 0, n_empty = 0, n_pad = 0, n_rows_live = 0;  // profiling counters (MI_ABLATE(32), profiling build only)
    const bool prof = MI_ABLATE(32);
    long long tmark = prof ? clock64() : 0;
#define TK(i) do { if (prof) { const long long t_ = clock64(); tk[i] += t_ - tmark; tmark = t_; } } while (0)
    // Everything the tile needs from memory is requested up front (tile header, per-pixel state, the gradient
    // image); the first batch of records follows as soon as the header is back.  A dependent global access costs
    // ~4 us in this kernel, so the prologue is organised as two round trips, not five.
    const uint2 range = ranges[tile];
    const int NS = (int)tile_nsurv[tile];
    const size_t pix_safe = inside ? pix_id : 0;
    const int last_contributor = inside ? (int)n_contrib[pix_safe] : 0;
    const float T_final = inside ? final_Ts[pix_safe] : 0;
    static_assert(CR == C || (C == 16 && CR == 3), "padded layout is the RGB case only");
    static_assert(!MASKGRAD || CR == 3, "the mask gradient belongs to the RGB (DEPTH variant) kernel");
    float dLpix[C];
#pragma unroll
    for (int ch = 0; ch < C; ch++) {
        if (ch < CR) dLpix[ch] = dL_dpixels[(size_t)ch * HW + pix_safe];
        else if (MASKGRAD && ch == CR) dLpix[ch] = dL_dout_mask[pix_safe];
        else dLpix[ch] = 0.f;
    }
    const BlendRec* rec = blend_rec + range.x;
    constexpr bool PIPE = BwdCfg<C>::PIPE;
    constexpr int NK = (RB2 * (C / 4) + BATCH - 1) / BATCH;  // float4 feature gathers per thread and batch
    static_assert(!PIPE || RB2 * 4 == BATCH, "PIPE: one 8-byte record quarter per thread");
    BlendRec cur;
    uint2 curq = make_uint2(0u, 0u);  // PIPE: quarter (tid & 3) of record (tid >> 2) of the batch
    float4 featpf[PIPE ? NK : 1];     // PIPE: this thread's parts of the next batch's feature rows
    uint32_t nid[PIPE ? NK : 1];      // PIPE: Gaussian ids of the batch after that
    if constexpr (PIPE) {
        if ((tid >> 2) < NS) curq = reinterpret_cast<const uint2*>(rec + (NS - 1 - (tid >> 2)))[tid & 3];
#pragma unroll
        for (int k = 0; k < NK; k++) {
            const int g = (tid + BATCH * k) / (C / 4);
            nid[k] = (g < RB2 && g < NS) ? rec[NS - 1 - g].id : 0u;
        }
    } else {
        if (tid < RB2 && tid < NS) cur = rec[NS - 1 - tid];
    }
    int wave_Lt = last_contributor;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) wave_Lt = max(wave_Lt, __shfl_xor(wave_Lt, o, 64));
    if (lane == 0) s_Lt[wave] = wave_Lt;
    float T = T_final;
    TK(0);

    // ---- gradient image of this quadrant: one coalesced pass (lane = pixel, 32 loads), staged through LDS into
    // the two MFMA operand layouts.  The staging rows alias the feature / w / u buffers, which are first written
    // after the barrier that opens the batch loop.
    const int n16 = lane & 15, kq = lane >> 4;
    float dLB[4][CPL];  // B of the S contraction:  dLB[pb][s] = dL[pixel 16*pb + n16][channel CPL*kq + s]
    float dLT[NB][16];  // B of the dF contraction: dLT[nb][s] = dL[pixel 16*kq + s][channel 16*nb + n16]
    float bg_dot_dpixel = 0.f;  // bg . dL of this lane's own pixel (backward.cu:533-535)
    {
        float* stage = reinterpret_cast<float*>(s_pool) + wave * (64 * DLROW);
        constexpr int PASS = C < 32 ? C : 32;  // channels per pass through the staging rows
#pragma unroll
        for (int h = 0; h < C / PASS; h++) {
#pragma unroll
            for (int c = 0; c < PASS; c++) {
                const float v = inside ? dLpix[PASS * h + c] : 0.f;
                if (PASS * h + c < CR) bg_dot_dpixel += bg_color[PASS * h + c] * v;
                stage[lane * DLROW + c] = v;
            }
            if (h == 0) {
                __syncthreads();  // also publishes s_Lt
                if (max(max(s_Lt[0], s_Lt[1]), max(s_Lt[2], s_Lt[3])) == 0) return;
            } else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            }
            // lanes whose CPL channels lie in this pass (all of them when C <= 32)
            const bool mine = (CPL * kq) / PASS == h;
            const int c0 = (CPL * kq) % PASS;
#pragma unroll
            for (int pb = 0; pb < 4; pb++)
#pragma unroll
                for (int s = 0; s < CPL; s++) {
                    float v = stage[(16 * pb + n16) * DLROW + (mine ? c0 + s : s)];
                    if (MASKGRAD && CPL * kq + s == CR) v = 0.f;  // the mask plane is no part of S
                    if (C <= 32 || mine) dLB[pb][s] = v;
                }
#pragma unroll
            for (int s = 0; s < 16; s++)
#pragma unroll
                for (int nb = 0; nb < PASS / 16; nb++) dLT[(PASS / 16) * h + nb][s] = stage[(16 * kq + s) * DLROW + 16 * nb + n16];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }
    }
    TK(1);
    const float nTb = -T_final * bg_dot_dpixel;  // the background term of dL/dalpha is nTb / (1 - alpha)
    const int last4 = last_contributor << 4, wave_Lt4 = wave_Lt << 4;  // compared with (position << 4 | mask)

    // Phi[pixel 16kq+s][j = n16] = monomial j (1, x, y, x^2, xy, y^2) about the quadrant centre, x = (s&7) - 3.5 (a
    // literal per unrolled step), y = 2kq - 3.5 + (s>>3):  phi = P[s>>3] + Q[s>>3] x + R x^2  (exact: small dyadics)
    float phP[2], phQ[2], phR;
    {
        const float c1 = n16 == 0, cx = n16 == 1, cy = n16 == 2, cxx = n16 == 3, cxy = n16 == 4, cyy = n16 == 5;
#pragma unroll
        for (int v = 0; v < 2; v++) {
            const float y = (float)(2 * kq + v) - 3.5f;
            phP[v] = c1 + cy * y + cyy * y * y;
            phQ[v] = cx + cxy * y;
        }
        phR = cxx;
    }

    float Rcur = 0.f;  // sum over the Gaussians behind the current one of (their colour . dL) * their share of what is behind
    const float ddelx_dx = 0.5 * W;  // backward.cu:460-461
    const float ddely_dy = 0.5 * H;
    const float cxq = (float)qx0 + 3.5f, cyq = (float)qy0 + 3.5f;  // moment origin: quadrant centre
    float* my_wa = reinterpret_cast<float*>(s_pool + FEAT4 + wave * (CHK * WROW / 4));
    float* my_ua = reinterpret_cast<float*>(s_pool + FEAT4 + (4 + wave) * (CHK * WROW / 4));
    float* my_mom = reinterpret_cast<float*>(s_mom4[wave]);
    const char* const par_bytes = reinterpret_cast<const char*>(s_par);

    if constexpr (PIPE) {  // first batch's feature parts, second batch's ids
#pragma unroll
        for (int k = 0; k < NK; k++) {
            const int e = tid + BATCH * k;
            const int g = e / F4, part = e % F4;
            featpf[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g < RB2 && g < NS) featpf[k] = reinterpret_cast<const float4*>(colors + (size_t)nid[k] * C)[part];
        }
#pragma unroll
        for (int k = 0; k < NK; k++) {
            const int g = (tid + BATCH * k) / F4;
            if (g < RB2 && RB2 + g < NS) nid[k] = rec[NS - 1 - (RB2 + g)].id;
        }
    }
    for (int b0 = 0; b0 < NS; b0 += RB2) {
        const int nr = min(RB2, NS - b0);  // records in this batch, walked back to front
        TK(4);
        __syncthreads();                   // LDS reuse (first batch: the gradient-image staging reads are done)
        // ---- A: records -> LDS; next batch's record -> registers; quadrant bitmaps
        if constexpr (PIPE) {
            // Staging indices and addresses are derived from an opaque copy of the thread index: hipcc otherwise keeps a
            // dozen of them live across the chunk loop, spills them, and every reload (scratch shares vmcnt) stalls the
            // requests below one by one.
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            // quarter 0 = {x, y} -> bytes 0..7; 1 = {id, pm} -> {pm, id} at 24; 2 = {a, b} -> {-a/2, -b} at 8; 3 = {c, opacity} -> {-c/2, opacity} at 16
            const int rq = tid >> 2, qq = tid & 3;
            if (rq < nr) {
                float2 v = make_float2(__uint_as_float(curq.x), __uint_as_float(curq.y));
                if (qq == 1) v = make_float2(__uint_as_float(curq.y), __uint_as_float(curq.x));
                if (qq == 2) v = make_float2(-0.5f * v.x, -v.y);
                if (qq == 3) v = make_float2(-0.5f * v.x, v.y);
                const int dst = qq == 0 ? 0 : (qq == 1 ? 24 : (qq == 2 ? 8 : 16));
                *reinterpret_cast<float2*>(reinterpret_cast<char*>(&s_par[rq]) + dst) = v;
            }
#pragma unroll
            for (int k = 0; k < NK; k++) {  // this batch's feature rows arrive from registers
                const int e = tid + BATCH * k;
                const int g = e / F4, part = e % F4;
                if (g < nr) s_feat4[g * (FROW / 4) + part] = featpf[k];
            }
            // Requests for the next batch (feature rows: ids known; records) and the ids of the one after.  The ids are
            // the youngest loads of the previous round: they are consumed (one wait, long satisfied) before anything
            // new is requested, so that nothing below waits on a request of this round.
            uint32_t idk[NK];
#pragma unroll
            for (int k = 0; k < NK; k++) idk[k] = nid[k];
#pragma unroll
            for (int k = 0; k < NK; k++) asm volatile("" : "+v"(idk[k]));
#pragma unroll
            for (int k = 0; k < NK; k++) {
                const int e = tid + BATCH * k;
                const int g = e / F4, part = e % F4;
                if (g < RB2 && b0 + RB2 + g < NS) featpf[k] = reinterpret_cast<const float4*>(colors + (size_t)idk[k] * C)[part];
            }
            if (b0 + RB2 + rq < NS) curq = reinterpret_cast<const uint2*>(rec + (NS - 1 - (b0 + RB2 + rq)))[qq];
#pragma unroll
            for (int k = 0; k < NK; k++) {
                const int g = (tid + BATCH * k) / F4;
                if (g < RB2 && b0 + 2 * RB2 + g < NS) nid[k] = rec[NS - 1 - (b0 + 2 * RB2 + g)].id;
            }
        } else if (tid < nr) {
            s_par[tid].q0 = make_float4(cur.xy.x, cur.xy.y, -0.5f * cur.co.x, -cur.co.y);
            s_par[tid].q1 = make_float4(-0.5f * cur.co.z, cur.co.w, __int_as_float((int)cur.pm), __int_as_float((int)cur.id));
        }
        if ((!PIPE || b0 == 0) && tid == RB2) {  // padding record: never valid ...
            s_par[RB2].q0 = make_float4(0.f, 0.f, -0.5f, 0.f);
            s_par[RB2].q1 = make_float4(-0.5f, 0.f, __int_as_float(0x7ffffff0), __int_as_float(0));
        }
        if ((!PIPE || b0 == 0) && tid >= 256 - FROW / 4)  // ... and its all-zero feature row (S = 0)
            s_feat4[RB2 * (FROW / 4) + (tid - (256 - FROW / 4))] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!PIPE && wave < NBITS) {
            const uint32_t pmv = tid < nr ? cur.pm : 0u;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint64_t b = ballot64((pmv >> q) & 1u);
                if (lane == 0) s_bits[q][wave] = b;
            }
        }
        TK(2);
        __syncthreads();
        // ---- B: feature rows (padded to FROW floats), gathered by the ids just staged.  All the loads are issued before
        // the first LDS write (unconditionally: a clamped row index keeps the address valid) -- written as one guarded
        // load-then-store per k, hipcc emits a full vmcnt(0) round trip per k.
        if constexpr (!PIPE) {
            float4 v[NK];
#pragma unroll
            for (int k = 0; k < NK; k++) {
                const int e = tid + BATCH * k;
                const int g = e / F4, part = e % F4;
                const size_t gid = (size_t)__float_as_int(s_par[g < nr ? g : 0].q1.w);
                if constexpr (CR == C) {
                    v[k] = reinterpret_cast<const float4*>(colors + gid * C)[part];
                } else {  // RGB: three floats per Gaussian, the other 13 operand channels are zero
                    v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (part == 0) v[k] = make_float4(colors[gid * 3 + 0], colors[gid * 3 + 1], colors[gid * 3 + 2], 0.f);
                }
            }
            // pins every loaded value in registers here: hipcc otherwise sinks each load into the guarded store below
            #pragma unroll
            for (int k = 0; k < NK; k++) asm volatile("" : "+v"(v[k].x), "+v"(v[k].y), "+v"(v[k].z), "+v"(v[k].w));
#pragma unroll
            for (int k = 0; k < NK; k++) {
                const int e = tid + BATCH * k;
                const int g = e / F4, part = e % F4;
                if (g < nr && !MI_ABLATE(4)) s_feat4[g * (FROW / 4) + part] = v[k];
            }
        }
        // The next batch's records are requested only now: vmcnt retires in order, so a request issued before the
        // gather would make the gather wait for it as well.
        if (!PIPE && tid < RB2 && b0 + RB2 + tid < NS) cur = rec[NS - 1 - (b0 + RB2 + tid)];
        if constexpr (!PIPE) __syncthreads();

        // ---- this quadrant's rows, back to front, restricted to positions below the quadrant's max n_contrib;
        //      the list is padded with CHK references to the padding record
        int cnt = 0;
#pragma unroll
        for (int h = 0; h < NBITS; h++) {
            const int ridx = 64 * h + lane;
            const int pmv = __float_as_int(s_par[ridx < nr ? ridx : 0].q1.z);
            const bool cand = (PIPE ? ((pmv >> wave) & 1) != 0 : ((s_bits[wave][h] >> lane) & 1ull) != 0) && ridx < nr && pmv < wave_Lt4;
            const uint64_t b = ballot64(cand);
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
            if (cand) s_list[wave][cnt + below] = (uint32_t)(ridx * sizeof(BwdPar));
            cnt += __builtin_popcountll(b);
        }
        cnt = __builtin_amdgcn_readfirstlane(cnt);
        if (lane < CHK) s_list[wave][cnt + lane] = (uint32_t)(RB2 * sizeof(BwdPar));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if MI_ABLATE(1) cnt = 0;
        TK(3);
        for (int j0 = 0; j0 < cnt; j0 += CHK) {
            // ---- 1. S = F . dL^T  (16 rows x 64 pixels, K = 32 channels); lane (n16, kq) feeds row n16
            const uint32_t my_off = s_list[wave][j0 + n16];
            v4f sacc[4];
            {
                const uint32_t km = my_off / (uint32_t)sizeof(BwdPar);
                float fa[CPL];
#pragma unroll
                for (int q = 0; q < CPL / 4; q++) {
                    const float4 f = s_feat4[km * (FROW / 4) + (CPL / 4) * kq + q];
                    fa[4 * q + 0] = f.x;
                    fa[4 * q + 1] = f.y;
                    fa[4 * q + 2] = f.z;
                    fa[4 * q + 3] = f.w;
                }
#pragma unroll
                for (int pb = 0; pb < 4; pb++) sacc[pb] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < CPL; s++)
#pragma unroll
                    for (int pb = 0; pb < 4; pb++)
                        sacc[pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s], dLB[pb][s], sacc[pb], 0, 0, 0);
            }
            // Transpose to lane = pixel through LDS (the w-row buffer is free until step 2 writes it): lane 16g+p
            // holds row 4g+r of pixel 16pb+p in sacc[pb][r]; both the writes and the row reads are conflict-free
            // (row stride 68 floats).  (A register-only v_permlane16/32_swap transpose is 16 instructions, but
            // hipcc 7.2 miscompiles that builtin sequence -- tools/s_probe.hip -- so LDS it is.)
            // Row m of S is read when step 2 reaches row m, just before w of the same row overwrites it (same address).
            {
#pragma unroll
                for (int pb = 0; pb < 4; pb++)
#pragma unroll
                    for (int r = 0; r < 4; r++) my_wa[(4 * kq + r) * WROW + 16 * pb + n16] = sacc[pb][r];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            }

            // ---- 2. scalar recurrences (lane = pixel), back to front.  Row parameters arrive by LDS broadcast reads.
            // A row that does not blend into this pixel runs the same arithmetic with alpha = 0: T, R stay put, w = u = 0.
            uint32_t rowmask = 0;
#pragma unroll
            for (int rr = 0; rr < CHK; rr++) {
                const uint32_t off = s_list[wave][j0 + rr];
                const float4 p0 = *reinterpret_cast<const float4*>(par_bytes + off);
                const float4 p1 = *reinterpret_cast<const float4*>(par_bytes + off + 16);
                const float dx = p0.x - pixfx, dy = p0.y - pixfy;
                const float power = gauss_power(p0.z, p0.w, p1.x, dx, dy);
                const float G = gauss_exp<false>(power);
                // opacity * G where the row can blend into this pixel at all, else 0; the 1/255 cut is the last test so
                // that its compare doubles as the ballot (min(0.99, t) >= 1/255  <=>  t >= 1/255)
                const float t0 = ((__float_as_int(p1.z) < last4) && power <= 0.0f) ? p1.y * G : 0.f;
                const bool valid = t0 >= (1.0f / 255.0f);
                const float tG = valid ? t0 : 0.f;   // opacity * G of a contributing row
                const float alpha = fminf(0.99f, tG);
                const float om = 1.f - alpha;
                const float inv = __builtin_amdgcn_rcpf(om);
                T = T * inv;
                const float w = alpha * T;  // dchannel_dcolor
                const float dS = my_wa[rr * WROW + lane] - Rcur;
                const float dL_dalpha = fmaf(nTb, inv, dS * T);
                Rcur = fmaf(alpha, dS, Rcur);  // = alpha S + (1 - alpha) Rcur
                const float u = tG * dL_dalpha;  // dL/dG * G  (dL/dG = opacity * dL/dalpha, clamp ignored as in the reference)
                rowmask |= (ballot64(valid) != 0 ? 1u : 0u) << rr;
                my_wa[rr * WROW + lane] = w;
                my_ua[rr * WROW + lane] = u;
            }
            rowmask = __builtin_amdgcn_readfirstlane(rowmask);
            if (prof) {
                n_chunks++;
                n_empty += rowmask == 0 ? 1 : 0;
                n_pad += max(0, j0 + CHK - cnt);
                n_rows_live += __builtin_popcount(rowmask);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (rowmask == 0 || MI_ABLATE(2)) continue;

            // ---- 3. dF = W^T . dL  and  M = U^T . Phi   (A rows from LDS, lane (m = n16, kq) reads pixels 16kq..16kq+15)
            v4f facc[NB];
#pragma unroll
            for (int nb = 0; nb < NB; nb++) facc[nb] = (v4f){0.f, 0.f, 0.f, 0.f};
            v4f macc = (v4f){0.f, 0.f, 0.f, 0.f};
            {
                const float4* wrow = reinterpret_cast<const float4*>(my_wa + n16 * WROW + 16 * kq);
                const float4* urow = reinterpret_cast<const float4*>(my_ua + n16 * WROW + 16 * kq);
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++) {
                    const float4 wv = wrow[s4];
                    const float4 uv = urow[s4];
                    const float wa[4] = {wv.x, wv.y, wv.z, wv.w};
                    const float ua[4] = {uv.x, uv.y, uv.z, uv.w};
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const int s = 4 * s4 + t;
#pragma unroll
                        for (int nb = 0; nb < NB; nb++)
                            facc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t], dLT[nb][s], facc[nb], 0, 0, 0);
                        const float x = (float)(s & 7) - 3.5f;
                        const float phi = fmaf(x, fmaf(x, phR, phQ[s >> 3]), phP[s >> 3]);
                        macc = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[t], phi, macc, 0, 0, 0);
                    }
                }
            }
            // ---- 4. outputs.  Result layout: lane l holds column n16 of rows 4*kq + r.
            const BwdPar mine = *reinterpret_cast<const BwdPar*>(par_bytes + my_off);  // lane n16 = row n16
            const int my_gid = __float_as_int(mine.q1.w);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 4 * kq + r;
                const bool act = (rowmask >> row) & 1u;
                const uint32_t gid = (uint32_t)__shfl(my_gid, row, 64);
                if (act && !MI_ABLATE(64)) {
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) {
                        const int ch = 16 * nb + n16;
                        if (ch < CR) atomicAdd(&dL_dcolors[(size_t)gid * CR + ch], facc[nb][r]);
                        else if (MASKGRAD && ch == CR) atomicAdd(&gpack[(size_t)gid * 8 + 6], facc[nb][r]);
                    }
                }
                if (n16 < 8) my_mom[row * 8 + n16] = macc[r];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            // moments -> fields: lane = row (16 lanes) rewrites its my_mom row in place, then the wave adds the rows
            // to the packed per-Gaussian records: lane -> (row = l / 8 (+8), field = l % 8), 24 contiguous bytes per row
            {
                const int row = lane & 15;
                const bool act = lane < 16 && ((rowmask >> row) & 1u);
                if (act) {
                    const float4 m0 = reinterpret_cast<const float4*>(my_mom + row * 8)[0];
                    const float4 m1 = reinterpret_cast<const float4*>(my_mom + row * 8)[1];
                    const float M0 = m0.x, M1 = m0.y, M2 = m0.z, M3 = m0.w, M4 = m1.x, M5 = m1.y;
                    const float ca = -2.f * mine.q0.z, cb = -mine.q0.w, cc = -2.f * mine.q1.x, op = mine.q1.y;
                    const float gx = mine.q0.x - cxq, gy = mine.q0.y - cyq;
                    // dx = gx - x', dy = gy - y'
                    const float Sdx = gx * M0 - M1;
                    const float Sdy = gy * M0 - M2;
                    const float Sdxx = gx * gx * M0 - 2.f * gx * M1 + M3;
                    const float Sdxy = gx * gy * M0 - gx * M2 - gy * M1 + M4;
                    const float Sdyy = gy * gy * M0 - 2.f * gy * M2 + M5;
                    float4 o0, o1;
                    o0.x = -ddelx_dx * (ca * Sdx + cb * Sdy);  // dL_dmean2D.x
                    o0.y = -ddely_dy * (cc * Sdy + cb * Sdx);  // dL_dmean2D.y
                    o0.z = -0.5f * Sdxx;                       // dL_dconic.x
                    o0.w = -0.5f * Sdxy;                       // dL_dconic.y
                    o1.x = -0.5f * Sdyy;                       // dL_dconic.w
                    o1.y = M0 * __builtin_amdgcn_rcpf(op);     // dL_dopacity = sum G dL_dalpha
                    o1.z = 0.f;
                    o1.w = 0.f;
                    reinterpret_cast<float4*>(my_mom + row * 8)[0] = o0;
                    reinterpret_cast<float4*>(my_mom + row * 8)[1] = o1;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
                for (int it = 0; it < 2; it++) {
                    const int row2 = 8 * it + (lane >> 3), f = lane & 7;
                    const uint32_t gid2 = (uint32_t)__shfl(my_gid, row2, 64);
                    if (((rowmask >> row2) & 1u) && f < 6 && !MI_ABLATE(128)) atomicAdd(&gpack[(size_t)gid2 * 8 + f], my_mom[row2 * 8 + f]);
                }
            }
        }
    }
    TK(4);
    if (prof && lane == 0)
    {
        for (int i = 0; i < 5; i++) atomicAdd(&gpack[8 * (i + 1) + 6], (float)tk[i]);
        atomicAdd(&gpack[8 * 6 + 6], (float)n_chunks);
        atomicAdd(&gpack[8 * 7 + 6], (float)n_empty);
        atomicAdd(&gpack[8 * 8 + 6], (float)n_pad);
        atomicAdd(&gpack[8 * 9 + 6], (float)n_rows_live);
    }
#undef TK
}

}  // namespace mirast
