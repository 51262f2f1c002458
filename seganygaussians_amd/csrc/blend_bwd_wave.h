// blend_bwd_wave.h -- per-QUADRANT back-to-front gradient pass: one wave per 8x8 pixel quadrant, no workgroup barriers.
//
// Same mathematics and the same 16-row MFMA chunk as blend_bwd_mfma.h (renderCUDA<C> backward,
// CF/cuda_rasterizer/backward.cu:399-559, with the three contractions S = F dL^T, dF = W^T dL, M = U^T Phi on
// v_mfma_f32_16x16x4_f32); what changes is who walks the tile's blend list and when.
//
// blend_bwd_mfma.h stages the list in batches of 64 records for the whole tile; each of the four waves then picks its
// quadrant's records out of the batch and pads them to a multiple of 16 rows PER BATCH: on cfg3 18 % of all chunk rows
// are padding, the waves of a tile meet at two barriers per batch (14 % of wave time), and a wave with few rows idles
// while the tile's busiest quadrant works.  Here a workgroup IS one wave:
//   * the wave scans the tile's blend list itself, 64 entries at a time (four bytes each: Gaussian id | quadrant mask << 28,
//     binning.h; prefetched one block ahead), and appends the entries of its quadrant to a ring in LDS (128 entries = 8 chunks
//     of run-ahead), so chunks are always full -- only the wave's very last chunk is padded;
//   * the 16 rows of the NEXT chunk (one 8-byte quarter of the geometry record index_rec[id] per lane + two float4 feature
//     parts per lane, both gathered by the id) are requested before the current chunk is processed and wait in registers;
//   * every chunk issues the SAME number of memory instructions (the gradient atomics are unconditional: rows without
//     a contributing pixel add exact zeros), so the in-order vmcnt wait for the staged rows never has to cover the
//     atomics issued after them;
//   * no __syncthreads anywhere; the quadrants of a tile are separate workgroups mapped to the same XCD
//     (blockIdx -> (tile, quadrant) below) so that they share the tile's records and feature rows in one L2.
#pragma once

#include <type_traits>

#include "blend_bwd_shared.h"
#include "blend_fwd.h"
#include "common.h"

namespace mirast {


// 1 / x to ~0.5 ulp: v_rcp_f32 (1 ulp) plus one Newton step (two FMAs).  T is divided by (1 - alpha) once per row and the
// quotients are chained through the whole list: the reference uses a correctly rounded division there (backward.cu:487).
__device__ __forceinline__ float rcp_refined(float x)
{
    const float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.0f), r, r);
}
// min(0.99, t) for t >= -1 as ONE instruction: v_med3_f32.  fminf() on a value that comes out of a select costs two (hipcc
// canonicalises it first, v_max_f32 x, x, because the kernel runs in IEEE mode); the result is the same operand either way.
__device__ __forceinline__ float alpha_clamp(float t) { return __builtin_amdgcn_fmed3f(t, 0.99f, -1.0f); }

template <int C>
struct BwvCfg {
    static constexpr int FROW = C + 4;           // padded feature row (floats): conflict-free 16-lane b128 operand reads
    static constexpr int WAVES = C == 64 ? 2 : 3;
    static constexpr int QCAP = 128;             // ring of {record index, Gaussian id}
    static constexpr int FEAT4 = CHK * FROW / 4;
};

// C: channels as the MFMA tiling sees them (16, 32, 64); CR: channels in memory (CR == C; 3 for RGB padded to C = 16; 0: `cr_arg`
// (1 .. 15) of a 16-channel block are real -- the last block of a feature whose width is no multiple of 16, the reference compiles
// ANY NUM_CHANNELS (config_contrastive_f.h:15) -- with rows `cstride_arg` floats apart).
// STRIDED: rows of `colors` / `dL_dcolors` are `cstride_arg` floats apart (one channel block of a wider feature); otherwise CR.
template <int C, int CR = C, bool MASKGRAD = false, bool XEXP = false, bool STRIDED = false>
__global__ void __launch_bounds__(64, BwvCfg<C>::WAVES) blend_bwd_wave_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ blend_list, const BlendRec* __restrict__ index_rec,
    const uint32_t* __restrict__ tile_nsurv, int W, int H, uint32_t horizontal_blocks, uint32_t ntiles, const float* __restrict__ bg_color,
    const float* __restrict__ colors, const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dpixels, const float* __restrict__ dL_dout_mask,
    float* __restrict__ gpack /*[P,8] packed field gradients*/, float* __restrict__ dL_dcolors,
    uint32_t* __restrict__ queue_ctr /* eight zeroed work-queue counters (common.h: xcd_grab) */,
    int cstride_arg /* STRIDED: floats between the rows of two Gaussians in `colors` and `dL_dcolors` = the full channel count of
                       the feature this launch handles one channel block of (both pointers then point at the block) */,
    int cr_arg /* CR == 0: channels of this block that exist in memory */,
    const uint32_t* __restrict__ run_bounds /* [9]: the XCDs' runs of tiles (common.h: XcdRuns), left in the image buffer by the forward */)
{
    constexpr int FROW = BwvCfg<C>::FROW, QCAP = BwvCfg<C>::QCAP, FEAT4 = BwvCfg<C>::FEAT4;
    const int cstride = STRIDED ? cstride_arg : CR;   // (a compile-time constant in the common case: no 64-bit multiply, no extra registers)
    const int cr = CR == 0 ? cr_arg : CR;             // channels in memory (a constant unless CR == 0)
    constexpr int CPL = C / 4;   // channels per lane in the S contraction: lane (n16, kq) holds channels CPL*kq .. +CPL-1
    constexpr int NB = C / 16;   // 16-channel blocks of the dF contraction
    constexpr int F4 = C / 4;    // float4s per feature row
    constexpr int NK = (CHK * F4 + 63) / 64;  // float4 feature parts per lane and chunk
    constexpr int MROW = 16;   // floats per row of the moment / field staging: 6 moments, then 8 fields
    static_assert(CR == C || (C == 16 && (CR == 3 || CR == 0)), "padded layouts: RGB, or a partial 16-channel block");
    static_assert(CR != 0 || STRIDED, "a partial block is a block of a wider (or narrower) feature: its row stride is an argument");
    static_assert(!MASKGRAD || CR == 3, "the mask gradient belongs to the RGB (DEPTH variant) kernel");
    static_assert(2 * CHK * WROW >= 64 * DLROW, "gradient-image staging must fit in the w/u rows");

    __shared__ BwdPar s_par[CHK];                 // the chunk's 16 records
    __shared__ float4 s_feat4[FEAT4];             // the chunk's 16 feature rows
    __shared__ float4 s_wu4[2 * CHK * WROW / 4];  // S / w rows | u rows   (prologue: gradient-image staging; after step 3: moments)
    __shared__ uint2 s_queue[QCAP];               // {walk index, entry = Gaussian id | quadrant mask << 28}

    // One (tile, quadrant) item: everything below.  Which item a wave gets is decided at the end of the kernel.
    auto quadrant = [&](const uint32_t tile, const uint32_t quad) __attribute__((always_inline)) {
    const uint32_t tile_x = tile % horizontal_blocks, tile_y = tile / horizontal_blocks;

    const int lane = threadIdx.x & 63;
    const uint32_t qx0 = tile_x * TILE_X + (quad & 1) * 8, qy0 = tile_y * TILE_Y + (quad >> 1) * 8;
    const uint32_t px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const uint32_t pix_id = W * py + px;
    const float pixfx = (float)px, pixfy = (float)py;
    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    const size_t HW = (size_t)H * W;

    const uint2 range = ranges[tile];
    const int NS_tile = (int)tile_nsurv[tile];
    const size_t pix_safe = inside ? pix_id : 0;
    // Per-pixel state AND the gradient image are requested at once, before anything waits: the quadrant's largest
    // n_contrib (needed for the early exit and for where the walk starts) then costs no round trip of its own.
    // (all loads unconditional -- pix_safe is a valid pixel for every lane -- so that they stay in this block, in this order)
    const int nc_raw = (int)n_contrib[pix_safe];
    const float T_raw = final_Ts[pix_safe];
    float dLpix[C];
#pragma unroll
    for (int ch = 0; ch < C; ch++) {
        if (ch < cr) dLpix[ch] = dL_dpixels[(size_t)ch * HW + pix_safe];
        else if (MASKGRAD && ch == CR) dLpix[ch] = dL_dout_mask[pix_safe];
        else dLpix[ch] = 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);  // the scheduler would otherwise move the 32 image loads behind the wait for n_contrib
    const int last_contributor = inside ? nc_raw : 0;
    const float T_final = inside ? T_raw : 0.f;
    int wave_Lt = last_contributor;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) wave_Lt = max(wave_Lt, __shfl_xor(wave_Lt, o, 64));
    wave_Lt = __builtin_amdgcn_readfirstlane(wave_Lt);
    // List positions (n_contrib's unit) are indices into the tile's blend list: no entry at index >= wave_Lt can reach this quadrant.
    const int NS = min(NS_tile, wave_Lt);   // 0: nothing blended into this quadrant -- the wave leaves after the staging
    // below (not here: with an exit in between, hipcc sinks the gradient-image loads behind it, i.e. behind this wait)
    // entry = Gaussian id | quadrant mask << 28 (binning.h); entry j of the walk (back to front) is lst[NS - 1 - j]
    const uint32_t* lst = NS > 0 ? blend_list + range.x : blend_list;

    // first scan block.  Scan loads are unconditional (clamped index): a conditional assignment makes hipcc copy the
    // register right behind the load, i.e. wait for it on the spot.
    uint32_t scan_reg = lst[max(0, NS - 1 - min(lane, NS - 1))];
    __builtin_amdgcn_sched_barrier(0);  // ... and the scan load behind the gradient-image staging
    float T = T_final;

    // ---- gradient image of this quadrant: one coalesced pass (lane = pixel), staged through LDS into the two MFMA
    // operand layouts.  The staging rows alias the w / u rows, which are first written in the chunk loop.
    const int n16 = lane & 15, kq = lane >> 4;
    float dLB[4][CPL];  // B of the S contraction:  dLB[pb][s] = dL[pixel 16*pb + n16][channel CPL*kq + s]
    float dLT[NB][16];  // B of the dF contraction: dLT[nb][s] = dL[pixel 16*kq + s][channel 16*nb + n16]
    float bg_dot_dpixel = 0.f;  // bg . dL of this lane's own pixel (backward.cu:533-535)
    {
        float* stage = reinterpret_cast<float*>(s_wu4);
        constexpr int PASS = C < 32 ? C : 32;  // channels per pass through the staging rows
#pragma unroll
        for (int h = 0; h < C / PASS; h++) {
#pragma unroll
            for (int c = 0; c < PASS; c++) {
                const float v = inside ? dLpix[PASS * h + c] : 0.f;
                if (PASS * h + c < cr) bg_dot_dpixel += bg_color[PASS * h + c] * v;
                stage[lane * DLROW + c] = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
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
    if (NS == 0) return;
    const float nTb = -T_final * bg_dot_dpixel;  // the background term of dL/dalpha is nTb / (1 - alpha)
    const bool has_bg = ballot64(nTb != 0.f) != 0;   // wave-uniform
    const int last4 = last_contributor << 4;  // compared with (position << 4 | mask)

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
    // wave-uniform constants, pinned to SGPRs (as VGPRs they get spilled, and a scratch reload in the middle of a chunk waits
    // for every atomic in flight: scratch shares vmcnt)
    // (inline asm: the builtin is folded away for a value the compiler already knows to be uniform -- and stays in a VGPR.
    // hipcc inserts no wait states around inline asm: the s_nop covers the VALU-write -> readfirstlane hazard.)
    auto uniform = [](float v) {
        int r;
        asm volatile("s_nop 7\n\tv_readfirstlane_b32 %0, %1\n\ts_nop 3" : "=s"(r) : "v"(v));
        return __builtin_bit_cast(float, r);
    };
    const float ddelx_dx = uniform((float)(0.5 * W));  // backward.cu:460-461
    const float ddely_dy = uniform((float)(0.5 * H));
    const float cxq = uniform((float)qx0 + 3.5f), cyq = uniform((float)qy0 + 3.5f);  // moment origin: quadrant centre
    float* const my_wa = reinterpret_cast<float*>(s_wu4);
    float* const my_ua = my_wa + CHK * WROW;
    float* const my_mom = my_wa;  // the w rows are dead between step 3 and the next chunk's step 1
    const char* const par_bytes = reinterpret_cast<const char*>(s_par);

    // ---- the queue of this quadrant's records
    int scanned = 0, qh = 0, qt = 0;
    // Consumes the prefetched scan block [scanned, scanned + 64): candidates = entries whose quadrant bit is set (every position
    // walked is below the quadrant's largest n_contrib: NS <= wave_Lt); requests the next block.
    auto consume_scan = [&]() {
        const int j = scanned + lane;
        const bool cand = j < NS && ((scan_reg >> (ID_BITS + quad)) & 1u) != 0;
        const uint64_t bal = ballot64(cand);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (cand) s_queue[(qt + (int)below) & (QCAP - 1)] = make_uint2((uint32_t)j, scan_reg);
        qt += __builtin_popcountll(bal);
        scanned += 64;
        scan_reg = lst[NS - 1 - min(scanned + lane, NS - 1)];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    };
    // Requests the rows queue[qh .. qh + n), n >= 1, of the next chunk: record quarter (lane & 3) of row (lane >> 2) and the
    // feature parts.  Every lane loads (rows >= n repeat row n - 1 and are replaced when the chunk is staged), and the
    // indices derive from an opaque copy of the lane id: hipcc otherwise hoists the per-lane addresses out of the chunk
    // loop, spills them, and the reload (scratch shares vmcnt) drains every outstanding atomic once per chunk.
    uint2 curq;
    float4 featpf[NK];
    auto request_rows = [&](int n) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int l = threadIdx.x & 63;
        asm volatile("" : "+v"(l));
        {
            const int rq = min(l >> 2, n - 1), qq = l & 3;
            const uint32_t gq = s_queue[(qh + rq) & (QCAP - 1)].y & ID_MASK;
            curq = reinterpret_cast<const uint2*>(index_rec + gq)[qq];
        }
#pragma unroll
        for (int k = 0; k < NK; k++) {
            const int e = l + 64 * k;
            const int g = min(e / F4, n - 1), part = e % F4;
            const size_t gid = (size_t)(s_queue[(qh + g) & (QCAP - 1)].y & ID_MASK);
            if constexpr (CR == C) {
                featpf[k] = reinterpret_cast<const float4*>(colors + gid * (size_t)cstride)[part];
            } else if constexpr (CR == 0) {  // partial block: channel by channel, zeros behind cr (never reads past the row)
                const float* row = colors + gid * (size_t)cstride + 4 * part;
                featpf[k] = make_float4(4 * part + 0 < cr ? row[0] : 0.f, 4 * part + 1 < cr ? row[1] : 0.f,
                                        4 * part + 2 < cr ? row[2] : 0.f, 4 * part + 3 < cr ? row[3] : 0.f);
            } else {  // RGB: three floats per Gaussian (part 0); the other 13 operand channels are zero
                featpf[k] = make_float4(colors[gid * (size_t)cstride + 0], colors[gid * (size_t)cstride + 1], colors[gid * (size_t)cstride + 2], 0.f);
            }
        }
    };

    while (qt - qh < CHK && scanned < NS) consume_scan();
    int nrows = min(CHK, qt - qh);
    if (nrows == 0) return;
    request_rows(nrows);

    // One chunk: stage the requested rows, request the next chunk's rows, process.  Called once before the loop and once
    // inside it (always inlined): hipcc computes the vmcnt wait for the staged rows from the FEWEST memory operations any
    // path issues behind their loads; with the first chunk peeled off, every path into the loop's copy has issued the
    // previous chunk's atomics (a fixed number) behind them, so the wait leaves those in flight.
    int nnext = 0;
    auto do_chunk = [&](auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;   // all 16 rows are real: no padding, unconditional atomics
        // ---- 1. the chunk's rows: registers -> LDS.  Rows beyond nrows (the wave's last chunk only) become padding
        // records: never valid (position 0x7ffffff), zero features, opacity 1; their atomics are masked off (adding their
        // exact zeros to some real row instead costs dearly: same-address atomics serialise at ~22 ns each).
        {
            const int rq = lane >> 2, qq = lane & 3;
            // quarter 0 = {x, y} -> bytes 0..7; 1 = {id, radius} -> {position << 4 | mask, id} at 24 (from the queue); 2 = {a, b} ->
            // {-a/2, -b} at 8; 3 = {c, opacity} -> {-c/2, opacity} at 16
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
#pragma unroll
            for (int k = 0; k < NK; k++) {
                const int e = lane + 64 * k;
                const int g = e / F4, part = e % F4;
                float4 f = featpf[k];
                if (CR == 3 && part != 0) f = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!FULL && g >= nrows) f = make_float4(0.f, 0.f, 0.f, 0.f);
                s_feat4[g * (FROW / 4) + part] = f;
            }
        }
        qh += nrows;
        // ---- 2. keep the queue ahead of the chunks (one scan block per chunk while there is room), then request the
        // next chunk's rows: everything below runs while they travel
        if (scanned < NS && qt - qh <= QCAP - 64) consume_scan();
        while (qt - qh < CHK && scanned < NS) consume_scan();
        nnext = min(CHK, qt - qh);
        if (nnext > 0) request_rows(nnext);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

        // ---- 3. S = F . dL^T  (16 rows x 64 pixels, K = C channels); lane (n16, kq) feeds row n16
        v4f sacc[4];
        {
            float fa[CPL];
#pragma unroll
            for (int q = 0; q < CPL / 4; q++) {
                const float4 f = s_feat4[n16 * (FROW / 4) + (CPL / 4) * kq + q];
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
        // Transpose to lane = pixel through LDS: lane 16g+p holds row 4g+r of pixel 16pb+p in sacc[pb][r]; writes and row
        // reads are conflict-free (row stride 68 floats).  Row m of S is read when step 4 reaches row m, just before w of
        // the same row overwrites it (same address).
#pragma unroll
        for (int pb = 0; pb < 4; pb++)
#pragma unroll
            for (int r = 0; r < 4; r++) my_wa[(4 * kq + r) * WROW + 16 * pb + n16] = sacc[pb][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

        // ---- 4. scalar recurrences (lane = pixel), back to front.  Row parameters arrive by LDS broadcast reads.
        // A row that does not blend into this pixel runs the same arithmetic with alpha = 0: T, R stay put, w = u = 0.
        auto row_step = [&](const int rr) __attribute__((always_inline)) {
            const float4 p0 = *reinterpret_cast<const float4*>(par_bytes + rr * (int)sizeof(BwdPar));
            const float4 p1 = *reinterpret_cast<const float4*>(par_bytes + rr * (int)sizeof(BwdPar) + 16);
            const float dx = p0.x - pixfx, dy = p0.y - pixfy;
            const float power = gauss_power(p0.z, p0.w, p1.x, dx, dy);
            const bool can = (__float_as_int(p1.z) < last4) && power <= 0.0f;
            // opacity * G where the row can blend into this pixel at all, else 0; the 1/255 cut is the last test
            // (min(0.99, t) >= 1/255  <=>  t >= 1/255)
            const float t0 = can ? p1.y * gauss_exp<XEXP>(power) : 0.f;
            const float tG = t0 >= ALPHA_CUT ? t0 : 0.f;   // opacity * G of a contributing row
            const float alpha = alpha_clamp(tG);
            const float om = 1.f - alpha;
            const float inv = rcp_refined(om);
            T = T * inv;
            const float w = alpha * T;  // dchannel_dcolor
            const float dS = my_wa[rr * WROW + lane] - Rcur;
            const float dL_dalpha = fmaf(nTb, inv, dS * T);
            Rcur = fmaf(alpha, dS, Rcur);  // = alpha S + (1 - alpha) Rcur
            const float u = tG * dL_dalpha;  // dL/dG * G  (dL/dG = opacity * dL/dalpha, clamp ignored as in the reference)
            my_wa[rr * WROW + lane] = w;
            my_ua[rr * WROW + lane] = u;
        };
        // Four rows with ONE reciprocal (v_rcp_f32 is a quarter-rate instruction): T behind the group = T / (om0 om1 om2 om3),
        // the T of the rows in between by multiplying back.  Only without a background term (nTb / (1 - alpha) needs each
        // row's own reciprocal): bg = 0 is SAGA's feature training (train_contrastive_feature.py:98).
        auto row_group4 = [&](const int r0) __attribute__((always_inline)) {
            float tG[4], al[4], om[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int rr = r0 + k;
                const float4 p0 = *reinterpret_cast<const float4*>(par_bytes + rr * (int)sizeof(BwdPar));
                const float4 p1 = *reinterpret_cast<const float4*>(par_bytes + rr * (int)sizeof(BwdPar) + 16);
                const float dx = p0.x - pixfx, dy = p0.y - pixfy;
                const float power = gauss_power(p0.z, p0.w, p1.x, dx, dy);
                const float G = gauss_exp<XEXP>(power);
                const float t0 = ((__float_as_int(p1.z) < last4) && power <= 0.0f) ? p1.y * G : 0.f;
                tG[k] = t0 >= ALPHA_CUT ? t0 : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                al[k] = alpha_clamp(tG[k]);
                om[k] = 1.f - al[k];
            }
            float Tk[4];
            Tk[3] = T * rcp_refined((om[0] * om[1]) * (om[2] * om[3]));
            Tk[2] = Tk[3] * om[3];
            Tk[1] = Tk[2] * om[2];
            Tk[0] = Tk[1] * om[1];
            T = Tk[3];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int rr = r0 + k;
                const float dS = my_wa[rr * WROW + lane] - Rcur;
                Rcur = fmaf(al[k], dS, Rcur);
                my_wa[rr * WROW + lane] = al[k] * Tk[k];
                my_ua[rr * WROW + lane] = tG[k] * (dS * Tk[k]);
            }
        };
        if constexpr (FULL) {  // straight-line code for the full chunk: the 16 rows' LDS reads overlap each other's arithmetic
            if (has_bg) {
#pragma unroll
                for (int rr = 0; rr < CHK; rr++) row_step(rr);
            } else {
#pragma unroll
                for (int r0 = 0; r0 < CHK; r0 += 4) row_group4(r0);
            }
        } else {             // the wave's last chunk.  S of a padding row is 0 (zero features), so its w row is done already
#pragma unroll
            for (int rr = 0; rr < CHK; rr++) {
                if (rr < nrows) row_step(rr);
                else my_ua[rr * WROW + lane] = 0.f;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

        // ---- 5. dF = W^T . dL  and  M = U^T . Phi   (A rows from LDS, lane (m = n16, kq) reads pixels 16kq..16kq+15)
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
                const float wa[4] = {wv.x, wv.y, wv.z, wv.w};
                const float4 uv = urow[s4];
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
        // ---- 6. outputs.  Result layout: lane l holds column n16 of rows 4*kq + r.  Every atomic below is issued
        // unconditionally (see the header): rows that contributed nothing add exact zeros.
        const BwdPar mine = *reinterpret_cast<const BwdPar*>(par_bytes + n16 * (int)sizeof(BwdPar));  // lane n16 = row n16
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the MFMA operand reads of the w rows are done: moments may land there
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = 4 * kq + r;
            // the row's Gaussian id, straight from the staged record (a __shfl would keep its lane arithmetic alive
            // across the whole kernel -- and spilled)
            const uint32_t gid = __float_as_uint(*reinterpret_cast<const float*>(par_bytes + row * (int)sizeof(BwdPar) + 28));
#pragma unroll
            for (int nb = 0; nb < NB; nb++) {
                const int ch = 16 * nb + n16;
                if (!FULL && row >= nrows) continue;
                if constexpr (CR == C) {
                    atomicAdd(&dL_dcolors[(size_t)gid * cstride + ch], facc[nb][r]);
                } else {
                    if (ch < cr) atomicAdd(&dL_dcolors[(size_t)gid * cstride + ch], facc[nb][r]);
                    else if (MASKGRAD && ch == CR) atomicAdd(&gpack[(size_t)gid * 8 + 6], facc[nb][r]);
                }
            }
        }
        // (a separate, unconditional loop: with the store inside the loop above and guarded by n16 < 8, hipcc clones the
        // atomics into both arms of the guard -- twice the memory instructions, and a count that depends on the path)
#pragma unroll
        for (int r = 0; r < 4; r++) my_mom[(4 * kq + r) * MROW + n16] = macc[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        // moments -> fields: lane = row (16 lanes) rewrites its my_mom row in place, then the wave adds the rows
        // to the packed per-Gaussian records: lane -> (row = l / 8 (+8), field = l % 8), 32 contiguous bytes per row
        {
            if (lane < 16) {
                const int row = lane;
                const float4 m0 = reinterpret_cast<const float4*>(my_mom + row * MROW)[0];
                const float4 m1 = reinterpret_cast<const float4*>(my_mom + row * MROW)[1];
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
                reinterpret_cast<float4*>(my_mom + row * MROW)[0] = o0;
                reinterpret_cast<float4*>(my_mom + row * MROW)[1] = o1;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const int row2 = 8 * it + (lane >> 3), f = lane & 7;
                const uint32_t gid2 = __float_as_uint(*reinterpret_cast<const float*>(par_bytes + row2 * (int)sizeof(BwdPar) + 28));
                if (FULL || row2 < nrows) atomicAdd(&gpack[(size_t)gid2 * 8 + f], my_mom[row2 * MROW + f]);   // fields 6, 7 receive +0
            }
        }
    };
    // Full chunks: the first one peeled off, the rest in a loop whose every iteration issues the same memory instructions.
    // The wave's last, partial chunk runs a third copy with padding rows and predicated atomics (nothing is staged behind it).
    if (nrows == CHK) {
        do_chunk(std::true_type{});
        while (nnext == CHK) {
            nrows = nnext;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // moment reads before the next chunk's S rows land there
            do_chunk(std::true_type{});
        }
        nrows = nnext;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    if (nrows != 0) do_chunk(std::false_type{});
    };

    {
        // workgroup -> (tile, quadrant).  Every XCD works through a contiguous run of tiles (common.h), and the four
        // quadrants of a tile go to four of its waves at about the same time: the records, the feature rows and the gradient
        // lines that the quadrants of a tile AND neighbouring tiles share stay in one L2.  The first half of a run
        // is assigned by the workgroup id (id = 8 (4 j + quad) + x: XCD x, j-th tile of its run); the items of the second
        // half are TAKEN from the XCD's queue, and from the other XCDs' queues when that one is empty (xcd_grab): the
        // runs stay contiguous while the XCDs keep pace, and nobody idles when the scene's density does not let them.
        // The runs are work-balanced (common.h): nine boundaries in the image buffer; the grid is sized for the longest run the
        // clamp allows, ids beyond a run's static part and takers beyond the queued items leave at once.
        const uint32_t b = blockIdx.x, nstatic = 32u * xcd_static_len_max(ntiles);
        uint32_t item;
        if (b < nstatic) {
            const uint32_t x = b & 7u, jj = b >> 3;
            const uint32_t start = run_bounds[x], len = run_bounds[x + 1u] - start;
            item = (jj >> 2) < len - len / XCD_QUEUE_DIV ? 4u * start + jj : 0xFFFFFFFFu;
        } else {
            item = xcd_grab_runs(queue_ctr, xcd_load_runs(run_bounds), 4u);
        }
        if (item != 0xFFFFFFFFu) {
            MI_XCD_STAMP(false);   // (profiling build: per-XCD start / end stamps, common.h)
            quadrant(item >> 2, item & 3u);
            MI_XCD_STAMP(true);
        }
    }
}

}  // namespace mirast
