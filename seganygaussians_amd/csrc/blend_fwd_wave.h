// blend_fwd_wave.h -- forward blend for C = 32 / 64, one WAVE per 8x8 quadrant, no workgroup barriers.
//
// Same per-pixel arithmetic as blend_fwd.h (renderCUDA<C> forward, CF/cuda_rasterizer/forward.cu:264-385: alpha, T, the
// 1/255 and 1e-4 tests, n_contrib and final_T on f32 VALU code, bit-identical) and the same accumulation (every f32 operand
// split exactly into three bf16 terms, six partial products on the bf16 matrix pipe, f32 accumulate).  What changes is who
// walks the tile's blend list and when -- the restructuring blend_bwd_wave.h applied to the backward:
//   * blend_fwd_x3.h stages the list in batches of 128 records for the whole tile: three workgroup barriers per batch, the
//     four quadrant waves of a tile in lockstep (a batch lasts as long as its busiest quadrant), every quadrant's list padded
//     to a multiple of 16 entries PER BATCH.  The kernel is VALU-bound (alpha evaluation: ~37 VALU per (wave, entry)) at four
//     waves per SIMD, so every cycle a wave waits at a barrier is a cycle the SIMD has three waves to issue from;
//   * here a workgroup IS one wave: it scans the tile's blend list itself (64 four-byte entries id | quadrant mask << 28 per
//     block, one block prefetched), queues the entries of its quadrant in an LDS ring, and works through them in FULL groups of 16
//     (the 32-byte geometry record index_rec[id] and the feature row of an entry are gathered side by side)
//     -- only a wave's last group is padded.  The 16 records and feature rows of the NEXT group are requested before the
//     current group is evaluated and wait in registers (2 + 2 C / 32 ... VGPRs); a wave leaves as soon as its 64 pixels are
//     done.  The four quadrants of a tile are four workgroups on the same XCD (blockIdx -> (tile, quadrant) as in the backward).
//   * the group's records sit at fixed LDS offsets (entry i at s_rec[i]): no per-quadrant index lists.
#pragma once

#include <type_traits>

#include "blend_fwd.h"
#include "blend_fwd_split.h"

namespace mirast {

// workgroup -> (tile, quadrant): every XCD works through contiguous runs of tiles, the four quadrants of a tile on four of its
// waves at about the same time (common.h; id = 8 (4 j + quad) + x: XCD x, j-th tile of its runs).  The row-major tile sequence is
// cut into 8 m runs of equal tile counts, run i to XCD i % 8: m = 1 is the one contiguous eighth of the image per XCD of rounds
// 3-4; with m > 1 every XCD works on m bands spread over the image -- what a tile costs varies over the image of a real scene, and
// nothing balances the XCDs' TIME but the statistics of what each is handed (a fast XCD cannot take more: the dispatcher deals
// workgroup b to XCD b % 8 whatever their progress; profiles/r04_xcd_balance.md) -- at the price of more run boundaries, where
// neighbouring tiles fetch the Gaussians they share into two L2s.
__host__ __device__ inline uint32_t fwd_runs_longest(uint32_t ntiles, uint32_t m)   // tiles of the XCD that is handed the most
{
    uint32_t longest = 0;
    for (uint32_t x = 0; x < 8u; x++) {
        uint32_t t = 0;
        for (uint32_t k = 0; k < m; k++) {
            const uint32_t i = x + 8u * k;
            t += (uint32_t)(((uint64_t)(i + 1u) * ntiles) / (8u * m)) - (uint32_t)(((uint64_t)i * ntiles) / (8u * m));
        }
        longest = t > longest ? t : longest;
    }
    return longest;
}
// m == 0: ONE run per XCD with the boundaries the range scan left in the image buffer (binning.h: tile_ranges_kernel, equal MODELLED
// work); the grid is sized for the longest run.
__device__ __forceinline__ bool fwd_wave_item(uint32_t b, uint32_t ntiles, uint32_t m, const uint32_t* __restrict__ run_bounds,
                                              uint32_t& tile, uint32_t& quad)
{
    const uint32_t x = b & 7u, jj = b >> 3;
    uint32_t tl = jj >> 2;   // index into the concatenation of XCD x's m runs
    quad = jj & 3u;
    if (m == 0u) {
        const uint32_t start = run_bounds[x], len = run_bounds[x + 1u] - start;
        tile = start + tl;
        return tl < len;
    }
    for (uint32_t k = 0; k < m; k++) {
        const uint32_t i = x + 8u * k;
        const uint32_t start = (uint32_t)(((uint64_t)i * ntiles) / (8u * m));
        const uint32_t len = (uint32_t)(((uint64_t)(i + 1u) * ntiles) / (8u * m)) - start;
        if (tl < len) {
            tile = start + tl;
            return true;
        }
        tl -= len;
    }
    return false;
}

// Zero-fill riding on the forward blend (include/mi_rast.h: dL_dcolor_next, MI_RAST_PREZERO_BWD).  The backward accumulates
// dL_dcolor and its packed per-Gaussian fields with atomics, so both must be zero when it starts: 160 MB of fill per cfg3 view
// (two fill kernels, 25 us of a 1.23-ms step) that this kernel -- bound by VALU issue, a fifth of the HBM rate in use -- stores on
// the side: workgroup b of nb writes the 1-KB pieces b, b + nb, ... before it starts on its quadrant (four or five
// dwordx4 stores per wave on cfg3).
struct FwdZeroFill {
    float4* a;      // region 1 (dL_dcolor_next) ...
    uint32_t na;    // ... in 16-byte units
    float4* b;      // region 2 (the geometry buffer's bwd_pack + work-queue counters)
    uint32_t nb;
};
__device__ __forceinline__ void fwd_zero_fill(const FwdZeroFill& z, uint32_t wg, uint32_t nwg, int lane)
{
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (uint32_t i = wg * 64u + (uint32_t)lane; i < z.na; i += nwg * 64u) z.a[i] = zero;
    for (uint32_t i = wg * 64u + (uint32_t)lane; i < z.nb; i += nwg * 64u) z.b[i] = zero;
}

// Waves per SIMD the instances are compiled for (register budget 512 / waves): 32 channels 4 (five: spills, no gain), 64 channels 3.
constexpr int FWD_WAVES32 = 4, FWD_WAVES64 = 3;
// Measured and settled in round 4: the rows of a group are requested TWO groups ahead at 32 channels (two
// register sets; at 64 channels the second set does not fit); the wave-uniform "every pixel is done" test sits in front of every
// pair of entries; a scheduling barrier behind every pair keeps the record reads of later pairs from piling up in registers.

// XM: how opacity * exp(power) is evaluated (common.h: ExpMode) -- EXP_HYBRID is the product default.
// PARTIAL: only the first `cr_arg` (1 .. 31) of the block's 32 channels exist in memory -- the last channel block of a feature whose
// width is no multiple of 32 (the reference compiles ANY NUM_CHANNELS, config_contrastive_f.h:15): feature rows are loaded channel by
// channel with zeros behind cr (zeros in the MFMA operands, as RGB has always been blended), and only cr planes are written.
template <int C, int XM = EXP_HYBRID, bool STRIDED = false, bool PARTIAL = false>
__global__ void __launch_bounds__(64, C == 32 ? FWD_WAVES32 : FWD_WAVES64) blend_fwd_wave_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ blend_list, const BlendRec* __restrict__ index_rec,
    int W, int H, uint32_t horizontal_blocks, uint32_t ntiles, const float* __restrict__ features, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_consumed /* zeroed: receives atomicMax */,
    uint32_t* __restrict__ tile_nsurv /* zeroed: receives atomicMax */, const float* __restrict__ bg_color,
    float* __restrict__ out_color, int cstride_arg /* STRIDED: floats between feature rows (blend_fwd.h) */, FwdZeroFill zfill,
    uint32_t runs_per_xcd /* fwd_wave_item's m */, const uint32_t* __restrict__ run_bounds /* [9], m == 0 */,
    int cr_arg /* PARTIAL: channels of this block that exist (1 .. 31) */)
{
    static_assert(C == 32 || C == 64, "32-channel accumulator blocks");
    static_assert(!PARTIAL || (C == 32 && STRIDED), "a partial block is the 32-channel remainder of a wider (or narrower) feature");
    const int cr = PARTIAL ? cr_arg : C;
    constexpr int F4 = C / 4, NCB = C / 32, XROW = 6 * C, PLANE = 2 * C;  // float4s per row; channel blocks; row / plane bytes
    constexpr int QCAP = 128;
    constexpr int NK = XG * F4 / 64;   // float4 feature parts per lane and group
    const int cstride = STRIDED ? cstride_arg : C;
    __shared__ XRec s_rec[XG];
    __shared__ uint4 s_feat4[XG * XROW / 16];
    __shared__ uint2 s_queue[QCAP];     // {position in the blend list, entry = Gaussian id | quadrant mask << 28}
    __shared__ uint32_t s_j[XG];        // blend-list index of the group's entries
    static_assert(XG * XROW >= 8 * 65 * 4, "feature buffer too small for the epilogue transpose");
    char* const featb = reinterpret_cast<char*>(s_feat4);
    const char* const rec_bytes = reinterpret_cast<const char*>(s_rec);

    fwd_zero_fill(zfill, blockIdx.x, gridDim.x, (int)(threadIdx.x & 63));   // (every workgroup, also those without an item)
    uint32_t tile, quad;
    if (!fwd_wave_item(blockIdx.x, ntiles, runs_per_xcd, run_bounds, tile, quad)) return;
    MI_XCD_STAMP(false);   // (profiling build: per-XCD start / end stamps, common.h)
    const int lane = threadIdx.x & 63;
    const uint32_t tile_x = tile % horizontal_blocks, tile_y = tile / horizontal_blocks;
    const uint32_t px = tile_x * TILE_X + (quad & 1) * 8 + (lane & 7);
    const uint32_t py = tile_y * TILE_Y + (quad >> 1) * 8 + (lane >> 3);
    const uint32_t pix_id = W * py + px;
    const float pixfx = (float)px, pixfy = (float)py;
    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    bool done = !inside;

    const uint2 range = ranges[tile];
    const int list_len = (int)(range.y - range.x);
    const bool any_inside = ballot64(inside) != 0;
    const int ns = any_inside ? list_len : 0;
    const uint32_t* lst = blend_list + range.x;   // entry = Gaussian id | quadrant mask << 28 (binning.h); its position is its index

    float T = 1.0f;
    uint32_t last_contributor = 0;
    int consumed = any_inside ? list_len : 0;   // raw list entries this wave consumed (counter E of SURVEY.md 8d)
    int walked = ns;                            // blend-list records it walked (the backward starts from there)
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

    // ---- the queue of this quadrant's records
    int scanned = 0, qh = 0, qt = 0;
    uint32_t scan_reg = 0u;
    if (ns > 0) scan_reg = lst[min(lane, ns - 1)];
    auto consume_scan = [&]() {
        const int j = scanned + lane;
        const bool cand = j < ns && ((scan_reg >> (ID_BITS + quad)) & 1u) != 0;
        const uint64_t bal = ballot64(cand);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (cand) s_queue[(qt + (int)below) & (QCAP - 1)] = make_uint2((uint32_t)j, scan_reg);
        qt += __builtin_popcountll(bal);
        scanned += 64;
        scan_reg = lst[min(scanned + lane, ns - 1)];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    };
    // Requests the rows queue[qh + ahead .. qh + ahead + n), n >= 1, of a coming group into one of the two register sets: record
    // quarter (lane & 3) of row (lane >> 2) and the feature parts.  Every lane loads (rows >= n repeat row n - 1 and are replaced
    // by padding when the group is staged).
    struct RowRegs {
        uint2 q;
        float4 f[NK];
    };
    RowRegs setA, setB;
    auto request_rows = [&](RowRegs& rr, int ahead, int n) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            const int rq = min(lane >> 2, n - 1), qq = lane & 3;
            const uint32_t gq = s_queue[(qh + ahead + rq) & (QCAP - 1)].y & ID_MASK;
            rr.q = reinterpret_cast<const uint2*>(index_rec + gq)[qq];
        }
#pragma unroll
        for (int k = 0; k < NK; k++) {
            const int e = lane + 64 * k;
            const int g = min(e / F4, n - 1), part = e % F4;
            const size_t gid = (size_t)(s_queue[(qh + ahead + g) & (QCAP - 1)].y & ID_MASK);
            if constexpr (PARTIAL) {   // channel by channel, zeros behind cr (never reads past the row)
                const float* row = features + gid * (size_t)cstride + 4 * part;
                rr.f[k] = make_float4(4 * part + 0 < cr ? row[0] : 0.f, 4 * part + 1 < cr ? row[1] : 0.f,
                                      4 * part + 2 < cr ? row[2] : 0.f, 4 * part + 3 < cr ? row[3] : 0.f);
            } else {
                rr.f[k] = reinterpret_cast<const float4*>(features + gid * (size_t)cstride)[part];
            }
        }
    };

    uint64_t live = 0;  // lanes still blending (wave-uniform copy of !done)
    bool finished = false;

    // One group of up to 16 entries whose rows wait in `rr`; `n_ahead` rows of the group behind it are in flight in the other
    // register set (32 channels) or none (n_ahead = 0).  Requests the group behind those into `rr` and returns its row count.
    auto do_group = [&](RowRegs& rr, const int n, const int n_ahead) __attribute__((always_inline)) -> int {
        // ---- 1. the group's rows: registers -> LDS.  Rows beyond n (the wave's last group only) become padding: opacity 0
        // (never blends) and zero features.
        {
            const uint2 curq = rr.q;
            const int rq = lane >> 2, qq = lane & 3;
            // quarter 0 = {x, y}; 1 = {id, radius} -> {position + 1, position << 4 | mask} (from the queue); 2 = {a, b} -> {-a/2, -b};
            // 3 = {c, opacity} -> {-c/2, opacity}
            float2 v = make_float2(__uint_as_float(curq.x), __uint_as_float(curq.y));
            const uint2 qe = s_queue[(qh + min(rq, n - 1)) & (QCAP - 1)];   // {position, id | mask << 28}
            if (qq == 1) v = make_float2(__uint_as_float(qe.x + 1u), __uint_as_float((qe.x << 4) | (qe.y >> ID_BITS)));
            if (qq == 2) v = make_float2(-0.5f * v.x, -v.y);
            if (qq == 3) v = make_float2(-0.5f * v.x, v.y);
            if (rq >= n) {
                v = make_float2(0.f, 0.f);
                if (qq == 2) v = make_float2(-0.5f, 0.f);
                if (qq == 3) v = make_float2(-0.5f, 0.f);
            }
            const int dst = qq == 0 ? 0 : (qq == 1 ? 24 : (qq == 2 ? 8 : 16));
            *reinterpret_cast<float2*>(reinterpret_cast<char*>(&s_rec[rq]) + dst) = v;
            if (qq == 1) s_j[rq] = qe.x;
#pragma unroll
            for (int k = 0; k < NK; k++) {
                const int e = lane + 64 * k;
                const int g = e / F4, part = e % F4;
                uint2 hi = make_uint2(0u, 0u), mid = hi, lo = hi;
                if (g < n) {
                    split3_bf16x2(rr.f[k].x, rr.f[k].y, hi.x, mid.x, lo.x);
                    split3_bf16x2(rr.f[k].z, rr.f[k].w, hi.y, mid.y, lo.y);
                }
                uint2* row = reinterpret_cast<uint2*>(featb + g * XROW);
                row[part] = hi;
                row[F4 + part] = mid;
                row[2 * F4 + part] = lo;
            }
        }
        qh += n;
        // ---- 2. keep the queue ahead of the groups, then request the rows of the group behind the ones in flight: everything
        // below runs while they travel
        if (scanned < ns && qt - qh <= QCAP - 64) consume_scan();
        while (qt - qh - n_ahead < XG && scanned < ns) consume_scan();
        const int nreq = min(XG, qt - qh - n_ahead);
        if (nreq > 0) request_rows(rr, n_ahead, nreq);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

        // ---- 3. alpha, T and w of the 16 entries (same f32 arithmetic as blend_fwd.h), w split and packed in pairs.
        // EXP_HYBRID: the whole group is evaluated with v_exp_f32 first (no branch inside: hipcc interleaves the pairs); `band`
        // collects the lanes of any entry whose opacity * G falls between the two bounds around the 1/255 cut, and in that (rare)
        // case the group is evaluated again from the saved pixel state with expf for every entry -- a wave-uniform branch per
        // GROUP (one per pair of entries was measured: the branches and their scalar mask arithmetic ate what the exp saved).
        uint32_t wp[3][XG / 2];
        // (Round 4 measured the bookkeeping on the vector side instead -- per lane "the entry at which I stopped" as one v_cndmask in
        // place of the seven scalar instructions of fin_j below, a running minimum of |t - 1/255| in place of the two-sided band
        // ballots: 112 -> 91 instructions per pair of entries, 43 -> 19 of them scalar -- and the kernel ran exactly as long at 32
        // channels and 7 % longer at 64: it is bound neither by the VALU pipe nor by what a wave issues)
        int fin_j = -1;
        auto eval_group = [&](auto fast_tag) __attribute__((always_inline)) -> uint64_t {
            constexpr bool FAST = decltype(fast_tag)::value;   // v_exp_f32 + two bounds; otherwise expf (or XM's own form) + the cut
            uint64_t band = 0;
#pragma unroll
            for (int i = 0; i < XG / 2; i++) {
                // padding pair of the wave's last group, or every pixel is done (wave-uniform): w = 0
                if (2 * i >= n || live == 0) {
                    wp[0][i] = wp[1][i] = wp[2][i] = 0u;
                    continue;
                }
                float w2[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const float4 p0 = *reinterpret_cast<const float4*>(rec_bytes + (2 * i + h) * (int)sizeof(XRec));
                    const float4 p1 = *reinterpret_cast<const float4*>(rec_bytes + (2 * i + h) * (int)sizeof(XRec) + 16);
                    const float dx = p0.x - pixfx, dy = p0.y - pixfy;
                    const float power = gauss_power(p0.z, p0.w, p1.x, dx, dy);
                    float t;
                    bool cut;
                    if constexpr (FAST) {
                        t = p1.y * gauss_exp_fast(power);
                        cut = t >= ALPHA_CUT_HI;
                        band |= ballot64(t >= ALPHA_CUT_LO) ^ ballot64(cut);
                    } else {
                        t = p1.y * gauss_exp<XM != EXP_FAST>(power);
                        cut = t >= ALPHA_CUT;
                    }
                    const float alpha = fminf(0.99f, t);
                    const bool ok = !done && power <= 0.0f && cut;
                    const float test_T = T * (1 - alpha);
                    // forward.cu:358-362: done once a contributor would push T below 1e-4 (that one is not blended).
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
            return band;
        };
        if constexpr (XM == EXP_HYBRID) {
            const float T0 = T;
            const bool done0 = done;
            const uint32_t last0 = last_contributor;
            const uint64_t live0 = live;
            if (eval_group(std::true_type{}) != 0) {   // some lane sits on the cut: expf decides (and supplies the values), for the whole group
                T = T0;
                done = done0;
                last_contributor = last0;
                live = live0;
                fin_j = -1;
                eval_group(std::false_type{});
            }
        } else {
            eval_group(std::false_type{});
        }
        if (fin_j >= 0) {
            consumed = (int)__builtin_amdgcn_readfirstlane(__float_as_uint(s_rec[fin_j].q1.z));
            walked = (int)__builtin_amdgcn_readfirstlane(s_j[fin_j]) + 1;
            finished = true;
        }
        // ---- 4. A operands (blend_fwd_x3.h): lane l and l ^ 32 exchange half of their packed terms
        v4u A0[3], A1[3];
        asm volatile("s_nop 3");
#pragma unroll
        for (int p = 0; p < 3; p++) {
#pragma unroll
            for (int d = 0; d < 4; d++) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(wp[p][d]), "+v"(wp[p][4 + d]));
            A0[p] = (v4u){wp[p][0], wp[p][1], wp[p][2], wp[p][3]};
            A1[p] = (v4u){wp[p][4], wp[p][5], wp[p][6], wp[p][7]};
        }
        // ---- 5. B operands: lane (channel, k half) gathers the three terms of its 8 entries' feature value (rows 8 khalf .. + 7)
        const char* const fb = featb + (lower ? 0 : 8 * XROW) + chan2;
#define X3_MFMA(ACC, AP, BP) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(AP), "v"(BP))
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
            v4u B[3];
            {
                uint32_t bb[3][4];
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const char* r0 = fb + (2 * d) * XROW + 64 * cb;
                    const char* r1 = fb + (2 * d + 1) * XROW + 64 * cb;
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
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // operand reads before the next group's rows land in the same LDS
        return nreq;
    };

    while (qt - qh < XG && scanned < ns) consume_scan();
    int nA = min(XG, qt - qh);
    if (nA > 0) request_rows(setA, 0, nA);
    live = ballot64(!done);
    // 32 channels: rows requested TWO groups ahead -- a group of 16 entries takes a wave about as long as a dependent gather
    // under load (~4 us), so one group of run-ahead leaves the wave waiting whenever the memory system is slower than its own
    // arithmetic (cfg3: 0.275 -> 0.268 ms).  At 64 channels the second register set (18 VGPRs) does not fit beside the accumulators
    // at three waves per SIMD (spills): one group ahead there.
    constexpr bool PF2 = C == 32;
    if constexpr (PF2) {
        int nB = 0;
        if (nA == XG) {
            while (qt - qh - nA < XG && scanned < ns) consume_scan();
            nB = min(XG, qt - qh - nA);
            if (nB > 0) request_rows(setB, nA, nB);
        }
        while (nA > 0 && !finished) {
            const int nA2 = do_group(setA, nA, nB);
            if (nB <= 0 || finished) break;
            const int nB2 = do_group(setB, nB, nA2);
            nA = nA2;
            nB = nB2;
        }
    } else {
        while (nA > 0 && !finished) nA = do_group(setA, nA, 0);
    }

    if (lane == 0 && any_inside) {
        atomicMax(&tile_consumed[tile], (uint32_t)consumed);
        atomicMax(&tile_nsurv[tile], (uint32_t)walked);
    }
    const size_t HW = (size_t)H * W;
    if (inside) {
        final_T[pix_id] = T;
        n_contrib[pix_id] = last_contributor;
    }
    // accumulator tiles: lane l holds channel (l & 31); register r of block b <-> pixel 32 b + (r & 3) + 8 (r >> 2) + 4 (l >> 5)
    // (tools/mfma_probe.hip).  Transposed through LDS, 8 channels at a time, then stored pixel-major.
    float* tp = reinterpret_cast<float*>(s_feat4);
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
                    if (!PARTIAL || ch < cr) out_color[ch * HW + pix_id] = tp[c * 65 + lane] + T * bg_color[ch];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }
    }
    MI_XCD_STAMP(true);
}

}  // namespace mirast

namespace mirast {

// The same walk for RGB (C = 3; EXTRA = 2: the DEPTH variant's mask and depth planes, DEPTH/cuda_rasterizer/forward.cu:308-309,
// 363-365, 384-385, no background term on them): accumulators on the VALU (acc[ch] = fmaf(f[ch], w, acc[ch]) in list order --
// the arithmetic of blend_fwd.h, bit for bit), the 16 rows of a group staged as {r, g, b, mask, depth} floats.
template <int EXTRA, int XM = EXP_HYBRID>
__global__ void __launch_bounds__(64, 8) blend_fwd_wave_rgb_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ blend_list, const BlendRec* __restrict__ index_rec,
    int W, int H, uint32_t horizontal_blocks, uint32_t ntiles, const float* __restrict__ features /* [P,3] */,
    const float* __restrict__ mask, const float* __restrict__ depths, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_consumed, uint32_t* __restrict__ tile_nsurv,
    const float* __restrict__ bg_color, float* __restrict__ out_color, float* __restrict__ out_mask, float* __restrict__ out_depth,
    FwdZeroFill zfill, uint32_t runs_per_xcd, const uint32_t* __restrict__ run_bounds)
{
    constexpr int C = 3, CE = C + EXTRA, QCAP = 128, FROW = 8;
    __shared__ XRec s_rec[XG];
    __shared__ float s_f[XG * FROW];
    __shared__ uint2 s_queue[QCAP];
    __shared__ uint32_t s_j[XG];
    const char* const rec_bytes = reinterpret_cast<const char*>(s_rec);

    fwd_zero_fill(zfill, blockIdx.x, gridDim.x, (int)(threadIdx.x & 63));   // (every workgroup, also those without an item)
    uint32_t tile, quad;
    if (!fwd_wave_item(blockIdx.x, ntiles, runs_per_xcd, run_bounds, tile, quad)) return;
    const int lane = threadIdx.x & 63;
    const uint32_t tile_x = tile % horizontal_blocks, tile_y = tile / horizontal_blocks;
    const uint32_t px = tile_x * TILE_X + (quad & 1) * 8 + (lane & 7);
    const uint32_t py = tile_y * TILE_Y + (quad >> 1) * 8 + (lane >> 3);
    const uint32_t pix_id = W * py + px;
    const float pixfx = (float)px, pixfy = (float)py;
    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    bool done = !inside;

    const uint2 range = ranges[tile];
    const int list_len = (int)(range.y - range.x);
    const bool any_inside = ballot64(inside) != 0;
    const int ns = any_inside ? list_len : 0;
    const uint32_t* lst = blend_list + range.x;   // entry = Gaussian id | quadrant mask << 28 (binning.h); its position is its index

    float T = 1.0f;
    uint32_t last_contributor = 0;
    int consumed = any_inside ? list_len : 0;
    int walked = ns;
    float acc[CE];
#pragma unroll
    for (int ch = 0; ch < CE; ch++) acc[ch] = 0.f;

    int scanned = 0, qh = 0, qt = 0;
    uint32_t scan_reg = 0u;
    if (ns > 0) scan_reg = lst[min(lane, ns - 1)];
    auto consume_scan = [&]() {
        const int j = scanned + lane;
        const bool cand = j < ns && ((scan_reg >> (ID_BITS + quad)) & 1u) != 0;
        const uint64_t bal = ballot64(cand);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (cand) s_queue[(qt + (int)below) & (QCAP - 1)] = make_uint2((uint32_t)j, scan_reg);
        qt += __builtin_popcountll(bal);
        scanned += 64;
        scan_reg = lst[min(scanned + lane, ns - 1)];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    };
    // next group's rows: record quarter (lane & 3) of row (lane >> 2); lane l feeds value (l & 7) of row (l >> 3) and (l >> 3) + 8
    uint2 curq;
    float fpf[2];
    auto request_rows = [&](int n) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            const int rq = min(lane >> 2, n - 1), qq = lane & 3;
            const uint32_t gq = s_queue[(qh + rq) & (QCAP - 1)].y & ID_MASK;
            curq = reinterpret_cast<const uint2*>(index_rec + gq)[qq];
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int g = min((lane >> 3) + 8 * k, n - 1), c = lane & 7;
            const size_t gid = (size_t)(s_queue[(qh + g) & (QCAP - 1)].y & ID_MASK);
            float v = 0.f;
            if (c < C) v = features[gid * 3 + c];
            else if (EXTRA >= 1 && c == C) v = mask[gid];
            else if (EXTRA >= 2 && c == C + 1) v = depths[gid];
            fpf[k] = v;
        }
    };

    while (qt - qh < XG && scanned < ns) consume_scan();
    int n = min(XG, qt - qh);
    if (n > 0) request_rows(n);
    uint64_t live = ballot64(!done);
    bool finished = false;

    while (n > 0 && !finished) {
        {
            const int rq = lane >> 2, qq = lane & 3;
            float2 v = make_float2(__uint_as_float(curq.x), __uint_as_float(curq.y));
            const uint2 qe = s_queue[(qh + min(rq, n - 1)) & (QCAP - 1)];   // {position, id | mask << 28}
            if (qq == 1) v = make_float2(__uint_as_float(qe.x + 1u), __uint_as_float((qe.x << 4) | (qe.y >> ID_BITS)));
            if (qq == 2) v = make_float2(-0.5f * v.x, -v.y);
            if (qq == 3) v = make_float2(-0.5f * v.x, v.y);
            if (rq >= n) {
                v = make_float2(0.f, 0.f);
                if (qq == 2) v = make_float2(-0.5f, 0.f);
                if (qq == 3) v = make_float2(-0.5f, 0.f);
            }
            const int dst = qq == 0 ? 0 : (qq == 1 ? 24 : (qq == 2 ? 8 : 16));
            *reinterpret_cast<float2*>(reinterpret_cast<char*>(&s_rec[rq]) + dst) = v;
            if (qq == 1) s_j[rq] = qe.x;
#pragma unroll
            for (int k = 0; k < 2; k++) s_f[((lane >> 3) + 8 * k) * FROW + (lane & 7)] = fpf[k];
        }
        qh += n;
        if (scanned < ns && qt - qh <= QCAP - 64) consume_scan();
        while (qt - qh < XG && scanned < ns) consume_scan();
        const int nnext = min(XG, qt - qh);
        if (nnext > 0) request_rows(nnext);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

        int fin_j = -1;
#pragma unroll
        for (int i = 0; i < XG; i++) {
            if (i >= n || live == 0) continue;   // padding of the wave's last group, or every pixel is done (wave-uniform)
            const float4 p0 = *reinterpret_cast<const float4*>(rec_bytes + i * (int)sizeof(XRec));
            const float4 p1 = *reinterpret_cast<const float4*>(rec_bytes + i * (int)sizeof(XRec) + 16);
            const float dx = p0.x - pixfx, dy = p0.y - pixfy;
            const float power = gauss_power(p0.z, p0.w, p1.x, dx, dy);
            float t;
            bool cut;
            if constexpr (XM == EXP_HYBRID) {   // common.h: v_exp_f32 away from the 1/255 cut, expf on it
                t = p1.y * gauss_exp_fast(power);
                cut = t >= ALPHA_CUT_HI;
                if ((ballot64(t >= ALPHA_CUT_LO) ^ ballot64(cut)) != 0) {
                    t = p1.y * gauss_exp<true>(power);
                    cut = t >= ALPHA_CUT;
                }
            } else {
                t = p1.y * gauss_exp<XM == EXP_EXACT>(power);
                cut = t >= ALPHA_CUT;
            }
            const float alpha = fminf(0.99f, t);
            const bool ok = !done && power <= 0.0f && cut;
            const float test_T = T * (1 - alpha);
            const float tt = ok ? test_T : 1.0f;
            live &= ~__builtin_amdgcn_fcmpf(tt, 0.0001f, 4 /* FCMP_OLT */);
            const bool stop = tt < 0.0001f;
            done = done || stop;
            const bool blend = ok && !stop;
            const float w = blend ? alpha * T : 0.f;
            if (ballot64(blend) != 0) {
                const float4 f0 = *reinterpret_cast<const float4*>(&s_f[i * FROW]);
                acc[0] = fmaf(f0.x, w, acc[0]);
                acc[1] = fmaf(f0.y, w, acc[1]);
                acc[2] = fmaf(f0.z, w, acc[2]);
                if constexpr (EXTRA >= 1) acc[3] = fmaf(f0.w, w, acc[3]);
                if constexpr (EXTRA >= 2) acc[4] = fmaf(s_f[i * FROW + 4], w, acc[4]);
            }
            T = blend ? test_T : T;
            last_contributor = blend ? __float_as_uint(p1.z) : last_contributor;
            fin_j = (live == 0 && fin_j < 0) ? i : fin_j;
        }
        if (fin_j >= 0) {
            consumed = (int)__builtin_amdgcn_readfirstlane(__float_as_uint(s_rec[fin_j].q1.z));
            walked = (int)__builtin_amdgcn_readfirstlane(s_j[fin_j]) + 1;
            finished = true;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        n = nnext;
    }

    if (lane == 0 && any_inside) {
        atomicMax(&tile_consumed[tile], (uint32_t)consumed);
        atomicMax(&tile_nsurv[tile], (uint32_t)walked);
    }
    const size_t HW = (size_t)H * W;
    if (inside) {
        final_T[pix_id] = T;
        n_contrib[pix_id] = last_contributor;
#pragma unroll
        for (int ch = 0; ch < C; ch++) out_color[ch * HW + pix_id] = acc[ch] + T * bg_color[ch];
        if constexpr (EXTRA >= 1) out_mask[pix_id] = acc[C];
        if constexpr (EXTRA >= 2) out_depth[pix_id] = acc[C + 1];
    }
}

}  // namespace mirast
