// depth_sort.h -- depth ranks of the Gaussians: sorted_idx[rank] and the per-rank geometry records.
//
// What the reference needs from its 64-bit (tile | depth) sort is, per tile, the order by (depth bits, Gaussian
// index) (rasterizer_impl.cu:70-113, 296-308).  binning.h gets that from ONE ordering of the P Gaussians by
// (depth bits, index): the dense rank.  A library radix sort of P key/value pairs is ~20 small launches (0.16 ms
// at P = 1 M, launch-bound); here the ordering is produced by the same count / scan / scatter machinery as the tile
// binning, with the float's own bit layout as the bucket function:
//   1. bucket = (depth_bits - min_bits) >> s, with s the smallest shift that maps the view's key range (min / max
//      from the preprocess pass) onto <= 16384 buckets: log-spaced in depth, monotonic in the key, between 8192 and
//      16384 of them in use; culled Gaussians go to one extra bucket;
//   2. count per (workgroup slice, bucket) in LDS, scan over (bucket, slice), scan over buckets (binning.h);
//   3. scatter (key - min_bits, index) pairs to their bucket with LDS cursors (arrival order arbitrary);
//   4. the pairs of a bucket are ordered by (key, index) and sorted_idx[rank] plus the 32-byte rank record (mean,
//      conic, opacity, radius, id: what the binning passes read coalesced) are written: buckets of up to 256 pairs by
//      one wave each (ranking by counting, no barriers); longer ones -- listed by the range scan -- by a workgroup each
//      (counting up to 512 pairs, LSD radix over the index digits and the s key bits that differ inside a bucket up to
//      2048 pairs in LDS, ping-ponging in HBM beyond): dense depth layers cost time, not correctness.
#pragma once

#include "binning.h"

namespace mirast {

constexpr int DS_NB = 16384;
constexpr int DS_NBK = DS_NB + 1;  // + the bucket of culled Gaussians
constexpr int DS_MAX_WG = 128;  // measured 16 / 32 / 64 / 128 / 192 / 256 slices on cfg3: depth order 0.122 / 0.092 / 0.078 / 0.074 / 0.076 / 0.077 ms
constexpr int DS_WAVE = 512;     // pairs per bucket ordered by ONE wave (four buckets per workgroup, no barriers)
constexpr int DS_WAVE_COUNT = 128;  // ... by counting up to this many pairs, by a bitonic network in LDS above
constexpr int DS_COUNTING = 512;  // buckets up to this size are ranked by counting instead of radix passes
constexpr int DS_LARGE = 2048;   // pairs per bucket the second kernel holds in LDS (36 KB: four workgroups per CU)

// Key range of the view -> {min key, bucket shift}.  Call from every thread (reads 2 x R_SLOTS words, L2-resident).
struct DepthMap {
    uint32_t kmin;
    int shift;
};
__device__ __forceinline__ DepthMap depth_map(const int* __restrict__ r_slots)
{
    const int lane = threadIdx.x & 63;
    uint32_t inv_min = (uint32_t)r_slots[(lane % R_SLOTS) * R_SLOT_STRIDE + 1];
    uint32_t mx = (uint32_t)r_slots[(lane % R_SLOTS) * R_SLOT_STRIDE + 2];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        inv_min = max(inv_min, (uint32_t)__shfl_xor((int)inv_min, o, 64));
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
    }
    DepthMap m;
    m.kmin = ~inv_min;
    const uint32_t span = mx >= m.kmin ? mx - m.kmin : 0u;  // no visible Gaussian: kmin = 0xFFFFFFFF, span 0
    int s = 0;
    while (s < 32 && (span >> s) >= (uint32_t)DS_NB) s++;
    m.shift = s;
    return m;
}
__device__ __forceinline__ uint32_t depth_bucket(uint32_t key, const DepthMap& m)
{
    if (key == 0xFFFFFFFFu) return (uint32_t)DS_NB;  // culled (geometry.h)
    return (key - m.kmin) >> m.shift;
}

inline int depth_workgroups(int P)
{
    const int blocks = (P + 1023) / 1024;
    return blocks < 1 ? 1 : (blocks > DS_MAX_WG ? DS_MAX_WG : blocks);
}

// Count pass (EMIT = false) and scatter pass (EMIT = true); same structure as bin_ranks_kernel, one item per Gaussian.
template <bool EMIT>
__global__ void __launch_bounds__(1024) depth_bucket_kernel(int P, const uint32_t* __restrict__ depth_key,
                                                            uint32_t* __restrict__ partial,
                                                            const uint2* __restrict__ ranges, uint2* __restrict__ pairs,
                                                            uint32_t* __restrict__ sorted_idx,
                                                            BlendRec* __restrict__ rank_rec,
                                                            const int* __restrict__ r_slots, int* __restrict__ host_r = nullptr)
{
    extern __shared__ uint32_t s_dyn[];  // [DS_NBK] counters / cursors
    const int tid = threadIdx.x;
    // The first kernel behind the preprocess pass hands the partial sums of R to the host: plain stores into its pinned buffer
    // (a copy command of 8 KB costs a 5-us blit kernel on the stream); the event behind this kernel tells the host they are there.
    if (!EMIT && host_r != nullptr && blockIdx.x == 0 && tid < R_SLOTS) host_r[tid * R_SLOT_STRIDE] = r_slots[tid * R_SLOT_STRIDE];
    const DepthMap dm = depth_map(r_slots);
    uint32_t* my_partial = partial + (size_t)slice_row(blockIdx.x, gridDim.x) * DS_NBK;  // slices of one XCD adjacent (binning.h)
    for (int b = tid; b < DS_NBK; b += 1024) s_dyn[b] = EMIT ? ranges[b].x + my_partial[b] : 0u;
    __syncthreads();
    for (int i = blockIdx.x * 1024 + tid; i < P; i += gridDim.x * 1024) {
        const uint32_t key = depth_key[i];
        const uint32_t b = depth_bucket(key, dm);
        if (EMIT) {
            const uint32_t slot = atomicAdd(&s_dyn[b], 1u);
            if (b < (uint32_t)DS_NB) {
                // the pair carries key - min key (same order as the key; the bucket is its high bits, so inside a
                // bucket only its low `shift` bits differ)
                pairs[slot] = make_uint2(key - dm.kmin, (uint32_t)i);
            } else {  // culled: ranks V..P-1 in arbitrary order, radius 0 (every later stage skips them)
                BlendRec rec;
                rec.xy = make_float2(0.f, 0.f);
                rec.id = (uint32_t)i;
                rec.pm = 0u;
                rec.co = make_float4(0.f, 0.f, 0.f, 0.f);
                rank_rec[slot] = rec;
                sorted_idx[slot] = (uint32_t)i;
            }
        } else {
            atomicAdd(&s_dyn[b], 1u);
        }
    }
    if (!EMIT) {
        __syncthreads();
        for (int b = tid; b < DS_NBK; b += 1024) my_partial[b] = s_dyn[b];
    }
}

// ---- (key, value) pair storage for the bucket sort: LDS arrays or one uint2 array in HBM ----------------
struct LdsPairs {
    uint32_t* k;
    uint32_t* v;
    __device__ __forceinline__ uint32_t key(int i) const { return k[i]; }
    __device__ __forceinline__ uint32_t val(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, uint32_t kk, uint32_t vv) const
    {
        k[i] = kk;
        v[i] = vv;
    }
};
struct GlobalPairs {
    uint2* p;
    __device__ __forceinline__ uint32_t key(int i) const { return p[i].x; }
    __device__ __forceinline__ uint32_t val(int i) const { return p[i].y; }
    __device__ __forceinline__ void set(int i, uint32_t kk, uint32_t vv) const { p[i] = make_uint2(kk, vv); }
};

// One stable 8-bit LSD pass over pairs (digit taken from the value when BY_VAL, else from the key); same scheme as
// radix_pass in binning.h: each wave owns a contiguous quarter, ballot-match ranking inside a wave.
template <bool BY_VAL, typename Src, typename Dst>
__device__ __forceinline__ void radix_pass_pairs(Src src, Dst dst, int n, int shift, uint32_t (*s_hist)[256],
                                                 uint32_t* s_wsum, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    const int chunks = (n + 63) >> 6;
    const int cpw = (chunks + 3) >> 2;
    const int begin = min(n, wave * cpw * 64), end = min(n, (wave + 1) * cpw * 64);
    for (int d = tid; d < 4 * 256; d += 256) (&s_hist[0][0])[d] = 0;
    __syncthreads();

    auto match = [&](uint32_t digit, bool active, uint32_t& rank_in_wave, uint32_t& cnt) {
        uint64_t m = ballot64(active);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const uint64_t bal = ballot64(active && ((digit >> b) & 1u));
            m &= ((digit >> b) & 1u) ? bal : ~bal;
        }
        rank_in_wave = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        cnt = (uint32_t)__builtin_popcountll(m);
    };

    for (int i0 = begin; i0 < end; i0 += 64) {
        const int i = i0 + lane;
        const bool active = i < end;
        const uint32_t word = active ? (BY_VAL ? src.val(i) : src.key(i)) : 0u;
        const uint32_t digit = (word >> shift) & 0xFFu;
        uint32_t rk, cnt;
        match(digit, active, rk, cnt);
        if (active && rk == 0) s_hist[wave][digit] += cnt;
    }
    __syncthreads();
    {
        const uint32_t c0 = s_hist[0][tid], c1 = s_hist[1][tid], c2 = s_hist[2][tid], c3 = s_hist[3][tid];
        const uint32_t tot = c0 + c1 + c2 + c3;
        const uint32_t incl = wave_inclusive_scan(tot, lane);
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; w++) woff += s_wsum[w];
        const uint32_t excl = woff + incl - tot;
        s_hist[0][tid] = excl;
        s_hist[1][tid] = excl + c0;
        s_hist[2][tid] = excl + c0 + c1;
        s_hist[3][tid] = excl + c0 + c1 + c2;
    }
    __syncthreads();
    for (int i0 = begin; i0 < end; i0 += 64) {
        const int i = i0 + lane;
        const bool active = i < end;
        const uint32_t kk = active ? src.key(i) : 0u, vv = active ? src.val(i) : 0u;
        const uint32_t digit = ((BY_VAL ? vv : kk) >> shift) & 0xFFu;
        uint32_t rk, cnt;
        match(digit, active, rk, cnt);
        uint32_t off = 0;
        if (active) off = s_hist[wave][digit] + rk;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (active && rk == cnt - 1) s_hist[wave][digit] = off + 1;
        if (active) dst.set((int)off, kk, vv);
    }
    __syncthreads();
}

// Buckets of up to DS_WAVE pairs: one WAVE per bucket, four buckets per workgroup, no workgroup barriers.
//   * up to DS_WAVE_COUNT pairs (the common case: ~100 pairs on a 1 M-Gaussian view): ranking by counting -- every pair is compared
//     with every other one through LDS broadcast reads; (key, index) pairs are distinct, so the ranks are a permutation;
//   * above (the common case of a 5 M-Gaussian view: ~300 pairs per bucket, where counting is O(n^2): 0.17 of cfg5's 0.33-ms depth
//     order until round 5): a bitonic network over the pairs padded to 256 / 512 in LDS, wave-synchronous -- log^2 steps of n / 2
//     compare-exchanges instead of n^2 comparisons.
// Either way the wave then scatters sorted_idx / the 32-byte records by rank.
__global__ void __launch_bounds__(256) depth_bucket_sort_wave_kernel(const uint2* __restrict__ ranges,
                                                                     const uint2* __restrict__ pairs,
                                                                     const BlendRec* __restrict__ index_rec,
                                                                     uint32_t* __restrict__ sorted_idx,
                                                                     BlendRec* __restrict__ rank_rec)
{
    __shared__ uint64_t s_p[4][DS_WAVE];   // (key << 32 | index) of the wave's bucket
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= DS_NB) return;
    const uint2 range = ranges[b];
    const int n = (int)(range.y - range.x);
    if (n == 0 || n > DS_WAVE) return;  // longer buckets are on the big-bucket list
    uint64_t* const sp = s_p[wave];
    if (n <= DS_WAVE_COUNT) {
        constexpr int EPL = DS_WAVE_COUNT / 64;   // elements per lane
        uint64_t mine[EPL];
#pragma unroll
        for (int t = 0; t < EPL; t++) {
            const int e = lane + 64 * t;
            uint2 p = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
            if (e < n) p = pairs[range.x + e];
            mine[t] = ((uint64_t)p.x << 32) | p.y;
            if (e < n) sp[e] = mine[t];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        int rank[EPL];
#pragma unroll
        for (int t = 0; t < EPL; t++) rank[t] = 0;
        for (int q = 0; q < n; q++) {
            const uint64_t other = sp[q];
#pragma unroll
            for (int t = 0; t < EPL; t++) rank[t] += other < mine[t] ? 1 : 0;
        }
#pragma unroll
        for (int t = 0; t < EPL; t++) {
            if (lane + 64 * t < n) {
                const uint32_t g = (uint32_t)mine[t];
                rank_rec[range.x + rank[t]] = index_rec[g];  // one 32-byte gather: {mean, id, radius, conic + opacity}
                sorted_idx[range.x + rank[t]] = g;
            }
        }
        return;
    }
    // bitonic network over m = 256 or 512 slots (padding: the largest word, which ends up behind every pair)
    const int m = n <= 256 ? 256 : 512;
    for (int e = lane; e < m; e += 64) {
        uint64_t w = 0xFFFFFFFFFFFFFFFFull;
        if (e < n) {
            const uint2 p = pairs[range.x + e];
            w = ((uint64_t)p.x << 32) | p.y;
        }
        sp[e] = w;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    for (int k = 2; k <= m; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < (m >> 1); t += 64) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // the t-th index whose bit j is clear; its partner is i + j
                const uint64_t a = sp[i], c = sp[i + j];
                const bool up = (i & k) == 0;                          // ascending block
                if ((a > c) == up) {
                    sp[i] = c;
                    sp[i + j] = a;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (the compare-exchanges of one step touch disjoint pairs)
        }
    }
    for (int e = lane; e < n; e += 64) {
        const uint32_t g = (uint32_t)sp[e];
        rank_rec[range.x + e] = index_rec[g];
        sorted_idx[range.x + e] = g;
    }
}

// Sorts the pairs of one bucket by (key, index) and writes sorted_idx / rank_rec for its ranks.
// BIG = true (the only instantiation in use): workgroups walk the list of buckets with more than LO pairs (built by
// tile_ranges_kernel); up to CAP pairs in LDS (counting up to 512, LSD radix above), more than that in HBM.
template <int LO, int CAP, bool BIG>
__global__ void __launch_bounds__(256) depth_bucket_sort_kernel(const uint2* __restrict__ ranges,
                                                                const uint32_t* __restrict__ big_list,
                                                                uint2* __restrict__ pairs, uint2* __restrict__ pairs_tmp,
                                                                int idx_passes, const BlendRec* __restrict__ index_rec,
                                                                uint32_t* __restrict__ sorted_idx,
                                                                BlendRec* __restrict__ rank_rec,
                                                                const int* __restrict__ r_slots)
{
    __shared__ uint32_t s_k[2][CAP];
    __shared__ uint32_t s_v[2][CAP];
    __shared__ uint32_t s_hist[4][256];
    __shared__ uint32_t s_wsum[4];
    const int tid = threadIdx.x;
    const int key_passes = (depth_map(r_slots).shift + 7) / 8;  // inside a bucket only the low `shift` key bits differ
    const int nwork = BIG ? (int)big_list[0] : 1;
    for (int j = BIG ? (int)blockIdx.x : 0; j < nwork; j += BIG ? (int)gridDim.x : 1) {
        const int b = BIG ? (int)big_list[1 + j] : (int)blockIdx.x;
        const uint2 range = ranges[b];
        const int n = (int)(range.y - range.x);
        if (!BIG && (n == 0 || n > CAP)) return;
        auto finalize = [&](auto src) {
            for (int i = tid; i < n; i += 256) {
                const uint32_t g = src.val(i);
                rank_rec[range.x + i] = index_rec[g];  // one 32-byte gather: {mean, id, radius, conic + opacity}
                sorted_idx[range.x + i] = g;
            }
        };
        if (n <= CAP) {
            for (int i = tid; i < n; i += 256) {
                const uint2 p = pairs[range.x + i];
                s_k[0][i] = p.x;
                s_v[0][i] = p.y;
            }
            __syncthreads();
            int cur = 0;
            if (n <= DS_COUNTING) {
                // short bucket (the common case): rank by counting -- every pair is compared with every other one
                // through LDS broadcast reads; (key, index) pairs are distinct, so the ranks are a permutation
                for (int e = tid; e < n; e += 256) {
                    const uint32_t mk = s_k[0][e], mv = s_v[0][e];
                    const uint64_t mine = ((uint64_t)mk << 32) | mv;
                    int rank = 0;
#pragma unroll 4
                    for (int q = 0; q < n; q++) rank += ((((uint64_t)s_k[0][q]) << 32) | s_v[0][q]) < mine ? 1 : 0;
                    s_k[1][rank] = mk;
                    s_v[1][rank] = mv;
                }
                __syncthreads();
                finalize(LdsPairs{s_k[1], s_v[1]});
                __syncthreads();
                continue;
            }
            for (int p = 0; p < idx_passes; p++, cur ^= 1)
                radix_pass_pairs<true>(LdsPairs{s_k[cur], s_v[cur]}, LdsPairs{s_k[cur ^ 1], s_v[cur ^ 1]}, n, 8 * p, s_hist, s_wsum, tid);
            for (int p = 0; p < key_passes; p++, cur ^= 1)
                radix_pass_pairs<false>(LdsPairs{s_k[cur], s_v[cur]}, LdsPairs{s_k[cur ^ 1], s_v[cur ^ 1]}, n, 8 * p, s_hist, s_wsum, tid);
            finalize(LdsPairs{s_k[cur], s_v[cur]});
        } else {
            uint2* a = pairs + range.x;
            uint2* t = pairs_tmp + range.x;
            for (int p = 0; p < idx_passes; p++) {
                radix_pass_pairs<true>(GlobalPairs{a}, GlobalPairs{t}, n, 8 * p, s_hist, s_wsum, tid);
                uint2* x = a;
                a = t;
                t = x;
            }
            for (int p = 0; p < key_passes; p++) {
                radix_pass_pairs<false>(GlobalPairs{a}, GlobalPairs{t}, n, 8 * p, s_hist, s_wsum, tid);
                uint2* x = a;
                a = t;
                t = x;
            }
            finalize(GlobalPairs{a});
        }
        __syncthreads();  // LDS reuse by the next bucket
    }
}

}  // namespace mirast
