// binning.h -- tile binning for gfx950.
//
// What the reference does (CF/cuda_rasterizer/rasterizer_impl.cu:70-138,277-317): prefix-sum the
// per-Gaussian tile counts, emit one 64-bit key (tile<<32 | depth_bits) + 32-bit value per overlap,
// run a 45-bit device-wide radix sort over all R pairs (6 passes x 24 B/pair of HBM traffic), then
// find per-tile ranges.  The RESULT -- point_list ordered by (tile, depth bits, Gaussian index) and
// ranges[tile] -- is part of the bit-exact integer contract; the way to get there is not.
//
// MI355X design (produces the identical point_list / ranges):
//   1. the per-Gaussian preprocess kernel sums tiles_touched into R (what the host needs to size the binning buffer);
//   2. ONE 32-bit radix sort of the P Gaussians by depth bits (stable, so ties keep index order) turns
//      every visible Gaussian into a dense RANK in [0,V): ordering by rank == ordering by (depth, index);
//   3. a count pass over rank slices (per-tile counters in LDS), a scan over (tile, slice), and the scan of the tile
//      totals give ranges[tile] and every slice's base inside every tile -- no global atomics;
//   4. rank emission: each (Gaussian, tile) overlap takes its slot from an LDS cursor and stores the 4-byte RANK
//      (arrival order inside a tile is arbitrary within a slice);
//   5. per-tile LDS radix sort of the ranks (<= 24 significant bits, 8-bit digits, stable wave-match
//      ranking), then blend_list[slot] = sorted_idx[rank] | quadrant mask << 28 (full lists: the low 28 bits are point_list).
// HBM traffic per overlap drops from ~172 B to ~16 B (4 B emit write, 4 B sort read, 4 B list
// write, 4 B L2-resident gather); the 64-bit keys are never materialised.
#pragma once

#include "common.h"
#include "cull.h"

namespace mirast {

// ---- workgroup-balanced enumeration of tile rects -------------------------------------------------------
// Each of the 1024 threads owns one Gaussian with `count` tiles (0 if culled) in rect [rmin, rmax).  Tile counts
// are extremely skewed along the depth-rank axis (BASELINE cfg3: median 6 tiles, 99.9th percentile 484, maximum
// 4056; the heaviest 64 consecutive ranks hold 12x the average), so the WORKGROUP walks the concatenation of all
// 1024 rects with every thread busy: item k belongs to the thread whose inclusive prefix first exceeds k (binary
// search in an LDS copy of the prefix), and its tile follows from k's offset inside that rect.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// Row of partial[][] (= position of the slice inside every tile's segment) of workgroup b.  Workgroup b runs on XCD b % 8
// (tools/xcc_probe.hip) and each XCD has its own L2: with the slices of one XCD next to each other, the 4-byte entries that
// share a 128-byte line of a tile's segment are mostly stored from ONE XCD instead of eight (measured: emit 0.086 -> 0.074 ms;
// the HBM write bytes of the pass did not change).  Any bijection is correct -- count, scan and emit only have to agree on it.
__device__ __forceinline__ uint32_t slice_row(uint32_t b, uint32_t nwg)
{
    return (nwg & 7u) ? b : (b & 7u) * (nwg >> 3) + (b >> 3);
}

struct RectWork {
    uint32_t* prefix;  // LDS [NT] inclusive prefix of counts
    uint32_t* rx;      // LDS [NT] rect_min.x | width << 16
    uint32_t* ry;      // LDS [NT] rect_min.y
    uint32_t* wsum;    // LDS [16]
};

// Contains workgroup barriers: call from all NT threads (NT = 1024, or 512: two workgroups per CU).  f(owner_thread, tile_x, tile_y)
// handles one item.
template <int NT = 1024, typename F>
__device__ __forceinline__ void for_each_tile_balanced(const RectWork& rw, int tid, uint2 rmin, uint2 rmax, uint32_t count, F&& f)
{
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t incl = wave_inclusive_scan(count, lane);
    if (lane == 63) rw.wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; w++) {
        const uint32_t c = rw.wsum[w];
        woff += w < wave ? c : 0u;
        total += c;
    }
    incl += woff;
    rw.prefix[tid] = incl;
    rw.rx[tid] = rmin.x | ((rmax.x - rmin.x) << 16);
    rw.ry[tid] = rmin.y;
    __syncthreads();
    for (uint32_t k = (uint32_t)tid; k < total; k += NT) {
        // first thread o with prefix[o] > k
        int lo = 0;
#pragma unroll
        for (int step = NT / 2; step >= 1; step >>= 1)
            if (rw.prefix[lo + step - 1] <= k) lo += step;
        const uint32_t packed = rw.rx[lo];
        const uint32_t w = packed >> 16, x0 = packed & 0xFFFFu, y0 = rw.ry[lo];
        const uint32_t prev = lo == 0 ? 0u : rw.prefix[lo - 1];
        const uint32_t i = k - prev;
        // row = i / w without an integer divide: (i + 0.5) / w is never within float error of an integer
        // boundary for i < 2^14 * w (a Gaussian covers at most grid_x * grid_y tiles)
        const uint32_t row = (uint32_t)(((float)i + 0.5f) * __builtin_amdgcn_rcpf((float)w));
        const uint32_t col = i - row * w;
        f((uint32_t)lo, x0 + col, y0 + row);
    }
    __syncthreads();  // LDS hand-off arrays are reused by the next call
}

// ---- 3. tile ranges: exclusive scan of the per-tile totals (single workgroup) ------------------------------
// Writes ranges[tile] = [base, base+count) (== identifyTileRanges' result, rasterizer_impl.cu:116-138, including
// {0,0} for empty tiles as left by the reference's cudaMemset), R and the longest list.
constexpr int RANK_BITS = 28;          // entry = depth rank | quadrant mask << 28
constexpr uint32_t RANK_MASK = (1u << RANK_BITS) - 1u;
constexpr int BIN_MAX_WG = 256;        // workgroups of the count / emit passes (rank slices): one per CU, all resident at once
                                       // (measured 512 / 256 / 128 / 64 slices on cfg3: count + emit 0.167 / 0.146 / 0.199 / 0.349 ms)
constexpr int BIN_THREADS = 1024;
constexpr int BIN_MAX_TILES = 26 * 1024 - 64;  // per launch of the count / emit passes: one LDS counter per tile + 57 KB of hand-off
                                               // arrays must fit in 160 KB; larger images are walked in bands of tile rows
constexpr int BIN_MAX_TILES_TOTAL = 40 * 1024 - 64;  // tile_ranges_kernel scans all tile totals in one workgroup's LDS

__global__ void __launch_bounds__(1024) tile_ranges_kernel(int ntiles_all, const uint32_t* __restrict__ tile_total,
                                                           uint2* __restrict__ ranges, int* __restrict__ num_rendered,
                                                           uint32_t big_threshold, int big_limit,
                                                           uint32_t* __restrict__ big_list, int* __restrict__ host_out = nullptr,
                                                           uint32_t* __restrict__ zero_a = nullptr, uint32_t* __restrict__ zero_b = nullptr,
                                                           uint32_t* __restrict__ run_bounds = nullptr /* [9]: the blend kernels' XCD runs
                                                               (common.h): equal tile counts, or equal MODELLED work when run_cap > 0 */,
                                                           uint32_t run_cap = 0, uint32_t run_fix = 0)
{
    // The totals are staged in LDS (coalesced), thread t scans the contiguous items [t*per, (t+1)*per) in place,
    // one workgroup scan joins the pieces, and the ranges leave coalesced again.  Items below big_limit with more
    // than big_threshold entries are appended to big_list (count in big_list[0], order arbitrary).
    // More than BIN_MAX_TILES_TOTAL items (images beyond 10 Mpx) are walked in segments of that many, one after the other, the
    // running total carried along: ONE pass of the loop below for every image up to 4096 x 2544.
    // run_cap > 0 (images of one segment): the XCD runs of the blend kernels are cut at equal sums of the
    // MODELLED cost of a tile, min(list length, run_cap) + run_fix -- a list is walked until its pixels are opaque, which takes about
    // run_cap entries where the scene is dense, and to its end where it is sparse; what a walk really covers is only known behind the
    // forward blend (run_bounds_from_walks_kernel) -- the second prefix sum rides on the first, at the granularity of the threads'
    // pieces (registers only); the thread whose piece holds a boundary walks its few tiles.
    extern __shared__ uint32_t s_val[];  // [min(ntiles_all, BIN_MAX_TILES_TOTAL) + 1]
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_wave_w[16];
    __shared__ uint32_t s_bound[9];
    const bool weighted = run_bounds != nullptr && run_cap > 0u && ntiles_all <= BIN_MAX_TILES_TOTAL;   // (one segment)
    __shared__ uint32_t s_maxcount;
    __shared__ uint32_t s_nbig;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 9) s_bound[tid] = xcd_run_start((uint32_t)tid, (uint32_t)ntiles_all);   // equal tile counts unless the model says otherwise
    if (tid == 0) {
        s_maxcount = 0;
        s_nbig = 0;
    }
    uint32_t carry = 0;   // entries in front of the segment (the same in every thread)
    for (int seg0 = 0; seg0 < ntiles_all; seg0 += BIN_MAX_TILES_TOTAL) {
        const int ntiles = min(BIN_MAX_TILES_TOTAL, ntiles_all - seg0);
        for (int i = tid; i < ntiles; i += 1024) s_val[i] = tile_total[seg0 + i];
        __syncthreads();
        const int per = (ntiles + 1023) / 1024;
        const int i0 = min(ntiles, tid * per), i1 = min(ntiles, i0 + per);
        uint32_t sum = 0, mx = 0, wsum = 0;
        for (int i = i0; i < i1; i++) {
            const uint32_t c = s_val[i];
            s_val[i] = sum;  // exclusive prefix inside the piece
            sum += c;
            mx = max(mx, c);
            wsum += min(c, run_cap) + run_fix;
        }
        const uint32_t incl = wave_inclusive_scan(sum, lane);
        const uint32_t incl_w = weighted ? wave_inclusive_scan(wsum, lane) : 0u;
        if (lane == 63) {
            s_wave[wave] = incl;
            s_wave_w[wave] = incl_w;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
        __syncthreads();
        if (lane == 0 && mx) atomicMax(&s_maxcount, mx);
        uint32_t woff = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const uint32_t c = s_wave[w];
            woff += w < wave ? c : 0u;
            total += c;
        }
        const uint32_t piece_base = carry + woff + incl - sum;
        for (int i = i0; i < i1; i++) s_val[i] += piece_base;
        if (tid == 0) s_val[ntiles] = carry + total;
        uint32_t woff_w = 0, total_w = 0;
        if (weighted) {
#pragma unroll
            for (int w = 0; w < 16; w++) {
                const uint32_t c = s_wave_w[w];
                woff_w += w < wave ? c : 0u;
                total_w += c;
            }
        }
        __syncthreads();
        if (weighted && total_w > 0u) {
            // boundary k = first tile i with 8 W(i) >= k W_total, W(i) = modelled work in front of tile i: it lies in (i0, i1] of exactly
            // one piece -- the one with 8 W(i0) < k W_total <= 8 W(i1) -- whose thread walks its tiles (list lengths = differences of
            // the finished prefix)
            const uint64_t w_lo = 8ull * (uint64_t)(woff_w + incl_w - wsum), w_hi = 8ull * (uint64_t)(woff_w + incl_w);
#pragma unroll
            for (uint32_t k = 1; k < 8u; k++) {
                const uint64_t want = (uint64_t)k * (uint64_t)total_w;
                if (w_lo < want && want <= w_hi) {
                    uint64_t wacc = w_lo;
                    int i = i0;
                    for (; i < i1; i++) {
                        wacc += 8ull * (uint64_t)(min(s_val[i + 1] - s_val[i], run_cap) + run_fix);
                        if (wacc >= want) break;
                    }
                    s_bound[k] = (uint32_t)min(i + 1, i1);
                }
            }
        }
        for (int i = tid; i < ntiles; i += 1024) {
            const uint32_t lo = s_val[i], hi = s_val[i + 1];
            ranges[seg0 + i] = hi > lo ? make_uint2(lo, hi) : make_uint2(0u, 0u);
            if (zero_a != nullptr) {   // per-tile words a later kernel accumulates into with atomicMax (blend_fwd_wave.h)
                zero_a[seg0 + i] = 0u;
                zero_b[seg0 + i] = 0u;
            }
            if (big_list != nullptr && seg0 + i < big_limit && hi - lo > big_threshold)
                big_list[1 + atomicAdd(&s_nbig, 1u)] = (uint32_t)(seg0 + i);
        }
        carry += total;
        __syncthreads();   // s_val / s_wave are rewritten by the next segment
    }
    if (run_bounds != nullptr) {
        const uint32_t n = (uint32_t)ntiles_all;   // (s_bound: behind the barrier that ends the segment loop)
        if (tid == 0) {   // the clamp in registers (xcd_clamp_runs on LDS words would be seven dependent round trips)
            uint32_t bb[9];
#pragma unroll
            for (int k = 1; k < 8; k++) bb[k] = s_bound[k];
            const uint32_t maxrun = xcd_max_run(n);
            bb[0] = 0u;
            bb[8] = n;
#pragma unroll
            for (uint32_t k = 1; k < 8u; k++) {
                const uint32_t prev = bb[k - 1], need = (8u - k) * maxrun;
                bb[k] = min(max(bb[k], max(prev, n > need ? n - need : 0u)), min(n, prev + maxrun));
            }
#pragma unroll
            for (int k = 0; k < 9; k++) {
                run_bounds[k] = bb[k];
                if (host_out != nullptr) host_out[4 + k] = (int)bb[k];   // the forward's grid is sized for its longest run
            }
        }
    }
    if (tid == 0) {
        num_rendered[0] = (int)carry;       // R (the host already has it from the preprocess pass; kept for checks)
        num_rendered[1] = (int)s_maxcount;  // longest list
        if (host_out != nullptr) {          // the same two words straight into the host's pinned buffer (no copy command)
            host_out[0] = (int)carry;
            host_out[1] = (int)s_maxcount;
        }
        if (big_list != nullptr) big_list[0] = s_nbig;
    }
}

// Run boundaries of the BACKWARD blend from what the forward walked (tile_nsurv: the entries of a tile's list the forward reached
// before every pixel was opaque -- the tile's cost in both blend kernels; common.h "WORK-balanced runs"), one workgroup, launched
// behind the forward blend of a view that will be differentiated: scan of (tile_nsurv + XCD_TILE_WEIGHT) over the row-major tile
// sequence, seven searches for the points of equal weight, the clamp that bounds the backward's grid.  Images above
// BIN_MAX_TILES_TOTAL tiles keep the equal-count boundaries of the range scan.
__global__ void __launch_bounds__(1024) run_bounds_from_walks_kernel(int ntiles, const uint32_t* __restrict__ tile_nsurv,
                                                                      uint32_t* __restrict__ run_bounds)
{
    extern __shared__ uint32_t s_w[];   // [ntiles + 1] exclusive prefix of the weights
    __shared__ uint32_t s_wave2[16];
    __shared__ uint32_t s_bound2[9];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n = (uint32_t)ntiles;
    for (int i = tid; i < ntiles; i += 1024) s_w[i] = tile_nsurv[i] + XCD_TILE_WEIGHT;
    __syncthreads();
    const int per = (ntiles + 1023) / 1024;
    const int i0 = min(ntiles, tid * per), i1 = min(ntiles, i0 + per);
    uint32_t sum = 0;
    for (int i = i0; i < i1; i++) {
        const uint32_t c = s_w[i];
        s_w[i] = sum;
        sum += c;
    }
    const uint32_t incl = wave_inclusive_scan(sum, lane);
    if (lane == 63) s_wave2[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
        const uint32_t c = s_wave2[w];
        woff += w < wave ? c : 0u;
        total += c;
    }
    const uint32_t base = woff + incl - sum;
    for (int i = i0; i < i1; i++) s_w[i] += base;
    if (tid == 0) s_w[ntiles] = total;
    __syncthreads();
    if (tid >= 1 && tid <= 7) {   // first tile i with 8 W(i) >= k W_total, W(i) = weight in front of tile i
        const uint64_t want = (uint64_t)tid * (uint64_t)total;
        uint32_t lo = 0, hi = n;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (8ull * (uint64_t)s_w[mid] >= want) hi = mid;
            else lo = mid + 1;
        }
        s_bound2[tid] = lo;
    }
    __syncthreads();
    if (tid == 0) {
        xcd_clamp_runs(s_bound2, n);
        for (int k = 0; k < 9; k++) run_bounds[k] = s_bound2[k];
    }
}

// Everything needed to evaluate a Gaussian at a pixel in ONE 32-byte record (the reference gathers id -> xy -> conic per batch
// with dependent loads, forward.cu:318-326): written per Gaussian by the preprocess pass (index_rec, `pm` = radius), permuted into
// depth-rank order for the binning passes (rank_rec), and what the blend kernels stage per list entry.
struct __attribute__((aligned(16))) BlendRec {
    float2 xy;       // pixel-space mean
    uint32_t id;     // Gaussian index (feature row)
    uint32_t pm;     // (position in the tile list) << 4 | quadrant mask
    float4 co;       // conic A,B,C + opacity
};
static_assert(sizeof(BlendRec) == 32, "BlendRec must be 32 bytes");

// Entry i of a tile's blend list (4 bytes: Gaussian id | quadrant mask << 28, see emit_blend_list below) as a full record: the
// geometry record of that Gaussian (index_rec, written by the preprocess pass) with `pm` = position << 4 | quadrant mask.  For the
// kernels that walk a list in whole batches (blend_fwd.h, blend_bwd.h); the wave kernels gather per quadrant.
__device__ __forceinline__ BlendRec list_record(const uint32_t* __restrict__ lst, const BlendRec* __restrict__ index_rec, int i)
{
    const uint32_t e = lst[i];
    BlendRec r = index_rec[e & RANK_MASK];
    r.pm = ((uint32_t)i << 4) | (e >> RANK_BITS);
    return r;
}

// ---- per-rank geometry records ------------------------------------------------------------------------
// The per-Gaussian data the binning stages need (pixel mean, conic, opacity, radius, id) is written by the preprocess pass as ONE
// 32-byte record per Gaussian (index_rec) and permuted once into depth-rank order by the depth sort (rank_rec, depth_sort.h): the count
// and emit passes read the ranks coalesced; the blend kernels gather index_rec[id] per list entry they reach.

// ---- 4. counting and rank emission -------------------------------------------------------------------
// Global atomics are the scarce resource of the binning stages (about 25 G scattered dword atomics/s on this part:
// one corner atomic per Gaussian and one cursor atomic per overlap were 0.11 + 0.27 ms per view).  So the bucketing
// is done the way a radix-sort pass does it.  The depth ranks are dealt to <= 512 workgroups of 1024 threads in
// chunks of 64 ranks, round robin; the set of ranks a workgroup owns is its "slice":
//   count pass  (EMIT = false): per-tile counters in LDS, one LDS atomic per overlap; the counters go to
//                partial[slice][tile] with plain stores;
//   scan        (scan_partials_kernel): per tile, the exclusive prefix over slices (in place) and the tile total;
//   emit pass   (EMIT = true): LDS cursors start at ranges[tile].x + partial[slice][tile]; every overlap takes
//                its slot with a returning LDS atomic and stores its 4-byte depth rank.
// Both passes enumerate the overlaps with the same code, so the counts cannot disagree with the emission.  Inside a
// tile the entries of one slice arrive in arbitrary order; the per-tile sort below does not care.
inline int bin_workgroups(int P)
{
    const int blocks = (P + BIN_THREADS - 1) / BIN_THREADS;
    return blocks < 1 ? 1 : (blocks > BIN_MAX_WG ? BIN_MAX_WG : blocks);
}

// Emit pass of the FULL lists (the reference's `debug` flag, MI_RAST_FULL_LISTS: parity tests): every overlap of the
// reference's rects is stored, with the quadrant mask the lean lists would give it (cull.h: span_tile_mask; 0 = culled) in
// the top 4 bits -- the per-tile sort then reproduces the reference's point_list (the low 28 bits of the blend list), and the blend
// kernels skip the entries without a mask bit.  NOCULL (testing aid, MI_RAST_NO_CULL): every overlap is kept with all four quadrant bits.
// The product default does not come here: bin_spans_kernel below.
template <bool NOCULL = false>
__global__ void __launch_bounds__(BIN_THREADS) bin_ranks_kernel(int P, const BlendRec* __restrict__ rank_rec,
                                                                const uint32_t* __restrict__ partial,
                                                                const uint2* __restrict__ ranges,
                                                                uint32_t* __restrict__ entries, uint32_t gx, uint32_t gy,
                                                                uint32_t by0, uint32_t by1)
{
    // One launch covers the tile rows [by0, by1) (the whole image unless it has more tiles than LDS counters: then the host
    // walks it in bands); the LDS cursors are indexed relative to the band, everything in memory by the global tile id.
    extern __shared__ uint32_t s_dyn[];  // [band tiles] cursors, then the rect hand-off arrays
    const int ntiles = (int)(gx * (by1 - by0));
    const int tile0 = (int)(gx * by0);
    const int ntiles_all = (int)(gx * gy);
    uint32_t* s_cnt = s_dyn;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* s_rw = s_dyn + ((ntiles + 3) & ~3);  // 16-byte aligned
    RectWork rw{s_rw, s_rw + 1024, s_rw + 2048, s_rw + 3072};
    float2* s_xy = reinterpret_cast<float2*>(s_rw + 3088);          // the owners' means ...
    float4* s_co = reinterpret_cast<float4*>(s_rw + 3088 + 2048);   // ... conics + opacities ...
    uint32_t* s_rad = s_rw + 3088 + 6144;                           // ... and radii
    const uint32_t* my_partial = partial + (size_t)slice_row(blockIdx.x, gridDim.x) * ntiles_all + tile0;
    for (int t = tid; t < ntiles; t += BIN_THREADS) s_cnt[t] = ranges[tile0 + t].x + my_partial[t];
    __syncthreads();
    // 64-rank chunks are dealt round robin over all the waves of all the workgroups: chunk c belongs to workgroup
    // c % nwg, wave slot (c / nwg) % 16, round (c / nwg) / 16 -- the heavy (near) chunks end up in different workgroups
    const int nwg = (int)gridDim.x;
    const int rounds = ((P + 63) / 64 + 16 * nwg - 1) / (16 * nwg);
    for (int it = 0; it < rounds; it++) {
        const int r = ((it * 16 + wave) * nwg + (int)blockIdx.x) * 64 + lane;
        uint2 rmin = make_uint2(0, 0), rmax = make_uint2(0, 0);
        uint32_t count = 0;
        if (r < P) {
            const BlendRec rec = rank_rec[r];
            const int rad = (int)rec.pm;
            if (rad > 0) {
                getRect(rec.xy.x, rec.xy.y, rad, rmin, rmax, gx, gy);
                rmin.y = max(rmin.y, by0);  // this band's rows only
                rmax.y = min(rmax.y, by1);
                count = rmax.y > rmin.y ? (rmax.x - rmin.x) * (rmax.y - rmin.y) : 0u;
            }
            s_xy[tid] = rec.xy;
            s_co[tid] = rec.co;
            s_rad[tid] = rec.pm;
        }
        for_each_tile_balanced(rw, tid, rmin, rmax, count, [&](uint32_t owner, uint32_t tx, uint32_t ty) {
            const uint32_t qmask = NOCULL ? 15u : span_tile_mask(s_xy[owner], s_co[owner], (int)s_rad[owner], tx, ty, gx, gy);
            const uint32_t rank = (uint32_t)(((it * 16 + (int)(owner >> 6)) * nwg + (int)blockIdx.x) * 64 + (int)(owner & 63u));
            const uint32_t slot = atomicAdd(&s_cnt[(ty - by0) * gx + tx], 1u);
            entries[slot] = rank | (qmask << RANK_BITS);
        });
    }
}

// Count pass without enumerating the overlaps: every rect adds +1 / -1 / -1 / +1 at its four corners of a
// (gy + 1) x (gx + 1) difference grid in LDS (four LDS atomics per Gaussian instead of one per covered tile plus a
// ten-step owner search), and a 2-D prefix sum of the grid is the number of rects covering each tile -- exactly what
// an enumeration would count.  Same slices (rank chunks dealt round robin) and the same rects as bin_ranks_kernel: the count
// pass of the FULL lists.
__host__ __device__ inline int count_grid_stride(uint32_t gx) { return (int)((gx + 1) | 1u); }  // odd row stride: column walks spread over the banks

__global__ void __launch_bounds__(BIN_THREADS) bin_count_kernel(int P, const BlendRec* __restrict__ rank_rec,
                                                                uint32_t* __restrict__ partial, uint32_t gx, uint32_t gy_all,
                                                                uint32_t by0, uint32_t by1)
{
    // tile rows [by0, by1) of the image (see bin_ranks_kernel); the difference grid covers the band only
    extern __shared__ int s_grid[];  // [(band rows + 1) * stride]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int stride = count_grid_stride(gx);
    const uint32_t gy = by1 - by0;
    const int cells = (int)(gy + 1) * stride;
    for (int c = tid; c < cells; c += BIN_THREADS) s_grid[c] = 0;
    __syncthreads();
    const int nwg = (int)gridDim.x;
    const int rounds = ((P + 63) / 64 + 16 * nwg - 1) / (16 * nwg);
    for (int it = 0; it < rounds; it++) {
        const int r = ((it * 16 + wave) * nwg + (int)blockIdx.x) * 64 + lane;  // same dealing as bin_ranks_kernel
        if (r >= P) continue;
        const BlendRec rec = rank_rec[r];
        const int rad = (int)rec.pm;
        if (rad <= 0) continue;
        uint2 rmin, rmax;
        getRect(rec.xy.x, rec.xy.y, rad, rmin, rmax, gx, gy_all);
        rmin.y = max(rmin.y, by0);
        rmax.y = min(rmax.y, by1);
        if (rmax.x <= rmin.x || rmax.y <= rmin.y) continue;
        rmin.y -= by0;
        rmax.y -= by0;
        atomicAdd(&s_grid[rmin.y * stride + rmin.x], 1);
        atomicAdd(&s_grid[rmin.y * stride + rmax.x], -1);
        atomicAdd(&s_grid[rmax.y * stride + rmin.x], -1);
        atomicAdd(&s_grid[rmax.y * stride + rmax.x], 1);
    }
    __syncthreads();
    // prefix along x: one wave per row, 64 cells at a time with a carry
    for (int y = wave; y < (int)gy; y += BIN_THREADS / 64) {
        int carry = 0;
        for (int x0 = 0; x0 < (int)gx; x0 += 64) {
            const int x = x0 + lane;
            int v = x < (int)gx ? s_grid[y * stride + x] : 0;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(v, o, 64);
                if (lane >= o) v += t;
            }
            v += carry;
            if (x < (int)gx) s_grid[y * stride + x] = v;
            carry = __shfl(v, 63, 64);
        }
    }
    __syncthreads();
    // prefix along y: one thread per column; the running sums are the per-tile counts of this slice
    uint32_t* my_partial = partial + (size_t)slice_row(blockIdx.x, gridDim.x) * (gx * gy_all) + (size_t)gx * by0;
    for (int x = tid; x < (int)gx; x += BIN_THREADS) {
        int run = 0;
        for (int y = 0; y < (int)gy; y++) {
            run += s_grid[y * stride + x];
            s_grid[y * stride + x] = run;
        }
    }
    __syncthreads();
    for (int t = tid; t < (int)(gx * gy); t += BIN_THREADS) my_partial[t] = (uint32_t)s_grid[(t / (int)gx) * stride + (t % (int)gx)];
}

// ---- lean lists from row spans (the product default) ------------------------------------------------------------
// The count and emit passes of the lean lists without a single per-tile test.  Items of the first level are (Gaussian,
// tile row) pairs, balanced over the workgroup like the tiles of bin_ranks_kernel (a Gaussian covers 1 .. 68 rows); each
// evaluates the two closed-form column intervals of its row's upper and lower 8-pixel band (cull.h: band_columns) and
// gets the tile span [x0, x1) that holds them.
//   COUNT (EMIT = false): +1 / -1 at the two ends of the span in a per-row difference grid in LDS; the prefix along x is
//                the number of spans covering each tile -- this slice's share of the tile's segment, EXACTLY: a tile is
//                counted iff the emit pass stores an entry for it, so the segments have no unused slots;
//   EMIT:        second level, the tiles of the 1024 spans of a window, balanced again: the quadrant mask of a tile is read
//                off the two column intervals (four range tests on integers), the slot comes from the LDS cursor.
// Measured on cfg3: 8.68 M tiles in the shrunk rects, 5.2 M in the spans; the enumerating emit pass spent 182 VALU
// instructions per 64 rect tiles on the whole-tile test and 386 per 64 survivors on the four quadrant tests.
// Slices (rank chunks dealt round robin), partial[][] and the cursors are those of bin_count_kernel / bin_ranks_kernel.
// LDS words of the hand-off arrays of a workgroup of nt threads: RectWork + means + conics + prefix / rect / radius / two bands' columns
__host__ __device__ constexpr int span_lds_words(int nt) { return (3 * nt + 16) + 2 * nt + 4 * nt + 5 * nt; }
constexpr int SPAN_LDS_WORDS = span_lds_words(1024);
// Threads per workgroup of the LEAN count / emit passes.  512 (profiling build, MI_RAST_BIN_NT): two workgroups share a CU (2 x 61 KB of
// LDS at 1080p) and overlap each other's workgroup-barrier phases, at the price of twice as many rank slices.  Measured in round 5
// (profiles/r05_front_half.md): emit 0.071 -> 0.065 ms, but count + scans 0.057 -> 0.068 (the partial[][] table and its scan double):
// 1024 stays.
constexpr int BIN_LEAN_THREADS = 1024;
constexpr int BIN_LEAN_MAX_WG = 512;
inline int bin_lean_workgroups(int P, int nt)
{
    const int blocks = (P + nt - 1) / nt;
    const int cap = nt == 1024 ? BIN_MAX_WG : BIN_LEAN_MAX_WG;
    return blocks < 1 ? 1 : (blocks > cap ? cap : blocks);
}

template <int NT = 1024>
__device__ __forceinline__ uint32_t workgroup_inclusive_scan(uint32_t v, int tid, uint32_t* s_wsum, uint32_t& total)
{
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t incl = wave_inclusive_scan(v, lane);
    __syncthreads();  // s_wsum may still be read by the previous caller
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; w++) {
        const uint32_t c = s_wsum[w];
        woff += w < wave ? c : 0u;
        tot += c;
    }
    total = tot;
    return incl + woff;
}

template <bool EMIT, int NT = 1024>
__global__ void __launch_bounds__(NT) bin_spans_kernel(int P, const BlendRec* __restrict__ rank_rec,
                                                                uint32_t* __restrict__ partial,
                                                                const uint2* __restrict__ ranges,
                                                                uint32_t* __restrict__ entries, uint32_t gx, uint32_t gy_all,
                                                                uint32_t by0, uint32_t by1, int ablate)
{
    // COUNT: s_dyn = difference grid [band rows][stride] (ints), then the hand-off arrays
    // EMIT : s_dyn = cursors [band tiles], then the hand-off arrays
    extern __shared__ uint32_t s_dyn[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t gy = by1 - by0;
    const int stride = count_grid_stride(gx);
    const int ntiles = (int)(gx * gy);
    const int tile0 = (int)(gx * by0);
    const int ntiles_all = (int)(gx * gy_all);
    const int head = EMIT ? ((ntiles + 3) & ~3) : (((int)gy * stride + 3) & ~3);
    uint32_t* s_cnt = s_dyn;
    int* s_grid = reinterpret_cast<int*>(s_dyn);
    uint32_t* s_rw = s_dyn + head;
    constexpr int NW = NT / 64;              // waves per workgroup
    static_assert(NT == 1024 || NT == 512, "rank slices are dealt in 64-rank chunks to 16 or 8 waves");
    RectWork rw{s_rw, s_rw + NT, s_rw + 2 * NT, s_rw + 3 * NT};
    float2* s_xy = reinterpret_cast<float2*>(s_rw + 3 * NT + 16);
    float4* s_co = reinterpret_cast<float4*>(s_rw + 3 * NT + 16 + 2 * NT);
    uint32_t* s_gpre = s_rw + 3 * NT + 16 + 6 * NT;   // inclusive prefix of the Gaussians' row counts
    uint32_t* s_grect = s_gpre + NT;         // clip columns x0 | x1 << 10, first row << 21
    uint32_t* s_grad = s_grect + NT;         // radius (the margin of tau needs it)
    uint32_t* s_q0 = s_grad + NT;            // EMIT, per span: upper band's columns lo | hi << 11, Gaussian slot << 22
    uint32_t* s_q1 = s_q0 + NT;              //                 lower band's columns lo | hi << 11
    uint32_t* my_partial = partial + (size_t)slice_row(blockIdx.x, gridDim.x) * ntiles_all + tile0;
    if (EMIT) {
        for (int t = tid; t < ntiles; t += NT) s_cnt[t] = ranges[tile0 + t].x + my_partial[t];
    } else {
        for (int c = tid; c < (int)gy * stride; c += NT) s_grid[c] = 0;
    }
    __syncthreads();
    const int nwg = (int)gridDim.x;
    const int rounds = ((P + 63) / 64 + NW * nwg - 1) / (NW * nwg);   // 64-rank chunks dealt round robin over all waves of all workgroups
    // The record of the NEXT round is requested before this round's items are walked (a round is a chain of workgroup barriers
    // and LDS searches with one workgroup per CU: nothing else would hide the load).  Unconditional, index clamped: a
    // conditionally assigned load result is waited for on the spot.
    BlendRec nxt = rank_rec[min((wave * nwg + (int)blockIdx.x) * 64 + lane, P - 1)];
    for (int it = 0; it < rounds; it++) {
        const int r = ((it * NW + wave) * nwg + (int)blockIdx.x) * 64 + lane;  // same dealing as bin_ranks_kernel
        const BlendRec rec = nxt;
        nxt = rank_rec[min((((it + 1) * NW + wave) * nwg + (int)blockIdx.x) * 64 + lane, P - 1)];
        uint32_t h = 0;
        if (r < P) {
            const int rad = (int)rec.pm;
            if (rad > 0) {
                uint2 rmin, rmax;
                getRect(rec.xy.x, rec.xy.y, rad, rmin, rmax, gx, gy_all);
                shrink_rect(rec.xy, rec.co, rad, rmin, rmax);
                rmin.y = max(rmin.y, by0);
                rmax.y = min(rmax.y, by1);
                if (rmax.x > rmin.x && rmax.y > rmin.y) {
                    h = rmax.y - rmin.y;
                    s_xy[tid] = rec.xy;
                    s_co[tid] = rec.co;
                    s_grect[tid] = rmin.x | (rmax.x << 10) | (rmin.y << 21);
                    s_grad[tid] = (uint32_t)rad;
                }
            }
        }
        uint32_t rows_total;
        s_gpre[tid] = workgroup_inclusive_scan<NT>(h, tid, rw.wsum, rows_total);
        __syncthreads();
        if MI_ABLATE(1 << 20) rows_total = 0;
        for (uint32_t w0 = 0; w0 < rows_total; w0 += NT) {  // windows of NT (Gaussian, tile row) items
            const uint32_t k = w0 + (uint32_t)tid;
            uint2 smin = make_uint2(0, 0), smax = make_uint2(0, 0);
            uint32_t width = 0;
            if (k < rows_total) {
                int g = 0;  // first Gaussian slot with prefix > k
#pragma unroll
                for (int step = NT / 2; step >= 1; step >>= 1)
                    if (s_gpre[g + step - 1] <= k) g += step;
                const uint32_t prev = g == 0 ? 0u : s_gpre[g - 1];
                const uint32_t packed = s_grect[g];
                const uint32_t cx0 = packed & 1023u, cx1 = (packed >> 10) & 2047u, ty = (packed >> 21) + (k - prev);
                const float2 xy = s_xy[g];
                const SpanPre pre = span_prepare(s_co[g], (int)s_grad[g]);
                int lo0, hi0, lo1, hi1;
                band_columns(pre, xy, (float)(ty * TILE_Y), (int)(2u * cx0), (int)(2u * cx1), lo0, hi0);
                band_columns(pre, xy, (float)(ty * TILE_Y + 8u), (int)(2u * cx0), (int)(2u * cx1), lo1, hi1);
                if (hi0 > lo0 || hi1 > lo1) {
                    const int lo = hi0 > lo0 ? (hi1 > lo1 ? min(lo0, lo1) : lo0) : lo1;
                    const int hi = hi0 > lo0 ? (hi1 > lo1 ? max(hi0, hi1) : hi0) : hi1;
                    smin = make_uint2((uint32_t)lo >> 1, ty);
                    smax = make_uint2(((uint32_t)hi + 1u) >> 1, ty + 1u);
                    width = smax.x - smin.x;
                    if (EMIT) {
                        s_q0[tid] = (uint32_t)lo0 | ((uint32_t)hi0 << 11) | ((uint32_t)g << 22);
                        s_q1[tid] = (uint32_t)lo1 | ((uint32_t)hi1 << 11);
                    } else {
                        // EXACT counts: a tile is counted iff one of the two intervals has a column in it, i.e. iff the emit
                        // pass finds a non-zero mask there -- the two bands' tile ranges separately when a gap lies between them
                        int* row = s_grid + (ty - by0) * stride;
                        const int a0 = lo0 >> 1, b0 = (hi0 + 1) >> 1, a1 = lo1 >> 1, b1 = (hi1 + 1) >> 1;
                        if (hi0 > lo0 && hi1 > lo1 && (b0 < a1 || b1 < a0)) {
                            atomicAdd(&row[a0], 1);
                            atomicAdd(&row[b0], -1);
                            atomicAdd(&row[a1], 1);
                            atomicAdd(&row[b1], -1);
                        } else {
                            atomicAdd(&row[smin.x], 1);
                            atomicAdd(&row[smax.x], -1);
                        }
                    }
                }
            }
            if (EMIT && !MI_ABLATE(1 << 16)) {
                for_each_tile_balanced<NT>(
                    rw, tid, smin, smax, width,
                    [&](uint32_t owner, uint32_t tx, uint32_t ty) {
                        if MI_ABLATE(1 << 17) return;
                        const uint32_t q0 = s_q0[owner], q1 = s_q1[owner];
                        const uint32_t lo0 = q0 & 2047u, n0 = ((q0 >> 11) & 2047u) - lo0, lo1 = q1 & 2047u, n1 = ((q1 >> 11) & 2047u) - lo1;
                        const uint32_t c = 2u * tx;
                        const uint32_t qmask = (uint32_t)(c - lo0 < n0) | ((uint32_t)(c + 1u - lo0 < n0) << 1) |
                                               ((uint32_t)(c - lo1 < n1) << 2) | ((uint32_t)(c + 1u - lo1 < n1) << 3);
                        if (qmask != 0u) {
                            const uint32_t slot_g = q0 >> 22;
                            const uint32_t rank = (uint32_t)(((it * NW + (int)(slot_g >> 6)) * nwg + (int)blockIdx.x) * 64 + (int)(slot_g & 63u));
                            if MI_ABLATE(1 << 18) return;
                            const uint32_t slot = atomicAdd(&s_cnt[(ty - by0) * gx + tx], 1u);
                            if MI_ABLATE(1 << 19) return;
                            entries[slot] = rank | (qmask << RANK_BITS);
                        }
                    });
            }
        }
        __syncthreads();  // the Gaussian arrays are rewritten by the next round
    }
    if (!EMIT) {
        __syncthreads();
        // prefix along x: one wave per row, 64 cells at a time with a carry -> the per-tile counts of this slice
        for (int y = wave; y < (int)gy; y += NW) {
            int carry = 0;
            for (int x0 = 0; x0 < (int)gx; x0 += 64) {
                const int x = x0 + lane;
                int v = x < (int)gx ? s_grid[y * stride + x] : 0;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int t = __shfl_up(v, o, 64);
                    if (lane >= o) v += t;
                }
                v += carry;
                if (x < (int)gx) my_partial[y * (int)gx + x] = (uint32_t)v;
                carry = __shfl(v, 63, 64);
            }
        }
    }
}

// Per tile: exclusive prefix of partial[slice][tile] over the slices (in place) and tile_total[tile].
// One workgroup handles 64 tiles x 16 groups of slices; every load is coalesced along the tile axis.
__global__ void __launch_bounds__(1024) scan_partials_kernel(int ntiles, int nwg, uint32_t* __restrict__ partial,
                                                             uint32_t* __restrict__ tile_total)
{
    __shared__ uint32_t s_sum[16][64];
    const int tl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + tl;
    const int wpg = (nwg + 15) / 16;
    const int w0 = g * wpg, w1 = min(nwg, w0 + wpg);
    uint32_t sum = 0;
    if (t < ntiles)
        for (int w = w0; w < w1; w++) sum += partial[(size_t)w * ntiles + t];
    s_sum[g][tl] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (int gg = 0; gg < g; gg++) run += s_sum[gg][tl];
    if (t < ntiles) {
        for (int w = w0; w < w1; w++) {
            const uint32_t c = partial[(size_t)w * ntiles + t];
            partial[(size_t)w * ntiles + t] = run;
            run += c;
        }
        if (g == 15) tile_total[t] = run;
    }
}

// ---- 5. per-tile LDS radix sort of the ranks --------------------------------------------------------
// One workgroup (256 threads for short lists, 1024 for long ones) per tile; handles tiles with LO < n <= CAP in LDS; when GLOBAL_FALLBACK is set it
// also handles n > CAP by ping-ponging between `entries` and `scratch` in HBM with the same code.
// LSD radix, 8-bit digits, `passes` = ceil(rank_bits / 8).  Each wave owns a contiguous quarter of the tile's
// list; per pass: per-wave digit histograms -> workgroup scan over (digit, wave) -> each wave scatters its quarter
// 64 keys at a time with a stable ballot-match rank.  Three workgroup barriers per pass.
// MAXB > 0 (lists held in LDS: a wave's share is at most MAXB batches of 64 keys): the keys and their match results
// (rank among the wave's equal digits, size of that group) stay in registers between the histogram and the scatter
// phase -- the eight-ballot match is evaluated once per key and pass instead of twice.  MAXB = 0: any length (the
// HBM fallback), everything re-read and re-matched in the scatter phase.
template <int NW, int MAXB, typename SrcPtr, typename DstPtr>
__device__ __forceinline__ void radix_pass(SrcPtr src, DstPtr dst, int n, int shift, uint32_t (*s_hist)[256], int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    // contiguous share per wave (NW waves), in multiples of 64 keys
    const int chunks = (n + 63) >> 6;
    const int cpw = (chunks + NW - 1) / NW;
    const int begin = min(n, wave * cpw * 64), end = min(n, (wave + 1) * cpw * 64);
    for (int d = tid; d < NW * 256; d += NW * 64) (&s_hist[0][0])[d] = 0;
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

    constexpr int NB = MAXB > 0 ? MAXB : 1;
    uint32_t c_key[NB], c_rk[NB], c_cnt[NB];
    // (a) per-wave histogram of this digit
    if (MAXB > 0) {
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const int i0 = begin + 64 * j;
            if (i0 >= end) break;  // wave-uniform
            const int i = i0 + lane;
            const bool active = i < end;
            c_key[j] = active ? (uint32_t)src[i] : 0u;
            const uint32_t digit = ((c_key[j] & RANK_MASK) >> shift) & 0xFFu;
            match(digit, active, c_rk[j], c_cnt[j]);
            if (active && c_rk[j] == 0) s_hist[wave][digit] += c_cnt[j];  // one lane per distinct digit, wave-private row
        }
    } else {
        for (int i0 = begin; i0 < end; i0 += 64) {
            const int i = i0 + lane;
            const bool active = i < end;
            const uint32_t key = active ? (uint32_t)src[i] : 0u;
            const uint32_t digit = ((key & RANK_MASK) >> shift) & 0xFFu;
            uint32_t rk, cnt;
            match(digit, active, rk, cnt);
            if (active && rk == 0) s_hist[wave][digit] += cnt;
        }
    }
    __syncthreads();
    // (b) exclusive scan over (digit major, wave minor): thread d < 256 owns digit d
    {
        __shared__ uint32_t s_wsum[4];
        uint32_t tot = 0, incl = 0;
        if (tid < 256) {
#pragma unroll
            for (int w = 0; w < NW; w++) tot += s_hist[w][tid];
            // workgroup exclusive scan of tot over the 256 digit threads
            incl = wave_inclusive_scan(tot, lane);
            if (lane == 63) s_wsum[wave] = incl;
        }
        __syncthreads();
        if (tid < 256) {
            uint32_t woff = 0;
            for (int w = 0; w < wave; w++) woff += s_wsum[w];
            uint32_t run = woff + incl - tot;
#pragma unroll
            for (int w = 0; w < NW; w++) {
                const uint32_t c = s_hist[w][tid];
                s_hist[w][tid] = run;
                run += c;
            }
        }
    }
    __syncthreads();
    // (c) stable scatter of this wave's share; s_hist[wave][digit] is now this wave's running cursor
    auto scatter = [&](uint32_t key, bool active, uint32_t rk, uint32_t cnt) {
        const uint32_t digit = ((key & RANK_MASK) >> shift) & 0xFFu;
        uint32_t off = 0;
        if (active) off = s_hist[wave][digit] + rk;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (active && rk == cnt - 1) s_hist[wave][digit] = off + 1;  // last lane of the group advances the cursor
        if (active) dst[off] = key;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    };
    if (MAXB > 0) {
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const int i0 = begin + 64 * j;
            if (i0 >= end) break;
            scatter(c_key[j], i0 + lane < end, c_rk[j], c_cnt[j]);
        }
    } else {
        for (int i0 = begin; i0 < end; i0 += 64) {
            const int i = i0 + lane;
            const bool active = i < end;
            const uint32_t key = active ? (uint32_t)src[i] : 0u;
            uint32_t rk, cnt;
            match(((key & RANK_MASK) >> shift) & 0xFFu, active, rk, cnt);
            scatter(key, active, rk, cnt);
        }
    }
    __syncthreads();
}

// The tile's BLEND LIST: the sorted entries with the depth rank replaced by the Gaussian id, id | quadrant mask << 28, four bytes
// per overlap at blend_list[range.x ...].  This is all the blend kernels get of a tile: they scan the entries 64 at a time (256
// contiguous bytes), queue the ones whose bit for their quadrant is set and gather the 32-byte geometry record index_rec[id] next to
// the feature row of the same id -- two independent gathers behind one scan.  (Rounds 1-4 materialised a 32-byte blend record per
// entry here: a 94-MB gather and a 94-MB store per cfg3 view for lists of which the blends walk a third.)  The position of an entry
// in its list is its index -- n_contrib's unit.  Full lists (bin_ranks_kernel): every overlap of the reference's rects is an entry,
// culled ones with mask 0, so blend_list & RANK_MASK IS the reference's point_list and the index its position there; lean lists
// (bin_spans_kernel): the entries that can blend only.
template <int NW, typename SrcPtr>
__device__ __forceinline__ void emit_blend_list(SrcPtr sorted_entries, int n, uint2 range, int tid,
                                                const uint32_t* __restrict__ sorted_idx, uint32_t* __restrict__ blend_list)
{
    uint32_t* out = blend_list + range.x;
    for (int i = tid; i < n; i += NW * 64) {
        const uint32_t e = sorted_entries[i];
        out[i] = sorted_idx[e & RANK_MASK] | (e & ~RANK_MASK);
    }
}

// MI_RAST_VERIFY_LISTS (include/mi_rast.h): the lean lists rest on the count pass and the emit pass taking bit-identical float
// decisions in two template instances of bin_spans_kernel -- a tile is counted iff the emit pass stores an entry for it.  With
// the flag the entries are zero-filled before the emit pass and this kernel counts the slots of every tile's segment that
// are still zero afterwards (a lean entry always carries a non-zero quadrant mask): any such slot is a decision that differed.
__global__ void __launch_bounds__(256) verify_entries_kernel(uint32_t ntiles, const uint2* __restrict__ ranges,
                                                              const uint32_t* __restrict__ entries, uint32_t* __restrict__ unwritten)
{
    const uint32_t tile = blockIdx.x;
    if (tile >= ntiles) return;
    const uint2 r = ranges[tile];
    uint32_t n = 0;
    for (uint32_t i = r.x + threadIdx.x; i < r.y; i += 256) n += entries[i] == 0u;
    if (n) atomicAdd(unwritten, n);
}

template <int LO, int CAP, bool GLOBAL_FALLBACK, int NT>
__global__ void __launch_bounds__(NT) tile_sort_kernel(uint32_t ntiles, const uint2* __restrict__ ranges,
                                                        uint32_t* __restrict__ entries,
                                                        uint32_t* __restrict__ scratch,
                                                        const uint32_t* __restrict__ sorted_idx, int passes,
                                                        uint32_t* __restrict__ blend_list)
{
    __shared__ uint32_t s_a[CAP];
    __shared__ uint32_t s_b[CAP];
    constexpr int NW = NT / 64;
    __shared__ uint32_t s_hist[NW][256];
    const int tid = threadIdx.x;
    // (tile = workgroup id: neighbouring tiles on different XCDs.  Contiguous runs per XCD -- common.h -- save 3 % here through
    // the rank records neighbouring tiles share, but the cost of a tile is its list length, and a static split would let a
    // scene's dense half wait for two of the eight XCDs.)
    const uint32_t tile = blockIdx.x;
    if (tile >= ntiles) return;
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    if (n <= LO) return;
    if (n > CAP && !GLOBAL_FALLBACK) return;
    uint32_t* seg = entries + range.x;
    if (n <= CAP) {
        // (lean lists too: their counts are exact -- bin_spans_kernel --, every slot of the segment holds an entry)
        for (int i = tid; i < n; i += NT) s_a[i] = seg[i];
        __syncthreads();
        uint32_t* a = s_a;
        uint32_t* b = s_b;
        for (int p = 0; p < passes; p++) {
            radix_pass<NW, (CAP + NW * 64 - 1) / (NW * 64)>(a, b, n, 8 * p, s_hist, tid);
            uint32_t* t = a;
            a = b;
            b = t;
        }
        emit_blend_list<NW>(a, n, range, tid, sorted_idx, blend_list);
    } else {
        uint32_t* a = seg;
        uint32_t* b = scratch + range.x;
        for (int p = 0; p < passes; p++) {
            radix_pass<NW, 0>(a, b, n, 8 * p, s_hist, tid);
            uint32_t* t = a;
            a = b;
            b = t;
        }
        emit_blend_list<NW>(a, n, range, tid, sorted_idx, blend_list);
    }
}

}  // namespace mirast
