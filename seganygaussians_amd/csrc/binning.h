// binning.h -- tile binning for gfx950.
//
// What the reference does (CF/cuda_rasterizer/rasterizer_impl.cu:70-138,277-317): prefix-sum the
// per-Gaussian tile counts, emit one 64-bit key (tile<<32 | depth_bits) + 32-bit value per overlap,
// run a 45-bit device-wide radix sort over all R pairs (6 passes x 24 B/pair of HBM traffic), then
// find per-tile ranges.  The RESULT -- point_list ordered by (tile, depth bits, Gaussian index) and
// ranges[tile] -- is part of the bit-exact integer contract; the way to get there is not.
//
// MI355X design (produces the identical point_list / ranges), five launches behind the preprocess pass:
//   1. the per-Gaussian preprocess kernel sums tiles_touched into R (what the host needs to size the binning buffer) and
//      leaves one 32-byte record {mean, id, radius, conic, opacity} and the depth bits of every visible Gaussian;
//   2. a count pass over slices of the Gaussians IN INDEX ORDER (per-tile counters in LDS), a scan over (tile, slice),
//      and the scan of the tile totals give ranges[tile] and every slice's base inside every tile -- no global atomics;
//   3. emission: each (Gaussian, tile) overlap takes its slot from an LDS cursor and stores the 8-byte entry
//      {depth bits, Gaussian id | quadrant mask << 28} (arrival order inside a tile is arbitrary);
//   4. per-tile LDS radix sort of the entries by their depth bits (only the bits in which the view's keys differ:
//      3-4 passes of 8 bits), entries of EQUAL depth ordered by id behind it, then blend_list[slot] = id | mask << 28
//      (full lists: the low 28 bits are point_list).
// Rounds 1-5 ordered the P Gaussians by depth first (six launches, 0.08 ms on a 1 M-Gaussian view) so that a tile's
// entries were 20-bit ranks; round 6 drops that stage: the per-tile sort has the depth bits themselves as its key, one
// radix pass more for six launches and 120 MB of traffic less.  The count / emit passes are wave-granular: a wave owns
// 64 consecutive Gaussians, walks their (Gaussian, tile row) items and the tiles of their spans with wave-synchronous LDS
// hand-offs, and meets the other waves of its workgroup only in the shared per-tile counters (LDS atomics) -- no workgroup
// barrier inside the loop (a barrier phase of a 1024-thread workgroup costs about 1 us on this part, and rounds 2-5 paid
// ~50 of them per workgroup).
// HBM traffic per overlap drops from ~172 B to ~24 B (8 B emit write, 8 B sort read, 4 B list write, 4 B blend scan);
// the 64-bit keys are never materialised.
#pragma once

#include <type_traits>

#include "common.h"
#include "cull.h"

namespace mirast {

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// The same scan on the VALU alone (seven DPP moves: row_shr 1 / 2 / 3, row_shr 4 and 8 under bank masks, row_bcast 15 and 31
// under row masks -- the sequence of AMD's GCN3 cross-lane note) instead of six trips through the LDS crossbar (__shfl_up is
// ds_bpermute): for the count / emit passes, whose waves run two scans per window.  MAX: running maximum instead of sum.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, false);   // lanes without a source / masked off: 0
}
template <bool MAX = false>
__device__ __forceinline__ uint32_t wave_inclusive_scan_dpp(uint32_t v0)
{
    auto op = [](uint32_t a, uint32_t b) { return MAX ? max(a, b) : a + b; };
    uint32_t v = op(v0, dpp_or_zero<0x111, 0xF, 0xF>(v0));   // row_shr:1
    v = op(v, dpp_or_zero<0x112, 0xF, 0xF>(v0));             // row_shr:2
    v = op(v, dpp_or_zero<0x113, 0xF, 0xF>(v0));             // row_shr:3  -> windows of four
    v = op(v, dpp_or_zero<0x114, 0xF, 0xE>(v));              // row_shr:4, lanes 4 .. 15 of a row -> windows of eight
    v = op(v, dpp_or_zero<0x118, 0xF, 0xC>(v));              // row_shr:8, lanes 8 .. 15 -> the row's prefix
    v = op(v, dpp_or_zero<0x142, 0xA, 0xF>(v));              // row_bcast:15 into rows 1 and 3
    v = op(v, dpp_or_zero<0x143, 0xC, 0xF>(v));              // row_bcast:31 into rows 2 and 3
    return v;
}

// Row of partial[][] (= position of the slice inside every tile's segment) of workgroup b.  Workgroup b runs on XCD b % 8
// (tools/xcc_probe.hip) and each XCD has its own L2: with the slices of one XCD next to each other, the entries that
// share a 128-byte line of a tile's segment are mostly stored from ONE XCD instead of eight (measured: emit 0.086 -> 0.074 ms;
// the HBM write bytes of the pass did not change).  Any bijection is correct -- count, scan and emit only have to agree on it.
__device__ __forceinline__ uint32_t slice_row(uint32_t b, uint32_t nwg)
{
    return (nwg & 7u) ? b : (b & 7u) * (nwg >> 3) + (b >> 3);
}

// ---- workgroup-balanced enumeration of tile rects (full lists only: parity tests, the reference's `debug` flag) ----------
// Each of the 1024 threads owns one Gaussian with `count` tiles (0 if culled) in rect [rmin, rmax).  Tile counts
// are extremely skewed (BASELINE cfg3: median 6 tiles, 99.9th percentile 484, maximum 4056), so the WORKGROUP walks the
// concatenation of all 1024 rects with every thread busy: item k belongs to the thread whose inclusive prefix first
// exceeds k (binary search in an LDS copy of the prefix), and its tile follows from k's offset inside that rect.
struct RectWork {
    uint32_t* prefix;  // LDS [NT] inclusive prefix of counts
    uint32_t* rx;      // LDS [NT] rect_min.x | width << 16
    uint32_t* ry;      // LDS [NT] rect_min.y
    uint32_t* wsum;    // LDS [16]
};

// Contains workgroup barriers: call from all NT threads.  f(owner_thread, tile_x, tile_y) handles one item.
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

// ---- tile ranges: exclusive scan of the per-tile totals (single workgroup) ---------------------------------
// Writes ranges[tile] = [base, base+count) (== identifyTileRanges' result, rasterizer_impl.cu:116-138, including
// {0,0} for empty tiles as left by the reference's cudaMemset), R and the longest list.
constexpr int ID_BITS = 28;            // list entry = Gaussian id | quadrant mask << 28
constexpr uint32_t ID_MASK = (1u << ID_BITS) - 1u;
constexpr int BIN_MAX_WG = 256;        // workgroups of the count / emit passes (slices of the Gaussians): one per CU, all resident at once
constexpr int BIN_THREADS = 1024;
constexpr int BIN_LEAN_WG = 256;       // workgroups of the LEAN count / emit passes (dealt to the bands by load, binning.h: band_plan)
constexpr int BIN_LEAN_WG_MAX = 1024;  // (upper bound of the profiling build's MI_RAST_LEAN_NWG knob)
constexpr int BIN_MAX_TILES = 22 * 1024 - 64;  // per launch of the count / emit passes: one LDS counter per tile + 61 KB of hand-off
                                               // arrays must fit in 160 KB; larger images are walked in bands of tile rows
constexpr int BIN_MAX_TILES_TOTAL = 40 * 1024 - 128;  // tile_ranges_kernel scans all tile totals in one workgroup's LDS

// ---- bands of tile rows (lean count / emit passes) ----------------------------------------------------------------------------
// The lean passes walk the image in <= MAX_BANDS bands of band_h tile rows.  A workgroup belongs to ONE band: its LDS counters / cursors
// cover that band's tiles only, and -- the point -- the entries it stores go to a few hundred tile segments instead of all of them:
// ~22 consecutive entries per (workgroup, tile) on cfg3 instead of 1.4, which the L2 of the workgroup's XCD merges into whole lines
// (tools/write_combine_probe.hip: the same 2.93 M eight-byte stores take 16.5 us banded, 30 us with every workgroup storing into every
// tile's segment; WRITE_SIZE of rounds 2-5 was 64 bytes per entry).  The preprocess pass leaves, per band, one bit per Gaussian (band_bits[band][index chunk]:
// a wave's ballot) and the number of Gaussians per band (r_slots); a band gets workgroups in proportion to that number (band_plan), every workgroup an
// equal share of the Gaussian INDEX range, of which it picks the Gaussians with its band's bit (64 Gaussians per word).
__host__ __device__ inline bool band_geometry(uint32_t gx, uint32_t gy, uint32_t max_head_words, uint32_t& band_h, uint32_t& nbands)
{
    band_h = (gy + 15u) / 16u;
    if (band_h < 1u) band_h = 1u;
    while (band_h > 1u && (band_h + 1u) * (gx + 2u) > max_head_words) band_h--;   // (+ 1: the count pass's difference grid has an odd stride >= gx + 1)
    nbands = (gy + band_h - 1u) / band_h;
    return nbands <= (uint32_t)MAX_BANDS && (band_h + 1u) * (gx + 2u) <= max_head_words;
}

// s_plan[b] = first workgroup of band b, s_plan[nbands] = workgroups in use (<= nwg): one workgroup per band plus the rest in
// proportion to the bands' Gaussian counts.  Call from one whole wave; every kernel that needs the plan recomputes it from the same
// counters with this same code.
__device__ __forceinline__ void band_plan(const int* __restrict__ r_slots, uint32_t nbands, uint32_t nwg, uint32_t* s_plan, int lane)
{
    uint32_t cnt = 0;
    if ((uint32_t)lane < nbands)
        for (int k = 0; k < R_SLOTS; k++) cnt += (uint32_t)r_slots[k * R_SLOT_STRIDE + R_SLOT_BANDS + lane];
    uint32_t total = cnt;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) total += (uint32_t)__shfl_xor((int)total, o, 64);
    const uint32_t avail = nwg > nbands ? nwg - nbands : 0u;
    uint32_t n = (uint32_t)lane < nbands ? 1u + (total ? (uint32_t)(((uint64_t)cnt * avail) / total) : 0u) : 0u;
    if (nwg < nbands) n = (uint32_t)lane < nwg ? 1u : 0u;   // (never launched so: the host gives at least one workgroup per band)
    uint32_t incl = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
        if (lane >= o) incl += t;
    }
    if ((uint32_t)lane < nbands) s_plan[lane] = incl - n;
    if ((uint32_t)lane == nbands) s_plan[nbands] = incl;   // (lanes >= nbands hold the grand total: n = 0 there)
}

// Words behind the R partial sums of the image buffer's num_rendered field (mi_rast.hip: ImgPtrs)
constexpr int NR_TOTAL = 0, NR_LONGEST = 1, NR_VERIFY = 2, NR_KEY_BITS = 3, NR_RUN_BOUNDS = 4;

__global__ void __launch_bounds__(1024) tile_ranges_kernel(int ntiles_all, const uint32_t* __restrict__ tile_total,
                                                           uint2* __restrict__ ranges, int* __restrict__ num_rendered,
                                                           const int* __restrict__ r_slots, int* __restrict__ host_out = nullptr,
                                                           uint32_t* __restrict__ zero_a = nullptr, uint32_t* __restrict__ zero_b = nullptr,
                                                           uint32_t* __restrict__ run_bounds = nullptr /* [9]: the blend kernels' XCD runs
                                                               (common.h): equal tile counts, or equal MODELLED work when run_cap > 0 */,
                                                           uint32_t run_cap = 0, uint32_t run_fix = 0)
{
    // The totals are staged in LDS (coalesced), thread t scans the contiguous items [t*per, (t+1)*per) in place,
    // one workgroup scan joins the pieces, and the ranges leave coalesced again.
    // More than BIN_MAX_TILES_TOTAL items (images beyond 10 Mpx) are walked in segments of that many, one after the other, the
    // running total carried along: ONE pass of the loop below for every image up to 4096 x 2544.
    // run_cap > 0 (images of one segment): the XCD runs of the blend kernels are cut at equal sums of the
    // MODELLED cost of a tile, min(list length, run_cap) + run_fix -- a list is walked until its pixels are opaque, which takes about
    // run_cap entries where the scene is dense, and to its end where it is sparse; what a walk really covers is only known behind the
    // forward blend (run_bounds_from_walks_kernel) -- the second prefix sum rides on the first, at the granularity of the threads'
    // pieces (registers only); the thread whose piece holds a boundary walks its few tiles.
    // Also left for the per-tile sort: the number of low key bits in which the view's depth keys differ at all (NR_KEY_BITS) --
    // min and max key share every bit above, and so does every key between them.
    extern __shared__ uint32_t s_val[];  // [min(ntiles_all, BIN_MAX_TILES_TOTAL) + 1]
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_wave_w[16];
    __shared__ uint32_t s_bound[9];
    const bool weighted = run_bounds != nullptr && run_cap > 0u && ntiles_all <= BIN_MAX_TILES_TOTAL;   // (one segment)
    __shared__ uint32_t s_maxcount;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 9) s_bound[tid] = xcd_run_start((uint32_t)tid, (uint32_t)ntiles_all);   // equal tile counts unless the model says otherwise
    if (tid == 0) s_maxcount = 0;
    if (wave == 15 && r_slots != nullptr) {   // (a wave that has the least to do below)
        uint32_t inv_min = (uint32_t)r_slots[(lane % R_SLOTS) * R_SLOT_STRIDE + 1];
        uint32_t mx = (uint32_t)r_slots[(lane % R_SLOTS) * R_SLOT_STRIDE + 2];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            inv_min = max(inv_min, (uint32_t)__shfl_xor((int)inv_min, o, 64));
            mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
        }
        const uint32_t kmin = ~inv_min;   // no visible Gaussian: kmin = 0xFFFFFFFF > mx = 0
        const uint32_t diff = mx >= kmin ? (mx ^ kmin) : 0u;
        if (lane == 0) num_rendered[NR_KEY_BITS] = diff ? 32 - __builtin_clz(diff) : 0;
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
                if (host_out != nullptr) host_out[NR_RUN_BOUNDS + k] = (int)bb[k];   // the forward's grid is sized for its longest run
            }
        }
    }
    if (tid == 0) {
        num_rendered[NR_TOTAL] = (int)carry;         // R (the host already has it from the preprocess pass; kept for checks)
        num_rendered[NR_LONGEST] = (int)s_maxcount;  // longest list
        if (host_out != nullptr) {                   // the same two words straight into the host's pinned buffer (no copy command)
            host_out[NR_TOTAL] = (int)carry;
            host_out[NR_LONGEST] = (int)s_maxcount;
        }
    }
}

// ---- content fingerprints (mi_rast_fingerprint, include/mi_rast.h) ---------------------------------------------------------
constexpr int MI_FP_MAX = 8;      // arrays per call
constexpr int FP_BLOCKS = 128;    // partial sums per array (summed by the host)
struct FingerprintArgs {
    const uint32_t* ptr[MI_FP_MAX];
    unsigned long long words[MI_FP_MAX];
};
// blockIdx.y = array; every word is hashed together with its position (murmur3's finaliser) and the hashes are summed: order of
// summation does not matter, order of the words does.
__global__ void __launch_bounds__(256) fingerprint_kernel(FingerprintArgs a, unsigned long long* __restrict__ out)
{
    const int k = blockIdx.y;
    const uint32_t* p = a.ptr[k];
    const unsigned long long n = a.words[k];
    unsigned long long sum = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; i < n; i += (unsigned long long)FP_BLOCKS * 256ull) {
        uint32_t x = p[i] ^ ((uint32_t)i * 0x9E3779B1u + (uint32_t)(i >> 32));
        x ^= x >> 16;
        x *= 0x85EBCA6Bu;
        x ^= x >> 13;
        x *= 0xC2B2AE35u;
        x ^= x >> 16;
        sum += (unsigned long long)x * 0x9E3779B97F4A7C15ull + i;
    }
    __shared__ unsigned long long s_part[4];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) out[k * FP_BLOCKS + blockIdx.x] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

// mi_rast_forward_reuse (include/mi_rast.h): a fresh image buffer takes over what the binning stages of an earlier forward of the same
// geometry and camera left in theirs -- ranges, {R, longest list, key bits}, the XCD run boundaries -- and gets the blend kernels'
// per-tile walk counters zeroed, as tile_ranges_kernel leaves them.
__global__ void __launch_bounds__(256) reuse_image_state_kernel(uint32_t ntiles, const uint2* __restrict__ src_ranges,
                                                                const int* __restrict__ src_words, uint2* __restrict__ ranges,
                                                                int* __restrict__ words, uint32_t* __restrict__ zero_a,
                                                                uint32_t* __restrict__ zero_b)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t < ntiles) {
        ranges[t] = src_ranges[t];
        zero_a[t] = 0u;
        zero_b[t] = 0u;
    }
    if (t < 16u) words[t] = src_words[t];   // {R, longest list, -, key bits, run boundaries}: the 16 words behind the R partial sums
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
// with dependent loads, forward.cu:318-326): written per VISIBLE Gaussian by the preprocess pass (index_rec, `pm` = radius; the
// slots of culled Gaussians are not written -- depth_key says which), read in index order by the count / emit passes, and what the
// blend kernels gather per list entry.
struct __attribute__((aligned(16))) BlendRec {
    float2 xy;       // pixel-space mean
    uint32_t id;     // index_rec: the DEPTH BITS of the Gaussian (its index is where the record lies; the emit pass gets the sort key with the
                     // record instead of a second 4-byte gather); a staged list record (list_record): the Gaussian index (feature row)
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
    BlendRec r = index_rec[e & ID_MASK];
    r.id = e & ID_MASK;   // (the stored word holds the depth bits)
    r.pm = ((uint32_t)i << 4) | (e >> ID_BITS);
    return r;
}

// ---- counting and emission ------------------------------------------------------------------------------
// Global atomics are the scarce resource of the binning stages (about 25 G scattered dword atomics/s on this part:
// one corner atomic per Gaussian and one cursor atomic per overlap were 0.11 + 0.27 ms per view).  So the bucketing
// is done the way a radix-sort pass does it.  The Gaussians are dealt to <= 256 workgroups of 1024 threads in
// chunks of 64 consecutive indices, round robin; the set of chunks a workgroup owns is its "slice":
//   count pass  (EMIT = false): per-tile counters in LDS, one LDS atomic per overlap; the counters go to
//                partial[slice][tile] with plain stores;
//   scan        (scan_partials_kernel): per tile, the exclusive prefix over slices (in place) and the tile total;
//   emit pass   (EMIT = true): LDS cursors start at ranges[tile].x + partial[slice][tile]; every overlap takes
//                its slot with a returning LDS atomic and stores its 8-byte entry {depth bits, id | mask << 28}.
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
__global__ void __launch_bounds__(BIN_THREADS) bin_ranks_kernel(int P, const BlendRec* __restrict__ index_rec,
                                                                const uint32_t* __restrict__ depth_key,
                                                                const uint32_t* __restrict__ partial,
                                                                const uint2* __restrict__ ranges,
                                                                uint2* __restrict__ entries, uint32_t gx, uint32_t gy,
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
    uint32_t* s_rad = s_rw + 3088 + 6144;                           // ... radii ...
    uint32_t* s_key = s_rw + 3088 + 7168;                           // ... and depth bits
    const uint32_t* my_partial = partial + (size_t)slice_row(blockIdx.x, gridDim.x) * ntiles_all + tile0;
    for (int t = tid; t < ntiles; t += BIN_THREADS) s_cnt[t] = ranges[tile0 + t].x + my_partial[t];
    __syncthreads();
    // 64-index chunks are dealt round robin over all the waves of all the workgroups: chunk c belongs to workgroup
    // c % nwg, wave slot (c / nwg) % 16, round (c / nwg) / 16
    const int nwg = (int)gridDim.x;
    const int rounds = ((P + 63) / 64 + 16 * nwg - 1) / (16 * nwg);
    for (int it = 0; it < rounds; it++) {
        const int r = ((it * 16 + wave) * nwg + (int)blockIdx.x) * 64 + lane;
        uint2 rmin = make_uint2(0, 0), rmax = make_uint2(0, 0);
        uint32_t count = 0;
        if (r < P) {
            const uint32_t key = depth_key[r];
            if (key != 0xFFFFFFFFu) {   // visible (geometry.h: radius > 0), so its record was written
                const BlendRec rec = index_rec[r];
                getRect(rec.xy.x, rec.xy.y, (int)rec.pm, rmin, rmax, gx, gy);
                rmin.y = max(rmin.y, by0);  // this band's rows only
                rmax.y = min(rmax.y, by1);
                count = rmax.y > rmin.y ? (rmax.x - rmin.x) * (rmax.y - rmin.y) : 0u;
                s_xy[tid] = rec.xy;
                s_co[tid] = rec.co;
                s_rad[tid] = rec.pm;
                s_key[tid] = key;
            }
        }
        for_each_tile_balanced(rw, tid, rmin, rmax, count, [&](uint32_t owner, uint32_t tx, uint32_t ty) {
            const uint32_t qmask = NOCULL ? 15u : span_tile_mask(s_xy[owner], s_co[owner], (int)s_rad[owner], tx, ty, gx, gy);
            const uint32_t id = (uint32_t)(((it * 16 + (int)(owner >> 6)) * nwg + (int)blockIdx.x) * 64 + (int)(owner & 63u));
            const uint32_t slot = atomicAdd(&s_cnt[(ty - by0) * gx + tx], 1u);
            entries[slot] = make_uint2(s_key[owner], id | (qmask << ID_BITS));
        });
    }
}

// Count pass without enumerating the overlaps: every rect adds +1 / -1 / -1 / +1 at its four corners of a
// (gy + 1) x (gx + 1) difference grid in LDS (four LDS atomics per Gaussian instead of one per covered tile plus a
// ten-step owner search), and a 2-D prefix sum of the grid is the number of rects covering each tile -- exactly what
// an enumeration would count.  Same slices (index chunks dealt round robin) and the same rects as bin_ranks_kernel: the count
// pass of the FULL lists.
__host__ __device__ inline int count_grid_stride(uint32_t gx) { return (int)((gx + 1) | 1u); }  // odd row stride: column walks spread over the banks

__global__ void __launch_bounds__(BIN_THREADS) bin_count_kernel(int P, const BlendRec* __restrict__ index_rec,
                                                                const uint32_t* __restrict__ depth_key,
                                                                uint32_t* __restrict__ partial, uint32_t gx, uint32_t gy_all,
                                                                uint32_t by0, uint32_t by1, const int* __restrict__ r_slots,
                                                                int* __restrict__ host_r)
{
    // tile rows [by0, by1) of the image (see bin_ranks_kernel); the difference grid covers the band only
    extern __shared__ int s_grid[];  // [(band rows + 1) * stride]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the partial sums of R for the host (see bin_spans_kernel)
    if (host_r != nullptr && blockIdx.x == 0 && tid < R_SLOTS) host_r[tid * R_SLOT_STRIDE] = r_slots[tid * R_SLOT_STRIDE];
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
        if (depth_key[r] == 0xFFFFFFFFu) continue;
        const BlendRec rec = index_rec[r];
        uint2 rmin, rmax;
        getRect(rec.xy.x, rec.xy.y, (int)rec.pm, rmin, rmax, gx, gy_all);
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
// The count and emit passes of the lean lists without a single per-tile test, WAVE-GRANULAR (round 6).  A wave takes a chunk of
// 64 consecutive Gaussians (the next chunk of its workgroup's slice from an LDS counter; its records are requested one chunk
// ahead), and walks
//   level 1: the (Gaussian, tile row) items of the chunk, 64 at a time, balanced over the lanes (a Gaussian covers 1 .. 68
//            rows): item k belongs to the lane whose inclusive prefix of row counts first exceeds k -- a six-step search in a
//            wave-private LDS copy of the prefix.  Each item evaluates the two closed-form column intervals of its row's upper
//            and lower 8-pixel band (cull.h: band_columns) and gets the tile span [x0, x1) that holds them;
//   COUNT (EMIT = false): +1 / -1 at the two ends of the span in a per-row difference grid in LDS (shared by the workgroup's
//            16 waves: LDS atomics); the prefix along x is the number of spans covering each tile -- this slice's share of the
//            tile's segment, EXACTLY: a tile is counted iff the emit pass stores an entry for it;
//   EMIT:    level 2, the tiles of the 64 spans of a window, balanced again: the quadrant mask of a tile is read off the two
//            column intervals (four range tests on integers), the slot comes from the LDS cursor of the tile.
// The waves of a workgroup share nothing but the counters / cursors: no workgroup barrier between the first store of a
// counter and the last (rounds 2-5 ran this as 1024-thread phases between workgroup barriers: count 0.036 ms, emit 0.071 ms
// on cfg3, ~1 us per phase whatever it computed).  Hand-offs between the lanes of a wave go through wave-private LDS words;
// LDS operations of one wave execute in program order, the wavefront fences keep the compiler from reordering them.
// Measured on cfg3: 2.27 M (Gaussian, row) items, 2.93 M entries.
constexpr int BW_WORDS = 64 * 18;  // LDS words per wave: prefix, rect, depth bits, id, owner scratch, two float4 of span constants + mean, the ids of a chunk; emit: span prefix, two bands' columns, span origin
__host__ __device__ constexpr size_t span_lds_bytes(size_t head_words) { return (((head_words + 3) & ~(size_t)3) + 4 + 16 * BW_WORDS) * sizeof(uint32_t); }
constexpr uint32_t SPAN_MAX_HEAD_WORDS = 40 * 1024 - 16 * BW_WORDS - 64;   // counters / cursors of one band: what is left of 160 KB

// Hand-off point between the lanes of ONE wave through LDS: LDS operations of a wave execute in program order, so no wait is needed --
// but the compiler must neither move LDS accesses across this point nor forward a lane's own store to its later load (another
// lane may have stored there in between): a full fence at wavefront scope (no instruction) plus a scheduling barrier.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Owner of item k = base + lane of a balanced walk: the lane o with excl[o] <= k < incl[o] (excl / incl: the lanes' exclusive /
// inclusive prefix of item counts; k < total).  The lane that owns position `base` is found by a ballot; every lane whose items
// START inside the window (base, base + 64) leaves its number at the start position in a wave-private LDS word, and a running
// maximum over the positions (lane numbers ascend with the prefix) spreads it to the positions behind: one LDS round trip and a
// DPP scan instead of a six-step binary search through LDS.
__device__ __forceinline__ uint32_t wave_owner(uint32_t* scratch /* [64] wave-private */, uint32_t excl, uint32_t incl, uint32_t base, int lane)
{
    const uint64_t first = ballot64(excl <= base && base < incl);
    const uint32_t owner0 = (uint32_t)__builtin_ctzll(first | (1ull << 63));
    // Lanes exchange words through LDS here: FULL fences (acquire + release, wavefront scope).  With release fences alone the
    // compiler forwards a lane's own `scratch[lane] = 0` to its own load of scratch[lane] -- another lane's store to that word is a
    // data race it may assume away -- and the owners come out wrong (round 6: a memory fault in the emit pass; tools/
    // wave_prims_probe.hip checks this function against a serial search on the GPU).
    scratch[lane] = 0u;
    wave_lds_fence();
    if (incl > excl && excl > base && excl - base < 64u) scratch[excl - base] = (uint32_t)lane;
    wave_lds_fence();
    const uint32_t here = scratch[lane];
    wave_lds_fence();
    return max(wave_inclusive_scan_dpp<true>(here), owner0);
}

template <bool EMIT>
__global__ void __launch_bounds__(1024) bin_spans_kernel(int P, const BlendRec* __restrict__ index_rec,
                                                         const uint32_t* __restrict__ depth_key, const unsigned long long* __restrict__ band_bits,
                                                         uint32_t* __restrict__ partial, uint32_t* __restrict__ tile_total, const uint2* __restrict__ ranges,
                                                         uint2* __restrict__ entries, uint32_t gx, uint32_t gy_all,
                                                         uint32_t band_h, uint32_t nbands, const int* __restrict__ r_slots,
                                                         int* __restrict__ host_r, int ablate)
{
    // COUNT: s_dyn = difference grid [band rows][stride] (ints), then the chunk counter and the waves' hand-off words
    // EMIT : s_dyn = cursors [band tiles], then the same
    extern __shared__ uint32_t s_dyn[];
    __shared__ uint32_t s_plan[MAX_BANDS + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // The first kernel behind the preprocess pass hands the partial sums of R to the host: plain stores into its pinned buffer
    // (a copy command of 8 KB costs a 5-us blit kernel on the stream); the event behind this kernel tells the host they are there.
    if (!EMIT && host_r != nullptr && blockIdx.x == 0 && tid < R_SLOTS) host_r[tid * R_SLOT_STRIDE] = r_slots[tid * R_SLOT_STRIDE];
    if (wave == 0) band_plan(r_slots, nbands, gridDim.x, s_plan, lane);
    __syncthreads();
    if (blockIdx.x >= s_plan[nbands]) return;   // (workgroups the plan leaves over)
    uint32_t band = 0;
    while (band + 1u < nbands && s_plan[band + 1u] <= blockIdx.x) band++;
    const uint32_t nsub = s_plan[band + 1u] - s_plan[band], sub = blockIdx.x - s_plan[band];
    const uint32_t by0 = band * band_h, by1 = min(gy_all, by0 + band_h);
    const uint32_t band_bit = 1u << band;
    const uint32_t gy = by1 - by0;
    const int stride = count_grid_stride(gx);
    const int ntiles = (int)(gx * gy);
    const int tile0 = (int)(gx * by0);
    const int head = EMIT ? ((ntiles + 3) & ~3) : (((int)gy * stride + 3) & ~3);
    uint32_t* s_cnt = s_dyn;
    int* s_grid = reinterpret_cast<int*>(s_dyn);
    uint32_t* s_next = s_dyn + head;   // [4]: next group of mask chunks of this workgroup's share
    uint32_t* wb = s_dyn + head + 4 + wave * BW_WORDS;
    uint32_t* w_pre = wb;              // inclusive prefix of the lanes' row counts
    uint32_t* w_rect = wb + 64;        // clip columns x0 | x1 << 10, first row << 21
    uint32_t* w_key = wb + 128;        // depth bits
    uint32_t* w_own = wb + 192;        // scratch of wave_owner
    float4* w_c0 = reinterpret_cast<float4*>(wb + 256);   // span constants of the Gaussian (cull.h: SpanPre): B, 1/A, 2 tau A, det
    float4* w_c1 = reinterpret_cast<float4*>(wb + 512);   //   ey (negative: no culling), y*, mean x, mean y
    uint32_t* w_tpre = wb + 768;       // EMIT, per span of the window: inclusive prefix of the spans' widths
    uint32_t* w_q0 = wb + 832;         //   upper band's columns lo | hi << 11, owner lane << 22
    uint32_t* w_q1 = wb + 896;         //   lower band's columns lo | hi << 11
    uint32_t* w_sx = wb + 960;         //   first tile column | tile row << 10
    uint32_t* w_id = wb + 1024;        // Gaussian id of the lane's record
    uint32_t* w_queue = wb + 1088;     // [64]: ids of the chunk being put together
    uint32_t* my_partial = partial + (size_t)blockIdx.x * (band_h * gx);   // [workgroup][tile of its band]
    if (EMIT) {
        for (int t = tid; t < ntiles; t += 1024) s_cnt[t] = ranges[tile0 + t].x + my_partial[t];
    } else {
        for (int c = tid; c < (int)gy * stride; c += 1024) s_grid[c] = 0;
    }
    if (tid == 0) s_next[0] = 0u;
    __syncthreads();
    // This workgroup's share of the Gaussians: index chunks (64 consecutive indices) [mc0, mc1) of the view's, an equal part of the
    // index range for each of the band's workgroups.
    const uint32_t nchunks_all = ((uint32_t)P + 63u) / 64u;
    const uint32_t mc0 = (uint32_t)(((uint64_t)sub * nchunks_all) / nsub), mc1 = (uint32_t)(((uint64_t)(sub + 1u) * nchunks_all) / nsub);
    // The band's Gaussians come as bit words: band_bits[band][c] = which of the 64 Gaussians of index chunk c reach this band
    // (geometry.h).  A wave takes UNITS of 32 words (2048 Gaussians) from the workgroup's LDS counter, one unit requested ahead; the
    // set bits of a unit are ranked (popcount + DPP scan over the lanes' words), and a chunk of <= 64 Gaussians is the next 64 ranks:
    // every lane whose word holds some of them walks its set bits and leaves the ids in the wave's queue.
    constexpr uint32_t UNIT = 32;
    const unsigned long long* my_bits = band_bits + (size_t)band * nchunks_all;
    uint32_t g_unit = 0, g_next;                    // first index chunk of the unit being sifted / of the one requested ahead
    unsigned long long word = 0ull, nword;          // lane L < 32: the word of chunk g_unit + L
    uint32_t cnt = 0, incl = 0, total = 0, r0 = 0;  // the lane's set bits, their inclusive prefix over the lanes, the unit's sum, ranks taken so far
    auto request_unit = [&]() __attribute__((always_inline)) {
        uint32_t u = 0;
        if (lane == 0) u = atomicAdd(s_next, 1u);
        g_next = mc0 + UNIT * (uint32_t)__builtin_amdgcn_readfirstlane(u);
        nword = my_bits[min(g_next + (uint32_t)lane, nchunks_all - 1u)];   // (unconditional, clamped; validity is applied in next_unit)
    };
    auto next_unit = [&]() __attribute__((always_inline)) {
        g_unit = g_next;
        word = ((uint32_t)lane < UNIT && g_unit + (uint32_t)lane < mc1) ? nword : 0ull;
        request_unit();
        cnt = (uint32_t)__builtin_popcountll(word);
        incl = wave_inclusive_scan_dpp(cnt);
        total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        r0 = 0;
    };
    // next chunk of <= 64 Gaussians of this band -> id (0xFFFFFFFF: lane without one); false: this workgroup's share is exhausted
    auto take_chunk = [&](uint32_t& id) __attribute__((always_inline)) -> bool {
        uint32_t qn = 0;
        while (qn < 64u) {
            if (r0 == total) {
                if (g_next >= mc1) break;
                next_unit();
                continue;
            }
            const uint32_t t = min(64u - qn, total - r0);
            uint32_t rk = incl - cnt;   // rank of this lane's first set bit
            if (incl > r0 && rk < r0 + t) {
                for (unsigned long long w = word; w != 0ull; w &= w - 1ull, rk++)
                    if (rk >= r0 && rk < r0 + t) w_queue[qn + rk - r0] = (g_unit + (uint32_t)lane) * 64u + (uint32_t)__builtin_ctzll(w);
            }
            r0 += t;
            qn += t;
        }
        if (qn == 0u) return false;
        wave_lds_fence();
        id = (uint32_t)lane < qn ? w_queue[lane] : 0xFFFFFFFFu;
        wave_lds_fence();
        return true;
    };
    request_unit();
    uint32_t nid = 0xFFFFFFFFu;
    bool have = take_chunk(nid);
    BlendRec nrec = index_rec[nid == 0xFFFFFFFFu ? 0u : nid];
    while (have) {
        const uint32_t id = nid;
        const BlendRec rec = nrec;
        // the next chunk's ids, and its records requested before this chunk's items are walked
        nid = 0xFFFFFFFFu;
        have = take_chunk(nid);
        nrec = index_rec[nid == 0xFFFFFFFFu ? 0u : nid];
        uint32_t h = 0;
        if (id != 0xFFFFFFFFu) {   // visible (its band bit was set): radius > 0, record written (geometry.h)
            const int rad = (int)rec.pm;
            uint2 rmin, rmax;
            getRect(rec.xy.x, rec.xy.y, rad, rmin, rmax, gx, gy_all);
            shrink_rect(rec.xy, rec.co, rad, rmin, rmax);
            rmin.y = max(rmin.y, by0);
            rmax.y = min(rmax.y, by1);
            if (rmax.x > rmin.x && rmax.y > rmin.y) {
                h = rmax.y - rmin.y;
                // the span constants ONCE per Gaussian (two square roots, a logarithm, three reciprocals), not once per row item
                const SpanPre pre = span_prepare(rec.co, rad);
                w_c0[lane] = make_float4(pre.B, pre.rcpA, pre.twotauA, pre.det);
                w_c1[lane] = make_float4(pre.cull ? pre.ey : -1.0f, pre.ystar, rec.xy.x, rec.xy.y);
                w_rect[lane] = rmin.x | (rmax.x << 10) | (rmin.y << 21);
                if (EMIT) {
                    w_key[lane] = rec.id;   // depth bits (BlendRec)
                    w_id[lane] = id;
                }
            }
        }
        const uint32_t hincl = wave_inclusive_scan_dpp(h);
        uint32_t rows_total = (uint32_t)__builtin_amdgcn_readlane((int)hincl, 63);
        w_pre[lane] = hincl;
        wave_lds_fence();
        if MI_ABLATE(1 << 20) rows_total = 0;
        for (uint32_t w0 = 0; w0 < rows_total; w0 += 64) {  // windows of 64 (Gaussian, tile row) items
            const uint32_t k = w0 + (uint32_t)lane;
            uint32_t width = 0;
            const uint32_t g = wave_owner(w_own, hincl - h, hincl, w0, lane);
            if (k < rows_total) {
                const uint32_t prev = g == 0u ? 0u : w_pre[g - 1];
                const uint32_t packed = w_rect[g];
                const float4 c0 = w_c0[g], c1 = w_c1[g];
                const uint32_t cx0 = packed & 1023u, cx1 = (packed >> 10) & 2047u, ty = (packed >> 21) + (k - prev);
                SpanPre pre;
                pre.B = c0.x;
                pre.rcpA = c0.y;
                pre.twotauA = c0.z;
                pre.det = c0.w;
                pre.cull = c1.x >= 0.f;
                pre.ey = c1.x;
                pre.ystar = c1.y;
                const float2 xy = make_float2(c1.z, c1.w);
                int lo0, hi0, lo1, hi1;
                band_columns(pre, xy, (float)(ty * TILE_Y), (int)(2u * cx0), (int)(2u * cx1), lo0, hi0);
                band_columns(pre, xy, (float)(ty * TILE_Y + 8u), (int)(2u * cx0), (int)(2u * cx1), lo1, hi1);
                if (hi0 > lo0 || hi1 > lo1) {
                    const int lo = hi0 > lo0 ? (hi1 > lo1 ? min(lo0, lo1) : lo0) : lo1;
                    const int hi = hi0 > lo0 ? (hi1 > lo1 ? max(hi0, hi1) : hi0) : hi1;
                    const uint32_t sx0 = (uint32_t)lo >> 1, sx1 = ((uint32_t)hi + 1u) >> 1;
                    if (EMIT) {
                        width = sx1 - sx0;
                        w_q0[lane] = (uint32_t)lo0 | ((uint32_t)hi0 << 11) | (g << 22);
                        w_q1[lane] = (uint32_t)lo1 | ((uint32_t)hi1 << 11);
                        w_sx[lane] = sx0 | (ty << 10);
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
                            atomicAdd(&row[sx0], 1);
                            atomicAdd(&row[sx1], -1);
                        }
                    }
                }
            }
            if (EMIT && !MI_ABLATE(1 << 16)) {
                const uint32_t tincl = wave_inclusive_scan_dpp(width);
                const uint32_t tiles_total = (uint32_t)__builtin_amdgcn_readlane((int)tincl, 63);
                w_tpre[lane] = tincl;
                wave_lds_fence();
                for (uint32_t t0 = 0; t0 < tiles_total; t0 += 64) {   // windows of 64 tiles of the spans
                    const uint32_t kk = t0 + (uint32_t)lane;
                    const uint32_t o = wave_owner(w_own, tincl - width, tincl, t0, lane);
                    if (kk < tiles_total) {
                        const uint32_t prevt = o == 0u ? 0u : w_tpre[o - 1];
                        const uint32_t q0 = w_q0[o], q1 = w_q1[o], sx = w_sx[o];
                        const uint32_t tx = (sx & 1023u) + (kk - prevt), ty = sx >> 10;
                        const uint32_t lo0 = q0 & 2047u, n0 = ((q0 >> 11) & 2047u) - lo0, lo1 = q1 & 2047u, n1 = ((q1 >> 11) & 2047u) - lo1;
                        const uint32_t c = 2u * tx;
                        const uint32_t qmask = (uint32_t)(c - lo0 < n0) | ((uint32_t)(c + 1u - lo0 < n0) << 1) |
                                               ((uint32_t)(c - lo1 < n1) << 2) | ((uint32_t)(c + 1u - lo1 < n1) << 3);
                        if (qmask != 0u && !MI_ABLATE(1 << 17)) {
                            const uint32_t gg = q0 >> 22;
                            const uint32_t slot = atomicAdd(&s_cnt[(ty - by0) * gx + tx], 1u);
                            if (!MI_ABLATE(1 << 19)) entries[slot] = make_uint2(w_key[gg], w_id[gg] | (qmask << ID_BITS));
                        }
                    }
                }
                wave_lds_fence();   // the spans' words are rewritten by the next window
            }
        }
        wave_lds_fence();   // the Gaussians' words are rewritten by the next chunk
    }
    if (!EMIT) {
        __syncthreads();
        // prefix along x: one wave per row, 64 cells at a time with a carry -> the per-tile counts of this workgroup
        for (int y = wave; y < (int)gy; y += 16) {
            int carry = 0;
            for (int x0 = 0; x0 < (int)gx; x0 += 64) {
                const int x = x0 + lane;
                int v = x < (int)gx ? s_grid[y * stride + x] : 0;
                v = (int)wave_inclusive_scan_dpp((uint32_t)v) + carry;   // (two's complement: the signed differences add up like unsigned words)
                // this workgroup's entries of the tile take the next v slots of the tile's segment: one returning atomic per
                // (workgroup, non-empty tile) on the tile's total -- ~130 K of them per cfg3 view, ~18 per address, in place of a
                // partial[slice][tile] table and its scan (the order of the workgroups inside a segment is whatever the atomics
                // make it: the per-tile sort does not care)
                if (x < (int)gx) my_partial[y * (int)gx + x] = v != 0 ? atomicAdd(&tile_total[tile0 + y * (int)gx + x], (uint32_t)v) : 0u;
                carry = __builtin_amdgcn_readlane(v, 63);
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

// ---- per-tile LDS radix sort of the entries by depth bits ------------------------------------------------
// (key, value) pair storage: two LDS arrays, or one uint2 array in HBM
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

// One workgroup (256 threads for short lists, 1024 for long ones) per tile; handles tiles with LO < n <= CAP in LDS; when GLOBAL_FALLBACK is set it
// also handles n > CAP by ping-ponging between `entries` and `scratch` in HBM with the same code.
// LSD radix over the depth bits, 8-bit digits, ceil(key_bits / 8) passes (key_bits: the low bits in which the view's keys differ,
// from the range scan).  Each wave owns a contiguous share of the tile's list; per pass: per-wave digit histograms -> workgroup scan
// over (digit, wave) -> each wave scatters its share 64 pairs at a time with a stable ballot-match rank.  Three workgroup barriers per pass.
// MAXB > 0 (lists held in LDS: a wave's share is at most MAXB batches of 64 pairs): the pairs and their match results
// (rank among the wave's equal digits, size of that group) stay in registers between the histogram and the scatter
// phase -- the eight-ballot match is evaluated once per pair and pass instead of twice.  MAXB = 0: any length (the
// HBM fallback), everything re-read and re-matched in the scatter phase.
template <int NW, int MAXB, typename Src, typename Dst>
__device__ __forceinline__ void radix_pass(Src src, Dst dst, int n, int shift, uint32_t (*s_hist)[256], int tid)
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
    uint32_t c_key[NB], c_val[NB], c_rk[NB], c_cnt[NB];
    // (a) per-wave histogram of this digit
    if (MAXB > 0) {
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const int i0 = begin + 64 * j;
            if (i0 >= end) break;  // wave-uniform
            const int i = i0 + lane;
            const bool active = i < end;
            c_key[j] = active ? src.key(i) : 0u;
            c_val[j] = active ? src.val(i) : 0u;
            const uint32_t digit = (c_key[j] >> shift) & 0xFFu;
            match(digit, active, c_rk[j], c_cnt[j]);
            if (active && c_rk[j] == 0) s_hist[wave][digit] += c_cnt[j];  // one lane per distinct digit, wave-private row
        }
    } else {
        for (int i0 = begin; i0 < end; i0 += 64) {
            const int i = i0 + lane;
            const bool active = i < end;
            const uint32_t key = active ? src.key(i) : 0u;
            const uint32_t digit = (key >> shift) & 0xFFu;
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
    auto scatter = [&](uint32_t key, uint32_t val, bool active, uint32_t rk, uint32_t cnt) {
        const uint32_t digit = (key >> shift) & 0xFFu;
        uint32_t off = 0;
        if (active) off = s_hist[wave][digit] + rk;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (active && rk == cnt - 1) s_hist[wave][digit] = off + 1;  // last lane of the group advances the cursor
        if (active) dst.set((int)off, key, val);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    };
    if (MAXB > 0) {
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const int i0 = begin + 64 * j;
            if (i0 >= end) break;
            scatter(c_key[j], c_val[j], i0 + lane < end, c_rk[j], c_cnt[j]);
        }
    } else {
        for (int i0 = begin; i0 < end; i0 += 64) {
            const int i = i0 + lane;
            const bool active = i < end;
            const uint32_t key = active ? src.key(i) : 0u, val = active ? src.val(i) : 0u;
            uint32_t rk, cnt;
            match((key >> shift) & 0xFFu, active, rk, cnt);
            scatter(key, val, active, rk, cnt);
        }
    }
    __syncthreads();
}

// The tile's BLEND LIST: the values of the sorted entries, Gaussian id | quadrant mask << 28, four bytes per overlap at
// blend_list[range.x ...].  This is all the blend kernels get of a tile: they scan the entries 64 at a time (256
// contiguous bytes), queue the ones whose bit for their quadrant is set and gather the 32-byte geometry record index_rec[id] next to
// the feature row of the same id -- two independent gathers behind one scan.  The position of an entry
// in its list is its index -- n_contrib's unit.  Full lists (bin_ranks_kernel): every overlap of the reference's rects is an entry,
// culled ones with mask 0, so blend_list & ID_MASK IS the reference's point_list and the index its position there; lean lists
// (bin_spans_kernel): the entries that can blend only.
// The radix passes order by depth bits and keep the ARRIVAL order of equal keys, which is arbitrary; the contract orders equal depths
// by Gaussian index (the reference's stable sort of (tile | depth) keys emitted in index order, rasterizer_impl.cu:96-113, 300-308).
// Entries whose neighbour in the sorted list has the same key -- two Gaussians of one tile at bit-identical depth: rare, but not
// excluded -- find their run and take the position of their id inside it.
template <int NW, typename Src>
__device__ __forceinline__ void emit_blend_list(Src sorted, int n, uint2 range, int tid, uint32_t* __restrict__ blend_list)
{
    uint32_t* out = blend_list + range.x;
    for (int i = tid; i < n; i += NW * 64) {
        const uint32_t k = sorted.key(i), v = sorted.val(i);
        int pos = i;
        if ((i > 0 && sorted.key(i - 1) == k) || (i + 1 < n && sorted.key(i + 1) == k)) {
            int lo = i, hi = i + 1;
            while (lo > 0 && sorted.key(lo - 1) == k) lo--;
            while (hi < n && sorted.key(hi) == k) hi++;
            int before = 0;
            for (int q = lo; q < hi; q++) before += (sorted.val(q) & ID_MASK) < (v & ID_MASK) ? 1 : 0;
            pos = lo + before;
        }
        out[pos] = v;
    }
}

// MI_RAST_VERIFY_LISTS (include/mi_rast.h): the lean lists rest on the count pass and the emit pass taking bit-identical float
// decisions in two template instances of bin_spans_kernel -- a tile is counted iff the emit pass stores an entry for it.  With
// the flag the entries are zero-filled before the emit pass and this kernel counts the slots of every tile's segment that
// are still zero afterwards (a lean entry always carries a non-zero quadrant mask): any such slot is a decision that differed.
__global__ void __launch_bounds__(256) verify_entries_kernel(uint32_t ntiles, const uint2* __restrict__ ranges,
                                                              const uint2* __restrict__ entries, uint32_t* __restrict__ unwritten)
{
    const uint32_t tile = blockIdx.x;
    if (tile >= ntiles) return;
    const uint2 r = ranges[tile];
    uint32_t n = 0;
    for (uint32_t i = r.x + threadIdx.x; i < r.y; i += 256) n += entries[i].y == 0u;
    if (n) atomicAdd(unwritten, n);
}

// ---- per-tile sort of short lists: one WAVE per tile, bitonic network in registers ------------------------------------
// Lists of up to 2048 entries (every list of a 1 M-Gaussian 1080p view; most of a 5 M-Gaussian one) are ordered by ONE wave
// each, with no LDS and no barrier: lane l holds the EPL consecutive elements l EPL .. l EPL + EPL - 1 of the list padded to
// N = 64 EPL (EPL = 4 / 8 / 16 / 32 by list length), each as one 64-bit word  depth bits << 32 | id << 4 | quadrant mask  -- ids
// are distinct inside a tile, so ordering the words orders by (depth bits, id): the contract's order, ties included, with no
// second pass.
//   * The words are compared as binary64 numbers: depth bits are those of a positive finite float (sign clear, exponent field
//     below 0xFF), so the word's top twelve bits are a binary64 exponent below 0x7FF -- never NaN or infinity --, and positive
//     binary64 numbers order like their bit patterns.  v_min_f64 / v_max_f64 are a 64-bit compare-exchange in TWO instructions;
//     v_cmp_lt_u64 + four v_cndmask_b32, the integer form, measured 0.087 ms per cfg3 view.
//   * The network is the all-ascending form of the bitonic sorter: stage K first compares element i with i ^ (K - 1) (the mirror
//     image inside its block of K), then i with i ^ j for j = K / 4 .. 1; the smaller word always goes to the smaller index, so no
//     step carries a direction.  Distances below EPL stay inside a lane (register pairs); the others pair lane l with
//     l ^ M: DPP moves for M = 1, 2, 3, 4, 7, 8, 15 (quad_perm, row_half_mirror, row_mirror, row_shl / row_shr under bank masks),
//     ds_swizzle for 16 and 31, ds_bpermute for 32 and 63 -- 21 of the 45 steps of a 512-entry list cross lanes.
template <int M>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v)
{
    // (full row / bank masks, every lane has a source: v_mov_b32_dpp without an `old` operand -- update_dpp(0, ...) costs a
    // v_mov_b32 0 in front of every move)
    if constexpr (M == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);         // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    else if constexpr (M == 3) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x1B, 0xF, 0xF, true);    // quad_perm [3,2,1,0]
    else if constexpr (M == 7) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    else if constexpr (M == 15) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xF, 0xF, true);  // row_mirror
    else if constexpr (M == 4) {   // lanes 0-3 / 8-11 of a row take lane + 4 (row_shl), the others lane - 4 (row_shr)
        const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x104, 0xF, 0x5, false);
        return (uint32_t)__builtin_amdgcn_update_dpp((int)t, (int)v, 0x114, 0xF, 0xA, false);
    } else if constexpr (M == 8) {
        const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x108, 0xF, 0x3, false);
        return (uint32_t)__builtin_amdgcn_update_dpp((int)t, (int)v, 0x118, 0xF, 0xC, false);
    } else if constexpr (M == 16 || M == 31) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (M << 10) | 0x1F);   // bit mode: lane ^ M inside 32 lanes
    else return (uint32_t)__shfl_xor((int)v, M, 64);   // 32, 63
}
template <int M>
__device__ __forceinline__ double lane_xor64(double v)
{
    const uint64_t w = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = lane_xor<M>((uint32_t)w), hi = lane_xor<M>((uint32_t)(w >> 32));
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ void minmax64(double& lo, double& hi)   // (inline asm: no canonicalisation of the operands in front of it)
{
    double mn;
    asm("v_min_f64 %0, %1, %2" : "=v"(mn) : "v"(lo), "v"(hi));
    asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(lo), "v"(hi));   // (may overwrite an operand in place: no copies around the pair)
    lo = mn;
}

// one step of stage K: FLIP -- partner i ^ (K - 1) -- or distance J
template <int EPL, int K, int J, bool FLIP>
__device__ __forceinline__ void bitonic_step(double (&a)[EPL], int lane)
{
    if constexpr (FLIP && K > EPL) {
        // i ^ (K - 1) = (lane ^ (K / EPL - 1)) * EPL + (EPL - 1 - r): the mirrored register of the mirrored lane
        constexpr int M = K / EPL - 1;
        const bool lower = (lane & ((M + 1) >> 1)) == 0;
#pragma unroll
        for (int r = 0; r < EPL / 2; r++) {   // registers r and EPL - 1 - r trade places with the partner lane's: two temporaries at a time
            double x0 = a[r], y0 = lane_xor64<M>(a[EPL - 1 - r]);
            double x1 = a[EPL - 1 - r], y1 = lane_xor64<M>(a[r]);
            minmax64(x0, y0);
            minmax64(x1, y1);
            a[r] = lower ? x0 : y0;
            a[EPL - 1 - r] = lower ? x1 : y1;
        }
    } else if constexpr (!FLIP && J >= EPL) {
        constexpr int M = J / EPL;
        const bool lower = (lane & M) == 0;
#pragma unroll
        for (int r = 0; r < EPL; r++) {
            double x = a[r], y = lane_xor64<M>(a[r]);
            minmax64(x, y);
            a[r] = lower ? x : y;
        }
    } else {
        constexpr int X = FLIP ? K - 1 : J;   // in-lane partner r ^ X
#pragma unroll
        for (int r = 0; r < EPL; r++)
            if ((r ^ X) > r) minmax64(a[r], a[r ^ X]);
    }
}
template <int EPL, int K, int J>
__device__ __forceinline__ void bitonic_tail(double (&a)[EPL], int lane)   // distances J, J / 2, .. 1 of stage K, then the next stage
{
    if constexpr (J >= 1) {
        bitonic_step<EPL, K, J, false>(a, lane);
        bitonic_tail<EPL, K, J / 2>(a, lane);
    } else if constexpr (K < 64 * EPL) {
        bitonic_step<EPL, 2 * K, 0, true>(a, lane);
        bitonic_tail<EPL, 2 * K, K / 2>(a, lane);
    }
}

template <int EPL>
__device__ __forceinline__ void tile_sort_in_wave(const uint2* __restrict__ seg, int n, uint32_t* __restrict__ out, int lane)
{
    double a[EPL];
#pragma unroll
    for (int r = 0; r < EPL; r++) {
        const int e = r * 64 + lane;   // (any order will do going IN: coalesced loads; the sorted sequence is indexed lane * EPL + r)
        uint64_t w = 0x7FEFFFFFFFFFFFFFull;   // padding: the largest finite binary64, behind every entry
        if (e < n) {
            const uint2 p = seg[e];
            w = ((uint64_t)p.x << 32) | __builtin_rotateleft32(p.y, 4);   // id | mask << 28  ->  id << 4 | mask
        }
        a[r] = __builtin_bit_cast(double, w);
    }
    bitonic_step<EPL, 2, 0, true>(a, lane);
    bitonic_tail<EPL, 2, 0>(a, lane);
#pragma unroll
    for (int r = 0; r < EPL; r++) {
        const int e = lane * EPL + r;
        if (e < n) out[e] = __builtin_rotateright32((uint32_t)__builtin_bit_cast(uint64_t, a[r]), 4);
    }
}

constexpr int TILE_SORT_WAVE_MAX = 4096;   // longest list the wave kernels order (EPL = 64); longer ones: tile_sort_kernel below
// CLASS 0: lists of 1 .. 1024 entries (EPL = 4 / 8 / 16: 52 VGPRs); 1: 1025 .. 2048 (EPL = 32); 2: 2049 .. 4096 (EPL = 64) -- the
// longer classes are launched only when a tile needs them: one kernel for all would run the short lists at the long ones' register count.
template <int CLASS>
__global__ void __launch_bounds__(256) tile_sort_wave_kernel(uint32_t ntiles, const uint2* __restrict__ ranges,
                                                              const uint2* __restrict__ entries, uint32_t* __restrict__ blend_list)
{
    const int lane = threadIdx.x & 63;
    const uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    const uint2* seg = entries + range.x;
    uint32_t* out = blend_list + range.x;
    if constexpr (CLASS == 2) {
        if (n > 2048 && n <= 4096) tile_sort_in_wave<64>(seg, n, out, lane);
    } else if constexpr (CLASS == 1) {
        if (n > 1024 && n <= 2048) tile_sort_in_wave<32>(seg, n, out, lane);
    } else {
        if (n == 0 || n > 1024) return;
        if (n <= 256) tile_sort_in_wave<4>(seg, n, out, lane);
        else if (n <= 512) tile_sort_in_wave<8>(seg, n, out, lane);
        else tile_sort_in_wave<16>(seg, n, out, lane);
    }
}

template <int LO, int CAP, bool GLOBAL_FALLBACK, int NT>
__global__ void __launch_bounds__(NT) tile_sort_kernel(uint32_t ntiles, const uint2* __restrict__ ranges,
                                                        uint2* __restrict__ entries, uint2* __restrict__ scratch,
                                                        const int* __restrict__ key_bits, uint32_t* __restrict__ blend_list)
{
    __shared__ uint32_t s_k[2][CAP];
    __shared__ uint32_t s_v[2][CAP];
    constexpr int NW = NT / 64;
    __shared__ uint32_t s_hist[NW][256];
    const int tid = threadIdx.x;
    // (tile = workgroup id: neighbouring tiles on different XCDs.  The cost of a tile is its list length, and a static split into
    // contiguous runs per XCD -- common.h -- would let a scene's dense half wait for two of the eight XCDs.)
    const uint32_t tile = blockIdx.x;
    if (tile >= ntiles) return;
    const uint2 range = ranges[tile];
    const int passes = (*key_bits + 7) >> 3;
    const int n = (int)(range.y - range.x);
    if (n <= LO) return;
    if (n > CAP && !GLOBAL_FALLBACK) return;
    uint2* seg = entries + range.x;
    if (n <= CAP) {
        // (lean lists too: their counts are exact -- bin_spans_kernel --, every slot of the segment holds an entry)
        for (int i = tid; i < n; i += NT) {
            const uint2 e = seg[i];
            s_k[0][i] = e.x;
            s_v[0][i] = e.y;
        }
        __syncthreads();
        int cur = 0;
        for (int p = 0; p < passes; p++, cur ^= 1)
            radix_pass<NW, (CAP + NW * 64 - 1) / (NW * 64)>(LdsPairs{s_k[cur], s_v[cur]}, LdsPairs{s_k[cur ^ 1], s_v[cur ^ 1]}, n, 8 * p, s_hist, tid);
        emit_blend_list<NW>(LdsPairs{s_k[cur], s_v[cur]}, n, range, tid, blend_list);
    } else {
        uint2* a = seg;
        uint2* b = scratch + range.x;
        for (int p = 0; p < passes; p++) {
            radix_pass<NW, 0>(GlobalPairs{a}, GlobalPairs{b}, n, 8 * p, s_hist, tid);
            uint2* t = a;
            a = b;
            b = t;
        }
        emit_blend_list<NW>(GlobalPairs{a}, n, range, tid, blend_list);
    }
}

}  // namespace mirast
