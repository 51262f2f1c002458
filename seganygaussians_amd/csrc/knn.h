// knn.h -- exact K-nearest-neighbour search over 3-D points on the GPU (include/mi_knn.h).
//
// Replaces, at the edges of the hot path:
//   * pytorch3d.ops.knn_points as SAGA calls it (scene/gaussian_model_ff.py:326,347,380: neighbour maps for the feature
//     smoothing, K = 16 / 4, queries == references or references = a subset), and
//   * simple_knn._C.distCUDA2 (submodules/simple-knn/simple_knn.cu:185-218, spatial.cu:16-25: mean squared distance to
//     the 3 nearest other points, used by create_from_pcd).
// Same idea as simple-knn (Morton order + bounding boxes + exact pruning), laid out for gfx950:
//   1. bounding box of the references (ordered-int atomics), 30-bit Morton code per point;
//   2. stable LSD radix sort of (code, index), 4 passes of 8 bits: per-block histograms -> one scan -> ordered scatter
//      (ballot-match ranking inside a wave, running digit counts across the waves of a block);
//   3. the sorted points as float4 {x, y, z, index}; leaf boxes of 64 consecutive points and super boxes of 64 leaves
//      (4096 points) with their axis-aligned bounds -- equal COUNT per box, so dense regions get small boxes;
//   4. one thread per query (queries processed in the references' Morton order when they are the references, so the
//      lanes of a wave walk the same boxes): best-K list in registers, seeded from the neighbours in Morton order,
//      then every super box / leaf whose box distance does not exceed the current K-th distance is scanned.
// The result is exact: K smallest squared distances (d.x*d.x + d.y*d.y + d.z*d.z, unfused, as simple_knn.cu:139-141),
// ascending, ties by index.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace mirast {

constexpr int KNN_LEAF = 64;      // points per leaf box
constexpr int KNN_FAN = 64;       // leaves per super box
constexpr int KNN_TILE = 1024;    // keys per block of the radix sort
constexpr int KNN_MAXK = 32;

struct KnnBox {
    float3 lo, hi;
};

__device__ __forceinline__ uint32_t f2ord(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// bbox[0..2] = min (ordered ints), bbox[3..5] = max; initialised to 0xffffffff / 0 by knn_init_kernel
__global__ void knn_init_kernel(uint32_t* bbox)
{
    if (threadIdx.x < 3) bbox[threadIdx.x] = 0xffffffffu;
    else if (threadIdx.x < 6) bbox[threadIdx.x] = 0u;
}

__global__ void __launch_bounds__(256) knn_bbox_kernel(int M, const float* __restrict__ pts, uint32_t* __restrict__ bbox)
{
    uint32_t lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const uint32_t o = f2ord(pts[3 * (size_t)i + a]);
            lo[a] = min(lo[a], o);
            hi[a] = max(hi[a], o);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            lo[a] = min(lo[a], (uint32_t)__shfl_xor((int)lo[a], o, 64));
            hi[a] = max(hi[a], (uint32_t)__shfl_xor((int)hi[a], o, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&bbox[a], lo[a]);
            atomicMax(&bbox[3 + a], hi[a]);
        }
    }
}

// simple_knn.cu:44-61 (prepMorton / coord2Morton): 10 bits per axis
__device__ __forceinline__ uint32_t prep_morton(uint32_t x)
{
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}
__device__ __forceinline__ uint32_t morton_of(float px, float py, float pz, const uint32_t* __restrict__ bbox)
{
    uint32_t c[3];
    const float p[3] = {px, py, pz};
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float lo = ord2f(bbox[a]), hi = ord2f(bbox[3 + a]);
        const float ext = hi - lo;
        float t = ext > 0.f ? (p[a] - lo) / ext : 0.f;
        t = fminf(fmaxf(t, 0.f), 1.f);  // queries may lie outside the references' box
        c[a] = prep_morton((uint32_t)(t * 1023.0f));
    }
    return c[0] | (c[1] << 1) | (c[2] << 2);
}

__global__ void __launch_bounds__(256) knn_morton_kernel(int M, const float* __restrict__ pts, const uint32_t* __restrict__ bbox,
                                                         uint32_t* __restrict__ codes, uint32_t* __restrict__ index)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    codes[i] = morton_of(pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], bbox);
    index[i] = (uint32_t)i;
}

// ---- stable LSD radix sort of (key, value) pairs, 8 bits per pass -------------------------------------------------
__global__ void __launch_bounds__(256) knn_radix_hist_kernel(int n, const uint32_t* __restrict__ keys, int shift, int nblocks,
                                                             uint32_t* __restrict__ hist /*[256][nblocks]*/)
{
    __shared__ uint32_t s_h[256];
    s_h[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * KNN_TILE;
#pragma unroll
    for (int r = 0; r < KNN_TILE / 256; r++) {
        const int e = base + r * 256 + threadIdx.x;
        if (e < n) atomicAdd(&s_h[(keys[e] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = s_h[threadIdx.x];
}

// exclusive scan of `count` values in place (one workgroup of 1024 threads)
__global__ void __launch_bounds__(1024) knn_scan_kernel(int count, uint32_t* __restrict__ data)
{
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < count; base += 1024) {
        const int i = base + tid;
        const uint32_t v = i < count ? data[i] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; w++) woff += s_w[w];
        const uint32_t carry = s_carry;
        if (i < count) data[i] = carry + woff + inc - v;
        __syncthreads();
        if (tid == 1023) s_carry = carry + woff + inc;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) knn_radix_scatter_kernel(int n, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                                int shift, int nblocks, const uint32_t* __restrict__ hist,
                                                                uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out)
{
    __shared__ uint32_t s_run[256];       // elements of each digit placed by earlier rounds / waves of this block
    __shared__ uint32_t s_wcnt[4][256];   // this round: elements of each digit per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    s_run[tid] = 0;
    const int base = blockIdx.x * KNN_TILE;
    for (int r = 0; r < KNN_TILE / 256; r++) {
#pragma unroll
        for (int w = 0; w < 4; w++) s_wcnt[w][tid] = 0;
        __syncthreads();
        const int e = base + r * 256 + tid;
        const bool valid = e < n;
        const uint32_t key = valid ? keys[e] : 0u;
        const uint32_t d = valid ? ((key >> shift) & 255u) : 256u;
        // lanes of this wave with the same digit (ballot-match over the 9 bits of d)
        uint64_t same = ~0ull;
#pragma unroll
        for (int b = 0; b < 9; b++) {
            const uint64_t bal = ballot64((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? bal : ~bal;
        }
        const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        const uint32_t rank_in_wave = (uint32_t)__builtin_popcountll(same & lt);
        if (valid && rank_in_wave == 0) s_wcnt[wave][d] = (uint32_t)__builtin_popcountll(same);
        __syncthreads();
        if (valid) {
            uint32_t pos = hist[(size_t)d * nblocks + blockIdx.x] + s_run[d] + rank_in_wave;
            for (int w = 0; w < wave; w++) pos += s_wcnt[w][d];
            keys_out[pos] = key;
            vals_out[pos] = vals[e];
        }
        __syncthreads();
        s_run[tid] += s_wcnt[0][tid] + s_wcnt[1][tid] + s_wcnt[2][tid] + s_wcnt[3][tid];
        __syncthreads();
    }
}

// ---- sorted points and the two box levels -------------------------------------------------------------------------
__global__ void __launch_bounds__(KNN_LEAF) knn_leaf_kernel(int M, const float* __restrict__ pts, const uint32_t* __restrict__ sorted_idx,
                                                            float4* __restrict__ sorted_pts, KnnBox* __restrict__ leaves)
{
    const int i = blockIdx.x * KNN_LEAF + threadIdx.x;
    float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    if (i < M) {
        const uint32_t id = sorted_idx[i];
        const float x = pts[3 * (size_t)id], y = pts[3 * (size_t)id + 1], z = pts[3 * (size_t)id + 2];
        sorted_pts[i] = make_float4(x, y, z, __uint_as_float(id));
        lo[0] = hi[0] = x;
        lo[1] = hi[1] = y;
        lo[2] = hi[2] = z;
    }
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64));
        }
    if (threadIdx.x == 0) {
        leaves[blockIdx.x].lo = make_float3(lo[0], lo[1], lo[2]);
        leaves[blockIdx.x].hi = make_float3(hi[0], hi[1], hi[2]);
    }
}

__global__ void __launch_bounds__(KNN_FAN) knn_super_kernel(int nleaf, const KnnBox* __restrict__ leaves, KnnBox* __restrict__ supers)
{
    const int i = blockIdx.x * KNN_FAN + threadIdx.x;
    float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    if (i < nleaf) {
        const KnnBox b = leaves[i];
        lo[0] = b.lo.x; lo[1] = b.lo.y; lo[2] = b.lo.z;
        hi[0] = b.hi.x; hi[1] = b.hi.y; hi[2] = b.hi.z;
    }
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64));
        }
    if (threadIdx.x == 0) {
        supers[blockIdx.x].lo = make_float3(lo[0], lo[1], lo[2]);
        supers[blockIdx.x].hi = make_float3(hi[0], hi[1], hi[2]);
    }
}

// simple_knn.cu:113-124 (distBoxPoint)
__device__ __forceinline__ float box_dist2(const KnnBox& b, float px, float py, float pz)
{
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (px < b.lo.x || px > b.hi.x) dx = fminf(fabsf(px - b.lo.x), fabsf(px - b.hi.x));
    if (py < b.lo.y || py > b.hi.y) dy = fminf(fabsf(py - b.lo.y), fabsf(py - b.hi.y));
    if (pz < b.lo.z || pz > b.hi.z) dz = fminf(fabsf(pz - b.lo.z), fabsf(pz - b.hi.z));
    return dx * dx + dy * dy + dz * dz;
}

// Best-K list in registers, ascending by (distance, index).
template <int K>
struct KBest {
    float d[K];
    uint32_t id[K];
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int j = 0; j < K; j++) {
            d[j] = 3.402823466e38f;
            id[j] = 0xffffffffu;
        }
    }
    __device__ __forceinline__ bool admits(float dist, uint32_t idx) const
    {
        return dist < d[K - 1] || (dist == d[K - 1] && idx < id[K - 1]);
    }
    __device__ __forceinline__ void insert(float dist, uint32_t idx)   // simple_knn.cu:126-142 (updateKBest), with ties by index
    {
#pragma unroll
        for (int j = 0; j < K; j++) {
            const bool sw = dist < d[j] || (dist == d[j] && idx < id[j]);
            const float td = d[j];
            const uint32_t ti = id[j];
            d[j] = sw ? dist : td;
            id[j] = sw ? idx : ti;
            dist = sw ? td : dist;
            idx = sw ? ti : idx;
        }
    }
};

// One thread per query.  SELF: query q IS reference sorted position q (queries == references, processed in Morton order);
// the point itself is skipped when exclude_self is set (distCUDA2) and kept otherwise (pytorch3d: the nearest neighbour
// of a point of the set is the point, distance 0).  !SELF: `queries` are arbitrary points.
// MEAN3: write the mean of the K = 3 distances (distCUDA2) instead of the lists.
template <int K, bool SELF, bool MEAN3>
__global__ void __launch_bounds__(64) knn_query_kernel(int N, const float* __restrict__ queries, int M, const float4* __restrict__ sorted_pts,
                                                       const uint32_t* __restrict__ sorted_codes, const uint32_t* __restrict__ bbox,
                                                       const KnnBox* __restrict__ leaves, const KnnBox* __restrict__ supers,
                                                       int exclude_self, int64_t* __restrict__ out_idx, float* __restrict__ out_d2)
{
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q >= N) return;
    float px, py, pz;
    uint32_t self = 0xffffffffu;
    int centre;
    size_t out_row;
    if (SELF) {
        const float4 p = sorted_pts[q];
        px = p.x; py = p.y; pz = p.z;
        self = __float_as_uint(p.w);
        centre = q;
        out_row = self;
    } else {
        px = queries[3 * (size_t)q]; py = queries[3 * (size_t)q + 1]; pz = queries[3 * (size_t)q + 2];
        const uint32_t code = morton_of(px, py, pz, bbox);
        int lo = 0, hi = M;  // first sorted position whose code is >= the query's
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (sorted_codes[mid] < code) lo = mid + 1;
            else hi = mid;
        }
        centre = min(lo, M - 1);
        out_row = (size_t)q;
    }
    const uint32_t skip = (SELF && exclude_self) ? self : 0xffffffffu;
    KBest<K> best;
    best.init();
    // seed: the neighbours in Morton order (simple_knn.cu:157-162)
    for (int i = max(0, centre - K); i <= min(M - 1, centre + K); i++) {
        const float4 c = sorted_pts[i];
        const uint32_t cid = __float_as_uint(c.w);
        if (cid == skip) continue;
        const float dx = c.x - px, dy = c.y - py, dz = c.z - pz;
        const float dist = dx * dx + dy * dy + dz * dz;
        if (best.admits(dist, cid)) best.insert(dist, cid);
    }
    const int seed_lo = max(0, centre - K), seed_hi = min(M - 1, centre + K);
    const int nleaf = (M + KNN_LEAF - 1) / KNN_LEAF, nsuper = (nleaf + KNN_FAN - 1) / KNN_FAN;
    for (int s = 0; s < nsuper; s++) {
        if (box_dist2(supers[s], px, py, pz) > best.d[K - 1]) continue;
        const int l1 = min(nleaf, (s + 1) * KNN_FAN);
        for (int l = s * KNN_FAN; l < l1; l++) {
            if (box_dist2(leaves[l], px, py, pz) > best.d[K - 1]) continue;
            const int i1 = min(M, (l + 1) * KNN_LEAF);
            for (int i = l * KNN_LEAF; i < i1; i++) {
                if (i >= seed_lo && i <= seed_hi) continue;   // already in the list
                const float4 c = sorted_pts[i];
                const uint32_t cid = __float_as_uint(c.w);
                if (cid == skip) continue;
                const float dx = c.x - px, dy = c.y - py, dz = c.z - pz;
                const float dist = dx * dx + dy * dy + dz * dz;
                if (best.admits(dist, cid)) best.insert(dist, cid);
            }
        }
    }
    if (MEAN3) {
        out_d2[out_row] = (best.d[0] + best.d[1] + best.d[2]) / 3.0f;   // simple_knn.cu:182
    } else {
#pragma unroll
        for (int j = 0; j < K; j++) {
            out_idx[out_row * K + j] = best.id[j] == 0xffffffffu ? (int64_t)-1 : (int64_t)best.id[j];
            out_d2[out_row * K + j] = best.d[j];
        }
    }
}

}  // namespace mirast
