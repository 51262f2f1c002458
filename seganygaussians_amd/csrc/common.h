// common.h -- shared device helpers for the gfx950 rasterizer kernels.
//
// Numeric contract: the per-Gaussian geometry path that feeds the INTEGER results
// (radii, tile rects, tiles_touched, depth key bits) is IEEE binary32 in the reference's source
// order with NO fused contraction: this translation unit is compiled with -ffp-contract=off and
// uses fmaf() only inside the blend kernels' accumulators, which are float-path (tolerance)
// quantities.  Division and sqrt are correctly rounded (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// Timing experiments (pieces of a kernel switched off by a run-time bit mask; results become wrong) exist only in the
// profiling build (-DMI_RAST_PROFILING -> libmi_rast_prof.so).  In the product build MI_ABLATE(bit) is the constant
// `false`, so the kernels carry no dead branches and the `ablate` argument is unused.
#ifdef MI_RAST_PROFILING
#define MI_ABLATE(bit) (((ablate) & (bit)) != 0)
#else
#define MI_ABLATE(bit) (false)
#endif

// Per-wave time stamps of the blend kernels, PROFILING build only (tools/xcd_stamps.py; mi_rast_xcd_stamps in mi_rast.hip): a wave's
// first lane stores the constant 100-MHz clock at its start (with the XCD it runs on) and at its end into ITS OWN two words -- plain
// stores: stamps kept with atomics on sixteen shared words slowed the kernels by half.  The product build compiles the macro to nothing.
#ifdef MI_RAST_PROFILING
constexpr unsigned MI_XCD_LOG_WAVES = 1u << 18;
__device__ unsigned long long g_xcd_log[2 * MI_XCD_LOG_WAVES];   // [2 b] = start << 3 | XCD, [2 b + 1] = end, b = workgroup id
#define MI_XCD_STAMP(AT_END)                                                                                          \
    do {                                                                                                              \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < MI_XCD_LOG_WAVES) {                                               \
            const unsigned long long t_ = __builtin_amdgcn_s_memrealtime();                                           \
            const unsigned x_ = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u; /* HW_REG_XCC_ID */        \
            g_xcd_log[2u * blockIdx.x + ((AT_END) ? 1u : 0u)] = (AT_END) ? t_ : ((t_ << 3) | x_);                     \
        }                                                                                                             \
    } while (0)
#else
#define MI_XCD_STAMP(AT_END) \
    do {                     \
    } while (0)
#endif

namespace mirast {

constexpr int R_SLOTS = 64;        // partial sums of R, one per 128-byte line
constexpr int R_SLOT_STRIDE = 32;  // ints: [0] R, [1] max of ~depth key, [2] max depth key, [R_SLOT_BANDS + b] Gaussians touching band b
constexpr int R_SLOT_BANDS = 4;
constexpr int MAX_BANDS = 24;      // bands of tile rows of the lean count / emit passes (binning.h); one bit per band in band_mask


constexpr int TILE_X = 16;  // CF/cuda_rasterizer/config_contrastive_f.h:16 -- part of the integer contract
constexpr int TILE_Y = 16;  // CF/cuda_rasterizer/config_contrastive_f.h:17
constexpr int TILE_PIXELS = TILE_X * TILE_Y;
constexpr int WAVE = 64;

struct Mat3 {  // glm layout m[col][row]
    float m[3][3];
};

// glm::mat3 * glm::mat3 (CF/third_party/glm/glm/detail/type_mat3x3.inl:486-520): three products
// summed left to right.
__device__ __forceinline__ Mat3 m3mul(const Mat3& A, const Mat3& B)
{
    Mat3 R;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++)
            R.m[c][r] = A.m[0][r] * B.m[c][0] + A.m[1][r] * B.m[c][1] + A.m[2][r] * B.m[c][2];
    return R;
}
__device__ __forceinline__ Mat3 m3transpose(const Mat3& A)
{
    Mat3 R;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++) R.m[c][r] = A.m[r][c];
    return R;
}

// float -> int32: v_cvt_i32_f32 saturates and maps NaN to 0, the same as nvcc's cvt.rzi.s32.f32.
__device__ __forceinline__ int f2i(float f) { return (int)f; }

// CF/cuda_rasterizer/auxiliary.h:41-44 -- double literals: evaluated in binary64, rounded once.
__device__ __forceinline__ float ndc2Pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

// CF/cuda_rasterizer/auxiliary.h:46-56
__device__ __forceinline__ void getRect(float px, float py, int max_radius, uint2& rect_min, uint2& rect_max,
                                        uint32_t gx, uint32_t gy)
{
    rect_min.x = min(gx, (uint32_t)max(0, f2i((px - (float)max_radius) / (float)TILE_X)));
    rect_min.y = min(gy, (uint32_t)max(0, f2i((py - (float)max_radius) / (float)TILE_Y)));
    rect_max.x = min(gx, (uint32_t)max(0, f2i((px + (float)max_radius + (float)TILE_X - (float)1) / (float)TILE_X)));
    rect_max.y = min(gy, (uint32_t)max(0, f2i((py + (float)max_radius + (float)TILE_Y - (float)1) / (float)TILE_Y)));
}

// CF/cuda_rasterizer/auxiliary.h:58-77
__device__ __forceinline__ float3 transformPoint4x3(const float3& p, const float* m)
{
    return make_float3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
                       m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
__device__ __forceinline__ float4 transformPoint4x4(const float3& p, const float* m)
{
    return make_float4(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
                       m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14],
                       m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]);
}
// CF/cuda_rasterizer/auxiliary.h:89-97
__device__ __forceinline__ float3 transformVec4x3Transpose(const float3& p, const float* m)
{
    return make_float3(m[0] * p.x + m[1] * p.y + m[2] * p.z,
                       m[4] * p.x + m[5] * p.y + m[6] * p.z,
                       m[8] * p.x + m[9] * p.y + m[10] * p.z);
}
// CF/cuda_rasterizer/auxiliary.h:107-117
__device__ __forceinline__ float3 dnormvdv(float3 v, float3 dv)
{
    float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    float3 r;
    r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
    r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
    r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
    return r;
}

// Per-view constants every kernel needs, passed by value (kernarg -> SGPRs).  The 4x4 matrices and the
// camera position stay in device memory exactly as the reference passes them; every thread reads them
// at wave-uniform addresses, which gfx950 serves with scalar loads (s_load_dwordx16) from the scalar
// cache -- no host round trip.
struct ViewParams {
    const float* view;    // [16] world->view, column-major for the kernels (scene/cameras.py:62)
    const float* proj;    // [16] full projection
    const float* campos;  // [3] or nullptr
    float tan_fovx, tan_fovy, focal_x, focal_y, scale_modifier;
    int W, H;
    uint32_t grid_x, grid_y;
};

// Exponent of a Gaussian at offset (dx, dy) from its mean, conic pre-scaled to (ha, nb, hc) = (-a/2, -b, -c/2):
// ((-a/2 dx) dx + (-c/2 dy) dy) + (-b dx) dy -- bit-identical to the reference expression -0.5f (a dx dx + c dy dy) - b dx dy
// (forward.cu:338, backward.cu:497; scaling by -1/2 and -1 commutes with every rounding), two VALU fewer.  ONE definition for
// every blend kernel: the forward's alpha >= 1/255 decisions are re-taken by the backward, so both must round this value
// identically.  Deliberately NOT fused: the fmaf form (two instructions fewer) was measured and bought nothing, while its
// 1-ulp differences against an unfused reference flip pairs at the 1/255 threshold -- at full cfg3 size the norm-wise error of
// dL_dmeans2D against the oracle went from 5.6e-7 to 5.8e-6 and the rows outside tolerance from 2.2e-4 to 1.2e-3.
__device__ __forceinline__ float gauss_power(float ha, float nb, float hc, float dx, float dy)
{
    return (ha * dx * dx + hc * dy * dy) + nb * dx * dy;
}

// exp() of the Gaussian exponent: ONE definition for every blend kernel (see gauss_power).  XEXP = true (product default): the
// device library's expf (<= 1 ulp, 13 VALU instructions; 10 as restated below) -- the function the reference's kernels call.  XEXP = false
// (MI_RAST_FAST_EXP): v_exp_f32(power * log2e), 2 instructions, ~5 ulp (the rounding of the product dominates).  The difference
// only matters at the alpha >= 1/255 cut: measured at full cfg3 size, with the 5-ulp exp the rows of dL_dmeans2D outside
// tolerance against the fp64 oracle are 2.2e-4 against the reference's own 1.9e-5, and ALL of that excess comes from pairs put
// on the other side of the cut; with expf the product's statistics are the reference's and n_contrib equals the reference's on
// every pixel.  Cost: +0.03 ms in the forward blend (ALU-bound), nothing measurable in the backward (atomic-bound): 2 % of a
// cfg3 step.
//
// The expf form is the device library's algorithm (ocml expf, f32 with denormals, as hipcc compiles it for gfx950) restated
// with ONE clamp in place of its two range selects (0 below -103.28, inf above 88.72; two v_cmp + two v_cndmask):
// exp(x) = ldexp(exp2((ph - e) + pl), e) with ph + pl = x log2(e) in two-term precision and e = rint(ph), on max(x, -104).
// For every f32 x in [-103.28, 0] the result is bit-identical to expf(x); below that, where expf gives exactly 0, it is 0 or
// the smallest denormal (tools/expf_check.hip runs all 2^32 inputs; tests/test_gpu_parity.py::test_lean_expf_is_the_device_
// expf).  The clamp is needed: without it a hugely negative exponent (a degenerate conic) with a positive rounding residual
// pl ends in ldexp(inf, INT_MIN) = inf.  Where the two differ the blend kernels do not use the value: power > 0 and NaN are
// rejected by their own `power <= 0` test (forward.cu:339), and opacity times a denormal fails the >= 1/255 test like
// opacity times 0.
//
// HYBRID evaluation (the forward blend's product default, blend_fwd_wave.h; include/mi_rast.h: MI_RAST_EXACT_EXP switches it off):
// opacity * exp(power) only has to be the device expf's value BIT FOR BIT where it decides something -- at the alpha >= 1/255
// cut, which the backward re-takes with expf.  Away from the cut a few ulp are float-path noise (alpha to ~6e-7 relative).  So:
// G = v_exp_f32(power * log2e) (2 VALU; relative error <= 8 * 2^-24 * ln 2 + 1 ulp < 1e-6 wherever opacity * G can reach 1/255,
// i.e. power >= -5.55), two compares against the cut widened by +-4e-6 relative, and a GROUP of 16 entries (32/64-channel kernel;
// one entry in the RGB kernel) in which ANY lane falls between the two bounds is re-evaluated with the exact form (a wave-uniform
// branch, taken for ~1e-3 of the groups).  A value outside the band lies on the same side of the cut in both forms, so every
// alpha >= 1/255 decision is the one expf takes: the forward and the backward agree on who blends, always.  ONLY that decision is
// protected: the T < 1e-4 stop test runs on the v_exp_f32 alphas, so where a pixel stops (n_contrib, final_T) can differ from an
// all-expf run -- on ~1e-6 of the pixels of the benchmark scenes, on up to 7.9e-5 of the pixels of an opaque scene of faint
// Gaussians (many pixels ending right at the threshold; the worst case of tools/fuzz_parity.py's sweeps).  A band around the stop
// test, like the one around the cut, would not help: what differs is T, the product of all earlier (1 - alpha), and re-evaluating
// the current group with expf does not restore it.  The default mode is not bit-identical to the reference on the image-state
// fields (MI_RAST_EXACT_EXP is, and that is the mode their parity is checked in).
constexpr float ALPHA_CUT = 1.0f / 255.0f;
constexpr float ALPHA_CUT_LO = (float)((1.0 / 255.0) * (1.0 - 4e-6));
constexpr float ALPHA_CUT_HI = (float)((1.0 / 255.0) * (1.0 + 4e-6));
enum ExpMode : int { EXP_FAST = 0, EXP_EXACT = 1, EXP_HYBRID = 2 };
__device__ __forceinline__ float gauss_exp_fast(float power) { return __builtin_amdgcn_exp2f(power * 0x1.715476p+0f); }

template <bool XEXP>
__device__ __forceinline__ float gauss_exp(float power)
{
    if constexpr (XEXP) {
        const float x = __builtin_fmaxf(power, -104.0f);
        const float ph = x * 0x1.715476p+0f;
        float pl = __builtin_fmaf(x, 0x1.715476p+0f, -ph);
        pl = __builtin_fmaf(x, 0x1.4ae0bep-26f, pl);
        const float e = __builtin_rintf(ph);
        return __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f((ph - e) + pl), (int)e);
    } else {
        return __expf(power);
    }
}

// ---- workgroup -> tile ----------------------------------------------------------------------------------------------------
// Workgroup b of a 1-D grid runs on XCD b % 8 (tools/xcc_probe.hip), and every XCD has its own L2.  Neighbouring tiles share most
// of their Gaussians.  The backward blend therefore gives every XCD a CONTIGUOUS run of tiles (row major:
// an eighth of the image, a band of whole tile rows): the records and feature rows neighbouring tiles share come through one L2,
// and the gradient atomics of a Gaussian hit lines that L2 already holds instead of lines that travel between L2s -- backward
// blend 0.647 -> 0.565 ms on cfg3.
//   static:              workgroup id = 8 j + x is the j-th tile of XCD x's run.  The kernel time is then the busiest XCD's.
//   dynamic (xcd_grab_runs): the backward blend's cost per tile is the scene's density, and a static split would hand one XCD the
//                        empty sky.  There the second half of every run is a QUEUE: a wave takes its item from the queue of
//                        the XCD it runs on (one returning atomic on that XCD's counter; HW_REG_XCC_ID says which) and, when
//                        that is empty, from the next XCD's.  As many such waves as queued items, each takes exactly one:
//                        every item is taken exactly once.  (All of a run queued: 0.597 ms instead of 0.565 -- the counter's
//                        answer is a round trip in front of every wave's work; persistent waves that request the next item
//                        ahead of time: 0.63, spills and an uneven tail.)
__host__ __device__ __forceinline__ uint32_t xcd_run_start(uint32_t x, uint32_t n) { return (uint32_t)(((uint64_t)x * n) >> 3); }
constexpr int XCD_QUEUE_STRIDE = 16;  // uint32 words between the eight queue counters (one 64-byte line each)
constexpr uint32_t XCD_QUEUE_DIV = 2;  // the queued part of a run is its last 1 / XCD_QUEUE_DIV
// WORK-balanced runs (round 5).  Equal tile counts per XCD are equal TIME only on a scene whose density does not vary over the image:
// on the second synthetic law (opaque surfaces + floaters; real scenes are of this kind) the slowest XCD ended 20-23 % behind the
// mean in both blend kernels (profiles/r04_xcd_balance.md), and nothing lets a fast XCD take more -- the dispatcher hands workgroup b
// to XCD b % 8 whatever the XCDs' progress.  The range scan (binning.h: tile_ranges_kernel) therefore cuts the row-major tile
// sequence into eight contiguous runs of equal MODELLED work, sum(min(list length, 768) + 128) -- the list LENGTH alone is the wrong
// weight: an opaque surface ends a long list early --, and leaves the nine boundaries in the image buffer for both blend kernels
// (profiles/r05_xcd_balance.md: cfg3s +7 %).  A knob (mi_rast.hip: BWD_RUNS_FROM_WALKS, off) replaces the backward's boundaries by
// ones cut at equal sums of what the forward really WALKED (tile_nsurv + XCD_TILE_WEIGHT; run_bounds_from_walks_kernel, one more
// launch behind the forward blend).  A run holds at most XCD_MAX_RUN_FACTOR times the equal share: that bounds the grid the
// (stateless) backward launches -- ids beyond a run's length exit at once.
constexpr uint32_t XCD_TILE_WEIGHT = 128;     // a tile's fixed cost (four waves' start-up) in list entries: 32 / 128 / 256 measured, round 4
constexpr uint32_t XCD_MAX_RUN_FACTOR = 2;
struct XcdRuns {
    uint32_t b[9];   // run x = tiles [b[x], b[x + 1])
};
__host__ __device__ inline uint32_t xcd_max_run(uint32_t n)
{
    const uint32_t m = XCD_MAX_RUN_FACTOR * ((n + 7u) >> 3);
    return m < n ? m : n;
}
__host__ __device__ inline uint32_t xcd_static_len_max(uint32_t n)
{
    const uint32_t longest = xcd_max_run(n);
    return longest - longest / XCD_QUEUE_DIV;
}
__host__ __device__ inline uint32_t xcd_queued_tiles_max(uint32_t n) { return n / XCD_QUEUE_DIV; }   // >= sum over the runs of len / DIV
__device__ __forceinline__ XcdRuns xcd_load_runs(const uint32_t* __restrict__ bounds)
{
    XcdRuns r;
#pragma unroll
    for (int k = 0; k < 9; k++) r.b[k] = bounds[k];   // wave-uniform address: scalar loads
    return r;
}
// Clamps eight proposed boundaries (s_bound[1..7], any values) so that every run holds at most xcd_max_run(n) tiles and the runs
// partition [0, n): one thread.
__device__ __forceinline__ void xcd_clamp_runs(uint32_t* s_bound, uint32_t n)
{
    const uint32_t maxrun = xcd_max_run(n);
    s_bound[0] = 0u;
    s_bound[8] = n;
    for (uint32_t k = 1; k < 8u; k++) {
        const uint32_t prev = s_bound[k - 1];
        const uint32_t need = (8u - k) * maxrun;               // what the runs behind boundary k can hold at most
        const uint32_t lo = max(prev, n > need ? n - need : 0u);
        const uint32_t hi = min(n, prev + maxrun);
        s_bound[k] = min(max(s_bound[k], lo), hi);
    }
}
// xcd_grab over given runs: as above; takers beyond the queued items of all eight runs (the grid is sized for the longest runs the
// clamp allows) find every queue empty after eight probes and get 0xFFFFFFFF.
__device__ __forceinline__ uint32_t xcd_grab_runs(uint32_t* __restrict__ counters, const XcdRuns& runs, uint32_t per)
{
    const uint32_t x0 = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;  // HW_REG_XCC_ID[3:0]
    const bool first = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u;
    for (uint32_t k = 0; k < 8u; k++) {
        const uint32_t x = (x0 + k) & 7u;
        uint32_t start = runs.b[0], end = runs.b[1];
#pragma unroll
        for (uint32_t j = 1; j < 8u; j++) {
            start = x == j ? runs.b[j] : start;
            end = x == j ? runs.b[j + 1] : end;
        }
        const uint32_t len = end - start;
        uint32_t got = 0;
        if (first) got = atomicAdd(&counters[x * XCD_QUEUE_STRIDE], 1u);
        got = __builtin_amdgcn_readfirstlane(got);
        if (got < (len / XCD_QUEUE_DIV) * per) return (start + len - len / XCD_QUEUE_DIV) * per + got;
    }
    return 0xFFFFFFFFu;
}

// ---- wave64 helpers -------------------------------------------------------------------------
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// Sum over the 64 lanes of a wave; the result is valid in EVERY lane.
// quad_perm xor1 / xor2, row_half_mirror, row_mirror stay inside a 16-lane DPP row (VALU, no LDS);
// the two cross-row steps use ds_swizzle-free __shfl_xor (ds_bpermute).
__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror
    v += dpp_mov<0x140>(v);  // row_mirror
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

}  // namespace mirast
