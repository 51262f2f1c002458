// mi_rast.hip -- C-ABI implementation (include/mi_rast.h) and host orchestration of the gfx950
// rasterizer kernels.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off
// -munsafe-fp-atomics -fPIC -shared (see seganygaussians_amd/build.py).  No torch, no pybind.
#include "../../include/mi_rast.h"
#include "../../include/mi_knn_smooth.h"
#include "../../include/mi_knn.h"
#include "../../include/mi_contrastive.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "binning.h"
#include "knn_smooth.h"
#include "knn.h"
#include "blend_bwd.h"
#include "blend_bwd_wave.h"   // (the profiling build compiles the PRODUCT kernel too: what tools/ measure is what ships; the ablation
                              // masks and rejected variants of rounds 2-5 are a record under tools/experiments/, compiled nowhere)
#include "blend_bwd_feat.h"   // features-only backward (MI_RAST_BWD_FEATURES_ONLY): one wave per half tile
#include "blend_fwd.h"
#include "blend_fwd_wave.h"
#ifdef MI_RAST_PROFILING
#include "blend_fwd_x3.h"   // round 2's tile-batched bf16x3 forward: A/B comparisons only (MI_RAST_TILE_FWD)
#endif
#include "common.h"
#include "contrastive.h"
#include "geometry.h"

using namespace mirast;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return fail(MI_RAST_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));           \
    } while (0)

// Mirrors CHECK_CUDA (CF/cuda_rasterizer/auxiliary.h:166-173): with debug set, synchronise after each
// stage and surface the error; without it only launch errors are caught.
#define STAGE_CHECK(name)                                                                              \
    do {                                                                                               \
        hipError_t _e = hipGetLastError();                                                             \
        if (_e == hipSuccess && debug) _e = hipStreamSynchronize(stream);                              \
        if (_e != hipSuccess)                                                                          \
            return fail(MI_RAST_ERR_HIP, std::string("[HIP ERROR] in stage ") + name + ": " + hipGetErrorString(_e)); \
    } while (0)

constexpr size_t ALIGN = 256;
inline size_t align_up(size_t v) { return (v + ALIGN - 1) & ~(ALIGN - 1); }

struct Carver {
    size_t off = 0;
    size_t take(size_t bytes)
    {
        size_t o = off;
        off = align_up(off + bytes);
        return o;
    }
};

// ---- host-side scratch for the num_rendered read-back: pinned words + two events per (host thread, device) ----
constexpr int MAX_DEVICES = 64;
struct HostSync {
    int* pinned = nullptr;   // R partial sums, then {R, longest tile list} of the range scan
    int* pinned_dev = nullptr;  // the same buffer as the kernels address it (they store into it directly)
    uint64_t* fp = nullptr;     // partial fingerprints (mi_rast_fingerprint), [MI_FP_MAX][FP_BLOCKS]
    uint64_t* fp_dev = nullptr;
    hipEvent_t ev = nullptr;
    hipEvent_t ev2 = nullptr;
    bool ok = false;
    bool init()
    {
        if (ok) return true;
        if (hipHostMalloc((void**)&pinned, (R_SLOTS * R_SLOT_STRIDE + 16) * sizeof(int), hipHostMallocMapped) != hipSuccess) return false;
        if (hipHostGetDevicePointer((void**)&pinned_dev, pinned, 0) != hipSuccess) return false;
        if (hipHostMalloc((void**)&fp, (size_t)MI_FP_MAX * FP_BLOCKS * sizeof(uint64_t), hipHostMallocMapped) != hipSuccess) return false;
        if (hipHostGetDevicePointer((void**)&fp_dev, fp, 0) != hipSuccess) return false;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return false;
        if (hipEventCreateWithFlags(&ev2, hipEventDisableTiming) != hipSuccess) return false;
        ok = true;
        return true;
    }
    // (a host thread's scratch goes with the thread; errors are ignored: at process exit the runtime may be gone already)
    ~HostSync()
    {
        if (ev) (void)hipEventDestroy(ev);
        if (ev2) (void)hipEventDestroy(ev2);
        if (pinned) (void)hipHostFree(pinned);
        if (fp) (void)hipHostFree(fp);
    }
};
thread_local HostSync g_host_sync_tl[MAX_DEVICES];
thread_local int g_last_longest_run[MAX_DEVICES];   // of this thread's last forward per device (mi_rast_last_longest_run)

int current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return -1;
    return dev;
}

// ---- per-stage profiling (bench.py) -----------------------------------------------------------
// A measurement aid, not part of the rendering state: one process-wide switch, one event set per device (created on
// first use on that device).  Toggle only while no call is in flight (include/mi_rast.h).
std::atomic<bool> g_profile{false};
struct ProfileEvents {
    hipEvent_t ev[MI_STAGE_COUNT][2];
    bool created = false;
    bool used[MI_STAGE_COUNT] = {};
};
ProfileEvents g_prof[MAX_DEVICES];
std::mutex g_prof_mutex;

ProfileEvents* profile_events(int dev)
{
    if (dev < 0) return nullptr;
    ProfileEvents& pe = g_prof[dev];
    if (!pe.created) {
        std::lock_guard<std::mutex> lock(g_prof_mutex);
        if (!pe.created) {
            for (int i = 0; i < MI_STAGE_COUNT; i++)
                for (int k = 0; k < 2; k++)
                    if (hipEventCreate(&pe.ev[i][k]) != hipSuccess) return nullptr;
            pe.created = true;
        }
    }
    return &pe;
}

struct StageTimer {
    hipStream_t s;
    int stage;
    ProfileEvents* pe;
    StageTimer(hipStream_t s_, int stage_) : s(s_), stage(stage_), pe(nullptr)
    {
        if (g_profile.load(std::memory_order_relaxed)) pe = profile_events(current_device());
        if (pe) (void)hipEventRecord(pe->ev[stage][0], s);
    }
    ~StageTimer()
    {
        if (pe) {
            (void)hipEventRecord(pe->ev[stage][1], s);
            pe->used[stage] = true;
        }
    }
};

ViewParams make_view(const float* view_d, const float* proj_d, const float* campos_d, float tan_fovx, float tan_fovy,
                     float scale_modifier, int W, int H)
{
    ViewParams vp;
    vp.view = view_d;
    vp.proj = proj_d;
    vp.campos = campos_d;
    vp.tan_fovx = tan_fovx;
    vp.tan_fovy = tan_fovy;
    // CF/cuda_rasterizer/rasterizer_impl.cu:222-223
    vp.focal_y = H / (2.0f * tan_fovy);
    vp.focal_x = W / (2.0f * tan_fovx);
    vp.scale_modifier = scale_modifier;
    vp.W = W;
    vp.H = H;
    vp.grid_x = (W + TILE_X - 1) / TILE_X;
    vp.grid_y = (H + TILE_Y - 1) / TILE_Y;
    return vp;
}

struct GeomPtrs {
    float* depths;
    float2* means2D;
    float4* conic_opacity;
    float* cov3D;
    float* rgb;
    uint8_t* clamped;
    uint32_t* tiles_touched;
    uint32_t* depth_key;
    BlendRec* index_rec;  // [P] {mean, id, radius, conic + opacity} in index order (written by the preprocess pass for visible Gaussians)
    int* cull_counter;    // one word: Gaussians culled although `prefiltered` was set
    unsigned long long* band_bits;  // [MAX_BANDS][ceil(P / 64)]: per band of tile rows, one bit per Gaussian whose rect reaches it (binning.h)
    float* bwd_pack;  // [P,8] packed per-Gaussian field gradients (backward scratch)
};
struct ImgPtrs {
    float* final_T;
    uint32_t* n_contrib;
    uint2* ranges;
    uint32_t* tile_consumed;  // per tile: list entries the forward blend consumed (counter E of SURVEY.md 8d)
    uint32_t* tile_count;
    uint32_t* tile_cursor;
    int* num_rendered;
    uint32_t* tile_nsurv;   // per tile: blend-list entries the forward walked
    uint32_t* run_bounds;   // [9]: the blend kernels' XCD runs of tiles (common.h: XcdRuns): from the range scan (equal counts, or equal
                            // modelled work), replaced by run_bounds_from_walks_kernel behind the forward blend when that is switched on
    uint32_t longest_run;   // host copy of the longest of those runs (forward only: read from the pinned words at the ev2 wait)
};
struct BinPtrs {
    uint2* entries;      // per overlap: {depth bits, Gaussian id | quadrant mask << 28}, bucketed by tile (unsorted inside a tile)
    uint2* scratch;      // ping-pong buffer for tiles too long for the LDS sort
    uint32_t* blend_list;  // per tile at range.x, in depth order: Gaussian id | quadrant mask << 28 (binning.h: emit_blend_list); full
                           // lists: the low 28 bits are the reference's point_list
};

GeomPtrs geom_from(char* base, int P)
{
    size_t off[MI_GEOM_NFIELDS];
    mi_rast_geometry_layout(P, off);
    GeomPtrs g;
    g.depths = (float*)(base + off[MI_GEOM_DEPTHS]);
    g.means2D = (float2*)(base + off[MI_GEOM_MEANS2D]);
    g.conic_opacity = (float4*)(base + off[MI_GEOM_CONIC_OPACITY]);
    g.cov3D = (float*)(base + off[MI_GEOM_COV3D]);
    g.rgb = (float*)(base + off[MI_GEOM_RGB]);
    g.clamped = (uint8_t*)(base + off[MI_GEOM_CLAMPED]);
    g.tiles_touched = (uint32_t*)(base + off[MI_GEOM_TILES_TOUCHED]);
    g.depth_key = (uint32_t*)(base + off[MI_GEOM_DEPTH_KEY]);
    g.index_rec = (BlendRec*)(base + off[MI_GEOM_INDEX_REC]);
    g.cull_counter = (int*)(base + off[MI_GEOM_CULL_COUNTER]);
    g.band_bits = (unsigned long long*)(base + off[MI_GEOM_BAND_BITS]);
    g.bwd_pack = (float*)(base + off[MI_GEOM_BWD_PACK]);
    return g;
}
ImgPtrs img_from(char* base, int W, int H)
{
    size_t off[MI_IMG_NFIELDS];
    mi_rast_image_layout(W, H, off);
    ImgPtrs m;
    m.final_T = (float*)(base + off[MI_IMG_FINAL_T]);
    m.n_contrib = (uint32_t*)(base + off[MI_IMG_N_CONTRIB]);
    m.ranges = (uint2*)(base + off[MI_IMG_RANGES]);
    m.tile_consumed = (uint32_t*)(base + off[MI_IMG_TILE_CONSUMED]);
    m.tile_count = (uint32_t*)(base + off[MI_IMG_TILE_COUNT]);
    m.tile_cursor = (uint32_t*)(base + off[MI_IMG_TILE_CURSOR]);
    m.num_rendered = (int*)(base + off[MI_IMG_NUM_RENDERED]);
    m.tile_nsurv = (uint32_t*)(base + off[MI_IMG_TILE_NSURV]);
    m.run_bounds = reinterpret_cast<uint32_t*>(m.num_rendered + R_SLOTS * R_SLOT_STRIDE + NR_RUN_BOUNDS);
    m.longest_run = 0;
    return m;
}
BinPtrs bin_from(char* base, int R)
{
    size_t off[MI_BIN_NFIELDS];
    mi_rast_binning_layout(R, off);
    BinPtrs b;
    b.entries = (uint2*)(base + off[MI_BIN_ENTRIES]);
    b.scratch = (uint2*)(base + off[MI_BIN_SCRATCH]);
    b.blend_list = (uint32_t*)(base + off[MI_BIN_BLEND_LIST]);
    return b;
}

// Timing experiments exist only in the profiling build (-DMI_RAST_PROFILING, seganygaussians_amd/build.py:
// libmi_rast_prof.so): MI_RAST_ABLATE / MI_RAST_ABLATE_FWD=<bitmask> disable pieces of the blend kernels (results become
// wrong).  The product build compiles none of it: no getenv, no run-time switches inside the kernels (common.h: MI_ABLATE).
#ifdef MI_RAST_PROFILING
int ablate_env(const char* name)
{
    const char* ab = getenv(name);
    return ab ? atoi(ab) : 0;
}
#else
constexpr int ablate_env(const char*) { return 0; }
#endif

// One-time opt-in to > 64 KB of dynamic LDS (gfx950: 160 KB per workgroup), per device.
std::atomic<bool> g_attr_set[MAX_DEVICES];
std::mutex g_attr_mutex;

// RGB (+ the DEPTH variant's mask / depth planes), or any multiple of 16 feature channels up to 256: wider features are blended
// in channel blocks of 64 / 32 / 16 (every pass re-evaluates alpha and T; the contraction over channels is linear, so the
// backward's per-block geometry gradients add up in the packed record).  The reference compiles ONE NUM_CHANNELS into its
// kernels (CF/cuda_rasterizer/config_contrastive_f.h:15).
constexpr int MAX_CHANNELS = 256;
constexpr int MAX_CHANNEL_BLOCKS = MAX_CHANNELS / 16;
// bytes of the geometry buffer's bwd_pack field: the packed per-Gaussian field gradients + the backward's work-queue counters (one set
// of eight per channel block); zero when mi_rast_backward starts its blend pass
inline size_t bwd_pack_bytes(int P)
{
    return (size_t)(P > 0 ? P : 1) * 8 * sizeof(float) + MAX_CHANNEL_BLOCKS * 8 * XCD_QUEUE_STRIDE * sizeof(uint32_t);
}
// Any width from 1 to MAX_CHANNELS (the reference compiles ANY NUM_CHANNELS into its kernels, config_contrastive_f.h:15): blocks of
// 64 / 32 / 16 channels, the last one possibly PARTIAL -- fewer than 16 real channels, zeros in the MFMA operands behind them, as
// RGB has always been rendered (3 of 16).
bool channels_supported(int c) { return c >= 1 && c <= MAX_CHANNELS; }
// next block of a feature with `rem` channels left (rem < 16: a 16-channel block of which `rem` exist)
int channel_block(int rem) { return rem >= 64 ? 64 : (rem >= 32 ? 32 : 16); }

// How the blend kernels' tiles are dealt to the eight XCDs (common.h, blend_fwd_wave.h: fwd_wave_item, binning.h: tile_ranges_kernel).
// Product constants; the profiling build reads overrides from the environment for A/B runs (MI_RAST_FWD_RUNS, MI_RAST_RUN_CAP,
// MI_RAST_RUN_FIX, MI_RAST_BWD_SCAN).
// Measured in round 5 (profiles/r05_xcd_balance.md; cfg3s = density varying over the image, cfg3 = uniform): equal tile counts 447 / 844
// views/s; this model 493 / 835 (cap 384 ... 1024, fix 32 ... 128 tried: 768 / 128 best; no cap -- the list length -- 428: a long list
// on an opaque surface is a SHORT walk); m = 4 interleaved equal-count runs in the forward + the walk scan for the backward 497 / 831.
constexpr int FWD_RUNS_PER_XCD = 0;   // forward: m interleaved runs of equal tile counts per XCD; 0: one run per XCD, boundaries from the range scan
constexpr int RUN_MODEL_CAP = 768;    // range scan: XCD runs of equal sum(min(list length, cap) + fix); 0: equal tile counts
constexpr int RUN_MODEL_FIX = 128;
// Forward of a view to be differentiated: the backward's runs from what the forward really WALKED (run_bounds_from_walks_kernel, one more
// 9-us launch behind the forward blend).  0 (product): never; 1: always; 2: when the range scan's model says the scene's density varies
// over the image (its longest run exceeds the equal share by more than an eighth).  Round 6 (tools/xcd_stamps.py: per-XCD finish times
// from per-wave stamps): with the model runs the backward's XCDs finish within 6-7 % of each other on the uniform law (cfg3) and within
// 21 % on the second law (cfg3s; the model balances the FORWARD: 11-13 %); the exact walks take 1.6 % off that backward (1.099 -> 1.081 ms)
// and put their launch on the forward: 530.2 / 529.1 / 530.8 views/s for 0 / 2 / 1 -- nothing, so it stays a knob.
constexpr int BWD_RUNS_FROM_WALKS = 0;
inline int knob(const char* name, int dflt)
{
#ifdef MI_RAST_PROFILING
    const char* e = getenv(name);
    if (e) return atoi(e);
#endif
    (void)name;
    return dflt;
}
inline uint32_t fwd_runs_per_xcd() { return (uint32_t)std::min(16, std::max(0, knob("MI_RAST_FWD_RUNS", FWD_RUNS_PER_XCD))); }
inline uint32_t longest_of(const int* bounds, uint32_t ntiles)
{
    uint32_t longest = 0;
    for (int x = 0; x < 8; x++) {
        const uint32_t a = (uint32_t)bounds[x], b = (uint32_t)bounds[x + 1];
        longest = std::max(longest, b >= a ? b - a : 0u);
    }
    return std::min(std::max(longest, (ntiles + 7u) >> 3), xcd_max_run(ntiles));
}

// Stages shared by forward and mask_forward: CF/cuda_rasterizer/rasterizer_impl.cu:246-317.
int geometry_and_binning(mi_rast_resize_fn geometry_buffer, void* geometry_user, mi_rast_resize_fn binning_buffer,
                         void* binning_user, mi_rast_resize_fn image_buffer, void* image_user, int P, int D, int M,
                         int W, int H, const float* means3D, const float* shs, int colors_given,
                         const float* opacities, const float* scales, const float* rotations,
                         const float* cov3D_precomp, const ViewParams& vp, int prefiltered, int* radii, int debug,
                         int flags, hipStream_t stream, GeomPtrs& geom, ImgPtrs& img, BinPtrs& bin, int* num_rendered)
{
    const int dev = current_device();
    if (dev < 0) return fail(MI_RAST_ERR_HIP, "hipGetDevice failed (or device ordinal >= 64)");
    HostSync& g_host_sync = g_host_sync_tl[dev];
    const int g_ablate_fwd = ablate_env("MI_RAST_ABLATE_FWD");
    (void)g_ablate_fwd;
    size_t goff[MI_GEOM_NFIELDS], ioff[MI_IMG_NFIELDS];
    const size_t geom_size = mi_rast_geometry_layout(P, goff);
    char* geom_base = geometry_buffer(geom_size, geometry_user);
    if (!geom_base) return fail(MI_RAST_ERR_ALLOC, "geometry buffer callback returned NULL");
    geom = geom_from(geom_base, P);
    const size_t img_size = mi_rast_image_layout(W, H, ioff);
    char* img_base = image_buffer(img_size, image_user);
    if (!img_base) return fail(MI_RAST_ERR_ALLOC, "image buffer callback returned NULL");
    img = img_from(img_base, W, H);

    int* cull_counter = geom.cull_counter;
    if (prefiltered) HIP_TRY(hipMemsetAsync(cull_counter, 0, sizeof(int), stream));
    const int ntiles = (int)(vp.grid_x * vp.grid_y);
    if (P >= (1 << ID_BITS)) return fail(MI_RAST_ERR_INVALID, "more than 2^28 Gaussians");
    if (vp.grid_x > 1023u || vp.grid_y > 2047u)
        return fail(MI_RAST_ERR_INVALID, "image too large: more than 1023 tiles across or 2047 tiles down");
    // FULL lists (parity tests): images with more tiles than one launch of the full count / emit passes has LDS counters for are walked
    // in bands of tile rows by the host (grid_x + 2: the count pass keeps band_rows (+ 1) rows of an odd stride >= grid_x + 1)
    const uint32_t band_rows = std::min<uint32_t>(vp.grid_y, std::max<uint32_t>(1u, (uint32_t)BIN_MAX_TILES / (vp.grid_x + 2u) - 1u));
    const uint32_t nbands = (vp.grid_y + band_rows - 1) / band_rows;
    // LEAN lists: <= MAX_BANDS device-side bands of lean_band_h tile rows, every workgroup of the count / emit passes in one of them
    uint32_t lean_band_h = 1, lean_nbands = 1;
    if (!band_geometry(vp.grid_x, vp.grid_y, SPAN_MAX_HEAD_WORDS, lean_band_h, lean_nbands))
        return fail(MI_RAST_ERR_INVALID, "image too large for the banded binning passes (more than 24 bands of tile rows at this width)");
    {
        StageTimer t(stream, MI_STAGE_PREPROCESS);
        HIP_TRY(hipMemsetAsync(img.num_rendered, 0, R_SLOTS * R_SLOT_STRIDE * sizeof(int), stream));
        // (the per-tile entry totals are zeroed by the preprocess kernel's first threads; fewer Gaussians than tiles: by a fill)
        if (P < ntiles) HIP_TRY(hipMemsetAsync(img.tile_cursor, 0, (size_t)ntiles * sizeof(uint32_t), stream));
        // SH colours: the workgroup's coefficient rows are staged in LDS (geometry.h), 256 rows of 3 M + 4 floats
        const size_t sh_lds = colors_given ? 0 : (size_t)256 * (3 * (size_t)M + 4) * sizeof(float);
        if (sh_lds > 64 * 1024) return fail(MI_RAST_ERR_INVALID, "too many SH coefficients per Gaussian (at most 16: degree 3)");
        hipLaunchKernelGGL(preprocess_fwd_kernel, dim3((P + 255) / 256), dim3(256), sh_lds, stream, P, D, M, means3D, scales,
                           rotations, opacities, shs, geom.clamped, cov3D_precomp, colors_given, vp, radii,
                           geom.means2D, geom.depths, geom.cov3D, geom.rgb, geom.conic_opacity, geom.tiles_touched,
                           geom.depth_key, geom.index_rec, img.num_rendered, prefiltered, cull_counter, geom.band_bits, lean_band_h, lean_nbands,
                           img.tile_cursor, (uint32_t)ntiles);
    }
    STAGE_CHECK("preprocess");
    if (prefiltered) {
        int culled = 0;
        HIP_TRY(hipMemcpyAsync(&culled, cull_counter, sizeof(int), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (culled)
            return fail(MI_RAST_ERR_INVALID, "Point is filtered although prefiltered is set. This shouldn't happen!");
    }
    if (!g_attr_set[dev].load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> lock(g_attr_mutex);
        if (!g_attr_set[dev].load(std::memory_order_relaxed)) {
            const int max_lds = (int)((BIN_MAX_TILES + 11 * 1024 + 16) * sizeof(uint32_t));
            HIP_TRY(hipFuncSetAttribute((const void*)bin_ranks_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
            HIP_TRY(hipFuncSetAttribute((const void*)bin_ranks_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
            HIP_TRY(hipFuncSetAttribute((const void*)bin_count_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
            // (160 KB per workgroup less the kernels' static words: the band plan)
            HIP_TRY(hipFuncSetAttribute((const void*)bin_spans_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
            HIP_TRY(hipFuncSetAttribute((const void*)bin_spans_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
            HIP_TRY(hipFuncSetAttribute((const void*)tile_ranges_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (BIN_MAX_TILES_TOTAL + 1) * (int)sizeof(uint32_t)));
            HIP_TRY(hipFuncSetAttribute((const void*)run_bounds_from_walks_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (BIN_MAX_TILES_TOTAL + 1) * (int)sizeof(uint32_t)));
            g_attr_set[dev].store(true, std::memory_order_release);
        }
    }
    // R is known after the preprocess pass; the count pass stores its partial sums into the host's pinned
    // buffer, and the host reads them while the scans run.
    if (!g_host_sync.init()) return fail(MI_RAST_ERR_HIP, "cannot allocate pinned host buffer / event");
    // Full lists (the reference's point_list and full-list positions) are materialised on request -- the reference's
    // `debug` flag or MI_RAST_FULL_LISTS -- ; otherwise only the overlaps that pass the cull are listed.
    const bool nocull = (flags & MI_RAST_NO_CULL) != 0;
    const bool full = debug != 0 || nocull || (flags & MI_RAST_FULL_LISTS) != 0;
    // full lists: <= 256 workgroups of 1024 threads, each a slice of the Gaussians (64-index chunks dealt round robin); lean lists:
    // workgroups dealt to the bands in proportion to the bands' Gaussian counts (binning.h: band_plan), at least one per band
    int nwg = full ? bin_workgroups(P) : std::min(knob("MI_RAST_LEAN_NWG", BIN_LEAN_WG), (int)lean_nbands + (P + 1023) / 1024);
#ifdef MI_RAST_PROFILING
    if (ablate_env("MI_RAST_NWG") > 0) nwg = std::min(nwg, ablate_env("MI_RAST_NWG"));
#endif
    const size_t band_tiles = (size_t)band_rows * vp.grid_x;
    const size_t bin_lds = ((size_t)((band_tiles + 3) & ~(size_t)3) + 3 * 1024 + 16) * sizeof(uint32_t);
    {
        // count pass over slices, scan over (tile, slice), scan over tiles -> ranges (binning.h)
        StageTimer t(stream, MI_STAGE_TILE_SCAN);
        const size_t cnt_lds = (size_t)(band_rows + 1) * count_grid_stride(vp.grid_x) * sizeof(int);
        for (uint32_t b = 0; full && b < nbands; b++) {
            const uint32_t by0 = b * band_rows, by1 = std::min(vp.grid_y, by0 + band_rows);
            if (full)
                hipLaunchKernelGGL(bin_count_kernel, dim3(nwg), dim3(BIN_THREADS), cnt_lds, stream, P, geom.index_rec, geom.depth_key,
                                   img.tile_count, vp.grid_x, vp.grid_y, by0, by1, (const int*)img.num_rendered,
                                   b == 0 ? g_host_sync.pinned_dev : (int*)nullptr);
        }
        if (!full)
            hipLaunchKernelGGL(bin_spans_kernel<false>, dim3(nwg), dim3(1024), span_lds_bytes((size_t)lean_band_h * count_grid_stride(vp.grid_x)),
                               stream, P, geom.index_rec, geom.depth_key, geom.band_bits, img.tile_count, img.tile_cursor, (const uint2*)nullptr, (uint2*)nullptr,
                               vp.grid_x, vp.grid_y, lean_band_h, lean_nbands, (const int*)img.num_rendered, g_host_sync.pinned_dev, g_ablate_fwd);
        HIP_TRY(hipEventRecord(g_host_sync.ev, stream));
        // (one launch for both scans -- every workgroup scans its tiles over the slices, the last one to finish scans the totals
        // behind a device-scope counter and agent-scope fences -- was built and measured in round 4: tile scan 0.058 -> 0.060 ms
        // on cfg3; the fences and the serial tail inside the kernel cost more than the launch they save)
        // full lists: scan over (tile, slice); lean lists: the count pass left the tile totals and every workgroup's offset itself
        if (full)
            hipLaunchKernelGGL(scan_partials_kernel, dim3((ntiles + 63) / 64), dim3(1024), 0, stream, ntiles, nwg,
                               img.tile_count, img.tile_cursor);
        const uint32_t run_cap = !(flags & MI_RAST_EQUAL_RUNS) ? (uint32_t)std::max(0, knob("MI_RAST_RUN_CAP", RUN_MODEL_CAP)) : 0u;
        const uint32_t run_fix = (uint32_t)std::max(0, knob("MI_RAST_RUN_FIX", RUN_MODEL_FIX));
        hipLaunchKernelGGL(tile_ranges_kernel, dim3(1), dim3(1024),
                           ((size_t)std::min(ntiles, BIN_MAX_TILES_TOTAL) + 1) * sizeof(uint32_t), stream, ntiles, img.tile_cursor,
                           img.ranges, img.num_rendered + R_SLOTS * R_SLOT_STRIDE, (const int*)img.num_rendered,
                           g_host_sync.pinned_dev + R_SLOTS * R_SLOT_STRIDE, img.tile_consumed, img.tile_nsurv, img.run_bounds, run_cap, run_fix);
    }
    STAGE_CHECK("tile scan");
    HIP_TRY(hipEventRecord(g_host_sync.ev2, stream));
    // rasterizer_impl.cu:280-281: the host needs num_rendered to size the binning buffer.  We wait only for
    // the copy (event), not for the work queued behind it.
    HIP_TRY(hipEventSynchronize(g_host_sync.ev));
    long long Rsum = 0;
    for (int k = 0; k < R_SLOTS; k++) Rsum += g_host_sync.pinned[k * R_SLOT_STRIDE];
    if (Rsum > 0x7fffffffll) return fail(MI_RAST_ERR_INVALID, "more than 2^31 tile overlaps");
    const int R = (int)Rsum;
    *num_rendered = R;

    size_t boff[MI_BIN_NFIELDS];
    const size_t bin_size = mi_rast_binning_layout(R, boff);
    char* bin_base = binning_buffer(bin_size, binning_user);
    if (!bin_base) return fail(MI_RAST_ERR_ALLOC, "binning buffer callback returned NULL");
    bin = bin_from(bin_base, R);
    const bool verify = !full && (flags & MI_RAST_VERIFY_LISTS) != 0;
    if (R > 0) {
        if (verify) HIP_TRY(hipMemsetAsync(bin.entries, 0, (size_t)R * sizeof(uint2), stream));
#ifdef MI_RAST_PROFILING
        // timing experiments that leave list slots unwritten (tools/abl_emit.sh): zeros instead of stale ids, so that the blends stay in bounds
        if (g_ablate_fwd & ((1 << 16) | (1 << 17) | (1 << 19) | (1 << 20))) HIP_TRY(hipMemsetAsync(bin.entries, 0, (size_t)R * sizeof(uint2), stream));
#endif
        {
            StageTimer t(stream, MI_STAGE_EMIT);
            const size_t emit_lds = bin_lds + 8 * 1024 * sizeof(uint32_t);
            for (uint32_t b = 0; full && b < nbands; b++) {
                const uint32_t by0 = b * band_rows, by1 = std::min(vp.grid_y, by0 + band_rows);
                if (nocull)
                    hipLaunchKernelGGL(bin_ranks_kernel<true>, dim3(nwg), dim3(BIN_THREADS), emit_lds, stream, P, geom.index_rec, geom.depth_key,
                                       img.tile_count, img.ranges, bin.entries, vp.grid_x, vp.grid_y, by0, by1);
                else if (full)
                    hipLaunchKernelGGL(bin_ranks_kernel<false>, dim3(nwg), dim3(BIN_THREADS), emit_lds, stream, P, geom.index_rec, geom.depth_key,
                                       img.tile_count, img.ranges, bin.entries, vp.grid_x, vp.grid_y, by0, by1);
            }
            if (!full)
                hipLaunchKernelGGL(bin_spans_kernel<true>, dim3(nwg), dim3(1024), span_lds_bytes((size_t)lean_band_h * vp.grid_x), stream, P,
                                   geom.index_rec, geom.depth_key, geom.band_bits, img.tile_count, img.tile_cursor, img.ranges, bin.entries, vp.grid_x, vp.grid_y,
                                   lean_band_h, lean_nbands, (const int*)img.num_rendered, (int*)nullptr, g_ablate_fwd);
        }
        STAGE_CHECK("emit entries");
        if (verify) {   // debugging aid: synchronous
            uint32_t* ctr = reinterpret_cast<uint32_t*>(img.num_rendered + R_SLOTS * R_SLOT_STRIDE + NR_VERIFY);
            HIP_TRY(hipMemsetAsync(ctr, 0, sizeof(uint32_t), stream));
            hipLaunchKernelGGL(verify_entries_kernel, dim3(ntiles), dim3(256), 0, stream, (uint32_t)ntiles, img.ranges, bin.entries, ctr);
            uint32_t unwritten = 0;
            HIP_TRY(hipMemcpyAsync(&unwritten, ctr, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            if (unwritten)
                return fail(MI_RAST_ERR_HIP, "internal error: " + std::to_string(unwritten) + " list slots were counted but not emitted "
                                             "(count / emit passes of bin_spans_kernel disagree)");
        }
        const int* key_bits = img.num_rendered + R_SLOTS * R_SLOT_STRIDE + NR_KEY_BITS;
        {
            StageTimer t(stream, MI_STAGE_TILE_SORT);
            // lists of up to 4096 entries: one wave per tile (bitonic network in registers, three length classes); longer ones: two classes of the LDS radix
            // sort (1024 threads + 112 KB, 1024 threads + 144 KB / HBM ping-pong), launched only if some tile needs them.  The longest
            // list was copied to the host right after the range scan, which finished before the emit pass above even started: this
            // wait does not stall the queue.
#define LAUNCH_TILE_SORT(LO, CAP, FB, NT)                                                                                 \
    hipLaunchKernelGGL((tile_sort_kernel<LO, CAP, FB, NT>), dim3(ntiles), dim3(NT), 0, stream, (uint32_t)ntiles, img.ranges,  \
                       bin.entries, bin.scratch, key_bits, bin.blend_list)
            hipLaunchKernelGGL(tile_sort_wave_kernel<0>, dim3((ntiles + 3) / 4), dim3(256), 0, stream, (uint32_t)ntiles, img.ranges,
                               (const uint2*)bin.entries, bin.blend_list);
            HIP_TRY(hipEventSynchronize(g_host_sync.ev2));
            img.longest_run = longest_of(g_host_sync.pinned + R_SLOTS * R_SLOT_STRIDE + NR_RUN_BOUNDS, (uint32_t)ntiles);
            g_last_longest_run[dev] = (int)img.longest_run;
            // lean lists: the counts are the lists' exact lengths (bin_spans_kernel), their sum a lower bound of R
            if (full ? g_host_sync.pinned[R_SLOTS * R_SLOT_STRIDE + NR_TOTAL] != R : g_host_sync.pinned[R_SLOTS * R_SLOT_STRIDE + NR_TOTAL] > R)
                return fail(MI_RAST_ERR_HIP, "internal error: tile counts do not add up to num_rendered");
            const int max_tile_count = g_host_sync.pinned[R_SLOTS * R_SLOT_STRIDE + NR_LONGEST];
            if (max_tile_count > 1024)
                hipLaunchKernelGGL(tile_sort_wave_kernel<1>, dim3((ntiles + 3) / 4), dim3(256), 0, stream, (uint32_t)ntiles, img.ranges,
                                   (const uint2*)bin.entries, bin.blend_list);
            if (max_tile_count > 2048)
                hipLaunchKernelGGL(tile_sort_wave_kernel<2>, dim3((ntiles + 3) / 4), dim3(256), 0, stream, (uint32_t)ntiles, img.ranges,
                                   (const uint2*)bin.entries, bin.blend_list);
            if (max_tile_count > TILE_SORT_WAVE_MAX) LAUNCH_TILE_SORT(TILE_SORT_WAVE_MAX, 6144, false, 1024);
            if (max_tile_count > 6144) LAUNCH_TILE_SORT(6144, 8192, true, 1024);
#undef LAUNCH_TILE_SORT
        }
        STAGE_CHECK("tile sort");
    } else {
        // no overlap at all (every tile's range is {0, 0}: the blend kernels read no list).  The range scan stores {total, longest list} into this thread's pinned words: never return while that store can still
        // land (the next forward of this thread, on another stream, would read them)
        HIP_TRY(hipEventSynchronize(g_host_sync.ev2));
        img.longest_run = longest_of(g_host_sync.pinned + R_SLOTS * R_SLOT_STRIDE + NR_RUN_BOUNDS, (uint32_t)ntiles);
        g_last_longest_run[dev] = (int)img.longest_run;
    }
    return MI_RAST_OK;
}

// xexp: the device library's expf (default); false under MI_RAST_FAST_EXP (include/mi_rast.h, common.h gauss_exp)
template <int C, int EXTRA>
void launch_blend_fwd(const ViewParams& vp, hipStream_t stream, const ImgPtrs& img, const BinPtrs& bin,
                      const GeomPtrs& geom, const float* features, const float* mask, const float* bg,
                      float* out_color, float* out_mask, float* out_depth, bool xexp, int cstride = C, int cr = C)
{
    const int g_ablate_fwd = ablate_env("MI_RAST_ABLATE_FWD");
#define TILE_FWD_LAUNCH(XE, PART)                                                                                                     \
    hipLaunchKernelGGL((blend_fwd_kernel<C, EXTRA, XE, PART>), dim3(vp.grid_x, vp.grid_y), dim3(256), 0, stream, img.ranges,              \
                       bin.blend_list, geom.index_rec, vp.W, vp.H, features, mask, geom.depths, img.final_T, img.n_contrib,              \
                       img.tile_consumed, img.tile_nsurv, bg, out_color, out_mask, out_depth, cstride, cr, g_ablate_fwd)
    if constexpr (C == 16 && EXTRA == 0) {   // the remainder block of a feature: `cr` < 16 of its channels may exist (blend_fwd.h PARTIAL)
        if (cr < C) {
            if (xexp) TILE_FWD_LAUNCH(true, true);
            else TILE_FWD_LAUNCH(false, true);
            return;
        }
    }
    if (xexp) TILE_FWD_LAUNCH(true, false);
    else TILE_FWD_LAUNCH(false, false);
#undef TILE_FWD_LAUNCH
}

#ifdef MI_RAST_PROFILING
template <int C>
void launch_blend_fwd_x3(const ViewParams& vp, hipStream_t stream, const ImgPtrs& img, const BinPtrs& bin, const GeomPtrs& geom,
                         const float* features, const float* bg, float* out_color, bool xexp, int cstride)
{
#define X3_LAUNCH(XE, ST)                                                                                                           \
    hipLaunchKernelGGL((blend_fwd_x3_kernel<C, XE, ST>), dim3(vp.grid_x, vp.grid_y), dim3(256), 0, stream, img.ranges, bin.blend_list, \
                       geom.index_rec, vp.W, vp.H, features, img.final_T, img.n_contrib, img.tile_consumed, img.tile_nsurv, bg,         \
                       out_color, cstride)
    if (cstride == C) {
        if (xexp) X3_LAUNCH(true, false);
        else X3_LAUNCH(false, false);
    } else {
        if (xexp) X3_LAUNCH(true, true);
        else X3_LAUNCH(false, true);
    }
#undef X3_LAUNCH
}
#endif

// One wave per (tile, quadrant): 32 x the longest XCD run of tiles workgroups (blend_fwd_wave.h).
// xm: common.h ExpMode -- EXP_HYBRID unless the caller's flags say otherwise (exp_mode_of)
template <int C>
void launch_blend_fwd_wave(const ViewParams& vp, hipStream_t stream, const ImgPtrs& img, const BinPtrs& bin, const GeomPtrs& geom,
                           const float* features, const float* bg, float* out_color, int xm, int cstride, FwdZeroFill& zfill, int cr = C)
{
    const uint32_t nt = vp.grid_x * vp.grid_y;
    const uint32_t fm = fwd_runs_per_xcd();
    const uint32_t grid = 32u * (fm ? fwd_runs_longest(nt, fm) : (img.longest_run ? img.longest_run : xcd_max_run(nt)));
    const FwdZeroFill zf = zfill;
    zfill = FwdZeroFill{nullptr, 0u, nullptr, 0u};   // taken: the launches of further channel blocks fill nothing
#define FW_LAUNCH(XM, ST, PT)                                                                                                 \
    hipLaunchKernelGGL((blend_fwd_wave_kernel<C, XM, ST, PT>), dim3(grid), dim3(64), 0, stream, img.ranges, bin.blend_list,       \
                       geom.index_rec, vp.W, vp.H, vp.grid_x, nt, features, img.final_T, img.n_contrib, img.tile_consumed,        \
                       img.tile_nsurv, bg, out_color, cstride, zf, fm, img.run_bounds, cr)
#define FW_LAUNCH_ST(ST, PT)                                  \
    do {                                                      \
        if (xm == EXP_HYBRID) FW_LAUNCH(EXP_HYBRID, ST, PT);  \
        else if (xm == EXP_EXACT) FW_LAUNCH(EXP_EXACT, ST, PT); \
        else FW_LAUNCH(EXP_FAST, ST, PT);                     \
    } while (0)
    if constexpr (C == 32) {
        if (cr < C) {   // the remainder block of a feature: `cr` of its 32 channels exist (blend_fwd_wave.h PARTIAL)
            FW_LAUNCH_ST(true, true);
            return;
        }
    }
    if (cstride == C) FW_LAUNCH_ST(false, false);
    else FW_LAUNCH_ST(true, false);
#undef FW_LAUNCH_ST
#undef FW_LAUNCH
}

template <int EXTRA>
void launch_blend_fwd_wave_rgb(const ViewParams& vp, hipStream_t stream, const ImgPtrs& img, const BinPtrs& bin, const GeomPtrs& geom,
                               const float* features, const float* mask, const float* bg, float* out_color, float* out_mask,
                               float* out_depth, int xm, FwdZeroFill& zfill)
{
    const uint32_t nt = vp.grid_x * vp.grid_y;
    const uint32_t fm = fwd_runs_per_xcd();
    const uint32_t grid = 32u * (fm ? fwd_runs_longest(nt, fm) : (img.longest_run ? img.longest_run : xcd_max_run(nt)));
    const FwdZeroFill zf = zfill;
    zfill = FwdZeroFill{nullptr, 0u, nullptr, 0u};
#define RGB_LAUNCH(XM)                                                                                                                       \
    hipLaunchKernelGGL((blend_fwd_wave_rgb_kernel<EXTRA, XM>), dim3(grid), dim3(64), 0, stream, img.ranges, bin.blend_list, geom.index_rec,   \
                       vp.W, vp.H, vp.grid_x, nt, features, mask, geom.depths, img.final_T, img.n_contrib, img.tile_consumed,                 \
                       img.tile_nsurv, bg, out_color, out_mask, out_depth, zf, fm, img.run_bounds)
    if (xm == EXP_HYBRID) RGB_LAUNCH(EXP_HYBRID);
    else if (xm == EXP_EXACT) RGB_LAUNCH(EXP_EXACT);
    else RGB_LAUNCH(EXP_FAST);
#undef RGB_LAUNCH
}

template <int C, bool MASKGRAD>
void launch_blend_bwd(const ViewParams& vp, hipStream_t stream, const ImgPtrs& img, const BinPtrs& bin,
                      const GeomPtrs& geom, const float* colors, const float* bg, const float* dL_dpix,
                      const float* dL_dout_mask, float* dL_dcolor, bool xexp)
{
    const int g_ablate = ablate_env("MI_RAST_ABLATE");
    if (xexp)
        hipLaunchKernelGGL((blend_bwd_kernel<C, MASKGRAD, true>), dim3(vp.grid_x, vp.grid_y), dim3(256), 0, stream, img.ranges,
                           bin.blend_list, geom.index_rec, img.tile_nsurv, vp.W, vp.H, bg, colors, img.final_T, img.n_contrib, dL_dpix,
                           dL_dout_mask, geom.bwd_pack, dL_dcolor, g_ablate);
    else
        hipLaunchKernelGGL((blend_bwd_kernel<C, MASKGRAD, false>), dim3(vp.grid_x, vp.grid_y), dim3(256), 0, stream, img.ranges,
                           bin.blend_list, geom.index_rec, img.tile_nsurv, vp.W, vp.H, bg, colors, img.final_T, img.n_contrib, dL_dpix,
                           dL_dout_mask, geom.bwd_pack, dL_dcolor, g_ablate);
}

}  // namespace

namespace {
struct KnnWs {
    uint32_t* bbox;        // [8]
    uint32_t* codes[2];    // [M] ping-pong
    uint32_t* index[2];    // [M]
    uint32_t* hist;        // [256 * nblocks]
    float4* sorted_pts;    // [M]
    KnnBox* leaves;        // [nleaf]
    KnnBox* supers;        // [nsuper]
    size_t bytes;
};
KnnWs knn_carve(char* base, int M)
{
    const size_t m = M > 0 ? (size_t)M : 1;
    const size_t nblocks = (m + KNN_TILE - 1) / KNN_TILE;
    const size_t nleaf = (m + KNN_LEAF - 1) / KNN_LEAF, nsuper = (nleaf + KNN_FAN - 1) / KNN_FAN;
    Carver c;
    KnnWs w;
    w.bbox = (uint32_t*)(base + c.take(8 * sizeof(uint32_t)));
    for (int k = 0; k < 2; k++) w.codes[k] = (uint32_t*)(base + c.take(m * sizeof(uint32_t)));
    for (int k = 0; k < 2; k++) w.index[k] = (uint32_t*)(base + c.take(m * sizeof(uint32_t)));
    w.hist = (uint32_t*)(base + c.take(256 * nblocks * sizeof(uint32_t)));
    w.sorted_pts = (float4*)(base + c.take(m * sizeof(float4)));
    w.leaves = (KnnBox*)(base + c.take(nleaf * sizeof(KnnBox)));
    w.supers = (KnnBox*)(base + c.take(nsuper * sizeof(KnnBox)));
    w.bytes = c.off;
    return w;
}
}  // namespace

namespace {
template <int K, bool SELF, bool MEAN3>
void knn_launch(int rows, const float* query, int M, const KnnWs& w, int exclude_self, int64_t* idx, float* d2, hipStream_t stream)
{
    hipLaunchKernelGGL((knn_query_kernel<K, SELF, MEAN3>), dim3((rows + 63) / 64), dim3(64), 0, stream, rows, query, M, w.sorted_pts,
                       w.codes[0], w.bbox, w.leaves, w.supers, exclude_self, idx, d2);
}
}  // namespace

extern "C" {

const char* mi_rast_last_error(void) { return g_last_error.c_str(); }
#ifndef MI_RAST_SRC_HASH
#define MI_RAST_SRC_HASH "unstamped"
#endif
// "... src:<hash>": seganygaussians_amd/build.py source_hash() of the sources, headers and flags this library was built from
const char* mi_rast_version(void) { return "mi_rast 0.3 (gfx950) src:" MI_RAST_SRC_HASH; }

int mi_rast_supported_channels(int* out, int n)
{
    int k = 0;
    for (int c = 1; c <= MAX_CHANNELS; c++, k++)
        if (k < n) out[k] = c;
    return k;
}

// ---- fused KNN feature smoothing (mi_knn_smooth.h, knn_smooth.h) ------------------------------------------
int mi_knn_smooth_forward(int P, int C, int K, const int* knn_idx, uint32_t sel_mask, const float* features, float* out,
                          int normalize_out, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (P < 0 || K < 1 || K > 32 || (C != 32 && C != 64)) return fail(MI_RAST_ERR_INVALID, "knn_smooth: need C in {32, 64} and 1 <= K <= 32");
    const uint32_t mask = K == 32 ? sel_mask : (sel_mask & ((1u << K) - 1u));
    const int k = __builtin_popcount(mask);
    if (k < 1) return fail(MI_RAST_ERR_INVALID, "knn_smooth: no neighbour column selected");
    if (P == 0) return MI_RAST_OK;
    const long long threads = (long long)P * (C / 4);
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (C == 32)
        hipLaunchKernelGGL(knn_smooth_fwd_kernel<32>, grid, dim3(256), 0, stream, P, K, knn_idx, mask, 1.0f / (float)k, features, out, normalize_out);
    else
        hipLaunchKernelGGL(knn_smooth_fwd_kernel<64>, grid, dim3(256), 0, stream, P, K, knn_idx, mask, 1.0f / (float)k, features, out, normalize_out);
    HIP_TRY(hipGetLastError());
    return MI_RAST_OK;
}

int mi_knn_smooth_backward(int P, int C, int K, const int* knn_idx, const int* inv_offsets, const uint32_t* inv_entries,
                           uint32_t sel_mask, const float* features, const float* dL_dout, float* dmean,
                           float* dL_dfeatures, int normalize_out, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (P < 0 || K < 1 || K > 32 || (C != 32 && C != 64)) return fail(MI_RAST_ERR_INVALID, "knn_smooth: need C in {32, 64} and 1 <= K <= 32");
    if (P >= (1 << 27)) return fail(MI_RAST_ERR_INVALID, "knn_smooth: more than 2^27 Gaussians");
    const uint32_t mask = K == 32 ? sel_mask : (sel_mask & ((1u << K) - 1u));
    const int k = __builtin_popcount(mask);
    if (k < 1) return fail(MI_RAST_ERR_INVALID, "knn_smooth: no neighbour column selected");
    if (P == 0) return MI_RAST_OK;
    const long long threads = (long long)P * (C / 4);
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (C == 32) {
        hipLaunchKernelGGL(knn_smooth_bwd_mean_kernel<32>, grid, dim3(256), 0, stream, P, K, knn_idx, mask, 1.0f / (float)k, features, dL_dout, dmean, normalize_out);
        hipLaunchKernelGGL(knn_smooth_bwd_feat_kernel<32>, grid, dim3(256), 0, stream, P, inv_offsets, inv_entries, mask, features, dmean, dL_dfeatures);
    } else {
        hipLaunchKernelGGL(knn_smooth_bwd_mean_kernel<64>, grid, dim3(256), 0, stream, P, K, knn_idx, mask, 1.0f / (float)k, features, dL_dout, dmean, normalize_out);
        hipLaunchKernelGGL(knn_smooth_bwd_feat_kernel<64>, grid, dim3(256), 0, stream, P, inv_offsets, inv_entries, mask, features, dmean, dL_dfeatures);
    }
    HIP_TRY(hipGetLastError());
    return MI_RAST_OK;
}


// ---- contrastive-loss front end (mi_contrastive.h, contrastive.h) --------------------------------------------------------
int mi_contrastive_forward(int C, int h, int w, const float* rendered, int H, int W, int S, const int* ray_yx, int N,
                           const float* gates, float* out, float* ray_feat, float* inv_len, float* inv_norm,
                           double* norm_sum, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (C < 1 || h < 1 || w < 1 || H < 1 || W < 1 || S < 0 || N < 1) return fail(MI_RAST_ERR_INVALID, "contrastive: need C, h, w, H, W, N >= 1 and S >= 0");
    if (S > 0 && C > 64 * CT_MAX_CPL) return fail(MI_RAST_ERR_INVALID, "contrastive: at most 256 channels");
    if (!rendered || !inv_norm || !norm_sum || !gates) return fail(MI_RAST_ERR_INVALID, "contrastive: null pointer");
    if (S > 0 && (!ray_yx || !out || !ray_feat || !inv_len)) return fail(MI_RAST_ERR_INVALID, "contrastive: null ray buffers");
    const size_t HW = (size_t)h * w;
    const bool vec = HW % 4 == 0 && ((uintptr_t)rendered % 16) == 0 && ((uintptr_t)inv_norm % 16) == 0;
    const size_t per_block = (size_t)CT_THREADS * (vec ? 4 : 1);
    const uint32_t dense_blocks = (uint32_t)((HW + per_block - 1) / per_block);
    const uint32_t ray_blocks = (uint32_t)((S + CT_THREADS / 64 - 1) / (CT_THREADS / 64));
    if (vec)
        hipLaunchKernelGGL(contrastive_fwd_kernel<4>, dim3(dense_blocks + ray_blocks), dim3(CT_THREADS), 0, stream, C, h, w, rendered, H, W, S,
                           ray_yx, N, gates, out, ray_feat, inv_len, inv_norm, norm_sum, dense_blocks);
    else
        hipLaunchKernelGGL(contrastive_fwd_kernel<1>, dim3(dense_blocks + ray_blocks), dim3(CT_THREADS), 0, stream, C, h, w, rendered, H, W, S,
                           ray_yx, N, gates, out, ray_feat, inv_len, inv_norm, norm_sum, dense_blocks);
    HIP_TRY(hipGetLastError());
    return MI_RAST_OK;
}

int mi_contrastive_backward(int C, int h, int w, const float* rendered, int H, int W, int S, const int* ray_yx, int N,
                            const float* gates, const float* out, const float* ray_feat, const float* inv_len,
                            const float* inv_norm, const float* dL_dout, const float* g_norm, float* dL_drendered,
                            float* dL_dgates, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (C < 1 || h < 1 || w < 1 || H < 1 || W < 1 || S < 0 || N < 1) return fail(MI_RAST_ERR_INVALID, "contrastive: need C, h, w, H, W, N >= 1 and S >= 0");
    if (!rendered || !inv_norm || !dL_drendered) return fail(MI_RAST_ERR_INVALID, "contrastive: null pointer");
    if (S > 0 && (!ray_yx || !out || !ray_feat || !inv_len || !dL_dout || !gates || !dL_dgates)) return fail(MI_RAST_ERR_INVALID, "contrastive: null ray buffers");
    const size_t lds = (size_t)(CT_THREADS / 64) * N * C * sizeof(float);
    if (S > 0 && lds > 64 * 1024) return fail(MI_RAST_ERR_INVALID, "contrastive: N * C too large for the gate-gradient reduction (4 N C floats of LDS)");
    const size_t HW = (size_t)h * w;
    const bool vec = HW % 4 == 0 && ((uintptr_t)rendered % 16) == 0 && ((uintptr_t)inv_norm % 16) == 0 && ((uintptr_t)dL_drendered % 16) == 0;
    const size_t per_block = (size_t)CT_THREADS * (vec ? 4 : 1);
    const uint32_t dense_blocks = (uint32_t)((HW + per_block - 1) / per_block);
    if (vec)
        hipLaunchKernelGGL(contrastive_bwd_dense_kernel<4>, dim3(dense_blocks), dim3(CT_THREADS), 0, stream, C, h, w, rendered, inv_norm, g_norm, dL_drendered);
    else
        hipLaunchKernelGGL(contrastive_bwd_dense_kernel<1>, dim3(dense_blocks), dim3(CT_THREADS), 0, stream, C, h, w, rendered, inv_norm, g_norm, dL_drendered);
    if (S > 0) {
        const uint32_t ray_blocks = (uint32_t)((S + CT_THREADS / 64 - 1) / (CT_THREADS / 64));
        hipLaunchKernelGGL(contrastive_bwd_rays_kernel, dim3(ray_blocks), dim3(CT_THREADS), lds, stream, C, h, w, H, W, S, ray_yx, N, gates, out,
                           ray_feat, inv_len, dL_dout, dL_drendered, dL_dgates);
    }
    HIP_TRY(hipGetLastError());
    return MI_RAST_OK;
}


// ---- exact KNN (mi_knn.h, knn.h) ---------------------------------------------------------------------------------------

size_t mi_knn_workspace_bytes(int M) { return knn_carve(nullptr, M).bytes; }

int mi_knn_build(int M, const float* ref, void* workspace, size_t workspace_bytes, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (M <= 0 || !ref || !workspace) return fail(MI_RAST_ERR_INVALID, "knn: need M > 0, reference points and a workspace");
    const KnnWs w = knn_carve((char*)workspace, M);
    if (workspace_bytes < w.bytes) return fail(MI_RAST_ERR_INVALID, "knn: workspace smaller than mi_knn_workspace_bytes(M)");
    const int nblocks = (M + KNN_TILE - 1) / KNN_TILE;
    const int nleaf = (M + KNN_LEAF - 1) / KNN_LEAF, nsuper = (nleaf + KNN_FAN - 1) / KNN_FAN;
    hipLaunchKernelGGL(knn_init_kernel, dim3(1), dim3(64), 0, stream, w.bbox);
    hipLaunchKernelGGL(knn_bbox_kernel, dim3(std::min(1024, (M + 255) / 256)), dim3(256), 0, stream, M, ref, w.bbox);
    hipLaunchKernelGGL(knn_morton_kernel, dim3((M + 255) / 256), dim3(256), 0, stream, M, ref, w.bbox, w.codes[0], w.index[0]);
    for (int pass = 0; pass < 4; pass++) {  // 30-bit codes: four 8-bit digits
        const int a = pass & 1, b = a ^ 1;
        hipLaunchKernelGGL(knn_radix_hist_kernel, dim3(nblocks), dim3(256), 0, stream, M, w.codes[a], 8 * pass, nblocks, w.hist);
        hipLaunchKernelGGL(knn_scan_kernel, dim3(1), dim3(1024), 0, stream, 256 * nblocks, w.hist);
        hipLaunchKernelGGL(knn_radix_scatter_kernel, dim3(nblocks), dim3(256), 0, stream, M, w.codes[a], w.index[a], 8 * pass, nblocks,
                           w.hist, w.codes[b], w.index[b]);
    }
    // four passes: the sorted pairs are back in buffer 0
    hipLaunchKernelGGL(knn_leaf_kernel, dim3(nleaf), dim3(KNN_LEAF), 0, stream, M, ref, w.index[0], w.sorted_pts, w.leaves);
    hipLaunchKernelGGL(knn_super_kernel, dim3(nsuper), dim3(KNN_FAN), 0, stream, nleaf, w.leaves, w.supers);
    HIP_TRY(hipGetLastError());
    return MI_RAST_OK;
}


int mi_knn_query(int N, const float* query, int M, const void* workspace, int K, int exclude_self, int64_t* idx,
                 float* dist2, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (M <= 0 || !workspace || !idx || !dist2) return fail(MI_RAST_ERR_INVALID, "knn: need M > 0, an index and output buffers");
    if (K < 1 || K > MI_KNN_MAX_K) return fail(MI_RAST_ERR_INVALID, "knn: 1 <= K <= 32");
    const KnnWs w = knn_carve((char*)const_cast<void*>(workspace), M);
    const bool self = query == nullptr;
    const int rows = self ? M : N;
    if (rows <= 0) return MI_RAST_OK;
    // the kernels keep a list of KT >= K candidates; they write KT columns, so K must be one of the compiled sizes
    if (K != 1 && K != 3 && K != 4 && K != 8 && K != 16 && K != 32)
        return fail(MI_RAST_ERR_INVALID, "knn: K must be one of 1, 3, 4, 8, 16, 32");
#define KNN_DISPATCH(KT)                                                                          \
    if (K == KT) {                                                                                \
        if (self) knn_launch<KT, true, false>(rows, nullptr, M, w, exclude_self, idx, dist2, stream); \
        else knn_launch<KT, false, false>(rows, query, M, w, 0, idx, dist2, stream);              \
    }
    KNN_DISPATCH(1) KNN_DISPATCH(3) KNN_DISPATCH(4) KNN_DISPATCH(8) KNN_DISPATCH(16) KNN_DISPATCH(32)
#undef KNN_DISPATCH
    HIP_TRY(hipGetLastError());
    return MI_RAST_OK;
}

int mi_knn_mean_dist2(int P, const float* points, void* workspace, size_t workspace_bytes, float* out, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (P <= 0) return MI_RAST_OK;
    if (!out) return fail(MI_RAST_ERR_INVALID, "knn: null output");
    const int rc = mi_knn_build(P, points, workspace, workspace_bytes, stream_);
    if (rc) return rc;
    const KnnWs w = knn_carve((char*)workspace, P);
    knn_launch<3, true, true>(P, nullptr, P, w, 1, nullptr, out, stream);
    HIP_TRY(hipGetLastError());
    return MI_RAST_OK;
}

// CF/cuda_rasterizer/rasterizer_impl.cu:35-50
uint32_t mi_rast_get_higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step;
        else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

size_t mi_rast_geometry_layout(int P, size_t* off)
{
    const size_t p = P > 0 ? (size_t)P : 1;
    Carver c;
    off[MI_GEOM_DEPTHS] = c.take(p * sizeof(float));
    off[MI_GEOM_MEANS2D] = c.take(p * sizeof(float2));
    off[MI_GEOM_CONIC_OPACITY] = c.take(p * sizeof(float4));
    off[MI_GEOM_COV3D] = c.take(p * 6 * sizeof(float));
    off[MI_GEOM_RGB] = c.take(p * 3 * sizeof(float));
    off[MI_GEOM_CLAMPED] = c.take(p * 3);
    off[MI_GEOM_TILES_TOUCHED] = c.take(p * sizeof(uint32_t));
    off[MI_GEOM_DEPTH_KEY] = c.take(p * sizeof(uint32_t));
    off[MI_GEOM_INDEX_REC] = c.take(p * sizeof(BlendRec));
    off[MI_GEOM_CULL_COUNTER] = c.take(16);
    off[MI_GEOM_BAND_BITS] = c.take((size_t)MAX_BANDS * ((p + 63) / 64) * sizeof(unsigned long long));
    off[MI_GEOM_BWD_PACK] = c.take(bwd_pack_bytes((int)p));  // + the backward's work-queue counters, one set per channel block
    return c.off;
}
size_t mi_rast_image_layout(int width, int height, size_t* off)
{
    const size_t n = (size_t)width * height > 0 ? (size_t)width * height : 1;
    const size_t tiles = (size_t)((width + TILE_X - 1) / TILE_X) * ((height + TILE_Y - 1) / TILE_Y);
    Carver c;
    off[MI_IMG_FINAL_T] = c.take(n * sizeof(float));
    off[MI_IMG_N_CONTRIB] = c.take(n * sizeof(uint32_t));
    off[MI_IMG_RANGES] = c.take((tiles ? tiles : 1) * sizeof(uint2));
    off[MI_IMG_TILE_CONSUMED] = c.take((tiles ? tiles : 1) * sizeof(uint32_t));
    // tile_count holds partial[slice][tile] of the count / emit passes (binning.h); tile_cursor the tile totals
    constexpr int max_slices = BIN_MAX_WG;   // (the lean passes' [workgroup][tile of its band] table is far smaller: BIN_LEAN_WG_MAX x a band's tiles)
    off[MI_IMG_TILE_COUNT] = c.take((size_t)max_slices * (tiles ? tiles : 1) * sizeof(uint32_t));
    off[MI_IMG_TILE_CURSOR] = c.take((tiles ? tiles : 1) * sizeof(uint32_t));
    off[MI_IMG_NUM_RENDERED] = c.take((R_SLOTS * R_SLOT_STRIDE + 16) * sizeof(int));  // R partial sums, then {R, longest list}, then the nine run boundaries
    off[MI_IMG_TILE_NSURV] = c.take((tiles ? tiles : 1) * sizeof(uint32_t));
    return c.off;
}
size_t mi_rast_binning_layout(int R, size_t* off)
{
    const size_t r = R > 0 ? (size_t)R : 1;
    Carver c;
    // the blend list first: it is all the blend kernels (and a caller that keeps a view's lists, mi_rast_forward_reuse) need of this
    // buffer, at an offset that does not depend on R
    off[MI_BIN_BLEND_LIST] = c.take(r * sizeof(uint32_t));
    off[MI_BIN_ENTRIES] = c.take(r * sizeof(uint2));
    off[MI_BIN_SCRATCH] = c.take(r * sizeof(uint2));
    return c.off;
}

#ifdef MI_RAST_PROFILING
// Profiling build only (tools/xcd_stamps.py; not part of include/mi_rast.h): from the blend kernels' per-wave stamps since the last reset,
// per XCD x: out[x] = latest end of a wave, out[8 + x] = earliest start, out[16 + x] = waves -- ticks of the constant 100-MHz clock.
int mi_rast_xcd_stamps(unsigned long long* out, int reset)
{
    static std::vector<unsigned long long> h(2 * (size_t)MI_XCD_LOG_WAVES);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_xcd_log), h.size() * sizeof(unsigned long long)));
    for (int k = 0; k < 24; k++) out[k] = 0ull;
    for (size_t b = 0; b < MI_XCD_LOG_WAVES; b++) {
        const unsigned long long s0 = h[2 * b], e0 = h[2 * b + 1];
        if (s0 == 0ull || e0 == 0ull) continue;
        const int x = (int)(s0 & 7ull);
        const unsigned long long st = s0 >> 3;
        out[x] = std::max(out[x], e0);
        out[8 + x] = out[8 + x] == 0ull ? st : std::min(out[8 + x], st);
        out[16 + x]++;
    }
    if (reset) {
        std::fill(h.begin(), h.end(), 0ull);
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_xcd_log), h.data(), h.size() * sizeof(unsigned long long)));
    }
    return MI_RAST_OK;
}
#endif

int mi_rast_profile_enable(int on)
{
    if (on && !profile_events(current_device())) return fail(MI_RAST_ERR_HIP, "hipEventCreate failed");
    g_profile.store(on != 0);
    for (int d = 0; d < MAX_DEVICES; d++)
        for (int i = 0; i < MI_STAGE_COUNT; i++) g_prof[d].used[i] = false;
    return MI_RAST_OK;
}

int mi_rast_profile_read(float* ms)
{
    const int dev = current_device();
    ProfileEvents* pe = dev >= 0 && g_prof[dev].created ? &g_prof[dev] : nullptr;
    for (int i = 0; i < MI_STAGE_COUNT; i++) {
        ms[i] = 0.f;
        if (pe && pe->used[i]) {
            HIP_TRY(hipEventSynchronize(pe->ev[i][1]));
            HIP_TRY(hipEventElapsedTime(&ms[i], pe->ev[i][0], pe->ev[i][1]));
            pe->used[i] = false;
        }
    }
    return MI_RAST_OK;
}

// The blend stage of a forward: CF/cuda_rasterizer/rasterizer_impl.cu:319-335 (+ the zero fills for this view's backward).  Shared by
// mi_rast_forward and mi_rast_forward_reuse.
static int blend_forward_stage(const ViewParams& vp, hipStream_t stream, GeomPtrs& geom, ImgPtrs& img, BinPtrs& bin, int P, int channels,
                               int width, int height, const float* background, const float* colors_precomp, const float* mask,
                               float* out_color, float* out_mask, float* out_depth, int debug, int flags, void* features_ready_event,
                               float* dL_dcolor_next)
{
    int rc;
    const float* feature_ptr = colors_precomp != nullptr ? colors_precomp : geom.rgb;  // rasterizer_impl.cu:321
    if (features_ready_event != nullptr) {
        // everything above depends on the geometry only; colors_precomp may still be in the making (mi_rast.h)
        HIP_TRY(hipStreamWaitEvent(stream, (hipEvent_t)features_ready_event, 0));
    }
    {
        StageTimer t(stream, MI_STAGE_BLEND_FWD);
        // zero-fill for the backward of this view (mi_rast.h: dL_dcolor_next, MI_RAST_PREZERO_BWD): stored by the first wave-per-quadrant
        // blend launch beside its own work (blend_fwd_wave.h); tails that are not whole 16-byte units, and everything when another
        // kernel blends, by fill commands
        FwdZeroFill zfill{nullptr, 0u, nullptr, 0u};
        {
            // The kernel takes the fill while it at most doubles the kernel's own stores (the image: 265 MB against 160 MB of
            // zeros on cfg3); beyond that the zeros are HBM time wherever they are stored, and the fill engine stores them faster
            // (cfg5, 5 M x 64 channels: 1.44 GB of zeros against a 435-MB image: blend 0.38 -> 0.67 ms with the fill on board,
            // 0.22 ms as fill commands).
            const size_t ride_limit = (size_t)channels * width * height * sizeof(float);
            size_t riding = 0;
            auto region = [&](float* ptr, size_t bytes, float4*& p, uint32_t& n16) -> int {
                if (ptr == nullptr || bytes == 0) return MI_RAST_OK;
                if ((reinterpret_cast<uintptr_t>(ptr) & 15u) != 0 || (bytes >> 4) > 0xFFFFFFFFull || riding + bytes > ride_limit) {
                    HIP_TRY(hipMemsetAsync(ptr, 0, bytes, stream));
                    return MI_RAST_OK;
                }
                p = reinterpret_cast<float4*>(ptr);
                n16 = (uint32_t)(bytes >> 4);
                riding += bytes;
                if (bytes & 15u) HIP_TRY(hipMemsetAsync(reinterpret_cast<char*>(ptr) + (bytes & ~(size_t)15), 0, bytes & 15u, stream));
                return MI_RAST_OK;
            };
            if ((rc = region(dL_dcolor_next, (size_t)P * channels * sizeof(float), zfill.a, zfill.na))) return rc;
            if (flags & MI_RAST_PREZERO_BWD)
                if ((rc = region(geom.bwd_pack, bwd_pack_bytes(P), zfill.b, zfill.nb))) return rc;
        }
        const bool xexp = (flags & MI_RAST_FAST_EXP) == 0;   // the kernels without a hybrid form: expf unless MI_RAST_FAST_EXP
        const int xm = (flags & MI_RAST_FAST_EXP) ? EXP_FAST : (flags & MI_RAST_EXACT_EXP) ? EXP_EXACT : EXP_HYBRID;   // the wave-per-quadrant kernels
        // Product kernels: the wave-per-quadrant forward (RGB, RGB + mask + depth, 64- and 32-channel blocks) and the tile-batched kernel for
        // the 16-channel remainder block.  The comparison kernels of earlier rounds -- MI_RAST_TILE_FWD (tile-batched bf16x3 / RGB),
        // MI_RAST_F32_BLEND (f32 FMA chain) -- exist in the profiling build only (libmi_rast_prof.so, seganygaussians_amd/build.py).
#ifdef MI_RAST_PROFILING
        const bool tile_fwd = (flags & MI_RAST_TILE_FWD) != 0, f32_blend = (flags & MI_RAST_F32_BLEND) != 0;
#else
        constexpr bool tile_fwd = false, f32_blend = false;
        if (flags & (MI_RAST_TILE_FWD | MI_RAST_F32_BLEND))
            return fail(MI_RAST_ERR_INVALID, "MI_RAST_TILE_FWD / MI_RAST_F32_BLEND select comparison kernels of the profiling build "
                                             "(libmi_rast_prof.so: python -m seganygaussians_amd.build --profiling, MI_RAST_LIB=<path>)");
#endif
        if (mask || channels == 3) {
#ifdef MI_RAST_PROFILING
            if (tile_fwd && mask) launch_blend_fwd<3, 2>(vp, stream, img, bin, geom, feature_ptr, mask, background, out_color, out_mask, out_depth, xexp);
            else if (tile_fwd) launch_blend_fwd<3, 0>(vp, stream, img, bin, geom, feature_ptr, nullptr, background, out_color, nullptr, nullptr, xexp);
            else
#endif
            if (mask) launch_blend_fwd_wave_rgb<2>(vp, stream, img, bin, geom, feature_ptr, mask, background, out_color, out_mask, out_depth, xm, zfill);
            else launch_blend_fwd_wave_rgb<0>(vp, stream, img, bin, geom, feature_ptr, nullptr, background, out_color, nullptr, nullptr, xm, zfill);
        } else {
            // feature channels in blocks of 64 / 32 (one launch per block; see channels_supported); what is left behind the last whole
            // block -- 1 .. 31 channels -- is a PARTIAL 32-channel block of the same wave-per-quadrant kernel
            const size_t HW = (size_t)width * height;
            for (int c0 = 0; c0 < channels;) {
                const int rem = channels - c0;
                const int cb = rem >= 64 ? 64 : 32;
                const float* f = feature_ptr + c0;
                const float* bgp = background + c0;
                float* out = out_color + (size_t)c0 * HW;
                if (cb == 64) {
#ifdef MI_RAST_PROFILING
                    if (f32_blend) launch_blend_fwd<64, 0>(vp, stream, img, bin, geom, f, nullptr, bgp, out, nullptr, nullptr, xexp, channels);
                    else if (tile_fwd) launch_blend_fwd_x3<64>(vp, stream, img, bin, geom, f, bgp, out, xexp, channels);
                    else
#endif
                    launch_blend_fwd_wave<64>(vp, stream, img, bin, geom, f, bgp, out, xm, channels, zfill);
                } else if (rem >= 32) {
#ifdef MI_RAST_PROFILING
                    if (f32_blend) launch_blend_fwd<32, 0>(vp, stream, img, bin, geom, f, nullptr, bgp, out, nullptr, nullptr, xexp, channels);
                    else if (tile_fwd) launch_blend_fwd_x3<32>(vp, stream, img, bin, geom, f, bgp, out, xexp, channels);
                    else
#endif
                    launch_blend_fwd_wave<32>(vp, stream, img, bin, geom, f, bgp, out, xm, channels, zfill);
                } else {
#ifdef MI_RAST_PROFILING
                    // (the comparison kernels have no partial 32-channel form: the tile-batched 16-channel kernel, in one or two blocks)
                    if (f32_blend || tile_fwd) {
                        for (int c1 = 0; c1 < rem; c1 += 16)
                            launch_blend_fwd<16, 0>(vp, stream, img, bin, geom, f + c1, nullptr, bgp + c1, out + (size_t)c1 * HW, nullptr, nullptr, xexp,
                                                    channels, std::min(16, rem - c1));
                    } else
#endif
                    launch_blend_fwd_wave<32>(vp, stream, img, bin, geom, f, bgp, out, xm, channels, zfill, rem);
                }
                c0 += cb;
            }
        }
        (void)tile_fwd;
        (void)f32_blend;
        // no wave-per-quadrant launch took the fill (tile-batched / f32 / 16-channel kernels): fill commands
        if (zfill.na) HIP_TRY(hipMemsetAsync(zfill.a, 0, (size_t)zfill.na << 4, stream));
        if (zfill.nb) HIP_TRY(hipMemsetAsync(zfill.b, 0, (size_t)zfill.nb << 4, stream));
        // A view that will be differentiated (the caller asked for the backward's buffers to be left zeroed): the backward blend's
        // XCD runs from what this forward walked (common.h "WORK-balanced runs"; inside the forward blend's stage time)
        const int nt_all = (int)(vp.grid_x * vp.grid_y);
        const int walk_scan = knob("MI_RAST_BWD_SCAN", BWD_RUNS_FROM_WALKS);
        const uint32_t equal_share = ((uint32_t)nt_all + 7u) >> 3;
        if (((flags & MI_RAST_PREZERO_BWD) || dL_dcolor_next != nullptr) && nt_all <= BIN_MAX_TILES_TOTAL && !(flags & MI_RAST_EQUAL_RUNS) &&
            (walk_scan == 1 || (walk_scan == 2 && img.longest_run > equal_share + (equal_share >> 3))))
            hipLaunchKernelGGL(run_bounds_from_walks_kernel, dim3(1), dim3(1024), ((size_t)nt_all + 1) * sizeof(uint32_t), stream, nt_all,
                               img.tile_nsurv, img.run_bounds);
    }
    STAGE_CHECK("render");
    return MI_RAST_OK;
}

int mi_rast_forward(mi_rast_resize_fn geometry_buffer, void* geometry_user, mi_rast_resize_fn binning_buffer,
                    void* binning_user, mi_rast_resize_fn image_buffer, void* image_user, int P, int D, int M,
                    int channels, const float* background, int width, int height, const float* means3D,
                    const float* shs, const float* colors_precomp, const float* opacities, const float* scales,
                    float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                    const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
                    const float* mask, float* out_color, float* out_mask, float* out_depth, int* radii, int debug,
                    int flags, void* features_ready_event, float* dL_dcolor_next, void* stream_, int* num_rendered)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (num_rendered) *num_rendered = 0;
    if (P <= 0 || width <= 0 || height <= 0) return fail(MI_RAST_ERR_INVALID, "P, width and height must be positive");
    if (!channels_supported(channels)) return fail(MI_RAST_ERR_INVALID, "unsupported channel count (supported: 1 .. 256)");
    if (mask && channels != 3) return fail(MI_RAST_ERR_INVALID, "mask/depth variant is built for 3 channels");
    // CF/cuda_rasterizer/rasterizer_impl.cu:242-245
    if (channels != 3 && colors_precomp == nullptr)
        return fail(MI_RAST_ERR_NON_RGB, "For non-RGB, provide precomputed Gaussian colors!");
    if (!num_rendered || !radii || !out_color) return fail(MI_RAST_ERR_INVALID, "null output pointer");

    const ViewParams vp = make_view(viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, scale_modifier, width, height);
    int rc;

    GeomPtrs geom;
    ImgPtrs img;
    BinPtrs bin;
    rc = geometry_and_binning(geometry_buffer, geometry_user, binning_buffer, binning_user, image_buffer, image_user, P,
                              D, M, width, height, means3D, shs, colors_precomp != nullptr, opacities, scales,
                              rotations, cov3D_precomp, vp, prefiltered, radii, debug, flags, stream, geom, img, bin,
                              num_rendered);
    if (rc) return rc;

    return blend_forward_stage(vp, stream, geom, img, bin, P, channels, width, height, background, colors_precomp, mask, out_color, out_mask,
                               out_depth, debug, flags, features_ready_event, dL_dcolor_next);
}

// Extension (no counterpart in the reference; include/mi_rast.h): the blend stage alone over the state a previous mi_rast_forward
// of the SAME geometry, camera and list mode left in its buffers.
int mi_rast_forward_reuse(int P, int channels, int R, const float* background, int width, int height, const float* colors_precomp,
                          char* geom_buffer, char* binning_buffer, const void* cached_ranges, const int* cached_words, char* img_buffer,
                          int longest_run, const float* mask, float* out_color, float* out_mask, float* out_depth, int flags,
                          void* features_ready_event, float* dL_dcolor_next, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (P <= 0 || width <= 0 || height <= 0) return fail(MI_RAST_ERR_INVALID, "P, width and height must be positive");
    if (!channels_supported(channels)) return fail(MI_RAST_ERR_INVALID, "unsupported channel count (supported: 1 .. 256)");
    if (mask && channels != 3) return fail(MI_RAST_ERR_INVALID, "mask/depth variant is built for 3 channels");
    if (channels != 3 && colors_precomp == nullptr)
        return fail(MI_RAST_ERR_NON_RGB, "For non-RGB, provide precomputed Gaussian colors!");
    if (!geom_buffer || !binning_buffer || !cached_ranges || !cached_words || !img_buffer || !out_color) return fail(MI_RAST_ERR_INVALID, "null pointer");
    ViewParams vp;
    std::memset(&vp, 0, sizeof(vp));
    vp.W = width;
    vp.H = height;
    vp.grid_x = (width + TILE_X - 1) / TILE_X;
    vp.grid_y = (height + TILE_Y - 1) / TILE_Y;
    const int debug = 0;
    GeomPtrs geom = geom_from(geom_buffer, P);
    BinPtrs bin = bin_from(binning_buffer, R);
    ImgPtrs img = img_from(img_buffer, width, height);
    const uint32_t ntiles = vp.grid_x * vp.grid_y;
    img.longest_run = longest_run > 0 ? std::min((uint32_t)longest_run, xcd_max_run(ntiles)) : 0u;
    {
        // the per-view words of the image buffer that the binning stages of the cached forward wrote: tile ranges, {R, longest list,
        // key bits}, the XCD run boundaries -- copied; the per-tile walk counters of the blend kernels -- zeroed (binning.h)
        StageTimer t(stream, MI_STAGE_TILE_SCAN);
        hipLaunchKernelGGL(reuse_image_state_kernel, dim3((ntiles + 255) / 256), dim3(256), 0, stream, ntiles, (const uint2*)cached_ranges,
                           cached_words, img.ranges, img.num_rendered + R_SLOTS * R_SLOT_STRIDE, img.tile_consumed, img.tile_nsurv);
    }
    STAGE_CHECK("reuse image state");
    return blend_forward_stage(vp, stream, geom, img, bin, P, channels, width, height, background, colors_precomp, mask, out_color, out_mask,
                               out_depth, debug, flags, features_ready_event, dL_dcolor_next);
}

// Content fingerprints of up to MI_FP_MAX device arrays (a 64-bit sum of position-dependent word hashes each): what a caller keys
// reuse decisions on when tensors are recomputed per call (activation outputs) and identity says nothing.  Synchronous: one kernel,
// then the calling thread waits for `stream`.
int mi_rast_fingerprint(int n, const void* const* ptrs, const size_t* nbytes, uint64_t* out, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || n > MI_FP_MAX) return fail(MI_RAST_ERR_INVALID, "mi_rast_fingerprint: at most 8 arrays");
    if (n == 0) return MI_RAST_OK;
    const int dev = current_device();
    if (dev < 0) return fail(MI_RAST_ERR_HIP, "hipGetDevice failed (or device ordinal >= 64)");
    HostSync& hs = g_host_sync_tl[dev];
    if (!hs.init()) return fail(MI_RAST_ERR_HIP, "cannot allocate pinned host buffer / event");
    FingerprintArgs a;
    for (int k = 0; k < MI_FP_MAX; k++) {
        a.ptr[k] = k < n ? (const uint32_t*)ptrs[k] : nullptr;
        a.words[k] = k < n ? (unsigned long long)(nbytes[k] / 4) : 0ull;
        if (k < n && ((nbytes[k] & 3) || (reinterpret_cast<uintptr_t>(ptrs[k]) & 3))) return fail(MI_RAST_ERR_INVALID, "mi_rast_fingerprint: arrays of 4-byte words");
    }
    hipLaunchKernelGGL(fingerprint_kernel, dim3(FP_BLOCKS, n), dim3(256), 0, stream, a, (unsigned long long*)hs.fp_dev);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));
    for (int k = 0; k < n; k++) {
        uint64_t sum = 0;
        for (int b = 0; b < FP_BLOCKS; b++) sum += hs.fp[k * FP_BLOCKS + b];
        out[k] = sum;
    }
    return MI_RAST_OK;
}

// The longest XCD run of tiles of the calling thread's last mi_rast_forward on the current device (what sizes the forward blend's
// grid): a caller that keeps the buffers of that forward for mi_rast_forward_reuse passes it back; 0 if unknown.
int mi_rast_last_longest_run(void)
{
    const int dev = current_device();
    return dev < 0 ? 0 : g_last_longest_run[dev];
}

int mi_rast_features_only_supported(int channels) { return channels >= 16 && channels <= 256 && channels % 16 == 0 ? 1 : 0; }

int mi_rast_backward(int P, int D, int M, int channels, int R, const float* background, int width, int height,
                     const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                     float scale_modifier, const float* rotations, const float* cov3D_precomp,
                     const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                     float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer, char* img_buffer,
                     const float* dL_dpix, const float* dL_dout_mask, float* dL_dmean2D, float* dL_dconic,
                     float* dL_dopacity, float* dL_dcolor, float* dL_dmask, float* dL_dmean3D, float* dL_dcov3D,
                     float* dL_dsh, float* dL_dscale, float* dL_drot, int debug, int flags, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (P <= 0) return MI_RAST_OK;
    const bool xexp = (flags & MI_RAST_FAST_EXP) == 0;
    const int g_ablate = ablate_env("MI_RAST_ABLATE");
    (void)g_ablate;
    if (!channels_supported(channels)) return fail(MI_RAST_ERR_INVALID, "unsupported channel count (supported: 1 .. 256)");
    const bool maskgrad = dL_dmask != nullptr;
    if (maskgrad && (channels != 3 || !dL_dout_mask)) return fail(MI_RAST_ERR_INVALID, "mask gradient needs 3 channels and dL_dout_mask");
    // EXTENSION (mi_rast.h): dL_dcolor alone -- blend_bwd_feat.h instead of the full blend kernel, no geometry backward
    const bool feat_only = (flags & MI_RAST_BWD_FEATURES_ONLY) != 0;
    if (feat_only && (!mi_rast_features_only_supported(channels) || colors_precomp == nullptr || maskgrad || dL_dcolor == nullptr))
        return fail(MI_RAST_ERR_INVALID, "MI_RAST_BWD_FEATURES_ONLY needs colors_precomp, dL_dcolor and a channel count that is a multiple of 16");

    const ViewParams vp = make_view(viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, scale_modifier, width, height);

    const GeomPtrs geom = geom_from(geom_buffer, P);
    const BinPtrs bin = bin_from(binning_buffer, R);
    const ImgPtrs img = img_from(img_buffer, width, height);
    const float* color_ptr = (colors_precomp != nullptr) ? colors_precomp : geom.rgb;  // rasterizer_impl.cu:389
    {
        StageTimer t(stream, MI_STAGE_BLEND_BWD);
        // (MI_RAST_PREZERO_BWD: the forward of this view left the block zeroed, and no backward has run on it since)
        // (features only: the packed fields are not used, the work-queue counters behind them are)
        if (!(flags & MI_RAST_PREZERO_BWD))
            HIP_TRY(hipMemsetAsync(feat_only ? (void*)(geom.bwd_pack + (size_t)P * 8) : (void*)geom.bwd_pack, 0,
                                   bwd_pack_bytes(P) - (feat_only ? (size_t)P * 8 * sizeof(float) : 0), stream));
        uint32_t* queue_ctr = reinterpret_cast<uint32_t*>(geom.bwd_pack + (size_t)P * 8);   // one set of eight counters per channel block
        const float* bg_blk = background;
        const float* colors_blk = color_ptr;
        const float* dpix_blk = dL_dpix;
        float* dcolor_blk = dL_dcolor;
        int cstride = channels;
        int cr_blk = 0;   // channels of a partial block that exist (blend_bwd_wave.h: CR == 0)
        const uint32_t nt_ = vp.grid_x * vp.grid_y;
#define LAUNCH_BWD_WAVE_(XE, ST, ...)                                                                                         \
    hipLaunchKernelGGL((blend_bwd_wave_kernel<__VA_ARGS__, XE, ST>), dim3(32u * xcd_static_len_max(nt_) + 4u * xcd_queued_tiles_max(nt_)), dim3(64), 0,    \
                       stream, img.ranges, bin.blend_list, geom.index_rec, img.tile_nsurv, vp.W, vp.H, vp.grid_x, nt_, bg_blk, colors_blk, \
                       img.final_T, img.n_contrib, dpix_blk, dL_dout_mask, geom.bwd_pack, dcolor_blk, queue_ctr, cstride, cr_blk, img.run_bounds)
#define LAUNCH_BWD_WAVE_ST(ST, ...)                                                 \
    do {                                                                            \
        if (xexp) LAUNCH_BWD_WAVE_(true, ST, __VA_ARGS__);                          \
        else LAUNCH_BWD_WAVE_(false, ST, __VA_ARGS__);                              \
    } while (0)
// (the row stride is a compile-time constant unless the launch handles one channel block of a wider feature)
#define LAUNCH_BWD_WAVE(C_, CR_, MG_)                                               \
    do {                                                                            \
        if (cstride == CR_) LAUNCH_BWD_WAVE_ST(false, C_, CR_, MG_);                \
        else LAUNCH_BWD_WAVE_ST(true, C_, CR_, MG_);                                \
    } while (0)
#ifdef MI_RAST_PROFILING
        if (g_ablate & 1024) {  // the VALU kernels (timing comparisons)
            if (maskgrad) launch_blend_bwd<3, true>(vp, stream, img, bin, geom, color_ptr, background, dL_dpix, dL_dout_mask, dL_dcolor, xexp);
            else if (channels == 3) launch_blend_bwd<3, false>(vp, stream, img, bin, geom, color_ptr, background, dL_dpix, nullptr, dL_dcolor, xexp);
            else if (channels == 32) launch_blend_bwd<32, false>(vp, stream, img, bin, geom, color_ptr, background, dL_dpix, nullptr, dL_dcolor, xexp);
            else launch_blend_bwd<64, false>(vp, stream, img, bin, geom, color_ptr, background, dL_dpix, nullptr, dL_dcolor, xexp);
        } else
#endif
        if (feat_only) {
            // one launch per channel block of 64 / 32 / 16 channels (blend_bwd_feat.h: one wave per half tile): alpha, T, dF = W^T dL and
            // the feature-row atomics
#define LAUNCH_BWD_HALF_(C_, XE, ST)                                                                                             \
    hipLaunchKernelGGL((blend_bwd_feat_kernel<C_, XE, ST>), dim3(16u * xcd_static_len_max(nt_) + 2u * xcd_queued_tiles_max(nt_)), dim3(64), 0,  \
                       stream, img.ranges, bin.blend_list, geom.index_rec, img.tile_nsurv, vp.W, vp.H, vp.grid_x, nt_, img.final_T,             \
                       img.n_contrib, dpix_blk, dcolor_blk, queue_ctr, cstride, img.run_bounds)
#define LAUNCH_BWD_HALF(C_)                                                                  \
    do {                                                                                     \
        if (cstride == C_) { if (xexp) LAUNCH_BWD_HALF_(C_, true, false); else LAUNCH_BWD_HALF_(C_, false, false); } \
        else { if (xexp) LAUNCH_BWD_HALF_(C_, true, true); else LAUNCH_BWD_HALF_(C_, false, true); }                 \
    } while (0)
            const size_t HW = (size_t)width * height;
            for (int c0 = 0; c0 < channels;) {
                const int cb = channel_block(channels - c0);
                dpix_blk = dL_dpix + (size_t)c0 * HW;
                dcolor_blk = dL_dcolor + c0;
                if (cb == 64) LAUNCH_BWD_HALF(64);
                else if (cb == 32) LAUNCH_BWD_HALF(32);
                else LAUNCH_BWD_HALF(16);
                queue_ctr += 8 * XCD_QUEUE_STRIDE;
                c0 += cb;
            }
#undef LAUNCH_BWD_HALF
#undef LAUNCH_BWD_HALF_
        } else if (maskgrad) LAUNCH_BWD_WAVE(16, 3, true);
        else if (channels == 3) LAUNCH_BWD_WAVE(16, 3, false);
        else {
            // one launch per channel block: the feature gradient of the block, and the block's share of the geometry gradients
            // (dL/dalpha is a sum over channels), which the launches accumulate in the packed record
            const size_t HW = (size_t)width * height;
            for (int c0 = 0; c0 < channels;) {
                const int cb = channel_block(channels - c0);
                bg_blk = background + c0;
                colors_blk = color_ptr + c0;
                dpix_blk = dL_dpix + (size_t)c0 * HW;
                dcolor_blk = dL_dcolor + c0;
                cr_blk = std::min(cb, channels - c0);
                if (cb == 64) LAUNCH_BWD_WAVE(64, 64, false);
                else if (cb == 32) LAUNCH_BWD_WAVE(32, 32, false);
                else if (cr_blk == 16) LAUNCH_BWD_WAVE(16, 16, false);
                else LAUNCH_BWD_WAVE_ST(true, 16, 0, false);   // partial block: cr_blk of its 16 channels exist
                queue_ctr += 8 * XCD_QUEUE_STRIDE;
                c0 += cb;
            }
        }
#undef LAUNCH_BWD_WAVE
#undef LAUNCH_BWD_WAVE_ST
#undef LAUNCH_BWD_WAVE_
    }
    STAGE_CHECK("render backward");

    if (feat_only) return MI_RAST_OK;
    const float* cov3D_ptr = (cov3D_precomp != nullptr) ? cov3D_precomp : geom.cov3D;  // rasterizer_impl.cu:411
    {
        StageTimer t(stream, MI_STAGE_GEOM_BWD);
        hipLaunchKernelGGL(geometry_bwd_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, D, M, means3D, radii, shs,
                           geom.clamped, scales, rotations, cov3D_ptr, vp, geom.bwd_pack, dL_dmean2D, dL_dconic, dL_dopacity,
                           dL_dmask, dL_dmean3D, dL_dcolor, dL_dcov3D, dL_dsh, dL_dscale, dL_drot);
    }
    STAGE_CHECK("preprocess backward");
    return MI_RAST_OK;
}

int mi_rast_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                         uint8_t* present, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (P <= 0) return MI_RAST_OK;
    const ViewParams vp = make_view(viewmatrix, projmatrix, nullptr, 1.f, 1.f, 1.f, 16, 16);
    hipLaunchKernelGGL(check_frustum_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, means3D, vp, present);
    HIP_TRY(hipGetLastError());
    return MI_RAST_OK;
}

int mi_rast_mask_forward(mi_rast_resize_fn geometry_buffer, void* geometry_user, mi_rast_resize_fn binning_buffer,
                         void* binning_user, mi_rast_resize_fn image_buffer, void* image_user, int P, int width,
                         int height, const float* means3D, const float* opacities, const float* mask,
                         const float* scales, float scale_modifier, const float* rotations,
                         const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, float tan_fovx,
                         float tan_fovy, int prefiltered, float* out_mask, int* radii, int debug, int flags,
                         void* stream_, int* num_rendered)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (num_rendered) *num_rendered = 0;
    if (P <= 0 || width <= 0 || height <= 0) return fail(MI_RAST_ERR_INVALID, "P, width and height must be positive");
    if (!num_rendered || !radii || !out_mask || !mask) return fail(MI_RAST_ERR_INVALID, "null pointer");
    const ViewParams vp = make_view(viewmatrix, projmatrix, nullptr, tan_fovx, tan_fovy, scale_modifier, width, height);
    int rc;
    GeomPtrs geom;
    ImgPtrs img;
    BinPtrs bin;
    // DEPTH/cuda_rasterizer/rasterizer_impl.cu:495-521: shs = nullptr, colours "given" (dummy pointer)
    rc = geometry_and_binning(geometry_buffer, geometry_user, binning_buffer, binning_user, image_buffer, image_user, P,
                              0, 0, width, height, means3D, nullptr, 1, opacities, scales, rotations, cov3D_precomp, vp,
                              prefiltered, radii, debug, flags, stream, geom, img, bin, num_rendered);
    if (rc) return rc;
    {
        StageTimer t(stream, MI_STAGE_BLEND_FWD);
        launch_blend_fwd<0, 1>(vp, stream, img, bin, geom, nullptr, mask, nullptr, nullptr, out_mask, nullptr, (flags & MI_RAST_FAST_EXP) == 0);
    }
    STAGE_CHECK("render_mask");
    return MI_RAST_OK;
}

int mi_rast_mask_backward(int P, int R, int width, int height, char* geom_buffer, char* binning_buffer,
                          char* img_buffer, const float* dL_dout_mask, float* dL_dmask, int debug, int flags, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (P <= 0) return MI_RAST_OK;
    ViewParams vp;
    std::memset(&vp, 0, sizeof(vp));
    vp.W = width;
    vp.H = height;
    vp.grid_x = (width + TILE_X - 1) / TILE_X;
    vp.grid_y = (height + TILE_Y - 1) / TILE_Y;
    const GeomPtrs geom = geom_from(geom_buffer, P);
    const BinPtrs bin = bin_from(binning_buffer, R);
    const ImgPtrs img = img_from(img_buffer, width, height);
    {
        StageTimer t(stream, MI_STAGE_BLEND_BWD);
        HIP_TRY(hipMemsetAsync(geom.bwd_pack, 0, (size_t)P * 8 * sizeof(float), stream));
        launch_blend_bwd<0, true>(vp, stream, img, bin, geom, nullptr, nullptr, nullptr, dL_dout_mask, nullptr, (flags & MI_RAST_FAST_EXP) == 0);
        hipLaunchKernelGGL(unpack_mask_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, geom.bwd_pack, dL_dmask);
    }
    STAGE_CHECK("render_mask backward");
    return MI_RAST_OK;
}

}  // extern "C"
