/*
 * mi_rast.h -- C-ABI of the MI355X-native differentiable Gaussian-splatting (feature) rasterizer.
 *
 * This is the drop-in boundary for the reference's native rasterizer core.  Each entry point
 * replaces one static method of CudaRasterizer::Rasterizer
 *   (CF/cuda_rasterizer/rasterizer.h:24-84, DEPTH/cuda_rasterizer/rasterizer.h),
 * where CF/ = submodules/diff-gaussian-rasterization_contrastive_f/ and
 * DEPTH/ = submodules/diff-gaussian-rasterization-depth/ of Jumpat/SegAnyGAussians.
 * The torch glue the reference keeps in rasterize_points.cu (tensor allocation, P==0 short
 * circuit, M = sh.size(1)) lives in Python on our side (seganygaussians_amd/rasterizer.py).
 *
 * Conventions
 *  - plain pointers and sizes only; no torch / C++ types; all data pointers are DEVICE pointers
 *    (HBM) unless marked [host]; optional inputs are NULL when absent, exactly like the
 *    reference's "empty tensor -> nullptr" convention (CF/.../__init__.py:196-206).
 *  - fp32, contiguous; out_color / dL_dpix are CHW; colors_precomp is (P,C); shs is (P,M,3).
 *  - `channels` is the reference's compile-time NUM_CHANNELS (config.h:15 = 3,
 *    config_contrastive_f.h:15 = 32), a RUN-TIME argument here; supported: every width from 1 to 256
 *    (mi_rast_supported_channels()).  32 and 64 run in one pass of the blend kernels, other widths in channel blocks, one pass each,
 *    and what is left behind the last whole block is a PARTIAL block (zeros in the matrix operands behind the channels that exist,
 *    loads / stores / atomics predicated on the channel index -- how RGB, 3 of 16, has always been rendered); the block widths
 *    differ per direction:
 *      forward : blocks of 64 and 32, the remainder a partial 32-channel block   (100 = 64 + 32 + 4 of 32;  8 = 8 of 32)
 *      backward: blocks of 64, 32 and 16, the remainder a partial 16-channel one (100 = 64 + 32 + 4 of 16; 40 = 32 + 8 of 16)
 *    Rows of a feature whose width is no multiple of 4 are not 16-byte aligned, which the vector loads of the full blocks tolerate
 *    at reduced speed; 0 or more than 256 returns MI_RAST_ERR_INVALID.
 *  - `mask != NULL` selects the DEPTH variant (adds out_mask/out_depth, dL_dmask).
 *  - `stream` is a hipStream_t (0 = null stream).  The rendering entry points are RE-ENTRANT: they keep no
 *    state between calls (every mode is a per-call argument: `flags`, `features_ready_event`), all memory is
 *    owned by the caller, and the three opaque buffers have a private layout that is self-contained given
 *    (P, W, H, R) and may be handed back verbatim to the backward call.  Concurrent calls from different host
 *    threads, on different streams and on different devices are safe (the only internal resources are a
 *    pinned word + two events per (host thread, device) for the num_rendered read-back).  The one
 *    process-level switch is the optional per-stage timing (mi_rast_profile_*), a measurement aid that must
 *    not be toggled while calls are in flight.
 *  - every function returns 0 on success; on failure a nonzero code, and mi_rast_last_error()
 *    (thread-local) holds the message.  MI_RAST_ERR_NON_RGB carries the reference's text
 *    "For non-RGB, provide precomputed Gaussian colors!" (rasterizer_impl.cu:242-245).
 */
#ifndef MI_RAST_H
#define MI_RAST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_RAST_OK 0
#define MI_RAST_ERR_INVALID 1    /* bad argument (shape / unsupported channel count) */
#define MI_RAST_ERR_NON_RGB 2    /* channels != 3 and colors_precomp == NULL          */
#define MI_RAST_ERR_HIP 3        /* a HIP runtime call or kernel failed               */
#define MI_RAST_ERR_ALLOC 4      /* a resize callback returned NULL                   */

/* `flags` of mi_rast_forward / mi_rast_mask_forward (0 = product default). */
#define MI_RAST_FULL_LISTS 1   /* also materialise the reference's point_list and full-list positions (see below) */
#define MI_RAST_NO_CULL 4      /* testing aid: the exact-conservative cull is switched off -- every overlap of the reference's
                                  tile lists reaches the blend kernels with all four quadrant bits (implies full lists).
                                  Results must not change: images bit for bit, gradients up to the order of the atomic sums */
#define MI_RAST_FAST_EXP 8     /* exp() of the blend kernels as v_exp_f32(x * log2e) (~5 ulp) instead of the device library's expf
                                  (<= 1 ulp, what the reference's kernels call and the product default): +2 % views/s on cfg3, but a
                                  few pairs that sit within a few ulp of the alpha >= 1/255 cut land on the other side of it than in
                                  the reference (n_contrib differs on ~1e-5 of the pixels, the per-row gradient noise is ~10x the
                                  reference's own; still inside the 1e-4 contract).  The backward MUST be given the flags of its
                                  forward: it re-takes the same decisions */
#define MI_RAST_EXACT_EXP 128  /* forward blend: the device library's expf for EVERY pair.  Product default (flag clear): the hybrid form of
                                  csrc/common.h -- v_exp_f32(x * log2e) away from the alpha >= 1/255 cut, and the whole 16-entry GROUP
                                  re-evaluated with expf when some pixel of it comes within 4e-6 (relative) of the cut (per entry in the
                                  RGB kernel).  What that protects is the alpha >= 1/255 decision only: those are expf's always (the
                                  backward, which uses expf, re-takes exactly them).  The T < 1e-4 STOP test runs on the v_exp_f32 alphas
                                  (<= 1e-6 relative off, accumulated in T), so n_contrib / final_T can differ from a build of the
                                  reference's kernels: on ~1e-6 of the pixels of the benchmark scenes (tests allow 5e-6), and on up to
                                  7.9e-5 of the pixels of an opaque scene of FAINT Gaussians, where many pixels end right at the 1e-4
                                  threshold (measured worst case of the randomised sweeps, tools/fuzz_parity.py; the accumulated error
                                  of T cannot be repaired by re-evaluating one group, unlike the alpha cut): the default is NOT
                                  bit-identical on the image-state fields, the image differs by ~1e-7 of its scale.  With the flag, alpha / T / n_contrib / final_T are
                                  bit-identical to a build of the reference's kernels (tests) */
#define MI_RAST_VERIFY_LISTS 16 /* debugging aid (lean lists only; synchronous): zero-fills the list entries before the emit pass and
                                * fails with MI_RAST_ERR_HIP if a slot the count pass reserved was not written by the emit pass */
#define MI_RAST_TILE_FWD 32    /* 32/64-channel forward on the tile-batched bf16x3 kernel (four lockstep waves per tile, blend_fwd_x3.h)
                                  instead of the wave-per-quadrant kernel (blend_fwd_wave.h); same alpha/T/n_contrib bit for bit */
#define MI_RAST_PREZERO_BWD 64  /* forward: leave the backward's packed gradient scratch (in the geometry buffer) zero-filled -- the wave-per-quadrant
                                  blend kernel stores the zeros beside its own work; backward: that was done by the forward of this view and no
                                  backward has run on its buffers since, so the fill command is skipped.  Pass it to ONE backward per forward. */
#define MI_RAST_F32_BLEND 2    /* 32/64-channel forward on the f32 FMA-chain kernel instead of the exactly split bf16x3
                                  matrix kernel (same alpha/T/n_contrib bit for bit; images agree to a few ulp) */
#define MI_RAST_EQUAL_RUNS 256 /* A/B aid: both blend kernels give every XCD (its own L2) one contiguous run of tiles; by default the runs
                                  are cut at equal MODELLED work -- sum over the tiles of min(list length, 768) + 128, csrc/binning.h:
                                  tile_ranges_kernel -- because the dispatcher deals workgroups to the XCDs whatever their progress and
                                  real scenes' density varies over the image; with this flag at equal tile counts (rounds 2-4).
                                  Results are the same either way (the order of the atomic sums apart) */
#define MI_RAST_BWD_FEATURES_ONLY 512 /* EXTENSION, mi_rast_backward only: the caller wants dL_dcolor and NOTHING else -- SAGA's contrastive feature
                                  training optimises the feature rows alone (scene/gaussian_model_ff.py:154-162); the reference computes the geometry
                                  gradients all the same and nobody reads them.  dL_dcolor[g] = sum over the pairs of alpha T dL_dpix needs alpha and T
                                  of every pair and no more: the backward blend then reads no feature row, forms no dL/dalpha, no moments, issues no
                                  packed-field atomics (a third of its atomic requests), and the per-Gaussian geometry backward does not run.  Every
                                  other dL_d* pointer may be NULL and is left untouched.  Needs colors_precomp != NULL and a channel count that is a
                                  multiple of 16 (mi_rast_features_only_supported); dL_dcolor equals the default mode's up to the order of the atomic
                                  sums.  seganygaussians_amd/rasterizer.py: enable_features_only_backward() / MI_RAST_FEATURES_ONLY_BACKWARD=1 */

/* Replaces std::function<char*(size_t)> (CF/rasterize_points.cu:27-33): must return a device
 * pointer to at least nbytes bytes (256-B aligned), valid until the caller frees it. */
typedef char* (*mi_rast_resize_fn)(size_t nbytes, void* user);

/* Replaces CudaRasterizer::Rasterizer::forward (CF/cuda_rasterizer/rasterizer_impl.cu:198-336,
 * declaration rasterizer.h:34-59; DEPTH variant DEPTH/cuda_rasterizer/rasterizer_impl.cu:198-343).
 * Stages: preprocess -> count pass + scans over (tile, slice) -> (host reads num_rendered) -> emission of
 * {depth bits, id | quadrant mask} entries -> per-tile sort by (depth bits, id) -> per-tile alpha blend; the resulting
 * point_list / ranges are identical to the reference's 64-bit (tile|depth) global sort.  Writes EVERY element of out_color (and out_mask /
 * out_depth), so the caller need not zero-fill them.  *num_rendered [host] receives R. */
int mi_rast_forward(
    mi_rast_resize_fn geometry_buffer, void* geometry_user,
    mi_rast_resize_fn binning_buffer, void* binning_user,
    mi_rast_resize_fn image_buffer, void* image_user,
    int P, int D, int M, int channels,
    const float* background,
    int width, int height,
    const float* means3D,
    const float* shs,
    const float* colors_precomp,
    const float* opacities,
    const float* scales,
    float scale_modifier,
    const float* rotations,
    const float* cov3D_precomp,
    const float* viewmatrix,
    const float* projmatrix,
    const float* cam_pos,
    float tan_fovx, float tan_fovy,
    int prefiltered,
    const float* mask,      /* DEPTH variant only, else NULL */
    float* out_color,       /* [channels, H, W] */
    float* out_mask,        /* [H, W]  (DEPTH variant) */
    float* out_depth,       /* [H, W]  (DEPTH variant) */
    int* radii,             /* [P] */
    int debug,
    int flags,                    /* MI_RAST_* bits above (0 = product default); `debug` implies MI_RAST_FULL_LISTS */
    void* features_ready_event,   /* hipEvent_t or NULL: see "List modes and the features-ready event" below */
    float* dL_dcolor_next,        /* [P, channels] or NULL: the buffer this view's mi_rast_backward will accumulate dL_dcolor into.  When not
                                     NULL the forward leaves it ZERO-FILLED (what torch::zeros does in CF/rasterize_points.cu:153): the
                                     blend kernel, which is bound by VALU issue, stores the zeros beside its own work instead of a
                                     separate 4 P channels-byte fill pass in front of the backward (while the zeros are no more
                                     than the image bytes it stores anyway; a larger buffer is filled by a fill command here) */
    void* stream,
    int* num_rendered /* [host] */);

/* List modes and the features-ready event (both per call).
 * flags & MI_RAST_FULL_LISTS == 0 (default): "lean" -- only the (Gaussian, tile) overlaps that pass the
 *    exact-conservative cull are listed and sorted; list positions (and n_contrib, tile_consumed) count the entries of
 *    those lists.  Every output of the reference API (images, radii, gradients, num_rendered) is identical to the full mode's.
 * flags & MI_RAST_FULL_LISTS: "full" -- every overlap of the reference's rects is listed (the culled ones with an empty
 *    quadrant mask, which the blend kernels skip): the lists are the reference's point_list
 *    (CF/cuda_rasterizer/rasterizer_impl.cu:300-317) and positions its full-list positions, so that the integer path can be
 *    compared bit-exactly with the oracle / the reference.
 * features_ready_event: training loops in which the geometry is frozen and only colors_precomp (the feature rows) is
 *    optimised -- SAGA's contrastive feature training, scene/gaussian_model_ff.py:154-162 -- may start a forward before
 *    the features are final: preprocess, binning and the per-tile sort read the geometry only.  When not NULL,
 *    the call makes `stream` wait for this hipEvent_t (recorded by the caller when colors_precomp is ready, e.g. after the
 *    gradient all-reduce and the optimizer step) right before its blend stage.  The caller keeps the event alive until the
 *    call returns; the library does not store it. */

/* Replaces CudaRasterizer::Rasterizer::backward (CF/cuda_rasterizer/rasterizer_impl.cu:340-434,
 * declaration rasterizer.h:61-84; DEPTH variant adds dL_dout_mask / dL_dmask).  dL_dcolor and dL_dsh are
 * accumulated into and must be zero on entry, as RasterizeGaussiansBackwardCUDA makes them with torch::zeros
 * (CF/rasterize_points.cu:151-159).  Every other dL_d* output is written in full (zeros for Gaussians that were not
 * rendered): the caller need not clear them -- 96 bytes per Gaussian less to write than the reference's glue does. */
int mi_rast_backward(
    int P, int D, int M, int channels, int R,
    const float* background,
    int width, int height,
    const float* means3D,
    const float* shs,
    const float* colors_precomp,
    const float* scales,
    float scale_modifier,
    const float* rotations,
    const float* cov3D_precomp,
    const float* viewmatrix,
    const float* projmatrix,
    const float* campos,
    float tan_fovx, float tan_fovy,
    const int* radii,
    char* geom_buffer,
    char* binning_buffer,
    char* img_buffer,
    const float* dL_dpix,        /* [channels, H, W] */
    const float* dL_dout_mask,   /* [H, W] (DEPTH variant) or NULL */
    float* dL_dmean2D,           /* [P,3] */
    float* dL_dconic,            /* [P,4] */
    float* dL_dopacity,          /* [P]   */
    float* dL_dcolor,            /* [P,channels] */
    float* dL_dmask,             /* [P] (DEPTH variant) or NULL */
    float* dL_dmean3D,           /* [P,3] */
    float* dL_dcov3D,            /* [P,6] */
    float* dL_dsh,               /* [P,M,3] */
    float* dL_dscale,            /* [P,3] */
    float* dL_drot,              /* [P,4] */
    int debug,
    int flags,                   /* the flags of the forward call that produced the buffers (MI_RAST_FAST_EXP matters) */
    void* stream);

/* EXTENSION, no counterpart in the reference: the blend stage of a forward alone, for a caller that renders the SAME geometry from
 * the SAME camera again with other colours / features -- SAGA's contrastive feature training optimises the feature rows only
 * (scene/gaussian_model_ff.py:154-162) and revisits each of its ~200 cameras ~50 times (train_contrastive_feature.py:231), so
 * preprocess, binning and the per-tile sort of a revisit reproduce what the first visit computed, bit for bit.
 *   geom_buffer: the geometry buffer a previous mi_rast_forward of that geometry, camera, image size and list mode filled (read-only
 *       here, apart from the backward's packed-gradient scratch under MI_RAST_PREZERO_BWD);
 *   binning_buffer: that forward's binning buffer, or a copy of its blend list alone -- mi_rast_binning_layout puts MI_BIN_BLEND_LIST
 *       first, and the words [0, 4 n) of it, n = the entries the lists hold (word 0 of cached_words), are all a blend kernel reads;
 *   cached_ranges: that forward's tile ranges (MI_IMG_RANGES of its image buffer, 8 bytes per tile), cached_words: the 16 words at
 *       MI_IMG_NUM_RENDERED + 8192 bytes of it ({lean entries, longest list, -, key bits, XCD run boundaries}) -- or copies of both;
 *   img_buffer: a fresh buffer of mi_rast_image_layout(width, height) bytes for THIS view's per-pixel state (final_T, n_contrib,
 *       walk counters): hand it -- with geom_buffer / binning_buffer -- to this view's mi_rast_backward;
 *   R: that forward's num_rendered; longest_run: mi_rast_last_longest_run() taken right behind it (0: unknown -- a larger grid);
 *   colors_precomp == NULL (channels == 3): the RGB colours that forward evaluated from its SHs (geometry buffer).
 * Everything else as mi_rast_forward.  Deciding WHEN the state may be reused is the caller's business (seganygaussians_amd/
 * rasterizer.py: GeometryCache keys it on content fingerprints of the geometry and camera tensors, mi_rast_fingerprint). */
int mi_rast_forward_reuse(
    int P, int channels, int R,
    const float* background,
    int width, int height,
    const float* colors_precomp,
    char* geom_buffer, char* binning_buffer, const void* cached_ranges, const int* cached_words, char* img_buffer,
    int longest_run,
    const float* mask, float* out_color, float* out_mask, float* out_depth,
    int flags, void* features_ready_event, float* dL_dcolor_next, void* stream);
int mi_rast_last_longest_run(void);
/* 1 if mi_rast_backward honours MI_RAST_BWD_FEATURES_ONLY for this channel count (a multiple of 16), else 0. */
int mi_rast_features_only_supported(int channels);
/* 64-bit content fingerprints of n <= 8 device arrays of 4-byte words (position-dependent word hashes, summed): out[k] [host].
 * Synchronous (one kernel on `stream`, then the calling thread waits for the stream). */
int mi_rast_fingerprint(int n, const void* const* ptrs, const size_t* nbytes, uint64_t* out /* [host] */, void* stream);

/* Replaces CudaRasterizer::Rasterizer::markVisible (CF/cuda_rasterizer/rasterizer_impl.cu:140-153).
 * present: one byte (0/1) per Gaussian == torch.bool storage. */
int mi_rast_mark_visible(int P, const float* means3D, const float* viewmatrix,
                         const float* projmatrix, uint8_t* present, void* stream);

/* Replaces CudaRasterizer::Rasterizer::mask_forward / mask_backward
 * (DEPTH/cuda_rasterizer/rasterizer_impl.cu:450-580, 587-635): mask-only render pair. */
int mi_rast_mask_forward(
    mi_rast_resize_fn geometry_buffer, void* geometry_user,
    mi_rast_resize_fn binning_buffer, void* binning_user,
    mi_rast_resize_fn image_buffer, void* image_user,
    int P, int width, int height,
    const float* means3D, const float* opacities, const float* mask,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix,
    float tan_fovx, float tan_fovy, int prefiltered,
    float* out_mask, int* radii, int debug, int flags, void* stream, int* num_rendered /* [host] */);

int mi_rast_mask_backward(
    int P, int R, int width, int height,
    char* geom_buffer, char* binning_buffer, char* img_buffer,
    const float* dL_dout_mask, float* dL_dmask, int debug, int flags, void* stream);

/* ---- introspection (used by the parity tests and bench.py; not part of the reference API) ---- */

const char* mi_rast_last_error(void);
const char* mi_rast_version(void);
/* Writes up to n supported channel counts into out, returns how many exist. */
int mi_rast_supported_channels(int* out, int n);
/* Reference helper getHigherMsb (CF/cuda_rasterizer/rasterizer_impl.cu:35-50), host side. */
uint32_t mi_rast_get_higher_msb(uint32_t n);
/* Private-layout maps of the three opaque buffers, so tests can compare the integer path
 * bit-exactly with the oracle.  Each call fills `offsets` (bytes from the buffer start) for the
 * fields listed, and returns the total size in bytes. */
enum { MI_GEOM_DEPTHS = 0, MI_GEOM_MEANS2D, MI_GEOM_CONIC_OPACITY, MI_GEOM_COV3D, MI_GEOM_RGB,
       MI_GEOM_CLAMPED, MI_GEOM_TILES_TOUCHED, MI_GEOM_DEPTH_KEY, MI_GEOM_INDEX_REC,
       MI_GEOM_CULL_COUNTER, MI_GEOM_BAND_BITS, MI_GEOM_BWD_PACK, MI_GEOM_NFIELDS };
enum { MI_IMG_FINAL_T = 0, MI_IMG_N_CONTRIB, MI_IMG_RANGES, MI_IMG_TILE_CONSUMED, MI_IMG_TILE_COUNT,
       MI_IMG_TILE_CURSOR, MI_IMG_NUM_RENDERED, MI_IMG_TILE_NSURV, MI_IMG_NFIELDS };
/* MI_BIN_BLEND_LIST: u32[R], per tile at ranges[tile].x in depth order: Gaussian id | quadrant mask << 28 -- all the blend kernels
 * read of a tile's list (they gather the 32-byte record MI_GEOM_INDEX_REC[id] next to the feature row).  Full lists: every overlap
 * of the reference's rects is an entry (culled ones with mask 0), so the low 28 bits ARE the reference's point_list. */
enum { MI_BIN_ENTRIES = 0, MI_BIN_SCRATCH, MI_BIN_BLEND_LIST, MI_BIN_NFIELDS };
size_t mi_rast_geometry_layout(int P, size_t* offsets /* [MI_GEOM_NFIELDS] */);
size_t mi_rast_image_layout(int width, int height, size_t* offsets /* [MI_IMG_NFIELDS] */);
size_t mi_rast_binning_layout(int R, size_t* offsets /* [MI_BIN_NFIELDS] */);

/* Per-stage HIP-event timing on the caller's stream (bench.py's live roofline measurement).
 * When enabled (process-wide switch; toggle it only while no call is in flight), forward/backward record
 * events between stages (one event set per device); mi_rast_profile_read synchronises the current device's
 * events of the last call and returns elapsed milliseconds per stage. */
enum { MI_STAGE_PREPROCESS = 0, MI_STAGE_TILE_SCAN, MI_STAGE_EMIT, MI_STAGE_TILE_SORT,
       MI_STAGE_BLEND_FWD, MI_STAGE_BLEND_BWD, MI_STAGE_GEOM_BWD, MI_STAGE_COUNT };
int mi_rast_profile_enable(int on);
int mi_rast_profile_read(float* ms /* [MI_STAGE_COUNT] */);

#ifdef __cplusplus
}
#endif
#endif
