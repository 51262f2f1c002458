/*
 * saga_rast_oracle.h -- CPU restatement ("oracle") of the SAGA / 3DGS differentiable
 * tile rasterizer.  TEST INFRASTRUCTURE ONLY.
 *
 * This library is the parity checker for the HIP product path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * package (seganygaussians_amd/) never imports, links or calls anything in oracle/.
 *
 * PARITY STATUS: PINNED against the reference implementation itself.  The reference ships no tests or
 * golden vectors, so the pin is an output-level one: oracle/build_ref.py
 * builds the reference's own rasterizer core (CF/, BASE/, DEPTH/ cuda_rasterizer/*.cu, translated
 * test-only with hipify-perl from the sources under /root/reference) into oracle/_ref/, and
 * tests/test_zz_reference_pin.py runs it on the MI355X next to this oracle: radii, tiles_touched,
 * depth / means2D bits, the sorted 64-bit key list, point_list, ranges and num_rendered bit-exact;
 * image, final_T, mask, depth and every gradient within 1e-4; up to the benchmarked size (1 M
 * Gaussians, 1080p, 32-D).  Secondary pins: an independent dense fp64 autograd re-derivation run
 * with the reference's own Python helpers imported (tests/dense_ref.py, tests/reference_helpers.py,
 * tests/golden/), analytic micro-scenes; see tests/test_oracle_*.py.
 *
 * Every function cites the reference file:line it restates, with
 *   CF/ = submodules/diff-gaussian-rasterization_contrastive_f/
 *   DEPTH/ = submodules/diff-gaussian-rasterization-depth/
 *
 * Arithmetic contract (also DESIGN.md "Numeric contract"): IEEE-754 binary32, every
 * operation individually rounded in the reference's source order, NO fused contraction
 * (build with -ffp-contract=off), correctly rounded sqrt and division, ndc2Pix in binary64.
 */
#ifndef SAGA_RAST_ORACLE_H
#define SAGA_RAST_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct saga_oracle_state saga_oracle_state;

/* Field ids for saga_oracle_field(). */
enum {
    SAGA_F_DEPTHS = 0,        /* float[P]   view-space z (valid where radii>0)            */
    SAGA_F_MEANS2D = 1,       /* float[2P]  pixel-space means                              */
    SAGA_F_COV3D = 2,         /* float[6P]  upper-triangular world covariance              */
    SAGA_F_CONIC_OPACITY = 3, /* float[4P]  conic.xyz, opacity                             */
    SAGA_F_RGB = 4,           /* float[3P]  SH-evaluated colour (only when shs given)      */
    SAGA_F_CLAMPED = 5,       /* uint8[3P]  colour clamp flags                             */
    SAGA_F_TILES_TOUCHED = 6, /* uint32[P]                                                 */
    SAGA_F_POINT_OFFSETS = 7, /* uint32[P]  inclusive scan of tiles_touched                */
    SAGA_F_KEYS_SORTED = 8,   /* uint64[R]  (tile<<32)|depth_bits, ascending, stable       */
    SAGA_F_POINT_LIST = 9,    /* uint32[R]  Gaussian ids in sorted order                   */
    SAGA_F_RANGES = 10,       /* uint32[2*tiles]  [start,end) per tile                     */
    SAGA_F_FINAL_T = 11,      /* float[W*H]                                                */
    SAGA_F_N_CONTRIB = 12,    /* uint32[W*H]                                               */
    SAGA_F_RADII = 13,        /* int32[P]                                                  */
    SAGA_F_KEYS_UNSORTED = 14,/* uint64[R]                                                 */
    SAGA_F_VALUES_UNSORTED = 15 /* uint32[R]                                               */
};

/* Counter ids for saga_oracle_counter() -- the quantities of SURVEY.md section 8(d). */
enum {
    SAGA_C_P = 0,   /* Gaussians                                                          */
    SAGA_C_V = 1,   /* visible (radius > 0)                                               */
    SAGA_C_R = 2,   /* num_rendered = sum tiles_touched                                   */
    SAGA_C_E = 3,   /* sum over tiles of list entries consumed before the tile finished   */
    SAGA_C_L = 4,   /* sum over tiles of max_pixel(n_contrib)                             */
    SAGA_C_PAIRS = 5,  /* pixel-Gaussian pairs that were blended (contributing pairs)     */
    SAGA_C_TILES = 6,
    SAGA_C_SORT_BITS = 7
};

/*
 * Forward.  Restates CudaRasterizer::Rasterizer::forward (CF/cuda_rasterizer/rasterizer_impl.cu:198-336)
 * and, when mask != NULL, DEPTH/cuda_rasterizer/rasterizer_impl.cu:198-343 (adds out_mask, out_depth).
 * All pointers are host pointers; optional inputs are NULL when absent (reference: empty tensor ->
 * nullptr, CF/diff_gaussian_rasterization_contrastive_f/__init__.py:196-206).
 * C is the channel count (reference compile-time NUM_CHANNELS).
 * Returns a state handle (never NULL unless out of memory); *rc_out: 0 ok, 1 = "For non-RGB,
 * provide precomputed Gaussian colors!" (rasterizer_impl.cu:242-245), 2 = prefiltered point was
 * culled (auxiliary.h:156-160, device trap in the reference).
 */
saga_oracle_state* saga_oracle_forward(
    int P, int D, int M, int C,
    const float* background, int W, int H,
    const float* means3D, const float* shs, const float* colors_precomp,
    const float* opacities, const float* scales, float scale_modifier,
    const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos,
    float tan_fovx, float tan_fovy, int prefiltered,
    const float* mask,      /* NULL for the BASE / CF rasterizers */
    float* out_color,       /* [C,H,W] */
    float* out_mask,        /* [H,W] or NULL */
    float* out_depth,       /* [H,W] or NULL */
    int* radii,             /* [P] */
    int* rc_out);

/*
 * Mask-only forward ("next" row; DEPTH/cuda_rasterizer/rasterizer_impl.cu:450-580,
 * DEPTH/cuda_rasterizer/forward.cu:390-498).
 */
saga_oracle_state* saga_oracle_mask_forward(
    int P, int W, int H,
    const float* means3D, const float* opacities, const float* mask,
    const float* scales, float scale_modifier, const float* rotations,
    const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
    float tan_fovx, float tan_fovy, int prefiltered,
    float* out_mask, int* radii, int* rc_out);

/*
 * Backward.  Restates Rasterizer::backward (CF/cuda_rasterizer/rasterizer_impl.cu:340-434;
 * DEPTH variant adds dL_dout_mask / dL_dmask).  Per-Gaussian sums that the reference forms with
 * float atomicAdd in nondeterministic order (backward.cu:525-556) are accumulated here in
 * binary64 when accum_double != 0 (the tolerance truth) or in binary32 in (tile, pixel, list)
 * order otherwise.  All output arrays must be zero-initialised by the caller exactly as
 * RasterizeGaussiansBackwardCUDA does (CF/rasterize_points.cu:151-159).
 */
void saga_oracle_backward(
    const saga_oracle_state* st,
    int P, int D, int M, int C,
    const float* background, int W, int H,
    const float* means3D, const float* shs, const float* colors_precomp,
    const float* scales, float scale_modifier, const float* rotations,
    const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
    const float* cam_pos, float tan_fovx, float tan_fovy,
    const float* dL_dpix,        /* [C,H,W] */
    const float* dL_dout_mask,   /* [H,W] or NULL */
    const float* mask,           /* unused by the gradient itself; non-NULL selects DEPTH variant */
    float* dL_dmean2D,           /* [P,3] */
    float* dL_dconic,            /* [P,4] */
    float* dL_dopacity,          /* [P]   */
    float* dL_dcolor,            /* [P,C] */
    float* dL_dmask,             /* [P] or NULL */
    float* dL_dmean3D,           /* [P,3] */
    float* dL_dcov3D,            /* [P,6] */
    float* dL_dsh,               /* [P,M,3] */
    float* dL_dscale,            /* [P,3] */
    float* dL_drot,              /* [P,4] */
    int accum_double);

/* Mask-only backward (DEPTH/cuda_rasterizer/backward.cu:568-660, rasterizer_impl.cu:587-635). */
void saga_oracle_mask_backward(
    const saga_oracle_state* st, int P, int W, int H,
    const float* dL_dout_mask, float* dL_dmask, int accum_double);

/* markVisible (CF/cuda_rasterizer/rasterizer_impl.cu:54-66,140-153). present: uint8[P]. */
void saga_oracle_mark_visible(int P, const float* means3D, const float* viewmatrix,
                              const float* projmatrix, uint8_t* present);

/* getHigherMsb (CF/cuda_rasterizer/rasterizer_impl.cu:35-50). */
uint32_t saga_oracle_get_higher_msb(uint32_t n);

const void* saga_oracle_field(const saga_oracle_state* st, int field, size_t* count);
int64_t saga_oracle_counter(const saga_oracle_state* st, int counter);
void saga_oracle_free(saga_oracle_state* st);

/* Number of OpenMP threads the oracle will use (for bench.py's cpu_baseline "cores"). */
int saga_oracle_num_threads(void);
void saga_oracle_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
