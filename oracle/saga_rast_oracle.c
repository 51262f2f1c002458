/*
 * saga_rast_oracle.c -- CPU restatement of the SAGA / 3DGS differentiable tile rasterizer.
 * TEST INFRASTRUCTURE ONLY (see saga_rast_oracle.h).  PARITY STATUS: pinned against the reference
 * implementation itself (oracle/_ref, tests/test_zz_reference_pin.py) -- details in the header.
 *
 * Citation prefixes: CF/ = submodules/diff-gaussian-rasterization_contrastive_f/,
 * DEPTH/ = submodules/diff-gaussian-rasterization-depth/ (both under /root/reference).
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC (oracle/Makefile).
 * -ffp-contract=off is part of the numeric contract: every binary32 op is rounded on its own.
 */
#include "saga_rast_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BLOCK_X 16 /* CF/cuda_rasterizer/config_contrastive_f.h:16 */
#define BLOCK_Y 16 /* CF/cuda_rasterizer/config_contrastive_f.h:17 */
#define BLOCK_SIZE (BLOCK_X * BLOCK_Y)

/* CF/cuda_rasterizer/auxiliary.h:21-39 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

struct saga_oracle_state {
    int P, W, H, C, tiles_x, tiles_y;
    int64_t R, V, E, L, pairs;
    int sort_bits;
    float* depths;
    float* means2D;
    float* cov3D;
    float* conic_opacity;
    float* rgb;
    uint8_t* clamped;
    uint32_t* tiles_touched;
    uint32_t* point_offsets;
    int32_t* radii;
    uint64_t* keys_unsorted;
    uint32_t* values_unsorted;
    uint64_t* keys_sorted;
    uint32_t* point_list;
    uint32_t* ranges;
    float* final_T;
    uint32_t* n_contrib;
    int has_shs; /* colours came from SH evaluation (feature pointer = rgb) */
};

typedef struct { float x, y, z; } f3;
typedef struct { float x, y, z, w; } f4;
typedef float mat3[3][3]; /* glm layout: m[col][row] */

/* float -> int32 with the saturating semantics both nvcc (cvt.rzi.s32.f32) and gfx950
 * (v_cvt_i32_f32) implement; NaN -> 0. */
static inline int32_t f2i_sat(float f)
{
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT32_MAX;
    if (f <= -2147483648.0f) return INT32_MIN;
    return (int32_t)f;
}

/* glm::mat3 * glm::mat3, CF/third_party/glm/glm/detail/type_mat3x3.inl:486-520 */
static inline void m3mul(const mat3 A, const mat3 B, mat3 Rr)
{
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++)
            Rr[c][r] = A[0][r] * B[c][0] + A[1][r] * B[c][1] + A[2][r] * B[c][2];
}
static inline void m3transpose(const mat3 A, mat3 Rr)
{
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++)
            Rr[c][r] = A[r][c];
}

/* CF/cuda_rasterizer/auxiliary.h:41-44 -- double literals: evaluated in binary64, rounded once */
static inline float ndc2Pix(float v, int S)
{
    return (float)(((v + 1.0) * S - 1.0) * 0.5);
}

/* CF/cuda_rasterizer/auxiliary.h:46-56 */
static inline void getRect(float px, float py, int max_radius, uint32_t* rmin, uint32_t* rmax,
                           uint32_t gx, uint32_t gy)
{
    int v;
    v = f2i_sat((px - (float)max_radius) / (float)BLOCK_X); v = v > 0 ? v : 0;
    rmin[0] = gx < (uint32_t)v ? gx : (uint32_t)v;
    v = f2i_sat((py - (float)max_radius) / (float)BLOCK_Y); v = v > 0 ? v : 0;
    rmin[1] = gy < (uint32_t)v ? gy : (uint32_t)v;
    v = f2i_sat((px + (float)max_radius + (float)BLOCK_X - (float)1) / (float)BLOCK_X); v = v > 0 ? v : 0;
    rmax[0] = gx < (uint32_t)v ? gx : (uint32_t)v;
    v = f2i_sat((py + (float)max_radius + (float)BLOCK_Y - (float)1) / (float)BLOCK_Y); v = v > 0 ? v : 0;
    rmax[1] = gy < (uint32_t)v ? gy : (uint32_t)v;
}

/* CF/cuda_rasterizer/auxiliary.h:58-66 */
static inline f3 transformPoint4x3(f3 p, const float* m)
{
    f3 t = {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
            m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
    return t;
}
/* CF/cuda_rasterizer/auxiliary.h:68-77 */
static inline f4 transformPoint4x4(f3 p, const float* m)
{
    f4 t = {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
            m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14],
            m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]};
    return t;
}
/* CF/cuda_rasterizer/auxiliary.h:89-97 */
static inline f3 transformVec4x3Transpose(f3 p, const float* m)
{
    f3 t = {m[0] * p.x + m[1] * p.y + m[2] * p.z,
            m[4] * p.x + m[5] * p.y + m[6] * p.z,
            m[8] * p.x + m[9] * p.y + m[10] * p.z};
    return t;
}
/* CF/cuda_rasterizer/auxiliary.h:107-117 */
static inline f3 dnormvdv3(f3 v, f3 dv)
{
    float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    f3 r;
    r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
    r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
    r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
    return r;
}

/* CF/cuda_rasterizer/auxiliary.h:139-164.  Returns 1 visible, 0 culled. */
static inline int in_frustum(int idx, const float* orig_points, const float* viewmatrix,
                             const float* projmatrix, f3* p_view)
{
    f3 p_orig = {orig_points[3 * idx], orig_points[3 * idx + 1], orig_points[3 * idx + 2]};
    /* p_hom / p_proj are computed by the reference but do not influence the test (the x/y clause is
     * commented out at auxiliary.h:154) */
    (void)projmatrix;
    *p_view = transformPoint4x3(p_orig, viewmatrix);
    if (p_view->z <= 0.2f) return 0;
    return 1;
}

/* CF/cuda_rasterizer/rasterizer_impl.cu:35-50 */
uint32_t saga_oracle_get_higher_msb(uint32_t n)
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

/* CF/cuda_rasterizer/forward.cu:121-155 */
static void computeCov3D(const float* scale, float mod, const float* rot, float* cov3D)
{
    mat3 S = {{1.0f, 0, 0}, {0, 1.0f, 0}, {0, 0, 1.0f}};
    S[0][0] = mod * scale[0];
    S[1][1] = mod * scale[1];
    S[2][2] = mod * scale[2];
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3]; /* not re-normalised (forward.cu:130) */
    mat3 Rm = {{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
               {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
               {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
    mat3 M, Mt, Sigma;
    m3mul(S, Rm, M);
    m3transpose(M, Mt);
    m3mul(Mt, M, Sigma);
    cov3D[0] = Sigma[0][0];
    cov3D[1] = Sigma[0][1];
    cov3D[2] = Sigma[0][2];
    cov3D[3] = Sigma[1][1];
    cov3D[4] = Sigma[1][2];
    cov3D[5] = Sigma[2][2];
}

/* Shared by forward cov2D (CF/cuda_rasterizer/forward.cu:77-116) and its backward
 * (CF/cuda_rasterizer/backward.cu:161-196): builds t (clamped), T = W*J, Vrk, cov2D (+0.3). */
static void cov2d_common(f3 mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy,
                         const float* cov3D, const float* vm, f3* t_out, float* txtz_o,
                         float* tytz_o, mat3 T, mat3 Vrk, mat3 Wm, mat3 cov)
{
    f3 t = transformPoint4x3(mean, vm);
    const float limx = 1.3f * tan_fovx;
    const float limy = 1.3f * tan_fovy;
    const float txtz = t.x / t.z;
    const float tytz = t.y / t.z;
    t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
    mat3 J = {{focal_x / t.z, 0.0f, -(focal_x * t.x) / (t.z * t.z)},
              {0.0f, focal_y / t.z, -(focal_y * t.y) / (t.z * t.z)},
              {0, 0, 0}};
    Wm[0][0] = vm[0]; Wm[0][1] = vm[4]; Wm[0][2] = vm[8];
    Wm[1][0] = vm[1]; Wm[1][1] = vm[5]; Wm[1][2] = vm[9];
    Wm[2][0] = vm[2]; Wm[2][1] = vm[6]; Wm[2][2] = vm[10];
    m3mul(Wm, J, T);
    Vrk[0][0] = cov3D[0]; Vrk[0][1] = cov3D[1]; Vrk[0][2] = cov3D[2];
    Vrk[1][0] = cov3D[1]; Vrk[1][1] = cov3D[3]; Vrk[1][2] = cov3D[4];
    Vrk[2][0] = cov3D[2]; Vrk[2][1] = cov3D[4]; Vrk[2][2] = cov3D[5];
    mat3 Tt, Vt, tmp;
    m3transpose(T, Tt);
    m3transpose(Vrk, Vt);
    m3mul(Tt, Vt, tmp);
    m3mul(tmp, T, cov);
    cov[0][0] += 0.3f;
    cov[1][1] += 0.3f;
    *t_out = t;
    *txtz_o = txtz;
    *tytz_o = tytz;
}

/* CF/cuda_rasterizer/forward.cu:23-74 */
static f3 computeColorFromSH(int idx, int deg, int max_coeffs, const float* means, const float* campos,
                             const float* shs, uint8_t* clamped)
{
    f3 pos = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
    f3 dir = {pos.x - campos[0], pos.y - campos[1], pos.z - campos[2]};
    float len = sqrtf(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
    dir.x = dir.x / len; dir.y = dir.y / len; dir.z = dir.z / len;
    const float* sh = shs + (size_t)idx * max_coeffs * 3;
    float res[3];
    float x = dir.x, y = dir.y, z = dir.z;
    for (int c = 0; c < 3; c++) {
#define SH(k) sh[(k) * 3 + c]
        float result = SH_C0 * SH(0);
        if (deg > 0) {
            result = result - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z;
                float xy = x * y, yz = y * z, xz = x * z;
                result = result + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) +
                         SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) + SH_C2[3] * xz * SH(7) +
                         SH_C2[4] * (xx - yy) * SH(8);
                if (deg > 2) {
                    result = result + SH_C3[0] * y * (3.0f * xx - yy) * SH(9) +
                             SH_C3[1] * xy * z * SH(10) +
                             SH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
                             SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                             SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) +
                             SH_C3[5] * z * (xx - yy) * SH(14) +
                             SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
                }
            }
        }
#undef SH
        result += 0.5f;
        clamped[3 * idx + c] = (result < 0);
        res[c] = fmaxf(result, 0.0f);
    }
    f3 r = {res[0], res[1], res[2]};
    return r;
}

/* CF/cuda_rasterizer/forward.cu:159-259 (one Gaussian).  Returns 0 ok, 2 prefiltered trap. */
static int preprocess_one(saga_oracle_state* st, int idx, int D, int M, int C, const float* orig_points,
                          const float* scales, float scale_modifier, const float* rotations,
                          const float* opacities, const float* shs, const float* cov3D_precomp,
                          int colors_given, const float* viewmatrix, const float* projmatrix,
                          const float* cam_pos, int W, int H, float tan_fovx, float tan_fovy,
                          float focal_x, float focal_y, int prefiltered)
{
    (void)C;
    st->radii[idx] = 0;
    st->tiles_touched[idx] = 0;
    f3 p_view;
    if (!in_frustum(idx, orig_points, viewmatrix, projmatrix, &p_view)) return prefiltered ? 2 : 0;

    f3 p_orig = {orig_points[3 * idx], orig_points[3 * idx + 1], orig_points[3 * idx + 2]};
    f4 p_hom = transformPoint4x4(p_orig, projmatrix);
    float p_w = 1.0f / (p_hom.w + 0.0000001f);
    f3 p_proj = {p_hom.x * p_w, p_hom.y * p_w, p_hom.z * p_w};

    const float* cov3D;
    if (cov3D_precomp != NULL) {
        cov3D = cov3D_precomp + (size_t)idx * 6;
    } else {
        computeCov3D(scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx,
                     st->cov3D + (size_t)idx * 6);
        cov3D = st->cov3D + (size_t)idx * 6;
    }
    f3 t;
    float txtz, tytz;
    mat3 T, Vrk, Wm, cov2;
    cov2d_common(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, &t, &txtz, &tytz, T,
                 Vrk, Wm, cov2);
    f3 cov = {cov2[0][0], cov2[0][1], cov2[1][1]};

    float det = (cov.x * cov.z - cov.y * cov.y);
    if (det == 0.0f) return 0;
    float det_inv = 1.f / det;
    f3 conic = {cov.z * det_inv, -cov.y * det_inv, cov.x * det_inv};

    float mid = 0.5f * (cov.x + cov.z);
    float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
    float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
    float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
    float pix = ndc2Pix(p_proj.x, W), piy = ndc2Pix(p_proj.y, H);
    uint32_t rmin[2], rmax[2];
    getRect(pix, piy, f2i_sat(my_radius), rmin, rmax, (uint32_t)st->tiles_x, (uint32_t)st->tiles_y);
    if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) return 0;

    if (!colors_given) {
        f3 c = computeColorFromSH(idx, D, M, orig_points, cam_pos, shs, st->clamped);
        st->rgb[idx * 3 + 0] = c.x; /* reference indexes rgb[idx*C+k] with C==3 enforced */
        st->rgb[idx * 3 + 1] = c.y;
        st->rgb[idx * 3 + 2] = c.z;
    }
    st->depths[idx] = p_view.z;
    st->radii[idx] = f2i_sat(my_radius);
    st->means2D[2 * idx] = pix;
    st->means2D[2 * idx + 1] = piy;
    st->conic_opacity[4 * idx + 0] = conic.x;
    st->conic_opacity[4 * idx + 1] = conic.y;
    st->conic_opacity[4 * idx + 2] = conic.z;
    st->conic_opacity[4 * idx + 3] = opacities[idx];
    st->tiles_touched[idx] = (rmax[1] - rmin[1]) * (rmax[0] - rmin[0]);
    return 0;
}

/* Stable LSD radix sort on bits [0, end_bit) == cub::DeviceRadixSort::SortPairs semantics
 * (CF/cuda_rasterizer/rasterizer_impl.cu:303-308; CUB 1.x from CUDA 11.6, not vendored: stable
 * ascending sort of (key,value) pairs on the given bit range). */
static void radix_sort_pairs(const uint64_t* kin, const uint32_t* vin, uint64_t* kout, uint32_t* vout,
                             size_t n, int end_bit)
{
    if (n == 0) return;
    uint64_t* k0 = (uint64_t*)malloc(n * sizeof(uint64_t));
    uint32_t* v0 = (uint32_t*)malloc(n * sizeof(uint32_t));
    uint64_t* k1 = (uint64_t*)malloc(n * sizeof(uint64_t));
    uint32_t* v1 = (uint32_t*)malloc(n * sizeof(uint32_t));
    memcpy(k0, kin, n * sizeof(uint64_t));
    memcpy(v0, vin, n * sizeof(uint32_t));
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    size_t* hist = (size_t*)malloc((size_t)nthreads * 256 * sizeof(size_t));
    for (int shift = 0; shift < end_bit; shift += 8) {
        int bits = end_bit - shift < 8 ? end_bit - shift : 8;
        uint64_t maskv = ((uint64_t)1 << bits) - 1;
        memset(hist, 0, (size_t)nthreads * 256 * sizeof(size_t));
#pragma omp parallel num_threads(nthreads)
        {
            int t = 0;
#ifdef _OPENMP
            t = omp_get_thread_num();
#endif
            size_t lo = n * (size_t)t / nthreads, hi = n * (size_t)(t + 1) / nthreads;
            size_t* h = hist + (size_t)t * 256;
            for (size_t i = lo; i < hi; i++) h[(k0[i] >> shift) & maskv]++;
#pragma omp barrier
#pragma omp single
            {
                size_t run = 0;
                for (int d = 0; d < 256; d++)
                    for (int tt = 0; tt < nthreads; tt++) {
                        size_t c = hist[(size_t)tt * 256 + d];
                        hist[(size_t)tt * 256 + d] = run;
                        run += c;
                    }
            }
            for (size_t i = lo; i < hi; i++) {
                size_t dst = h[(k0[i] >> shift) & maskv]++;
                k1[dst] = k0[i];
                v1[dst] = v0[i];
            }
        }
        uint64_t* tk = k0; k0 = k1; k1 = tk;
        uint32_t* tv = v0; v0 = v1; v1 = tv;
    }
    memcpy(kout, k0, n * sizeof(uint64_t));
    memcpy(vout, v0, n * sizeof(uint32_t));
    free(k0); free(v0); free(k1); free(v1); free(hist);
}

static saga_oracle_state* state_alloc(int P, int W, int H, int C)
{
    saga_oracle_state* st = (saga_oracle_state*)calloc(1, sizeof(*st));
    if (!st) return NULL;
    st->P = P; st->W = W; st->H = H; st->C = C;
    st->tiles_x = (W + BLOCK_X - 1) / BLOCK_X;
    st->tiles_y = (H + BLOCK_Y - 1) / BLOCK_Y;
    size_t p = (size_t)(P > 0 ? P : 1), n = (size_t)W * H;
    if (n == 0) n = 1;
    st->depths = (float*)calloc(p, sizeof(float));
    st->means2D = (float*)calloc(2 * p, sizeof(float));
    st->cov3D = (float*)calloc(6 * p, sizeof(float));
    st->conic_opacity = (float*)calloc(4 * p, sizeof(float));
    st->rgb = (float*)calloc(3 * p, sizeof(float));
    st->clamped = (uint8_t*)calloc(3 * p, 1);
    st->tiles_touched = (uint32_t*)calloc(p, sizeof(uint32_t));
    st->point_offsets = (uint32_t*)calloc(p, sizeof(uint32_t));
    st->radii = (int32_t*)calloc(p, sizeof(int32_t));
    st->ranges = (uint32_t*)calloc((size_t)2 * st->tiles_x * st->tiles_y + 2, sizeof(uint32_t));
    st->final_T = (float*)calloc(n, sizeof(float));
    st->n_contrib = (uint32_t*)calloc(n, sizeof(uint32_t));
    return st;
}

/* Stages shared by forward and mask_forward: preprocess -> scan -> keys -> sort -> ranges.
 * CF/cuda_rasterizer/rasterizer_impl.cu:246-317. */
static int geometry_and_binning(saga_oracle_state* st, int P, int D, int M, int C, int W, int H,
                                const float* means3D, const float* shs, int colors_given,
                                const float* opacities, const float* scales, float scale_modifier,
                                const float* rotations, const float* cov3D_precomp,
                                const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                                float tan_fovx, float tan_fovy, int prefiltered)
{
    /* rasterizer_impl.cu:222-223 */
    const float focal_y = H / (2.0f * tan_fovy);
    const float focal_x = W / (2.0f * tan_fovx);
    int rc = 0;
#pragma omp parallel for schedule(static) reduction(max : rc)
    for (int idx = 0; idx < P; idx++) {
        int r = preprocess_one(st, idx, D, M, C, means3D, scales, scale_modifier, rotations, opacities, shs,
                               cov3D_precomp, colors_given, viewmatrix, projmatrix, cam_pos, W, H, tan_fovx,
                               tan_fovy, focal_x, focal_y, prefiltered);
        if (r > rc) rc = r;
    }
    if (rc) return rc;

    /* InclusiveSum, rasterizer_impl.cu:277 */
    uint32_t run = 0;
    int64_t V = 0;
    for (int i = 0; i < P; i++) {
        run += st->tiles_touched[i];
        st->point_offsets[i] = run;
        if (st->radii[i] > 0) V++;
    }
    st->V = V;
    /* rasterizer_impl.cu:280-281: num_rendered is read back as int */
    st->R = P > 0 ? (int64_t)(int32_t)st->point_offsets[P - 1] : 0;
    size_t R = (size_t)st->R;
    size_t ra = R ? R : 1;
    st->keys_unsorted = (uint64_t*)calloc(ra, sizeof(uint64_t));
    st->values_unsorted = (uint32_t*)calloc(ra, sizeof(uint32_t));
    st->keys_sorted = (uint64_t*)calloc(ra, sizeof(uint64_t));
    st->point_list = (uint32_t*)calloc(ra, sizeof(uint32_t));

    /* duplicateWithKeys, rasterizer_impl.cu:70-111 */
    const uint32_t gx = (uint32_t)st->tiles_x, gy = (uint32_t)st->tiles_y;
#pragma omp parallel for schedule(dynamic, 4096)
    for (int idx = 0; idx < P; idx++) {
        if (st->radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : st->point_offsets[idx - 1];
            uint32_t rmin[2], rmax[2];
            getRect(st->means2D[2 * idx], st->means2D[2 * idx + 1], st->radii[idx], rmin, rmax, gx, gy);
            uint32_t dbits;
            memcpy(&dbits, &st->depths[idx], 4);
            for (int y = (int)rmin[1]; y < (int)rmax[1]; y++)
                for (int x = (int)rmin[0]; x < (int)rmax[0]; x++) {
                    uint64_t key = (uint64_t)((uint32_t)y * gx + (uint32_t)x);
                    key <<= 32;
                    key |= dbits;
                    st->keys_unsorted[off] = key;
                    st->values_unsorted[off] = (uint32_t)idx;
                    off++;
                }
        }
    }
    /* rasterizer_impl.cu:300-308 */
    int bit = (int)saga_oracle_get_higher_msb(gx * gy);
    st->sort_bits = 32 + bit;
    radix_sort_pairs(st->keys_unsorted, st->values_unsorted, st->keys_sorted, st->point_list, R, 32 + bit);

    /* cudaMemset + identifyTileRanges, rasterizer_impl.cu:116-138,310-317 */
    memset(st->ranges, 0, (size_t)2 * gx * gy * sizeof(uint32_t));
    for (size_t i = 0; i < R; i++) {
        uint32_t currtile = (uint32_t)(st->keys_sorted[i] >> 32);
        if (i == 0) st->ranges[2 * currtile] = 0;
        else {
            uint32_t prevtile = (uint32_t)(st->keys_sorted[i - 1] >> 32);
            if (currtile != prevtile) {
                st->ranges[2 * prevtile + 1] = (uint32_t)i;
                st->ranges[2 * currtile] = (uint32_t)i;
            }
        }
        if (i == R - 1) st->ranges[2 * currtile + 1] = (uint32_t)R;
    }
    return 0;
}

/* renderCUDA forward: CF/cuda_rasterizer/forward.cu:264-385; DEPTH variant adds M/D accumulators
 * (DEPTH/cuda_rasterizer/forward.cu:308-309,363-365,384-385); mask-only variant
 * DEPTH/cuda_rasterizer/forward.cu:390-498 (features == NULL, C == 0). */
static void render_forward(saga_oracle_state* st, int C, const float* features, const float* bg,
                           const float* mask, float* out_color, float* out_mask, float* out_depth)
{
    const int W = st->W, H = st->H;
    const int ntiles = st->tiles_x * st->tiles_y;
    int64_t E = 0, L = 0, pairs = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : E, L, pairs)
    for (int tile = 0; tile < ntiles; tile++) {
        const int tx = tile % st->tiles_x, ty = tile / st->tiles_x;
        const uint32_t r0 = st->ranges[2 * tile], r1 = st->ranges[2 * tile + 1];
        const int n = (int)(r1 - r0);
        float Cacc[256]; /* C <= 256 */
        int tile_consumed = 0, tile_maxcontrib = 0;
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                if (!(px < W && py < H)) continue;
                const int pix_id = W * py + px;
                const float pixfx = (float)px, pixfy = (float)py;
                float T = 1.0f;
                uint32_t contributor = 0, last_contributor = 0;
                float Macc = 0, Dacc = 0;
                for (int ch = 0; ch < C; ch++) Cacc[ch] = 0;
                int consumed = n;
                for (int j = 0; j < n; j++) {
                    contributor++;
                    const uint32_t id = st->point_list[r0 + j];
                    const float dx = st->means2D[2 * id] - pixfx;
                    const float dy = st->means2D[2 * id + 1] - pixfy;
                    const float* co = st->conic_opacity + 4 * (size_t)id;
                    const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0.0f) continue;
                    const float alpha = fminf(0.99f, co[3] * expf(power));
                    if (alpha < 1.0f / 255.0f) continue;
                    const float test_T = T * (1 - alpha);
                    if (test_T < 0.0001f) {
                        consumed = j + 1;
                        break; /* done = true */
                    }
                    for (int ch = 0; ch < C; ch++)
                        Cacc[ch] += features[(size_t)id * C + ch] * alpha * T;
                    if (mask) {
                        Macc += mask[id] * alpha * T;
                        if (out_depth) Dacc += st->depths[id] * alpha * T;
                    }
                    T = test_T;
                    last_contributor = contributor;
                    pairs++;
                }
                st->final_T[pix_id] = T;
                st->n_contrib[pix_id] = last_contributor;
                for (int ch = 0; ch < C; ch++)
                    out_color[(size_t)ch * H * W + pix_id] = Cacc[ch] + T * bg[ch];
                if (mask && out_mask) out_mask[pix_id] = Macc;
                if (mask && out_depth) out_depth[pix_id] = Dacc;
                if (consumed > tile_consumed) tile_consumed = consumed;
                if ((int)last_contributor > tile_maxcontrib) tile_maxcontrib = (int)last_contributor;
            }
        E += tile_consumed;
        L += tile_maxcontrib;
    }
    st->E = E; st->L = L; st->pairs = pairs;
}

saga_oracle_state* saga_oracle_forward(int P, int D, int M, int C, const float* background, int W, int H,
                                       const float* means3D, const float* shs, const float* colors_precomp,
                                       const float* opacities, const float* scales, float scale_modifier,
                                       const float* rotations, const float* cov3D_precomp,
                                       const float* viewmatrix, const float* projmatrix,
                                       const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
                                       const float* mask, float* out_color, float* out_mask,
                                       float* out_depth, int* radii, int* rc_out)
{
    saga_oracle_state* st = state_alloc(P, W, H, C);
    if (rc_out) *rc_out = 0;
    if (!st) return NULL;
    /* CF/rasterize_points.cu:68: out_color is zero-filled; P==0 short-circuit (:80) leaves zeros */
    memset(out_color, 0, (size_t)C * H * W * sizeof(float));
    if (out_mask) memset(out_mask, 0, (size_t)H * W * sizeof(float));
    if (out_depth) memset(out_depth, 0, (size_t)H * W * sizeof(float));
    if (P == 0) return st;
    /* rasterizer_impl.cu:242-245 */
    if (C != 3 && colors_precomp == NULL) {
        if (rc_out) *rc_out = 1;
        return st;
    }
    st->has_shs = (colors_precomp == NULL);
    int rc = geometry_and_binning(st, P, D, M, C, W, H, means3D, shs, colors_precomp != NULL, opacities,
                                  scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix,
                                  cam_pos, tan_fovx, tan_fovy, prefiltered);
    if (radii) memcpy(radii, st->radii, (size_t)P * sizeof(int32_t));
    if (rc) {
        if (rc_out) *rc_out = rc;
        return st;
    }
    const float* feature_ptr = colors_precomp != NULL ? colors_precomp : st->rgb; /* rasterizer_impl.cu:321 */
    render_forward(st, C, feature_ptr, background, mask, out_color, out_mask, out_depth);
    return st;
}

saga_oracle_state* saga_oracle_mask_forward(int P, int W, int H, const float* means3D,
                                            const float* opacities, const float* mask, const float* scales,
                                            float scale_modifier, const float* rotations,
                                            const float* cov3D_precomp, const float* viewmatrix,
                                            const float* projmatrix, float tan_fovx, float tan_fovy,
                                            int prefiltered, float* out_mask, int* radii, int* rc_out)
{
    saga_oracle_state* st = state_alloc(P, W, H, 0);
    if (rc_out) *rc_out = 0;
    if (!st) return NULL;
    memset(out_mask, 0, (size_t)H * W * sizeof(float));
    if (P == 0) return st;
    /* DEPTH/cuda_rasterizer/rasterizer_impl.cu:495-521: shs = nullptr, dummy non-null colours */
    int rc = geometry_and_binning(st, P, 0, 0, 3, W, H, means3D, NULL, 1, opacities, scales, scale_modifier,
                                  rotations, cov3D_precomp, viewmatrix, projmatrix, NULL, tan_fovx, tan_fovy,
                                  prefiltered);
    if (radii) memcpy(radii, st->radii, (size_t)P * sizeof(int32_t));
    if (rc) {
        if (rc_out) *rc_out = rc;
        return st;
    }
    render_forward(st, 0, NULL, NULL, mask, NULL, out_mask, NULL);
    return st;
}

/* ------------------------------------------------------------------------------------------ */
/* Backward                                                                                    */
/* ------------------------------------------------------------------------------------------ */

static inline void atomic_add_d(double* p, double v)
{
#pragma omp atomic
    *p += v;
}

/* renderCUDA backward: CF/cuda_rasterizer/backward.cu:399-559; DEPTH variant
 * DEPTH/cuda_rasterizer/backward.cu:457,516 (dL_dmask); mask-only DEPTH/.../backward.cu:568-660.
 * Per-Gaussian sums go to binary64 arrays acc_* (size P each field). */
/* exact_pairs: the per-pair VALUES (G, alpha, T, accum_rec, dL/dalpha, the six geometric terms) are evaluated in binary64 from
 * the same binary32 inputs, while every DECISION (power > 0, alpha < 1/255, the contributor range) is taken by the binary32
 * expressions above, exactly as the reference takes it.  The default mode rounds every per-pair value like the reference's
 * kernel does and is therefore a yardstick that shares the reference's per-pair rounding; this mode shares nobody's, which
 * is what a comparison of two implementations' noise needs (tests/test_zz_reference_pin.py: error statistics). */
static void render_backward(const saga_oracle_state* st, int C, const float* bg, const float* colors,
                            const float* dL_dpixels, const float* dL_dout_mask, int mask_only,
                            double* acc_color /*P*C*/, double* acc_mean2D /*P*2*/,
                            double* acc_conic /*P*3*/, double* acc_opacity /*P*/, double* acc_mask /*P*/,
                            int exact_pairs)
{
    const int W = st->W, H = st->H;
    const int ntiles = st->tiles_x * st->tiles_y;
    const int NF = C + 7; /* per-entry local fields: C colours, mean2D xy, conic xyw, opacity, mask */
    const float ddelx_dx = (float)(0.5 * W); /* backward.cu:460-461: 0.5 * W in double, stored float */
    const float ddely_dy = (float)(0.5 * H);
#pragma omp parallel
    {
        double* local = NULL;
        size_t local_cap = 0;
        float accum_rec[256], last_color[256], dL_dpixel[256];
        double accum_d[256], last_color_d[256];
#pragma omp for schedule(dynamic, 1)
        for (int tile = 0; tile < ntiles; tile++) {
            const int tx = tile % st->tiles_x, ty = tile / st->tiles_x;
            const uint32_t r0 = st->ranges[2 * tile], r1 = st->ranges[2 * tile + 1];
            const int n = (int)(r1 - r0);
            if (n == 0) continue;
            /* entries this tile's pixels can touch: max n_contrib */
            int maxc = 0;
            for (int ly = 0; ly < BLOCK_Y; ly++)
                for (int lx = 0; lx < BLOCK_X; lx++) {
                    const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                    if (px < W && py < H) {
                        int nc = (int)st->n_contrib[W * py + px];
                        if (nc > maxc) maxc = nc;
                    }
                }
            if (maxc == 0) continue;
            size_t need = (size_t)maxc * NF;
            if (need > local_cap) {
                free(local);
                local_cap = need * 2;
                local = (double*)malloc(local_cap * sizeof(double));
            }
            memset(local, 0, need * sizeof(double));
            for (int ly = 0; ly < BLOCK_Y; ly++)
                for (int lx = 0; lx < BLOCK_X; lx++) {
                    const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                    if (!(px < W && py < H)) continue;
                    const int pix_id = W * py + px;
                    const float pixfx = (float)px, pixfy = (float)py;
                    const float T_final = st->final_T[pix_id];
                    float T = T_final;
                    const int last_contributor = (int)st->n_contrib[pix_id];
                    for (int ch = 0; ch < C; ch++) {
                        accum_rec[ch] = 0;
                        last_color[ch] = 0;
                        dL_dpixel[ch] = dL_dpixels[(size_t)ch * H * W + pix_id];
                    }
                    const float dL_dout_mask_i = dL_dout_mask ? dL_dout_mask[pix_id] : 0;
                    float last_alpha = 0;
                    if (exact_pairs && !mask_only) {
                        double Td = (double)T_final, last_alpha_d = 0.0, bg_dot_d = 0.0;
                        for (int ch = 0; ch < C; ch++) {
                            accum_d[ch] = 0.0;
                            last_color_d[ch] = 0.0;
                            bg_dot_d += (double)bg[ch] * (double)dL_dpixel[ch];
                        }
                        for (int j = last_contributor - 1; j >= 0; j--) {
                            const uint32_t id = st->point_list[r0 + j];
                            const float* co = st->conic_opacity + 4 * (size_t)id;
                            {   /* decisions: the binary32 expressions of backward.cu:478-492 */
                                const float dx = st->means2D[2 * id] - pixfx;
                                const float dy = st->means2D[2 * id + 1] - pixfy;
                                const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                                if (power > 0.0f) continue;
                                if (fminf(0.99f, co[3] * expf(power)) < 1.0f / 255.0f) continue;
                            }
                            const double dx = (double)st->means2D[2 * id] - (double)pixfx;
                            const double dy = (double)st->means2D[2 * id + 1] - (double)pixfy;
                            const double power = -0.5 * ((double)co[0] * dx * dx + (double)co[2] * dy * dy) - (double)co[1] * dx * dy;
                            const double G = exp(power);
                            const double alpha = fmin((double)0.99f, (double)co[3] * G);
                            Td = Td / (1.0 - alpha);
                            const double dchannel_dcolor = alpha * Td;
                            double* loc = local + (size_t)j * NF;
                            double dL_dalpha = 0.0;
                            for (int ch = 0; ch < C; ch++) {
                                const double c = (double)colors[(size_t)id * C + ch];
                                accum_d[ch] = last_alpha_d * last_color_d[ch] + (1.0 - last_alpha_d) * accum_d[ch];
                                last_color_d[ch] = c;
                                dL_dalpha += (c - accum_d[ch]) * (double)dL_dpixel[ch];
                                loc[ch] += dchannel_dcolor * (double)dL_dpixel[ch];
                            }
                            if (dL_dout_mask) loc[C + 6] += dchannel_dcolor * (double)dL_dout_mask_i;
                            dL_dalpha *= Td;
                            last_alpha_d = alpha;
                            dL_dalpha += (-(double)T_final / (1.0 - alpha)) * bg_dot_d;
                            const double dL_dG = (double)co[3] * dL_dalpha;
                            const double gdx = G * dx, gdy = G * dy;
                            const double dG_ddelx = -gdx * (double)co[0] - gdy * (double)co[1];
                            const double dG_ddely = -gdy * (double)co[2] - gdx * (double)co[1];
                            loc[C + 0] += dL_dG * dG_ddelx * (double)ddelx_dx;
                            loc[C + 1] += dL_dG * dG_ddely * (double)ddely_dy;
                            loc[C + 2] += -0.5 * gdx * dx * dL_dG;
                            loc[C + 3] += -0.5 * gdx * dy * dL_dG;
                            loc[C + 4] += -0.5 * gdy * dy * dL_dG;
                            loc[C + 5] += G * dL_dalpha;
                        }
                        continue;
                    }
                    /* back to front; entries with contributor >= last_contributor are skipped
                     * (backward.cu:485-487), so start at list position last_contributor-1 */
                    for (int j = last_contributor - 1; j >= 0; j--) {
                        const uint32_t id = st->point_list[r0 + j];
                        const float dx = st->means2D[2 * id] - pixfx;
                        const float dy = st->means2D[2 * id + 1] - pixfy;
                        const float* co = st->conic_opacity + 4 * (size_t)id;
                        const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > 0.0f) continue;
                        const float G = expf(power);
                        const float alpha = fminf(0.99f, co[3] * G);
                        if (alpha < 1.0f / 255.0f) continue;
                        T = T / (1.f - alpha);
                        const float dchannel_dcolor = alpha * T;
                        double* loc = local + (size_t)j * NF;
                        if (mask_only) {
                            loc[C + 6] += (double)(dchannel_dcolor * dL_dout_mask_i);
                            continue;
                        }
                        float dL_dalpha = 0.0f;
                        for (int ch = 0; ch < C; ch++) {
                            const float c = colors[(size_t)id * C + ch];
                            accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                            last_color[ch] = c;
                            const float dL_dchannel = dL_dpixel[ch];
                            dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
                            loc[ch] += (double)(dchannel_dcolor * dL_dchannel);
                        }
                        if (dL_dout_mask) loc[C + 6] += (double)(dchannel_dcolor * dL_dout_mask_i);
                        dL_dalpha *= T;
                        last_alpha = alpha;
                        float bg_dot_dpixel = 0;
                        for (int i = 0; i < C; i++) bg_dot_dpixel += bg[i] * dL_dpixel[i];
                        dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
                        const float dL_dG = co[3] * dL_dalpha;
                        const float gdx = G * dx;
                        const float gdy = G * dy;
                        const float dG_ddelx = -gdx * co[0] - gdy * co[1];
                        const float dG_ddely = -gdy * co[2] - gdx * co[1];
                        loc[C + 0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
                        loc[C + 1] += (double)(dL_dG * dG_ddely * ddely_dy);
                        loc[C + 2] += (double)(-0.5f * gdx * dx * dL_dG);
                        loc[C + 3] += (double)(-0.5f * gdx * dy * dL_dG);
                        loc[C + 4] += (double)(-0.5f * gdy * dy * dL_dG);
                        loc[C + 5] += (double)(G * dL_dalpha);
                    }
                }
            /* flush tile-local sums */
            for (int j = 0; j < maxc; j++) {
                const uint32_t id = st->point_list[r0 + j];
                const double* loc = local + (size_t)j * NF;
                if (!mask_only) {
                    for (int ch = 0; ch < C; ch++)
                        if (loc[ch] != 0.0) atomic_add_d(&acc_color[(size_t)id * C + ch], loc[ch]);
                    if (loc[C + 0] != 0.0) atomic_add_d(&acc_mean2D[2 * (size_t)id + 0], loc[C + 0]);
                    if (loc[C + 1] != 0.0) atomic_add_d(&acc_mean2D[2 * (size_t)id + 1], loc[C + 1]);
                    if (loc[C + 2] != 0.0) atomic_add_d(&acc_conic[3 * (size_t)id + 0], loc[C + 2]);
                    if (loc[C + 3] != 0.0) atomic_add_d(&acc_conic[3 * (size_t)id + 1], loc[C + 3]);
                    if (loc[C + 4] != 0.0) atomic_add_d(&acc_conic[3 * (size_t)id + 2], loc[C + 4]);
                    if (loc[C + 5] != 0.0) atomic_add_d(&acc_opacity[id], loc[C + 5]);
                }
                if (acc_mask && loc[C + 6] != 0.0) atomic_add_d(&acc_mask[id], loc[C + 6]);
            }
        }
        free(local);
    }
}

/* computeCov2DCUDA, CF/cuda_rasterizer/backward.cu:144-274 (one Gaussian) */
static void cov2d_backward_one(int idx, const float* means, const int32_t* radii, const float* cov3Ds,
                               float h_x, float h_y, float tan_fovx, float tan_fovy,
                               const float* view_matrix, const float* dL_dconics, float* dL_dmeans,
                               float* dL_dcov)
{
    if (!(radii[idx] > 0)) return;
    const float* cov3D = cov3Ds + 6 * (size_t)idx;
    f3 mean = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
    f3 dL_dconic = {dL_dconics[4 * idx], dL_dconics[4 * idx + 1], dL_dconics[4 * idx + 3]};
    f3 t;
    float txtz, tytz;
    mat3 T, Vrk, Wm, cov2D;
    cov2d_common(mean, h_x, h_y, tan_fovx, tan_fovy, cov3D, view_matrix, &t, &txtz, &tytz, T, Vrk, Wm, cov2D);
    const float limx = 1.3f * tan_fovx;
    const float limy = 1.3f * tan_fovy;
    const float x_grad_mul = txtz < -limx || txtz > limx ? 0 : 1;
    const float y_grad_mul = tytz < -limy || tytz > limy ? 0 : 1;

    float a = cov2D[0][0]; /* already += 0.3f */
    float b = cov2D[0][1];
    float c = cov2D[1][1];
    float denom = a * c - b * b;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float* o = dL_dcov + 6 * (size_t)idx;
    if (denom2inv != 0) {
        dL_da = denom2inv * (-c * c * dL_dconic.x + 2 * b * c * dL_dconic.y + (denom - a * c) * dL_dconic.z);
        dL_dc = denom2inv * (-a * a * dL_dconic.z + 2 * a * b * dL_dconic.y + (denom - a * c) * dL_dconic.x);
        dL_db = denom2inv * 2 * (b * c * dL_dconic.x - (denom + 2 * b * b) * dL_dconic.y + a * b * dL_dconic.z);
        o[0] = (T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc);
        o[3] = (T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc);
        o[5] = (T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc);
        o[1] = 2 * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][1] * dL_dc;
        o[2] = 2 * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][2] * dL_dc;
        o[4] = 2 * T[0][2] * T[0][1] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db + 2 * T[1][1] * T[1][2] * dL_dc;
    } else {
        for (int i = 0; i < 6; i++) o[i] = 0;
    }
    float dL_dT00 = 2 * (T[0][0] * Vrk[0][0] + T[0][1] * Vrk[0][1] + T[0][2] * Vrk[0][2]) * dL_da +
                    (T[1][0] * Vrk[0][0] + T[1][1] * Vrk[0][1] + T[1][2] * Vrk[0][2]) * dL_db;
    float dL_dT01 = 2 * (T[0][0] * Vrk[1][0] + T[0][1] * Vrk[1][1] + T[0][2] * Vrk[1][2]) * dL_da +
                    (T[1][0] * Vrk[1][0] + T[1][1] * Vrk[1][1] + T[1][2] * Vrk[1][2]) * dL_db;
    float dL_dT02 = 2 * (T[0][0] * Vrk[2][0] + T[0][1] * Vrk[2][1] + T[0][2] * Vrk[2][2]) * dL_da +
                    (T[1][0] * Vrk[2][0] + T[1][1] * Vrk[2][1] + T[1][2] * Vrk[2][2]) * dL_db;
    float dL_dT10 = 2 * (T[1][0] * Vrk[0][0] + T[1][1] * Vrk[0][1] + T[1][2] * Vrk[0][2]) * dL_dc +
                    (T[0][0] * Vrk[0][0] + T[0][1] * Vrk[0][1] + T[0][2] * Vrk[0][2]) * dL_db;
    float dL_dT11 = 2 * (T[1][0] * Vrk[1][0] + T[1][1] * Vrk[1][1] + T[1][2] * Vrk[1][2]) * dL_dc +
                    (T[0][0] * Vrk[1][0] + T[0][1] * Vrk[1][1] + T[0][2] * Vrk[1][2]) * dL_db;
    float dL_dT12 = 2 * (T[1][0] * Vrk[2][0] + T[1][1] * Vrk[2][1] + T[1][2] * Vrk[2][2]) * dL_dc +
                    (T[0][0] * Vrk[2][0] + T[0][1] * Vrk[2][1] + T[0][2] * Vrk[2][2]) * dL_db;
    float dL_dJ00 = Wm[0][0] * dL_dT00 + Wm[0][1] * dL_dT01 + Wm[0][2] * dL_dT02;
    float dL_dJ02 = Wm[2][0] * dL_dT00 + Wm[2][1] * dL_dT01 + Wm[2][2] * dL_dT02;
    float dL_dJ11 = Wm[1][0] * dL_dT10 + Wm[1][1] * dL_dT11 + Wm[1][2] * dL_dT12;
    float dL_dJ12 = Wm[2][0] * dL_dT10 + Wm[2][1] * dL_dT11 + Wm[2][2] * dL_dT12;
    float tz = 1.f / t.z;
    float tz2 = tz * tz;
    float tz3 = tz2 * tz;
    float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
    float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
    float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 +
                   (2 * h_y * t.y) * tz3 * dL_dJ12;
    f3 dt = {dL_dtx, dL_dty, dL_dtz};
    f3 dL_dmean = transformVec4x3Transpose(dt, view_matrix);
    dL_dmeans[3 * idx + 0] = dL_dmean.x; /* assignment, backward.cu:273 */
    dL_dmeans[3 * idx + 1] = dL_dmean.y;
    dL_dmeans[3 * idx + 2] = dL_dmean.z;
}

/* computeColorFromSH backward, CF/cuda_rasterizer/backward.cu:20-139 */
static void sh_backward_one(int idx, int deg, int max_coeffs, const float* means, const float* campos,
                            const float* shs, const uint8_t* clamped, const float* dL_dcolor,
                            float* dL_dmeans, float* dL_dshs)
{
    f3 pos = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
    f3 dir_orig = {pos.x - campos[0], pos.y - campos[1], pos.z - campos[2]};
    float len = sqrtf(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
    f3 dir = {dir_orig.x / len, dir_orig.y / len, dir_orig.z / len};
    const float* sh = shs + (size_t)idx * max_coeffs * 3;
    float dL_dRGB[3] = {dL_dcolor[3 * idx + 0], dL_dcolor[3 * idx + 1], dL_dcolor[3 * idx + 2]};
    dL_dRGB[0] *= clamped[3 * idx + 0] ? 0 : 1;
    dL_dRGB[1] *= clamped[3 * idx + 1] ? 0 : 1;
    dL_dRGB[2] *= clamped[3 * idx + 2] ? 0 : 1;
    float dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
    float x = dir.x, y = dir.y, z = dir.z;
    float* dL_dsh = dL_dshs + (size_t)idx * max_coeffs * 3;
#define SH(k) sh[(k) * 3 + c]
#define DSH(k, v) for (int c = 0; c < 3; c++) dL_dsh[(k) * 3 + c] = (v) * dL_dRGB[c]
    float dRGBdsh0 = SH_C0;
    DSH(0, dRGBdsh0);
    if (deg > 0) {
        float dRGBdsh1 = -SH_C1 * y;
        float dRGBdsh2 = SH_C1 * z;
        float dRGBdsh3 = -SH_C1 * x;
        DSH(1, dRGBdsh1);
        DSH(2, dRGBdsh2);
        DSH(3, dRGBdsh3);
        for (int c = 0; c < 3; c++) {
            dRGBdx[c] = -SH_C1 * SH(3);
            dRGBdy[c] = -SH_C1 * SH(1);
            dRGBdz[c] = SH_C1 * SH(2);
        }
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            float dRGBdsh4 = SH_C2[0] * xy;
            float dRGBdsh5 = SH_C2[1] * yz;
            float dRGBdsh6 = SH_C2[2] * (2.f * zz - xx - yy);
            float dRGBdsh7 = SH_C2[3] * xz;
            float dRGBdsh8 = SH_C2[4] * (xx - yy);
            DSH(4, dRGBdsh4);
            DSH(5, dRGBdsh5);
            DSH(6, dRGBdsh6);
            DSH(7, dRGBdsh7);
            DSH(8, dRGBdsh8);
            for (int c = 0; c < 3; c++) {
                dRGBdx[c] += SH_C2[0] * y * SH(4) + SH_C2[2] * 2.f * -x * SH(6) + SH_C2[3] * z * SH(7) + SH_C2[4] * 2.f * x * SH(8);
                dRGBdy[c] += SH_C2[0] * x * SH(4) + SH_C2[1] * z * SH(5) + SH_C2[2] * 2.f * -y * SH(6) + SH_C2[4] * 2.f * -y * SH(8);
                dRGBdz[c] += SH_C2[1] * y * SH(5) + SH_C2[2] * 2.f * 2.f * z * SH(6) + SH_C2[3] * x * SH(7);
            }
            if (deg > 2) {
                float dRGBdsh9 = SH_C3[0] * y * (3.f * xx - yy);
                float dRGBdsh10 = SH_C3[1] * xy * z;
                float dRGBdsh11 = SH_C3[2] * y * (4.f * zz - xx - yy);
                float dRGBdsh12 = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                float dRGBdsh13 = SH_C3[4] * x * (4.f * zz - xx - yy);
                float dRGBdsh14 = SH_C3[5] * z * (xx - yy);
                float dRGBdsh15 = SH_C3[6] * x * (xx - 3.f * yy);
                DSH(9, dRGBdsh9);
                DSH(10, dRGBdsh10);
                DSH(11, dRGBdsh11);
                DSH(12, dRGBdsh12);
                DSH(13, dRGBdsh13);
                DSH(14, dRGBdsh14);
                DSH(15, dRGBdsh15);
                for (int c = 0; c < 3; c++) {
                    dRGBdx[c] += (SH_C3[0] * SH(9) * 3.f * 2.f * xy + SH_C3[1] * SH(10) * yz +
                                  SH_C3[2] * SH(11) * -2.f * xy + SH_C3[3] * SH(12) * -3.f * 2.f * xz +
                                  SH_C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) +
                                  SH_C3[5] * SH(14) * 2.f * xz + SH_C3[6] * SH(15) * 3.f * (xx - yy));
                    dRGBdy[c] += (SH_C3[0] * SH(9) * 3.f * (xx - yy) + SH_C3[1] * SH(10) * xz +
                                  SH_C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) +
                                  SH_C3[3] * SH(12) * -3.f * 2.f * yz + SH_C3[4] * SH(13) * -2.f * xy +
                                  SH_C3[5] * SH(14) * -2.f * yz + SH_C3[6] * SH(15) * -3.f * 2.f * xy);
                    dRGBdz[c] += (SH_C3[1] * SH(10) * xy + SH_C3[2] * SH(11) * 4.f * 2.f * yz +
                                  SH_C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) +
                                  SH_C3[4] * SH(13) * 4.f * 2.f * xz + SH_C3[5] * SH(14) * (xx - yy));
                }
            }
        }
    }
#undef SH
#undef DSH
    f3 dL_ddir = {dRGBdx[0] * dL_dRGB[0] + dRGBdx[1] * dL_dRGB[1] + dRGBdx[2] * dL_dRGB[2],
                  dRGBdy[0] * dL_dRGB[0] + dRGBdy[1] * dL_dRGB[1] + dRGBdy[2] * dL_dRGB[2],
                  dRGBdz[0] * dL_dRGB[0] + dRGBdz[1] * dL_dRGB[1] + dRGBdz[2] * dL_dRGB[2]};
    f3 dL_dmean = dnormvdv3(dir_orig, dL_ddir);
    dL_dmeans[3 * idx + 0] += dL_dmean.x;
    dL_dmeans[3 * idx + 1] += dL_dmean.y;
    dL_dmeans[3 * idx + 2] += dL_dmean.z;
}

/* computeCov3D backward, CF/cuda_rasterizer/backward.cu:278-341 */
static void cov3d_backward_one(int idx, const float* scale, float mod, const float* rot,
                               const float* dL_dcov3Ds, float* dL_dscales, float* dL_drots)
{
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    mat3 Rm = {{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
               {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
               {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
    mat3 S = {{1.0f, 0, 0}, {0, 1.0f, 0}, {0, 0, 1.0f}};
    float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    S[0][0] = s[0]; S[1][1] = s[1]; S[2][2] = s[2];
    mat3 M;
    m3mul(S, Rm, M);
    const float* d = dL_dcov3Ds + 6 * (size_t)idx;
    mat3 dL_dSigma = {{d[0], 0.5f * d[1], 0.5f * d[2]},
                      {0.5f * d[1], d[3], 0.5f * d[4]},
                      {0.5f * d[2], 0.5f * d[4], d[5]}};
    mat3 M2, dL_dM, Rt, dL_dMt;
    for (int c = 0; c < 3; c++)
        for (int rr = 0; rr < 3; rr++) M2[c][rr] = 2.0f * M[c][rr]; /* glm scalar * mat */
    m3mul(M2, dL_dSigma, dL_dM);
    m3transpose(Rm, Rt);
    m3transpose(dL_dM, dL_dMt);
    float* ds = dL_dscales + 3 * (size_t)idx;
    ds[0] = Rt[0][0] * dL_dMt[0][0] + Rt[0][1] * dL_dMt[0][1] + Rt[0][2] * dL_dMt[0][2];
    ds[1] = Rt[1][0] * dL_dMt[1][0] + Rt[1][1] * dL_dMt[1][1] + Rt[1][2] * dL_dMt[1][2];
    ds[2] = Rt[2][0] * dL_dMt[2][0] + Rt[2][1] * dL_dMt[2][1] + Rt[2][2] * dL_dMt[2][2];
    for (int k = 0; k < 3; k++) {
        dL_dMt[0][k] *= s[0];
        dL_dMt[1][k] *= s[1];
        dL_dMt[2][k] *= s[2];
    }
    float* dq = dL_drots + 4 * (size_t)idx;
    dq[0] = 2 * z * (dL_dMt[0][1] - dL_dMt[1][0]) + 2 * y * (dL_dMt[2][0] - dL_dMt[0][2]) + 2 * x * (dL_dMt[1][2] - dL_dMt[2][1]);
    dq[1] = 2 * y * (dL_dMt[1][0] + dL_dMt[0][1]) + 2 * z * (dL_dMt[2][0] + dL_dMt[0][2]) + 2 * r * (dL_dMt[1][2] - dL_dMt[2][1]) - 4 * x * (dL_dMt[2][2] + dL_dMt[1][1]);
    dq[2] = 2 * x * (dL_dMt[1][0] + dL_dMt[0][1]) + 2 * r * (dL_dMt[2][0] - dL_dMt[0][2]) + 2 * z * (dL_dMt[1][2] + dL_dMt[2][1]) - 4 * y * (dL_dMt[2][2] + dL_dMt[0][0]);
    dq[3] = 2 * r * (dL_dMt[0][1] - dL_dMt[1][0]) + 2 * x * (dL_dMt[2][0] + dL_dMt[0][2]) + 2 * y * (dL_dMt[1][2] + dL_dMt[2][1]) - 4 * z * (dL_dMt[1][1] + dL_dMt[0][0]);
}

/* preprocessCUDA backward, CF/cuda_rasterizer/backward.cu:346-396 (one Gaussian) */
static void preprocess_backward_one(int idx, int D, int M, const float* means, const int32_t* radii,
                                    const float* shs, const uint8_t* clamped, const float* scales,
                                    const float* rotations, float scale_modifier, const float* proj,
                                    const float* campos, const float* dL_dmean2D, float* dL_dmeans,
                                    float* dL_dcolor, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                                    float* dL_drot)
{
    if (!(radii[idx] > 0)) return;
    f3 m = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
    f4 m_hom = transformPoint4x4(m, proj);
    float m_w = 1.0f / (m_hom.w + 0.0000001f);
    float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
    float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
    const float gx = dL_dmean2D[3 * idx], gy = dL_dmean2D[3 * idx + 1];
    f3 dL_dmean;
    dL_dmean.x = (proj[0] * m_w - proj[3] * mul1) * gx + (proj[1] * m_w - proj[3] * mul2) * gy;
    dL_dmean.y = (proj[4] * m_w - proj[7] * mul1) * gx + (proj[5] * m_w - proj[7] * mul2) * gy;
    dL_dmean.z = (proj[8] * m_w - proj[11] * mul1) * gx + (proj[9] * m_w - proj[11] * mul2) * gy;
    dL_dmeans[3 * idx + 0] += dL_dmean.x;
    dL_dmeans[3 * idx + 1] += dL_dmean.y;
    dL_dmeans[3 * idx + 2] += dL_dmean.z;
    if (shs) sh_backward_one(idx, D, M, means, campos, shs, clamped, dL_dcolor, dL_dmeans, dL_dsh);
    if (scales)
        cov3d_backward_one(idx, scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx,
                           dL_dcov3D, dL_dscale, dL_drot);
}

void saga_oracle_backward(const saga_oracle_state* st, int P, int D, int M, int C, const float* background,
                          int W, int H, const float* means3D, const float* shs, const float* colors_precomp,
                          const float* scales, float scale_modifier, const float* rotations,
                          const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                          const float* cam_pos, float tan_fovx, float tan_fovy, const float* dL_dpix,
                          const float* dL_dout_mask, const float* mask, float* dL_dmean2D, float* dL_dconic,
                          float* dL_dopacity, float* dL_dcolor, float* dL_dmask, float* dL_dmean3D,
                          float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                          int accum_double)
{
    (void)mask;   /* accum_double: 1 = binary64 sums of binary32 per-pair terms (default); 2 = binary64 per-pair values too */
    if (P == 0) return; /* CF/rasterize_points.cu:161 */
    const float focal_y = H / (2.0f * tan_fovy);
    const float focal_x = W / (2.0f * tan_fovx);
    const float* color_ptr = (colors_precomp != NULL) ? colors_precomp : st->rgb;
    size_t p = (size_t)P;
    double* acc_color = (double*)calloc(p * (size_t)C, sizeof(double));
    double* acc_mean2D = (double*)calloc(p * 2, sizeof(double));
    double* acc_conic = (double*)calloc(p * 3, sizeof(double));
    double* acc_opacity = (double*)calloc(p, sizeof(double));
    double* acc_mask = dL_dmask ? (double*)calloc(p, sizeof(double)) : NULL;
    render_backward(st, C, background, color_ptr, dL_dpix, dL_dmask ? dL_dout_mask : NULL, 0, acc_color,
                    acc_mean2D, acc_conic, acc_opacity, acc_mask, accum_double == 2);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        for (int ch = 0; ch < C; ch++) dL_dcolor[(size_t)i * C + ch] += (float)acc_color[(size_t)i * C + ch];
        dL_dmean2D[3 * i + 0] += (float)acc_mean2D[2 * i + 0];
        dL_dmean2D[3 * i + 1] += (float)acc_mean2D[2 * i + 1];
        dL_dconic[4 * i + 0] += (float)acc_conic[3 * i + 0];
        dL_dconic[4 * i + 1] += (float)acc_conic[3 * i + 1];
        dL_dconic[4 * i + 3] += (float)acc_conic[3 * i + 2];
        dL_dopacity[i] += (float)acc_opacity[i];
        if (dL_dmask) dL_dmask[i] += (float)acc_mask[i];
    }
    free(acc_color); free(acc_mean2D); free(acc_conic); free(acc_opacity); free(acc_mask);

    /* BACKWARD::preprocess, CF/cuda_rasterizer/backward.cu:561-624 */
    const float* cov3D_ptr = (cov3D_precomp != NULL) ? cov3D_precomp : st->cov3D;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++)
        cov2d_backward_one(i, means3D, st->radii, cov3D_ptr, focal_x, focal_y, tan_fovx, tan_fovy, viewmatrix,
                           dL_dconic, dL_dmean3D, dL_dcov3D);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++)
        preprocess_backward_one(i, D, M, means3D, st->radii, shs, st->clamped, scales, rotations,
                                scale_modifier, projmatrix, cam_pos, dL_dmean2D, dL_dmean3D, dL_dcolor,
                                dL_dcov3D, dL_dsh, dL_dscale, dL_drot);
}

void saga_oracle_mask_backward(const saga_oracle_state* st, int P, int W, int H, const float* dL_dout_mask,
                               float* dL_dmask, int accum_double)
{
    (void)W; (void)H; (void)accum_double;
    if (P == 0) return;
    double* acc_mask = (double*)calloc((size_t)P, sizeof(double));
    render_backward(st, 0, NULL, NULL, NULL, dL_dout_mask, 1, NULL, NULL, NULL, NULL, acc_mask, 0);
    for (int i = 0; i < P; i++) dL_dmask[i] += (float)acc_mask[i];
    free(acc_mask);
}

void saga_oracle_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                              uint8_t* present)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        f3 pv;
        present[i] = (uint8_t)in_frustum(i, means3D, viewmatrix, projmatrix, &pv);
    }
}

const void* saga_oracle_field(const saga_oracle_state* st, int field, size_t* count)
{
    size_t P = (size_t)st->P, N = (size_t)st->W * st->H, R = (size_t)st->R;
    size_t T = (size_t)st->tiles_x * st->tiles_y;
    const void* p = NULL;
    size_t n = 0;
    switch (field) {
    case SAGA_F_DEPTHS: p = st->depths; n = P; break;
    case SAGA_F_MEANS2D: p = st->means2D; n = 2 * P; break;
    case SAGA_F_COV3D: p = st->cov3D; n = 6 * P; break;
    case SAGA_F_CONIC_OPACITY: p = st->conic_opacity; n = 4 * P; break;
    case SAGA_F_RGB: p = st->rgb; n = 3 * P; break;
    case SAGA_F_CLAMPED: p = st->clamped; n = 3 * P; break;
    case SAGA_F_TILES_TOUCHED: p = st->tiles_touched; n = P; break;
    case SAGA_F_POINT_OFFSETS: p = st->point_offsets; n = P; break;
    case SAGA_F_KEYS_SORTED: p = st->keys_sorted; n = R; break;
    case SAGA_F_POINT_LIST: p = st->point_list; n = R; break;
    case SAGA_F_RANGES: p = st->ranges; n = 2 * T; break;
    case SAGA_F_FINAL_T: p = st->final_T; n = N; break;
    case SAGA_F_N_CONTRIB: p = st->n_contrib; n = N; break;
    case SAGA_F_RADII: p = st->radii; n = P; break;
    case SAGA_F_KEYS_UNSORTED: p = st->keys_unsorted; n = R; break;
    case SAGA_F_VALUES_UNSORTED: p = st->values_unsorted; n = R; break;
    default: break;
    }
    if (p == NULL) n = 0;
    if (count) *count = n;
    return p;
}

int64_t saga_oracle_counter(const saga_oracle_state* st, int counter)
{
    switch (counter) {
    case SAGA_C_P: return st->P;
    case SAGA_C_V: return st->V;
    case SAGA_C_R: return st->R;
    case SAGA_C_E: return st->E;
    case SAGA_C_L: return st->L;
    case SAGA_C_PAIRS: return st->pairs;
    case SAGA_C_TILES: return (int64_t)st->tiles_x * st->tiles_y;
    case SAGA_C_SORT_BITS: return st->sort_bits;
    default: return -1;
    }
}

void saga_oracle_free(saga_oracle_state* st)
{
    if (!st) return;
    free(st->depths); free(st->means2D); free(st->cov3D); free(st->conic_opacity); free(st->rgb);
    free(st->clamped); free(st->tiles_touched); free(st->point_offsets); free(st->radii);
    free(st->keys_unsorted); free(st->values_unsorted); free(st->keys_sorted); free(st->point_list);
    free(st->ranges); free(st->final_T); free(st->n_contrib);
    free(st);
}

int saga_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void saga_oracle_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
