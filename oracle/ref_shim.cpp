// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or shipped with the product (seganygaussians_amd/).
//
// C-ABI shim around the REFERENCE's own rasterizer core, `CudaRasterizer::Rasterizer` (declared in
// CF|BASE|DEPTH /cuda_rasterizer/rasterizer.h:24-84), so that tests can run the reference implementation itself on the
// MI355X and compare the oracle and the HIP product path against it.  This file is OUR code; the reference sources are
// NOT in this repository: oracle/build_ref.py hipifies them from /root/reference into a temporary directory at build
// time (test-only translation, exactly what torch's CUDAExtension does under ROCm), compiles them together with this
// shim and leaves only the shared object in oracle/_ref/ (git-ignored).
//
// The shim replaces the reference's torch glue (rasterize_points.cu:35-216): plain device pointers in, std::function
// buffer callbacks built from C callbacks, same argument order as the core API.  Built once per variant:
//   -DREF_DEPTH            DEPTH/ (RGB + mask + depth, mask-only pair)   else CF/ == BASE/ core (only NUM_CHANNELS differs)
//   NUM_CHANNELS           comes from the variant's (hipified) config header; build_ref.py rewrites it for the 64-channel build.
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <exception>
#include <functional>
#include <string>

#include "rasterizer.h"
#include "rasterizer_impl.h"
#ifdef REF_DEPTH
#include "config.h"
#elif defined(REF_BASE)
#include "config.h"
#else
#include "config_contrastive_f.h"
#endif

#define REF_API extern "C" __attribute__((visibility("default")))

typedef char* (*ref_resize_fn)(size_t nbytes, void* ctx);

static thread_local std::string g_err;

static std::function<char*(size_t)> wrap(ref_resize_fn fn, void* ctx) {
    return [fn, ctx](size_t n) -> char* { return fn(n, ctx); };
}

REF_API int saga_ref_channels(void) { return NUM_CHANNELS; }
REF_API int saga_ref_is_depth(void) {
#ifdef REF_DEPTH
    return 1;
#else
    return 0;
#endif
}
REF_API const char* saga_ref_last_error(void) { return g_err.c_str(); }

// Offsets of the reference's private arrays inside its three byte buffers (GeometryState / ImageState /
// BinningState::fromChunk, rasterizer_impl.cu:155-194), for a 128-byte aligned base address.
// geom[10]: depths, clamped, internal_radii, means2D, cov3D, conic_opacity, rgb, tiles_touched, scanning_space, point_offsets
// img[3]:   accum_alpha, n_contrib, ranges
// bin[5]:   point_list, point_list_unsorted, point_list_keys, point_list_keys_unsorted, list_sorting_space
REF_API void saga_ref_layout(size_t P, size_t N, size_t R, size_t* geom, size_t* img, size_t* bin) {
    using namespace CudaRasterizer;
    char* c = nullptr;
    GeometryState g = GeometryState::fromChunk(c, P);
    geom[0] = (size_t)g.depths; geom[1] = (size_t)g.clamped; geom[2] = (size_t)g.internal_radii;
    geom[3] = (size_t)g.means2D; geom[4] = (size_t)g.cov3D; geom[5] = (size_t)g.conic_opacity;
    geom[6] = (size_t)g.rgb; geom[7] = (size_t)g.tiles_touched; geom[8] = (size_t)g.scanning_space;
    geom[9] = (size_t)g.point_offsets;
    c = nullptr;
    ImageState im = ImageState::fromChunk(c, N);
    img[0] = (size_t)im.accum_alpha; img[1] = (size_t)im.n_contrib; img[2] = (size_t)im.ranges;
    c = nullptr;
    BinningState b = BinningState::fromChunk(c, R);
    bin[0] = (size_t)b.point_list; bin[1] = (size_t)b.point_list_unsorted; bin[2] = (size_t)b.point_list_keys;
    bin[3] = (size_t)b.point_list_keys_unsorted; bin[4] = (size_t)b.list_sorting_space;
}

REF_API int saga_ref_mark_visible(int P, float* means3D, float* view, float* proj, bool* present) {
    try {
        CudaRasterizer::Rasterizer::markVisible(P, means3D, view, proj, present);
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
    return 0;
}

// Returns num_rendered (>= 0) or -1 on an exception (text in saga_ref_last_error()).
// mask / out_mask / out_depth are used by the DEPTH build only.
REF_API int saga_ref_forward(ref_resize_fn geom, void* gctx, ref_resize_fn bin, void* bctx, ref_resize_fn img, void* ictx,
                             int P, int D, int M, const float* bg, int W, int H, const float* means3D, const float* shs,
                             const float* colors_precomp, const float* opacities, const float* mask, const float* scales,
                             float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* view,
                             const float* proj, const float* campos, float tan_fovx, float tan_fovy, int prefiltered,
                             float* out_color, float* out_mask, float* out_depth, int* radii, int debug) {
    try {
#ifdef REF_DEPTH
        return CudaRasterizer::Rasterizer::forward(wrap(geom, gctx), wrap(bin, bctx), wrap(img, ictx), P, D, M, bg, W, H,
                                                   means3D, shs, colors_precomp, opacities, mask, scales, scale_modifier,
                                                   rotations, cov3D_precomp, view, proj, campos, tan_fovx, tan_fovy,
                                                   prefiltered != 0, out_color, out_mask, out_depth, radii, debug != 0);
#else
        (void)mask; (void)out_mask; (void)out_depth;
        return CudaRasterizer::Rasterizer::forward(wrap(geom, gctx), wrap(bin, bctx), wrap(img, ictx), P, D, M, bg, W, H,
                                                   means3D, shs, colors_precomp, opacities, scales, scale_modifier,
                                                   rotations, cov3D_precomp, view, proj, campos, tan_fovx, tan_fovy,
                                                   prefiltered != 0, out_color, radii, debug != 0);
#endif
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

REF_API int saga_ref_backward(int P, int D, int M, int R, const float* bg, int W, int H, const float* means3D,
                              const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                              const float* rotations, const float* cov3D_precomp, const float* view, const float* proj,
                              const float* campos, float tan_fovx, float tan_fovy, const int* radii, char* geom_buffer,
                              char* binning_buffer, char* image_buffer, const float* dL_dpix, const float* dL_dout_mask,
                              float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dmask, float* dL_dcolor,
                              float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                              int debug) {
    try {
#ifdef REF_DEPTH
        CudaRasterizer::Rasterizer::backward(P, D, M, R, bg, W, H, means3D, shs, colors_precomp, scales, scale_modifier,
                                             rotations, cov3D_precomp, view, proj, campos, tan_fovx, tan_fovy, radii,
                                             geom_buffer, binning_buffer, image_buffer, dL_dpix, dL_dout_mask, dL_dmean2D,
                                             dL_dconic, dL_dopacity, dL_dmask, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh,
                                             dL_dscale, dL_drot, debug != 0);
#else
        (void)dL_dout_mask; (void)dL_dmask;
        CudaRasterizer::Rasterizer::backward(P, D, M, R, bg, W, H, means3D, shs, colors_precomp, scales, scale_modifier,
                                             rotations, cov3D_precomp, view, proj, campos, tan_fovx, tan_fovy, radii,
                                             geom_buffer, binning_buffer, image_buffer, dL_dpix, dL_dmean2D, dL_dconic,
                                             dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot,
                                             debug != 0);
#endif
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
    return 0;
}

// Mask-only pair (DEPTH/cuda_rasterizer/rasterizer.h: mask_forward / mask_backward); -2 in the other builds.
REF_API int saga_ref_mask_forward(ref_resize_fn geom, void* gctx, ref_resize_fn bin, void* bctx, ref_resize_fn img,
                                  void* ictx, int P, int D, const float* bg, int W, int H, const float* means3D,
                                  const float* opacities, const float* mask, const float* scales, float scale_modifier,
                                  const float* rotations, const float* cov3D_precomp, const float* view, const float* proj,
                                  const float* campos, float tan_fovx, float tan_fovy, int prefiltered, float* out_mask,
                                  int* radii, int debug) {
#ifdef REF_DEPTH
    try {
        return CudaRasterizer::Rasterizer::mask_forward(wrap(geom, gctx), wrap(bin, bctx), wrap(img, ictx), P, D, bg, W, H,
                                                        means3D, opacities, mask, scales, scale_modifier, rotations,
                                                        cov3D_precomp, view, proj, campos, tan_fovx, tan_fovy,
                                                        prefiltered != 0, out_mask, radii, debug != 0);
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
#else
    g_err = "mask_forward exists in the DEPTH build only";
    return -2;
#endif
}

REF_API int saga_ref_mask_backward(int P, int D, int R, const float* bg, int W, int H, const float* means3D,
                                   const float* scales, float scale_modifier, const float* rotations,
                                   const float* cov3D_precomp, const float* view, const float* proj, const float* campos,
                                   float tan_fovx, float tan_fovy, const int* radii, char* geom_buffer,
                                   char* binning_buffer, char* image_buffer, const float* dL_dout_mask, float* dL_dmask,
                                   int debug) {
#ifdef REF_DEPTH
    try {
        CudaRasterizer::Rasterizer::mask_backward(P, D, R, bg, W, H, means3D, scales, scale_modifier, rotations,
                                                  cov3D_precomp, view, proj, campos, tan_fovx, tan_fovy, radii,
                                                  geom_buffer, binning_buffer, image_buffer, dL_dout_mask, dL_dmask,
                                                  debug != 0);
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
    return 0;
#else
    g_err = "mask_backward exists in the DEPTH build only";
    return -2;
#endif
}
