/* mi_contrastive.h -- C-ABI of the contrastive-loss front end, in libmi_rast.so.
 *
 * Replaces, for the caller of the rasterizer's output, train_contrastive_feature.py:234-253:
 *
 *     rendered_feature_norm = rendered_features.norm(dim=0, p=2).mean()                       (:234)
 *     rendered_features = F.interpolate(rendered_features[None], (H, W), mode='bilinear')[0]  (:237)
 *     feature_with_scale = rendered_features[None].repeat(N, 1, 1, 1) * gates[:, :, None, None]   (:247-248)
 *     sampled = feature_with_scale[:, :, sampled_ray].permute(0, 2, 1)                        (:250-252)
 *     out = F.normalize(sampled, dim=-1, p=2)                                                 (:254)
 *
 * and their autograd backward.  The reference materialises an (N, C, H, W) tensor (2.6 GB at 10 x 32 x 1080p) before
 * it keeps S ~ 1000 rays; every step is per pixel, so the rays are bilinear-sampled FIRST (same weights as
 * F.interpolate(..., align_corners=False)).  The dense part -- the feature-norm regulariser -- is one streaming pass.
 *
 * forward : ONE launch.  Reads `rendered` (C, h, w) once: norm_sum[16 k] += partial sums over pixels of ||f(:, p)||_2 for
 *           k < MI_CONTRASTIVE_NORM_SLOTS (doubles, one per 128-byte line: same-address atomics serialise; the caller zeroes the
 *           MI_CONTRASTIVE_NORM_SLOTS * 16 doubles, adds the slots up and divides by h w), inv_norm[p] = 1 / ||f(:, p)|| (0 where the norm is 0, as torch's norm
 *           backward); and for the S rays `ray_yx` (pixel coordinates in the (H, W) mask grid, row-major order = the order
 *           boolean-mask indexing produces): ray_feat (S, C) = the bilinear samples, out (N, S, C) = normalize(ray * gate),
 *           inv_len (N, S) = 1 / max(||ray * gate||, 1e-12).
 * backward: two launches.  dL_drendered (C, h, w) is WRITTEN IN FULL: g_norm / (h w) * f * inv_norm (the regulariser term;
 *           g_norm = dL/d rendered_feature_norm, a device scalar, may be NULL = 0), then the 4 S C tap gradients of the rays
 *           are added with float atomics; dL_dgates (N, C) must be zeroed by the caller and receives atomics.
 *
 * All pointers are device pointers, fp32 unless noted, contiguous; `stream` is a hipStream_t.  1 <= C <= 256, N >= 1,
 * S >= 0.  Algorithmic bytes: forward 4 C h w read (+ 4 h w written), backward 4 C h w read + 4 C h w written: three
 * streams of the feature image (265 MB each at 32 x 1080p); HBM-bound.
 * Returns 0 or an MI_RAST_ERR_* code (mi_rast_last_error() holds the text). */
#ifndef MI_CONTRASTIVE_H
#define MI_CONTRASTIVE_H

#define MI_CONTRASTIVE_NORM_SLOTS 64

#ifdef __cplusplus
extern "C" {
#endif

int mi_contrastive_forward(int C, int h, int w, const float* rendered, int H, int W, int S, const int* ray_yx /* [S,2] (y, x) */,
                           int N, const float* gates /* [N,C] */, float* out /* [N,S,C] */, float* ray_feat /* [S,C] */,
                           float* inv_len /* [N,S] */, float* inv_norm /* [h w] */, double* norm_sum /* [MI_CONTRASTIVE_NORM_SLOTS * 16], zeroed by the caller */,
                           void* stream);

int mi_contrastive_backward(int C, int h, int w, const float* rendered, int H, int W, int S, const int* ray_yx, int N,
                            const float* gates, const float* out, const float* ray_feat, const float* inv_len,
                            const float* inv_norm, const float* dL_dout /* [N,S,C] */, const float* g_norm /* [1] or NULL */,
                            float* dL_drendered /* [C,h,w], written in full */, float* dL_dgates /* [N,C], zeroed by the caller */,
                            void* stream);

#ifdef __cplusplus
}
#endif
#endif
