/* mi_knn_smooth.h -- C-ABI of the fused KNN feature smoothing, in libmi_rast.so.
 *
 * Replaces the PyTorch expression of FeatureGaussianModel.get_smoothed_point_features
 * (scene/gaussian_model_ff.py:338-364: normalize -> gather selected neighbour columns -> mean) fused with the
 * renderer's re-normalisation of its result (gaussian_renderer/__init__.py:362-363), and its autograd backward.
 * All pointers are device pointers (fp32 / int32), row-major; `stream` is a hipStream_t.  C must be 32 or 64,
 * K <= 32.  sel_mask: bit s set <=> neighbour column s takes part (the reference draws int(K*dropout) columns with
 * torch.randperm; all K columns when dropout is outside (0, 1)).  Returns 0 or an MI_RAST_ERR_* code
 * (mi_rast_last_error() holds the text). */
#ifndef MI_KNN_SMOOTH_H
#define MI_KNN_SMOOTH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int mi_knn_smooth_forward(int P, int C, int K, const int* knn_idx /* [P,K] */, uint32_t sel_mask,
                          const float* features /* [P,C] */, float* out /* [P,C] */, int normalize_out, void* stream);

/* inv_offsets[P+1] / inv_entries[P*K]: the inverse neighbour lists -- entries (i << 5 | column) with
 * knn_idx[i][column] == j, grouped by j -- built once per neighbour map by the caller.
 * dmean: scratch [P,C].  dL_dfeatures is overwritten (not accumulated). */
int mi_knn_smooth_backward(int P, int C, int K, const int* knn_idx, const int* inv_offsets,
                           const uint32_t* inv_entries, uint32_t sel_mask, const float* features,
                           const float* dL_dout, float* dmean, float* dL_dfeatures, int normalize_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
