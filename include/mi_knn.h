/*
 * mi_knn.h -- C-ABI of the exact K-nearest-neighbour search at the edges of the rasterizer hot path
 * (SURVEY.md 8(f) rows 1 and 4).  Same conventions as mi_rast.h: device pointers, fp32, contiguous; 0 on success,
 * else a code and mi_rast_last_error(); `stream` is a hipStream_t; no state between calls.
 *
 * Replaces
 *   pytorch3d.ops.knn_points(p1, p2, K).idx / .dists   as SAGA calls it: scene/gaussian_model_ff.py:326,347,380
 *                                                      (batch of one; p1 == p2, or p2 = a subset of p1)
 *   simple_knn._C.distCUDA2(points)                    submodules/simple-knn/spatial.cu:16-25 -> SimpleKNN::knn
 *                                                      (simple_knn.cu:185-218): scene/gaussian_model.py:20,
 *                                                      gaussian_model_ff.py:21 (create_from_pcd)
 * The search is exact (K smallest squared Euclidean distances, ascending, ties by index); see csrc/knn.h.
 */
#ifndef MI_KNN_H
#define MI_KNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_KNN_MAX_K 32

/* Bytes of device scratch the index over M reference points needs (Morton codes, sorted points, two box levels). */
size_t mi_knn_workspace_bytes(int M);

/* Builds the index over ref [M,3] into `workspace` (>= mi_knn_workspace_bytes(M), 256-B aligned).  The index refers to
 * `ref` only through the copies it makes: ref may be freed afterwards. */
int mi_knn_build(int M, const float* ref, void* workspace, size_t workspace_bytes, void* stream);

/* K nearest references of every query.
 *   query == NULL : the queries ARE the references (N is ignored, M rows are written, row i = reference i).
 *                   exclude_self != 0 leaves the point itself out of its own list.
 *   query [N,3]   : arbitrary points; exclude_self is ignored.
 * idx [rows,K] int64 (reference indices; -1 where fewer than K references exist), dist2 [rows,K] squared distances. */
int mi_knn_query(int N, const float* query, int M, const void* workspace, int K, int exclude_self,
                 int64_t* idx, float* dist2, void* stream);

/* distCUDA2: out[i] = mean of the squared distances from point i to its 3 nearest OTHER points
 * (simple_knn.cu:145-183).  Builds its own index in `workspace` (>= mi_knn_workspace_bytes(P)). */
int mi_knn_mean_dist2(int P, const float* points, void* workspace, size_t workspace_bytes, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
