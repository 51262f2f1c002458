// TEST INFRASTRUCTURE ONLY (see oracle/ref_shim.cpp, oracle/build_ref.py).
// C-ABI shim around the reference's `SimpleKNN::knn` (submodules/simple-knn/simple_knn.h:17-21, simple_knn.cu:185-218),
// the core of `simple_knn._C.distCUDA2` (spatial.cu:16-25): mean squared distance to the 3 nearest neighbours.
#include <hip/hip_runtime.h>
#include <exception>
#include <string>

#include "simple_knn.h"

static thread_local std::string g_err;

extern "C" __attribute__((visibility("default"))) const char* saga_ref_knn_last_error(void) { return g_err.c_str(); }

// points: device float[3P]; mean_dists: device float[P] (written).
extern "C" __attribute__((visibility("default"))) int saga_ref_knn(int P, float* points, float* mean_dists) {
    try {
        SimpleKNN::knn(P, (float3*)points, mean_dists);
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
    return 0;
}
