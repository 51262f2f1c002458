// binning.h -- tile binning: key emission and tile-range identification.
// (The prefix scan and the 64-bit radix sort are hipcub device primitives, called from mi_rast.hip,
// exactly where the reference calls cub::DeviceScan / cub::DeviceRadixSort.)
#pragma once

#include "common.h"

namespace mirast {

// CF/cuda_rasterizer/rasterizer_impl.cu:70-111.  One thread per Gaussian walks its tile rect
// row-major and emits (tile<<32 | depth_bits, idx) at offsets[idx-1] + k.
__global__ void __launch_bounds__(256) duplicate_with_keys_kernel(
    int P, const float2* __restrict__ points_xy, const float* __restrict__ depths, const uint32_t* __restrict__ offsets,
    uint64_t* __restrict__ keys_unsorted, uint32_t* __restrict__ values_unsorted, const int* __restrict__ radii,
    uint32_t gx, uint32_t gy)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    const int r = radii[idx];
    if (r > 0) {
        uint32_t off = (idx == 0) ? 0 : offsets[idx - 1];
        uint2 rect_min, rect_max;
        const float2 p = points_xy[idx];
        getRect(p.x, p.y, r, rect_min, rect_max, gx, gy);
        const uint32_t dbits = __float_as_uint(depths[idx]);
        for (int y = rect_min.y; y < (int)rect_max.y; y++) {
            for (int x = rect_min.x; x < (int)rect_max.x; x++) {
                uint64_t key = (uint64_t)((uint32_t)y * gx + (uint32_t)x);
                key <<= 32;
                key |= dbits;
                keys_unsorted[off] = key;
                values_unsorted[off] = (uint32_t)idx;
                off++;
            }
        }
    }
}

// CF/cuda_rasterizer/rasterizer_impl.cu:116-138
__global__ void __launch_bounds__(256) identify_tile_ranges_kernel(int L, const uint64_t* __restrict__ keys,
                                                                   uint2* __restrict__ ranges)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= L) return;
    const uint32_t currtile = (uint32_t)(keys[idx] >> 32);
    if (idx == 0)
        ranges[currtile].x = 0;
    else {
        const uint32_t prevtile = (uint32_t)(keys[idx - 1] >> 32);
        if (currtile != prevtile) {
            ranges[prevtile].y = idx;
            ranges[currtile].x = idx;
        }
    }
    if (idx == L - 1) ranges[currtile].y = L;
}

}  // namespace mirast
