// Does the L2 combine scattered 8-byte stores into whole lines when a workgroup's store frontier is small?
// The emit pass of the binning (csrc/binning.h: bin_spans_kernel<true>) stores 2.9 M 8-byte entries per cfg3 view, each into a tile's
// segment at the cursor of the storing workgroup's slice: 256 slices x 8160 tiles, ~1.4 entries per (slice, tile) -- every store is
// its own 64-byte write (WRITE_SIZE = 64 B x stores, profiles/r05_pmc.md) and the pass is bound by them (profiles/r06_front_half.md).
// Pattern A restates that.  Pattern B gives every workgroup a BAND of tiles (16 bands x 16 sub-slices): ~22 entries per (workgroup,
// tile), contiguous, written one at a time in random tile order -- the same number of stores, a frontier of 510 lines per workgroup.
//   hipcc --offload-arch=gfx950 -O3 tools/write_combine_probe.hip -o tools/write_combine_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int TILES = 8160, NWG = 256, PER_WG = 11456;   // 2.93 M entries

__device__ inline uint32_t rng(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

// cursors in LDS, one per tile this workgroup writes; slot = base(tile) + my offset inside the tile's segment + cursor++
template <bool BANDED>
__global__ void __launch_bounds__(1024) emit_like(uint2* __restrict__ entries, int seg /* entries per tile segment */)
{
    extern __shared__ uint32_t s_cur[];
    const int wg = blockIdx.x, tid = threadIdx.x;
    const int ntile = BANDED ? TILES / 16 : TILES;
    for (int t = tid; t < ntile; t += 1024) s_cur[t] = 0;
    __syncthreads();
    uint32_t s = 0x9E3779B9u * (wg * 1024 + tid + 1);
    const int band = wg >> 4, sub = wg & 15;
    const int per_tile = BANDED ? seg / 16 : seg / NWG;   // this workgroup's share of a tile's segment
    for (int k = tid; k < PER_WG; k += 1024) {
        const int t = (int)(rng(s) % (uint32_t)ntile);
        const uint32_t c = atomicAdd(&s_cur[t], 1u);
        if ((int)c >= per_tile) continue;
        const size_t tile = BANDED ? (size_t)band * ntile + t : (size_t)t;
        const size_t slot = tile * seg + (size_t)(BANDED ? sub : wg) * per_tile + c;
        entries[slot] = make_uint2(s, (uint32_t)k);
    }
}

int main()
{
    const int seg = 512;   // entries per tile segment (mean load 359)
    uint2* d;
    hipMalloc(&d, (size_t)TILES * seg * sizeof(uint2));
    hipFuncSetAttribute((const void*)emit_like<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        for (int banded = 0; banded < 2; banded++) {
            hipMemset(d, 0, (size_t)TILES * seg * sizeof(uint2));
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int it = 0; it < 10; it++) {
                if (banded) hipLaunchKernelGGL(emit_like<true>, dim3(NWG), dim3(1024), (TILES / 16) * 4, 0, d, seg);
                else hipLaunchKernelGGL(emit_like<false>, dim3(NWG), dim3(1024), TILES * 4, 0, d, seg);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.1f us per launch (%d stores of 8 bytes)\n", banded ? "B banded (16 bands x 16 sub-slices)" : "A all tiles per workgroup (256 slices)    ", 1e3 * ms / 10, NWG * PER_WG);
        }
    }
    return 0;
}
