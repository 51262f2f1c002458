// Where do float atomics saturate on gfx950 -- in the CU's memory pipeline or in the L2 channels -- and what is a line worth that
// is already in the issuing XCD's L2?  (round 4; tools/atomic_bench.hip measured ~20 G 64-byte segment-atomics/s device-wide on a
// 128-MB working set, the figure the backward blend's 8.7 M requests per view are priced with in DESIGN.md section 11.)
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_limit_probe.hip -o tools/atomic_limit_probe.bin
//   A  device rate against the number of workgroups (1024 threads, 150 KB of LDS: one per CU): a per-CU limit scales with the CUs in
//      use, an L2 / fabric limit is reached by a fraction of them
//   B  working set 128 MB .. 1 MB, rows drawn from the whole set by every XCD (lines shared between the eight L2s)
//   C  the same sizes, every XCD drawing from an eighth of its own (lines private to one L2; workgroup b runs on XCD b % 8)
//   D  integer instead of float adds; returning instead of non-returning
//   E  the backward blend's chunk: 8 feature instructions (2 rows x 128 B) + 2 geometry instructions (8 rows x 24 B of 32) per 16 rows
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// MODE 0: 2 rows x 128 B per instruction (feature rows), float, no return; 1: the same, uint32; 2: float, returning;
// 3: chunk mix (8 feature + 2 geometry instructions per iteration); 4: 4 rows x 64 B
template <int MODE, bool PRIVATE>
__global__ void __launch_bounds__(1024) k(float* buf, uint32_t rows, int iters, float* sink)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t per = rows >> 3;
    float acc = 0.f;
    auto pick = [&](uint32_t h) { return PRIVATE ? xcd * per + h % per : h % rows; };
    for (int it = 0; it < iters; it++) {
        const uint32_t h0 = hash32(wave * 7919u + it * 104729u);
        if (MODE == 0) {
            const uint32_t r = pick(hash32(h0 * 2 + (lane >> 5)));
            atomicAdd(buf + (size_t)r * 32 + (lane & 31), 1.0f);
        } else if (MODE == 1) {
            const uint32_t r = pick(hash32(h0 * 2 + (lane >> 5)));
            atomicAdd(reinterpret_cast<uint32_t*>(buf) + (size_t)r * 32 + (lane & 31), 1u);
        } else if (MODE == 2) {
            const uint32_t r = pick(hash32(h0 * 2 + (lane >> 5)));
            acc += atomicAdd(buf + (size_t)r * 32 + (lane & 31), 1.0f);
        } else if (MODE == 3) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const uint32_t r = pick(hash32(h0 * 16 + 2 * q + (lane >> 5)));
                atomicAdd(buf + (size_t)r * 32 + (lane & 31), 1.0f);
            }
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const uint32_t r = pick(hash32(h0 * 16 + 8 * q + (lane >> 3)));
                atomicAdd(buf + (size_t)rows * 32 + (size_t)r * 8 + (lane & 7), 1.0f);
            }
        } else {
            const uint32_t r = pick(hash32(h0 * 4 + (lane >> 4)));
            atomicAdd(buf + (size_t)r * 32 + (lane & 15) + 16 * (it & 1), 1.0f);
        }
    }
    if (MODE == 2 && acc == -1.f) sink[0] = acc + lds[0];
}

template <int MODE, bool PRIVATE>
double run(float* buf, uint32_t rows, int blocks, int iters, double segs_per_iter, size_t lds)
{
    hipFuncSetAttribute((const void*)k<MODE, PRIVATE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, PRIVATE>), dim3(blocks), dim3(1024), lds, 0, buf, rows, 4, buf);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE, PRIVATE>), dim3(blocks), dim3(1024), lds, 0, buf, rows, iters, buf);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return (double)blocks * 16 * iters * segs_per_iter / ms / 1e6;  // G segments / s
}

int main()
{
    const uint32_t max_rows = 1u << 20;
    float* buf;
    hipMalloc(&buf, (size_t)max_rows * 40 * 4);
    hipMemset(buf, 0, (size_t)max_rows * 40 * 4);
    const size_t big_lds = 150 * 1024;
    printf("A  2 rows x 128 B, 128-MB set, one 1024-thread workgroup per CU: G segment-atomics/s against workgroups\n");
    for (int blocks : {32, 64, 128, 192, 256}) printf("   %4d workgroups  %7.2f\n", blocks, run<0, false>(buf, max_rows, blocks, 256, 4, big_lds));
    printf("A' the same with 4 / 8 waves per CU in use (256 workgroups of 1024 threads is 16)\n");
    printf("B/C  2 rows x 128 B, 2048 workgroups: working set, shared rows | XCD-private rows\n");
    for (uint32_t rows : {1u << 20, 1u << 18, 1u << 16, 1u << 14, 1u << 13})
        printf("   %6.1f MB   shared %7.2f   private %7.2f\n", rows * 128.0 / 1e6, run<0, false>(buf, rows, 2048, 64, 4, 0),
               run<0, true>(buf, rows, 2048, 64, 4, 0));
    printf("D  128-MB set, 2048 workgroups: float %7.2f  uint32 %7.2f  float returning %7.2f  4 rows x 64 B %7.2f\n",
           run<0, false>(buf, max_rows, 2048, 64, 4, 0), run<1, false>(buf, max_rows, 2048, 64, 4, 0), run<2, false>(buf, max_rows, 2048, 64, 4, 0),
           run<4, false>(buf, max_rows, 2048, 64, 4, 0));
    printf("E  chunk mix (32 feature + 16 geometry segments per iteration), 2048 workgroups:\n");
    for (uint32_t rows : {1u << 20, 1u << 16, 1u << 14})
        printf("   %6.1f MB   shared %7.2f   private %7.2f\n", rows * 160.0 / 1e6, run<3, false>(buf, rows, 2048, 16, 48, 0),
               run<3, true>(buf, rows, 2048, 16, 48, 0));
    return 0;
}
