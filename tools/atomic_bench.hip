// Micro-benchmark: throughput of global float atomics in the patterns the backward blend uses.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_bench.hip -o /tmp/atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int MODE>
__global__ void k(float* buf, uint32_t rows, int iters, int stride_rows)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    for (int it = 0; it < iters; it++) {
        uint32_t row = hash32(wave * 7919u + it * 104729u) % rows;
        if (MODE == 4) row = (wave * iters + it) % rows;  // sequential rows, no contention
        float* p = buf + (size_t)row * 32;
        if (MODE == 0) { if (lane < 32) atomicAdd(p + lane, 1.0f); }              // 32 lanes, one 128-B line
        if (MODE == 1) { atomicAdd(p + (lane & 31) + (lane >> 5) * 32 * stride_rows, 1.0f); }  // 64 lanes, two lines
        if (MODE == 2) { if (lane < 6) atomicAdd(buf + (size_t)(hash32(row + lane * 977u) % (rows * 32)), 1.0f); }  // 6 scattered
        if (MODE == 3) { atomicAdd(buf + (size_t)(hash32(row * 64 + lane) % (rows * 32)), 1.0f); }  // 64 scattered
        if (MODE == 4) { if (lane < 32) atomicAdd(p + lane, 1.0f); }
        if (MODE == 5) { if (lane < 32) p[lane] += 1.0f; }                          // plain RMW (no atomic) for reference
        if (MODE == 6) { if (lane < 8) atomicAdd(p + lane, 1.0f); }                 // 8 lanes, one 32-B sector
        // the blend backward's real instruction shapes: every lane group of one instruction goes to a different random row
        if (MODE == 7) { const uint32_t r4 = hash32(row * 4 + (lane >> 4)) % rows; atomicAdd(buf + (size_t)r4 * 32 + (lane & 15) + 16 * (it & 1), 1.0f); }  // 4 rows x 64 B
        if (MODE == 8) { const uint32_t r2 = hash32(row * 2 + (lane >> 5)) % rows; atomicAdd(buf + (size_t)r2 * 32 + (lane & 31), 1.0f); }                  // 2 rows x 128 B
        if (MODE == 9) { const uint32_t r8 = hash32(row * 8 + (lane >> 3)) % rows; atomicAdd(buf + (size_t)r8 * 8 + (lane & 7), 1.0f); }                    // 8 rows x 32 B
        if (MODE == 10) { const uint32_t r8 = hash32(row * 8 + (lane >> 3)) % rows; if ((lane & 7) < 6) atomicAdd(buf + (size_t)r8 * 8 + (lane & 7), 1.0f); }  // 8 rows x 24 B
    }
}

template <int MODE>
void run(const char* name, float* buf, uint32_t rows, int lanes_per_instr)
{
    const int blocks = 4096, threads = 256, iters = 64;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, threads>>>(buf, rows, 4, 1);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<blocks, threads>>>(buf, rows, iters, 1);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr = (double)blocks * threads / 64 * iters;
    printf("%-44s %8.3f ms  %7.2f G wave-instr/s  %8.2f G lane-atomics/s\n", name, ms, instr / ms / 1e6, instr * lanes_per_instr / ms / 1e6);
}

int main()
{
    const uint32_t rows = 1u << 20;
    float* buf; hipMalloc(&buf, (size_t)rows * 32 * 4); hipMemset(buf, 0, (size_t)rows * 32 * 4);
    run<0>("32 lanes -> one 128B line, random row", buf, rows, 32);
    run<1>("64 lanes -> two adjacent 128B lines", buf, rows, 64);
    run<2>("6 scattered lanes", buf, rows, 6);
    run<3>("64 scattered lanes", buf, rows, 64);
    run<4>("32 lanes -> one line, sequential rows", buf, rows, 32);
    run<5>("32 lanes plain RMW (no atomic)", buf, rows, 32);
    run<6>("8 lanes -> 32B", buf, rows, 8);
    run<7>("64 lanes -> 4 random rows x 64B", buf, rows, 64);
    run<8>("64 lanes -> 2 random rows x 128B", buf, rows, 64);
    run<9>("64 lanes -> 8 random rows x 32B", buf, rows, 64);
    run<10>("48 lanes -> 8 random rows x 24B", buf, rows, 48);
    return 0;
}
