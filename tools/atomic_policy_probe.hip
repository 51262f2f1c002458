// Does any encoding of a float atomic get past the ~20 G requests/s of the L2's atomic units (tools/atomic_limit_probe.hip)?
// The same instruction shape as the backward blend's feature atomics (64 lanes -> 2 random rows x 128 B = 4 segment requests),
// 128-MB working set, issued through inline asm with the cache-policy / scope bits of the gfx940 ISA (for atomics: sc0 = return the
// old value, sc1 = system scope, nt = non-temporal), next to packed-bf16 (two values per dword: half the requests per value), f64
// and plain stores of the same shape.  Round 4; build:
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_policy_probe.hip -o tools/atomic_policy_probe.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int MODE>
__global__ void __launch_bounds__(256) k(float* buf, uint32_t rows, int iters)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    for (int it = 0; it < iters; it++) {
        const uint32_t h0 = hash32(wave * 7919u + it * 104729u);
        const uint32_t r = hash32(h0 * 2 + (lane >> 5)) % rows;
        float* p = buf + (size_t)r * 32 + (lane & 31);
        const float one = 1.0f;
        if (MODE == 0) asm volatile("global_atomic_add_f32 %0, %1, off" ::"v"(p), "v"(one) : "memory");
        if (MODE == 1) asm volatile("global_atomic_add_f32 %0, %1, off sc1" ::"v"(p), "v"(one) : "memory");
        if (MODE == 2) asm volatile("global_atomic_add_f32 %0, %1, off nt" ::"v"(p), "v"(one) : "memory");
        if (MODE == 3) asm volatile("global_atomic_add_f32 %0, %1, off sc1 nt" ::"v"(p), "v"(one) : "memory");
        if (MODE == 4) {  // packed bf16: the row is 32 bf16 pairs = 16 dwords... here 32 dwords of pairs (same bytes as f32: requests per BYTE)
            const uint32_t two = 0x3f803f80u;
            asm volatile("global_atomic_pk_add_bf16 %0, %1, off" ::"v"(p), "v"(two) : "memory");
        }
        if (MODE == 5) {  // f64: 32 lanes x 8 B per row -> the wave covers 4 rows x 128 B?  keep the bytes: lane -> 8 B of 2 rows x 256 B
            double* q = reinterpret_cast<double*>(buf) + (size_t)(r & ~1u) * 16 + (lane & 31);
            const double oned = 1.0;
            asm volatile("global_atomic_add_f64 %0, %1, off" ::"v"(q), "v"(oned) : "memory");
        }
        if (MODE == 6) *p = one;                                               // plain store, same shape
        if (MODE == 7) __builtin_nontemporal_store(one, p);                    // non-temporal store
        if (MODE == 8) asm volatile("global_atomic_add %0, %1, off" ::"v"(p), "v"(1u) : "memory");
    }
}

template <int MODE>
void run(const char* name, float* buf, uint32_t rows, double segs_per_iter)
{
    const int blocks = 8192, iters = 64;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, rows, 4);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, rows, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double instr = (double)blocks * 4 * iters;
    printf("%-52s %8.3f ms  %7.2f G wave-instr/s  %7.2f G 64-byte requests/s\n", name, ms, instr / ms / 1e6, instr * segs_per_iter / ms / 1e6);
}

int main()
{
    const uint32_t rows = 1u << 20;
    float* buf;
    hipMalloc(&buf, (size_t)rows * 32 * 4 + 4096);
    hipMemset(buf, 0, (size_t)rows * 32 * 4 + 4096);
    run<0>("global_atomic_add_f32", buf, rows, 4);
    run<1>("global_atomic_add_f32 sc1 (system scope)", buf, rows, 4);
    run<2>("global_atomic_add_f32 nt", buf, rows, 4);
    run<3>("global_atomic_add_f32 sc1 nt", buf, rows, 4);
    run<4>("global_atomic_pk_add_bf16 (2 values per dword)", buf, rows, 4);
    run<5>("global_atomic_add_f64 (2 rows x 256 B)", buf, rows, 8);
    run<6>("plain global_store_dword, same shape", buf, rows, 4);
    run<7>("non-temporal store, same shape", buf, rows, 4);
    run<8>("global_atomic_add_u32", buf, rows, 4);
    return 0;
}
