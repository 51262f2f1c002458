// VALU issue rate per SIMD vs. number of resident VALU-issuing waves (gfx950).  One block on one CU.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void __launch_bounds__(1024) probe(int iters, long long* cyc, float* sink)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0)
            asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                         "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        else if (KIND == 1)
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                         "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        else if (KIND == 2)
            asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n"
                         "v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n"
                         : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6));
    }
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.f) sink[0] = a0;
}
template <int KIND> void run(const char* name)
{
    long long* d; float* s; hipMalloc(&d, 16 * 8); hipMalloc(&s, 4);
    long long h[16];
    const int iters = 4096;
    for (int threads : {256, 512, 768, 1024}) {
        probe<KIND><<<1, threads>>>(iters, d, s);
        probe<KIND><<<1, threads>>>(iters, d, s);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        long long mx = 0; for (int w = 0; w < threads / 64; w++) mx = h[w] > mx ? h[w] : mx;
        printf("%-10s %d waves/SIMD: %8lld cyc for %d instr/wave -> %.2f cyc/instr/wave, %.2f cyc/instr/SIMD\n", name, threads / 256, mx,
               iters * 8, (double)mx / (iters * 8), (double)mx / (iters * 8) / (threads / 256));
    }
}
int main() { run<0>("v_fma"); run<1>("v_exp"); run<2>("v_pk_fma"); return 0; }
