// In-wave interleave: 1 MFMA (16x16x4 f32, 32 cyc) + NV VALU per group; every wave runs the same stream (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int NV, bool MF, bool SAL>
__global__ void __launch_bounds__(1024) probe(int iters, long long* cyc, float* sink)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    int s0 = 1;
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            if (MF) {
                if (g == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %1, %0" : "+v"(c0) : "v"(a0));
                if (g == 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %1, %0" : "+v"(c1) : "v"(a0));
                if (g == 2) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %1, %0" : "+v"(c2) : "v"(a0));
                if (g == 3) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %1, %0" : "+v"(c3) : "v"(a0));
            }
#pragma unroll
            for (int v = 0; v < NV; v++) {
                if ((v & 3) == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a0));
                if ((v & 3) == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a1));
                if ((v & 3) == 2) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a2));
                if ((v & 3) == 3) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a3));
                if (SAL) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s0));
            }
        }
    }
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
    if (a0 + a1 + a2 + a3 + c0[0] + c1[0] + c2[0] + c3[0] + s0 == 12345.f) sink[0] = a0;
}
template <int NV, bool MF, bool SAL> void run()
{
    long long* d; float* s; hipMalloc(&d, 16 * 8); hipMalloc(&s, 4);
    long long h[16];
    const int iters = 2048;
    printf("group = %s%d VALU%s:", MF ? "1 MFMA + " : "", NV, SAL ? " (+1 SALU each)" : "");
    for (int threads : {256, 512, 768, 1024}) {
        probe<NV, MF, SAL><<<1, threads>>>(iters, d, s);
        probe<NV, MF, SAL><<<1, threads>>>(iters, d, s);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        long long mx = 0; for (int w = 0; w < threads / 64; w++) mx = h[w] > mx ? h[w] : mx;
        printf("  W=%d: %6.1f cyc/group/wave (%5.1f /SIMD)", threads / 256, (double)mx / (iters * 4), (double)mx / (iters * 4) / (threads / 256));
    }
    printf("\n");
}
int main()
{
    run<8, false, false>(); run<8, true, false>(); run<4, true, false>(); run<12, true, false>(); run<16, true, false>();
    run<8, false, true>(); run<8, true, true>();
    return 0;
}
