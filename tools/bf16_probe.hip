// bf16 MFMA (32x32x16) next to VALU on gfx950: (a) a VALU-only wave beside a bf16-MFMA-only wave on the same SIMD,
// (b) one wave alternating 1 MFMA + N VALU, (c) v_permlane32_swap semantics.  Companion of overlap_probe.hip (f32 MFMA).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float valu8(int iters, float x)
{
    float a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
    for (int i = 0; i < iters; i++)
        asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                     "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    return a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int F32>
__device__ __forceinline__ float mfma4(int iters, float x)
{
    v16f c0, c1;
    for (int r = 0; r < 16; r++) { c0[r] = 0; c1[r] = 0; }
    v4f a = {x, x, x, x};
    for (int i = 0; i < iters; i++) {
        if (F32)
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %2, %2, %0\n v_mfma_f32_32x32x2_f32 %1, %2, %2, %1\n"
                         "v_mfma_f32_32x32x2_f32 %0, %2, %2, %0\n v_mfma_f32_32x32x2_f32 %1, %2, %2, %1\n"
                         : "+v"(c0), "+v"(c1) : "v"(x));
        else
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %2, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %2, %1\n"
                         "v_mfma_f32_32x32x16_bf16 %0, %2, %2, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %2, %1\n"
                         : "+v"(c0), "+v"(c1) : "v"(a));
    }
    return c0[0] + c1[1];
}

// one wave: 1 MFMA then N independent VALU, repeated
template <int N, int F32>
__device__ __forceinline__ float mixed(int iters, float x)
{
    v16f c0, c1;
    for (int r = 0; r < 16; r++) { c0[r] = 0; c1[r] = 0; }
    v4f a = {x, x, x, x};
    float a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3;
    for (int i = 0; i < iters; i++) {
        if (F32) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %1, %0" : "+v"(c0) : "v"(x));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %1, %0" : "+v"(c0) : "v"(a));
        for (int k = 0; k < N / 4; k++)
            asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        if (F32) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %1, %0" : "+v"(c1) : "v"(x));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %1, %0" : "+v"(c1) : "v"(a));
        for (int k = 0; k < N / 4; k++)
            asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    }
    return c0[0] + c1[1] + a0 + a1 + a2 + a3;
}

template <int F32>
__global__ void __launch_bounds__(512) side_by_side(int mode, int iters, long long* cyc, float* sink)
{
    const int wave = threadIdx.x >> 6;
    const bool is_m = wave >= 4;
    __syncthreads();
    const long long t0 = clock64();
    float r = 0;
    if (is_m) { if (mode & 2) r = mfma4<F32>(iters / 4, (float)threadIdx.x); }
    else      { if (mode & 1) r = valu8(iters, (float)threadIdx.x); }
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
    if (r == 12345.f) sink[0] = r;
}

template <int N, int F32>
__global__ void __launch_bounds__(256) one_wave_mix(int iters, long long* cyc, float* sink, int nw)
{
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    const long long t0 = clock64();
    float r = mixed<N, F32>(iters, (float)threadIdx.x);
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
    if (r == 12345.f) sink[0] = r;
}

__global__ void swap_test(unsigned* out, int variant)
{
    unsigned a = 100 + threadIdx.x, b = 200 + threadIdx.x;
    if (variant == 0) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    else if (variant == 1) asm volatile("s_nop 7\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 7" : "+v"(a), "+v"(b));
    else asm volatile("s_nop 7\n\tv_permlane32_swap_b32_e64 %0, %1\n\ts_nop 7" : "+v"(a), "+v"(b));
    out[threadIdx.x] = a;
    out[64 + threadIdx.x] = b;
}

int main()
{
    long long* d; float* s; hipMalloc(&d, 64 * 8); hipMalloc(&s, 4);
    long long h[8];
    const int iters = 4096;
    for (int f32 = 0; f32 < 2; f32++) {
        printf("%s MFMA waves (4 per CU, 1 per SIMD) beside 4 VALU waves; V = %d fma, M = %d mfma\n", f32 ? "f32 32x32x2" : "bf16 32x32x16", iters * 8, iters);
        for (int mode = 1; mode <= 3; mode++) {
            for (int rep = 0; rep < 2; rep++) {
                if (f32) side_by_side<1><<<1, 512>>>(mode, iters, d, s); else side_by_side<0><<<1, 512>>>(mode, iters, d, s);
            }
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            printf("  mode %d: V wave %8lld ticks  M wave %8lld ticks\n", mode, h[0], h[4]);
        }
    }
#define MIX(N, F, NT) { one_wave_mix<N, F><<<1, NT>>>(iters, d, s, NT / 64); one_wave_mix<N, F><<<1, NT>>>(iters, d, s, NT / 64); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); \
    printf("  %s: 1 mfma + %2d fma per step, %d wave(s)/SIMD... block %3d thr: %8lld ticks for %d steps = %.1f per step\n", F ? "f32 " : "bf16", N, NT > 256 ? 2 : 1, NT, h[0], iters * 2, (double)h[0] / (iters * 2)); }
    printf("one wave alternating MFMA and VALU (256 threads = 1 wave per SIMD)\n");
    MIX(0, 0, 256) MIX(4, 0, 256) MIX(8, 0, 256) MIX(16, 0, 256) MIX(32, 0, 256)
    MIX(0, 1, 256) MIX(4, 1, 256) MIX(8, 1, 256) MIX(16, 1, 256) MIX(32, 1, 256)
    unsigned* o; hipMalloc(&o, 128 * 4); unsigned ho[128];
    for (int variant = 0; variant < 3; variant++) {
        swap_test<<<1, 64>>>(o, variant);
        hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
        printf("permlane32_swap v%d (a = 100 + lane, b = 200 + lane): a[0]=%u a[31]=%u a[32]=%u a[63]=%u | b[0]=%u b[31]=%u b[32]=%u b[63]=%u\n", variant,
               ho[0], ho[31], ho[32], ho[63], ho[64], ho[95], ho[96], ho[127]);
    }
    return 0;
}
