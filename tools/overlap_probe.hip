// Does a VALU-only wave overlap with an MFMA-only wave on the SAME SIMD (gfx950)?  One 512-thread block on one CU:
// waves 0..3 run VALU loops, waves 4..7 run f32 MFMA loops; cycles per wave for each role alone and together.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int KIND>
__device__ __forceinline__ float valu_body(int iters, float x)
{
    float a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {  // 8 independent fma
            asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                         "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 1) {  // 8 dependent mul
            asm volatile("v_mul_f32 %0, %0, %0\n v_mul_f32 %0, %0, %0\n v_mul_f32 %0, %0, %0\n v_mul_f32 %0, %0, %0\n"
                         "v_mul_f32 %0, %0, %0\n v_mul_f32 %0, %0, %0\n v_mul_f32 %0, %0, %0\n v_mul_f32 %0, %0, %0\n" : "+v"(a0));
        } else if (KIND == 2) {  // 8 independent exp
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                         "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 3) {  // readlane -> sgpr -> mul with sgpr (4 pairs)
            int s0, s1, s2, s3;
            asm volatile("v_readlane_b32 %4, %0, 3\n v_readlane_b32 %5, %1, 5\n v_readlane_b32 %6, %2, 7\n v_readlane_b32 %7, %3, 9\n"
                         "v_mul_f32 %0, %4, %0\n v_mul_f32 %1, %5, %1\n v_mul_f32 %2, %6, %2\n v_mul_f32 %3, %7, %3\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3));
        } else if (KIND == 4) {  // cmp + cndmask pairs (VCC dependency)
            asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n"
                         "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
        }
    }
    return a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int NOPS>
__device__ __forceinline__ float mfma_body(int iters, float x)
{
    v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; i++) {
#define NOPSTR(n) (n == 0 ? "" : n == 1 ? "s_nop 7\n" : n == 2 ? "s_nop 7\n s_nop 7\n" : "s_nop 7\n s_nop 7\n s_nop 7\n")
        if (NOPS == 0)
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %4, %4, %0\n v_mfma_f32_16x16x4_f32 %1, %4, %4, %1\n"
                         "v_mfma_f32_16x16x4_f32 %2, %4, %4, %2\n v_mfma_f32_16x16x4_f32 %3, %4, %4, %3\n"
                         "v_mfma_f32_16x16x4_f32 %0, %4, %4, %0\n v_mfma_f32_16x16x4_f32 %1, %4, %4, %1\n"
                         "v_mfma_f32_16x16x4_f32 %2, %4, %4, %2\n v_mfma_f32_16x16x4_f32 %3, %4, %4, %3\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x));
        else if (NOPS == 1)
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %4, %4, %0\n s_nop 7\n v_mfma_f32_16x16x4_f32 %1, %4, %4, %1\n s_nop 7\n"
                         "v_mfma_f32_16x16x4_f32 %2, %4, %4, %2\n s_nop 7\n v_mfma_f32_16x16x4_f32 %3, %4, %4, %3\n s_nop 7\n"
                         "v_mfma_f32_16x16x4_f32 %0, %4, %4, %0\n s_nop 7\n v_mfma_f32_16x16x4_f32 %1, %4, %4, %1\n s_nop 7\n"
                         "v_mfma_f32_16x16x4_f32 %2, %4, %4, %2\n s_nop 7\n v_mfma_f32_16x16x4_f32 %3, %4, %4, %3\n s_nop 7\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x));
        else if (NOPS == 2)
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %4, %4, %0\n s_nop 7\n s_nop 7\n v_mfma_f32_16x16x4_f32 %1, %4, %4, %1\n s_nop 7\n s_nop 7\n"
                         "v_mfma_f32_16x16x4_f32 %2, %4, %4, %2\n s_nop 7\n s_nop 7\n v_mfma_f32_16x16x4_f32 %3, %4, %4, %3\n s_nop 7\n s_nop 7\n"
                         "v_mfma_f32_16x16x4_f32 %0, %4, %4, %0\n s_nop 7\n s_nop 7\n v_mfma_f32_16x16x4_f32 %1, %4, %4, %1\n s_nop 7\n s_nop 7\n"
                         "v_mfma_f32_16x16x4_f32 %2, %4, %4, %2\n s_nop 7\n s_nop 7\n v_mfma_f32_16x16x4_f32 %3, %4, %4, %3\n s_nop 7\n s_nop 7\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x));
        else
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %4, %4, %0\n s_nop 7\n s_nop 7\n s_nop 7\n v_mfma_f32_16x16x4_f32 %1, %4, %4, %1\n s_nop 7\n s_nop 7\n s_nop 7\n"
                         "v_mfma_f32_16x16x4_f32 %2, %4, %4, %2\n s_nop 7\n s_nop 7\n s_nop 7\n v_mfma_f32_16x16x4_f32 %3, %4, %4, %3\n s_nop 7\n s_nop 7\n s_nop 7\n"
                         "v_mfma_f32_16x16x4_f32 %0, %4, %4, %0\n s_nop 7\n s_nop 7\n s_nop 7\n v_mfma_f32_16x16x4_f32 %1, %4, %4, %1\n s_nop 7\n s_nop 7\n s_nop 7\n"
                         "v_mfma_f32_16x16x4_f32 %2, %4, %4, %2\n s_nop 7\n s_nop 7\n s_nop 7\n v_mfma_f32_16x16x4_f32 %3, %4, %4, %3\n s_nop 7\n s_nop 7\n s_nop 7\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x));
    }
    return c0[0] + c1[1] + c2[2] + c3[3];
}

template <int KIND, int NOPS, int VPRIO, int MPRIO>
__global__ void __launch_bounds__(512) probe(int mode, int iters, long long* cyc, float* sink)
{
    const int wave = threadIdx.x >> 6;
    const bool is_m = wave >= 4;
    if (!is_m && VPRIO) __builtin_amdgcn_s_setprio(VPRIO);
    if (is_m && MPRIO) __builtin_amdgcn_s_setprio(MPRIO);
    __syncthreads();
    const long long t0 = clock64();
    float r = 0;
    if (is_m) { if (mode & 2) r = mfma_body<NOPS>(iters / 4, (float)threadIdx.x); }     // 8 MFMA x 32 cyc = 256 cyc per iter
    else      { if (mode & 1) r = valu_body<KIND>(iters, (float)threadIdx.x); }   // 8 VALU per iter
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    if (r == 12345.f) sink[0] = r;
}

template <int KIND, int NOPS = 0, int VPRIO = 0, int MPRIO = 0> void run(const char* name, int iters)
{
    long long* d; float* s; hipMalloc(&d, 64 * 8); hipMalloc(&s, 4);
    long long h[8];
    printf("%-28s", name);
    for (int mode = 1; mode <= 3; mode++) {
        probe<KIND, NOPS, VPRIO, MPRIO><<<1, 512>>>(mode, iters, d, s);
        probe<KIND, NOPS, VPRIO, MPRIO><<<1, 512>>>(mode, iters, d, s);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf(" | mode %d: V %8lld  M %8lld", mode, h[0], h[4]);
    }
    printf("   (per VALU instr alone: %.2f cyc; per MFMA alone: ", 0.0);
    printf(")\n");
    hipFree(d); hipFree(s);
}
int main()
{
    const int iters = 4096;  // V: 32768 VALU instr ; M: 8192 MFMA
    printf("iters=%d: V wave = %d VALU instr, M wave = %d MFMA (16x16x4 f32)\n", iters, iters * 8, iters / 4 * 8);
    run<0>("8 indep v_fma", iters);
    run<1>("8 dependent v_mul", iters);
    run<2>("8 indep v_exp", iters);
    run<3>("4 readlane + 4 mul(sgpr)", iters);
    run<4>("4 cmp + 4 cndmask (vcc)", iters);
    run<0, 1>("fma | mfma + 1 nop7", iters);
    run<0, 2>("fma | mfma + 2 nop7", iters);
    run<0, 3>("fma | mfma + 3 nop7", iters);
    run<0, 0, 3, 0>("fma prio3 | mfma prio0", iters);
    run<0, 0, 0, 3>("fma prio0 | mfma prio3", iters);
    run<0, 2, 3, 0>("fma prio3 | mfma + 2 nop7", iters);
    run<2, 0, 3, 0>("exp prio3 | mfma", iters);
    return 0;
}
