// Lane/register layout of v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 blocks, K = 1) on gfx950.
// Hypothesis (blend_bwd_wave.h, separable moments): D[block][i][j] = A[block][i] * B[block][j] with A, B at lane 4 block + i / j
// and D[block][i][j] in register i of lane 4 block + j.  One-hot A and B: prints where the single 1 lands.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void probe(int la, int lb, float* out, long long* cyc)
{
    const int l = threadIdx.x;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(l == la ? 1.f : 0.f, l == lb ? 1.f : 0.f, acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) out[4 * l + r] = acc[r];
    // issue rate: 64 dependent-free MFMAs on 4 accumulators
    v4f a0 = acc, a1 = acc, a2 = acc, a3 = acc;
    const long long t0 = clock64();
#pragma unroll
    for (int k = 0; k < 64; k++) {
        a0 = __builtin_amdgcn_mfma_f32_4x4x1f32((float)l, 1.f, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_4x4x1f32((float)l, 2.f, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_4x4x1f32((float)l, 3.f, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_4x4x1f32((float)l, 4.f, a3, 0, 0, 0);
    }
    const long long t1 = clock64();
    if (l == 0) cyc[0] = t1 - t0;
    out[256 + l] = a0[0] + a1[1] + a2[2] + a3[3];
}
int main()
{
    float* out; long long* cyc;
    hipMalloc(&out, 512 * sizeof(float)); hipMalloc(&cyc, 8);
    const int cases[][2] = {{0, 0}, {1, 0}, {0, 2}, {5, 6}, {5, 9}, {62, 61}, {35, 32}};
    for (auto& c : cases) {
        probe<<<1, 64>>>(c[0], c[1], out, cyc);
        float h[256]; long long hc;
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        printf("A one-hot at lane %d, B one-hot at lane %d ->", c[0], c[1]);
        int n = 0;
        for (int i = 0; i < 256; i++) if (h[i] != 0.f) { printf(" lane %d reg %d = %g", i / 4, i % 4, h[i]); n++; }
        printf("%s   [256 MFMAs: %lld clocks (s_memtime units)]\n", n ? "" : " nothing", hc);
    }
    return 0;
}
