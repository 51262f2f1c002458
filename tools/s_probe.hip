// Isolated check of the S = F . dL^T MFMA phase + permlane transpose used by blend_bwd_mfma.h
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __host__ inline float Fv(int m, int ch) { return 0.01f * (float)((m * 37 + ch * 11) % 101) - 0.3f; }
__device__ __host__ inline float dLv(int p, int ch) { return 0.02f * (float)((p * 13 + ch * 7) % 53) - 0.5f; }
__global__ void k(float* out /*[16 rows][64 pix]*/)
{
    const int lane = threadIdx.x, n16 = lane & 15, kq = lane >> 4;
    float dLB[4][8];
    for (int pb = 0; pb < 4; pb++) for (int s = 0; s < 8; s++) dLB[pb][s] = dLv(16 * pb + n16, 8 * kq + s);
    float fa[8];
    for (int s = 0; s < 8; s++) fa[s] = Fv(n16, 8 * kq + s);
    v4f sacc[4];
    for (int pb = 0; pb < 4; pb++) sacc[pb] = (v4f){0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < 8; s++)
        for (int pb = 0; pb < 4; pb++) sacc[pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s], dLB[pb][s], sacc[pb], 0, 0, 0);
    float Srow[16];
    for (int r = 0; r < 4; r++) {
        const auto s01 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, sacc[0][r]), __builtin_bit_cast(unsigned, sacc[1][r]), false, false);
        const auto s23 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, sacc[2][r]), __builtin_bit_cast(unsigned, sacc[3][r]), false, false);
        const auto ax = __builtin_amdgcn_permlane32_swap(s01[0], s23[0], false, false);
        const auto ay = __builtin_amdgcn_permlane32_swap(s01[1], s23[1], false, false);
        Srow[0 + r] = __builtin_bit_cast(float, ax[0]);
        Srow[4 + r] = __builtin_bit_cast(float, ay[0]);
        Srow[8 + r] = __builtin_bit_cast(float, ax[1]);
        Srow[12 + r] = __builtin_bit_cast(float, ay[1]);
    }
    for (int m = 0; m < 16; m++) out[m * 64 + lane] = Srow[m];
    // also raw layout for diagnosis
    for (int pb = 0; pb < 4; pb++) for (int r = 0; r < 4; r++) out[1024 + (pb * 4 + r) * 64 + lane] = sacc[pb][r];
}
int main()
{
    float* d; hipMalloc(&d, 2048 * 4);
    k<<<1, 64>>>(d);
    static float h[2048]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 16; m++) for (int p = 0; p < 64; p++) {
        float ref = 0; for (int ch = 0; ch < 32; ch++) ref += Fv(m, ch) * dLv(p, ch);
        if (fabsf(ref - h[m * 64 + p]) > 1e-4f) { if (bad < 8) printf("transposed mismatch row %d pix %d got %f want %f\n", m, p, h[m * 64 + p], ref); bad++; }
    }
    printf("transposed bad: %d / 1024\n", bad);
    int bad2 = 0;
    for (int pb = 0; pb < 4; pb++) for (int r = 0; r < 4; r++) for (int l = 0; l < 64; l++) {
        int row = 4 * (l / 16) + r, pix = 16 * pb + (l % 16);
        float ref = 0; for (int ch = 0; ch < 32; ch++) ref += Fv(row, ch) * dLv(pix, ch);
        if (fabsf(ref - h[1024 + (pb * 4 + r) * 64 + l]) > 1e-4f) { if (bad2 < 8) printf("raw mismatch pb %d r %d lane %d got %f want %f\n", pb, r, l, h[1024 + (pb * 4 + r) * 64 + l], ref); bad2++; }
    }
    printf("raw bad: %d / 1024\n", bad2);
    return 0;
}
