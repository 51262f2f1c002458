// Probe the lane/register layouts of the f32 MFMA forms and permlane swaps on gfx950 (prints maps).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v32f __attribute__((ext_vector_type(32)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

// out[kind][lane][reg]
__global__ void probe(float* out)
{
    const int l = threadIdx.x;
    // 32x32x1 (2 blocks): A value encodes (lane), B encodes lane -> D = a*b ; use a = 1000 + l (as m id), b = 1 + l*0.001? use two runs
    {   // run 1: b = 1 -> D = a[m,block]  => tells which A lane feeds each D element
        v32f c = {0};
        c = __builtin_amdgcn_mfma_f32_32x32x1f32((float)l, 1.0f, c, 0, 0, 0);
        for (int r = 0; r < 32; r++) out[(0 * 64 + l) * 32 + r] = c[r];
        v32f d = {0};
        d = __builtin_amdgcn_mfma_f32_32x32x1f32(1.0f, (float)l, d, 0, 0, 0);
        for (int r = 0; r < 32; r++) out[(1 * 64 + l) * 32 + r] = d[r];
    }
    {   // 32x32x2: K=2: D[m][n] = sum_k A[m][k] B[k][n]. run a: A = lane id, B = (k==0) -> D = A lane with k=0
        v16f c = {0};
        c = __builtin_amdgcn_mfma_f32_32x32x2f32((float)l, (l < 32) ? 1.0f : 0.0f, c, 0, 0, 0);   // guess k = l/32
        for (int r = 0; r < 16; r++) out[(2 * 64 + l) * 32 + r] = c[r];
        v16f d = {0};
        d = __builtin_amdgcn_mfma_f32_32x32x2f32((l < 32) ? 1.0f : 0.0f, (float)l, d, 0, 0, 0);
        for (int r = 0; r < 16; r++) out[(3 * 64 + l) * 32 + r] = d[r];
        v16f e = {0};
        e = __builtin_amdgcn_mfma_f32_32x32x2f32((float)l, (l >= 32) ? 1.0f : 0.0f, e, 0, 0, 0);
        for (int r = 0; r < 16; r++) out[(4 * 64 + l) * 32 + r] = e[r];
    }
    {   // 16x16x4: guess m = l%16, k = l/16
        v4f c = {0};
        c = __builtin_amdgcn_mfma_f32_16x16x4f32((float)l, (l / 16 == 1) ? 1.0f : 0.0f, c, 0, 0, 0);
        for (int r = 0; r < 4; r++) out[(5 * 64 + l) * 32 + r] = c[r];
        v4f d = {0};
        d = __builtin_amdgcn_mfma_f32_16x16x4f32((l / 16 == 2) ? 1.0f : 0.0f, (float)l, d, 0, 0, 0);
        for (int r = 0; r < 4; r++) out[(6 * 64 + l) * 32 + r] = d[r];
    }
    {   // permlane32_swap(a = l, b = 100 + l)
        auto r = __builtin_amdgcn_permlane32_swap((unsigned)l, (unsigned)(100 + l), false, false);
        out[(7 * 64 + l) * 32 + 0] = (float)r[0];
        out[(7 * 64 + l) * 32 + 1] = (float)r[1];
        auto q = __builtin_amdgcn_permlane16_swap((unsigned)l, (unsigned)(100 + l), false, false);
        out[(7 * 64 + l) * 32 + 2] = (float)q[0];
        out[(7 * 64 + l) * 32 + 3] = (float)q[1];
    }
}
int main()
{
    float* d; hipMalloc(&d, 8 * 64 * 32 * 4); hipMemset(d, 0, 8 * 64 * 32 * 4);
    probe<<<1, 64>>>(d);
    static float h[8][64][32];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"32x32x1 D<-A lane (b=1)", "32x32x1 D<-B lane (a=1)", "32x32x2 D<-A lane, B picks k=l/32==0", "32x32x2 D<-B lane, A picks k==0", "32x32x2 D<-A lane, B picks k==1", "16x16x4 D<-A lane, B picks k=l/16==1", "16x16x4 D<-B lane, A picks k==2"};
    int regs[] = {32, 32, 16, 16, 16, 4, 4};
    for (int k = 0; k < 7; k++) {
        printf("== %s\n", names[k]);
        for (int l : {0, 1, 5, 31, 32, 33, 63}) { printf(" lane %2d:", l); for (int r = 0; r < regs[k]; r++) printf(" %3.0f", h[k][l][r]); printf("\n"); }
    }
    printf("== permlane32_swap(a=l, b=100+l) -> r0 r1 ; permlane16_swap -> q0 q1\n");
    for (int l : {0, 1, 15, 16, 17, 31, 32, 33, 47, 48, 63}) printf(" lane %2d: %3.0f %3.0f | %3.0f %3.0f\n", l, h[7][l][0], h[7][l][1], h[7][l][2], h[7][l][3]);
    return 0;
}
