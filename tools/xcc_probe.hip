// Which XCD does workgroup b of a 1-D grid run on?  (HW_REG_XCC_ID, gfx940+)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int* out)
{
    if (threadIdx.x == 0) {
        unsigned v;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
        out[blockIdx.x] = (int)(v & 0xF);
    }
}
int main()
{
    const int n = 600;
    int* d; hipMalloc(&d, n * 4);
    for (int threads : {256, 1024}) {
        probe<<<n, threads>>>(d);
        int h[n]; hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
        printf("threads=%d: first 40 blocks -> xcc:", threads);
        for (int i = 0; i < 40; i++) printf(" %d", h[i]);
        int ok = 0; for (int i = 0; i < n; i++) ok += (h[i] == i % 8);
        printf("\n  blocks with xcc == b %% 8: %d of %d\n", ok, n);
    }
    return 0;
}
