// LDS float-atomic throughput on gfx950: ds_add_f32 (no return) vs ds_write_b32, conflict-free addresses, 1..12 waves per CU.
// Question behind it: can the four quadrant waves of a tile sum their gradient rows in an LDS table
// (640 floats per 16-row chunk and wave) instead of issuing global atomics per (quadrant, record) row?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void probe(long long* out, float* sink, int iters)
{
    __shared__ float tab[12 * 1024];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 12 * 1024; i += blockDim.x) tab[i] = 0.f;
    __syncthreads();
    float* base = tab + w * 1024 + l;
    const float v = (float)l;
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 10; k++) {
            if (MODE == 0) atomicAdd(base + 64 * k, v);                       // ds_add_f32
            else if (MODE == 1) base[64 * k] = v + (float)it;                  // ds_write_b32
            else { const float o = base[64 * k]; base[64 * k] = o + v; }       // read + add + write
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = tab[threadIdx.x];
}
int main()
{
    long long* out; float* sink;
    hipMalloc(&out, 1024 * 8); hipMalloc(&sink, 1024 * 1024 * 4);
    const int iters = 2000;
    for (int waves : {1, 4, 8, 12}) {
        long long h[3];
        for (int m = 0; m < 3; m++) {
            if (m == 0) probe<0><<<256, 64 * waves>>>(out, sink, iters);
            if (m == 1) probe<1><<<256, 64 * waves>>>(out, sink, iters);
            if (m == 2) probe<2><<<256, 64 * waves>>>(out, sink, iters);
            hipDeviceSynchronize();
            hipMemcpy(&h[m], out, 8, hipMemcpyDeviceToHost);
        }
        const double n = (double)iters * 10 * waves;  // wave-instructions per CU
        printf("%2d waves/CU: clocks per wave-instruction (per CU): ds_add_f32 %.1f, ds_write_b32 %.1f, read+add+write %.1f   (clock64 ticks)\n",
               waves, h[0] / n, h[1] / n, h[2] / n);
    }
    return 0;
}
