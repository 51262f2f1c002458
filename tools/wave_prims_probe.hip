// Checks the wave-level primitives of csrc/binning.h on the GPU against serial restatements: the seven-DPP inclusive scan (sum / max),
// the scatter + max-scan owner search of the count / emit passes, and the lane exchanges of the wave-per-tile bitonic sorter.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I seganygaussians_amd/csrc tools/wave_prims_probe.hip -o tools/wave_prims_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "binning.h"
using namespace mirast;

__global__ void probe(const uint32_t* in, uint32_t* out_sum, uint32_t* out_max, uint32_t* out_xor, uint32_t* out_owner, int nbase)
{
    __shared__ uint32_t s_scr[64];
    const int lane = threadIdx.x;
    const uint32_t h = in[blockIdx.x * 64 + lane];
    const uint32_t incl = wave_inclusive_scan_dpp(h);
    out_sum[blockIdx.x * 64 + lane] = incl;
    out_max[blockIdx.x * 64 + lane] = wave_inclusive_scan_dpp<true>(h * 7u % 13u);
    uint32_t* x = out_xor + (size_t)blockIdx.x * 64 * 12 + lane * 12;
    const uint32_t v = (uint32_t)lane * 3u + 1u;
    x[0] = lane_xor<1>(v); x[1] = lane_xor<2>(v); x[2] = lane_xor<3>(v); x[3] = lane_xor<4>(v); x[4] = lane_xor<7>(v); x[5] = lane_xor<8>(v);
    x[6] = lane_xor<15>(v); x[7] = lane_xor<16>(v); x[8] = lane_xor<31>(v); x[9] = lane_xor<32>(v); x[10] = lane_xor<63>(v); x[11] = v;
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    for (int b = 0; b < nbase; b++) {
        const uint32_t base = 64u * (uint32_t)b;
        uint32_t o = 0xFFFFFFFFu;
        if (base < total) o = wave_owner(s_scr, incl - h, incl, base, lane);   // wave-uniform condition
        out_owner[((size_t)blockIdx.x * nbase + b) * 64 + lane] = o;
    }
}

int main()
{
    const int NB = 512, NBASE = 8;
    std::vector<uint32_t> in(NB * 64);
    srand(1);
    for (int b = 0; b < NB; b++)
        for (int l = 0; l < 64; l++) {
            uint32_t h = (rand() % 100 < 55) ? rand() % 5 : 0;
            if (b % 7 == 3 && l == (b * 11) % 64) h = 150;
            if (b % 5 == 0) h = rand() % 2;
            in[b * 64 + l] = h;
        }
    uint32_t *d_in, *d_sum, *d_max, *d_xor, *d_own;
    hipMalloc(&d_in, in.size() * 4); hipMalloc(&d_sum, in.size() * 4); hipMalloc(&d_max, in.size() * 4);
    hipMalloc(&d_xor, in.size() * 12 * 4); hipMalloc(&d_own, (size_t)NB * NBASE * 64 * 4);
    hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(NB), dim3(64), 0, 0, d_in, d_sum, d_max, d_xor, d_own, NBASE);
    std::vector<uint32_t> sum(in.size()), mx(in.size()), xr(in.size() * 12), own((size_t)NB * NBASE * 64);
    hipMemcpy(sum.data(), d_sum, sum.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(mx.data(), d_max, mx.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(xr.data(), d_xor, xr.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(own.data(), d_own, own.size() * 4, hipMemcpyDeviceToHost);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    long bad_sum = 0, bad_max = 0, bad_xor = 0, bad_own = 0;
    const int M[11] = {1, 2, 3, 4, 7, 8, 15, 16, 31, 32, 63};
    for (int b = 0; b < NB; b++) {
        uint32_t run = 0, rm = 0, incl[64];
        for (int l = 0; l < 64; l++) {
            run += in[b * 64 + l]; incl[l] = run;
            rm = std::max(rm, in[b * 64 + l] * 7u % 13u);
            bad_sum += sum[b * 64 + l] != run; bad_max += mx[b * 64 + l] != rm;
            for (int m = 0; m < 11; m++) bad_xor += xr[((size_t)b * 64 + l) * 12 + m] != (uint32_t)(l ^ M[m]) * 3u + 1u;
        }
        for (int bb = 0; bb < NBASE; bb++)
            for (int l = 0; l < 64; l++) {
                const uint32_t k = 64u * bb + l, got = own[((size_t)b * NBASE + bb) * 64 + l];
                if (k >= run) continue;
                int want = 0; while (incl[want] <= k) want++;
                if (got != (uint32_t)want) { if (bad_own < 5) printf("owner: block %d base %d lane %d got %u want %d\n", b, bb, l, got, want); bad_own++; }
            }
    }
    for (int m = 0; m < 11 && bad_xor; m++) { long c = 0; for (size_t i = 0; i < in.size(); i++) c += xr[i * 12 + m] != (uint32_t)((i % 64) ^ M[m]) * 3u + 1u; if (c) printf("lane_xor<%d>: %ld wrong\n", M[m], c); }
    printf("wave primitives: scan(sum) wrong %ld, scan(max) wrong %ld, lane_xor wrong %ld, owner wrong %ld  -> %s\n", bad_sum, bad_max, bad_xor, bad_own,
           (bad_sum | bad_max | bad_xor | bad_own) ? "FAIL" : "ok");
    return (bad_sum | bad_max | bad_xor | bad_own) ? 1 : 0;
}
