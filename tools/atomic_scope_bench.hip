// Micro-benchmark: float atomics executed in the XCD's own L2 (workgroup scope: no sc1 bit) against device-scope atomics
// (sc1: performed memory-side, behind the fabric), in the backward blend's instruction shapes -- and a correctness check of
// the L2-local form when every address is only ever touched from ONE XCD (the 32-byte pattern interleaves the XCDs' records
// inside 128-byte lines: byte-masked write-back).  Also reports whether workgroup b runs on XCD b % 8.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_scope_bench.hip -o /tmp/atomic_scope_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ inline uint32_t xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u; }  // HW_REG_XCC_ID[3:0]

template <bool LOCAL>
__device__ inline void add(float* p, float v)
{
    if (LOCAL) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// PATTERN 0: 64 lanes -> 2 rows x 128 B (dF of a 32-channel row pair); rows of XCD x live in [x rows/8, (x+1) rows/8)
// PATTERN 1: 64 lanes -> 8 records x 32 B; record r belongs to XCD r % 8 (four XCDs share every 128-B line)
// PATTERN 2: PATTERN 0 + PATTERN 1 per iteration (3 segments per row, like the kernel: here 2 rows + 8 records)
template <bool LOCAL, int PATTERN>
__global__ void k(float* buf, float* rec, uint32_t rows, int iters, uint32_t* wrong_xcc)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t x = xcc_id();
    if (threadIdx.x == 0 && x != (blockIdx.x & 7u)) atomicAdd(wrong_xcc, 1u);
    const uint32_t per = rows / 8;
    for (int it = 0; it < iters; it++) {
        const uint32_t h = hash32(wave * 7919u + it * 104729u);
        if (PATTERN == 0 || PATTERN == 2) {
            const uint32_t r2 = x * per + hash32(h * 2 + (lane >> 5)) % per;
            add<LOCAL>(buf + (size_t)r2 * 32 + (lane & 31), 1.0f);
        }
        if (PATTERN == 1 || PATTERN == 2) {
            const uint32_t r8 = (hash32(h * 8 + (lane >> 3)) % per) * 8 + x;
            add<LOCAL>(rec + (size_t)r8 * 8 + (lane & 7), 1.0f);
        }
    }
}

template <bool LOCAL, int PATTERN>
void run(const char* name, float* buf, float* rec, uint32_t rows, uint32_t* wrong)
{
    const int blocks = 8192, threads = 64, iters = 256;
    const size_t n = (size_t)rows * 32;
    hipMemset(buf, 0, n * 4); hipMemset(rec, 0, (size_t)rows * 8 * 4); hipMemset(wrong, 0, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    k<LOCAL, PATTERN><<<blocks, threads>>>(buf, rec, rows, iters, wrong);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<float> h(n), hr((size_t)rows * 8);
    hipMemcpy(h.data(), buf, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hr.data(), rec, (size_t)rows * 8 * 4, hipMemcpyDeviceToHost);
    uint32_t w; hipMemcpy(&w, wrong, 4, hipMemcpyDeviceToHost);
    double s = 0, sr = 0;
    for (float v : h) s += v;
    for (float v : hr) sr += v;
    const double instr = (double)blocks * iters;
    const double want = (PATTERN != 1 ? instr * 64 : 0), want_r = (PATTERN != 0 ? instr * 64 : 0);
    const double segs = instr * ((PATTERN != 1 ? 4 : 0) + (PATTERN != 0 ? 4 : 0));   // 64-byte segments touched
    printf("%-58s %8.3f ms  %7.2f G segment-atomics/s  sums %s (%.0f/%.0f, %.0f/%.0f)  blocks off their XCD: %u\n", name, ms,
           segs / ms / 1e6, (s == want && sr == want_r) ? "EXACT" : "WRONG", s, want, sr, want_r, w);
}

int main()
{
    const uint32_t rows = 1u << 20;
    float *buf, *rec; uint32_t* wrong;
    hipMalloc(&buf, (size_t)rows * 32 * 4); hipMalloc(&rec, (size_t)rows * 8 * 4); hipMalloc(&wrong, 4);
    for (int rep = 0; rep < 2; rep++) {
        run<false, 0>("device scope: 2 rows x 128 B", buf, rec, rows, wrong);
        run<true, 0>("XCD-local    : 2 rows x 128 B", buf, rec, rows, wrong);
        run<false, 1>("device scope: 8 records x 32 B (lines shared by 4 XCDs)", buf, rec, rows, wrong);
        run<true, 1>("XCD-local    : 8 records x 32 B (lines shared by 4 XCDs)", buf, rec, rows, wrong);
        run<false, 2>("device scope: both", buf, rec, rows, wrong);
        run<true, 2>("XCD-local    : both", buf, rec, rows, wrong);
    }
    return 0;
}
