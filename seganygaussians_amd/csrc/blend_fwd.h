// blend_fwd.h -- per-tile front-to-back alpha compositing (forward).
//
// Restates renderCUDA<C> forward (CF/cuda_rasterizer/forward.cu:264-385); with EXTRA==2 the DEPTH
// variant (DEPTH/cuda_rasterizer/forward.cu:308-309,363-365,384-385: mask + depth accumulators,
// no background term); with C==0, EXTRA==1 the mask-only render (DEPTH/.../forward.cu:390-498).
//
// gfx950 mapping: one 256-thread workgroup per 16x16 tile (the tile size is part of the integer
// contract), four wave64s each owning an 8x8 pixel quadrant.  The kernel does not walk the raw tile list:
// it streams the tile's BLEND LIST (binning.h), i.e. only the entries that can reach alpha >= 1/255
// somewhere in the tile (about one third of the list on the benchmark scene): four-byte entries id | quadrant mask,
// each completed by ONE gather of the Gaussian's 32-byte geometry record (xy / conic / opacity; binning.h: list_record).
// Per batch of FB records:
//   A. record -> LDS; the NEXT batch's record is already in flight in registers while this one is blended;
//   B. the C feature floats of every record are staged into LDS with coalesced 16-B loads (8 lanes cover one
//      Gaussian's 128 B at C=32) -- the reference re-reads features from global memory per pixel (forward.cu:359);
//   C. each wave walks the records, skipping those whose mask bit for its quadrant is clear (scalar test),
//      with the next record's geometry prefetched from LDS while the current one is blended.
// For C = 32 / 64 the accumulation  acc[pixel][ch] += w[pixel] * f[g][ch]  is a rank-1 update and runs on the
// matrix pipe: v_mfma_f32_32x32x1_2b_f32 takes A = w (lane = pixel: two 32-pixel blocks) and B = f[g][lane & 31]
// (ONE ds_read_b32 instead of eight broadcast ds_read_b128) and keeps the 64x32 accumulator tile in 32 VGPRs.
// f32 MFMA is bit-for-bit an fmaf chain, so numerics equal the VALU path; VALU and LDS are left to the alpha
// evaluation.  The accumulator tile (lane = channel) is transposed through LDS once per tile for the CHW stores.
// Early termination is per wave (ballot) and per workgroup (__syncthreads_and).
#pragma once

#include "binning.h"
#include "common.h"

namespace mirast {

constexpr int BATCH = 256;  // threads per tile workgroup
constexpr int FB = 128;     // blend-list records per forward batch

template <int CE>
struct FeatStage {
    // Feature rows padded to a multiple of 4 floats so that a row is read with ds_read_b128.
    static constexpr int ROW = (CE + 3) & ~3;
};

// PARTIAL: only the first `cr` (< C) of the block's C channels exist in memory -- the last channel block of a feature whose width
// is no multiple of 16 (the reference compiles ANY NUM_CHANNELS, config_contrastive_f.h:15): feature rows are staged channel by
// channel with zeros behind cr (a vector load would run into the next Gaussian's row, or past the end of the tensor), and only cr
// planes are written.
template <int C, int EXTRA, bool XEXP = false, bool PARTIAL = false>
__global__ void __launch_bounds__(256, (C == 32 && EXTRA == 0) ? 4 : 1) blend_fwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ blend_list, const BlendRec* __restrict__ index_rec,
    int W, int H, const float* __restrict__ features, const float* __restrict__ mask, const float* __restrict__ depths,
    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_consumed,
    uint32_t* __restrict__ tile_nsurv, const float* __restrict__ bg_color, float* __restrict__ out_color,
    float* __restrict__ out_mask, float* __restrict__ out_depth,
    int cstride /* floats between the feature rows of two Gaussians: C, or the full channel count when this launch renders one
                   channel block of a wider feature (mi_rast.hip: channel blocks; `features` then points at the block) */,
    int cr_arg /* PARTIAL: channels of this block that exist (1 .. C - 1) */,
    int ablate /* timing experiments only (MI_RAST_ABLATE_FWD); 0 in production */)
{
    const int cr = PARTIAL ? cr_arg : C;
    constexpr int CE = C + EXTRA;            // accumulated values per pixel
    constexpr int ROW = FeatStage<CE>::ROW;  // LDS floats per staged Gaussian
    constexpr bool VEC_STAGE = (EXTRA == 0) && (C % 4 == 0) && (C >= 4) && !PARTIAL;
    static_assert(!PARTIAL || (EXTRA == 0 && C == 16), "partial blocks are the 16-channel remainder of a feature");
    constexpr bool USE_MFMA = (EXTRA == 0) && (C == 32 || C == 64);
    constexpr int NACC = USE_MFMA ? C / 32 : 1;  // 32-channel accumulator tiles
    typedef float v32f __attribute__((ext_vector_type(32)));

    __shared__ float2 s_xy[FB];
    __shared__ float4 s_co[FB];
    __shared__ uint32_t s_id[FB];
    __shared__ uint32_t s_pm[FB];  // (position in the tile list) << 4 | quadrant mask
    __shared__ float4 s_feat4[FB * ROW / 4];
    __shared__ uint64_t s_bits[4][FB / 64];  // [quadrant][staging wave]: records that may touch the quadrant
    __shared__ uint8_t s_list[4][FB];        // per quadrant: indices of those records, in list order
    __shared__ int s_consumed, s_walked;
    float* s_feat = reinterpret_cast<float*>(s_feat4);

    const int tid = threadIdx.x;
    if (tid == 0) {
        s_consumed = 0;
        s_walked = 0;
    }
    const int wave = tid >> 6, lane = tid & 63;
    const uint32_t horizontal_blocks = (W + TILE_X - 1) / TILE_X;
    const uint32_t tile = blockIdx.y * horizontal_blocks + blockIdx.x;
    // 8x8 quadrant per wave
    const uint32_t px = blockIdx.x * TILE_X + (wave & 1) * 8 + (lane & 7);
    const uint32_t py = blockIdx.y * TILE_Y + (wave >> 1) * 8 + (lane >> 3);
    const uint32_t pix_id = W * py + px;
    float pixfx = (float)px, pixfy = (float)py;
    asm volatile("" : "+v"(pixfx), "+v"(pixfy));  // opaque: hipcc otherwise re-converts both in every iteration of the blend loop
    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    bool done = !inside;

    const uint2 range = ranges[tile];
    const int list_len = (int)(range.y - range.x);
    const int ns_total = list_len;
    const uint32_t* lst = blend_list + range.x;   // four-byte entries; list_record (binning.h) gathers the geometry record of one

    float T = 1.0f;
    uint32_t last_contributor = 0;
    // wave-uniform counters: raw list entries this wave walked (counter E of SURVEY.md 8d: a wave that never
    // finishes walks the whole list) and blend-list records it walked (the backward starts from there)
    int consumed = (ballot64(!done) != 0) ? list_len : 0;
    int walked = 0;
    float acc[(CE > 0 && !USE_MFMA) ? CE : 1];
    v32f accv[NACC];
    if constexpr (USE_MFMA) {
#pragma unroll
        for (int a = 0; a < NACC; a++)
#pragma unroll
            for (int r = 0; r < 32; r++) accv[a][r] = 0.f;
    } else {
#pragma unroll
        for (int ch = 0; ch < CE; ch++) acc[ch] = 0.f;
    }

    BlendRec cur;
    if (tid < FB && tid < ns_total) cur = list_record(lst, index_rec, tid);

    for (int b0 = 0; b0 < ns_total; b0 += FB) {
        // whole workgroup finished? (also the barrier that protects LDS reuse)
        if (__syncthreads_and(done)) break;
        const int nb = min(FB, ns_total - b0);

        // ---- A: this batch's records -> LDS; next batch's records -> registers (in flight during B and C)
        if (tid < nb) {
            s_xy[tid] = cur.xy;
            // conic pre-scaled to (-a/2, -b, -c/2) for gauss_power (common.h)
            s_co[tid] = make_float4(-0.5f * cur.co.x, -cur.co.y, -0.5f * cur.co.z, cur.co.w);
            s_id[tid] = cur.id;
            s_pm[tid] = cur.pm;
            if constexpr (!VEC_STAGE) {
#pragma unroll
                for (int ch = 0; ch < C; ch++)
                    s_feat[tid * ROW + ch] = (!PARTIAL || ch < cr) ? features[(size_t)cur.id * cstride + ch] : 0.f;
                if constexpr (EXTRA >= 1) s_feat[tid * ROW + C] = mask[cur.id];
                if constexpr (EXTRA >= 2) s_feat[tid * ROW + C + 1] = depths[cur.id];
            }
        }
        if (wave < FB / 64) {
            // per-quadrant record bitmaps: wave q later walks only the set bits (scalar ctz loop)
            const uint32_t pmv = tid < nb ? cur.pm : 0u;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint64_t b = ballot64((pmv >> q) & 1u);
                if (lane == 0) s_bits[q][wave] = b;
            }
        }
        if (tid < FB && b0 + FB + tid < ns_total) cur = list_record(lst, index_rec, b0 + FB + tid);
        if constexpr (VEC_STAGE) {
            // ---- B: features, gathered by the ids just staged in LDS (a second trip to the records in memory
            // would put one more ~4 us dependent access in front of every batch)
            __syncthreads();
            constexpr int F4 = C / 4;  // float4s per Gaussian
            // All the loads are issued before the first LDS write (unconditionally: a clamped row index keeps the
            // address valid) -- written as one guarded load-then-store per k, hipcc emits a full vmcnt(0) round trip per k.
            if (!MI_ABLATE(4)) {
                constexpr int NK = FB * F4 / BATCH;
                float4 v[NK];
#pragma unroll
                for (int k = 0; k < NK; k++) {
                    const int q = tid + BATCH * k;
                    const int g = q / F4, part = q % F4;
                    v[k] = reinterpret_cast<const float4*>(features + (size_t)s_id[g < nb ? g : 0] * cstride)[part];
                }
                // pins every loaded value in registers here: hipcc otherwise sinks each load into the guarded store below
                #pragma unroll
                for (int k = 0; k < NK; k++) asm volatile("" : "+v"(v[k].x), "+v"(v[k].y), "+v"(v[k].z), "+v"(v[k].w));
#pragma unroll
                for (int k = 0; k < NK; k++) {
                    const int q = tid + BATCH * k;
                    const int g = q / F4, part = q % F4;
                    if (g < nb) s_feat4[g * F4 + part] = v[k];
                }
            }
        }
        __syncthreads();

        // ---- C: blend.  The wave walks the set bits of its quadrant's bitmap; the body is branch-free around the
        // accumulate so that the 32-register MFMA accumulator tile stays pinned (no phi copies).
        if (ballot64(!done) != 0) {
            walked = b0 + nb;
            // compact this quadrant's record indices (wave-local: bitmap -> index list via mbcnt)
            int cnt = 0;
#pragma unroll
            for (int h = 0; h < FB / 64; h++) {
                const uint64_t b = s_bits[wave][h];
                const bool set = (b >> lane) & 1ull;
                const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
                if (set) s_list[wave][cnt + below] = (uint8_t)(64 * h + lane);
                cnt += __builtin_popcountll(b);
            }
            cnt = __builtin_amdgcn_readfirstlane(cnt);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            int j = 0;
            int k = cnt > 0 ? (int)__builtin_amdgcn_readfirstlane((uint32_t)s_list[wave][0]) : -1;
            float2 nxy = s_xy[k < 0 ? 0 : k];
            float4 nco = s_co[k < 0 ? 0 : k];
            uint32_t npm = s_pm[k < 0 ? 0 : k];
            bool finished = false;
            uint64_t live = ballot64(!done);  // lanes still blending (wave-uniform copy of !done: the loop exit needs no ballot)
            uint32_t last_pm = 0;
            int last_k = 0;
            while (k >= 0 && !finished) {
                const float2 cxy = nxy;
                const float4 cco = nco;
                const uint32_t pm = __builtin_amdgcn_readfirstlane(npm);
                j++;
                const int kn = j < cnt ? (int)__builtin_amdgcn_readfirstlane((uint32_t)s_list[wave][j]) : -1;
                const int kp = kn < 0 ? k : kn;  // prefetch the next record of this quadrant
                nxy = s_xy[kp];
                nco = s_co[kp];
                npm = s_pm[kp];
                const float dx = cxy.x - pixfx, dy = cxy.y - pixfy;
                const float power = gauss_power(cco.x, cco.y, cco.z, dx, dy);
                const float t = cco.w * gauss_exp<XEXP>(power);
                const float alpha = fminf(0.99f, t);
                const bool ok = !done && power <= 0.0f && t >= (1.0f / 255.0f);  // min(0.99, t) >= 1/255  <=>  t >= 1/255
                const float test_T = T * (1 - alpha);
                // forward.cu:358-362: the pixel is done once a contributor would push T below 1e-4 (that one is not blended)
                const bool stop = (ok ? test_T : 1.0f) < 0.0001f;
                live &= ~ballot64(stop);
                done = done || stop;
                const bool blend = ok && !stop;
                const float w = blend ? alpha * T : 0.f;
                if constexpr (USE_MFMA) {
#pragma unroll
                    for (int a = 0; a < NACC; a++) {
                        // Inline asm with a tied "+v" accumulator: with the builtin hipcc shuttles all 32 accumulator
                        // registers VGPR<->AGPR (or v_mov's them) around every MFMA that sits behind a branch.
                        // Hazards handled here (cdna_hip_programming.md 5.7): s_nop 1 covers a just-written VGPR
                        // operand -> MFMA; an accumulate chain on the same registers needs 0 states; the epilogue
                        // waits before its VALU reads the tile.  Lanes that do not blend contribute w = 0.
                        const float fb = s_feat[k * ROW + 32 * a + (lane & 31)];
                        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x1_2b_f32 %0, %1, %2, %0" : "+v"(accv[a]) : "v"(w), "v"(fb));
                    }
                } else {
                    if (ballot64(blend) != 0) {
#pragma unroll
                        for (int ch = 0; ch < CE; ch++) acc[ch] = fmaf(s_feat[k * ROW + ch], w, acc[ch]);
                    }
                }
                T = blend ? test_T : T;
                last_contributor = blend ? (pm >> 4) + 1u : last_contributor;
                finished = live == 0;  // this wave is finished
                last_pm = pm;
                last_k = k;
                k = kn;
            }
            if (finished) {
                consumed = (int)(last_pm >> 4) + 1;
                walked = b0 + last_k + 1;
            }
        }
    }

    if (lane == 0) {
        atomicMax(&s_consumed, consumed);
        atomicMax(&s_walked, walked);
    }
    __syncthreads();
    if (tid == 0) {
        tile_consumed[tile] = (uint32_t)s_consumed;
        tile_nsurv[tile] = (uint32_t)s_walked;
    }
    const size_t HW = (size_t)H * W;
    if (inside) {
        final_T[pix_id] = T;
        n_contrib[pix_id] = last_contributor;
    }
    if constexpr (USE_MFMA) {
        // accumulator tile: lane l holds channel (l & 31) of 32 pixels: register r <-> pixel lane index
        // L(r, l) = 32*(r>>4) + (r&3) + 8*((r&15)>>2) + 4*(l>>5)   (probed on gfx950, tools/mfma_probe.hip).
        // Transpose through LDS, 8 channels at a time per wave ([8][65] floats, reusing the feature buffer
        // -- every wave is past the last barrier-protected use of it), then store pixel-major.
        static_assert(FB * ROW >= 4 * 8 * 65, "feature buffer too small for the epilogue transpose");
        float* tp = s_feat + wave * 8 * 65;
        // 16-pass MFMA result -> VALU reader needs 18 wait states (the barrier above makes this moot in
        // practice; kept so correctness never depends on timing)
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
        for (int a = 0; a < NACC; a++) {
#pragma unroll
            for (int part = 0; part < 4; part++) {
                const int chl = lane & 31;  // channel within the 32-tile
                if ((chl >> 3) == part) {
#pragma unroll
                    for (int r = 0; r < 32; r++) {
                        const int L = 32 * (r >> 4) + (r & 3) + 8 * ((r & 15) >> 2) + 4 * (lane >> 5);
                        tp[(chl & 7) * 65 + L] = accv[a][r];
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                if (inside) {
#pragma unroll
                    for (int c = 0; c < 8; c++) {
                        const int ch = 32 * a + 8 * part + c;
                        out_color[ch * HW + pix_id] = tp[c * 65 + lane] + T * bg_color[ch];
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            }
        }
    } else if (inside) {
#pragma unroll
        for (int ch = 0; ch < C; ch++)
            if (!PARTIAL || ch < cr) out_color[ch * HW + pix_id] = acc[ch] + T * bg_color[ch];
        if constexpr (EXTRA >= 1) out_mask[pix_id] = acc[C];
        if constexpr (EXTRA >= 2) out_depth[pix_id] = acc[C + 1];
    }
}

}  // namespace mirast
