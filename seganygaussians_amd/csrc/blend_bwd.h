// blend_bwd.h -- per-tile back-to-front gradient pass.
//
// Restates renderCUDA<C> backward (CF/cuda_rasterizer/backward.cu:399-559); MASKGRAD adds the DEPTH
// variant's dL_dmask (DEPTH/cuda_rasterizer/backward.cu:457,516); C==0 is the mask-only pair
// (DEPTH/cuda_rasterizer/backward.cu:568-660).
//
// What is different from the reference, by design (all float-path, tolerance-checked):
//  * The per-channel recurrence  accum_rec[ch] = last_alpha*last_color[ch] + (1-last_alpha)*accum_rec[ch]
//    followed by  dL_dalpha += (c[ch]-accum_rec[ch])*dL_dpixel[ch]  (backward.cu:517-522) is algebraically
//    dL_dalpha = S_k - R_k with  S_k = <c_k, dL_dpixel>  and the SCALAR recurrence
//    R_k = last_alpha*S_{k+1} + (1-last_alpha)*R_{k+1}.  That removes 2C of the 3C per-pair flops and 2C live
//    registers per pixel (accum_rec, last_color).
//  * The reference issues C+6 float atomicAdds per contributing pixel-Gaussian pair (backward.cu:525-556).
//    Measured on MI355X (tools/atomic_bench.hip) the L2 executes ~10 G atomic LINE operations / s (one per
//    128-B line an instruction touches, ~20 G/s for scattered dwords) -- at ~10^9 pair atomics per view that
//    alone would be > 50 ms.  Here every gradient is reduced three times before it reaches HBM:
//      1. over the 64 pixels of a wave: blend weights w[g][pixel] of up to SLOTS contributing Gaussians are
//         parked in LDS, then the wave switches to a lane = channel layout (dL_dpixel held transposed in
//         registers) and each lane sums w*dL over the pixels (the dF = W^T * dL_dpix contraction); the 6
//         geometric fields (+mask) are quad-reduced with DPP, parked, and finished by lanes [4f, 4f+3];
//      2. over the 4 waves of the tile: LDS float atomics (ds_add_f32) into per-entry accumulator rows shared
//         by the workgroup;
//      3. one flush per batch: touched rows go to HBM as ONE 128-B atomic per (tile, Gaussian) for the C=32
//         channels plus ONE 32-B atomic into a packed per-Gaussian record {mean2D.xy, conic.xyw, opacity,
//         mask} (unpacked later by the streaming geometry-backward kernel).
//  * bg_dot_dpixel is hoisted out of the pair loop (backward.cu:533-535 recomputes it per pair).
//  * The tile only walks entries below max_pixel(n_contrib) (the reference walks from the end of the range
//    and skips, backward.cu:485-487), each wave only those below ITS quadrant's maximum, and entries are
//    culled per quadrant exactly as in the forward (cull.h); features are staged for survivors only.
#pragma once

#include "blend_fwd.h"
#include "common.h"
#include "cull.h"

namespace mirast {

constexpr int SLOTS = 4;     // contributing Gaussians parked per wave before a transposed flush
constexpr int NFIELD = 8;    // packed record: mean2D.x, mean2D.y, conic.x, conic.y, conic.w, opacity, mask, pad
constexpr int ROWS = 128;    // survivor rows (features + gradient accumulators) resident in LDS at a time

template <int C, bool MASKGRAD, bool XEXP = false>
__global__ void __launch_bounds__(256) blend_bwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ blend_list, const BlendRec* __restrict__ index_rec,
    const uint32_t* __restrict__ tile_nsurv,
    int W, int H, const float* __restrict__ bg_color, const float* __restrict__ colors,
    const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
    const float* __restrict__ dL_dout_mask, float* __restrict__ gpack /*[P,8] packed field gradients*/,
    float* __restrict__ dL_dcolors, int ablate /* timing experiments only (MI_RAST_ABLATE); 0 in production */)
{
    constexpr bool WIDE = (C >= 32);           // transposed LDS reduction for the colour channels
    constexpr bool GEO = (C > 0);              // geometric gradients exist (not the mask-only pair)
    constexpr int ROW = FeatStage<C>::ROW;
    constexpr int NH = WIDE ? (64 / C) : 1;    // lane groups over pixels in the transposed layout (C=32: 2, C=64: 1)
    constexpr int PPL = WIDE ? (64 / NH) : 1;  // pixels summed per lane in the transposed layout
    static_assert(!WIDE || C == 32 || C == 64, "wide path supports C = 32 or 64");
    constexpr int GC = C > 0 ? C : 1;

    __shared__ uint32_t s_id[ROWS];
    __shared__ float2 s_xy[ROWS];
    __shared__ float4 s_co[ROWS];
    __shared__ uint32_t s_pm[ROWS];
    __shared__ float4 s_feat4[(C > 0 ? ROWS * ROW / 4 : 1)];
    __shared__ float s_gcol[WIDE ? 1 : ROWS * GC];  // per-entry colour gradient shared by the 4 waves (narrow path)
    __shared__ float4 s_gfld4[ROWS * NFIELD / 4];  // per-entry packed field gradients
    __shared__ uint32_t s_touched[ROWS];
    __shared__ float4 s_w4[WIDE ? 4 * SLOTS * 64 / 4 : 1];
    __shared__ float4 s_f4[4 * SLOTS * NFIELD * 16 / 4];  // quad-partials of the fields
    __shared__ int s_slot_row[4 * SLOTS];
    __shared__ int s_maxc;
    float* s_feat = reinterpret_cast<float*>(s_feat4);
    float* s_gfld = reinterpret_cast<float*>(s_gfld4);
    float* s_w = reinterpret_cast<float*>(s_w4);
    float* s_f = reinterpret_cast<float*>(s_f4);

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const uint32_t horizontal_blocks = (W + TILE_X - 1) / TILE_X;
    const uint32_t tile = blockIdx.y * horizontal_blocks + blockIdx.x;
    const uint32_t qx0 = blockIdx.x * TILE_X + (wave & 1) * 8, qy0 = blockIdx.y * TILE_Y + (wave >> 1) * 8;
    const uint32_t px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const uint32_t pix_id = W * py + px;
    const float pixfx = (float)px, pixfy = (float)py;
    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    const size_t HW = (size_t)H * W;

    const uint2 range = ranges[tile];
    const int last_contributor = inside ? (int)n_contrib[pix_id] : 0;

    // entries this wave / this tile has to visit: max over pixels of n_contrib
    if (tid == 0) s_maxc = 0;
    __syncthreads();
    int wave_Lt = last_contributor;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) wave_Lt = max(wave_Lt, __shfl_xor(wave_Lt, o, 64));
    if (lane == 0) atomicMax(&s_maxc, wave_Lt);
    __syncthreads();
    const int Lt = s_maxc;
    if (Lt == 0) return;
    // blend-list records the forward walked for this tile: a superset of every contributing entry
    const int NS = (int)tile_nsurv[tile];
    const uint32_t* lst = blend_list + range.x;   // four-byte entries; list_record (binning.h) gathers the geometry record of one

    const float T_final = inside ? final_Ts[pix_id] : 0;
    float T = T_final;

    // pixel-major gradient of this lane's pixel (clamped address + select: no per-channel branches)
    const size_t pix_safe = inside ? pix_id : 0;
    float dL_dpixel[GC];
    float bg_dot_dpixel = 0;
#pragma unroll
    for (int ch = 0; ch < C; ch++) {
        const float v = dL_dpixels[ch * HW + pix_safe];
        dL_dpixel[ch] = inside ? v : 0.f;
        bg_dot_dpixel += bg_color[ch] * dL_dpixel[ch];
    }
    float dL_dout_mask_i = 0.f;
    if constexpr (MASKGRAD) {
        const float v = dL_dout_mask[pix_safe];
        dL_dout_mask_i = inside ? v : 0.f;
    }

    // channel-major (transposed) gradient: lane -> channel n, pixel group h
    float dLT[WIDE ? PPL : 1];
    const int tn = WIDE ? (lane % GC) : 0, th = WIDE ? (lane / GC) : 0;
    if constexpr (WIDE) {
#pragma unroll
        for (int p = 0; p < PPL; p++) {
            const int l = th * PPL + p;  // pixel (lane index) inside this wave's quadrant
            const uint32_t x = qx0 + (l & 7), y = qy0 + (l >> 3);
            const bool in = x < (uint32_t)W && y < (uint32_t)H;
            const float v = dL_dpixels[tn * HW + (in ? (size_t)W * y + x : 0)];
            dLT[p] = in ? v : 0.f;
        }
    }

    float last_alpha = 0.f, S_last = 0.f, Rrec = 0.f;
    const float ddelx_dx = 0.5 * W;  // backward.cu:460-461
    const float ddely_dy = 0.5 * H;
    int nslots = 0;  // wave-uniform
    float* my_w = s_w + wave * SLOTS * 64;
    float* my_f = s_f + wave * SLOTS * NFIELD * 16;
    int* my_slot_row = s_slot_row + wave * SLOTS;

    // Finish the per-wave reductions of the parked Gaussians and add them to the tile's LDS accumulators.
    // Unrolled over the slots so that the LDS reads of all parked rows are in flight together.
    auto flush = [&]() {
        // LDS rows written by this wave's own lanes: wave-local, no workgroup barrier needed
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        const int n = MI_ABLATE(2) ? 0 : __builtin_amdgcn_readfirstlane(nslots);
        if constexpr (WIDE) {
            float sum[SLOTS];
#pragma unroll
            for (int s = 0; s < SLOTS; s++) {
                sum[s] = 0.f;
                if (s < n) {
                    const float4* wrow = reinterpret_cast<const float4*>(my_w + s * 64 + th * PPL);
                    float sum0 = 0.f, sum1 = 0.f;
#pragma unroll
                    for (int p4 = 0; p4 < PPL / 4; p4++) {
                        const float4 wv = wrow[p4];
                        sum0 = fmaf(wv.x, dLT[4 * p4 + 0], sum0);
                        sum1 = fmaf(wv.y, dLT[4 * p4 + 1], sum1);
                        sum0 = fmaf(wv.z, dLT[4 * p4 + 2], sum0);
                        sum1 = fmaf(wv.w, dLT[4 * p4 + 3], sum1);
                    }
                    sum[s] = sum0 + sum1;
                }
            }
#pragma unroll
            for (int s = 0; s < SLOTS; s++) {
                if (s < n) {
                    // LDS float atomics cost ~1.5 LDS cycles per lane (measured: they were 60% of this kernel's LDS
                    // time), while a contiguous 128-B global atomic is ONE L2 line operation: send each wave's
                    // channel sums straight to HBM.
                    float v = sum[s];
                    if constexpr (NH == 2) v += __shfl_xor(v, 32, 64);
                    if (th == 0) atomicAdd(&dL_dcolors[(size_t)s_id[my_slot_row[s]] * C + tn], v);
                }
            }
        }
        // fields: lane l < 4*NQ handles field fi = l/4, partials [4*(l%4), +4)
        constexpr int NQ = GEO ? (MASKGRAD ? 7 : 6) : 1;  // quads at work (mask-only pair: field 6 only)
        if (lane < 4 * NQ) {
            const int fi = GEO ? (lane >> 2) : 6;
#pragma unroll
            for (int s = 0; s < SLOTS; s++) {
                if (s < n) {
                    const float4 v = reinterpret_cast<const float4*>(my_f + (s * NFIELD + fi) * 16)[lane & 3];
                    float sum = (v.x + v.y) + (v.z + v.w);
                    sum += dpp_mov<0xB1>(sum);  // quad_perm [1,0,3,2]
                    sum += dpp_mov<0x4E>(sum);  // quad_perm [2,3,0,1]
                    if ((lane & 3) == 0) atomicAdd(&s_gfld[my_slot_row[s] * NFIELD + fi], sum);
                }
            }
        }
        if (lane < n) s_touched[my_slot_row[lane]] = 1u;
        nslots = 0;
    };
    // quad-reduce a per-pixel value and park the 16 partials of field `fi` of the current slot
    auto park = [&](int fi, float v) {
        if MI_ABLATE(4) return;
        v += dpp_mov<0xB1>(v);
        v += dpp_mov<0x4E>(v);
        const int n = __builtin_amdgcn_readfirstlane(nslots);
        if ((lane & 3) == 0) my_f[(n * NFIELD + fi) * 16 + (lane >> 2)] = v;
    };

    BlendRec cur;
    if (tid < ROWS && tid < NS) cur = list_record(lst, index_rec, NS - 1 - tid);

    for (int b0 = 0; b0 < NS; b0 += ROWS) {
        const int nr = min(ROWS, NS - b0);  // records in this batch, walked back to front
        __syncthreads();                    // previous batch's flush to HBM done: LDS may be reused
        // ---- A: this batch's records -> LDS; the next batch's record -> registers (in flight during B and C)
        if (tid < nr) {
            s_xy[tid] = cur.xy;
            s_co[tid] = cur.co;
            s_id[tid] = cur.id;
            s_pm[tid] = cur.pm;
            s_touched[tid] = 0u;
            if constexpr (C > 0 && !WIDE) {
#pragma unroll
                for (int ch = 0; ch < C; ch++) s_feat[tid * ROW + ch] = colors[(size_t)cur.id * C + ch];
            }
        }
        if (tid < ROWS && b0 + ROWS + tid < NS) cur = list_record(lst, index_rec, NS - 1 - (b0 + ROWS + tid));
        // ---- B: features of this batch; clear the tile accumulators
        if constexpr (WIDE) {
            constexpr int F4 = C / 4;
            if (!MI_ABLATE(32))
#pragma unroll
            for (int k = 0; k < ROWS * F4 / BATCH; k++) {
                const int q = tid + BATCH * k;
                const int g = q / F4, part = q % F4;
                if (g < nr)
                    s_feat4[g * F4 + part] = reinterpret_cast<const float4*>(colors + (size_t)(lst[NS - 1 - (b0 + g)] & ID_MASK) * C)[part];
            }
        }
        if constexpr (C > 0 && !WIDE)
            for (int q = tid; q < nr * GC; q += BATCH) s_gcol[q] = 0.f;
        for (int q = tid; q < nr * NFIELD; q += BATCH) s_gfld[q] = 0.f;
        __syncthreads();

        // ---- C: gradients of this batch
        {
            float2 nxy = s_xy[0];
            float4 nco = s_co[0];
            uint32_t npm = s_pm[0];
            for (int r = 0; r < nr; r++) {
                const float2 cxy = nxy;
                const float4 cco = nco;
                const uint32_t pm = __builtin_amdgcn_readfirstlane(npm);
                const int rn = r + 1 < nr ? r + 1 : r;
                nxy = s_xy[rn];
                nco = s_co[rn];
                npm = s_pm[rn];
                const int pos = (int)(pm >> 4);  // 0-based forward position in the tile list
                if (!((pm >> wave) & 1u) || pos >= wave_Lt) continue;
                if MI_ABLATE(64) continue;
                const float dx = cxy.x - pixfx, dy = cxy.y - pixfy;
                const float power = gauss_power(-0.5f * cco.x, -cco.y, -0.5f * cco.z, dx, dy);
                const float G = gauss_exp<XEXP>(power);
                const float t = cco.w * G;
                const float alpha = fminf(0.99f, t);
                // the forward's test, on t (min(0.99, t) >= 1/255 <=> t >= 1/255; false for a NaN t like there)
                const bool valid = (pos < last_contributor) && power <= 0.0f && t >= (1.0f / 255.0f);
                if (ballot64(valid) == 0) continue;
                if MI_ABLATE(16) continue;

                const float one_m_alpha_inv = __builtin_amdgcn_rcpf(1.f - alpha);
                T = valid ? T * one_m_alpha_inv : T;
                const float w = valid ? alpha * T : 0.f;  // dchannel_dcolor

                if constexpr (!GEO) {
                    park(6, w * dL_dout_mask_i);
                } else {
                    float S0 = 0.f, S1 = 0.f;
                    if (!MI_ABLATE(8))
#pragma unroll
                    for (int ch = 0; ch < C; ch += 2) {
                        S0 = fmaf(s_feat[r * ROW + ch], dL_dpixel[ch], S0);
                        if (ch + 1 < C) S1 = fmaf(s_feat[r * ROW + ch + 1], dL_dpixel[ch + 1], S1);
                    }
                    const float S = S0 + S1;
                    Rrec = valid ? (last_alpha * S_last + (1.f - last_alpha) * Rrec) : Rrec;
                    float dL_dalpha = (S - Rrec) * T;
                    S_last = valid ? S : S_last;
                    last_alpha = valid ? alpha : last_alpha;
                    dL_dalpha += (-T_final * one_m_alpha_inv) * bg_dot_dpixel;
                    dL_dalpha = valid ? dL_dalpha : 0.f;

                    const float dL_dG = cco.w * dL_dalpha;
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * cco.x - gdy * cco.y;
                    const float dG_ddely = -gdy * cco.z - gdx * cco.y;
                    park(0, dL_dG * dG_ddelx * ddelx_dx);   // dL_dmean2D.x
                    park(1, dL_dG * dG_ddely * ddely_dy);   // dL_dmean2D.y
                    park(2, -0.5f * gdx * dx * dL_dG);      // dL_dconic.x
                    park(3, -0.5f * gdx * dy * dL_dG);      // dL_dconic.y
                    park(4, -0.5f * gdy * dy * dL_dG);      // dL_dconic.w
                    park(5, G * dL_dalpha);                 // dL_dopacity
                    if constexpr (MASKGRAD) park(6, w * dL_dout_mask_i);
                    if constexpr (WIDE) {
                        my_w[__builtin_amdgcn_readfirstlane(nslots) * 64 + lane] = w;
                    } else {
                        // narrow path (C = 3): per-channel wave reduction, one lane adds to the tile accumulator
#pragma unroll
                        for (int ch = 0; ch < C; ch++) {
                            const float gc = wave_sum(w * dL_dpixel[ch]);
                            if (lane == 0) atomicAdd(&s_gcol[r * GC + ch], gc);
                        }
                    }
                }
                if (lane == 0) my_slot_row[__builtin_amdgcn_readfirstlane(nslots)] = r;
                nslots++;
                if (nslots == SLOTS) flush();
            }
            flush();
        }
        __syncthreads();
        if MI_ABLATE(1) continue;

        // ---- D: one HBM atomic per touched (tile, Gaussian): C channels (128 B at C=32) + one packed 32-B record
        if constexpr (C > 0 && !WIDE) {
            constexpr int RPP = BATCH / GC > 0 ? BATCH / GC : 1;  // rows per pass
            const int rr = tid / GC, ch = tid % GC;
            if (rr < RPP)
                for (int r = rr; r < nr; r += RPP)
                    if (s_touched[r]) atomicAdd(&dL_dcolors[(size_t)s_id[r] * C + ch], s_gcol[r * GC + ch]);
        }
        {
            const int rr = tid / NFIELD, f = tid % NFIELD;
            constexpr int NF = GEO ? (MASKGRAD ? 7 : 6) : 7;
            for (int r = rr; r < nr; r += BATCH / NFIELD)
                if (s_touched[r] && f < NF && (GEO || f == 6))
                    atomicAdd(&gpack[(size_t)s_id[r] * NFIELD + f], s_gfld[r * NFIELD + f]);
        }
    }
}

}  // namespace mirast
