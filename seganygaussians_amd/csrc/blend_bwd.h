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
//    Here every gradient field is first reduced over the wave64's 64 pixels and only then added to HBM:
//      - the 6 geometric fields with a DPP wave reduction (common.h wave_sum), lanes 0..5 issue ONE
//        global_atomic_add_f32 instruction together;
//      - the C colour/feature channels with a transposed accumulation: blend weights w[g][pixel] of up to
//        SLOTS contributing Gaussians are parked in LDS, then the wave switches to a lane = channel layout
//        (dL_dpixel held transposed in registers) and each lane sums w*dL over the pixels; the wave then
//        issues ONE coalesced 128-B atomic per Gaussian (C=32).  This is the dF = W^T * dL_dpix contraction.
//    => (C+6)*64 atomics per (wave, Gaussian) become 2 atomic instructions.
//  * bg_dot_dpixel is hoisted out of the pair loop (backward.cu:533-535 recomputes it per pair).
//  * The tile only walks entries below max_pixel(n_contrib) (the reference walks from the end of the range
//    and skips, backward.cu:485-487).
#pragma once

#include "blend_fwd.h"
#include "common.h"

namespace mirast {

constexpr int SLOTS = 8;  // contributing Gaussians parked per wave before a transposed flush

template <int C, bool MASKGRAD>
__global__ void __launch_bounds__(256) blend_bwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H,
    const float* __restrict__ bg_color, const float2* __restrict__ points_xy_image,
    const float4* __restrict__ conic_opacity, const float* __restrict__ colors, const float* __restrict__ final_Ts,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
    const float* __restrict__ dL_dout_mask, float* __restrict__ dL_dmean2D /*[P,3]*/,
    float* __restrict__ dL_dconic2D /*[P,4]*/, float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolors,
    float* __restrict__ dL_dmask)
{
    constexpr bool WIDE = (C >= 32);           // transposed LDS flush for the colour channels
    constexpr int ROW = FeatStage<C>::ROW;
    constexpr int NH = WIDE ? (64 / C) : 1;    // lane groups over pixels in the transposed layout (C=32: 2, C=64: 1)
    constexpr int PPL = WIDE ? (64 / NH) : 1;  // pixels summed per lane in the transposed layout
    static_assert(!WIDE || C == 32 || C == 64, "wide path supports C = 32 or 64");

    __shared__ int s_id[BATCH];
    __shared__ float2 s_xy[BATCH];
    __shared__ float4 s_co[BATCH];
    __shared__ float4 s_feat4[(C > 0 ? BATCH * ROW / 4 : 1)];
    __shared__ float4 s_w4[WIDE ? 4 * SLOTS * 64 / 4 : 1];
    __shared__ int s_slot_id[WIDE ? 4 * SLOTS : 1];
    __shared__ int s_maxc;
    float* s_feat = reinterpret_cast<float*>(s_feat4);
    float* s_w = reinterpret_cast<float*>(s_w4);

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

    // entries this tile has to visit: max over its pixels of n_contrib
    if (tid == 0) s_maxc = 0;
    __syncthreads();
    {
        int m = last_contributor;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
        if (lane == 0) atomicMax(&s_maxc, m);
    }
    __syncthreads();
    const int Lt = s_maxc;
    if (Lt == 0) return;
    const int rounds = (Lt + BATCH - 1) / BATCH;

    const float T_final = inside ? final_Ts[pix_id] : 0;
    float T = T_final;

    // pixel-major gradient of this lane's pixel
    float dL_dpixel[C > 0 ? C : 1];
    float bg_dot_dpixel = 0;
#pragma unroll
    for (int ch = 0; ch < C; ch++) {
        dL_dpixel[ch] = inside ? dL_dpixels[ch * HW + pix_id] : 0.f;
        bg_dot_dpixel += bg_color[ch] * dL_dpixel[ch];
    }
    const float dL_dout_mask_i = (MASKGRAD && inside) ? dL_dout_mask[pix_id] : 0.f;

    // channel-major (transposed) gradient: lane -> channel n, pixel group h
    float dLT[WIDE ? PPL : 1];
    const int tn = WIDE ? (lane % C) : 0, th = WIDE ? (lane / C) : 0;
    if constexpr (WIDE) {
#pragma unroll
        for (int p = 0; p < PPL; p++) {
            const int l = th * PPL + p;  // pixel (lane index) inside this wave's quadrant
            const uint32_t x = qx0 + (l & 7), y = qy0 + (l >> 3);
            dLT[p] = (x < (uint32_t)W && y < (uint32_t)H) ? dL_dpixels[tn * HW + (size_t)W * y + x] : 0.f;
        }
    }

    float last_alpha = 0.f, S_last = 0.f, Rrec = 0.f;
    const float ddelx_dx = 0.5 * W;  // backward.cu:460-461
    const float ddely_dy = 0.5 * H;
    int nslots = 0;  // wave-uniform
    float* my_w = s_w + wave * SLOTS * 64;
    int* my_slot_id = s_slot_id + wave * SLOTS;

    auto flush = [&]() {
        if constexpr (WIDE) {
            // LDS rows written by this wave's own lanes: wave-local, no workgroup barrier needed
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            for (int s = 0; s < nslots; s++) {
                const float4* wrow = reinterpret_cast<const float4*>(my_w + s * 64 + th * PPL);
                float sum = 0.f;
#pragma unroll
                for (int p4 = 0; p4 < PPL / 4; p4++) {
                    const float4 wv = wrow[p4];
                    sum = fmaf(wv.x, dLT[4 * p4 + 0], sum);
                    sum = fmaf(wv.y, dLT[4 * p4 + 1], sum);
                    sum = fmaf(wv.z, dLT[4 * p4 + 2], sum);
                    sum = fmaf(wv.w, dLT[4 * p4 + 3], sum);
                }
                if constexpr (NH == 2) sum += __shfl_xor(sum, 32, 64);
                const int gid = my_slot_id[s];
                if (th == 0) atomicAdd(&dL_dcolors[(size_t)gid * C + tn], sum);
            }
            nslots = 0;
        }
    };

    for (int i = 0; i < rounds; i++) {
        __syncthreads();
        const int base = i * BATCH;               // entries already consumed (from the back)
        const int nb = min(BATCH, Lt - base);     // entries in this batch
        if (tid < nb) {
            const int coll_id = point_list[range.x + (Lt - 1 - base - tid)];
            s_id[tid] = coll_id;
            s_xy[tid] = points_xy_image[coll_id];
            s_co[tid] = conic_opacity[coll_id];
            if constexpr (C > 0 && !WIDE) {
#pragma unroll
                for (int ch = 0; ch < C; ch++) s_feat[tid * ROW + ch] = colors[(size_t)coll_id * C + ch];
            }
        }
        __syncthreads();
        if constexpr (WIDE) {
            constexpr int F4 = C / 4;
#pragma unroll
            for (int k = 0; k < F4; k++) {
                const int q = tid + BATCH * k;
                const int g = q / F4, part = q % F4;
                if (g < nb) s_feat4[g * F4 + part] = reinterpret_cast<const float4*>(colors + (size_t)s_id[g] * C)[part];
            }
            __syncthreads();
        }

        for (int j = 0; j < nb; j++) {
            const int pos = Lt - 1 - base - j;  // 0-based forward position in the tile list
            const float2 xy = s_xy[j];
            const float4 con_o = s_co[j];
            const float dx = xy.x - pixfx, dy = xy.y - pixfy;
            const float power = -0.5f * (con_o.x * dx * dx + con_o.z * dy * dy) - con_o.y * dx * dy;
            const float G = __expf(power);
            const float alpha = fminf(0.99f, con_o.w * G);
            const bool valid = (pos < last_contributor) && power <= 0.0f && alpha >= (1.0f / 255.0f);
            if (ballot64(valid) == 0) continue;

            const float one_m_alpha_inv = __builtin_amdgcn_rcpf(1.f - alpha);
            T = valid ? T * one_m_alpha_inv : T;
            const float w = valid ? alpha * T : 0.f;  // dchannel_dcolor
            const int global_id = s_id[j];

            if constexpr (C == 0) {
                // mask-only pair
                const float gm = wave_sum(w * dL_dout_mask_i);
                if (lane == 0) atomicAdd(&dL_dmask[global_id], gm);
                continue;
            } else {
                float S = 0.f;
#pragma unroll
                for (int ch = 0; ch < C; ch++) S = fmaf(s_feat[j * ROW + ch], dL_dpixel[ch], S);
                Rrec = valid ? (last_alpha * S_last + (1.f - last_alpha) * Rrec) : Rrec;
                float dL_dalpha = (S - Rrec) * T;
                S_last = valid ? S : S_last;
                last_alpha = valid ? alpha : last_alpha;
                dL_dalpha += (-T_final * one_m_alpha_inv) * bg_dot_dpixel;
                dL_dalpha = valid ? dL_dalpha : 0.f;

                const float dL_dG = con_o.w * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * con_o.x - gdy * con_o.y;
                const float dG_ddely = -gdy * con_o.z - gdx * con_o.y;
                float g0 = wave_sum(dL_dG * dG_ddelx * ddelx_dx);   // dL_dmean2D.x
                float g1 = wave_sum(dL_dG * dG_ddely * ddely_dy);   // dL_dmean2D.y
                float g2 = wave_sum(-0.5f * gdx * dx * dL_dG);      // dL_dconic.x
                float g3 = wave_sum(-0.5f * gdx * dy * dL_dG);      // dL_dconic.y
                float g4 = wave_sum(-0.5f * gdy * dy * dL_dG);      // dL_dconic.w
                float g5 = wave_sum(G * dL_dalpha);                 // dL_dopacity
                float g6 = 0.f;
                if constexpr (MASKGRAD) g6 = wave_sum(w * dL_dout_mask_i);
                {
                    // lanes 0..5(6) carry one field each: a single atomic instruction per (wave, Gaussian)
                    float* addr = &dL_dmean2D[(size_t)global_id * 3];
                    float v = g0;
                    if (lane == 1) { addr = &dL_dmean2D[(size_t)global_id * 3 + 1]; v = g1; }
                    if (lane == 2) { addr = &dL_dconic2D[(size_t)global_id * 4 + 0]; v = g2; }
                    if (lane == 3) { addr = &dL_dconic2D[(size_t)global_id * 4 + 1]; v = g3; }
                    if (lane == 4) { addr = &dL_dconic2D[(size_t)global_id * 4 + 3]; v = g4; }
                    if (lane == 5) { addr = &dL_dopacity[global_id]; v = g5; }
                    if constexpr (MASKGRAD) {
                        if (lane == 6) { addr = &dL_dmask[global_id]; v = g6; }
                    }
                    if (lane < (MASKGRAD ? 7 : 6)) atomicAdd(addr, v);
                }

                if constexpr (WIDE) {
                    my_w[nslots * 64 + lane] = w;
                    if (lane == 0) my_slot_id[nslots] = global_id;
                    nslots++;
                    if (nslots == SLOTS) flush();
                } else {
#pragma unroll
                    for (int ch = 0; ch < C; ch++) {
                        const float gc = wave_sum(w * dL_dpixel[ch]);
                        if (lane == 0) atomicAdd(&dL_dcolors[(size_t)global_id * C + ch], gc);
                    }
                }
            }
        }
        if constexpr (WIDE) flush();  // slot ids / feature rows of this batch die at the next barrier
    }
}

}  // namespace mirast
