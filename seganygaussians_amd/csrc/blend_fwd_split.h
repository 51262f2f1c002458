// blend_fwd_split.h -- the exact three-term bf16 split of f32 operands and the staged-record type shared by the forward blend
// kernels that accumulate on the bf16 matrix pipe (blend_fwd_wave.h: the product kernel; blend_fwd_x3.h: round 2's tile-batched
// kernel, profiling build only).  Every f32 operand x is split EXACTLY into x = hi + mid + lo (8 + 8 + 8 significant bits, each the
// round-to-nearest bf16 of the remainder); a product is the sum of the six partial products hi.hi + hi.mid + mid.hi + hi.lo + lo.hi
// + mid.mid, each exact in f32 inside the MFMA and accumulated in f32 -- the three dropped terms are below 2^-26 of |w f|.
#pragma once

#include "common.h"

namespace mirast {

// a staged feature row: bf16 hi[C] | mid[C] | lo[C], i.e. 6 C bytes (192 at C = 32)
constexpr int XG = 16;     // Gaussians per MFMA group

// One staged record: {x, y, -a/2, -b} {-c/2, opacity, list position + 1, (position << 4 | quadrant mask)}
struct XRec {
    float4 q0, q1;
};

typedef float v16f __attribute__((ext_vector_type(16)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));  // one MFMA operand: 8 bf16
typedef float v2fx __attribute__((ext_vector_type(2)));
typedef __bf16 v2bfx __attribute__((ext_vector_type(2)));

// (a, b) -> three dwords holding a's term in the low and b's term in the high half-word: a = hi + mid + lo exactly
__device__ __forceinline__ void split3_bf16x2(float a, float b, uint32_t& hi, uint32_t& mid, uint32_t& lo)
{
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector((v2fx){a, b}, v2bfx));
    const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
    mid = __builtin_bit_cast(uint32_t, __builtin_convertvector((v2fx){ra, rb}, v2bfx));
    const float sa = ra - __uint_as_float(mid << 16), sb = rb - __uint_as_float(mid & 0xffff0000u);
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector((v2fx){sa, sb}, v2bfx));
}

}  // namespace mirast
