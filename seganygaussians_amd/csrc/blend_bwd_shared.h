// blend_bwd_shared.h -- constants and the staged-record type shared by the MFMA backward blend kernels
// (blend_bwd_wave.h: the product kernel; blend_bwd_mfma.h: round 1's tile-batched kernel, profiling build only).
#pragma once

#include "common.h"

namespace mirast {

constexpr int WROW = 68;   // padded w/u row (floats)
constexpr int CHK = 16;    // rows per MFMA chunk
constexpr int DLROW = 33;  // padded gradient-image staging row (floats; 32 channels at a time)

typedef float v4f __attribute__((ext_vector_type(4)));

// One staged record: {x, y, -a/2, -b} {-c/2, opacity, list position << 4 | quadrant mask (int bits), Gaussian id (int bits)} with the conic
// (a, b, c) pre-scaled to (-a/2, -b, -c/2) for gauss_power (common.h).
struct BwdPar {
    float4 q0, q1;
};

}  // namespace mirast
