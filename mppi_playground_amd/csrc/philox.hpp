// philox.hpp — counter-based device RNG of the sampler (step 1 of MPPI.forward, mppi.py:261-263).
//
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11).
// key     = (seed_lo, seed_hi)
// counter = (global sample index lo, hi, float4 group r of the [T*dc] row, solve index)
// The four 32-bit outputs become four normals through two Box-Muller pairs, so one call fills one
// float4 of the lane-major noise tile and the value of eps[i][t][k] depends only on (seed, solve,
// global i, t, k) — not on how num_samples is sharded across GPUs.
#pragma once
#include <stdint.h>

#ifndef MPPI_PHILOX_ROUNDS
#define MPPI_PHILOX_ROUNDS 10
#endif

namespace mppi {

struct u32x4 {
    uint32_t x, y, z, w;
};

// k0v / k1v / k0w: the key again — (k0, k1) for round 0, k0 + 0x9E3779B9 for round 1.  With a wave-uniform group and solve
// index (c2, c3: every caller's) round 0 XORs two uniform words per output, and so does the first output of round 1 (its
// c1 is the low half of the uniform product of round 0); a VALU instruction reads ONE scalar register, so one of the two
// needs a vector copy.  A caller with a hot loop hands in copies it pinned in VGPRs once (trajectory_cost); everybody
// else passes nothing and the compiler makes the copies where it needs them.
__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1, uint32_t k0v, uint32_t k1v, uint32_t k0w) {
#pragma unroll
    for (int r = 0; r < MPPI_PHILOX_ROUNDS; ++r) {
        // one 32x32->64 multiply each (v_mad_u64_u32): integer multiplies are quarter rate on CDNA
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        // three-input XOR in one v_bitop3_b32 (truth table 0x96)
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32(hi1, c1, r == 0 ? k0v : r == 1 ? k0w : k0, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32(hi0, c3, r == 0 ? k1v : k1, 0x96);
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}
__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1) {
    return philox4x32_10(c0, c1, c2, c3, k0, k1, k0, k1, k0 + 0x9E3779B9u);
}

// Box-Muller on (a, b): u1 = ((a>>8)+1) * 2^-24 in (0,1], u2 = (b>>9) * 2^-23 in [0,1).
// Hardware transcendentals: v_log_f32 (log2), v_sqrt_f32, v_sin_f32 / v_cos_f32.  The latter take their argument in
// REVOLUTIONS and are periodic in 1, so the angle needs neither the 2*pi multiply nor a conversion: the 23 bits of u2 are
// the mantissa of a float in [1, 2) (one shift + one OR), and cos(2 pi (1 + u2)) = cos(2 pi u2).
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
    const float u1 = (float)((a >> 8) + 1u) * (1.0f / 16777216.0f);
    const float rev = __uint_as_float(0x3f800000u | (b >> 9));  // 1 + u2
    const float rad = __builtin_amdgcn_sqrtf(-1.38629436112f * __builtin_amdgcn_logf(u1));  // -2 ln2 log2(u1)
    z0 = rad * __builtin_amdgcn_cosf(rev);
    z1 = rad * __builtin_amdgcn_sinf(rev);
}

}  // namespace mppi
