// mppi_common.hpp — Shared definitions of the gfx950 kernels: launch geometry (Dims), the noise identity (GenCtx), wave reductions.
// Part of the MPPI.forward() hot path for gfx950; see mppi_kernels.hpp for the map of the files.
#pragma once
#include <hip/hip_runtime.h>
#include "host_search.hpp"
#include "mppi_models.hpp"
#include "philox.hpp"

namespace mppi {

constexpr int WAVE = 64;
constexpr int BLOCK = 256;  // 4 waves

struct Dims {
    int64_t N;              // local samples
    int64_t tiles;          // ceil(N/64)
    int64_t sample_offset;  // global index of local sample 0
    int64_t inherit_count;  // global threshold of mppi.py:266
    int32_t T, R, row, dc;  // horizon, float4 groups per trajectory, row = T*dc, dim_control
    float u_min[MPPI_MAX_DIM_CONTROL], u_max[MPPI_MAX_DIM_CONTROL], sigma[MPPI_MAX_DIM_CONTROL];
};

// Identity of the noise of one solve: eps[i][t][k] is a pure function of (seed, solve, global i, t, k).
struct SgFilter {       // device half of the Savitzky-Golay step (window == 0: off)
    const float* coeffs;  // [window]
    float* history;       // [T-1][dc], updated by finalize_kernel
    int window;
};

struct GenCtx {
    uint32_t seed_lo, seed_hi, solve_idx;
};

// control dimension of column j of a float4 group (flat horizon index 4r + j): dc is 1, 2 or 4, so it
// does not depend on r (a row-dependent index would make the per-column bounds 4*CH distinct
// loop-invariant scalars, which the compiler hoists and spills).  dim_control = 3 is padded to 4 by
// the caller.
__device__ __forceinline__ int ctrl_index(int j, int dc) { return j & (dc - 1); }

__device__ __forceinline__ unsigned float_to_key(float f) {  // order-preserving map for atomicMin
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// The SAME butterflies (partner lane ^ 32, ^ 16, ^ 8, ^ 4, ^ 2, ^ 1 in this order, every lane ends with the total) without
// the LDS crossbar: __shfl_xor is a ds_bpermute (~100 cycles of latency per dependent stage, 0.5 us per reduction measured
// in lbps_brent_kernel's probe); here the partner arrives through v_permlane32_swap / v_permlane16_swap (gfx950) and DPP
// (row_ror:8 = lane ^ 8 inside a row of 16; lane ^ 4 = row_shl:4 into banks 0, 2 + row_shr:4 into banks 1, 3; quad_perm for
// ^ 2, ^ 1).  a + b = b + a to the bit, so each lane computes exactly what wave_sum's stage computes: bit-identical results
// (scripts/ubench/bfly_check.hip holds the two against each other on the device).
// One stage: op(v, v of lane ^ STAGE) in every lane.  For 32 / 16 the two swapped copies hold (lower, lower) and (upper,
// upper) halves / (even, even) and (odd, odd) rows, so op of the two IS the stage in every lane (op commutes).
template <int STAGE>
__device__ __forceinline__ unsigned bfly_dpp(unsigned u) {  // v of lane ^ STAGE, STAGE <= 8
    if constexpr (STAGE == 8) {
        return (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x128, 0xF, 0xF, false);  // row_ror:8
    } else if constexpr (STAGE == 4) {
        int t = __builtin_amdgcn_update_dpp(0, (int)u, 0x104, 0xF, 0x5, false);            // row_shl:4 -> lanes with bit 2 clear
        t = __builtin_amdgcn_update_dpp(t, (int)u, 0x114, 0xF, 0xA, false);                // row_shr:4 -> lanes with bit 2 set
        return (unsigned)t;
    } else if constexpr (STAGE == 2) {
        return (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x4E, 0xF, 0xF, false);    // quad_perm:[2,3,0,1]
    } else {
        static_assert(STAGE == 1, "stage");
        return (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0xB1, 0xF, 0xF, false);    // quad_perm:[1,0,3,2]
    }
}
template <int STAGE, class OP>
__device__ __forceinline__ float bfly_stage(float v, OP op) {
    const unsigned u = __float_as_uint(v);
    if constexpr (STAGE == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        return op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    } else if constexpr (STAGE == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        return op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    } else {
        return op(v, __uint_as_float(bfly_dpp<STAGE>(u)));
    }
}
template <int STAGE, class OP>
__device__ __forceinline__ double bfly_stage(double v, OP op) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
    const auto join = [](unsigned l, unsigned h) { return __longlong_as_double((long long)(((unsigned long long)h << 32) | l)); };
    if constexpr (STAGE == 32) {
        const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        return op(join(rl[0], rh[0]), join(rl[1], rh[1]));
    } else if constexpr (STAGE == 16) {
        const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        return op(join(rl[0], rh[0]), join(rl[1], rh[1]));
    } else {
        return op(v, join(bfly_dpp<STAGE>(lo), bfly_dpp<STAGE>(hi)));
    }
}
template <class T, class OP>
__device__ __forceinline__ T wave_bfly(T v, OP op) {
    v = bfly_stage<32>(v, op); v = bfly_stage<16>(v, op); v = bfly_stage<8>(v, op);
    v = bfly_stage<4>(v, op); v = bfly_stage<2>(v, op); v = bfly_stage<1>(v, op);
    return v;
}
__device__ __forceinline__ float wave_sum_bfly(float v) { return wave_bfly(v, [](float a, float b) { return a + b; }); }
__device__ __forceinline__ double wave_sum_bfly(double v) { return wave_bfly(v, [](double a, double b) { return a + b; }); }
__device__ __forceinline__ float wave_max_bfly(float v) { return wave_bfly(v, [](float a, float b) { return fmaxf(a, b); }); }

}  // namespace mppi
