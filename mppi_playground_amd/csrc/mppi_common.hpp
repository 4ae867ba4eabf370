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

}  // namespace mppi
