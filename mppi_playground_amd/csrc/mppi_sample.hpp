// mppi_sample.hpp — Step 1 (mppi.py:255-263): the device noise stream (gen_noise4 is its definition), sample_kernel, posterior draws.
// Part of the MPPI.forward() hot path for gfx950; see mppi_kernels.hpp for the map of the files.
#pragma once
#include "mppi_common.hpp"

namespace mppi {

// ------------------------------------------------------------------------------------------
// Step 1: eps ~ N(0, diag(sigma^2)).  gen_noise4() is THE definition of the device noise: float4
// group r of global sample gi.  It is used by sample_kernel (materialise the lane-major tiles) and,
// in "regen" mode, directly by the rollout and reduction kernels, which then never touch HBM for
// the noise (Philox + Box-Muller is ~25 VALU per normal, cheaper than a 16 B/lane HBM round trip).
// WIDE (generic handles whose dim_control is not 1, 2 or 4): the control index of a flat column depends on the
// group, so sigma / bounds come from the per-column table `coltab` = {sigma[4R], lo[4R], hi[4R]} (built on the host,
// zeros past the row) instead of the launch constants in Dims.
// The two halves of gen_noise4 (the integer hash and the Box-Muller transform of its output), separately callable so
// that the rollout loop can run them one group apart (software pipeline: see trajectory_cost).
__device__ __forceinline__ u32x4 noise_bits(uint64_t gi, int r, const GenCtx& g) {
    return philox4x32_10((uint32_t)gi, (uint32_t)(gi >> 32), (uint32_t)r, g.solve_idx, g.seed_lo, g.seed_hi);
}
template <bool WIDE = false>
__device__ __forceinline__ float4 noise_from_bits(const u32x4& x, int r, const Dims& d,
                                                  const float* __restrict__ sig_cols = nullptr) {
    float z[4];
    box_muller(x.x, x.y, z[0], z[1]);
    box_muller(x.z, x.w, z[2], z[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) z[j] *= WIDE ? sig_cols[4 * r + j] : d.sigma[ctrl_index(j, d.dc)];
    return make_float4(z[0], z[1], z[2], z[3]);
}
struct KeyPins { uint32_t k0v, k1v, k0w; };  // key words in VGPRs (philox4x32_10: rounds 0 and 1), pinned once by a caller with a hot loop
template <bool WIDE = false>
__device__ __forceinline__ float4 gen_noise4(uint64_t gi, int r, const GenCtx& g, const Dims& d,
                                             const float* __restrict__ sig_cols = nullptr, const KeyPins* pins = nullptr) {
    const u32x4 x = philox4x32_10((uint32_t)gi, (uint32_t)(gi >> 32), (uint32_t)r, g.solve_idx, g.seed_lo, g.seed_hi,
                                  pins ? pins->k0v : g.seed_lo, pins ? pins->k1v : g.seed_hi,
                                  pins ? pins->k0w : g.seed_lo + 0x9E3779B9u);
    float z[4];
    box_muller(x.x, x.y, z[0], z[1]);
    box_muller(x.z, x.w, z[2], z[3]);
    // columns past the row length (row % 4 != 0) carry unused values: no consumer reads them
#pragma unroll
    for (int j = 0; j < 4; ++j) z[j] *= WIDE ? sig_cols[4 * r + j] : d.sigma[ctrl_index(j, d.dc)];
    return make_float4(z[0], z[1], z[2], z[3]);
}

// HBM-write bound: 16 B per lane per Philox call, one 1 KiB store per wave instruction.
template <bool WIDE>
__global__ __launch_bounds__(BLOCK) void sample_kernel(float4* __restrict__ noise, Dims d, GenCtx g,
                                                       const float* __restrict__ coltab) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6);
    if (tile >= d.tiles) return;
    const uint64_t gi = (uint64_t)(d.sample_offset + tile * 64 + lane);
    float4* out = noise + tile * d.R * 64 + lane;
    for (int r = 0; r < d.R; ++r) out[(int64_t)r * 64] = gen_noise4<WIDE>(gi, r, g, d, coltab);
}

// get_samples_from_posterior (mppi.py:489-506): samples[q][f] = loc[f] + eps_q[f] with eps ~ N(0, diag(sigma^2)) from
// the Philox stream of a solve index RESERVED for this call (counter = (q, group, solve_idx): the draw advances the
// solver's stream exactly like a forward() would, and every shard draws the same k samples).  Unclamped, like the
// reference's MultivariateNormal(loc=optimal_solution).sample().  One thread per (sample, float4 group).
template <bool WIDE>
__global__ __launch_bounds__(BLOCK) void posterior_sample_kernel(const float* __restrict__ loc, int k,
                                                                 float* __restrict__ samples, Dims d, GenCtx g,
                                                                 const float* __restrict__ coltab) {
    const int64_t idx = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= (int64_t)k * d.R) return;
    const int q = (int)(idx / d.R), r = (int)(idx - (int64_t)q * d.R);
    const float4 n4 = gen_noise4<WIDE>((uint64_t)q, r, g, d, coltab);
    const float nv[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = 4 * r + j;
        if (f < d.row) samples[(int64_t)q * d.row + f] = loc[f] + nv[j];
    }
}

// One float4 group of a lane's noise row: from the tiles (GEN=false) or regenerated (GEN=true).
template <bool GEN>
__device__ __forceinline__ float4 noise_group(const float4* __restrict__ np, int r, uint64_t gi, const GenCtx& g,
                                              const Dims& d, const KeyPins* pins = nullptr) {
    if (GEN) return gen_noise4(gi, r, g, d, nullptr, pins);
    return np[(int64_t)r * 64];
}

}  // namespace mppi
