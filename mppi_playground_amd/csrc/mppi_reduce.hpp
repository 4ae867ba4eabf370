// mppi_reduce.hpp — Steps 5-6 (mppi.py:376-384): softmax weights and the weighted sum of the clamped actions as per-block partial rows — weights_reduce_kernel.
// Part of the MPPI.forward() hot path for gfx950; see mppi_kernels.hpp for the map of the files.
#pragma once
#include "mppi_sample.hpp"

namespace mppi {

// ------------------------------------------------------------------------------------------
// Steps 5-6: e_i = exp((-c_i)/lambda - max_j(-c_j)/lambda) and A = sum_i e_i * clamp(mean + eps_i)
// (mppi.py:376-384, un-normalised) as per-block partial rows.
//
// Phase A (per wave): the costs of TPW tiles are loaded together (one memory latency instead of TPW in a
// chain), turned into weights and a wave-uniform bitmask of the tiles that carry any weight; the weights of
// those tiles are parked in LDS.  Tiles whose 64 weights are all exactly zero are never touched again
// (exact: they add 0) — with a sharp softmax (racing, lambda = 1) that is all but a handful of tiles.
// Phase B (per block): every live tile of the block is accumulated by ALL waves, wave w taking the float4
// groups r = w, w+NW, ...: a single heavy tile is a 4x shorter dependent chain than one wave walking the whole
// row, and each column is owned by exactly one wave, so no cross-wave sum is needed.  Each lane accumulates
// its trajectory in GPW*4 registers; the 64 lanes are combined through a padded LDS tile, 32 accumulators at
// a time (a fully unrolled register butterfly is ~40 KB of straight-line code executed once per wave and
// ran instruction-fetch bound).
// Blocks publish one partial row only if they saw a live tile (heads[b][3] is the flag); summarize_kernel
// folds the published rows.
// partials layout: [gridDim.x][colsp] with colsp = gridDim.y * NW * GPW * 4; heads: [gridDim.x][4].
constexpr int REDUCE_MAX_BLOCKS = 2048;
// WIDE: per-column clamp bounds from `coltab` (see gen_noise4) staged in LDS next to the mean; tiles only (GEN = false).
#ifndef MPPI_REDUCE_ATTR
#define MPPI_REDUCE_ATTR  // (A/B knob of scripts/build_variant.sh, e.g. __attribute__((amdgpu_waves_per_eu(2,3))))
#endif
// CHAINS: Philox + Box-Muller chains per basic block of the regenerating reduction.  4: a lone wave per SIMD (grids of a few
// hundred blocks: C2) hides the chains' latencies inside its own instruction stream, 104 VGPRs; 2: 72 VGPRs = seven waves
// per SIMD, the interleaving comes from the other waves (C3 / C5 sizes).  The host picks by the tile count.
// REM: some chunk of the row has groups left over (R % 4 != 0): without them (cart-pole's 16-group rows, ...) the remainder's
// accumulators and its loop-carried state are compiled out — they cost the 16-group reduction 66 instructions per tile and wave
// (a second block of accumulator moves at the loop edge): C5 +2 us.
template <int GPW, bool GEN, bool WIDE = false, int CHAINS = 2, bool REM = true>  // GPW: float4 groups per wave and column chunk (8: the host launches ceil(R / 32) chunks)
__global__ __launch_bounds__(BLOCK) MPPI_REDUCE_ATTR void weights_reduce_kernel(const float4* __restrict__ noise,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ costs,
                                                               const unsigned* __restrict__ min_key,
                                                               float* __restrict__ partials,
                                                               float* __restrict__ heads, Dims d, GenCtx gen,
                                                               float lambda_arg, const float* __restrict__ lambda_dev,
                                                               const float* __restrict__ coltab) {
    static_assert(!(GEN && WIDE), "wide control rows are reduced from the materialised tiles");
    // the temperature: a launch constant, or (ESSPS searched on the device) the value the search left in HBM
    const float lambda = lambda_dev ? *lambda_dev : lambda_arg;
    constexpr int NACC = GPW * 4;
    constexpr int NW = BLOCK / WAVE;
    constexpr int CHG = NW * GPW;  // float4 groups per column chunk
    constexpr int TPW = 8;
    constexpr int RP = 8;  // accumulators combined per pass of the cross-lane sum (round 5: 8, was 32 — 33 KB of LDS for a
                           // tile used once after the loop held the kernel at three waves per SIMD)
    __shared__ float s_red[NW][RP][WAVE + 1];
    __shared__ float s_e[NW][TPW][WAVE];
    __shared__ unsigned s_live[NW];
    __shared__ float s_head[NW][4];
    // this chunk's mean groups and an all-zero copy for samples that do not inherit the mean.  Read
    // from LDS inside the tile loop (with an opaque offset) so that the compiler does not hoist the
    // loop-invariant scalar loads into SGPRs: that spilled ~450 SGPRs in every wave's prologue.
    __shared__ __attribute__((aligned(16))) float s_mean[2][CHG * 4];
    __shared__ __attribute__((aligned(16))) float s_bnd[2][WIDE ? CHG * 4 : 4];  // WIDE: lo / hi of this chunk's columns
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: lives in an SGPR
    const int r0 = blockIdx.y * CHG;  // first float4 group of this column chunk
    for (int j = threadIdx.x; j < CHG * 4; j += BLOCK) {
        const int f = 4 * r0 + j;
        s_mean[0][j] = f < d.row ? mean[f] : 0.0f;
        s_mean[1][j] = 0.0f;
        if (WIDE) {
            s_bnd[0][j] = f < d.row ? coltab[4 * d.R + f] : 0.0f;
            s_bnd[1][j] = f < d.row ? coltab[8 * d.R + f] : 0.0f;
        }
    }
    const float cmin = key_to_float(*min_key);
    const float xmax = (-cmin) / lambda;
    float acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = 0.0f;
    // Ownership of this chunk's `ng` float4 groups (round 5): wave w owns r0 + w + NW*m for m < full = ng / NW — the same
    // count for every wave — and the rem = ng % NW groups left over are spread over the TILES: with one left over, the
    // wave (tile % 4) takes it; with two, waves {0, 1} take them on even tiles and {2, 3} on odd ones; three stay with
    // waves 0..2.  Either way a wave only ever sees ONE remainder group (gx), so it needs one more accumulator set
    // (accx), and the waves that share a group add theirs up at the end.  With the old static split a 25-group row
    // (racing, nav2d) gave wave 0 seven groups and the others six: every block waited for its wave 0 at the round
    // barrier, and all wave 0s share a SIMD — 12 % of a dense reduction.
    float accx[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    __shared__ float s_x[NW][4];
    const int ng = min(CHG, d.R - r0);
    float se = 0.0f, se2 = 0.0f, sec = 0.0f;
    const int64_t nwaves = (int64_t)gridDim.x * NW;
    bool block_live = false;  // block-uniform
    for (int64_t base0 = (int64_t)blockIdx.x * NW; base0 < d.tiles; base0 += nwaves * TPW) {
        // ---- phase A: this wave's TPW tiles
        float cc[TPW];
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            const int64_t i = (base0 + wid + q * nwaves) * 64 + lane;
            cc[q] = (i < d.N) ? costs[i] : INFINITY;  // tiles past the end have i >= N as well
        }
        unsigned live = 0;
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            const float e = expf((-cc[q]) / lambda - xmax);  // exp(-inf) = 0 for the padding lanes
            const bool tile_live = __ballot(e != 0.0f) != 0ull;
            live |= (tile_live ? 1u : 0u) << q;
            if (tile_live) {  // wave-uniform
                s_e[wid][q][lane] = e;
                const float c = e != 0.0f ? cc[q] : 0.0f;  // (keeps 0 * inf out of the padding lanes)
                se += e;
                se2 = fmaf(e, e, se2);
                sec = fmaf(e, c, sec);
            }
        }
        if (lane == 0) s_live[wid] = live;
        __syncthreads();
        // ---- phase B: the block's live tiles, this wave's groups
        for (int w2 = 0; w2 < NW; ++w2) {
            const unsigned lv = __builtin_amdgcn_readfirstlane(s_live[w2]);
            if (lv == 0u) continue;
            block_live = true;
            for (int q = 0; q < TPW; ++q) {
                if (!((lv >> q) & 1u)) continue;
                const int64_t tile = base0 + w2 + q * nwaves;
                const int64_t i = tile * 64 + lane;
                const float e = s_e[w2][q][lane];
                const uint64_t gi = (uint64_t)(d.sample_offset + i);
                const bool inherit = (d.sample_offset + i) < d.inherit_count;
                const float4* np = noise + (tile * d.R) * 64 + lane;
                int moff = inherit ? 0 : CHG;           // float4 offset of this lane's copy of the mean groups
                asm volatile("" : "+v"(moff));          // opaque: keeps the LDS reads inside the loop
                const float4* mp = reinterpret_cast<const float4*>(&s_mean[0][0]) + moff;
                int full = ng / NW;                     // groups r0 + wid + NW*m, m < full, exist for every wave
                asm volatile("" : "+s"(full));          // opaque: keeps the group predicates out of SGPRs
                const int rem = ng - full * NW;
                // one float4 group (index g inside the chunk) of this tile into four accumulators
                const auto accumulate4 = [&](float* a4, int g, const float4& n4) {
                    const float4 m4 = mp[g];
                    const float nv[4] = {n4.x, n4.y, n4.z, n4.w};
                    const float mv[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float u;
                        if (WIDE) {
                            const int cj = 4 * g + j;  // column inside this chunk
                            u = clampf(mv[j] + nv[j], s_bnd[0][cj], s_bnd[1][cj]);
                        } else {
                            const int k = ctrl_index(j, d.dc);
                            u = clampf(mv[j] + nv[j], d.u_min[k], d.u_max[k]);
                        }
                        a4[j] = fmaf(e, u, a4[j]);
                    }
                };
                const auto accumulate = [&](int m, const float4& n4) { accumulate4(&acc[4 * m], wid + NW * m, n4); };
                if constexpr (GEN) {
                    // Regenerated noise: the wave's groups are taken CHAINS at a time while that many exist, so that
                    // independent Philox + Box-Muller chains (10 dependent 64-bit multiplies each) sit in one basic block
                    // and overlap — a per-group branch serialised them.  Only groups of the row are generated (round 5:
                    // predicating whole quads generated 32 groups per tile for racing's / nav2d's 25-group rows).
                    const auto quad = [&](int m0) {
                        float4 n4[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) n4[k] = noise_group<true>(np, r0 + wid + NW * (m0 + k), gi, gen, d);
#pragma unroll
                        for (int k = 0; k < 4; ++k) accumulate(m0 + k, n4[k]);
                    };
                    const auto pair = [&](int m0) {
                        float4 n4[2];
#pragma unroll
                        for (int k = 0; k < 2; ++k) n4[k] = noise_group<true>(np, r0 + wid + NW * (m0 + k), gi, gen, d);
#pragma unroll
                        for (int k = 0; k < 2; ++k) accumulate(m0 + k, n4[k]);
                    };
                    const auto single = [&](int m) { accumulate(m, noise_group<true>(np, r0 + wid + NW * m, gi, gen, d)); };
                    static_assert(GPW == 8, "the cases below are written for eight groups per wave");
                    // (independent `if`s, no else branches: every region updates its accumulators in place or not at all —
                    // nested if / else chains made the compiler add a second block of 28 accumulator moves per tile)
                    if constexpr (CHAINS == 4) {
                        if (full >= 4) quad(0);
                        if (full >= 8) quad(4);
                        if (full == 6 || full == 7) pair(4);
                        if (full == 7) single(6);
                        if (full == 5) single(4);
                        if (full == 2 || full == 3) pair(0);
                        if (full == 3) single(2);
                        if (full == 1) single(0);
                    } else {
                        (void)quad;
#pragma unroll
                        for (int m0 = 0; m0 < GPW; m0 += 2) {
                            if (full >= m0 + 2) pair(m0);
                            if (full == m0 + 1) single(m0);
                        }
                    }
                } else {
#pragma unroll
                    for (int m = 0; m < GPW; ++m)
                        if (m < full) accumulate(m, noise_group<false>(np, r0 + wid + NW * m, gi, gen, d));
                }
                // this wave's share of the remainder groups (see `accx` above)
                const bool mine = REM && (rem == 1 ? ((int)tile & 3) == wid : rem == 2 ? ((int)tile & 1) == (wid >> 1) : wid < rem);
                if (mine) {  // wave-uniform
                    const int g = NW * full + (rem == 1 ? 0 : rem == 2 ? (wid & 1) : wid);
                    accumulate4(accx, g, noise_group<GEN>(np, r0 + g, gi, gen, d));
                }
            }
        }
        __syncthreads();  // s_e / s_live are rewritten by the next round
    }
    // cross-lane reduction, RP = 8 accumulators per pass: every lane stores its 8 values as a column of
    // s_red[wid][j][lane]; lane l then sums row j = l & 7 over the 8 lanes [8*(l>>3), +8) (row stride 65 floats:
    // conflict-free), and the eight segments are added with three shuffles.  Accumulator 4*m + j of wave
    // w is column 4*(r0 + w + NW*m) + j of the row (m < ng / NW); the remainder groups' accumulators are summed over the
    // waves that took them (fixed order) and are columns 4*(r0 + NW*(ng/NW) + group) + j.
    const int colsp = gridDim.y * CHG * 4;
    const auto lane_sum8 = [&](const float* a8) {  // lanes 0..7 return the sums over the wave of a8[0..7]
#pragma unroll
        for (int j = 0; j < RP; ++j) s_red[wid][j][lane] = a8[j];
        __builtin_amdgcn_wave_barrier();
        const float* rowp = &s_red[wid][lane & (RP - 1)][(lane >> 3) * 8];
        float v0 = rowp[0] + rowp[1], v1 = rowp[2] + rowp[3], v2 = rowp[4] + rowp[5], v3 = rowp[6] + rowp[7];
        float v = (v0 + v1) + (v2 + v3);
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        __builtin_amdgcn_wave_barrier();
        return v;
    };
    if (block_live) {  // (block-uniform)
        const int full = ng / NW, rem = ng - full * NW;
#pragma unroll
        for (int p = 0; p < NACC / RP; ++p) {
            const float v = lane_sum8(&acc[p * RP]);
            const int a = p * RP + lane;
            if (lane < RP && (a >> 2) < full) partials[(int64_t)blockIdx.x * colsp + 4 * (r0 + wid + NW * (a >> 2)) + (a & 3)] = v;
        }
        if (REM && rem) {
            float x8[RP];
#pragma unroll
            for (int j = 0; j < RP; ++j) x8[j] = j < 4 ? accx[j & 3] : 0.0f;
            const float v = lane_sum8(x8);
            if (lane < 4) s_x[wid][lane] = v;
            __syncthreads();
            if (threadIdx.x < 4 * rem) {  // remainder group j = threadIdx.x >> 2: the waves that took it, in order
                const int j = threadIdx.x >> 2, jj = threadIdx.x & 3;
                float t = 0.0f;
#pragma unroll
                for (int w = 0; w < NW; ++w)
                    if (rem == 1 || (rem == 2 ? (w & 1) == j : w == j)) t += s_x[w][jj];
                partials[(int64_t)blockIdx.x * colsp + 4 * (r0 + NW * full + j) + jj] = t;
            }
        }
    }
    se = wave_sum(se);
    se2 = wave_sum(se2);
    sec = wave_sum(sec);
    if (lane == 0) { s_head[wid][0] = se; s_head[wid][1] = se2; s_head[wid][2] = sec; }
    __syncthreads();
    if (blockIdx.y == 0 && threadIdx.x < 4) {
        float v = block_live ? 1.0f : 0.0f;
        if (threadIdx.x < 3) {
            v = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += s_head[w][threadIdx.x];
        }
        heads[(int64_t)blockIdx.x * 4 + threadIdx.x] = v;
    }
}

}  // namespace mppi
