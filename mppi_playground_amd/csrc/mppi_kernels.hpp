// mppi_kernels.hpp — gfx950 kernels of the MPPI.forward() hot path.
//
// Mapping (CDNA4, wave64): one LANE per trajectory, one WAVEFRONT per tile of 64 trajectories.
// The horizon recurrence is serial in t, so the 64 lanes of a wave advance 64 independent
// trajectories in lock-step; the noise is stored lane-major (see include/mppi_hip.h) so each
// wave-level load/store is one contiguous 1 KiB segment, no LDS transpose is needed on the hot
// path, and state stays in VGPRs for the whole horizon.  LDS is used only for the block-level
// reductions and for the [N][T][dc] <-> tile layout conversions (inject/export).
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

// ------------------------------------------------------------------------------------------
// Steps 1b-3 fused: U = clamp(mean + eps), rollout, stage + terminal cost (mppi.py:266-336).
// Reads the noise once (16 B per lane per 4/dc steps), writes costs[N] and the shard minimum.
//
// trajectory_cost(): one lane walks one trajectory.  `np` points at the lane's first float4 of the
// tile; consecutive groups are 64 float4 apart.
// `mean4` (R float4 groups, same grouping as the noise row; lanes beyond the exploration threshold
// are handed an all-zero copy, mppi.py:266-270) and `ktab` (KROW floats per step) are the
// block's LDS copies of the wave-uniform per-step inputs: LDS returns in order, so the compiler can
// keep the fetch of the next group / next row in flight (lgkmcnt(N)) while the current step computes,
// which scalar (SMEM) loads — out of order, lgkmcnt(0) only — do not allow.
// UC: the solver's clamp range lies inside the model's own action clamp (compile-time so that the
// second clamp disappears).
// VAR: a launch-uniform model variant the kernel has branched on OUTSIDE the horizon loop (racing: unit wheel base).
// X0OUT (models with EntryGeneral only): the start lies outside the model's position clamp — launch-uniform, x0 is the
// same for every lane — so the stage cost of step 0 takes the bounds-tested map lookup (every later state is clamped).
template <int MODEL, int FAST, bool GEN, bool UC, bool VAR = false, bool X0OUT = false>
__device__ __forceinline__ float trajectory_cost(const float4* __restrict__ np, uint64_t gi, const GenCtx& gen,
                                                 const float4* mean4, const float* ktab,
                                                 const float* __restrict__ x0, const Dims& d, const ModelCtx& ctx_in,
                                                 bool& bad) {
    using M = ModelT<MODEL, FAST>;
    using K = typename M::K;
    constexpr int DS = M::DS, DC = M::DC, SPG = 4 / DC;
    // wave-uniform operands that would otherwise cost a v_mov per use inside the loop (a VALU instruction reads one scalar
    // register): the model's picks of its launch constants (Model::pin_hot) and the Philox key of round 0
    ModelCtx ctx = ctx_in;
    if constexpr (FAST != 0 && EntryGeneral<M>::value) M::pin_hot(ctx);
    KeyPins pins{gen.seed_lo, gen.seed_hi, gen.seed_lo + 0x9E3779B9u};
    if (GEN) asm volatile("" : "+v"(pins.k0v), "+v"(pins.k1v), "+v"(pins.k0w));
    float s[DS], pu[DC], pl[DC];
#pragma unroll
    for (int j = 0; j < DS; ++j) s[j] = x0[j];
    if (FAST) {
        if constexpr (EntryGeneral<M>::value) {
            M::enter_any(s);  // any finite heading, wrapped once by the reference's own operation (exact)
        } else {
            M::check_state(ctx, s, bad);
            M::enter(s);  // kinematic models: wrap the heading once; every later heading is a fixed point of that wrap
        }
    }
    // clamp bounds live in VGPRs: v_med3_f32 takes one SGPR operand only, and the compiler would
    // otherwise re-materialise the second bound with a v_mov in every step
    float lo[DC], hi[DC];
#pragma unroll
    for (int k = 0; k < DC; ++k) {
        lo[k] = d.u_min[k]; hi[k] = d.u_max[k];
        asm volatile("" : "+v"(hi[k]));
    }
    CostSum<exact_cost_sum(MODEL)> acc;  // sum of the stage costs (mppi.py:333): exactly rounded (racing: sequential fp32)
    K knext = M::load_k(ktab, 0);
    float4 e = noise_group<GEN>(np, 0, gi, gen, d, &pins);
    float4 m4 = mean4[0];
    {   // info["prev_action"] of step 0 is U[:, 0] itself (mppi.py:299-301)
        const float e0[4] = {e.x, e.y, e.z, e.w}, m0[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
        for (int k = 0; k < DC; ++k) pu[k] = pl[k] = clampf(m0[k] + e0[k], lo[k], hi[k]);
    }
    int t = 0;
    auto one_step = [&](const float* ev, const float* mv) {
        const K kcur = knext;
        knext = M::load_k(ktab, min(t + 1, d.T - 1));
        float u[DC];
#pragma unroll
        for (int k = 0; k < DC; ++k) u[k] = clampf(mv[k] + ev[k], lo[k], hi[k]);
        float sn[DS], ss[DS];
        if constexpr (MODEL == MPPI_MODEL_RACING) M::step(ctx, s, u, sn, ss, bad, UC, FAST != 0, VAR);
        else M::step(ctx, s, u, sn, ss, bad, UC, FAST != 0);
        if constexpr (X0OUT) acc.add(M::cost(ctx, kcur, ss, u, pu, bad, t == 0));
        else acc.add(M::cost(ctx, kcur, ss, u, pu, bad));
#pragma unroll
        for (int k = 0; k < DC; ++k) { pl[k] = pu[k]; pu[k] = u[k]; }
#pragma unroll
        for (int j = 0; j < DS; ++j) s[j] = sn[j];
        ++t;
    };
    // groups that lie completely inside the horizon: SPG steps each, no per-step bound checks
    const int full = d.T / SPG;
    if (GEN) {
        for (int r = 0; r < full; ++r) {
            const int rn = min(r + 1, d.R - 1);
            const float4 en = noise_group<GEN>(np, rn, gi, gen, d, &pins);  // independent chain, interleaved with the steps
            const float4 m4n = mean4[rn];
            const float ev[4] = {e.x, e.y, e.z, e.w};
            const float mv[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
            for (int g = 0; g < SPG; ++g) one_step(ev + g * DC, mv + g * DC);
            e = en;
            m4 = m4n;
        }
    } else {
        // Tiles: loads return in order (one vmcnt), so the first map gather consumed after a noise load also
        // waits for that load.  Keep two groups in flight and issue the load of group r+2 at the very END of
        // iteration r (pinned by a fake dependency on the iteration's result): it then has most of iteration r+1 to arrive.
        float4 e1 = noise_group<GEN>(np, min(1, d.R - 1), gi, gen, d);
        for (int r = 0; r < full; ++r) {
            const float4 m4n = mean4[min(r + 1, d.R - 1)];
            const float ev[4] = {e.x, e.y, e.z, e.w};
            const float mv[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
            for (int g = 0; g < SPG; ++g) one_step(ev + g * DC, mv + g * DC);
            e = e1;
            m4 = m4n;
            const float4* nptr = np + (int64_t)min(r + 2, d.R - 1) * 64;
            asm volatile("" : "+v"(nptr) : "v"(acc.a));  // the address "depends" on this iteration's last cost
            e1 = *nptr;
        }
    }
    if (t < d.T) {  // ragged last group (T*dc not a multiple of 4)
        const float ev[4] = {e.x, e.y, e.z, e.w};
        const float mv[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
        for (int g = 0; g < SPG; ++g)
            if (t < d.T) one_step(ev + g * DC, mv + g * DC);
    }
    // terminal cost: zero action, stale prev_action U[:, max(T-2,0)] and stale t = T-1
    // (mppi.py:318-328); knext already holds the constants of row T-1
    float zero[DC];
#pragma unroll
    for (int k = 0; k < DC; ++k) zero[k] = 0.0f;
    const float term = M::cost(ctx, knext, s, zero, pl, bad);
    return acc.total(term);
}

// Total cost of one lane's trajectory: picks the launch-uniform copy of the horizon loop (racing: unit wheel base; a start
// outside the position clamp) and — for the models whose fast paths have per-lane validity ranges (pendulum, cart-poles,
// mountain car, goal zone) — redoes a lane that left one with the library math.  Racing and nav2d take any finite start
// (EntryGeneral) and carry no redo: inlining the library-math walk next to the hot loop cost the racing kernel 18 VGPRs,
// two waves per SIMD and 3.6 % of its time (profiles/r04_experiments.md).
template <int MODEL, int FAST, bool GEN, bool UC>
__device__ __forceinline__ float lane_cost(const float4* __restrict__ np, uint64_t gi, const GenCtx& gen, const float4* mp,
                                           const float* s_ktab, const float* __restrict__ x0, const Dims& d,
                                           const ModelCtx& ctx) {
    using M = ModelT<MODEL, FAST>;
    bool bad = false;
    float total;
    if constexpr (FAST != 0 && EntryGeneral<M>::value) {
        if (!M::start_in_box(ctx, x0))  // launch-uniform (x0 is shared): the copy whose first stage cost is bounds-tested
            return trajectory_cost<MODEL, FAST, GEN, UC, false, true>(np, gi, gen, mp, s_ktab, x0, d, ctx, bad);
    }
    // (racing, fast math: the unit wheel base of the reference is a launch-uniform branch around two copies of the loop)
    if (MODEL == MPPI_MODEL_RACING && FAST != 0 && ctx.unit_L)
        total = trajectory_cost<MODEL, FAST, GEN, UC, true>(np, gi, gen, mp, s_ktab, x0, d, ctx, bad);
    else
        total = trajectory_cost<MODEL, FAST, GEN, UC>(np, gi, gen, mp, s_ktab, x0, d, ctx, bad);
    if constexpr (FAST != 0 && !EntryGeneral<M>::value) {
        if (bad) {  // a fast path left its validity range: redo this lane with the library math
            bool ignore = false;
            total = trajectory_cost<MODEL, 0, GEN, false>(np, gi, gen, mp, s_ktab, x0, d, ctx, ignore);
        }
    }
    return total;
}

template <int MODEL, int FAST>  // (defined with the solve's tail below)
__device__ __forceinline__ void batch1_rollout(const ModelCtx& ctx, const float* s_x0, const float* s_act, int T,
                                               float* __restrict__ state_out);

#ifndef MPPI_ROLLOUT_ATTR
#define MPPI_ROLLOUT_ATTR  // e.g. __attribute__((amdgpu_waves_per_eu(8))) for occupancy experiments
#endif
template <int MODEL, int FAST, bool GEN, bool UC>
__global__ __launch_bounds__(BLOCK) MPPI_ROLLOUT_ATTR void rollout_cost_kernel(const float4* __restrict__ noise,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ x0,
                                                             float* __restrict__ costs,
                                                             unsigned* __restrict__ min_key,
                                                             unsigned* __restrict__ next_min_key,
                                                             float* __restrict__ mean_used,
                                                             float* __restrict__ x0_used, Dims d, GenCtx gen,
                                                             ModelCtx ctx, const float* __restrict__ b1_in,
                                                             float* __restrict__ b1_state_out) {
    using M = ModelT<MODEL, FAST>;
    __shared__ float s_min[BLOCK / WAVE];
    // [4*R] mean groups, [4*R] zeros (samples that do not inherit the mean), then [T*KROW] step rows
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    // One extra block (the last) when the PREVIOUS solve left its state sequence pending (option "lazy_state_seq"): the
    // batch-1 rollout of that solution (mppi.py:448-449) from the inputs finalize_kernel left in b1_in — T dependent steps
    // of one wave, hidden behind this launch's N-sample rollout instead of extending the previous solve's tail.
    if (b1_state_out != nullptr && blockIdx.x == gridDim.x - 1) {
        for (int i = threadIdx.x; i < d.row + M::DS; i += BLOCK) s_dyn[i] = b1_in[i];
        __syncthreads();
        batch1_rollout<MODEL, FAST>(ctx, s_dyn + d.row, s_dyn, d.T, b1_state_out);
        return;
    }
#ifdef MPPI_AB_VGPR_FLOOR  // (A/B knob of scripts/build_variant.sh: same code at the occupancy of an 85-VGPR build)
    asm volatile("; vgpr floor" ::: "v84");
#endif
    // the state this solve starts from outlives the caller's buffer (mppi_bind_state is zero-copy): later
    // re-rolls of this solve's samples (get_top_samples, _state_seq_batch) read the snapshot
    if (blockIdx.x == 0 && threadIdx.x < M::DS) x0_used[threadIdx.x] = x0[threadIdx.x];
    float4* s_mean4 = reinterpret_cast<float4*>(s_dyn);
    float* s_ktab = s_dyn + 8 * d.R;
    for (int f = threadIdx.x; f < 4 * d.R; f += BLOCK) {
        const float m = f < d.row ? mean[f] : 0.0f;
        s_dyn[f] = m;
        s_dyn[4 * d.R + f] = 0.0f;
        // the mean this solve samples around outlives the warm-start update (get_top_samples re-rolls with it)
        if (blockIdx.x == 0 && f < d.row) mean_used[f] = m;
    }
    for (int f = threadIdx.x; f < d.T * M::KROW; f += BLOCK) s_ktab[f] = ctx.ref[f];
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t tile = (int64_t)blockIdx.x * (BLOCK / WAVE) + wid;
    // the minimum key is double-buffered: this launch accumulates into `min_key` (reset by the
    // previous launch) and resets the other slot for the next one -> no memset between solves
    if (blockIdx.x == 0 && threadIdx.x == 0) *next_min_key = 0xFFFFFFFFu;
    float total = INFINITY;
    if (tile < d.tiles) {
        const int64_t i = tile * 64 + lane;
        const uint64_t gi = (uint64_t)(d.sample_offset + i);
        const bool inherit = (d.sample_offset + i) < d.inherit_count;
        const float4* np = noise + tile * d.R * 64 + lane;
        bool bad = false;
        const float4* mp = inherit ? s_mean4 : s_mean4 + d.R;
        total = lane_cost<MODEL, FAST, GEN, UC>(np, gi, gen, mp, s_ktab, x0, d, ctx);
        if (i < d.N) costs[i] = total;
        else total = INFINITY;
    }
    const float wm = wave_min(total);
    if (lane == 0) s_min[wid] = wm;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = s_min[0];
#pragma unroll
        for (int w = 1; w < BLOCK / WAVE; ++w) m = fminf(m, s_min[w]);
        if (m < INFINITY) atomicMin(min_key, float_to_key(m));
    }
}

// ------------------------------------------------------------------------------------------
// The north star's literal mapping, kept for comparison (mppi_set_option("mapping", 1)): ONE WAVEFRONT
// PER TRAJECTORY.  The wave loads the trajectory's [T*dc] noise row from the reference layout
// [N][T][dc] with coalesced float4 loads and stages U = clamp(mean + eps) in LDS; lane 0 walks the
// serial recurrence S[t+1] = f(S[t], U[t]) writing the states to LDS (63 lanes idle: the recurrence
// cannot be spread over lanes); then lane t evaluates the stage cost of step t (lane T the terminal
// cost) and a wavefront shuffle reduction sums them.  Same model functors, same results up to the
// summation order of the T+1 stage costs.  Measured 20x slower than the lane-per-trajectory mapping
//  because the recurrence runs on 1/64 of the machine.
template <int MODEL, int FAST>
__global__ __launch_bounds__(BLOCK) void rollout_cost_wave_kernel(const float* __restrict__ eps_std,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ x0,
                                                                  float* __restrict__ costs,
                                                                  unsigned* __restrict__ min_key,
                                                                  unsigned* __restrict__ next_min_key, Dims d,
                                                                  ModelCtx ctx) {
    using M = ModelT<MODEL, FAST>;
    using K = typename M::K;
    constexpr int DS = M::DS, DC = M::DC, NW = BLOCK / WAVE;
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];  // per wave: U[row4] then S[(T+1)*DS]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int row4 = 4 * d.R;
    float* sU = s_dyn + (size_t)wid * (row4 + (d.T + 1) * DS);
    float* sS = sU + row4;
    if (blockIdx.x == 0 && threadIdx.x == 0) *next_min_key = 0xFFFFFFFFu;
    float wmin = INFINITY;
    const int64_t nwaves = (int64_t)gridDim.x * NW;
    for (int64_t i = (int64_t)blockIdx.x * NW + wid; i < d.N; i += nwaves) {
        const bool inherit = (d.sample_offset + i) < d.inherit_count;  // wave-uniform
        const float* erow = eps_std + i * d.row;
        for (int f = lane; f < d.row; f += WAVE) {  // coalesced row load, clamp, stage in LDS
            const float m = inherit ? mean[f] : 0.0f;
            sU[f] = clampf(m + erow[f], d.u_min[f % DC], d.u_max[f % DC]);
        }
        __builtin_amdgcn_wave_barrier();
        bool bad = false;
        if (lane == 0) {  // the serial recurrence: one lane
            float s[DS];
#pragma unroll
            for (int j = 0; j < DS; ++j) s[j] = x0[j];
            if (FAST) M::check_state(ctx, s, bad);
            for (int t = 0; t < d.T; ++t) {
                float u[DC], sn[DS], ss[DS];
#pragma unroll
                for (int k = 0; k < DC; ++k) u[k] = sU[t * DC + k];
                M::step(ctx, s, u, sn, ss, bad, false);
#pragma unroll
                for (int j = 0; j < DS; ++j) { sS[t * DS + j] = ss[j]; s[j] = sn[j]; }
            }
#pragma unroll
            for (int j = 0; j < DS; ++j) sS[d.T * DS + j] = s[j];
        }
        __builtin_amdgcn_wave_barrier();
        float part = 0.0f;
        for (int t = lane; t <= d.T; t += WAVE) {  // time-parallel stage costs
            float st[DS], u[DC], pu[DC];
#pragma unroll
            for (int j = 0; j < DS; ++j) st[j] = sS[t * DS + j];
            const bool term = t == d.T;
            const int tp = term ? max(d.T - 2, 0) : max(t - 1, 0);
#pragma unroll
            for (int k = 0; k < DC; ++k) { u[k] = term ? 0.0f : sU[t * DC + k]; pu[k] = sU[tp * DC + k]; }
            const K kk = M::load_k(ctx.ref, term ? d.T - 1 : t);
            part += M::cost(ctx, kk, st, u, pu, bad);
        }
        const float total = wave_sum(part);
        if (lane == 0) { costs[i] = total; wmin = fminf(wmin, total); }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0 && wmin < INFINITY) atomicMin(min_key, float_to_key(wmin));
}

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
template <int GPW, bool GEN, bool WIDE = false, int CHAINS = 2>  // GPW: float4 groups per wave and column chunk (8: the host launches ceil(R / 32) chunks)
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
                    if constexpr (CHAINS == 4) {
                        if (full >= 8) { quad(0); quad(4); }
                        else if (full >= 4) {
                            quad(0);
                            if (full >= 6) { pair(4); if (full >= 7) single(6); }
                            else if (full >= 5) single(4);
                        } else {
                            if (full >= 2) { pair(0); if (full >= 3) single(2); }
                            else if (full >= 1) single(0);
                        }
                    } else {
                        (void)quad;
#pragma unroll
                        for (int m0 = 0; m0 < GPW; m0 += 2) {
                            if (full >= m0 + 2) pair(m0);
                            else if (full >= m0 + 1) single(m0);
                        }
                    }
                } else {
#pragma unroll
                    for (int m = 0; m < GPW; ++m)
                        if (m < full) accumulate(m, noise_group<false>(np, r0 + wid + NW * m, gi, gen, d));
                }
                // this wave's share of the remainder groups (see `accx` above)
                const bool mine = rem == 1 ? ((int)tile & 3) == wid : rem == 2 ? ((int)tile & 1) == (wid >> 1) : wid < rem;
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
        if (rem) {
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

// ------------------------------------------------------------------------------------------
// One-shot peer-to-peer exchange of the shard summaries (the sharded solve's only exchange) without a collective
// launch: every rank stores its summary straight into all peers' exchange buffers over xGMI and the consumer
// polls its own buffer.  Cells are 8 bytes {fp32 value, 32-bit sequence number} written with ONE store, so data and
// "ready" flag cannot be seen apart (the idea of RCCL's low-latency protocol): no fence ordering is relied on.
// Buffer of rank r (fine-grained device memory, IPC-mapped into every peer): cells[2][W][lenp]; solve `seq` uses
// parity seq & 1 — a rank can be at most one solve ahead of the slowest one, because its next finalize needs
// everybody's summary of that solve.
struct P2pCtx {
    unsigned long long* const* peers;  // [W] base of every rank's buffer as mapped here (device array)
    unsigned long long* local;         // this rank's buffer
    int* error;                        // mapped host flag: set when a poll timed out
    int world, rank, lenp;
    unsigned seq;                      // 0 = exchange off
};

__device__ __forceinline__ void p2p_store(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long p2p_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Ascending list of the blocks that published a partial row (heads[b][3] != 0), built by a whole block of
// NT threads: per-wave ballots, wave counts through LDS, exclusive prefix.  Returns the list length.
template <int NT>
__device__ __forceinline__ int compact_live_rows(const float* __restrict__ heads, int nblocks,
                                                 unsigned short* __restrict__ s_list, int* __restrict__ s_wcnt) {
    constexpr int NWV = NT / WAVE;
    constexpr int MAXCH = REDUCE_MAX_BLOCKS / NT;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nch = (nblocks + NT - 1) / NT;
    unsigned long long mine = 0ull;  // bit ch: this thread's block of chunk ch is live
    for (int ch = 0; ch < nch; ++ch) {
        const int bb = ch * NT + threadIdx.x;
        const bool f = bb < nblocks && heads[(int64_t)bb * 4 + 3] != 0.0f;
        const unsigned long long mask = __ballot(f);
        if (f) mine |= 1ull << ch;
        if (lane == 0) s_wcnt[ch * NWV + wv] = __popcll(mask);
    }
    __syncthreads();
    int nlive = 0;
    for (int ch = 0; ch < nch; ++ch) {
        int off = 0;
        for (int w = 0; w < nch * NWV; ++w) {
            const int cnt = s_wcnt[w];
            if (w < ch * NWV + wv) off += cnt;
            if (ch == 0) nlive += cnt;
        }
        const bool f = (mine >> ch) & 1ull;
        const unsigned long long mask = __ballot(f);
        if (f) s_list[off + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)(ch * NT + threadIdx.x);
    }
    __syncthreads();
    static_assert(MAXCH <= 64, "chunk bitmask");
    return nlive;
}

// Sum the per-block partial rows into the shard summary {min c, sum e, sum e^2, sum e*c, A[row]}.  Only blocks
// that saw a live tile published a row (heads[b][3]); every block of this kernel first compacts the ascending
// list of those rows, then thread (c = tid & 15, g = tid >> 4) of block x sums list entries g, g+64, ... of
// column 16x + c (64 B coalesced row segments, 8 loads in flight) and the 64 row groups combine through LDS.
// The last block folds the three scalar heads.  Deterministic (fixed order).  With a sharp softmax the list
// holds a handful of rows and the kernel is launch-latency only.
constexpr int SUM_COLS = 16;
constexpr int SUM_BLOCK = 1024;
__global__ __launch_bounds__(SUM_BLOCK) void summarize_kernel(const float* __restrict__ partials,
                                                          const float* __restrict__ heads,
                                                          const unsigned* __restrict__ min_key, int nblocks,
                                                          int colsp, int row, float* __restrict__ summary,
                                                          float* __restrict__ summary_copy,
                                                          int* __restrict__ nlive_out, P2pCtx p2p) {
    constexpr int NG = SUM_BLOCK / SUM_COLS;
    __shared__ float s_part[NG][SUM_COLS + 1];
    __shared__ unsigned short s_list[REDUCE_MAX_BLOCKS];
    __shared__ int s_wcnt[REDUCE_MAX_BLOCKS / WAVE];
    const int nlive = compact_live_rows<SUM_BLOCK>(heads, nblocks, s_list, s_wcnt);
    const int c = threadIdx.x & (SUM_COLS - 1), g = threadIdx.x / SUM_COLS;
    const bool head_block = blockIdx.x == gridDim.x - 1;
    float a[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = 0.f;
    const int col = blockIdx.x * SUM_COLS + c;
    const bool active = head_block ? c < 3 : col < colsp;
    const float* base = head_block ? heads + c : partials + col;
    const int64_t ld = head_block ? 4 : colsp;
    if (active) {
        for (int k = g; k < nlive; k += 8 * NG) {  // 8 independent loads in flight per thread
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int kk = k + q * NG;
                if (kk < nlive) a[q] += base[(int64_t)s_list[kk] * ld];
            }
        }
    }
    s_part[g][c] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (threadIdx.x < SUM_COLS) {
        float v = 0.f;
        for (int q = 0; q < NG; ++q) v += s_part[q][threadIdx.x];
        int dst = -1;
        if (!head_block) {
            const int cc = blockIdx.x * SUM_COLS + threadIdx.x;
            if (cc < row) dst = MPPI_SUMMARY_HEAD + cc;
        } else {
            if (threadIdx.x < 3) dst = 1 + threadIdx.x;
            if (threadIdx.x == 3) {
                dst = 0;
                v = key_to_float(*min_key);
                if (nlive_out) *nlive_out = nlive;
            }
        }
        if (dst >= 0) {
            summary[dst] = v;
            if (summary_copy) summary_copy[dst] = v;
            if (p2p.seq) {  // cells are self-contained: every block hands its own columns to the peers right away
                const size_t slot = ((size_t)(p2p.seq & 1u) * p2p.world + p2p.rank) * p2p.lenp + dst;
                const unsigned long long cell = ((unsigned long long)p2p.seq << 32) | (unsigned long long)__float_as_uint(v);
                for (int w = 0; w < p2p.world; ++w) p2p_store(p2p.peers[w] + slot, cell);
            }
        }
    }
}

// One trajectory rolled out from explicit actions (reference layout row) or from noise, writing the
// states the reference would leave in its state buffer.  GETU(t, u) fills the action of step t.
template <int MODEL, int FAST, class GETU>
__device__ __forceinline__ bool rollout_states(const float* __restrict__ x0, int T, const ModelCtx& ctx,
                                               float* __restrict__ out, GETU getu) {
    using M = ModelT<MODEL, FAST>;
    constexpr int DS = M::DS, DC = M::DC;
    bool bad = false;
    float s[DS];
#pragma unroll
    for (int j = 0; j < DS; ++j) s[j] = x0[j];
    if constexpr (FAST != 0 && EntryGeneral<M>::value) {
        // any finite start (see Model::enter_any): the heading is wrapped once by the reference's own operation and every
        // later one is a fixed point of the wrap; row 0 keeps the caller's state as given (mppi.py:280-283: S[:, 0] = x0)
        const float raw_heading = s[2];
        M::enter_any(s);
        for (int t = 0; t < T; ++t) {
            float u[DC], sn[DS], ss[DS];
            getu(t, u);
            M::step(ctx, s, u, sn, ss, bad, false, true);
            if (t == 0) ss[2] = raw_heading;
#pragma unroll
            for (int j = 0; j < DS; ++j) { out[t * DS + j] = ss[j]; s[j] = sn[j]; }
        }
        if (T == 0) s[2] = raw_heading;
    } else {
        if (FAST) M::check_state(ctx, s, bad);
        for (int t = 0; t < T; ++t) {
            float u[DC], sn[DS], ss[DS];
            getu(t, u);
            M::step(ctx, s, u, sn, ss, bad);
#pragma unroll
            for (int j = 0; j < DS; ++j) { out[t * DS + j] = ss[j]; s[j] = sn[j]; }
        }
    }
#pragma unroll
    for (int j = 0; j < DS; ++j) out[T * DS + j] = s[j];
    return bad;
}
template <int MODEL, int FAST, class GETU>
__device__ __forceinline__ void rollout_states_checked(const float* __restrict__ x0, int T, const ModelCtx& ctx,
                                                       float* __restrict__ out, GETU getu) {
    const bool bad = rollout_states<MODEL, FAST>(x0, T, ctx, out, getu);
    if constexpr (FAST != 0 && !EntryGeneral<ModelT<MODEL, FAST>>::value) {  // (EntryGeneral models cannot leave a fast path)
        if (bad) (void)rollout_states<MODEL, 0>(x0, T, ctx, out, getu);
    }
}

__global__ __launch_bounds__(BLOCK) void p2p_publish_kernel(const float* __restrict__ summary, int len, P2pCtx x) {
    const size_t slot = ((size_t)(x.seq & 1u) * x.world + x.rank) * x.lenp;
    for (int j = threadIdx.x; j < len; j += BLOCK) {
        const unsigned long long cell = ((unsigned long long)x.seq << 32) | (unsigned long long)__float_as_uint(summary[j]);
        for (int w = 0; w < x.world; ++w) p2p_store(x.peers[w] + slot + j, cell);
    }
}

// Block-wide: wait for the `len` cells of every rank of solve x.seq and unpack them to out[w * stride + j].
// Polls give up after ~20 s of wall clock (100 MHz counter) and raise *x.error; the caller's results are then void.
template <int NT>
__device__ __forceinline__ void p2p_collect(const P2pCtx& x, int len, float* __restrict__ out, int stride,
                                            int* __restrict__ s_timed_out = nullptr) {
    const long long t0 = wall_clock64();
    for (int idx = threadIdx.x; idx < x.world * len; idx += NT) {
        const int w = idx / len, j = idx - w * len;
        const unsigned long long* cellp = x.local + ((size_t)(x.seq & 1u) * x.world + w) * x.lenp + j;
        unsigned long long cell = p2p_load(cellp);
        unsigned spins = 0;
        while ((unsigned)(cell >> 32) != x.seq) {
            if ((++spins & 1023u) == 0u && wall_clock64() - t0 > 2000000000ll) {
                *x.error = 1;
                if (s_timed_out) *s_timed_out = 1;  // (LDS) the block voids this solve's outputs
                break;
            }
            __builtin_amdgcn_s_sleep(2);
            cell = p2p_load(cellp);
        }
        out[w * stride + j] = __uint_as_float((unsigned)cell);
    }
    __syncthreads();
}

// self-test / generic use: collect into a plain device array [W][len]
__global__ __launch_bounds__(BLOCK) void p2p_collect_kernel(P2pCtx x, int len, float* __restrict__ out) {
    p2p_collect<BLOCK>(x, len, out, len);
}

// Step 8 (mppi.py:448-449,508-524): the batch-1 rollout of the solution `s_act` [T][dc] from `s_x0`, by the calling
// block's first wave (racing / fast math: spread over the wave, see Model::rollout_wave; else lane 0 walks the T steps).
// Shared by finalize_tail (in the solve's last kernel), state_seq_kernel and the extra block of rollout_cost_kernel (the same
// rollout completed lazily): one code path, so all of them produce the same bits.
template <int MODEL, int FAST>
__device__ __forceinline__ void batch1_rollout(const ModelCtx& ctx, const float* s_x0, const float* s_act, int T,
                                               float* __restrict__ state_out) {
    constexpr int DC = ModelT<MODEL, FAST>::DC;
    const auto getu = [&](int t, float* u) {
#pragma unroll
        for (int k = 0; k < DC; ++k) u[k] = s_act[t * DC + k];
    };
    if constexpr (MODEL == MPPI_MODEL_RACING && FAST) {
        if (T <= 63) {  // the serial part of the batch-1 rollout shrinks to the heading/speed recurrences
            if (threadIdx.x >= WAVE) return;
            ModelT<MODEL, FAST>::rollout_wave(ctx, s_x0, s_act, T, state_out);  // (any finite start: no library-math redo)
            return;
        }
    }
    if (threadIdx.x == 0) rollout_states_checked<MODEL, FAST>(s_x0, T, ctx, state_out, getu);
}

// The same rollout as its own one-wave kernel: `b1_in` = [row] action sequence, then [ds] start state, left behind by
// finalize_kernel (its `b1_out`) under option "lazy_state_seq".  The 50 dependent steps are not needed by anything on the
// solve's critical path (the next solve samples around the mean, env.step applies a[0]): they normally ride in an extra
// block of the NEXT rollout launch (rollout_cost_kernel), and this kernel runs only when somebody reads the state
// sequence before that (mppi_join_state_seq).
template <int MODEL, int FAST>
__global__ __launch_bounds__(WAVE) void state_seq_kernel(const float* __restrict__ b1_in, int row, int T,
                                                         float* __restrict__ state_out, ModelCtx ctx) {
    constexpr int DS = ModelT<MODEL, FAST>::DS;
    extern __shared__ __attribute__((aligned(16))) float s_b1[];  // [row] action, [DS] start state
    for (int i = threadIdx.x; i < row + DS; i += WAVE) s_b1[i] = b1_in[i];
    __syncthreads();
    batch1_rollout<MODEL, FAST>(ctx, s_b1 + row, s_b1, T, state_out);
}

constexpr int FIN_BLOCK = 1024;
// The tail of a solve once the shard summaries are at hand (block-wide, FIN_BLOCK threads): combine the shards, normalise,
// Savitzky-Golay step, warm start, outputs, batch-1 rollout.  Shared by finalize_kernel and solve_fused_kernel.
// s_act [row] and s_yp (filter staging) are LDS; `summaries` may be LDS or global.
template <int MODEL, int FAST>
__device__ __forceinline__ void finalize_tail(const float* summaries, int num_shards, float lambda, int row, int T,
                                              const float* s_x0, float* s_act, float* s_yp,
                                              float* __restrict__ mean_store, float* __restrict__ action_out,
                                              float* __restrict__ state_out, float* __restrict__ stats_out,
                                              float* __restrict__ stats_keep, const SgFilter& sg, const ModelCtx& ctx,
                                              float* __restrict__ b1_out = nullptr, float* __restrict__ poison_out = nullptr) {
    const int stride = MPPI_SUMMARY_HEAD + row;
    float xmax = -INFINITY, cmin = INFINITY;
    for (int g = 0; g < num_shards; ++g) {
        const float m = summaries[(int64_t)g * stride];
        xmax = fmaxf(xmax, (-m) / lambda);
        cmin = fminf(cmin, m);
    }
    float se = 0.f, se2 = 0.f, sec = 0.f;
    for (int g = 0; g < num_shards; ++g) {
        const float* sm = summaries + (int64_t)g * stride;
        const float f = expf((-sm[0]) / lambda - xmax);
        se = fmaf(f, sm[1], se);
        se2 = fmaf(f * f, sm[2], se2);
        sec = fmaf(f, sm[3], sec);
    }
    for (int cidx = threadIdx.x; cidx < row; cidx += FIN_BLOCK) {
        float a = 0.f;
        for (int g = 0; g < num_shards; ++g) {
            const float* sm = summaries + (int64_t)g * stride;
            const float f = expf((-sm[0]) / lambda - xmax);
            a = fmaf(f, sm[MPPI_SUMMARY_HEAD + cidx], a);
        }
        a = a / se;
        s_act[cidx] = a;
        if (sg.window == 0) {
            if (action_out) action_out[cidx] = a;
            if (mean_store) mean_store[cidx] = a;
        }
    }
    if (threadIdx.x == 0) {
        if (stats_out) { stats_out[0] = cmin; stats_out[1] = se; stats_out[2] = se2; stats_out[3] = sec; }
        stats_keep[0] = cmin; stats_keep[1] = se; stats_keep[2] = se2; stats_keep[3] = sec;
        stats_keep[4] = lambda;  // the temperature these weights used (later queries: get_top_samples, _weights)
    }
    __syncthreads();
    if (sg.window > 0) {
        // Step 7 (mppi.py:423-443,598-620): Savitzky-Golay smoothing of [history(T-1); a(T)] per control dimension,
        // symmetric-flip padding by w/2, valid cross-correlation accumulated tap by tap in fp32 (the operation
        // order of the host statement in pi_mpc/_host.py), keep the last T; then shift a'[0] into the history.
        const int dcn = row / T, p = sg.window / 2, n = 2 * T - 1;
        for (int idx = threadIdx.x; idx < n * dcn; idx += FIN_BLOCK) {
            const int i = idx / dcn, k = idx - i * dcn;
            s_yp[(p + i) * dcn + k] = i < T - 1 ? sg.history[i * dcn + k] : s_act[(i - (T - 1)) * dcn + k];
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < p * dcn; idx += FIN_BLOCK) {
            const int j = idx / dcn, k = idx - j * dcn;
            s_yp[(p - 1 - j) * dcn + k] = s_yp[(p + j) * dcn + k];                  // front: y[p-1], ..., y[0]
            s_yp[(p + n + j) * dcn + k] = s_yp[(p + n - 1 - j) * dcn + k];          // back:  y[n-1], ..., y[n-p]
        }
        __syncthreads();
        float filt = 0.0f;
        const int cidx = threadIdx.x;  // row <= FIN_BLOCK is checked on the host for the filter
        if (cidx < row) {
            const int t = cidx / dcn, k = cidx - t * dcn;
            for (int j = 0; j < sg.window; ++j) filt = filt + s_yp[(T - 1 + t + j) * dcn + k] * sg.coeffs[j];
        }
        __syncthreads();
        if (cidx < row) {
            s_act[cidx] = filt;
            if (action_out) action_out[cidx] = filt;
            if (mean_store) mean_store[cidx] = filt;
        }
        for (int idx = threadIdx.x; idx < (T - 1) * dcn; idx += FIN_BLOCK) {  // history <- [history[1:]; a'[0]]
            const int i = idx / dcn, k = idx - i * dcn;
            sg.history[idx] = i < T - 2 ? s_yp[(p + i + 1) * dcn + k] : 0.0f;
        }
        __syncthreads();
        if (threadIdx.x < dcn && T >= 2) sg.history[(T - 2) * dcn + threadIdx.x] = s_act[threadIdx.x];
        __syncthreads();
    }
    if (b1_out) {  // the batch-1 rollout is deferred to state_seq_kernel: leave its inputs behind (final action, start state)
        for (int cidx = threadIdx.x; cidx < row; cidx += FIN_BLOCK) b1_out[cidx] = s_act[cidx];
        if (threadIdx.x < ModelT<MODEL, FAST>::DS) b1_out[row + threadIdx.x] = s_x0[threadIdx.x];
        // ... and void the caller's buffer until the rollout lands in it: a reader that bypasses the join (a raw pointer
        // handed to another library, a different stream) sees NaN, not the previous solve's states or uninitialised memory
        if (poison_out)
            for (int c = threadIdx.x; c < (T + 1) * ModelT<MODEL, FAST>::DS; c += FIN_BLOCK) poison_out[c] = __uint_as_float(0x7fc00000u);
    }
    if (!state_out) return;
    batch1_rollout<MODEL, FAST>(ctx, s_x0, s_act, T, state_out);
}

// Combine shard summaries, normalise, store the warm start, roll the result out with batch 1
// (mppi.py:381-385,448-452,508-524).
// `summaries` != nullptr: `num_shards` summary vectors (the all_gathered shards, or this handle's own summary
// from summarize_kernel).  `summaries` == nullptr: the kernel first folds this handle's published partial rows
// itself — no summarize launch; with a sharp softmax that is a handful of rows.  The fold uses summarize_kernel's
// summation tree (64 row groups x 8 interleaved accumulators per column over the ascending live list, then the
// groups in order), so the summary is bit-identical whichever of the two paths the host picks.  The summary is
// also written to `summary_out` for later readers and the number of live rows to `nlive_out` (mapped host memory:
// the host's hint for the next solve).  A timed-out peer-to-peer poll voids the outputs (NaN) instead of
// returning a partial combine.
template <int MODEL, int FAST>
__global__ __launch_bounds__(FIN_BLOCK) void finalize_kernel(const float* __restrict__ summaries, int num_shards,
                                                             const float* __restrict__ partials,
                                                             const float* __restrict__ heads,
                                                             const unsigned* __restrict__ min_key, int nblocks,
                                                             int colsp, float* __restrict__ summary_out,
                                                             int* __restrict__ nlive_out, float lambda_arg,
                                                             const float* __restrict__ lambda_dev, int row, int T,
                                                             const float* __restrict__ x0,
                                                             float* __restrict__ mean_store,
                                                             float* __restrict__ action_out,
                                                             float* __restrict__ state_out,
                                                             float* __restrict__ stats_out,
                                                             float* __restrict__ stats_keep, SgFilter sg,
                                                             P2pCtx p2p, ModelCtx ctx, float* __restrict__ b1_out,
                                                             float* __restrict__ poison_out) {
    const float lambda = lambda_dev ? *lambda_dev : lambda_arg;
    // issued before the first barrier so that their latency hides behind the fold: the shard minimum and the start
    // state of the batch-1 rollout (both would otherwise be dependent loads at the end of the chain)
    const unsigned min_key_now = *min_key;
    __shared__ float s_x0[MPPI_MAX_DIM_STATE];
    if (threadIdx.x < ModelT<MODEL, FAST>::DS) s_x0[threadIdx.x] = x0[threadIdx.x];
    // [row] action, [max(1, W) * (4 + row)] own / collected summaries, then (SG filter) [(2T-1+2*(w/2))*dc]
    extern __shared__ __attribute__((aligned(16))) float s_fin[];
    const int stride = MPPI_SUMMARY_HEAD + row;
    float* s_act = s_fin;
    float* s_sum = s_fin + row;
    float* s_yp = s_sum + (p2p.seq ? p2p.world : 1) * stride;
    if (p2p.seq) {  // the shards' summaries arrive through the peer-to-peer exchange buffer
        __shared__ int s_timed_out;
        if (threadIdx.x == 0) s_timed_out = 0;
        __syncthreads();
        p2p_collect<FIN_BLOCK>(p2p, stride, s_sum, stride, &s_timed_out);
        if (s_timed_out) {  // a rank is missing or stalled: no partial answer leaves this kernel
            const float nanv = __uint_as_float(0x7fc00000u);
            constexpr int DSN = ModelT<MODEL, FAST>::DS;
            for (int c = threadIdx.x; c < row; c += FIN_BLOCK) if (action_out) action_out[c] = nanv;
            for (int c = threadIdx.x; c < (T + 1) * DSN; c += FIN_BLOCK) if (state_out) state_out[c] = nanv;
            for (int c = threadIdx.x; c < row + DSN; c += FIN_BLOCK) if (b1_out) b1_out[c] = nanv;  // (a lazily completed state sequence is void too)
            for (int c = threadIdx.x; c < (T + 1) * DSN; c += FIN_BLOCK) if (poison_out) poison_out[c] = nanv;
            if (threadIdx.x < 4 && stats_out) stats_out[threadIdx.x] = nanv;
            return;
        }
        summaries = s_sum;
        num_shards = p2p.world;
    } else if (summaries == nullptr) {
        constexpr int NG = SUM_BLOCK / SUM_COLS;  // 64 row groups: summarize_kernel's tree
        __shared__ unsigned short s_list[REDUCE_MAX_BLOCKS];
        __shared__ int s_wcnt[REDUCE_MAX_BLOCKS / WAVE];
        const int nlive = compact_live_rows<FIN_BLOCK>(heads, nblocks, s_list, s_wcnt);
        const int ncols = row + 3;  // the last 3 "columns" are the heads
        // [NG][ncols] group sums, behind the filter staging (the host sizes the dynamic LDS for it)
        float* s_fold = s_yp + (sg.window ? (2 * T - 1 + 2 * (sg.window / 2)) * (row / T) : 0);
        // row groups g >= nlive hold no row: their sums are +0 and adding them changes nothing, so only the first
        // min(NG, nlive) groups are formed and summed (one pass of row + 3 threads when one or two blocks were live)
        const int ng = min(NG, nlive);
        for (int p = threadIdx.x; p < ng * ncols; p += FIN_BLOCK) {
            const int g = p / ncols, col = p - g * ncols;  // consecutive lanes: consecutive columns of one row
            const bool is_head = col >= row;
            const float* base = is_head ? heads + (col - row) : partials + col;
            const int64_t ld = is_head ? 4 : colsp;
            float a[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = 0.f;
            for (int k = g; k < nlive; k += 8 * NG) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int kk = k + q * NG;
                    if (kk < nlive) a[q] += base[(int64_t)s_list[kk] * ld];
                }
            }
            s_fold[p] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        }
        __syncthreads();
        for (int col = threadIdx.x; col < ncols; col += FIN_BLOCK) {
            float v = 0.f;
            for (int q = 0; q < ng; ++q) v += s_fold[q * ncols + col];
            const int dst = col < row ? MPPI_SUMMARY_HEAD + col : 1 + (col - row);
            s_sum[dst] = v;
            if (summary_out) summary_out[dst] = v;
        }
        if (threadIdx.x == 0) {
            s_sum[0] = key_to_float(min_key_now);
            if (summary_out) summary_out[0] = s_sum[0];
            if (nlive_out) *nlive_out = nlive;
        }
        __syncthreads();
        summaries = s_sum;  // (generic address space: LDS)
        num_shards = 1;
    }
    finalize_tail<MODEL, FAST>(summaries, num_shards, lambda, row, T, s_x0, s_act, s_yp, mean_store, action_out, state_out,
                               stats_out, stats_keep, sg, ctx, b1_out, poison_out);
}

// Softmax statistics of the cost vector for one temperature — the device half of the auto-lambda
// searches (ESSPS / LBPS / MPO, mppi.py:341-370,387-398,526-566): the root-finders stay on the host
// and ask for {sum e, sum e^2, sum e*c, max c} with e_i = exp((-c_i)/lambda - (-cmin)/lambda), instead
// of pulling costs[N] over PCIe and running ~10-40 softmaxes on the CPU.  Two tiny launches
// (per-block partials, then a fixed-order combine written to mapped host memory): deterministic.
constexpr int STATS_BLOCKS = 256;
__global__ __launch_bounds__(BLOCK) void stats_partial_kernel(const float* __restrict__ costs, int64_t N,
                                                             const unsigned* __restrict__ min_key, float lambda_arg,
                                                             const float* __restrict__ lambda_dev /* nullable */,
                                                             float* __restrict__ part /*[STATS_BLOCKS][4]*/) {
    __shared__ float s_p[BLOCK / WAVE][4];
    const float lambda = lambda_dev ? *lambda_dev : lambda_arg;
    const float cmin = key_to_float(*min_key);
    const float xmax = (-cmin) / lambda;
    float se = 0.f, se2 = 0.f, sec = 0.f, cmax = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < N; i += (int64_t)gridDim.x * BLOCK) {
        const float c = costs[i];
        const float e = expf((-c) / lambda - xmax);
        se += e;
        se2 = fmaf(e, e, se2);
        sec = fmaf(e, c, sec);
        cmax = fmaxf(cmax, c);
    }
    se = wave_sum(se); se2 = wave_sum(se2); sec = wave_sum(sec);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) cmax = fmaxf(cmax, __shfl_xor(cmax, m));
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { s_p[wid][0] = se; s_p[wid][1] = se2; s_p[wid][2] = sec; s_p[wid][3] = cmax; }
    __syncthreads();
    if (threadIdx.x < 4) {
        float v = s_p[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < BLOCK / WAVE; ++w) v = threadIdx.x == 3 ? fmaxf(v, s_p[w][3]) : v + s_p[w][threadIdx.x];
        part[blockIdx.x * 4 + threadIdx.x] = v;
    }
}
__global__ __launch_bounds__(WAVE) void stats_combine_kernel(const float* __restrict__ part, int nblocks,
                                                            const unsigned* __restrict__ min_key,
                                                            double* __restrict__ out /*[5] mapped host*/) {
    double se = 0.0, se2 = 0.0, sec = 0.0;
    float cmax = -INFINITY;
    for (int b = threadIdx.x; b < nblocks; b += WAVE) {
        se += part[b * 4]; se2 += part[b * 4 + 1]; sec += part[b * 4 + 2];
        cmax = fmaxf(cmax, part[b * 4 + 3]);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        se += __shfl_xor(se, m); se2 += __shfl_xor(se2, m); sec += __shfl_xor(sec, m);
        cmax = fmaxf(cmax, __shfl_xor(cmax, m));
    }
    if (threadIdx.x == 0) {
        out[0] = (double)key_to_float(*min_key); out[1] = (double)cmax; out[2] = se; out[3] = se2; out[4] = sec;
    }
}

// The same statistics for up to STATS_L temperatures in one pass over the costs (a grid of lambdas
// for the bracketing search of ESSPS): part [blocks][STATS_L][3] = {sum e, sum e^2, sum e*c}.
//
// Mapping: a block of 1024 threads stages 1024 costs in LDS per round; thread (l = tid & 31, chunk = tid >> 5) then
// walks the 32 costs of its chunk for ITS temperature l (LDS broadcast reads: the 32 lanes of a half-wave share the
// address).  Every lane therefore owns one temperature and the cross-lane work at the end is one shuffle (the two
// half-waves) plus a 16-way sum through LDS — instead of 96 full wave reductions per thread when every lane carried
// all 32 temperatures (12.6 us -> launch-bound at N = 65 536, profiles/r02_visitA_c2_c5_dense_path.md).
// `lams` is a DEVICE array [STATS_L] (entries past the caller's count hold 1): the temperatures of the second ESSPS
// grid are produced on the device (essps_select_kernel) and never visit the host.
constexpr int STATS_L = 32;
constexpr int STATS_THREADS = 1024;
// One block's share: thread j < 96 returns the block's partial sum of column j (0 elsewhere); part_max as the kernel's.
struct StatsLds {
    float c[STATS_THREADS];
    float p[STATS_THREADS / WAVE][STATS_L][3];
    float mx[STATS_THREADS / WAVE];
};
__device__ __forceinline__ float stats_multi_block(const float* __restrict__ costs, int64_t N, float cmin,
                                                   const float* __restrict__ lams, float* __restrict__ part_max,
                                                   StatsLds& lds) {
    constexpr int NWV = STATS_THREADS / WAVE;
    float (&s_c)[STATS_THREADS] = lds.c;
    float (&s_p)[NWV][STATS_L][3] = lds.p;
    float (&s_mx)[NWV] = lds.mx;
    float cmaxv = -INFINITY;
    const int l = threadIdx.x & (STATS_L - 1), chunk = threadIdx.x >> 5;
    const float inv_lam = 1.0f / lams[l];
    float se = 0.0f, se2 = 0.0f, sec = 0.0f;
    for (int64_t base = (int64_t)blockIdx.x * STATS_THREADS; base < N; base += (int64_t)gridDim.x * STATS_THREADS) {
        __syncthreads();
        const int64_t i = base + threadIdx.x;
        // padding: a huge finite cost -> e = exp(-inf) = 0 and 0 * c = 0
        const float cv = i < N ? costs[i] : 3.0e38f;
        if (i < N) cmaxv = fmaxf(cmaxv, cv);
        s_c[threadIdx.x] = cv;
        __syncthreads();
        const float* cc = s_c + chunk * 32;
#pragma unroll 8
        for (int j = 0; j < 32; ++j) {
            const float c = cc[j];
            // exp(-(c - cmin)/lambda) with the reciprocal of lambda: this kernel only brackets the temperature (the
            // weights themselves use the reference's (-c)/lambda - max form); cmin - c is exact within a factor 2
            const float e = expf((cmin - c) * inv_lam);
            se += e;
            se2 = fmaf(e, e, se2);
            sec = fmaf(e, c, sec);
        }
    }
    se += __shfl_xor(se, 32); se2 += __shfl_xor(se2, 32); sec += __shfl_xor(sec, 32);  // the wave's two chunks
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane < STATS_L) { s_p[wid][lane][0] = se; s_p[wid][lane][1] = se2; s_p[wid][lane][2] = sec; }
    if (part_max) {  // (uniform) the LBPS objective needs the cost range
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) cmaxv = fmaxf(cmaxv, __shfl_xor(cmaxv, m));
        if (lane == 0) s_mx[wid] = cmaxv;
    }
    __syncthreads();
    if (part_max && threadIdx.x == 0) {
        float v = s_mx[0];
#pragma unroll
        for (int w = 1; w < NWV; ++w) v = fmaxf(v, s_mx[w]);
        part_max[blockIdx.x] = v;
    }
    float v = 0.0f;
    if (threadIdx.x < STATS_L * 3) {
#pragma unroll
        for (int w = 0; w < NWV; ++w) v += (&s_p[w][0][0])[threadIdx.x];
    }
    return v;
}
__global__ __launch_bounds__(STATS_THREADS) void stats_multi_kernel(const float* __restrict__ costs, int64_t N,
                                                                    const unsigned* __restrict__ min_key,
                                                                    const float* __restrict__ lams,
                                                                    float* __restrict__ part,
                                                                    float* __restrict__ part_max /* nullable: [blocks] max c */) {
    __shared__ StatsLds lds;
    const float v = stats_multi_block(costs, N, key_to_float(*min_key), lams, part_max, lds);
    if (threadIdx.x < STATS_L * 3) part[(int64_t)blockIdx.x * STATS_L * 3 + threadIdx.x] = v;
}
// Block-wide (960 of 1024 threads = 24 column quads x 40 row groups): column sums of part[nblocks][96] in double, fixed
// order -> out[96] (LDS or global).  The partial rows were written by other XCDs a moment ago, so every load is a
// trip to memory: one float4 per (row, quad) and up to eight rows per thread in flight make it ONE round of latency
// for up to 320 rows (a thread per (row group, column) with two loads in flight needed 13).  Ends with a barrier.
constexpr int STATS_COMB_THREADS = 960;
constexpr int STATS_COMB_GROUPS = 40;
// Where the partial rows come from: the array a statistics kernel wrote before this one started ...
struct PartRows {
    const float* __restrict__ part;
    static constexpr int K = 8;  // rows in flight per thread: 8 x 40 groups = one round of latency for up to 320 rows
    struct Raw { float4 v; };
    __device__ __forceinline__ void issue(int bb, int quad, Raw& r) const {
        r.v = *reinterpret_cast<const float4*>(part + (int64_t)bb * (STATS_L * 3) + 4 * quad);
    }
    __device__ __forceinline__ float4 finish(int, int, const Raw& r) const { return r.v; }
};
// ... or 8-byte {value, launch number} cells the blocks of THIS launch are still writing (relaxed agent-scope stores: data
// and readiness in one store, no fence — the hand-off of the single-launch solve): polled until the tag is this launch's.
struct CellRows {
    const unsigned long long* cells;  // [blocks][96]
    unsigned seq;
    static constexpr int K = 7;  // 7 x 40 >= STATS_BLOCKS: still one round (8 spills under the kernel's 128-VGPR cap)
    struct Raw { unsigned long long c[4]; };
    __device__ __forceinline__ const unsigned long long* at(int bb, int quad) const {
        return cells + (int64_t)bb * (STATS_L * 3) + 4 * quad;
    }
    __device__ __forceinline__ void issue(int bb, int quad, Raw& r) const {
#pragma unroll
        for (int c = 0; c < 4; ++c) r.c[c] = __hip_atomic_load(at(bb, quad) + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ float4 finish(int bb, int quad, const Raw& r) const {
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            unsigned long long cell = r.c[c];
            // (no time-out: the writers wait for nothing, every one of them gets its turn on the device)
            while ((unsigned)(cell >> 32) != seq) {
                __builtin_amdgcn_s_sleep(2);
                cell = __hip_atomic_load(at(bb, quad) + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            o[c] = __uint_as_float((unsigned)cell);
        }
        return make_float4(o[0], o[1], o[2], o[3]);
    }
};
template <class Rows>
__device__ __forceinline__ void stats_combine_columns(const Rows rows, int nblocks,
                                                      double* s_acc /*[STATS_COMB_GROUPS][96]*/, double* out /*[96]*/) {
    constexpr int COLS = STATS_L * 3, QUADS = COLS / 4, GROUPS = STATS_COMB_GROUPS;
    static_assert(QUADS * GROUPS == STATS_COMB_THREADS, "thread layout");
    const int quad = threadIdx.x % QUADS, g = threadIdx.x / QUADS;
    if (g < GROUPS) {
        double v[4] = {0.0, 0.0, 0.0, 0.0};
        constexpr int K = Rows::K;
        for (int b0 = g; b0 < nblocks; b0 += K * GROUPS) {
            typename Rows::Raw raw[K];
#pragma unroll
            for (int q = 0; q < K; ++q) {
                const int bb = b0 + q * GROUPS;
                if (bb < nblocks) rows.issue(bb, quad, raw[q]);
            }
#pragma unroll
            for (int q = 0; q < K; ++q) {
                const int bb = b0 + q * GROUPS;
                const float4 r = bb < nblocks ? rows.finish(bb, quad, raw[q]) : make_float4(0.f, 0.f, 0.f, 0.f);
                v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) s_acc[g * COLS + 4 * quad + c] = v[c];
    }
    __syncthreads();
    if (threadIdx.x < COLS) {
        double v = 0.0;
        for (int q = 0; q < GROUPS; ++q) v += s_acc[q * COLS + threadIdx.x];
        out[threadIdx.x] = v;
    }
    __syncthreads();
}
__device__ __forceinline__ void stats_combine_columns(const float* __restrict__ part, int nblocks, double* s_acc,
                                                      double* out) {
    stats_combine_columns(PartRows{part}, nblocks, s_acc, out);
}
__global__ __launch_bounds__(1024) void stats_multi_combine_kernel(const float* __restrict__ part, int nblocks,
                                                                   double* __restrict__ out /*[STATS_L][3] mapped*/) {
    __shared__ double s_acc[STATS_COMB_GROUPS * STATS_L * 3];
    stats_combine_columns(part, nblocks, s_acc, out);
}

// ESSPS without leaving the device (mppi.py:351-370): after each 32-temperature statistics pass one block combines
// the partial sums and runs the scalar step of the search (host_search.hpp: the same functions the host loop of
// mppi_essps_lambda calls) — round 0 applies the end-point rules or writes the refined grid for the second pass,
// round 1 interpolates the root.  The temperature ends up in `lambda_out` (device, fp32: what weights_reduce_kernel
// and finalize_kernel read) and in mapped host memory (double) for whoever asks later; the host never waits.
// The scalar steps of the search with the lanes of ONE wave sharing the work (call with all 64 lanes active; every lane
// returns the same values).  Same arithmetic and the same order of the sums as host::essps_round0 / essps_round1, which
// one lane would take ~3 us for (32 dependent LDS reads for the bracket, 10 double divisions and ~100 dependent double
// multiplications for the two polynomials): the bracket is a ballot, every polynomial term has its own lane.
template <int P>
__device__ __forceinline__ int essps_bracket_wave(const double* ess, double target_ess, int lane) {
    const unsigned long long above = __ballot(lane < P && ess[lane < P ? lane : 0] >= target_ess);
    const int i = above ? __ffsll((long long)above) - 1 : P - 1;
    return i < 1 ? 1 : i;
}
template <int P>
__device__ __forceinline__ bool essps_round0_wave(const double* lgrid, const double* ess, double target_ess,
                                                  const mppi::host::EsspsRange& r, int lane, int& i, mppi::host::EsspsRoot& root) {
    using namespace mppi::host;
    if (target_ess <= ess[0]) { root = EsspsRoot{r.lam_min, r.lmin, false}; return true; }
    if (target_ess >= ess[P - 1]) { root = EsspsRoot{r.lam_max, r.lmax, false}; return true; }
    i = essps_bracket_wave<P>(ess, target_ess, lane);
    constexpr int H = ESSPS_NPT / 2;
    if (i < H || i > P - H) return false;
    bool ok = true;  // lanes 0 .. NPT-2 own one pair of neighbours each: close, and ESS strictly increasing
    if (lane < ESSPS_NPT - 1) {
        const int k = i - H + lane;
        ok = lgrid[k + 1] - lgrid[k] <= ESSPS_LOG_FINE_RATIO && ess[k + 1] > ess[k];
    }
    if (!__all(ok)) return false;
    double term = 0.0;  // lanes 0..5: the terms of the six-point polynomial, lanes 8..11: of the four-point one
    if (lane < ESSPS_NPT) term = essps_poly_term<ESSPS_NPT>(lgrid, ess, target_ess, i - H, lane);
    else if (lane >= 8 && lane < 12) term = essps_poly_term<4>(lgrid, ess, target_ess, i - 2, lane - 8);
    double x6 = 0.0, x4 = 0.0;
#pragma unroll
    for (int a = 0; a < ESSPS_NPT; ++a) x6 += __shfl(term, a);
#pragma unroll
    for (int a = 0; a < 4; ++a) x4 += __shfl(term, 8 + a);
    if (x6 >= lgrid[i - 1] && x6 <= lgrid[i] && fabs(x6 - x4) <= ESSPS_AGREE) {
        root = EsspsRoot{exp(x6), x6, true};
        return true;
    }
    return false;
}
template <int P>
__device__ __forceinline__ mppi::host::EsspsRoot essps_round1_wave(const double* grid, const double* lgrid, const double* ess,
                                                                   double target_ess, int lane) {
    using namespace mppi::host;
    const int i = essps_bracket_wave<P>(ess, target_ess, lane);
    constexpr int H = ESSPS_NPT / 2;
    const int j0 = (i - H < 0 ? 0 : (i - H > P - ESSPS_NPT ? P - ESSPS_NPT : i - H));
    bool ok = true;
    if (lane < ESSPS_NPT - 1) ok = ess[j0 + lane + 1] > ess[j0 + lane];
    if (__all(ok)) {
        const double term = lane < ESSPS_NPT ? essps_poly_term<ESSPS_NPT>(lgrid, ess, target_ess, j0, lane) : 0.0;
        double x = 0.0;
#pragma unroll
        for (int a = 0; a < ESSPS_NPT; ++a) x += __shfl(term, a);
        if (x >= lgrid[i - 1] && x <= lgrid[i]) return EsspsRoot{exp(x), x, true};
    }
    return essps_linear(grid, ess, target_ess, i);
}
struct EsspsDev {
    // first grid of the NEXT search and its logs: geometric over [lam_min, lam_max] at first (host), then rewritten by
    // every finished search around its root (host_search.hpp: essps_first_grid)
    double grid0[STATS_L], lgrid0[STATS_L];
    double grid1[STATS_L], lgrid1[STATS_L];  // the refined grid round 0 wrote (`lams` holds the fp32 casts)
    double lam;                              // result
    int32_t done, pad;                       // round 0 already finished the search
};
// The scalar step after the sums of round ROUND are in s_sum (call with one full wave; j = lane).
template <int ROUND>
__device__ __forceinline__ void essps_select_step(const double* s_sum, double* s_ess, double* s_grid, double* s_lgrid,
                                                  double target_ess, const mppi::host::EsspsRange& range,
                                                  EsspsDev* __restrict__ st, float* __restrict__ lams,
                                                  float* __restrict__ lams0, float* __restrict__ lambda_out,
                                                  double* __restrict__ lambda_host, int j) {
    if (j < STATS_L) {
        s_ess[j] = s_sum[3 * j] * s_sum[3 * j] / s_sum[3 * j + 1];  // 32 double divisions, one per lane
        s_grid[j] = ROUND == 0 ? st->grid0[j] : st->grid1[j];
        s_lgrid[j] = ROUND == 0 ? st->lgrid0[j] : st->lgrid1[j];
    }
    __builtin_amdgcn_wave_barrier();
    mppi::host::EsspsRoot root{0.0, 0.0, false};  // (wave-uniform from here on)
    int i = 1;
    bool have = true;
    if (ROUND == 0) have = essps_round0_wave<STATS_L>(s_lgrid, s_ess, target_ess, range, j, i, root);
    else root = essps_round1_wave<STATS_L>(s_grid, s_lgrid, s_ess, target_ess, j);
    if (j == 0) {
        if (ROUND == 0) st->done = have ? 1 : 0;
        if (have) {
            st->lam = root.lam;
            *lambda_out = (float)root.lam;
            lambda_host[0] = root.lam; lambda_host[1] = root.lam; lambda_host[2] = (double)(ROUND + 1);
        }
    }
    if (j < STATS_L) {  // one grid point (an exp in double) per lane
        double g, lg;
        if (have) {  // the search is over: the next one starts from a grid around this root
            mppi::host::essps_first_point<STATS_L>(root.warm, root.log_lam, range, j, g, lg);
            st->grid0[j] = g; st->lgrid0[j] = lg;
            lams0[j] = (float)g;
        } else {     // the refined grid over the bracket
            mppi::host::essps_point<STATS_L>(s_grid[i - 1], s_grid[i], s_lgrid[i - 1], s_lgrid[i], j, g, lg);
            st->grid1[j] = g; st->lgrid1[j] = lg;
            lams[j] = (float)g;
        }
    }
}
__global__ __launch_bounds__(1024) void essps_select_kernel(const float* __restrict__ part, int nblocks, double target_ess,
                                                            mppi::host::EsspsRange range, EsspsDev* __restrict__ st,
                                                            float* __restrict__ lams, float* __restrict__ lams0,
                                                            float* __restrict__ lambda_out,
                                                            double* __restrict__ lambda_host) {
    __shared__ double s_acc[STATS_COMB_GROUPS * STATS_L * 3];
    __shared__ double s_sum[STATS_L * 3];
    __shared__ double s_ess[STATS_L], s_grid[STATS_L], s_lgrid[STATS_L];
    stats_combine_columns(part, nblocks, s_acc, s_sum);
    if (threadIdx.x >= WAVE) return;  // the scalar step: one wave, lane j owns temperature j where that helps
    essps_select_step<0>(s_sum, s_ess, s_grid, s_lgrid, target_ess, range, st, lams, lams0, lambda_out, lambda_host,
                         (int)threadIdx.x);
}
// Round 1 — statistics over the refined grid AND its select step — as ONE launch that costs its launch floor when round 0
// already finished the search (the warm-started first grid usually does: every block returns at once; as two kernels the
// skipped pair cost two floors, and the select kernel combined stale partial rows before it looked at `done`).  When the
// round runs, block 0 gathers the other blocks' 96 partial sums through tagged cells (CellRows) in the order and with the
// arithmetic of the two-kernel chain: the same temperature to the bit.  No `done` is written here, so reading it at the
// top does not race with block 0's step.
template <int ROUND>
__global__ __launch_bounds__(STATS_THREADS) void essps_round_kernel(const float* __restrict__ costs, int64_t N,
                                                                    const unsigned* __restrict__ min_key, double target_ess,
                                                                    mppi::host::EsspsRange range, EsspsDev* __restrict__ st,
                                                                    float* __restrict__ lams, float* __restrict__ lams0,
                                                                    float* __restrict__ lambda_out,
                                                                    double* __restrict__ lambda_host,
                                                                    unsigned long long* __restrict__ cells, unsigned seq) {
    if (ROUND == 1 && st->done) return;
    __shared__ union {
        StatsLds stats;
        double acc[STATS_COMB_GROUPS * STATS_L * 3];
    } u;
    __shared__ double s_sum[STATS_L * 3];
    __shared__ double s_ess[STATS_L], s_grid[STATS_L], s_lgrid[STATS_L];
    // (block 0's step rewrites lams0 — the NEXT search's first grid — only after every block published its sums, i.e.
    // after the last read of this round's temperatures)
    const float v = stats_multi_block(costs, N, key_to_float(*min_key), ROUND == 0 ? lams0 : lams, nullptr, u.stats);
    if (threadIdx.x < STATS_L * 3)
        __hip_atomic_store(cells + (int64_t)blockIdx.x * STATS_L * 3 + threadIdx.x,
                           ((unsigned long long)seq << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    if (blockIdx.x != 0) return;
    __syncthreads();  // u.stats is dead from here on
    stats_combine_columns(CellRows{cells, seq}, (int)gridDim.x, u.acc, s_sum);
    if (threadIdx.x >= WAVE) return;
    essps_select_step<ROUND>(s_sum, s_ess, s_grid, s_lgrid, target_ess, range, st, lams, lams0, lambda_out, lambda_host,
                             (int)threadIdx.x);
}

// LBPS without leaving the device (mppi.py:341-349,534-557).  The reference minimises the lower-bound objective with
// scipy's bounded Brent search, ~25 dependent probes; here every round evaluates the objective on a 32-temperature
// geometric grid in ONE pass over the costs (stats_multi_kernel), one block picks the grid minimum and writes the next
// grid over the two intervals around it; after LBPS_ROUNDS grids (spacing 25 % -> 1.4 % of lambda over [0.01, 10]) the
// last round minimises the quartic through the five points around the minimum in log(lambda)
// (host_search.hpp: lbps_grid_step — the same code the CPU tests run against scipy; round 4: two rounds instead of
// three + a parabola: same accuracy, two launches fewer).  The temperature stays in
// `lambda_out` (device) + mapped host memory; the host never waits.
constexpr int LBPS_ROUNDS = mppi::host::LBPS_GRID_ROUNDS;
struct LbpsDev {
    double grid0[STATS_L];  // round-0 temperatures: geometric over [lam_min, lam_max], written once by the host
    double grid[STATS_L];   // temperatures of the round in flight (`lams` holds their fp32 casts)
};
template <bool LAST, bool FIRST>
__global__ __launch_bounds__(1024) void lbps_select_kernel(const float* __restrict__ part,
                                                           const float* __restrict__ part_max, int nblocks,
                                                           const unsigned* __restrict__ min_key, double delta,
                                                           LbpsDev* __restrict__ st, float* __restrict__ lams,
                                                           float* __restrict__ lambda_out,
                                                           double* __restrict__ lambda_host /*[2]: next, used*/) {
    __shared__ double s_acc[STATS_COMB_GROUPS * STATS_L * 3];
    __shared__ double s_sum[STATS_L * 3];
    __shared__ double s_obj[STATS_L], s_grid[STATS_L];
    __shared__ double s_bracket[2];
    __shared__ float s_cmax;
    stats_combine_columns(part, nblocks, s_acc, s_sum);
    if (threadIdx.x >= WAVE) return;
    const int j = threadIdx.x;
    {   // the cost range: per-block maxima -> one wave
        float m = -INFINITY;
        for (int b = j; b < nblocks; b += WAVE) m = fmaxf(m, part_max[b]);
#pragma unroll
        for (int q = 32; q >= 1; q >>= 1) m = fmaxf(m, __shfl_xor(m, q));
        if (j == 0) s_cmax = m;
    }
    __builtin_amdgcn_wave_barrier();
    if (j < STATS_L) {
        const double g = FIRST ? st->grid0[j] : st->grid[j];
        s_grid[j] = g;
        const mppi::host::SoftmaxStats ss{(double)key_to_float(*min_key), (double)s_cmax, s_sum[3 * j], s_sum[3 * j + 1],
                                          s_sum[3 * j + 2]};
        s_obj[j] = mppi::host::lbps_objective(ss, delta);
    }
    __builtin_amdgcn_wave_barrier();
    if (j == 0) {
        double lo, hi, lam;
        mppi::host::lbps_grid_step<STATS_L>(s_grid, s_obj, LAST, lo, hi, lam);
        s_bracket[0] = lo; s_bracket[1] = hi;
        if (LAST) { *lambda_out = (float)lam; lambda_host[0] = lam; lambda_host[1] = lam; lambda_host[2] = (double)LBPS_ROUNDS; }
    }
    __builtin_amdgcn_wave_barrier();
    if (!LAST && j < STATS_L) {
        const double gj = mppi::host::essps_grid_point<STATS_L>(s_bracket[0], s_bracket[1], j);
        st->grid[j] = gj;
        lams[j] = (float)gj;
    }
}

// MPO without leaving the device (mppi.py:191-200,387-398): the dual variable and its Adam moments live in device
// memory; after the solve's weights one statistics pass at T = softplus(log T) (stats_partial_kernel reading T from
// `temp_dev`) and this one-thread step (host_search.hpp: mpo_step — the arithmetic the CPU tests pin to the reference)
// leave lambda = exp(log T) for the NEXT solve in `lambda_out`.
__global__ __launch_bounds__(WAVE) void mpo_step_kernel(const float* __restrict__ part, int nblocks,
                                                        const unsigned* __restrict__ min_key,
                                                        mppi::host::MpoState* __restrict__ st,
                                                        float* __restrict__ lambda_out, float* __restrict__ temp_dev,
                                                        double* __restrict__ lambda_host /*[2]: next, used*/) {
    double se = 0.0, se2 = 0.0, sec = 0.0;
    float cmax = -INFINITY;
    for (int b = threadIdx.x; b < nblocks; b += WAVE) {
        se += part[b * 4]; se2 += part[b * 4 + 1]; sec += part[b * 4 + 2];
        cmax = fmaxf(cmax, part[b * 4 + 3]);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        se += __shfl_xor(se, m); se2 += __shfl_xor(se2, m); sec += __shfl_xor(sec, m);
        cmax = fmaxf(cmax, __shfl_xor(cmax, m));
    }
    if (threadIdx.x == 0) {
        mppi::host::MpoState s = *st;
        const double used = (double)*lambda_out;
        const mppi::host::SoftmaxStats ss{(double)key_to_float(*min_key), (double)cmax, se, se2, sec};
        const double lam = mppi::host::mpo_step(s, ss);
        *st = s;
        *lambda_out = (float)lam;
        *temp_dev = s.temperature();
        lambda_host[0] = lam; lambda_host[1] = used;
    }
}

// ------------------------------------------------------------------------------------------
// MPPI.forward() as ONE launch (mppi.py:223-460) for N <= 512 x (number of CUs): a cooperative kernel.
//
// The multi-kernel solve of a small or medium problem is a chain of 3-9 dependent, latency-bound launches (launch + the
// first load of data another XCD just wrote ~ 4-5 us each; a captured hipGraph replays the same chain:
// profiles/r03_experiments.md).  Here the whole problem is resident at once — G = min(#CUs, ceil(N/64)) blocks of 512
// threads (at most 32 blocks up to 4096 trajectories), block b owning `spb` consecutive trajectories (one per thread of
// its first spb/64 waves — ONE wave as long as CUs are left: small problems spread over many CUs as lone waves, exactly
// like the stand-alone rollout kernel; the other waves of a block share its reductions and the regeneration of its
// weighted noise rows) — and the blocks talk through CELLS in HBM instead of kernel boundaries: an 8-byte word
// {fp32 value, 32-bit solve number} written with ONE relaxed agent-scope store and polled with agent-scope loads, so that
// data and "ready" cannot be seen apart and no fence or grid barrier is needed (the protocol of the peer-to-peer
// exchange, P2pCtx; a device-scope fence per block costs far more than a kernel boundary on this part).  A round trip
// through a cell costs about as much as a kernel boundary (~2.5 us), so the exchanges are arranged in as few DEPENDENT
// round trips as possible and every reader issues all its loads before it looks at the first one (fx_get_many):
//   more than 32 blocks:
//   1. every block publishes its minimum cost; every block reads all of them                              (1 round trip)
//   2. ESSPS / LBPS only, per round: every block publishes the 96 partial sums of its 32-temperature statistics; block 0
//      combines them (fixed order, double), runs the scalar step of the search (fused_scalar_step: host_search.hpp, the
//      code of essps_select_kernel / lbps_select_kernel) and broadcasts the next grid or the temperature    (2 each)
//   3. every block publishes its partial row sum_i e_i U_i and {sum e, sum e^2, sum e c} (zeros without a weight);
//      block 0 folds them (fixed order) and runs the tail of the solve: normalise, filter, warm start, batch-1 rollout.
//   up to 32 blocks: hop 1 and the broadcast disappear — a block's exponents are relative to its OWN minimum, published
//   next to its sums; whoever adds the blocks' sums rescales them by exp((c_min - c_ref,b) / lambda) (finalize_tail's
//   combine of shard summaries, applied to blocks), and EVERY block gathers the statistics and runs the scalar step
//   itself (same inputs, same order: the same temperature in every block).
// Costs and the minimum are BIT-IDENTICAL to the multi-kernel path (same device functions); the statistics and the
// weighted row are summed over another partition, i.e. the temperature and the action agree to rounding.  Deterministic.
// A poll that does not complete within ~2 s (a block that never became resident: the device is shared with another
// cooperative kernel) raises *error, voids the outputs and returns — no hang.
// 512 threads, not 1024: at 1024 the kernel is capped at 128 VGPRs, spilled to scratch memory, and every wave executed
// the double-precision invariants the compiler hoisted out of the rounds loop for the scalar step (7 us per round on a
// 28 us solve; profiles/r03_experiments.md) — hence also fused_scalar_step as a non-inlined function.
constexpr int FUSED_BLOCK = 512;
constexpr int FUSED_MAX_BLOCKS = 256;
constexpr int FUSED_MAX_ROW = 128;
constexpr int FUSED_SMALL_BLOCKS = 32;       // up to this many blocks no hop is spent on the global minimum or on a broadcast
constexpr int FX_CELLS = FUSED_MAX_ROW + 8;  // per (phase, block): >= 4 + row, >= 97
enum { FX_MIN = 0, FX_STATS = 1 /* +2*round */, FX_BCAST = 2 /* +2*round */, FX_ROW = 7, FX_PHASES = 8 };
enum { FUSED_RULE_NONE = 0, FUSED_RULE_ESSPS = 1, FUSED_RULE_LBPS = 2 };
// A poll that cannot complete within `timeout_ticks` gives up (100 MHz wall clock; default 20 ms — three orders of magnitude
// above the ~30 us a healthy single-launch solve takes, short enough for a control loop to notice within a tick or two;
// option "fused_timeout_us" for a GPU that is shared or preempted for longer): a block of this launch is not resident, i.e.
// something else holds the GPU's CUs.
constexpr long long FUSED_TIMEOUT_TICKS = 2000000ll;
struct FusedCtx {
    unsigned long long* cells;  // [FX_PHASES][FUSED_MAX_BLOCKS][FX_CELLS]
    int* error;                 // mapped host flag
    unsigned seq;               // this solve's number (never 0)
    long long timeout_ticks;    // poll budget
};
__device__ __forceinline__ unsigned long long* fx_cell(const FusedCtx& x, int phase, int b, int j) {
    return x.cells + ((size_t)phase * FUSED_MAX_BLOCKS + b) * FX_CELLS + j;
}
__device__ __forceinline__ void fx_put(const FusedCtx& x, int phase, int b, int j, float v) {
    __hip_atomic_store(fx_cell(x, phase, b, j), ((unsigned long long)x.seq << 32) | (unsigned long long)__float_as_uint(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float fx_wait(const FusedCtx& x, const unsigned long long* p, unsigned long long cell, long long t0,
                                         bool& timed_out) {
    unsigned spins = 0;
    while ((unsigned)(cell >> 32) != x.seq) {
        if ((++spins & 255u) == 0u && wall_clock64() - t0 > x.timeout_ticks) { timed_out = true; break; }
        __builtin_amdgcn_s_sleep(2);
        cell = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return __uint_as_float((unsigned)cell);
}
__device__ __forceinline__ float fx_get(const FusedCtx& x, int phase, int b, int j, long long t0, bool& timed_out) {
    const unsigned long long* p = fx_cell(x, phase, b, j);
    return fx_wait(x, p, __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), t0, timed_out);
}
// cell j of blocks b0, b0 + bstep, ... (n <= K of them): ALL loads are issued before the first tag is looked at, so the
// K cells cost one round trip, not K
template <int K>
__device__ __forceinline__ void fx_get_many(const FusedCtx& x, int phase, int b0, int bstep, int n, int j, float (&out)[K],
                                            long long t0, bool& timed_out) {
    unsigned long long c[K];
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (k < n) c[k] = __hip_atomic_load(fx_cell(x, phase, b0 + k * bstep, j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int k = 0; k < K; ++k) out[k] = k < n ? fx_wait(x, fx_cell(x, phase, b0 + k * bstep, j), c[k], t0, timed_out) : 0.0f;
}

// -DMPPI_FUSED_TRACE (experiments only): block 0 stamps the 100 MHz clock at its phase boundaries into error[1 + k]
#ifdef MPPI_FUSED_TRACE
#define FX_TRACE(k) do { if (b == 0 && tid == 0) s_trace[k] = (int)(wall_clock64() - t0); } while (0)  // (dumped at the end)
#else
#define FX_TRACE(k) do { } while (0)
#endif
struct FusedArgs {
    const float* mean;      // warm start [row] (read), then overwritten through mean_store
    const float* x0;        // [ds]
    float* costs;           // [N]
    unsigned* min_key;      // slot this solve's minimum goes to (later queries read it)
    unsigned* next_min_key; // the other slot, reset for the next multi-kernel rollout (it accumulates with atomicMin)
    float* mean_used;       // snapshots for later re-rolls (get_top_samples)
    float* x0_used;
    int spb;                // trajectories per block (a multiple of 64, <= FUSED_BLOCK)
    int rule;               // FUSED_RULE_*
    double rule_param, lam_min, lam_max;
    float lambda_arg;       // rule == NONE: > 0, or MPPI_LAMBDA_DEVICE = read *lambda_dev
    float* lambda_dev;      // device copy of the temperature (written by ESSPS / LBPS)
    double* lambda_host;    // mapped host [2]
    double* grid0;          // device [32]: round-0 grid (ESSPS: essps->grid0, rewritten around the root for the next search; LBPS: fixed)
    EsspsDev* essps;        // ESSPS: the search state shared with the multi-kernel path (first grid of the next search + logs)
    mppi::host::EsspsRange range;
    float* lams0;           // ESSPS: fp32 copy of grid0 for the multi-kernel path's statistics pass (kept in step)
    float* mean_store;
    float* action_out;
    float* state_out;
    float* stats_out;
    float* stats_keep;
    float* summary_out;     // [4 + row] the shard summary, for later readers
};

// The scalar step of a search round of the single-launch solve, for ONE wave (lane j): statistics sums -> ESS / LBPS
// objective per temperature -> essps_round0/1 (wave-parallel) or lbps_grid_step -> the next grid or the temperature, left
// in s_lams[0..31] (next grid as fp32, zeros once the temperature is known), s_lams[32] (1 = known), s_lams[33] (it).
// NOT inlined: its double-precision code (and what the compiler would hoist out of the rounds loop for it) stays out of
// the register budget and the loop pre-header of solve_fused_kernel, where all the other waves would execute it too.
struct FusedSearchLds {  // the search's staging in LDS (ONE pointer for the call: arguments beyond 32 dwords travel through scratch memory)
    double sumd[STATS_L * 3];
    double vald[STATS_L], gridd[STATS_L], lgridd[STATS_L];
    float lams[STATS_L + 2];
    float bc[4];  // [0] block minimum, [1] block maximum, [2] global minimum, [3] global maximum
};
// (every argument a scalar: 27 dwords, all in registers — a struct by value, like anything beyond 32 dwords, would travel
// through scratch memory, a store -> load round trip at the head of the call)
__device__ __noinline__ void fused_scalar_step(int rule, int r, int rounds, bool first_block, int j, double rule_param, double lam_min,
                                               double lam_max, double range_lmin, double range_lmax, EsspsDev* essps, float* lams0,
                                               float* lambda_dev, FusedSearchLds* S) {
    struct { int rule; double rule_param, lam_min, lam_max; mppi::host::EsspsRange range; EsspsDev* essps; float* lams0; float* lambda_dev; }
        A{rule, rule_param, lam_min, lam_max, mppi::host::EsspsRange{lam_min, lam_max, range_lmin, range_lmax}, essps, lams0, lambda_dev};
    double* s_sumd = S->sumd; double* s_vald = S->vald; double* s_gridd = S->gridd; double* s_lgridd = S->lgridd;
    float* s_lams = S->lams; const float* s_bc = S->bc;
    if (j < STATS_L) {
        if (A.rule == FUSED_RULE_ESSPS) s_vald[j] = s_sumd[3 * j] * s_sumd[3 * j] / s_sumd[3 * j + 1];
        else s_vald[j] = mppi::host::lbps_objective(
            mppi::host::SoftmaxStats{(double)s_bc[2], (double)s_bc[3], s_sumd[3 * j], s_sumd[3 * j + 1], s_sumd[3 * j + 2]},
            A.rule_param);
    }
    __builtin_amdgcn_wave_barrier();
    double lam = 0.0, gj = 0.0, lgj = 0.0;
    bool have;
    if (A.rule == FUSED_RULE_ESSPS) {
        mppi::host::EsspsRoot root{0.0, 0.0, false};  // (wave-uniform)
        int i = 1;
        have = true;
        if (r == 0) have = essps_round0_wave<STATS_L>(s_lgridd, s_vald, A.rule_param, A.range, j, i, root);
        else root = essps_round1_wave<STATS_L>(s_gridd, s_lgridd, s_vald, A.rule_param, j);
        lam = root.lam;
        if (j < STATS_L) {  // the next grid, one point per lane
            if (!have) {
                const double lo = s_gridd[i - 1], hi = s_gridd[i], llo = s_lgridd[i - 1], lhi = s_lgridd[i];
                mppi::host::essps_point<STATS_L>(lo, hi, llo, lhi, j, gj, lgj);
                __builtin_amdgcn_wave_barrier();
                s_gridd[j] = gj; s_lgridd[j] = lgj;
            } else if (first_block) {  // the next ESSPS search starts around this root
                double g0, lg0;
                mppi::host::essps_first_point<STATS_L>(root.warm, root.log_lam, A.range, j, g0, lg0);
                A.essps->grid0[j] = g0; A.essps->lgrid0[j] = lg0;
                A.lams0[j] = (float)g0;
            }
        }
    } else {
        if (j == 0) {
            double lo = A.lam_min, hi = A.lam_max;
            mppi::host::lbps_grid_step<STATS_L>(s_gridd, s_vald, r == rounds - 1, lo, hi, lam);
            s_sumd[0] = lo; s_sumd[1] = hi; s_sumd[2] = lam;
        }
        __builtin_amdgcn_wave_barrier();
        have = r == rounds - 1;
        lam = s_sumd[2];
        if (!have && j < STATS_L) {
            gj = mppi::host::essps_grid_point<STATS_L>(s_sumd[0], s_sumd[1], j);
            s_gridd[j] = gj;
        }
    }
    if (j < STATS_L) s_lams[j] = have ? 0.0f : (float)gj;  // (zeros once the temperature is known)
    if (j == 0) {
        if (have && first_block) {
            *A.lambda_dev = (float)lam;
            s_vald[0] = lam; s_vald[1] = (double)(r + 1);  // (block 0 copies them to the host's mirror at the very end of the kernel:
                                                           // a store to host memory holds up every later wait on memory of this wave)
        }
        s_lams[STATS_L + 1] = have ? (float)lam : 0.0f;
        s_lams[STATS_L] = have ? 1.0f : 0.0f;
    }
}

template <int MODEL, int FAST>
__global__ __launch_bounds__(FUSED_BLOCK) void solve_fused_kernel(FusedArgs A, Dims d, GenCtx gen, ModelCtx ctx,
                                                                  SgFilter sg, FusedCtx fx) {
    using M = ModelT<MODEL, FAST>;
    constexpr int NWV = FUSED_BLOCK / WAVE;
    constexpr bool UC = FAST != 0;
    constexpr int KG = 32;                         // cells a thread keeps in flight: every gather is ONE round trip (G <= 256)
    constexpr int KS = (FUSED_SMALL_BLOCKS + FUSED_BLOCK / FX_CELLS - 1) / (FUSED_BLOCK / FX_CELLS);                          // ... with few blocks (G <= 32 over >= 7 thread groups)
    constexpr int COLS = STATS_L * 3;              // 96 statistics columns
    constexpr int SPARTS = FUSED_BLOCK / COLS;     // 10 row groups of the statistics combine
    constexpr int CW = FX_CELLS;                   // column slots of the row fold (>= 4 + row)
    constexpr int RPARTS = FUSED_BLOCK / CW;       // 7 row groups of the row fold
    __shared__ float s_c[FUSED_BLOCK];             // this block's costs (padded), later its weights
    __shared__ float s_p[NWV][STATS_L][3];
    __shared__ float s_w[NWV][4];                  // per-wave scalars
    __shared__ double s_scratch[2048];             // statistics combine [SPARTS][COLS] doubles; aliased: row partials, 4096 floats
    __shared__ float s_fold[RPARTS][CW];           // block 0's row fold
    __shared__ FusedSearchLds s_search;
    double* const s_sumd = s_search.sumd; double* const s_vald = s_search.vald;
    double* const s_gridd = s_search.gridd; double* const s_lgridd = s_search.lgridd;
    float* const s_lams = s_search.lams; float* const s_bc = s_search.bc;
    __shared__ float s_ref[2][FUSED_SMALL_BLOCKS]; // few blocks: the blocks' reference costs / their maxima or rescale factors
    __shared__ int s_flag;
    __shared__ float s_x0[MPPI_MAX_DIM_STATE];
#ifdef MPPI_FUSED_TRACE
    __shared__ int s_trace[24];
    __shared__ int s_wtrace[16][2];  // (per wave: start / end of the round-0 statistics)
    if (threadIdx.x < 24) s_trace[threadIdx.x] = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];  // [8R] mean groups, [T*KROW] step rows, then the tail's staging
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, b = blockIdx.x, G = gridDim.x;
    const long long t0 = wall_clock64();
    bool timed_out = false;

    // ---- stage the wave-uniform per-step inputs (like rollout_cost_kernel)
    float4* s_mean4 = reinterpret_cast<float4*>(s_dyn);
    float* s_ktab = s_dyn + 8 * d.R;
    for (int f = tid; f < 4 * d.R; f += FUSED_BLOCK) {
        const float m = f < d.row ? A.mean[f] : 0.0f;
        s_dyn[f] = m;
        s_dyn[4 * d.R + f] = 0.0f;
        if (b == 0 && f < d.row) A.mean_used[f] = m;
    }
    for (int f = tid; f < d.T * M::KROW; f += FUSED_BLOCK) s_ktab[f] = ctx.ref[f];
    if (tid < M::DS) { s_x0[tid] = A.x0[tid]; if (b == 0) A.x0_used[tid] = A.x0[tid]; }
    if (A.rule != FUSED_RULE_NONE && tid >= FUSED_BLOCK - STATS_L) {  // the search's first grid (its loads hide behind the rollout)
        const int j = tid - (FUSED_BLOCK - STATS_L);
        s_gridd[j] = A.grid0[j];
        if (A.rule == FUSED_RULE_ESSPS) s_lgridd[j] = A.essps->lgrid0[j];
    }
    __syncthreads();
    FX_TRACE(0);

    // ---- steps 1-3: one trajectory per thread of the block's first spb/64 waves
    const int64_t i = (int64_t)b * A.spb + tid;
    const bool mine = tid < A.spb && i < d.N;
    float total = INFINITY;
    if (mine) {
        const uint64_t gi = (uint64_t)(d.sample_offset + i);
        const bool inherit = (d.sample_offset + i) < d.inherit_count;
        const float4* mp = inherit ? s_mean4 : s_mean4 + d.R;
        total = lane_cost<MODEL, FAST, true, UC>(nullptr, gi, gen, mp, s_ktab, s_x0, d, ctx);
        A.costs[i] = total;
    }
    FX_TRACE(1);
    // ---- the block's minimum and maximum; hop 1 (G > FUSED_SMALL_BLOCKS only): the global ones
    // With few blocks no hop is spent on the minimum: a block's exponents are taken relative to its OWN minimum
    // (`cref`), which it publishes next to its sums, and whoever adds the blocks' sums rescales them by
    // exp((c_min - cref_b) / lambda) — the combine of the shard summaries (finalize_tail) applied to blocks.
    const bool small = G <= FUSED_SMALL_BLOCKS;
    {
        const float wm = wave_min(total);
        float wx = mine ? total : -INFINITY;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) wx = fmaxf(wx, __shfl_xor(wx, m));
        if (lane == 0) { s_w[wid][0] = wm; s_w[wid][1] = wx; }
        __syncthreads();
        if (tid == 0) {
            float m = s_w[0][0], mx = s_w[0][1];
#pragma unroll
            for (int w = 1; w < NWV; ++w) { m = fminf(m, s_w[w][0]); mx = fmaxf(mx, s_w[w][1]); }
            s_bc[0] = m; s_bc[1] = mx;
            s_bc[2] = m; s_bc[3] = mx;  // (small: until the first gather knows better)
            if (!small) { fx_put(fx, FX_MIN, b, 0, m); fx_put(fx, FX_MIN, b, 1, mx); }
        }
        if (!small) {
            float gm = INFINITY, gx = -INFINITY;
            if (tid < G) gm = fx_get(fx, FX_MIN, tid, 0, t0, timed_out);                                   // G <= 256
            else if (tid >= FUSED_BLOCK / 2 && tid - FUSED_BLOCK / 2 < G) gx = fx_get(fx, FX_MIN, tid - FUSED_BLOCK / 2, 1, t0, timed_out);
            gm = wave_min(gm);
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) gx = fmaxf(gx, __shfl_xor(gx, m));
            __syncthreads();
            if (lane == 0) { s_w[wid][0] = gm; s_w[wid][1] = gx; }
            __syncthreads();
            if (tid == 0) {
                float m = s_w[0][0], mx = s_w[0][1];
#pragma unroll
                for (int w = 1; w < NWV; ++w) { m = fminf(m, s_w[w][0]); mx = fmaxf(mx, s_w[w][1]); }
                s_bc[2] = m; s_bc[3] = mx;
            }
        }
        __syncthreads();
    }
    // what this block's exponents are relative to: the global minimum once it is known (a block without trajectories
    // publishes +inf as its reference — a factor 0 wherever its zeros are added — and uses 0 itself)
    float cref = s_bc[2], cpub = s_bc[2];
    if (!(cref < INFINITY)) cref = 0.0f;
    bool cmin_known = !small;
    FX_TRACE(2);

    // ---- step 4: the temperature
    float lambda = A.lambda_arg;
    if (A.rule == FUSED_RULE_NONE && !(lambda > 0.0f)) lambda = *A.lambda_dev;  // (MPO: the dual's temperature)
    if (A.rule != FUSED_RULE_NONE) {
        const int rounds = A.rule == FUSED_RULE_ESSPS ? 2 : LBPS_ROUNDS;
        s_c[tid] = mine ? total : 3.0e38f;  // padding: e = exp(-inf) = 0 and 0 * c = 0
        __syncthreads();
        FX_TRACE(14);
        for (int r = 0; r < rounds; ++r) {
            // statistics of this block's costs for the 32 temperatures of round r (stats_multi_kernel's arithmetic)
            // (thread = temperature l x one of 32 runs of spb/32 consecutive costs: every thread of the block works)
#ifdef MPPI_FUSED_TRACE
            if (b == 0 && r == 0 && lane == 0) s_wtrace[wid][0] = (int)(wall_clock64() - t0);
#endif
            const int l = tid & (STATS_L - 1), chunk = tid >> 5, per = A.spb / (FUSED_BLOCK / STATS_L);
            const float lam_l = r == 0 ? (float)s_gridd[l] : s_lams[l];
            const float inv_lam = 1.0f / lam_l;
            float se = 0.0f, se2 = 0.0f, sec = 0.0f;
            {
                const float* cc = s_c + chunk * per;
#pragma unroll 2
                for (int j = 0; j < per; ++j) {
                    const float c = cc[j];
                    const float e = expf((cref - c) * inv_lam);
                    se += e;
                    se2 = fmaf(e, e, se2);
                    sec = fmaf(e, c, sec);
                }
            }
            se += __shfl_xor(se, 32); se2 += __shfl_xor(se2, 32); sec += __shfl_xor(sec, 32);
            if (lane < STATS_L) { s_p[wid][lane][0] = se; s_p[wid][lane][1] = se2; s_p[wid][lane][2] = sec; }
            if (r == 0) FX_TRACE(15);
#ifdef MPPI_FUSED_TRACE
            if (b == 0 && r == 0 && lane == 0) s_wtrace[wid][1] = (int)(wall_clock64() - t0);
#endif
            __syncthreads();
            if (r == 0) FX_TRACE(16);
            if (tid < COLS) {
                float v = 0.0f;
#pragma unroll
                for (int w = 0; w < NWV; ++w) v += (&s_p[w][0][0])[tid];
                fx_put(fx, FX_STATS + 2 * r, b, tid, v);
            } else if (small && tid == COLS) {
                fx_put(fx, FX_STATS + 2 * r, b, COLS, cpub);
                fx_put(fx, FX_STATS + 2 * r, b, COLS + 1, s_bc[1]);
            }
            if (r == 0) FX_TRACE(10);
            if (small || b == 0) {  // (few blocks: EVERY block gathers and runs the scalar step itself — no broadcast hop)
                if (small) {        // the blocks' reference costs first: the global minimum / maximum
                    if (tid < G) s_ref[0][tid] = fx_get(fx, FX_STATS + 2 * r, tid, COLS, t0, timed_out);
                    else if (tid >= WAVE && tid - WAVE < G) s_ref[1][tid - WAVE] = fx_get(fx, FX_STATS + 2 * r, tid - WAVE, COLS + 1, t0, timed_out);
                    __syncthreads();
                    if (wid < 2) {
                        float v = lane < G ? s_ref[wid][lane] : (wid == 0 ? INFINITY : -INFINITY);
#pragma unroll
                        for (int m = 32; m >= 1; m >>= 1) v = wid == 0 ? fminf(v, __shfl_xor(v, m)) : fmaxf(v, __shfl_xor(v, m));
                        if (lane == 0) s_bc[2 + wid] = v;
                    }
                    __syncthreads();
                }
                // combine: thread (col, part) sums blocks part, part + SPARTS, ... in ascending order, KG cells in flight
                const int col = tid % COLS, part = tid / COLS;
                if (part < SPARTS) {
                    double v = 0.0;
                    const float gmin = s_bc[2];
                    const float lam_c = r == 0 ? (float)s_gridd[col / 3] : s_lams[col / 3];
                    const float inv_c = 1.0f / lam_c;
                    if (small) {  // <= KS blocks per thread; sums relative to the block's reference -> relative to the global minimum
                        float vals[KS];
                        const int n = min(KS, (G - part + SPARTS - 1) / SPARTS);
                        fx_get_many<KS>(fx, FX_STATS + 2 * r, part, SPARTS, n, col, vals, t0, timed_out);
#pragma unroll
                        for (int k = 0; k < KS; ++k)
                            if (k < n) {
                                const float f = expf((gmin - s_ref[0][part + k * SPARTS]) * inv_c);
                                v += (double)vals[k] * (double)(col % 3 == 1 ? f * f : f);
                            }
                    } else {
                        for (int b0 = part; b0 < G; b0 += KG * SPARTS) {
                            float vals[KG];
                            const int n = min(KG, (G - b0 + SPARTS - 1) / SPARTS);
                            fx_get_many<KG>(fx, FX_STATS + 2 * r, b0, SPARTS, n, col, vals, t0, timed_out);
#pragma unroll
                            for (int k = 0; k < KG; ++k) v += (double)vals[k];
                        }
                    }
                    s_scratch[part * COLS + col] = v;
                }
                __syncthreads();
                if (r == 0) FX_TRACE(11);
                if (tid < COLS) {
                    double v = 0.0;
                    for (int q = 0; q < SPARTS; ++q) v += s_scratch[q * COLS + tid];
                    s_sumd[tid] = v;
                }
                __syncthreads();
                if (r == 0) FX_TRACE(12);
                if (tid < WAVE) {  // the scalar step: one wave (essps_select_kernel / lbps_select_kernel)
                    if (r == 0) FX_TRACE(17);
                    fused_scalar_step(A.rule, r, rounds, b == 0, tid, A.rule_param, A.lam_min, A.lam_max, A.range.lmin, A.range.lmax, A.essps,
                                      A.lams0, A.lambda_dev, &s_search);
                    if (r == 0) FX_TRACE(18);
                }
                __syncthreads();
                if (r == 0) FX_TRACE(13);
                // broadcast: every block gets its OWN copy of the 34 cells (nobody polls a shared address)
                if (!small)
                    for (int q = tid; q < G * (STATS_L + 2); q += FUSED_BLOCK)
                        fx_put(fx, FX_BCAST + 2 * r, q / (STATS_L + 2), q % (STATS_L + 2), s_lams[q % (STATS_L + 2)]);
            } else {
                if (tid < STATS_L + 2) s_lams[tid] = fx_get(fx, FX_BCAST + 2 * r, b, tid, t0, timed_out);
            }
            __syncthreads();
            FX_TRACE(3 + r);
            if (small) { cref = cpub = s_bc[2]; cmin_known = true; }  // (every block has seen all the minima by now)
            if (s_lams[STATS_L] != 0.0f) { lambda = s_lams[STATS_L + 1]; break; }
        }
        __syncthreads();
    }
    FX_TRACE(6);

    // ---- steps 5-6: weights (relative to cref) and this block's share of sum_i e_i U_i
    const float xmax = (-cref) / lambda;
    const float e = mine ? expf((-total) / lambda - xmax) : 0.0f;
    s_c[tid] = e;
    {
        const float cz = e != 0.0f ? total : 0.0f;
        const float se = wave_sum(e), se2 = wave_sum(e * e), sec = wave_sum(e * cz);
        if (lane == 0) { s_w[wid][0] = se; s_w[wid][1] = se2; s_w[wid][2] = sec; }
    }
    __syncthreads();
    float bse = 0.0f, bse2 = 0.0f, bsec = 0.0f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) { bse += s_w[w][0]; bse2 += s_w[w][1]; bsec += s_w[w][2]; }
    {
        int RP = 1;
        while (RP < d.R) RP <<= 1;  // float4 groups per row, rounded up to a power of two (<= 32)
        const int r = tid & (RP - 1), slice = tid / RP, nsl = FUSED_BLOCK / RP;
        float* s_part = reinterpret_cast<float*>(s_scratch);  // [nsl][4 * RP] = 4096 floats
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (bse != 0.0f && r < d.R) {
            for (int sidx = slice; sidx < A.spb; sidx += nsl) {
                const float es = s_c[sidx];
                if (es != 0.0f) {
                    const int64_t i2 = (int64_t)b * A.spb + sidx;
                    const uint64_t gi2 = (uint64_t)(d.sample_offset + i2);
                    const float4 n4 = gen_noise4(gi2, r, gen, d);
                    const float4 m4 = ((d.sample_offset + i2) < d.inherit_count) ? s_mean4[r] : s_mean4[d.R + r];
                    const float nv[4] = {n4.x, n4.y, n4.z, n4.w}, mv[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = ctrl_index(j, d.dc);
                        acc[j] = fmaf(es, clampf(mv[j] + nv[j], d.u_min[k], d.u_max[k]), acc[j]);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) s_part[slice * 4 * RP + 4 * r + j] = acc[j];
        __syncthreads();
        {   // fold the slices in two steps (fixed order): W = 4 RP columns x Q = 1024 / W groups of nsl / Q = 4 slices each
            const int W = 4 * RP, Q = FUSED_BLOCK / W, c = tid & (W - 1), q = tid / W;
            float* s_half = &s_p[0][0][0];  // [Q][W] = 1024 floats (the statistics' staging is free by now)
            float v = 0.0f;
            for (int sl = q; sl < nsl; sl += Q) v += s_part[sl * W + c];
            s_half[q * W + c] = v;
            __syncthreads();
            if (tid < d.row) {
                float t = 0.0f;
                for (int g = 0; g < Q; ++g) t += s_half[g * W + tid];
                fx_put(fx, FX_ROW, b, MPPI_SUMMARY_HEAD + tid, t);
            }
        }
        if (tid == FUSED_BLOCK - 1) {
            fx_put(fx, FX_ROW, b, 0, cpub);
            fx_put(fx, FX_ROW, b, 1, bse); fx_put(fx, FX_ROW, b, 2, bse2); fx_put(fx, FX_ROW, b, 3, bsec);
        }
    }
    FX_TRACE(7);
    if (b != 0) {
        if (timed_out) *fx.error = 1;
        return;
    }

    // ---- block 0: fold the blocks' rows in ascending order, then the tail of the solve
    float* s_act = s_dyn + 8 * d.R + d.T * M::KROW;  // [row]
    float* s_sum = s_act + d.row;                     // [4 + row]
    float* s_yp = s_sum + MPPI_SUMMARY_HEAD + d.row;  // filter staging
    const bool rescale = small && !cmin_known;  // the blocks' exponents are relative to their own minima
    float cmin = cref;
    if (rescale) {  // finalize_tail's combine of shard summaries, applied to the blocks: f_b = exp((-cref_b)/lambda - max)
        if (tid < G) s_ref[0][tid] = fx_get(fx, FX_ROW, tid, 0, t0, timed_out);
        __syncthreads();
        if (wid == 0) {
            float v = lane < G ? s_ref[0][lane] : INFINITY;
            v = wave_min(v);
            if (lane == 0) s_bc[2] = v;
            if (lane < G) s_ref[1][lane] = expf((-s_ref[0][lane]) / lambda - (-v) / lambda);
        }
        __syncthreads();
        cmin = s_bc[2];
    }
    {
        const int col = tid % CW, part = tid / CW;    // cell slot (1 .. 3 + row are used), row group
        if (part < RPARTS) {
            float v = 0.0f;
            if (col >= 1 && col < MPPI_SUMMARY_HEAD + d.row) {
                if (small) {
                    float vals[KS];
                    const int n = min(KS, (G - part + RPARTS - 1) / RPARTS);
                    fx_get_many<KS>(fx, FX_ROW, part, RPARTS, n, col, vals, t0, timed_out);
#pragma unroll
                    for (int k = 0; k < KS; ++k)
                        if (k < n) {
                            const float f = rescale ? s_ref[1][part + k * RPARTS] : 1.0f;
                            v = rescale ? fmaf(col == 2 ? f * f : f, vals[k], v) : v + vals[k];
                        }
                } else {
                    for (int b0 = part; b0 < G; b0 += KG * RPARTS) {
                        float vals[KG];
                        const int n = min(KG, (G - b0 + RPARTS - 1) / RPARTS);
                        fx_get_many<KG>(fx, FX_ROW, b0, RPARTS, n, col, vals, t0, timed_out);
#pragma unroll
                        for (int k = 0; k < KG; ++k) v += vals[k];
                    }
                }
            }
            s_fold[part][col] = v;
        }
    }
    if (tid == 0) { *A.min_key = float_to_key(cmin); *A.next_min_key = 0xFFFFFFFFu; }
    __syncthreads();
    if (tid >= 1 && tid < MPPI_SUMMARY_HEAD + d.row) {
        float v = 0.0f;
#pragma unroll
        for (int q = 0; q < RPARTS; ++q) v += s_fold[q][tid];
        s_sum[tid] = v;
        if (A.summary_out) A.summary_out[tid] = v;
    }
    if (tid == 0) { s_sum[0] = cmin; if (A.summary_out) A.summary_out[0] = cmin; }
    s_flag = 0;
    __syncthreads();
    FX_TRACE(8);
    if (timed_out) s_flag = 1;
    __syncthreads();
    if (s_flag) {
        // A block is missing: no partial answer leaves this kernel — but no NaN reaches an actuator either.  The outputs
        // become the PREVIOUS plan (the warm start this solve sampled around, which stays the warm start: nothing is stored)
        // and its rollout from the current state; the statistics are NaN and the error flag is raised (mapped host memory:
        // mppi_fused_error; the handle returns to the multi-kernel path for good).
        for (int c = tid; c < d.row; c += FUSED_BLOCK) {
            s_act[c] = A.mean[c];
            if (A.action_out) A.action_out[c] = s_act[c];
        }
        if (tid < 4 && A.stats_out) A.stats_out[tid] = __uint_as_float(0x7fc00000u);
        if (tid == 0) *fx.error = 1;
        __syncthreads();
        if (A.state_out) batch1_rollout<MODEL, FAST>(ctx, s_x0, s_act, d.T, A.state_out);
        return;
    }
    finalize_tail<MODEL, FAST>(s_sum, 1, lambda, d.row, d.T, s_x0, s_act, s_yp, A.mean_store, A.action_out, A.state_out,
                               A.stats_out, A.stats_keep, sg, ctx);
    if (tid == 0 && A.rule != FUSED_RULE_NONE) { A.lambda_host[0] = s_vald[0]; A.lambda_host[1] = s_vald[0]; A.lambda_host[2] = s_vald[1]; }
    FX_TRACE(9);
#ifdef MPPI_FUSED_TRACE
    if (tid == 0) for (int k = 0; k < 24; ++k) fx.error[1 + k] = s_trace[k];
    if (tid < 32) fx.error[32 + tid] = s_wtrace[tid >> 1][tid & 1];
#endif
}

// `_weights` (mppi.py:376) given the global min cost and sum e.
__global__ __launch_bounds__(BLOCK) void weights_kernel(const float* __restrict__ costs, int64_t N, float lambda,
                                                        float cmin, float sum_e, float* __restrict__ w) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < N) w[i] = expf((-costs[i]) / lambda - (-cmin) / lambda) / sum_e;
}

// `_states_prediction` (mppi.py:508-524) for k action sequences in the reference layout.
template <int MODEL, int FAST>
__global__ __launch_bounds__(WAVE) void rollout_actions_kernel(const float* __restrict__ actions, int k, int T,
                                                               const float* __restrict__ x0,
                                                               float* __restrict__ states, ModelCtx ctx) {
    constexpr int DS = ModelT<MODEL, FAST>::DS, DC = ModelT<MODEL, FAST>::DC;
    const int q = blockIdx.x * WAVE + threadIdx.x;
    if (q >= k) return;
    const float* a = actions + (int64_t)q * T * DC;
    float* out = states + (int64_t)q * (T + 1) * DS;
    rollout_states_checked<MODEL, FAST>(x0, T, ctx, out, [&](int t, float* u) {
#pragma unroll
        for (int kk = 0; kk < DC; ++kk) u[kk] = a[t * DC + kk];
    });
}

// `_state_seq_batch[idx]` (mppi.py:481) re-rolled from the resident noise.
template <int MODEL, int FAST>
__global__ __launch_bounds__(WAVE) void rollout_samples_kernel(const float4* __restrict__ noise,
                                                               const float* __restrict__ mean,
                                                               const int64_t* __restrict__ idx, int k,
                                                               const float* __restrict__ x0,
                                                               float* __restrict__ states, Dims d, ModelCtx ctx) {
    constexpr int DS = ModelT<MODEL, FAST>::DS, DC = ModelT<MODEL, FAST>::DC;
    const int q = blockIdx.x * WAVE + threadIdx.x;
    if (q >= k) return;
    const int64_t i = idx[q];
    const bool inherit = (d.sample_offset + i) < d.inherit_count;
    const float* np = reinterpret_cast<const float*>(noise + (i >> 6) * d.R * 64 + (i & 63));
    float* out = states + (int64_t)q * (d.T + 1) * DS;
    rollout_states_checked<MODEL, FAST>(x0, d.T, ctx, out, [&](int t, float* u) {
#pragma unroll
        for (int kk = 0; kk < DC; ++kk) {
            const int f = t * DC + kk;
            const float e = np[(int64_t)(f >> 2) * 256 + (f & 3)];
            const float m = inherit ? mean[f] : 0.0f;
            u[kk] = clampf(m + e, d.u_min[kk], d.u_max[kk]);
        }
    });
}

// ------------------------------------------------------------------------------------------
// get_top_samples (mppi.py:462-487) on the device: the k largest weights are the k smallest costs.
// Radix select on the order-preserving cost keys (11 + 11 + 10 bits): three histogram passes over costs[N]
// (LDS histograms merged into a global one; passes 1 and 2 count only keys under the prefix chosen so far, which
// every block re-derives from the previous histogram), a collect pass that gathers the keys below the k-th key
// plus as many ties as are needed, and one block that sorts the k candidates by (key, index) — the order does
// not depend on the atomics that gathered them — and re-rolls their trajectories from the regenerated (or
// resident) noise around the mean the solve sampled.  State trajectories S[N,T+1,ds] are never materialised.
constexpr int TOPK_BINS = 2048;
constexpr int TOPK_MAX = 1024;
struct TopkSel { unsigned prefix, krem; };  // high bits selected so far; how many keys to take under that prefix
__device__ __forceinline__ constexpr int topk_shift(int pass) { return pass == 0 ? 21 : pass == 1 ? 10 : 0; }
__device__ __forceinline__ constexpr int topk_bits(int pass) { return pass == 2 ? 10 : 11; }

// Block-wide (NT threads): the bin whose cumulative count first reaches krem, and the count below that bin.
template <int NT = BLOCK>
__device__ __forceinline__ void topk_pick(const unsigned* __restrict__ hist, int nbins, unsigned krem,
                                          unsigned* __restrict__ s_scan /*[NT + 2]*/, unsigned& bin,
                                          unsigned& below) {
    constexpr int BLOCK = NT;  // (shadows the global block size inside this function)
    const int per = (nbins + BLOCK - 1) / BLOCK;
    const int b0 = threadIdx.x * per;
    unsigned loc = 0;
    for (int b = b0; b < min(b0 + per, nbins); ++b) loc += hist[b];
    s_scan[threadIdx.x] = loc;
    __syncthreads();
    if (threadIdx.x < WAVE) {  // exclusive scan of the BLOCK partial sums by one wave (BLOCK / WAVE each)
        constexpr int PER = BLOCK / WAVE;
        unsigned v[PER], sum = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) { v[q] = s_scan[threadIdx.x * PER + q]; sum += v[q]; }
        unsigned incl = sum;
#pragma unroll
        for (int m = 1; m < WAVE; m <<= 1) {
            const unsigned o = __shfl_up(incl, m);
            if ((int)threadIdx.x >= m) incl += o;
        }
        unsigned run = incl - sum;
#pragma unroll
        for (int q = 0; q < PER; ++q) { s_scan[threadIdx.x * PER + q] = run; run += v[q]; }
    }
    __syncthreads();
    const unsigned excl = s_scan[threadIdx.x];
    __syncthreads();
    if (excl < krem && krem <= excl + loc) {  // exactly one thread (loc > 0 there)
        unsigned run = excl;
        for (int b = b0; b < min(b0 + per, nbins); ++b) {
            const unsigned hcount = hist[b];
            if (krem <= run + hcount) { s_scan[BLOCK] = (unsigned)b; s_scan[BLOCK + 1] = run; break; }
            run += hcount;
        }
    }
    __syncthreads();
    bin = s_scan[BLOCK];
    below = s_scan[BLOCK + 1];
}

// prefix/krem entering pass PASS (derived from the histogram of pass PASS-1); block 0 records it in sel[PASS-1]
template <int PASS>
__device__ __forceinline__ TopkSel topk_enter(const unsigned* __restrict__ hist, TopkSel* __restrict__ sel, unsigned k,
                                              unsigned* __restrict__ s_scan) {
    TopkSel cur{0u, k};
    if (PASS > 0) {
        if (PASS > 1) cur = sel[PASS - 2];
        unsigned bin, below;
        topk_pick(hist + (PASS - 1) * TOPK_BINS, 1 << topk_bits(PASS - 1), cur.krem, s_scan, bin, below);
        cur.prefix = (cur.prefix << topk_bits(PASS - 1)) | bin;
        cur.krem -= below;
        if (blockIdx.x == 0 && threadIdx.x == 0) sel[PASS - 1] = cur;
    }
    return cur;
}

template <int PASS>
__global__ __launch_bounds__(BLOCK) void topk_hist_kernel(const float* __restrict__ costs, int64_t N, unsigned k,
                                                          unsigned* __restrict__ hist, TopkSel* __restrict__ sel) {
    __shared__ unsigned s_hist[TOPK_BINS];
    __shared__ unsigned s_scan[BLOCK + 2];
    constexpr int NB = 1 << topk_bits(PASS);
    for (int b = threadIdx.x; b < NB; b += BLOCK) s_hist[b] = 0u;
    const TopkSel cur = topk_enter<PASS>(hist, sel, k, s_scan);
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < N; i += (int64_t)gridDim.x * BLOCK) {
        const unsigned key = float_to_key(costs[i]);
        if (PASS == 0 || (key >> (topk_shift(PASS) + topk_bits(PASS))) == cur.prefix)
            atomicAdd(&s_hist[(key >> topk_shift(PASS)) & (NB - 1)], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < NB; b += BLOCK)
        if (s_hist[b]) atomicAdd(&hist[PASS * TOPK_BINS + b], s_hist[b]);
}

// cand[j] = (key << 32) | GLOBAL sample index for the k selected samples (unordered); counters = {#below, #ties taken}.
// The key is the cost itself (order-preserving bijection), so a candidate is self-contained: any rank can weigh and
// re-roll it without the owner's cost vector.
__global__ __launch_bounds__(BLOCK) void topk_collect_kernel(const float* __restrict__ costs, int64_t N, unsigned k,
                                                             int64_t sample_offset,
                                                             const unsigned* __restrict__ hist,
                                                             TopkSel* __restrict__ sel,
                                                             unsigned long long* __restrict__ cand,
                                                             unsigned* __restrict__ counters) {
    __shared__ unsigned s_scan[BLOCK + 2];
    const TopkSel cur = topk_enter<3>(hist, sel, k, s_scan);  // prefix = the k-th smallest key, krem = ties to take
    const unsigned nbelow = k - cur.krem;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < N; i += (int64_t)gridDim.x * BLOCK) {
        const unsigned key = float_to_key(costs[i]);
        if (key < cur.prefix) {
            const unsigned slot = atomicAdd(&counters[0], 1u);
            cand[slot] = ((unsigned long long)key << 32) | (unsigned long long)(sample_offset + i);
        } else if (key == cur.prefix) {
            const unsigned t = atomicAdd(&counters[1], 1u);
            if (t < cur.krem) cand[nbelow + t] = ((unsigned long long)key << 32) | (unsigned long long)(sample_offset + i);
        }
    }
}

// Ascending bitonic sort of one 64-bit word per thread across the block's 1024 threads, NV independent sorts in lockstep
// (v[r] of thread t = element t of row r).  Strides below 64 are wave shuffles (no barrier); only the 10 stages with a
// stride >= 64 go through LDS (s_x [NV][1024]) — a plain LDS bitonic sort pays a 1024-thread barrier for each of its 55
// stages.  first_size = 2: full sort; = 1024: the final merge only (rows that are bitonic already).
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
    const unsigned lo = __shfl_xor((unsigned)v, m), hi = __shfl_xor((unsigned)(v >> 32), m);
    return ((unsigned long long)hi << 32) | lo;
}
template <int NV>
__device__ __forceinline__ void block_bitonic_1024(unsigned long long (&v)[NV], unsigned long long* s_x, int tid, int first_size) {
    for (int size = first_size; size <= TOPK_MAX; size <<= 1) {
        const bool up = (tid & size) == 0;  // (size = 1024: every thread)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            unsigned long long o[NV];
            if (stride >= WAVE) {
#pragma unroll
                for (int r = 0; r < NV; ++r) s_x[r * TOPK_MAX + tid] = v[r];
                __syncthreads();
#pragma unroll
                for (int r = 0; r < NV; ++r) o[r] = s_x[r * TOPK_MAX + (tid ^ stride)];
                __syncthreads();
            } else {
#pragma unroll
                for (int r = 0; r < NV; ++r) o[r] = shfl_xor_u64(v[r], stride);
            }
            const bool keep_min = ((tid & stride) == 0) == up;
#pragma unroll
            for (int r = 0; r < NV; ++r) v[r] = keep_min ? (v[r] < o[r] ? v[r] : o[r]) : (v[r] > o[r] ? v[r] : o[r]);
        }
    }
}
// SORTED = false: every block of the grid (ceil(k / 64) blocks of 1024 threads) selects and sorts the same k <= TOPK_MAX
//   candidates itself and re-rolls 64 of them with ONE wave (the re-roll is a serial chain of T steps per lane, ~0.36 us per
//   step: spread over CUs, not stacked on the SIMDs of one).  The candidates are the k words of `cand` (radix select by
//   topk_hist_kernel / topk_collect_kernel, any N), or — `costs` != nullptr, n_direct <= TOPK_DIRECT_MAX samples: the sizes
//   of the reference's examples, which call get_top_samples every tick — they are selected from the costs right here (one
//   row: sorted directly; up to four rows: radix select inside the block): ONE launch instead of five;
// SORTED = true: `cand` is already ascending (topk_sort_* below: any k) and the grid's threads take one candidate each.
// lambda <= 0: the temperature the last solve's weights used (stats[4], left by finalize_tail) — no host read-back.
constexpr int TOPK_DIRECT_MAX = 4096;
template <int MODEL, int FAST, bool SORTED>
__global__ __launch_bounds__(TOPK_MAX) void topk_rollout_kernel(const unsigned long long* __restrict__ cand, int k,
                                                                const float* __restrict__ costs, int n_direct,
                                                                const float4* __restrict__ noise, bool gen_noise,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ x0,
                                                                const float* __restrict__ stats, float lambda_arg,
                                                                float* __restrict__ states,
                                                                float* __restrict__ weights,
                                                                unsigned* __restrict__ hist,
                                                                unsigned* __restrict__ counters, Dims d, GenCtx gen,
                                                                ModelCtx ctx) {
    constexpr int DS = ModelT<MODEL, FAST>::DS, DC = ModelT<MODEL, FAST>::DC;
    __shared__ unsigned long long s_key[SORTED ? 1 : TOPK_MAX];
    if (hist && blockIdx.x == 0) {  // leave the select state clean for the next call
        for (int b = threadIdx.x; b < 3 * TOPK_BINS; b += blockDim.x) hist[b] = 0u;
        if (threadIdx.x < 2) counters[threadIdx.x] = 0u;
    }
    const float lambda = lambda_arg > 0.0f ? lambda_arg : stats[4];
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long mine;
    if (!SORTED) {
        // row r of the words = samples r * 1024 + t (direct) / the k candidates (one row); padding = the largest word
        const int tid = threadIdx.x;
        const int rows = costs ? (n_direct + TOPK_MAX - 1) / TOPK_MAX : 1;
        unsigned long long v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = r * TOPK_MAX + tid;
            if (costs) v[r] = i < n_direct ? ((unsigned long long)float_to_key(costs[i]) << 32) | (unsigned long long)(d.sample_offset + i) : ~0ull;
            else v[r] = (r == 0 && tid < k) ? cand[tid] : ~0ull;
        }
        if (rows <= 1) {
            unsigned long long w1[1] = {v[0]};
            block_bitonic_1024<1>(w1, s_key, tid, 2);
            v[0] = w1[0];
        } else {
            // 2-4 rows: radix select of the k smallest keys INSIDE the block (three passes of 11 / 11 / 10 bits over the <= 4
            // keys a thread holds, histogram in LDS: the scheme of topk_hist_kernel / topk_collect_kernel without their four
            // launches), then one row to sort.  (Sorting all four rows and pruning was measured at ~25 us: 4x the work.)
            __shared__ unsigned s_hist[TOPK_BINS];
            __shared__ unsigned s_scan[TOPK_MAX + 2];
            __shared__ unsigned s_cnt[2];
            unsigned prefix = 0u, krem = (unsigned)k;
#pragma unroll
            for (int pass = 0; pass < 3; ++pass) {
                const int nb = 1 << topk_bits(pass), shift = topk_shift(pass);
                for (int b = tid; b < nb; b += TOPK_MAX) s_hist[b] = 0u;
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned key = (unsigned)(v[r] >> 32);
                    const bool valid = r * TOPK_MAX + tid < n_direct;
                    const unsigned bin = (key >> shift) & (nb - 1);
                    if (pass == 0) {
                        // the top 11 bits of costs of one solve fall into a handful of bins: one atomic per (wave, distinct bin)
                        // instead of 64 serialised ones on the same LDS word
                        unsigned long long todo = __ballot(valid);
                        while (todo) {
                            const int leader = __ffsll((long long)todo) - 1;
                            const unsigned b = __shfl(bin, leader);
                            const unsigned long long same = __ballot(valid && bin == b) & todo;
                            if ((tid & 63) == leader) atomicAdd(&s_hist[b], (unsigned)__popcll(same));
                            todo &= ~same;
                        }
                    } else if (valid && (key >> (shift + topk_bits(pass))) == prefix) {
                        atomicAdd(&s_hist[bin], 1u);
                    }
                }
                __syncthreads();
                unsigned bin, below;
                topk_pick<TOPK_MAX>(s_hist, nb, krem, s_scan, bin, below);
                prefix = (prefix << topk_bits(pass)) | bin;
                krem -= below;
            }
            // prefix = the k-th smallest key, krem = how many samples with exactly that key to take
            if (tid < 2) s_cnt[tid] = 0u;
            __syncthreads();
            const unsigned nbelow = (unsigned)k - krem;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned key = (unsigned)(v[r] >> 32);
                if (r * TOPK_MAX + tid < n_direct) {
                    if (key < prefix) s_key[atomicAdd(&s_cnt[0], 1u)] = v[r];
                    else if (key == prefix) { const unsigned t = atomicAdd(&s_cnt[1], 1u); if (t < krem) s_key[nbelow + t] = v[r]; }
                }
            }
            __syncthreads();
            unsigned long long w1[1] = {tid < k ? s_key[tid] : ~0ull};
            __syncthreads();
            block_bitonic_1024<1>(w1, s_key, tid, 2);
            v[0] = w1[0];
        }
        // Every block of the grid has sorted the same words; block b re-rolls candidates 64 b .. 64 b + 63 with ONE wave.
        // (The re-roll is a serial chain of T steps per lane, ~12 us for a lone wave; k = 300 candidates in the first five
        // waves of one block put two of them on one SIMD: 25 us.  One wave per block = one CU each.)
        s_key[tid] = v[0];
        __syncthreads();
        q = blockIdx.x * WAVE + tid;
        if (tid >= WAVE || q >= k) return;
        mine = s_key[q];
    } else {
        if (q >= k) return;
        mine = cand[q];
    }
    const uint64_t gi = mine & 0xFFFFFFFFull;            // global sample index
    const float c = key_to_float((unsigned)(mine >> 32));  // its cost
    weights[q] = expf((-c) / lambda - (-stats[0]) / lambda) / stats[1];  // softmax(-c/lambda)_i (mppi.py:376)
    const bool inherit = (int64_t)gi < d.inherit_count;
    const int64_t i = (int64_t)gi - d.sample_offset;  // local index: only meaningful when the tiles are read
    const float4* np = noise + ((i >> 6) * d.R) * 64 + (i & 63);
    float* out = states + (int64_t)q * (d.T + 1) * DS;
    int have = -1;
    float grp[4] = {0.f, 0.f, 0.f, 0.f};
    rollout_states_checked<MODEL, FAST>(x0, d.T, ctx, out, [&](int t, float* u) {
#pragma unroll
        for (int kk = 0; kk < DC; ++kk) {
            const int f = t * DC + kk;
            if ((f >> 2) != have) {
                have = f >> 2;
                const float4 n4 = gen_noise ? gen_noise4(gi, have, gen, d) : np[(int64_t)have * 64];
                grp[0] = n4.x; grp[1] = n4.y; grp[2] = n4.z; grp[3] = n4.w;
            }
            const float m = inherit ? mean[f] : 0.0f;
            const int c4 = f & 3;
            const float e = c4 == 0 ? grp[0] : c4 == 1 ? grp[1] : c4 == 2 ? grp[2] : grp[3];
            u[kk] = clampf(m + e, d.u_min[kk], d.u_max[kk]);
        }
    });
}

// Ascending sort of P = 2^m >= 2048 candidate words in global memory (k > TOPK_MAX; the tail past k holds ~0).  Bitonic:
// topk_sort_local_kernel<true> sorts every 1024-word chunk completely in LDS (all stages up to 1024, direction by the
// chunk's position), then for size = 2048, 4096, ... P the strides >= 1024 are one global compare-exchange pass each
// (topk_sort_global_kernel) and the strides 512 ... 1 of that stage run in LDS again (topk_sort_local_kernel<false>).
template <bool FULL>
__global__ __launch_bounds__(TOPK_MAX) void topk_sort_local_kernel(unsigned long long* __restrict__ cand, int size_arg) {
    __shared__ unsigned long long s_key[TOPK_MAX];
    const int g = blockIdx.x * TOPK_MAX + threadIdx.x;
    s_key[threadIdx.x] = cand[g];
    __syncthreads();
    for (int size = FULL ? 2 : size_arg; size <= (FULL ? TOPK_MAX : size_arg); size <<= 1) {
        for (int stride = min(size >> 1, TOPK_MAX >> 1); stride > 0; stride >>= 1) {
            const int j = threadIdx.x ^ stride;
            if (j > (int)threadIdx.x) {
                const unsigned long long a = s_key[threadIdx.x], b = s_key[j];
                const bool up = (g & size) == 0;
                if ((a > b) == up) { s_key[threadIdx.x] = b; s_key[j] = a; }
            }
            __syncthreads();
        }
    }
    cand[g] = s_key[threadIdx.x];
}
__global__ __launch_bounds__(BLOCK) void topk_sort_global_kernel(unsigned long long* __restrict__ cand, int P, int size,
                                                                 int stride) {
    const int t = blockIdx.x * BLOCK + threadIdx.x;  // one thread per pair
    if (t >= P / 2) return;
    const int i = ((t / stride) * 2 * stride) + (t % stride), j = i + stride;
    const unsigned long long a = cand[i], b = cand[j];
    const bool up = (i & size) == 0;
    if ((a > b) == up) { cand[i] = b; cand[j] = a; }
}
__global__ __launch_bounds__(BLOCK) void topk_pad_kernel(unsigned long long* __restrict__ cand, int k, int P) {
    const int t = k + blockIdx.x * BLOCK + threadIdx.x;
    if (t < P) cand[t] = ~0ull;
}

// ObstacleMap.compute_cost / LaneMap.compute_cost (src/envs/obstacle_map_2d.py:168-200, src/envs/lane_map_2d.py:90-122) for
// callers OUTSIDE the solver — env.collision_check of the examples' loops, cost plugins on the generic path: one thread
// per point, the reference's arithmetic (fp32 division by the cell size, + origin, round half to even, out of the grid
// = 1, else the map's value) instead of ~15 torch launches.
__global__ __launch_bounds__(BLOCK) void grid_lookup_kernel(const float* __restrict__ map, int nx, int ny, float cell_size,
                                                            float ox, float oy, const float* __restrict__ xy, int64_t n,
                                                            int64_t stride, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const float qx = rintf(xy[i * stride] / cell_size + ox), qy = rintf(xy[i * stride + 1] / cell_size + oy);
    const bool inb = qx >= 0.0f && qx < (float)nx && qy >= 0.0f && qy < (float)ny;  // (NaN: out of the grid, like .long() of it)
    out[i] = inb ? map[(int64_t)qx * ny + (int64_t)qy] : 1.0f;
}

// ------------------------------------------------------------------------------------------
// The racing control tick without the host (example/racing.py:161-218,221-266).
//
// racing_controller.calc_ref_trajectory: nearest centre-line point to the vehicle (the reference's Python
// `min(range(len(path)), key=np.hypot(...))`: FIRST minimum of the fp32 hypot), `ind = max(cind, ind)` with the index
// carried from tick to tick, then the T+1 window rows path[ind + dind[i]] (dind = int(round(travel / DL)), a function
// of the call's constants: computed once by the caller), target speed V_MAX — or, once the window runs past the end of
// the course, the last point repeated and the WHOLE speed column zeroed (`xref[:, 3] = 0.0`, :213-216).
// One block: thread i computes the distance of points i, i + 1024, ...; np.hypot on float32 is glibc's hypotf =
// (float)sqrt((double)dx*dx + (double)dy*dy) (checked bit for bit on 10^7 inputs, tests/test_host_logic.py), which
// the fp64 units reproduce exactly; the (distance bits, index) pair is reduced as one 64-bit key, so ties resolve to
// the lowest index like the reference's min().  Rows are written in the layout the rollout kernel reads
// (ModelCtx::ref: x, y, yaw, v, sin yaw, cos yaw) from a per-point table whose sin/cos were evaluated on the host by
// the same calls mppi_set_reference makes: the window is bit-identical to the host path's.
// The vehicle state is read from device memory and the path index lives there: no host synchronisation per tick.
struct RefWindowCtx {
    const float* path8;   // [n][8] = x, y, yaw, 0, sin(yaw), cos(yaw), 0, 0
    const int32_t* dind;  // [rows] index offsets of the window rows (monotone)
    int32_t* cind;        // current path index (`racing_controller.current_path_index`)
    int32_t n, rows;
    float v_target;       // env.V_MAX
};
constexpr int REFWIN_BLOCK = 1024;
__global__ __launch_bounds__(REFWIN_BLOCK) void ref_window_kernel(RefWindowCtx w, const float* __restrict__ state,
                                                                  float* __restrict__ ref_out /*[rows][8]*/) {
    __shared__ unsigned long long s_best[REFWIN_BLOCK / WAVE];
    __shared__ int s_ind;
    const float sx = state[0], sy = state[1];
    unsigned long long best = ~0ull;
    for (int i = threadIdx.x; i < w.n; i += REFWIN_BLOCK) {
        const float2 p = *reinterpret_cast<const float2*>(w.path8 + 8 * (int64_t)i);
        const float dx = p.x - sx, dy = p.y - sy;  // fp32 differences, like the reference's numpy scalars
        const float hd = (float)sqrt((double)dx * (double)dx + (double)dy * (double)dy);
        const unsigned long long key = ((unsigned long long)__float_as_uint(hd) << 32) | (unsigned)i;  // hd >= 0
        best = key < best ? key : best;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned long long o = __shfl_xor(best, m);
        best = o < best ? o : best;
    }
    if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = s_best[0];
#pragma unroll
        for (int q = 1; q < REFWIN_BLOCK / WAVE; ++q) b = s_best[q] < b ? s_best[q] : b;
        const int ind = max(*w.cind, (int)(unsigned)b);  // "ensure the index is not less than the current index"
        *w.cind = ind;
        s_ind = ind;
    }
    __syncthreads();
    const int ind = s_ind;
    const bool all_inside = ind + w.dind[w.rows - 1] < w.n;  // dind is increasing
    for (int i = threadIdx.x; i < w.rows; i += REFWIN_BLOCK) {
        const int idx = ind + w.dind[i];
        const float4* src = reinterpret_cast<const float4*>(w.path8 + 8 * (int64_t)(idx < w.n ? idx : w.n - 1));
        float4 a = src[0];
        const float4 b = src[1];
        a.w = all_inside ? w.v_target : 0.0f;
        float4* dst = reinterpret_cast<float4*>(ref_out + 8 * (int64_t)i);
        dst[0] = a;
        dst[1] = b;
    }
}

// RacingEnv.step / Navigation2DEnv.step (src/envs/racing_env.py:142-163, navigation_2d.py): the plant's batch-1
// dynamics call as ONE launch instead of ~20 batch-1 torch kernels — next = dynamics(state, clamp(u)) with the library
// math in the reference's operation order (the FAST=false functor the parity pins cover), plus the goal test
// `norm(next[:2] - goal) < threshold`.  `state` and `next` may alias.
struct StepBounds { float lo[MPPI_MAX_DIM_CONTROL], hi[MPPI_MAX_DIM_CONTROL]; };
template <int MODEL>
__global__ __launch_bounds__(WAVE) void model_step_kernel(ModelCtx ctx, const float* __restrict__ state,
                                                          const float* __restrict__ action, StepBounds ub,
                                                          float* next, float gx, float gy, float goal_threshold,
                                                          uint8_t* __restrict__ reached) {
    using M = ModelT<MODEL, false>;
    if (threadIdx.x != 0) return;
    float s[M::DS], u[M::DC], sn[M::DS], ss[M::DS];
#pragma unroll
    for (int j = 0; j < M::DS; ++j) s[j] = state[j];
#pragma unroll
    for (int k = 0; k < M::DC; ++k) u[k] = fminf(fmaxf(action[k], ub.lo[k]), ub.hi[k]);  // `torch.clamp(u, u_min, u_max)`
    bool bad = false;
    M::step(ctx, s, u, sn, ss, bad);
#pragma unroll
    for (int j = 0; j < M::DS; ++j) next[j] = sn[j];
    if (reached) {
        const float dx = sn[0] - gx, dy = sn[1] - gy;
        *reached = sqrtf(dx * dx + dy * dy) < goal_threshold ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------
// Layout conversions between the reference layout [N][T][dc] and the lane-major tiles, staged
// through LDS so that both the global reads and the global writes are coalesced.
// One block per (tile, 128-float column chunk); LDS row stride 129 floats (bank-conflict free).
constexpr int CONV_COLS = 128;
__global__ __launch_bounds__(BLOCK) void inject_kernel(const float* __restrict__ eps, float4* __restrict__ noise,
                                                       Dims d) {
    __shared__ float tilebuf[64][CONV_COLS + 1];
    const int64_t tile = blockIdx.x;
    const int c0 = blockIdx.y * CONV_COLS;
    const int nc = min(CONV_COLS, d.row - c0);
    for (int idx = threadIdx.x; idx < 64 * CONV_COLS; idx += BLOCK) {
        const int l = idx / CONV_COLS, cc = idx % CONV_COLS;
        const int64_t i = tile * 64 + l;
        float v = 0.0f;
        if (i < d.N && cc < nc) v = eps[i * d.row + c0 + cc];
        tilebuf[l][cc] = v;
    }
    __syncthreads();
    const int ngroups = (nc + 3) / 4;
    for (int idx = threadIdx.x; idx < 64 * ngroups; idx += BLOCK) {
        const int g = idx / 64, l = idx % 64;
        const float4 v = make_float4(tilebuf[l][4 * g], tilebuf[l][4 * g + 1], tilebuf[l][4 * g + 2],
                                     tilebuf[l][4 * g + 3]);
        noise[(tile * d.R + (c0 / 4) + g) * 64 + l] = v;
    }
}

__global__ __launch_bounds__(BLOCK) void export_kernel(const float4* __restrict__ noise,
                                                       const float* __restrict__ mean, float* __restrict__ eps_out,
                                                       float* __restrict__ act_out, Dims d,
                                                       const float* __restrict__ coltab /* wide rows, else null */) {
    __shared__ float tilebuf[64][CONV_COLS + 1];
    const int64_t tile = blockIdx.x;
    const int c0 = blockIdx.y * CONV_COLS;
    const int nc = min(CONV_COLS, d.row - c0);
    const int ngroups = (nc + 3) / 4;
    const int dc = d.dc;
    for (int idx = threadIdx.x; idx < 64 * ngroups; idx += BLOCK) {
        const int g = idx / 64, l = idx % 64;
        const float4 v = noise[(tile * d.R + (c0 / 4) + g) * 64 + l];
        tilebuf[l][4 * g] = v.x; tilebuf[l][4 * g + 1] = v.y; tilebuf[l][4 * g + 2] = v.z; tilebuf[l][4 * g + 3] = v.w;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * CONV_COLS; idx += BLOCK) {
        const int l = idx / CONV_COLS, cc = idx % CONV_COLS;
        const int64_t i = tile * 64 + l;
        if (i < d.N && cc < nc) {
            const float e = tilebuf[l][cc];
            const int f = c0 + cc;
            if (eps_out) eps_out[i * d.row + f] = e;
            if (act_out) {
                const bool inherit = (d.sample_offset + i) < d.inherit_count;
                const float m = inherit ? mean[f] : 0.0f;
                float lo, hi;
                if (coltab) { lo = coltab[4 * d.R + f]; hi = coltab[8 * d.R + f]; }
                else { const int k = f % dc; lo = d.u_min[k]; hi = d.u_max[k]; }
                act_out[i * d.row + f] = clampf(m + e, lo, hi);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Map construction on the device (integer / byte work, bit-exact with the reference's host loops).
// Grids are cells[ix * ny + iy]; consecutive lanes own consecutive iy, so stores are coalesced.

// ObstacleMap.add_circle_obstacle / add_rectangle_obstacle (obstacle_map_2d.py:103-158) for a whole
// obstacle list at once.  circles[c] = (ci, cj, r) in cells; the reference writes the disc
// {(i,j): i^2+j^2 <= r^2} at clip(ci+i), clip(cj+j), so a border cell collects every disc cell that was
// clipped onto it: the cell is set iff the disc offset of SMALLEST magnitude that maps onto it is inside.
// rects[q] = (x0, x1, y0, y1): the already clipped half-open slice map[x0:x1, y0:y1] = 1.
__global__ __launch_bounds__(BLOCK) void raster_obstacles_kernel(uint8_t* __restrict__ cells, int nx, int ny,
                                                                 const int32_t* __restrict__ circles, int n_circles,
                                                                 const int32_t* __restrict__ rects, int n_rects) {
    const int iy = blockIdx.x * BLOCK + threadIdx.x;
    const int ix = blockIdx.y;
    if (iy >= ny) return;
    bool occ = false;
    for (int c = 0; c < n_circles; ++c) {
        const int ci = circles[3 * c], cj = circles[3 * c + 1], r = circles[3 * c + 2];
        // offsets i with clip(ci + i, 0, nx-1) == ix form [lo, hi]; the one closest to 0 decides
        const int lo_i = (ix == 0) ? INT32_MIN / 2 : ix - ci, hi_i = (ix == nx - 1) ? INT32_MAX / 2 : ix - ci;
        const int lo_j = (iy == 0) ? INT32_MIN / 2 : iy - cj, hi_j = (iy == ny - 1) ? INT32_MAX / 2 : iy - cj;
        const int64_t i = min(max(0, lo_i), hi_i), j = min(max(0, lo_j), hi_j);
        occ |= i * i + j * j <= (int64_t)r * r;
    }
    for (int q = 0; q < n_rects; ++q)
        occ |= (ix >= rects[4 * q]) && (ix < rects[4 * q + 1]) && (iy >= rects[4 * q + 2]) && (iy < rects[4 * q + 3]);
    cells[(size_t)ix * ny + iy] = occ ? 1 : 0;
}

// LaneMap.populate_map (lane_map_2d.py:68-88): seeds = centre-line cells; the Euclidean distance transform
// of the seed grid is sqrt(min over seeds of the integer squared cell distance), and `distance <= max_distance`
// is the integer test d2 <= max_d2 with max_d2 = the largest k whose float64 sqrt is <= max_distance (host).
// Brute force over the seeds staged through LDS: nx*ny*ns integer mads (2.4e9 for the racing map).
constexpr int LANE_CHUNK = 1024;
__global__ __launch_bounds__(BLOCK) void lane_map_kernel(uint8_t* __restrict__ cells, int nx, int ny,
                                                         const int32_t* __restrict__ seeds, int n_seeds,
                                                         int64_t max_d2) {
    __shared__ int32_t sx[LANE_CHUNK], sy[LANE_CHUNK];
    const int iy = blockIdx.x * BLOCK + threadIdx.x;
    const int ix = blockIdx.y;
    int64_t best = INT64_MAX;
    for (int base = 0; base < n_seeds; base += LANE_CHUNK) {
        const int n = min(LANE_CHUNK, n_seeds - base);
        __syncthreads();
        for (int k = threadIdx.x; k < n; k += BLOCK) {
            sx[k] = seeds[2 * (base + k)];
            sy[k] = seeds[2 * (base + k) + 1];
        }
        __syncthreads();
        for (int k = 0; k < n; ++k) {
            const int64_t dx = ix - sx[k], dy = iy - sy[k];
            best = min(best, dx * dx + dy * dy);
        }
    }
    if (iy < ny) cells[(size_t)ix * ny + iy] = (best <= max_d2) ? 0 : 1;
}

// The grid of the FAST lookup (occ_lookup_pad): a (nx+1) x (ny+1) copy of the occupancy grid — for racing the
// obstacle and lane grids summed per cell (0..2), one gather instead of two — whose extra row and column hold
// the out-of-bounds value.
__global__ __launch_bounds__(BLOCK) void pad_map_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                                        int nx, int ny, uint8_t oob, uint8_t* __restrict__ out) {
    const int iy = blockIdx.x * BLOCK + threadIdx.x;
    const int ix = blockIdx.y;
    if (iy > ny) return;
    uint8_t v = oob;
    if (ix < nx && iy < ny) {
        v = a[(size_t)ix * ny + iy];
        if (b) v = (uint8_t)(v + b[(size_t)ix * ny + iy]);
    }
    out[(size_t)ix * (ny + 1) + iy] = v;
}

}  // namespace mppi
