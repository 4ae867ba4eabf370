// mppi_rollout.hpp — Steps 1b-3 (mppi.py:266-336): clamp, N x T rollout and stage / terminal costs — rollout_cost_kernel (lane per trajectory) and the literal wavefront-per-trajectory variant.
// Part of the MPPI.forward() hot path for gfx950; see mppi_kernels.hpp for the map of the files.
#pragma once
#include "mppi_sample.hpp"

namespace mppi {

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

}  // namespace mppi
