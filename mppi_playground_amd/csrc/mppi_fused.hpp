// mppi_fused.hpp — MPPI.forward() as ONE cooperative launch for small problems: solve_fused_kernel.
// Part of the MPPI.forward() hot path for gfx950; see mppi_kernels.hpp for the map of the files.
#pragma once
#include "mppi_rollout.hpp"
#include "mppi_finalize.hpp"
#include "mppi_search.hpp"

namespace mppi {

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

}  // namespace mppi
