// mppi_finalize.hpp — Fold of the partial rows (summarize_kernel), combine of the shard summaries, normalise, Savitzky-Golay step, warm start, batch-1 rollout (mppi.py:381-452,508-524): finalize_kernel, state_seq_kernel.
// Part of the MPPI.forward() hot path for gfx950; see mppi_kernels.hpp for the map of the files.
#pragma once
#include "mppi_exchange.hpp"
#include "mppi_reduce.hpp"

namespace mppi {

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
// The same rollout driven by a sample's NOISE, software-pipelined like the rollout kernel (round 5): LOADG(g) returns float4 group
// g of the sample's noise row (regenerated or read from the tiles — the caller picks, no branch in here); the group of the NEXT
// 4 / DC steps is requested at the top of the loop body and the current group's steps follow in the SAME basic block, so that a
// lone wave overlaps the Philox + Box-Muller chain with the state recurrence (with the action fetched step by step behind an
// "is this a new group?" branch the two chains ran one after the other: 0.36 us per step in get_top_samples' re-roll).
// `mp`: the mean row as float4 groups in LDS (zeros past the row; an all-zero copy for samples beyond the exploration split,
// mppi.py:266-270).  u = clamp(mean + eps) exactly as rollout_cost_kernel forms it; same step functions, same order: the same
// bits as rollout_states with a per-step GETU.
template <int MODEL, int FAST, class LOADG>
__device__ __forceinline__ bool rollout_states_noise(const float* __restrict__ x0, const Dims& d, const ModelCtx& ctx,
                                                     float* __restrict__ out, const float4* mp, LOADG loadg) {
    using M = ModelT<MODEL, FAST>;
    constexpr int DS = M::DS, DC = M::DC, SPG = 4 / DC;  // steps per float4 group
    static_assert(DC == 1 || DC == 2 || DC == 4, "a step's controls lie inside one float4 group");
    const int T = d.T;
    bool bad = false;
    float s[DS];
#pragma unroll
    for (int j = 0; j < DS; ++j) s[j] = x0[j];
    constexpr bool GENERAL = FAST != 0 && EntryGeneral<M>::value;
    float raw_heading = 0.0f;
    if constexpr (GENERAL) { raw_heading = s[2]; M::enter_any(s); }
    else if (FAST) M::check_state(ctx, s, bad);
    const auto step = [&](int t, const float4& e, const float4& m, int j) {
        const float e4[4] = {e.x, e.y, e.z, e.w}, m4[4] = {m.x, m.y, m.z, m.w};
        float u[DC], sn[DS], ss[DS];
#pragma unroll
        for (int kk = 0; kk < DC; ++kk) u[kk] = clampf(m4[j * DC + kk] + e4[j * DC + kk], d.u_min[kk], d.u_max[kk]);
        if constexpr (GENERAL) {
            M::step(ctx, s, u, sn, ss, bad, false, true);
            if (t == 0) ss[2] = raw_heading;
        } else {
            M::step(ctx, s, u, sn, ss, bad);
        }
#pragma unroll
        for (int j2 = 0; j2 < DS; ++j2) { out[t * DS + j2] = ss[j2]; s[j2] = sn[j2]; }
    };
    float4 nxt = loadg(0);
    const int gfull = T / SPG;  // groups whose SPG steps all exist
    for (int g = 0; g < gfull; ++g) {
        const float4 cur = nxt;
        nxt = loadg(g + 1 < d.R ? g + 1 : g);  // (the last request repeats a group: nobody reads it)
        const float4 m = mp[g];
#pragma unroll
        for (int j = 0; j < SPG; ++j) step(g * SPG + j, cur, m, j);
    }
    if (gfull * SPG < T) {  // the row's last, partial group
        const float4 m = mp[gfull];
        for (int j = 0; gfull * SPG + j < T; ++j) step(gfull * SPG + j, nxt, m, j);
    }
    if constexpr (GENERAL) { if (T == 0) s[2] = raw_heading; }
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

}  // namespace mppi
