// mppi_topk.hpp — Queries after a solve: weights, re-rolls of given actions / samples, get_top_samples (mppi.py:462-487: radix select, sort, re-roll).
// Part of the MPPI.forward() hot path for gfx950; see mppi_kernels.hpp for the map of the files.
#pragma once
#include "mppi_rollout.hpp"
#include "mppi_finalize.hpp"

namespace mppi {

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

// Wave-level prefix sums / reductions through DPP (row_shr within the rows of 16 lanes, then row_bcast15 / row_bcast31 across
// them — CDNA keeps both): six VALU instructions with a DPP operand, against six ds_bpermute round trips (~100 cycles of
// latency each) for the __shfl forms.  A lane whose source lies outside its row (or whose row is masked off) gets `old`.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ unsigned dpp_u32(unsigned old, unsigned src) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ unsigned wave_inclusive_sum_dpp(unsigned x) {
    x += dpp_u32<0x111>(0u, x);  // row_shr:1
    x += dpp_u32<0x112>(0u, x);  // row_shr:2
    x += dpp_u32<0x114>(0u, x);  // row_shr:4
    x += dpp_u32<0x118>(0u, x);  // row_shr:8  -> inclusive sums inside every row
    x += dpp_u32<0x142, 0xA>(0u, x);  // row_bcast15 into rows 1 and 3: the total of the row before
    x += dpp_u32<0x143, 0xC>(0u, x);  // row_bcast31 into rows 2 and 3: the total of lanes 0 .. 31
    return x;
}
// op over the wave's 64 values, the same in every lane (lane 63 holds it after the six steps)
template <class OP>
__device__ __forceinline__ float wave_reduce_dpp(float v, OP op) {
    const auto step = [&](auto ctrl, auto rows) {
        const unsigned u = __float_as_uint(v);
        v = op(v, __uint_as_float(dpp_u32<decltype(ctrl)::value, decltype(rows)::value>(u, u)));
    };
    using std::integral_constant;
    step(integral_constant<int, 0x111>{}, integral_constant<int, 0xF>{});
    step(integral_constant<int, 0x112>{}, integral_constant<int, 0xF>{});
    step(integral_constant<int, 0x114>{}, integral_constant<int, 0xF>{});
    step(integral_constant<int, 0x118>{}, integral_constant<int, 0xF>{});
    step(integral_constant<int, 0x142>{}, integral_constant<int, 0xA>{});
    step(integral_constant<int, 0x143>{}, integral_constant<int, 0xC>{});
    return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}
// topk_pick for 2048 bins and 1024 threads (the in-block select of topk_rollout_kernel): two bins per thread, the wave's prefix
// sums through DPP, the sixteen wave totals through LDS — two barriers instead of four and no serial tail.  Also leaves the
// exclusive prefix sums of ALL bins in s_pref[0 .. 2048] (s_pref[2048] = the number of keys) and reports whether a bin up to
// the chosen one holds more than `crowd` keys.
__device__ __forceinline__ void topk_pick_2048(const unsigned* __restrict__ hist, unsigned krem, unsigned* __restrict__ s_part /*[19]*/,
                                               unsigned* __restrict__ s_pref /*[2049]*/, unsigned crowd, int tid, unsigned& bin,
                                               unsigned& below, bool& crowded) {
    const uint2 h = *reinterpret_cast<const uint2*>(hist + 2 * tid);
    const unsigned loc = h.x + h.y, incl = wave_inclusive_sum_dpp(loc);
    if ((tid & 63) == 63) s_part[tid >> 6] = incl;
    if (tid == 0) s_part[18] = 0u;
    __syncthreads();
    unsigned before = 0u;  // the totals of the waves below this one
#pragma unroll
    for (int w = 0; w < TOPK_MAX / WAVE; ++w) before += w < (tid >> 6) ? s_part[w] : 0u;
    const unsigned excl = before + incl - loc;
    *reinterpret_cast<uint2*>(s_pref + 2 * tid) = make_uint2(excl, excl + h.x);
    if (tid == TOPK_MAX - 1) s_pref[2 * TOPK_MAX] = excl + loc;
    if (excl < krem && krem <= excl + loc) {  // exactly one thread (loc > 0 there)
        const bool first = krem <= excl + h.x;
        s_part[16] = (unsigned)(2 * tid) + (first ? 0u : 1u);
        s_part[17] = first ? excl : excl + h.x;
    }
    // (a bin lies at or below the chosen one exactly when keys of it come before the krem-th: its exclusive prefix < krem)
    if ((h.x > crowd && excl < krem) || (h.y > crowd && excl + h.x < krem)) s_part[18] = 1u;
    __syncthreads();
    bin = s_part[16];
    below = s_part[17];
    crowded = s_part[18] != 0u;
}
// Ascending sort of the n <= 1024 DISTINCT words a block of 1024 threads holds (thread t < n: word w) — round 5, replaces the
// 55-stage bitonic network over 1024 padded words (10 of its stages through LDS behind 1024-thread barriers: 8.8 us of the
// examples' 20 us get_top_samples kernel, measured with phase stamps: load 1.8, radix select 3.6, compaction 0.8, sort 8.8,
// re-roll 4.9 us).  Merge by rank in two levels:
//   (1) inside a run of 64 words (one wave): every thread counts the words of its run below its own — 64 broadcast LDS reads
//       that do not depend on each other — and stores its word at that position of the run in the second buffer (an in-wave
//       bitonic network is 21 stages of two cross-lane permutes each, ~100 cycles of latency apiece: 3.3 us of this kernel
//       under the phase stamps of -DMPPI_TOPK_TRACE, against ~1 us this way);
//   (2) across runs: how many words of each OTHER (now sorted) run lie below its own, by a branch-free binary search (7 LDS
//       reads per run, eight runs in flight); rank = position in the run + those counts.
// Three barriers in all; only the ceil(n / 64) runs that hold words are looked at.  Returns the t-th smallest word (~0 for
// t >= n); the sorted words are left in s_x[0 .. n).  s_y: a second buffer of 1024 words.
__device__ __forceinline__ unsigned long long block_rank_sort_1024(unsigned long long w, int n, unsigned long long* s_x,
                                                                   unsigned long long* s_y, int tid) {
    const int nruns = (n + WAVE - 1) / WAVE, my = tid >> 6;
    if (tid >= n) w = ~0ull;
    s_x[tid] = w;
    __syncthreads();
    int rank = 0;
    if (my < nruns) {  // (wave-uniform)
        const ulonglong2* own = reinterpret_cast<const ulonglong2*>(s_x + my * WAVE);
#pragma unroll
        for (int j = 0; j < WAVE / 2; ++j) {
            const ulonglong2 p = own[j];
            rank += p.x < w ? 1 : 0;
            rank += p.y < w ? 1 : 0;
        }
        // (the words of a run's tail that are padding sit in its last lanes: they keep their places)
        s_y[w != ~0ull ? my * WAVE + rank : tid] = w;
    }
    __syncthreads();
    if (w != ~0ull) {
        // eight runs at a time: the seven reads of one search depend on each other (7 LDS latencies), those of different
        // runs do not (a run past the last one repeats it and does not count, nor does the thread's own)
        constexpr int G = 8;
        for (int r0 = 0; r0 < nruns; r0 += G) {
            const unsigned long long* run[G];
            int pos[G];
#pragma unroll
            for (int j = 0; j < G; ++j) { run[j] = s_y + min(r0 + j, nruns - 1) * WAVE; pos[j] = 0; }
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) {
#pragma unroll
                for (int j = 0; j < G; ++j) pos[j] += run[j][pos[j] + step - 1] < w ? step : 0;
            }
#pragma unroll
            for (int j = 0; j < G; ++j) {
                pos[j] += run[j][pos[j]] < w ? 1 : 0;  // (pos = 63 here when all of the first 63 are smaller)
                rank += (r0 + j < nruns && r0 + j != my) ? pos[j] : 0;
            }
        }
        s_x[rank] = w;  // (nobody reads s_x any more: its last readers passed the barrier above)
    }
    __syncthreads();
    return s_x[tid];
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
// -DMPPI_TOPK_TRACE (experiments only, scripts/build_variant.sh): block 0 stamps the 100 MHz clock at the phase boundaries of
// topk_rollout_kernel<.., false> and prints the differences (10 ns units) when it is done
#ifdef MPPI_TOPK_TRACE
#define TK_TRACE(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) s_tk[i] = (long long)wall_clock64(); } while (0)
#else
#define TK_TRACE(i) do { } while (0)
#endif
constexpr int TOPK_PREGEN_MAX_R = 32;  // rows of up to 32 float4 groups (T <= 64 at two controls): 32 KiB of LDS
// dynamic LDS of topk_rollout_kernel: the two mean rows [8R floats] + (unsorted candidates, regen mode) the block's noise
inline size_t topk_rollout_lds(int R, bool sorted, bool gen_noise) {
    return sizeof(float) * 8 * (size_t)R + (!sorted && gen_noise && R <= TOPK_PREGEN_MAX_R ? sizeof(float4) * 64 * (size_t)R : 0);
}
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
    constexpr int DS = ModelT<MODEL, FAST>::DS;
    __shared__ __attribute__((aligned(16))) unsigned long long s_key[SORTED ? 1 : TOPK_MAX];
    __shared__ __attribute__((aligned(16))) unsigned long long s_key2[SORTED ? 1 : TOPK_MAX];  // the sort's second buffer; the select's histogram before that
    // [4R] the mean row the solve sampled around (zeros past the row), [4R] zeros (samples beyond the exploration split): the
    // re-roll reads its mean groups from here (rollout_states_noise)
    extern __shared__ __attribute__((aligned(16))) float s_mrow[];
    __shared__ float s_x0[DS];
#ifdef MPPI_TOPK_TRACE
    __shared__ long long s_tk[10];
    TK_TRACE(0);
#endif
    // (the unsorted path calls this AFTER it has requested its candidates: one memory round trip for both, and the barrier
    // that publishes the sorted words publishes these as well)
    const auto stage_inputs = [&]() {
        for (int f = threadIdx.x; f < 4 * d.R; f += blockDim.x) {
            s_mrow[f] = f < d.row ? mean[f] : 0.0f;
            s_mrow[4 * d.R + f] = 0.0f;
        }
        if (threadIdx.x < DS) s_x0[threadIdx.x] = x0[threadIdx.x];
    };
    if (SORTED) { stage_inputs(); __syncthreads(); }
    // (dynamic LDS behind the mean rows: the block's noise, see below; sized by topk_rollout_lds())
    const bool pregen = !SORTED && gen_noise && d.R <= TOPK_PREGEN_MAX_R;
    float4* s_noise = reinterpret_cast<float4*>(s_mrow + 8 * d.R);
    if (hist && blockIdx.x == 0) {  // leave the select state clean for the next call
        for (int b = threadIdx.x; b < 3 * TOPK_BINS; b += blockDim.x) hist[b] = 0u;
        if (threadIdx.x < 2) counters[threadIdx.x] = 0u;
    }
    const float lambda = lambda_arg > 0.0f ? lambda_arg : stats[4];
    const float st_min = stats[0], st_sum = stats[1];  // (requested here: not a round trip of their own before the re-roll)
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
        stage_inputs();
        if (rows <= 1) {
            v[0] = block_rank_sort_1024(v[0], costs ? n_direct : k, s_key, s_key2, tid);
        } else {
            // 2-4 rows: radix select of the k smallest keys INSIDE the block (three passes of 11 / 11 / 10 bits over the <= 4
            // keys a thread holds, histogram in LDS: the scheme of topk_hist_kernel / topk_collect_kernel without their four
            // launches), then one row to sort.  (Sorting all four rows and pruning was measured at ~25 us: 4x the work.)
            static_assert(sizeof(unsigned) * TOPK_BINS <= sizeof(unsigned long long) * TOPK_MAX, "the histogram lives in s_key2");
            unsigned* s_hist = reinterpret_cast<unsigned*>(s_key2);
            __shared__ unsigned s_scan[TOPK_MAX + 2];
            __shared__ __attribute__((aligned(8))) unsigned s_pref[TOPK_BINS + 2];  // exclusive prefix sums of the value bins
            __shared__ unsigned s_cnt[2];
            __shared__ float s_mm[2][TOPK_MAX / WAVE];
            // (a) ONE pass over bins of the cost VALUE (round 5): 2048 bins, monotone in the cost, between the block's minimum
            // and maximum (see bin_of below) — the top 11 BITS of the keys of one solve fall into a handful of bins (costs of
            // 77 k .. 110 k share their exponent), so the bitwise select below needs all three passes; a value bin holds a few
            // keys around the k-th smallest.  Every key in a lower bin is smaller than every key of that bin, so {keys in bins <= b*} is a
            // superset of the k smallest: when it fits one row it is sorted — by counting when no bin is crowded, else by the
            // merge by rank — and the first k are the answer (exact).  Costs that are all equal, infinite or NaN, or more than
            // one row of keys up to the boundary bin, take the bitwise passes (b).
            bool done = false;
            {
                float cmn = INFINITY, cmx = -INFINITY, cf[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool valid = r * TOPK_MAX + tid < n_direct;
                    cf[r] = valid ? key_to_float((unsigned)(v[r] >> 32)) : 0.0f;
                    if (valid) { cmn = fminf(cmn, cf[r]); cmx = fmaxf(cmx, cf[r]); if (cf[r] != cf[r]) cmx = INFINITY; }  // (a NaN: not usable)
                }
                cmn = wave_reduce_dpp(cmn, [](float a, float b) { return fminf(a, b); });
                cmx = wave_reduce_dpp(cmx, [](float a, float b) { return fmaxf(a, b); });
                if ((tid & 63) == 0) { s_mm[0][tid >> 6] = cmn; s_mm[1][tid >> 6] = cmx; }
                for (int b = tid; b < TOPK_BINS; b += TOPK_MAX) s_hist[b] = 0u;
                if (tid < 2) s_cnt[tid] = 0u;
                __syncthreads();
#pragma unroll
                for (int w = 0; w < TOPK_MAX / WAVE; ++w) { cmn = fminf(cmn, s_mm[0][w]); cmx = fmaxf(cmx, s_mm[1][w]); }
                TK_TRACE(1);  // the candidates' costs have arrived, their range is known
                // bin = the exponent and the first six mantissa bits of (cost - min), counted down from those of (max - min):
                // 64 bins per octave over the 32 octaves below the range's top (anything smaller: bin 0) — monotone in the cost
                // like equal bins, but a handful of collision penalties (+10^4 per step: costs of 300 .. 250 000 in a running
                // racing loop) no longer push every candidate into bins 0 and 1 (400 .. 1 300 keys there: measured)
                const bool usable = cmx > cmn && cmx < INFINITY && cmn > -INFINITY && (cmx - cmn) < INFINITY;  // (block-uniform)
                const int bin_base = (int)(__float_as_uint(cmx - cmn) >> 17) - (TOPK_BINS - 1);
                const auto bin_of = [&](float c) { return min(max((int)(__float_as_uint(c - cmn) >> 17) - bin_base, 0), TOPK_BINS - 1); };
                if (usable) {
                    int vb[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        vb[r] = bin_of(cf[r]);
                        if (r * TOPK_MAX + tid < n_direct) atomicAdd(&s_hist[vb[r]], 1u);
                    }
                    __syncthreads();
                    TK_TRACE(2);  // histogram
                    unsigned bstar, below;
                    bool crowded;
                    static_assert(TOPK_BINS == 2 * TOPK_MAX, "topk_pick_2048: two bins per thread");
                    topk_pick_2048(s_hist, (unsigned)k, s_scan, s_pref, 64u, tid, bstar, below, crowded);
                    TK_TRACE(3);  // the boundary bin
                    const unsigned total = s_pref[bstar + 1];  // keys in bins <= b*  (>= k)
                    if (total <= (unsigned)TOPK_MAX && !crowded) {  // (block-uniform)
                        // Counting sort (no bin up to b* holds more than 64 keys; a few at the examples' sizes): the histogram is
                        // its own set of cursors — a key takes the next free place of its bin's segment [pref[b], pref[b + 1]) —
                        // and the words, now ordered by bin, find their rank inside their segment by looking at its few words.
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (r * TOPK_MAX + tid < n_direct && vb[r] <= (int)bstar)
                                s_key[s_pref[vb[r]] + atomicSub(&s_hist[vb[r]], 1u) - 1u] = v[r];
                        __syncthreads();
                        TK_TRACE(4);  // scatter by bin
                        unsigned long long w = ~0ull;
                        unsigned rank = (unsigned)tid;  // (threads without a word pad their own places)
                        if (tid < (int)total) {
                            w = s_key[tid];
                            const int bin = bin_of(key_to_float((unsigned)(w >> 32)));
                            const unsigned lo = s_pref[bin], hi = s_pref[bin + 1];
                            rank = lo;
                            for (unsigned j = lo; j < hi; ++j) rank += s_key[j] < w ? 1u : 0u;
                        }
                        s_key2[rank] = w;  // (the histogram's memory: its cursors were last touched before the barrier above)
                        __syncthreads();
                        v[0] = s_key2[tid];  // the k smallest are the first k
                        TK_TRACE(5);  // ranks inside the bins
                        done = true;
                    } else if (total <= (unsigned)TOPK_MAX) {  // a crowded bin: compaction + merge by rank
                        {  // (ONE LDS atomic per wave, not one per key: the order of the words is free)
                            const int lane = tid & 63;
                            const unsigned long long below_me = (1ull << lane) - 1ull;
                            bool take[4];
                            unsigned at[4], cnt = 0u;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                take[r] = r * TOPK_MAX + tid < n_direct && vb[r] <= (int)bstar;
                                const unsigned long long m = __ballot(take[r]);
                                at[r] = cnt + (unsigned)__popcll(m & below_me);
                                cnt += (unsigned)__popcll(m);
                            }
                            unsigned base = 0u;
                            if (lane == 0 && cnt) base = atomicAdd(&s_cnt[0], cnt);
                            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (take[r]) s_key[base + at[r]] = v[r];
                        }
                        __syncthreads();
                        const unsigned long long wk = tid < (int)total ? s_key[tid] : ~0ull;
                        __syncthreads();
                        TK_TRACE(4);  // compaction
                        v[0] = block_rank_sort_1024(wk, (int)total, s_key, s_key2, tid);  // the k smallest are its first k
                        TK_TRACE(5);  // sort
                        done = true;
                    }
                }
            }
            if (!done) {
            // (b) bitwise radix select
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
            const unsigned long long wk = tid < k ? s_key[tid] : ~0ull;
            __syncthreads();
            v[0] = block_rank_sort_1024(wk, k, s_key, s_key2, tid);
            }
        }
        // Every block of the grid has sorted the same words; block b re-rolls candidates 64 b .. 64 b + 63 with ONE wave.
        // (The re-roll is a serial chain of T steps per lane, ~12 us for a lone wave; k = 300 candidates in the first five
        // waves of one block put two of them on one SIMD: 25 us.  One wave per block = one CU each.)
        s_key[tid] = v[0];
        __syncthreads();
        q = blockIdx.x * WAVE + tid;
        // The fifteen waves that do not re-roll generate the block's noise first (regen mode, rows of <= TOPK_PREGEN_MAX_R
        // groups): group g of candidate 64 b + lane by wave g mod 16, into LDS as [g][lane] — the serial chain of the
        // re-roll is then the model's steps alone (the Philox + Box-Muller of a group is about as long as its two steps).
        if (pregen) {
            const int lane = tid & 63, cq = blockIdx.x * WAVE + lane;
            if (cq < k) {
                const uint64_t cgi = s_key[cq] & 0xFFFFFFFFull;
                for (int g = tid >> 6; g < d.R; g += TOPK_MAX / WAVE) s_noise[g * 64 + lane] = gen_noise4(cgi, g, gen, d);
            }
            __syncthreads();
        }
        TK_TRACE(6);  // the block's noise
        if (tid >= WAVE || q >= k) return;
        mine = s_key[q];
    } else {
        if (q >= k) return;
        mine = cand[q];
    }
    const uint64_t gi = mine & 0xFFFFFFFFull;            // global sample index
    const float c = key_to_float((unsigned)(mine >> 32));  // its cost
    weights[q] = expf((-c) / lambda - (-st_min) / lambda) / st_sum;  // softmax(-c/lambda)_i (mppi.py:376)
    const bool inherit = (int64_t)gi < d.inherit_count;
    const int64_t i = (int64_t)gi - d.sample_offset;  // local index: only meaningful when the tiles are read
    const float4* np = noise + ((i >> 6) * d.R) * 64 + (i & 63);
    float* out = states + (int64_t)q * (d.T + 1) * DS;
    const float4* mp = reinterpret_cast<const float4*>(s_mrow) + (inherit ? 0 : d.R);
    const auto roll = [&](auto loadg) {
        const bool bad = rollout_states_noise<MODEL, FAST>(s_x0, d, ctx, out, mp, loadg);
        if constexpr (FAST != 0 && !EntryGeneral<ModelT<MODEL, FAST>>::value) {  // (a lane that left a fast path: the library-math walk)
            if (bad) (void)rollout_states_noise<MODEL, 0>(s_x0, d, ctx, out, mp, loadg);
        }
    };
    if (pregen) roll([&](int g) { return s_noise[g * 64 + (int)(threadIdx.x & 63)]; });
    else if (gen_noise) roll([&](int g) { return gen_noise4(gi, g, gen, d); });
    else roll([&](int g) { return np[(int64_t)g * 64]; });
#ifdef MPPI_TOPK_TRACE
    if (!SORTED && blockIdx.x == 0 && threadIdx.x == 0) {
        const long long t7 = (long long)wall_clock64();
        // (phases 4 / 5 are "scatter by bin" / "ranks inside the bins" of the counting sort, or "compaction" / "sort" of the
        // crowded-bin fallback: the two paths stamp the same slots)
        printf("topk trace (10 ns): costs %lld, histogram %lld, boundary bin %lld, scatter | compaction %lld, ranks | sort %lld, noise %lld, re-roll %lld, total %lld\n",
               s_tk[1] - s_tk[0], s_tk[2] - s_tk[1], s_tk[3] - s_tk[2], s_tk[4] - s_tk[3], s_tk[5] - s_tk[4], s_tk[6] - s_tk[5],
               t7 - s_tk[6], t7 - s_tk[0]);
    }
#endif
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

}  // namespace mppi
