// mppi_search.hpp — Step 4 on the device (mppi.py:341-370,387-398): softmax statistics for 1 / 32 temperatures, the ESSPS / LBPS searches and the MPO step as kernels.
// Part of the MPPI.forward() hot path for gfx950; see mppi_kernels.hpp for the map of the files.
#pragma once
#include "mppi_common.hpp"

namespace mppi {

// Softmax statistics of the cost vector for one temperature — the device half of the auto-lambda
// searches (ESSPS / LBPS / MPO, mppi.py:341-370,387-398,526-566): the root-finders stay on the host
// and ask for {sum e, sum e^2, sum e*c, max c} with e_i = exp((-c_i)/lambda - (-cmin)/lambda), instead
// of pulling costs[N] over PCIe and running ~10-40 softmaxes on the CPU.  Two tiny launches
// (per-block partials, then a fixed-order combine written to mapped host memory): deterministic.
constexpr int STATS_BLOCKS = 256;
// One thread's share of the single-temperature statistics: the costs start, start + stride, ... in this order.  `load(i)`
// fetches costs[i] (global memory, or a copy a block staged once).  stats_partial_kernel AND lbps_brent_kernel: the same
// arithmetic in the same order, so that the Brent search on the device sees the statistics of mppi_softmax_stats bit for bit.
template <bool WITH_MAX = true, class Load>
__device__ __forceinline__ void stats_partial_thread(Load&& load, int64_t N, int64_t start, int64_t stride, float lambda,
                                                     float xmax, float& se, float& se2, float& sec, float& cmax) {
    se = 0.f; se2 = 0.f; sec = 0.f; cmax = -INFINITY;
    int m = 0;
    for (int64_t i = start; i < N; i += stride, ++m) {
        const float c = load(i, m);
        const float e = expf((-c) / lambda - xmax);
        se += e;
        se2 = fmaf(e, e, se2);
        sec = fmaf(e, c, sec);
        if (WITH_MAX) cmax = fmaxf(cmax, c);  // (the maximum does not depend on the temperature: a search needs it once)
    }
}
template <bool WITH_MAX = true>
__device__ __forceinline__ void stats_partial_wave(float& se, float& se2, float& sec, float& cmax) {
    // (the butterflies of wave_sum / __shfl_xor to the bit, through DPP instead of the LDS crossbar: mppi_common.hpp)
    se = wave_sum_bfly(se); se2 = wave_sum_bfly(se2); sec = wave_sum_bfly(sec);
    if (WITH_MAX) cmax = wave_max_bfly(cmax);
}
// column j (0..2 sums, 3 the maximum) of a 256-thread block's partial row from its four waves' values s_p[w][j]
__device__ __forceinline__ float stats_partial_fold(const float (*s_p)[4], int j) {
    float v = s_p[0][j];
#pragma unroll
    for (int w = 1; w < BLOCK / WAVE; ++w) v = j == 3 ? fmaxf(v, s_p[w][3]) : v + s_p[w][j];
    return v;
}
__global__ __launch_bounds__(BLOCK) void stats_partial_kernel(const float* __restrict__ costs, int64_t N,
                                                             const unsigned* __restrict__ min_key, float lambda_arg,
                                                             const float* __restrict__ lambda_dev /* nullable */,
                                                             float* __restrict__ part /*[STATS_BLOCKS][4]*/) {
    __shared__ float s_p[BLOCK / WAVE][4];
    const float lambda = lambda_dev ? *lambda_dev : lambda_arg;
    const float cmin = key_to_float(*min_key);
    const float xmax = (-cmin) / lambda;
    float se, se2, sec, cmax;
    stats_partial_thread([&](int64_t i, int) { return costs[i]; }, N, (int64_t)blockIdx.x * BLOCK + threadIdx.x,
                         (int64_t)gridDim.x * BLOCK, lambda, xmax, se, se2, sec, cmax);
    stats_partial_wave(se, se2, sec, cmax);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { s_p[wid][0] = se; s_p[wid][1] = se2; s_p[wid][2] = sec; s_p[wid][3] = cmax; }
    __syncthreads();
    if (threadIdx.x < 4) part[blockIdx.x * 4 + threadIdx.x] = stats_partial_fold(s_p, threadIdx.x);
}
// One wave: the partial rows part[b][0..3], b < nblocks, summed in double in a fixed order (lane l takes rows l, l + 64, ...,
// then a butterfly).  Every lane returns the totals.
template <class Row>
__device__ __forceinline__ void stats_combine_wave(Row&& row, int nblocks, int lane, double& se, double& se2, double& sec,
                                                   float& cmax) {
    se = 0.0; se2 = 0.0; sec = 0.0; cmax = -INFINITY;
    for (int b = lane; b < nblocks; b += WAVE) {
        se += row(b, 0); se2 += row(b, 1); sec += row(b, 2);
        cmax = fmaxf(cmax, row(b, 3));
    }
    se = wave_sum_bfly(se); se2 = wave_sum_bfly(se2); sec = wave_sum_bfly(sec);
    cmax = wave_max_bfly(cmax);
}
__global__ __launch_bounds__(WAVE) void stats_combine_kernel(const float* __restrict__ part, int nblocks,
                                                            const unsigned* __restrict__ min_key,
                                                            double* __restrict__ out /*[5] mapped host*/) {
    double se, se2, sec;
    float cmax;
    stats_combine_wave([&](int b, int j) { return part[b * 4 + j]; }, nblocks, (int)threadIdx.x, se, se2, sec, cmax);
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

// LBPS as the REFERENCE searches it — scipy's bounded Brent (mppi.py:341-349; host_search.hpp: fminbound, ported step for
// step), ~22-31 DEPENDENT probes of the objective — without leaving the device: lbps_brent_kernel.
//
// mppi_lbps_lambda runs that search on the host: per probe two launches (stats_partial_kernel, stats_combine_kernel) and a
// read-back, ~19 us each — 0.58 ms per solve at 65 536 samples, 12x the rest of the solve.  Here the search is ONE
// launch of G = min(64, nvb) blocks, nvb = the blocks of stats_partial_kernel's grid ("virtual blocks": 256 threads, the
// same threads own the same costs and add them in the same order), the costs staged once in LDS.  Block l runs the virtual
// blocks l, l + 64, l + 128, l + 192 — exactly the partial rows LANE l of stats_combine_kernel adds up — so per probe it
// publishes that lane's three double sums as six 8-byte {half, probe tag} cells (one relaxed agent-scope store each: data
// and readiness cannot be seen apart, no fence — the protocol of essps_round_kernel / solve_fused_kernel); wave 0 of EVERY
// block then gathers the G lanes' sums (lane l polls block l's cells: one wave per CU on the memory system, one round of
// latency), finishes the sum with stats_combine_kernel's butterfly and takes the SAME Brent step in double precision:
// identical inputs, identical code, so all blocks agree on the next temperature and nothing is broadcast — one dependent
// hop through memory per probe instead of launch + launch + PCIe.  The temperature is the host search's TO THE BIT
// (same partial sums, same order, same fp64 steps; tests/test_gpu_parity.py::test_device_brent_*).
// (First form, measured: every virtual block published its fp32 row and all 1024 threads of every block gathered 8 KB of
// cells — an all-to-all, 2.2-2.8 us per probe for the hop alone; several polls in flight per thread made it worse.)
// Cells are double-buffered by probe parity: a block can run at most one probe ahead of the slowest one (it needs that
// block's sums to finish a probe).  A poll that does not complete within the budget (a block that never became resident)
// raises *error and leaves NaN — no hang.
constexpr int BRENT_LANES = WAVE;                    // blocks of the launch = lanes of the combine (at most)
constexpr int BRENT_GROUPS = STATS_BLOCKS / BRENT_LANES;  // virtual blocks per block (at most): 4
constexpr int BRENT_THREADS = BLOCK * BRENT_GROUPS;  // 1024
constexpr int BRENT_STAGE_MAX = 32;                  // costs per thread staged in LDS (128 KB); beyond that they are re-read (L2)
constexpr int BRENT_MAXITER = 500;                   // scipy's maxiter: a search takes at most this many probes
constexpr unsigned BRENT_SEQ_STRIDE = 512;           // probe tags of one launch: seq0 + 1 .. seq0 + BRENT_MAXITER
constexpr int BRENT_CELLS = 8;                       // per (parity, block): 3 doubles as halves, the block's maximum cost, one spare
struct BrentCtx {
    unsigned long long* cells;  // [2][BRENT_LANES][BRENT_CELLS]
    int* error;                 // mapped host flag
    unsigned seq0;              // tags of this launch start above it (a multiple of BRENT_SEQ_STRIDE)
    long long timeout_ticks;    // poll budget (100 MHz)
};
// -DMPPI_BRENT_TRACE (experiments only): block 0's lane 0 adds up the 100 MHz clock per phase of a probe into bx.error[1..8]
#ifdef MPPI_BRENT_TRACE
#define BRENT_TRACE(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const long long now_ = wall_clock64(); L.trace[k] += (int)(now_ - L.tlast); L.tlast = now_; } } while (0)
#else
#define BRENT_TRACE(k) do { } while (0)
#endif
struct BrentLds {
    float p[BRENT_THREADS / WAVE][4];  // per-wave partial sums of the probe
    float lam, xmax;                   // the temperature of the probe in flight (fp32: what the host search hands the device), (-cmin) / it
    int go;
#ifdef MPPI_BRENT_TRACE
    long long tlast;
    int trace[8];
#endif
};
// (polls: one round of loads in flight per lane and a short sleep between rounds — more loads in flight per lane were
// measured SLOWER: the polls of 64 CUs queue up behind each other in the memory system)
#ifndef BRENT_POLL_SLEEP
#define BRENT_POLL_SLEEP 1
#endif
__device__ __forceinline__ void brent_publish_wave(int nvb, const BrentCtx& bx, unsigned probe, const BrentLds& L, int lane);
// The block's share of one probe, by every thread: the partial sums of its virtual blocks, per wave, into L.p (ends with
// the barrier that publishes them to wave 0).
__device__ __forceinline__ void brent_partials_block(const float* __restrict__ costs, const float* s_cost, bool staged, int64_t N,
                                                     int nvb, const BrentCtx& bx, unsigned probe, BrentLds& L) {
    const int tid = threadIdx.x, g = tid >> 8, t = tid & (BLOCK - 1);
    const int vb = (int)blockIdx.x + BRENT_LANES * g;
    const float lambda = L.lam;
    const float xmax = L.xmax;  // (-cmin) / lambda, divided once per block instead of once per thread (the same bits)
    float se, se2, sec, cmax;
    const int64_t n_eff = vb < nvb ? N : 0, start = (int64_t)vb * BLOCK + t, stride = (int64_t)nvb * BLOCK;
    const auto from_lds = [&](int64_t, int m) { return s_cost[m * (int)blockDim.x + tid]; };
    const auto from_mem = [&](int64_t i, int) { return costs[i]; };
    // (the maximum cost — the objective's range term — does not depend on the temperature: the first probe computes and
    // publishes it, the later ones skip its share of the loop, its butterfly and its cell: ~25 of ~100 instructions per wave)
    if (probe == 1) {
        if (staged) stats_partial_thread<true>(from_lds, n_eff, start, stride, lambda, xmax, se, se2, sec, cmax);
        else stats_partial_thread<true>(from_mem, n_eff, start, stride, lambda, xmax, se, se2, sec, cmax);
        BRENT_TRACE(1);  // exp + sums of the thread
        stats_partial_wave<true>(se, se2, sec, cmax);
    } else {
        if (staged) stats_partial_thread<false>(from_lds, n_eff, start, stride, lambda, xmax, se, se2, sec, cmax);
        else stats_partial_thread<false>(from_mem, n_eff, start, stride, lambda, xmax, se, se2, sec, cmax);
        BRENT_TRACE(1);
        stats_partial_wave<false>(se, se2, sec, cmax);
    }
    const int lane = tid & 63, wid = tid >> 6;
    if (lane == 0) { L.p[wid][0] = se; L.p[wid][1] = se2; L.p[wid][2] = sec; L.p[wid][3] = cmax; }
    BRENT_TRACE(2);  // wave reduction
    __syncthreads();  // (B)
    // wave 1 publishes (wave 0 gathers meanwhile: a wave that stores waits for the store's acknowledgement — a trip to
    // memory for a write-through store — before it can look at a load issued after it: one counter, in order)
    if (wid == 1) brent_publish_wave(nvb, bx, probe, L, lane);
}
// Wave 1 after (B): this block's lane sums (stats_combine_wave's loop over its rows: blockIdx.x, + 64, ... ascending, like
// lane blockIdx.x of the combine) and its maximum, published as seven tagged cells.
__device__ __forceinline__ void brent_publish_wave(int nvb, const BrentCtx& bx, unsigned probe, const BrentLds& L, int lane) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    float mx = -INFINITY;
    for (int g = 0; g < BRENT_GROUPS; ++g)
        if ((int)blockIdx.x + BRENT_LANES * g < nvb) {
            const float (*wp)[4] = L.p + g * (BLOCK / WAVE);
            a0 += stats_partial_fold(wp, 0); a1 += stats_partial_fold(wp, 1); a2 += stats_partial_fold(wp, 2);
            mx = fmaxf(mx, stats_partial_fold(wp, 3));
        }
    const unsigned tag = bx.seq0 + probe;
    unsigned long long* mine = bx.cells + ((size_t)(probe & 1u) * BRENT_LANES + blockIdx.x) * BRENT_CELLS;
    if (lane < (probe == 1 ? 7 : 6)) {
        const unsigned long long b0 = (unsigned long long)__double_as_longlong(a0), b1 = (unsigned long long)__double_as_longlong(a1),
                                 b2 = (unsigned long long)__double_as_longlong(a2);
        const unsigned half = lane == 0 ? (unsigned)b0 : lane == 1 ? (unsigned)(b0 >> 32) : lane == 2 ? (unsigned)b1
                            : lane == 3 ? (unsigned)(b1 >> 32) : lane == 4 ? (unsigned)b2 : lane == 5 ? (unsigned)(b2 >> 32)
                            : __float_as_uint(mx);
        __hip_atomic_store(mine + lane, ((unsigned long long)tag << 32) | (unsigned long long)half, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
}
// Wave 0 after (B): every lane's sums gathered (lane l polls block l's cells), then the butterfly.  Every lane returns the
// totals; false = a poll timed out.
__device__ __forceinline__ bool brent_gather_wave(int nvb, const BrentCtx& bx, unsigned probe, BrentLds& L, long long t0,
                                                  double& se, double& se2, double& sec, float& cmax) {
    const int lane = threadIdx.x;  // (wave 0)
    const int G = nvb < BRENT_LANES ? nvb : BRENT_LANES;
    const unsigned tag = bx.seq0 + probe;
    bool timed_out = false;
    se = 0.0; se2 = 0.0; sec = 0.0; cmax = -INFINITY;
    if (lane < G) {
        const unsigned long long* theirs = bx.cells + ((size_t)(probe & 1u) * BRENT_LANES + lane) * BRENT_CELLS;
        const int ncell = probe == 1 ? 7 : 6;  // (the maximum travels with the first probe only)
        unsigned long long c[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) c[j] = j < ncell ? __hip_atomic_load(theirs + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        for (unsigned spins = 1;; ++spins) {
            bool all = true;
#pragma unroll
            for (int j = 0; j < 7; ++j) all = all && (j >= ncell || (unsigned)(c[j] >> 32) == tag);
            if (all) break;
            // (the clock is a trip to the memory clock domain: looked at once in 64 rounds, not in every one)
            if ((spins & 63u) == 0u && wall_clock64() - t0 > bx.timeout_ticks) { timed_out = true; break; }
            __builtin_amdgcn_s_sleep(BRENT_POLL_SLEEP);
#pragma unroll
            for (int j = 0; j < 7; ++j)
                if (j < ncell) c[j] = __hip_atomic_load(theirs + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        se = __longlong_as_double((long long)((c[1] << 32) | (c[0] & 0xFFFFFFFFull)));
        se2 = __longlong_as_double((long long)((c[3] << 32) | (c[2] & 0xFFFFFFFFull)));
        sec = __longlong_as_double((long long)((c[5] << 32) | (c[4] & 0xFFFFFFFFull)));
        if (probe == 1) cmax = __uint_as_float((unsigned)c[6]);
    }
    BRENT_TRACE(4);  // gather
    se = wave_sum_bfly(se); se2 = wave_sum_bfly(se2); sec = wave_sum_bfly(sec);
    if (probe == 1) cmax = wave_max_bfly(cmax);
    BRENT_TRACE(5);  // butterfly
    return !__any(timed_out);
}
__global__ __launch_bounds__(BRENT_THREADS) void lbps_brent_kernel(const float* __restrict__ costs, int64_t N,
                                                                   const unsigned* __restrict__ min_key, int nvb, int per_thread,
                                                                   double delta, double lam_min, double lam_max, BrentCtx bx,
                                                                   float* __restrict__ lambda_out,
                                                                   double* __restrict__ lambda_host /*[3]: next, used, probes*/) {
    extern __shared__ float s_cost[];  // [per_thread][blockDim.x] when staged
    __shared__ BrentLds L;
    const int tid = threadIdx.x;
    const bool staged = per_thread <= BRENT_STAGE_MAX;
    const long long t0 = wall_clock64();
    if (staged) {
        const int vb = (int)blockIdx.x + BRENT_LANES * (tid >> 8);
        int m = 0;
        if (vb < nvb)
            for (int64_t i = (int64_t)vb * BLOCK + (tid & (BLOCK - 1)); i < N; i += (int64_t)nvb * BLOCK, ++m)
                s_cost[m * (int)blockDim.x + tid] = costs[i];  // (read back by the same thread only: no barrier needed)
    }
    if (tid == 0) {
        L.go = 0;
#ifdef MPPI_BRENT_TRACE
        L.tlast = wall_clock64();
        for (int k = 0; k < 8; ++k) L.trace[k] = 0;
#endif
    }
    const float cmin = key_to_float(*min_key);
    __syncthreads();
    if (tid < WAVE) {  // wave 0: the search itself (every lane the same scalars); the probe's barriers pair with the loop below
        unsigned probe = 0;
        double lam = 0.0;
        int nfev = 0;
        float cmax_all = -INFINITY;  // (gathered with the first probe)
        const bool ok = mppi::host::lbps_lambda(
            [&](double x, mppi::host::SoftmaxStats& st) {
                if (tid == 0) { L.lam = (float)x; L.xmax = (-cmin) / (float)x; L.go = 1; }
                ++probe;
                BRENT_TRACE(6);  // objective + Brent step
                __syncthreads();  // (A) the other waves pick the temperature up
                BRENT_TRACE(0);  // barrier A
                brent_partials_block(costs, s_cost, staged, N, nvb, bx, probe, L);
                BRENT_TRACE(3);  // barrier B
                double se, se2, sec;
                float cmax;
                if (!brent_gather_wave(nvb, bx, probe, L, t0, se, se2, sec, cmax)) return false;
                if (probe == 1) cmax_all = cmax;
                st = mppi::host::SoftmaxStats{(double)cmin, (double)cmax_all, se, se2, sec};
                return true;
            },
            delta, lam_min, lam_max, lam, &nfev);
        if (tid == 0) L.go = 0;
        __syncthreads();  // (A) releases the other waves for good
        if (blockIdx.x == 0 && tid == 0) {
            if (!ok) { lam = NAN; *bx.error = 1; }
            *lambda_out = (float)lam;
            lambda_host[0] = lam; lambda_host[1] = lam; lambda_host[2] = (double)nfev;
#ifdef MPPI_BRENT_TRACE
            BRENT_TRACE(7);
            for (int k = 0; k < 8; ++k) bx.error[1 + k] = L.trace[k];
#endif
        }
    } else {
        unsigned probe = 0;
        for (;;) {
            __syncthreads();  // (A)
            if (!L.go) break;
            ++probe;
            brent_partials_block(costs, s_cost, staged, N, nvb, bx, probe, L);
        }
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

}  // namespace mppi
