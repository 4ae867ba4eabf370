// mppi_env.hpp — The control tick around the solver without the host: map lookups for callers outside the solver, calc_ref_trajectory, env.step (example/racing.py:161-266).
// Part of the MPPI.forward() hot path for gfx950; see mppi_kernels.hpp for the map of the files.
#pragma once
#include "mppi_common.hpp"

namespace mppi {

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
    // everything that does not depend on the search is requested before it (the carried index, this thread's window
    // offset, the last offset): the kernel is a chain of memory round trips, two instead of four this way
    const int cind0 = *w.cind;
    const int dmine = (int)threadIdx.x < w.rows ? w.dind[threadIdx.x] : 0, dlast = w.dind[w.rows - 1];
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
        const int ind = max(cind0, (int)(unsigned)b);  // "ensure the index is not less than the current index"
        *w.cind = ind;
        s_ind = ind;
    }
    __syncthreads();
    const int ind = s_ind;
    const bool all_inside = ind + dlast < w.n;  // dind is increasing
    for (int i = threadIdx.x; i < w.rows; i += REFWIN_BLOCK) {
        const int idx = ind + (i == (int)threadIdx.x ? dmine : w.dind[i]);
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

}  // namespace mppi
