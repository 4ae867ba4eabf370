// mppi_layout.hpp — Reference layout [N][T][dc] <-> lane-major noise tiles (inject / export), staged through LDS.
// Part of the MPPI.forward() hot path for gfx950; see mppi_kernels.hpp for the map of the files.
#pragma once
#include "mppi_common.hpp"

namespace mppi {

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

}  // namespace mppi
