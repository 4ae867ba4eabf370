// mppi_maps.hpp — Map construction on the device: obstacle rasteriser, lane corridor, padded copies (bit-exact with the reference's host loops).
// Part of the MPPI.forward() hot path for gfx950; see mppi_kernels.hpp for the map of the files.
#pragma once
#include "mppi_common.hpp"

namespace mppi {

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
