// What does a read of the 100 MHz clock (s_memrealtime, wall_clock64()) cost — one wave, and 16 waves of a block at once?
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/clock_cost scripts/ubench/clock_cost.hip && scripts/ubench/clock_cost
// Each wave reads the clock R times in a dependent chain (the next read is issued only after the previous value is used);
// reported: (last - first) / (R - 1) per wave, in ns, and clock64() (s_memtime: the shader clock) for comparison.
#include <hip/hip_runtime.h>
#include <cstdio>

template <bool WALL>
__global__ void k(long long* out, int reps) {
    long long first = 0, last = 0, acc = 0;
    for (int i = 0; i < reps; ++i) {
        long long v = WALL ? wall_clock64() : clock64();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (i == 0) first = v;
        last = v;
        acc += v & 1;
        __builtin_amdgcn_sched_barrier(0);
    }
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2] = last - first;
        out[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2 + 1] = acc;
    }
}

int main() {
    long long* out;
    hipHostMalloc((void**)&out, 2 * 16 * 8, hipHostMallocMapped);
    const int reps = 65;
    for (int waves : {1, 4, 16}) {
        for (int wall = 1; wall >= 0; --wall) {
            for (int rep = 0; rep < 2; ++rep) {
                if (wall) hipLaunchKernelGGL(k<true>, dim3(1), dim3(64 * waves), 0, 0, out, reps);
                else hipLaunchKernelGGL(k<false>, dim3(1), dim3(64 * waves), 0, 0, out, reps);
                (void)hipDeviceSynchronize();
            }
            double mx = 0, mn = 1e30;
            for (int w = 0; w < waves; ++w) { double v = (double)out[2 * w] / (reps - 1); if (v > mx) mx = v; if (v < mn) mn = v; }
            if (wall) printf("%2d wave(s), wall_clock64 (100 MHz ticks = 10 ns): %.1f .. %.1f ns per dependent read\n", waves, mn * 10, mx * 10);
            else printf("%2d wave(s), clock64 (shader clock cycles):            %.0f .. %.0f cycles per dependent read\n", waves, mn, mx);
        }
    }
    return 0;
}
