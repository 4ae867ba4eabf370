// How much slower is straight-line code the first time a CU executes it — and does another wave's earlier pass help?
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/icache_cold scripts/ubench/icache_cold.hip && scripts/ubench/icache_cold
// One block, two waves.  Wave 0 runs N dependent FMAs (8 bytes of code each, fully unrolled) twice through the SAME
// instructions (a non-unrolled 2-trip loop): pass 0 is cold, pass 1 warm.  After a barrier wave 1 runs them once: warm only
// if the instruction cache is shared.  The kernel is launched several times: is a LATER launch cold again?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int N>
__global__ __launch_bounds__(128) void k(float* out, long long* t, float seed) {
    const int w = threadIdx.x >> 6;
    float x = seed + threadIdx.x * 1e-3f;
    if (w == 1) __syncthreads();  // wave 1 waits for wave 0's two passes
#pragma clang loop unroll(disable)
    for (int rep = 0; rep < (w == 0 ? 2 : 1); ++rep) {
        const long long a = wall_clock64();
#pragma unroll
        for (int i = 0; i < N; ++i) x = fmaf(x, 0.999f, 1e-3f * (float)(i & 7));
        const long long b = wall_clock64();
        if ((threadIdx.x & 63) == 0) t[w * 2 + rep] = b - a;
    }
    if (w == 0) __syncthreads();
    out[threadIdx.x] = x;
}

template <int N>
void run(const char* name) {
    float* out; long long* t;
    hipMalloc(&out, 128 * 4);
    hipHostMalloc((void**)&t, 4 * 8, hipHostMallocMapped);
    for (int launch = 0; launch < 4; ++launch) {
        for (int i = 0; i < 4; ++i) t[i] = 0;
        hipLaunchKernelGGL(k<N>, dim3(1), dim3(128), 0, 0, out, t, 1.0f + launch);
        hipDeviceSynchronize();
        printf("%s launch %d: wave 0 pass 0 %6.2f us, pass 1 %6.2f us; wave 1 (after) %6.2f us   [%d FMAs, ~%d KB of code]\n", name, launch,
               t[0] / 100.0, t[1] / 100.0, t[2] / 100.0, N, N * 8 / 1024);
    }
}

int main() {
    run<256>("2KB ");
    run<1024>("8KB ");
    run<4096>("32KB");
    return 0;
}
