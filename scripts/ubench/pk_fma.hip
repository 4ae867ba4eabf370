// micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 (and a mixed stream) on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters) {
    float x[8]; v2f y[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3f + i; y[i] = v2f{x[i], x[i] + 0.5f}; }
    v2f av = {a, a}, bv = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);
            if (MODE == 1) y[i] = __builtin_elementwise_fma(y[i], av, bv);
            if (MODE == 2) { x[i] = __builtin_fmaf(x[i], a, b); x[i] = fminf(x[i], 3.0f); }   // fma + min
            if (MODE == 3) { x[i] = x[i] * a; x[i] = x[i] + b; }                              // mul + add unfused
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096, blocks = 4096;
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, iters);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double insts = (double)blocks * 4 * iters * 8 * (mode >= 2 ? 2 : 1);  // wave-instructions
            if (rep) printf("mode %d: %.3f ms, %.1f G wave-inst/s, %.2f cycles/inst/SIMD @2.4GHz, lane-ops %.1f T/s\n", mode, ms,
                            insts / ms / 1e6, 1024 * 2.4e9 / (insts / (ms * 1e-3)), insts * 64 * (mode == 1 ? 2 : 1) / ms / 1e9);
        }
    }
    return 0;
}
