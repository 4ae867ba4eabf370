// micro-benchmark: what do a few transcendental instructions cost inside a long FMA stream, at the rollout kernel's
// residency (5 waves/SIMD) and at 16?  Body = 64 v_fma_f32 on 8 independent chains (+ the extra instructions of the mode).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters) {
    extern __shared__ float pad[];
    float x[8], y0 = threadIdx.x * 1e-3f + 1.5f, y1 = y0 + 0.25f, y2 = y0 + 0.5f, y3 = y0 + 0.75f;
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
    if (threadIdx.x == 1023) pad[0] = 1.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], a, b);
            if (MODE == 1 && rep == 3) {  // two independent trans ops
                y0 = __builtin_amdgcn_logf(y0); y1 = __builtin_amdgcn_sqrtf(y1);
            }
            if (MODE == 2 && (rep == 1 || rep == 5)) {  // Box-Muller-like dependent chain: log -> mul -> sqrt -> mul, + sin, cos
                const float r = __builtin_amdgcn_sqrtf(-1.386f * __builtin_amdgcn_logf(y0));
                const float c = __builtin_amdgcn_cosf(y1), s = __builtin_amdgcn_sinf(y1);
                y2 = r * c; y3 = r * s; y0 = y2 * y2 + 0.01f; y1 = y3 + 0.3f;
            }
            if (MODE == 3 && (rep == 1 || rep == 5)) {  // the same number of extra PLAIN instructions (8 fma-class)
                const float r = -1.386f * y0 + y1;
                const float c = y1 * a + b, s = y1 * b + a;
                y2 = r * c; y3 = r * s; y0 = y2 * y2 + 0.01f; y1 = y3 + 0.3f;
            }
        }
    }
    float s = y0 + y1 + y2 + y3;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
static void run(float* d, hipEvent_t e0, hipEvent_t e1, size_t lds, const char* occ, float& base_ms) {
    static const char* names[] = {"64 fma", "64 fma + 2 independent trans", "64 fma + 2 x (log,mul,sqrt,cos,sin,mul,mul,fma,add)",
                                  "64 fma + 2 x 9 plain ops"};
    const int iters = 1024, blocks = 4096;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d, 1.0001f, 0.5f, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    if (MODE == 0) base_ms = ms;
    // per SIMD: 16 waves in total; cycles per loop iteration per wave-slot
    const double cyc_iter = ms * 1e-3 * 2.4e9 / (16.0 * iters);
    printf("[%s] %-52s %7.3f ms  %6.1f SIMD-cycles per wave-iteration  (+%.1f over the fma body)\n", occ, names[MODE], ms, cyc_iter,
           (ms - base_ms) * 1e-3 * 2.4e9 / (16.0 * iters));
    if constexpr (MODE < 3) run<MODE + 1>(d, e0, e1, lds, occ, base_ms);
}
int main() {
    float* d;
    (void)hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float base = 1;
    run<0>(d, e0, e1, 0, "16 waves/SIMD", base);
    run<0>(d, e0, e1, 32 * 1024, " 5 waves/SIMD", base);  // 32 KB LDS per block -> 5 blocks of 4 waves per CU
    run<0>(d, e0, e1, 64 * 1024, " 2 waves/SIMD", base);
    return 0;
}
