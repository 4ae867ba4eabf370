// micro-benchmark: issue cost of the instruction classes the regenerating sampler uses, relative to v_fma_f32,
// on gfx950.  Eight independent chains per lane, 16 waves per SIMD: measures issue rate, not latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>

enum { FMA, MAD_U64, MUL_LO, MUL_HI, MUL_U24, LOG, SIN, SQRT, RCP, EXP, CVT, BITOP3, MED3, PHILOX_ROUND, N_MODES };
static const char* kNames[N_MODES] = {"v_fma_f32", "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24",
                                      "v_log_f32", "v_sin_f32", "v_sqrt_f32", "v_rcp_f32", "v_exp_f32",
                                      "v_cvt_f32_u32", "v_bitop3_b32", "v_med3_f32", "philox round (2 mad_u64 + 2 bitop3)"};
static const int kInstPerIter[N_MODES] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 4};

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, uint32_t m, int iters) {
    float x[8];
    uint32_t u[8], w[8];
    for (int i = 0; i < 8; ++i) {
        x[i] = threadIdx.x * 1e-3f + i + 1.0f;
        u[i] = threadIdx.x * 2654435761u + i;
        w[i] = u[i] ^ 0x9E3779B9u;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == FMA) x[i] = __builtin_fmaf(x[i], a, b);
            if (MODE == MAD_U64) {
                const uint64_t p = (uint64_t)m * (uint64_t)u[i] + (uint64_t)w[i];
                u[i] = (uint32_t)p; w[i] = (uint32_t)(p >> 32);
            }
            if (MODE == MUL_LO) u[i] = u[i] * w[i];
            if (MODE == MUL_HI) u[i] = __umulhi(u[i], w[i]) | 0x10001u;
            if (MODE == MUL_U24) u[i] = __umul24(u[i], w[i]) | 1u;
            if (MODE == LOG) x[i] = __builtin_amdgcn_logf(x[i]);
            if (MODE == SIN) x[i] = __builtin_amdgcn_sinf(x[i]);
            if (MODE == SQRT) x[i] = __builtin_amdgcn_sqrtf(x[i]);
            if (MODE == RCP) x[i] = __builtin_amdgcn_rcpf(x[i]);
            if (MODE == EXP) x[i] = __builtin_amdgcn_exp2f(x[i]);
            if (MODE == CVT) { x[i] = (float)u[i]; u[i] = __float_as_uint(x[i]); }
            if (MODE == BITOP3) u[i] = __builtin_amdgcn_bitop3_b32(u[i], w[i], m, 0x96);
            if (MODE == MED3) x[i] = __builtin_amdgcn_fmed3f(x[i], a, b);
            if (MODE == PHILOX_ROUND) {
                const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)u[i];
                const uint64_t p1 = (uint64_t)0xCD9E8D57u * (uint64_t)w[i];
                u[i] = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), (uint32_t)p0, m, 0x96);
                w[i] = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), (uint32_t)p1, m, 0x96);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + (float)u[i] + (float)w[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(float* d, hipEvent_t e0, hipEvent_t e1, double& fma_ms) {
    const int iters = 2048, blocks = 4096;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, 0xD2511F53u, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double insts = (double)blocks * 4 * iters * 8 * kInstPerIter[MODE];
    if (MODE == FMA) fma_ms = ms;
    printf("%-40s %8.3f ms  %7.1f G wave-inst/s  %5.2f x v_fma_f32 per inst\n", kNames[MODE], ms, insts / ms / 1e6,
           (ms / kInstPerIter[MODE]) / fma_ms);
    if constexpr (MODE + 1 < N_MODES) run<MODE + 1>(d, e0, e1, fma_ms);
}

int main() {
    float* d;
    hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double fma_ms = 1;
    run<0>(d, e0, e1, fma_ms);
    return 0;
}
