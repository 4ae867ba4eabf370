// Accuracy of the hardware v_sin_f32 / v_cos_f32 (argument in revolutions) against the FAST polynomial sincos_f of the
// models and against fp64, for wrapped headings x in [-pi, pi): every 2^-22-spaced float plus the floats around 0.
// Prints max / rms absolute error and the max error in ulps of the result for both.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o hw_sincos_acc hw_sincos_acc.hip && ./hw_sincos_acc
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../../mppi_playground_amd/csrc/mppi_models.hpp"

struct Acc { double max_abs[4], sum_sq[4], max_ulp[4]; unsigned long long n; };

__global__ void k(Acc* out, long long n0, long long n) {
    __shared__ double s_max[4][256], s_sq[4][256], s_ulp[4][256];
    double mx[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0}, mu[4] = {0, 0, 0, 0};
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) {
        const float x = (float)((double)(n0 + i) * (1.0 / 4194304.0));  // 2^-22 spacing
        if (!(x >= -mppi::PI_F && x < mppi::PI_F)) continue;
        const double sd = sin((double)x), cd = cos((double)x);
        bool bad = false;
        float sp, cp;
        mppi::fused::sincos_f<true, false>(x, sp, cp, bad);
        const float rev = x * 0.159154943f;
        const float sh = __builtin_amdgcn_sinf(rev), ch = __builtin_amdgcn_cosf(rev);
        const float v[4] = {sp, cp, sh, ch};
        const double ref[4] = {sd, cd, sd, cd};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double e = fabs((double)v[q] - ref[q]);
            mx[q] = fmax(mx[q], e);
            sq[q] += e * e;
            const float rf = (float)ref[q];
            const double ulp = (double)(__uint_as_float(__float_as_uint(fabsf(rf)) + 1) - fabsf(rf));
            if (fabs(ref[q]) > 1e-3) mu[q] = fmax(mu[q], e / ulp);
        }
    }
    for (int q = 0; q < 4; ++q) { s_max[q][threadIdx.x] = mx[q]; s_sq[q][threadIdx.x] = sq[q]; s_ulp[q][threadIdx.x] = mu[q]; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int q = threadIdx.x;
        double a = 0, b = 0, c = 0;
        for (int t = 0; t < 256; ++t) { a = fmax(a, s_max[q][t]); b += s_sq[q][t]; c = fmax(c, s_ulp[q][t]); }
        out[blockIdx.x].max_abs[q] = a; out[blockIdx.x].sum_sq[q] = b; out[blockIdx.x].max_ulp[q] = c;
    }
}

int main() {
    const int blocks = 2048;
    Acc* d;
    (void)hipMalloc(&d, sizeof(Acc) * blocks);
    (void)hipMemset(d, 0, sizeof(Acc) * blocks);
    const long long n0 = -13176796, n = 2 * 13176796 + 1;  // x = i * 2^-22 covers [-pi, pi]
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, n0, n);
    std::vector<Acc> h(blocks);
    (void)hipMemcpy(h.data(), d, sizeof(Acc) * blocks, hipMemcpyDeviceToHost);
    const char* names[4] = {"poly sin", "poly cos", "v_sin_f32(x/2pi)", "v_cos_f32(x/2pi)"};
    for (int q = 0; q < 4; ++q) {
        double a = 0, b = 0, c = 0;
        for (auto& e : h) { a = fmax(a, e.max_abs[q]); b += e.sum_sq[q]; c = fmax(c, e.max_ulp[q]); }
        printf("%-18s max abs err %.3e  rms %.3e  max ulp (|ref| > 1e-3) %.2f\n", names[q], a, sqrt(b / (double)n), c);
    }
    return 0;
}
