// bfly_check — the DPP / permlane-swap butterflies of mppi_common.hpp (wave_sum_bfly, wave_max_bfly) against the
// __shfl_xor butterflies they replace, bit for bit in every lane, on random floats / doubles of mixed magnitude and sign.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I mppi_playground_amd/csrc scripts/ubench/bfly_check.hip -o scripts/ubench/bfly_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "mppi_common.hpp"

__global__ void check(const float* f, const double* d, int n, unsigned* bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = f[i];
    double w = d[i];
    float s0 = mppi::wave_sum(v), s1 = mppi::wave_sum_bfly(v);
    float m0 = v;
    for (int m = 32; m >= 1; m >>= 1) m0 = fmaxf(m0, __shfl_xor(m0, m));
    float m1 = mppi::wave_max_bfly(v);
    double t0 = w;
    for (int m = 32; m >= 1; m >>= 1) t0 += __shfl_xor(t0, m);
    double t1 = mppi::wave_sum_bfly(w);
    if (__float_as_uint(s0) != __float_as_uint(s1)) atomicAdd(bad, 1u);
    if (__float_as_uint(m0) != __float_as_uint(m1)) atomicAdd(bad + 1, 1u);
    if (__double_as_longlong(t0) != __double_as_longlong(t1)) atomicAdd(bad + 2, 1u);
}

int main() {
    const int n = 1 << 20;
    std::mt19937_64 rng(7);
    std::vector<float> f(n);
    std::vector<double> d(n);
    std::uniform_real_distribution<double> u(-1.0, 1.0), e(-30.0, 30.0);
    for (int i = 0; i < n; ++i) { f[i] = (float)(u(rng) * std::exp2(e(rng))); d[i] = u(rng) * std::exp2(e(rng)); }
    float* fd; double* dd; unsigned* bad;
    (void)hipMalloc(&fd, n * 4); (void)hipMalloc(&dd, n * 8); (void)hipMalloc(&bad, 12);
    (void)hipMemcpy(fd, f.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dd, d.data(), n * 8, hipMemcpyHostToDevice);
    (void)hipMemset(bad, 0, 12);
    check<<<n / 256, 256>>>(fd, dd, n, bad);
    unsigned b[3];
    (void)hipMemcpy(b, bad, 12, hipMemcpyDeviceToHost);
    std::printf("bfly_check: %d lanes (%d waves): float sum mismatches %u, float max mismatches %u, double sum mismatches %u\n", n, n / 64, b[0], b[1], b[2]);
    return (b[0] | b[1] | b[2]) ? 1 : 0;
}
