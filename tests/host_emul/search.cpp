// Host build of the product's temperature searches (mppi_playground_amd/csrc/host_search.hpp) for the CPU suite.
// TEST-ONLY: the product calls the same header from mppi_capi.hip with statistics reduced on the device; here the
// statistics callback is a plain double-precision loop over a cost array, so that the search logic (Brent's bounded
// minimiser, the ESSPS grid + cubic, the MPO Adam step) can be compared with scipy / pi_mpc/_host.py without a GPU.
#include <cmath>
#include <vector>

#include "../../mppi_playground_amd/csrc/host_search.hpp"

using namespace mppi::host;

// softmax statistics as mppi_softmax_stats defines them: e_i = exp((-c_i)/lam - (-cmin)/lam) with the fp32 quotients
// of the device kernel, sums in double
static SoftmaxStats stats_of(const float* c, int n, double lam) {
    float cmin = INFINITY, cmax = -INFINITY;
    for (int i = 0; i < n; ++i) { cmin = std::fmin(cmin, c[i]); cmax = std::fmax(cmax, c[i]); }
    const float lamf = (float)lam, xmax = (-cmin) / lamf;
    double se = 0, se2 = 0, sec = 0;
    for (int i = 0; i < n; ++i) {
        const double e = (double)std::exp((-c[i]) / lamf - xmax);
        se += e; se2 += e * e; sec += e * (double)c[i];
    }
    return SoftmaxStats{cmin, cmax, se, se2, sec};
}

extern "C" {

int search_lbps(const float* costs, int n, double delta, double lo, double hi, double* lam_out, int* nfev) {
    return lbps_lambda([&](double lam, SoftmaxStats& st) { st = stats_of(costs, n, lam); return true; }, delta, lo, hi,
                       *lam_out, nfev) ? 0 : -1;
}

// the device-resident variant of the LBPS search (lbps_select_kernel): `rounds` geometric grids of 32 temperatures + the
// final parabola, with the statistics the device pass produces (e = exp((cmin - c) * (1 / lam)) in fp32, sums in double)
int search_lbps_grid(const float* costs, int n, double delta, double lo, double hi, double* lam_out) {
    float cmin = INFINITY, cmax = -INFINITY;
    for (int i = 0; i < n; ++i) { cmin = std::fmin(cmin, costs[i]); cmax = std::fmax(cmax, costs[i]); }
    return lbps_lambda_grid<32, LBPS_GRID_ROUNDS>(
               [&](const double* grid, double* obj) {
                   for (int j = 0; j < 32; ++j) {
                       const float inv_lam = 1.0f / (float)grid[j];
                       double se = 0, se2 = 0, sec = 0;
                       for (int i = 0; i < n; ++i) {
                           const double e = (double)std::exp((cmin - costs[i]) * inv_lam);
                           se += e; se2 += e * e; sec += e * (double)costs[i];
                       }
                       obj[j] = lbps_objective(SoftmaxStats{cmin, cmax, se, se2, sec}, delta);
                   }
                   return true;
               },
               lo, hi, *lam_out) ? 0 : -1;
}

// generic check of the minimiser on f(x) = (x - a)^2 * (1 + b * sin(c * x)) over [lo, hi]
int search_fminbound_poly(double a, double b, double c, double lo, double hi, double* xmin, int* nfev) {
    return fminbound([&](double x, double& out) { out = (x - a) * (x - a) * (1.0 + b * std::sin(c * x)) + 0.1 * x; return true; },
                     lo, hi, 1e-5, 500, *xmin, nfev) ? 0 : -1;
}

// lam_prev > 0: warm start from the previous root; passes_out = number of grids (passes over the costs) the search took
int search_essps(const float* costs, int n, double target, double lo, double hi, double lam_prev, double* lam_out,
                 int* passes_out) {
    int passes = 0;
    EsspsRoot prev{lam_prev, lam_prev > 0.0 ? std::log(lam_prev) : 0.0, lam_prev > 0.0};
    const bool ok = essps_lambda<32>(
        [&](const double* grid, double* ess) {
            ++passes;
            for (int j = 0; j < 32; ++j) ess[j] = stats_of(costs, n, (double)(float)grid[j]).ess();
            return true;
        },
        target, lo, hi, *lam_out, prev);
    if (passes_out) *passes_out = passes;
    return ok ? 0 : -1;
}
int search_essps_first_grid(double lam_prev, double lo, double hi, double* grid32) {
    double lg[32];
    essps_first_grid<32>(lam_prev > 0.0, lam_prev > 0.0 ? std::log(lam_prev) : 0.0, essps_range(lo, hi), grid32, lg);
    return 0;
}

// `steps` MPO updates, each on its own cost vector costs[s][n]; lambdas_out[s] = exp(logT) after step s
int search_mpo(const float* costs, int n, int steps, double lam0, double epsilon, double lr, double* lambdas_out) {
    MpoState s;
    mpo_reset(s, lam0, epsilon, lr);
    for (int k = 0; k < steps; ++k)
        lambdas_out[k] = mpo_step(s, stats_of(costs + (size_t)k * n, n, s.temperature()));
    return 0;
}
}
