// host_search.hpp — the scalar halves of the automatic temperature rules (ESSPS / LBPS / MPO,
// src/pi_mpc/mppi.py:341-370,387-398,526-566 of the reference), as host C++ over softmax statistics that the
// device reduces (mppi_softmax_stats / mppi_softmax_stats_multi).  Pure C++17, no HIP: mppi_capi.hip calls
// these with device-backed statistics callbacks, tests/host_emul compiles the same header with g++ and checks
// it against scipy / the numpy statements in pi_mpc/_host.py on a machine without a GPU.
//
// The reference runs scipy.optimize.brentq (ESSPS) and scipy.optimize.minimize_scalar(method="bounded")
// (LBPS) and one torch.optim.Adam step (MPO); scipy 1.15.3 and torch are third-party dependencies of the
// reference (uv.lock), so their published algorithms are restated here:
//   * bounded scalar minimisation = Brent's fmin (Brent 1973, ch. 5; Forsythe, Malcolm & Moler 1977 "FMIN"),
//     golden-section steps with parabolic interpolation, termination |x - xm| <= 2*tol1 - (b-a)/2 with
//     tol1 = sqrt(eps)*|x| + xatol/3 — scipy's `_minimize_scalar_bounded` uses xatol = 1e-5, maxiter = 500;
//   * Adam (Kingma & Ba 2015) with bias correction in torch's operation order, lr 0.2, betas (0.9, 0.999),
//     eps 1e-8, scalar parameter in fp32.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define MPPI_SEARCH_HD __host__ __device__ inline  // the ESSPS steps also run inside essps_select_kernel
#else
#define MPPI_SEARCH_HD inline
#endif

namespace mppi {
namespace host {

struct SoftmaxStats {  // of softmax(-c/lambda) over all samples, e_i = exp(-(c_i - cmin)/lambda)
    double cmin, cmax, se, se2, sec;
    MPPI_SEARCH_HD double ess() const { return se * se / se2; }
};

// ---------------------------------------------------------------------------------------------------------
// Brent's bounded minimiser.  f: double -> double (may fail: returns false), minimum of f on [x1, x2].
// Returns false if an evaluation failed.
// (Host AND device: lbps_brent_kernel in mppi_search.hpp runs this very function in one wave per block, with the probe
// f(x) = a pass of the whole grid over the costs.  Only +, -, *, /, sqrt, fabs and compares on doubles, every one
// correctly rounded on both sides and nothing contracted (-ffp-contract=off): the same x sequence to the bit.)
template <class F>
MPPI_SEARCH_HD bool fminbound(F&& f, double x1, double x2, double xatol, int maxiter, double& xmin, int* nfev = nullptr) {
    const double sqrt_eps = sqrt(2.2e-16);
    const double golden_mean = 0.5 * (3.0 - sqrt(5.0));
    double a = x1, b = x2;
    double fulc = a + golden_mean * (b - a);
    double nfc = fulc, xf = fulc;
    double rat = 0.0, e = 0.0;
    double x = xf, fx = 0.0;
    if (!f(x, fx)) return false;
    int num = 1;
    double fu = INFINITY;
    double ffulc = fx, fnfc = fx;
    double xm = 0.5 * (a + b);
    double tol1 = sqrt_eps * fabs(xf) + xatol / 3.0;
    double tol2 = 2.0 * tol1;
    while (fabs(xf - xm) > (tol2 - 0.5 * (b - a))) {
        bool golden = true;
        if (fabs(e) > tol1) {  // try a parabolic step through (fulc, nfc, xf)
            golden = false;
            double r = (xf - nfc) * (fx - ffulc);
            double q = (xf - fulc) * (fx - fnfc);
            double p = (xf - fulc) * q - (xf - nfc) * r;
            q = 2.0 * (q - r);
            if (q > 0.0) p = -p;
            q = fabs(q);
            r = e;
            e = rat;
            if (fabs(p) < fabs(0.5 * q * r) && p > q * (a - xf) && p < q * (b - xf)) {
                rat = p / q;
                x = xf + rat;
                if ((x - a) < tol2 || (b - x) < tol2) {  // too close to an end point: step tol1 towards the middle
                    const double d = xm - xf;
                    const double si = (d > 0.0 ? 1.0 : (d < 0.0 ? -1.0 : 0.0)) + (d == 0.0 ? 1.0 : 0.0);
                    rat = tol1 * si;
                }
            } else {
                golden = true;
            }
        }
        if (golden) {
            e = (xf >= xm) ? a - xf : b - xf;
            rat = golden_mean * e;
        }
        const double si = (rat > 0.0 ? 1.0 : (rat < 0.0 ? -1.0 : 0.0)) + (rat == 0.0 ? 1.0 : 0.0);
        x = xf + si * (fabs(rat) > tol1 ? fabs(rat) : tol1);
        if (!f(x, fu)) return false;
        ++num;
        if (fu <= fx) {
            if (x >= xf) a = xf; else b = xf;
            fulc = nfc; ffulc = fnfc;
            nfc = xf; fnfc = fx;
            xf = x; fx = fu;
        } else {
            if (x < xf) a = x; else b = x;
            if (fu <= fnfc || nfc == xf) {
                fulc = nfc; ffulc = fnfc;
                nfc = x; fnfc = fu;
            } else if (fu <= ffulc || fulc == xf || fulc == nfc) {
                fulc = x; ffulc = fu;
            }
        }
        xm = 0.5 * (a + b);
        tol1 = sqrt_eps * fabs(xf) + xatol / 3.0;
        tol2 = 2.0 * tol1;
        if (num >= maxiter) break;
    }
    xmin = xf;
    if (nfev) *nfev = num;
    return true;
}

// LBPS objective (mppi.py:534-557): -(E_w[-c] - range * sqrt((1-delta)/delta) / sqrt(ESS)).
MPPI_SEARCH_HD double lbps_objective(const SoftmaxStats& st, double delta) {
    const double expected_return = -st.sec / st.se;
    const double penalty = (st.cmax - st.cmin) * sqrt((1.0 - delta) / delta) / sqrt(st.se * st.se / st.se2);
    return -(expected_return - penalty);
}

// LBPS temperature (mppi.py:341-349): bounded minimisation of the objective over [lam_min, lam_max].
// stats(lambda, SoftmaxStats&) -> bool evaluates the softmax sums (one device round trip per probe).
template <class S>
MPPI_SEARCH_HD bool lbps_lambda(S&& stats, double delta, double lam_min, double lam_max, double& lam, int* nfev = nullptr) {
    return fminbound(
        [&](double l, double& out) {
            SoftmaxStats st{};
            if (!stats(l, st)) return false;
            out = lbps_objective(st, delta);
            return true;
        },
        lam_min, lam_max, 1e-5, 500, lam, nfev);
}

// ---------------------------------------------------------------------------------------------------------
// ESSPS (mppi.py:351-370,559-566): the root of ESS(lambda) = target on [lam_min, lam_max] with the reference's
// end-point rules, from two geometric grids of P temperatures (ONE pass over the costs on the device per grid) and an
// inverse polynomial interpolation in (ESS, log lambda).  Same algorithm as pi_mpc/_host.py::essps_lambda_grid (within
// ~1e-7 relative of scipy's brentq on the same statistics).  The search is written as three steps so that the host
// loop below (statistics read back per grid) and the device-resident chain in mppi_kernels.hpp (essps_select_kernel:
// no read-back at all) run the very same arithmetic.
// point j of the geometric grid over [lo, hi] (end points exact); the device evaluates one point per lane
template <int P>
MPPI_SEARCH_HD double essps_grid_point(double lo, double hi, int j) {
    if (j == 0) return lo;
    if (j == P - 1) return hi;
    const double llo = log(lo), lhi = log(hi);
    return exp(llo + (lhi - llo) * (double)j / (double)(P - 1));
}
template <int P>
MPPI_SEARCH_HD void essps_make_grid(double lo, double hi, double* grid) {
    for (int j = 0; j < P; ++j) grid[j] = essps_grid_point<P>(lo, hi, j);
}
// first bracketing step of a round: the grid interval [grid[i-1], grid[i]] that holds the root
template <int P>
MPPI_SEARCH_HD int essps_bracket(const double* ess, double target_ess) {
    int i = P - 1;
    for (int j = 0; j < P; ++j) if (ess[j] >= target_ess) { i = j; break; }
    return i < 1 ? 1 : i;
}
// The ESSPS steps work on the grid in the LOG domain: every grid is generated from lg[j] = log(grid[j]) anyway, the
// interpolation needs exactly those logs, and its result is log(root), which is what the next first grid is built
// around — so no step takes a logarithm of a grid value again (the scalar step is one lane of one wave on the device:
// a double-precision log there costs as much as the rest of it).
struct EsspsRange {  // [lam_min, lam_max] and its logs
    double lam_min, lam_max, lmin, lmax;
};
MPPI_SEARCH_HD EsspsRange essps_range(double lam_min, double lam_max) { return EsspsRange{lam_min, lam_max, log(lam_min), log(lam_max)}; }
struct EsspsRoot {   // the result of a search: lambda, log(lambda), and whether the next first grid may be clustered around it
    double lam, log_lam;
    bool warm;
};
// point j of the geometric grid over [lo, hi] given llo = log(lo), lhi = log(hi) (end points exact)
template <int P>
MPPI_SEARCH_HD void essps_point(double lo, double hi, double llo, double lhi, int j, double& g, double& lg) {
    if (j == 0) { g = lo; lg = llo; return; }
    if (j == P - 1) { g = hi; lg = lhi; return; }
    lg = llo + (lhi - llo) * (double)j / (double)(P - 1);
    g = exp(lg);
}
// Warm start.  In a control loop the temperature moves little from solve to solve, so the FIRST grid of a search is
// clustered around the previous solve's root: ESSPS_CLUSTER points geometric over [prev / F, prev * F] (F = 1.5: 3.9 %
// apart), and the other P - ESSPS_CLUSTER spread geometrically over the rest of [lam_min, lam_max] on either side in
// proportion to its log-length (the end points themselves always included: the reference's end-point rules need ESS
// there).  If the six grid points around the root are all close neighbours and the interpolation has visibly
// converged there, the search is over after ONE pass over the costs; anywhere else (first solve, a jump of the
// temperature, a target of a few samples out of very many) the bracket — never wider than ~1/10 of the log-range — is
// refined by a second grid exactly as in a cold search.  Same root either way (to ~1e-6 relative).
constexpr double ESSPS_LOG_WARM_FACTOR = 0.4054651081081644;  // log(1.5)
constexpr double ESSPS_LOG_FINE_RATIO = 0.04879016416943205;  // log(1.05): a spacing this narrow can be interpolated right away
constexpr int ESSPS_CLUSTER = 22;
MPPI_SEARCH_HD bool essps_warm(bool have_prev, double log_prev, const EsspsRange& r) {
    return have_prev && log_prev - ESSPS_LOG_WARM_FACTOR > r.lmin + 1.0e-3 && log_prev + ESSPS_LOG_WARM_FACTOR < r.lmax - 1.0e-3;
}
template <int P>
MPPI_SEARCH_HD void essps_first_point(bool have_prev, double log_prev, const EsspsRange& r, int j, double& g, double& lg) {
    static_assert(P >= ESSPS_CLUSTER + 2, "grid too small for the clustered first grid");
    if (!essps_warm(have_prev, log_prev, r)) { essps_point<P>(r.lam_min, r.lam_max, r.lmin, r.lmax, j, g, lg); return; }
    if (j == 0) { g = r.lam_min; lg = r.lmin; return; }
    if (j == P - 1) { g = r.lam_max; lg = r.lmax; return; }
    constexpr int S = P - ESSPS_CLUSTER;  // points outside the cluster, >= 1 on each side
    const double clo = log_prev - ESSPS_LOG_WARM_FACTOR, chi = log_prev + ESSPS_LOG_WARM_FACTOR;
    int nb = (int)((double)S * (clo - r.lmin) / ((clo - r.lmin) + (r.lmax - chi)) + 0.5);
    nb = nb < 1 ? 1 : (nb > S - 1 ? S - 1 : nb);
    const int na = S - nb;
    if (j < nb) lg = r.lmin + (clo - r.lmin) * (double)j / (double)nb;
    else if (j < nb + ESSPS_CLUSTER) lg = clo + (chi - clo) * (double)(j - nb) / (double)(ESSPS_CLUSTER - 1);
    else lg = chi + (r.lmax - chi) * (double)(j - (nb + ESSPS_CLUSTER - 1)) / (double)na;
    g = exp(lg);
}
template <int P>
MPPI_SEARCH_HD void essps_first_grid(bool have_prev, double log_prev, const EsspsRange& r, double* grid, double* lgrid) {
    for (int j = 0; j < P; ++j) essps_first_point<P>(have_prev, log_prev, r, j, grid[j], lgrid[j]);
}
// the Lagrange form of log(lambda) as a function of ESS at ESS = target through the NPT grid points j0 .. j0+NPT-1:
// term a of the sum (one division per point) ...
template <int NPT>
MPPI_SEARCH_HD double essps_poly_term(const double* lgrid, const double* ess, double target_ess, int j0, int a) {
    double num = 1.0, den = 1.0;
    for (int b = 0; b < NPT; ++b)
        if (b != a) { num *= target_ess - ess[j0 + b]; den *= ess[j0 + a] - ess[j0 + b]; }
    return num / den * lgrid[j0 + a];
}
// ... and the sum in ascending order (the device's scalar step computes the terms on NPT lanes and adds them in the same
// order: mppi_kernels.hpp, essps_round0_wave); false when ESS is not strictly increasing over the points
template <int NPT>
MPPI_SEARCH_HD bool essps_poly(const double* lgrid, const double* ess, double target_ess, int j0, double& log_lam) {
    bool increasing = true;
    for (int a = 0; a < NPT - 1; ++a) increasing = increasing && ess[j0 + a + 1] > ess[j0 + a];
    if (!increasing) return false;
    double x = 0.0;
    for (int a = 0; a < NPT; ++a) x += essps_poly_term<NPT>(lgrid, ess, target_ess, j0, a);
    log_lam = x;
    return true;
}
// the root inside the bracket [grid[i-1], grid[i]]: the polynomial through ESSPS_NPT points (three either side of the root
// away from the grid's ends), or linear interpolation when ESS is not strictly increasing there or the polynomial leaves
// the bracket
constexpr int ESSPS_NPT = 6;
MPPI_SEARCH_HD EsspsRoot essps_linear(const double* grid, const double* ess, double target_ess, int i) {
    const double lo = grid[i - 1], hi = grid[i], e0 = ess[i - 1], e1 = ess[i];
    const double lam = (e1 == e0) ? 0.5 * (lo + hi) : lo + (hi - lo) * (target_ess - e0) / (e1 - e0);
    return EsspsRoot{lam, log(lam), true};
}
MPPI_SEARCH_HD EsspsRoot essps_interpolate(const double* grid, const double* lgrid, const double* ess, double target_ess, int i, int j0) {
    double x;
    if (essps_poly<ESSPS_NPT>(lgrid, ess, target_ess, j0, x) && x >= lgrid[i - 1] && x <= lgrid[i]) return EsspsRoot{exp(x), x, true};
    return essps_linear(grid, ess, target_ess, i);
}
// round 0 (mppi.py:361-364): true when the search is over (root set) — an end-point rule decided, or the
// ESSPS_NPT grid points around the root are all close neighbours (a clustered first grid, or a narrow
// [lam_min, lam_max]) AND the interpolation has visibly converged there: the polynomials through the six and through
// the inner four points agree to ESSPS_AGREE (ESS(lambda) varies on the scale lambda / (c - c_min) of the samples
// that carry the weight, which can be short against the cluster's spacing when the target is a few samples out of
// very many) — else i = the bracket [grid[i-1], grid[i]] for the second grid
constexpr double ESSPS_AGREE = 1.0e-5;
template <int P>
MPPI_SEARCH_HD bool essps_round0(const double* lgrid, const double* ess, double target_ess, const EsspsRange& r, int& i, EsspsRoot& root) {
    static_assert(P >= 2 * ESSPS_NPT, "grid too small");
    if (target_ess <= ess[0]) { root = EsspsRoot{r.lam_min, r.lmin, false}; return true; }
    if (target_ess >= ess[P - 1]) { root = EsspsRoot{r.lam_max, r.lmax, false}; return true; }
    i = essps_bracket<P>(ess, target_ess);
    constexpr int H = ESSPS_NPT / 2;
    if (i >= H && i <= P - H) {
        bool fine = true;
        for (int k = i - H; k < i + H - 1; ++k) fine = fine && lgrid[k + 1] - lgrid[k] <= ESSPS_LOG_FINE_RATIO;
        double x6, x4;
        if (fine && essps_poly<ESSPS_NPT>(lgrid, ess, target_ess, i - H, x6) && essps_poly<4>(lgrid, ess, target_ess, i - 2, x4) &&
            x6 >= lgrid[i - 1] && x6 <= lgrid[i] && fabs(x6 - x4) <= ESSPS_AGREE) {
            root = EsspsRoot{exp(x6), x6, true};
            return true;
        }
    }
    return false;
}
// round 1: bracket on the refined grid, then the polynomial through the grid points around the root
template <int P>
MPPI_SEARCH_HD EsspsRoot essps_round1(const double* grid, const double* lgrid, const double* ess, double target_ess) {
    static_assert(P >= ESSPS_NPT, "grid too small for the interpolation");
    const int i = essps_bracket<P>(ess, target_ess);
    constexpr int H = ESSPS_NPT / 2;
    const int j0 = (i - H < 0 ? 0 : (i - H > P - ESSPS_NPT ? P - ESSPS_NPT : i - H));
    return essps_interpolate(grid, lgrid, ess, target_ess, i, j0);
}
// host loop: ess_grid(lams[P], ess_out[P]) -> bool evaluates ESS for P temperatures (one device pass + read-back)
// (prev: the root of the previous search on this solver — `warm` false = none — replaced by this search's)
template <int P, class G>
bool essps_lambda(G&& ess_grid, double target_ess, double lam_min, double lam_max, double& lam_out, EsspsRoot& prev) {
    double grid[P], lgrid[P], ess[P];
    const EsspsRange r = essps_range(lam_min, lam_max);
    essps_first_grid<P>(prev.warm, prev.log_lam, r, grid, lgrid);
    if (!ess_grid(grid, ess)) return false;
    int i = 1;
    if (!essps_round0<P>(lgrid, ess, target_ess, r, i, prev)) {
        const double lo = grid[i - 1], hi = grid[i], llo = lgrid[i - 1], lhi = lgrid[i];
        for (int j = 0; j < P; ++j) essps_point<P>(lo, hi, llo, lhi, j, grid[j], lgrid[j]);
        if (!ess_grid(grid, ess)) return false;
        prev = essps_round1<P>(grid, lgrid, ess, target_ess);
    }
    lam_out = prev.lam;
    return true;
}
template <int P, class G>
bool essps_lambda(G&& ess_grid, double target_ess, double lam_min, double lam_max, double& lam_out) {  // cold
    EsspsRoot none{0.0, 0.0, false};
    return essps_lambda<P>(ess_grid, target_ess, lam_min, lam_max, lam_out, none);
}

// ---------------------------------------------------------------------------------------------------------
// LBPS, device-resident variant.  The same minimisation as a grid search (one pass over the costs per grid of P temperatures
// instead of one per probe).  One step: obj[j] = objective at grid[j] (geometric grid).  Not the last round: [lo, hi] =
// the two grid intervals around the first grid minimum (the next round's geometric grid spans them).  Last round:
// lam = the minimiser of the QUARTIC that interpolates the five grid points around the minimum in (log lambda,
// objective) (uniform spacing h; Newton on its derivative from the parabola's vertex), kept inside the two neighbouring
// intervals; within two points of an end of the grid the parabola through three points; the grid point itself when the
// minimum sits at an end of the grid or the points are not convex.  Two rounds (spacing 25 % -> 1.4 % of lambda over
// [0.01, 10]) + the quartic land within 3e-7 of the float64 minimiser on exact statistics — what three rounds + a
// parabola did (2e-7; two rounds + a parabola: 1e-4) — and on the device's fp32-summed statistics every variant is
// noise-limited alike (~1e-3 on nav2d's flat objective), so the third round bought nothing but two launches.
constexpr int LBPS_GRID_ROUNDS = 2;
template <int P>
MPPI_SEARCH_HD void lbps_grid_step(const double* grid, const double* obj, bool last, double& lo, double& hi, double& lam) {
    int i = 0;
    for (int j = 1; j < P; ++j) if (obj[j] < obj[i]) i = j;
    const int a = i > 0 ? i - 1 : 0, b = i < P - 1 ? i + 1 : P - 1;
    lo = grid[a]; hi = grid[b];
    lam = grid[i];
    if (last && a < i && i < b) {
        const double x0 = log(grid[a]), x1 = log(grid[i]), x2 = log(grid[b]);
        const double d01 = (obj[i] - obj[a]) / (x1 - x0), d12 = (obj[b] - obj[i]) / (x2 - x1);
        const double curv = (d12 - d01) / (x2 - x0);
        if (curv > 0.0) {
            const double xv = 0.5 * (x0 + x1) - 0.5 * d01 / curv;
            if (xv > x0 && xv < x2) lam = exp(xv);
            if (i >= 2 && i <= P - 3) {  // five points: p(t) = y2 + a1 t + a2 t^2 + a3 t^3 + a4 t^4, t = (x - x1) / h
                const double* y = obj + (i - 2);
                const double h = (log(grid[i + 2]) - log(grid[i - 2])) * 0.25;
                const double a1 = (y[0] - 8.0 * y[1] + 8.0 * y[3] - y[4]) / 12.0;
                const double a2 = (-y[0] + 16.0 * y[1] - 30.0 * y[2] + 16.0 * y[3] - y[4]) / 24.0;
                const double a3 = (-y[0] + 2.0 * y[1] - 2.0 * y[3] + y[4]) / 12.0;
                const double a4 = (y[0] - 4.0 * y[1] + 6.0 * y[2] - 4.0 * y[3] + y[4]) / 24.0;
                if (a2 > 0.0) {
                    double t = -a1 / (2.0 * a2);
                    bool ok = true;
                    for (int it = 0; it < 4 && ok; ++it) {
                        const double dp = a1 + t * (2.0 * a2 + t * (3.0 * a3 + t * 4.0 * a4));
                        const double ddp = 2.0 * a2 + t * (6.0 * a3 + t * 12.0 * a4);
                        ok = ddp > 0.0;
                        if (ok) t -= dp / ddp;
                    }
                    if (ok && t > -1.0 && t < 1.0) lam = exp(x1 + t * h);
                }
            }
        }
    }
}
// host loop over the grid rounds: obj_grid(lams[P], obj_out[P]) -> bool (one device pass + read-back per round)
template <int P, int ROUNDS, class G>
bool lbps_lambda_grid(G&& obj_grid, double lam_min, double lam_max, double& lam_out) {
    double grid[P], obj[P], lo = lam_min, hi = lam_max;
    for (int r = 0; r < ROUNDS; ++r) {
        essps_make_grid<P>(lo, hi, grid);
        if (!obj_grid(grid, obj)) return false;
        lbps_grid_step<P>(grid, obj, r == ROUNDS - 1, lo, hi, lam_out);
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------
// MPO temperature (mppi.py:191-200,387-398): one Adam(lr) step per solve on
//   loss(logT) = T * (epsilon + logsumexp(-c / T)),  T = softplus(logT),  then lambda = exp(logT)   (B-Q12)
// with the gradient written out: dL/dT = epsilon + LSE + E_w[c] / T, dT/dlogT = sigmoid(logT).  The two large terms
// cancel (|LSE| ~ |c|/T against |dL/dT| ~ 1), so WHERE the reference rounds to fp32 decides the result: its autograd
// keeps LSE as an fp32 scalar and forms the backward pass's weights as exp(x_i - fl32(LSE)), which sum to
// exp(LSE - fl32(LSE)) instead of 1.  Both roundings are reproduced (lse32, scale): the same formula in float64 is
// 2-10 % of the gradient away from the reference on nav2d costs, this statement ~1e-4.  Same arithmetic as
// pi_mpc/_host.py::MpoTemperature.  Scalars are fp32 where torch keeps them in fp32 (parameter, gradient, moments).
struct MpoState {
    float log_temperature = 0.0f, m = 0.0f, v = 0.0f;
    int32_t t = 0;
    double epsilon = 0.1, lr = 0.2;
    MPPI_SEARCH_HD float temperature() const { return (float)log1p(exp((double)log_temperature)); }  // softplus(logT), fp32
};
inline void mpo_reset(MpoState& s, double lam0, double epsilon, double lr) {
    s = MpoState{};
    s.log_temperature = (float)std::log(lam0);
    s.epsilon = epsilon; s.lr = lr;
}
// `st` = the softmax statistics at lambda = s.temperature().  Returns the new lambda = exp(logT).
MPPI_SEARCH_HD double mpo_step(MpoState& s, const SoftmaxStats& st) {
    const double b1 = 0.9, b2 = 0.999, adam_eps = 1e-8;
    const double lt = (double)s.log_temperature;
    const float T = s.temperature();
    const float xmax32 = (-(float)st.cmin) / T;
    const float lse32 = xmax32 + (float)log((double)(float)st.se);
    const double scale = exp((double)xmax32 - (double)lse32);
    const float dL_dT = ((float)s.epsilon + lse32) + (float)(scale * st.sec / (double)T);
    const float g = (float)((double)dL_dT * (1.0 / (1.0 + exp(-lt))));
    s.t += 1;
    s.m = (float)(b1 * (double)s.m + (1.0 - b1) * (double)g);
    s.v = (float)(b2 * (double)s.v + (1.0 - b2) * (double)g * (double)g);
    const double bc1 = 1.0 - pow(b1, (double)s.t), bc2 = 1.0 - pow(b2, (double)s.t);
    const double denom = sqrt((double)s.v) / sqrt(bc2) + adam_eps;
    s.log_temperature = (float)(lt - (s.lr / bc1) * (double)s.m / denom);
    return (double)exp(s.log_temperature);  // np.exp of the fp32 scalar: fp32 result
}

}  // namespace host
}  // namespace mppi
