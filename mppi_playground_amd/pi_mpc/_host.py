"""Host-side pieces of MPPI.forward() that the north star keeps on the CPU: automatic temperature
tuning (ESSPS / LBPS / MPO) and Savitzky-Golay smoothing.  numpy/scipy only, fp32 where the reference
computes in fp32.  Citations are relative to the reference repo.
"""
from __future__ import annotations

import math

import numpy as np
from scipy.optimize import brentq, minimize_scalar

F32 = np.float32


def softmax_weights(costs: np.ndarray, lam: float) -> np.ndarray:
    """torch.softmax(-costs / lambda_, dim=0) in fp32 (src/pi_mpc/mppi.py:355,376,543,564)."""
    x = (-costs) / F32(lam)
    e = np.exp(x - x.max())
    return e / e.sum(dtype=F32)


def compute_ess(weights: np.ndarray) -> float:
    """ESS = 1 / sum(w^2) (src/pi_mpc/mppi.py:526-532)."""
    return 1.0 / float(np.sum(weights * weights, dtype=F32))


def essps_lambda(costs: np.ndarray, target_ess: float, lam_min: float, lam_max: float) -> float:
    """Effective-sample-size policy search (src/pi_mpc/mppi.py:351-370,559-566)."""
    ess_at_min = compute_ess(softmax_weights(costs, lam_min))
    ess_at_max = compute_ess(softmax_weights(costs, lam_max))
    if target_ess <= ess_at_min:
        return lam_min
    if target_ess >= ess_at_max:
        return lam_max
    return brentq(lambda lam: compute_ess(softmax_weights(costs, lam)) - target_ess, lam_min, lam_max)


def lbps_objective(lam: float, costs: np.ndarray, delta: float) -> float:
    """Negative lower bound of the expected return (src/pi_mpc/mppi.py:534-557)."""
    w = softmax_weights(costs, lam)
    ess = compute_ess(w)
    expected_return = -float(np.sum(w * costs, dtype=F32))
    cost_range = float(costs.max() - costs.min())
    penalty = cost_range * math.sqrt((1 - delta) / delta) / math.sqrt(ess)
    return -(expected_return - penalty)


def lbps_lambda(costs: np.ndarray, delta: float, lam_min: float, lam_max: float) -> float:
    """Lower-bound policy search (src/pi_mpc/mppi.py:341-349)."""
    res = minimize_scalar(lambda lam: lbps_objective(lam, costs, delta), bounds=(lam_min, lam_max),
                          method="bounded")
    return float(res.x)


# ---- the same searches driven by device-side softmax statistics ---------------------------------------
# `stats(lam)` returns {cmin, cmax, se, se2, sec} with e_i = exp((-c_i)/lam - (-cmin)/lam) summed over all
# samples (mppi_softmax_stats, combined across shards): the O(N) work stays on the GPU, the scalar
# root-finding stays here.
def ess_from_stats(st) -> float:
    return st["se"] * st["se"] / st["se2"]


def essps_lambda_stats(stats, target_ess: float, lam_min: float, lam_max: float) -> float:
    """src/pi_mpc/mppi.py:351-370 with ESS(lambda) evaluated on the device."""
    if target_ess <= ess_from_stats(stats(lam_min)):
        return lam_min
    if target_ess >= ess_from_stats(stats(lam_max)):
        return lam_max
    return brentq(lambda lam: ess_from_stats(stats(lam)) - target_ess, lam_min, lam_max)


ESSPS_LOG_WARM_FACTOR = math.log(1.5)  # csrc/host_search.hpp: the same constants, the same steps
ESSPS_LOG_FINE_RATIO = math.log(1.05)
ESSPS_CLUSTER = 22
ESSPS_NPT = 6
ESSPS_AGREE = 1.0e-5


def essps_first_grid(lam_prev, lam_min: float, lam_max: float, points: int = 32):
    """First grid of a search and its logs (csrc/host_search.hpp::essps_first_point): geometric over [lam_min, lam_max], or —
    given the previous root well inside the range — ESSPS_CLUSTER points over [prev/1.5, prev*1.5] and the rest spread over
    what is left on either side in proportion to its log-length, the end points included."""
    lo, hi = float(lam_min), float(lam_max)
    lmin, lmax = math.log(lo), math.log(hi)
    lf = ESSPS_LOG_WARM_FACTOR
    lp = math.log(lam_prev) if lam_prev is not None and lam_prev > 0.0 else None
    warm = lp is not None and points >= ESSPS_CLUSTER + 2 and lp - lf > lmin + 1e-3 and lp + lf < lmax - 1e-3
    j = np.arange(points, dtype=np.float64)
    if not warm:
        lg = lmin + (lmax - lmin) * j / (points - 1.0)
    else:
        s = points - ESSPS_CLUSTER
        clo, chi = lp - lf, lp + lf
        nb = min(max(int(s * (clo - lmin) / ((clo - lmin) + (lmax - chi)) + 0.5), 1), s - 1)
        na = s - nb
        below = lmin + (clo - lmin) * j / nb
        cluster = clo + (chi - clo) * (j - nb) / (ESSPS_CLUSTER - 1)
        above = chi + (lmax - chi) * (j - (nb + ESSPS_CLUSTER - 1)) / na
        lg = np.where(j < nb, below, np.where(j < nb + ESSPS_CLUSTER, cluster, above))
    lg[0], lg[-1] = lmin, lmax
    grid = np.exp(lg)
    grid[0], grid[-1] = lo, hi
    return grid, lg


def _essps_poly(lgrid, ess, target_ess, j0, npt):
    """Lagrange form of log(lambda)(ESS) at ESS = target through grid points j0 .. j0+npt-1; None unless ESS increases."""
    xs, ys = lgrid[j0:j0 + npt], ess[j0:j0 + npt]
    if not np.all(np.diff(ys) > 0):
        return None
    x = 0.0
    for a in range(npt):
        num = den = 1.0
        for b in range(npt):
            if b != a:
                num *= target_ess - ys[b]
                den *= ys[a] - ys[b]
        x += num / den * xs[a]
    return float(x)


def _essps_interpolate(grid, lgrid, ess, target_ess, i, j0):
    x = _essps_poly(lgrid, ess, target_ess, j0, ESSPS_NPT)
    if x is not None and lgrid[i - 1] <= x <= lgrid[i]:
        return math.exp(x)
    lo, hi, e0, e1 = float(grid[i - 1]), float(grid[i]), float(ess[i - 1]), float(ess[i])
    if e1 == e0:
        return 0.5 * (lo + hi)
    return lo + (hi - lo) * (target_ess - e0) / (e1 - e0)


def essps_lambda_grid(stats_multi, target_ess: float, lam_min: float, lam_max: float, points: int = 32,
                      lam_prev=None) -> float:
    """The same root as essps_lambda_stats (ESS(lambda) = target, ESS increasing in lambda), bracketed on
    geometric grids: `stats_multi(lams)` evaluates ESS for up to 32 temperatures in ONE pass over the costs, so
    two round trips shrink the bracket from [lam_min, lam_max] to a ratio of (lam_max/lam_min)**(1/31**2)
    (1.007 for [0.01, 10]) and an inverse polynomial interpolation in (ESS, log lambda) through the six grid points
    around it lands within ~1e-7 relative of brentq's answer over the whole range — instead of ~18 sequential
    probes, each a device round trip.  With `lam_prev` (the previous solve's root) the first grid is clustered around
    it (essps_first_grid) and ONE round trip is enough whenever the root has not left the cluster and the
    interpolation has visibly converged there (~1e-6 relative).  Same steps as csrc/host_search.hpp::essps_lambda."""
    assert points >= 2 * ESSPS_NPT
    grid, lgrid = essps_first_grid(lam_prev, lam_min, lam_max, points)
    ess = np.asarray(stats_multi(grid), np.float64)
    if target_ess <= ess[0]:  # same end-point rules as the reference (mppi.py:361-364)
        return lam_min
    if target_ess >= ess[-1]:
        return lam_max

    def bracket(ess):
        above = np.nonzero(ess >= target_ess)[0]
        return max(int(above[0]) if len(above) else points - 1, 1)

    i = bracket(ess)
    h = ESSPS_NPT // 2
    if h <= i <= points - h and np.all(np.diff(lgrid[i - h:i + h]) <= ESSPS_LOG_FINE_RATIO):
        x6, x4 = _essps_poly(lgrid, ess, target_ess, i - h, ESSPS_NPT), _essps_poly(lgrid, ess, target_ess, i - 2, 4)
        if x6 is not None and x4 is not None and lgrid[i - 1] <= x6 <= lgrid[i] and abs(x6 - x4) <= ESSPS_AGREE:
            return math.exp(x6)  # the first grid was fine around the root and the interpolation has converged: one pass
    lo, hi, llo, lhi = float(grid[i - 1]), float(grid[i]), float(lgrid[i - 1]), float(lgrid[i])
    lgrid = llo + (lhi - llo) * np.arange(points) / (points - 1.0)
    lgrid[0], lgrid[-1] = llo, lhi
    grid = np.exp(lgrid)
    grid[0], grid[-1] = lo, hi
    ess = np.asarray(stats_multi(grid), np.float64)
    i = bracket(ess)
    return _essps_interpolate(grid, lgrid, ess, target_ess, i, min(max(i - ESSPS_NPT // 2, 0), points - ESSPS_NPT))


def lbps_lambda_stats(stats, delta: float, lam_min: float, lam_max: float) -> float:
    """src/pi_mpc/mppi.py:341-349,534-557 with the softmax sums evaluated on the device."""

    def objective(lam):
        st = stats(lam)
        expected_return = -st["sec"] / st["se"]
        penalty = (st["cmax"] - st["cmin"]) * math.sqrt((1 - delta) / delta) / math.sqrt(ess_from_stats(st))
        return -(expected_return - penalty)

    return float(minimize_scalar(objective, bounds=(lam_min, lam_max), method="bounded").x)


class MpoTemperature:
    """MPO E-step dual on the temperature (src/pi_mpc/mppi.py:191-200,387-398): one Adam(lr=0.2) step
    per solve on loss = softplus(logT) * (eps + logsumexp(-c / softplus(logT))), then
    lambda = exp(logT).  The gradient is written out instead of using autograd:
      dL/dT = eps + LSE + (sum_i w_i c_i) / T,   dT/dlogT = sigmoid(logT).
    The two large terms cancel (|LSE| ~ |c|/T against |dL/dT| ~ 1), so WHERE the reference rounds to fp32 decides
    the result: its autograd keeps LSE as an fp32 scalar and forms the weights of the backward pass as
    exp(x_i - fl32(LSE)) — they sum to exp(LSE - fl32(LSE)), not to 1.  Both roundings are reproduced here
    (`lse32`, `scale`); evaluating the same formula in float64 differs from the reference by 2-10 % of the gradient
    on nav2d costs (c ~ 770, T ~ 0.8), this statement by ~1e-4.
    """

    def __init__(self, lam0: float = 1.0, epsilon: float = 0.1, lr: float = 0.2):
        self.log_temperature = F32(math.log(lam0))
        self.epsilon = epsilon
        self.lr, self.b1, self.b2, self.eps = lr, 0.9, 0.999, 1e-8
        self.m = F32(0.0)
        self.v = F32(0.0)
        self.t = 0

    def temperature(self) -> float:
        """softplus(logT) as the fp32 scalar of the dual (not the lambda used for the weights)."""
        return float(F32(math.log1p(math.exp(float(self.log_temperature)))))

    def step(self, costs: np.ndarray) -> float:
        T = F32(self.temperature())
        c = costs.astype(F32)
        x = ((-c) / T).astype(np.float64)  # fp32 quotients, like the device statistics
        e = np.exp(x - x.max())
        return self.step_from_sums(float(c.min()), float(e.sum()), float((e * c).sum()))

    def step_from_stats(self, st) -> float:
        """`st` = softmax statistics at lambda = self.temperature()."""
        return self.step_from_sums(st["cmin"], st["se"], st["sec"])

    def step_from_sums(self, cmin: float, se: float, sec: float) -> float:
        """cmin, se = sum e_i, sec = sum e_i c_i with e_i = exp((-c_i)/T - (-cmin)/T), T = self.temperature()."""
        lt = float(self.log_temperature)
        T = F32(self.temperature())
        xmax32 = F32(-F32(cmin)) / T
        lse32 = F32(xmax32 + F32(math.log(F32(se))))           # torch.logsumexp keeps an fp32 scalar
        scale = math.exp(float(xmax32) - float(lse32))          # what the backward pass's fp32 weights sum to / se
        dL_dT = F32(F32(F32(self.epsilon) + lse32) + F32(scale * sec / float(T)))
        g = F32(float(dL_dT) * (1.0 / (1.0 + math.exp(-lt))))
        self.t += 1
        self.m = F32(self.b1 * float(self.m) + (1 - self.b1) * float(g))
        self.v = F32(self.b2 * float(self.v) + (1 - self.b2) * float(g) * float(g))
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        denom = math.sqrt(float(self.v)) / math.sqrt(bc2) + self.eps
        self.log_temperature = F32(lt - (self.lr / bc1) * float(self.m) / denom)
        return float(np.exp(self.log_temperature))


def savitzky_golay_coeffs(window_size: int, poly_order: int) -> np.ndarray:
    """First row of pinv(vander(-h..h)) (src/pi_mpc/mppi.py:568-596)."""
    if window_size % 2 == 0 or window_size <= poly_order:
        raise ValueError("window_size must be odd and greater than poly_order.")
    half = (window_size - 1) // 2
    idx = np.arange(-half, half + 1, dtype=np.float64)
    A = np.vander(idx, N=poly_order + 1, increasing=True)
    return np.linalg.pinv(A)[0].astype(F32)


def apply_savitzky_golay(y: np.ndarray, coeffs: np.ndarray) -> np.ndarray:
    """Symmetric-flip padding + valid cross-correlation (src/pi_mpc/mppi.py:598-620)."""
    pad = len(coeffs) // 2
    yp = np.concatenate([y[:pad][::-1], y, y[-pad:][::-1]]).astype(F32)
    n = len(y)
    out = np.zeros(n, F32)
    for j, cj in enumerate(coeffs):
        out += yp[j:j + n] * cj
    return out


def sg_filter_sequence(history: np.ndarray, action_seq: np.ndarray, coeffs: np.ndarray) -> np.ndarray:
    """Step 7 of forward() (src/pi_mpc/mppi.py:423-443): filter [history(T-1); a(T)], keep the last T."""
    T = action_seq.shape[0]
    prolonged = np.concatenate([history, action_seq], axis=0).astype(F32)
    out = np.zeros_like(prolonged)
    for i in range(prolonged.shape[1]):
        out[:, i] = apply_savitzky_golay(prolonged[:, i], coeffs)
    return out[-T:]
