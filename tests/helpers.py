"""Shared test helpers: golden-fixture loading and oracle problem construction."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402

# case name -> (model, ctor kwargs as used by tests/golden/make_golden.py)
CASES = {
    "pendulum_T50_N1000_essps": dict(model="pendulum", T=50, N=1000, lambda_="ESSPS"),
    "pendulum_T15_N256_fixed": dict(model="pendulum", T=15, N=256, lambda_=1.0),
    "pendulum_T15_N256_lbps": dict(model="pendulum", T=15, N=256, lambda_="LBPS"),
    "pendulum_T15_N256_mpo": dict(model="pendulum", T=15, N=256, lambda_="MPO"),
    "pendulum_T15_N200_explore": dict(model="pendulum", T=15, N=200, lambda_=0.5, exploration=0.25),
    "cartpole_T64_N1024_essps_sg": dict(model="cartpole", T=64, N=1024, lambda_="ESSPS", use_sg_filter=True),
    "cartpole_T10_N100_fixed": dict(model="cartpole", T=10, N=100, lambda_=0.001),
    "mountaincar_T100_N256_fixed": dict(model="mountaincar", T=100, N=256, lambda_=0.1),
    "nav2d_T50_N512_essps": dict(model="nav2d", T=50, N=512, lambda_="ESSPS"),
    "nav2d_T30_N256_fixed_explore": dict(model="nav2d", T=30, N=256, lambda_=1.0, exploration=0.25),
    "racing_T50_N512_fixed": dict(model="racing", T=50, N=512, lambda_=1.0),
    "racing_T25_N256_fixed": dict(model="racing", T=25, N=256, lambda_=1.0),
    "mjcartpole_T50_N256_fixed": dict(model="mjcartpole", T=50, N=256, lambda_=1.0),
    "goalzone_T30_N256_fixed": dict(model="goalzone", T=30, N=256, lambda_=1.0),
    # round 2: dense softmax at N = 4096, racing / nav2d with SG + exploration, the other
    # temperature rules on nav2d, ESSPS running into lambda_max, a posterior draw between two solves
    "racing_T25_N4096_dense": dict(model="racing", T=25, N=4096, lambda_=500.0),
    "racing_T25_N512_explore_sg": dict(model="racing", T=25, N=512, lambda_=200.0, exploration=0.25,
                                       use_sg_filter=True),
    "racing_T25_N1024_essps": dict(model="racing", T=25, N=1024, lambda_="ESSPS"),
    "nav2d_T30_N4096_essps": dict(model="nav2d", T=30, N=4096, lambda_="ESSPS"),
    "nav2d_T30_N512_lbps": dict(model="nav2d", T=30, N=512, lambda_="LBPS"),
    "nav2d_T30_N512_mpo": dict(model="nav2d", T=30, N=512, lambda_="MPO"),
    "nav2d_T30_N512_sg": dict(model="nav2d", T=30, N=512, lambda_=5.0, use_sg_filter=True, sg_window_size=7,
                              sg_poly_order=2),
    "nav2d_T20_N256_posterior": dict(model="nav2d", T=20, N=256, lambda_=5.0),
}
SOLVER_KW = ("exploration", "use_sg_filter", "sg_window_size", "sg_poly_order")  # ctor kwargs a case may carry


def sg_coeffs(cfg):
    from pi_mpc import _host

    return _host.savitzky_golay_coeffs(cfg.get("sg_window_size", 5), cfg.get("sg_poly_order", 3))

MODEL_CFG = {
    "pendulum": dict(u_min=[-2.0], u_max=[2.0], sigmas=[1.0]),
    "cartpole": dict(u_min=[-3.0], u_max=[3.0], sigmas=[1.0]),
    "mountaincar": dict(u_min=[-1.0], u_max=[1.0], sigmas=[1.0]),
    "nav2d": dict(u_min=[0.0, -1.0], u_max=[2.0, 1.0], sigmas=[0.5, 0.5]),
    "racing": dict(u_min=[-2.0, -0.25], u_max=[2.0, 0.25], sigmas=[0.5, 0.1]),
    "mjcartpole": dict(u_min=[-3.0], u_max=[3.0], sigmas=[1.0]),
    "goalzone": dict(u_min=[-1.0, -1.0], u_max=[1.0, 1.0], sigmas=[0.5, 0.5]),
}


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def unpack_bits(bits, shape):
    n = int(shape[0]) * int(shape[1])
    return np.unpackbits(bits)[:n].reshape(int(shape[0]), int(shape[1])).astype(np.uint8)


_env_cache = {}


def racing_env_fixture():
    if "racing" not in _env_cache:
        e = load("racing_env")
        shape = e["map_shape"]
        _env_cache["racing"] = dict(
            obst=unpack_bits(e["obst_bits"], shape), lane=unpack_bits(e["lane_bits"], shape),
            cell=float(e["cell_size"]), origin=e["origin"].astype(np.float64),
            center_path=e["center_path"], center_path_f64=e["center_path_f64"], circles=e["circles"],
            start_state=e["start_state"], x_lim=e["x_lim"], y_lim=e["y_lim"],
            lane_width=float(e["lane_width"]))
    return _env_cache["racing"]


def nav2d_env_fixture():
    if "nav2d" not in _env_cache:
        e = load("nav2d_env")
        _env_cache["nav2d"] = dict(
            map=unpack_bits(e["map_bits"], e["map_shape"]), cell=float(e["cell_size"]),
            origin=e["origin"].astype(np.float64), circles=e["circles"], rects=e["rects"],
            goal=e["goal"], start_state=e["start_state"], x_lim=e["x_lim"], y_lim=e["y_lim"])
    return _env_cache["nav2d"]


def goalzone_env_fixture():
    if "goalzone" not in _env_cache:
        e = load("goalzone_env")
        _env_cache["goalzone"] = dict(goal=e["goal"], center=e["center"], radius=float(e["radius"]), x0=e["x0"])
    return _env_cache["goalzone"]


def oracle_problem(model, N, T, exploration=0.0, ref_path=None):
    cfg = MODEL_CFG[model]
    params, maps = (), ()
    if model == "racing":
        e = racing_env_fixture()
        params = orc.racing_params()
        maps = [(e["obst"], e["cell"], e["origin"]), (e["lane"], e["cell"], e["origin"])]
    elif model == "nav2d":
        e = nav2d_env_fixture()
        params = orc.nav2d_params()
        maps = [(e["map"], e["cell"], e["origin"])]
    elif model == "goalzone":
        e = goalzone_env_fixture()
        params = orc.goalzone_params(np.float32(e["goal"]), np.float32(e["center"]), e["radius"])
    return orc.Problem(model, N, T, cfg["u_min"], cfg["u_max"], exploration=exploration, params=params,
                       maps=maps, ref_path=ref_path)


# ---- the reference's own measured sensitivity (tests/golden/make_golden.py: BAND_VARIANTS; 24 probes per solve)
def band_fixed(g, k):
    """(action, state): how far the REFERENCE's action_seq / state_seq of solve k move when its total costs are replaced
    by equally valid fp32 evaluations of the same sums (1-ulp changes, other summation orders), inputs and temperature
    held fixed.  Maximum over the recorded probes."""
    b = g[f"band_fixed_{k}"]
    return float(b[:, 0].max()), float(b[:, 1].max())


def band_rule_lambda(g, k):
    """The same probes with the reference's temperature rule re-run: relative spread of the temperature of solve k."""
    return float(g[f"band_rule_{k}"][:, 2].max()) if f"band_rule_{k}" in g.files else 0.0


def band_closed_loop(g, k):
    """dict(x0, action, state, lam): the reference's whole K-solve closed loop re-run per probe (states, warm start, SG
    history and the rule's memory evolve on their own), relative distance of solve k to the unperturbed loop."""
    b = g["band_closed_loop"][k].max(axis=0)
    return dict(x0=float(b[0]), action=float(b[1]), state=float(b[2]), lam=float(b[3]))


def rel_err(a, b):
    """max |a-b| relative to max |b| (the tolerance convention of SURVEY Appendix D)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def lbps_objective64(costs, lam, delta=0.01):
    """The LBPS objective (src/pi_mpc/mppi.py:534-557) in float64 — the yardstick for a temperature found on the
    reference's fp32, noise-flat objective: two minimisers are "the same" when this differs by a few fp32 ulps."""
    c = np.asarray(costs, np.float64)
    x = -c / lam
    e = np.exp(x - x.max())
    w = e / e.sum()
    ess = 1.0 / np.sum(w * w)
    return float(-(-np.sum(w * c) - (c.max() - c.min()) * np.sqrt((1 - delta) / delta) / np.sqrt(ess)))


def same_lbps_minimum(costs, lam, lam_ref, delta=0.01, tol=2e-2):
    """LBPS temperatures agree: within 1e-3 relative, or — where the fp32 objective is flat to its own rounding noise
    (nav2d: 1e-8 relative over a 1 % change of lambda) — within 2 % AND no worse than the reference's own lambda by
    more than 4 fp32 ulps of the float64 objective."""
    if abs(lam - lam_ref) <= 1e-3 * lam_ref:
        return True
    f, f_ref = lbps_objective64(costs, lam, delta), lbps_objective64(costs, lam_ref, delta)
    return abs(lam - lam_ref) <= tol * lam_ref and f - f_ref <= 4 * float(np.finfo(np.float32).eps) * abs(f_ref)
