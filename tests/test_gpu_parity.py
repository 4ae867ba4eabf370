"""Parity of the HIP path (through the C ABI / pi_mpc.mppi.MPPI) against the oracle and the reference
fixtures, on a real MI355X.  All tests here are marked `gpu`.

Tolerances (fp32, north star: 1e-5 relative):
  * costs: |gpu - oracle| <= 1e-5 * max|oracle| for every sample that is not within 1e-3 cell of a
    map-rounding boundary (`margin`, computed by the oracle); at most a handful of boundary samples
    may flip an occupancy cell under <=1.5-ulp sin/cos differences.
  * action_seq / state_seq given the same costs: 1e-5 relative to max-abs.
  * end-to-end against the reference fixtures (`check_end_to_end`): 1e-5, or — where softmax(-c / lambda) amplifies the
    last bits of the costs beyond that (nav2d, goal zone, pendulum at lambda ~ 1: |c| / lambda ~ 1e3) — the REFERENCE'S
    OWN measured spread: every fixture records how far the reference's action_seq / state_seq move when its total costs
    are replaced by equally valid fp32 evaluations of the same sums (256 probes per solve since round 5: 1-ulp changes, other
    summation orders; tests/golden/make_golden.py `band_fixed_k`, `band_rule_k`, `band_closed_loop`).  limit = max(1e-5,
    1.0 x the sample maximum of the probes, see BAND_MARGIN); nothing is derived analytically.  Where the softmax is an arg-min (racing at lambda = 1) the winning sample must be
    the reference's and the action must equal its clamped action sequence to 1e-6; only a top-2 cost gap under 4 ulps
    lifts that.
  * automatic temperatures are compared with the reference's on their own terms; the end-to-end check of those
    cases then re-solves with the reference's lambda so that it is not blurred by the search tolerance.
"""
import ctypes as C

import numpy as np
import pytest
import torch

import parity_report
from helpers import (CASES, MODEL_CFG, SOLVER_KW, Band, band_closed_loop, band_fixed, band_rule, band_rule_lambda, load, oracle_problem, orc,
                     rel_err, same_lbps_minimum, sg_coeffs)

pytestmark = pytest.mark.gpu

TOL = 1e-5
COST_TOL = 2e-6  # costs against the oracle on identical inputs (see check_costs)
EPS32 = float(np.finfo(np.float32).eps)


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X; torch.cuda.is_available() is False")


# ------------------------------------------------------------------------------ solver factory
_envs = {}


def make_solver(model, T, N, lambda_=1.0, **kw):
    """Product MPPI with the shipped native plugins (no call-site changes vs the reference examples)."""
    _need_gpu()
    from pi_mpc.mppi import MPPI

    cfg = MODEL_CFG[model]
    common = dict(horizon=T, num_samples=N, u_min=torch.tensor(cfg["u_min"]), u_max=torch.tensor(cfg["u_max"]),
                  sigmas=torch.tensor(cfg["sigmas"]), lambda_=lambda_, **kw)
    ctrl = None
    if model == "goalzone":
        from envs.goal_in_danger_zone import GoalInDangerZoneEnv
        from helpers import goalzone_env_fixture

        env = GoalInDangerZoneEnv()
        env._goal = goalzone_env_fixture()["goal"]
        solver = MPPI(dim_state=7, dim_control=2, dynamics=env.parallel_step, cost_func=env.parallel_cost, **common)
        return solver, None
    if model in ("pendulum", "cartpole", "mountaincar", "mjcartpole"):
        from envs import classic_control as cc

        dyn, cost = getattr(cc, f"{model}_dynamics"), getattr(cc, f"{model}_cost")
        ds, dc = orc.MODEL_DIMS[orc.MODEL_IDS[model]]
        solver = MPPI(dim_state=ds, dim_control=dc, dynamics=dyn, cost_func=cost, **common)
    elif model == "nav2d":
        from envs.navigation_2d import Navigation2DEnv

        env = _envs.setdefault("nav2d", Navigation2DEnv())
        solver = MPPI(dim_state=3, dim_control=2, dynamics=env.dynamics, cost_func=env.cost_function, **common)
    else:
        from envs.racing_controller import racing_controller
        from envs.racing_env import RacingEnv

        env = _envs.setdefault("racing", RacingEnv())
        kw2 = {k: v for k, v in common.items() if k not in ("horizon", "num_samples", "u_min", "u_max", "sigmas",
                                                             "lambda_")}
        ctrl = racing_controller(env, horizon=T, num_samples=N, lambda_=lambda_, **kw2)
        ctrl.set_cost_map(env._obstacle_map, env._lane_map)
        solver = ctrl.solver
    return solver, ctrl


def used_lambda(g, cfg, k):
    if cfg["lambda_"] == "MPO":
        return 1.0 if k == 0 else float(g[f"lambda_{k - 1}"])
    return float(g[f"lambda_{k}"])


def check_costs(c_gpu, r, max_flips=None):
    if max_flips is None:
        max_flips = max(1, int(2e-5 * len(c_gpu)))  # boundary samples only (measured: 0 in every check of rounds 3-4); see module docstring
    scale = np.abs(r["costs"]).max()
    diff = np.abs(c_gpu - r["costs"])
    clear = r["margin"] > 1e-3
    if clear.any():  # (a start pinned exactly onto a cell boundary by the position clamp leaves no clear sample)
        # band-independent tripwire: the costs themselves are held to COST_TOL = 2e-6 of the scale — a few fp32 ulps
        # of a sum of T stage costs, 5x what has ever been measured (4.3e-7) — not to the north star's 1e-5, so that a
        # regression smaller than the end-to-end bands still fails here
        parity_report.record("cost_rel_err_clear_samples", np.max(diff[clear]) / scale, COST_TOL, n=int(len(c_gpu)))
        assert np.max(diff[clear]) <= COST_TOL * scale, f"clear-sample cost error {np.max(diff[clear]) / scale:.2e}"
    nflip = int((diff > TOL * scale).sum())
    parity_report.record("map_cell_flips", nflip, max_flips, n=int(len(c_gpu)),
                         flips_on_clear_samples=int(((diff > TOL * scale) & clear).sum()))
    assert nflip <= max_flips, f"{nflip} samples differ beyond tolerance (all must be boundary samples)"
    return nflip


def check_rel(quantity, got, want, tol):
    """rel_err(got, want) < tol, with the measured value kept for the parity report."""
    err = rel_err(got, want)
    parity_report.record(quantity, err, tol)
    assert err < tol, f"{quantity}: {err:.2e} >= {tol:.2e}"
    return err


# A fixture band is the MAXIMUM over 256 probes of the reference (round 5; 24 until round 4, when the tests allowed 1.5x the
# sample maximum): a measurement of the reference's own spread, held at 1.0x.  The parity report also counts the checks
# beyond 1e-5 and gives each band's 99th percentile.
BAND_MARGIN = 1.0
# ... except for the OPT-IN grid search of LBPS (lbps_search="grid"): it returns the float64 minimiser of the
# objective, which is not where the reference's Brent stops (MPPI's docstring); end to end it is held to 1.5x the band
BAND_MARGIN_FAST_LBPS = 1.5


def _band_position(band, err):
    """Where `err` falls in the reference's own probes (helpers.Band): rank, probe count, and the probes it equals bit for bit."""
    if not hasattr(band, "rank"):
        return {}
    return dict(rank=band.rank(err), band_probes=band.probes, coincides_with=band.coincides(err))


def check_banded(quantity, got, want, band, floor=TOL, margin=None, strict_ok=False):
    """rel_err(got, want) <= max(floor, BAND_MARGIN * band) where `band` is the reference's own measured spread of that
    quantity (committed with the fixture); the report keeps the value, the band, whether the plain 1e-5 held, the RANK of the
    value among the reference's probes and the probes it coincides with bit for bit (a value sitting exactly on the band is one
    of the probes: the device computes that very evaluation of the costs — tests/test_gpu_zz_report.py requires every check
    on its band to be explained that way, and holds every rank to the committed report's).  `strict_ok`: under
    --strict-parity this check (a full-size configuration) is held to the plain floor."""
    err = rel_err(got, want)
    margin = BAND_MARGIN if margin is None else margin
    limit = max(floor, margin * band)
    if strict_ok and parity_report.strict:
        limit = floor
    parity_report.record(quantity, err, limit, reference_band=float(band), reference_band_p99=getattr(band, "p99", None),
                         within_1e5=bool(err <= TOL), within_band=bool(err <= max(floor, band)),
                         above_band_itself=bool(err > band), **_band_position(band, err))
    assert err <= limit, f"{quantity}: {err:.2e} > max({floor:.0e}, {margin} x reference band {band:.2e})"
    return err


def check_end_to_end(a, s, c_gpu, g, k, cfg, band_a, band_s, tag="", margin=None):
    """Action / state sequence of solve k against the reference fixture: 1e-5, or the reference's own measured spread
    under rounding-level changes of its costs where that is larger (see the module docstring)."""
    from pi_mpc import _host

    a_ref, s_ref, c_ref = g[f"action_seq_{k}"], g[f"state_seq_{k}"], g[f"costs_{k}"]
    w_ref = g[f"weights_{k}"].astype(np.float64)
    scale = max(float(np.abs(c_ref).max()), 1e-30)
    order = np.argsort(c_ref, kind="stable")
    i_ref = int(order[0])
    if w_ref[i_ref] >= 1.0 - 1e-6 and len(c_ref) > 1:  # arg-min regime (SURVEY B-Q9): an exact statement exists
        gap = float(c_ref[order[1]] - c_ref[order[0]])
        if gap >= 4 * EPS32 * scale:
            assert int(np.argmin(c_gpu)) == i_ref, "the winning sample differs from the reference's"
            mc = MODEL_CFG[cfg["model"]]  # U[i] = clamp(mean + eps_i) (mean only below the exploration split, mppi.py:266-275)
            inherit = i_ref < int(cfg["N"] * (1 - cfg.get("exploration", 0.0)))
            U = np.clip((g[f"mean_in_{k}"] if inherit else 0.0) + g[f"eps_{k}"][i_ref], np.float32(mc["u_min"]),
                        np.float32(mc["u_max"])).astype(np.float32)
            if cfg.get("use_sg_filter"):
                U = _host.sg_filter_sequence(g[f"sg_hist_in_{k}"], U, sg_coeffs(cfg))
            parity_report.record("argmin_action_vs_reference_sample", np.abs(a - U).max() / max(np.abs(U).max(), 1e-30), 1e-6 + BAND_MARGIN * band_a)
            assert np.abs(a - U).max() <= (1e-6 + BAND_MARGIN * band_a) * max(np.abs(U).max(), 1e-30), "action != U[argmin]"
    check_banded("action_seq_vs_reference_fixture" + tag, a, a_ref, band_a, margin=margin)
    check_banded("state_seq_vs_reference_fixture" + tag, s, s_ref, band_s, margin=margin)


# the library's own search tolerances (the reference-side spread of the temperature comes from the fixture bands):
# ESSPS: grid + inverse interpolation, checked to 1e-5 against brentq (test_host_logic); LBPS: scipy's bounded Brent stops
# within xatol = 1e-5 ABSOLUTE + sqrt(eps) of a minimum of the reference's fp32 objective, short of a bound it never
# evaluates (pendulum: lambda_max = 10 -> 9.9994..9.9998), while the library's opt-in grid search (lbps_search="grid")
# returns the float64 minimiser to 1e-7: 1e-3 for that one; the default — the port of scipy's Brent, on the device since
# round 6, or as a host loop (lbps_search="brent_host") — is held to 1e-4
LBPS_TOL = 1e-3
LBPS_TOL_BRENT = 1e-4
LAMBDA_TOL = {"ESSPS": 1e-4}


def grid_lbps(solver):
    return getattr(solver, "_rule_on_device", None) == "LBPS" and solver._lbps_search == "grid"


def lbps_floor(solver):
    return LBPS_TOL if grid_lbps(solver) else LBPS_TOL_BRENT


# ------------------------------------------------------------------------------ whole solve vs oracle / golden
@pytest.mark.parametrize("math", [2, 1, 0])  # 2 = default (hardware sin/cos of the wrapped headings), 1 = polynomials, 0 = library
@pytest.mark.parametrize("name", list(CASES))
def test_forward_parity(name, math):
    cfg, g = CASES[name], load(name)
    model, T, N = cfg["model"], cfg["T"], cfg["N"]
    kw = {k: cfg[k] for k in SOLVER_KW if k in cfg}
    solver, ctrl = make_solver(model, T, N, lambda_=cfg["lambda_"], **kw)
    solver.set_option("math", math)
    auto = isinstance(cfg["lambda_"], str)
    twin = twin_ctrl = None
    if auto:  # the same solve at the reference's own temperature, for the end-to-end check
        twin, twin_ctrl = make_solver(model, T, N, lambda_=1.0, **kw)
        twin.set_option("math", math)
    P = oracle_problem(model, N, T, cfg.get("exploration", 0.0))
    from pi_mpc import _host

    flipped = []
    for k in range(int(g["K"])):
        x0, mean, eps = g[f"x0_{k}"], g[f"mean_in_{k}"], g[f"eps_{k}"]
        if ctrl is not None:
            ctrl.set_reference(g[f"ref_path_{k}"])
            P.set_ref_path(g[f"ref_path_{k}"])
        solver.set_warm_start(mean, g[f"sg_hist_in_{k}"])
        if cfg["lambda_"] == "MPO" and k > 0:
            solver._lambda = float(g[f"lambda_{k - 1}"])
        solver.inject_noise(torch.from_numpy(eps))
        a, s = solver.forward(torch.from_numpy(x0))
        assert a.shape == (T, P.dc) and s.shape == (1, T + 1, P.ds) and a.is_cuda
        a, s = a.cpu().numpy(), s.cpu().numpy()
        c_gpu = solver._costs.cpu().numpy()

        # (1) costs against the oracle on identical inputs
        r = P.rollout_cost(x0, mean, eps, want_margin=True)
        nflip = check_costs(c_gpu, r)

        # (2) temperature against the reference (MPO: the temperature the NEXT solve will use, mppi.py:387-398)
        lam = solver._last_lambda
        lam_ref = used_lambda(g, cfg, k)
        if cfg["lambda_"] == "LBPS":
            # the reference's own temperature moves by band_rule (nav2d: up to 1e-2) under 1-ulp changes of its costs
            lim = max(lbps_floor(solver), BAND_MARGIN * band_rule_lambda(g, k))
            parity_report.record("lambda_rel_err_LBPS", abs(lam - lam_ref) / lam_ref, lim, reference_band=band_rule_lambda(g, k),
                                 **_band_position(band_rule_lambda(g, k), abs(lam - lam_ref) / lam_ref))
            assert abs(lam - lam_ref) <= lim * lam_ref, (lam, lam_ref, lim)
            assert same_lbps_minimum(c_gpu, lam, lam_ref, tol=lim), (lam, lam_ref)  # ... and it is no worse a minimiser
        elif cfg["lambda_"] == "MPO":
            assert lam == lam_ref  # (fed from the fixture below; the rule itself is checked on lambda_next)
            lam_next, lam_next_ref = float(solver._lambda), float(g[f"lambda_{k}"])
            # the dual's Adam state is this solver's own (it has seen the reference's cost vectors up to fp32 rounding),
            # so the reference-side spread is the closed-loop band of the temperature, not the one-solve band
            mpo_band = Band.merge(band_rule_lambda(g, k), band_closed_loop(g, k)["lam"])
            lim = max(1e-4, BAND_MARGIN * mpo_band)
            parity_report.record("lambda_rel_err_MPO", abs(lam_next - lam_next_ref) / lam_next_ref, lim, reference_band=float(mpo_band),
                                 **_band_position(mpo_band, abs(lam_next - lam_next_ref) / lam_next_ref))
            assert abs(lam_next - lam_next_ref) <= lim * lam_next_ref, (k, lam_next, lam_next_ref, lim)
        else:
            if cfg["lambda_"] in LAMBDA_TOL:
                parity_report.record("lambda_rel_err_" + cfg["lambda_"], abs(lam - lam_ref) / lam_ref, LAMBDA_TOL[cfg["lambda_"]])
            assert abs(lam - lam_ref) <= LAMBDA_TOL.get(cfg["lambda_"], 0.0) * lam_ref + 1e-12

        # (3) weights/reduction/finalize against the oracle fed with the GPU's own costs
        w, st = orc.softmax_weights(c_gpu, lam)
        a_or = P.weighted_actions(w, mean, eps)
        if cfg.get("use_sg_filter"):
            a_or = _host.sg_filter_sequence(g[f"sg_hist_in_{k}"], a_or, sg_coeffs(cfg))
        check_rel("action_seq_vs_oracle_given_costs", a, a_or, TOL)
        stats = solver.last_stats()
        assert abs(stats["cmin"] - st["cmin"]) <= 1e-6 * abs(st["cmin"]) + 1e-12
        parity_report.record("ess_rel_err", abs(stats["ess"] - st["ess"]) / st["ess"], 1e-4)
        assert abs(stats["ess"] - st["ess"]) <= 1e-4 * st["ess"]
        check_rel("weights_vs_oracle", solver._weights.cpu().numpy(), w, TOL)
        check_rel("state_seq_vs_oracle_rollout", s[0], P.rollout_single(x0, a), TOL)

        # (4) end to end against the reference fixture, at the reference's temperature
        if auto:
            if twin_ctrl is not None:
                twin_ctrl.set_reference(g[f"ref_path_{k}"])
            twin._lambda = lam_ref
            twin.set_warm_start(mean, g[f"sg_hist_in_{k}"])
            twin.inject_noise(torch.from_numpy(eps))
            a, s = (t.cpu().numpy() for t in twin.forward(torch.from_numpy(x0)))
            assert np.array_equal(twin._costs.cpu().numpy(), c_gpu)
        if nflip:  # an occupancy cell flipped under <=1.5-ulp sin/cos differences: the two runs saw different maps
            flipped.append((k, nflip))
        else:
            check_end_to_end(a, s, c_gpu, g, k, cfg, *band_fixed(g, k))
    if flipped:  # every other check of every solve ran; the omitted one is reported, not passed over
        pytest.skip(f"{name}: boundary sample(s) flipped a map cell in solve(s) {flipped}; the end-to-end comparison "
                    "with the reference fixture was not applicable there (all other checks passed)")


@pytest.mark.parametrize("name", ["pendulum_T15_N256_fixed", "pendulum_T15_N200_explore", "cartpole_T10_N100_fixed",
                                  "mountaincar_T100_N256_fixed", "nav2d_T30_N256_fixed_explore", "racing_T25_N256_fixed",
                                  "pendulum_T50_N1000_essps", "cartpole_T64_N1024_essps_sg", "mjcartpole_T50_N256_fixed",
                                  "goalzone_T30_N256_fixed", "racing_T25_N4096_dense", "racing_T25_N512_explore_sg",
                                  "racing_T25_N1024_essps", "nav2d_T30_N4096_essps", "nav2d_T30_N512_lbps",
                                  "nav2d_T30_N512_mpo", "nav2d_T30_N512_sg", "nav2d_T20_N256_posterior",
                                  "pendulum_T15_N256_lbps", "pendulum_T15_N256_mpo", "nav2d_T50_N512_essps",
                                  "nav2d_T30_N4096_lbps", "nav2d_T30_N4096_mpo", "racing_T25_N4096_lbps",
                                  "racing_T25_N4096_mpo", "nav2d_T30_N512_essps_at_min", "nav2d_T30_N512_essps_at_max"])
def test_identical_seed_closed_loop_matches_reference(name):
    """`noise_source="torch_cpu"`, seed 42: the solver draws the reference's own CPU noise stream (the
    constructor consumes one draw, mppi.py:146-148) — nothing is injected.  The closed-loop solves must
    reproduce the reference's noise (bit for bit), temperatures, action and state sequences, warm start and SG
    history included; `get_samples_from_posterior` between two solves draws from the same stream (mppi.py:489-506):
    same samples, same states, and the next solve's noise is the reference's."""
    _identical_seed_closed_loop(name)


_HOST_TEMPERATURE_CASES = ["pendulum_T50_N1000_essps", "cartpole_T64_N1024_essps_sg", "racing_T25_N1024_essps",
                           "nav2d_T30_N4096_essps", "nav2d_T50_N512_essps", "pendulum_T15_N256_lbps", "nav2d_T30_N512_lbps",
                           "pendulum_T15_N256_mpo", "nav2d_T30_N512_mpo", "nav2d_T30_N4096_lbps", "racing_T25_N4096_lbps",
                           "nav2d_T30_N4096_mpo", "racing_T25_N4096_mpo", "nav2d_T30_N512_essps_at_min",
                           "nav2d_T30_N512_essps_at_max"]


def _host_temperature_params():
    """(case, mode) pairs that exist: "host" for every rule; "brent" (the library's ports probing device statistics) for
    the two searches — MPO has no search, its device-statistics step is the default path; "grid" for LBPS only — the
    device-resident ESSPS search is the default and runs in test_identical_seed_closed_loop_matches_reference."""
    out = []
    for name in _HOST_TEMPERATURE_CASES:
        rule = CASES[name]["lambda_"]
        out.append((name, "host"))
        if rule != "MPO":
            out.append((name, "brent"))
        if rule == "LBPS":
            out.append((name, "grid"))
    return out


@pytest.mark.parametrize("name,mode", _host_temperature_params())
def test_identical_seed_closed_loop_with_the_temperature_on_the_host(name, mode):
    """The north star's literal split — auto-lambda on the HOST — through the same identical-seed closed loops:
    mode "host": `auto_lambda_stats="host"`: costs[N] copied to the CPU and searched with scipy's brentq / bounded Brent /
    the Adam step in numpy fp32, the reference's own calls (mppi.py:341-370,387-398; pi_mpc/_host.py);
    mode "brent": the same root-finders inside the library (csrc/host_search.hpp ports of brentq's bracket rule and of
    scipy's bounded Brent) probing the device-side softmax statistics one temperature at a time, as host loops (LBPS: the
    DEFAULT runs this very search on the device — test_identical_seed_closed_loop_matches_reference — and must return the
    host loop's temperature to the bit: test_device_brent_*);
    mode "grid": LBPS's opt-in grid search as kernels."""
    kw = {"host": dict(auto_lambda_stats="host"), "brent": dict(lbps_search="brent_host", essps_search="brentq"),
          "grid": dict(lbps_search="grid")}[mode]
    _identical_seed_closed_loop(name, tag="_" + mode, **kw)


def _identical_seed_closed_loop(name, tag="", **solver_kw):
    cfg, g = CASES[name], load(name)
    model, T, N = cfg["model"], cfg["T"], cfg["N"]
    kw = {k: cfg[k] for k in SOLVER_KW if k in cfg}
    kw.update(solver_kw)
    solver, ctrl = make_solver(model, T, N, lambda_=cfg["lambda_"], noise_source="torch_cpu", seed=42, **kw)
    P = oracle_problem(model, N, T, cfg.get("exploration", 0.0))
    state = torch.from_numpy(g["x0_0"])
    for k in range(int(g["K"])):
        band = band_closed_loop(g, k)  # the reference's own closed loop under rounding-level changes of its costs
        assert rel_err(state.cpu().numpy(), g[f"x0_{k}"]) <= max(TOL, BAND_MARGIN * band["x0"])
        if ctrl is not None:
            env = _envs["racing"]
            ref, ctrl.current_path_index = ctrl.calc_ref_trajectory(state, env.racing_center_path,
                                                                    ctrl.current_path_index, T, DL=0.1,
                                                                    lookahead_distance=3, reference_path_interval=0.85)
            ctrl.set_reference(ref)
            assert np.array_equal(ref.numpy(), g[f"ref_path_{k}"])
        a, s = solver.forward(state)
        assert np.abs(solver._action_noises.cpu().numpy() - g[f"eps_{k}"]).max() == 0.0  # same stream, bit for bit
        c = solver._costs.cpu().numpy()
        lam, lam_ref = solver._last_lambda, used_lambda(g, cfg, k)
        if cfg["lambda_"] in ("ESSPS", "LBPS", "MPO"):
            kk = k - 1 if cfg["lambda_"] == "MPO" else k  # (MPO: this solve's weights use the temperature solve k-1 left)
            lam_band = band_closed_loop(g, kk)["lam"] if kk >= 0 else 0.0
            lim = max({"ESSPS": 1e-4, "LBPS": lbps_floor(solver), "MPO": 1e-4}[cfg["lambda_"]], BAND_MARGIN * lam_band)
            parity_report.record("closed_loop_lambda_rel_err_" + cfg["lambda_"] + tag, abs(lam - lam_ref) / lam_ref, lim,
                                 reference_band=float(lam_band), **_band_position(lam_band, abs(lam - lam_ref) / lam_ref))
            assert abs(lam - lam_ref) <= lim * lam_ref, (k, lam, lam_ref, lim)
        else:
            assert lam == lam_ref
        check_end_to_end(a.cpu().numpy(), s.cpu().numpy(), c, g, k, cfg, band["action"], band["state"], tag=tag,
                         margin=BAND_MARGIN_FAST_LBPS if grid_lbps(solver) else None)
        if "posterior_after" in g.files and int(g["posterior_after"]) == k:
            ps, pst = solver.get_samples_from_posterior(a, state, g["posterior_samples"].shape[0])
            assert rel_err(ps.cpu().numpy(), g["posterior_samples"]) <= max(TOL, BAND_MARGIN * band["action"])
            assert rel_err(pst.cpu().numpy(), g["posterior_states"]) <= max(TOL, BAND_MARGIN * max(band["action"], band["state"]))
            assert np.abs((ps - a[None]).cpu().numpy() - (g["posterior_samples"] - g[f"action_seq_{k}"][None])).max() < 1e-6
        if ctrl is not None:  # env.step of the reference loop (example/racing.py:233)
            env = _envs["racing"]
            u = torch.clamp(a[0], env.u_min, env.u_max)
            state = env.dynamics(state.cuda().unsqueeze(0), u.unsqueeze(0)).squeeze(0)
        else:
            state = s[0, 1].clone()


# BASELINE.json's configurations at FULL size against the reference ITSELF (round 5): tests/golden/make_golden.py fullsize
# ran the real reference (seed 42, two closed-loop solves) and kept outputs and summaries only; the solver draws the same
# torch-CPU stream (noise_source="torch_cpu") and must reproduce them.
FULL_SIZE = {
    "c2": ("full_c2_nav2d_T50_N65536_essps", "nav2d", dict(lambda_="ESSPS")),
    "c5": ("full_c5_cartpole_T64_N262144_essps_sg", "cartpole", dict(lambda_="ESSPS", use_sg_filter=True)),
    "c3": ("full_c3_racing_T50_N1048576_lambda1", "racing", dict(lambda_=1.0)),
    # configs[1]'s size under the reference's other search rule (round 6: the default LBPS search is the device-resident Brent)
    "c2_lbps": ("full_c2_nav2d_T50_N65536_lbps", "nav2d", dict(lambda_="LBPS")),
}


def full_size_band(g, k):
    """Reference spread of solve k: the per-solve probes (fixed temperature and rule re-run) and the closed-loop probes
    together (helpers.Band: maximum, ranks and names of all of them)."""
    fa, fs = band_fixed(g, k)
    cl = band_closed_loop(g, k)
    rl = band_rule(g, k)
    zero = Band([0.0])
    return dict(x0=cl["x0"], action=Band.merge(fa, cl["action"], *(rl[:1] if rl else ())),
                state=Band.merge(fs, cl["state"], *(rl[1:2] if rl else ())),
                lam=Band.merge(cl["lam"], rl[2] if rl else zero))


@pytest.mark.parametrize("which", ["c2", "c5", "c3", "c2_lbps"])
def test_identical_seed_full_size_matches_reference(which):
    """North star: "action_seq / state_seq match the PyTorch reference on identical RNG seeds within 1e-5" at the sizes the
    metric is quoted on (/root/reference/src/pi_mpc/mppi.py:255-460 run in the build container, outputs only).  Checked per
    solve: the noise block (float64 sum and sum of squares of all N*T*dc values, first / last rows, the rows of the
    reference's 32 best samples: bit for bit), the N costs through their minimum, maximum, float64 sum, eight order
    statistics, a 64-bin histogram and the 32 smallest (index, cost) pairs — the arg-min among 2^20 samples must be the
    reference's — then temperature, ESS, action_seq and state_seq."""
    name, model, kw = FULL_SIZE[which]
    g = load(name)
    N, T, K = int(g["N"]), int(g["T"]), int(g["K"])
    solver, ctrl = make_solver(model, T, N, noise_source="torch_cpu", seed=int(g["seed"]), **kw)
    state = torch.from_numpy(g["x0_0"])
    mc = MODEL_CFG[model]
    for k in range(K):
        band = full_size_band(g, k)
        assert rel_err(state.cpu().numpy(), g[f"x0_{k}"]) <= max(TOL, band["x0"])
        if ctrl is not None:
            env = _envs["racing"]
            ref, ctrl.current_path_index = ctrl.calc_ref_trajectory(state, env.racing_center_path, ctrl.current_path_index,
                                                                    T, DL=0.1, lookahead_distance=3,
                                                                    reference_path_interval=0.85)
            ctrl.set_reference(ref)
            assert np.array_equal(ref.numpy(), g[f"ref_path_{k}"]) and ctrl.current_path_index == int(g[f"cind_out_{k}"])
        a, s = solver.forward(state)
        # ---- the noise: the reference's block, bit for bit
        eps = solver._action_noises.cpu().numpy()
        e64 = eps.astype(np.float64)
        assert float(e64.sum()) == float(g[f"eps_sum64_{k}"]) and float((e64 * e64).sum()) == float(g[f"eps_sumsq64_{k}"])
        del e64
        assert np.array_equal(eps[:2], g[f"eps_head_{k}"]) and np.array_equal(eps[-1:], g[f"eps_tail_{k}"])
        top_i = g[f"top32_idx_{k}"]
        assert np.array_equal(eps[top_i], g[f"top32_eps_{k}"])
        # ---- the costs
        c = solver._costs.cpu().numpy()
        c_ref_top = g[f"top32_cost_{k}"]
        scale = float(g[f"cmax_{k}"])
        tag = f"_full_{which}"
        parity_report.record("cost_rel_err_reference_top32" + tag, np.abs(c[top_i] - c_ref_top).max() / scale, TOL)
        assert np.abs(c[top_i] - c_ref_top).max() <= TOL * scale
        assert abs(float(c.min()) - float(g[f"cmin_{k}"])) <= TOL * scale and abs(float(c.max()) - scale) <= TOL * scale
        rel_sum = abs(float(c.astype(np.float64).sum()) - float(g[f"costs_sum64_{k}"])) / abs(float(g[f"costs_sum64_{k}"]))
        # (from the second solve on the start state is the closed loop's own — equal to the reference's to band["x0"] — and the
        # costs follow it)
        sum_tol = max(1e-6, float(band["x0"]))
        parity_report.record("cost_sum_rel_err_vs_reference" + tag, rel_sum, sum_tol)
        assert rel_sum <= sum_tol
        q = np.sort(c)[g[f"quantile_ranks_{k}"]]
        assert np.abs(q - g[f"quantiles_{k}"]).max() <= TOL * scale
        hist = np.histogram(c.astype(np.float64), bins=g[f"hist_edges_{k}"])[0]
        moved = int(np.abs(hist - g[f"hist_{k}"]).sum())
        parity_report.record("cost_histogram_L1_vs_reference" + tag, moved / N, 2e-3)
        assert moved <= 2e-3 * N, (moved, N)
        # the 32 smallest (cost, index) pairs in the reference's order, up to the first pair of neighbours the
        # reference itself separates by less than 4 ulps of the cost scale
        order = np.lexsort((np.arange(N), c))[:32]
        gaps = np.diff(c_ref_top.astype(np.float64))
        amb = np.nonzero(gaps < 4 * EPS32 * float(np.abs(c_ref_top).max()))[0]
        need = int(amb[0]) if len(amb) else 32
        agree = int(np.argmin(np.append(order == top_i, False)))
        parity_report.record("reference_top32_order_reproduced" + tag, need - min(agree, need), 0, agree=agree, needed=need)
        assert agree >= need, (agree, need, order[:8], top_i[:8])
        # ---- temperature and effective sample size
        lam, lam_ref = solver._last_lambda, float(g[f"lambda_{k}"])
        if isinstance(kw["lambda_"], str):
            lim = max(LAMBDA_TOL.get(kw["lambda_"], LBPS_TOL_BRENT), band["lam"])
            parity_report.record("lambda_rel_err_" + kw["lambda_"] + tag, abs(lam - lam_ref) / lam_ref, lim,
                                 reference_band=float(band["lam"]), **_band_position(band["lam"], abs(lam - lam_ref) / lam_ref))
            assert abs(lam - lam_ref) <= lim * lam_ref, (lam, lam_ref)
        else:
            assert lam == lam_ref
        ess, ess_ref = solver.last_stats()["ess"], float(g[f"ess_{k}"])
        w64, _ = orc.softmax_weights(c, lam)
        ess_own = 1.0 / float(np.sum(w64.astype(np.float64) ** 2))
        assert abs(ess - ess_own) <= 1e-4 * ess_own  # the device's statistic against a float64 evaluation of its own costs
        if isinstance(kw["lambda_"], str):
            # (ESSPS pins the ESS itself; LBPS's temperature moves by its band under 1-ulp changes of the reference's costs
            # — 5e-3 at this size — and the ESS follows it with a logarithmic slope of a few)
            ess_tol = 1e-3 if kw["lambda_"] == "ESSPS" else max(1e-3, 4.0 * float(band["lam"]))
            assert abs(ess - ess_ref) <= ess_tol * ess_ref, (ess, ess_ref, ess_tol)
        # ---- action_seq / state_seq
        if float(g[f"top32_weight_{k}"][0]) >= 1.0 - 1e-6 and need >= 1:  # arg-min regime: an exact statement exists
            U = np.clip(g[f"mean_in_{k}"] + g[f"top32_eps_{k}"][0], np.float32(mc["u_min"]), np.float32(mc["u_max"]))
            assert int(np.argmin(c)) == int(top_i[0])
            assert np.abs(a.cpu().numpy() - U).max() <= 1e-6 * np.abs(U).max()
        # (--strict-parity: BASELINE's configurations are held to the north star's plain 1e-5, without the band.  Not the LBPS
        # twin of C2: there the REFERENCE's own temperature moves by 5e-3 under 1-ulp changes of its costs — its objective is
        # flat to fp32 noise — and its action_seq with it, by 2.3e-3; the device's distance is 6e-5)
        strict = kw["lambda_"] != "LBPS"
        check_banded("action_seq_vs_reference_fixture" + tag, a.cpu().numpy(), g[f"action_seq_{k}"], band["action"], strict_ok=strict)
        check_banded("state_seq_vs_reference_fixture" + tag, s.cpu().numpy(), g[f"state_seq_{k}"], band["state"], strict_ok=strict)
        # get_top_samples (mppi.py:462-487) at full size: the 32 largest weights are the reference's, in its order
        ts, tw = solver.get_top_samples(32)
        w_ref = g[f"top32_weight_{k}"]
        werr = float(np.abs(tw.cpu().numpy() - w_ref).max() / w_ref.max())
        # (under LBPS the reference's own temperature moves by band["lam"] — 5e-3 at this size — when its costs move by one ulp,
        # and a weight follows the temperature with a logarithmic slope of (c_i - E_w[c]) / lambda = O(1) for the best samples)
        wtol = 1e-4 if kw["lambda_"] != "LBPS" else max(1e-4, float(band["lam"]))
        parity_report.record("top32_weights_vs_reference" + tag, werr, wtol)
        assert werr <= wtol, (werr, wtol)
        assert ts.shape == (32, T + 1, s.shape[-1]) and torch.equal(ts[:, 0, :], state.to(ts.device).expand(32, -1))
        if float(w_ref[0]) >= 1.0 - 1e-6:  # arg-min regime: the best sample's trajectory IS the solution's rollout
            assert rel_err(ts[0].cpu().numpy(), g[f"state_seq_{k}"][0]) <= TOL
        if ctrl is not None:
            env = _envs["racing"]
            u = torch.clamp(a[0], env.u_min, env.u_max)
            state = env.dynamics(state.cuda().unsqueeze(0), u.unsqueeze(0)).squeeze(0)
        else:
            state = s[0, 1].clone()


def test_top_samples_match_reference():
    name = "pendulum_T15_N256_fixed"
    cfg, g = CASES[name], load(name)
    solver, _ = make_solver("pendulum", 15, 256, lambda_=1.0)
    solver.set_warm_start(g["mean_in_0"])
    solver.inject_noise(torch.from_numpy(g["eps_0"]))
    solver.forward(torch.from_numpy(g["x0_0"]))
    ts, tw = solver.get_top_samples(8)
    assert ts.shape == (8, 16, 2) and tw.shape == (8,)
    assert rel_err(tw.cpu().numpy(), g["top8_weights_0"]) < TOL
    assert rel_err(ts.cpu().numpy(), g["top8_states_0"]) < TOL
    # mountaincar: the stored trajectories are the mutated views (SURVEY B-Q7)
    g = load("mountaincar_T100_N256_fixed")
    solver, _ = make_solver("mountaincar", 100, 256, lambda_=0.1)
    solver.set_warm_start(g["mean_in_0"])
    solver.inject_noise(torch.from_numpy(g["eps_0"]))
    solver.forward(torch.from_numpy(g["x0_0"]))
    ts, tw = solver.get_top_samples(8)
    assert rel_err(ts.cpu().numpy(), g["top8_states_0"]) < TOL


@pytest.mark.parametrize("N,k,lam", [(1000, 1, 1.0), (4096, 300, 50.0), (777, 777, 500.0), (1 << 20, 300, 1.0),
                                     (1 << 20, 1024, 2000.0), (5000, 1025, 800.0), (1 << 17, 4096, 2000.0),
                                     (3000, 3000, 500.0), (1 << 16, 40000, 5000.0),
                                     # the one-launch query of small problems: one row (<= 1024 samples) sorted directly, two
                                     # to four rows through the radix select inside the block; sizes around the row edges
                                     (1024, 1024, 500.0), (1025, 7, 50.0), (2048, 1000, 500.0), (2500, 64, 50.0),
                                     (3000, 300, 50.0), (4096, 1024, 2000.0), (4097, 300, 50.0)])
def test_device_top_k_selects_the_smallest_costs(N, k, lam):
    """mppi_top_samples (radix select + sort + re-roll on the device) against a host sort of the same costs,
    the softmax weights, and the index-driven re-roll path; twice, to check the select state is left clean.  Any
    k <= N like the reference (mppi.py:462-487): up to 1024 candidates are sorted in LDS, more by the multi-pass bitonic
    sort in HBM (k = 1025, 4096, N = 3000 and 40 000 of 65 536); up to 4096 samples and k <= 1024 the whole query is one
    launch (the reference examples' sizes)."""
    solver, ctrl = make_solver("racing", 50, N, lambda_=lam)
    env = _envs["racing"]
    x0 = env._robot_state.clone()
    ref, _ = ctrl.calc_ref_trajectory(x0, env.racing_center_path, 0, 50, DL=0.1, lookahead_distance=3,
                                      reference_path_interval=0.85)
    ctrl.set_reference(ref)
    for _ in range(2):
        solver.forward(x0)
        ts, tw = solver.get_top_samples(k)
        costs = solver._costs.cpu().numpy()
        order = np.lexsort((np.arange(N), costs))[:k]  # ascending cost, then index
        x = (-costs.astype(np.float64)) / lam
        w_all = np.exp(x - x.max())
        w_all /= w_all.sum()
        got_w = tw.cpu().numpy()
        assert got_w.shape == (k,) and np.all(np.diff(got_w) <= 0)
        assert np.abs(got_w - w_all[order]).max() <= 2e-5 * w_all.max()
        # the same samples re-rolled through the index path (noise tiles materialised on demand): identical
        out2 = torch.empty_like(ts)
        idx = torch.from_numpy(order.astype(np.int64)).to(ts.device)
        solver._h.call("mppi_rollout_samples", idx.data_ptr(), k, out2.data_ptr(), solver._stream())
        tie_free = len(np.unique(costs[order])) == k and (k == N or costs[order][-1] < np.partition(costs, k)[k])
        if tie_free:
            assert torch.equal(ts, out2)
        assert ts.shape == (k, 51, 4) and torch.isfinite(ts).all()
        assert torch.equal(ts[:, 0, :], x0.to(ts.device).expand(k, 4))


@pytest.mark.parametrize("model,T,N,expl", [("pendulum", 1, 5, 0.0), ("pendulum", 2, 64, 0.5), ("pendulum", 7, 65, 1.0),
                                            ("racing", 1, 3, 0.0), ("racing", 3, 130, 0.3), ("nav2d", 2, 1, 0.0),
                                            ("cartpole", 5, 63, 0.0), ("mountaincar", 9, 200, 0.9),
                                            # long rows: 32 groups per wave (T*dc > 128), two column chunks (> 512)
                                            ("racing", 80, 300, 0.2), ("nav2d", 300, 130, 0.1)])
def test_edge_sizes_against_oracle(model, T, N, expl):
    """Ragged shapes: horizons shorter than one float4 group, sample counts that are not a multiple of 64
    (or smaller than a wave), exploration splits that fall inside a tile, all against the oracle."""
    solver, ctrl = make_solver(model, T, N, lambda_=2.5, exploration=expl)
    x0 = {"pendulum": [3.0, 0.1], "cartpole": [0.01, 0.0, 0.02, 0.0], "mountaincar": [-0.5, 0.0],
          "nav2d": [-9.0, -9.0, 0.785], "racing": None}[model]
    P = oracle_problem(model, N, T, expl)
    if ctrl is not None:
        env = _envs["racing"]
        x0 = env._robot_state.cpu().numpy()
        ref, _ = ctrl.calc_ref_trajectory(env._robot_state, env.racing_center_path, 0, T, DL=0.1,
                                          lookahead_distance=3, reference_path_interval=0.85)
        ctrl.set_reference(ref)
        P.set_ref_path(ref.numpy())
    x0 = np.asarray(x0, np.float32)
    mean = (np.random.default_rng(T * 131 + N).standard_normal((T, P.dc)) * 0.2).astype(np.float32)
    for k in range(2):
        solver.set_warm_start(mean)
        a, s = solver.forward(torch.from_numpy(x0))
        eps = solver._action_noises.cpu().numpy()
        c = solver._costs.cpu().numpy()
        assert eps.shape == (N, T, P.dc) and c.shape == (N,)
        r = P.rollout_cost(x0, mean, eps, want_margin=True, want_U=True)
        check_costs(c, r)
        w, st = orc.softmax_weights(c, 2.5)
        assert rel_err(a.cpu().numpy(), P.weighted_actions(w, mean, eps)) < TOL
        assert rel_err(s.cpu().numpy()[0], P.rollout_single(x0, a.cpu().numpy())) < TOL
        assert np.array_equal(solver._perturbed_actions_for(torch.from_numpy(mean).cuda()).cpu().numpy(), r["U"])
        if k == 0:  # the reference's per-sample buffers, rebuilt on demand (U bit-exact, S to tolerance)
            assert np.array_equal(solver._perturbed_action_seqs.cpu().numpy(), r["U"])
            S = P.rollout_cost(x0, mean, eps, want_S=True)["S"]
            assert solver._state_seq_batch.shape == S.shape and rel_err(solver._state_seq_batch.cpu().numpy(), S) < TOL
        k8 = min(8, N)
        ts, tw = solver.get_top_samples(k8)
        assert ts.shape == (k8, T + 1, P.ds) and rel_err(tw.cpu().numpy(), np.sort(w)[::-1][:k8]) < TOL
        mean = a.cpu().numpy()


# ------------------------------------------------------------------------------ sampler / layouts
def test_sampler_matches_philox_restatement():
    solver, _ = make_solver("racing", 50, 1000, lambda_=1.0)
    h, st = solver._h, solver._stream()
    h.call("mppi_sample", 3, st)
    eps = solver._action_noises.cpu().numpy()
    ref = orc.philox_normal(42, 3, 0, 1000, 50, 2, [0.5, 0.1])
    err = np.abs(eps - ref) / np.array([0.5, 0.1], np.float32)
    print("sampler max abs err / sigma:", err.max())
    assert err.max() < 1e-4  # hardware v_sin/v_cos/v_log vs libm
    # dc = 1 model, odd row length (T*dc = 15 -> 4 groups, one padded)
    solver, _ = make_solver("pendulum", 15, 200, lambda_=1.0)
    solver._h.call("mppi_sample", 1, solver._stream())
    eps = solver._action_noises.cpu().numpy()
    assert np.abs(eps - orc.philox_normal(42, 1, 0, 200, 15, 1, [1.0])).max() < 1e-4


def test_sampler_moments_full_size():
    N, T = 1 << 20, 50
    solver, _ = make_solver("racing", T, N, lambda_=1.0)
    solver._h.call("mppi_sample", 1, solver._stream())
    e = solver._action_noises
    for k, sg in enumerate((0.5, 0.1)):
        x = e[..., k].double()
        assert abs(float(x.mean())) < 5 * sg / np.sqrt(N * T)
        assert abs(float(x.std()) - sg) < 5e-4 * sg
        assert abs(float((x ** 4).mean()) / sg ** 4 - 3.0) < 0.01  # kurtosis of a normal
    # lag-1 correlation across time and across samples
    x = e[..., 0].double()
    assert abs(float((x[:, 1:] * x[:, :-1]).mean())) / 0.25 < 1e-3
    assert abs(float((x[1:] * x[:-1]).mean())) / 0.25 < 1e-3
    # a different solve index gives a different draw
    solver._h.call("mppi_sample", 2, solver._stream())
    assert float((solver._action_noises - e).abs().max()) > 0.1


@pytest.mark.parametrize("model,T,N", [("racing", 50, 5000), ("nav2d", 30, 1000), ("pendulum", 15, 777),
                                       ("cartpole", 64, 640), ("mountaincar", 100, 320)])
def test_regenerated_noise_equals_materialised_tiles(model, T, N):
    """`noise_regen` = 1 (Philox regenerated in the rollout / reduction registers) and = 0 (tiles written
    by sample_kernel and read back) are the same computation: costs and action must be bit-identical."""
    outs = []
    for regen in (1, 0):
        solver, ctrl = make_solver(model, T, N, lambda_=50.0 if model in ("racing", "nav2d") else 1.0)
        solver.set_option("noise_regen", regen)
        solver.set_option("fused_solve", 0)  # (the single-launch solve sums the weighted rows in another order)
        if ctrl is not None:
            env = _envs["racing"]
            ref, _ = ctrl.calc_ref_trajectory(env._robot_state, env.racing_center_path, 0, T, DL=0.1,
                                              lookahead_distance=3, reference_path_interval=0.85)
            ctrl.set_reference(ref)
        x0 = {"racing": None, "nav2d": [-9.0, -9.0, 0.785], "pendulum": [3.0, 0.1], "cartpole": [0.01, 0, 0.02, 0],
              "mountaincar": [-0.5, 0.0]}[model]
        x0 = _envs["racing"]._robot_state.clone() if x0 is None else torch.tensor(x0, dtype=torch.float32)
        a1, s1 = solver.forward(x0)
        a2, s2 = solver.forward(x0)  # second solve: warm start + next solve index
        outs.append((solver._costs.cpu(), a1.cpu(), s1.cpu(), a2.cpu(), s2.cpu(), solver._action_noises.cpu()))
    for x, y in zip(*outs):
        assert torch.equal(x, y)


@pytest.mark.parametrize("model,T,N", [("racing", 50, 3000), ("pendulum", 15, 500), ("mountaincar", 100, 300),
                                       ("nav2d", 30, 777), ("cartpole", 64, 256)])
def test_wavefront_per_trajectory_variant_agrees(model, T, N):
    """The north star's literal mapping (one wavefront per trajectory, serial recurrence on lane 0,
    time-parallel costs, shuffle reduction) computes the same costs as the lane-per-trajectory kernel up
    to the summation order of the stage costs."""
    solver, ctrl = make_solver(model, T, N, lambda_=3.0, exploration=0.1)
    if ctrl is not None:
        env = _envs["racing"]
        ref, _ = ctrl.calc_ref_trajectory(env._robot_state, env.racing_center_path, 0, T, DL=0.1,
                                          lookahead_distance=3, reference_path_interval=0.85)
        ctrl.set_reference(ref)
        x0 = env._robot_state.clone()
    else:
        x0 = torch.tensor({"pendulum": [3.0, 0.1], "mountaincar": [-0.5, 0.0], "nav2d": [-9.0, -9.0, 0.785],
                           "cartpole": [0.01, 0.0, 0.02, 0.0]}[model])
    mean = (np.random.default_rng(N).standard_normal((T, solver._dim_control)) * 0.2).astype(np.float32)
    res = []
    for mapping in (0, 1):
        solver.set_option("mapping", mapping)
        solver.set_warm_start(mean)
        solver._solve_idx = 5
        a, s = solver.forward(x0)
        res.append((solver._costs.cpu().numpy(), a.cpu().numpy()))
    assert rel_err(res[1][0], res[0][0]) < 2e-6
    assert rel_err(res[1][1], res[0][1]) < 1e-4


def test_inject_export_roundtrip_and_clamp():
    rng = np.random.default_rng(5)
    for model, T, N in (("racing", 50, 1000), ("pendulum", 15, 130), ("mountaincar", 100, 65), ("cartpole", 64, 64)):
        solver, _ = make_solver(model, T, N, lambda_=1.0, exploration=0.25)
        dc = solver._dim_control
        eps = rng.standard_normal((N, T, dc)).astype(np.float32)
        mean = rng.standard_normal((T, dc)).astype(np.float32) * 0.3
        solver.set_warm_start(mean)
        solver._h.call("mppi_inject_noise", C.c_void_p(torch.from_numpy(eps).cuda().data_ptr()), solver._stream())
        torch.cuda.synchronize()
        assert np.array_equal(solver._action_noises.cpu().numpy(), eps)
        u = solver._perturbed_actions_for(torch.from_numpy(mean).cuda()).cpu().numpy()
        thr = int(N * 0.75)
        cfg = MODEL_CFG[model]
        ref = eps.copy()
        ref[:thr] = mean + eps[:thr]
        ref = np.minimum(np.maximum(ref, np.array(cfg["u_min"], np.float32)), np.array(cfg["u_max"], np.float32))
        assert np.array_equal(u, ref)


def _summary(solver, lam):
    out = torch.zeros(4 + solver._horizon * solver._dim_control, device="cuda")
    solver._h.call("mppi_weights_reduce", float(lam), C.c_void_p(out.data_ptr()), solver._stream())
    return out


def test_shard_invariance_and_combine():
    """Two half shards (sample_offset) reproduce the unsharded noise bit for bit and, combined by
    mppi_finalize(num_shards=2), the unsharded action."""
    from mppi_playground_amd import _capi

    N, T = 4096, 50
    full, ctrl = make_solver("racing", T, N, lambda_=5000.0)  # large lambda: many samples carry weight
    x0 = _envs["racing"]._robot_state.clone()
    ref, _ = ctrl.calc_ref_trajectory(x0, _envs["racing"].racing_center_path, 0, T, DL=0.1, lookahead_distance=3,
                                      reference_path_interval=0.85)
    ctrl.set_reference(ref)
    a_full, s_full = full.forward(x0)
    eps_full = full._action_noises.cpu().numpy()
    stats_full = full.last_stats()

    halves, sums = [], []
    for r in range(2):
        sol, c2 = make_solver("racing", T, N // 2, lambda_=5000.0)
        # re-create the handle as shard r of a global N
        cfg = _capi.MppiConfig()
        cfg.model, cfg.horizon, cfg.dim_state, cfg.dim_control = 4, T, 4, 2
        cfg.num_samples, cfg.sample_offset, cfg.inherit_count = N // 2, r * (N // 2), N
        for k in range(2):
            cfg.u_min[k], cfg.u_max[k], cfg.sigmas[k] = (MODEL_CFG["racing"][q][k] for q in ("u_min", "u_max", "sigmas"))
        cfg.seed, cfg.device = 42, 0
        sol._h.close()
        sol._h = _capi.Handle(cfg)
        sol._uploaded, sol._params_set = {}, None
        c2.set_reference(ref)
        sol._h.call("mppi_set_state", C.c_void_p(x0.data_ptr()), 1, sol._stream())
        sol._refresh_model_inputs()
        sol._h.call("mppi_sample", 1, sol._stream())
        sol._h.call("mppi_rollout_cost", sol._stream())
        sums.append(_summary(sol, 5000.0))
        halves.append(sol)
        assert np.array_equal(sol._action_noises.cpu().numpy(), eps_full[r * (N // 2):(r + 1) * (N // 2)])
    both = torch.stack(sums).contiguous()
    a = torch.zeros(T, 2, device="cuda")
    s = torch.zeros(1, T + 1, 4, device="cuda")
    stats = torch.zeros(4, device="cuda")
    halves[0]._h.call("mppi_finalize", C.c_void_p(both.data_ptr()), 2, 5000.0, 0, C.c_void_p(a.data_ptr()),
                      C.c_void_p(s.data_ptr()), C.c_void_p(stats.data_ptr()), halves[0]._stream())
    halves[0].join_state_seq()  # (raw C-ABI use of a handle with lazily completed state sequences: join before reading)
    assert rel_err(a.cpu().numpy(), a_full.cpu().numpy()) < 2e-6
    assert rel_err(s.cpu().numpy(), s_full.cpu().numpy()) < 2e-6
    st = stats.cpu().numpy()
    assert abs(st[0] - stats_full["cmin"]) == 0.0
    assert abs(st[1] - stats_full["sum_e"]) <= 1e-5 * stats_full["sum_e"]


def _sharded_worker(rank, world, port, q, exchange="nccl"):
    import os

    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["MPPI_EXCHANGE"] = exchange
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import mppi_playground_amd  # noqa: F401

        # (lazy_state_seq forced on: the sharded finalize — gathered summaries, or the peer-to-peer poll — with the batch-1
        # rollout completed by the next rollout launch / on first use, like a rank of a large sharded solve)
        solver, ctrl = make_solver("racing", 50, 8192, lambda_=5000.0, shard_samples=True, lazy_state_seq=True)
        assert solver._p2p == (exchange == "p2p") and solver._lazy_state
        env = _envs["racing"]
        x0 = env._robot_state.clone()
        ref, _ = ctrl.calc_ref_trajectory(x0, env.racing_center_path, 0, 50, DL=0.1, lookahead_distance=3,
                                          reference_path_interval=0.85)
        ctrl.set_reference(ref)
        a1, s1 = solver.forward(x0)
        a2, s2 = solver.forward(x0)
        st = solver.last_stats()
        ts, tw = solver.get_top_samples(24)  # sharded: candidates merged across ranks, re-rolled on every rank
        tsb, twb = solver.get_top_samples(5000)  # more than a rank owns (4096) and more than one block sorts (1024)
        for _ in range(50):  # many back-to-back solves: the exchange buffers alternate, ranks drift apart freely
            a_prev = solver._previous_action_seq.clone()
            idx_last = solver._solve_idx
            a3, _ = solver.forward(x0)
        # a second model on the same ranks: nav2d with an exploration split and the ESSPS search (sharded
        # statistics: one all_gather per 32-temperature grid)
        nav, _ = make_solver("nav2d", 30, 4096, lambda_="ESSPS", exploration=0.25, shard_samples=True)
        xn = torch.tensor([-9.0, -9.0, 0.785])
        nav.forward(xn)
        an, sn = nav.forward(xn)
        q.put((rank, a1.cpu().numpy(), s1.cpu().numpy(), a2.cpu().numpy(), st["sum_e"], st["cmin"],
               ts.cpu().numpy(), tw.cpu().numpy(), a3.cpu().numpy(), an.cpu().numpy(), sn.cpu().numpy(),
               nav._last_lambda, a_prev.cpu().numpy(), idx_last, tsb.cpu().numpy(), twb.cpu().numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["nccl", "p2p"])
def test_two_rank_sharded_solver_matches_single(exchange):
    """The whole sharded forward() (shard_samples=True) with two ranks — both on this GPU, gloo instead of RCCL —
    against the unsharded solver; exchange = one all_gather of the 4+T*dc summary per solve, or the library's
    peer-to-peer buffer exchange (IPC-mapped fine-grained buffers, polled by finalize_kernel)."""
    _need_gpu()
    import socket

    import torch.multiprocessing as mp

    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q, exchange)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    single, ctrl = make_solver("racing", 50, 8192, lambda_=5000.0)
    env = _envs["racing"]
    x0 = env._robot_state.clone()
    ref, _ = ctrl.calc_ref_trajectory(x0, env.racing_center_path, 0, 50, DL=0.1, lookahead_distance=3,
                                      reference_path_interval=0.85)
    ctrl.set_reference(ref)
    a1, s1 = single.forward(x0)
    a2, _ = single.forward(x0)
    cmin_after_two = single.last_stats()["cmin"]
    ts, tw = single.get_top_samples(24)
    tsb, twb = single.get_top_samples(5000)
    for r in res:  # ... also when k exceeds what one rank owns and what one block sorts
        assert rel_err(r[14], tsb.cpu().numpy()) < 1e-5 and rel_err(r[15], twb.cpu().numpy()) < 1e-5
    assert np.array_equal(res[0][14], res[1][14]) and np.array_equal(res[0][15], res[1][15])
    for r in res:  # the sharded top samples are the unsharded ones (same global indices, same noise; the warm
        # start they are rolled around differs in the last bits between the sharded and the single combine)
        assert rel_err(r[6], ts.cpu().numpy()) < 1e-5
        assert rel_err(r[7], tw.cpu().numpy()) < 1e-5
    assert np.array_equal(res[0][6], res[1][6]) and np.array_equal(res[0][7], res[1][7])
    # solve 52 of the sharded run, repeated unsharded from the SAME warm start and noise index: the two combines
    # (per-shard minima rescaled vs one global minimum) differ by rounding only.  (Left to themselves the two runs
    # drift apart over 50 warm-started solves — each step feeds its last-bit differences to the next — which says
    # nothing about either; the ranks, which must agree exactly, do.)
    assert np.array_equal(res[0][8], res[1][8]) and np.array_equal(res[0][12], res[1][12])
    single.set_warm_start(res[0][12])
    single._solve_idx = res[0][13]
    a3, _ = single.forward(x0)
    for r in res:
        assert rel_err(r[8], a3.cpu().numpy()) < 4e-6
    nav, _ = make_solver("nav2d", 30, 4096, lambda_="ESSPS", exploration=0.25)
    xn = torch.tensor([-9.0, -9.0, 0.785])
    nav.forward(xn)
    an, sn = nav.forward(xn)
    for r in res:
        assert abs(r[11] - nav._last_lambda) <= 1e-5 * nav._last_lambda
        assert rel_err(r[9], an.cpu().numpy()) < 2e-5 and rel_err(r[10], sn.cpu().numpy()) < 2e-5
    assert np.array_equal(res[0][9], res[1][9]) and res[0][11] == res[1][11]
    for r in res:  # every rank ends up with the same, correct answer
        assert rel_err(r[1], a1.cpu().numpy()) < 2e-6 and rel_err(r[2], s1.cpu().numpy()) < 2e-6
        assert rel_err(r[3], a2.cpu().numpy()) < 4e-6
        assert abs(r[5] - cmin_after_two) <= 1e-6 * abs(cmin_after_two)  # (second solve: warm starts differ in the last bits)
    assert np.array_equal(res[0][1], res[1][1])


# ------------------------------------------------------------------------------ full size (BASELINE configs)
def test_racing_full_size_against_oracle():
    """C3: racing N = 1,048,576, T = 50, lambda = 1.  The oracle (8 threads) takes a few seconds."""
    N, T = 1 << 20, 50
    solver, ctrl = make_solver("racing", T, N, lambda_=1.0)
    env = _envs["racing"]
    x0 = env._robot_state.clone()
    ref, _ = ctrl.calc_ref_trajectory(x0, env.racing_center_path, 0, T, DL=0.1, lookahead_distance=3,
                                      reference_path_interval=0.85)
    ctrl.set_reference(ref)
    a1, s1 = solver.forward(x0)
    c_gpu = solver._costs.cpu().numpy()
    eps = solver._action_noises.cpu().numpy()
    stats = solver.last_stats()
    P = oracle_problem("racing", N, T, ref_path=ref.numpy())
    mean = np.zeros((T, 2), np.float32)
    r = P.rollout_cost(x0.cpu().numpy(), mean, eps, want_margin=True)
    nflip = check_costs(c_gpu, r)
    print("full-size racing: boundary flips", nflip, "of", N)
    w, st = orc.softmax_weights(c_gpu, 1.0)
    check_rel("action_seq_vs_oracle_given_costs", a1.cpu().numpy(), P.weighted_actions(w, mean, eps), TOL)
    assert abs(stats["ess"] - st["ess"]) <= 1e-4 * st["ess"]
    check_rel("state_seq_vs_oracle_rollout", s1.cpu().numpy()[0], P.rollout_single(x0.cpu().numpy(), a1.cpu().numpy()), TOL)
    # the same costs under a DENSE softmax (lambda = 5000: thousands of samples carry weight): the weighted reduction over
    # all 2^20 samples against the oracle's float64 sums
    dense, cd = make_solver("racing", T, N, lambda_=5000.0)
    cd.set_reference(ref)
    ad, sd = dense.forward(x0)
    assert torch.equal(dense._costs, solver._costs)  # same seed and solve index: the same noise, the same costs
    wd, std = orc.softmax_weights(c_gpu, 5000.0)
    assert std["ess"] > 100
    check_rel("action_seq_vs_oracle_given_costs", ad.cpu().numpy(), P.weighted_actions(wd, mean, eps), TOL)
    parity_report.record("ess_rel_err", abs(dense.last_stats()["ess"] - std["ess"]) / std["ess"], 1e-4)
    assert abs(dense.last_stats()["ess"] - std["ess"]) <= 1e-4 * std["ess"]
    check_rel("state_seq_vs_oracle_rollout", sd.cpu().numpy()[0], P.rollout_single(x0.cpu().numpy(), ad.cpu().numpy()), TOL)
    # size-independent properties: weights sum to one, bounds respected, determinism
    assert abs(float(solver._weights.double().sum()) - 1.0) < 1e-5
    lo, hi = np.array(MODEL_CFG["racing"]["u_min"]), np.array(MODEL_CFG["racing"]["u_max"])
    assert np.all(a1.cpu().numpy() >= lo - 1e-6) and np.all(a1.cpu().numpy() <= hi + 1e-6)
    solver2, ctrl2 = make_solver("racing", T, N, lambda_=1.0)
    ctrl2.set_reference(ref)
    a2, s2 = solver2.forward(x0)
    assert torch.equal(a1, a2) and torch.equal(s1, s2)  # same seed, same solve index -> bit-identical
    # second solve warm-starts from the first (no time shift, mppi.py:452)
    a3, _ = solver.forward(x0)
    assert torch.isfinite(a3).all()
    mean_now = torch.empty(T, 2, device="cuda")
    solver._h.call("mppi_get_mean", C.c_void_p(mean_now.data_ptr()), 1, solver._stream())
    assert torch.equal(mean_now, a3)


def test_uniform_costs_give_sample_mean_full_size():
    """Linearity check of the weighted reduction at C5 size: equal costs -> plain mean of U."""
    N, T = 262144, 64
    solver, _ = make_solver("cartpole", T, N, lambda_=1.0)
    h, st = solver._h, solver._stream()
    h.call("mppi_sample", 1, st)
    c = torch.full((N,), 3.25, device="cuda")
    h.call("mppi_set_costs", C.c_void_p(c.data_ptr()), 1, st)
    h.call("mppi_weights_reduce", 1.0, None, st)
    a = torch.zeros(T, 1, device="cuda")
    stats = torch.zeros(4, device="cuda")
    h.call("mppi_finalize", None, 1, 1.0, 0, C.c_void_p(a.data_ptr()), None, C.c_void_p(stats.data_ptr()), st)
    u = solver._perturbed_actions_for(torch.zeros(T, 1, device="cuda"))
    assert rel_err(a.cpu().numpy(), u.double().mean(dim=0).cpu().numpy()) < 1e-5
    s = stats.cpu().numpy()
    assert s[0] == 3.25 and abs(s[1] - N) < 1e-3 * N and abs(s[1] * s[1] / s[2] - N) < 1e-3 * N  # ESS = N


@pytest.mark.parametrize("model,T,N,lam", [("nav2d", 50, 65536, "ESSPS"), ("cartpole", 64, 262144, "ESSPS")])
def test_baseline_configs_against_oracle(model, T, N, lam):
    """C2 / C5 at full size: costs and action against the oracle on the device-drawn noise."""
    kw = dict(use_sg_filter=True) if model == "cartpole" else {}
    solver, _ = make_solver(model, T, N, lambda_=lam, **kw)
    x0 = {"nav2d": np.array([-9.0, -9.0, np.pi / 4], np.float32),
          "cartpole": np.array([0.01, 0.0, 0.02, 0.0], np.float32)}[model]
    a, s = solver.forward(torch.from_numpy(x0))
    c_gpu = solver._costs.cpu().numpy()
    eps = solver._action_noises.cpu().numpy()
    P = oracle_problem(model, N, T)
    mean = np.zeros((T, P.dc), np.float32)
    r = P.rollout_cost(x0, mean, eps, want_margin=True)
    check_costs(c_gpu, r)
    lam_used = solver._last_lambda
    from pi_mpc import _host

    assert abs(_host.compute_ess(_host.softmax_weights(c_gpu, lam_used)) - N / 10) < 1e-3 * N / 10
    w, _ = orc.softmax_weights(c_gpu, lam_used)
    a_or = P.weighted_actions(w, mean, eps)
    if kw:
        a_or = _host.sg_filter_sequence(np.zeros((T - 1, P.dc), np.float32), a_or, _host.savitzky_golay_coeffs(5, 3))
    check_rel("action_seq_vs_oracle_given_costs", a.cpu().numpy(), a_or, TOL)
    check_rel("state_seq_vs_oracle_rollout", s.cpu().numpy()[0], P.rollout_single(x0, a.cpu().numpy()), TOL)


def brent_cost_vector(rng, N, kind):
    """Cost vectors for the LBPS search: the shapes of the shipped models' costs and awkward ones."""
    if kind == 0:    # nav2d-like: distances + collision penalties
        c = rng.uniform(10, 40, N) + 1e4 * rng.integers(0, 30, N) * (rng.random(N) < 0.5)
    elif kind == 1:  # racing-like
        c = rng.uniform(300, 3000, N) + 1e4 * rng.integers(0, 25, N) * (rng.random(N) < 0.4)
    elif kind == 2:  # pendulum / cartpole-like: a smooth, narrow range
        c = rng.gamma(2.0, rng.uniform(0.5, 50.0), N) + rng.uniform(0, 100)
    elif kind == 3:  # a range of e^40
        c = np.exp(rng.uniform(-20, 20, N))
    elif kind == 4:  # mixed signs, any scale
        c = rng.standard_normal(N) * 10.0 ** rng.integers(-3, 6)
    elif kind == 5:  # all equal: the objective has no range term
        c = np.full(N, float(rng.uniform(-5, 5)))
    elif kind == 6:  # few distinct values
        c = rng.integers(0, max(2, N // 50), N).astype(np.float64)
    else:            # one clear winner
        c = rng.uniform(100, 200, N)
        c[int(rng.integers(0, N))] = 1.0
    return np.ascontiguousarray(c, dtype=np.float32)


def brent_both(solver, costs, delta=0.01, lo=0.01, hi=10.0):
    """(host loop's temperature, device search's temperature, probes of either) on the same uploaded cost vector."""
    st = solver._stream()
    c = torch.from_numpy(costs).cuda()
    solver._h.call("mppi_set_costs", c.data_ptr(), 1, st)
    lam_host = C.c_double(0.0)
    solver._h.call("mppi_lbps_lambda", delta, lo, hi, C.byref(lam_host), st)
    solver._h.call("mppi_lbps_brent_device", delta, lo, hi, st)
    lam_dev, used = C.c_double(0.0), C.c_double(0.0)
    solver._h.call("mppi_get_lambda", C.byref(lam_dev), C.byref(used), st)
    assert not solver._h.lib.mppi_search_error(solver._h.h)
    return lam_host.value, lam_dev.value, solver._h.lib.mppi_search_passes(solver._h.h, st)


@pytest.mark.parametrize("N", [1, 63, 256, 257, 1000, 4096, 65536, 65537, 262144, 1048576, 3000001])
def test_device_brent_equals_the_host_loop_to_the_bit(N):
    """LBPS's bounded Brent search as ONE kernel (mppi_lbps_brent_device, the default of lambda_="LBPS") against the same
    search as a host loop over mppi_softmax_stats (mppi_lbps_lambda; csrc/host_search.hpp::fminbound on both sides): the
    temperature must be IDENTICAL, float64 bit for bit — same partial sums in the same order, same double-precision steps —
    for sample counts on every side of the kernel's geometry (one virtual block, ragged tails, 256 virtual blocks with 1,
    4, 16 costs per thread staged in LDS, and beyond the staging limit), for every shape of cost vector, other deltas and
    ranges.  (A short form of scripts/brent_soak.py, whose 4 000 cases are recorded in profiles/r06_visitB_brent_soak.txt.)"""
    _need_gpu()
    rng = np.random.default_rng(N)
    solver, _ = make_solver("pendulum", 5, N, lambda_=1.0)
    solver.forward(torch.tensor([1.0, 0.0]))
    reps = 3 if N <= 65537 else 1
    for kind in range(8):
        for _ in range(reps):
            costs = brent_cost_vector(rng, N, kind)
            lh, ld, probes = brent_both(solver, costs)
            assert lh == ld and 3 <= probes <= 500, (N, kind, lh, ld, probes)
    costs = brent_cost_vector(rng, N, 0)
    for delta, lo, hi in ((0.1, 0.01, 10.0), (0.01, 0.5, 2.0), (0.001, 1e-3, 1e3), (0.5, 5.0, 5.5)):
        lh, ld, probes = brent_both(solver, costs, delta, lo, hi)
        assert lh == ld, (N, delta, lo, hi, lh, ld, probes)


@pytest.mark.parametrize("name", ["pendulum_T15_N256_lbps", "nav2d_T30_N512_lbps", "nav2d_T30_N4096_lbps", "racing_T25_N4096_lbps"])
def test_device_brent_on_the_reference_fixtures(name):
    """The same on the cost vectors of the reference's own LBPS solves (tests/golden/): device search == host loop to the bit,
    and both within the fixture's band of the temperature the reference found."""
    _need_gpu()
    cfg, g = CASES[name], load(name)
    solver, _ = make_solver("pendulum", 5, cfg["N"], lambda_=1.0)
    solver.forward(torch.tensor([1.0, 0.0]))
    for k in range(int(g["K"])):
        lh, ld, probes = brent_both(solver, np.ascontiguousarray(g[f"costs_{k}"], np.float32))
        assert lh == ld, (name, k, lh, ld)
        assert same_lbps_minimum(g[f"costs_{k}"], ld, float(g[f"lambda_{k}"])), (name, k, ld, float(g[f"lambda_{k}"]))
        parity_report.record("device_brent_probes", probes, 500)


def test_device_brent_gives_up_instead_of_hanging():
    """A block of the search that never becomes resident (here: one block too few is launched — test hook
    `search_test_drop_block`) cannot be waited for: every poll runs into the budget (`fused_timeout_us`), the kernel ends, the
    temperature of that solve is NaN (never a stale or partial value), `mppi_search_error` is raised and the NEXT solve of the
    Python class raises a message that names the remedy; after that the solver works again."""
    _need_gpu()
    import time

    from mppi_playground_amd import _capi

    solver, _ = make_solver("nav2d", 20, 65536, lambda_="LBPS")
    x0 = torch.tensor([-9.0, -9.0, 0.785])
    a0, _ = solver.forward(x0)
    lam0 = solver._last_lambda
    assert np.isfinite(lam0) and torch.isfinite(a0).all()
    solver.set_option("fused_timeout_us", 2000)
    solver.set_option("search_test_drop_block", 1)
    t0 = time.perf_counter()
    a1, s1 = solver.forward(x0)
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 1.0  # (gave up after ~2 ms, did not hang)
    assert solver._h.lib.mppi_search_error(solver._h.h) == 1
    assert np.isnan(solver._last_lambda) and torch.isnan(a1).all()
    solver.set_option("search_test_drop_block", 0)
    with pytest.raises(_capi.MppiError, match="brent_host"):
        solver.forward(x0)
    assert not torch.isnan(solver._previous_action_seq).any()  # (the NaN plan is gone: the raise reset the warm start)
    a2, _ = solver.forward(x0)
    assert torch.isfinite(a2).all() and np.isfinite(solver._last_lambda) and not solver._h.lib.mppi_search_error(solver._h.h)


def test_device_brent_in_a_captured_graph_and_back_to_back():
    """The search is one launch with no host wait: 200 searches enqueued back to back on alternating cost vectors (the
    probe tags and the double-buffered cells carry over from launch to launch) return the host loop's temperatures, and a
    whole LBPS solve can be captured into a hipGraph and replayed."""
    _need_gpu()
    N = 65536
    rng = np.random.default_rng(3)
    solver, _ = make_solver("pendulum", 5, N, lambda_=1.0)
    solver.forward(torch.tensor([1.0, 0.0]))
    vecs = [brent_cost_vector(rng, N, kind) for kind in (0, 2, 3)]
    want = [brent_both(solver, v)[0] for v in vecs]
    st = solver._stream()
    dev = [torch.from_numpy(v).cuda() for v in vecs]
    got = torch.empty(200, device="cuda")
    lam_ptr = C.c_void_p(0)
    for i in range(200):
        solver._h.call("mppi_set_costs", dev[i % 3].data_ptr(), 1, st)
        solver._h.call("mppi_lbps_brent_device", 0.01, 0.01, 10.0, st)
        lam, used = C.c_double(0.0), C.c_double(0.0)
        if i % 50 == 49:  # (most of the searches are never waited for individually)
            solver._h.call("mppi_get_lambda", C.byref(lam), C.byref(used), st)
            assert lam.value == want[i % 3], (i, lam.value, want[i % 3])
    torch.cuda.synchronize()
    assert not solver._h.lib.mppi_search_error(solver._h.h)
    lbps, _ = make_solver("nav2d", 30, 32768, lambda_="LBPS")
    twin, _ = make_solver("nav2d", 30, 32768, lambda_="LBPS")
    x0 = torch.tensor([-9.0, -9.0, 0.785], device="cuda")
    for _ in range(3):
        lbps.forward(x0)
        twin.forward(x0)
    a_t, s_t = twin.forward(x0)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            a, s = lbps.forward(x0)
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(a, a_t) and torch.equal(torch.as_tensor(s), torch.as_tensor(s_t))
    assert lbps._last_lambda == twin._last_lambda and 0.01 <= twin._last_lambda <= 10.0


@pytest.mark.parametrize("lam_mode", ["ESSPS", "LBPS", "MPO"])
def test_device_softmax_stats_drive_the_same_temperature(lam_mode):
    """auto_lambda_stats='device' (sums on the GPU) and 'host' (costs copied to numpy) find the same lambda
    and the same action; the raw statistics match a float64 numpy evaluation."""
    N, T = 65536, 50
    outs = []
    for mode in ("device", "host"):
        solver, _ = make_solver("nav2d", T, N, lambda_=lam_mode, auto_lambda_stats=mode)
        x0 = torch.tensor([-9.0, -9.0, 0.785])
        a, s = solver.forward(x0)
        a2, _ = solver.forward(x0)
        outs.append((solver._last_lambda, a.cpu().numpy(), a2.cpu().numpy(), solver))
    tol = {"ESSPS": 1e-4, "LBPS": 5e-3, "MPO": 1e-3}[lam_mode]
    assert abs(outs[0][0] - outs[1][0]) <= tol * outs[1][0]
    if lam_mode == "ESSPS":  # grid bracketing and one-lambda-at-a-time brentq find the same root
        sb, _ = make_solver("nav2d", T, N, lambda_="ESSPS", essps_search="brentq")
        sb.forward(torch.tensor([-9.0, -9.0, 0.785]))
        sg, _ = make_solver("nav2d", T, N, lambda_="ESSPS", essps_search="grid")
        sg.forward(torch.tensor([-9.0, -9.0, 0.785]))
        assert abs(sb._last_lambda - sg._last_lambda) <= 1e-6 * sb._last_lambda
        # the default: the same search with its scalar steps on the device too (no read-back during the solve) — same
        # arithmetic (csrc/host_search.hpp on both sides), so the same temperature and the same action
        sd, _ = make_solver("nav2d", T, N, lambda_="ESSPS", essps_search="device")
        ad, _ = sd.forward(torch.tensor([-9.0, -9.0, 0.785]))
        assert sd._lambda_pending and outs[0][3]._essps_search == "device"
        assert abs(sd._last_lambda - sg._last_lambda) <= 1e-12 * sg._last_lambda and not sd._lambda_pending
        assert rel_err(ad.cpu().numpy(), outs[0][1]) == 0.0
        # end-point rules decided on the device (mppi.py:361-364): racing costs never reach ESS = N/10 below lambda_max
        sr, cr = make_solver("racing", 25, 4096, lambda_="ESSPS")
        env = _envs["racing"]
        ref, _ = cr.calc_ref_trajectory(env.reset(), env.racing_center_path, 0, 25, DL=0.1, lookahead_distance=3,
                                        reference_path_interval=0.85)
        cr.set_reference(ref)
        sr.forward(env.reset().clone())
        assert sr._last_lambda == 10.0
        st, _ = make_solver("pendulum", 15, 256, lambda_="ESSPS", essps_target_ess=1.0)
        st.forward(torch.tensor([3.0, 0.0]))
        assert st._last_lambda == 0.01
        # the library's own search (mppi_essps_lambda, what forward() used) == the host statement of it
        from pi_mpc import _host
        lam_py = _host.essps_lambda_grid(sg._ess_grid, sg._essps_target_ess, sg._lambda_min, sg._lambda_max)
        assert abs(lam_py - sg._last_lambda) <= 1e-9 * lam_py
        ess = sg._ess_grid([sg._last_lambda])[0]
        assert abs(ess - N / 10) <= 1e-4 * N / 10
    assert rel_err(outs[0][1], outs[1][1]) < 20 * tol
    x0 = torch.tensor([-9.0, -9.0, 0.785])
    if lam_mode == "LBPS":
        # the default is the reference's own algorithm — scipy's bounded Brent — as ONE kernel (round 6; no host wait);
        # lbps_search="brent_host" runs the same search as a host loop with one read-back per probe (round 5's default):
        # the same temperature and the same action TO THE BIT; lbps_search="grid" (two 32-temperature grids + a quartic)
        # lands on the same minimum
        sb = outs[0][3]
        assert sb._rule_on_device == "LBPS" and sb._lbps_search == "brent" and sb._one_call
        c = sb._costs.cpu().numpy()
        sh, _ = make_solver("nav2d", T, N, lambda_="LBPS", lbps_search="brent_host")
        assert sh._rule_on_device is None and not sh._one_call
        sh.forward(x0)
        ah, _ = sh.forward(x0)
        assert sh._last_lambda == sb._last_lambda and torch.equal(ah, torch.from_numpy(outs[0][2]).cuda())
        probes = sb._h.lib.mppi_search_passes(sb._h.h, None)
        assert 10 <= probes <= 60, probes
        sd, _ = make_solver("nav2d", T, N, lambda_="LBPS", lbps_search="grid")
        assert sd._rule_on_device == "LBPS" and sd._one_call
        sd.forward(x0)
        ad, _ = sd.forward(x0)
        assert sd._lambda_pending
        assert same_lbps_minimum(c, sd._last_lambda, sb._last_lambda) and not sd._lambda_pending
        assert sd._lambda == sd._last_lambda
    if lam_mode == "MPO":
        # the dual and its Adam moments live on the device; the step-by-step entry points (what an injected-noise solve
        # takes) and the one-call path agree bit for bit, and `_lambda` / `_last_lambda` follow the reference's
        # bookkeeping: the weights of solve k use the temperature the dual had BEFORE its k-th step (mppi.py:387-398)
        sa, _ = make_solver("nav2d", T, N, lambda_="MPO")
        sb, _ = make_solver("nav2d", T, N, lambda_="MPO")
        sb._one_call = False
        lams = []
        for k in range(4):
            a1, _ = sa.forward(x0)
            a2, _ = sb.forward(x0)
            assert torch.equal(a1, a2)
            assert sa._last_lambda == sb._last_lambda == (1.0 if k == 0 else lams[-1]) and sa._lambda == sb._lambda
            lams.append(sa._lambda)
        assert len(set(lams)) == 4
        st4 = (C.c_double * 4)()
        sa._h.call("mppi_mpo_state", st4)
        assert int(st4[3]) == 4 and abs(np.exp(np.float32(st4[0])) - lams[-1]) <= 1e-6 * lams[-1]
        # a temperature assigned by the caller is used by the next solve's weights; the dual steps on regardless
        sa._lambda = 2.5
        sa.forward(x0)
        assert sa._last_lambda == 2.5 and sa._lambda != 2.5
        sa._h.call("mppi_mpo_state", st4)
        assert int(st4[3]) == 5
    solver = outs[0][3]
    c = solver._costs.cpu().numpy().astype(np.float64)
    st = solver._softmax_stats(3.0)
    e = np.exp(-(c - c.min()) / 3.0)
    assert st["cmin"] == c.min() and st["cmax"] == c.max()
    assert abs(st["se"] - e.sum()) <= 1e-5 * e.sum() and abs(st["se2"] - (e * e).sum()) <= 1e-5 * (e * e).sum()
    assert abs(st["sec"] - (e * c).sum()) <= 1e-5 * (e * c).sum()


# ------------------------------------------------------------------------------ generic (opaque callables) path
def _untagged(fn):
    """A plain closure around a plugin: what the reference examples pass (no native tag)."""
    return lambda *a: fn(*a)


@pytest.mark.parametrize("name", ["pendulum_T15_N256_fixed", "mountaincar_T100_N256_fixed", "cartpole_T10_N100_fixed",
                                  "pendulum_T15_N200_explore"])
def test_generic_callable_path_matches_reference(name):
    """Opaque torch callables (the reference's plugin surface as-is): library sampling/softmax/
    reduction around the user's T-step loops.  Must agree with the reference fixture and with the
    fused native kernels (mountaincar: including the in-place mutation quirk)."""
    _need_gpu()
    from envs import classic_control as cc
    from pi_mpc.mppi import MPPI

    cfg, g = CASES[name], load(name)
    model, T, N = cfg["model"], cfg["T"], cfg["N"]
    mc = MODEL_CFG[model]
    ds, dc = orc.MODEL_DIMS[orc.MODEL_IDS[model]]
    kw = {k: cfg[k] for k in ("exploration",) if k in cfg}
    gen = MPPI(horizon=T, num_samples=N, dim_state=ds, dim_control=dc,
               dynamics=_untagged(getattr(cc, f"{model}_dynamics")), cost_func=_untagged(getattr(cc, f"{model}_cost")),
               u_min=torch.tensor(mc["u_min"]), u_max=torch.tensor(mc["u_max"]), sigmas=torch.tensor(mc["sigmas"]),
               lambda_=cfg["lambda_"], **kw)
    assert gen._model is None
    nat, _ = make_solver(model, T, N, lambda_=cfg["lambda_"], **kw)
    for k in range(int(g["K"])):
        outs = []
        for sol in (gen, nat):
            sol.set_warm_start(g[f"mean_in_{k}"])
            sol.inject_noise(torch.from_numpy(g[f"eps_{k}"]))
            a, s = sol.forward(torch.from_numpy(g[f"x0_{k}"]))
            outs.append((a.cpu().numpy(), s.cpu().numpy(), sol._costs.cpu().numpy()))
        (ag, sg, cg), (an, sn, cn) = outs
        assert rel_err(cg, g[f"costs_{k}"]) < TOL and rel_err(cg, cn) < TOL
        cond = 8 * EPS32 * float(np.abs(cg).max()) / float(cfg["lambda_"])
        assert rel_err(ag, g[f"action_seq_{k}"]) < max(TOL, cond) and rel_err(ag, an) < max(TOL, cond)
        assert rel_err(sg, g[f"state_seq_{k}"]) < max(TOL, cond) and rel_err(sg, sn) < max(TOL, cond)
        if f"S_{k}" in g.files:
            assert rel_err(gen._state_seq_batch_buf.cpu().numpy(), g[f"S_{k}"]) < TOL
            assert np.array_equal(gen._perturbed_action_seqs.cpu().numpy(), g[f"U_{k}"])
    ts, tw = gen.get_top_samples(8)
    assert ts.shape == (8, T + 1, ds)


def test_recognised_closures_run_the_fused_model():
    """Untagged callables whose source fingerprint is listed for a model AND that agree with that model's shipped plugin on the
    probe batches (pi_mpc/recognize.py: how the reference examples' own closures are recognised; the real ones are checked in the
    build container, tests/test_host_logic.py) run as the fused native model: same bits as the tagged plugins, and the generic
    path's answer to 1e-5.  A listed fingerprint whose callable computes something else stays on the generic path."""
    _need_gpu()
    from envs import classic_control as cc
    from pi_mpc import recognize
    from pi_mpc.mppi import MPPI

    def dynamics(state: torch.Tensor, action: torch.Tensor) -> torch.Tensor:  # (a closure, no native tag)
        th, thdot = state[:, 0:1], state[:, 1:2]
        u = torch.clamp(action[:, 0:1], -2, 2)
        newthdot = thdot + (-3 * 10.0 / (2 * 1.0) * torch.sin(th + torch.pi) + 3.0 / (1.0 * 1.0 ** 2) * u) * 0.05
        return torch.cat((th + newthdot * 0.05, torch.clamp(newthdot, -8, 8)), dim=1)

    def cost(state, action, info):
        return (torch.remainder(state[:, 0] + torch.pi, 2 * torch.pi) - torch.pi) ** 2 + 0.1 * state[:, 1] ** 2

    def other_cost(state, action, info):
        return state[:, 0] ** 2

    kw = dict(horizon=15, num_samples=4096, dim_state=2, dim_control=1, u_min=torch.tensor([-2.0]), u_max=torch.tensor([2.0]),
              sigmas=torch.tensor([1.0]), lambda_=1.0)
    table = recognize._table()
    try:
        recognize._table_cache = {"pendulum": {"dynamics": [recognize.fingerprint(dynamics)],
                                               "cost": [recognize.fingerprint(cost), recognize.fingerprint(other_cost)]}}
        rec = MPPI(dynamics=dynamics, cost_func=cost, **kw)
        wrong = MPPI(dynamics=dynamics, cost_func=other_cost, **kw)   # listed fingerprint, other values: not recognised
        off = MPPI(dynamics=dynamics, cost_func=cost, recognize_closures=False, **kw)
    finally:
        recognize._table_cache = table
    tagged = MPPI(dynamics=cc.pendulum_dynamics, cost_func=cc.pendulum_cost, **kw)
    assert rec._model == "pendulum" and rec._recognized == (dynamics, cost) and tagged._model == "pendulum"
    assert wrong._model is None and off._model is None
    x = torch.tensor([3.0, 0.5])
    for _ in range(3):
        a, s = rec.forward(x)
        b, sb = tagged.forward(x)
        g, sg = off.forward(x)
        assert torch.equal(a, b) and torch.equal(s, sb)
        assert rel_err(g.cpu().numpy(), a.cpu().numpy()) <= 1e-5 and rel_err(sg.cpu().numpy(), s.cpu().numpy()) <= 1e-5
        x = sb[0, 1].clone()


def test_shipped_fingerprints_recognise_a_transcribed_pendulum_closure_on_this_torch():
    """The SHIPPED closure_fingerprints.json against tests/closure_transcription.py (the pendulum example's closures written
    anew: a TorchScript dynamics nested in a function, a plain cost over a module-level TorchScript angle wrap) scripted by
    THIS machine's torch: recognised through the version-independent fingerprints + the probe batches, so the solver runs the
    fused model — same bits as the tagged plugin — and says how it decided (`_recognition`)."""
    _need_gpu()
    import warnings

    import closure_transcription as ct
    from envs import classic_control as cc
    from pi_mpc import recognize
    from pi_mpc.mppi import MPPI

    step, cost = ct.build()
    assert isinstance(step, torch.jit.ScriptFunction) and resolve_tag(step) is None and resolve_tag(cost) is None
    kw = dict(horizon=20, num_samples=4096, dim_state=2, dim_control=1, u_min=torch.tensor([-2.0]), u_max=torch.tensor([2.0]),
              sigmas=torch.tensor([1.0]), lambda_="ESSPS")
    with warnings.catch_warnings():
        warnings.simplefilter("error", recognize.RecognitionWarning)
        rec = MPPI(dynamics=step, cost_func=cost, **kw)
    assert rec._model == "pendulum" and rec._recognized == (step, cost)
    assert rec._recognition["fingerprint"] and rec._recognition["behaviour"] and rec._recognition["torch"] == torch.__version__
    tagged = MPPI(dynamics=cc.pendulum_dynamics, cost_func=cc.pendulum_cost, **kw)
    x = torch.tensor([3.0, 0.5])
    for _ in range(3):
        a, s = rec.forward(x)
        b, sb = tagged.forward(x)
        assert torch.equal(a, b) and torch.equal(s, sb)
        x = sb[0, 1].clone()


def resolve_tag(fn):
    from pi_mpc.native import resolve

    return resolve(fn)


@pytest.mark.parametrize("dc", [4, 3, 6, 7])
def test_generic_path_any_control_dimension(dc):
    """dim_control = 3, 4, 6, 7 (the reference accepts any, mppi.py:96-98), dim_state = 5, on a toy linear model,
    checked against plain torch on the exported noise: the control index of a flat column depends on the float4 group
    once dim_control is not 1, 2 or 4 (per-column sigma / bounds table), and more than four controls do not fit the
    config struct (mppi_set_control_limits)."""
    _need_gpu()
    from pi_mpc.mppi import MPPI

    T, N, ds = 11, 500, 5
    B = torch.arange(ds * dc, dtype=torch.float32).reshape(ds, dc).cuda() / (2.5 * dc)

    def dyn(s, u):
        return s + 0.1 * (u @ B.T)

    def cost(s, u, info):
        return (s ** 2).sum(dim=1) + 0.01 * (u ** 2).sum(dim=1)

    sig = torch.tensor([0.5, 1.0, 0.2, 0.7, 0.9, 0.3, 0.6][:dc])
    u_min = torch.tensor([-1.0, -2.0, -0.3, -1.5, -0.8, -0.2, -1.1][:dc])
    u_max = torch.tensor([1.0, 2.0, 0.3, 0.5, 0.4, 0.9, 1.3][:dc])
    sol = MPPI(horizon=T, num_samples=N, dim_state=ds, dim_control=dc, dynamics=dyn, cost_func=cost,
               u_min=u_min, u_max=u_max, sigmas=sig, lambda_=0.7, exploration=0.1)
    x0 = torch.ones(ds)
    mean = None
    for tick in range(2):  # second solve: warm start carried (mean != 0 below the exploration split)
        mean_in = sol._previous_action_seq.clone()
        a, s = sol.forward(x0)
        assert a.shape == (T, dc) and s.shape == (1, T + 1, ds)
        eps = sol._action_noises
        for k in range(dc):
            assert abs(float(eps[..., k].std()) - float(sig[k])) < 0.06 * float(sig[k])
        ref = orc.philox_normal(42, 1 + tick, 0, N, T, dc, sig.numpy())
        assert np.abs(eps.cpu().numpy() - ref).max() < 1e-4
        thr = int(N * 0.9)
        U = eps.clone()
        U[:thr] += mean_in
        U = torch.maximum(torch.minimum(U, sol._u_max), sol._u_min)
        assert torch.equal(sol._perturbed_action_seqs, U)
        S = torch.zeros(N, T + 1, ds, device="cuda")
        S[:, 0] = x0.cuda()
        c = torch.zeros(N, device="cuda")
        for t in range(T):
            S[:, t + 1] = dyn(S[:, t], U[:, t])
            c += cost(S[:, t], U[:, t], None)
        c += cost(S[:, T], torch.zeros(N, dc, device="cuda"), None)
        w = torch.softmax(-c.double() / 0.7, dim=0)
        a_ref = (w.view(N, 1, 1) * U.double()).sum(0)
        assert rel_err(a.cpu().numpy(), a_ref.cpu().numpy()) < 1e-5
        assert rel_err(sol._weights.cpu().numpy(), w.cpu().numpy()) < 1e-5
    ps, pst = sol.get_samples_from_posterior(a, x0, 8)  # generic path: library sampling, user dynamics
    assert ps.shape == (8, T, dc) and pst.shape == (8, T + 1, ds)
    want = orc.philox_normal(42, 3, 0, 8, T, dc, sig.numpy()) + a.cpu().numpy()[None]
    assert np.abs(ps.cpu().numpy() - want).max() < 1e-4
    ts, tw = sol.get_top_samples(5)
    assert ts.shape == (5, T + 1, ds) and bool((tw[:-1] >= tw[1:]).all())
    with pytest.raises(ValueError):
        MPPI(horizon=T, num_samples=N, dim_state=ds, dim_control=65, dynamics=dyn, cost_func=cost,
             u_min=torch.zeros(65), u_max=torch.ones(65), sigmas=torch.ones(65), lambda_=0.7)


def test_wide_control_rows_through_the_c_abi():
    """dim_control = 6 through the raw C ABI: sampling is refused until mppi_set_control_limits hands over all six
    bounds (the config struct holds four); afterwards noise scale, clamp and the weighted reduction use the table.
    Long rows (T*dc = 600: 32 float4 groups per wave, two column chunks) against numpy."""
    _need_gpu()
    from mppi_playground_amd import _capi

    T, N, dc, ds = 100, 300, 6, 3
    f4 = C.c_float * 4
    cfg = _capi.MppiConfig(model=_capi.MODEL_GENERIC, horizon=T, dim_state=ds, dim_control=dc, num_samples=N,
                           sample_offset=0, inherit_count=N, u_min=f4(-1, -1, -1, -1), u_max=f4(1, 1, 1, 1),
                           sigmas=f4(1, 1, 1, 1), seed=7, device=0)
    h = _capi.Handle(cfg)
    with pytest.raises(_capi.MppiError):
        h.call("mppi_sample", 1, None)
    lo = np.array([-1.0, -0.5, -2.0, -0.1, -3.0, -0.7], np.float32)
    hi = np.array([0.5, 0.6, 1.0, 0.2, 3.0, 0.1], np.float32)
    sg = np.array([0.5, 1.0, 2.0, 0.1, 1.5, 0.3], np.float32)
    with pytest.raises(_capi.MppiError):
        h.call("mppi_set_control_limits", lo.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p),
               sg.ctypes.data_as(C.c_void_p), 4)
    h.call("mppi_set_control_limits", lo.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p),
           sg.ctypes.data_as(C.c_void_p), dc)
    rng = np.random.default_rng(2)
    mean = (rng.standard_normal((T, dc)) * 0.3).astype(np.float32)
    md = torch.from_numpy(mean).cuda()
    h.call("mppi_set_mean", md.data_ptr(), 1, None)
    h.call("mppi_sample", 1, None)
    eps = torch.empty(N, T, dc, device="cuda")
    U = torch.empty(N, T, dc, device="cuda")
    h.call("mppi_export_noise", eps.data_ptr(), U.data_ptr(), None)
    e = eps.cpu().numpy()
    assert np.abs(e - orc.philox_normal(7, 1, 0, N, T, dc, sg)).max() < 1e-4 * sg.max()
    assert np.array_equal(U.cpu().numpy(), np.clip(mean[None] + e, lo, hi))
    costs = (rng.random(N) * 5).astype(np.float32)
    cd = torch.from_numpy(costs).cuda()
    h.call("mppi_set_costs", cd.data_ptr(), 1, None)
    h.call("mppi_weights_reduce", 0.8, None, None)
    a = torch.empty(T, dc, device="cuda")
    h.call("mppi_finalize", None, 1, 0.8, 1, a.data_ptr(), None, None, None)
    w = np.exp(-(costs.astype(np.float64) - costs.min()) / 0.8)
    w /= w.sum()
    a_ref = (w[:, None, None] * U.cpu().numpy().astype(np.float64)).sum(0)
    assert rel_err(a.cpu().numpy(), a_ref) < 1e-5
    h.close()


# ------------------------------------------------------------------------------ API / error behaviour
def test_error_behaviour_matches_reference():
    _need_gpu()
    from envs import classic_control as cc
    from pi_mpc.mppi import MPPI

    base = dict(horizon=15, num_samples=100, dim_state=2, dim_control=1, dynamics=cc.pendulum_dynamics,
                cost_func=cc.pendulum_cost, u_min=torch.tensor([-2.0]), u_max=torch.tensor([2.0]),
                sigmas=torch.tensor([1.0]))
    with pytest.raises(ValueError):
        MPPI(lambda_=1, **base)  # int is not accepted (mppi.py:205-210)
    with pytest.raises(ValueError):
        MPPI(lambda_="nope", **base)
    with pytest.raises(ValueError):
        MPPI(lambda_=1.0, sg_window_size=4, **base)
    with pytest.raises(AssertionError):
        MPPI(lambda_=1.0, **{**base, "u_min": torch.tensor([-2.0, 0.0])})
    with pytest.raises(ValueError):  # no CPU path: a CPU device is refused, not silently run on the GPU
        MPPI(lambda_=1.0, device=torch.device("cpu"), **base)
    assert MPPI(lambda_=1.0, device="cuda:0", **base)._device == torch.device("cuda", 0)
    s = MPPI(lambda_=1.0, **base)
    with pytest.raises(AssertionError):
        s.forward(torch.zeros(3))
    a, st = s(np.array([np.pi, 0.0]))  # nn.Module.__call__, float64 ndarray state (example/pendulum.py:73-77)
    assert a.shape == (15, 1) and st.shape == (1, 16, 2)
    s.reset()
    assert float(s._previous_action_seq.abs().max()) == 0.0


def test_racing_controller_closed_loop_smoke():
    """The reference's control loop (example/racing.py:221-266) minus rendering, default sizes."""
    _need_gpu()
    from envs.racing_controller import racing_controller
    from envs.racing_env import RacingEnv

    env = RacingEnv()
    ctrl = racing_controller(env)
    with pytest.raises(ValueError):
        ctrl.update(env.reset(), env.racing_center_path)  # maps not set (example/racing.py:83-90)
    ctrl.set_cost_map(env._obstacle_map, env._lane_map)
    state = env.reset()
    for _ in range(20):
        action_seq, state_seq = ctrl.update(state, env.racing_center_path)
        state, done = env.step(action_seq[0, :])
        coll = env.collision_check(state=state_seq)
        top, tw = ctrl.get_top_samples(num_samples=300)
        assert top.shape == (300, 26, 4) and coll.shape[-1] == 26
    assert float(state[3]) > 0.5  # the car accelerates along the track


# ------------------------------------------------------------------------------ map construction on the device
def test_device_built_maps_are_bit_exact_with_reference_fixtures():
    """raster_obstacles_kernel / lane_map_kernel (through mppi_build_*_map, the path the solver takes when a map
    carries a recipe) against the reference's own 800x800 racing maps and 200x200 nav2d map."""
    from helpers import nav2d_env_fixture, racing_env_fixture

    solver, ctrl = make_solver("racing", 10, 64)
    ctrl.set_reference(np.zeros((11, 4), np.float32))
    solver.forward(torch.zeros(4))
    e = racing_env_fixture()
    assert np.array_equal(solver.device_map(0), e["obst"])
    assert np.array_equal(solver.device_map(1), e["lane"])
    solver, _ = make_solver("nav2d", 10, 64)
    solver.forward(torch.tensor([-9.0, -9.0, 0.785]))
    assert np.array_equal(solver.device_map(0), nav2d_env_fixture()["map"])


def test_device_map_builders_clip_like_the_reference_loops():
    """Obstacles sticking out of the grid (clipped onto border cells), degenerate radii, a centre line that
    leaves the grid — against the literal restatement of the reference loops; and upload == build."""
    _need_gpu()
    from envs.lane_map_2d import LaneMap
    from envs.obstacle_map_2d import ObstacleMap
    from mppi_playground_amd import _capi

    rng = np.random.default_rng(5)
    f4 = C.c_float * 4
    cfg = _capi.MppiConfig(model=_capi.MODEL_IDS["racing"], horizon=4, dim_state=4, dim_control=2, num_samples=64,
                           sample_offset=0, inherit_count=64, u_min=f4(-1, -1, 0, 0), u_max=f4(1, 1, 0, 0),
                           sigmas=f4(1, 1, 0, 0), seed=1, device=0)
    h = _capi.Handle(cfg)
    for trial, (metres, cell) in enumerate([(2, 0.05), (2, 0.04), (4, 0.1), (2, 0.025), (4, 0.05), (6, 0.1)]):
        m = ObstacleMap(map_size=(metres, metres), cell_size=cell, device="cpu")
        nx, ny = m._map.shape
        size = metres / 2
        circles = [(rng.uniform(-1.3 * size, 1.3 * size, 2), rng.uniform(0.2 * cell, 0.3 * size)) for _ in range(6)]
        rects = [(rng.uniform(-1.3 * size, 1.3 * size, 2), rng.uniform(cell, 0.5 * size), rng.uniform(cell, 0.5 * size))
                 for _ in range(5)]
        for c, r in circles:
            m.add_circle_obstacle(c, r)
        for c, w, hh in rects:
            m.add_rectangle_obstacle(c, w, hh)
        lit, _ = orc.obstacle_map_literal(nx, ny, cell, circles, rects)
        assert 0 < lit.sum() < lit.size
        rec = m.grid_spec().recipe
        ci, re_ = np.ascontiguousarray(rec["circles"]), np.ascontiguousarray(rec["rects"])
        h.call("mppi_build_obstacle_map", 0, nx, ny, cell, float(nx // 2), float(ny // 2),
               ci.ctypes.data_as(C.c_void_p), len(ci), re_.ctypes.data_as(C.c_void_p), len(re_), None)
        out = np.empty((nx, ny), np.uint8)
        h.call("mppi_download_map", 0, out.ctypes.data_as(C.c_void_p), None, None)
        assert np.array_equal(out, lit), f"trial {trial}"

        t = np.linspace(0, 2 * np.pi, 50)
        lane = np.stack([1.1 * size * np.cos(t), 0.7 * size * np.sin(t) + 0.4 * size, t], axis=1)  # leaves the grid
        width = float(rng.uniform(2 * cell, 0.5 * size))
        lit, _ = orc.lane_map_literal(nx, ny, cell, lane, width)
        lm = LaneMap(lane, lane_width=width, map_size=(metres, metres), cell_size=cell, device="cpu")
        rec = lm.grid_spec().recipe
        assert lm.grid_spec().cells.shape == (nx, ny)
        seeds = np.ascontiguousarray(rec["seeds"])
        h.call("mppi_build_lane_map", 1, nx, ny, cell, float(nx // 2), float(ny // 2),
               seeds.ctypes.data_as(C.c_void_p), len(seeds), int(rec["max_d2"]), None)
        h.call("mppi_download_map", 1, out.ctypes.data_as(C.c_void_p), None, None)
        assert np.array_equal(out, lit), f"lane trial {trial}"

    # byte upload and device build give the same grid; bad arguments are refused
    h.call("mppi_upload_map", 1, lit.ctypes.data_as(C.c_void_p), nx, ny, 0.05, float(nx // 2), float(ny // 2))
    h.call("mppi_download_map", 1, out.ctypes.data_as(C.c_void_p), None, None)
    assert np.array_equal(out, lit)
    with pytest.raises(_capi.MppiError):
        h.call("mppi_build_lane_map", 1, nx, ny, 0.05, 0.0, 0.0, None, 0, 4, None)
    with pytest.raises(_capi.MppiError):
        h.call("mppi_build_obstacle_map", 2, nx, ny, 0.05, 0.0, 0.0, None, 0, None, 0, None)
    bad = np.array([[1, 1, -2]], np.int32)
    with pytest.raises(_capi.MppiError):
        h.call("mppi_build_obstacle_map", 0, nx, ny, 0.05, 0.0, 0.0, bad.ctypes.data_as(C.c_void_p), 1, None, 0, None)
    h.close()


def test_wave_parallel_batch1_rollout_is_bit_identical_to_the_serial_one():
    """finalize_kernel<racing, fast> spreads the batch-1 rollout over a wave (Model::rollout_wave); the states must
    equal, bit for bit, the lane-serial rollout of the same actions (mppi_rollout_actions), along a closed loop
    that visits many headings / speeds and with actions outside the bounds."""
    solver, ctrl = make_solver("racing", 50, 2048, lambda_=20.0)
    env = _envs["racing"]
    state = env.reset()
    ctrl.current_path_index = 0
    for tick in range(40):
        a, s = ctrl.update(state, env.racing_center_path)
        s2 = torch.empty_like(s)
        solver._h.call("mppi_rollout_actions", a.data_ptr(), 1, None, s2.data_ptr(), solver._stream())
        assert torch.equal(s, s2), f"tick {tick}"
        state, _ = env.step(a[0, :])
    # actions beyond the bounds and a heading just inside the wrap
    solver2, ctrl2 = make_solver("racing", 50, 256, lambda_=1e6)
    ctrl2.set_reference(np.zeros((51, 4), np.float32))
    x0 = torch.tensor([39.0, -39.5, 3.1415, 7.9])
    solver2.set_warm_start(np.tile(np.array([[3.0, -0.3]], np.float32), (50, 1)))
    a, s = solver2.forward(x0)
    s2 = torch.empty_like(s)
    solver2._h.call("mppi_rollout_actions", a.data_ptr(), 1, None, s2.data_ptr(), solver2._stream())
    assert torch.equal(s, s2)
    P = oracle_problem("racing", 1, 50)
    ref = P.rollout_single(x0.numpy(), a.cpu().numpy())
    assert rel_err(s.cpu().numpy()[0], ref) < TOL


@pytest.mark.parametrize("model,T,N,kw", [("racing", 50, 1 << 17, {}), ("nav2d", 50, 65536, dict(lambda_="ESSPS")),
                                          ("cartpole", 64, 32768, dict(lambda_="ESSPS", use_sg_filter=True)),
                                          ("pendulum", 15, 20000, dict(lambda_=2.0))])
def test_lazy_state_seq_same_bits_read_early_late_or_never(model, T, N, kw):
    """Opt-in (`lazy_state_seq=True`; the default returns a completed plain tensor like the reference): the batch-1 rollout of
    the solution (mppi.py:448-449) leaves the solve's last kernel: it rides in one extra block of the NEXT solve's rollout
    launch, or is launched on the spot when the returned `state_seq` is used first.  Read at once, after later solves, or
    never: the same bits as a solver that rolls out inside finalize_kernel (the default), the same actions, and nothing else
    changes.  Until it is completed the buffer holds NaN (a reader that bypasses the join cannot mistake it for states), and
    a completion requested from ANOTHER stream is ordered behind the solve's stream."""
    from pi_mpc.mppi import _DeferredStateSeq

    kw = dict(kw)
    lam = kw.pop("lambda_", 1.0)
    a_s, a_ctrl = make_solver(model, T, N, lambda_=lam, lazy_state_seq=True, **kw)
    b_s, b_ctrl = make_solver(model, T, N, lambda_=lam, **kw)
    assert a_s._lazy_state and not b_s._lazy_state
    a_s.set_option("timing", 1)
    if model == "racing":
        env = _envs["racing"]
        x0 = env._robot_state.clone()
        ref, _ = a_ctrl.calc_ref_trajectory(x0, env.racing_center_path, 0, T, DL=0.1, lookahead_distance=3,
                                            reference_path_interval=0.85)
        a_ctrl.set_reference(ref)
        b_ctrl.set_reference(ref)
    elif model == "nav2d":
        x0 = _envs["nav2d"].reset().clone()
    elif model == "cartpole":
        x0 = torch.tensor([0.01, 0.0, 0.02, 0.0], device="cuda")
    else:
        x0 = torch.tensor([3.0, 0.5], device="cuda")
    kept, want = [], []
    for k in range(12):
        a, s = a_s.forward(x0)
        b, sb = b_s.forward(x0)
        assert type(s) is _DeferredStateSeq and type(sb) is torch.Tensor
        assert torch.equal(a, b)
        if k % 3 == 0:  # read at once: the first use launches the stand-alone rollout
            assert torch.equal(s, sb), k
        elif k % 3 == 1:  # read after later solves: the next solve's rollout launch completed it
            kept.append(s)
            want.append(sb)
        # k % 3 == 2: never read (completed by the next rollout launch all the same; its tensor is dropped)
        x0 = sb[0, 1].clone()
    for s, sb in zip(kept, want):
        assert s.__dict__["_mppi_join"] is not None  # untouched so far
        assert torch.equal(s, sb)
    a, s = a_s.forward(x0)
    b, sb = b_s.forward(x0)
    assert torch.isnan(a_s._state_out.clone()).all()  # a reader of the plain buffer (bypassing the join) sees NaN, never stale states
    a_s.join_state_seq()  # (what a reader outside torch calls before it uses the raw pointer)
    assert torch.equal(s, sb)
    st = a_s.stage_times_ms()
    # stand-alone launches: the four reads at once + the explicit join; every other state sequence rode in a rollout launch
    assert st["state_seq_standalone_launches"] == 5.0, st
    # a consumer on ANOTHER stream: its join waits for the solve's stream (finalize's write of the rollout's inputs)
    side = torch.cuda.Stream()
    for _ in range(20):
        a, s = a_s.forward(x0)
        b, sb = b_s.forward(x0)
        with torch.cuda.stream(side):
            got = s.clone()
        side.synchronize()
        assert torch.equal(got, sb)


def test_info_dict_and_log_temperature_like_the_reference():
    """Side effects the reference's forward() has besides its return values (src/pi_mpc/mppi.py:299-306,318-322,194-199):
    the CALLER's `info` dict ends with t = T-1, initial_state = S[:, 0], prev_state = S[:, T-1], prev_action = U[:, T-2]
    (here: built on first use from the solve's noise — checked against the materialised buffers), and with lambda_="MPO"
    the solver carries `log_temperature` as a registered nn.Parameter (here: a view of the dual on the device)."""
    from pi_mpc.mppi import _LazyInfoTensor

    solver, _ = make_solver("nav2d", 20, 512, lambda_=5.0)
    x0 = torch.tensor([-9.0, -9.0, 0.785], device="cuda")
    info = {"mine": 1}
    a, s = solver.forward(x0, info)
    assert info["t"] == 19 and info["mine"] == 1
    assert type(info["prev_state"]) is _LazyInfoTensor and info["prev_state"].__dict__["_mppi_value"] is None  # not built yet
    S, U = solver._state_seq_batch, solver._perturbed_action_seqs
    assert torch.equal(info["initial_state"], S[:, 0, :]) and info["initial_state"].shape == (512, 3)
    assert torch.equal(info["prev_state"] + 0.0, S[:, 19, :])
    assert torch.equal(torch.stack([info["prev_action"], info["prev_action"]])[1], U[:, 18, :])
    assert info["prev_action"].shape == (512, 2)
    solver.forward(x0)  # the default dict: nothing to leave anywhere
    # opaque callables fill the dict in their own loop, like the reference
    from envs import classic_control as cc
    from pi_mpc.mppi import MPPI

    g = MPPI(15, 64, 2, 1, lambda s_, a_: cc.pendulum_dynamics(s_, a_), lambda s_, a_, i_: cc.pendulum_cost(s_, a_, i_),
             torch.tensor([-2.0]), torch.tensor([2.0]), torch.tensor([1.0]), 1.0)
    gi = {}
    g.forward(torch.tensor([3.0, 0.0]), gi)
    assert gi["t"] == 14 and gi["prev_state"].shape == (64, 2) and gi["prev_action"].shape == (64, 1)
    # MPO: the dual as a Parameter
    m, _ = make_solver("nav2d", 20, 512, lambda_="MPO")
    assert isinstance(m.log_temperature, torch.nn.Parameter) and "log_temperature" in dict(m.named_parameters())
    assert float(m.log_temperature) == 0.0  # log(1.0), mppi.py:194-199
    m.forward(x0)
    lam = float(m._lambda)  # (fetching the temperature waits for the solve's stream)
    assert abs(float(torch.exp(m.log_temperature)) - lam) <= 1e-6 * lam  # lambda = exp(log T), mppi.py:398
    m.forward(x0)
    lam2 = float(m._lambda)
    assert lam2 != lam and abs(float(torch.exp(m.log_temperature)) - lam2) <= 1e-6 * lam2


def test_log_temperature_follows_load_state_dict_and_module_conversions():
    """`log_temperature` (MPO, mppi.py:194-199) is a Parameter whose storage IS the library's dual.  load_state_dict() writes
    into it: the library restarts the dual from the loaded value (temperature of the next solve = exp(loaded), Adam moments
    zero) instead of keeping a stale derived temperature; module.to() / .float() would replace the Parameter by a detached
    copy: it is re-bound to the dual afterwards."""
    a, _ = make_solver("pendulum", 15, 1000, lambda_="MPO")
    b, _ = make_solver("pendulum", 15, 1000, lambda_="MPO")
    x = torch.tensor([3.0, 0.0])
    for _ in range(4):
        a.forward(x)
    # (a state dict as the REFERENCE's solver writes it: `log_temperature` only — this build's own also carries `_extra_state`,
    # see test_state_dict_round_trip_continues_bit_identically)
    sd = {k: v.clone() for k, v in a.state_dict().items() if torch.is_tensor(v)}
    assert list(sd) == ["log_temperature"] and abs(float(np.exp(float(sd["log_temperature"][0]))) - a._lambda) <= 1e-6 * a._lambda
    b.load_state_dict(sd)  # (strict: the missing `_extra_state` of a reference checkpoint is not an error)
    assert abs(b._lambda - a._lambda) <= 1e-6 * a._lambda            # the dual was restarted from the loaded value ...
    b.forward(x)
    assert abs(b._last_lambda - a._lambda) <= 1e-6 * a._lambda       # ... and the next solve's weights use exp(loaded)
    ptr = b.log_temperature.data_ptr()
    b.float()
    b.to(torch.device("cuda"))
    assert b.log_temperature.data_ptr() == ptr                       # still the library's dual, not a detached copy
    lam_before = b._lambda
    b.forward(x)
    assert abs(float(np.exp(float(b.log_temperature.detach().cpu()[0]))) - b._lambda) <= 1e-6 * b._lambda and b._lambda != lam_before


def _protocol_cases():
    return [("pendulum", 15, 1000, dict(lambda_="ESSPS")), ("nav2d", 30, 8192, dict(lambda_="LBPS")),
            ("nav2d", 30, 20000, dict(lambda_="MPO")), ("cartpole", 32, 4096, dict(lambda_="ESSPS", use_sg_filter=True)),
            ("nav2d", 20, 4096, dict(lambda_=1.0, noise_source="torch_cpu", exploration=0.1)),
            ("racing", 25, 4000, dict(lambda_=1.0)), ("goalzone", 20, 2000, dict(lambda_=5.0)),
            ("cartpole", 32, 70000, dict(lambda_="ESSPS", use_sg_filter=True, lazy_state_seq=True))]


def _x0_for(model):
    if model == "racing":
        return _envs["racing"].reset().clone().cuda()
    if model == "goalzone":
        from helpers import goalzone_env_fixture
        return torch.from_numpy(np.asarray(goalzone_env_fixture()["x0"], np.float32)).cuda()
    return {"pendulum": torch.tensor([3.0, 0.0]), "nav2d": torch.tensor([-9.0, -9.0, 0.785]),
            "cartpole": torch.tensor([0.01, 0.0, 0.02, 0.0])}[model].cuda()


@pytest.mark.parametrize("model,T,N,kw", _protocol_cases())
def test_deepcopy_mid_loop_continues_bit_identically(model, T, N, kw):
    """copy.deepcopy(solver) — for the reference a plain nn.Module copy (mppi.py:16), here a second handle with the device
    state cloned (mppi_clone_state) — in the middle of a closed loop: original and copy continue with IDENTICAL actions,
    state sequences and temperatures (warm start, RNG stream position, Savitzky-Golay history, ESSPS warm grid, the MPO dual
    with its Adam moments, the racing window's path index all travel), the copy answers queries about the solve it never
    ran (get_top_samples), and the two are independent afterwards.  For racing the CONTROLLER is deep-copied (the solver is
    copied with it and its cost plugin re-bound to the controller's copy)."""
    import copy

    solver, ctrl = make_solver(model, T, N, **kw)
    env = _envs.get("racing")
    x = _x0_for(model)

    def tick(s, c, x):
        if c is not None:
            a, st = c.update(x, env.racing_center_path)
        else:
            a, st = s.forward(x)
        return a, st, torch.as_tensor(st)[0, 1].clone()

    for _ in range(3):
        a, st, x = tick(solver, ctrl, x)
    if ctrl is not None:
        ctrl2 = copy.deepcopy(ctrl)
        twin = ctrl2.solver
        assert twin is not solver and twin._cost_owner is ctrl2 and ctrl2.env is not ctrl.env
    else:
        ctrl2 = None
        twin = copy.deepcopy(solver)
    assert twin._h.h.value != solver._h.h.value and twin._solve_idx == solver._solve_idx
    assert twin._lambda == solver._lambda and twin._last_lambda == solver._last_lambda
    assert torch.equal(twin._previous_action_seq, solver._previous_action_seq)
    t1, w1 = solver.get_top_samples(16)
    t2, w2 = twin.get_top_samples(16)  # (the copy never ran that solve: costs, noise identity and start state were cloned)
    assert torch.equal(t1, t2) and torch.equal(w1, w2)
    x2 = x.clone()
    for k in range(3):
        a1, s1, x = tick(solver, ctrl, x)
        a2, s2, x2 = tick(twin, ctrl2, x2)
        assert torch.equal(a1, a2) and torch.equal(torch.as_tensor(s1), torch.as_tensor(s2)), (model, k)
        assert solver._last_lambda == twin._last_lambda and solver._lambda == twin._lambda
    if kw.get("use_sg_filter"):
        assert np.array_equal(solver._actions_history_for_sg, twin._actions_history_for_sg)
    # independent from here on: another solve on the copy alone moves the copy only
    before = solver._previous_action_seq.clone()
    tick(twin, ctrl2, x2)
    assert torch.equal(solver._previous_action_seq, before) and twin._solve_idx == solver._solve_idx + 1


def test_deepcopy_of_a_generic_path_solver():
    """The same for OPAQUE callables (untagged, not recognised: the generic path keeps its costs from the user's torch loops):
    the copy continues identically — the functions themselves are shared, like copy.deepcopy shares any function."""
    import copy

    from envs import classic_control as cc
    from pi_mpc.mppi import MPPI

    def dyn(s, a):
        return cc.pendulum_dynamics(s, a)

    def cost(s, a, info):
        return cc.pendulum_cost(s, a, info)

    solver = MPPI(horizon=12, num_samples=2048, dim_state=2, dim_control=1, dynamics=dyn, cost_func=cost, u_min=torch.tensor([-2.0]),
                  u_max=torch.tensor([2.0]), sigmas=torch.tensor([1.0]), lambda_="ESSPS", recognize_closures=False)
    assert solver._model is None
    x = torch.tensor([3.0, 0.2]).cuda()
    for _ in range(2):
        _, st = solver.forward(x)
        x = st[0, 1].clone()
    twin = copy.deepcopy(solver)
    assert twin._model is None and twin._dynamics is dyn and twin._h.h.value != solver._h.h.value
    x2 = x.clone()
    for k in range(3):
        a1, s1 = solver.forward(x)
        a2, s2 = twin.forward(x2)
        assert torch.equal(a1, a2) and torch.equal(s1, s2) and solver._last_lambda == twin._last_lambda, k
        x, x2 = s1[0, 1].clone(), s2[0, 1].clone()


def test_top_samples_of_a_very_long_horizon_fail_with_a_message():
    """get_top_samples stages the mean rows in LDS (32 bytes per float4 group of a control row): a horizon whose rows no
    longer fit next to the kernel's static LDS must fail with a message that says so, not with a bare launch error
    (ADVICE r5).  The solve itself still works at that length."""
    from mppi_playground_amd import _capi

    T, N = 5000, 256  # rows of 5 000 floats: 40 KB of staging + ~28 KB static > 64 KB
    solver, _ = make_solver("pendulum", T, N, lambda_=1.0)
    a, s = solver.forward(torch.tensor([3.0, 0.0]))
    assert a.shape == (T, 1) and torch.isfinite(a).all() and torch.isfinite(s).all()
    with pytest.raises(_capi.MppiError, match="LDS"):
        solver.get_top_samples(8)
    short, _ = make_solver("pendulum", 1000, N, lambda_=1.0)  # (8 KB of staging: fine)
    short.forward(torch.tensor([3.0, 0.0]))
    ts, tw = short.get_top_samples(8)
    assert ts.shape == (8, 1001, 2) and abs(float(tw.sum())) > 0


def test_pickling_a_solver_raises_a_message_that_names_the_handle():
    """pickle.dumps(solver) / torch.save(solver) cannot work (the buffers live behind mppi_handle_t): a TypeError that says so
    and names the two supported routes, not ctypes' "objects containing pointers cannot be pickled"."""
    import io
    import pickle

    solver, _ = make_solver("pendulum", 10, 256, lambda_=1.0)
    for dump in (lambda: pickle.dumps(solver), lambda: torch.save(solver, io.BytesIO())):
        with pytest.raises(TypeError) as e:
            dump()
        assert "mppi_handle_t" in str(e.value) and "deepcopy" in str(e.value) and "state_dict" in str(e.value)
    import copy
    with pytest.raises(TypeError):
        copy.copy(solver)  # (a shallow copy would share the handle)


@pytest.mark.parametrize("model,T,N,kw", [("pendulum", 15, 1000, dict(lambda_=0.5)), ("nav2d", 30, 20000, dict(lambda_="MPO")),
                                          ("cartpole", 32, 4096, dict(lambda_=2.0, use_sg_filter=True)),
                                          ("nav2d", 20, 4096, dict(lambda_=1.0, noise_source="torch_cpu"))])
def test_state_dict_round_trip_continues_bit_identically(model, T, N, kw):
    """torch.save(solver.state_dict()) -> a freshly constructed solver -> load_state_dict(): the loaded solver continues like
    the saved one — warm start, Savitzky-Golay history, RNG stream position (Philox index / torch-CPU generator) and the MPO
    dual WITH its Adam moments travel in `_extra_state` (the reference keeps them as plain attributes its state_dict drops)."""
    import io

    a, _ = make_solver(model, T, N, **kw)
    x = _x0_for(model)
    for _ in range(3):
        _, st = a.forward(x)
        x = torch.as_tensor(st)[0, 1].clone()
    buf = io.BytesIO()
    torch.save(a.state_dict(), buf)
    buf.seek(0)
    b, _ = make_solver(model, T, N, **kw)
    b.load_state_dict(torch.load(buf))  # (torch's default weights_only=True: the state is tensors, numbers, lists and None only)
    assert b._solve_idx == a._solve_idx and b._lambda == a._lambda
    xb = x.clone()
    for k in range(3):
        a1, s1 = a.forward(x)
        a2, s2 = b.forward(xb)
        assert torch.equal(a1, a2) and torch.equal(torch.as_tensor(s1), torch.as_tensor(s2)), (model, k)
        assert a._last_lambda == b._last_lambda
        x, xb = torch.as_tensor(s1)[0, 1].clone(), torch.as_tensor(s2)[0, 1].clone()


def test_device_sg_filter_equals_the_host_statement():
    """Step 7 inside finalize_kernel (sg_filter="device", default) against the host numpy statement of the reference's
    filter (sg_filter="host"): same taps, same accumulation order -> identical actions, states and history, over a
    closed loop that fills the history; also a wider window and two control dimensions."""
    for model, T, N, kw in (("cartpole", 64, 1024, dict()), ("cartpole", 12, 256, dict(sg_window_size=9, sg_poly_order=4)),
                            ("nav2d", 30, 512, dict(sg_window_size=7, sg_poly_order=2))):
        dev, _ = make_solver(model, T, N, lambda_=1.0, use_sg_filter=True, **kw)
        dev.set_option("fused_solve", 0)  # (bit-identity is a statement about the same summation order: multi-kernel path)
        host, _ = make_solver(model, T, N, lambda_=1.0, use_sg_filter=True, sg_filter="host", **kw)
        assert dev._sg_on_device and not host._sg_on_device
        x = torch.tensor([0.01, 0.0, 0.02, 0.0]) if model == "cartpole" else torch.tensor([-9.0, -9.0, 0.785])
        for tick in range(8):
            a_d, s_d = dev.forward(x)
            a_h, s_h = host.forward(x)
            assert torch.equal(a_d, a_h), f"{model} tick {tick}"
            assert torch.equal(s_d, s_h)
            assert np.array_equal(dev._actions_history_for_sg, host._actions_history_for_sg)
            x = s_d[0, 1].clone()
        dev.reset()
        assert not dev._actions_history_for_sg.any()


def test_reset_and_posterior_samples():
    """reset() (mppi.py:212-221) zeroes the warm start; get_samples_from_posterior (mppi.py:489-506) returns
    N(a, Sigma) action sequences (unclamped) drawn from the solver's own Philox stream (the call consumes one solve
    index) and their batch rollouts from the GIVEN state — each checked against the oracle — without disturbing
    what get_top_samples reports about the last solve."""
    solver, _ = make_solver("nav2d", 20, 512, lambda_=1.0)
    x0 = torch.tensor([-9.0, -9.0, 0.785])
    a, _ = solver.forward(x0)
    assert float(a.abs().max()) > 0
    top_before = solver.get_top_samples(8)
    idx = solver._solve_idx
    other = torch.tensor([-5.0, -6.0, 0.3])
    samples, states = solver.get_samples_from_posterior(a, other, 16)
    assert solver._solve_idx == idx + 1
    assert samples.shape == (16, 20, 2) and states.shape == (16, 21, 3)
    want = orc.philox_normal(42, idx, 0, 16, 20, 2, [0.5, 0.5]) + a.cpu().numpy()[None]
    assert np.abs(samples.cpu().numpy() - want).max() < 1e-4
    P = oracle_problem("nav2d", 1, 20)
    for i in range(16):
        ref = P.rollout_single(other.numpy(), samples[i].cpu().numpy())
        assert rel_err(states[i].cpu().numpy(), ref) < TOL
    top_after = solver.get_top_samples(8)
    assert torch.equal(top_before[0], top_after[0]) and torch.equal(top_before[1], top_after[1])
    # the next solve draws the index after the posterior's
    solver.forward(x0)
    assert np.abs(solver._action_noises.cpu().numpy() - orc.philox_normal(42, idx + 1, 0, 512, 20, 2, [0.5, 0.5])).max() < 1e-4
    solver.reset()
    h = (C.c_float * 40)()
    solver._h.call("mppi_get_mean", h, 0, solver._stream())
    assert not any(h) and not solver._previous_action_seq.any()
    a2, _ = solver.forward(x0)  # solving again from a zero warm start works
    assert torch.isfinite(a2).all()


def test_queries_do_not_depend_on_the_callers_state_tensor():
    """forward() binds a CUDA state tensor zero-copy; the reference keeps `_state_seq_batch` (mppi.py:280-286,481), so
    its get_top_samples is unaffected by what the caller does to that tensor afterwards.  Here the rollout kernel
    snapshots the state: overwriting the tensor in place (as env.reset() and the mountain-car dynamics do) must not
    change the re-rolled trajectories."""
    for model, T, x in (("racing", 25, None), ("mountaincar", 30, [-0.5, 0.0])):
        solver, ctrl = make_solver(model, T, 2048, lambda_=5.0)
        if ctrl is not None:
            env = _envs["racing"]
            x0 = env.reset().clone()
            ref, _ = ctrl.calc_ref_trajectory(x0, env.racing_center_path, 0, T, DL=0.1, lookahead_distance=3,
                                              reference_path_interval=0.85)
            ctrl.set_reference(ref)
        else:
            x0 = torch.tensor(x)
        xs = x0.cuda().clone()
        solver.forward(xs)
        torch.cuda.synchronize()
        ts, tw = solver.get_top_samples(16)
        S = solver._state_seq_batch[:32].clone()
        xs.mul_(-3.0).add_(1.0)  # the caller reuses its tensor
        torch.cuda.synchronize()
        ts2, tw2 = solver.get_top_samples(16)
        assert torch.equal(ts, ts2) and torch.equal(tw, tw2)
        assert torch.equal(S, solver._state_seq_batch[:32])
        if model != "mountaincar":  # (its stored trajectories hold the mutated views, SURVEY B-Q7)
            assert torch.equal(ts[:, 0, :], x0.cuda().expand(16, -1))


@pytest.mark.parametrize("model,T,N,lam", [("racing", 50, 1 << 17, 2000.0), ("cartpole", 64, 1 << 16, 1.0),
                                           ("racing", 50, 1 << 17, 1.0), ("nav2d", 300, 4096, 50.0)])
def test_both_folds_of_the_partial_rows_give_the_same_bits(model, T, N, lam):
    """The published partial rows are folded either by summarize_kernel or inside finalize_kernel (sparse softmax,
    chosen from a host-side hint of earlier solves): both use one summation tree, so action, state sequence and
    statistics must be bit-identical — results never depend on which path an earlier solve steered to.  (Rows too
    long for the in-kernel fold always take summarize_kernel: nav2d T = 300.)"""
    solver, ctrl = make_solver(model, T, N, lambda_=lam)
    if ctrl is not None:
        env = _envs["racing"]
        x0 = env.reset().clone()
        ref, _ = ctrl.calc_ref_trajectory(x0, env.racing_center_path, 0, T, DL=0.1, lookahead_distance=3,
                                          reference_path_interval=0.85)
        ctrl.set_reference(ref)
    else:
        x0 = torch.tensor({"cartpole": [0.01, 0.0, 0.02, 0.0], "nav2d": [-9.0, -9.0, 0.785]}[model])
    solver.forward(x0)  # costs + noise identity of solve 1 are now resident
    h, st = solver._h, solver._stream()
    ds, dc = solver._dim_state, solver._dim_control
    outs = []
    for path in (1, 2, 0):  # fold inside finalize (when it fits), separate summarize kernel, the default choice
        solver.set_option("fold_path", path)
        a, s, stats = torch.empty(T, dc, device="cuda"), torch.empty(1, T + 1, ds, device="cuda"), torch.empty(4, device="cuda")
        h.call("mppi_weights_reduce", float(lam), None, st)
        h.call("mppi_finalize", None, 1, float(lam), 0, a.data_ptr(), s.data_ptr(), stats.data_ptr(), st)
        solver.join_state_seq()  # (raw C-ABI use of a handle with lazily completed state sequences: join before reading)
        torch.cuda.synchronize()
        outs.append((a.clone(), s.clone(), stats.clone()))
    summ = torch.zeros(4 + T * dc, device="cuda")  # a caller-provided summary buffer (the sharded path) as well
    h.call("mppi_weights_reduce", float(lam), C.c_void_p(summ.data_ptr()), st)
    a2 = torch.empty(T, dc, device="cuda")
    h.call("mppi_finalize", C.c_void_p(summ.data_ptr()), 1, float(lam), 0, a2.data_ptr(), None, None, st)
    assert torch.equal(a2, outs[0][0])
    assert torch.equal(outs[0][0], outs[2][0])
    for x, y in zip(outs[0], outs[1]):
        assert torch.equal(x, y)


# ------------------------------------------------------------------------------ randomised model parameters (C ABI)
def _capi_rollout(model, T, N, params, maps, u_min, u_max, x0, mean, eps, ref=None, math=1):
    """One rollout+cost, reduction and finalize through the raw C ABI with explicit parameters."""
    from mppi_playground_amd import _capi

    ds, dc = orc.MODEL_DIMS[orc.MODEL_IDS[model]]
    f4 = C.c_float * 4
    pad = lambda v, fill: f4(*(list(v) + [fill] * (4 - len(v))))  # noqa: E731
    cfg = _capi.MppiConfig(model=_capi.MODEL_IDS[model], horizon=T, dim_state=ds, dim_control=dc, num_samples=N,
                           sample_offset=0, inherit_count=N, u_min=pad(u_min, 0.0), u_max=pad(u_max, 0.0),
                           sigmas=pad([1.0] * dc, 0.0), seed=3, device=0)
    h = _capi.Handle(cfg)
    h.call("mppi_set_option", b"math", math)
    p = (C.c_float * len(params))(*params)
    h.call("mppi_set_model_params", p, len(params))
    for slot, (cells, cell, origin) in enumerate(maps):
        cells = np.ascontiguousarray(cells, np.uint8)
        h.call("mppi_upload_map", slot, cells.ctypes.data_as(C.c_void_p), cells.shape[0], cells.shape[1], float(cell),
               float(origin[0]), float(origin[1]))
    if ref is not None:
        r = np.ascontiguousarray(ref, np.float32)
        h.call("mppi_set_reference", r.ctypes.data_as(C.c_void_p), r.shape[0], None)
    xd, md, ed = (torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda() for a in (x0, mean, eps))
    h.call("mppi_set_state", xd.data_ptr(), 1, None)
    h.call("mppi_set_mean", md.data_ptr(), 1, None)
    h.call("mppi_inject_noise", ed.data_ptr(), None)
    h.call("mppi_rollout_cost", None)
    costs = torch.empty(N, device="cuda")
    h.call("mppi_get_costs", costs.data_ptr(), 1, None)
    h.call("mppi_weights_reduce", 3.0, None, None)
    a = torch.empty(T, dc, device="cuda")
    s = torch.empty(1, T + 1, ds, device="cuda")
    h.call("mppi_finalize", None, 1, 3.0, 1, a.data_ptr(), s.data_ptr(), None, None)
    torch.cuda.synchronize()
    out = costs.cpu().numpy(), a.cpu().numpy(), s.cpu().numpy()[0]
    h.close()
    return out


@pytest.mark.parametrize("seed", range(20))
def test_randomised_model_parameters_against_oracle(seed):
    """Random racing / nav2d parameter sets that break the launch-uniform preconditions of the fast kernels one way
    or another (steering beyond the tangent polynomial, solver bounds wider than the model's clamp, position limits
    beyond the map so that the padded-grid lookup does not apply, headings increments near pi, odd map sizes): the
    library must fall back on its own and still match the oracle, in both math modes."""
    _need_gpu()
    rng = np.random.default_rng(1000 + seed)
    T, N = int(rng.integers(3, 40)), int(rng.integers(65, 700))
    nx, ny = int(rng.integers(60, 140)), int(rng.integers(60, 140))
    cell = float(rng.choice([0.05, 0.1, 0.125, 0.2]))
    half_x, half_y = nx * cell / 2, ny * cell / 2
    grid = (rng.random((nx, ny)) < 0.08).astype(np.uint8)
    origin = (nx // 2, ny // 2)
    beyond = 1.0 if seed % 3 == 0 else (0.6 if seed % 3 == 1 else 1.3)  # limits inside / beyond the map
    if seed % 5 == 0:
        beyond = 0.6  # (so that the out-of-range start below is still on the map)
    x_lim, y_lim = (-half_x * beyond, half_x * beyond), (-half_y * beyond, half_y * beyond)
    if seed % 2 == 0:
        model, dt = "racing", float(rng.choice([0.05, 0.1, 0.2]))
        steer = float(rng.choice([0.2, 0.25, 0.5, 1.2]))
        mu_min, mu_max = (-float(rng.uniform(1, 3)), -steer), (float(rng.uniform(1, 3)), steer)
        params = orc.racing_params(mu_min, mu_max, L=float(rng.uniform(0.4, 2.5)), v_max=float(rng.uniform(3, 12)), dt=dt,
                                   x_lim=x_lim, y_lim=y_lim, Qc=float(rng.uniform(0, 4)), Ql=float(rng.uniform(0, 4)),
                                   Qv=float(rng.uniform(0, 3)), Qo=float(rng.uniform(10, 1e4)),
                                   Qin=float(rng.uniform(0, 1)), Qdin=float(rng.uniform(0, 1)))
        lane = (rng.random((nx, ny)) < 0.3).astype(np.uint8)
        maps = [(grid, cell, origin), (lane, cell, origin)]
        ref = np.cumsum(rng.standard_normal((T + 1, 4)) * 0.2, axis=0).astype(np.float32)
        ref[:, 2] = np.arctan2(np.sin(ref[:, 2]), np.cos(ref[:, 2]))
        x0 = np.array([rng.uniform(*x_lim) * 0.5, rng.uniform(*y_lim) * 0.5, rng.uniform(-3.1, 3.1), rng.uniform(0, 3)])
        scale = np.array([1.0, 0.2])
    else:
        model, dt, ref = "nav2d", float(rng.choice([0.05, 0.1, 0.3])), None
        mu_min, mu_max = (0.0, -float(rng.uniform(0.5, 9.0))), (float(rng.uniform(1, 4)), float(rng.uniform(0.5, 9.0)))
        params = orc.nav2d_params(mu_min, mu_max, dt=dt, x_lim=x_lim, y_lim=y_lim,
                                  goal=(float(rng.uniform(*x_lim)), float(rng.uniform(*y_lim))), Qo=float(rng.uniform(10, 1e4)))
        maps = [(grid, cell, origin)]
        x0 = np.array([rng.uniform(*x_lim) * 0.5, rng.uniform(*y_lim) * 0.5, rng.uniform(-3.1, 3.1)])
        scale = np.array([1.0, 1.0])
    wide = seed % 4 == 1  # solver bounds wider than the model's own clamp
    u_min = [m * (1.5 if wide else 1.0) - (0.5 if wide else 0.0) for m in mu_min]
    u_max = [m * (1.5 if wide else 1.0) + (0.5 if wide else 0.0) for m in mu_max]
    if seed % 5 == 0:  # initial position outside the model's clamp range: the fast kernels' per-lane redo path
        x0[0], x0[1] = x_lim[1] * 1.2, y_lim[0] * 1.1
    x0[:2] = np.round(x0[:2] / cell) * cell  # a cell centre: the shared first lookup is far from a rounding boundary
    x0 = x0.astype(np.float32)
    mean = (rng.standard_normal((T, 2)) * 0.3 * scale).astype(np.float32)
    eps = (rng.standard_normal((N, T, 2)) * scale).astype(np.float32)
    P = orc.Problem(model, N, T, u_min, u_max, params=params, maps=maps, ref_path=ref)
    r = P.rollout_cost(x0, mean, eps, want_margin=True)
    for math in (2, 1, 0):
        c, a, s = _capi_rollout(model, T, N, params, maps, u_min, u_max, x0, mean, eps, ref, math)
        check_costs(c, r, max_flips=3 + N // 50)
        w, _ = orc.softmax_weights(c, 3.0)
        a_or = P.weighted_actions(w, mean, eps)
        assert rel_err(a, a_or) < TOL, (model, math)
        assert rel_err(s, P.rollout_single(x0, a)) < TOL, (model, math)


@pytest.mark.parametrize("model", ["pendulum", "cartpole", "mountaincar", "mjcartpole", "goalzone"])
def test_extreme_initial_states_against_oracle(model):
    """Initial states far outside the nominal ranges (many turns of angle, states at or beyond the model's clamps,
    huge velocities): every fast path either covers them exactly or flags the lane for the library-math redo."""
    rng = np.random.default_rng(7)
    T, N = 25, 192
    cases = {
        "pendulum": [[3.0, 0.1], [150.0, -3.0], [-199.0, 8.0], [250.0, 1.0], [-9.9e4, 7.9], [1.2e5, -8.0], [0.0, 100.0]],
        "cartpole": [[0.0, 0.0, 0.1, 0.0], [2.4, 5.0, 190.0, -7.0], [-30.0, -20.0, -250.0, 30.0], [0.0, 0.0, 3.1415927, 0.0],
                     [1.0, 0.0, 1e4, 2.0]],
        "mountaincar": [[-0.5, 0.0], [-1.2, -0.07], [0.6, 0.07], [-1.5, 0.3], [0.45, 0.0]],
        "mjcartpole": [[0.0, 0.0, 0.05, 0.0], [3.0, 4.0, 170.0, -9.0], [-5.0, 0.0, -230.0, 40.0], [0.1, 0.2, 5e3, 0.0]],
        "goalzone": [[0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0], [4.0, -3.0, 3.1, 0.5, -0.5, 1.0, 0.0],
                     [-12.0, 15.0, -3.14159, 0.0, 0.0, 0.0, 1.0], [2.0, 2.0, 100.0, 0.3, 0.2, 0.0, 0.0]],
    }[model]
    solver, _ = make_solver(model, T, N, lambda_=2.0)
    P = oracle_problem(model, N, T)
    for x0 in cases:
        x0 = np.asarray(x0, np.float32)
        mean = (rng.standard_normal((T, P.dc)) * 0.3).astype(np.float32)
        eps = (rng.standard_normal((N, T, P.dc)) * np.asarray(MODEL_CFG[model]["sigmas"])).astype(np.float32)
        solver.set_warm_start(mean)
        solver.inject_noise(torch.from_numpy(eps))
        a, s = solver.forward(torch.from_numpy(x0))
        c = solver._costs.cpu().numpy()
        r = P.rollout_cost(x0, mean, eps, want_margin=True)
        assert np.all(np.isfinite(c)), x0
        scale = max(np.abs(r["costs"]).max(), 1e-30)
        # (angles of hundreds of radians carry ulps of 1e-5 rad: 1e-4 is the conditioning of these cases, not a precision claim)
        assert np.abs(c - r["costs"]).max() <= 1e-4 * scale, (model, x0.tolist(), np.abs(c - r["costs"]).max() / scale)
        w, _ = orc.softmax_weights(c, 2.0)
        assert rel_err(a.cpu().numpy(), P.weighted_actions(w, mean, eps)) < TOL
        assert rel_err(s.cpu().numpy()[0], P.rollout_single(x0, a.cpu().numpy())) < 1e-4


def test_top_samples_with_tied_costs():
    """Radix select when many costs are bit-identical: all equal, two plateaus with the k-th rank inside a plateau,
    and negative / zero costs (key order across the sign)."""
    solver, _ = make_solver("pendulum", 10, 4096, lambda_=1.0)
    x0 = torch.tensor([1.0, 0.0])
    solver.forward(x0)
    st = solver._stream()
    for costs, k in ((np.full(4096, 5.0, np.float32), 300),
                     (np.where(np.arange(4096) % 3 == 0, 1.0, 2.0).astype(np.float32), 1000),
                     (np.concatenate([np.full(100, -3.0), np.zeros(100), -np.zeros(100), np.linspace(0.5, 9, 3796)]).astype(np.float32), 250)):
        c = torch.from_numpy(costs).cuda()
        solver._h.call("mppi_set_costs", c.data_ptr(), 1, st)
        solver._h.call("mppi_weights_reduce", 1.0, None, st)   # refreshes the statistics the weights are normalised with
        a = torch.empty(10, 1, device="cuda")
        solver._h.call("mppi_finalize", None, 1, 1.0, 0, a.data_ptr(), None, None, st)
        out = torch.empty(k, 11, 2, device="cuda")
        w = torch.empty(k, device="cuda")
        solver._h.call("mppi_top_samples", k, 1.0, out.data_ptr(), w.data_ptr(), st)
        x = -costs.astype(np.float64)
        ref_w = np.exp(x - x.max())
        ref_w /= ref_w.sum()
        want = np.sort(ref_w)[::-1][:k]
        assert rel_err(w.cpu().numpy(), want) < TOL and torch.isfinite(out).all()
        assert torch.equal(out[:, 0, :], x0.cuda().expand(k, 2))


def test_reference_side_binding_is_self_sufficient():
    """example/reference_binding.py — the ctypes stub INTEGRATION.md shows a maintainer of the reference, using only
    libmppi_hip.so through include/mppi_hip.h — drives the same solves as this build's MPPI class: bit-identical
    action / state sequences over a closed loop, step-by-step calls and the one-call entry point, fixed lambda and ESSPS."""
    _need_gpu()
    import importlib.util
    import os

    from helpers import ROOT
    from mppi_playground_amd import _capi

    spec = importlib.util.spec_from_file_location("reference_binding", os.path.join(ROOT, "example", "reference_binding.py"))
    rb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rb)
    for model, T, N, x0 in (("pendulum", 15, 1000, [3.0, 0.1]), ("cartpole", 20, 4096, [0.01, 0.0, 0.02, 0.0])):
        mc = MODEL_CFG[model]
        for essps in (False, True):
            ours, _ = make_solver(model, T, N, lambda_="ESSPS" if essps else 0.7)
            ours.set_option("fused_solve", 0)  # (the single launch sums in another order: compare like with like)
            hips = [rb.HipForward(_capi.LIB_PATH, model, T, N, mc["u_min"], mc["u_max"], mc["sigmas"], seed=42) for _ in range(2)]
            for hp in hips:
                hp.lib.mppi_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
                assert hp.lib.mppi_set_option(hp.h, b"fused_solve", 0) == 0
            states = [torch.tensor(x0, device="cuda") for _ in range(3)]
            for tick in range(4):
                a0, s0 = ours.forward(states[0])
                outs = [hips[i].forward(states[1 + i], lam=0.7, essps_target=N / 10 if essps else None, one_call=bool(i))
                        for i in range(2)]
                for a, s in outs:
                    assert torch.equal(a, a0) and torch.equal(s, s0), (model, essps, tick)
                states = [s0[0, 1].clone()] + [s[0, 1].clone() for _, s in outs]
            for hp in hips:
                hp.close()


@pytest.mark.parametrize("lam", [5000.0, 1.0])
def test_c4_eight_shards_at_full_size_on_one_device(lam):
    """BASELINE configs[3] (racing, N = 8 388 608 over 8 GPUs) minus the transport: eight shard handles of 2^20 samples
    each (sample_offset = r * 2^20, global exploration threshold) run one after the other on THIS device, their
    summaries combined by mppi_finalize(num_shards = 8), against one unsharded handle of 2^23 samples.  The noise is
    a function of the global index, so each shard's costs must equal its slice of the full run bit for bit; the
    combined action / state sequence differs from the unsharded one by the rounding of the combine only."""
    _need_gpu()
    from mppi_playground_amd import _capi

    W, NL, T = 8, 1 << 20, 50
    N = W * NL
    # lambda = 5000: thousands of samples carry weight (every shard contributes); lambda = 1 (BASELINE configs[3]): arg-min
    full, ctrl = make_solver("racing", T, N, lambda_=lam, exploration=0.1)
    env = _envs["racing"]
    x0 = env.reset().clone()
    ref, _ = ctrl.calc_ref_trajectory(x0, env.racing_center_path, 0, T, DL=0.1, lookahead_distance=3,
                                      reference_path_interval=0.85)
    ctrl.set_reference(ref)
    mean = (np.random.default_rng(4).standard_normal((T, 2)) * np.array([0.3, 0.05])).astype(np.float32)
    full.set_warm_start(mean)
    a_full, s_full = full.forward(x0)
    c_full = full._costs
    st_full = full.last_stats()
    assert st_full["ess"] > 100 or lam == 1.0  # a dense softmax: every shard contributes
    sums, shard0 = [], None
    for r in range(W):
        sol, c2 = make_solver("racing", T, NL, lambda_=lam)
        cfg = _capi.MppiConfig()
        cfg.model, cfg.horizon, cfg.dim_state, cfg.dim_control = 4, T, 4, 2
        cfg.num_samples, cfg.sample_offset, cfg.inherit_count = NL, r * NL, int(N * 0.9)
        for k in range(2):
            cfg.u_min[k], cfg.u_max[k], cfg.sigmas[k] = (MODEL_CFG["racing"][q][k] for q in ("u_min", "u_max", "sigmas"))
        cfg.seed, cfg.device = 42, 0
        sol._h.close()
        sol._h = _capi.Handle(cfg)
        sol._uploaded, sol._params_set, sol._ref_uploaded = {}, None, None
        c2.set_reference(ref)
        sol._h.call("mppi_set_state", C.c_void_p(x0.data_ptr()), 1, sol._stream())
        sol._refresh_model_inputs()
        md = torch.from_numpy(mean).cuda()
        sol._h.call("mppi_set_mean", C.c_void_p(md.data_ptr()), 1, sol._stream())
        sol._h.call("mppi_sample", 1, sol._stream())
        sol._h.call("mppi_rollout_cost", sol._stream())
        assert torch.equal(sol._costs, c_full[r * NL:(r + 1) * NL]), f"shard {r}: costs differ from the unsharded slice"
        sums.append(_summary(sol, lam))
        torch.cuda.synchronize()
        if r == 0:
            shard0 = sol  # (kept: its handle runs the combine)
        else:
            sol._h.close()
    allsum = torch.stack(sums).contiguous()
    a = torch.zeros(T, 2, device="cuda")
    s = torch.zeros(1, T + 1, 4, device="cuda")
    stats = torch.zeros(4, device="cuda")
    shard0._h.call("mppi_finalize", C.c_void_p(allsum.data_ptr()), W, lam, 0, C.c_void_p(a.data_ptr()),
                   C.c_void_p(s.data_ptr()), C.c_void_p(stats.data_ptr()), shard0._stream())
    shard0.join_state_seq()
    check_rel("sharded_action_seq_vs_unsharded", a.cpu().numpy(), a_full.cpu().numpy(), 4e-6)
    check_rel("sharded_state_seq_vs_unsharded", s.cpu().numpy(), s_full.cpu().numpy(), 4e-6)
    st = stats.cpu().numpy()
    assert st[0] == st_full["cmin"] and abs(st[1] - st_full["sum_e"]) <= 2e-5 * st_full["sum_e"]
    assert abs(st[1] * st[1] / st[2] - st_full["ess"]) <= 1e-4 * st_full["ess"]


def test_c4_eight_shards_match_the_reference_at_full_size():
    """BASELINE configs[3] against the reference ITSELF: the 8 x 2^20 = 8 388 608 racing samples of the sharded run were put
    through the real reference UNSHARDED in the build container (tests/golden/make_golden.py fullsize c4: seed 42, two
    closed-loop solves, ~35 GB of RSS; outputs and summaries only).  Here eight shard handles (sample_offset = r * 2^20) on
    this one device are fed their slices of the same torch-CPU noise stream, and their summaries are combined by
    mppi_finalize(num_shards = 8) — the sharded solve minus the transport.  Checked per solve: the noise block's checksums,
    the 32 smallest (index, cost) pairs over all shards in the reference's order, minimum / maximum / float64 sum / order
    statistics of the 8 M costs, and action_seq / state_seq (arg-min regime: action == U[argmin] of the reference)."""
    import os

    from helpers import GOLDEN
    from mppi_playground_amd import _capi

    name = "full_c4_racing_T50_N8388608_lambda1"
    if not os.path.exists(os.path.join(GOLDEN, name + ".npz")):
        pytest.skip(f"{name}.npz not generated (make_golden.py fullsize c4: ~1.5 h, 35 GB)")
    _need_gpu()
    g = load(name)
    W, T = 8, int(g["T"])
    N, K = int(g["N"]), int(g["K"])
    NL = N // W
    mc = MODEL_CFG["racing"]
    sig = torch.tensor(mc["sigmas"])
    gen = torch.Generator(device="cpu").manual_seed(int(g["seed"]))
    ctor = torch.randn(N, T, 2, generator=gen)  # the constructor's draw (mppi.py:146-148)
    assert float((ctor * sig).numpy().astype(np.float64).sum()) == float(g["ctor_eps_sum64"])
    del ctor
    shards = []
    for r in range(W):
        sol, c2 = make_solver("racing", T, NL, lambda_=1.0)
        cfg = _capi.MppiConfig()
        cfg.model, cfg.horizon, cfg.dim_state, cfg.dim_control = 4, T, 4, 2
        cfg.num_samples, cfg.sample_offset, cfg.inherit_count = NL, r * NL, N
        for k in range(2):
            cfg.u_min[k], cfg.u_max[k], cfg.sigmas[k] = (mc[q][k] for q in ("u_min", "u_max", "sigmas"))
        cfg.seed, cfg.device = 42, 0
        sol._h.close()
        sol._h = _capi.Handle(cfg)
        sol._uploaded, sol._params_set, sol._ref_uploaded = {}, None, None
        shards.append((sol, c2))
    env = _envs["racing"]
    ctrl0 = shards[0][1]
    state = torch.from_numpy(g["x0_0"])
    lo, hi = np.float32(mc["u_min"]), np.float32(mc["u_max"])
    for k in range(K):
        assert rel_err(state.cpu().numpy(), g[f"x0_{k}"]) <= TOL
        ref, ctrl0.current_path_index = ctrl0.calc_ref_trajectory(state, env.racing_center_path, ctrl0.current_path_index, T,
                                                                  DL=0.1, lookahead_distance=3, reference_path_interval=0.85)
        assert np.array_equal(ref.numpy(), g[f"ref_path_{k}"])
        eps = torch.randn(N, T, 2, generator=gen) * sig  # this solve's block of the reference's stream
        e64 = eps.numpy().astype(np.float64)
        assert float(e64.sum()) == float(g[f"eps_sum64_{k}"]) and float((e64 * e64).sum()) == float(g[f"eps_sumsq64_{k}"])
        del e64
        top_i = g[f"top32_idx_{k}"]
        assert np.array_equal(eps.numpy()[top_i], g[f"top32_eps_{k}"])
        x0d = state.cuda().contiguous()
        md = torch.from_numpy(g[f"mean_in_{k}"]).cuda()
        sums, costs = [], []
        for r, (sol, c2) in enumerate(shards):
            c2.set_reference(ref)
            sol._h.call("mppi_set_state", C.c_void_p(x0d.data_ptr()), 1, sol._stream())
            sol._refresh_model_inputs()
            sol._h.call("mppi_set_mean", C.c_void_p(md.data_ptr()), 1, sol._stream())
            sl = eps[r * NL:(r + 1) * NL].cuda().contiguous()
            sol._h.call("mppi_inject_noise", C.c_void_p(sl.data_ptr()), sol._stream())
            sol._h.call("mppi_rollout_cost", sol._stream())
            costs.append(sol._costs.cpu().numpy())
            sums.append(_summary(sol, 1.0))
            torch.cuda.synchronize()
            del sl
        c = np.concatenate(costs)
        scale = float(g[f"cmax_{k}"])
        assert np.abs(c[top_i] - g[f"top32_cost_{k}"]).max() <= TOL * scale
        order = np.lexsort((np.arange(N), c))[:32]
        assert np.array_equal(order, top_i), (order[:8], top_i[:8])  # (no pair of the reference's top 32 is closer than 4 ulps: checked below)
        assert np.diff(g[f"top32_cost_{k}"].astype(np.float64)).min() >= 4 * EPS32 * float(np.abs(g[f"top32_cost_{k}"]).max())
        assert abs(float(c.min()) - float(g[f"cmin_{k}"])) <= TOL * scale and abs(float(c.max()) - scale) <= TOL * scale
        rel_sum = abs(float(c.astype(np.float64).sum()) - float(g[f"costs_sum64_{k}"])) / abs(float(g[f"costs_sum64_{k}"]))
        parity_report.record("cost_sum_rel_err_vs_reference_full_c4", rel_sum, 1e-6)
        assert rel_sum <= 1e-6
        assert np.abs(np.sort(c)[g[f"quantile_ranks_{k}"]] - g[f"quantiles_{k}"]).max() <= TOL * scale
        allsum = torch.stack(sums).contiguous()
        a = torch.zeros(T, 2, device="cuda")
        s = torch.zeros(1, T + 1, 4, device="cuda")
        stats = torch.zeros(4, device="cuda")
        h0 = shards[0][0]
        h0._h.call("mppi_finalize", C.c_void_p(allsum.data_ptr()), W, 1.0, 0, C.c_void_p(a.data_ptr()), C.c_void_p(s.data_ptr()),
                   C.c_void_p(stats.data_ptr()), h0._stream())
        h0.join_state_seq()
        U = np.clip(g[f"mean_in_{k}"] + g[f"top32_eps_{k}"][0], lo, hi)
        assert float(g[f"top32_weight_{k}"][0]) >= 1.0 - 1e-6 and int(np.argmin(c)) == int(top_i[0])
        assert np.abs(a.cpu().numpy() - U).max() <= 1e-6 * np.abs(U).max()
        band = full_size_band(g, k)
        check_banded("action_seq_vs_reference_fixture_full_c4", a.cpu().numpy(), g[f"action_seq_{k}"], band["action"])
        check_banded("state_seq_vs_reference_fixture_full_c4", s.cpu().numpy(), g[f"state_seq_{k}"], band["state"])
        u = torch.clamp(a[0], env.u_min, env.u_max)
        state = env.dynamics(state.cuda().unsqueeze(0), u.unsqueeze(0)).squeeze(0)
        del eps


def test_generic_path_hipgraph_capture_of_the_callables():
    """graph_callables=True: the reference's two T-step Python loops over opaque callables, captured once into a
    hipGraph (after one eager warm-up solve) and replayed — bit-identical to the eager loops over a closed loop
    (pendulum, mountain car with its in-place mutation, cart-pole with its masked assignments) and faster; a callable
    that is not capturable (it synchronises with the host) falls back to the eager loops with a warning, same results."""
    _need_gpu()
    import time
    import warnings

    from envs import classic_control as cc
    from pi_mpc.mppi import MPPI

    def host_sync_cost(s, u, info):  # .item() cannot be captured
        return cc.pendulum_cost(s, u, info) * (1.0 + 0.0 * float(s[0, 0].item()))

    for model, T, N, x0, cost_fn in (("pendulum", 30, 8192, [3.0, 0.1], None), ("mountaincar", 40, 4096, [-0.5, 0.0], None),
                                     ("cartpole", 20, 2048, [0.01, 0.0, 0.02, 0.0], None),
                                     ("pendulum", 10, 512, [3.0, 0.1], host_sync_cost)):
        mc = MODEL_CFG[model]
        ds, dc = orc.MODEL_DIMS[orc.MODEL_IDS[model]]
        make = lambda **kw: MPPI(horizon=T, num_samples=N, dim_state=ds, dim_control=dc,  # noqa: E731
                                 dynamics=_untagged(getattr(cc, f"{model}_dynamics")),
                                 cost_func=cost_fn or _untagged(getattr(cc, f"{model}_cost")), u_min=torch.tensor(mc["u_min"]),
                                 u_max=torch.tensor(mc["u_max"]), sigmas=torch.tensor(mc["sigmas"]), lambda_=0.5, **kw)
        eager, graph = make(), make(graph_callables=True)
        assert eager._model is None and graph._model is None
        xe = xg = torch.tensor(x0)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            for tick in range(6):
                ae, se = eager.forward(xe)
                ag, sg = graph.forward(xg)
                assert torch.equal(ae, ag) and torch.equal(se, sg), (model, tick)
                assert torch.equal(eager._costs, graph._costs)
                xe, xg = se[0, 1].clone(), sg[0, 1].clone()
        warned = [w for w in caught if "graph_callables" in str(w.message)]
        if cost_fn is None:
            assert graph._graph_state == "replay" and not warned
            times = []
            for sol, x in ((eager, xe), (graph, xg)):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    sol.forward(x)
                torch.cuda.synchronize()
                times.append((time.perf_counter() - t0) / 20 * 1e3)
            print(f"generic {model} N={N} T={T}: eager loops {times[0]:.2f} ms/solve, hipGraph replay {times[1]:.2f} ms/solve")
            assert times[1] < times[0]
            # a replay fills the caller's dict like the eager loop leaves it ...
            info = {}
            graph.forward(xg, info)
            assert set(info) == {"prev_state", "prev_action", "initial_state", "t"} and info["t"] == T - 1
            # ... and refuses entries of the caller's that it cannot have seen at capture time
            with pytest.raises(RuntimeError, match="recapture"):
                graph.forward(xg, {"weights_table": torch.ones(3, device="cuda")})
            graph.recapture()
            assert graph._graph_state == "warmup"
            a_w, _ = graph.forward(xg)   # eager warm-up
            a_c, _ = graph.forward(xg)   # captures again
            assert graph._graph_state == "replay" and torch.isfinite(a_c).all()
        else:
            assert graph._graph_state == "failed" and warned


def test_rccl_backed_exchange_path_with_one_rank():
    """The multi-GPU exchanges on the REAL backend: a one-rank `nccl` (= RCCL) process group — RCCL cannot put two ranks
    on one device, which is why the two-rank tests above use gloo — with the solver's `_force_exchange` hook, so that
    every solve takes the N-GPU code path: summary written by summarize_kernel -> all_gather (torch.distributed on
    ProcessGroupNCCL's stream, or the library's own ncclAllGather on the solve's stream) -> mppi_finalize on the gathered
    buffer.  scripts/nccl_single_rank.py checks the results against the unsharded solve and prints the fixed cost of each
    path per solve."""
    _need_gpu()
    import os
    import subprocess
    import sys

    from helpers import ROOT

    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "nccl_single_rank.py")], env=env, capture_output=True,
                       text=True, timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if "forced exchange" in ln or "RCCL all_gather path" in ln]
    print("\n".join(lines))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert len(lines) == 5


def test_essps_device_search_equals_the_host_loop_on_random_costs():
    """mppi_essps_lambda_device (both statistics passes AND both scalar steps as kernels, the default of forward()) against
    mppi_essps_lambda (the same steps in a host loop over read-back statistics) and against scipy's brentq on a float64
    evaluation of ESS(lambda), over cost vectors of very different shapes: Gaussian, heavy-tailed, two clusters, a huge
    common offset, all equal (ESS = N at every temperature: lambda_min), one far outlier (ESS stays near 1: lambda_max)."""
    _need_gpu()
    from scipy.optimize import brentq

    rng = np.random.default_rng(11)
    solver, _ = make_solver("pendulum", 10, 300_000, lambda_=1.0)
    solver.forward(torch.tensor([1.0, 0.0]))  # (a handle with costs[N] allocated and a noise identity)
    h, st = solver._h, solver._stream()
    N = 300_000
    shapes = {
        "gauss": lambda: rng.standard_normal(N) * 3.0 + 50.0,
        "gauss_small_spread": lambda: rng.standard_normal(N) * 0.02 + 7.0,
        "exponential": lambda: rng.exponential(5.0, N),
        "lognormal": lambda: np.exp(rng.standard_normal(N) * 1.5),
        "two_clusters": lambda: np.where(rng.random(N) < 0.05, rng.standard_normal(N) * 0.5, 40.0 + rng.standard_normal(N)),
        "offset_1e6": lambda: 1.0e6 + rng.standard_normal(N) * 4.0,
        "all_equal": lambda: np.full(N, 3.25),
        "one_outlier": lambda: np.concatenate([[-1.0e4], 100.0 + rng.standard_normal(N - 1)]),
    }
    for name, draw in shapes.items():
        c = draw().astype(np.float32)
        cd = torch.from_numpy(c).cuda()
        h.call("mppi_set_costs", cd.data_ptr(), 1, st)
        for target in (N / 10, 50.0, N * 0.9):
            lam_host = C.c_double(0.0)
            h.call("mppi_set_option", b"essps_cold", 1)  # (both searches are warm-started from their last root otherwise)
            h.call("mppi_essps_lambda", float(target), 0.01, 10.0, C.byref(lam_host), st)
            h.call("mppi_essps_lambda_device", float(target), 0.01, 10.0, st)
            lam_dev = C.c_double(0.0)
            h.call("mppi_get_lambda", C.byref(lam_dev), None, st)
            assert abs(lam_dev.value - lam_host.value) <= 1e-12 * lam_host.value, (name, target, lam_dev.value, lam_host.value)
            # again, now from the grid the finished search left behind (clustered around its root: one pass over the
            # costs), and once more from THAT search's grid: the same root
            for _ in range(2):
                h.call("mppi_essps_lambda_device", float(target), 0.01, 10.0, st)
                lam_warm = C.c_double(0.0)
                h.call("mppi_get_lambda", C.byref(lam_warm), None, st)
                assert abs(lam_warm.value - lam_host.value) <= 5e-6 * lam_host.value, (name, target, lam_warm.value, lam_host.value)
            c64 = c.astype(np.float64)
            ess = lambda lam: (lambda e: e.sum() ** 2 / (e * e).sum())(np.exp(-(c64 - c64.min()) / lam))  # noqa: E731
            if target <= ess(0.01):
                want = 0.01
            elif target >= ess(10.0):
                want = 10.0
            else:
                want = brentq(lambda lam: ess(lam) - target, 0.01, 10.0, xtol=1e-12)
            assert abs(lam_dev.value - want) <= 2e-4 * want, (name, target, lam_dev.value, want)
        # a search that starts from the grid around ANOTHER problem's root (the next target, never made cold)
        for target in (N / 10, 50.0, N * 0.9, 2000.0, N / 3):
            lam_host = C.c_double(0.0)
            h.call("mppi_essps_lambda", float(target), 0.01, 10.0, C.byref(lam_host), st)
            h.call("mppi_essps_lambda_device", float(target), 0.01, 10.0, st)
            lam_dev = C.c_double(0.0)
            h.call("mppi_get_lambda", C.byref(lam_dev), None, st)
            assert abs(lam_dev.value - lam_host.value) <= 5e-6 * lam_host.value, (name, target, lam_dev.value, lam_host.value)


def test_map_lookup_outside_the_solver_is_one_native_launch():
    """ObstacleMap.compute_cost / LaneMap.compute_cost on device tensors (env.collision_check of the examples' loops,
    cost plugins on the generic path) go through mppi_grid_lookup: bit-identical to the reference's arithmetic evaluated
    in numpy float32 (true division, round half to even, out of the grid = 1) for contiguous points, the [:, :, :2]
    view of state rows, one time step of a state buffer, an irregular view, points outside the grid and NaN."""
    _need_gpu()
    env = _envs["racing"]
    rng = np.random.default_rng(5)
    for m in (env._obstacle_map, env._lane_map):
        grid = m._map_torch
        g = grid.cpu().numpy()
        cell = np.float32(m._cell_size)
        org = m._torch_cell_map_origin.cpu().numpy().astype(np.float32)

        def want(p):
            q = np.rint(p.astype(np.float32) / cell + org)  # float32 throughout; rint = round half to even
            inb = (q[..., 0] >= 0) & (q[..., 0] < g.shape[0]) & (q[..., 1] >= 0) & (q[..., 1] < g.shape[1])
            ix = np.where(inb, q[..., 0], 0).astype(np.int64)
            iy = np.where(inb, q[..., 1], 0).astype(np.int64)
            return np.where(inb, g[ix, iy], np.float32(1.0)).astype(np.float32)

        lim = 1.2 * max(abs(float(m.x_lim[0])), abs(float(m.x_lim[1])))  # some points outside the map
        S = torch.from_numpy((rng.uniform(-lim, lim, (257, 31, 4))).astype(np.float32)).cuda()
        S[3, 4, 0] = float("nan")
        S[5, 6, 1] = float("inf")
        # half-integer cell coordinates: the tie cases of the rounding
        S[7, :, 0] = torch.from_numpy(((np.arange(31) + 0.5 - org[0]) * cell).astype(np.float32)).cuda()
        cases = {"state rows [:, :, :2]": S[:, :, :2], "contiguous": S[:, :, :2].contiguous(), "one step of a buffer": S[:, 9, None, :2],
                 "irregular view": S[:, ::3, 1:3], "a single trajectory": S[:1, :, :2]}
        for name, x in cases.items():
            got = m.compute_cost(x)
            assert got.shape == x.shape[:-1] and got.dtype == torch.float32, name
            assert np.array_equal(got.cpu().numpy(), want(x.cpu().numpy())), name
    # the env call of the examples' loops
    st = torch.zeros(1, 26, 4, device="cuda")
    st[0, :, :2] = torch.from_numpy(rng.uniform(-30, 30, (26, 2)).astype(np.float32)).cuda()
    c = env.collision_check(state=st)
    assert c.shape == (1, 26) and bool(((c == 0) | (c == 1)).all())


@pytest.mark.parametrize("fused", [0, 2])
def test_essps_warm_start_in_a_closed_loop(fused):
    """From the second solve on the device-resident ESSPS search starts from the grid the previous search left around its
    root: ONE pass over the costs (mppi_search_passes) while the temperature moves slowly, and every tick the temperature
    of the reference's brentq on a float64 evaluation of the same costs (<= 1e-5 relative); reset() / option "essps_cold"
    bring the next search back to the geometric grid (two passes)."""
    _need_gpu()
    from scipy.optimize import brentq

    solver, _ = make_solver("nav2d", 30, 4096, lambda_="ESSPS")
    solver.set_option("fused_solve", fused)
    h = solver._h
    x = torch.tensor([-9.0, -9.0, 0.785]).cuda()
    passes = []
    for k in range(12):
        if k == 8:
            solver.set_option("essps_cold", 1)
        a, st = solver.forward(x)
        passes.append(h.lib.mppi_search_passes(h.h, None))
        lam = solver._last_lambda
        c = solver._costs.cpu().numpy().astype(np.float64)
        ess = lambda l: (lambda e: e.sum() ** 2 / (e * e).sum())(np.exp(-(c - c.min()) / l))  # noqa: E731
        assert 0.01 < lam < 10.0
        want = brentq(lambda l: ess(l) - 409.6, 0.01, 10.0, xtol=1e-13)
        assert abs(lam - want) <= 1e-5 * want, (k, lam, want)
        x = st[0, 1].clone()
    assert passes[0] == 2 and passes[8] == 2, passes          # cold searches
    assert passes[1:8].count(1) >= 4 and passes[9:].count(1) >= 1, passes  # warm ones (the first ticks of a loop move the temperature most)
    solver.reset()
    solver.forward(x)
    assert h.lib.mppi_search_passes(h.h, None) == 2


@pytest.mark.parametrize("N", [65536, 262144 + 1000])
def test_essps_rounds_as_one_launch_same_temperature_to_the_bit(N):
    """Round 1 of the device-resident ESSPS chain is ONE launch (statistics pass + select step: block 0 gathers the other
    blocks' partial sums through tagged cells; every block returns at once when round 0 finished the search), and option
    "essps_merge0" makes round 0 one as well.  Same sums in the same order as a statistics kernel followed by a one-block
    select kernel: the temperatures, the number of passes and the actions are bit-equal over a loop with cold (two-pass)
    and warm searches, and every temperature is the root brentq finds on a float64 evaluation of the same costs."""
    _need_gpu()
    from scipy.optimize import brentq

    a_s, _ = make_solver("nav2d", 30, N, lambda_="ESSPS")
    b_s, _ = make_solver("nav2d", 30, N, lambda_="ESSPS")
    a_s.set_option("fused_solve", 0)
    b_s.set_option("fused_solve", 0)
    b_s.set_option("essps_merge0", 1)
    x = torch.tensor([-9.0, -9.0, 0.785]).cuda()
    passes = []
    for k in range(10):
        if k in (4, 5):
            a_s.set_option("essps_cold", 1)
            b_s.set_option("essps_cold", 1)
        a, st = a_s.forward(x)
        b, sb = b_s.forward(x)
        pa = a_s._h.lib.mppi_search_passes(a_s._h.h, None)
        assert pa == b_s._h.lib.mppi_search_passes(b_s._h.h, None)
        passes.append(pa)
        assert a_s._last_lambda == b_s._last_lambda, (k, a_s._last_lambda, b_s._last_lambda)
        assert torch.equal(a, b) and torch.equal(torch.as_tensor(st), torch.as_tensor(sb))
        c = a_s._costs.cpu().numpy().astype(np.float64)
        ess = lambda l: (lambda e: e.sum() ** 2 / (e * e).sum())(np.exp(-(c - c.min()) / l))  # noqa: E731
        want = brentq(lambda l: ess(l) - 0.1 * N, 0.01, 10.0, xtol=1e-13)
        assert abs(a_s._last_lambda - want) <= 1e-5 * want, (k, a_s._last_lambda, want)
        x = sb[0, 1].clone()
    assert passes[0] == 2 and passes[4] == 2 and passes[5] == 2 and 1 in passes, passes


# ------------------------------------------------------------------------------ device-resident racing tick
def _device_window(solver, ctrl, env, T, state, cind):
    """calc_ref_trajectory through the library: (reference_path [T+1,4], path index)."""
    solver.path_index = int(cind)
    solver.update_reference_window(torch.as_tensor(np.asarray(state, np.float32)).cuda())
    return solver.reference_window().cpu().numpy(), solver.path_index


def test_device_reference_window_matches_fixtures_and_host():
    """mppi_ref_window == racing_controller.calc_ref_trajectory (example/racing.py:161-218), bit for bit: the
    reference's own (state, cind) -> (window, index) pairs, random vehicle positions (nearest-point search: first
    minimum of the fp32 hypot), the monotone index guard, and the end of the course (last point repeated, whole speed
    column zeroed)."""
    env_np = None
    for name in ("racing_T50_N512_fixed", "racing_T25_N256_fixed"):
        g, T = load(name), CASES[name]["T"]
        solver, ctrl = make_solver("racing", T, 256)
        env = _envs["racing"]
        env_np = env.racing_center_path.cpu().numpy()
        solver.set_center_path(env_np, ctrl._window_offsets(T, 0.1, 3, 0.85), ctrl._v_max())
        for k in range(3):
            ref, ind = _device_window(solver, ctrl, env, T, g[f"x0_{k}"], int(g[f"cind_in_{k}"]))
            assert np.array_equal(ref, g[f"ref_path_{k}"]) and ind == int(g[f"cind_out_{k}"])
    rng = np.random.default_rng(11)
    n = len(env_np)
    for trial in range(60):
        j = int(rng.integers(0, n))
        s = np.zeros(4, np.float32)
        s[:2] = env_np[j, :2] + rng.standard_normal(2).astype(np.float32) * (0.05 if trial % 3 else 6.0)
        cind = int(rng.integers(0, n)) if trial % 4 == 0 else 0
        ref_h, ind_h = ctrl.calc_ref_trajectory(torch.from_numpy(s), env.racing_center_path, cind, T, DL=0.1,
                                                lookahead_distance=3, reference_path_interval=0.85)
        ref_d, ind_d = _device_window(solver, ctrl, env, T, s, cind)
        assert ind_d == ind_h and np.array_equal(ref_d, ref_h.numpy()), (trial, ind_d, ind_h)
    # exact ties: a state ON a centre-line point and the midpoint of two neighbours
    for s2 in (env_np[100, :2], 0.5 * (env_np[200, :2] + env_np[201, :2])):
        s = np.array([s2[0], s2[1], 0, 0], np.float32)
        ref_h, ind_h = ctrl.calc_ref_trajectory(torch.from_numpy(s), env.racing_center_path, 0, T, DL=0.1,
                                                lookahead_distance=3, reference_path_interval=0.85)
        ref_d, ind_d = _device_window(solver, ctrl, env, T, s, 0)
        assert ind_d == ind_h and np.array_equal(ref_d, ref_h.numpy())
    # past the end of the course
    s = np.concatenate([env_np[n - 5, :3], [0.0]]).astype(np.float32)
    ref_h, ind_h = ctrl.calc_ref_trajectory(torch.from_numpy(s), env.racing_center_path, n - 5, T, DL=0.1,
                                            lookahead_distance=3, reference_path_interval=0.85)
    ref_d, ind_d = _device_window(solver, ctrl, env, T, s, n - 5)
    assert ind_d == ind_h and np.array_equal(ref_d, ref_h.numpy()) and float(np.abs(ref_d[:, 3]).max()) == 0.0


def test_device_tick_equals_host_tick_in_a_closed_loop():
    """The racing control loop (example/racing.py:221-266) with the tick resident on the device — reference window by
    mppi_ref_window from the state in HBM, solve, plant step by mppi_model_step — against the host statement of the
    same loop fed the same states: windows, path indices, costs and actions bit-identical every tick; the native plant
    step within 1e-6 of the env's torch dynamics (src/envs/racing_env.py:142-163)."""
    from envs.racing_controller import racing_controller
    from envs.racing_env import RacingEnv

    T, N, ticks = 50, 4096, 40
    env = RacingEnv()
    env_t = RacingEnv(native_step=False)
    ctrls = []
    for device_tick in (True, False):
        c = racing_controller(env, horizon=T, num_samples=N, lambda_=1.0)
        c.set_cost_map(env._obstacle_map, env._lane_map)
        c.device_tick = device_tick
        ctrls.append(c)
    cd, ch = ctrls
    state = env.reset().clone()
    for k in range(ticks):
        a_d, s_d = cd.update(state, env.racing_center_path)
        a_h, s_h = ch.update(state, env.racing_center_path)
        assert cd._window_on_device and not ch._window_on_device
        assert np.array_equal(cd.reference_path.cpu().numpy(), ch.reference_path.numpy()), k
        assert cd.current_path_index == ch.current_path_index
        assert torch.equal(cd.solver._costs, ch.solver._costs), k
        assert torch.equal(a_d, a_h) and torch.equal(s_d, s_h), k
        env._robot_state = state.clone()
        nxt, reached = env.step(a_d[0, :])          # one native launch
        env_t._robot_state = state.clone()
        nxt_t, reached_t = env_t.step(a_d[0, :])    # the torch ops of the plugin
        assert rel_err(nxt.cpu().numpy(), nxt_t.cpu().numpy()) <= 1e-6
        assert bool(reached) == bool(reached_t)
        # the solver's own prediction of that step (batch-1 rollout, fast math) agrees to the parity tolerance
        assert rel_err(s_d[0, 1].cpu().numpy(), nxt.cpu().numpy()) <= 1e-5
        state = nxt
    assert cd.current_path_index > 0 and float(state[3]) > 1.0  # the vehicle actually moved along the course
    # switching a controller between the two ticks hands the path index over
    ind = cd.current_path_index
    cd.device_tick = False
    cd.update(state, env.racing_center_path)
    assert not cd._window_on_device and cd.current_path_index >= ind
    cd.device_tick = True
    cd.update(state, env.racing_center_path)
    assert cd._window_on_device and cd.current_path_index >= ind


def test_native_env_step_matches_torch_dynamics_nav2d():
    from envs.navigation_2d import Navigation2DEnv

    e1, e2 = Navigation2DEnv(), Navigation2DEnv(native_step=False)
    rng = np.random.default_rng(5)
    for _ in range(25):
        u = torch.tensor(rng.uniform([-0.5, -1.5], [2.5, 1.5]).astype(np.float32)).cuda()
        s1, r1 = e1.step(u)
        s2, r2 = e2.step(u)
        assert rel_err(s1.cpu().numpy(), s2.cpu().numpy()) <= 1e-6 and bool(r1) == bool(r2)
        e2._robot_state = s1.clone()


# ------------------------------------------------------------------------------ uneven shards
def _uneven_worker(rank, world, port, q):
    import os

    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["MPPI_EXCHANGE"] = "nccl"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import mppi_playground_amd  # noqa: F401

        out = [rank]
        xn = torch.tensor([-9.0, -9.0, 0.785])
        for stats in ("device", "host"):
            nav, _ = make_solver("nav2d", 30, 1001, lambda_="ESSPS", exploration=0.25, shard_samples=True,
                                 auto_lambda_stats=stats)
            assert nav._local_samples == (334 if rank < 2 else 333) and nav._sample_offset == (0, 334, 668)[rank]
            nav.forward(xn)
            a, s = nav.forward(xn)
            out += [a.cpu().numpy(), s.cpu().numpy(), nav._last_lambda]
        # opaque callables, sharded: get_top_samples with k beyond what a rank owns gathers the winners' stored rows
        from envs import classic_control as cc
        from pi_mpc.mppi import MPPI

        gen = MPPI(15, 1001, 2, 1, _untagged(cc.pendulum_dynamics), _untagged(cc.pendulum_cost), torch.tensor([-2.0]),
                   torch.tensor([2.0]), torch.tensor([1.0]), 5.0, shard_samples=True)
        gen.forward(torch.tensor([3.0, 0.0]))
        ts, tw = gen.get_top_samples(400)
        out += [ts.cpu().numpy(), tw.cpu().numpy()]
        q.put(tuple(out))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_uneven_shards_match_the_unsharded_solver():
    """num_samples that the world size does not divide (the reference accepts any num_samples, mppi.py:96-98): three
    ranks own 334 + 334 + 333 of 1001 nav2d samples; ESSPS from device statistics and from gathered host costs."""
    _need_gpu()
    import socket

    import torch.multiprocessing as mp

    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_uneven_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(3)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    nav, _ = make_solver("nav2d", 30, 1001, lambda_="ESSPS", exploration=0.25)
    xn = torch.tensor([-9.0, -9.0, 0.785])
    nav.forward(xn)
    a, s = nav.forward(xn)
    for r in res:
        for base in (1, 4):
            assert abs(r[base + 2] - nav._last_lambda) <= 1e-4 * nav._last_lambda
            assert rel_err(r[base], a.cpu().numpy()) < 5e-5 and rel_err(r[base + 1], s.cpu().numpy()) < 5e-5
        assert np.array_equal(r[1], res[0][1])  # the ranks agree exactly
    from envs import classic_control as cc
    from pi_mpc.mppi import MPPI

    gen = MPPI(15, 1001, 2, 1, _untagged(cc.pendulum_dynamics), _untagged(cc.pendulum_cost), torch.tensor([-2.0]),
               torch.tensor([2.0]), torch.tensor([1.0]), 5.0)
    gen.forward(torch.tensor([3.0, 0.0]))
    ts, tw = gen.get_top_samples(400)
    for r in res:
        assert rel_err(r[8], tw.cpu().numpy()) < 1e-5 and r[7].shape == (400, 16, 2)
        distinct = np.abs(np.diff(tw.cpu().numpy())) > 1e-7 * tw.cpu().numpy()[:-1]  # (ties may come in any order)
        keep = np.concatenate([[True], distinct]) & np.concatenate([distinct, [True]])
        assert rel_err(r[7][keep], ts.cpu().numpy()[keep]) < 1e-5
        assert np.array_equal(r[7], res[0][7])


# ------------------------------------------------------------------------------ the single-launch solve
@pytest.mark.parametrize("model,T,N,lam,kw", [
    ("pendulum", 50, 1000, "ESSPS", {}), ("pendulum", 15, 256, "LBPS", {}), ("pendulum", 15, 1000, "MPO", {}),
    ("racing", 25, 4000, 1.0, {}), ("racing", 50, 5000, 300.0, dict(exploration=0.2, use_sg_filter=True)),
    ("nav2d", 50, 65536, "ESSPS", {}), ("nav2d", 30, 3000, "LBPS", dict(use_sg_filter=True, sg_window_size=7, sg_poly_order=2)),
    ("cartpole", 64, 262144, "ESSPS", dict(use_sg_filter=True)), ("mountaincar", 100, 1025, 0.1, {}),
    ("goalzone", 30, 3000, 1.0, {}), ("mjcartpole", 50, 1000, 1.0, {}), ("cartpole", 10, 100, 0.001, {})])
def test_single_launch_solve_equals_the_multi_kernel_path(model, T, N, lam, kw):
    """mppi_solve as ONE cooperative kernel (solve_fused_kernel, the default whenever the problem is resident at once)
    against the same solve as separate launches (option fused_solve = 0), closed loop over the warm start: costs,
    minimum and the searched temperature bit-identical, action and state sequences equal to the rounding of the two
    summation orders; the queries that read the solve's state afterwards (top samples, weights) agree as well."""
    if lam == "LBPS":
        kw = dict(kw, lbps_search="grid")  # (the single launch searches on grids; the default — Brent — is a kernel of its own)
    fused, cf = make_solver(model, T, N, lambda_=lam, **kw)
    fused.set_option("fused_solve", 2)  # (1, the default, takes the single launch up to 4096 samples only)
    multi, cm = make_solver(model, T, N, lambda_=lam, **kw)
    multi.set_option("fused_solve", 0)
    assert fused._one_call and multi._one_call
    x0 = {"pendulum": [3.0, 0.0], "racing": None, "nav2d": [-9.0, -9.0, 0.785], "cartpole": [0.01, 0.0, 0.02, 0.0],
          "mountaincar": [-0.5, 0.0], "mjcartpole": [0.0, 0.0, 0.05, 0.0],
          "goalzone": None}[model]
    if model == "racing":
        env = _envs["racing"]
        state = env.reset().clone()
        ref, _ = cf.calc_ref_trajectory(state, env.racing_center_path, 0, T, DL=0.1, lookahead_distance=3,
                                        reference_path_interval=0.85)
        cf.set_reference(ref)
        cm.set_reference(ref)
    elif model == "goalzone":
        from helpers import goalzone_env_fixture
        state = torch.from_numpy(np.asarray(goalzone_env_fixture()["x0"], np.float32)).cuda()
    else:
        state = torch.tensor(x0).cuda()
    for k in range(4):
        if k:  # both solve from the SAME warm start / filter history every tick (the fused solver's), so that each tick
            # compares the two paths on identical inputs instead of two diverging closed loops
            multi.set_warm_start(fused._previous_action_seq.cpu().numpy(),
                                 fused._actions_history_for_sg if kw.get("use_sg_filter") else None)
        a1, s1 = fused.forward(state)
        a2, s2 = multi.forward(state)
        assert not fused._h.lib.mppi_fused_error(fused._h.h)
        # the rollout is the same arithmetic; the sums behind the temperature and the action run over another partition
        assert torch.equal(fused._costs, multi._costs)
        assert fused.last_stats()["cmin"] == multi.last_stats()["cmin"]
        if lam == "LBPS":  # (flat to fp32 rounding around its minimum: see same_lbps_minimum)
            assert same_lbps_minimum(fused._costs.cpu().numpy(), fused._last_lambda, multi._last_lambda)
        else:
            # (ESSPS from the second tick on: each search starts from the grid around its own previous root)
            assert abs(fused._last_lambda - multi._last_lambda) <= (3e-6 if lam == "ESSPS" and k else 1e-6) * multi._last_lambda, \
                (k, fused._last_lambda, multi._last_lambda)
        dl = abs(fused._last_lambda - multi._last_lambda) / multi._last_lambda
        tol = 2e-6 + 20 * dl
        check_rel("single_launch_action_seq_vs_multi_kernel", a1.cpu().numpy(), a2.cpu().numpy(), tol)
        check_rel("single_launch_state_seq_vs_multi_kernel", s1.cpu().numpy(), s2.cpu().numpy(), tol if model != "mjcartpole" else 50 * tol)
        assert abs(fused.last_stats()["ess"] - multi.last_stats()["ess"]) <= (1e-4 + 20 * dl) * multi.last_stats()["ess"]
    kq = min(N, 50)
    ts1, tw1 = fused.get_top_samples(kq)
    ts2, tw2 = multi.get_top_samples(kq)
    assert rel_err(tw1.cpu().numpy(), tw2.cpu().numpy()) < 1e-4 and ts1.shape == ts2.shape
    check_rel("weights_vs_oracle", fused._weights.cpu().numpy(), orc.softmax_weights(fused._costs.cpu().numpy(), fused._last_lambda)[0], TOL)


@pytest.mark.parametrize("model,T,N,expl", [("pendulum", 1, 1, 0.0), ("pendulum", 1, 5, 0.0), ("pendulum", 2, 64, 0.5),
                                            ("pendulum", 7, 65, 1.0), ("racing", 1, 3, 0.0), ("racing", 3, 130, 0.3),
                                            ("nav2d", 2, 1, 0.0), ("nav2d", 30, 255, 0.1), ("nav2d", 64, 257, 0.0),
                                            ("cartpole", 64, 1023, 0.0), ("mountaincar", 128, 4096, 0.0),
                                            ("goalzone", 5, 1025, 0.2)])
def test_single_launch_solve_at_edge_sizes_against_oracle(model, T, N, expl):
    """The default path of small problems (solve_fused_kernel, one cooperative launch) at ragged sizes — one sample,
    one step, sizes around the 64-trajectory tiles and the 256-per-block split, the widest row it takes (T*dc = 128), an
    exploration split through a block — against the ORACLE on the device-drawn noise: costs, action and state sequence."""
    lam = 0.7
    solver, ctrl = make_solver(model, T, N, lambda_=lam, exploration=expl)
    assert solver._one_call
    if model == "racing":
        env = _envs["racing"]
        x0 = env.reset().clone()
        ref, _ = ctrl.calc_ref_trajectory(x0, env.racing_center_path, 0, T, DL=0.1, lookahead_distance=3,
                                          reference_path_interval=0.85)
        ctrl.set_reference(ref)
        ref_np = ref.numpy()
    else:
        from helpers import goalzone_env_fixture
        x0 = {"pendulum": [3.0, 0.1], "nav2d": [-9.0, -9.0, 0.785], "cartpole": [0.01, 0.0, 0.02, 0.0],
              "mountaincar": [-0.5, 0.0], "goalzone": list(np.asarray(goalzone_env_fixture()["x0"], np.float32))}[model]
        x0 = torch.tensor(x0, dtype=torch.float32).cuda()
        ref_np = None
    mean = np.zeros((T, solver._dim_control), np.float32)
    P = oracle_problem(model, N, T, expl, ref_path=ref_np)
    for k in range(2):
        a, s = solver.forward(x0)
        assert not solver._h.lib.mppi_fused_error(solver._h.h)
        c_gpu = solver._costs.cpu().numpy()
        eps = solver._action_noises.cpu().numpy()
        r = P.rollout_cost(x0.cpu().numpy(), mean, eps, want_margin=True)
        check_costs(c_gpu, r)
        w, st = orc.softmax_weights(c_gpu, lam)
        check_rel("action_seq_vs_oracle_given_costs", a.cpu().numpy(), P.weighted_actions(w, mean, eps), TOL)
        check_rel("state_seq_vs_oracle_rollout", s.cpu().numpy()[0], P.rollout_single(x0.cpu().numpy(), a.cpu().numpy()), TOL)
        assert abs(solver.last_stats()["ess"] - st["ess"]) <= 1e-4 * st["ess"]
        mean = a.cpu().numpy()  # the warm start of the next solve


def test_one_launch_top_k_on_randomised_cost_vectors():
    """The one-launch get_top_samples (N <= 4096, k <= 1024: value-binned select, compaction, two-level rank sort, re-roll) on
    cost vectors of awkward shapes — one exponent, a range of e^40, 40 distinct values a few ulps apart, plateaus, mixed signs,
    all equal, infinite costs, sorted input, duplicates around the boundary, collision penalties on top of a narrow range — against a host sort of the same costs: the weights
    in order, and without ties the trajectories bit-equal to the index-driven re-roll of the host's order.  (A short form of
    scripts/topk_soak.py, whose 4 000 cases are recorded in profiles/r05_topk_soak.txt.)"""
    _need_gpu()
    rng = np.random.default_rng(7)

    def draw(N, kind):
        if kind == 0:
            c = rng.uniform(77e3, 110e3, N)
        elif kind == 1:
            c = np.exp(rng.uniform(-20, 20, N))
        elif kind == 2:
            c = 5.0 + rng.integers(0, 40, N) * np.float32(4.8e-7)
        elif kind == 3:
            c = rng.integers(0, max(2, N // 50), N).astype(np.float64)
        elif kind == 4:
            c = rng.standard_normal(N) * 10.0 ** rng.integers(-3, 6)
        elif kind == 5:
            c = np.full(N, float(rng.uniform(-5, 5)))
        elif kind == 6:
            c = rng.uniform(0, 100, N)
            c[rng.random(N) < 0.2] = np.inf
        elif kind == 7:
            c = np.sort(rng.uniform(0, 1e4, N))[:: (1 if rng.random() < 0.5 else -1)]
        elif kind == 8:
            c = rng.uniform(0, 1, N)
            c[rng.integers(0, N, max(1, N // 8))] = c[rng.integers(0, N)]
        else:  # a running racing loop: collision penalties of 10^4 per step on top of a few thousand
            c = rng.uniform(300, 3000, N) + 1e4 * rng.integers(0, 25, N) * (rng.random(N) < 0.4)
        return np.ascontiguousarray(c, dtype=np.float32)

    x0 = torch.tensor([1.0, 0.0])
    for N in (4000, 1024, 1025, 4096, 2731, 77):
        solver, _ = make_solver("pendulum", 10, N, lambda_=1.0)
        solver.forward(x0)
        st = solver._stream()
        for kind in range(10):
            for k in sorted({1, min(N, 64), min(N, 300), min(N, 1024), int(rng.integers(1, min(N, 1024) + 1))}):
                costs = draw(N, kind)
                fin = costs[np.isfinite(costs)]
                lam = float(max(1e-3, np.ptp(fin))) if fin.size else 1.0
                c = torch.from_numpy(costs).cuda()
                solver._h.call("mppi_set_costs", c.data_ptr(), 1, st)
                solver._h.call("mppi_weights_reduce", lam, None, st)
                a = torch.empty(10, 1, device="cuda")
                solver._h.call("mppi_finalize", None, 1, lam, 0, a.data_ptr(), None, None, st)
                out, w = torch.empty(k, 11, 2, device="cuda"), torch.empty(k, device="cuda")
                solver._h.call("mppi_top_samples", k, lam, out.data_ptr(), w.data_ptr(), st)
                order = np.lexsort((np.arange(N), costs))[:k]
                x = (-costs) / np.float32(lam)  # (fp32 quotients like the device's)
                ref = np.exp((x - x.max()).astype(np.float64))
                ref /= ref.sum()
                got = w.cpu().numpy()
                assert np.all(np.isfinite(got)) and np.abs(got - ref[order]).max() <= 2e-5 * ref.max(), (N, kind, k)
                if len(np.unique(costs[order])) == k and (k == N or costs[order][-1] < np.partition(costs, k)[k]):
                    out2 = torch.empty_like(out)
                    idx = torch.from_numpy(order.astype(np.int64)).cuda()
                    solver._h.call("mppi_rollout_samples", idx.data_ptr(), k, out2.data_ptr(), st)
                    assert torch.equal(out, out2), (N, kind, k)


def test_row_pool_steps_aside_under_stream_capture():
    """mppi_playground_amd/_pool.py: a pooled row is ordinary memory that goes back to the allocator when its block's rows
    are dropped, so it must never be baked into a graph — under capture `take` is a plain torch.empty (the graph's private
    pool), and env.collision_check captured in a caller's graph replays correctly after the eager outputs are gone."""
    _need_gpu()
    from mppi_playground_amd._pool import RowPool

    dev = torch.device("cuda", torch.cuda.current_device())
    pool = RowPool((8,), dev, torch.float32)
    raw = torch._C._cuda_getCurrentRawStream(dev.index)
    eager = pool.take(raw)
    assert eager.untyped_storage().nbytes() > 32  # a row of a block
    make_solver("racing", 5, 64)
    env = _envs["racing"]
    pts = torch.zeros(1, 26, 4, device=dev)
    pts[0, :, 0] = torch.linspace(-3.0, 3.0, 26, device=dev)
    want = env.collision_check(state=pts).clone()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        inside = pool.take(torch._C._cuda_getCurrentRawStream(dev.index))
        got = env.collision_check(state=pts)
    assert inside.untyped_storage().nbytes() == 32 and got.untyped_storage().nbytes() == got.numel() * 4
    junk = [env.collision_check(state=pts) for _ in range(600)]  # eager rows come and go around the replays
    del junk
    pts[0, :, 1] = 0.5
    want2 = env.collision_check(state=pts).clone()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(got, want2) and want.shape == want2.shape
