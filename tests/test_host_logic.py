"""Host-side logic that needs no GPU: temperature tuning + SG filter against the reference fixtures,
the env set-up code against the reference's maps, reference-window selection, the C-ABI symbol table,
and the loud failure without a device."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from helpers import (CASES, ROOT, load, nav2d_env_fixture, oracle_problem, orc, racing_env_fixture, rel_err,
                     same_lbps_minimum)
from pi_mpc import _host


@pytest.mark.parametrize("name", ["pendulum_T50_N1000_essps", "nav2d_T50_N512_essps", "cartpole_T64_N1024_essps_sg"])
def test_essps_lambda(name):
    g, cfg = load(name), CASES[name]
    for k in range(3):
        lam = _host.essps_lambda(g[f"costs_{k}"], cfg["N"] / 10, 0.01, 10.0)
        assert abs(lam - float(g[f"lambda_{k}"])) <= 1e-5 * float(g[f"lambda_{k}"])


@pytest.mark.parametrize("name", ["pendulum_T15_N256_lbps", "nav2d_T30_N512_lbps"])
def test_lbps_lambda(name):
    # the bounded scalar minimiser stops at xatol=1e-5 on a flat, fp32-noisy objective
    g = load(name)
    for k in range(3):
        lam = _host.lbps_lambda(g[f"costs_{k}"], 0.01, 0.01, 10.0)
        assert same_lbps_minimum(g[f"costs_{k}"], lam, float(g[f"lambda_{k}"]))


@pytest.mark.parametrize("name", ["pendulum_T15_N256_mpo", "nav2d_T30_N512_mpo"])
def test_mpo_temperature(name):
    # the reference's fp32 autograd gradient cancels ~1000 against ~1000 (|g| ~ 1); with its two fp32 roundings of the
    # log-sum-exp reproduced (see MpoTemperature) three chained Adam steps agree to ~1e-5
    g = load(name)
    m = _host.MpoTemperature()
    for k in range(3):
        lam = m.step(g[f"costs_{k}"])
        assert abs(lam - float(g[f"lambda_{k}"])) <= 3e-5 * float(g[f"lambda_{k}"])


def _np_stats(costs):
    """numpy stand-in of mppi_softmax_stats (float64 sums)."""
    def stats(lam):
        x = (-costs.astype(np.float32)) / np.float32(lam)
        e = np.exp((x - x.max()).astype(np.float32)).astype(np.float64)
        return dict(cmin=float(costs.min()), cmax=float(costs.max()), se=e.sum(), se2=(e * e).sum(),
                    sec=(e * costs).sum())
    return stats


def test_stats_driven_searches_match_reference():
    """The device-statistics form of the three temperature searches gives the reference's lambdas."""
    for name in ("pendulum_T50_N1000_essps", "nav2d_T50_N512_essps", "cartpole_T64_N1024_essps_sg"):
        g, cfg = load(name), CASES[name]
        for k in range(3):
            lam = _host.essps_lambda_stats(_np_stats(g[f"costs_{k}"]), cfg["N"] / 10, 0.01, 10.0)
            assert abs(lam - float(g[f"lambda_{k}"])) <= 1e-5 * float(g[f"lambda_{k}"])
            multi = lambda lams, c=g[f"costs_{k}"]: np.array([_host.ess_from_stats(_np_stats(c)(l)) for l in lams])  # noqa: E731
            lam_g = _host.essps_lambda_grid(multi, cfg["N"] / 10, 0.01, 10.0)
            assert abs(lam_g - float(g[f"lambda_{k}"])) <= 1e-5 * float(g[f"lambda_{k}"])
    g = load("pendulum_T15_N256_lbps")
    for k in range(3):
        lam = _host.lbps_lambda_stats(_np_stats(g[f"costs_{k}"]), 0.01, 0.01, 10.0)
        assert abs(lam - float(g[f"lambda_{k}"])) <= 1e-3 * float(g[f"lambda_{k}"])
    g = load("pendulum_T15_N256_mpo")
    m = _host.MpoTemperature()
    for k in range(3):
        lam = m.step_from_stats(_np_stats(g[f"costs_{k}"])(m.temperature()))
        assert abs(lam - float(g[f"lambda_{k}"])) <= 2e-4 * float(g[f"lambda_{k}"])


def test_savitzky_golay():
    co = _host.savitzky_golay_coeffs(5, 3)
    assert np.allclose(co * 35, [-3, 12, 17, 12, -3], atol=1e-5)
    with pytest.raises(ValueError):
        _host.savitzky_golay_coeffs(4, 3)
    with pytest.raises(ValueError):
        _host.savitzky_golay_coeffs(3, 3)
    g = load("cartpole_T64_N1024_essps_sg")
    P = oracle_problem("cartpole", 1024, 64)
    for k in range(3):
        a = P.weighted_actions(g[f"weights_{k}"], g[f"mean_in_{k}"], g[f"eps_{k}"])
        f = _host.sg_filter_sequence(g[f"sg_hist_in_{k}"], a, co)
        assert rel_err(f, g[f"action_seq_{k}"]) < 1e-5
        if k + 1 < 3:  # history shift-in of the applied first action (mppi.py:455-458)
            hist = np.concatenate([g[f"sg_hist_in_{k}"][1:], f[:1]])
            assert rel_err(hist, g[f"sg_hist_in_{k + 1}"]) < 1e-5 or np.abs(hist).max() < 1e-6


@pytest.fixture(scope="module")
def racing_env():
    from envs.racing_env import RacingEnv

    return RacingEnv(device="cpu")


def test_racing_env_matches_reference_maps(racing_env):
    e = racing_env_fixture()
    assert np.array_equal(racing_env._obstacle_map.grid_spec().cells, e["obst"])
    assert np.array_equal(racing_env._lane_map.grid_spec().cells, e["lane"])
    assert np.array_equal(racing_env.racing_center_path.numpy(), e["center_path"])
    assert np.array_equal(racing_env._robot_state.numpy(), e["start_state"])
    assert racing_env._obstacle_map.x_lim == [-40.0, 40.0]
    c = np.array([[c[0][0], c[0][1], c[1]] for c in racing_env._obstacle_map.circle_obs_list])
    assert np.array_equal(c, e["circles"])


def test_nav2d_env_matches_reference_map():
    from envs.navigation_2d import Navigation2DEnv

    env, n = Navigation2DEnv(device="cpu"), nav2d_env_fixture()
    assert np.array_equal(env._obstacle_map.grid_spec().cells, n["map"])
    assert np.array_equal(env._robot_state.numpy(), n["start_state"])
    assert np.array_equal(env._goal_pos.numpy(), n["goal"])


def test_calc_ref_trajectory(racing_env):
    from envs.racing_controller import racing_controller

    ctrl = racing_controller.__new__(racing_controller)
    ctrl.env = racing_env
    for name in ("racing_T50_N512_fixed", "racing_T25_N256_fixed"):
        g, T = load(name), CASES[name]["T"]
        for k in range(3):
            ref, ind = ctrl.calc_ref_trajectory(torch.tensor(g[f"x0_{k}"]), racing_env.racing_center_path,
                                                int(g[f"cind_in_{k}"]), T, DL=0.1, lookahead_distance=3,
                                                reference_path_interval=0.85)
            assert np.array_equal(ref.numpy(), g[f"ref_path_{k}"]) and ind == int(g[f"cind_out_{k}"])
    # past the end of the course the whole target-velocity column drops to zero
    n = len(racing_env.racing_center_path)
    ref, _ = ctrl.calc_ref_trajectory(racing_env.racing_center_path[n - 5, :3].clone().repeat(2)[:4],
                                      racing_env.racing_center_path, n - 5, 10, DL=0.1, lookahead_distance=3,
                                      reference_path_interval=0.85)
    assert float(ref[:, 3].abs().max()) == 0.0


def test_torch_plugins_match_oracle_states(racing_env):
    """The torch callables of the shipped plugins (used by env.step and any torch caller) follow the
    same math as the oracle."""
    from envs import classic_control as cc

    rng = np.random.default_rng(3)
    for model, dyn in (("pendulum", cc.pendulum_dynamics), ("cartpole", cc.cartpole_dynamics)):
        P = oracle_problem(model, 1, 1)
        s = rng.standard_normal(P.ds).astype(np.float32) * 0.1
        u = rng.standard_normal((1, 1)).astype(np.float32)
        ours = dyn(torch.tensor(s[None]), torch.tensor(u)).numpy()[0]
        assert rel_err(ours, P.rollout_single(s, u)[1]) < 1e-6
    s = np.array([-15.7, -23.9, 2.1, 3.0], np.float32)
    u = np.array([[1.0, 0.1]], np.float32)
    P = oracle_problem("racing", 1, 1)
    ours = racing_env.dynamics(torch.tensor(s[None]), torch.tensor(u)).numpy()[0]
    assert rel_err(ours, P.rollout_single(s, u)[1]) < 1e-6


def test_goalzone_and_mjcartpole_plugins_match_oracle():
    from envs import classic_control as cc
    from envs.goal_in_danger_zone import GoalInDangerZoneEnv
    from helpers import goalzone_env_fixture

    e = goalzone_env_fixture()
    env = GoalInDangerZoneEnv()
    env._goal = e["goal"]
    P = oracle_problem("goalzone", 1, 1)
    u = np.array([[0.7, -0.4]], np.float32)
    ours = env.parallel_step(torch.tensor(e["x0"][None]), torch.tensor(u)).numpy()[0]
    assert rel_err(ours, P.rollout_single(e["x0"], u)[1]) < 1e-6
    c = env.parallel_cost(torch.tensor(e["x0"][None]), torch.tensor(u), {}).numpy()
    r = P.rollout_cost(e["x0"], np.zeros((1, 2), np.float32), u[None], want_stage=True)
    assert abs(c[0] - r["stage"][0, 0]) < 1e-4
    P = oracle_problem("mjcartpole", 1, 1)
    s = np.array([0.01, 0.1, 0.05, -0.2], np.float32)
    u = np.array([[1.3]], np.float32)
    ours = cc.mjcartpole_dynamics(torch.tensor(s[None]), torch.tensor(u)).numpy()[0]
    assert rel_err(ours, P.rollout_single(s, u)[1]) < 1e-6


def test_capi_exports_every_declared_symbol():
    """The built library loads without a GPU and exports exactly what include/mppi_hip.h declares."""
    from mppi_playground_amd import _build, _capi

    _build.build()
    hdr = open(os.path.join(ROOT, "include", "mppi_hip.h")).read()
    declared = set(re.findall(r"\b(mppi_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_capi.SYMBOLS)
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in _capi.load().mppi_version()


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_fails_loudly_without_gpu():
    from envs.classic_control import pendulum_cost, pendulum_dynamics
    from mppi_playground_amd import _capi
    from pi_mpc.mppi import MPPI

    with pytest.raises(_capi.MppiError):
        MPPI(15, 100, 2, 1, pendulum_dynamics, pendulum_cost, torch.tensor([-2.0]), torch.tensor([2.0]),
             torch.tensor([1.0]), 1.0)


# ------------------------------------------------------------------ map recipes (what the device rasterises)
def _recipe_raster(nx, ny, recipe):
    """numpy statement of the integer rules of raster_obstacles_kernel / lane_map_kernel (mppi_kernels.hpp)."""
    ix, iy = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    if recipe["kind"] == "lane":
        best = np.full((nx, ny), np.iinfo(np.int64).max)
        for sx, sy in recipe["seeds"].astype(np.int64):
            best = np.minimum(best, (ix - sx) ** 2 + (iy - sy) ** 2)
        return (best > recipe["max_d2"]).astype(np.uint8)
    big = 1 << 40
    occ = np.zeros((nx, ny), bool)
    for ci, cj, r in recipe["circles"].astype(np.int64):
        lo_i, hi_i = np.where(ix == 0, -big, ix - ci), np.where(ix == nx - 1, big, ix - ci)
        lo_j, hi_j = np.where(iy == 0, -big, iy - cj), np.where(iy == ny - 1, big, iy - cj)
        i, j = np.minimum(np.maximum(0, lo_i), hi_i), np.minimum(np.maximum(0, lo_j), hi_j)
        occ |= i * i + j * j <= r * r
    for x0, x1, y0, y1 in recipe["rects"]:
        occ |= (ix >= x0) & (ix < x1) & (iy >= y0) & (iy < y1)
    return occ.astype(np.uint8)


def test_map_recipes_reproduce_reference_maps(racing_env):
    from envs.navigation_2d import Navigation2DEnv

    e, n = racing_env_fixture(), nav2d_env_fixture()
    spec = racing_env._obstacle_map.grid_spec()
    assert np.array_equal(_recipe_raster(*spec.cells.shape, spec.recipe), e["obst"])
    spec = Navigation2DEnv(device="cpu")._obstacle_map.grid_spec()
    assert len(spec.recipe["rects"]) == 7 and len(spec.recipe["circles"]) == 7
    assert np.array_equal(_recipe_raster(*spec.cells.shape, spec.recipe), n["map"])
    # the lane rule on a crop around a stretch of the centre line (the full 800x800x3678 case runs on the GPU)
    lane = racing_env._lane_map.grid_spec()
    seeds, k = lane.recipe["seeds"], lane.recipe["max_d2"]
    x0, y0 = seeds[:, 0].min(), int(np.median(seeds[:, 1]))
    sl = (slice(x0, x0 + 90), slice(y0 - 60, y0 + 60))
    near = seeds[(seeds[:, 0] < x0 + 90 + 40) & (np.abs(seeds[:, 1] - y0) < 100)]
    crop = _recipe_raster(800, 800, {"kind": "lane", "seeds": near, "max_d2": k})[sl]
    assert np.array_equal(crop, e["lane"][sl])


def test_map_recipes_border_clipping_and_threshold():
    """Discs and rectangles that stick out of the grid are clipped onto the border cells by the reference's
    loops (obstacle_map_2d.py:118-123,146-156); the recipe rule must pile them up the same way."""
    from envs.lane_map_2d import LaneMap, largest_square_within
    from envs.obstacle_map_2d import ObstacleMap

    circles = [(np.array([-0.95, 0.2]), 0.3), (np.array([0.9, -0.97]), 0.25), (np.array([1.4, 1.4]), 0.7),
               (np.array([0.0, 0.0]), 0.04), (np.array([-3.0, 0.5]), 0.5)]
    rects = [(np.array([0.9, 0.9]), 0.5, 0.3), (np.array([-1.2, -0.3]), 0.6, 0.2), (np.array([0.2, -0.4]), 0.11, 0.33)]
    m = ObstacleMap(map_size=(2, 2), cell_size=0.05, device="cpu")
    for c, r in circles:
        m.add_circle_obstacle(c, r)
    for c, w, h in rects:
        m.add_rectangle_obstacle(c, w, h)
    lit, origin = orc.obstacle_map_literal(40, 40, 0.05, circles, rects)
    spec = m.grid_spec()
    assert np.array_equal(spec.cells, lit) and tuple(origin) == spec.origin
    assert np.array_equal(_recipe_raster(40, 40, spec.recipe), lit)

    for md in (0.0, 0.5, 1.0, 25.999999, 26.0, 26.000001, np.sqrt(2.0), 2.5 ** 0.5, 1e3 + 0.5):
        k = largest_square_within(md)
        assert np.sqrt(np.float64(k)) <= md < np.sqrt(np.float64(k + 1))

    t = np.linspace(0, 2 * np.pi, 40)
    lane = np.stack([0.6 * np.cos(t), 0.5 * np.sin(t) + 0.45, t], axis=1)  # partly outside the 2 x 2 m grid
    lm = LaneMap(lane, lane_width=0.33, map_size=(2, 2), cell_size=0.05, device="cpu")
    lit, _ = orc.lane_map_literal(40, 40, 0.05, lane, 0.33)
    spec = lm.grid_spec()
    assert np.array_equal(spec.cells, lit)
    assert np.array_equal(_recipe_raster(40, 40, spec.recipe), lit)


# ------------------------------------------------------------------ the library's own searches (csrc/host_search.hpp)
def test_library_bounded_minimiser_is_scipys():
    """host::fminbound (what mppi_lbps_lambda runs) against scipy.optimize.minimize_scalar(method="bounded") on the
    same objectives: same minimiser, same number of function evaluations (the same published algorithm, step by
    step), including minima at an end point and objectives with several local minima."""
    import emul
    from scipy.optimize import minimize_scalar

    rng = np.random.default_rng(3)
    for trial in range(40):
        a, b, c = rng.uniform(-2, 12), rng.uniform(0, 0.9), rng.uniform(0.1, 6)
        lo, hi = 0.01, 10.0
        f = lambda x: (x - a) ** 2 * (1.0 + b * np.sin(c * x)) + 0.1 * x  # noqa: E731
        res = minimize_scalar(f, bounds=(lo, hi), method="bounded")
        x, nfev = emul.fminbound_poly(a, b, c, lo, hi)
        assert nfev == res.nfev, (trial, nfev, res.nfev)
        assert abs(x - res.x) <= 1e-12 * max(1.0, abs(res.x)), (trial, x, res.x)


def test_library_searches_match_reference_fixtures():
    """LBPS / ESSPS / MPO as the library computes them (host C++ over softmax statistics) on the reference's own
    cost vectors -> the reference's lambdas (same tolerances as the numpy statements in pi_mpc/_host.py)."""
    import emul

    for name in ("pendulum_T15_N256_lbps", "nav2d_T30_N512_lbps"):
        g = load(name)
        for k in range(3):
            c = g[f"costs_{k}"]
            lam, nfev = emul.lbps(c, 0.01, 0.01, 10.0)
            assert same_lbps_minimum(c, lam, float(g[f"lambda_{k}"]))
            assert same_lbps_minimum(c, lam, _host.lbps_lambda_stats(_np_stats(c), 0.01, 0.01, 10.0))
            assert 5 <= nfev <= 60
            # the device-resident variant (three 32-temperature grids + a parabola, lbps_select_kernel) lands on the
            # reference's minimum too
            lam_grid = emul.lbps_grid(c, 0.01, 0.01, 10.0)
            assert same_lbps_minimum(c, lam_grid, float(g[f"lambda_{k}"])), (name, k, lam_grid, float(g[f"lambda_{k}"]))
    for name in ("pendulum_T50_N1000_essps", "nav2d_T50_N512_essps", "cartpole_T64_N1024_essps_sg",
                 "nav2d_T30_N4096_essps", "racing_T25_N1024_essps"):
        g, cfg = load(name), CASES[name]
        for k in range(int(g["K"])):
            lam = emul.essps(g[f"costs_{k}"], cfg["N"] / 10, 0.01, 10.0)
            assert abs(lam - float(g[f"lambda_{k}"])) <= 1e-5 * float(g[f"lambda_{k}"])
    # end-point rules of mppi.py:361-364
    c = load("pendulum_T50_N1000_essps")["costs_0"]
    assert emul.essps(c, 1.0, 0.01, 10.0) == 0.01 and emul.essps(c, 999.99, 0.01, 10.0) == 10.0
    for name in ("pendulum_T15_N256_mpo", "nav2d_T30_N512_mpo"):
        g = load(name)
        lams = emul.mpo(np.stack([g[f"costs_{k}"] for k in range(3)]))
        m = _host.MpoTemperature()
        for k in range(3):
            assert abs(lams[k] - float(g[f"lambda_{k}"])) <= 3e-5 * float(g[f"lambda_{k}"])
            assert abs(lams[k] - m.step(g[f"costs_{k}"])) <= 2e-6 * lams[k]


def test_bench_launches_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` without a launcher re-runs itself under torch.distributed.run with one rank per GPU
    on 127.0.0.1 (the driver's N=1 command shape must work for N>1 too), with a wall-clock limit on the launcher."""
    import importlib
    import sys

    bench = importlib.import_module("bench")
    seen = {}
    monkeypatch.setattr(bench, "run_with_deadline",
                        lambda cmd, env, timeout_s, on_timeout_line: seen.update(cmd=cmd, env=env, timeout=timeout_s,
                                                                                 line=on_timeout_line()) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "50", "--warmup", "10"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "50", "--warmup", "10"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert 0 < seen["timeout"] < 1800  # below the driver's own limit
    line = seen["line"]
    assert line["value"] is None and line["n_gpus"] == 8 and line["metric"] == "sample_steps_per_sec" and "error" in line


def test_bench_launcher_deadline_leaves_a_line():
    """A launcher that never returns (a rank stuck in a communicator's set-up) is killed at the deadline and a
    contract-shaped line with value null is printed; a launcher that printed its line keeps it."""
    import json
    import subprocess
    import sys
    code = (
        "import sys, json\n"
        f"sys.path.insert(0, {str(ROOT)!r})\n"
        "import bench\n"
        "bench.own_stdout()\n"
        "child = [sys.executable, '-c', sys.argv[1]]\n"
        "rc = bench.run_with_deadline(child, None, float(sys.argv[2]), lambda: {'value': None, 'error': 'deadline'})\n"
        "sys.exit(0 if rc == int(sys.argv[3]) else 1)\n"
    )
    hang = "import time; print('chatter', flush=True); time.sleep(600)"
    r = subprocess.run([sys.executable, "-c", code, hang, "1.0", "124"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"value": None, "error": "deadline"}
    good = "print('{\"value\": 1.0}', flush=True)"
    r = subprocess.run([sys.executable, "-c", code, good, "30", "0"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")] == [{"value": 1.0}]


@pytest.mark.parametrize("hang_at", [0, 1])
def test_bench_watchdog_reports_when_a_transport_blocks_forever(hang_at):
    """The multi-GPU leg of bench.py with a transport that never returns (monkey-patched: the first, then the second): the
    watchdog — armed BEFORE the first transport — must leave a line: a null-value line carrying the phase that hung when
    nothing had completed, the best complete run otherwise; and end the process through its exit hook."""
    import argparse
    import importlib
    import threading

    bench = importlib.import_module("bench")
    args = argparse.Namespace(steps=20, warmup=5, samples=1 << 20, horizon=50)
    out, release, runs = {}, threading.Event(), []

    def timed_run(mode):
        if ["rccl", "nccl", "p2p"].index(mode) == hang_at:
            release.wait(30)  # "blocks forever": only the watchdog's exit hook ends it
            return {"exchange": mode, "requested": mode, "error": "ended by the watchdog"}
        return {"exchange": mode, "requested": mode, "dt": 0.01, "finite": True}

    def on_expiry(phase, seconds):
        best = bench.best_of(runs)
        why = f"{phase} exceeded its {seconds:.1f} s budget"
        out["line"] = ({"value": 1.0, "from": best["exchange"], "transports": runs + [{"exchange": phase, "error": why}]}
                       if best is not None else bench.null_line(args, 8, why, runs + [{"exchange": phase, "error": why}]))
        return 0 if best is not None else 1

    def exit_fn(code):
        out["code"] = code
        release.set()

    dog = bench.Watchdog(on_expiry, exit_fn=exit_fn)
    t = threading.Thread(target=lambda: bench.run_transports(["rccl", "nccl", "p2p"], timed_run, dog, 0.3, 0.3, runs),
                         daemon=True)
    t.start()
    t.join(20)
    assert not t.is_alive() and "line" in out, "the watchdog did not fire"
    line = out["line"]
    if hang_at == 0:
        assert out["code"] == 1 and line["value"] is None and line["n_gpus"] == 8 and line["steps"] == 20
        assert "transport rccl exceeded" in line["error"] and line["transports"][-1]["error"].startswith("transport rccl")
    else:
        assert out["code"] == 0 and line["value"] == 1.0 and line["from"] == "rccl"
        assert line["transports"][-1]["exchange"] == "transport nccl"


def test_bench_runs_every_transport_in_its_own_process(monkeypatch, capsys):
    """`bench.py --gpus N --exchange all` (the default): one child process per transport on every rank (orchestrate_transports).
    With a transport that is killed at its budget and one that reports an error, rank 0 still prints the best complete run with
    all three listed; with none complete it prints a null line and exits 1; other ranks print nothing."""
    import argparse
    import importlib
    import json

    bench = importlib.import_module("bench")
    args = argparse.Namespace(steps=20, warmup=5, samples=1 << 20, horizon=50, first_budget_s=300.0)
    printed = []
    monkeypatch.setattr(bench, "print_line", printed.append)

    def line(mode, value):
        return json.dumps({"metric": "sample_steps_per_sec", "value": value, "ms_per_step": 0.14, "n_gpus": 8,
                           "config": {"exchange_used": mode}, "exchange_us": 3.0, "rccl_ranks": None, "strong": {"value": 1.0},
                           "transports": [{"per_rank_stages_ms": []}]})

    def run_child(mode, index, budget_s):
        assert budget_s == 300.0
        if mode == "rccl":
            return 0, "chatter\n" + line("rccl", 10.0) + "\n", 12.0
        if mode == "nccl":
            return 124, "", 300.0                                       # hung: killed at the budget
        return 1, json.dumps(bench.null_line(args, 8, "MPPI_EXCHANGE=p2p: not usable here")) + "\n", 3.0

    assert bench.orchestrate_transports(args, 8, 0, ["rccl", "nccl", "p2p"], run_child=run_child) == 0
    out = json.loads(printed[-1])
    assert out["value"] == 10.0 and out["config"]["exchange_used"] == "rccl" and len(out["transports"]) == 3
    assert out["transports"][1]["exit_code"] == 124 and "killed at its 300 s budget" in out["transports"][1]["error"]
    assert "not usable" in out["transports"][2]["error"] and out["transports"][0]["isolated_process"] is True
    printed.clear()
    assert bench.orchestrate_transports(args, 8, 3, ["rccl", "nccl", "p2p"], run_child=run_child) == 0 and printed == []
    assert bench.orchestrate_transports(args, 8, 0, ["nccl", "p2p"], run_child=run_child) == 1
    out = json.loads(printed[-1])
    assert out["value"] is None and out["n_gpus"] == 8 and len(out["transports"]) == 2


def test_lbps_grid_search_finds_the_brent_minimum_on_random_costs():
    """The grid variant of the LBPS search against scipy's bounded Brent minimiser (the reference's call) on a float64
    evaluation of the objective, over cost vectors of different shapes; where Brent stops in a local minimum the grid may
    only be better."""
    import emul
    from scipy.optimize import minimize_scalar

    from helpers import lbps_objective64

    rng = np.random.default_rng(9)
    N = 2000
    shapes = [lambda: rng.standard_normal(N) * 3 + 20, lambda: rng.standard_exponential(N) * 5,
              lambda: np.concatenate([rng.standard_normal(N // 2), 8 + rng.standard_normal(N - N // 2)]),
              lambda: 1e4 + rng.standard_normal(N) * 30, lambda: rng.standard_normal(N) * 0.05,
              lambda: np.abs(rng.standard_cauchy(N)) * 2]
    for i, draw in enumerate(shapes):
        c = draw().astype(np.float32)
        res = minimize_scalar(lambda lam: lbps_objective64(c, lam), bounds=(0.01, 10.0), method="bounded")
        lam = emul.lbps_grid(c, 0.01, 0.01, 10.0)
        f, f_ref = lbps_objective64(c, lam), lbps_objective64(c, res.x)
        assert f <= f_ref + 1e-6 * abs(f_ref), (i, lam, res.x, f, f_ref)
        if abs(lam - res.x) > 2e-3 * res.x:  # a different temperature only where the objective is no worse
            assert f <= f_ref + 1e-7 * abs(f_ref), (i, lam, res.x)


def test_essps_warm_start_lands_on_the_same_root():
    """The clustered first grid (host_search.hpp::essps_first_grid, mirrored by _host.essps_first_grid): same points in C++
    and numpy, sorted, end points exact; a search warm-started from anywhere gives the reference's lambda (the fixtures'
    brentq roots) — in ONE pass over the costs when the previous root is within the cluster, in two otherwise."""
    import emul

    for prev in (0.0, 0.02, 0.3, 1.0, 2.5, 6.5, 9.9):
        for lo, hi in ((0.01, 10.0), (1e-3, 1e3), (0.5, 2.0)):
            gc, gp = emul.essps_first_grid(prev, lo, hi), _host.essps_first_grid(prev, lo, hi)[0]
            assert np.allclose(gc, gp, rtol=1e-14, atol=0)
            assert gc[0] == lo and gc[-1] == hi and np.all(np.diff(gc) > 0)
            if prev and lo * 1.6 < prev < hi / 1.6:  # warm: the cluster is there, and no gap is wider than ~1/10 of the log-range
                near = (gc > prev / 1.51) & (gc < prev * 1.51)
                assert near.sum() == 22 and np.max(gc[near][1:] / gc[near][:-1]) < 1.05
                assert np.max(np.log(gc[1:] / gc[:-1])) <= np.log(hi / lo) / 8
    one_pass = 0
    for name in ("pendulum_T50_N1000_essps", "nav2d_T50_N512_essps", "cartpole_T64_N1024_essps_sg",
                 "nav2d_T30_N4096_essps", "racing_T25_N1024_essps"):
        g, cfg = load(name), CASES[name]
        for k in range(int(g["K"])):
            c, want = g[f"costs_{k}"], float(g[f"lambda_{k}"])
            multi = lambda lams, c=c: np.array([_host.ess_from_stats(_np_stats(c)(l)) for l in lams])  # noqa: E731
            for f in (1.0, 0.97, 1.2, 0.7, 1.49, 0.1, 3.0, 30.0):
                lam, passes = emul.essps(c, cfg["N"] / 10, 0.01, 10.0, lam_prev=want * f, with_passes=True)
                assert abs(lam - want) <= 1e-5 * want, (name, k, f, lam, want)
                lam_py = _host.essps_lambda_grid(multi, cfg["N"] / 10, 0.01, 10.0, lam_prev=want * f)
                assert abs(lam_py - lam) <= 2e-6 * lam, (name, k, f, lam_py, lam)
                one_pass += passes == 1
                assert passes in (1, 2) and (passes == 2 or lam in (0.01, 10.0) or 1 / 1.5 < f < 1.5), (name, k, f, passes)
    assert one_pass > 20  # (a sharp ESS curve — few samples — may fail the convergence check and take the second grid)


def test_deferred_state_seq_wrapper_completes_on_first_use_only():
    """`_DeferredStateSeq` (what forward() returns as state_seq while its batch-1 rollout is still pending): a tensor whose
    FIRST use through torch — directly, nested in a list, or as a keyword argument — calls the completion hook once;
    results are plain tensors; an untouched wrapper never calls it."""
    from pi_mpc.mppi import _DeferredStateSeq

    calls = []
    base = torch.arange(24.0).reshape(1, 4, 6)
    w = _DeferredStateSeq.wrap(base, lambda t: calls.append("a"))
    assert isinstance(w, torch.Tensor) and calls == []
    row = w[0, 1]
    assert calls == ["a"] and type(row) is torch.Tensor and torch.equal(row, base[0, 1])
    assert type(w + 1) is torch.Tensor and bool(torch.isfinite(w).all()) and calls == ["a"]  # joined once
    w2 = _DeferredStateSeq.wrap(base, lambda t: calls.append("b"))
    assert type(torch.cat([w2, base])) is torch.Tensor and calls == ["a", "b"]
    w3 = _DeferredStateSeq.wrap(base, lambda t: calls.append("c"))
    assert torch.equal(torch.sum(input=w3, dim=1), base.sum(1)) and calls == ["a", "b", "c"]
    w4 = _DeferredStateSeq.wrap(base, lambda t: calls.append("d"))
    del w4
    assert calls == ["a", "b", "c"]
    assert np.array_equal(w.numpy(), base.numpy())  # (CPU tensors only: what a caller does after .cpu())


def test_graph_replay_compares_the_callers_info_by_value():
    """graph_callables: a replayed graph cannot see changes of the caller's `info` entries; the check compares tensors by
    storage (an equal-valued NEW tensor is a change, an in-place update is not) and Python scalars by value (a caller may
    rebuild an equal dict every tick)."""
    from pi_mpc.mppi import MPPI

    t = torch.zeros(3)
    a = MPPI._info_signature({"w": 2.0, "ref": t, "mode": "x", "prev_state": object(), "t": 5})
    assert a == MPPI._info_signature({"w": 2.0, "ref": t, "mode": "x", "t": 7})       # rebuilt dict, the solver's own keys ignored
    t.add_(1.0)
    assert a == MPPI._info_signature({"w": 2.0, "ref": t, "mode": "x"})                # in-place update: same storage
    assert a != MPPI._info_signature({"w": 3.0, "ref": t, "mode": "x"})                # a scalar changed value
    assert a != MPPI._info_signature({"w": 2.0, "ref": t.clone(), "mode": "x"})        # another tensor object / storage
    assert a != MPPI._info_signature({"w": 2.0, "ref": t})                             # an entry disappeared


def test_fixture_bands_are_complete():
    """Every solve fixture carries the reference's own measured spread (tests/golden/make_golden.py): 256 probes per solve
    at the reference's temperature, with its rule re-run where it has one, and per closed loop; racing at lambda = 1 is an
    arg-min (no spread at all), the ill-conditioned cases show the 1e-5 of the north star is below the reference's own
    rounding noise."""
    for name, cfg in CASES.items():
        g = load(name)
        K = int(g["K"])
        assert g["band_closed_loop"].shape == (K, 256, 4)
        for k in range(K):
            assert g[f"band_fixed_{k}"].shape == (256, 2) and np.isfinite(g[f"band_fixed_{k}"]).all()
            assert (f"band_rule_{k}" in g.files) == isinstance(cfg["lambda_"], str)
    assert load("racing_T50_N512_fixed")["band_closed_loop"].max() == 0.0
    assert load("nav2d_T30_N256_fixed_explore")["band_fixed_1"][:, 0].max() > 1e-5
    assert load("nav2d_T30_N512_lbps")["band_rule_0"][:, 2].max() > 1e-3  # the reference's own LBPS temperature under 1-ulp changes


def _undefined_globals(path):
    """Names a module's functions read that are neither local, nor defined at module level, nor builtins (a poor man's
    pyflakes: the GPU-only scripts cannot be exercised by the CPU suite, but a deleted helper must not survive a commit)."""
    import ast
    import builtins

    tree = ast.parse(open(path).read())
    module_names = set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
    for node in ast.walk(tree):
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            module_names.add(node.name)
        elif isinstance(node, (ast.Import, ast.ImportFrom)):
            for a in node.names:
                module_names.add((a.asname or a.name).split(".")[0])
        elif isinstance(node, ast.Name) and isinstance(node.ctx, (ast.Store, ast.Del)):
            module_names.add(node.id)
        elif isinstance(node, ast.arg):
            module_names.add(node.arg)
        elif isinstance(node, ast.ExceptHandler) and node.name:
            module_names.add(node.name)
    loads = {n.id for n in ast.walk(tree) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load)}
    return sorted(loads - module_names)


@pytest.mark.parametrize("rel", ["bench.py", "__graft_entry__.py", "mppi_playground_amd/pi_mpc/mppi.py",
                                 "mppi_playground_amd/pi_mpc/_lazy.py", "mppi_playground_amd/pi_mpc/_exchange.py",
                                 "mppi_playground_amd/pi_mpc/_generic.py", "mppi_playground_amd/pi_mpc/_queries.py",
                                 "mppi_playground_amd/pi_mpc/recognize.py", "scripts/closure_fingerprints.py",
                                 "mppi_playground_amd/_capi.py", "mppi_playground_amd/envs/racing_controller.py",
                                 "mppi_playground_amd/_pool.py", "mppi_playground_amd/envs/common.py",
                                 "mppi_playground_amd/envs/obstacle_map_2d.py", "mppi_playground_amd/envs/lane_map_2d.py",
                                 "mppi_playground_amd/envs/racing_env.py", "mppi_playground_amd/envs/navigation_2d.py",
                                 "scripts/make_visit_docs.py", "scripts/pmc_constants.py", "tests/golden/make_golden.py"])
def test_no_function_reads_an_undefined_name(rel):
    assert _undefined_globals(os.path.join(ROOT, rel)) == []


def test_lazy_info_tensor_is_built_on_first_use():
    """`_LazyInfoTensor` (the reference's `info["prev_state"]` / `info["prev_action"]` on the native path): nothing is
    computed until a torch function touches it; then it behaves like the tensor its thunk returns, built once."""
    from pi_mpc.mppi import _LazyInfoTensor

    built = []

    def thunk():
        built.append(1)
        return torch.arange(6.0).reshape(3, 2)

    t = _LazyInfoTensor.make(thunk)
    assert isinstance(t, torch.Tensor) and built == []
    assert t.shape == (3, 2) and built == [1]
    assert torch.equal(t * 2, torch.arange(6.0).reshape(3, 2) * 2) and float(t.sum()) == 15.0
    assert torch.equal(torch.cat([t, t]), torch.arange(6.0).reshape(3, 2).repeat(2, 1)) and built == [1]
    assert type(t[0]) is torch.Tensor and t[:, 1].tolist() == [1.0, 3.0, 5.0]


def test_bench_stdout_carries_only_its_line():
    """bench.py: whatever a library prints from C (RCCL's version banner, flushed at exit) lands on stderr; stdout is the
    JSON line and nothing else."""
    import subprocess
    import sys
    code = (
        "import ctypes, os, sys\n"
        f"sys.path.insert(0, {str(ROOT)!r})\n"
        "import bench\n"
        "bench.own_stdout()\n"
        "libc = ctypes.CDLL(None)\n"
        "libc.printf(b'banner from C\\n')\n"      # stays in the stdio buffer until exit, like RCCL's
        "print('python chatter')\n"
        "bench.print_line('{\"metric\": 1}')\n"
    )
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout == '{"metric": 1}\n'
    assert "banner from C" in r.stderr and "python chatter" in r.stderr


# ------------------------------------------------------------------------------ the reference examples' own closures
def _reference_example_closures(model):
    """(dynamics, cost) as example/<model>.py holds them — nested definitions inside main(), `dynamics` a TorchScript function,
    the cost a closure over the module-level TorchScript angle_normalize — rebuilt from the reference's file in a scratch
    module (build container only; nothing is copied into the repo)."""
    import ast
    import importlib.util
    import tempfile
    import textwrap

    fn, dyn_name, cost_name = {"pendulum": ("pendulum.py", "dynamics", "cost_function"),
                               "cartpole": ("cartpole.py", "dynamics", "stage_cost"),
                               "mountaincar": ("mountaincar.py", "dynamics", "cost_func"),
                               "mjcartpole": ("mujoco_cartpole.py", "dynamics", "cost_func")}[model]
    tree = ast.parse(open(os.path.join("/root/reference/example", fn)).read())
    body = ["import torch\n"]
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == "angle_normalize":
            body.append(ast.unparse(node) + "\n")
    main = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main"][0]
    inner = [textwrap.indent(textwrap.dedent(ast.unparse(n)), "    ") for n in main.body
             if isinstance(n, ast.FunctionDef) and n.name in (dyn_name, cost_name)]
    body.append("def main():\n" + "\n".join(inner) + f"\n    return {dyn_name}, {cost_name}\n")  # nested, like the example
    d = tempfile.mkdtemp(prefix="mppi_ref_example_")
    path = os.path.join(d, f"ref_example_{model}.py")
    open(path, "w").write("\n".join(body))
    spec = importlib.util.spec_from_file_location(f"ref_example_{model}", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.main()


@pytest.mark.parametrize("model", ["pendulum", "cartpole", "mountaincar", "mjcartpole"])
def test_reference_example_closures_are_recognised(model):
    """pi_mpc/recognize.py on the REAL closures of the reference's classic-control examples (container only): the committed
    fingerprints match what the installed torch prints for them, the shipped torch plugin of that model agrees with them on the
    probe batches (values returned AND — mountain car — what the call leaves in its argument), and an edited copy is not
    recognised."""
    if not os.path.isdir("/root/reference/example"):
        pytest.skip("needs the reference's example files (build container)")
    import torch

    from envs import classic_control as cc
    from pi_mpc import recognize

    dyn, cost = _reference_example_closures(model)
    assert isinstance(dyn, torch.jit.ScriptFunction) and not isinstance(cost, torch.jit.ScriptFunction)
    ds, dc = recognize._MODELS[model][:2]
    twin = recognize.match(dyn, cost, ds, dc, torch.device("cpu"))
    assert twin == (getattr(cc, f"{model}_dynamics"), getattr(cc, f"{model}_cost"))
    assert recognize.match(dyn, cost, ds + 1, dc, torch.device("cpu")) is None      # another problem shape
    assert recognize.match(cost, dyn, ds, dc, torch.device("cpu")) is None          # roles swapped
    edited = lambda s, a, info: cost(s, a, info) * 1.0001                           # noqa: E731  (another function: another fingerprint)
    assert recognize.match(dyn, edited, ds, dc, torch.device("cpu")) is None
    # a callable with the right fingerprint but other values cannot exist by construction; the behaviour test is what
    # protects against a whitelist entry that does not belong to the plugin it names:
    other = "cartpole" if model != "cartpole" else "mjcartpole"
    table = dict(recognize._table())
    try:
        recognize._table_cache = {other: table[model]} if recognize._MODELS[other][:2] == (ds, dc) else {}
        assert recognize.match(dyn, cost, ds, dc, torch.device("cpu")) is None
    finally:
        recognize._table_cache = table


def test_transcribed_pendulum_closure_is_recognised_through_the_version_independent_fingerprints():
    """tests/closure_transcription.py — the pendulum example's closures written anew with other names — against the SHIPPED
    table: only the "g2" forms match (operator sequence + sorted constants of the scripted graph; structure of the cost's AST
    with identifiers numbered), the torch printer's text ("v1") does not; together with the behaviour on the probe batches that
    is a recognition.  The two tests disagreeing warns once: a TorchScript dynamics that behaves like the shipped model but
    is not listed (here: the shipped plugin itself, scripted — other operators), and a listed fingerprint whose callable
    computes something else."""
    import warnings

    import torch

    import closure_transcription as ct
    from envs import classic_control as cc
    from pi_mpc import recognize

    step, cost = ct.build()
    table = recognize._table()
    fd, fc = recognize.fingerprints(step), recognize.fingerprints(cost)
    assert fd & set(table["pendulum"]["dynamics"]) == {table["pendulum"]["dynamics"][1]}   # g2 only
    assert fc & set(table["pendulum"]["cost"]) == {table["pendulum"]["cost"][1]}
    assert recognize.fingerprint(step) not in table["pendulum"]["dynamics"]                  # (v1: the printer's text differs)
    with warnings.catch_warnings():
        warnings.simplefilter("error", recognize.RecognitionWarning)
        assert recognize.match(step, cost, 2, 1, torch.device("cpu")) == (cc.pendulum_dynamics, cc.pendulum_cost)
        assert recognize.last_report["model"] == "pendulum" and recognize.last_report["fingerprint"] and recognize.last_report["behaviour"]

        # another constant: another fingerprint, other values -> simply not one of the examples (no warning)
        @torch.jit.script
        def other_step(x: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
            return torch.cat((x[:, 0:1] + 0.07 * x[:, 1:2], torch.clamp(x[:, 1:2] + u[:, 0:1], -8, 8)), dim=1)

        assert recognize.match(other_step, cost, 2, 1, torch.device("cpu")) is None
    # behaves like the model, not listed: the generic path, and a warning that says what it costs
    recognize._warned.clear()
    scripted_plugin = torch.jit.script(lambda_free(cc.pendulum_dynamics))
    with pytest.warns(recognize.RecognitionWarning, match="GENERIC path"):
        assert recognize.match(scripted_plugin, cost, 2, 1, torch.device("cpu")) is None
    assert recognize.last_report == {**recognize.last_report, "model": "pendulum", "fingerprint": False, "behaviour": True}
    with warnings.catch_warnings():  # ... once per process
        warnings.simplefilter("error", recognize.RecognitionWarning)
        assert recognize.match(scripted_plugin, cost, 2, 1, torch.device("cpu")) is None
    # listed, but other values
    try:
        recognize._table_cache = {"_torch": "x", "pendulum": {"dynamics": sorted(recognize.fingerprints(other_step)), "cost": sorted(fc)}}
        with pytest.warns(recognize.RecognitionWarning, match=r"values\s+differ"):
            assert recognize.match(other_step, cost, 2, 1, torch.device("cpu")) is None
    finally:
        recognize._table_cache = table
        recognize._warned.clear()


def lambda_free(fn):
    """The undecorated Python function behind a tagged plugin (torch.jit.script wants a plain function)."""
    return getattr(fn, "__wrapped__", fn)


def test_probe_comparison_is_relative_per_column():
    """recognize._same: 1e-6 of EACH column's scale: an error of 1e-6 x the position scale in the velocity column
    of a mountain-car-like state (velocity ~ 0.07, position ~ 1.4) is 20x that column's tolerance and must fail."""
    import torch

    from pi_mpc import recognize

    g = torch.Generator().manual_seed(1)
    y = torch.stack([1.4 * (2 * torch.rand(256, generator=g) - 1), 0.07 * (2 * torch.rand(256, generator=g) - 1)], dim=1)
    assert recognize._same(y.clone(), y)
    x = y.clone()
    x[17, 1] += 1.0e-6 * float(y[:, 0].abs().max())
    assert not recognize._same(x, y)
    x = y.clone()
    x[17, 0] += 0.5e-6 * float(y[:, 0].abs().max())
    assert recognize._same(x, y)
    assert not recognize._same(y[:, :1], y)


def test_predicted_scaling_curve_is_self_consistent():
    """bench.py's predicted 1 -> 8 GPU curve (no multi-GPU node has been available to any round): one row per
    W and transport, efficiencies in (0, 1], monotone in W, the assumed collective latency's range brackets the central value,
    the strong-scaling solve never beats its measured one-GPU solve of 2^20 / W samples, live inputs override the committed
    ones, and every file its inputs cite exists."""
    import importlib.util
    import re

    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    p = bench.predict_scaling()
    n, T = 1 << 20, 50
    for tr in ("rccl", "nccl", "p2p"):
        rows = p["transports"][tr]
        assert list(rows) == ["1", "2", "4", "8"]
        t_prev, eff_prev = 0.0, 1.0 + 1e-12
        for W in (1, 2, 4, 8):
            r = rows[str(W)]
            assert r["value"] == pytest.approx(W * n * T / (r["ms_per_step"] * 1e-3))
            assert 0.0 < r["efficiency"] <= 1.0 and r["efficiency"] <= eff_prev and r["ms_per_step"] >= t_prev
            t_prev, eff_prev = r["ms_per_step"], r["efficiency"]
            if W > 1:
                assert r["ms_per_step_high"] <= r["ms_per_step"] <= r["ms_per_step_low"]
                assert r["value_low"] <= r["value"] <= r["value_high"]
                local = bench.SCALING_INPUTS["solve_ms_by_local_samples"][n // W]
                assert r["strong_ms_per_step"] > local and 0.0 < r["strong_efficiency"] < 1.0
    assert p["transports"]["p2p"]["8"]["ms_per_step"] < p["transports"]["rccl"]["8"]["ms_per_step"] < p["transports"]["nccl"]["8"]["ms_per_step"]
    q = bench.predict_scaling(solve_ms_1gpu=0.2)
    assert q["inputs"]["solve_ms_1gpu"] == 0.2 and q["transports"]["rccl"]["1"]["ms_per_step"] == 0.2
    assert q["transports"]["rccl"]["8"]["ms_per_step"] == pytest.approx(0.2 + p["transports"]["rccl"]["8"]["ms_per_step"] - p["transports"]["rccl"]["1"]["ms_per_step"])
    cited = set()
    for v in bench.SCALING_INPUTS.values():
        cited |= set(re.findall(r"(?:profiles/[A-Za-z0-9_.]+|BENCH_r\d+\.json)", v.get("source", "")))
    assert cited and all(os.path.exists(os.path.join(ROOT, f)) for f in cited), cited


def test_row_pool_hands_out_fresh_aligned_rows():
    """mppi_playground_amd/_pool.py: every row is handed out once, contiguous and 256 bytes apart, rows never overlap, a new
    block starts when the rows run out or the stream changes, and a dropped block's rows stay valid while somebody holds one."""
    import torch
    from mppi_playground_amd._pool import RowPool

    pool = RowPool((25, 2), torch.device("cpu"), torch.float32, block_bytes=4096)
    rows = [pool.take(7) for _ in range(3 * pool._per_block + 1)]
    assert pool._per_block == 16 and all(r.shape == (25, 2) and r.is_contiguous() and r.data_ptr() % 64 == 0 for r in rows)
    assert (rows[1].data_ptr() - rows[0].data_ptr()) % 256 == 0  # (blocks of torch's GPU allocator start on 512-byte boundaries)
    spans = sorted((r.data_ptr(), r.data_ptr() + 200) for r in rows)
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))
    for i, r in enumerate(rows):
        r.fill_(float(i))
    assert all(float(r[0, 0]) == float(i) and float(r[-1, -1]) == float(i) for i, r in enumerate(rows))
    left = pool.rows_left
    other = pool.take(8)  # another stream: a new block
    assert pool.rows_left == pool._per_block - 1 and left != pool.rows_left + 1 and other.data_ptr() not in {r.data_ptr() for r in rows}
    scalar = RowPool((), torch.device("cpu"), torch.bool).take(0)
    assert scalar.shape == () and scalar.dtype == torch.bool
    big = RowPool((300, 26, 4), torch.device("cpu"), torch.float32)
    assert big._per_block == 8 and big.take(0).numel() == 300 * 26 * 4


def test_value_bins_of_the_one_launch_top_k_are_monotone_and_exact():
    """csrc/mppi_topk.hpp `bin_of` (the exponent and six mantissa bits of cost - min, counted down from those of max - min,
    clamped to 0 .. 2047), restated in numpy: monotone in the cost, hence {keys in bins <= b*} is a superset of the k smallest
    and ordering that set by (bin, word) is ordering it by word — the property the device select + counting sort rely on
    (the implementation itself is checked on the GPU: test_one_launch_top_k_on_randomised_cost_vectors, scripts/topk_soak.py)."""
    rng = np.random.default_rng(11)

    def bins(c):
        cmn, cmx = c.min(), c.max()
        base = int(np.float32(cmx - cmn).view(np.uint32) >> 17) - 2047
        u = ((c - cmn).astype(np.float32).view(np.uint32) >> 17).astype(np.int64)
        return np.clip(u - base, 0, 2047)

    for trial in range(200):
        N = int(rng.integers(2, 4097))
        kind = trial % 4
        if kind == 0:
            c = rng.uniform(77e3, 110e3, N)
        elif kind == 1:
            c = rng.uniform(300, 3000, N) + 1e4 * rng.integers(0, 25, N) * (rng.random(N) < 0.4)
        elif kind == 2:
            c = rng.standard_normal(N) * 10.0 ** rng.integers(-3, 6)
        else:
            c = np.exp(rng.uniform(-20, 20, N))
        c = c.astype(np.float32)
        if c.max() == c.min():
            continue
        b = bins(c)
        order = np.argsort(c, kind="stable")
        assert np.all(np.diff(b[order]) >= 0)  # monotone
        k = int(rng.integers(1, min(N, 1024) + 1))
        words = (c.view(np.uint32).astype(np.uint64) ^ np.where(c.view(np.int32) < 0, 0xFFFFFFFF, 0x80000000).astype(np.uint64)) << np.uint64(32) \
            | np.arange(N, dtype=np.uint64)  # (order-preserving key of the float, then the index)
        bstar = int(np.searchsorted(np.cumsum(np.bincount(b, minlength=2048)), k))
        chosen = np.flatnonzero(b <= bstar)
        assert len(chosen) >= k
        by_bin_then_word = chosen[np.lexsort((words[chosen], b[chosen]))][:k]
        assert np.array_equal(by_bin_then_word, np.argsort(words, kind="stable")[:k])
        if kind == 1 and N >= 2000 and k >= 300:  # the running racing loop: the boundary set stays within one row of 1024 words
            assert len(chosen) <= 1024
