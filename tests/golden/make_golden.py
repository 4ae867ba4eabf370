#!/usr/bin/env python3
"""Golden-vector generator (TEST INFRASTRUCTURE — runs only in the build container).

Imports the *real* reference (``/root/reference``, read-only, CPU) and records the
inputs/outputs of ``pi_mpc.mppi.MPPI.forward`` for the five shipped models into small
``.npz`` fixtures under ``tests/golden/``.  Nothing from the reference (source,
bytecode) is written into the fixtures: they hold arrays only.

Usage (container only; the GPU box has no /root/reference):
    python tests/golden/make_golden.py            # everything
    python tests/golden/make_golden.py round2     # only the cases added in round 2 (dense softmax at N = 4096,
                                                  # racing / nav2d with SG, exploration, LBPS, MPO, ESSPS end point,
                                                  # get_samples_from_posterior between two solves)

Recipe follows SURVEY.md Appendix C: stub the UI-only imports (moviepy, fire,
gymnasium), run with cwd=/root/reference (racing_env.py:47-49 opens a relative CSV),
extract the nested model closures of example/{pendulum,cartpole,mountaincar}.py with
``ast`` and run them eagerly.
"""
from __future__ import annotations

import ast
import os
import sys
import textwrap
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    sys.path[:0] = [f"{REF}/src", f"{REF}/example"]
    for n in [
        "moviepy",
        "moviepy.video",
        "moviepy.video.io",
        "moviepy.video.io.ImageSequenceClip",
        "fire",
        "gymnasium",
    ]:
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.modules["moviepy.video.io.ImageSequenceClip"].ImageSequenceClip = object
    sys.modules["fire"].Fire = lambda f: None
    gym = sys.modules["gymnasium"]  # GoalInDangerZoneEnv subclasses gym.Env and builds spaces.Box in __init__
    gym.Env = object
    gym.spaces = types.ModuleType("gymnasium.spaces")
    gym.spaces.Box = lambda *a, **k: None
    sys.modules["gymnasium.spaces"] = gym.spaces
    import matplotlib

    matplotlib.use("Agg")
    os.chdir(REF)


def _extract_closures(path: str, names):
    """Pull nested FunctionDefs out of ``main()`` of an example script (eager, no jit)."""
    import torch

    src = open(path).read()
    tree = ast.parse(src)
    ns = {"torch": torch}
    # module-level angle_normalize (drop the jit decorator)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == "angle_normalize":
            node.decorator_list = []
            seg = ast.unparse(node)
            exec(seg, ns)
    main = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main"][0]
    out = {}
    for node in main.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            node.decorator_list = []
            seg = textwrap.dedent(ast.unparse(node))
            exec(seg, ns)
            out[node.name] = ns[node.name]
    return out


def pack_bits(m: np.ndarray) -> np.ndarray:
    return np.packbits((m != 0).astype(np.uint8).ravel())


class Recorder:
    """Wraps cost_func to capture the per-call costs the solver sums (mppi.py:307-336).

    `feed`: when set to a tensor [N], the wrapped cost is NOT evaluated: the first call of the solve returns `feed` and
    the other T calls return zeros, so that `torch.sum(costs, dim=1) + terminal` (mppi.py:333-334) is exactly `feed` and
    the REFERENCE's own steps 4-8 (temperature rule, softmax, weighted mean, SG filter, batch-1 rollout, warm start:
    mppi.py:341-460) run on prescribed total costs — the sensitivity probes below."""

    def __init__(self, fn):
        self.fn = fn
        self.calls = []
        self.feed = None

    def __call__(self, state, action, info):
        if self.feed is not None:
            first = not self.calls
            self.calls.append(None)
            return self.feed.clone() if first else torch_zeros_like(self.feed)
        c = self.fn(state, action, info)
        self.calls.append(c.detach().clone())
        return c


def torch_zeros_like(t):
    import torch

    return torch.zeros_like(t)


# ---------------------------------------------------------------------------------------------------------------------
# The reference's OWN sensitivity to fp32 rounding of its costs ("bands").  forward() turns the N total costs into the
# action sequence through softmax(-c / lambda): a cost change dc moves a weight by w * dc / lambda, so the 1e-5 the
# north star asks of action_seq / state_seq is a statement about the costs to ~1e-5 * lambda — below one fp32 ulp of the
# cost for nav2d / goal zone at lambda = 1.  Instead of deriving a tolerance analytically, every fixture records how far
# the reference's own outputs move when its total costs are replaced by equally valid fp32 evaluations of the same
# sums.  VARIANTS (index = column of every band array):
#   0-15  every total cost moved to the next fp32 value up or down (seeded random signs): a 1-ulp change
#   16     the T stage costs + terminal re-summed in float64, rounded once to fp32 (the exactly rounded sum — what the
#          HIP kernels compute for every model but racing, mppi_models.hpp: CostSum)
#   17     the same terms summed sequentially in fp32, t = 0..T-1 then the terminal (another valid fp32 order — the
#          racing kernel's; torch.sum's own order is a vectorised cascade that depends on the CPU's ISA)
#   18-23  every stage cost and the terminal moved by one fp32 ulp (seeded signs), then summed by the reference's own
#          torch.sum(dim=1) + terminal: what a different-but-valid fp32 evaluation of each stage cost would do
# A band is the maximum over the variants: a sample of the reference's spread under rounding-level changes of its costs
# (24 probes), against which the tests hold the HIP path's distance to the reference: err <= max(1e-5, band).
BAND_VARIANTS = tuple(f"ulp_{i}" for i in range(16)) + ("f64sum", "seqsum") + tuple(f"stage_ulp_{i}" for i in range(6))


def _variant_costs(v: int, costs, stage, terminal):
    import torch

    name = BAND_VARIANTS[v]
    rng = np.random.default_rng(9000 + v)
    if name.startswith("ulp_"):
        sign = torch.from_numpy(rng.choice(np.float32([-1.0, 1.0]), size=costs.shape[0]))
        return torch.nextafter(costs, costs + sign * float("inf"))
    if name == "f64sum":
        return (stage.double().sum(dim=1) + terminal.double()).float()
    if name == "seqsum":
        acc = torch.zeros_like(terminal)
        for t in range(stage.shape[1]):
            acc = acc + stage[:, t]
        return acc + terminal
    sign = torch.from_numpy(rng.choice(np.float32([-1.0, 1.0]), size=tuple(stage.shape)))
    st = torch.nextafter(stage, stage + sign * float("inf"))
    sign_t = torch.from_numpy(rng.choice(np.float32([-1.0, 1.0]), size=terminal.shape[0]))
    return torch.sum(st, dim=1) + torch.nextafter(terminal, terminal + sign_t * float("inf"))


def _snapshot(solver):
    import copy

    import torch

    s = dict(prev=solver._previous_action_seq.detach().clone(), hist=solver._actions_history_for_sg.detach().clone(),
             lam=solver._lambda, auto=solver._auto_lambda, rng=torch.get_rng_state())
    if hasattr(solver, "optimizer"):  # MPO: the dual and its Adam moments (mppi.py:191-200)
        s["logT"] = solver.log_temperature.data.clone()
        s["opt"] = copy.deepcopy(solver.optimizer.state_dict())
    return s


def _restore(solver, s):
    import copy

    import torch

    solver._previous_action_seq = s["prev"].clone()
    solver._actions_history_for_sg = s["hist"].clone()
    solver._lambda, solver._auto_lambda = s["lam"], s["auto"]
    torch.set_rng_state(s["rng"])
    if "opt" in s:
        solver.log_temperature.data.copy_(s["logT"])
        solver.optimizer.load_state_dict(copy.deepcopy(s["opt"]))


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def _feed_solve(solver, rec, state, costs):
    rec.calls.clear()
    rec.feed = costs
    try:
        a, s = solver.forward(state=state.clone())
    finally:
        rec.feed = None
        rec.calls.clear()
    return a.detach().numpy().copy(), s.detach().numpy().copy()


ONLY_ROUND2 = len(sys.argv) > 1 and sys.argv[1] == "round2"
ROUND2 = set()  # names registered with round2=True


def run_case(name, make_solver, x0, K, next_state, keep_S, before_solve=None, round2=False, posterior_after=None):
    """Run K closed-loop solves and dump everything the parity tests need.  posterior_after = (k, n): call
    get_samples_from_posterior(action_seq_k, state, n) right after solve k (mppi.py:489-506) and record it — the draw
    comes from the same global generator, so the noise of solve k+1 pins the stream position."""
    import torch

    if round2:
        ROUND2.add(name)
    if ONLY_ROUND2 and not round2:
        return

    solver, rec, extra = make_solver()
    N, T = solver._num_samples, solver._horizon
    d = dict(extra)
    d["ctor_eps"] = solver._action_noises.numpy().copy()  # Q2: ctor consumes one draw
    state = torch.as_tensor(np.asarray(x0), dtype=torch.float32)
    auto = solver._auto_lambda
    for k in range(K):
        if before_solve is not None:
            for kk, vv in before_solve(state, k).items():
                d[f"{kk}_{k}"] = vv
        rec.calls.clear()
        mean_in = solver._previous_action_seq.detach().clone().numpy()
        hist_in = solver._actions_history_for_sg.detach().clone().numpy()
        pre = _snapshot(solver)
        a, s = solver.forward(state=state.clone())
        calls = list(rec.calls)
        # first T+1 calls belong to the N-sample pass; (no cost calls in the B=1 rollout)
        stage = torch.stack(calls[:T], dim=1)
        terminal = calls[T]
        costs = torch.sum(stage, dim=1) + terminal  # exactly mppi.py:333-334
        d[f"x0_{k}"] = state.numpy().copy()
        d[f"mean_in_{k}"] = mean_in
        d[f"sg_hist_in_{k}"] = hist_in
        d[f"eps_{k}"] = solver._action_noises.numpy().copy()
        d[f"costs_{k}"] = costs.numpy().copy()
        d[f"stage_costs_{k}"] = stage.numpy().copy() if keep_S else np.zeros(0, np.float32)
        d[f"lambda_{k}"] = np.float64(solver._lambda)
        d[f"weights_{k}"] = solver._weights.detach().numpy().copy()
        d[f"action_seq_{k}"] = a.detach().numpy().copy()
        d[f"state_seq_{k}"] = s.detach().numpy().copy()
        if keep_S:
            d[f"U_{k}"] = solver._perturbed_action_seqs.numpy().copy()
            d[f"S_{k}"] = solver._state_seq_batch.numpy().copy()
        if k == 0 and N >= 8:
            ts, tw = solver.get_top_samples(8)
            d["top8_states_0"] = ts.numpy().copy()
            d["top8_weights_0"] = tw.detach().numpy().copy()
        if posterior_after is not None and posterior_after[0] == k:
            ps, pst = solver.get_samples_from_posterior(a, state.clone(), posterior_after[1])
            d["posterior_after"] = np.int64(k)
            d["posterior_samples"] = ps.detach().numpy().copy()
            d["posterior_states"] = pst.detach().numpy().copy()
        # ---- per-solve bands: the reference's steps 4-8 re-run on the SAME inputs with perturbed total costs
        post = _snapshot(solver)
        lam_used = pre["lam"] if auto == "MPO" else solver._lambda  # the temperature of this solve's weights
        a_np, s_np = a.detach().numpy(), s.detach().numpy()
        _restore(solver, pre)  # (sanity: feeding the recorded costs back reproduces the solve bit for bit)
        a_chk, s_chk = _feed_solve(solver, rec, state, costs)
        assert np.array_equal(a_chk, a_np) and np.array_equal(s_chk, s_np) and solver._lambda == post["lam"], name
        nv = len(BAND_VARIANTS)
        fixed = np.zeros((nv, 2))   # [variant] (action, state) at the temperature the reference used
        ruled = np.zeros((nv, 3))   # [variant] (action, state, lambda) with the temperature rule re-run
        for v in range(nv):
            cv = _variant_costs(v, costs, stage, terminal)
            _restore(solver, pre)
            solver._auto_lambda, solver._lambda = None, float(lam_used)
            av, sv = _feed_solve(solver, rec, state, cv)
            fixed[v] = _rel(av, a_np), _rel(sv, s_np)
            if auto is not None:
                _restore(solver, pre)
                av, sv = _feed_solve(solver, rec, state, cv)
                ruled[v] = _rel(av, a_np), _rel(sv, s_np), abs(solver._lambda - post["lam"]) / post["lam"]
        _restore(solver, post)
        d[f"band_fixed_{k}"] = fixed
        if auto is not None:
            d[f"band_rule_{k}"] = ruled
        state = next_state(state, a, s)
    d["K"] = np.int64(K)
    # ---- closed-loop bands: the whole K-solve loop re-run per variant (same seed -> same noise; every solve's total
    # costs replaced by the variant's evaluation of the SAME terms; states, warm start, SG history and the temperature
    # rule's memory all evolve on their own), compared with the unperturbed loop above
    cl = np.zeros((K, len(BAND_VARIANTS), 4))  # [solve][variant] (x0, action, state, lambda)
    for v in range(len(BAND_VARIANTS)):
        solver2, rec2, _ = make_solver()
        st2 = torch.as_tensor(np.asarray(x0), dtype=torch.float32)
        for k in range(K):
            if before_solve is not None:
                before_solve(st2, k)
            rec2.calls.clear()
            pre = _snapshot(solver2)
            solver2.forward(state=st2.clone())
            calls = list(rec2.calls)
            stage2, term2 = torch.stack(calls[:T], dim=1), calls[T]
            costs2 = torch.sum(stage2, dim=1) + term2
            _restore(solver2, pre)
            av, sv = _feed_solve(solver2, rec2, st2, _variant_costs(v, costs2, stage2, term2))
            lam_ref = float(d[f"lambda_{k}"])
            cl[k, v] = (_rel(st2.numpy(), d[f"x0_{k}"]), _rel(av, d[f"action_seq_{k}"]), _rel(sv, d[f"state_seq_{k}"]),
                        abs(float(solver2._lambda) - lam_ref) / lam_ref)
            a2, s2 = torch.from_numpy(av), torch.from_numpy(sv)
            if posterior_after is not None and posterior_after[0] == k:  # keeps the noise stream aligned
                solver2.get_samples_from_posterior(a2, st2.clone(), posterior_after[1])
            st2 = next_state(st2, a2, s2)
    d["band_closed_loop"] = cl
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.0f} KiB)")


def main():
    _import_reference()
    import torch

    torch.set_num_threads(1)
    from pi_mpc.mppi import MPPI

    cpu = torch.device("cpu")

    def pred_next(state, a, s):
        # closed loop without a simulator: apply the predicted next state.
        return s[0, 1].detach().clone()

    # ------------------------------------------------------------ classic control
    def classic(example, dyn_name, cost_name, **kw):
        fns = _extract_closures(f"{REF}/example/{example}.py", [dyn_name, cost_name])

        def make():
            rec = Recorder(fns[cost_name])
            solver = MPPI(dynamics=fns[dyn_name], cost_func=rec, device=cpu, **kw)
            return solver, rec, {}

        return make

    pend = dict(dim_state=2, dim_control=1, u_min=torch.tensor([-2.0]), u_max=torch.tensor([2.0]),
                sigmas=torch.tensor([1.0]))
    x0_pend = [np.pi, 0.0]
    run_case("pendulum_T50_N1000_essps",
             classic("pendulum", "dynamics", "cost_function", horizon=50, num_samples=1000,
                     lambda_="ESSPS", **pend), x0_pend, 3, pred_next, keep_S=False)
    run_case("pendulum_T15_N256_fixed",
             classic("pendulum", "dynamics", "cost_function", horizon=15, num_samples=256,
                     lambda_=1.0, **pend), x0_pend, 3, pred_next, keep_S=True)
    run_case("pendulum_T15_N256_lbps",
             classic("pendulum", "dynamics", "cost_function", horizon=15, num_samples=256,
                     lambda_="LBPS", **pend), x0_pend, 3, pred_next, keep_S=False)
    run_case("pendulum_T15_N256_mpo",
             classic("pendulum", "dynamics", "cost_function", horizon=15, num_samples=256,
                     lambda_="MPO", **pend), x0_pend, 3, pred_next, keep_S=False)
    run_case("pendulum_T15_N200_explore",
             classic("pendulum", "dynamics", "cost_function", horizon=15, num_samples=200,
                     lambda_=0.5, exploration=0.25, **pend), x0_pend, 3, pred_next, keep_S=True)

    cart = dict(dim_state=4, dim_control=1, u_min=torch.tensor([-3.0]), u_max=torch.tensor([3.0]),
                sigmas=torch.tensor([1.0]))
    x0_cart = [0.01, 0.0, 0.02, 0.0]
    run_case("cartpole_T64_N1024_essps_sg",
             classic("cartpole", "dynamics", "stage_cost", horizon=64, num_samples=1024,
                     lambda_="ESSPS", use_sg_filter=True, **cart), x0_cart, 3, pred_next, keep_S=False)
    run_case("cartpole_T10_N100_fixed",
             classic("cartpole", "dynamics", "stage_cost", horizon=10, num_samples=100,
                     lambda_=0.001, **cart), x0_cart, 3, pred_next, keep_S=True)

    mc = dict(dim_state=2, dim_control=1, u_min=torch.tensor([-1.0]), u_max=torch.tensor([1.0]),
              sigmas=torch.tensor([1.0]))
    run_case("mountaincar_T100_N256_fixed",
             classic("mountaincar", "dynamics", "cost_func", horizon=100, num_samples=256,
                     lambda_=0.1, **mc), [-0.5, 0.0], 3, pred_next, keep_S=True)

    # MuJoCo-style cart-pole (example/mujoco_cartpole.py:20-79): continuous force, masspole = 1
    run_case("mjcartpole_T50_N256_fixed",
             classic("mujoco_cartpole", "dynamics", "cost_func", horizon=50, num_samples=256,
                     lambda_=1.0, **cart), [0.01, 0.0, 0.05, 0.0], 3, pred_next, keep_S=True)

    # ------------------------------------------------------------ goal in danger zone
    from envs.goal_in_danger_zone import GoalInDangerZoneEnv

    np.random.seed(42)
    gz = GoalInDangerZoneEnv(render_mode="rgb_array", seed=42)
    gz._set_goal(gz._danger_zone)
    gz._set_initial_state(gz._danger_zone)
    gz_x0 = np.concatenate([gz._pos, [gz._angle], gz._goal - gz._pos,
                            np.array(gz._danger_zone.center) - gz._pos]).astype(np.float32)
    gz_extra = {"goal": np.asarray(gz._goal, np.float64), "center": np.asarray(gz._danger_zone.center, np.float64),
                "radius": np.float64(gz._danger_zone.radius), "x0": gz_x0}
    if not ONLY_ROUND2:
        np.savez_compressed(os.path.join(OUT, "goalzone_env.npz"), **gz_extra)

    def goalzone(**kw):
        def make():
            rec = Recorder(gz.parallel_cost)
            solver = MPPI(dim_state=7, dim_control=2, dynamics=gz.parallel_step, cost_func=rec,
                          u_min=torch.tensor([-1.0, -1.0]), u_max=torch.tensor([1.0, 1.0]),
                          sigmas=torch.tensor([0.5, 0.5]), device=cpu, **kw)
            return solver, rec, {}

        return make

    run_case("goalzone_T30_N256_fixed", goalzone(horizon=30, num_samples=256, lambda_=1.0), gz_x0, 3, pred_next,
             keep_S=True)

    # ------------------------------------------------------------ navigation 2d
    from envs.navigation_2d import Navigation2DEnv

    nav_env = Navigation2DEnv(device=cpu)
    nav_map = nav_env._obstacle_map._map
    nav_extra = {
        "map_bits": pack_bits(nav_map),
        "map_shape": np.array(nav_map.shape, np.int64),
        "cell_size": np.float64(nav_env._obstacle_map._cell_size),
        "origin": np.array(nav_env._obstacle_map._cell_map_origin, np.int64),
        "x_lim": np.array(nav_env._obstacle_map.x_lim, np.float64),
        "y_lim": np.array(nav_env._obstacle_map.y_lim, np.float64),
        "circles": np.array([[c.center[0], c.center[1], c.radius]
                             for c in nav_env._obstacle_map.circle_obs_list], np.float64),
        "rects": np.array([[r.center[0], r.center[1], r.width, r.height]
                           for r in nav_env._obstacle_map.rectangle_obs_list], np.float64),
        "goal": nav_env._goal_pos.numpy().copy(),
        "start_state": nav_env._robot_state.numpy().copy(),
    }

    def nav(**kw):
        def make():
            rec = Recorder(nav_env.cost_function)
            solver = MPPI(dim_state=3, dim_control=2, dynamics=nav_env.dynamics, cost_func=rec,
                          u_min=nav_env.u_min, u_max=nav_env.u_max,
                          sigmas=torch.tensor([0.5, 0.5]), device=cpu, **kw)
            return solver, rec, {}

        return make

    if not ONLY_ROUND2:
        np.savez_compressed(os.path.join(OUT, "nav2d_env.npz"), **nav_extra)

    x0_nav = nav_env._robot_state.numpy().copy()
    run_case("nav2d_T50_N512_essps", nav(horizon=50, num_samples=512, lambda_="ESSPS"),
             x0_nav, 3, pred_next, keep_S=False)
    run_case("nav2d_T30_N256_fixed_explore", nav(horizon=30, num_samples=256, lambda_=1.0,
                                                 exploration=0.25),
             x0_nav, 3, pred_next, keep_S=True)
    # round 2: dense softmax at N = 4096, the other temperature rules, SG, the posterior draw
    run_case("nav2d_T30_N4096_essps", nav(horizon=30, num_samples=4096, lambda_="ESSPS"), x0_nav, 2, pred_next,
             keep_S=False, round2=True)
    run_case("nav2d_T30_N512_lbps", nav(horizon=30, num_samples=512, lambda_="LBPS"), x0_nav, 3, pred_next,
             keep_S=False, round2=True)
    run_case("nav2d_T30_N512_mpo", nav(horizon=30, num_samples=512, lambda_="MPO"), x0_nav, 3, pred_next,
             keep_S=False, round2=True)
    run_case("nav2d_T30_N512_sg", nav(horizon=30, num_samples=512, lambda_=5.0, use_sg_filter=True,
                                      sg_window_size=7, sg_poly_order=2), x0_nav, 3, pred_next, keep_S=False,
             round2=True)
    run_case("nav2d_T20_N256_posterior", nav(horizon=20, num_samples=256, lambda_=5.0), x0_nav, 3, pred_next,
             keep_S=False, round2=True, posterior_after=(0, 16))

    # ------------------------------------------------------------ racing
    import racing as racing_example  # example/racing.py (fire stubbed)
    from envs.racing_env import RacingEnv

    env = RacingEnv(device=cpu)
    obst = env._obstacle_map._map
    lane = env._lane_map._map
    racing_extra = {
        "center_path": env.racing_center_path.numpy().copy(),  # [3678,3] f32
        "center_path_f64": None,
        "obst_bits": pack_bits(obst),
        "lane_bits": pack_bits(lane),
        "map_shape": np.array(obst.shape, np.int64),
        "cell_size": np.float64(env.cell_size),
        "origin": np.array(env._obstacle_map._cell_map_origin, np.int64),
        "lane_origin": np.array(env._lane_map._cell_map_origin, np.int64),
        "x_lim": np.array(env._obstacle_map.x_lim, np.float64),
        "y_lim": np.array(env._obstacle_map.y_lim, np.float64),
        "circles": np.array([[c.center[0], c.center[1], c.radius]
                             for c in env._obstacle_map.circle_obs_list], np.float64),
        "start_state": env._robot_state.numpy().copy(),
        "lane_width": np.float64(env.line_width * 0.8),
    }
    # float64 centre path as produced by the circuit generator (LaneMap input)
    from envs.circuit_generator.path_generate import make_csv_paths

    cp64, _, _ = make_csv_paths("src/envs/circuit_generator/circuit.csv")
    racing_extra["center_path_f64"] = cp64
    if not ONLY_ROUND2:
        np.savez_compressed(os.path.join(OUT, "racing_env.npz"), **racing_extra)

    def racing_case(name, T, N, K, keep_S, lambda_=1.0, round2=False, **kw):
        ctrl_box = {}
        if ONLY_ROUND2 and not round2:
            return

        def make():
            ctrl = racing_example.racing_controller(env, debug=False, device=cpu)
            ctrl.set_cost_map(env._obstacle_map, env._lane_map)
            rec = Recorder(ctrl.cost_function)
            ctrl.solver = MPPI(horizon=T, num_samples=N, dim_state=4, dim_control=2,
                               dynamics=env.dynamics, cost_func=rec, u_min=env.u_min,
                               u_max=env.u_max, sigmas=torch.tensor([0.5, 0.1]), lambda_=lambda_,
                               device=cpu, **kw)
            ctrl_box["c"] = ctrl
            return ctrl.solver, rec, {}

        def before(state, k):
            ctrl = ctrl_box["c"]
            cind_in = ctrl.current_path_index
            ref, ind = ctrl.calc_ref_trajectory(state, env.racing_center_path, cind_in,
                                                ctrl.solver._horizon, DL=0.1,
                                                lookahead_distance=3,
                                                reference_path_interval=0.85)
            ctrl.reference_path, ctrl.current_path_index = ref, ind
            return {"ref_path": ref.numpy().copy(), "cind_in": np.int64(cind_in),
                    "cind_out": np.int64(ind)}

        def nxt(state, a, s):
            u = torch.clamp(a[0], env.u_min, env.u_max)
            return env.dynamics(state.unsqueeze(0), u.unsqueeze(0)).squeeze(0).detach().clone()

        run_case(name, make, env._robot_state.numpy().copy(), K, nxt, keep_S, before_solve=before, round2=round2)

    racing_case("racing_T50_N512_fixed", 50, 512, 3, keep_S=False)
    racing_case("racing_T25_N256_fixed", 25, 256, 3, keep_S=True)
    # round 2: a dense softmax at the example's sample count, exploration + SG, ESSPS end point
    racing_case("racing_T25_N4096_dense", 25, 4096, 3, keep_S=False, lambda_=500.0, round2=True)
    racing_case("racing_T25_N512_explore_sg", 25, 512, 3, keep_S=False, lambda_=200.0, round2=True,
                exploration=0.25, use_sg_filter=True)
    racing_case("racing_T25_N1024_essps", 25, 1024, 2, keep_S=False, lambda_="ESSPS", round2=True)

    if ONLY_ROUND2:
        print("done (round-2 cases only):", sorted(ROUND2))
        return
    # ------------------------------------------------------------ torch-CPU RNG stream
    rng = {}
    for seed in (0, 42):
        for n in (1000, 15000):
            torch.manual_seed(seed)
            rng[f"randn_seed{seed}_n{n}_a"] = torch.randn(n).numpy().copy()
            rng[f"randn_seed{seed}_n{n}_b"] = torch.randn(n).numpy().copy()  # consecutive draw
    np.savez_compressed(os.path.join(OUT, "torch_cpu_randn.npz"), **rng)

    # ------------------------------------------------------------ model-level pins
    # angle_normalize / map lookup on random inputs (Appendix A pins)
    g = torch.Generator().manual_seed(7)
    xs = (torch.rand(20000, generator=g) - 0.5) * 40.0
    from envs.racing_env import angle_normalize

    pins = {"an_in": xs.numpy().copy(), "an_out": angle_normalize(xs).numpy().copy()}
    pts = (torch.rand(20000, 1, 2, generator=g) - 0.5) * 90.0
    pins["occ_pts"] = pts.numpy().copy()
    pins["occ_obst"] = env._obstacle_map.compute_cost(pts).numpy().copy()
    pins["occ_lane"] = env._lane_map.compute_cost(pts).numpy().copy()
    pts2 = (torch.rand(20000, 1, 2, generator=g) - 0.5) * 22.0
    pins["occ_nav_pts"] = pts2.numpy().copy()
    pins["occ_nav"] = nav_env._obstacle_map.compute_cost(pts2).numpy().copy()
    np.savez_compressed(os.path.join(OUT, "model_pins.npz"), **pins)
    print("done")


if __name__ == "__main__":
    main()
