"""The CPU oracle (oracle/mppi_oracle.c) against the fixtures captured from the real reference
(tests/golden/make_golden.py).  Tolerance: 1e-5 relative to the max-abs of each tensor (SURVEY
Appendix D); observed ~2e-7."""
import numpy as np
import pytest

from helpers import CASES, load, oracle_problem, orc, racing_env_fixture, nav2d_env_fixture, rel_err, sg_coeffs

TOL = 1e-5


def used_lambda(g, cfg, k):
    """lambda the reference used for the weights of solve k (MPO updates it AFTER the weights)."""
    if cfg["lambda_"] == "MPO":
        return 1.0 if k == 0 else float(g[f"lambda_{k - 1}"])
    return float(g[f"lambda_{k}"])


@pytest.mark.parametrize("name", list(CASES))
def test_rollout_costs_states(name):
    cfg, g = CASES[name], load(name)
    P = oracle_problem(cfg["model"], cfg["N"], cfg["T"], cfg.get("exploration", 0.0))
    for k in range(int(g["K"])):
        if cfg["model"] == "racing":
            P.set_ref_path(g[f"ref_path_{k}"])
        keep = f"S_{k}" in g.files
        r = P.rollout_cost(g[f"x0_{k}"], g[f"mean_in_{k}"], g[f"eps_{k}"], want_U=keep, want_S=keep,
                           want_stage=keep)
        assert rel_err(r["costs"], g[f"costs_{k}"]) < TOL
        if keep:
            assert np.array_equal(r["U"], g[f"U_{k}"])  # clamp(mean + eps) is exact
            assert rel_err(r["S"], g[f"S_{k}"]) < TOL
            assert rel_err(r["stage"], g[f"stage_costs_{k}"]) < TOL


@pytest.mark.parametrize("name", list(CASES))
def test_weights_action_state_seq(name):
    cfg, g = CASES[name], load(name)
    P = oracle_problem(cfg["model"], cfg["N"], cfg["T"], cfg.get("exploration", 0.0))
    for k in range(int(g["K"])):
        lam = used_lambda(g, cfg, k)
        w, st = orc.softmax_weights(g[f"costs_{k}"], lam)
        assert rel_err(w, g[f"weights_{k}"]) < TOL
        a = P.weighted_actions(g[f"weights_{k}"], g[f"mean_in_{k}"], g[f"eps_{k}"])
        if cfg.get("use_sg_filter"):  # step 7 (mppi.py:423-443) with the fixture's own history
            from pi_mpc import _host

            a = _host.sg_filter_sequence(g[f"sg_hist_in_{k}"], a, sg_coeffs(cfg))
        assert rel_err(a, g[f"action_seq_{k}"]) < TOL
        s = P.rollout_single(g[f"x0_{k}"], g[f"action_seq_{k}"])
        assert rel_err(s, g[f"state_seq_{k}"][0]) < TOL


def test_top_samples_are_reference_rows():
    # get_top_samples returns rows of _state_seq_batch sorted by weight (mppi.py:462-487)
    g = load("pendulum_T15_N256_fixed")
    order = np.argsort(-g["weights_0"], kind="stable")[:8]
    assert np.allclose(g["top8_weights_0"], g["weights_0"][order])
    assert np.allclose(g["top8_states_0"], g["S_0"][order])


def test_angle_normalize_and_occupancy_pins():
    pins = load("model_pins")
    assert np.array_equal(orc.angle_normalize(pins["an_in"]), pins["an_out"])
    e, n = racing_env_fixture(), nav2d_env_fixture()
    assert np.array_equal(orc.occ(e["obst"], e["cell"], e["origin"], pins["occ_pts"]), pins["occ_obst"].ravel())
    assert np.array_equal(orc.occ(e["lane"], e["cell"], e["origin"], pins["occ_pts"]), pins["occ_lane"].ravel())
    assert np.array_equal(orc.occ(n["map"], n["cell"], n["origin"], pins["occ_nav_pts"]), pins["occ_nav"].ravel())


@pytest.mark.parametrize("seed", [0, 42])
@pytest.mark.parametrize("n", [1000, 15000])
def test_torch_cpu_normal_stream(seed, n):
    """mt19937 + Box-Muller restatement of torch's CPU normal_() (SURVEY B-Q1): two consecutive draws."""
    g = load("torch_cpu_randn")
    s = orc.TorchCpuStream(seed)
    a, b = s.randn(n), s.randn(n)
    assert np.max(np.abs(a - g[f"randn_seed{seed}_n{n}_a"])) < 2e-6
    assert np.max(np.abs(b - g[f"randn_seed{seed}_n{n}_b"])) < 2e-6


def test_ctor_draw_is_part_of_the_stream():
    """The constructor consumes one [N,T,dc] draw before the first solve (mppi.py:146-148)."""
    g = load("pendulum_T15_N256_fixed")
    s = orc.TorchCpuStream(42)
    n = 256 * 15
    assert np.max(np.abs(s.randn(n).reshape(256, 15, 1) - g["ctor_eps"])) < 2e-6
    assert np.max(np.abs(s.randn(n).reshape(256, 15, 1) - g["eps_0"])) < 2e-6


def test_philox_restatement_statistics():
    eps = orc.philox_normal(42, 1, 0, 4096, 50, 2, [0.5, 0.1])
    assert abs(eps[..., 0].std() - 0.5) < 0.01 and abs(eps[..., 1].std() - 0.1) < 0.002
    assert abs(eps.mean()) < 0.005
    # counter-based: a shard starting at 1000 reproduces rows 1000.. of the full draw
    part = orc.philox_normal(42, 1, 1000, 64, 50, 2, [0.5, 0.1])
    assert np.array_equal(part, eps[1000:1064])


def test_posterior_samples_roll_out_through_the_oracle():
    """get_samples_from_posterior (mppi.py:489-506): the reference's samples are loc + sigma * (the next normals of
    the stream) and its states are the batch rollout of those unclamped actions."""
    g, cfg = load("nav2d_T20_N256_posterior"), CASES["nav2d_T20_N256_posterior"]
    k0 = int(g["posterior_after"])
    samples, states = g["posterior_samples"], g["posterior_states"]
    P = oracle_problem("nav2d", 1, cfg["T"])
    for i in range(samples.shape[0]):
        assert rel_err(P.rollout_single(g[f"x0_{k0}"], samples[i]), states[i]) < TOL
    # stream position: ctor draw, solve 0, the posterior's [16, T, dc] normals, then solve 1
    s = orc.TorchCpuStream(42)
    n = cfg["N"] * cfg["T"] * 2
    s.randn(n), s.randn(n)
    z = s.randn(samples.size).reshape(samples.shape) * np.float32(0.5)
    assert np.max(np.abs((g[f"action_seq_{k0}"][None] + z) - samples)) < 2e-6
    assert np.max(np.abs(s.randn(n).reshape(cfg["N"], cfg["T"], 2) * np.float32(0.5) - g[f"eps_{k0 + 1}"])) < 2e-6


@pytest.mark.parametrize("name", ["pendulum_T15_N256_fixed", "pendulum_T15_N200_explore", "cartpole_T64_N1024_essps_sg",
                                  "nav2d_T30_N256_fixed_explore", "racing_T25_N256_fixed", "racing_T25_N4096_dense"])
def test_torch_cpu_restatement_of_the_reference_loop(name):
    """oracle/torch_reference_loop.py (the `cpu_baseline_torch` leg of bench.py: the reference's op structure over the
    product's torch plugins, on CPU) reproduces the reference run for run: same global seed -> the same noise bit for
    bit, and the same temperatures, action and state sequences over the closed loop."""
    import torch

    from helpers import MODEL_CFG, SOLVER_KW
    from oracle.torch_reference_loop import TorchReferenceLoop

    cfg, g = CASES[name], load(name)
    model, T, N = cfg["model"], cfg["T"], cfg["N"]
    mc = MODEL_CFG[model]
    kw = {k: cfg[k] for k in SOLVER_KW if k in cfg}
    common = dict(horizon=T, num_samples=N, lambda_=cfg["lambda_"], seed=42, **kw)
    cpu = torch.device("cpu")
    ctrl = env = None
    if model == "racing":
        from envs.racing_controller import racing_controller
        from envs.racing_env import RacingEnv

        env = RacingEnv(device=cpu)
        ctrl = racing_controller(env, device=cpu, mppi_cls=TorchReferenceLoop, **common)
        ctrl.set_cost_map(env._obstacle_map, env._lane_map)
        solver = ctrl.solver
    elif model == "nav2d":
        from envs.navigation_2d import Navigation2DEnv

        env = Navigation2DEnv(device=cpu)
        solver = TorchReferenceLoop(dim_state=3, dim_control=2, dynamics=env.dynamics, cost_func=env.cost_function,
                                    u_min=env.u_min, u_max=env.u_max, sigmas=torch.tensor(mc["sigmas"]), **common)
    else:
        from envs import classic_control as cc

        ds, dc = orc.MODEL_DIMS[orc.MODEL_IDS[model]]
        solver = TorchReferenceLoop(dim_state=ds, dim_control=dc, dynamics=getattr(cc, f"{model}_dynamics"),
                                    cost_func=getattr(cc, f"{model}_cost"), u_min=torch.tensor(mc["u_min"]),
                                    u_max=torch.tensor(mc["u_max"]), sigmas=torch.tensor(mc["sigmas"]), **common)
    assert np.array_equal(solver._action_noises.numpy(), g["ctor_eps"])
    state = torch.from_numpy(g["x0_0"])
    for k in range(int(g["K"])):
        if ctrl is not None:
            ref, ctrl.current_path_index = ctrl.calc_ref_trajectory(state, env.racing_center_path, ctrl.current_path_index,
                                                                    T, DL=0.1, lookahead_distance=3,
                                                                    reference_path_interval=0.85)
            ctrl.set_reference(ref)
        a, s = solver.forward(state.clone())
        assert np.array_equal(solver._action_noises.numpy(), g[f"eps_{k}"])
        assert rel_err(solver._costs.numpy(), g[f"costs_{k}"]) < 1e-6
        assert abs(float(solver._lambda) - float(g[f"lambda_{k}"])) <= 1e-5 * float(g[f"lambda_{k}"])
        assert rel_err(a.numpy(), g[f"action_seq_{k}"]) < 2e-5 * (k + 1)
        assert rel_err(s.numpy(), g[f"state_seq_{k}"]) < 2e-5 * (k + 1)
        if ctrl is not None:
            u = torch.clamp(a[0], env.u_min, env.u_max)
            state = env.dynamics(state.unsqueeze(0), u.unsqueeze(0)).squeeze(0)
        else:
            state = s[0, 1].clone()


FULL_SIZE = {
    "c2": ("full_c2_nav2d_T50_N65536_essps", "nav2d", {}),
    "c5": ("full_c5_cartpole_T64_N262144_essps_sg", "cartpole", dict(use_sg_filter=True)),
    "c3": ("full_c3_racing_T50_N1048576_lambda1", "racing", {}),
    "c2_lbps": ("full_c2_nav2d_T50_N65536_lbps", "nav2d", {}),
}


@pytest.mark.parametrize("which", ["c2", "c5", "c3", "c2_lbps"])
def test_oracle_at_full_size_against_the_reference(which):
    """The oracle at BASELINE.json's sizes against the real reference (tests/golden/make_golden.py fullsize: outputs and
    summaries only).  The noise is drawn again with torch's CPU generator from the fixture's seed — the reference's own
    sampler — and verified bit for bit against the fixture's checksums before it is used."""
    import os

    import torch

    from helpers import GOLDEN, MODEL_CFG

    name, model, kw = FULL_SIZE[which]
    if not os.path.exists(os.path.join(GOLDEN, name + ".npz")):
        pytest.skip(f"{name}.npz not generated")
    g = load(name)
    N, T, K = int(g["N"]), int(g["T"]), int(g["K"])
    sig = torch.tensor(MODEL_CFG[model]["sigmas"])
    gen = torch.Generator(device="cpu").manual_seed(int(g["seed"]))
    ctor = torch.randn(N, T, len(sig), generator=gen) * sig
    assert float(ctor.numpy().astype(np.float64).sum()) == float(g["ctor_eps_sum64"])
    del ctor
    P = oracle_problem(model, N, T)
    for k in range(K):
        eps = (torch.randn(N, T, len(sig), generator=gen) * sig).numpy()
        assert float(eps.astype(np.float64).sum()) == float(g[f"eps_sum64_{k}"])
        top_i = g[f"top32_idx_{k}"]
        assert np.array_equal(eps[top_i], g[f"top32_eps_{k}"])
        if model == "racing":
            P.set_ref_path(g[f"ref_path_{k}"])
        r = P.rollout_cost(g[f"x0_{k}"], g[f"mean_in_{k}"], eps)
        c = r["costs"]
        scale = float(g[f"cmax_{k}"])
        assert np.abs(c[top_i] - g[f"top32_cost_{k}"]).max() <= TOL * scale
        assert abs(float(c.min()) - float(g[f"cmin_{k}"])) <= TOL * scale and abs(float(c.max()) - scale) <= TOL * scale
        assert abs(float(c.astype(np.float64).sum()) - float(g[f"costs_sum64_{k}"])) <= 1e-6 * abs(float(g[f"costs_sum64_{k}"]))
        assert np.abs(np.sort(c)[g[f"quantile_ranks_{k}"]] - g[f"quantiles_{k}"]).max() <= TOL * scale
        assert int(np.argmin(c)) == int(top_i[0]) or float(np.diff(g[f"top32_cost_{k}"][:2])[0]) < 4e-7 * scale
        # steps 5-8 at the reference's temperature; the limit is the reference's own spread where that exceeds 1e-5
        band = max(float(g[f"band_fixed_{k}"][:, 0].max()), TOL)
        w, st = orc.softmax_weights(c, float(g[f"lambda_{k}"]))
        a = P.weighted_actions(w, g[f"mean_in_{k}"], eps)
        if kw.get("use_sg_filter"):
            from pi_mpc import _host

            a = _host.sg_filter_sequence(g[f"sg_hist_in_{k}"], a, _host.savitzky_golay_coeffs(5, 3))
        assert rel_err(a, g[f"action_seq_{k}"]) <= band, (rel_err(a, g[f"action_seq_{k}"]), band)
        s = P.rollout_single(g[f"x0_{k}"], g[f"action_seq_{k}"])
        assert rel_err(s, g[f"state_seq_{k}"][0]) < TOL
        del eps
