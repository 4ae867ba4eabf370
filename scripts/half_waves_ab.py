#!/usr/bin/env python3
"""Same-box A/B of the rollout kernel's half-filled waves (option half_waves: 32 trajectories per wave, twice the waves) against
full waves, interleaved, min of three 50-solve loops, three repetitions; rollout stage from HIP events.
Usage (GPU box): python scripts/half_waves_ab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import bench
import mppi_playground_amd  # noqa: F401
from envs import classic_control as cc
from envs.navigation_2d import Navigation2DEnv
from envs.racing_controller import racing_controller
from envs.racing_env import RacingEnv
from pi_mpc.mppi import MPPI

nav = Navigation2DEnv()
env = RacingEnv()
t = torch.tensor


def racing(n):
    ctrl = racing_controller(env, horizon=50, num_samples=n, lambda_=1.0, lazy_state_seq=True)
    ctrl.set_cost_map(env._obstacle_map, env._lane_map)
    x0 = env.reset().clone()
    ref, _ = ctrl.calc_ref_trajectory(x0, env.racing_center_path, 0, 50, DL=0.1, lookahead_distance=3, reference_path_interval=0.85)
    ctrl.set_reference(ref)
    keep.append(ctrl)
    return ctrl.solver


keep = []
LZ = dict(lazy_state_seq=True)
cases = {}
for n in (16384, 32768, 65536, 131072, 262144):
    cases[f"nav2d N={n} T=50 lambda=1"] = (lambda n=n: MPPI(50, n, 3, 2, nav.dynamics, nav.cost_function, nav.u_min, nav.u_max, t([0.5, 0.5]), 1.0, **LZ), nav.reset().clone())
cases["c2_essps nav2d N=65536 T=50"] = (lambda: MPPI(50, 65536, 3, 2, nav.dynamics, nav.cost_function, nav.u_min, nav.u_max, t([0.5, 0.5]), "ESSPS", **LZ), nav.reset().clone())
for n in (65536, 131072):
    cases[f"cartpole N={n} T=64 ESSPS"] = (lambda n=n: MPPI(64, n, 4, 1, cc.cartpole_dynamics, cc.cartpole_cost, t([-3.0]), t([3.0]), t([1.0]), "ESSPS", **LZ), t([0.01, 0.0, 0.02, 0.0]).cuda())
cases["pendulum N=65536 T=50 lambda=1"] = (lambda: MPPI(50, 65536, 2, 1, cc.pendulum_dynamics, cc.pendulum_cost, t([-2.0]), t([2.0]), t([1.0]), 1.0, **LZ), t([np.pi, 0.0], dtype=torch.float32).cuda())
for n in (65536, 131072, 262144):
    cases[f"racing N={n} T=50 lambda=1"] = (lambda n=n: racing(n), env.reset().clone())
for name, (make, x0) in cases.items():
    rows = {0: [], 2: []}
    st = {}
    for rep in range(3):
        for hw in (2, 0):
            s = make()
            s.set_option("half_waves", hw)
            rows[hw].append(bench._time_solver(torch, s, x0, n=50, warm=20) * 1e6)
            if rep == 0:
                st[hw] = bench._stage_times(torch, s, x0, n=30)["rollout_cost"] * 1e3
            del s
            keep.clear()
    print(f"{name:34s} half waves {min(rows[2]):6.1f} us ({' '.join('%.1f' % v for v in rows[2])}) rollout {st[2]:5.1f} | full waves "
          f"{min(rows[0]):6.1f} us ({' '.join('%.1f' % v for v in rows[0])}) rollout {st[0]:5.1f}", flush=True)
