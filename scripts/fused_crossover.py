#!/usr/bin/env python3
"""Where does the single-launch solve stop paying?  racing (lambda = 1) and nav2d (ESSPS) at N = 2^10 .. 2^16, us per solve
with option fused_solve = 0 (multi-kernel) and 2 (single launch).  Usage (GPU box): python scripts/fused_crossover.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch

import mppi_playground_amd  # noqa: F401
from bench import _time_solver
from envs.navigation_2d import Navigation2DEnv
from envs.racing_controller import racing_controller
from envs.racing_env import RacingEnv
from pi_mpc.mppi import MPPI

env, nav = RacingEnv(), Navigation2DEnv()
t = torch.tensor


def racing(n, T=50):
    c = racing_controller(env, horizon=T, num_samples=n, lambda_=1.0)
    c.set_cost_map(env._obstacle_map, env._lane_map)
    ref, _ = c.calc_ref_trajectory(env.reset(), env.racing_center_path, 0, T, DL=0.1, lookahead_distance=3, reference_path_interval=0.85)
    c.set_reference(ref)
    c.solver._keep_ctrl = c
    return c.solver, env.reset().clone()


def nav_essps(n, T=50):
    return MPPI(T, n, 3, 2, nav.dynamics, nav.cost_function, nav.u_min, nav.u_max, t([0.5, 0.5]), "ESSPS"), nav.reset().clone()


for name, make in (("racing T=50 lambda=1", racing), ("nav2d T=50 ESSPS", nav_essps)):
    for n in (1024, 2048, 4096, 6144, 8192, 16384, 32768, 65536):
        out = []
        for fused in (0, 2):
            s, x0 = make(n)
            s.set_option("fused_solve", fused)
            out.append(_time_solver(torch, s, x0, n=200, warm=30) * 1e6)
            del s
        print(f"{name} N={n}: multi-kernel {out[0]:.1f} us, single launch {out[1]:.1f} us", flush=True)
