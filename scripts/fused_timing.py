#!/usr/bin/env python3
"""Single-launch solve (option fused_solve = 1) against the multi-kernel path, per config: us per solve, open loop."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import mppi_playground_amd  # noqa: F401
from bench import _other_solvers, _time_solver
from envs.racing_controller import racing_controller
from envs.racing_env import RacingEnv

rows = list(_other_solvers(torch, np))
env = RacingEnv()


def racing(n, T):
    def make():
        c = racing_controller(env, horizon=T, num_samples=n, lambda_=1.0)
        c.set_cost_map(env._obstacle_map, env._lane_map)
        ref, _ = c.calc_ref_trajectory(env.reset(), env.racing_center_path, 0, T, DL=0.1, lookahead_distance=3, reference_path_interval=0.85)
        c.set_reference(ref)
        c.solver._keep_ctrl = c
        return c.solver
    return make


rows += [("racing4000", "racing T=25 N=4000 lambda=1 (the example's size)", 0, 0, racing(4000, 25), env.reset().clone()),
         ("racing64k", "racing T=50 N=65536 lambda=1", 0, 0, racing(65536, 50), env.reset().clone()),
         ("racing256k", "racing T=50 N=262144 lambda=1", 0, 0, racing(262144, 50), env.reset().clone())]
for key, label, work, b_alg, make, x0 in rows:
    out = []
    for fused in (0, 2):
        s = make()
        s.set_option("fused_solve", fused)
        out.append(_time_solver(torch, s, x0, n=200, warm=30) * 1e6)
        err = s._h.lib.mppi_fused_error(s._h.h)
        del s
    print(f"{label}: multi-kernel {out[0]:.1f} us, single launch {out[1]:.1f} us{'  (FUSED ERROR FLAG)' if err else ''}", flush=True)
