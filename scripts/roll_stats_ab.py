#!/usr/bin/env python3
"""Same-box A/B of the first-grid ESSPS statistics: from the rollout kernel's epilogue (option roll_stats = 1, the default up to
1024 rollout blocks) against the separate 32-temperature statistics pass (roll_stats = 0).  Interleaved, best of three 50-solve
loops each, three repetitions.  Usage (GPU box): python scripts/roll_stats_ab.py"""
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
from pi_mpc.mppi import MPPI

nav = Navigation2DEnv()
t = torch.tensor
cases = {
    "c2_essps nav2d N=65536 T=50": (lambda: MPPI(50, 65536, 3, 2, nav.dynamics, nav.cost_function, nav.u_min, nav.u_max, t([0.5, 0.5]), "ESSPS",
                                                 lazy_state_seq=True), nav.reset().clone()),
    "nav2d N=131072 T=50": (lambda: MPPI(50, 131072, 3, 2, nav.dynamics, nav.cost_function, nav.u_min, nav.u_max, t([0.5, 0.5]), "ESSPS",
                                         lazy_state_seq=True), nav.reset().clone()),
    "c5 cartpole N=262144 T=64 +SG": (lambda: MPPI(64, 262144, 4, 1, cc.cartpole_dynamics, cc.cartpole_cost, t([-3.0]), t([3.0]), t([1.0]), "ESSPS",
                                                   use_sg_filter=True, lazy_state_seq=True), t([0.01, 0.0, 0.02, 0.0]).cuda()),
    "cartpole N=65536 T=64": (lambda: MPPI(64, 65536, 4, 1, cc.cartpole_dynamics, cc.cartpole_cost, t([-3.0]), t([3.0]), t([1.0]), "ESSPS",
                                           lazy_state_seq=True), t([0.01, 0.0, 0.02, 0.0]).cuda()),
    "pendulum N=32768 T=50": (lambda: MPPI(50, 32768, 2, 1, cc.pendulum_dynamics, cc.pendulum_cost, t([-2.0]), t([2.0]), t([1.0]), "ESSPS",
                                           lazy_state_seq=True), t([np.pi, 0.0], dtype=torch.float32).cuda()),
}
for name, (make, x0) in cases.items():
    rows = {0: [], 1: []}
    lam = {}
    for rep in range(3):
        for rs in (1, 0):
            s = make()
            s.set_option("roll_stats", rs)
            rows[rs].append(bench._time_solver(torch, s, x0, n=50, warm=20) * 1e6)
            lam[rs] = s._last_lambda
            if rep == 0:
                st = bench._stage_times(torch, s, x0, n=30)
                lam[("st", rs)] = {k: round(v * 1e3, 1) for k, v in st.items() if k != "sample"}
            del s
    print(f"{name:32s} epilogue {min(rows[1]):6.1f} us ({' '.join('%.1f' % v for v in rows[1])}) | statistics pass {min(rows[0]):6.1f} us "
          f"({' '.join('%.1f' % v for v in rows[0])}) | lambda {lam[1]:.7f} / {lam[0]:.7f} | stages {lam[('st', 1)]} / {lam[('st', 0)]}", flush=True)
