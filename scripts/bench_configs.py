#!/usr/bin/env python3
"""Solve rates of the other BASELINE configs (C1, C2, C5) — reported in DESIGN.md, not bench lines."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import mppi_playground_amd  # noqa: F401
from envs import classic_control as cc
from envs.navigation_2d import Navigation2DEnv
from pi_mpc.mppi import MPPI


def timeit(solver, x0, n=50, warm=10):
    for _ in range(warm):
        solver.forward(x0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        solver.forward(x0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    rows = []
    s = MPPI(50, 1000, 2, 1, cc.pendulum_dynamics, cc.pendulum_cost, torch.tensor([-2.0]), torch.tensor([2.0]),
             torch.tensor([1.0]), "ESSPS")
    rows.append(("C1 pendulum T=50 N=1000 ESSPS", 1000 * 50, timeit(s, torch.tensor([np.pi, 0.0], device="cuda"))))
    env = Navigation2DEnv()
    for lam in ("ESSPS", 1.0):
        s = MPPI(50, 65536, 3, 2, env.dynamics, env.cost_function, env.u_min, env.u_max, torch.tensor([0.5, 0.5]), lam)
        rows.append((f"C2 nav2d T=50 N=65536 lambda={lam}", 65536 * 50, timeit(s, env.reset().clone())))
    for stats in ("device", "host"):
        s = MPPI(64, 262144, 4, 1, cc.cartpole_dynamics, cc.cartpole_cost, torch.tensor([-3.0]), torch.tensor([3.0]),
                 torch.tensor([1.0]), "ESSPS", use_sg_filter=True, auto_lambda_stats=stats)
        rows.append((f"C5 cartpole T=64 N=262144 ESSPS+SG (stats on {stats})", 262144 * 64,
                     timeit(s, torch.tensor([0.01, 0.0, 0.02, 0.0], device="cuda"), n=20 if stats == "host" else 50)))
    for name, work, t in rows:
        print(f"{name:62s} {t * 1e3:9.3f} ms/solve {1 / t:10.1f} solves/s {work / t:12.4g} sample-steps/s")


if __name__ == "__main__":
    main()
