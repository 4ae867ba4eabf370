#!/usr/bin/env python3
"""Host-side breakdown of an ESSPS solve (nav2d N=65536 and pendulum N=1000): where do the microseconds go?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cProfile, pstats
import torch
import mppi_playground_amd  # noqa
from pi_mpc.mppi import MPPI
from envs.navigation_2d import Navigation2DEnv
from envs import classic_control as cc

env = Navigation2DEnv()
s1 = MPPI(horizon=50, num_samples=65536, dim_state=3, dim_control=2, dynamics=env.dynamics, cost_func=env.cost_function,
          u_min=env.u_min, u_max=env.u_max, sigmas=torch.tensor([0.5, 0.5]), lambda_="ESSPS")
s2 = MPPI(horizon=50, num_samples=1000, dim_state=2, dim_control=1, dynamics=cc.pendulum_dynamics, cost_func=cc.pendulum_cost,
          u_min=torch.tensor([-2.0]), u_max=torch.tensor([2.0]), sigmas=torch.tensor([1.0]), lambda_="ESSPS")
for name, s, x0 in (("nav2d N=65536", s1, torch.tensor([-9.0, -9.0, 0.785]).cuda()), ("pendulum N=1000", s2, torch.tensor([3.14, 0.0]).cuda())):
    for _ in range(30):
        s.forward(x0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        s.forward(x0)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 300 * 1e6:.1f} us/solve, lambda {s._last_lambda:.4f}")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        s.forward(x0)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative")
    st.print_stats(14)
