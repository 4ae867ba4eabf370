#!/usr/bin/env python3
"""get_top_samples alone (200 calls back to back after one solve), us per call, over N and k; racing T = 25 and 50.
Usage (GPU box): python scripts/top_samples_breakdown.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mppi_playground_amd  # noqa
from envs.racing_controller import racing_controller
from envs.racing_env import RacingEnv

env = RacingEnv()
for T in (25, 50):
    for N in (1000, 3000, 4000, 65536):
        ctrl = racing_controller(env, horizon=T, num_samples=N, lambda_=1.0)
        ctrl.set_cost_map(env._obstacle_map, env._lane_map)
        state = env.reset()
        ctrl.update(state, env.racing_center_path)
        out = []
        for k in (1, 64, 100, 300, 1000):
            for _ in range(10):
                ctrl.get_top_samples(num_samples=k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                ctrl.get_top_samples(num_samples=k)
            torch.cuda.synchronize()
            out.append(f"k={k}: {(time.perf_counter() - t0) / 200 * 1e6:.1f}")
        print(f"racing T={T} N={N}: " + ", ".join(out) + " us per call", flush=True)

# the same query on the costs of a RUNNING loop (collision penalties of 10^4 per step next to costs of a few hundred: the
# shape of the cost vector decides how well the value bins of the one-launch select separate the candidates)
ctrl = racing_controller(env, horizon=25, num_samples=4000, lambda_=1.0)
ctrl.set_cost_map(env._obstacle_map, env._lane_map)
state = env.reset()
for tick in range(200):
    a, s = ctrl.update(state, env.racing_center_path)
    state, _ = env.step(a[0, :])
out = []
for k in (1, 64, 100, 300, 1000):
    for _ in range(10):
        ctrl.get_top_samples(num_samples=k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        ctrl.get_top_samples(num_samples=k)
    torch.cuda.synchronize()
    out.append(f"k={k}: {(time.perf_counter() - t0) / 200 * 1e6:.1f}")
c = ctrl.solver._costs
print(f"racing T=25 N=4000 after 200 ticks of the loop (costs {float(c.min()):.0f} .. {float(c.max()):.0f}): " + ", ".join(out) + " us per call", flush=True)
