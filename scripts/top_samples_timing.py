#!/usr/bin/env python3
"""Time get_top_samples(300) after a racing solve at N=2^20 (what example/racing.py does every tick)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mppi_playground_amd  # noqa
from envs.racing_controller import racing_controller
from envs.racing_env import RacingEnv

env = RacingEnv()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
ctrl = racing_controller(env, horizon=50, num_samples=N, lambda_=1.0)
ctrl.set_cost_map(env._obstacle_map, env._lane_map)
state = env.reset()
for _ in range(5):
    a, s = ctrl.update(state, env.racing_center_path)
    top, w = ctrl.get_top_samples(num_samples=300)
torch.cuda.synchronize()
for label, do_top in (("solve only", False), ("solve + get_top_samples(300)", True)):
    t0 = time.perf_counter()
    for _ in range(50):
        a, s = ctrl.solver.forward(state)
        if do_top:
            top, w = ctrl.get_top_samples(num_samples=300)
    torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per tick")
print("top weights", w[:4].tolist(), "shape", tuple(top.shape))
