#!/usr/bin/env python3
"""Wall time per tick of the reference's racing control loop (example/racing.py:221-266 minus rendering) at
N = 2^20, broken down by call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mppi_playground_amd  # noqa
from envs.racing_controller import racing_controller
from envs.racing_env import RacingEnv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
env = RacingEnv()
ctrl = racing_controller(env, horizon=50, num_samples=N, lambda_=1.0)
ctrl.set_cost_map(env._obstacle_map, env._lane_map)
state = env.reset()
acc = {}


def timed(name, fn, *a, **k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn(*a, **k)
    torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return r


ticks = 0
for tick in range(120):
    if tick == 20:
        acc.clear(); ticks = 0
    action_seq, state_seq = timed("controller.update (ref window + solve)", ctrl.update, state, env.racing_center_path)
    state, done = timed("env.step", env.step, action_seq[0, :])
    timed("env.collision_check", env.collision_check, state=state_seq)
    timed("get_top_samples(300)", ctrl.get_top_samples, num_samples=300)
    ticks += 1
tot = sum(acc.values())
for k, v in acc.items():
    print(f"{k:45s} {1e3 * v / ticks:8.3f} ms/tick")
print(f"{'total':45s} {1e3 * tot / ticks:8.3f} ms/tick  ({ticks} ticks, N={N})")
