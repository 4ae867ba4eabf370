#!/usr/bin/env python3
"""Where the example tick's time goes on the HOST: wall time until the loop has ENQUEUED its ticks against the time until the
device has finished them, and the host time spent inside each of the tick's four calls."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch

import mppi_playground_amd  # noqa: F401
from envs.racing_controller import racing_controller
from envs.racing_env import RacingEnv

ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 400
env = RacingEnv()
ctrl = racing_controller(env, horizon=25, num_samples=4000, lambda_=1.0)
ctrl.set_cost_map(env._obstacle_map, env._lane_map)
pc = time.perf_counter
for rep in range(3):
    state = env.reset()
    acc = [0.0] * 4
    for tick in range(20 + ticks):
        if tick == 20:
            torch.cuda.synchronize()
            acc = [0.0] * 4
            t0 = pc()
        a0 = pc()
        a, s = ctrl.update(state, env.racing_center_path)
        a1 = pc()
        state, _ = env.step(a[0, :])
        a2 = pc()
        env.collision_check(state=s)
        a3 = pc()
        ctrl.get_top_samples(num_samples=300)
        a4 = pc()
        acc[0] += a1 - a0; acc[1] += a2 - a1; acc[2] += a3 - a2; acc[3] += a4 - a3
    t1 = pc()
    torch.cuda.synchronize()
    t2 = pc()
    print(f"enqueued after {(t1 - t0) / ticks * 1e6:.1f} us per tick, finished after {(t2 - t0) / ticks * 1e6:.1f}; host time in "
          f"update {acc[0] / ticks * 1e6:.1f}, env.step {acc[1] / ticks * 1e6:.1f}, collision_check {acc[2] / ticks * 1e6:.1f}, "
          f"get_top_samples {acc[3] / ticks * 1e6:.1f} us")
