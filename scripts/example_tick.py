#!/usr/bin/env python3
"""The reference's racing example loop at its own size (T = 25, N = 4000: update + env.step + collision_check +
get_top_samples(300) per tick, bench.py's `example_loop`) as a stand-alone command for rocprofv3:
    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o tick -- python scripts/example_tick.py [ticks]"""
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
state = env.reset()
for tick in range(20 + ticks):
    if tick == 20:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    a, s = ctrl.update(state, env.racing_center_path)
    state, _ = env.step(a[0, :])
    env.collision_check(state=s)
    ctrl.get_top_samples(num_samples=300)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / ticks * 1e6:.1f} us per tick over {ticks} ticks")
