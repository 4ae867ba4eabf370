#!/usr/bin/env python3
"""cProfile of the example tick's HOST side (the loop is host-bound at the example's size: scripts/example_tick_host.py)."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch

import mppi_playground_amd  # noqa: F401
from envs.racing_controller import racing_controller
from envs.racing_env import RacingEnv

ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
env = RacingEnv()
ctrl = racing_controller(env, horizon=25, num_samples=4000, lambda_=1.0)
ctrl.set_cost_map(env._obstacle_map, env._lane_map)


def loop(n):
    state = env.reset()
    for _ in range(n):
        a, s = ctrl.update(state, env.racing_center_path)
        state, _ = env.step(a[0, :])
        env.collision_check(state=s)
        ctrl.get_top_samples(num_samples=300)
    torch.cuda.synchronize()


loop(50)
pr = cProfile.Profile()
pr.enable()
loop(ticks)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
