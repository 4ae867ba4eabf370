#!/usr/bin/env python3
"""Phase stamps of the one-launch get_top_samples kernel (needs a library built with -DMPPI_TOPK_TRACE:
scripts/build_variant.sh topktrace -DMPPI_TOPK_TRACE; MPPI_HIP_LIB=.../lib_topktrace.so python scripts/topk_trace.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import mppi_playground_amd  # noqa: F401
from envs.racing_controller import racing_controller
from envs.racing_env import RacingEnv

env = RacingEnv()
for T, N, k in ((25, 4000, 300), (25, 4000, 1000), (50, 4000, 300)):
    ctrl = racing_controller(env, horizon=T, num_samples=N, lambda_=1.0)
    ctrl.set_cost_map(env._obstacle_map, env._lane_map)
    state = env.reset()
    ctrl.update(state, env.racing_center_path)
    torch.cuda.synchronize()
    print(f"--- T={T} N={N} k={k}", flush=True)
    for _ in range(6):
        ctrl.get_top_samples(num_samples=k)
        torch.cuda.synchronize()
