"""Rollout kernel time against the number of tile-waves per SIMD (N = 65 536 x k): is there a round-quantisation loss
at the bench size (16 waves per SIMD, 5 resident)?  Usage (GPU box): python scripts/occ_curve.py"""
import os, sys, time
sys.path[:0] = ['/root/repo', '/root/repo/tests']
import torch
import mppi_playground_amd
from envs.racing_controller import racing_controller
from envs.racing_env import RacingEnv
env = RacingEnv()
for waves_per_simd in (1, 2, 3, 4, 5, 6, 8, 10, 11, 15, 16):
    N = 65536 * waves_per_simd
    for bal in (0,):
        ctrl = racing_controller(env, horizon=50, num_samples=N, lambda_=1.0)
        ctrl.set_cost_map(env._obstacle_map, env._lane_map)
        s = ctrl.solver
        x0 = env.reset().clone()
        ref, _ = ctrl.calc_ref_trajectory(x0, env.racing_center_path, 0, 50, DL=0.1, lookahead_distance=3, reference_path_interval=0.85)
        ctrl.set_reference(ref)
        for _ in range(300):
            s.forward(x0)
        torch.cuda.synchronize()
        s.set_option("timing", 2); s.stage_times_ms()
        for _ in range(200):
            s.forward(x0)
        torch.cuda.synchronize()
        print(f"waves/SIMD {waves_per_simd:2d} N={N:8d}: rollout {s.stage_times_ms()['rollout_cost']*1e3:7.1f} us", flush=True)
        del s, ctrl
