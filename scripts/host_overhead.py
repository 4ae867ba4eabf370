"""How long does the host take to enqueue one solve (no synchronisation inside the loop)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import mppi_playground_amd  # noqa
from envs.racing_controller import racing_controller
from envs.racing_env import RacingEnv

env = RacingEnv()
for N in (4096, 1 << 20):
    ctrl = racing_controller(env, horizon=50, num_samples=N, lambda_=1.0)
    ctrl.set_cost_map(env._obstacle_map, env._lane_map)
    x0 = env.reset().clone()
    ref, _ = ctrl.calc_ref_trajectory(x0, env.racing_center_path, 0, 50, DL=0.1, lookahead_distance=3, reference_path_interval=0.85)
    ctrl.set_reference(ref)
    s = ctrl.solver
    for _ in range(20):
        s.forward(x0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        s.forward(x0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"N={N}: enqueue {1e6*(t1-t0)/300:.1f} us/solve, wall {1e6*(t2-t0)/300:.1f} us/solve")
