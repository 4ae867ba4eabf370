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

# small problems: is the solve loop bound by the host's enqueue rate?
import numpy as np
from envs import classic_control as cc
from envs.navigation_2d import Navigation2DEnv
from pi_mpc.mppi import MPPI

nav = Navigation2DEnv()
t = torch.tensor
for name, s, x0 in (
        ("C1 pendulum N=1000 ESSPS", MPPI(50, 1000, 2, 1, cc.pendulum_dynamics, cc.pendulum_cost, t([-2.0]), t([2.0]), t([1.0]), "ESSPS"),
         t([np.pi, 0.0], dtype=torch.float32).cuda()),
        ("C1 pendulum N=1000 lambda=1", MPPI(50, 1000, 2, 1, cc.pendulum_dynamics, cc.pendulum_cost, t([-2.0]), t([2.0]), t([1.0]), 1.0),
         t([np.pi, 0.0], dtype=torch.float32).cuda()),
        ("C2 nav2d N=65536 lambda=1", MPPI(50, 65536, 3, 2, nav.dynamics, nav.cost_function, nav.u_min, nav.u_max, t([0.5, 0.5]), 1.0),
         nav.reset().clone()),
        ("C2 nav2d N=65536 ESSPS", MPPI(50, 65536, 3, 2, nav.dynamics, nav.cost_function, nav.u_min, nav.u_max, t([0.5, 0.5]), "ESSPS"),
         nav.reset().clone())):
    for _ in range(30):
        s.forward(x0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        s.forward(x0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: enqueue {1e6*(t1-t0)/300:.1f} us/solve, wall {1e6*(t2-t0)/300:.1f} us/solve")
