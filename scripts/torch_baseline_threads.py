import sys, time
sys.path[:0]=['/root/repo','/root/repo/tests']
import torch, numpy as np
import mppi_playground_amd
from envs.racing_controller import racing_controller
from envs.racing_env import RacingEnv
from oracle.torch_reference_loop import TorchReferenceLoop
cpu=torch.device('cpu')
env=RacingEnv(device=cpu)
for nt in (8, 16, 32, 64):
    torch.set_num_threads(nt)
    ctrl=racing_controller(env, device=cpu, horizon=50, num_samples=1<<20, lambda_=1.0, mppi_cls=TorchReferenceLoop)
    ctrl.set_cost_map(env._obstacle_map, env._lane_map)
    state=env.reset()
    ref,_=ctrl.calc_ref_trajectory(state, env.racing_center_path,0,50,DL=0.1,lookahead_distance=3,reference_path_interval=0.85)
    ctrl.set_reference(ref)
    ctrl.solver.forward(state.clone())
    t0=time.perf_counter(); ctrl.solver.forward(state.clone()); dt=time.perf_counter()-t0
    print(nt, 'threads:', round(dt,2), 's/solve', flush=True)
    del ctrl
