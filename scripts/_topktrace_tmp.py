import os, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path[:0] = [ROOT]
import torch
import mppi_playground_amd  # noqa
from envs.racing_controller import racing_controller
from envs.racing_env import RacingEnv
env = RacingEnv()
ctrl = racing_controller(env, horizon=25, num_samples=4000, lambda_=1.0)
ctrl.set_cost_map(env._obstacle_map, env._lane_map)
state = env.reset()
for _ in range(6):
    a, s = ctrl.update(state, env.racing_center_path)
    ctrl.get_top_samples(num_samples=300)
    torch.cuda.synchronize()
