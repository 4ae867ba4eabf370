"""Goal in a danger zone (counterpart of the reference's example/goal_in_danger_zone.py, no rendering)."""
import random

import numpy as np
import torch

import _common  # noqa: F401
from envs.goal_in_danger_zone import GoalInDangerZoneEnv
from pi_mpc.mppi import MPPI


def main():
    seed = 42
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    env = GoalInDangerZoneEnv(seed=seed)
    solver = MPPI(horizon=30, num_samples=3000, dim_state=7, dim_control=2, dynamics=env.parallel_step,
                  cost_func=env.parallel_cost, u_min=torch.tensor([-1.0, -1.0]), u_max=torch.tensor([1.0, 1.0]),
                  sigmas=torch.tensor([0.5, 0.5]), lambda_=1.0)
    obs, info = env.reset(seed=seed)
    episodic_reward = episodic_cost = 0.0
    for i in range(env.max_episode_steps):
        action_seq, predicted_traj = solver.forward(state=torch.tensor(obs, dtype=torch.float32))
        obs, reward, terminated, truncated, info = env.step(action_seq[0, :].cpu().numpy())
        episodic_reward += reward
        episodic_cost += info["cost"]
        top_samples, top_weights = solver.get_top_samples(num_samples=100)
        if truncated or terminated:
            obs, info = env.reset()
    print("episodic reward: ", episodic_reward)
    print("episodic cost: ", episodic_cost)


if __name__ == "__main__":
    main()
