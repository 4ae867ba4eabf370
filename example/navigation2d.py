"""2-D navigation (counterpart of the reference's example/navigation2d.py, rendering removed)."""
import torch

import _common  # noqa: F401
from envs.navigation_2d import Navigation2DEnv
from pi_mpc.mppi import MPPI


def main(max_steps: int = 500):
    env = Navigation2DEnv()
    solver = MPPI(horizon=30, num_samples=3000, dim_state=3, dim_control=2, dynamics=env.dynamics,
                  cost_func=env.cost_function, u_min=env.u_min, u_max=env.u_max, sigmas=torch.tensor([0.5, 0.5]),
                  lambda_="ESSPS")
    state = env.reset()
    for i in range(max_steps):
        action_seq, state_seq = solver.forward(state=state)
        state, is_goal_reached = env.step(action_seq[0, :])
        is_collisions = env.collision_check(state=state_seq)
        top_samples, top_weights = solver.get_top_samples(num_samples=300)
        if is_goal_reached:
            print(f"Goal Reached! ({i + 1} steps, collisions along the way: {int(is_collisions.sum())})")
            break
    print("final state", state.tolist())


if __name__ == "__main__":
    main()
