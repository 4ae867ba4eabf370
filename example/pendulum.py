"""Pendulum swing-up (counterpart of the reference's example/pendulum.py, its model as the simulator)."""
import numpy as np
import torch

from _common import run_loop
from envs.classic_control import pendulum_cost, pendulum_dynamics
from pi_mpc.mppi import MPPI


def main(steps: int = 200):
    solver = MPPI(horizon=15, num_samples=1000, dim_state=2, dim_control=1, dynamics=pendulum_dynamics,
                  cost_func=pendulum_cost, u_min=torch.tensor([-2.0]), u_max=torch.tensor([2.0]),
                  sigmas=torch.tensor([1.0]), lambda_="ESSPS")
    step = lambda s, u: pendulum_dynamics(s.view(1, -1), u.view(1, -1)).view(-1)  # noqa: E731
    run_loop(solver, step, torch.tensor([np.pi, 0.0], device="cuda"), steps, "pendulum")


if __name__ == "__main__":
    main()
