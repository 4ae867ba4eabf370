"""Cart-pole balancing (counterpart of the reference's example/cartpole.py)."""
import torch

from _common import run_loop
from envs.classic_control import cartpole_cost, cartpole_dynamics
from pi_mpc.mppi import MPPI


def main(steps: int = 200):
    solver = MPPI(horizon=10, num_samples=100, dim_state=4, dim_control=1, dynamics=cartpole_dynamics,
                  cost_func=cartpole_cost, u_min=torch.tensor([-3.0]), u_max=torch.tensor([3.0]),
                  sigmas=torch.tensor([1.0]), lambda_=0.001)
    step = lambda s, u: cartpole_dynamics(s.view(1, -1), u.view(1, -1)).view(-1)  # noqa: E731
    run_loop(solver, step, torch.tensor([0.01, 0.0, 0.02, 0.0], device="cuda"), steps, "cartpole")


if __name__ == "__main__":
    main()
