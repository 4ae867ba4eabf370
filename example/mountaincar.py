"""Mountain car (counterpart of the reference's example/mountaincar.py)."""
import torch

from _common import run_loop
from envs.classic_control import mountaincar_cost, mountaincar_dynamics
from pi_mpc.mppi import MPPI


def main(steps: int = 300):
    solver = MPPI(horizon=100, num_samples=1000, dim_state=2, dim_control=1, dynamics=mountaincar_dynamics,
                  cost_func=mountaincar_cost, u_min=torch.tensor([-1.0]), u_max=torch.tensor([1.0]),
                  sigmas=torch.tensor([1.0]), lambda_=0.1)
    step = lambda s, u: mountaincar_dynamics(s.clone().view(1, -1), u.view(1, -1)).view(-1)  # noqa: E731
    run_loop(solver, step, torch.tensor([-0.5, 0.0], device="cuda"), steps, "mountaincar")


if __name__ == "__main__":
    main()
