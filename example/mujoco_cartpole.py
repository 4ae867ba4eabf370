"""MuJoCo-style cart-pole (counterpart of the reference's example/mujoco_cartpole.py; the model itself is
the simulator here, InvertedPendulum-v4 needs mujoco)."""
import torch

from _common import run_loop
from envs.classic_control import mjcartpole_cost, mjcartpole_dynamics
from pi_mpc.mppi import MPPI


def main(steps: int = 200):
    solver = MPPI(horizon=50, num_samples=1000, dim_state=4, dim_control=1, dynamics=mjcartpole_dynamics,
                  cost_func=mjcartpole_cost, u_min=torch.tensor([-3.0]), u_max=torch.tensor([3.0]),
                  sigmas=torch.tensor([1.0]), lambda_=1.0)
    step = lambda s, u: mjcartpole_dynamics(s.view(1, -1), u.view(1, -1)).view(-1)  # noqa: E731
    run_loop(solver, step, torch.tensor([0.0, 0.0, 0.05, 0.0], device="cuda"), steps, "mujoco-style cartpole")


if __name__ == "__main__":
    main()
