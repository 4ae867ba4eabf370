"""Drive the unicycle of Navigation2DEnv to its goal with the MI355X MPPI solver (no rendering).

Same public calls as the reference's navigation example — env plugin callables into MPPI(...), one solve per
tick, the first action applied to the env — written as a small report instead of a video."""
import time

import torch

import _common  # noqa: F401  (puts mppi_playground_amd on the path)
from envs.navigation_2d import Navigation2DEnv
from pi_mpc.mppi import MPPI

SETTINGS = dict(horizon=30, num_samples=3000, dim_state=3, dim_control=2, lambda_="ESSPS")


def run(tick_limit: int = 500, shown_rollouts: int = 300) -> bool:
    world = Navigation2DEnv()
    mppi = MPPI(dynamics=world.dynamics, cost_func=world.cost_function, u_min=world.u_min, u_max=world.u_max,
                sigmas=torch.tensor([0.5, 0.5]), **SETTINGS)
    pose, arrived, hits, t0 = world.reset(), False, 0, None
    for tick in range(1, tick_limit + 1):
        if tick == 2:  # (the first tick pays the one-time set-up: module load, buffers, map rasterisation)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        plan, predicted = mppi(pose)
        pose, arrived = world.step(plan[0])
        hits += int(world.collision_check(state=predicted).sum())
        mppi.get_top_samples(num_samples=shown_rollouts)  # what the reference draws every tick
        if arrived:
            print(f"Goal Reached! ({tick} steps, collisions along the way: {hits})")
            break
    torch.cuda.synchronize()
    per_tick = (time.perf_counter() - t0) / max(tick - 1, 1) * 1e3 if t0 is not None else float("nan")
    print(f"final state {pose.tolist()}  ({per_tick:.3f} ms per tick after the first)")
    return bool(arrived)


if __name__ == "__main__":
    run()
