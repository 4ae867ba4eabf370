"""2-D navigation over an obstacle map: unicycle model + goal/obstacle cost as MPPI plugins.

Counterpart of the reference's src/envs/navigation_2d.py (set-up :23-72, dynamics :218-255, cost
:257-279); simulator rendering is UI and not built.  `dynamics` / `cost_function` are torch callables
with the reference's contract and carry a native tag so MPPI runs them fused on the device.
"""
from __future__ import annotations

from typing import Tuple

import torch

from envs.common import NativeStep, angle_normalize
from envs.obstacle_map_2d import ObstacleMap, _device, generate_random_obstacles
from pi_mpc.native import native_model


def _nav_inputs(env: "Navigation2DEnv") -> dict:
    m = env._obstacle_map
    if env._params is None:  # read the device tensors back once, not per solve
        env._params = [float(env.u_min[0]), float(env.u_max[0]), float(env.u_min[1]), float(env.u_max[1]),
                       env.delta_t, m.x_lim[0], m.x_lim[1], m.y_lim[0], m.y_lim[1], float(env._goal_pos[0]),
                       float(env._goal_pos[1]), env.obstacle_weight]
    return {"params": env._params, "maps": [m.grid_spec()], "ref_path": None}


class Navigation2DEnv:
    def __init__(self, device=torch.device("cuda"), dtype=torch.float32, seed: int = 42, native_step: bool = True) -> None:
        """`native_step`: on a GPU, step() is one launch of the library's nav2d functor (envs.common.NativeStep)."""
        self._device, self._dtype = _device(device), dtype
        self._native_step, self._step_fn = bool(native_step), None
        self._obstacle_map = ObstacleMap(map_size=(20, 20), cell_size=0.1, device=self._device, dtype=dtype)
        self._seed = seed
        generate_random_obstacles(self._obstacle_map, random_x_range=(-7.5, 7.5), random_y_range=(-7.5, 7.5),
                                  num_circle_obs=7, radius_range=(1, 1), num_rectangle_obs=7, width_range=(2, 2),
                                  height_range=(2, 2), max_iteration=1000, seed=seed)
        self._obstacle_map.convert_to_torch()
        self.delta_t = 0.1
        self.obstacle_weight = 10000.0
        self._start_pos = torch.tensor([-9.0, -9.0], device=self._device, dtype=dtype)
        self._goal_pos = torch.tensor([9.0, 9.0], device=self._device, dtype=dtype)
        self.u_min = torch.tensor([0.0, -1.0], device=self._device, dtype=dtype)
        self.u_max = torch.tensor([2.0, 1.0], device=self._device, dtype=dtype)
        self._x_lim = torch.tensor(self._obstacle_map.x_lim, device=self._device, dtype=dtype)
        self._y_lim = torch.tensor(self._obstacle_map.y_lim, device=self._device, dtype=dtype)
        self._robot_state = torch.zeros(3, device=self._device, dtype=dtype)
        self._params = None
        self.reset()

    def reset(self) -> torch.Tensor:
        d = self._goal_pos - self._start_pos
        self._robot_state[:2] = self._start_pos
        self._robot_state[2] = angle_normalize(torch.atan2(d[1], d[0]))
        return self._robot_state

    def step(self, u: torch.Tensor) -> Tuple[torch.Tensor, bool]:
        if self._native_step and self._device.type == "cuda" and torch.is_tensor(u) and u.is_cuda:
            if self._step_fn is None:
                self._step_fn = NativeStep("nav2d", _nav_inputs(self)["params"], self.u_min, self.u_max, self._goal_pos,
                                           0.5, 3, self._device, self._dtype)
            self._robot_state, reached = self._step_fn(self._robot_state, u)
            return self._robot_state, reached
        u = torch.clamp(u, self.u_min, self.u_max)
        self._robot_state = self.dynamics(self._robot_state.unsqueeze(0), u.unsqueeze(0)).squeeze(0)
        reached = torch.norm(self._robot_state[:2] - self._goal_pos) < 0.5
        return self._robot_state, reached

    @native_model("nav2d", "dynamics", _nav_inputs)
    def dynamics(self, state: torch.Tensor, action: torch.Tensor, delta_t: float = 0.1) -> torch.Tensor:
        x, y, theta = state[:, 0:1], state[:, 1:2], angle_normalize(state[:, 2:3])
        v = torch.clamp(action[:, 0:1], self.u_min[0], self.u_max[0])
        omega = torch.clamp(action[:, 1:2], self.u_min[1], self.u_max[1])
        new_x = x + v * torch.cos(theta) * delta_t
        new_y = y + v * torch.sin(theta) * delta_t
        new_theta = angle_normalize(theta + omega * delta_t)
        new_x = torch.clamp(new_x, self._x_lim[0], self._x_lim[1])
        new_y = torch.clamp(new_y, self._y_lim[0], self._y_lim[1])
        return torch.cat([new_x, new_y, new_theta], dim=1)

    @native_model("nav2d", "cost", _nav_inputs)
    def cost_function(self, state: torch.Tensor, action: torch.Tensor, info: dict) -> torch.Tensor:
        goal_cost = torch.norm(state[:, :2] - self._goal_pos, dim=1)
        occ = self._obstacle_map.compute_cost(state[:, :2].unsqueeze(1)).squeeze(1)
        return goal_cost + self.obstacle_weight * occ

    def collision_check(self, state: torch.Tensor) -> torch.Tensor:
        return self._obstacle_map.compute_cost(state[:, :, :2]).squeeze(1)

    def render(self, *args, **kwargs) -> None:  # UI: out of scope
        return None

    def close(self, *args, **kwargs) -> None:
        return None
