"""Racing: kinematic-bicycle model on a closed circuit with lane and obstacle maps.

Counterpart of the reference's src/envs/racing_env.py (set-up :26-115, dynamics :327-372).  The
circuit centre line (0.1 m resampling of the reference's circuit.csv through its generator) is shipped
as data (envs/data/racing_center_path.npy); the maps are built here.  Rendering is UI and not built.
"""
from __future__ import annotations

import os
from typing import Tuple

import numpy as np
import torch

from envs.common import NativeStep, angle_normalize
from envs.lane_map_2d import LaneMap
from envs.obstacle_map_2d import ObstacleMap, _device, generate_random_obstacles
from pi_mpc.native import native_model

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def _racing_dyn_inputs(env: "RacingEnv") -> dict:
    return {"params": None, "maps": (), "ref_path": None}


class RacingEnv:
    def __init__(self, device=torch.device("cuda"), dtype=torch.float32, seed: int = 42, native_step: bool = True) -> None:
        """`native_step`: on a GPU, step() runs the plant's batch-1 dynamics call and the goal test as ONE launch of
        the library's racing functor (mppi_model_step, library math in the reference's operation order) instead of
        ~20 batch-1 torch kernels; False keeps the torch ops of `dynamics` (the two agree to fp32 rounding)."""
        self._device, self._dtype = _device(device), dtype
        self._native_step = bool(native_step)
        self._step_args = None
        self.u_min = torch.tensor([-2.0, -0.25], device=self._device, dtype=dtype)  # [accel, steer]
        self.u_max = torch.tensor([2.0, 0.25], device=self._device, dtype=dtype)
        self.L = torch.tensor(1, device=self._device, dtype=dtype)
        self.V_MAX = torch.tensor(8.0, device=self._device, dtype=dtype)
        self.delta_t = 0.1
        self.dl = 0.1
        self.line_width = 6.5
        center = np.load(os.path.join(_DATA, "racing_center_path.npy"))  # [n,3] float64 (x, y, yaw)
        self.racing_center_path = torch.tensor(center, device=self._device, dtype=dtype)
        self.racing_center_path_np = self.racing_center_path.cpu().numpy()
        self.map_size, self.cell_size = (80, 80), 0.1
        self._lane_map = LaneMap(lane=center, lane_width=self.line_width * 0.8, map_size=self.map_size,
                                 cell_size=self.cell_size, device=self._device, dtype=dtype)
        self._obstacle_map = ObstacleMap(map_size=self.map_size, cell_size=self.cell_size, device=self._device,
                                         dtype=dtype)
        self._seed = seed
        generate_random_obstacles(self._obstacle_map, random_x_range=(-35, 35), random_y_range=(-35, 35),
                                  num_circle_obs=50, radius_range=(0.9, 1.2), num_rectangle_obs=0,
                                  width_range=(1.5, 2.0), height_range=(1.5, 2.0), max_iteration=1000, seed=seed)
        self._obstacle_map.convert_to_torch()
        self._x_lim = torch.tensor(self._obstacle_map.x_lim, device=self._device, dtype=dtype)
        self._y_lim = torch.tensor(self._obstacle_map.y_lim, device=self._device, dtype=dtype)
        self._start_pos = self.racing_center_path[0, :2].clone()
        self._goal_pos = self.racing_center_path[-1, :2].clone()
        self._robot_state = torch.zeros(4, device=self._device, dtype=dtype)
        self._dyn_params = None
        self.reset()

    def reset(self) -> torch.Tensor:
        d = self.racing_center_path[1, :2] - self._start_pos
        self._robot_state[:2] = self._start_pos
        self._robot_state[2] = angle_normalize(torch.atan2(d[1], d[0]))
        self._robot_state[3] = 0.0
        return self._robot_state

    def step(self, u: torch.Tensor) -> Tuple[torch.Tensor, bool]:
        """src/envs/racing_env.py:142-163: clamp the control, advance the plant one step, test the goal."""
        if self._native_step and self._device.type == "cuda" and torch.is_tensor(u) and u.is_cuda:
            return self._step_native(u)
        u = torch.clamp(u, self.u_min, self.u_max)
        self._robot_state = self.dynamics(self._robot_state.unsqueeze(0), u.unsqueeze(0)).squeeze(0)
        reached = torch.norm(self._robot_state[:2] - self._goal_pos) < 1.0
        return self._robot_state, reached

    def _step_native(self, u: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if self._step_args is None:
            self._step_args = NativeStep("racing", self.model_params([0.0] * 6), self.u_min, self.u_max, self._goal_pos,
                                         1.0, 4, self._device, self._dtype)
        self._robot_state, reached = self._step_args(self._robot_state, u)
        return self._robot_state, reached

    def model_params(self, weights) -> list:
        """MPPI_RP_* vector: dynamics constants here + the controller's six cost weights.  The env
        constants are read back from the device tensors once (no per-solve synchronisation)."""
        if self._dyn_params is None:
            m = self._obstacle_map
            self._dyn_params = [float(self.u_min[0]), float(self.u_max[0]), float(self.u_min[1]),
                                float(self.u_max[1]), float(self.L), float(self.V_MAX), self.delta_t,
                                m.x_lim[0], m.x_lim[1], m.y_lim[0], m.y_lim[1]]
        return [*self._dyn_params, *[float(w) for w in weights]]

    @native_model("racing", "dynamics", _racing_dyn_inputs)
    def dynamics(self, state: torch.Tensor, action: torch.Tensor, delta_t: float = 0.1) -> torch.Tensor:
        x, y, v = state[:, 0:1], state[:, 1:2], state[:, 3:4]
        theta = angle_normalize(state[:, 2:3])
        accel = torch.clamp(action[:, 0:1], self.u_min[0], self.u_max[0])
        steer = torch.clamp(action[:, 1:2], self.u_min[1], self.u_max[1])
        new_x = x + (v * torch.cos(theta)) * delta_t
        new_y = y + (v * torch.sin(theta)) * delta_t
        new_theta = angle_normalize(theta + (v * torch.tan(steer) / self.L) * delta_t)
        new_v = v + accel * delta_t
        new_x = torch.clamp(new_x, self._x_lim[0], self._x_lim[1])
        new_y = torch.clamp(new_y, self._y_lim[0], self._y_lim[1])
        new_v = torch.clamp(new_v, -self.V_MAX, self.V_MAX)
        return torch.cat([new_x, new_y, new_theta, new_v], dim=1)

    def collision_check(self, state: torch.Tensor) -> torch.Tensor:
        return self._obstacle_map.compute_cost(state[:, :, :2]).squeeze(1)

    def render(self, *args, **kwargs) -> None:  # UI: out of scope
        return None

    def close(self, *args, **kwargs) -> None:
        return None
