"""Goal inside a circular danger zone: 7-state unicycle (x, y, heading, vector to the goal, vector to the
zone centre) with a distance-to-goal + zone-penalty cost.

Counterpart of the reference's src/envs/goal_in_danger_zone.py (model :113-156, episode logic
:158-230) without gymnasium and without rendering.  `parallel_step` / `parallel_cost` are the MPPI
plugins (torch callables + native tag).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch

from envs.common import angle_normalize
from pi_mpc.native import native_model


class DangerZone:
    def __init__(self, radius: float = 10.0, center=(0.0, 0.0)):
        self.radius, self.center = float(radius), [float(center[0]), float(center[1])]

    def _point(self, r_lo, r_hi):
        angle = np.random.uniform(0, 2 * np.pi)
        radius = np.random.uniform(r_lo, r_hi)
        return np.array([radius * np.cos(angle) + self.center[0], radius * np.sin(angle) + self.center[1]])

    def get_random_inside_point(self):
        return self._point(0, self.radius)

    def get_random_outside_point(self):
        return self._point(self.radius, 2 * self.radius)

    def is_inside(self, pos) -> bool:
        return bool(np.linalg.norm(np.asarray(pos) - self.center) < self.radius)


def _goalzone_inputs(env: "GoalInDangerZoneEnv") -> dict:
    g, c = np.float32(env._goal), np.float32(env._danger_zone.center)
    return {"params": [env._v_min, env._v_max, env._omega_min, env._omega_max, env._dt, float(g[0]), float(g[1]),
                       float(c[0]), float(c[1]), env._danger_zone.radius, env.zone_penalty],
            "maps": (), "ref_path": None}


class GoalInDangerZoneEnv:
    def __init__(self, seed: int = 42, cfg: dict = {"shape": "circle", "radius": 10.0, "center": [0.0, 0.0]}):
        if cfg.get("shape", "circle") != "circle":
            raise ValueError(f"Invalid shape: {cfg['shape']}")
        self._danger_zone = DangerZone(cfg["radius"], cfg["center"])
        self._v_max, self._omega_max, self._v_min, self._omega_min = 1.0, 1.0, -1.0, -1.0
        self._dt = 0.1
        self.zone_penalty = 1000.0
        self.max_episode_steps = 100
        self._goal = np.zeros(2)
        self._pos, self._angle = np.zeros(2), 0.0
        self._steps = 0

    def reset(self, seed: int = None) -> Tuple[np.ndarray, dict]:
        if seed is not None:
            np.random.seed(seed)
        self._goal = self._danger_zone.get_random_inside_point()
        self._pos = self._danger_zone.get_random_outside_point()
        self._angle = np.random.uniform(-np.pi, np.pi)
        self._steps = 0
        return self._obs(), {}

    def _obs(self) -> np.ndarray:
        return np.concatenate([self._pos, [self._angle], self._goal - self._pos,
                               np.array(self._danger_zone.center) - self._pos]).astype(np.float32)

    def step(self, action: np.ndarray):
        v = float(np.clip(action[0], self._v_min, self._v_max))
        omega = float(np.clip(action[1], self._omega_min, self._omega_max))
        prev = np.linalg.norm(self._goal - self._pos)
        self._angle = float(angle_normalize(torch.tensor(self._angle + omega * self._dt)))
        self._pos = self._pos + v * np.array([np.cos(self._angle), np.sin(self._angle)]) * self._dt
        dist = np.linalg.norm(self._goal - self._pos)
        self._steps += 1
        cost = 1.0 if self._danger_zone.is_inside(self._pos) else 0.0
        terminated = bool(dist < 0.5)
        truncated = self._steps >= self.max_episode_steps
        return self._obs(), float(prev - dist), terminated, truncated, {"cost": cost}

    @native_model("goalzone", "dynamics", _goalzone_inputs)
    def parallel_step(self, obs: torch.Tensor, action: torch.Tensor) -> torch.Tensor:
        x, y, theta = obs[:, 0:1], obs[:, 1:2], obs[:, 2:3]
        v = torch.clamp(action[:, 0:1], self._v_min, self._v_max)
        omega = torch.clamp(action[:, 1:2], self._omega_min, self._omega_max)
        theta = angle_normalize(theta + omega * self._dt)  # heading first, then move along it
        new_x = x + v * torch.cos(theta) * self._dt
        new_y = y + v * torch.sin(theta) * self._dt
        pos = torch.cat((new_x, new_y), dim=-1)
        goal = torch.tensor(self._goal, device=obs.device, dtype=obs.dtype)
        center = torch.tensor(self._danger_zone.center, device=obs.device, dtype=obs.dtype)
        return torch.cat((new_x, new_y, theta, goal - pos, center - pos), dim=-1)

    @native_model("goalzone", "cost", _goalzone_inputs)
    def parallel_cost(self, obs: torch.Tensor, action: torch.Tensor, info: dict) -> torch.Tensor:
        cost = torch.norm(obs[:, 3:5], dim=-1)
        inside = torch.norm(obs[:, 5:7], dim=-1) < self._danger_zone.radius
        return cost + inside.float() * self.zone_penalty

    def render(self, *args, **kwargs) -> None:  # UI: out of scope
        return None

    def set_render_info(self, *args, **kwargs) -> None:
        return None

    def close(self) -> None:
        return None
