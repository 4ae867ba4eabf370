"""Pendulum, cart-pole and mountain-car models as MPPI plugins.

The reference defines these as TorchScript closures inside example/pendulum.py:17-47,
example/cartpole.py:17-81 and example/mountaincar.py:17-55; here they are module-level torch
functions with the same contract plus a native tag.
"""
from __future__ import annotations

import torch

from envs.common import angle_normalize
from pi_mpc.native import native_model


# ------------------------------------------------------------------ pendulum (gymnasium Pendulum-v1)
@native_model("pendulum", "dynamics")
def pendulum_dynamics(state: torch.Tensor, action: torch.Tensor) -> torch.Tensor:
    th, thdot = state[:, 0:1], state[:, 1:2]
    g, m, length, dt = 10.0, 1.0, 1.0, 0.05
    u = torch.clamp(action[:, 0:1], -2, 2)
    newthdot = thdot + (-3 * g / (2 * length) * torch.sin(th + torch.pi) + 3.0 / (m * length ** 2) * u) * dt
    newth = th + newthdot * dt
    return torch.cat((newth, torch.clamp(newthdot, -8, 8)), dim=1)


@native_model("pendulum", "cost")
def pendulum_cost(state: torch.Tensor, action: torch.Tensor, info) -> torch.Tensor:
    return angle_normalize(state[:, 0]) ** 2 + 0.1 * state[:, 1] ** 2


# ------------------------------------------------------------------ cart-pole (gymnasium CartPole-v1)
@native_model("cartpole", "dynamics")
def cartpole_dynamics(state: torch.Tensor, action: torch.Tensor) -> torch.Tensor:
    x, x_dt, theta, theta_dt = state[:, 0:1], state[:, 1:2], state[:, 2:3], state[:, 3:4]
    gravity, masscart, masspole, length, force_mag, tau = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
    total_mass, polemass_length = masspole + masscart, masspole * length
    a = action[:, 0:1]
    force = torch.zeros_like(a)
    force[a >= 0] = force_mag  # bang-bang: the simulator's action is discrete
    force[a < 0] = -force_mag
    cos, sin = torch.cos(theta), torch.sin(theta)
    temp = (force + polemass_length * theta_dt ** 2 * sin) / total_mass
    thetaacc = (gravity * sin - cos * temp) / (length * (4.0 / 3.0 - masspole * cos ** 2 / total_mass))
    xacc = temp - polemass_length * thetaacc * cos / total_mass
    newx = torch.clamp(x + tau * x_dt, -2.4, 2.4)
    lim = 12 * 2 * torch.pi / 360
    newtheta = torch.clamp(theta + tau * theta_dt, -lim, lim)
    return torch.cat((newx, x_dt + tau * xacc, newtheta, theta_dt + tau * thetaacc), dim=1)


@native_model("cartpole", "cost")
def cartpole_cost(state: torch.Tensor, action: torch.Tensor, info) -> torch.Tensor:
    return angle_normalize(state[:, 2]) ** 2 + 0.1 * state[:, 3] ** 2 + 0.1 * state[:, 0] ** 2


# ------------------------------------------------------------------ mountain car (MountainCarContinuous-v0)
@native_model("mountaincar", "dynamics")
def mountaincar_dynamics(state: torch.Tensor, action: torch.Tensor) -> torch.Tensor:
    """NOTE: like the reference closure this updates `state` in place through views (position and
    velocity end up holding the un-clamped updates); the native kernel reproduces that behaviour."""
    position, velocity = state[:, 0].view(-1, 1), state[:, 1].view(-1, 1)
    force = torch.clamp(action[:, 0].view(-1, 1), -1.0, 1.0)
    velocity += force * 0.0015 - 0.0025 * torch.cos(3 * position)
    velocity = torch.clamp(velocity, -0.07, 0.07)
    position += velocity
    position = torch.clamp(position, -1.2, 0.6)
    return torch.cat((position, velocity), dim=1)


@native_model("mountaincar", "cost")
def mountaincar_cost(state: torch.Tensor, action: torch.Tensor, info) -> torch.Tensor:
    return (0.45 - state[:, 0]) ** 2


# ------------------------------------------------------------------ MuJoCo-style cart-pole (InvertedPendulum-v4)
@native_model("mjcartpole", "dynamics")
def mjcartpole_dynamics(state: torch.Tensor, action: torch.Tensor) -> torch.Tensor:
    """Continuous force, pole mass 1 (the reference's example/mujoco_cartpole.py:20-68)."""
    x, x_dt, theta, theta_dt = state[:, 0:1], state[:, 1:2], state[:, 2:3], state[:, 3:4]
    force = action[:, 0:1]
    gravity, masscart, masspole, length, tau = 9.8, 1.0, 1.0, 0.5, 0.02
    total_mass, polemass_length = masspole + masscart, masspole * length
    cos, sin = torch.cos(theta), torch.sin(theta)
    temp = (force + polemass_length * theta_dt ** 2 * sin) / total_mass
    thetaacc = (gravity * sin - cos * temp) / (length * (4.0 / 3.0 - masspole * cos ** 2 / total_mass))
    xacc = temp - polemass_length * thetaacc * cos / total_mass
    newx = torch.clamp(x + tau * x_dt, -1.0, 1.0)
    lim = 12 * 2 * torch.pi / 360
    newtheta = torch.clamp(theta + tau * theta_dt, -lim, lim)
    return torch.cat((newx, x_dt + tau * xacc, newtheta, theta_dt + tau * thetaacc), dim=1)


@native_model("mjcartpole", "cost")
def mjcartpole_cost(state: torch.Tensor, action: torch.Tensor, info) -> torch.Tensor:
    return angle_normalize(state[:, 2]) ** 2 + 0.1 * state[:, 3] ** 2 + 0.1 * state[:, 0] ** 2
