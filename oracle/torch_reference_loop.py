"""TEST / BASELINE INFRASTRUCTURE — never imported by the product (mppi_playground_amd/).

A torch-CPU restatement of the reference's MPPI.forward() with the reference's OWN op structure
(/root/reference/src/pi_mpc/mppi.py:223-460): one [N,T,dc] normal draw from torch's global generator, the
exploration split + clamp, a Python loop of T batched `dynamics` calls writing strided [N,ds] slices of
S[N,T+1,ds], a second Python loop of T batched `cost_func` calls with the reference's `info` protocol — including
the dead `mean[t] @ inv_cov[t] @ U[:,t].T` product the reference computes and never uses (:312-316) — the terminal
cost with its stale `t` / `prev_action`, softmax weights, the weighted sum and the batch-1 rollout.

Why it exists: SURVEY.md section 8(d) / BASELINE.md section 3 name "the build's own torch-CPU restatement of the
reference algorithm (same op structure), all cores" as the CPU baseline to time on the GPU box's host, next to the C
port (oracle/mppi_oracle.c) — the Python reference itself cannot travel.  `bench.py` times it (`cpu_baseline_torch`);
tests/test_oracle_vs_golden.py pins it against the reference fixtures (same global seed -> same noise stream, same
actions), so what is timed is the reference's arithmetic.

The plugins it drives are the product's torch callables (envs/*: same contract as the reference's), on CPU tensors.
"""
from __future__ import annotations

import os
import sys
from typing import Callable, Dict, Optional, Tuple

import numpy as np
import torch

_PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mppi_playground_amd")
if _PKG not in sys.path:  # envs/ and pi_mpc/_host.py (numpy statements of the temperature rules / SG filter)
    sys.path.append(_PKG)


class TorchReferenceLoop:
    """Constructor arguments follow the reference's MPPI (mppi.py:24-47); CPU only, fp32."""

    def __init__(self, horizon: int, num_samples: int, dim_state: int, dim_control: int,
                 dynamics: Callable, cost_func: Callable, u_min: torch.Tensor, u_max: torch.Tensor,
                 sigmas: torch.Tensor, lambda_, essps_target_ess: Optional[float] = None, lambda_min: float = 0.01,
                 lambda_max: float = 10.0, exploration: float = 0.0, use_sg_filter: bool = False,
                 sg_window_size: int = 5, sg_poly_order: int = 3, seed: int = 42, dead_action_cost: bool = True,
                 **_ignored) -> None:
        from pi_mpc import _host

        self._host = _host
        self._horizon, self._num_samples = horizon, num_samples
        self._dim_state, self._dim_control = dim_state, dim_control
        self._dynamics, self._cost_func = dynamics, cost_func
        f32 = torch.float32
        self._u_min, self._u_max = u_min.detach().cpu().to(f32), u_max.detach().cpu().to(f32)
        self._sigmas = sigmas.detach().cpu().to(f32)
        self._exploration = exploration
        self._lambda = lambda_
        self._auto = lambda_ if isinstance(lambda_, str) else None
        if self._auto not in (None, "ESSPS") or (self._auto is None and not isinstance(lambda_, float)):
            raise ValueError("TorchReferenceLoop: fixed lambda or 'ESSPS'")
        self._target_ess = essps_target_ess if essps_target_ess is not None else num_samples / 10
        self._lambda_min, self._lambda_max = lambda_min, lambda_max
        self._use_sg = use_sg_filter
        self._coeffs = _host.savitzky_golay_coeffs(sg_window_size, sg_poly_order)
        self._sg_hist = np.zeros((horizon - 1, dim_control), np.float32)
        self._dead = dead_action_cost
        torch.manual_seed(seed)  # the reference seeds the GLOBAL generator (mppi.py:93) ...
        self._inv_cov = torch.zeros(horizon, dim_control, dim_control)  # (:132-137: row 0 stays zero)
        for t in range(1, horizon):
            self._inv_cov[t] = torch.diag(1.0 / self._sigmas ** 2)
        self._action_noises = self._draw()  # ... and its constructor consumes one draw (:146-148)
        self._previous_action_seq = torch.zeros(horizon, dim_control)
        self._state_seq_batch = torch.zeros(num_samples, horizon + 1, dim_state)
        self._weights = torch.zeros(num_samples)

    def _draw(self) -> torch.Tensor:
        # MultivariateNormal(0, diag(sigma^2)).rsample([N]) == randn(N,T,dc) * sigma bit for bit (SURVEY B-Q1)
        return torch.randn(self._num_samples, self._horizon, self._dim_control) * self._sigmas

    def forward(self, state, info: Dict = {}) -> Tuple[torch.Tensor, torch.Tensor]:
        N, T, dc = self._num_samples, self._horizon, self._dim_control
        state = torch.as_tensor(np.asarray(state) if not torch.is_tensor(state) else state).detach().cpu().float()
        assert state.shape == (self._dim_state,)
        mean = self._previous_action_seq.clone()
        # step 1 (:255-275)
        eps = self._action_noises = self._draw()
        thr = int(N * (1 - self._exploration))
        U = torch.clamp(torch.cat([mean + eps[:thr], eps[thr:]]), self._u_min, self._u_max)
        self._perturbed_action_seqs = U
        # step 2 (:280-286)
        S = self._state_seq_batch
        S[:, 0, :] = state.repeat(N, 1)
        for t in range(T):
            S[:, t + 1, :] = self._dynamics(S[:, t, :], U[:, t, :])
        # step 3 (:291-336)
        costs = torch.zeros(N, T)
        action_costs = torch.zeros(N, T)
        initial_state = S[:, 0, :]
        for t in range(T):
            p = t - 1 if t > 0 else 0
            info["prev_state"], info["prev_action"] = S[:, p, :], U[:, p, :]
            info["initial_state"], info["t"] = initial_state, t
            costs[:, t] = self._cost_func(S[:, t, :], U[:, t, :], info)
            if self._dead:  # computed and discarded by the reference (:312-316,335)
                action_costs[:, t] = mean[t] @ self._inv_cov[t] @ U[:, t].T
        info["prev_state"] = S[:, -2, :]
        terminal = self._cost_func(S[:, -1, :], torch.zeros(N, dc), info)
        c = torch.sum(costs, dim=1) + terminal
        self._costs = c
        # step 4 (:351-370)
        if self._auto == "ESSPS":
            self._lambda = self._host.essps_lambda(c.numpy(), self._target_ess, self._lambda_min, self._lambda_max)
        # steps 5-6 (:376-385)
        w = self._weights = torch.softmax(-c / self._lambda, dim=0)
        a = torch.sum(w.view(N, 1, 1) * U, dim=0)
        # step 7 (:423-443)
        if self._use_sg:
            a = torch.from_numpy(self._host.sg_filter_sequence(self._sg_hist, a.numpy(), self._coeffs))
        # step 8 (:448-458)
        s = torch.zeros(1, T + 1, self._dim_state)
        s[:, 0, :] = state
        for t in range(T):
            s[:, t + 1, :] = self._dynamics(s[:, t, :], a[t].view(1, dc))
        self._previous_action_seq = a
        self._sg_hist = np.concatenate([self._sg_hist[1:], a[0].numpy()[None, :]])
        return a, s
