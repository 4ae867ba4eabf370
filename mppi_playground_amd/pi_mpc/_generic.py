"""The generic path of pi_mpc.mppi.MPPI (mix-in): opaque `dynamics` / `cost_func` callables run as the reference runs them
(src/pi_mpc/mppi.py:280-336: two T-step loops of batched torch ops), optionally captured once into hipGraphs; sampling,
softmax, reduction and warm start stay in the library."""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from pi_mpc._lazy import _ptr


class GenericPathMixin:
    def _generic_rollout_costs(self, state, info: Dict) -> None:
        """Steps 2-3 of forward() with the user's torch callables on GPU tensors, same call sequence and
        `info` protocol as the reference (src/pi_mpc/mppi.py:280-336); the summed costs go back to the
        library with mppi_set_costs.  With graph_callables the two loops are one hipGraph replay."""
        N, T = self._local_samples, self._horizon
        x0 = torch.as_tensor(np.asarray(state) if not torch.is_tensor(state) else state)
        x0 = x0.to(self._device, self._dtype)
        if self._x0_tensor is None:
            self._x0_tensor = torch.empty(self._dim_state, device=self._device, dtype=self._dtype)
        self._x0_tensor.copy_(x0)  # static buffers: the same storage every solve (what a captured graph replays on)
        # clamp(mean + eps) in the reference layout [N,T,dc]; the handle's warm start still holds the
        # mean of this solve (it is replaced by mppi_finalize)
        if self._state_seq_batch_buf is None:  # (the reference allocates `_state_seq_batch` once, too: mppi.py:168-174)
            self._state_seq_batch_buf = torch.zeros(N, T + 1, self._dim_state, device=self._device, dtype=self._dtype)
            self._generic_costs_keep = torch.empty(N, device=self._device, dtype=self._dtype)
        if self._perturbed_action_seqs_buf is None or not self._graph_callables:
            # a new `_perturbed_action_seqs` tensor every solve like the reference (mppi.py:266-275); a captured graph
            # needs the same storage every solve instead
            self._perturbed_action_seqs_buf = torch.empty(N, T, self._dim_control, device=self._device, dtype=self._dtype)
        U = self._perturbed_action_seqs_buf
        self._h.call("mppi_export_noise", None, _ptr(U), self._stream())
        if self._graph_state == "replay":
            self._check_replay_info(info)
            self._graph.replay()
        elif self._graph_state == "capture":
            self._capture_callables(info)
        else:
            self._callable_loops(info)
            if self._graph_state == "warmup":
                self._graph_state = "capture"  # the next solve captures (this one warmed the allocator / kernels up)
        self._h.call("mppi_set_costs", _ptr(self._generic_costs_keep), 1, self._stream())

    def _callable_loops(self, info: Dict) -> None:
        """src/pi_mpc/mppi.py:280-336 on the static buffers: S <- rollout of U from x0, total costs -> _generic_costs_keep."""
        N, T = self._local_samples, self._horizon
        U, S = self._perturbed_action_seqs_buf, self._state_seq_batch_buf
        S[:, 0, :] = self._x0_tensor.repeat(N, 1)
        for t in range(T):
            S[:, t + 1, :] = self._dynamics(S[:, t, :], U[:, t, :])
        costs = torch.zeros(N, T, device=self._device, dtype=self._dtype)
        initial_state = S[:, 0, :]
        for t in range(T):
            p = t - 1 if t > 0 else 0
            info["prev_state"] = S[:, p, :]
            info["prev_action"] = U[:, p, :]
            info["initial_state"] = initial_state
            info["t"] = t
            costs[:, t] = self._cost_func(S[:, t, :], U[:, t, :], info)
        info["prev_state"] = S[:, -2, :]
        zero_action = torch.zeros(N, self._dim_control, device=self._device, dtype=self._dtype)
        terminal = self._cost_func(S[:, -1, :], zero_action, info)
        self._generic_costs_keep.copy_(torch.sum(costs, dim=1) + terminal)

    def _states_prediction_graphed(self) -> torch.Tensor:
        """Step 8 (the batch-1 rollout of the solution through the user's dynamics, T launch-bound calls) as a second
        captured graph on static buffers; returns a fresh tensor like the eager path.  A dynamics that cannot be captured
        at batch 1 keeps the eager rollout (warned once), like the N-sample loops."""
        import warnings

        if self._graph_b1 is None:
            self._b1_actions = torch.empty(1, self._horizon, self._dim_control, device=self._device, dtype=self._dtype)
            self._b1_actions.copy_(self._action_out)
            try:
                side = torch.cuda.Stream(device=self._device)
                side.wait_stream(torch.cuda.current_stream(self._device))
                with torch.cuda.stream(side):  # warm-up at batch 1 on the stream the capture will use
                    self._states_prediction(self._x0_tensor, self._b1_actions)
                torch.cuda.current_stream(self._device).wait_stream(side)
                torch.cuda.synchronize(self._device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    self._b1_states = self._states_prediction(self._x0_tensor, self._b1_actions)
                self._graph_b1 = g
            except Exception as e:  # noqa: BLE001  not capturable at batch 1: stay eager for this step
                torch.cuda.synchronize(self._device)
                self._graph_b1 = False
                warnings.warn(f"graph_callables: the batch-1 rollout of the solution could not be captured "
                              f"({type(e).__name__}); it stays on the eager loop")
        if self._graph_b1 is False:
            return self._states_prediction(self._x0_tensor, self._action_out.repeat(1, 1, 1))
        self._b1_actions.copy_(self._action_out)
        self._graph_b1.replay()
        return self._b1_states.clone()

    def recapture(self) -> None:
        """graph_callables: drop the captured loops; the next solve runs eagerly (warm-up) and the one after captures
        again.  Call it after REBINDING anything the callables read (a new tensor object for a reference path, a
        changed Python scalar): a replay reads the storage that was captured — update tensors in place (`copy_`) to
        change what a captured graph sees without recapturing."""
        if self._graph_callables:
            self._graph = self._graph_b1 = None
            self._graph_info_keys = None
            self._graph_state = "warmup"

    def _capture_callables(self, info: Dict) -> None:
        """Capture _callable_loops into a hipGraph and run it once; on failure fall back to the eager loops for good."""
        import warnings

        try:
            torch.cuda.synchronize(self._device)
            side = torch.cuda.Stream(device=self._device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                self._callable_loops(info)
            self._graph = g
            self._graph_state = "replay"
            # what the caller's dict held besides the solver's own four keys when the loops were captured: a replay cannot
            # see later changes of it (see _check_replay_info)
            self._graph_info_keys = self._info_signature(info)
            g.replay()
        except Exception as e:  # not capturable (host sync, data-dependent shapes, ...): stay eager
            self._graph, self._graph_state = None, "failed"
            torch.cuda.synchronize(self._device)
            warnings.warn(f"graph_callables: the dynamics / cost_func loops could not be captured ({type(e).__name__}: "
                          f"{str(e).splitlines()[0] if str(e) else ''}); staying on the eager loops")
            self._callable_loops(info)

    @staticmethod
    def _info_signature(info: Dict) -> Dict:
        """What a captured graph saw of the CALLER's entries of `info`, by value: a tensor is its storage (address, shape,
        in-place version counter: a replay reads that storage, so an equal-valued NEW tensor is a change and an in-place
        update of the captured one is not — but bumps the version, which is allowed), a Python scalar / string / None is
        its value (a caller may rebuild an equal dict every tick), anything else its identity."""
        sig = {}
        for k, v in info.items():
            if k in ("prev_state", "prev_action", "initial_state", "t"):
                continue
            if torch.is_tensor(v):
                sig[k] = ("tensor", v.data_ptr(), tuple(v.shape), v.dtype)
            elif isinstance(v, (bool, int, float, str, bytes, type(None))):
                sig[k] = ("value", v)
            else:
                sig[k] = ("object", id(v))
        return sig

    def _check_replay_info(self, info: Dict) -> None:
        """A replayed graph ignores the `info` dict it is handed: the solver's own keys are views of the static buffers
        (filled in below like the eager loop leaves them), but entries the CALLER put there were read at capture time.  If
        those changed identity since, the replay would silently use the old objects: refuse instead."""
        now = self._info_signature(info)
        if now != self._graph_info_keys:
            raise RuntimeError("graph_callables: the caller's entries of `info` changed since the loops were captured "
                               f"({sorted(set(now) ^ set(self._graph_info_keys)) or sorted(now)}); update tensors in place "
                               "or call solver.recapture()")
        U, S = self._perturbed_action_seqs_buf, self._state_seq_batch_buf  # what the eager loop leaves in the dict
        info["prev_state"], info["prev_action"] = S[:, -2, :], U[:, max(self._horizon - 2, 0), :]
        info["initial_state"], info["t"] = S[:, 0, :], self._horizon - 1

    def _states_prediction(self, state: torch.Tensor, action_seqs: torch.Tensor) -> torch.Tensor:
        """src/pi_mpc/mppi.py:508-524 with the user's dynamics."""
        out = torch.zeros(action_seqs.shape[0], self._horizon + 1, self._dim_state, device=self._device,
                          dtype=self._dtype)
        out[:, 0, :] = state
        for t in range(self._horizon):
            out[:, t + 1, :] = self._dynamics(out[:, t, :], action_seqs[:, t, :])
        return out
