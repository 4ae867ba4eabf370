"""Racing MPC controller: contouring/lag/velocity/obstacle/input cost + reference-window selection.

Counterpart of `racing_controller` in the reference's example/racing.py (:16-218).  The cost is an
MPPI plugin (torch callable + native tag); `calc_ref_trajectory` is vectorised on the host.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch

from envs.obstacle_map_2d import _device
from pi_mpc.mppi import MPPI
from pi_mpc.native import native_model


def _racing_cost_inputs(ctrl: "racing_controller") -> dict:
    on_device = ctrl._window_on_device  # the window was written by the library itself (update_reference_window)
    if (not on_device and ctrl._reference_path_np is None) or ctrl.obstacle_map is None or ctrl.lane_map is None:
        raise ValueError("reference path, obstacle map, and lane map must be set before calling solve method.")
    weights = [ctrl.Qc, ctrl.Ql, ctrl.Qv, ctrl.Qo, ctrl.Qin, ctrl.Qdin]
    return {"params": ctrl.env.model_params(weights),
            "maps": [ctrl.obstacle_map.grid_spec(), ctrl.lane_map.grid_spec()],
            "ref_path": None if on_device else ctrl._reference_path_np}


class racing_controller:
    def __init__(self, env, debug=False, device=torch.device("cuda"), dtype=torch.float32, horizon: int = 25,
                 num_samples: int = 4000, lambda_: float = 1.0, mppi_cls=MPPI, **mppi_kwargs) -> None:
        """`mppi_cls`: the solver class to construct (default: the MI355X MPPI).  Anything with the reference's
        constructor / forward() contract works — the CPU baseline in bench.py passes its torch restatement of the
        reference loop here, so that it runs this very controller's cost function."""
        self.debug = debug
        self._window_on_device = False   # reference window + path index live in the library (device tick)
        self._path_index_host = 0
        self._center_on_device = None    # (path tensor, version) the library holds
        self.device_tick = True          # False: always take the host statement of calc_ref_trajectory
        self.env = env
        self.Qc, self.Ql, self.Qv = 2.0, 3.0, 2.0  # contouring, lag, velocity
        self.Qo, self.Qin, self.Qdin = 10000.0, 0.01, 0.5  # obstacle, input, input rate
        self._device, self._dtype = _device(device), dtype
        self._reference_path_t: torch.Tensor = None
        self._reference_path_np: np.ndarray = None
        self.obstacle_map = None
        self.lane_map = None
        self.solver = mppi_cls(horizon=horizon, num_samples=num_samples, dim_state=4, dim_control=2,
                           dynamics=env.dynamics, cost_func=self.cost_function, u_min=env.u_min, u_max=env.u_max,
                           sigmas=torch.tensor([0.5, 0.1]), lambda_=lambda_, **mppi_kwargs)

    # the reference's plain attributes (example/racing.py:31,57), here views of state that may live on the device
    @property
    def current_path_index(self) -> int:
        if self._window_on_device:
            return self.solver.path_index  # (synchronises: nothing in the control loop reads it)
        return self._path_index_host

    @current_path_index.setter
    def current_path_index(self, value: int) -> None:
        self._path_index_host = int(value)
        if self._center_on_device is not None:
            self.solver.path_index = int(value)

    @property
    def reference_path(self):
        if self._window_on_device:
            return self.solver.reference_window()
        return self._reference_path_t

    @reference_path.setter
    def reference_path(self, ref) -> None:
        self.set_reference(ref)

    def update(self, state: torch.Tensor, racing_center_path: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """example/racing.py:73-81.  With the state on the GPU the whole tick stays there: the reference window is
        rebuilt by one kernel from the state in device memory (MPPI.update_reference_window) and the solve follows on
        the same stream; otherwise the host statement of calc_ref_trajectory below runs."""
        T = self.solver._horizon
        if (self.device_tick and torch.is_tensor(state) and state.is_cuda and torch.is_tensor(racing_center_path)
                and getattr(self.solver, "_model", None) == "racing"):
            key = (racing_center_path, racing_center_path._version)
            held = self._center_on_device
            if held is None or held[0] is not key[0] or held[1] != key[1]:
                if self._window_on_device:  # the monotone `ind = max(cind, ind)` carry lives on the device: fetch it first
                    self._path_index_host = self.solver.path_index
                self.solver.set_center_path(racing_center_path.detach().cpu().numpy(),
                                            self._window_offsets(T, 0.1, 3, 0.85), self._v_max())
                self._center_on_device = key
                self.solver.path_index = self._path_index_host
            elif not self._window_on_device:  # coming back from a host tick: hand the index over
                self.solver.path_index = self._path_index_host
            self.solver.update_reference_window(state)
            self._window_on_device = True
            return self.solver.forward(state=state)
        cind = self.current_path_index
        self._window_on_device = False
        ref, self._path_index_host = self.calc_ref_trajectory(
            state, racing_center_path, cind, T, DL=0.1, lookahead_distance=3, reference_path_interval=0.85)
        self.set_reference(ref)
        return self.solver.forward(state=state)

    def set_reference(self, ref) -> None:
        if self._window_on_device:
            self._path_index_host = self.solver.path_index
        self._window_on_device = False
        self._reference_path_np = np.ascontiguousarray(
            ref.detach().cpu().numpy() if torch.is_tensor(ref) else ref, dtype=np.float32)
        self._reference_path_t = torch.as_tensor(self._reference_path_np)

    def _window_offsets(self, horizon: int, DL: float, lookahead_distance: float, reference_path_interval: float):
        """dind[i] of example/racing.py:201-205: travel = lookahead + (i+1) intervals accumulated one by one in float64,
        dind = int(round(travel / DL)) (round half to even) — a function of the arguments only."""
        cache = self.__dict__.setdefault("_ref_cache", {})
        key = (horizon, float(DL), float(lookahead_distance), float(reference_path_interval))
        dind = cache.get(key)
        if dind is None:
            travel = np.cumsum(np.concatenate([[float(lookahead_distance)],
                                               np.full(horizon + 1, float(reference_path_interval))]))[1:]
            dind = cache[key] = np.rint(travel / DL).astype(np.int64)
        return dind

    def _v_max(self) -> float:
        """env.V_MAX (a device scalar) read back once per tensor value, not once per tick."""
        cache = self.__dict__.setdefault("_ref_cache", {})
        v = self.env.V_MAX
        key = (id(v), getattr(v, "_version", None))
        hit = cache.get("v_max")
        if hit is None or hit[0] != key:
            hit = cache["v_max"] = (key, float(v))
        return hit[1]

    def get_top_samples(self, num_samples=300):
        return self.solver.get_top_samples(num_samples=num_samples)

    def set_cost_map(self, obstacle_map, lane_map) -> None:
        self.obstacle_map, self.lane_map = obstacle_map, lane_map

    @native_model("racing", "cost", _racing_cost_inputs)
    def cost_function(self, state: torch.Tensor, action: torch.Tensor, info: dict) -> torch.Tensor:
        ref = self.reference_path.to(state.device)[info["t"]]
        prev_action = info["prev_action"]
        sp, cp = torch.sin(ref[2]), torch.cos(ref[2])
        ex, ey = state[:, 0] - ref[0], state[:, 1] - ref[1]
        ec = sp * ex - cp * ey
        el = -cp * ex - sp * ey
        path_cost = self.Qc * ec.pow(2) + self.Ql * el.pow(2)
        velocity_cost = self.Qv * (state[:, 3] - ref[3]).pow(2)
        pos = state[:, :2].unsqueeze(1)
        occ = self.obstacle_map.compute_cost(pos).squeeze(1) + self.lane_map.compute_cost(pos).squeeze(1)
        input_cost = self.Qin * action.pow(2).sum(dim=1)
        input_cost = input_cost + self.Qdin * (action - prev_action).pow(2).sum(dim=1)
        return path_cost + velocity_cost + self.Qo * occ + input_cost

    def calc_ref_trajectory(self, state, path, cind: int, horizon: int, DL=0.1, lookahead_distance=1.0,
                            reference_path_interval=0.5):
        """Reference window [horizon+1, 4] = (x, y, yaw, v_target) ahead of the nearest path point
        (example/racing.py:161-218).  Nearest-point search is one vectorised fp32 hypot + argmin (first minimum, like
        the reference's Python min over indices); the window rows are one gather.  What does not change between ticks
        is kept: the host copy of the centre line (the reference re-reads it element by element from the device every
        tick), the row offsets, V_MAX."""
        cache = self.__dict__.setdefault("_ref_cache", {})
        if torch.is_tensor(path):
            hit = cache.get("path")
            if hit is None or hit[0] is not path or hit[1] != path._version:
                hit = cache["path"] = (path, path._version, path.detach().cpu().numpy().astype(np.float32))
            p = hit[2]
        else:
            p = np.asarray(path, np.float32)
        s = state.detach().cpu().numpy().astype(np.float32) if torch.is_tensor(state) else np.asarray(state, np.float32)
        ncourse = len(p)
        ind = int(np.argmin(np.hypot(p[:, 0] - s[0], p[:, 1] - s[1])))
        ind = max(cind, ind)
        dind = self._window_offsets(horizon, DL, lookahead_distance, reference_path_interval)
        idx = ind + dind
        inside = idx < ncourse
        xref = np.zeros((horizon + 1, s.shape[0]), np.float32)
        xref[:, :3] = p[np.where(inside, idx, ncourse - 1)]
        if inside.all():  # past the end of the course the reference zeroes the whole target-velocity column: stop
            xref[:, 3] = self._v_max()
        return torch.from_numpy(xref), ind
