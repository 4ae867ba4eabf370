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
    if ctrl.reference_path is None or ctrl.obstacle_map is None or ctrl.lane_map is None:
        raise ValueError("reference path, obstacle map, and lane map must be set before calling solve method.")
    weights = [ctrl.Qc, ctrl.Ql, ctrl.Qv, ctrl.Qo, ctrl.Qin, ctrl.Qdin]
    return {"params": ctrl.env.model_params(weights),
            "maps": [ctrl.obstacle_map.grid_spec(), ctrl.lane_map.grid_spec()],
            "ref_path": ctrl._reference_path_np}


class racing_controller:
    def __init__(self, env, debug=False, device=torch.device("cuda"), dtype=torch.float32, horizon: int = 25,
                 num_samples: int = 4000, lambda_: float = 1.0, mppi_cls=MPPI, **mppi_kwargs) -> None:
        """`mppi_cls`: the solver class to construct (default: the MI355X MPPI).  Anything with the reference's
        constructor / forward() contract works — the CPU baseline in bench.py passes its torch restatement of the
        reference loop here, so that it runs this very controller's cost function."""
        self.debug = debug
        self.current_path_index = 0
        self.env = env
        self.Qc, self.Ql, self.Qv = 2.0, 3.0, 2.0  # contouring, lag, velocity
        self.Qo, self.Qin, self.Qdin = 10000.0, 0.01, 0.5  # obstacle, input, input rate
        self._device, self._dtype = _device(device), dtype
        self.reference_path: torch.Tensor = None
        self._reference_path_np: np.ndarray = None
        self.obstacle_map = None
        self.lane_map = None
        self.solver = mppi_cls(horizon=horizon, num_samples=num_samples, dim_state=4, dim_control=2,
                           dynamics=env.dynamics, cost_func=self.cost_function, u_min=env.u_min, u_max=env.u_max,
                           sigmas=torch.tensor([0.5, 0.1]), lambda_=lambda_, **mppi_kwargs)

    def update(self, state: torch.Tensor, racing_center_path: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        ref, self.current_path_index = self.calc_ref_trajectory(
            state, racing_center_path, self.current_path_index, self.solver._horizon, DL=0.1,
            lookahead_distance=3, reference_path_interval=0.85)
        self.set_reference(ref)
        return self.solver.forward(state=state)

    def set_reference(self, ref) -> None:
        self._reference_path_np = np.ascontiguousarray(
            ref.detach().cpu().numpy() if torch.is_tensor(ref) else ref, dtype=np.float32)
        self.reference_path = torch.as_tensor(self._reference_path_np)

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
        # index offsets of the window rows: travel = lookahead + (i+1) intervals accumulated one by one in float64,
        # dind = int(round(travel / DL)) (round half to even) — a function of the arguments only
        key = (horizon, float(DL), float(lookahead_distance), float(reference_path_interval))
        dind = cache.get(key)
        if dind is None:
            travel = np.cumsum(np.concatenate([[float(lookahead_distance)],
                                               np.full(horizon + 1, float(reference_path_interval))]))[1:]
            dind = cache[key] = np.rint(travel / DL).astype(np.int64)
        if "v_max" not in cache:
            cache["v_max"] = float(self.env.V_MAX)  # (a device scalar: read back once)
        idx = ind + dind
        inside = idx < ncourse
        xref = np.zeros((horizon + 1, s.shape[0]), np.float32)
        xref[:, :3] = p[np.where(inside, idx, ncourse - 1)]
        if inside.all():  # past the end of the course the reference zeroes the whole target-velocity column: stop
            xref[:, 3] = cache["v_max"]
        return torch.from_numpy(xref), ind
