"""Lane-corridor occupancy grid: cells farther than lane_width/2 from the centre line are blocked.

Counterpart of the reference's src/envs/lane_map_2d.py (:48-88 construction via a Euclidean distance
transform, :90-122 lookup); pinned against the reference's 800x800 racing lane map.
"""
from __future__ import annotations

from math import ceil
from typing import Tuple

import numpy as np
import torch
from scipy.ndimage import distance_transform_edt

from envs.obstacle_map_2d import _device, grid_lookup
from pi_mpc.native import GridSpec


def largest_square_within(max_distance: float) -> int:
    """Largest integer k with float64 sqrt(k) <= max_distance: `edt <= max_distance` (lane_map_2d.py:80-82)
    on a grid whose EDT values are sqrt(integer) becomes `d2 <= k`."""
    k = int(max_distance * max_distance)
    while np.sqrt(np.float64(k + 1)) <= max_distance:
        k += 1
    while k >= 0 and np.sqrt(np.float64(k)) > max_distance:
        k -= 1
    return k


class LaneMap:
    def __init__(self, lane: np.ndarray, lane_width: float, map_size: Tuple[int, int] = (20, 20),
                 cell_size: float = 0.01, device=torch.device("cuda"), dtype=torch.float32) -> None:
        assert lane_width > 0
        assert lane.ndim == 2 and lane.shape[1] == 3
        self._device, self._dtype = _device(device), dtype
        nx, ny = ceil(map_size[0] / cell_size), ceil(map_size[1] / cell_size)
        self._cell_size = cell_size
        self._cell_map_origin = np.array([nx // 2, ny // 2])
        self._torch_cell_map_origin = torch.from_numpy(self._cell_map_origin).to(self._device, self._dtype)
        self.x_lim = [-map_size[0] / 2, map_size[0] / 2]
        self.y_lim = [-map_size[1] / 2, map_size[1] / 2]

        seeds = np.ones((nx, ny))
        # python round() (half to even) of the float64 cell coordinate, like the reference
        cx = np.array([int(round(v / cell_size)) for v in lane[:, 0]]) + self._cell_map_origin[0]
        cy = np.array([int(round(v / cell_size)) for v in lane[:, 1]]) + self._cell_map_origin[1]
        ok = (cx >= 0) & (cx < nx) & (cy >= 0) & (cy < ny)
        seeds[cx[ok], cy[ok]] = 0
        dist = distance_transform_edt(seeds)
        max_distance = (lane_width / 2) / cell_size
        self._map = np.where(dist <= max_distance, 0, 1)
        # the same test on integer squared cell distances, for the device builder
        self._seed_cells = np.unique(np.stack([cx[ok], cy[ok]], axis=1), axis=0).astype(np.int32)
        self._max_d2 = largest_square_within(max_distance)
        self._map_torch = torch.tensor(self._map, device=self._device, dtype=self._dtype)
        self._spec = GridSpec(np.ascontiguousarray(self._map != 0, dtype=np.uint8), float(cell_size),
                              (float(self._cell_map_origin[0]), float(self._cell_map_origin[1])), 0,
                              {"kind": "lane", "seeds": self._seed_cells, "max_d2": self._max_d2})

    def grid_spec(self) -> GridSpec:
        return self._spec

    def compute_cost(self, x: torch.Tensor) -> torch.Tensor:
        if x.device != self._device or x.dtype != self._dtype:
            x = x.to(self._device, self._dtype)
        return grid_lookup(self._map_torch, x, self._cell_size, self._torch_cell_map_origin,
                           (float(self._cell_map_origin[0]), float(self._cell_map_origin[1])))
