"""Occupancy-grid obstacle map: host-side construction (numpy) + torch lookup + native GridSpec.

Counterpart of the reference's src/envs/obstacle_map_2d.py (construction semantics :46-162,
lookup :168-200, random placement :235-345), written for this build; pinned against the reference's
maps by tests/test_host_logic.py.
"""
from __future__ import annotations

from math import ceil
from typing import List, Sequence, Tuple

import numpy as np
import torch

from mppi_playground_amd._pool import RowPool

from pi_mpc.native import GridSpec


def _device(device) -> torch.device:
    # same rule as the reference (obstacle_map_2d.py:63-67): cuda iff available and requested
    if torch.cuda.is_available() and torch.device(device) == torch.device("cuda"):
        return torch.device("cuda")
    return torch.device("cpu")


class ObstacleMap:
    """Grid of 0/1 cells over [-w/2, w/2] x [-h/2, h/2]; first index is x."""

    def __init__(self, map_size: Tuple[int, int] = (20, 20), cell_size: float = 0.01,
                 device=torch.device("cuda"), dtype=torch.float32) -> None:
        assert len(map_size) == 2 and cell_size > 0
        assert map_size[0] % 2 == 0 and map_size[1] % 2 == 0
        self._device, self._dtype = _device(device), dtype
        nx, ny = ceil(map_size[0] / cell_size), ceil(map_size[1] / cell_size)
        self._map = np.zeros((nx, ny))
        self._cell_size = cell_size
        self._cell_map_origin = np.array([nx / 2, ny / 2]).astype(int)
        self._torch_cell_map_origin = torch.from_numpy(self._cell_map_origin).to(self._device, self._dtype)
        self.x_lim = [-cell_size * nx / 2, cell_size * nx / 2]
        self.y_lim = [-cell_size * ny / 2, cell_size * ny / 2]
        self._map_torch: torch.Tensor = None
        self.circle_obs_list: List[Tuple[np.ndarray, float]] = []
        self.rectangle_obs_list: List[Tuple[np.ndarray, float, float]] = []
        self._version = 0
        self._spec = None
        self._circle_cells: List[Tuple[int, int, int]] = []    # (ci, cj, r) in cells, for the device rasteriser
        self._rect_cells: List[Tuple[int, int, int, int]] = []  # clipped (x0, x1, y0, y1)

    # -- rasterisers ---------------------------------------------------------------------------
    def add_circle_obstacle(self, center: np.ndarray, radius: float) -> None:
        assert len(center) == 2 and radius > 0
        c = np.round(center / self._cell_size + self._cell_map_origin).astype(int)
        r = ceil(radius / self._cell_size)
        off = np.arange(-r, r + 1)
        ii, jj = np.meshgrid(off, off, indexing="ij")
        inside = ii ** 2 + jj ** 2 <= r ** 2
        xi = np.clip(c[0] + ii[inside], 0, self._map.shape[0] - 1)
        yi = np.clip(c[1] + jj[inside], 0, self._map.shape[1] - 1)
        self._map[xi, yi] = 1
        self._circle_cells.append((int(c[0]), int(c[1]), int(r)))
        self.circle_obs_list.append((np.asarray(center, float), float(radius)))
        self._touch()

    def add_rectangle_obstacle(self, center: np.ndarray, width: float, height: float) -> None:
        assert len(center) == 2 and width > 0 and height > 0
        c = np.ceil(center / self._cell_size + self._cell_map_origin).astype(int)
        hw = ceil(ceil(width / self._cell_size) / 2)
        hh = ceil(ceil(height / self._cell_size) / 2)
        x0, x1 = np.clip([c[0] - hw, c[0] + hw], 0, self._map.shape[0] - 1)
        y0, y1 = np.clip([c[1] - hh, c[1] + hh], 0, self._map.shape[1] - 1)
        self._map[x0:x1, y0:y1] = 1
        self._rect_cells.append((int(x0), int(x1), int(y0), int(y1)))
        self.rectangle_obs_list.append((np.asarray(center, float), float(width), float(height)))
        self._touch()

    def _touch(self):
        self._version += 1
        self._spec = None

    # -- device views --------------------------------------------------------------------------
    def convert_to_torch(self) -> torch.Tensor:
        self._map_torch = torch.from_numpy(self._map).to(self._device, self._dtype)
        return self._map_torch

    def grid_spec(self) -> GridSpec:
        """Plain host description for the native path (cached per version; the solver rasterises the integer
        recipe on the device, `cells` is the host-built twin)."""
        if self._spec is None:
            cells = np.ascontiguousarray(self._map != 0, dtype=np.uint8)
            recipe = {"kind": "obstacles", "circles": np.asarray(self._circle_cells, np.int32).reshape(-1, 3),
                      "rects": np.asarray(self._rect_cells, np.int32).reshape(-1, 4)}
            self._spec = GridSpec(cells, float(self._cell_size),
                                      (float(self._cell_map_origin[0]), float(self._cell_map_origin[1])),
                                      self._version, recipe)
        return self._spec

    def compute_cost(self, x: torch.Tensor) -> torch.Tensor:
        """x [B, L, 2] -> occupancy [B, L]; out of the grid counts as occupied."""
        assert self._map_torch is not None
        if x.device != self._device or x.dtype != self._dtype:
            x = x.to(self._device, self._dtype)
        return grid_lookup(self._map_torch, x, self._cell_size, self._torch_cell_map_origin,
                           (float(self._cell_map_origin[0]), float(self._cell_map_origin[1])))

    def render(self, *args, **kwargs) -> None:  # UI: out of scope
        return None


_native = None  # (binding module, library handle) once a device lookup has been asked for
_POOLED_POINTS = 4096
_out_pools = {}  # (shape, device index) -> RowPool of small outputs
_stride_of_layout = {}  # (shape, strides) -> _uniform_point_stride


def _uniform_point_stride(x: torch.Tensor):
    """Floats between consecutive [..., 2] points of `x` when they are evenly spaced in memory (a contiguous tensor, the
    [:, :, :2] view of contiguous state rows, S[:, t, None, :2] of a state buffer, ...), else None."""
    if x.stride(-1) != 1:
        return None
    dims = [(x.size(d), x.stride(d)) for d in range(x.dim() - 1) if x.size(d) != 1]  # outer -> inner, size-1 dims do not matter
    if not dims:
        return 2
    step = dims[-1][1]
    if step < 2:
        return None
    expect = step * dims[-1][0]
    for size, stride in reversed(dims[:-1]):
        if stride != expect:
            return None
        expect = stride * size
    return step


def _grid_lookup_device(grid: torch.Tensor, x: torch.Tensor, cell_size: float, origin_xy) -> torch.Tensor:
    """The lookup as ONE launch of the library (mppi_grid_lookup: the reference's arithmetic, fp32 division included)
    instead of ~15 torch kernels: env.collision_check runs every tick of the examples' loops, and cost plugins on the
    generic path call it once per step.  Raises when the extension is missing (no silent fallback on a GPU box)."""
    global _native
    if _native is None:
        from mppi_playground_amd import _capi

        _native = (_capi, _capi.load())
    capi, lib = _native
    o = origin_xy  # host floats owned by the map object (never a cache keyed on device addresses: those get reused)
    layout = (x.shape, x.stride())
    step = _stride_of_layout.get(layout, 0)
    if step == 0:  # (a per-tick caller passes the same layout every time)
        if len(_stride_of_layout) >= 256:
            _stride_of_layout.clear()
        step = _stride_of_layout[layout] = _uniform_point_stride(x)
    if step is None:
        x = x.contiguous()
        step = 2
    st = torch._C._cuda_getCurrentRawStream(x.device.index)
    shape = x.shape[:-1]
    n = shape.numel()
    if n <= _POOLED_POINTS:  # (per-tick callers: env.collision_check of a predicted trajectory; see _pool.py)
        key = (shape, x.device.index)
        pool = _out_pools.get(key)
        if pool is None:
            if len(_out_pools) >= 64:
                _out_pools.clear()
            pool = _out_pools[key] = RowPool(shape, x.device, torch.float32)
        out = pool.take(st)
    else:
        out = torch.empty(shape, device=x.device, dtype=torch.float32)
    rc = lib.mppi_grid_lookup(grid.data_ptr(), grid.shape[0], grid.shape[1], float(cell_size), o[0], o[1], x.data_ptr(),
                              n, step, out.data_ptr(), st)
    if rc != 0:
        raise capi.MppiError(f"mppi_grid_lookup failed ({rc})")
    return out


def grid_lookup(grid: torch.Tensor, x: torch.Tensor, cell_size: float, origin: torch.Tensor, origin_xy=None) -> torch.Tensor:
    """round-half-even(x / cell + origin) gather with out-of-bound = 1 (shared by both map types).  `origin_xy`: the
    origin as two host floats (the maps keep it as `_cell_map_origin`); without it the tensor is read back."""
    if (x.is_cuda and grid.is_cuda and x.dtype == torch.float32 and grid.dtype == torch.float32 and grid.is_contiguous()
            and x.shape[-1] == 2 and x.device == grid.device and not x.requires_grad):
        if origin_xy is None:
            origin_xy = tuple(float(v) for v in origin.detach().cpu().tolist())
        return _grid_lookup_device(grid, x, cell_size, origin_xy)
    idx = torch.round(x / cell_size + origin).long()
    ix, iy = idx[..., 0], idx[..., 1]
    oob = (ix < 0) | (ix >= grid.shape[0]) | (iy < 0) | (iy >= grid.shape[1])
    val = grid[ix.clamp(0, grid.shape[0] - 1), iy.clamp(0, grid.shape[1] - 1)]
    return torch.where(oob, torch.ones_like(val), val)


def generate_random_obstacles(obstacle_map: ObstacleMap, random_x_range: Sequence[float],
                              random_y_range: Sequence[float], num_circle_obs: int,
                              radius_range: Sequence[float], num_rectangle_obs: int,
                              width_range: Sequence[float], height_range: Sequence[float],
                              max_iteration: int, seed: int) -> None:
    """Rejection-sample non-overlapping circles then rectangles from numpy's default_rng(seed), drawing
    (cx, cy, r) resp. (cx, cy, w, h) per trial in that order so the stream matches the reference."""
    rng = np.random.default_rng(seed)
    xr = (max(random_x_range[0], obstacle_map.x_lim[0]), min(random_x_range[1], obstacle_map.x_lim[1]))
    yr = (max(random_y_range[0], obstacle_map.y_lim[0]), min(random_y_range[1], obstacle_map.y_lim[1]))

    def dist(a, b):
        return np.linalg.norm(a - b)

    for _ in range(num_circle_obs):
        for trial in range(max_iteration + 1):
            if trial == max_iteration:
                raise RuntimeError("Cannot generate random obstacles due to reach max iteration.")
            center = np.array([rng.uniform(xr[0], xr[1]), rng.uniform(yr[0], yr[1])])
            radius = rng.uniform(radius_range[0], radius_range[1])
            hit = any(dist(c, center) <= r + radius for c, r in obstacle_map.circle_obs_list)
            hit = hit or any(dist(c, center) <= w / 2 + radius and dist(c, center) <= h / 2 + radius
                             for c, w, h in obstacle_map.rectangle_obs_list)
            if not hit:
                break
        obstacle_map.add_circle_obstacle(center, radius)

    for _ in range(num_rectangle_obs):
        for trial in range(max_iteration + 1):
            if trial == max_iteration:
                raise RuntimeError("Cannot generate random obstacles due to reach max iteration.")
            center = np.array([rng.uniform(xr[0], xr[1]), rng.uniform(yr[0], yr[1])])
            width = rng.uniform(width_range[0], width_range[1])
            height = rng.uniform(height_range[0], height_range[1])
            hit = any(dist(c, center) <= r + width / 2 and dist(c, center) <= r + height / 2
                      for c, r in obstacle_map.circle_obs_list)
            hit = hit or any(dist(c, center) <= w / 2 + width / 2 and dist(c, center) <= h / 2 + height / 2
                             for c, w, h in obstacle_map.rectangle_obs_list)
            if not hit:
                break
        obstacle_map.add_rectangle_obstacle(center, width, height)
