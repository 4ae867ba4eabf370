"""ctypes front-end of the CPU oracle (oracle/mppi_oracle.c).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

PENDULUM, CARTPOLE, MOUNTAINCAR, NAV2D, RACING, MJCARTPOLE, GOALZONE = range(7)
MODEL_IDS = {"pendulum": PENDULUM, "cartpole": CARTPOLE, "mountaincar": MOUNTAINCAR,
             "nav2d": NAV2D, "racing": RACING, "mjcartpole": MJCARTPOLE, "goalzone": GOALZONE}
MODEL_DIMS = {PENDULUM: (2, 1), CARTPOLE: (4, 1), MOUNTAINCAR: (2, 1), NAV2D: (3, 2), RACING: (4, 2),
              MJCARTPOLE: (4, 1), GOALZONE: (7, 2)}


class OracleMap(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("cell", C.c_float), ("ox", C.c_float),
                ("oy", C.c_float), ("cells", C.c_void_p)]


class OracleProblem(C.Structure):
    _fields_ = [("model", C.c_int32), ("N", C.c_int32), ("T", C.c_int32), ("ds", C.c_int32),
                ("dc", C.c_int32), ("threshold", C.c_int32), ("u_min", C.c_float * 4),
                ("u_max", C.c_float * 4), ("params", C.c_float * 32), ("maps", OracleMap * 2),
                ("ref_path", C.c_void_p)]


def build(force: bool = False) -> str:
    """Compile oracle/mppi_oracle.c with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "mppi_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "_build/liboracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_rollout_cost.restype = C.c_int
        _lib.oracle_softmax_weights.restype = C.c_int
        _lib.oracle_softmax_weights.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        _lib.oracle_weighted_actions.restype = C.c_int
        _lib.oracle_rollout_single.restype = C.c_int
        _lib.oracle_torch_randn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        _lib.oracle_torch_seed.argtypes = [C.c_void_p, C.c_uint64]
        _lib.oracle_philox_normal.argtypes = [C.c_uint64, C.c_uint32, C.c_int64, C.c_int64, C.c_int,
                                              C.c_int, C.c_void_p, C.c_void_p]
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# -- model parameter packs (values cited in mppi_oracle.c) -------------------------------------
def racing_params(u_min=(-2.0, -0.25), u_max=(2.0, 0.25), L=1.0, v_max=8.0, dt=0.1,
                  x_lim=(-40.0, 40.0), y_lim=(-40.0, 40.0), Qc=2.0, Ql=3.0, Qv=2.0, Qo=10000.0,
                  Qin=0.01, Qdin=0.5):
    return [u_min[0], u_max[0], u_min[1], u_max[1], L, v_max, dt, x_lim[0], x_lim[1], y_lim[0],
            y_lim[1], Qc, Ql, Qv, Qo, Qin, Qdin]


def nav2d_params(u_min=(0.0, -1.0), u_max=(2.0, 1.0), dt=0.1, x_lim=(-10.0, 10.0),
                 y_lim=(-10.0, 10.0), goal=(9.0, 9.0), Qo=10000.0):
    return [u_min[0], u_max[0], u_min[1], u_max[1], dt, x_lim[0], x_lim[1], y_lim[0], y_lim[1],
            goal[0], goal[1], Qo]


def goalzone_params(goal, center=(0.0, 0.0), radius=10.0, u_min=(-1.0, -1.0), u_max=(1.0, 1.0), dt=0.1,
                    penalty=1000.0):
    return [u_min[0], u_max[0], u_min[1], u_max[1], dt, goal[0], goal[1], center[0], center[1], radius, penalty]


class Problem:
    """Owns the numpy buffers an OracleProblem points at."""

    def __init__(self, model, N, T, u_min, u_max, exploration=0.0, params=(), maps=(), ref_path=None):
        model = MODEL_IDS[model] if isinstance(model, str) else model
        ds, dc = MODEL_DIMS[model]
        self.model, self.N, self.T, self.ds, self.dc = model, int(N), int(T), ds, dc
        p = OracleProblem()
        p.model, p.N, p.T, p.ds, p.dc = model, int(N), int(T), ds, dc
        p.threshold = int(N * (1 - exploration))  # mppi.py:266
        for k in range(dc):
            p.u_min[k] = float(u_min[k])
            p.u_max[k] = float(u_max[k])
        for i, v in enumerate(params):
            p.params[i] = float(v)
        self._keep = []
        for i, m in enumerate(maps):
            cells, cell, origin = m
            cells = np.ascontiguousarray(cells, dtype=np.uint8)
            self._keep.append(cells)
            p.maps[i].nx, p.maps[i].ny = cells.shape
            p.maps[i].cell = float(cell)
            p.maps[i].ox, p.maps[i].oy = float(origin[0]), float(origin[1])
            p.maps[i].cells = cells.ctypes.data
        self.p = p
        self.set_ref_path(ref_path)

    def set_ref_path(self, ref_path):
        if ref_path is None:
            self.p.ref_path = None
            return
        r = _f32(ref_path)
        assert r.shape == (self.T + 1, 4)
        self._ref = r
        self.p.ref_path = r.ctypes.data

    # -- forward() steps 1-3
    def rollout_cost(self, x0, mean, eps, want_U=False, want_S=False, want_stage=False, want_margin=False):
        N, T, ds, dc = self.N, self.T, self.ds, self.dc
        x0, mean, eps = _f32(x0), _f32(mean), _f32(eps)
        assert eps.shape == (N, T, dc) and mean.shape == (T, dc) and x0.shape == (ds,)
        assert self.model != "racing" or self.p.ref_path, "racing: set the reference window first (set_ref_path)"
        U = np.empty((N, T, dc), np.float32) if want_U else None
        S = np.empty((N, T + 1, ds), np.float32) if want_S else None
        stage = np.empty((N, T), np.float32) if want_stage else None
        margin = np.empty(N, np.float32) if want_margin else None
        costs = np.empty(N, np.float32)
        rc = lib().oracle_rollout_cost(C.byref(self.p), _ptr(x0), _ptr(mean), _ptr(eps), _ptr(U), _ptr(S),
                                       _ptr(stage), _ptr(costs), _ptr(margin))
        assert rc == 0, rc
        return dict(costs=costs, U=U, S=S, stage=stage, margin=margin)

    # -- step 6
    def weighted_actions(self, w, mean, eps):
        w, mean, eps = _f32(w), _f32(mean), _f32(eps)
        out = np.empty((self.T, self.dc), np.float32)
        rc = lib().oracle_weighted_actions(C.byref(self.p), _ptr(w), _ptr(mean), _ptr(eps), _ptr(out))
        assert rc == 0
        return out

    # -- step 8
    def rollout_single(self, x0, actions):
        x0, actions = _f32(x0), _f32(actions)
        out = np.empty((self.T + 1, self.ds), np.float32)
        rc = lib().oracle_rollout_single(C.byref(self.p), _ptr(x0), _ptr(actions), _ptr(out))
        assert rc == 0
        return out


def num_threads() -> int:
    return int(lib().oracle_num_threads())


def softmax_weights(costs, lam):
    """mppi.py:376; returns (w[N] f32, stats dict)."""
    costs = _f32(costs)
    w = np.empty_like(costs)
    st = np.zeros(4, np.float64)
    lib().oracle_softmax_weights(_ptr(costs), costs.shape[0], float(lam), _ptr(w), _ptr(st))
    return w, dict(cmin=st[0], sum_e=st[1], sum_e2=st[2], sum_ec=st[3], ess=st[1] * st[1] / st[2])


def angle_normalize(x):
    x = _f32(x)
    y = np.empty_like(x)
    lib().oracle_angle_normalize(_ptr(x), _ptr(y), C.c_int(x.size))
    return y


def occ(cells, cell, origin, pts):
    cells = np.ascontiguousarray(cells, dtype=np.uint8)
    m = OracleMap(cells.shape[0], cells.shape[1], float(cell), float(origin[0]), float(origin[1]),
                  cells.ctypes.data)
    pts = _f32(pts).reshape(-1, 2)
    out = np.empty(pts.shape[0], np.float32)
    lib().oracle_occ(C.byref(m), _ptr(pts), _ptr(out), C.c_int(pts.shape[0]))
    return out


class TorchCpuStream:
    """torch's CPU generator for tensor.normal_() restated (mt19937 + Box-Muller)."""

    def __init__(self, seed: int):
        self.state = np.zeros(625, np.uint32)
        lib().oracle_torch_seed(_ptr(self.state), seed)

    def randn(self, n: int) -> np.ndarray:
        out = np.empty(n, np.float32)
        rc = lib().oracle_torch_randn(_ptr(self.state), _ptr(out), n)
        assert rc == 0
        return out


def philox_normal(seed, solve_idx, first, n, T, dc, sigma):
    sigma = _f32(sigma)
    out = np.empty((n, T, dc), np.float32)
    lib().oracle_philox_normal(int(seed), int(solve_idx), int(first), int(n), int(T), int(dc), _ptr(sigma),
                               _ptr(out))
    return out


# ---- map construction (test oracle; literal restatement of the reference's host loops) -----------------------
def obstacle_map_literal(nx, ny, cell_size, circles=(), rects=()):
    """ObstacleMap.__init__ + add_circle_obstacle + add_rectangle_obstacle, cell by cell as the reference writes
    them (src/envs/obstacle_map_2d.py:83-95,103-158).  circles: [(center(2,), radius)], rects: [(center, w, h)]
    in metres.  Returns (uint8 [nx, ny], origin)."""
    from math import ceil

    grid = np.zeros((nx, ny), np.uint8)
    origin = np.array([nx / 2, ny / 2]).astype(int)
    for center, radius in circles:
        c = np.round(np.asarray(center, float) / cell_size + origin).astype(int)
        r = ceil(radius / cell_size)
        for i in range(-r, r + 1):
            for j in range(-r, r + 1):
                if i ** 2 + j ** 2 <= r ** 2:
                    grid[np.clip(c[0] + i, 0, nx - 1), np.clip(c[1] + j, 0, ny - 1)] = 1
    for center, w, h in rects:
        c = np.ceil(np.asarray(center, float) / cell_size + origin).astype(int)
        wo, ho = ceil(w / cell_size), ceil(h / cell_size)
        x0, x1 = c[0] - ceil(wo / 2), c[0] + ceil(wo / 2)
        y0, y1 = c[1] - ceil(ho / 2), c[1] + ceil(ho / 2)
        x0, x1 = np.clip(x0, 0, nx - 1), np.clip(x1, 0, nx - 1)
        y0, y1 = np.clip(y0, 0, ny - 1), np.clip(y1, 0, ny - 1)
        grid[x0:x1, y0:y1] = 1
    return grid, origin


def lane_map_literal(nx, ny, cell_size, lane, lane_width):
    """LaneMap.__init__ + populate_map (src/envs/lane_map_2d.py:48-82): centre-line cells -> Euclidean distance
    transform (scipy, as the reference) -> threshold.  Returns (uint8 [nx, ny], origin)."""
    from scipy.ndimage import distance_transform_edt

    grid = np.ones((nx, ny))
    origin = np.array([nx // 2, ny // 2])
    for x, y, _ in lane:
        cx = int(round(x / cell_size)) + origin[0]
        cy = int(round(y / cell_size)) + origin[1]
        if 0 <= cx < nx and 0 <= cy < ny:
            grid[cx, cy] = 0
    dist = distance_transform_edt(grid)
    return np.where(dist <= (lane_width / 2) / cell_size, 0, 1).astype(np.uint8), origin
