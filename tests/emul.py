"""ctypes front-end of tests/host_emul (host build of the product's model functors; test-only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_emul")
_SO = os.path.join(_HERE, "libemul.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "emul.cpp")
        hdr = os.path.join(_HERE, "..", "..", "mppi_playground_amd", "csrc", "mppi_models.hpp")
        if (not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mfma",
                                   "-o", _SO, src])
        _lib = C.CDLL(_SO)
        _lib.emul_div_cell.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def rollout_cost(model_id, fast, x0, mean, eps, u_min, u_max, threshold, params=(), maps=(), geom=None,
                 ref=None, want_S=False, ds=None):
    f32 = np.float32
    N, T, dc = eps.shape
    x0, mean, eps = (np.ascontiguousarray(a, f32) for a in (x0, mean, eps))
    umin = np.zeros(4, f32); umax = np.zeros(4, f32)
    umin[:dc] = u_min; umax[:dc] = u_max
    params = np.asarray(list(params) + [0.0], f32)
    m0 = np.ascontiguousarray(maps[0], np.uint8) if len(maps) > 0 else None
    m1 = np.ascontiguousarray(maps[1], np.uint8) if len(maps) > 1 else None
    dims = np.array(m0.shape if m0 is not None else (0, 0), np.int32)
    g = np.asarray(geom if geom is not None else (1, 0, 0), f32)
    r = np.ascontiguousarray(ref, f32) if ref is not None else None
    costs = np.empty(N, f32); bad = np.empty(N, np.uint8)
    S = np.empty((N, T + 1, ds), f32) if want_S else None
    rc = lib().emul_rollout_cost(model_id, int(fast), N, T, int(threshold), _p(x0), _p(mean), _p(eps), _p(umin),
                                 _p(umax), _p(params), len(params) - 1, _p(m0), _p(m1), _p(dims), _p(g), _p(r),
                                 0 if r is None else r.shape[0], _p(costs), _p(bad), _p(S))
    assert rc == 0
    return costs, bad, S
