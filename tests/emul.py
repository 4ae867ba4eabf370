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


_SEARCH_SO = os.path.join(_HERE, "libsearch.so")
_search = None


def search_lib():
    """Host build of csrc/host_search.hpp (the library's LBPS / ESSPS / MPO searches) over plain cost arrays."""
    global _search
    if _search is None:
        src = os.path.join(_HERE, "search.cpp")
        hdr = os.path.join(_HERE, "..", "..", "mppi_playground_amd", "csrc", "host_search.hpp")
        if (not os.path.exists(_SEARCH_SO)
                or os.path.getmtime(_SEARCH_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", _SEARCH_SO, src])
        _search = C.CDLL(_SEARCH_SO)
        d, vp, i = C.c_double, C.c_void_p, C.c_int
        _search.search_lbps.argtypes = [vp, i, d, d, d, vp, vp]
        _search.search_lbps_grid.argtypes = [vp, i, d, d, d, vp]
        _search.search_fminbound_poly.argtypes = [d, d, d, d, d, vp, vp]
        _search.search_essps.argtypes = [vp, i, d, d, d, d, vp, vp]
        _search.search_essps_first_grid.argtypes = [d, d, d, vp]
        _search.search_mpo.argtypes = [vp, i, i, d, d, d, vp]
    return _search


def lbps(costs, delta, lo, hi):
    c = np.ascontiguousarray(costs, np.float32)
    out, nf = C.c_double(0), C.c_int(0)
    assert search_lib().search_lbps(_p(c), len(c), delta, lo, hi, C.byref(out), C.byref(nf)) == 0
    return out.value, nf.value


def lbps_grid(costs, delta, lo, hi):
    c = np.ascontiguousarray(costs, np.float32)
    out = C.c_double(0)
    assert search_lib().search_lbps_grid(_p(c), len(c), delta, lo, hi, C.byref(out)) == 0
    return out.value


def fminbound_poly(a, b, c, lo, hi):
    out, nf = C.c_double(0), C.c_int(0)
    assert search_lib().search_fminbound_poly(a, b, c, lo, hi, C.byref(out), C.byref(nf)) == 0
    return out.value, nf.value


def essps(costs, target, lo, hi, lam_prev=0.0, with_passes=False):
    c = np.ascontiguousarray(costs, np.float32)
    out, passes = C.c_double(0), C.c_int(0)
    assert search_lib().search_essps(_p(c), len(c), target, lo, hi, lam_prev, C.byref(out), C.byref(passes)) == 0
    return (out.value, passes.value) if with_passes else out.value


def essps_first_grid(lam_prev, lo, hi):
    g = np.zeros(32, np.float64)
    assert search_lib().search_essps_first_grid(lam_prev, lo, hi, _p(g)) == 0
    return g


def mpo(cost_rows, lam0=1.0, epsilon=0.1, lr=0.2):
    c = np.ascontiguousarray(cost_rows, np.float32)
    out = np.zeros(c.shape[0], np.float64)
    assert search_lib().search_mpo(_p(c), c.shape[1], c.shape[0], lam0, epsilon, lr, _p(out)) == 0
    return out
