"""Plugin recognition: how a `dynamics` / `cost_func` callable tells the solver that it is one of the
shipped models and can run fused on the device.

The reference's plugin surface is two opaque Python callables (src/pi_mpc/mppi.py:30-31,55-56).  The
call site `MPPI(..., dynamics=env.dynamics, cost_func=ctrl.cost_function)` stays unchanged: a callable
(or the function object behind a bound method) carries an attribute

    __mppi_native__ = NativeTag(model, role, provider)

where `provider(owner)` returns the model's current inputs as plain host data:
    {"params": [floats], "maps": [GridSpec, ...], "ref_path": ndarray[rows,4] | None}
`owner` is the bound method's `__self__` (None for plain functions).  The solver calls the provider
of the cost callable before every solve (the racing reference window changes every tick,
example/racing.py:73-81) and re-uploads only what changed.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np


@dataclass(frozen=True)
class NativeTag:
    model: str                    # "pendulum" | "cartpole" | "mountaincar" | "nav2d" | "racing"
    role: str                     # "dynamics" | "cost"
    provider: Optional[Callable]  # provider(owner) -> dict


@dataclass
class GridSpec:
    cells: np.ndarray   # uint8 [nx][ny] occupancy (0/1)
    cell_size: float
    origin: tuple       # (ox, oy) cell index of world (0, 0)
    version: int = 0    # bump when cells change
    # optional integer description the solver can rasterise on the device instead of uploading `cells`:
    #   {"kind": "obstacles", "circles": int32[nc,3] (ci, cj, r cells), "rects": int32[nr,4] (x0, x1, y0, y1)}
    #   {"kind": "lane", "seeds": int32[ns,2], "max_d2": int}
    recipe: Optional[dict] = None


def native_model(model: str, role: str, provider: Optional[Callable] = None):
    """Decorator: tag a function / method as the native `role` of `model`."""

    def deco(fn):
        fn.__mppi_native__ = NativeTag(model, role, provider)
        return fn

    return deco


def resolve(fn):
    """-> (NativeTag, owner) or None.  Works for plain functions, bound methods, callable objects."""
    tag = getattr(fn, "__mppi_native__", None)
    if tag is None and hasattr(fn, "__func__"):
        tag = getattr(fn.__func__, "__mppi_native__", None)
    if not isinstance(tag, NativeTag):
        return None
    return tag, getattr(fn, "__self__", None)
