#!/usr/bin/env python3
"""Phase timeline of the single-launch solve (block 0, 100 MHz clock) from a -DMPPI_FUSED_TRACE build:
    scripts/build_variant.sh trace -DMPPI_FUSED_TRACE
    MPPI_HIP_LIB=mppi_playground_amd/csrc/variants/lib_trace.so python scripts/fused_trace.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import mppi_playground_amd  # noqa: F401
from bench import _other_solvers, _time_solver

NAMES = ["staged", "rollout", "min hop", "search r0", "search r1", "search r2", "temperature", "row published", "rows folded", "tail done"]
for key, label, work, b_alg, make, x0 in _other_solvers(torch, np, which=("c1", "c2", "c2_essps")):
    s = make()
    s.set_option("fused_solve", 2)
    us = _time_solver(torch, s, x0, n=100, warm=30) * 1e6
    acc = np.zeros(56)
    n = 20
    for _ in range(n):
        s.forward(x0)
        out = (C.c_int * 56)()
        assert s._h.lib.mppi_debug_fused_trace(s._h.h, out) == 0
        acc += np.array(list(out), np.float64)
    acc = acc / n / 100.0  # us
    prev = 0.0
    parts = []
    for k, name in enumerate(NAMES):
        if k < 10 and acc[k] >= prev and acc[k] > 0:
            parts.append(f"{name} +{acc[k] - prev:.1f}")
            prev = acc[k]
    f = acc - acc[2]
    if acc[10] > 0:
        parts.append(f"| inside search r0 (us from the min hop): costs staged {f[14]:.1f}, sums computed {f[15]:.1f}, block-synced {f[16]:.1f}, " +
                     (f"[first pass done {f[19]:.1f}] " if acc[19] > 0 else "") + f"published {f[10]:.1f}, gathered {f[11]:.1f}, combined {f[12]:.1f}, ESS per lane {f[17]:.1f}, lane-0 step {f[18]:.1f}, next grid + sync {f[13]:.1f}")
    if acc[10] > 0:
        wt = (acc[24:56].reshape(16, 2) - acc[2])[:8]  # (8 waves per block)
        parts.append("| per wave, statistics of round 0 (start -> done, us from the min hop): " + " ".join(f"{a:.1f}->{b:.1f}" for a, b in wt))
    print(f"{label}: {us:.1f} us/solve end to end; block 0 from its start: " + ", ".join(parts) + f" = {prev:.1f} us", flush=True)
