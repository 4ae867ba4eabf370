#!/usr/bin/env python3
"""Warm-started ESSPS on the device: passes over the costs per solve and the temperature's path, open loop and closed
loop, single launch and multi-kernel.  Usage (GPU box): python scripts/essps_passes.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import mppi_playground_amd  # noqa: F401
from bench import _other_solvers, _time_solver

for key, label, work, b_alg, make, x0 in _other_solvers(torch, np):
    if "ESSPS" not in label:
        continue
    for fused in (0, 2):
        s = make()
        s.set_option("fused_solve", fused)
        passes, lams = [], []
        x = x0.clone()
        for k in range(60):
            a, st = s.forward(x)
            passes.append(s._h.lib.mppi_search_passes(s._h.h, None))
            lams.append(s._last_lambda)
            if k >= 30:  # closed loop for the second half: the state moves along the predicted trajectory
                x = st[0, 1].clone()
        us = _time_solver(torch, s, x0, n=200, warm=30) * 1e6
        l = np.array(lams)
        print(f"{label} fused={fused}: {us:.1f} us/solve; passes open {passes[:30].count(1)}/30 one-pass, closed {passes[30:].count(1)}/30; "
              f"max |dlam/lam| open {np.max(np.abs(np.diff(l[:30]) / l[:29])):.3f} closed {np.max(np.abs(np.diff(l[30:]) / l[30:-1])):.3f}", flush=True)
        del s
