#!/usr/bin/env python3
"""A/B on one box: the ESSPS chain with round 0 as one launch (statistics pass + select step, block 0 gathers through
tagged cells) against round 0 as two kernels.  Round 1 is one conditional launch in both.  Also times a cold search per
solve (essps_cold: both rounds run) so the un-skipped round is measured too."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
import mppi_playground_amd  # noqa: F401

rows = {k: (label, make, x0) for k, label, _, _, make, x0 in bench._other_solvers(torch, np, ("c2_essps", "c5"))}
for key, (label, make, x0) in rows.items():
    for rep in range(2):
        for merge in (1, 0):
            s = make()
            s.set_option("essps_merge0", merge)
            dt = bench._time_solver(torch, s, x0, n=300, warm=50)
            # cold: every search starts from the geometric grid and takes both rounds
            import time
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(300):
                s.set_option("essps_cold", 1)
                s.forward(x0)
            torch.cuda.synchronize()
            cold = (time.perf_counter() - t0) / 300
            print(f"{key} merge0={merge}: warm {dt * 1e6:.1f} us  cold {cold * 1e6:.1f} us  lambda {s._last_lambda:.6f}", flush=True)
