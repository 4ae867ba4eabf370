#!/usr/bin/env python3
"""Same-box A/B of the weighted reduction's launch shape on the dense-weight configurations: reduction grid
(`reduce_blocks`: partial rows = blocks) x Philox chains per basic block (`reduce_chains`).  Prints us per solve (best of
three 50-solve loops) and the weights+reduce stage's device time (HIP events).  Usage (GPU box): python scripts/reduce_ab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import bench
import mppi_playground_amd  # noqa: F401

which = sys.argv[1:] or ["c3_dense", "c3_essps", "c5", "c2_essps", "c2"]
grids = [(512, 0), (512, 2), (512, 4), (1024, 2)]


def solver_of(key):
    if key == "c3_dense":
        ctrl, x0 = bench._racing_c3(torch, 5000.0)
        return ctrl, ctrl.solver, x0
    if key == "c3_essps":
        ctrl, x0 = bench._racing_c3(torch, "ESSPS", lambda_max=1.0e5)
        return ctrl, ctrl.solver, x0
    (k, label, nt, balg, make, x0), = bench._other_solvers(torch, np, which=(key,))
    s = make()
    return s, s, x0


for key in which:
    for blocks, chains in grids:
        keep, s, x0 = solver_of(key)
        s.set_option("reduce_blocks", blocks)
        s.set_option("reduce_chains", chains)
        t = bench._time_solver(torch, s, x0, n=50, warm=20)
        st = bench._stage_times(torch, s, x0, n=30)
        print(f"{key:9s} reduce_blocks {blocks:5d} chains {chains}: {t * 1e6:7.1f} us/solve | rollout {st['rollout_cost'] * 1e3:6.1f} "
              f"weights_reduce {st['weights_reduce'] * 1e3:6.1f} finalize {st['finalize'] * 1e3:5.1f} us", flush=True)
        del keep, s
        torch.cuda.empty_cache()
