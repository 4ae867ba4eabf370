#!/usr/bin/env python3
"""Would a captured hipGraph of the native solve help the launch-bound configs?  TIMING EXPERIMENT ONLY: two consecutive
mppi_solve calls (the minimum-key slots alternate) are captured into one torch.cuda.CUDAGraph and replayed; the solve
index is baked into the captured kernel arguments, so every replay draws the same noise — fine for a clock, not a product."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import mppi_playground_amd  # noqa: F401
from bench import _other_solvers


def clock(fn, n):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for key, label, work, b_alg, make, x0 in _other_solvers(torch, np):
    s = make()
    for _ in range(5):
        s.forward(x0)
    eager = clock(lambda: s.forward(x0), 200)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    try:
        with torch.cuda.stream(side):
            s.forward(x0); s.forward(x0)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                s.forward(x0)
                s.forward(x0)
        torch.cuda.current_stream().wait_stream(side)
        graph = clock(g.replay, 100) / 2
        print(f"{label}: eager {eager:.1f} us/solve, captured graph {graph:.1f} us/solve", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"{label}: eager {eager:.1f} us/solve, capture failed: {type(e).__name__}: {str(e)[:200]}", flush=True)
    del s
