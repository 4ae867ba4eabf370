#!/usr/bin/env python3
"""Soak of the ESSPS rounds that run as one launch (essps_round_kernel: block 0 gathers the other blocks' partial sums through
tagged cells): K closed-loop solves per size, a cold (two-round) search every third solve, next to a solver whose round 0 is
merged as well (option essps_merge0) — temperatures and actions must stay bit-equal, and every 500th temperature is checked
against brentq on a float64 evaluation of the same costs.  Usage: python scripts/essps_round_stress.py [K]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from scipy.optimize import brentq

import mppi_playground_amd  # noqa: F401
from envs.navigation_2d import Navigation2DEnv
from pi_mpc.mppi import MPPI

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
nav = Navigation2DEnv()
t = torch.tensor
for N in (4096 + 64, 65536, 262144 + 1000):
    a_s = MPPI(30, N, 3, 2, nav.dynamics, nav.cost_function, nav.u_min, nav.u_max, t([0.5, 0.5]), "ESSPS", seed=11)
    b_s = MPPI(30, N, 3, 2, nav.dynamics, nav.cost_function, nav.u_min, nav.u_max, t([0.5, 0.5]), "ESSPS", seed=11)
    for s in (a_s, b_s):
        s.set_option("fused_solve", 0)
    b_s.set_option("essps_merge0", 1)
    x = nav.reset().clone()
    two_pass = checked = 0
    worst = 0.0
    t0 = time.perf_counter()
    for k in range(K):
        if k % 3 == 0:
            a_s.set_option("essps_cold", 1)
            b_s.set_option("essps_cold", 1)
        a, st = a_s.forward(x)
        b, sb = b_s.forward(x)
        if k % 50 == 0 or k == K - 1:
            assert a_s._last_lambda == b_s._last_lambda, (N, k, a_s._last_lambda, b_s._last_lambda)
            assert torch.equal(a, b), (N, k)
            two_pass += a_s._h.lib.mppi_search_passes(a_s._h.h, None) == 2
        if k % 500 == 0:
            c = a_s._costs.cpu().numpy().astype(np.float64)
            ess = lambda l: (lambda e: e.sum() ** 2 / (e * e).sum())(np.exp(-(c - c.min()) / l))  # noqa: E731
            lo, hi = 0.01, 10.0
            if ess(lo) < 0.1 * N < ess(hi):
                want = brentq(lambda l: ess(l) - 0.1 * N, lo, hi, xtol=1e-13)
                worst = max(worst, abs(a_s._last_lambda - want) / want)
                checked += 1
        x = torch.as_tensor(sb)[0, 1].clone()
        if k % 400 == 399:
            x = nav.reset().clone()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert worst <= 1e-5, worst
    print(f"N = {N}: {K} closed-loop solves x 2 solvers in {dt:.1f} s, every 50th compared (bit-equal), {two_pass} of the compared "
          f"searches took two rounds, {checked} temperatures against brentq: worst relative error {worst:.1e}", flush=True)
print("ok")
