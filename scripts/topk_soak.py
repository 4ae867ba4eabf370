#!/usr/bin/env python3
"""Randomised soak of the one-launch get_top_samples (mppi_top_samples, N <= 4096, k <= 1024): synthetic cost vectors of
awkward shapes handed to the library (mppi_set_costs), the k winners against a host sort of the same costs —
weights in order, and (costs without ties) the re-rolled trajectories bit-equal to the index-driven re-roll of the
host's order.  Usage (GPU box): python scripts/topk_soak.py [cases]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
import torch

import mppi_playground_amd  # noqa: F401
from envs import classic_control as cc
from pi_mpc.mppi import MPPI

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = np.random.default_rng(20260927)


def draw(N):
    kind = rng.integers(0, 10)
    if kind == 0:    # one exponent, like the costs of a racing solve
        c = rng.uniform(77e3, 110e3, N)
    elif kind == 1:  # a huge range: almost everything in the first value bin
        c = np.exp(rng.uniform(-20, 20, N))
    elif kind == 2:  # a crowded boundary: many values within a few ulps
        c = 5.0 + rng.integers(0, 40, N) * np.float32(4.8e-7)
    elif kind == 3:  # plateaus (ties)
        c = rng.integers(0, max(2, N // 50), N).astype(np.float64)
    elif kind == 4:  # negative and positive
        c = rng.standard_normal(N) * 10.0 ** rng.integers(-3, 6)
    elif kind == 5:  # all equal
        c = np.full(N, float(rng.uniform(-5, 5)))
    elif kind == 6:  # a few infinite costs (collisions of a cost plugin)
        c = rng.uniform(0, 100, N)
        c[rng.random(N) < 0.2] = np.inf
    elif kind == 7:  # sorted / reversed input
        c = np.sort(rng.uniform(0, 1e4, N))[:: (1 if rng.random() < 0.5 else -1)]
    elif kind == 8:  # the k-th value duplicated around the boundary
        c = rng.uniform(0, 1, N)
        c[rng.integers(0, N, max(1, N // 8))] = c[rng.integers(0, N)]
    else:            # a running racing loop: a few hundred to a few thousand, plus collision penalties of 10^4 per step
        c = rng.uniform(300, 3000, N) + 1e4 * rng.integers(0, 25, N) * (rng.random(N) < 0.4)
    return np.ascontiguousarray(c, dtype=np.float32), int(kind)


solvers = {}
bad = 0
by_kind = [0] * 10
for case in range(cases):
    N = int(rng.choice([rng.integers(1, 4097), rng.integers(1000, 4097), 1024, 1025, 2048, 4096, 4000]))
    k = int(min(N, rng.choice([rng.integers(1, 1025), 1, 64, 300, 1000, 1024, N])))
    k = min(k, 1024)
    key = N
    if key not in solvers:
        if len(solvers) > 24:
            solvers.clear()
            torch.cuda.empty_cache()
        s = MPPI(horizon=10, num_samples=N, dim_state=2, dim_control=1, dynamics=cc.pendulum_dynamics, cost_func=cc.pendulum_cost,
                 u_min=torch.tensor([-2.0]), u_max=torch.tensor([2.0]), sigmas=torch.tensor([1.0]), lambda_=1.0)
        s.forward(torch.tensor([1.0, 0.0]))
        solvers[key] = s
    s = solvers[key]
    st = s._stream()
    costs, kind = draw(N)
    by_kind[kind] += 1
    lam = float(max(1e-3, np.ptp(costs[np.isfinite(costs)])) if np.isfinite(costs).any() else 1.0)
    c = torch.from_numpy(costs).cuda()
    s._h.call("mppi_set_costs", c.data_ptr(), 1, st)
    s._h.call("mppi_weights_reduce", lam, None, st)
    a = torch.empty(10, 1, device="cuda")
    s._h.call("mppi_finalize", None, 1, lam, 0, a.data_ptr(), None, None, st)
    out = torch.empty(k, 11, 2, device="cuda")
    w = torch.empty(k, device="cuda")
    s._h.call("mppi_top_samples", k, lam, out.data_ptr(), w.data_ptr(), st)
    order = np.lexsort((np.arange(N), costs))[:k]
    x = (-costs) / np.float32(lam)  # (fp32 quotients like the device's: at |cost / lambda| ~ 5000 their rounding is 5e-4 of a weight)
    ref = np.exp((x - x.max()).astype(np.float64))
    ref /= ref.sum()
    got = w.cpu().numpy()
    ok = got.shape == (k,) and np.all(np.isfinite(got)) and np.abs(got - ref[order]).max() <= 2e-5 * ref.max()
    tie_free = len(np.unique(costs[order])) == k and (k == N or costs[order][-1] < np.partition(costs, k)[k])
    if ok and tie_free:
        out2 = torch.empty_like(out)
        idx = torch.from_numpy(order.astype(np.int64)).cuda()
        s._h.call("mppi_rollout_samples", idx.data_ptr(), k, out2.data_ptr(), st)
        ok = bool(torch.equal(out, out2))
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: N={N} k={k} kind={kind} tie_free={tie_free}", flush=True)
print(f"{cases} cases (N in 1..4096, k in 1..1024; per kind of cost vector: {by_kind}): {bad} mismatches")
sys.exit(1 if bad else 0)
