#!/usr/bin/env python3
"""Randomised soak of the device-resident LBPS search (mppi_lbps_brent_device: scipy's bounded Brent as ONE kernel) against
the same search as a host loop over mppi_softmax_stats (mppi_lbps_lambda): synthetic cost vectors of the shipped models'
shapes and of awkward ones, random sample counts on every side of the kernel's geometry, random deltas and ranges — the two
temperatures must be IDENTICAL (float64 bit for bit).  Also times both on the nav2d-sized vector (N = 65 536).
Usage (GPU box): python scripts/brent_soak.py [cases]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import mppi_playground_amd  # noqa: F401
from envs import classic_control as cc
from pi_mpc.mppi import MPPI

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
rng = np.random.default_rng(20261001)


def draw(N, kind):
    if kind == 0:    # nav2d-like: distances + collision penalties
        c = rng.uniform(10, 40, N) + 1e4 * rng.integers(0, 30, N) * (rng.random(N) < 0.5)
    elif kind == 1:  # racing-like
        c = rng.uniform(300, 3000, N) + 1e4 * rng.integers(0, 25, N) * (rng.random(N) < 0.4)
    elif kind == 2:  # pendulum / cartpole-like: a smooth, narrow range
        c = rng.gamma(2.0, rng.uniform(0.5, 50.0), N) + rng.uniform(0, 100)
    elif kind == 3:  # a range of e^40
        c = np.exp(rng.uniform(-20, 20, N))
    elif kind == 4:  # mixed signs, any scale
        c = rng.standard_normal(N) * 10.0 ** rng.integers(-3, 6)
    elif kind == 5:  # all equal
        c = np.full(N, float(rng.uniform(-5, 5)))
    elif kind == 6:  # few distinct values
        c = rng.integers(0, max(2, N // 50), N).astype(np.float64)
    else:            # one clear winner
        c = rng.uniform(100, 200, N)
        c[int(rng.integers(0, N))] = 1.0
    return np.ascontiguousarray(c, dtype=np.float32)


def solver_for(N):
    s = MPPI(horizon=5, num_samples=N, dim_state=2, dim_control=1, dynamics=cc.pendulum_dynamics, cost_func=cc.pendulum_cost,
             u_min=torch.tensor([-2.0]), u_max=torch.tensor([2.0]), sigmas=torch.tensor([1.0]), lambda_=1.0)
    s.forward(torch.tensor([1.0, 0.0]))
    return s


def both(s, costs, delta, lo, hi):
    st = s._stream()
    c = torch.from_numpy(costs).cuda()
    s._h.call("mppi_set_costs", c.data_ptr(), 1, st)
    lh = C.c_double(0.0)
    s._h.call("mppi_lbps_lambda", delta, lo, hi, C.byref(lh), st)
    s._h.call("mppi_lbps_brent_device", delta, lo, hi, st)
    ld, used = C.c_double(0.0), C.c_double(0.0)
    s._h.call("mppi_get_lambda", C.byref(ld), C.byref(used), st)
    return lh.value, ld.value, s._h.lib.mppi_search_passes(s._h.h, st)


solvers, bad, probes_hist = {}, 0, []
by_kind = [0] * 8
t_start = time.time()
for case in range(cases):
    N = int(rng.choice([rng.integers(1, 70000), rng.integers(1, 2000), 65536, 65537, 256, 4096, 262144, 1048576, rng.integers(65536, 2_200_000)],
                       p=[0.3, 0.2, 0.15, 0.05, 0.05, 0.1, 0.05, 0.03, 0.07]))
    if N not in solvers:
        if len(solvers) > 16:
            solvers.clear()
            torch.cuda.empty_cache()
        solvers[N] = solver_for(N)
    kind = int(rng.integers(0, 8))
    costs = draw(N, kind)
    if rng.random() < 0.7:
        delta, lo, hi = 0.01, 0.01, 10.0
    else:
        delta = float(rng.choice([0.001, 0.01, 0.1, 0.5]))
        lo = float(10.0 ** rng.uniform(-3, 0.5))
        hi = lo * float(10.0 ** rng.uniform(0.02, 4))
    lh, ld, probes = both(solvers[N], costs, delta, lo, hi)
    by_kind[kind] += 1
    probes_hist.append(probes)
    if not (lh == ld):
        bad += 1
        print(f"MISMATCH case {case}: N={N} kind={kind} delta={delta} range=[{lo},{hi}] host={lh!r} device={ld!r} probes={probes}")
if cases:
  print(f"{cases} cases ({by_kind} per kind) in {time.time() - t_start:.0f} s: {bad} mismatches (device search vs host loop, float64 bits); "
        f"probes per search min/median/max = {min(probes_hist)}/{int(np.median(probes_hist))}/{max(probes_hist)}")
err = sum(s._h.lib.mppi_search_error(s._h.h) for s in solvers.values())
print("search_error flags raised:", err)

# timing at nav2d's size
for N in (4096, 65536, 262144, 1048576):
    s = solver_for(N)
    st = s._stream()
    c = torch.from_numpy(draw(N, 0)).cuda()
    s._h.call("mppi_set_costs", c.data_ptr(), 1, st)
    lh = C.c_double(0.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        s._h.call("mppi_lbps_lambda", 0.01, 0.01, 10.0, C.byref(lh), st)
    t_host = (time.perf_counter() - t0) / 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(10):
        s._h.call("mppi_lbps_brent_device", 0.01, 0.01, 10.0, st)
    e0.record()
    for _ in range(200):
        s._h.call("mppi_lbps_brent_device", 0.01, 0.01, 10.0, st)
    e1.record()
    torch.cuda.synchronize()
    probes = s._h.lib.mppi_search_passes(s._h.h, st)
    t_dev = e0.elapsed_time(e1) / 200 * 1e3
    print(f"N={N}: host loop {t_host * 1e6:.1f} us per search, device kernel {t_dev:.1f} us per search "
          f"({probes} probes, {t_dev / probes:.2f} us per probe)")
sys.exit(1 if bad or err else 0)
