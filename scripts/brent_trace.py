#!/usr/bin/env python3
"""Where a probe of lbps_brent_kernel spends its time: block 0's per-phase 100 MHz clock sums from a -DMPPI_BRENT_TRACE build
(scripts/build_variant.sh brenttrace -DMPPI_BRENT_TRACE; MPPI_HIP_LIB=mppi_playground_amd/csrc/variants/lib_brenttrace.so).
Usage (GPU box): MPPI_HIP_LIB=... python scripts/brent_trace.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
import torch

import mppi_playground_amd  # noqa: F401
from envs import classic_control as cc
from pi_mpc.mppi import MPPI

PHASES = ["barrier A", "exp + thread sums", "wave reduction", "barrier B", "gather", "butterfly (fp64)",
          "objective + Brent step (fp64)", "tail"]
rng = np.random.default_rng(1)
for N in (4096, 65536, 1048576):
    s = MPPI(horizon=5, num_samples=N, dim_state=2, dim_control=1, dynamics=cc.pendulum_dynamics, cost_func=cc.pendulum_cost,
             u_min=torch.tensor([-2.0]), u_max=torch.tensor([2.0]), sigmas=torch.tensor([1.0]), lambda_=1.0)
    s.forward(torch.tensor([1.0, 0.0]))
    st = s._stream()
    c = torch.from_numpy((rng.uniform(10, 40, N) + 1e4 * rng.integers(0, 30, N) * (rng.random(N) < 0.5)).astype(np.float32)).cuda()
    s._h.call("mppi_set_costs", c.data_ptr(), 1, st)
    for _ in range(20):
        s._h.call("mppi_lbps_brent_device", 0.01, 0.01, 10.0, st)
    out = (C.c_int * 8)()
    s._h.lib.mppi_debug_brent_trace.argtypes = [C.c_void_p, C.c_void_p]
    assert s._h.lib.mppi_debug_brent_trace(s._h.h, out) == 0
    probes = s._h.lib.mppi_search_passes(s._h.h, st)
    print(f"N={N}: {probes} probes, {sum(out) / 100:.1f} us in block 0; per probe [us]: " +
          ", ".join(f"{n} {out[k] / 100 / probes:.2f}" for k, n in enumerate(PHASES)))
