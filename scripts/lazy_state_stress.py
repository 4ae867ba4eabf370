#!/usr/bin/env python3
"""Soak of the lazily completed state sequence: a lazy solver and one that rolls out inside finalize_kernel run the same
closed loops (same seeds -> same noise); the lazy solver's state_seq is read at once, late (after up to 5 more solves, in
a random order), or dropped unread — with allocator churn in between, so that a completion that wrote into a freed and
reused tensor would corrupt something that is checked.  Every read must equal the eager solver's bits.
Usage (GPU box): python scripts/lazy_state_stress.py [solves]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch

import mppi_playground_amd  # noqa: F401
from envs import classic_control as cc
from envs.navigation_2d import Navigation2DEnv
from pi_mpc.mppi import MPPI

n_solves = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
nav = Navigation2DEnv()
t = torch.tensor
rng = random.Random(7)
total, t0 = 0, time.perf_counter()
cases = (
    ("nav2d N=32768 T=30 ESSPS", lambda **k: MPPI(30, 32768, 3, 2, nav.dynamics, nav.cost_function, nav.u_min, nav.u_max, t([0.5, 0.5]), "ESSPS", **k),
     lambda: nav.reset().clone()),
    ("nav2d N=65536 T=50 lambda=5", lambda **k: MPPI(50, 65536, 3, 2, nav.dynamics, nav.cost_function, nav.u_min, nav.u_max, t([0.5, 0.5]), 5.0, **k),
     lambda: nav.reset().clone()),
    ("cartpole N=20000 T=64 ESSPS + SG", lambda **k: MPPI(64, 20000, 4, 1, cc.cartpole_dynamics, cc.cartpole_cost, t([-3.0]), t([3.0]), t([1.0]), "ESSPS",
                                                          use_sg_filter=True, **k), lambda: t([0.01, 0.0, 0.02, 0.0]).cuda()),
)
for name, make, start in cases:
    lazy, eager = make(lazy_state_seq=True), make()
    assert lazy._lazy_state and not eager._lazy_state
    x = start()
    per = n_solves // len(cases)
    kept, reads, churn = [], 0, []
    for k in range(per):
        a, s = lazy.forward(x)
        b, sb = eager.forward(x)
        assert torch.equal(a, b), (name, k)
        mode = rng.random()
        if mode < 0.3:  # read at once
            assert torch.equal(s, sb), (name, k, "early")
            reads += 1
        elif mode < 0.6:  # read later
            kept.append((s, sb))
        # else: dropped unread
        if len(kept) > rng.randint(0, 5):
            rng.shuffle(kept)
            while kept:
                s1, sb1 = kept.pop()
                assert torch.equal(s1, sb1), (name, k, "late")
                reads += 1
        # allocator churn of the sizes in play: a completion into a freed tensor would land in one of these
        churn = [torch.full((1, lazy._horizon + 1, lazy._dim_state), float(k), device="cuda") for _ in range(rng.randint(0, 3))]
        for c in churn:
            assert float(c.min()) == float(k) == float(c.max())
        x = sb[0, 1].clone() if k % 40 else start()
    for s1, sb1 in kept:
        assert torch.equal(s1, sb1)
    torch.cuda.synchronize()
    for c in churn:
        assert float(c.min()) == float(per - 1)
    total += per
    print(f"{name}: {per} solves, {reads} state sequences read (early / late), all bit-equal to the eager solver", flush=True)
print(f"{total} solves in {time.perf_counter() - t0:.1f} s: ok")
