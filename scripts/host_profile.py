"""cProfile of the host side of forward() for a tiny problem (C1 pendulum, lambda = 1): where do the ~26 us go?"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
import mppi_playground_amd  # noqa
from envs import classic_control as cc
from pi_mpc.mppi import MPPI

t = torch.tensor
s = MPPI(50, 1000, 2, 1, cc.pendulum_dynamics, cc.pendulum_cost, t([-2.0]), t([2.0]), t([1.0]), 1.0)
x0 = t([np.pi, 0.0], dtype=torch.float32).cuda()
for _ in range(200):
    s.forward(x0)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    s.forward(x0)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
# raw library call cost without Python around it
import ctypes as C
h, st = s._h, s._stream()
a = torch.empty(50, 1, device="cuda"); so = torch.empty(1, 51, 2, device="cuda")
t0 = time.perf_counter()
for i in range(2000):
    h.lib.mppi_solve(h.h, C.c_void_p(x0.data_ptr()), 5 + i, 1.0, C.c_void_p(a.data_ptr()), C.c_void_p(so.data_ptr()), None, st)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"bare mppi_solve: {1e6 * (t1 - t0) / 2000:.1f} us per call (enqueue), {1e6 * (time.perf_counter() - t0) / 2000:.1f} us wall")
