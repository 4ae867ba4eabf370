import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import mppi_playground_amd  # noqa
from envs import classic_control as cc
from pi_mpc.mppi import MPPI
s = MPPI(64, 262144, 4, 1, cc.cartpole_dynamics, cc.cartpole_cost, torch.tensor([-3.0]), torch.tensor([3.0]),
         torch.tensor([1.0]), "ESSPS", use_sg_filter=True)
x0 = torch.tensor([0.01, 0.0, 0.02, 0.0], device="cuda")
for _ in range(10): s.forward(x0)
torch.cuda.synchronize()
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(50): s.forward(x0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 50
pr.disable()
print("C5 ms/solve", dt * 1e3)
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
