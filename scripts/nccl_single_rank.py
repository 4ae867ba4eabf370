#!/usr/bin/env python3
"""The real RCCL-backed exchanges on ONE GPU: a one-rank `nccl` process group (RCCL cannot put two ranks on one device)
and the solver's private `_force_exchange` hook, so that every solve takes the code path of an N-GPU run:
  nccl  summary -> all_gather_into_tensor (ProcessGroupNCCL: its own stream, two event hand-offs) -> combine
  rccl  the library's own communicator: ncclAllGather issued by mppi_weights_reduce on the solve's stream (one library
        call per solve, mppi_solve)
Reports the fixed cost of each path per solve against the unsharded solve, and checks that all agree.
Usage (GPU box): python scripts/nccl_single_rank.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29517")
import torch
import torch.distributed as dist

import mppi_playground_amd  # noqa: F401
from envs.racing_controller import racing_controller
from envs.racing_env import RacingEnv

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
env = RacingEnv()
out = {}
for forced in ("none", "nccl", "rccl"):
    os.environ["MPPI_EXCHANGE"] = forced if forced != "none" else "nccl"
    ctrl = racing_controller(env, horizon=50, num_samples=1 << 20, lambda_=1.0, shard_samples=True,
                             _force_exchange=forced != "none")
    ctrl.set_cost_map(env._obstacle_map, env._lane_map)
    s = ctrl.solver
    assert s._force_exchange == (forced != "none") and s._comm == (forced == "rccl")
    x0 = env.reset().clone()
    ref, _ = ctrl.calc_ref_trajectory(x0, env.racing_center_path, 0, 50, DL=0.1, lookahead_distance=3, reference_path_interval=0.85)
    ctrl.set_reference(ref)
    a, st = s.forward(x0)
    first = (a.clone(), st.clone())
    for _ in range(300):
        s.forward(x0)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        s.forward(x0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    out[forced] = (first, (t2 - t0) / 200 * 1e6, (t1 - t0) / 200 * 1e6)
    print(f"forced exchange = {forced}: {out[forced][1]:.1f} us/solve wall, host enqueue {out[forced][2]:.1f} us/solve", flush=True)
    del s, ctrl
a0, s0 = out["none"][0]
for mode in ("nccl", "rccl"):
    a1, s1 = out[mode][0]
    err = float((a0 - a1).abs().max() / a0.abs().max())
    print(f"RCCL all_gather path [{mode}] (1 rank) vs unsharded: max rel action difference {err:.2e}; "
          f"fixed cost of the exchange path {out[mode][1] - out['none'][1]:.1f} us/solve")
    assert err < 2e-6 and bool(torch.isfinite(s1).all())
dist.destroy_process_group()
