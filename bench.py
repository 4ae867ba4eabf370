#!/usr/bin/env python3
"""bench.py — MPPI hot-path throughput on MI355X (BASELINE.json metric).

Workload (config.workload): racing kinematic-bicycle, horizon T=50, num_samples N=1,048,576 per GPU,
lambda=1.0 (BASELINE configs[2] = C3; with --gpus G it is C4: G*N samples sharded over G ranks, weak
scaling, one all_gather of 4+T*dc floats per solve).  A "step" is one MPPI solve = one pass of the hot
path: sample -> rollout+cost -> weights+reduce -> finalize, with every input resident in HBM.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        --master-port 29500 bench.py --gpus 8 --steps 50 --warmup 10

Prints ONE JSON line on rank 0.  `value` = sample-steps/s (N_total * T * solves/s), whole job.
`roofline` is for the dominant kernel (rollout_cost_kernel), measured with HIP events on the launch
stream inside the timed region; `cpu_baseline` is the oracle (C restatement of the reference
algorithm, OpenMP over samples) timed on this host, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured copy


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--samples", type=int, default=1 << 20, help="samples per GPU")
    ap.add_argument("--horizon", type=int, default=50)
    ap.add_argument("--math", type=int, default=1, help="1 = fast-path math (default), 0 = library math")
    ap.add_argument("--noise-regen", type=int, default=1,
                    help="1 = regenerate the Philox noise in registers (default), 0 = materialise the noise tiles")
    ap.add_argument("--mapping", type=int, default=0, help="0 = lane per trajectory (default), 1 = the literal "
                    "wavefront-per-trajectory rollout (comparison only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timing", type=int, default=2, help="HIP-event instrumentation inside the timed region: "
                    "1 = every stage, 2 = dominant kernel only, 0 = none (stage times from a second pass)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import mppi_playground_amd  # noqa: F401
    from envs.racing_controller import racing_controller
    from envs.racing_env import RacingEnv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1:
        assert world == args.gpus, f"launch with torch.distributed.run --nproc-per-node {args.gpus}"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # MPPI_BENCH_BACKEND=gloo + MPPI_BENCH_ONE_DEVICE=1: dry run of the multi-rank path on a 1-GPU box
        backend = os.environ.get("MPPI_BENCH_BACKEND", "nccl")
        dev = 0 if os.environ.get("MPPI_BENCH_ONE_DEVICE") else local_rank
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)

    N_local, T = args.samples, args.horizon
    N_total = N_local * world
    env = RacingEnv()
    ctrl = racing_controller(env, horizon=T, num_samples=N_total, lambda_=1.0, shard_samples=world > 1)
    ctrl.set_cost_map(env._obstacle_map, env._lane_map)
    solver = ctrl.solver
    solver.set_option("math", args.math)
    solver.set_option("noise_regen", args.noise_regen)
    solver.set_option("mapping", args.mapping)
    state = env.reset()
    ref, _ = ctrl.calc_ref_trajectory(state, env.racing_center_path, 0, T, DL=0.1, lookahead_distance=3,
                                      reference_path_interval=0.85)
    ctrl.set_reference(ref)
    x0 = state.clone()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, before the contract's W warm-up steps: bring the device out of its idle power state (the first
    # ~20 ms of load run at lower clocks) so that short --warmup values do not time the clock ramp
    for _ in range(200):
        solver.forward(x0)
    sync()
    for _ in range(args.warmup):
        solver.forward(x0)
    sync()
    solver.set_option("timing", args.timing)
    solver.stage_times_ms()  # drain
    t0 = time.perf_counter()
    for _ in range(args.steps):
        a, s = solver.forward(x0)
    sync()
    dt = time.perf_counter() - t0
    stages = solver.stage_times_ms()
    if args.timing != 1:  # complete the per-stage picture with a separate instrumented pass
        solver.set_option("timing", 1)
        for _ in range(min(args.steps, 50)):
            solver.forward(x0)
        torch.cuda.synchronize()
        extra = solver.stage_times_ms()
        if args.timing == 2:
            extra["rollout_cost"] = stages["rollout_cost"]
        stages = extra
    solver.set_option("timing", 0)
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(a).all() and torch.isfinite(s).all()

    ms_per_step = dt / args.steps * 1e3
    solves_per_s = args.steps / dt
    value = N_total * T * solves_per_s

    if rank == 0:
        dc = 2
        # algorithmic bytes: per sample-step 4*dc B noise written by the sampler, read by the
        # rollout, read again by the weighted reduction, + 8 B/sample of costs -> per solve and per GPU:
        b_alg_solve = 3 * 4 * dc * N_local * T + 8 * N_local
        # dominant kernel = rollout_cost_kernel: reads the noise once, writes costs once
        b_alg_rollout = 4 * dc * N_local * T + 4 * N_local
        t_roll = stages["rollout_cost"] * 1e-3
        achieved = b_alg_rollout / t_roll / 1e9
        dev_solve_ms = sum(stages[k] for k in ("sample", "rollout_cost", "weights_reduce", "finalize"))
        # PMC-derived constants of the dominant kernel for THIS configuration (profiles/pmc_constants.json)
        traffic, valu = None, None
        try:
            if (N_local, T, args.math) == (1 << 20, 50, 1):
                pc = json.load(open(os.path.join(ROOT, "profiles", "pmc_constants.json")))
                k = pc["rollout_regen" if args.noise_regen else "rollout_tiles"]
                traffic = int((2 * k["fetch_kb"] + k["write_kb"]) * 1024)
                peak = 1024 * 2.4e9 / 2  # wave64 VALU instructions/s: 1024 SIMD32s, 2 cycles each, 2.4 GHz
                valu = {"kernel": "rollout_cost_kernel<racing>", "valu_insts_per_launch": k["valu_insts"],
                        "achieved_Ginst_per_s": k["valu_insts"] / t_roll / 1e9, "peak_Ginst_per_s": peak / 1e9,
                        "frac": k["valu_insts"] / t_roll / peak,
                        "measured_peak_Ginst_per_s": pc.get("valu_issue_ubench", {}).get("mul_add_Ginst_per_s"),
                        "cycles_per_inst_per_simd_at_2p4GHz": 1024 * 2.4e9 * t_roll / k["valu_insts"],
                        "note": "wave64 VALU instructions (SQ_INSTS_VALU, rocprofv3) / live kernel time; the kernel "
                                "is VALU-issue bound, not HBM bound (profiles/r01_experiments.md)"}
        except Exception:
            pass
        out = {
            "metric": "sample_steps_per_sec", "value": value, "unit": "sample-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "racing kinematic-bicycle MPPI solve (BASELINE configs[2]/[3])",
                       "num_samples_per_gpu": N_local, "num_samples_total": N_total, "horizon": T,
                       "lambda": 1.0, "noise": "device philox4x32-10 (" + ("regenerated in registers" if args.noise_regen else "materialised tiles") + ")", "math": "fast" if args.math else "library",
                       "mapping": "lane-per-trajectory" if not args.mapping else "wavefront-per-trajectory",
                       "sharding": f"num_samples x{world}" if world > 1 else "none",
                       "exchange": ("peer-to-peer buffers" if solver._p2p else "all_gather") if world > 1 else "none"},
            "solves_per_sec": solves_per_s,
            "roofline": {"bound": "hbm", "kernel": "rollout_cost_kernel<racing>", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "algorithmic_bytes_per_launch": b_alg_rollout,
                         "kernel_ms": stages["rollout_cost"]},
            "solve_roofline": {"algorithmic_bytes_per_solve": b_alg_solve, "device_ms_per_solve": dev_solve_ms,
                               "achieved_GBps": b_alg_solve / (dev_solve_ms * 1e-3) / 1e9,
                               "frac_of_8TBps": b_alg_solve / (dev_solve_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "stages_ms": {k: stages[k] for k in ("sample", "rollout_cost", "weights_reduce", "finalize")},
        }
        if valu is not None:
            out["valu_roofline"] = valu
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(np, T, ref.numpy(), x0.cpu().numpy())
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(np, T, ref, x0):
    """The oracle (oracle/mppi_oracle.c: C restatement of the reference algorithm, OpenMP over the
    samples) on this host: full racing solves (N=1,048,576, T=50: clamp, rollout, costs, softmax,
    weighted mean) repeated for ~10 s; reported in the metric's unit.  Noise generation is excluded
    (the oracle's Philox restatement is single-threaded test code)."""
    from helpers import oracle_problem, orc

    n = 1 << 20
    P = oracle_problem("racing", n, T, ref_path=ref)
    eps = orc.philox_normal(42, 1, 0, n, T, 2, [0.5, 0.1])
    mean = np.zeros((T, 2), np.float32)
    P.rollout_cost(x0, mean, eps)  # page in
    t0 = time.perf_counter()
    reps = 0
    while reps < 2 or time.perf_counter() - t0 < 10.0:
        r = P.rollout_cost(x0, mean, eps)
        w, _ = orc.softmax_weights(r["costs"], 1.0)
        P.weighted_actions(w, mean, eps)
        reps += 1
    dt = time.perf_counter() - t0
    return {"value": reps * n * T / dt, "unit": "sample-steps/s", "cores": orc.num_threads(), "kind": "port",
            "sample": f"{reps} full solves of racing N={n} T={T} in {dt:.1f} s (oracle C port, OpenMP over samples; "
                      "noise generation excluded)", "solves_per_sec": reps / dt}


if __name__ == "__main__":
    main()
