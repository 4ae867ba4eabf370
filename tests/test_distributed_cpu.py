"""The N>1 path on CPU: two gloo ranks each own half of num_samples, build their shard summary (from
oracle costs), exchange it with the product's one all_gather, and combine — the result must equal the
unsharded oracle solve.  (The device kernels' side of the same contract is covered by
tests/test_gpu_parity.py::test_shard_invariance_and_combine.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import CASES, ROOT, load, oracle_problem, orc, rel_err


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, q):
    import mppi_playground_amd  # noqa: F401
    from pi_mpc.sharding import all_gather_summaries, combine_summaries, local_summary, shard_range

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg, g = CASES[name], load(name)
        N, T = cfg["N"], cfg["T"]
        lam = float(g["lambda_0"])
        off, n = shard_range(N, world, rank)
        # every rank evaluates only its block of the global sample range (threshold is global)
        P = oracle_problem(cfg["model"], N, T, cfg.get("exploration", 0.0))
        if cfg["model"] == "racing":
            P.set_ref_path(g["ref_path_0"])
        full = P.rollout_cost(g["x0_0"], g["mean_in_0"], g["eps_0"], want_U=True)
        c, U = full["costs"][off:off + n], full["U"][off:off + n]
        summ = torch.from_numpy(local_summary(c, U, lam))
        gathered = all_gather_summaries(summ)
        assert gathered.shape == (world, 4 + T * P.dc)
        a, stats = combine_summaries(gathered.numpy(), lam)
        if rank == 0:
            w, st = orc.softmax_weights(full["costs"], lam)
            a_ref = P.weighted_actions(w, g["mean_in_0"], g["eps_0"])
            q.put((rel_err(a.reshape(T, -1), a_ref), abs(stats["cmin"] - st["cmin"]),
                   abs(stats["sum_e"] - st["sum_e"]) / st["sum_e"], rel_err(a.reshape(T, -1), g["action_seq_0"])))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["nav2d_T30_N256_fixed_explore", "racing_T50_N512_fixed", "pendulum_T15_N256_fixed"])
def test_two_rank_shard_combine(name):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    err_a, err_cmin, err_se, err_gold = q.get(timeout=10)
    assert err_a < 2e-6 and err_cmin == 0.0 and err_se < 1e-6
    assert err_gold < 5e-5  # against the reference fixture (softmax conditioning at lambda=1)


def test_shard_range():
    from pi_mpc.sharding import shard_range

    assert shard_range(8388608, 8, 3) == (3 * 1048576, 1048576)
    # any num_samples (the reference accepts any, src/pi_mpc/mppi.py:96-98): contiguous blocks that differ by at most one
    for n, w in ((10, 4), (1000, 7), (5, 5), (1 << 20, 3)):
        blocks = [shard_range(n, w, r) for r in range(w)]
        assert blocks[0][0] == 0 and sum(c for _, c in blocks) == n
        assert all(blocks[r + 1][0] == blocks[r][0] + blocks[r][1] for r in range(w - 1))
        assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1
    with pytest.raises(ValueError):
        shard_range(3, 4, 0)


def test_bench_transport_children_rendezvous_under_a_launcher(tmp_path):
    """bench.py's per-transport process isolation, end to end on the CPU: two ranks under torch.distributed.run, every rank's
    parent spawns one child per transport (bench.run_child_transport: shifted MASTER_PORT, the launcher's TORCHELASTIC_*
    variables removed — with them the children would wait for a store that does not exist), the children of a transport form
    their own gloo process group; transport "b" hangs on one rank and is killed at its budget on both, "c" runs after it.  Rank 0
    prints ONE merged line."""
    import json
    import subprocess
    import sys

    child = tmp_path / "child.py"
    child.write_text(
        "import json, os, sys, time\n"
        "import torch.distributed as dist\n"
        "mode = sys.argv[sys.argv.index('--exchange') + 1]\n"
        "dist.init_process_group('gloo')\n"
        "if mode == 'b' and dist.get_rank() == 1:\n"
        "    time.sleep(600)\n"                      # a rank that never arrives: the other one blocks in the barrier
        "dist.barrier()\n"
        "if dist.get_rank() == 0:\n"
        "    print(json.dumps({'value': {'a': 5.0, 'c': 7.0}.get(mode), 'ms_per_step': 1.0, 'config': {'exchange_used': mode}}), flush=True)\n"
        "dist.destroy_process_group()\n")
    driver = tmp_path / "driver.py"
    driver.write_text(
        "import argparse, os, sys\n"
        f"sys.path.insert(0, {str(ROOT)!r})\n"
        "import bench\n"
        "bench.own_stdout()\n"
        "args = argparse.Namespace(steps=20, warmup=5, samples=1 << 20, horizon=50, first_budget_s=12.0)\n"
        f"run = lambda mode, i, b: bench.run_child_transport(mode, i, b, script={str(child)!r})\n"
        "sys.exit(bench.orchestrate_transports(args, 2, int(os.environ['RANK']), ['a', 'b', 'c'], run_child=run))\n")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29733", str(driver)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["value"] == 7.0 and d["config"]["exchange_used"] == "c"
    t = {e["requested"]: e for e in d["transports"]}
    assert t["a"]["value"] == 5.0 and t["b"]["exit_code"] == 124 and "killed" in t["b"]["error"] and t["c"]["value"] == 7.0
