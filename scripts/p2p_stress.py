#!/usr/bin/env python3
"""Two ranks on one GPU, peer-to-peer exchange, many solves with random host-side skew between the ranks: the final
actions must be identical on both ranks and no poll may time out."""
import os, socket, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
import torch.multiprocessing as mp


def worker(rank, world, port, q, solves):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["MPPI_EXCHANGE"] = "p2p"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mppi_playground_amd  # noqa
    from envs.racing_controller import racing_controller
    from envs.racing_env import RacingEnv
    env = RacingEnv()
    ctrl = racing_controller(env, horizon=50, num_samples=1 << 16, lambda_=50.0, shard_samples=True)
    ctrl.set_cost_map(env._obstacle_map, env._lane_map)
    s = ctrl.solver
    assert s._p2p
    state = env.reset()
    rng = np.random.default_rng(rank)
    t0 = time.perf_counter()
    for i in range(solves):
        a, st = ctrl.update(state, env.racing_center_path)
        state, _ = env.step(a[0])  # closed loop: every rank must see exactly the same action
        if rng.random() < 0.02:
            time.sleep(rng.random() * 0.02)  # skew
    torch.cuda.synchronize()
    q.put((rank, a.cpu().numpy(), state.cpu().numpy(), s._h.lib.mppi_p2p_error(s._h.h), time.perf_counter() - t0))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    solves = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q, solves)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda r: r[0])
    [p.join(60) for p in procs]
    same = np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    print(f"{solves} closed-loop solves per rank: identical on both ranks = {same}, timeouts = {[r[3] for r in res]}, "
          f"{res[0][4] / solves * 1e3:.3f} ms per tick; final state {res[0][2].tolist()}")
    sys.exit(0 if same and not any(r[3] for r in res) else 1)
