"""Shared bits of the example control loops (no rendering: UI is out of scope of this build)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import mppi_playground_amd  # noqa: E402,F401  (puts pi_mpc/ and envs/ on sys.path, like the reference's src/)


def run_loop(solver, step_fn, state, steps, label):
    """solve -> apply the first action -> repeat; prints the average solve time like the reference."""
    import torch

    total = 0.0
    for i in range(steps):
        torch.cuda.synchronize()
        t0 = time.time()
        action_seq, state_seq = solver.forward(state=state)
        torch.cuda.synchronize()
        total += time.time() - t0
        state = step_fn(state, action_seq[0])
    print(f"{label}: average solve time {total / steps * 1e3:.3f} ms over {steps} steps; final state {state.tolist()}")
    return state
