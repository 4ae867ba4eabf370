#!/usr/bin/env python3
"""Measured parity of the racing rollout at C3 size against the oracle, for whichever build MPPI_HIP_LIB selects:
    python tests/parity_probe.py [--solves K]
prints, per solve of a short closed loop, the number of samples whose cost differs beyond 1e-5 * max|c| (map cells
flipped by <= few-ulp differences upstream), how many of them the oracle marks as boundary samples, and the largest
relative cost error among the clear samples.  Test infrastructure (uses oracle/); not a pytest module."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import mppi_playground_amd  # noqa: E402,F401
from helpers import oracle_problem  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--solves", type=int, default=2)
    ap.add_argument("--samples", type=int, default=1 << 20)
    args = ap.parse_args()
    from envs.racing_controller import racing_controller
    from envs.racing_env import RacingEnv

    N, T = args.samples, 50
    env = RacingEnv()
    ctrl = racing_controller(env, horizon=T, num_samples=N, lambda_=1.0)
    ctrl.set_cost_map(env._obstacle_map, env._lane_map)
    ctrl.device_tick = False
    state = env.reset().clone()
    out = []
    for k in range(args.solves):
        mean = ctrl.solver._previous_action_seq.cpu().numpy().copy()
        a, s = ctrl.update(state, env.racing_center_path)
        c = ctrl.solver._costs.cpu().numpy()
        eps = ctrl.solver._action_noises.cpu().numpy()
        P = oracle_problem("racing", N, T, ref_path=ctrl._reference_path_np)
        r = P.rollout_cost(state.cpu().numpy(), mean, eps, want_margin=True)
        scale = float(np.abs(r["costs"]).max())
        diff = np.abs(c - r["costs"])
        bad = diff > 1e-5 * scale
        clear = r["margin"] > 1e-3
        out.append(dict(solve=k, flips=int(bad.sum()), flips_on_clear_samples=int((bad & clear).sum()),
                        allowed=3 + int(2e-4 * N), max_rel_err_clear=float(diff[clear].max() / scale),
                        median_rel_err=float(np.median(diff) / scale), argmin_same=bool(np.argmin(c) == np.argmin(r["costs"]))))
        state, _ = env.step(a[0])
        state = state.clone()
    print(json.dumps({"lib": os.environ.get("MPPI_HIP_LIB", "default"), "N": N, "T": T, "solves": out}))


if __name__ == "__main__":
    main()
