"""Sharding of num_samples across ranks (one process per GPU) — host-side logic only.

Trajectories are independent until the softmax, so rank r owns the contiguous block
[r*N/W, (r+1)*N/W) of the GLOBAL sample index space (the device noise is a function of the global
index, so results do not depend on W).  Per solve each rank produces one summary vector
    [min c, sum e, sum e^2, sum e*c, A[T*dc]]   with  e_i = exp((-c_i)/lambda - (-min c)/lambda)
over its shard; ONE all_gather of these 4+T*dc floats (408 B/rank for racing T=50) is the only
exchange, and every rank combines them locally (mppi_finalize on the device; combine_summaries()
below is the numpy statement of the same arithmetic, used by tests).
"""
from __future__ import annotations

import numpy as np
import torch


def shard_range(num_samples: int, world: int, rank: int):
    """(offset, count) of rank's contiguous block.  Any num_samples works, like in the reference (src/pi_mpc/mppi.py:
    96-98 puts no constraint on it): the first num_samples % world ranks own one sample more; a rank may own none only
    if num_samples < world, which is refused."""
    if num_samples < world:
        raise ValueError("num_samples must be at least the world size")
    q, rem = divmod(num_samples, world)
    return rank * q + min(rank, rem), q + (1 if rank < rem else 0)


def local_summary(costs: np.ndarray, actions: np.ndarray, lam: float) -> np.ndarray:
    """Summary of one shard from its costs [n] and clamped actions [n, T, dc] (numpy, float64 sums)."""
    lam32 = np.float32(lam)
    x = (-costs.astype(np.float32)) / lam32
    e = np.exp(x - x.max()).astype(np.float32)
    A = (e.astype(np.float64)[:, None] * actions.reshape(len(costs), -1).astype(np.float64)).sum(0)
    head = [costs.min(), e.astype(np.float64).sum(), (e.astype(np.float64) ** 2).sum(),
            (e.astype(np.float64) * costs).sum()]
    return np.concatenate([np.array(head), A]).astype(np.float32)


def combine_summaries(summaries: np.ndarray, lam: float):
    """[G, 4+row] -> (action[row], stats{cmin, sum_e, sum_e2, sum_ec}); mirrors finalize_kernel."""
    s = np.asarray(summaries, np.float64)
    lam32 = np.float32(lam)
    x = ((-s[:, 0].astype(np.float32)) / lam32).astype(np.float64)
    f = np.exp(x - x.max())
    se = float((f * s[:, 1]).sum())
    A = (f[:, None] * s[:, 4:]).sum(0)
    stats = dict(cmin=float(s[:, 0].min()), sum_e=se, sum_e2=float((f * f * s[:, 2]).sum()),
                 sum_ec=float((f * s[:, 3]).sum()))
    return (A / se).astype(np.float32), stats


def all_gather_summaries(summary: torch.Tensor, group=None, out: torch.Tensor = None) -> torch.Tensor:
    """One collective per solve: [len] on every rank -> [W, len] on every rank (RCCL on GPU, gloo on CPU).
    `out`: a reusable [W, len] buffer (the solver keeps one, so the per-solve path allocates nothing)."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty(world, summary.numel(), device=summary.device, dtype=summary.dtype)
    dist.all_gather_into_tensor(out.view(-1), summary.contiguous().view(-1), group=group)
    return out
