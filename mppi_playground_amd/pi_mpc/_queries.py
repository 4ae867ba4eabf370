"""What a caller can ask pi_mpc.mppi.MPPI after a solve (mix-in): softmax statistics, the reference's inspection attributes
(`_costs`, `_weights`, `_action_noises`, `_perturbed_action_seqs`, `_state_seq_batch`: src/pi_mpc/mppi.py:261-336,376),
get_top_samples (mppi.py:462-487), get_samples_from_posterior (mppi.py:489-506), last_stats."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Tuple

import numpy as np
import torch

from mppi_playground_amd import _capi
from mppi_playground_amd._pool import RowPool, capturing
from pi_mpc._lazy import _ptr
from pi_mpc.sharding import shard_range


class QueriesMixin:
    def _softmax_stats(self, lam: float) -> Dict[str, float]:
        """{cmin, cmax, se, se2, sec} of softmax(-costs/lam) over ALL samples, reduced on the device
        (one 40-byte read-back per probe; one all_gather of 5 doubles per probe when sharded)."""
        out = (C.c_double * 5)()
        self._h.call("mppi_softmax_stats", float(lam), out, self._stream())
        cmin, cmax, se, se2, sec = (float(v) for v in out)
        if self._world > 1:
            import torch.distributed as dist

            mine = torch.tensor([cmin, cmax, se, se2, sec], dtype=torch.float64, device=self._device)
            allv = torch.empty(self._world * 5, dtype=torch.float64, device=self._device)
            dist.all_gather_into_tensor(allv, mine, group=self._pg)
            a = allv.view(self._world, 5).cpu().numpy()
            lam32 = np.float32(lam)
            x = ((-a[:, 0].astype(np.float32)) / lam32).astype(np.float64)
            f = np.exp(x - x.max())
            cmin, cmax = float(a[:, 0].min()), float(a[:, 1].max())
            se, se2, sec = float((f * a[:, 2]).sum()), float((f * f * a[:, 3]).sum()), float((f * a[:, 4]).sum())
        return dict(cmin=cmin, cmax=cmax, se=se, se2=se2, sec=sec)

    def _ess_grid(self, lams) -> np.ndarray:
        """ESS(lambda) for up to 32 lambdas from ONE pass over the costs on the device
        (mppi_softmax_stats_multi; shards combined with one all_gather per call)."""
        lams = np.ascontiguousarray(lams, dtype=np.float32)
        L = len(lams)
        out = np.zeros((L, 3), np.float64)
        self._h.call("mppi_softmax_stats_multi", lams.ctypes.data_as(C.c_void_p), L, out.ctypes.data_as(C.c_void_p),
                     self._stream())
        se, se2 = out[:, 0], out[:, 1]
        if self._world > 1:
            import torch.distributed as dist

            cmin = self.last_local_cmin()
            mine = torch.from_numpy(np.concatenate([[cmin], out.ravel()])).to(self._device)
            allv = torch.empty(self._world * mine.numel(), dtype=torch.float64, device=self._device)
            dist.all_gather_into_tensor(allv, mine, group=self._pg)
            a = allv.view(self._world, -1).cpu().numpy()
            cm, st = a[:, 0], a[:, 1:].reshape(self._world, L, 3)
            # every shard's sums are relative to ITS minimum, e = exp((cmin_w - c) / lam) (stats_multi_partial_kernel):
            # rescale by exp((cmin - cmin_w) / lam), difference first, in float64 — rounding the two quotients
            # separately would put ulp(cmin / lam) into the exponent
            f = np.exp((cm.min() - cm)[:, None] * (1.0 / lams.astype(np.float64))[None, :])
            se, se2 = (f * st[:, :, 0]).sum(0), (f * f * st[:, :, 1]).sum(0)
        return se * se / se2

    def last_local_cmin(self) -> float:
        out = (C.c_double * 5)()
        self._h.call("mppi_softmax_stats", 1.0, out, self._stream())
        return float(out[0])

    def _gather_costs_host(self) -> np.ndarray:
        """costs[N] on the host for the temperature search (all shards when sharded)."""
        c = np.empty(self._local_samples, np.float32)
        self._h.call("mppi_get_costs", c.ctypes.data_as(C.c_void_p), 0, self._stream())
        if self._world > 1:
            import torch.distributed as dist

            counts = [shard_range(self._num_samples, self._world, r)[1] for r in range(self._world)]
            width = max(counts)  # shards may differ by one sample: gather equal-sized rows, then drop the padding
            row = np.full(width, np.nan, np.float32)
            row[:len(c)] = c
            out = torch.empty(self._world * width, device=self._device, dtype=torch.float32)
            dist.all_gather_into_tensor(out, torch.from_numpy(row).to(self._device), group=self._pg)
            rows = out.cpu().numpy().reshape(self._world, width)
            c = np.concatenate([rows[r, :counts[r]] for r in range(self._world)])
        return c

    # ------------------------------------------------------------------ lazily materialised state
    @property
    def _costs(self) -> torch.Tensor:
        c = torch.empty(self._local_samples, device=self._device, dtype=self._dtype)
        self._h.call("mppi_get_costs", _ptr(c), 1, self._stream())
        return c

    @property
    def _weights(self) -> torch.Tensor:
        """softmax(-costs/lambda) of the last solve (src/pi_mpc/mppi.py:376), this shard's slice."""
        w = torch.empty(self._local_samples, device=self._device, dtype=self._dtype)
        stats = self._stats.cpu().numpy()
        self._h.call("mppi_weights", float(self._last_lambda), float(stats[0]), float(stats[1]), _ptr(w),
                     self._stream())
        return w

    @property
    def _action_noises(self) -> torch.Tensor:
        e = torch.empty(self._local_samples, self._horizon, self._dim_control, device=self._device,
                        dtype=self._dtype)
        self._h.call("mppi_export_noise", _ptr(e), None, self._stream())
        return e

    @property
    def _perturbed_action_seqs(self) -> torch.Tensor:
        """clamp(mean + eps) of the last solve, [N,T,dc] (src/pi_mpc/mppi.py:266-275): kept by the generic path,
        rebuilt on demand from the solve's noise and the mean it sampled around for the native models."""
        if self._model is None:
            return self._perturbed_action_seqs_buf
        return self._perturbed_actions_for(self._mean_of_last_solve)

    @property
    def _state_seq_batch(self) -> torch.Tensor:
        """All N state trajectories of the last solve, [N,T+1,ds] (src/pi_mpc/mppi.py:280-286).  The native
        path never stores them (856 MB at N = 2^20, T = 50): they are re-rolled on demand."""
        if self._model is None:
            return self._state_seq_batch_buf
        n = self._local_samples
        out = torch.empty(n, self._horizon + 1, self._dim_state, device=self._device, dtype=self._dtype)
        idx = torch.arange(n, device=self._device, dtype=torch.int64)
        self._h.call("mppi_rollout_samples", _ptr(idx), n, _ptr(out), self._stream())
        return out

    def _perturbed_actions_for(self, mean: torch.Tensor) -> torch.Tensor:
        """clamp(mean + eps) for the resident noise with an explicit mean (the mean of the LAST solve
        has been overwritten by the warm start when store_mean was on)."""
        u = torch.empty(self._local_samples, self._horizon, self._dim_control, device=self._device,
                        dtype=self._dtype)
        cur = torch.empty(self._horizon, self._dim_control, device=self._device, dtype=self._dtype)
        st = self._stream()
        self._h.call("mppi_get_mean", _ptr(cur), 1, st)
        self._h.call("mppi_set_mean", _ptr(mean.contiguous()), 1, st)
        self._h.call("mppi_export_noise", None, _ptr(u), st)
        self._h.call("mppi_set_mean", _ptr(cur), 1, st)
        return u

    # ------------------------------------------------------------------ queries
    def get_top_samples(self, num_samples: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """Top-weighted trajectories of the last solve (src/pi_mpc/mppi.py:462-487).  The N state
        trajectories are not kept in HBM: the k winners are selected on the device (largest weight =
        smallest cost) and re-rolled from the noise of that solve around the mean it sampled."""
        assert num_samples <= self._num_samples
        if self._world > 1:
            return self._top_samples_sharded(num_samples)
        if self._model is None:  # the generic path keeps _state_seq_batch like the reference
            top = torch.topk(self._weights, num_samples)
            order = torch.argsort(top.values, descending=True)
            return self._state_seq_batch_buf[top.indices][order], top.values[order]
        st = self._stream()
        pools = self._top_pools.get(num_samples)
        if pools is None:  # (the examples ask for the same k every tick: see _pool.py)
            if len(self._top_pools) >= 16:
                self._top_pools.clear()
            pools = self._top_pools[num_samples] = (
                RowPool((num_samples, self._horizon + 1, self._dim_state), self._device, self._dtype),
                RowPool((num_samples,), self._device, self._dtype))
        cap = capturing()
        out, w = pools[0].take(st.value, cap), pools[1].take(st.value, cap)
        # one library call for any k: radix select + sort (one block up to 1024, multi-pass beyond) + re-roll + weights — ONE
        # launch up to 4096 samples; the weights use the temperature the solve left on the device (no read-back, no wait)
        self._h.call("mppi_top_samples", num_samples, _capi.LAMBDA_DEVICE, _ptr(out), _ptr(w), st)
        return out, w

    def _top_samples_sharded(self, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """Sharded get_top_samples: every rank selects its min(k, local) best candidates ((cost key << 32) | global
        index, padded to k with the largest word), one all_gather merges them, and — the device noise being a function
        of the global sample index — every rank re-rolls the k global winners itself: all ranks return the same
        tensors.  Opaque callables keep their state trajectories (like the reference): there the winners' rows are
        gathered instead of re-rolled."""
        import torch.distributed as dist

        st = self._stream()
        kk = min(k, self._local_samples)
        flip = torch.tensor(-(1 << 63), dtype=torch.int64, device=self._device)  # unsigned order through a signed sort
        if self._model is None:
            wl = self._weights  # this shard's slice of the global softmax
            top = torch.topk(wl, kk)
            mine_w = torch.full((k,), -1.0, device=self._device, dtype=self._dtype)
            mine_s = torch.zeros(k, self._horizon + 1, self._dim_state, device=self._device, dtype=self._dtype)
            mine_w[:kk], mine_s[:kk] = top.values, self._state_seq_batch_buf[top.indices]
            all_w = torch.empty(self._world * k, device=self._device, dtype=self._dtype)
            all_s = torch.empty(self._world * k, self._horizon + 1, self._dim_state, device=self._device, dtype=self._dtype)
            dist.all_gather_into_tensor(all_w, mine_w, group=self._pg)
            dist.all_gather_into_tensor(all_s, mine_s, group=self._pg)
            best = torch.sort(all_w, descending=True, stable=True)
            return all_s[best.indices[:k]], best.values[:k]
        mine = torch.full((k,), -1, dtype=torch.int64, device=self._device)  # uint64 bit patterns; -1 = the largest word
        if kk == k:
            self._h.call("mppi_top_candidates", kk, _ptr(mine), st)
        else:
            part = torch.empty(kk, dtype=torch.int64, device=self._device)
            self._h.call("mppi_top_candidates", kk, _ptr(part), st)
            mine[:kk] = part
        allc = torch.empty(self._world * k, dtype=torch.int64, device=self._device)
        dist.all_gather_into_tensor(allc, mine, group=self._pg)
        best = (torch.sort(allc ^ flip).values[:k] ^ flip).contiguous()
        out = torch.empty(k, self._horizon + 1, self._dim_state, device=self._device, dtype=self._dtype)
        w = torch.empty(k, device=self._device, dtype=self._dtype)
        self._h.call("mppi_rollout_candidates", _ptr(best), k, _capi.LAMBDA_DEVICE, _ptr(out), _ptr(w), st)
        return out, w

    def get_samples_from_posterior(self, optimal_solution: torch.Tensor, state: torch.Tensor,
                                   num_samples: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """N(optimal_solution, Sigma) samples (unclamped) and their rollouts (src/pi_mpc/mppi.py:489-506).

        The draw comes from the SOLVER'S noise stream, like the reference's MultivariateNormal.sample() on torch's
        global generator: in "torch_cpu" mode the next [k,T,dc] normals of the solver's CPU generator (so an
        identically seeded reference run sees the same samples and the same noise in the following solve), otherwise
        the Philox stream at the next solve index, which this call consumes.  The solver's own state is untouched:
        a later get_top_samples still describes the last solve."""
        assert num_samples <= self._num_samples
        k, T, dc = num_samples, self._horizon, self._dim_control
        st = self._stream()
        loc = torch.as_tensor(optimal_solution, dtype=self._dtype).to(self._device).contiguous()
        assert loc.shape == (T, dc)
        if self._noise_source == "torch_cpu":
            eps = torch.randn(k, T, dc, generator=self._cpu_gen, dtype=torch.float32) * self._sigmas.cpu()
            samples = (loc + eps.to(self._device)).contiguous()
        else:
            samples = torch.empty(k, T, dc, device=self._device, dtype=self._dtype)
            self._h.call("mppi_sample_posterior", self._solve_idx, _ptr(loc), k, _ptr(samples), st)
            self._solve_idx += 1
        x0 = torch.as_tensor(np.asarray(state) if not torch.is_tensor(state) else state).to(
            self._device, self._dtype).contiguous()
        assert x0.shape == (self._dim_state,)
        if self._model is None:
            return samples, self._states_prediction(x0, samples)
        out = torch.empty(k, T + 1, self._dim_state, device=self._device, dtype=self._dtype)
        self._h.call("mppi_rollout_actions", _ptr(samples), k, _ptr(x0), _ptr(out), st)
        self._posterior_keep = (x0, samples)  # alive until the enqueued kernels ran
        return samples, out

    # ------------------------------------------------------------------ diagnostics
    def last_stats(self) -> Dict[str, float]:
        """{min cost, sum e, sum e^2, sum e*c, ess, lambda} of the last solve (synchronises)."""
        s = self._stats.cpu().numpy().astype(np.float64)
        return dict(cmin=s[0], sum_e=s[1], sum_e2=s[2], sum_ec=s[3], ess=s[1] * s[1] / s[2],
                    lambda_=self._last_lambda)
