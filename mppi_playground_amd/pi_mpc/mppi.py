"""pi_mpc.mppi.MPPI — host-side mirror of the reference solver (src/pi_mpc/mppi.py) whose
forward() hot path runs as hand-written HIP on MI355X through the C ABI in include/mppi_hip.h.

Same constructor, `forward(state, info={}) -> (action_seq[T,dc], state_seq[1,T+1,ds])`, `reset()`,
`get_top_samples(k)`, `get_samples_from_posterior(...)` and error behaviour as the reference; the
`dynamics` / `cost_func` callables are the plugin surface (pi_mpc/native.py explains how the shipped
models are recognised without changing the call site).

Device work per solve (native models): rollout+cost (the noise is regenerated in registers) ->
weights+reduce -> finalize (fold, normalise, Savitzky-Golay step if enabled, warm start, batch-1 rollout):
three kernel launches on torch's current stream with no host synchronisation when lambda is fixed.  The
auto-lambda searches (ESSPS/LBPS/MPO) are host code as in the reference, fed by softmax statistics reduced on
the device.  There is NO CPU fallback: without the built extension or without a GPU the constructor raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from mppi_playground_amd import _capi
from mppi_playground_amd._pool import RowPool, capturing
from pi_mpc import _host
from pi_mpc.native import resolve
from pi_mpc.sharding import all_gather_summaries, shard_range


from pi_mpc._exchange import ExchangeMixin
from pi_mpc._generic import GenericPathMixin
from pi_mpc._lazy import _DeferredStateSeq, _LazyInfoTensor, _ptr  # noqa: F401  (re-exported: tests, callers)
from pi_mpc._module import ModuleProtocolMixin
from pi_mpc._queries import QueriesMixin


_NO_INFO: Dict = {}  # forward()'s default `info` (the reference's shared mutable default, mppi.py:224): nobody can read it back


class MPPI(ModuleProtocolMixin, ExchangeMixin, GenericPathMixin, QueriesMixin, nn.Module):
    """Model Predictive Path Integral control (Williams et al., T-RO 2017) — MI355X-native."""

    # Private state ("_name") never holds Parameters, sub-modules or buffers, so it skips nn.Module's attribute
    # bookkeeping: forward() assigns about ten such attributes per solve and Module.__setattr__ costs ~1 us each — half of
    # the host time of a small solve (scripts/host_profile.py).  Properties with setters keep the normal route.
    _SETTER_PROPERTIES = frozenset(("_lambda", "_last_lambda", "_actions_history_for_sg"))

    def __setattr__(self, name, value):
        if name[0] == "_" and name[1:2] != "_" and name not in MPPI._SETTER_PROPERTIES:
            self.__dict__[name] = value
        else:
            super().__setattr__(name, value)

    def __init__(
        self,
        horizon: int,
        num_samples: int,
        dim_state: int,
        dim_control: int,
        dynamics: Callable[[torch.Tensor, torch.Tensor], torch.Tensor],
        cost_func: Callable[[torch.Tensor, torch.Tensor, Dict], torch.Tensor],
        u_min: torch.Tensor,
        u_max: torch.Tensor,
        sigmas: torch.Tensor,
        lambda_: float | str,
        lbps_delta: float = 0.01,
        essps_target_ess: Optional[float] = None,
        lambda_min: float = 0.01,
        lambda_max: float = 10.0,
        exploration: float = 0.0,
        use_sg_filter: bool = False,
        sg_window_size: int = 5,
        sg_poly_order: int = 3,
        device=torch.device("cuda"),
        dtype=torch.float32,
        seed: int = 42,
        *,
        noise_source: str = "philox",
        shard_samples: bool = False,
        process_group=None,
        auto_lambda_stats: str = "device",
        essps_search: str = "device",
        lbps_search: str = "brent",
        recognize_closures: bool = True,
        sg_filter: str = "device",
        graph_callables: bool = False,
        lazy_state_seq: Optional[bool] = None,
        _force_exchange: bool = False,
    ) -> None:
        """Arguments up to `seed` are the reference's (src/pi_mpc/mppi.py:24-47).

        Extensions (keyword-only):
            noise_source: "philox" (device Philox4x32-10 stream, default) or "torch_cpu" (draw with
                torch's CPU generator exactly like the reference does on CPU — same seed, same
                numbers — and upload; for parity runs).
            auto_lambda_stats: "device" (default) evaluates the softmax sums of the ESSPS/LBPS/MPO searches
                on the GPU (mppi_softmax_stats; the root-finders stay on the host), "host" copies
                costs[N] to the CPU and evaluates them in numpy like the reference does.
            essps_search: with device statistics, "device" (default) and "grid" bracket the ESSPS root with
                32-temperature grids (one pass over the costs each: two geometric ones, or ONE clustered around the
                previous solve's root when the temperature has moved < 1.5x) and an inverse polynomial
                interpolation — "device" runs the scalar steps of that search as kernels too, so the solve never
                waits for the host (the temperature stays in HBM; reading `_lambda` fetches it), "grid" reads the
                statistics back after each pass; "brentq" probes one lambda at a time like the reference's scipy
                call.  All return the same root (to ~1e-6 relative).  Sharded solvers combine the shards'
                statistics on the host ("device" behaves like "grid" there).
            lbps_search: with device statistics on one GPU.  "brent" (default) is the reference's own algorithm
                (mppi.py:341-349: scipy's bounded Brent, ported step for step in csrc/host_search.hpp) run ON THE DEVICE
                (round 6, mppi_lbps_brent_device: one launch for all ~22-31 dependent probes, the temperature stays in HBM,
                no host wait, capturable) — the rule that lands where the reference lands: within its own measured spread on
                every LBPS fixture.  "brent_host" is the same search as a host loop inside the library (mppi_lbps_lambda:
                one read-back of the device's statistics per probe, ~0.5 ms per solve; round 5's default) — the two return
                the same temperature TO THE BIT (same partial sums, same fp64 steps), which is what the GPU tests hold the
                device search to.  "grid" (round 3-5: "device", still accepted) is the two-grid search + quartic
                (mppi_lbps_lambda_device): on exact statistics the float64 minimiser to 3e-7, but the REFERENCE's Brent
                stops 6e-5..5e-3 away from that (xatol = 1e-5 absolute on an objective that is flat to fp32 noise for
                nav2d), so "grid" agrees with the reference to that spread only; kept for comparison.
                The MPO dual always steps on the device when the statistics are the device's own.
            recognize_closures: True (default): untagged callables that ARE the closures of the reference's classic-control
                examples (example/pendulum.py, cartpole.py, mountaincar.py, mujoco_cartpole.py: same source fingerprint AND
                the same values as the shipped plugin on probe batches, pi_mpc/recognize.py) run as the fused native model
                instead of on the generic path; False: untagged callables always take the generic path.
            sg_filter: "device" (default) runs the Savitzky-Golay step inside the finalize kernel
                (bit-identical to the host statement), "host" keeps the reference's numpy-style round trip.
            graph_callables: opaque (untagged) callables only.  The reference's two T-step Python loops over the user's
                `dynamics` / `cost_func` are ~2*T*(10..40) tiny kernels bound by launch latency; with True they are
                captured ONCE into a hipGraph (torch.cuda.CUDAGraph on static [N,T,*] buffers, after one eager solve as
                warm-up) and replayed every solve.  Requires capturable callables: no host synchronisation, no
                data-dependent shapes (boolean-mask assignment is not capturable), and an `info` dict used as the
                reference documents it (tensor views + the integer `t`).  If the capture fails the solver says so once
                and stays on the eager loops.
            lazy_state_seq: native models on the multi-kernel path; OPT-IN (default None / False: forward() returns a
                completed plain tensor, like the reference).  True: the batch-1 rollout of the solution (`state_seq`,
                mppi.py:448-449: T dependent steps of one wave, 5.8 us of a 146 us racing solve) leaves the solve's last
                kernel; it rides in one extra block of the NEXT solve's rollout launch, or is launched on the spot when
                `state_seq` is used first through torch (_DeferredStateSeq) — same code, same bits.  Until then the
                returned buffer holds NaN (a reader that bypasses torch — a raw data_ptr(), another library — must call
                `join_state_seq()` first; a completion on another stream is ordered behind the solve's stream).  For
                control loops that do not look at `state_seq` every tick (bench.py passes True and says so in its line).
            shard_samples: treat `num_samples` as the GLOBAL sample count and let this rank own the
                contiguous block rank*N/W .. (rank+1)*N/W of it (torch.distributed must be
                initialised); the 4+T*dc-float shard summaries are exchanged once per solve with one RCCL
                all_gather (default), or through the library's peer-to-peer buffers (environment variable
                MPPI_EXCHANGE = nccl | p2p | auto; "auto" takes the buffers when their start-up self-test passes
                on every rank and falls back to the all_gather otherwise).

        `device`: the hot path exists on MI355X only.  The reference falls back to the CPU silently when CUDA is
        unavailable or another device is asked for (src/pi_mpc/mppi.py:102-105); this class raises instead — a
        CPU device is refused; "cuda" / "cuda:i" must name the process's current device (one process per GPU).
        """
        super().__init__()
        # (what copy.deepcopy constructs the copy from: pi_mpc/_module.py)
        self._ctor = dict(horizon=horizon, num_samples=num_samples, dim_state=dim_state, dim_control=dim_control, dynamics=dynamics,
                          cost_func=cost_func, u_min=u_min, u_max=u_max, sigmas=sigmas, lambda_=lambda_, lbps_delta=lbps_delta,
                          essps_target_ess=essps_target_ess, lambda_min=lambda_min, lambda_max=lambda_max, exploration=exploration,
                          use_sg_filter=use_sg_filter, sg_window_size=sg_window_size, sg_poly_order=sg_poly_order, device=device,
                          dtype=dtype, seed=seed, noise_source=noise_source, shard_samples=shard_samples,
                          process_group=process_group, auto_lambda_stats=auto_lambda_stats, essps_search=essps_search,
                          lbps_search=lbps_search, recognize_closures=recognize_closures, sg_filter=sg_filter,
                          graph_callables=graph_callables, lazy_state_seq=lazy_state_seq, _force_exchange=_force_exchange)
        assert u_min.shape == (dim_control,)
        assert u_max.shape == (dim_control,)
        assert sigmas.shape == (dim_control,)
        if dtype != torch.float32:
            raise ValueError("the HIP path computes in float32 (the reference default dtype)")
        dev = torch.device(device)
        if dev.type != "cuda":
            raise ValueError(f"device={dev}: this MPPI runs its hot path on MI355X only; there is no CPU path "
                             "(the reference's silent CPU fallback, src/pi_mpc/mppi.py:102-105, is not reproduced)")
        if not torch.cuda.is_available():
            raise _capi.MppiError("no GPU visible: this MPPI runs its hot path on MI355X only "
                                  "(no CPU fallback)")
        _capi.load()  # fail loudly if the extension is missing
        if dev.index is not None and dev.index != torch.cuda.current_device():
            # one process drives one GPU: the library launches on the calling thread's current device
            raise ValueError(f"device={dev} is not the current device (cuda:{torch.cuda.current_device()}): select it "
                             "with torch.cuda.set_device() before constructing the solver")
        self._device = torch.device("cuda", torch.cuda.current_device())
        self._device_index = self._device.index
        self._dtype = dtype

        self._horizon = horizon
        self._num_samples = num_samples
        self._dim_state = dim_state
        self._dim_control = dim_control
        self._dynamics = dynamics
        self._cost_func = cost_func
        self._u_min = u_min.clone().detach().to(self._device, self._dtype)
        self._u_max = u_max.clone().detach().to(self._device, self._dtype)
        self._sigmas = sigmas.clone().detach().to(self._device, self._dtype)
        self._exploration = exploration
        self._use_sg_filter = use_sg_filter
        self._sg_window_size = sg_window_size
        self._sg_poly_order = sg_poly_order
        self._seed = int(seed)
        if auto_lambda_stats not in ("device", "host"):
            raise ValueError("auto_lambda_stats must be 'device' or 'host'")
        self._auto_lambda_stats = auto_lambda_stats
        self._graph_callables = bool(graph_callables)
        self._graph = None           # captured loops (generic path)
        self._graph_b1 = None        # captured batch-1 rollout of the solution (generic path); False = not capturable
        self._graph_info_keys = None
        self._graph_state = "off" if not graph_callables else "warmup"  # warmup -> capture -> replay | failed
        if essps_search not in ("device", "grid", "brentq"):
            raise ValueError("essps_search must be 'device', 'grid' or 'brentq'")
        self._essps_search = essps_search
        if lbps_search == "device":  # rounds 3-5 name of the grid search
            lbps_search = "grid"
        if lbps_search not in ("brent", "brent_host", "grid"):
            raise ValueError("lbps_search must be 'brent', 'brent_host' or 'grid'")
        self._lbps_search = lbps_search
        assert sg_filter in ("device", "host")
        self._sg_on_device = sg_filter == "device"
        if noise_source not in ("philox", "torch_cpu"):
            raise ValueError("noise_source must be 'philox' or 'torch_cpu'")
        self._noise_source = noise_source

        # ---- sharding of num_samples
        self._pg = process_group
        self._world, self._rank = 1, 0
        if shard_samples:
            import torch.distributed as dist

            if not dist.is_initialized():
                raise RuntimeError("shard_samples=True needs torch.distributed to be initialised")
            self._world = dist.get_world_size(process_group)
            self._rank = dist.get_rank(process_group)
        self._sample_offset, self._local_samples = shard_range(num_samples, self._world, self._rank)
        # measurement / test hook (private keyword, not an ambient switch): run the per-solve exchange (summary ->
        # all_gather -> combine) even with ONE rank, so that the real RCCL-backed paths can be exercised and their fixed
        # cost measured on a single GPU
        self._force_exchange = bool(_force_exchange and shard_samples and self._world == 1)

        # ---- plugin recognition: a native tag (pi_mpc/native.py), or the reference examples' own closures (pi_mpc/recognize.py:
        # fingerprint of their source AND agreement with the shipped plugin on probe batches)
        dyn, cst = resolve(dynamics), resolve(cost_func)
        self._recognized = None
        self._recognition = None
        if dyn is None and cst is None and recognize_closures:
            from pi_mpc import recognize

            twin = recognize.match(dynamics, cost_func, dim_state, dim_control, self._device)
            self._recognition = dict(recognize.last_report)  # what was decided and why (inspection; a disagreement warns once)
            if twin is not None:
                self._recognized = (dynamics, cost_func)  # (kept for inspection; the fused model runs instead)
                dyn, cst = resolve(twin[0]), resolve(twin[1])
        if (dyn and cst and dyn[0].model == cst[0].model and dyn[0].role == "dynamics"
                and cst[0].role == "cost"):
            self._model = dyn[0].model  # fused on the device
            self._cost_tag, self._cost_owner = cst
            self._dyn_tag, self._dyn_owner = dyn
            ds, dc = _capi.MODEL_DIMS[self._model]
            if (ds, dc) != (dim_state, dim_control):
                raise AssertionError(f"model {self._model} has dim_state={ds}, dim_control={dc}")
        else:
            # opaque callables (any torch dynamics / cost, e.g. the reference examples' TorchScript
            # closures): sampling, softmax, reduction and warm start run in the library, the two
            # T-step loops run the user's callables on GPU tensors exactly like the reference
            self._model = None
            self._cost_tag = self._cost_owner = self._dyn_tag = self._dyn_owner = None
            if not 1 <= dim_control <= _capi.MAX_DIM_CONTROL_GENERIC:
                raise ValueError(f"dim_control must be in 1..{_capi.MAX_DIM_CONTROL_GENERIC}")

        # ---- auto lambda (src/pi_mpc/mppi.py:183-210)
        self._lambda_pending = False  # the temperature of the last solve still lives on the device only
        self._lambda_stream = None    # ... and this is the stream that solve was enqueued on
        self._lambda_override = None  # MPO on the device: a temperature assigned to `_lambda` by the caller
        self._essps_prev = None       # root of the last host-side ESSPS grid search (warm start of the next one)
        self._used_known = True       # `_last_lambda_value` already holds the last solve's temperature
        self._lambda_value = self._last_lambda_value = None
        self._h = None
        self._lambda = lambda_
        self._lbps_delta = lbps_delta
        self._essps_target_ess = essps_target_ess if essps_target_ess is not None else num_samples / 10
        self._lambda_min = lambda_min
        self._lambda_max = lambda_max
        if self._lambda == "MPO":
            self._auto_lambda = "MPO"
            self._lambda = 1.0
            self._mpo = _host.MpoTemperature(1.0, 0.1, 0.2)  # host statement (sharded / host statistics)
        elif self._lambda == "LBPS":
            self._auto_lambda = "LBPS"
        elif self._lambda == "ESSPS":
            self._auto_lambda = "ESSPS"
        elif isinstance(self._lambda, float):
            self._auto_lambda = None
        else:
            raise ValueError("lambda_ must be 'MPO', 'LBPS', 'ESSPS', or a float value.")

        # ---- Savitzky-Golay (src/pi_mpc/mppi.py:160-165); raises ValueError like the reference
        self._coeffs = _host.savitzky_golay_coeffs(sg_window_size, sg_poly_order)
        self._sg_history_host = np.zeros((horizon - 1, dim_control), np.float32)
        # the device filter holds one thread per element of the action sequence and needs T >= 2
        self._sg_on_device = (self._sg_on_device and use_sg_filter and horizon >= 2
                              and horizon * dim_control <= 1024)

        # ---- device handle
        cfg = _capi.MppiConfig()
        cfg.model = _capi.MODEL_IDS[self._model] if self._model is not None else _capi.MODEL_GENERIC
        cfg.horizon, cfg.dim_state, cfg.dim_control = horizon, dim_state, dim_control
        cfg.num_samples = self._local_samples
        cfg.sample_offset = self._sample_offset
        cfg.inherit_count = int(num_samples * (1 - exploration))  # src/pi_mpc/mppi.py:266
        for k in range(min(dim_control, _capi.MAX_DIM_CONTROL)):
            cfg.u_min[k] = float(u_min[k])
            cfg.u_max[k] = float(u_max[k])
            cfg.sigmas[k] = float(sigmas[k])
        cfg.seed = self._seed
        cfg.device = self._device.index
        self._h = _capi.Handle(cfg)
        if dim_control > _capi.MAX_DIM_CONTROL:  # the config holds four controls: hand over the full vectors
            f = lambda t: (C.c_float * dim_control)(*[float(v) for v in t])  # noqa: E731
            self._h.call("mppi_set_control_limits", f(u_min), f(u_max), f(sigmas), dim_control)
        # the LBPS search and the MPO step run inside the library (no interpreter work per probe) whenever the
        # statistics come from this device alone; sharded solvers combine the shards' statistics in Python
        self._search_in_library = auto_lambda_stats == "device" and self._world == 1
        # which rule runs as kernels with the temperature resident in HBM (no host wait): mppi_set_auto_lambda
        self._rule_on_device = None
        self._auto_params_sent = None  # (rule parameter, lambda_min, lambda_max) the handle was last given
        if self._search_in_library:
            if self._auto_lambda == "ESSPS" and essps_search == "device":
                self._rule_on_device = "ESSPS"
                self._push_auto_lambda()
            elif self._auto_lambda == "LBPS" and lbps_search in ("brent", "grid"):
                self._rule_on_device = "LBPS"
                self._h.call("mppi_set_option", b"lbps_search", 1 if lbps_search == "grid" else 0)
                self._push_auto_lambda()
            elif self._auto_lambda == "MPO":
                self._rule_on_device = "MPO"
                self._h.call("mppi_mpo_reset", 1.0, 0.1, 0.2)  # mppi.py:191-200
                self._h.call("mppi_set_auto_lambda", _capi.AUTO_RULES["MPO"], 0.0, 0.0, 0.0)
                self._bind_log_temperature()
        if self._sg_on_device:  # step 7 runs inside mppi_finalize (taps computed above, history zero)
            self._h.call("mppi_set_sg_filter", self._coeffs.ctypes.data_as(C.c_void_p), int(len(self._coeffs)), None)
        self._uploaded = {}  # slot -> (id(cells), version)
        self._ref_uploaded = None
        self._x0_keep = None
        self._params_set = None

        # ---- RNG stream bookkeeping (src/pi_mpc/mppi.py:93,146-148; SURVEY B-Q1/Q2)
        self._solve_idx = 0
        # outputs of the per-tick calls (see _pool.py: the example loops are bound by the host at their own sizes)
        self._action_pool = RowPool((horizon, dim_control), self._device, self._dtype)
        self._state_pool = RowPool((1, horizon + 1, dim_state), self._device, self._dtype)
        self._top_pools = {}
        self._cpu_gen = None
        if noise_source == "torch_cpu":
            self._cpu_gen = torch.Generator(device="cpu")
            self._cpu_gen.manual_seed(self._seed)
            self._draw_torch_cpu()  # the constructor consumes one [N,T,dc] draw
        self._solve_idx = 1          # Philox: solve index 0 is the constructor's draw

        # ---- outputs / lazily materialised attributes
        T, dcn = horizon, dim_control
        self._action_out = torch.zeros(T, dcn, device=self._device, dtype=dtype)
        self._state_out = torch.zeros(1, T + 1, dim_state, device=self._device, dtype=dtype)
        self._stats = torch.zeros(4, device=self._device, dtype=dtype)
        self._summary = torch.zeros(_capi.SUMMARY_HEAD + T * dcn, device=self._device, dtype=dtype)
        self._gathered = None
        self._p2p = self._comm = False
        if self._world > 1 or self._force_exchange:
            self._setup_exchange()
        self._previous_action_seq = torch.zeros(T, dcn, device=self._device, dtype=dtype)
        # forward() in one library call (mppi_solve) when nothing needs the host between the steps
        in_library_exchange = self._p2p or self._comm  # the sharded solve needs nothing from the host either
        self._one_call = (self._model is not None and noise_source == "philox"
                          and ((self._world == 1 and not self._force_exchange) or in_library_exchange)
                          and not (self._world > 1 and lambda_ in ("MPO", "LBPS", "ESSPS"))
                          and not (use_sg_filter and not self._sg_on_device)
                          and (self._auto_lambda is None or self._rule_on_device is not None))
        # the batch-1 rollout of the solution completed lazily (see `lazy_state_seq` above)
        self._lazy_state = bool(lazy_state_seq and self._model is not None and not (use_sg_filter and not self._sg_on_device))
        if self._lazy_state:
            self._h.call("mppi_set_option", b"lazy_state_seq", 1)
        self._last_lambda = None
        self._fused_error_seen = False
        self._injected = None
        self._mean_of_last_solve = self._previous_action_seq
        self._state_seq_batch_buf = None
        self._perturbed_action_seqs_buf = None
        self._x0_tensor = None
        self._costs_host = None
        self._pick_strategy()

    def _bind_log_temperature(self) -> None:
        """`log_temperature` of the reference (an nn.Parameter, mppi.py:194-199) as a registered Parameter whose storage IS
        the dual the library steps on the device (zero-copy view through __cuda_array_interface__): always current on the
        solve's stream, listed by parameters() / state_dict(), never copied (valid while this solver is alive: the memory belongs to
        its handle; `state_dict()` values should be cloned by callers that outlive it).  Read-only in effect — the library derives the
        temperatures it uses when the dual steps; assigning `_lambda` is how a caller overrides a solve's temperature."""
        ptr = C.c_void_p(0)
        self._h.call("mppi_mpo_log_temperature_ptr", C.byref(ptr))

        class _View:  # (one fp32 at a device address owned by the handle, which this module keeps alive)
            __cuda_array_interface__ = {"shape": (1,), "typestr": "<f4", "data": (int(ptr.value), False), "version": 2}

        self._log_temperature_is_view = True
        try:
            t = torch.as_tensor(_View(), device=self._device)
        except Exception:  # noqa: BLE001  (a torch build without the interface: a copy, refreshed whenever the temperature is fetched)
            t = torch.zeros(1, device=self._device, dtype=torch.float32)
            self._log_temperature_is_view = False
        self.log_temperature = nn.Parameter(t, requires_grad=False)
        if not self.__dict__.get("_log_temperature_hooked"):
            self._log_temperature_hooked = True
            self.register_load_state_dict_post_hook(MPPI._after_load_state_dict)

    @staticmethod
    def _after_load_state_dict(module, incompatible_keys) -> None:
        """load_state_dict() copied a value into `log_temperature` — i.e. straight into the library's dual.  The temperature
        the next solve uses and the Adam moments are derived state the library keeps next to it: restart the dual from the
        loaded value (mppi_mpo_reset: lambda = exp(log T), moments zero, like a freshly constructed reference solver whose
        parameter was loaded before its optimizer state) instead of leaving them stale."""
        if module.__dict__.pop("_mpo_restored", False):  # `_extra_state` carried the whole dual (moments included): nothing to restart
            module._bind_log_temperature()
            return
        if module.__dict__.get("_auto_lambda") == "MPO" and module._rule_on_device == "MPO":
            v = float(module.log_temperature.detach().cpu()[0])
            module._h.call("mppi_mpo_reset", float(np.exp(v)), 0.1, 0.2)
            module._lambda_value = float(np.exp(v))
            module._bind_log_temperature()

    def _apply(self, fn, *args, **kwargs):
        """module.to() / .float() / .cpu() replace every Parameter by a converted COPY — for `log_temperature` that would be a
        detached value going stale while the library keeps stepping its own dual.  The solver is bound to its device and
        dtype (the reference's constructor arguments, mppi.py:45-46), so after the conversion the parameter is re-bound to
        the library's dual."""
        out = super()._apply(fn, *args, **kwargs)
        if self.__dict__.get("_log_temperature_is_view") is not None and "log_temperature" in self._parameters:
            self._bind_log_temperature()
        return out

    def _push_auto_lambda(self) -> None:
        """The device-resident ESSPS / LBPS rule reads its parameters from the handle (mppi_set_auto_lambda), the reference
        reads `_essps_target_ess` / `_lbps_delta` / `_lambda_min` / `_lambda_max` on every solve (mppi.py:341-370): hand them
        over again whenever a caller changed one of these attributes (a tuple compare per solve)."""
        p = (float(self._essps_target_ess if self._rule_on_device == "ESSPS" else self._lbps_delta), float(self._lambda_min),
             float(self._lambda_max))
        if p != self._auto_params_sent:
            self._h.call("mppi_set_auto_lambda", _capi.AUTO_RULES[self._rule_on_device], *p)
            self._auto_params_sent = p

    # ------------------------------------------------------------------ temperature (lazily fetched)
    def _fetch_lambda(self) -> None:
        """The temperature rule of the last solve ran on the device without a read-back: fetch its result now (waits for
        the stream that solve was enqueued on)."""
        nxt, used = C.c_double(0.0), C.c_double(0.0)
        self._h.call("mppi_get_lambda", C.byref(nxt), C.byref(used), self._lambda_stream)
        self._lambda_value = nxt.value
        if not self._used_known:  # (an explicitly given temperature is already on record)
            self._last_lambda_value = used.value
        self._lambda_pending = False
        if self.__dict__.get("_log_temperature_is_view") is False:  # MPO: lambda = exp(log T) (mppi.py:398)
            self.log_temperature.data.fill_(float(np.log(nxt.value)))

    @property
    def _lambda(self):
        """`_lambda` of the reference (mppi.py:183,349,370,398): the current temperature."""
        if self._lambda_pending:
            self._fetch_lambda()
        return self._lambda_value

    @_lambda.setter
    def _lambda(self, value) -> None:
        if self._lambda_pending:  # settle what the last solve left on the device first (its `_last_lambda`)
            self._fetch_lambda()
        self._lambda_value = value
        if getattr(self, "_rule_on_device", None) == "MPO":  # the next solve's weights use the caller's value
            self._lambda_override = float(value)

    @property
    def _last_lambda(self):
        """The temperature the weights of the last solve used."""
        if self._lambda_pending:
            self._fetch_lambda()
        return self._last_lambda_value

    @_last_lambda.setter
    def _last_lambda(self, value) -> None:
        self._last_lambda_value = value

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        # (the raw handle of torch's current stream on this device; torch.cuda.current_stream() builds a Stream object: ~4 us)
        return C.c_void_p(torch._C._cuda_getCurrentRawStream(self._device_index))

    def _draw_torch_cpu(self) -> torch.Tensor:
        """`MultivariateNormal(0, diag(sigma^2)).rsample([N])` on torch's CPU generator: bit-identical
        to randn(N,T,dc) * sigma (SURVEY B-Q1).  Always draws the GLOBAL [N,T,dc] block so that every
        shard sees the same stream."""
        eps = torch.randn(self._num_samples, self._horizon, self._dim_control, generator=self._cpu_gen,
                          dtype=torch.float32)
        return eps * self._sigmas.cpu()

    def _refresh_model_inputs(self):
        prov = self._cost_tag.provider if self._cost_tag is not None else None
        if prov is None:
            return
        spec = prov(self._cost_owner)
        params = spec.get("params")
        if params is not None:
            p = tuple(params)
            if p != self._params_set:
                arr = (C.c_float * len(p))(*p)
                self._h.call("mppi_set_model_params", arr, len(p))
                self._params_set = p
        for slot, grid in enumerate(spec.get("maps", ())):
            key = (id(grid.cells), grid.version)
            if self._uploaded.get(slot) != key:
                self._put_map(slot, grid)
                self._uploaded[slot] = key
        ref = spec.get("ref_path")
        if ref is not None and ref is not self._ref_uploaded:  # re-upload only a NEW reference window
            r = np.ascontiguousarray(ref, dtype=np.float32)
            self._h.call("mppi_set_reference", r.ctypes.data_as(C.c_void_p), r.shape[0], self._stream())
            self._ref_uploaded = ref

    def _put_map(self, slot: int, grid) -> None:
        """Occupancy grid -> device slot: rasterised on the device from the integer recipe when the map
        provides one (obstacle_map_2d.py:103-158, lane_map_2d.py:68-88), else uploaded as bytes."""
        geom = (grid.cells.shape[0], grid.cells.shape[1], float(grid.cell_size), float(grid.origin[0]),
                float(grid.origin[1]))
        rec = getattr(grid, "recipe", None)
        ip = lambda a: a.ctypes.data_as(C.c_void_p) if a.size else None  # noqa: E731
        if rec and rec["kind"] == "obstacles":
            circles = np.ascontiguousarray(rec["circles"], np.int32)
            rects = np.ascontiguousarray(rec["rects"], np.int32)
            self._h.call("mppi_build_obstacle_map", slot, *geom, ip(circles), len(circles), ip(rects), len(rects),
                         self._stream())
        elif rec and rec["kind"] == "lane" and len(rec["seeds"]):
            seeds = np.ascontiguousarray(rec["seeds"], np.int32)
            self._h.call("mppi_build_lane_map", slot, *geom, ip(seeds), len(seeds), int(rec["max_d2"]), self._stream())
        else:
            cells = np.ascontiguousarray(grid.cells, dtype=np.uint8)
            self._h.call("mppi_upload_map", slot, cells.ctypes.data_as(C.c_void_p), *geom)

    def device_map(self, slot: int) -> np.ndarray:
        """The occupancy grid the kernels read (uint8 [nx, ny]) — inspection / tests."""
        nx, ny = C.c_int(0), C.c_int(0)
        self._h.call("mppi_download_map", slot, None, C.byref(nx), C.byref(ny))
        out = np.empty((nx.value, ny.value), np.uint8)
        self._h.call("mppi_download_map", slot, out.ctypes.data_as(C.c_void_p), None, None)
        return out

    # ------------------------------------------------------------------ device-resident racing tick
    def set_center_path(self, path_xyyaw: np.ndarray, dind: np.ndarray, v_target: float) -> None:
        """Hand the racing centre line [n,3] and the window row offsets of calc_ref_trajectory (example/racing.py:
        161-218: dind[i] = int(round(travel_i / DL))) to the library, so that `update_reference_window` can rebuild
        `reference_path` on the device every tick.  Set-up path (synchronises)."""
        p = np.ascontiguousarray(path_xyyaw, dtype=np.float32)
        d = np.ascontiguousarray(dind, dtype=np.int32)
        assert p.ndim == 2 and p.shape[1] == 3 and d.ndim == 1
        self._h.call("mppi_set_center_path", p.ctypes.data_as(C.c_void_p), p.shape[0], d.ctypes.data_as(C.c_void_p),
                     d.shape[0], float(v_target))
        self._ref_uploaded = None

    def update_reference_window(self, state: torch.Tensor) -> None:
        """calc_ref_trajectory(state, path, current_path_index, ...) (example/racing.py:73-81,161-218) as one kernel
        on the current stream: the window goes straight into the cost's reference table and the path index stays on
        the device — no read-back of the state, no host work."""
        assert state.is_cuda and state.dtype == torch.float32 and state.shape == (self._dim_state,)
        self._window_state_keep = state  # alive until the enqueued kernel ran
        self._h.call("mppi_ref_window", _ptr(state), self._stream())
        self._ref_uploaded = None  # a later host-side reference must be uploaded again

    def reference_window(self) -> torch.Tensor:
        """The window the next solve's cost reads, as `reference_path` [T+1, 4] (x, y, yaw, v_target) on the device."""
        out = torch.empty(self._horizon + 1, 4, device=self._device, dtype=self._dtype)
        self._h.call("mppi_get_reference", _ptr(out), self._horizon + 1, 1, self._stream())
        return out

    @property
    def path_index(self) -> int:
        """`racing_controller.current_path_index` as kept on the device (reading synchronises)."""
        out = C.c_int32(0)
        self._h.call("mppi_get_path_index", C.byref(out), self._stream())
        return int(out.value)

    @path_index.setter
    def path_index(self, value: int) -> None:
        self._h.call("mppi_set_path_index", int(value), self._stream())

    def inject_noise(self, eps: torch.Tensor) -> None:
        """Parity hook: use `eps` [N_local,T,dc] (already scaled by sigma) for the next solve instead of
        drawing it."""
        assert eps.shape == (self._local_samples, self._horizon, self._dim_control)
        self._injected = eps.to(self._device, torch.float32).contiguous()

    def set_warm_start(self, mean, sg_history=None) -> None:
        """Overwrite `_previous_action_seq` (and optionally the SG history) — parity/test hook."""
        m = torch.as_tensor(np.asarray(mean, np.float32)).to(self._device).contiguous()
        assert m.shape == (self._horizon, self._dim_control)
        self._previous_action_seq = m
        self._h.call("mppi_set_mean", _ptr(m), 1, self._stream())
        torch.cuda.current_stream(self._device).synchronize()
        if sg_history is not None:
            self._actions_history_for_sg = np.asarray(sg_history, np.float32)

    def stage_times_ms(self) -> Dict[str, float]:
        """Mean device time per stage since the last call (needs set_option('timing', 1))."""
        out = (C.c_float * 8)()
        self._h.call("mppi_get_timing", out)
        names = ("sample", "rollout_cost", "weights_reduce", "finalize")
        # a stage that launched nothing (e.g. `sample` when the noise is regenerated in registers) is 0
        res = {n: max(float(out[i]), 0.0) for i, n in enumerate(names)} | {"calls": float(out[4 + 1])}
        if self._lazy_state:  # stand-alone completions of a pending state sequence (not part of `finalize`)
            o2 = (C.c_float * 2)()
            self._h.call("mppi_get_state_seq_timing", o2)
            res["state_seq_standalone"], res["state_seq_standalone_launches"] = max(float(o2[0]), 0.0), float(o2[1])
        return res

    def set_option(self, key: str, value: int) -> None:
        self._h.call("mppi_set_option", key.encode(), int(value))
        if key == "fused_rearm":
            self._fused_error_seen = False

    def reset(self):
        """Reset the previous action sequence (src/pi_mpc/mppi.py:212-221)."""
        self._previous_action_seq = torch.zeros(self._horizon, self._dim_control, device=self._device,
                                                dtype=self._dtype)
        self._h.call("mppi_set_mean", _ptr(self._previous_action_seq), 1, self._stream())
        self._actions_history_for_sg = np.zeros((self._horizon - 1, self._dim_control), np.float32)
        self._essps_prev = None  # the next ESSPS search starts cold (no grid clustered around the last temperature)
        self._h.call("mppi_set_option", b"essps_cold", 1)

    @property
    def _actions_history_for_sg(self) -> np.ndarray:
        """The last T-1 applied (filtered) first actions (src/pi_mpc/mppi.py:166,441-443); lives on the device
        when the filter runs there."""
        if self._sg_on_device:
            out = np.empty((self._horizon - 1, self._dim_control), np.float32)
            self._h.call("mppi_get_sg_history", out.ctypes.data_as(C.c_void_p))
            return out
        return self._sg_history_host

    @_actions_history_for_sg.setter
    def _actions_history_for_sg(self, value) -> None:
        hist = np.ascontiguousarray(value, dtype=np.float32).reshape(self._horizon - 1, self._dim_control).copy()
        if self._sg_on_device:
            self._h.call("mppi_set_sg_filter", self._coeffs.ctypes.data_as(C.c_void_p), int(len(self._coeffs)),
                         hist.ctypes.data_as(C.c_void_p))
        else:
            self._sg_history_host = hist

    # ------------------------------------------------------------------ lazily completed state sequence
    def _returned_state_seq(self):
        """What forward() returns as `state_seq`: the tensor itself, or — rollout pending — its completing wrapper."""
        if not self._lazy_state:
            return self._state_out
        serial, pending = C.c_uint32(0), C.c_int(0)
        self._h.call("mppi_state_seq_serial", C.byref(serial), C.byref(pending))
        if not pending.value:  # (this solve ran as a single launch: nothing left to do)
            return self._state_out
        n = int(serial.value)
        return _DeferredStateSeq.wrap(self._state_out, lambda _t: self._h.call("mppi_join_state_seq", n, self._stream()))

    def join_state_seq(self) -> None:
        """Complete the newest `state_seq` on torch's current stream if it is still pending (for consumers that read it
        outside torch)."""
        if self._lazy_state:
            self._h.call("mppi_join_state_seq", 0, self._stream())

    # ------------------------------------------------------------------ forward
    def forward(self, state: torch.Tensor, info: Dict = _NO_INFO) -> Tuple[torch.Tensor, torch.Tensor]:
        """Solve one MPPI step (src/pi_mpc/mppi.py:223-460).

        One of three strategies, resolved once in __init__ (`_solve`, see _pick_strategy):
          _solve_one_call   native model and nothing needs the host between the steps: ONE library call (mppi_solve);
          _solve_by_steps   native model through the individual entry points, with host work in between (torch-CPU noise,
                            a host-side temperature rule, the exchange through torch.distributed, the host SG filter);
          _solve_generic    opaque `dynamics` / `cost_func` callables around the library's sampler / softmax / reduction.
        Injected noise (parity hook) sends a one-call solver through the entry points for that one solve."""
        assert state.shape == (self._dim_state,)
        if self._injected is not None and self._one_call:
            out = self._solve_by_steps(state, info)
        else:
            out = self._solve(state, info)
        if info is not _NO_INFO and self._model is not None:
            self._leave_info(info, state)
        return out

    def _leave_info(self, info: Dict, state) -> None:
        """What the reference's cost loop leaves in the caller's dict (mppi.py:299-306,318-322) — the native path has no
        such loop: `t` = T-1, `initial_state` = the start state seen by every sample (an expanded view), and `prev_state` =
        S[:, T-1] / `prev_action` = U[:, max(T-2, 0)] as tensors that are built on first use (_LazyInfoTensor).  Only for a
        dict the caller passed: the default one cannot be read back."""
        N, T = self._local_samples, self._horizon
        info["t"] = T - 1
        x0 = state if (torch.is_tensor(state) and state.is_cuda) else torch.as_tensor(
            np.asarray(state.detach().cpu() if torch.is_tensor(state) else state, dtype=np.float32)).to(self._device)
        info["initial_state"] = x0.detach().to(self._dtype).reshape(1, -1).expand(N, -1)
        info["prev_state"] = _LazyInfoTensor.make(lambda: self._state_seq_batch[:, T - 1, :])
        info["prev_action"] = _LazyInfoTensor.make(lambda: self._perturbed_action_seqs[:, max(T - 2, 0), :])

    def _pick_strategy(self) -> None:
        """Resolve every per-solve mode decision ONCE (called at the end of __init__): the solve strategy and the step that
        produces this solve's temperature (src/pi_mpc/mppi.py:341-370; MPO: the temperature the previous solve left)."""
        if self._model is None:
            self._solve = self._solve_generic
        elif self._one_call:
            self._solve = self._solve_one_call
        else:
            self._solve = self._solve_by_steps
        on_dev = self._auto_lambda_stats == "device"
        rule = self._auto_lambda
        if rule is None:
            step = self._lam_fixed
        elif self._rule_on_device == "LBPS":
            step = self._lam_lbps_device
        elif self._rule_on_device == "MPO":
            step = self._lam_mpo_device
        elif self._rule_on_device == "ESSPS":
            step = self._lam_essps_device
        elif rule == "LBPS":
            step = self._lam_lbps_library if self._search_in_library else self._lam_lbps_host
        elif rule == "ESSPS":
            step = (self._lam_essps_library if (on_dev and self._essps_search == "grid" and self._world == 1)
                    else self._lam_essps_host)
        else:  # MPO with the dual on the host: nothing before the weights (the dual steps after them)
            step = self._lam_fixed
        self._temperature_step = step
        self._stats_on_device = on_dev
        self._sharded = self._world > 1 or self._force_exchange
        self._host_sg = self._use_sg_filter and not self._sg_on_device  # host round trip only for sg_filter="host"

    # ---- stages shared by the step-by-step strategies
    def _bind_state(self, state, st) -> None:
        if torch.is_tensor(state) and state.is_cuda:
            # zero-copy: this solve's kernels read the caller's tensor (kept alive until the next solve); the rollout
            # kernel snapshots it, so later re-rolls (get_top_samples, _state_seq_batch) do not depend on it
            self._x0_keep = state.detach().to(self._device, self._dtype).contiguous()
            self._h.call("mppi_bind_state", _ptr(self._x0_keep))
        else:
            x0h = np.ascontiguousarray(state.detach().cpu().numpy() if torch.is_tensor(state) else state,
                                       dtype=np.float32)
            self._h.call("mppi_set_state", x0h.ctypes.data_as(C.c_void_p), 0, st)
        self._refresh_model_inputs()
        self._mean_of_last_solve = self._previous_action_seq  # the mean this solve samples around

    def _draw_noise(self, st) -> None:
        """Step 1 (src/pi_mpc/mppi.py:261-263): injected, the reference's CPU stream, or the device Philox stream."""
        h = self._h
        if self._injected is not None:
            h.call("mppi_inject_noise", _ptr(self._injected), st)
            self._injected = None
        elif self._noise_source == "torch_cpu":
            eps = self._draw_torch_cpu()
            lo = self._sample_offset
            eps = eps[lo:lo + self._local_samples].to(self._device).contiguous()
            h.call("mppi_inject_noise", _ptr(eps), st)
        else:
            h.call("mppi_sample", self._solve_idx, st)
        self._solve_idx += 1

    # ---- Step 4, one method per way of getting this solve's temperature (picked once: _pick_strategy)
    def _lam_fixed(self, st) -> None:
        pass

    def _lam_lbps_device(self, st) -> None:  # the Brent search (or the grid rounds) as kernels: nothing is read back
        self._check_search_error()
        self._h.call("mppi_lbps_lambda_device" if self._lbps_search == "grid" else "mppi_lbps_brent_device",
                     float(self._lbps_delta), float(self._lambda_min), float(self._lambda_max), st)
        self._lambda_pending, self._lambda_stream = True, st

    def _check_search_error(self) -> None:
        """A poll of the device-resident Brent search gave up on an EARLIER solve (one of its blocks never became resident: the
        GPU is shared with other work) — that solve's temperature and outputs are NaN.  Raised once per occurrence."""
        h = self._h
        if self._rule_on_device == "LBPS" and self._lbps_search == "brent" and h.lib.mppi_search_error(h.h):
            h.call("mppi_set_option", b"search_rearm", 1)
            self.reset()  # (that solve's NaN plan was stored as the warm start: start over from zeros, like reset())
            raise _capi.MppiError("the device-resident LBPS search gave up waiting for one of its blocks on an earlier solve "
                                  "(budget: set_option('fused_timeout_us', ...), default 20 ms; is the GPU shared with other "
                                  "work?): that solve returned NaN and the warm start has been reset to zeros.  "
                                  "lbps_search='brent_host' searches with a host loop instead")

    def _lam_mpo_device(self, st) -> None:  # this solve uses the temperature the dual left in HBM (or the caller's)
        self._lambda_pending, self._lambda_stream = self._lambda_override is None, st

    def _lam_essps_device(self, st) -> None:  # the whole search as kernels on this stream; the temperature stays in HBM
        self._h.call("mppi_essps_lambda_device", float(self._essps_target_ess), float(self._lambda_min),
                     float(self._lambda_max), st)
        self._lambda_pending, self._lambda_stream = True, st

    def _lam_lbps_library(self, st) -> None:  # bounded Brent inside the library (same algorithm as scipy's, host C++)
        lam_out = C.c_double(0.0)
        self._h.call("mppi_lbps_lambda", float(self._lbps_delta), float(self._lambda_min), float(self._lambda_max),
                     C.byref(lam_out), st)
        self._lambda = lam_out.value

    def _lam_lbps_host(self, st) -> None:  # scipy's bounded Brent over device statistics or over the costs on the host
        self._lambda = (_host.lbps_lambda_stats(self._softmax_stats, self._lbps_delta, self._lambda_min, self._lambda_max)
                        if self._stats_on_device else
                        _host.lbps_lambda(self._costs_host, self._lbps_delta, self._lambda_min, self._lambda_max))

    def _lam_essps_library(self, st) -> None:  # the grid search as a host loop inside the library (one read-back per grid)
        lam_out = C.c_double(0.0)
        self._h.call("mppi_essps_lambda", float(self._essps_target_ess), float(self._lambda_min), float(self._lambda_max),
                     C.byref(lam_out), st)
        self._lambda = lam_out.value

    def _lam_essps_host(self, st) -> None:
        if not self._stats_on_device:  # the reference's own calls on the costs (mppi.py:351-370)
            self._lambda = _host.essps_lambda(self._costs_host, self._essps_target_ess, self._lambda_min, self._lambda_max)
        elif self._essps_search == "brentq":
            self._lambda = _host.essps_lambda_stats(self._softmax_stats, self._essps_target_ess, self._lambda_min,
                                                    self._lambda_max)
        else:
            self._lambda = _host.essps_lambda_grid(self._ess_grid, self._essps_target_ess, self._lambda_min,
                                                   self._lambda_max, lam_prev=self._essps_prev)
        self._essps_prev = float(self._lambda)  # the next grid search starts around it (one exchange instead of two)

    def _temperature(self, st) -> float:
        """Run the rule's step and return what the reduce / finalize entry points are given: a value, or the marker for
        "read it from device memory"."""
        self._costs_host = None
        if self._auto_lambda is not None and not self._stats_on_device:
            self._costs_host = self._gather_costs_host()
        self._temperature_step(st)
        if self._lambda_pending:
            self._used_known = False
            return _capi.LAMBDA_DEVICE  # weights_reduce / finalize read the temperature from device memory
        lam = float(self._lambda_override if self._lambda_override is not None else self._lambda)
        self._last_lambda = lam
        self._used_known = True
        return lam

    def _reduce_and_exchange(self, lam, st):
        """Steps 5-6 (src/pi_mpc/mppi.py:376-385): weights + weighted mean of this shard and, when sharded, the solve's only
        exchange.  Returns (gathered summaries or None, number of shards)."""
        h = self._h
        if self._p2p or self._comm:  # the library exchanges the shard summaries itself (buffers / its own all_gather)
            if self._p2p and h.lib.mppi_p2p_error(h.h):
                raise _capi.MppiError("peer-to-peer exchange timed out on an earlier solve (a rank is missing or stalled)")
            h.call("mppi_weights_reduce", lam, None, st)
            return None, self._world
        h.call("mppi_weights_reduce", lam, _ptr(self._summary) if self._sharded else None, st)
        if not self._sharded:
            return None, 1
        if self._gathered is None:  # 4+T*dc floats per rank over RCCL/xGMI
            self._gathered = torch.empty(self._world, self._summary.numel(), device=self._device, dtype=self._dtype)
        return all_gather_summaries(self._summary, self._pg, out=self._gathered), self._world

    def _finalize(self, summaries, nsh, lam, st, native: bool) -> None:
        """Steps 6-8 (src/pi_mpc/mppi.py:381-385,448-452): normalise, warm start, batch-1 rollout — into fresh output tensors
        (the kernel writes straight into what is returned)."""
        self._action_out = torch.empty(self._horizon, self._dim_control, device=self._device, dtype=self._dtype)
        self._state_out = torch.empty(1, self._horizon + 1, self._dim_state, device=self._device, dtype=self._dtype)
        self._h.call("mppi_finalize", _ptr(summaries), nsh, lam, 0 if self._host_sg else 1, _ptr(self._action_out),
                     _ptr(self._state_out) if (native and not self._host_sg) else None, _ptr(self._stats), st)

    def _mpo_after_weights(self, st) -> None:
        """MPO steps its dual AFTER the weights; the result is the next solve's temperature (mppi.py:387-398)."""
        if self._auto_lambda != "MPO":
            return
        if self._rule_on_device == "MPO":
            self._h.call("mppi_mpo_step_device", st)  # dual, moments and the next temperature stay in HBM
            self._lambda_pending, self._lambda_stream = True, st
        else:
            self._lambda = (self._mpo.step_from_stats(self._softmax_stats(self._mpo.temperature())) if self._stats_on_device
                            else self._mpo.step(self._costs_host))

    def _host_sg_step(self, st, native: bool) -> None:
        """Step 7 on the host (sg_filter="host"; src/pi_mpc/mppi.py:423-443)."""
        a = self._action_out.cpu().numpy()
        a = _host.sg_filter_sequence(self._actions_history_for_sg, a, self._coeffs)
        self._action_out.copy_(torch.from_numpy(a))
        self._h.call("mppi_set_mean", _ptr(self._action_out), 1, st)
        if native:
            self._h.call("mppi_rollout_actions", _ptr(self._action_out), 1, None, _ptr(self._state_out), st)
        self._actions_history_for_sg = np.concatenate([self._actions_history_for_sg[1:], a[0][None, :]])

    # ---- the strategies
    def _solve_by_steps(self, state, info):
        """A native model through the individual entry points (same kernels as mppi_solve)."""
        st = self._stream()
        self._bind_state(state, st)
        self._draw_noise(st)
        self._h.call("mppi_rollout_cost", st)  # Steps 1b-3: clamp, rollout, costs (src/pi_mpc/mppi.py:266-336)
        lam = self._temperature(st)
        summaries, nsh = self._reduce_and_exchange(lam, st)
        self._finalize(summaries, nsh, lam, st, native=True)
        self._mpo_after_weights(st)
        if self._host_sg:
            self._host_sg_step(st, native=True)
        self._lambda_override = None
        self._previous_action_seq = self._action_out
        return self._action_out, (self._state_out if self._host_sg else self._returned_state_seq())

    def _solve_generic(self, state, info):
        """Opaque callables: the reference's two T-step loops over the user's torch `dynamics` / `cost_func` on GPU tensors
        (or their captured hipGraph) between the library's sampler and its softmax / reduction / warm start."""
        st = self._stream()
        self._bind_state(state, st)
        self._draw_noise(st)
        self._generic_rollout_costs(state, info)  # Steps 1b-3 with the callables; the summed costs go back to the library
        lam = self._temperature(st)
        summaries, nsh = self._reduce_and_exchange(lam, st)
        self._finalize(summaries, nsh, lam, st, native=False)
        self._mpo_after_weights(st)
        if self._host_sg:
            self._host_sg_step(st, native=False)
        # Step 8 with the user's dynamics (src/pi_mpc/mppi.py:448-449,508-524)
        if self._graph_state == "replay":
            self._state_out = self._states_prediction_graphed()
        else:
            self._state_out = self._states_prediction(self._x0_tensor, self._action_out.repeat(1, 1, 1))
        self._lambda_override = None
        self._previous_action_seq = self._action_out
        return self._action_out, self._state_out

    def _solve_one_call(self, state, info=None):
        """forward() through mppi_solve: the same kernel sequence as _solve_by_steps in one library call (native model,
        device noise, a fixed temperature or a device-resident rule, one GPU or an in-library exchange)."""
        h, st = self._h, self._stream()
        me = self.__dict__  # (private state is written straight into the instance dict: see __setattr__ above)
        if torch.is_tensor(state) and state.is_cuda:
            if state.dtype is self._dtype and state.device == self._device and state.is_contiguous():
                me["_x0_keep"] = state  # zero-copy as it is (kept alive until the next solve)
            else:
                me["_x0_keep"] = state.detach().to(self._device, self._dtype).contiguous()
            x0p = _ptr(self._x0_keep)
        else:
            x0h = np.ascontiguousarray(state.detach().cpu().numpy() if torch.is_tensor(state) else state,
                                       dtype=np.float32)
            h.call("mppi_set_state", x0h.ctypes.data_as(C.c_void_p), 0, st)
            x0p = None
        self._refresh_model_inputs()
        if not self._fused_error_seen and h.lib.mppi_fused_error(h.h):
            me["_fused_error_seen"] = True  # (from now on the library stays on the multi-kernel path)
            raise _capi.MppiError("a single-launch solve gave up waiting for one of its blocks (default budget 20 ms; is the GPU "
                                  "shared with other work?): that solve returned the previous plan instead of a new one and NaN "
                                  "statistics (last_stats()); later solves use the multi-kernel path.  "
                                  "set_option('fused_timeout_us', ...) widens the budget, set_option('fused_rearm', 1) allows "
                                  "the single launch again")
        me["_mean_of_last_solve"] = self._previous_action_seq
        if self._auto_lambda is None:
            lam = float(self._lambda_value)
            me["_last_lambda_value"], me["_used_known"] = lam, True
        elif self._lambda_override is not None:  # MPO with a temperature assigned by the caller (the dual still steps)
            lam = me["_last_lambda_value"] = self._lambda_override
            me["_lambda_pending"], me["_lambda_stream"], me["_used_known"] = True, st, True
            me["_lambda_override"] = None
        else:  # the configured rule runs on the device; the temperature is fetched when somebody asks for it
            if self._rule_on_device != "MPO":
                self._push_auto_lambda()
                if self._rule_on_device == "LBPS":
                    self._check_search_error()
            lam = _capi.LAMBDA_DEVICE
            me["_lambda_pending"], me["_lambda_stream"], me["_used_known"] = True, st, False
        # (the previous solve's state tensor stays alive across this launch: with a lazily completed state sequence this
        # solve's rollout launch may still write it)
        prev_state_keep = self._state_out
        cap = capturing()
        me["_action_out"], me["_state_out"] = self._action_pool.take(st.value, cap), self._state_pool.take(st.value, cap)
        h.call("mppi_solve", x0p, self._solve_idx, lam, _ptr(self._action_out), _ptr(self._state_out), _ptr(self._stats), st)
        del prev_state_keep
        me["_solve_idx"] = self._solve_idx + 1
        me["_previous_action_seq"] = self._action_out
        return self._action_out, self._returned_state_seq()
