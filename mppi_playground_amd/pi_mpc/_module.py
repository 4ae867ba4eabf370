"""What the reference's MPPI gets for free from being a plain nn.Module (src/pi_mpc/mppi.py:16: every attribute is a tensor or
a Python value) and this one has to provide itself, because its state lives behind a library handle: `copy.deepcopy`, pickling
(`pickle`, `torch.save(solver)`), and a `state_dict()` that round-trips the solver's dynamic state."""
from __future__ import annotations

import copy
import ctypes as C
from typing import Any, Dict

import numpy as np
import torch

from mppi_playground_amd import _capi
from pi_mpc.native import resolve


class ModuleProtocolMixin:
    # ------------------------------------------------------------------ copy.deepcopy
    def __deepcopy__(self, memo):
        """A second solver that continues EXACTLY like this one: same constructor arguments, a handle of its own, and every
        piece of dynamic state copied on the device (mppi_clone_state: warm start, noise identity, the last solve's costs,
        Savitzky-Golay history, the temperature and the device-resident search / dual state, model parameters, maps,
        reference window and path index, options) or on the host (RNG stream position, the torch-CPU generator, temperatures
        already fetched).  The callables are deep-copied like the reference's attributes would be — with them the environment
        / controller objects that own the model's parameters — and the native tags re-resolved on the copies."""
        if self._world > 1 or self._force_exchange:
            raise TypeError("a sharded MPPI (shard_samples=True) is bound to its process group and its peers' buffers and cannot "
                            "be deep-copied; construct a second solver on every rank instead")
        if self._lambda_pending:  # settle what the last solve left on the device (so that the host mirrors are current)
            self._fetch_lambda()
        new = type(self)(**self._ctor)
        memo[id(self)] = new
        dyn, cst = copy.deepcopy(self._dynamics, memo), copy.deepcopy(self._cost_func, memo)
        me, it = self.__dict__, new.__dict__
        it["_dynamics"], it["_cost_func"] = dyn, cst
        it["_ctor"] = dict(self._ctor, dynamics=dyn, cost_func=cst)  # (a copy of the copy starts from ITS callables, and the originals are not kept alive)
        if self._model is not None and self._recognized is None:
            d, c = resolve(dyn), resolve(cst)
            if d and c:
                it["_dyn_tag"], it["_dyn_owner"] = d
                it["_cost_tag"], it["_cost_owner"] = c
        elif self._recognized is not None:
            it["_recognized"] = (dyn, cst)
        _capi.load().mppi_clone_state.restype = C.c_int
        rc = new._h.lib.mppi_clone_state(new._h.h, self._h.h)
        if rc != 0:
            raise _capi.MppiError(f"mppi_clone_state failed ({rc}): {new._h.lib.mppi_last_error(new._h.h).decode()}")
        # host-side dynamic state
        for k in ("_solve_idx", "_lambda_value", "_last_lambda_value", "_lambda_override", "_used_known", "_essps_prev",
                  "_params_set", "_fused_error_seen", "_auto_params_sent", "_lbps_delta", "_essps_target_ess", "_lambda_min",
                  "_lambda_max"):
            it[k] = me[k]
        it["_lambda_pending"], it["_lambda_stream"] = False, None
        for k in ("_previous_action_seq", "_action_out", "_state_out", "_stats", "_summary", "_mean_of_last_solve", "_injected",
                  "_x0_tensor"):
            v = me.get(k)
            it[k] = v.detach().clone() if torch.is_tensor(v) else v
        if me.get("_previous_action_seq") is me.get("_action_out"):  # (the reference keeps ONE tensor for both, mppi.py:452,460)
            it["_previous_action_seq"] = it["_action_out"]
        if me.get("_mean_of_last_solve") is me.get("_previous_action_seq"):
            it["_mean_of_last_solve"] = it["_previous_action_seq"]
        x0 = me.get("_x0_keep")
        it["_x0_keep"] = None  # (the library copied a borrowed state into its own buffer)
        if torch.is_tensor(x0):
            it["_x0_keep"] = x0.detach().clone()
        it["_sg_history_host"] = np.array(me["_sg_history_host"], copy=True)
        if me.get("_mpo") is not None:
            it["_mpo"] = copy.deepcopy(me["_mpo"])
        if self._cpu_gen is not None:
            new._cpu_gen.set_state(self._cpu_gen.get_state())
        it["_uploaded"], it["_ref_uploaded"] = {}, None  # (maps / a host-side reference are sent again when the copies' ids differ)
        return new

    # ------------------------------------------------------------------ pickle / torch.save(solver)
    def __getstate__(self):
        raise TypeError(
            "an MPPI solver cannot be pickled (pickle.dumps / torch.save(solver)): its buffers live behind a device handle of "
            "libmppi_hip.so (mppi_handle_t), not in Python attributes.  In-process copy: copy.deepcopy(solver).  Across "
            "processes: torch.save(solver.state_dict()) and load_state_dict() into a solver constructed with the same "
            "arguments — the state dict carries the warm start, the Savitzky-Golay history, the RNG stream position and the "
            "temperature state")

    # ------------------------------------------------------------------ state_dict round trip
    def get_extra_state(self) -> Dict[str, Any]:
        """The solver's dynamic state for state_dict() (entry `_extra_state`; the reference keeps the same things as plain
        attributes, which its state_dict() silently drops): warm start, Savitzky-Golay history, RNG stream position (Philox
        solve index and the torch-CPU generator), the temperature and — MPO — the dual with its Adam moments."""
        if self._lambda_pending:
            self._fetch_lambda()
        mean = torch.empty(self._horizon, self._dim_control, device=self._device, dtype=self._dtype)
        self._h.call("mppi_get_mean", mean.data_ptr(), 1, self._stream())
        out = {"version": 1, "previous_action_seq": mean.cpu(),
               "sg_history": torch.from_numpy(np.array(self._actions_history_for_sg, copy=True)),
               "solve_idx": int(self._solve_idx), "lambda": self._lambda_value, "last_lambda": self._last_lambda_value,
               "cpu_generator": None if self._cpu_gen is None else self._cpu_gen.get_state(), "mpo": None}
        if self._auto_lambda == "MPO":
            if self._rule_on_device == "MPO":
                st4 = (C.c_double * 4)()
                self._h.call("mppi_mpo_state", st4)
                out["mpo"] = [float(v) for v in st4]
            else:
                m = self._mpo
                out["mpo"] = [float(m.log_temperature), float(m.m), float(m.v), float(m.t)]
        return out

    def set_extra_state(self, state: Dict[str, Any]) -> None:
        if not state:
            return
        if state.get("version") != 1:
            raise RuntimeError(f"MPPI state of version {state.get('version')!r}: this build reads version 1")
        a = state["previous_action_seq"]
        if tuple(a.shape) != (self._horizon, self._dim_control):
            raise RuntimeError(f"previous_action_seq of shape {tuple(a.shape)} does not fit horizon {self._horizon} x "
                               f"dim_control {self._dim_control}")
        self.set_warm_start(a.numpy(), sg_history=state["sg_history"].numpy())
        self._solve_idx = int(state["solve_idx"])
        if state.get("cpu_generator") is not None and self._cpu_gen is not None:
            self._cpu_gen.set_state(state["cpu_generator"])
        if state.get("mpo") is not None and self._auto_lambda == "MPO":
            if self._rule_on_device == "MPO":
                self._h.call("mppi_mpo_set_state", (C.c_double * 4)(*state["mpo"]))
                self.__dict__["_mpo_restored"] = True  # (tells the load_state_dict hook not to restart the dual)
            else:
                m = self._mpo
                m.log_temperature, m.m, m.v, m.t = (np.float32(state["mpo"][0]), np.float32(state["mpo"][1]),
                                                    np.float32(state["mpo"][2]), int(state["mpo"][3]))
        self._lambda_pending = False
        if state.get("lambda") is not None:
            self._lambda_value = state["lambda"]
            if self._rule_on_device == "MPO":
                self._lambda_override = None
        self._last_lambda_value = state.get("last_lambda")

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """A state dict written by the REFERENCE's solver has no `_extra_state` entry (it holds `log_temperature` at most):
        loading it is not an error — the dynamic state simply keeps its current values."""
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)
        key = prefix + torch.nn.modules.module._EXTRA_STATE_KEY_SUFFIX
        if key in missing_keys:
            missing_keys.remove(key)
