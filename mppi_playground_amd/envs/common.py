"""Shared model math in torch (used by env.step with batch 1 and by any torch caller of the plugins)."""
import torch

from mppi_playground_amd._pool import RowPool, capturing


def angle_normalize(x: torch.Tensor) -> torch.Tensor:
    """Wrap to [-pi, pi): ((x + pi) mod 2 pi) - pi with Python-style modulo."""
    return torch.remainder(x + torch.pi, 2 * torch.pi) - torch.pi


class NativeStep:
    """env.step() as ONE launch of the library's model functor (mppi_model_step: next = dynamics(state, clamp(u)) with
    the library math in the reference's operation order, plus the goal test) instead of ~20 batch-1 torch kernels.
    Holds the constants of the call (read back from the env's device tensors once).  Raises when the extension is
    missing: there is no silent torch fallback for a GPU env constructed with native_step=True."""

    def __init__(self, model: str, params, u_min, u_max, goal_xy, goal_threshold: float, dim_state: int, device, dtype):
        import ctypes as C

        from mppi_playground_amd import _capi

        f = lambda v: (C.c_float * len(v))(*[float(x) for x in v])  # noqa: E731
        self._capi, self._lib = _capi, _capi.load()
        self._model = _capi.MODEL_IDS[model]
        self._params, self._lo, self._hi, self._goal = f(params), f(u_min), f(u_max), f(goal_xy)
        self._thr, self._ds, self._device, self._dtype = float(goal_threshold), dim_state, device, dtype
        self._n_params = len(self._params)
        # (the tick is bound by the host at the examples' sizes: outputs come from row pools, the stream as a raw handle —
        # both of the device the STATE lives on: the env's device is the reference's unindexed "cuda")
        self._pool_device, self._next_pool, self._reached_pool = None, None, None

    def __deepcopy__(self, memo):
        """copy.deepcopy(env) (e.g. as part of copy.deepcopy(controller), pi_mpc/_module.py): the constants are immutable and
        shared, the library module / handle are process-wide, the output pools start empty."""
        new = type(self).__new__(type(self))
        new.__dict__.update(self.__dict__)
        new._pool_device, new._next_pool, new._reached_pool = None, None, None
        memo[id(self)] = new
        return new

    def __call__(self, state: torch.Tensor, u: torch.Tensor):
        assert u.dtype == torch.float32 and state.dtype == torch.float32 and state.is_contiguous() and state.is_cuda
        u = u if u.is_contiguous() else u.contiguous()
        dev = state.device
        if dev != self._pool_device:
            self._pool_device = dev
            self._next_pool, self._reached_pool = RowPool((self._ds,), dev, self._dtype), RowPool((), dev, torch.bool)
        st = torch._C._cuda_getCurrentRawStream(dev.index)
        cap = capturing()
        nxt, reached = self._next_pool.take(st, cap), self._reached_pool.take(st, cap)
        rc = self._lib.mppi_model_step(self._model, self._params, self._n_params, self._lo, self._hi, state.data_ptr(),
                                       u.data_ptr(), nxt.data_ptr(), self._goal, self._thr, reached.data_ptr(), st)
        if rc != 0:
            raise self._capi.MppiError(f"mppi_model_step failed ({rc})")
        return nxt, reached
