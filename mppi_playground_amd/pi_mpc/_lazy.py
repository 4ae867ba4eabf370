"""Tensors that complete themselves on first use through torch (host-side helpers of pi_mpc.mppi.MPPI): the lazily completed
`state_seq` of a solve (opt-in, `lazy_state_seq=True`) and the entries the reference leaves in the caller's `info` dict."""
from __future__ import annotations

from typing import Optional

import torch


def _ptr(t: Optional[torch.Tensor]):
    # (a plain int: every entry point declares its argtypes, so ctypes converts it to void* itself — building a c_void_p
    # here cost ~0.25 us, seven times per tick of the example loop)
    return None if t is None else t.data_ptr()


class _DeferredStateSeq(torch.Tensor):
    """`state_seq` of a solve whose batch-1 rollout (src/pi_mpc/mppi.py:448-449) is completed lazily
    (MPPI(..., lazy_state_seq=True); mppi_set_option("lazy_state_seq")): a plain float32 tensor [1, T+1, ds] whose FIRST
    use through torch (indexing, .cpu(), arithmetic, printing, ...) completes it on the consumer's current stream if the
    next solve has not done so already (mppi_join_state_seq: at most one small kernel launch, no host synchronisation).
    The T dependent steps are off the solve's critical path: in a control loop they ride in one extra block of the NEXT
    solve's rollout launch.  Results of operations on it are ordinary tensors.  Consumers that bypass torch (a raw
    data_ptr() handed to another library) must call `solver.join_state_seq()` first."""

    @staticmethod
    def wrap(t: torch.Tensor, join) -> "_DeferredStateSeq":
        r = torch.Tensor._make_subclass(_DeferredStateSeq, t)
        r.__dict__["_mppi_join"] = join
        return r

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        def join(a):
            if isinstance(a, _DeferredStateSeq):
                j = a.__dict__.get("_mppi_join")
                if j is not None:
                    a.__dict__["_mppi_join"] = None
                    j(a)
            elif isinstance(a, (list, tuple)):  # torch.cat([...]), torch.stack((...))
                for b in a:
                    join(b)

        for a in args:
            join(a)
        if kwargs:
            for a in kwargs.values():
                join(a)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **(kwargs or {}))


class _LazyInfoTensor(torch.Tensor):
    """An entry the reference leaves in the CALLER's `info` dict after a solve (src/pi_mpc/mppi.py:299-306,318-322:
    `prev_state` = S[:, T-1], `prev_action` = U[:, T-2]) whose value the native path never holds — the N state trajectories
    are not materialised, the clamped actions are regenerated in registers.  It stands in the dict as a tensor that is
    built from the solve's noise (a re-roll / an export launch) the first time a torch function touches it; nobody pays
    for it otherwise.  Like the reference's views it describes the LAST solve: build it before the next one."""

    @staticmethod
    def make(thunk) -> "_LazyInfoTensor":
        r = torch.Tensor._make_subclass(_LazyInfoTensor, torch.empty(0))
        r.__dict__["_mppi_thunk"], r.__dict__["_mppi_value"] = thunk, None
        return r

    def materialize(self) -> torch.Tensor:
        d = self.__dict__
        if d["_mppi_value"] is None:
            d["_mppi_value"], d["_mppi_thunk"] = d["_mppi_thunk"](), None
        return d["_mppi_value"]

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        def real(a):
            if isinstance(a, _LazyInfoTensor):
                return a.materialize()
            if isinstance(a, (list, tuple)):
                return type(a)(real(b) for b in a)
            return a

        with torch._C.DisableTorchFunctionSubclass():
            return func(*[real(a) for a in args], **{k: real(v) for k, v in (kwargs or {}).items()})
