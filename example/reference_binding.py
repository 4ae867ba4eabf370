"""The ctypes binding a maintainer of kohonda/mppi_playground would add as `src/pi_mpc/_hip.py` (INTEGRATION.md,
section B), as a runnable file: it uses NOTHING of this repository's Python — only `libmppi_hip.so` through the C ABI of
include/mppi_hip.h — so it is also the proof that the boundary is self-sufficient.  tests/test_gpu_parity.py runs it
against this build's own `pi_mpc.mppi.MPPI` (same seed, same solve index: bit-identical action and state sequences).

    hip = HipForward(lib_path, model="pendulum", horizon=15, num_samples=1000, u_min=[-2], u_max=[2], sigmas=[1], seed=42)
    a, x = hip.forward(state_cuda_tensor, lam=1.0)       # replaces mppi.py:255-336, :376-385, :448-452
"""
import ctypes as C

import torch

MODELS = {"pendulum": (0, 2, 1), "cartpole": (1, 4, 1), "mountaincar": (2, 2, 1), "nav2d": (3, 3, 2), "racing": (4, 4, 2)}
MPPI_LAMBDA_DEVICE = -1.0


class MppiConfig(C.Structure):  # include/mppi_hip.h: struct MppiConfig
    _fields_ = [("model", C.c_int32), ("horizon", C.c_int32), ("dim_state", C.c_int32), ("dim_control", C.c_int32),
                ("num_samples", C.c_int64), ("sample_offset", C.c_int64), ("inherit_count", C.c_int64),
                ("u_min", C.c_float * 4), ("u_max", C.c_float * 4), ("sigmas", C.c_float * 4),
                ("seed", C.c_uint64), ("device", C.c_int32), ("reserved", C.c_int32)]


class HipForward:
    def __init__(self, lib_path, model, horizon, num_samples, u_min, u_max, sigmas, seed=42, exploration=0.0):
        vp = C.c_void_p
        lib = self.lib = C.CDLL(lib_path)
        lib.mppi_last_error.restype = C.c_char_p
        lib.mppi_last_error.argtypes = [vp]
        lib.mppi_create.argtypes = [C.POINTER(MppiConfig), C.POINTER(vp)]
        lib.mppi_destroy.argtypes = [vp]
        lib.mppi_bind_state.argtypes = [vp, vp]
        lib.mppi_sample.argtypes = [vp, C.c_uint32, vp]
        lib.mppi_rollout_cost.argtypes = [vp, vp]
        lib.mppi_essps_lambda.argtypes = [vp, C.c_double, C.c_double, C.c_double, vp, vp]
        lib.mppi_weights_reduce.argtypes = [vp, C.c_float, vp, vp]
        lib.mppi_finalize.argtypes = [vp, vp, C.c_int, C.c_float, C.c_int, vp, vp, vp, vp]
        lib.mppi_set_auto_lambda.argtypes = [vp, C.c_int, C.c_double, C.c_double, C.c_double]
        lib.mppi_solve.argtypes = [vp, vp, C.c_uint32, C.c_float, vp, vp, vp, vp]
        model_id, ds, dc = MODELS[model]
        f4 = lambda v: (C.c_float * 4)(*(list(v) + [0.0] * (4 - len(v))))  # noqa: E731
        cfg = MppiConfig(model_id, horizon, ds, dc, num_samples, 0, int(num_samples * (1 - exploration)),
                         f4(u_min), f4(u_max), f4(sigmas), seed, torch.cuda.current_device(), 0)
        self.h = vp()
        self._check(lib.mppi_create(C.byref(cfg), C.byref(self.h)))
        self.T, self.ds, self.dc = horizon, ds, dc
        self.solve_idx = 1  # index 0 is the constructor's draw (mppi.py:146-148)

    def _check(self, rc):
        if rc:
            raise RuntimeError(self.lib.mppi_last_error(self.h).decode())

    def forward(self, state, lam=1.0, essps_target=None, lam_min=0.01, lam_max=10.0, one_call=False):
        """mppi.py:255-336 (sample, clamp, rollout, costs), :341-370 (ESSPS when essps_target is given), :376-385
        (weights, weighted mean), :448-452 (batch-1 rollout, warm start)."""
        lib, h = self.lib, self.h
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        a = torch.empty(self.T, self.dc, device="cuda")
        x = torch.empty(1, self.T + 1, self.ds, device="cuda")
        self._keep = state  # the bound tensor stays alive until the enqueued kernels ran
        if one_call:
            lam_arg = lam
            if essps_target is not None:  # lambda_ = "ESSPS": the rule runs on the device inside mppi_solve (MPPI_AUTO_ESSPS = 1)
                self._check(lib.mppi_set_auto_lambda(h, 1, essps_target, lam_min, lam_max))
                lam_arg = MPPI_LAMBDA_DEVICE
            self._check(lib.mppi_solve(h, C.c_void_p(state.data_ptr()), self.solve_idx, lam_arg, C.c_void_p(a.data_ptr()),
                                       C.c_void_p(x.data_ptr()), None, s))
        else:
            self._check(lib.mppi_bind_state(h, C.c_void_p(state.data_ptr())))
            self._check(lib.mppi_sample(h, self.solve_idx, s))
            self._check(lib.mppi_rollout_cost(h, s))
            if essps_target is not None:
                out = C.c_double(0.0)
                self._check(lib.mppi_essps_lambda(h, essps_target, lam_min, lam_max, C.byref(out), s))
                lam = out.value
            self._check(lib.mppi_weights_reduce(h, lam, None, s))
            self._check(lib.mppi_finalize(h, None, 1, lam, 1, C.c_void_p(a.data_ptr()), C.c_void_p(x.data_ptr()), None, s))
        self.solve_idx += 1
        return a, x

    def close(self):
        if self.h:
            self.lib.mppi_destroy(self.h)
            self.h = C.c_void_p()
