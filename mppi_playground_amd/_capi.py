"""ctypes binding of include/mppi_hip.h (libmppi_hip.so).  Thin: no compute happens here.

The library is required: there is no CPU or PyTorch fallback for the hot path.  Importing this module
without the built extension, or creating a solver without a visible MI355X, raises immediately.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MPPI_HIP_LIB selects an alternative build of the same library (compiler-flag experiments)
LIB_PATH = os.environ.get("MPPI_HIP_LIB") or os.path.join(_HERE, "csrc", "libmppi_hip.so")

MODEL_GENERIC = -1
MAX_DIM_CONTROL = 4           # controls held by MppiConfig
MAX_DIM_CONTROL_GENERIC = 64  # opaque callables: any dim_control up to this
MODEL_IDS = {"pendulum": 0, "cartpole": 1, "mountaincar": 2, "nav2d": 3, "racing": 4, "mjcartpole": 5, "goalzone": 6}
MODEL_DIMS = {"pendulum": (2, 1), "cartpole": (4, 1), "mountaincar": (2, 1), "nav2d": (3, 2), "racing": (4, 2),
              "mjcartpole": (4, 1), "goalzone": (7, 2)}
SUMMARY_HEAD = 4
LAMBDA_DEVICE = -1.0  # MPPI_LAMBDA_DEVICE: "the temperature a device-resident rule left on the device"
AUTO_RULES = {None: 0, "ESSPS": 1, "LBPS": 2, "MPO": 3}  # MPPI_AUTO_*

# every symbol include/mppi_hip.h declares
ABI_VERSION = 10  # MPPI_ABI_VERSION of the header this binding was written against

SYMBOLS = [
    "mppi_version", "mppi_abi_version", "mppi_device_count", "mppi_last_error", "mppi_create", "mppi_destroy", "mppi_set_control_limits",
    "mppi_set_model_params", "mppi_upload_map", "mppi_build_obstacle_map", "mppi_build_lane_map",
    "mppi_download_map", "mppi_set_reference", "mppi_set_mean", "mppi_get_mean",
    "mppi_set_state", "mppi_bind_state", "mppi_sample", "mppi_inject_noise", "mppi_export_noise", "mppi_rollout_cost",
    "mppi_get_costs", "mppi_set_costs", "mppi_weights_reduce", "mppi_finalize", "mppi_solve", "mppi_set_sg_filter", "mppi_get_sg_history", "mppi_softmax_stats", "mppi_softmax_stats_multi", "mppi_essps_lambda", "mppi_essps_lambda_device", "mppi_get_lambda", "mppi_lbps_lambda", "mppi_mpo_reset", "mppi_mpo_step", "mppi_mpo_state", "mppi_weights", "mppi_sample_posterior",
    "mppi_p2p_alloc", "mppi_p2p_connect", "mppi_p2p_exchange", "mppi_p2p_error", "mppi_rollout_actions", "mppi_rollout_samples", "mppi_top_samples", "mppi_top_candidates", "mppi_rollout_candidates", "mppi_set_option", "mppi_get_timing",
    "mppi_set_center_path", "mppi_ref_window", "mppi_set_path_index", "mppi_get_path_index", "mppi_get_reference",
    "mppi_model_step", "mppi_comm_unique_id", "mppi_comm_init", "mppi_comm_exchange", "mppi_comm_destroy",
    "mppi_set_auto_lambda", "mppi_lbps_lambda_device", "mppi_lbps_brent_device", "mppi_search_error", "mppi_clone_state", "mppi_mpo_set_state", "mppi_mpo_step_device", "mppi_fused_error",
    "mppi_search_passes", "mppi_grid_lookup", "mppi_mpo_log_temperature_ptr", "mppi_join_state_seq", "mppi_state_seq_serial", "mppi_get_state_seq_timing", "mppi_comm_info",
]


class MppiConfig(C.Structure):
    _fields_ = [
        ("model", C.c_int32), ("horizon", C.c_int32), ("dim_state", C.c_int32), ("dim_control", C.c_int32),
        ("num_samples", C.c_int64), ("sample_offset", C.c_int64), ("inherit_count", C.c_int64),
        ("u_min", C.c_float * 4), ("u_max", C.c_float * 4), ("sigmas", C.c_float * 4),
        ("seed", C.c_uint64), ("device", C.c_int32), ("reserved", C.c_int32),
    ]


class MppiError(RuntimeError):
    pass


_lib = None


def load():
    """dlopen the in-tree extension; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MppiError(
            f"HIP extension missing: {LIB_PATH}. Build it with `python -m mppi_playground_amd._build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the MPPI hot path.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u32, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_float
    try:
        got = int(lib.mppi_abi_version())
    except AttributeError:
        got = None
    if got != ABI_VERSION:
        raise MppiError(f"{LIB_PATH} exports ABI version {got}, this binding needs {ABI_VERSION}: rebuild the extension "
                        "(python -m mppi_playground_amd._build)")
    lib.mppi_version.restype = C.c_char_p
    lib.mppi_last_error.restype = C.c_char_p
    lib.mppi_last_error.argtypes = [vp]
    lib.mppi_create.argtypes = [C.POINTER(MppiConfig), C.POINTER(vp)]
    lib.mppi_destroy.argtypes = [vp]
    lib.mppi_set_model_params.argtypes = [vp, vp, i32]
    lib.mppi_upload_map.argtypes = [vp, i32, vp, i32, i32, f32, f32, f32]
    lib.mppi_build_obstacle_map.argtypes = [vp, i32, i32, i32, f32, f32, f32, vp, i32, vp, i32, vp]
    lib.mppi_build_lane_map.argtypes = [vp, i32, i32, i32, f32, f32, f32, vp, i32, C.c_int64, vp]
    lib.mppi_download_map.argtypes = [vp, i32, vp, vp, vp]
    lib.mppi_p2p_alloc.argtypes = [vp, i32, i32, vp]
    lib.mppi_p2p_connect.argtypes = [vp, vp, vp]
    lib.mppi_p2p_exchange.argtypes = [vp, vp, vp, vp]
    lib.mppi_p2p_error.argtypes = [vp]
    lib.mppi_set_sg_filter.argtypes = [vp, vp, i32, vp]
    lib.mppi_get_sg_history.argtypes = [vp, vp]
    lib.mppi_top_samples.argtypes = [vp, i32, f32, vp, vp, vp]
    lib.mppi_top_candidates.argtypes = [vp, i32, vp, vp]
    lib.mppi_rollout_candidates.argtypes = [vp, vp, i32, f32, vp, vp, vp]
    lib.mppi_essps_lambda.argtypes = [vp, C.c_double, C.c_double, C.c_double, vp, vp]
    lib.mppi_lbps_lambda.argtypes = [vp, C.c_double, C.c_double, C.c_double, vp, vp]
    lib.mppi_essps_lambda_device.argtypes = [vp, C.c_double, C.c_double, C.c_double, vp]
    lib.mppi_get_lambda.argtypes = [vp, vp, vp, vp]
    lib.mppi_set_auto_lambda.argtypes = [vp, i32, C.c_double, C.c_double, C.c_double]
    lib.mppi_lbps_lambda_device.argtypes = [vp, C.c_double, C.c_double, C.c_double, vp]
    lib.mppi_lbps_brent_device.argtypes = [vp, C.c_double, C.c_double, C.c_double, vp]
    lib.mppi_search_error.argtypes = [vp]
    lib.mppi_clone_state.argtypes = [vp, vp]
    lib.mppi_mpo_set_state.argtypes = [vp, vp]
    lib.mppi_mpo_step_device.argtypes = [vp, vp]
    lib.mppi_fused_error.argtypes = [vp]
    lib.mppi_search_passes.argtypes = [vp, vp]
    lib.mppi_comm_info.argtypes = [vp, vp, vp]
    lib.mppi_join_state_seq.argtypes = [vp, u32, vp]
    lib.mppi_state_seq_serial.argtypes = [vp, vp, vp]
    lib.mppi_get_state_seq_timing.argtypes = [vp, vp]
    lib.mppi_grid_lookup.argtypes = [vp, i32, i32, f32, f32, f32, vp, C.c_int64, C.c_int64, vp, vp]
    lib.mppi_mpo_reset.argtypes = [vp, C.c_double, C.c_double, C.c_double]
    lib.mppi_mpo_step.argtypes = [vp, vp, vp]
    lib.mppi_mpo_state.argtypes = [vp, vp]
    lib.mppi_mpo_log_temperature_ptr.argtypes = [vp, vp]
    lib.mppi_set_control_limits.argtypes = [vp, vp, vp, vp, i32]
    lib.mppi_sample_posterior.argtypes = [vp, u32, vp, i32, vp, vp]
    lib.mppi_set_reference.argtypes = [vp, vp, i32, vp]
    lib.mppi_set_mean.argtypes = [vp, vp, i32, vp]
    lib.mppi_get_mean.argtypes = [vp, vp, i32, vp]
    lib.mppi_set_state.argtypes = [vp, vp, i32, vp]
    lib.mppi_bind_state.argtypes = [vp, vp]
    lib.mppi_sample.argtypes = [vp, u32, vp]
    lib.mppi_inject_noise.argtypes = [vp, vp, vp]
    lib.mppi_export_noise.argtypes = [vp, vp, vp, vp]
    lib.mppi_rollout_cost.argtypes = [vp, vp]
    lib.mppi_get_costs.argtypes = [vp, vp, i32, vp]
    lib.mppi_set_costs.argtypes = [vp, vp, i32, vp]
    lib.mppi_weights_reduce.argtypes = [vp, f32, vp, vp]
    lib.mppi_finalize.argtypes = [vp, vp, i32, f32, i32, vp, vp, vp, vp]
    lib.mppi_solve.argtypes = [vp, vp, u32, f32, vp, vp, vp, vp]
    lib.mppi_softmax_stats.argtypes = [vp, f32, vp, vp]
    lib.mppi_softmax_stats_multi.argtypes = [vp, vp, i32, vp, vp]
    lib.mppi_weights.argtypes = [vp, f32, f32, f32, vp, vp]
    lib.mppi_rollout_actions.argtypes = [vp, vp, i32, vp, vp, vp]
    lib.mppi_rollout_samples.argtypes = [vp, vp, i32, vp, vp]
    lib.mppi_set_center_path.argtypes = [vp, vp, i32, vp, i32, f32]
    lib.mppi_ref_window.argtypes = [vp, vp, vp]
    lib.mppi_set_path_index.argtypes = [vp, C.c_int32, vp]
    lib.mppi_get_path_index.argtypes = [vp, vp, vp]
    lib.mppi_get_reference.argtypes = [vp, vp, i32, i32, vp]
    lib.mppi_model_step.argtypes = [i32, vp, i32, vp, vp, vp, vp, vp, vp, f32, vp, vp]
    lib.mppi_comm_unique_id.argtypes = [vp]
    lib.mppi_comm_init.argtypes = [vp, i32, i32, vp]
    lib.mppi_comm_exchange.argtypes = [vp, vp, vp, vp]
    lib.mppi_comm_destroy.argtypes = [vp]
    lib.mppi_set_option.argtypes = [vp, C.c_char_p, i64]
    lib.mppi_get_timing.argtypes = [vp, vp]
    for name in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the .so does not export what the header declares
        if name not in ("mppi_version", "mppi_last_error"):
            fn.restype = C.c_int
    _lib = lib
    return lib


# Handles whose owner was dropped while a stream of this process was being captured into a graph (torch.cuda.graph): freeing
# device memory there is not permitted (hipFree inside a capture invalidates it — a garbage collection that happened to run
# inside another solver's capture took the process down), so the destruction waits for the next handle operation outside a capture.
_deferred_destroy = []


def _capturing() -> bool:
    try:
        import torch

        return bool(torch.cuda.is_available() and torch._C._cuda_isCurrentStreamCapturing())
    except Exception:  # noqa: BLE001  (interpreter shutdown, a torch build without the query)
        return False


def _flush_deferred() -> None:
    if _deferred_destroy and not _capturing():
        lib = _lib
        while _deferred_destroy:
            h = _deferred_destroy.pop()
            if lib is not None:
                lib.mppi_destroy(h)


class Handle:
    """RAII wrapper of mppi_handle_t; every call checks the return code and raises MppiError."""

    def __init__(self, cfg: MppiConfig):
        self.lib = load()
        if self.lib.mppi_device_count() <= 0:
            raise MppiError("no HIP device visible: the MPPI hot path needs an MI355X (gfx950); "
                            "there is no CPU fallback")
        _flush_deferred()
        self.h = C.c_void_p()
        rc = self.lib.mppi_create(C.byref(cfg), C.byref(self.h))
        if rc != 0:
            msg = self.lib.mppi_last_error(self.h).decode() if self.h else f"code {rc}"
            if self.h:
                self.lib.mppi_destroy(self.h)
                self.h = C.c_void_p()
            raise MppiError(f"mppi_create failed ({rc}): {msg}")

    def call(self, name, *args):
        rc = getattr(self.lib, name)(self.h, *args)
        if rc != 0:
            raise MppiError(f"{name} failed ({rc}): {self.lib.mppi_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None):
            h, self.h = self.h, C.c_void_p()
            if _capturing():  # (dropped — typically by the garbage collector — inside somebody's graph capture: not now)
                _deferred_destroy.append(h)
                return
            self.lib.mppi_destroy(h)
            _flush_deferred()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
