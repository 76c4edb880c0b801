"""ctypes binding of libivp_hip.so (the C ABI in include/ivp_hip.h).

The library is built in-tree by ``ivp_amd/csrc/Makefile`` (hipcc, gfx950).  There is no Python or
CPU fallback: if the shared object is missing or no HIP device is present, every compute entry
point raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# IVP_AMD_LIB: an alternative build of the SAME library for A/B measurements (tools/, exp_libs/); never a fallback
LIB_PATH = os.environ.get("IVP_AMD_LIB") or os.path.join(_HERE, "libivp_hip.so")
CSRC = os.path.join(_HERE, "csrc")

c_double_p = C.POINTER(C.c_double)


class ProblemT(C.Structure):
    _fields_ = [("rhs_id", C.c_int32), ("n", C.c_int32), ("n_params", C.c_int32), ("jit", C.c_void_p)]


class OptionsT(C.Structure):
    _fields_ = [
        ("method", C.c_int32),
        ("rtol", C.c_double), ("atol", C.c_double),
        ("rtol_vec", c_double_p), ("atol_vec", c_double_p),
        ("rtol_vec_len", C.c_int32), ("atol_vec_len", C.c_int32),
        ("max_steps", C.c_uint64),
        ("t_eval", c_double_p), ("n_eval", C.c_int64),
        ("has_first_step", C.c_int32), ("first_step", C.c_double),
        ("has_max_step", C.c_int32), ("max_step", C.c_double),
        ("dense_output", C.c_int32),
        ("ev_direction", C.c_int32 * 4), ("ev_terminal", C.c_uint32 * 4), ("max_events", C.c_uint32),
        ("has_min_step", C.c_int32), ("min_step", C.c_double),
        ("fp_mode", C.c_int32), ("chunk_attempts", C.c_int32), ("max_log", C.c_uint32), ("variant", C.c_int32), ("profile", C.c_int32),
        ("has_settings", C.c_int32), ("uround", C.c_double), ("safety_factor", C.c_double), ("scale_min", C.c_double),
        ("scale_max", C.c_double), ("beta", C.c_double), ("stiff_test", C.c_uint64),
        ("count_log", C.c_int32),
        ("t_eval_offsets", C.POINTER(C.c_uint64)), ("ev_direction_vec", C.POINTER(C.c_int32)), ("ev_terminal_vec", C.POINTER(C.c_uint32)),
        ("n_event_cfg", C.c_int32),
    ]


class BatchResultT(C.Structure):
    _fields_ = [(name, C.c_void_p) for name in (
        "y_end", "t_end", "status", "nfev", "nstep", "naccpt", "nrejct", "h_next",
        "y_eval", "eval_idx", "n_filled", "t_log", "y_log", "n_log",
        "seg_cont", "seg_xold", "seg_h", "n_seg", "t_events", "y_events", "n_event_hits", "t_term", "njev", "nlu", "log_offsets")]


class StepLogT(C.Structure):
    """ivp_step_log_t: the CSR accepted-step log of ivp_batch_solve_logged*() (Solution.t / Solution.y of a batch)."""
    _fields_ = [("offsets", C.c_void_p), ("t", C.c_void_p), ("y", C.c_void_p), ("capacity", C.c_uint64), ("reserve", C.c_uint64),
                ("defer", C.c_int32), ("owned", C.c_int32), ("device", C.c_int32), ("passes", C.c_uint32), ("total", C.c_uint64),
                ("pool_bytes", C.c_uint64), ("pool_used_bytes", C.c_uint64), ("page_slots", C.c_uint32)]


class ShardT(C.Structure):
    """ivp_shard_t: trajectories [first, first + count) of a batch, resident on ctx's device (SoA stride count)."""
    _fields_ = [("ctx", C.c_void_p), ("first", C.c_size_t), ("count", C.c_size_t),
                ("y0", C.c_void_p), ("params", C.c_void_p), ("t0", C.c_void_p), ("t0_len", C.c_size_t),
                ("t1", C.c_void_p), ("t1_len", C.c_size_t), ("out", BatchResultT), ("hip_stream", C.c_void_p)]


class RunStatsT(C.Structure):
    _fields_ = [
        ("launches", C.c_uint32), ("init_launches", C.c_uint32),
        ("step_kernel_ms", C.c_double), ("init_kernel_ms", C.c_double), ("total_ms", C.c_double),
        ("total_accepted", C.c_uint64), ("total_attempts", C.c_uint64), ("lane_attempt_slots", C.c_uint64),
        ("lane_launches", C.c_uint64), ("coop_launches", C.c_uint32), ("coop_kernel_ms", C.c_double),
        ("declined_launches", C.c_uint32), ("declined_coop_launches", C.c_uint32), ("declined_ms", C.c_double), ("declined_coop_ms", C.c_double),
    ]


# every symbol include/ivp_hip.h declares
EXPORTS = (
    "ivp_abi_version", "ivp_device_count", "ivp_ctx_create", "ivp_ctx_destroy", "ivp_last_error_string",
    "ivp_ctx_get_stats", "ivp_options_default", "ivp_options_method_defaults", "ivp_rhs_dims", "ivp_rhs_n_events", "ivp_batch_solve",
    "ivp_batch_solve_device", "ivp_batch_submit_device", "ivp_batch_poll", "ivp_batch_wait", "ivp_batch_solve_multi", "ivp_batch_solve_multi_host",
    "ivp_rhs_compile", "ivp_rhs_compile_events", "ivp_rhs_compile_ex", "ivp_rhs_free",
    "ivp_batch_solve_logged", "ivp_batch_solve_logged_device", "ivp_step_log_fetch_device", "ivp_step_log_free", "ivp_batch_solve_logged_multi", "ivp_step_log_fetch_multi",
)

ERRORS = {
    0: "IVP_OK", -1: "IVP_ERR_MUST_BE_POSITIVE", -2: "IVP_ERR_OUT_OF_RANGE", -3: "IVP_ERR_NEGATIVE_TOLERANCE",
    -4: "IVP_ERR_TOLERANCE_SIZE_MISMATCH", -5: "IVP_ERR_INVALID_STEP_SIZE", -6: "IVP_ERR_INVALID_SCALE_FACTORS",
    -100: "IVP_ERR_BAD_ARGUMENT", -101: "IVP_ERR_UNSUPPORTED_METHOD", -102: "IVP_ERR_NO_DEVICE",
    -103: "IVP_ERR_HIP", -104: "IVP_ERR_JIT", -105: "IVP_ERR_LOG_CAPACITY",
}

_lib = None


def build(force: bool = False) -> str:
    """Compile libivp_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, f"-j{min(8, os.cpu_count() or 4)}"]   # 8 translation units, ~2.5 min from scratch on 8 cores
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


def load():
    """Load the shared library; raises if it has not been built (there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `make -C ivp_amd/csrc` (or __graft_entry__.build()). "
            "ivp_amd has no CPU fallback.")
    # PyTorch-ROCm ships its own libamdhip64; two HIP runtimes in one process cannot both own the
    # device.  Loading torch's first lets the dynamic linker resolve our DT_NEEDED libamdhip64.so.N
    # to the copy that is already mapped, so tensors and this library share one runtime.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    L.ivp_abi_version.restype = C.c_int
    L.ivp_device_count.restype = C.c_int
    L.ivp_ctx_create.restype = C.c_int
    L.ivp_ctx_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    L.ivp_ctx_destroy.restype = None
    L.ivp_ctx_destroy.argtypes = [C.c_void_p]
    L.ivp_last_error_string.restype = C.c_char_p
    L.ivp_last_error_string.argtypes = [C.c_void_p]
    L.ivp_ctx_get_stats.restype = C.c_int
    L.ivp_ctx_get_stats.argtypes = [C.c_void_p, C.POINTER(RunStatsT)]
    L.ivp_options_default.restype = None
    L.ivp_options_default.argtypes = [C.POINTER(OptionsT)]
    L.ivp_options_method_defaults.restype = C.c_int
    L.ivp_options_method_defaults.argtypes = [C.POINTER(OptionsT), C.c_int32]
    L.ivp_rhs_dims.restype = C.c_int
    L.ivp_rhs_dims.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    solve_args = [C.c_void_p, C.POINTER(ProblemT), C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                  C.c_void_p, C.c_size_t, C.POINTER(OptionsT), C.POINTER(BatchResultT)]
    L.ivp_batch_solve.restype = C.c_int
    L.ivp_batch_solve.argtypes = solve_args
    L.ivp_batch_solve_device.restype = C.c_int
    L.ivp_batch_solve_device.argtypes = solve_args + [C.c_void_p]
    L.ivp_batch_submit_device.restype = C.c_int
    L.ivp_batch_submit_device.argtypes = solve_args + [C.c_void_p]
    L.ivp_batch_poll.restype = C.c_int
    L.ivp_batch_poll.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.ivp_batch_wait.restype = C.c_int
    L.ivp_batch_wait.argtypes = [C.c_void_p]
    L.ivp_batch_solve_multi.restype = C.c_int
    L.ivp_batch_solve_multi.argtypes = [C.POINTER(ShardT), C.c_int32, C.POINTER(ProblemT), C.c_size_t, C.POINTER(OptionsT),
                                        C.c_int32, C.POINTER(BatchResultT)]
    L.ivp_batch_solve_multi_host.restype = C.c_int
    L.ivp_batch_solve_multi_host.argtypes = [C.POINTER(C.c_void_p), C.c_int32] + solve_args[1:]
    L.ivp_batch_solve_logged.restype = C.c_int
    L.ivp_batch_solve_logged.argtypes = solve_args + [C.POINTER(StepLogT)]
    L.ivp_batch_solve_logged_device.restype = C.c_int
    L.ivp_batch_solve_logged_device.argtypes = solve_args + [C.POINTER(StepLogT), C.c_void_p]
    L.ivp_step_log_fetch_device.restype = C.c_int
    L.ivp_step_log_fetch_device.argtypes = [C.c_void_p, C.POINTER(StepLogT), C.c_void_p]
    L.ivp_step_log_free.restype = None
    L.ivp_step_log_free.argtypes = [C.POINTER(StepLogT)]
    L.ivp_batch_solve_logged_multi.restype = C.c_int
    L.ivp_batch_solve_logged_multi.argtypes = [C.POINTER(ShardT), C.c_int32, C.POINTER(ProblemT), C.c_size_t, C.POINTER(OptionsT),
                                               C.c_int32, C.POINTER(BatchResultT), C.POINTER(StepLogT)]
    L.ivp_step_log_fetch_multi.restype = C.c_int
    L.ivp_step_log_fetch_multi.argtypes = [C.POINTER(ShardT), C.c_int32, C.POINTER(ProblemT), C.c_size_t, C.POINTER(OptionsT), C.c_int32, C.POINTER(StepLogT)]
    L.ivp_rhs_compile.restype = C.c_int
    L.ivp_rhs_compile.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    L.ivp_rhs_compile_events.restype = C.c_int
    L.ivp_rhs_compile_events.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    L.ivp_rhs_compile_ex.restype = C.c_int
    L.ivp_rhs_compile_ex.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.POINTER(C.c_void_p)]
    L.ivp_rhs_n_events.restype = C.c_int
    L.ivp_rhs_n_events.argtypes = [C.c_int32]
    L.ivp_rhs_free.restype = None
    L.ivp_rhs_free.argtypes = [C.c_void_p]
    _lib = L
    return L
