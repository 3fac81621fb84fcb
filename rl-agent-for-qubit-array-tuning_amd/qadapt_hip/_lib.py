"""ctypes binding of libqdsim.so (include/qdsim.h).  There is NO fallback: if the
library is missing or a call fails, an exception is raised."""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
LIB_PATH = os.environ.get("QDSIM_LIB", os.path.join(CSRC, "libqdsim.so"))   # QDSIM_LIB: diagnostic builds

QD_FLAG_VALIDATE = 1
QD_FLAG_PIXEL_SEARCH = 2
QD_NOISE_SENSOR = 1
QD_NOISE_RADIAL = 2
QD_NOISE_LATCH = 4

EXPORTS = [
    "qd_param_block_doubles", "qd_state_block_doubles", "qd_layout_query", "qd_create", "qd_destroy",
    "qd_last_error", "qd_bind_outputs", "qd_load_episodes", "qd_apply_actions", "qd_observe",
    "qd_update_capacitance", "qd_step", "qd_get_state", "qd_set_state", "qd_get_raw",
    "qd_get_occupations", "qd_get_candidates", "qd_get_eigen", "qd_get_search_stats", "qd_get_solver_stats", "qd_get_rng_state", "qd_set_rng_state",
    "qd_time_ground_kernel", "qd_time_candidates_kernel", "qd_time_kernels", "qd_timed_kernel_name", "qd_chunk_envs",
]

QD_CURVES = {"constant": 0, "polynomial": 1, "exponential": 2, "linear": 3}
QD_UPDATE_KALMAN, QD_UPDATE_DIRECT = 0, 1


class QdConfig(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_int32), ("n_dot", ctypes.c_int32), ("resolution", ctypes.c_int32),
        ("batch", ctypes.c_int32), ("max_steps", ctypes.c_int32), ("env_chunk", ctypes.c_int32),
        ("flags", ctypes.c_int32), ("noise_flags", ctypes.c_int32),
        ("gate_ramp_start", ctypes.c_double), ("gate_quadratic_start", ctypes.c_double),
        ("barrier_ramp_start", ctypes.c_double), ("kalman_prior_mean", ctypes.c_double),
        ("kalman_prior_variance", ctypes.c_double), ("kalman_prior_mean_nnn", ctypes.c_double),
        ("kalman_variance_threshold", ctypes.c_double), ("kalman_process_noise", ctypes.c_double),
        ("rng_seed", ctypes.c_uint64), ("env_id_offset", ctypes.c_int64),
        ("use_deltas", ctypes.c_int32), ("sparse_reward", ctypes.c_int32), ("gate_curve_type", ctypes.c_int32),
        ("update_method", ctypes.c_int32), ("cnn_outputs", ctypes.c_int32), ("reserved0", ctypes.c_int32),
        ("delta_max", ctypes.c_double), ("gate_curve_exponent", ctypes.c_double),
        ("plunger_radius", ctypes.c_double), ("outer_plunger_radius", ctypes.c_double),
        ("outer_plunger_reward_max", ctypes.c_double), ("barrier_radius", ctypes.c_double),
    ]


class QdError(RuntimeError):
    pass


def build(force=False):
    """Compile libqdsim.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.check_call(["make", "-s", "-C", CSRC, "libqdsim.so"])
    return LIB_PATH


_LIB = None


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise QdError(
            f"{LIB_PATH} not found: build it with `make -C {CSRC}` (or __graft_entry__.build()). "
            "qadapt_hip has no CPU fallback.")
    # libqdsim.so needs libamdhip64.so.7.  PyTorch-ROCm ships its own copy with that
    # soname; loading torch first makes the dynamic loader reuse it, so device
    # pointers and streams are shared with torch instead of living in a second runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, ip, fp, dp = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32), ctypes.c_void_p, ctypes.c_void_p
    L.qd_param_block_doubles.argtypes = [ctypes.c_int]; L.qd_param_block_doubles.restype = ctypes.c_int
    L.qd_state_block_doubles.argtypes = [ctypes.c_int]; L.qd_state_block_doubles.restype = ctypes.c_int
    L.qd_layout_query.argtypes = [ctypes.c_int, ip]; L.qd_layout_query.restype = ctypes.c_int
    L.qd_create.argtypes = [ctypes.POINTER(QdConfig), ctypes.c_int, ctypes.POINTER(vp)]
    L.qd_create.restype = ctypes.c_int
    L.qd_destroy.argtypes = [vp]; L.qd_destroy.restype = ctypes.c_int
    L.qd_last_error.argtypes = [vp]; L.qd_last_error.restype = ctypes.c_char_p
    L.qd_bind_outputs.argtypes = [vp, fp, fp, fp, fp]; L.qd_bind_outputs.restype = ctypes.c_int
    L.qd_load_episodes.argtypes = [vp, ip, ctypes.c_int, dp, dp, ctypes.c_int, vp]
    L.qd_load_episodes.restype = ctypes.c_int
    L.qd_apply_actions.argtypes = [vp, fp, dp, vp, vp]; L.qd_apply_actions.restype = ctypes.c_int
    L.qd_observe.argtypes = [vp, vp, ctypes.c_int, vp]; L.qd_observe.restype = ctypes.c_int
    L.qd_update_capacitance.argtypes = [vp, vp, ctypes.c_int, fp, fp, ctypes.c_int, vp]
    L.qd_update_capacitance.restype = ctypes.c_int
    L.qd_step.argtypes = [vp, fp, fp, fp, dp, vp, vp]; L.qd_step.restype = ctypes.c_int
    L.qd_get_state.argtypes = [vp, dp, vp]; L.qd_get_state.restype = ctypes.c_int
    L.qd_set_state.argtypes = [vp, dp, vp]; L.qd_set_state.restype = ctypes.c_int
    L.qd_get_raw.argtypes = [vp, dp, dp]; L.qd_get_raw.restype = ctypes.c_int
    L.qd_get_occupations.argtypes = [vp, dp]; L.qd_get_occupations.restype = ctypes.c_int
    L.qd_get_candidates.argtypes = [vp, vp]; L.qd_get_candidates.restype = ctypes.c_int
    L.qd_get_eigen.argtypes = [vp, dp]; L.qd_get_eigen.restype = ctypes.c_int
    L.qd_get_search_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]; L.qd_get_search_stats.restype = ctypes.c_int
    L.qd_get_solver_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]; L.qd_get_solver_stats.restype = ctypes.c_int
    L.qd_get_rng_state.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]; L.qd_get_rng_state.restype = ctypes.c_int
    L.qd_set_rng_state.argtypes = [vp, ctypes.c_uint64]; L.qd_set_rng_state.restype = ctypes.c_int
    L.qd_time_ground_kernel.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float), vp]
    L.qd_time_ground_kernel.restype = ctypes.c_int
    L.qd_time_candidates_kernel.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float), vp]
    L.qd_time_candidates_kernel.restype = ctypes.c_int
    L.qd_time_kernels.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float), vp]; L.qd_time_kernels.restype = ctypes.c_int
    L.qd_timed_kernel_name.argtypes = [ctypes.c_int]; L.qd_timed_kernel_name.restype = ctypes.c_char_p
    L.qd_chunk_envs.argtypes = [vp]; L.qd_chunk_envs.restype = ctypes.c_int
    _LIB = L
    return L


def check(handle, rc, what):
    if rc != 0:
        msg = lib().qd_last_error(handle).decode() if handle else "no handle"
        raise QdError(f"{what} failed (code {rc}): {msg}")
