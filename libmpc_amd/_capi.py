"""ctypes binding of the mpcx C ABI (include/mpcx.h).  Product code: loads
libmpc_amd/libmpcx.so and fails loudly when it is missing -- there is no fallback."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MPCX_LIBRARY", os.path.join(_HERE, "libmpcx.so"))   # override: testing aid for build variants

OK = 0
E_INVALID, E_UNSUPPORTED, E_DEVICE, E_NUMERIC, E_STATE = -1, -2, -3, -4, -5
STATUS_SUCCESS, STATUS_MAX_ITERATION, STATUS_INFEASIBLE, STATUS_ERROR, STATUS_UNKNOWN = range(5)
REF_SHARED, REF_PER_INSTANCE, REF_PER_STEP = 0, 1, 2


class Dims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("nx", "nu", "ndu", "ny", "ph", "ch")]


class LParams(C.Structure):
    """POD mirror of mpc::LParameters (reference include/mpc/Types.hpp:99-161)."""
    _fields_ = [("maximum_iteration", C.c_int), ("time_limit", C.c_double), ("enable_warm_start", C.c_int),
                ("alpha", C.c_double), ("rho", C.c_double), ("eps_rel", C.c_double), ("eps_abs", C.c_double),
                ("eps_prim_inf", C.c_double), ("eps_dual_inf", C.c_double),
                ("verbose", C.c_int), ("adaptive_rho", C.c_int), ("polish", C.c_int)]


class Batch(C.Structure):
    _fields_ = [("batch", C.c_int),
                ("x0", C.c_void_p), ("u0", C.c_void_p),
                ("yref", C.c_void_p), ("yref_mode", C.c_int),
                ("uref", C.c_void_p), ("uref_mode", C.c_int),
                ("duref", C.c_void_p), ("duref_mode", C.c_int),
                ("dmeas", C.c_void_p), ("dmeas_mode", C.c_int),
                ("cmd", C.c_void_p), ("cost", C.c_void_p),
                ("status", C.c_void_p), ("solver_status", C.c_void_p), ("is_feasible", C.c_void_p),
                ("iterations", C.c_void_p),
                ("active_lower", C.c_void_p), ("active_upper", C.c_void_p),
                ("seq_state", C.c_void_p), ("seq_output", C.c_void_p), ("seq_input", C.c_void_p),
                ("polish_rounds", C.c_void_p), ("active_count", C.c_void_p),
                ("warm_active_lower", C.c_void_p), ("warm_active_upper", C.c_void_p), ("warm_shift", C.c_int)]


class Info(C.Structure):
    _fields_ = [("n_ref", C.c_int), ("m_ref", C.c_int), ("neq_ref", C.c_int), ("nz", C.c_int), ("mg", C.c_int),
                ("active_words", C.c_int), ("kernel_variant", C.c_int),
                ("flops_setup", C.c_double), ("flops_per_admm_iter", C.c_double),
                ("flops_fixed_per_solve", C.c_double), ("bytes_per_solve", C.c_double)]


EXPORTS = [
    "mpcx_lmpc_create", "mpcx_lmpc_destroy", "mpcx_lparams_default", "mpcx_last_error",
    "mpcx_lmpc_set_state_space_model", "mpcx_lmpc_set_disturbances",
    "mpcx_lmpc_set_objective_weights", "mpcx_lmpc_set_objective_weights_slice",
    "mpcx_lmpc_set_state_bounds", "mpcx_lmpc_set_state_bounds_slice",
    "mpcx_lmpc_set_input_bounds", "mpcx_lmpc_set_input_bounds_slice",
    "mpcx_lmpc_set_output_bounds", "mpcx_lmpc_set_output_bounds_slice",
    "mpcx_lmpc_set_scalar_constraint_slice", "mpcx_lmpc_set_scalar_constraint_index",
    "mpcx_lmpc_set_references", "mpcx_lmpc_set_references_slice",
    "mpcx_lmpc_set_exogenous_inputs", "mpcx_lmpc_set_exogenous_inputs_slice",
    "mpcx_lmpc_set_optimizer_parameters", "mpcx_lmpc_set_strict_infeasibility", "mpcx_lmpc_setup", "mpcx_lmpc_solve_batch",
    "mpcx_lmpc_time_solve_batch", "mpcx_lmpc_solve_host", "mpcx_lmpc_get_info", "mpcx_version",
    "mpcx_lmpc_graph_create", "mpcx_lmpc_graph_launch", "mpcx_lmpc_graph_destroy",
    "mpcx_nlmpc_create", "mpcx_nlmpc_destroy", "mpcx_nlmpc_get_dims", "mpcx_nlmpc_evaluate_batch",
    "mpcx_nlparams_default", "mpcx_nlmpc_set_optimizer_parameters", "mpcx_nlmpc_solve_batch", "mpcx_nlmpc_time_solve_batch", "mpcx_discretize_batch",
    "mpcx_nlmpc_set_state_bounds_slice", "mpcx_nlmpc_set_input_bounds_slice", "mpcx_nlmpc_solve_host",
    "mpcx_nlmpc_create_custom", "mpcx_nlmpc_create_from_source", "mpcx_nlmpc_set_input_scale", "mpcx_nlmpc_set_state_scale",
    "mpcx_comm_get_unique_id", "mpcx_comm_create", "mpcx_comm_destroy", "mpcx_comm_rank", "mpcx_comm_world", "mpcx_allgather_u",
    "mpcx_lmpc_hetero_create", "mpcx_lmpc_hetero_destroy", "mpcx_lmpc_hetero_get_info", "mpcx_lmpc_hetero_solve_batch",
    "mpcx_lmpc_hetero_time_solve_batch", "mpcx_lmpc_hetero_create_ex", "mpcx_lmpc_hetero_debug_get",
    # profiling and testing aids (declared in include/mpcx.h under that heading)
    "mpcx_lmpc_set_total_batch", "mpcx_lmpc_debug_time_kernels", "mpcx_lmpc_debug_get", "mpcx_lmpc_debug_setup_counts", "mpcx_lmpc_debug_use_fused",
    "mpcx_lmpc_debug_force_generic", "mpcx_lmpc_debug_set_rounds", "mpcx_lmpc_debug_set_cycle_buffer",
    "mpcx_nlmpc_debug_set_tolerances", "mpcx_nlmpc_debug_last_form", "mpcx_nlmpc_last_form", "mpcx_nlmpc_debug_get_ws", "mpcx_nlmpc_debug_generated_source", "mpcx_nlmpc_debug_compile_source",
]


class NLParams(C.Structure):
    """mpcx_nlparams == mpc::NLParameters (Types.hpp:99-144)"""
    _fields_ = [("maximum_iteration", C.c_int), ("time_limit", C.c_double), ("enable_warm_start", C.c_int),
                ("relative_ftol", C.c_double), ("relative_xtol", C.c_double), ("absolute_ftol", C.c_double),
                ("absolute_xtol", C.c_double), ("hard_constraints", C.c_int)]


class NlmpcBatch(C.Structure):
    _fields_ = [("batch", C.c_int), ("x0", C.c_void_p), ("u0", C.c_void_p), ("z_warm", C.c_void_p), ("cmd", C.c_void_p),
                ("cost", C.c_void_p), ("status", C.c_void_p), ("solver_status", C.c_void_p), ("is_feasible", C.c_void_p),
                ("iterations", C.c_void_p), ("z", C.c_void_p), ("seq_state", C.c_void_p), ("seq_input", C.c_void_p),
                ("seq_output", C.c_void_p), ("warm_curvature", C.c_int), ("multipliers", C.c_void_p), ("params", C.c_void_p)]


class NlmpcDims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("nx", "nu", "ph", "ch", "nz", "neq", "nineq", "jeq_w", "neq_user", "ny", "nbnd", "n_params")]


class NlmpcSource(C.Structure):
    """mpcx_nlmpc_source: user hooks as C++ text (the bodies of the reference's lambdas), compiled at run time"""
    _fields_ = ([(n, C.c_int) for n in ("nx", "nu", "ny", "ph", "ch", "nineq", "neq_user")] +
                [(n, C.c_char_p) for n in ("preamble", "state_fn", "objective_fn", "ineq_fn", "eq_fn", "output_fn")])

_lib = None


class MpcxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mpcx error {code}: {msg}")
        self.code = code


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  libmpc_amd has no CPU fallback.")
        # PyTorch-ROCm bundles its own libamdhip64.so.7; it must be in the process first so
        # that libmpcx.so binds to the same HIP runtime instead of loading a second one.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = C.CDLL(LIB_PATH)
        _lib.mpcx_last_error.restype = C.c_char_p
        _lib.mpcx_version.restype = C.c_char_p
        _lib.mpcx_lmpc_set_scalar_constraint_slice.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        _lib.mpcx_lmpc_set_scalar_constraint_index.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        _lib.mpcx_nlmpc_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _lib.mpcx_nlmpc_destroy.argtypes = [C.c_void_p]
        _lib.mpcx_lmpc_graph_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.mpcx_lmpc_graph_launch.argtypes = [C.c_void_p, C.c_void_p]
        _lib.mpcx_lmpc_graph_destroy.argtypes = [C.c_void_p]
        _lib.mpcx_nlmpc_get_dims.argtypes = [C.c_void_p, C.c_void_p]
        _lib.mpcx_nlmpc_evaluate_batch.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 9
        _lib.mpcx_nlparams_default.restype = None
        for _n in ("mpcx_nlmpc_set_state_bounds_slice", "mpcx_nlmpc_set_input_bounds_slice"):
            getattr(_lib, _n).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        _lib.mpcx_discretize_batch.argtypes = [C.c_int] * 4 + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3
        _lib.mpcx_nlmpc_set_optimizer_parameters.argtypes = [C.c_void_p, C.c_void_p]
        _lib.mpcx_nlmpc_solve_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.mpcx_nlmpc_time_solve_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _lib.mpcx_nlmpc_create_from_source.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_void_p]
        _lib.mpcx_nlmpc_set_input_scale.argtypes = [C.c_void_p, C.c_void_p]
        _lib.mpcx_nlmpc_set_state_scale.argtypes = [C.c_void_p, C.c_void_p]
        _lib.mpcx_comm_get_unique_id.argtypes = [C.c_void_p]
        _lib.mpcx_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _lib.mpcx_comm_destroy.argtypes = [C.c_void_p]
        _lib.mpcx_comm_rank.argtypes = [C.c_void_p]
        _lib.mpcx_comm_world.argtypes = [C.c_void_p]
        _lib.mpcx_allgather_u.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return _lib


def check(rc):
    if rc < 0:
        raise MpcxError(rc, lib().mpcx_last_error().decode())
    return rc
