"""ctypes binding of libdpgo_hip.so (include/dpgo_hip.h).

The product path has NO CPU fallback: if the shared library is missing, or a compute
entry point is called without a HIP device, this raises.  Nothing here imports oracle/.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DPGO_LIB") or os.path.join(_HERE, "libdpgo_hip.so")  # DPGO_LIB: kernel A/B builds

OK, ERR_INVALID, ERR_HIP, ERR_UNSUPPORTED, ERR_STATE = 0, 1, 2, 3, 4
METHOD_RTR, METHOD_RGD = 0, 1
PRECOND_NONE, PRECOND_BLOCK_JACOBI, PRECOND_MULTILEVEL, PRECOND_AUTO, PRECOND_ADDITIVE = 0, 1, 2, 3, 4
PRECOND_NAMES = ["none", "jacobi", "multilevel", "auto", "additive"]
ML_P_BLOCKS, ML_A_ROWPTR, ML_A_COLIDX, ML_A_VALUES, ML_DENSE_INVERSE, ML_AGG_LABELS, ML_AP_NNZB = 0, 1, 2, 3, 4, 5, 6
ML_RESTRICT_PARTIALS = 7
TCG_STATUS = ["NEGCURVTURE", "EXCREGION", "LCON", "SCON", "MAXITER"]


class DpgoError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("dpgo_hip error %d: %s" % (code, msg))
        self.code = code


class RoptParamsC(C.Structure):
    _fields_ = [("method", C.c_int), ("verbose", C.c_int), ("gradnorm_tol", C.c_double),
                ("RGD_stepsize", C.c_double), ("RGD_use_preconditioner", C.c_int),
                ("RTR_iterations", C.c_int), ("RTR_tCG_iterations", C.c_int),
                ("RTR_initial_radius", C.c_double), ("precond", C.c_int), ("precond_shift", C.c_double),
                ("accept_tiny_decrease", C.c_int), ("tcg_poll_interval", C.c_int), ("time_bound_s", C.c_double)]


class RoptResultC(C.Structure):
    _fields_ = [("success", C.c_int), ("fInit", C.c_double), ("gradNormInit", C.c_double),
                ("fOpt", C.c_double), ("gradNormOpt", C.c_double), ("elapsedMs", C.c_double),
                ("tCGStatus", C.c_int), ("rtr_iterations", C.c_int), ("rtr_accepted", C.c_int),
                ("tcg_iterations", C.c_int), ("spmm_count", C.c_int), ("latest_step_accepted", C.c_int),
                ("precond_used", C.c_int)]


_P = C.c_void_p
_I = C.c_int
_D = C.c_double
_PI32 = C.POINTER(C.c_int32)
_PD = C.POINTER(C.c_double)

# name -> argtypes ; every symbol declared in include/dpgo_hip.h
SIGNATURES = {
    "dpgo_version": ([], C.c_char_p),
    "dpgo_last_error": ([], C.c_char_p),
    "dpgo_device_count": ([C.POINTER(C.c_int)], _I),
    "dpgo_ropt_params_default": ([C.POINTER(RoptParamsC)], None),
    "dpgo_supported": ([_I, _I], _I),
    "dpgo_describe_options": ([C.c_char_p, _I], _I),
    "dpgo_options_reload": ([], _I),
    "dpgo_problem_describe": ([_P, C.c_char_p, _I], _I),
    "dpgo_problem_create": ([C.POINTER(_P), _I, _I, _I, _I], _I),
    "dpgo_problem_destroy": ([_P], _I),
    "dpgo_problem_set_stream": ([_P, _P], _I),
    "dpgo_problem_use_own_stream": ([_P], _I),
    "dpgo_problem_dims": ([_P, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)], _I),
    "dpgo_problem_set_Q_bsr": ([_P, _I, _P, _P, _P], _I),
    "dpgo_problem_set_Q_csr": ([_P, _P, _P, _P], _I),
    "dpgo_problem_update_Q_values": ([_P, _P], _I),
    "dpgo_problem_set_reweightable_edges": ([_P, _I, _P, _P, _P, _P, _P, _P, _P, _P], _I),
    "dpgo_problem_set_reweightable_edges_ex": ([_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P], _I),
    "dpgo_problem_gnc_reweight_device": ([_P, _P, _P, _D, _D, _D, _I, C.POINTER(C.c_int * 3), C.POINTER(_D)], _I),
    "dpgo_problem_gnc_reweight": ([_P, _P, _D, _D, _D, _I, C.POINTER(C.c_int * 3), C.POINTER(_D)], _I),
    "dpgo_problem_set_edge_weights": ([_P, _P], _I),
    "dpgo_problem_get_edge_weights": ([_P, _P, _P], _I),
    "dpgo_problem_get_Q_values": ([_P, _P], _I),
    "dpgo_multilevel_default_ks": ([_I, _I, _P, C.POINTER(_I)], _I),
    "dpgo_multilevel_graph_aggregates": ([_I, _P, _P, _I, _P, _P, C.POINTER(_I)], _I),
    "dpgo_multilevel_merged_aggregates": ([_I, _P, _P, _I, _I, _P, _P, C.POINTER(_I)], _I),
    "dpgo_problem_additive_plan": ([_P] + [C.POINTER(_I)] * 6, _I),
    "dpgo_problem_setup_multilevel": ([_P, _I, _P, _D, _D], _I),
    "dpgo_problem_multilevel_info": ([_P, C.POINTER(_I), _P, _P, _P], _I),
    "dpgo_problem_multilevel_get": ([_P, _I, _I, _P], _I),
    "dpgo_dense_spd_inverse": ([_I, _P, _P, _I, _I], _I),
    "dpgo_problem_multilevel_path": ([_P, C.POINTER(_I)], _I),
    "dpgo_problem_multilevel_coarse_bits": ([_P, C.POINTER(_I)], _I),
    "dpgo_problem_multilevel_operator_bits": ([_P, C.POINTER(_I), C.POINTER(_I)], _I),
    "dpgo_problem_auto_state": ([_P, C.POINTER(_I)], _I),
    "dpgo_problem_auto_info": ([_P, C.POINTER(_I), C.POINTER(C.c_longlong)] + [C.POINTER(_I)] * 5, _I),
    "dpgo_auto_rule_constants": ([C.POINTER(_I)] * 4, _I),
    "dpgo_problem_set_G": ([_P, _P], _I),
    "dpgo_problem_set_G_device": ([_P, _P], _I),
    "dpgo_problem_set_G_coupling": ([_P, _I, _I, _P, _P, _P, _P], _I),
    "dpgo_problem_update_G_from_neighbors_device": ([_P, _P], _I),
    "dpgo_problem_f": ([_P, _P, C.POINTER(_D)], _I),
    "dpgo_problem_euc_grad": ([_P, _P, _P], _I),
    "dpgo_problem_euc_hess": ([_P, _P, _P], _I),
    "dpgo_problem_rie_grad": ([_P, _P, _P], _I),
    "dpgo_problem_rie_grad_norm": ([_P, _P, C.POINTER(_D)], _I),
    "dpgo_problem_rie_hess": ([_P, _P, _P, _P], _I),
    "dpgo_problem_precondition": ([_P, _I, _D, _P, _P, _P], _I),
    "dpgo_optimize": ([_P, C.POINTER(RoptParamsC), _P, _P, C.POINTER(RoptResultC)], _I),
    "dpgo_optimize_device": ([_P, C.POINTER(RoptParamsC), _P, C.POINTER(RoptResultC)], _I),
    "dpgo_warning_count": ([], _I),
    "dpgo_optimize_device_begin": ([_P, C.POINTER(RoptParamsC), _P, _P], _I),
    "dpgo_optimize_device_end": ([_P, C.POINTER(RoptResultC)], _I),
    "dpgo_optimize_device_many": ([_I, _P, C.POINTER(RoptParamsC), _P, _P, _P, _P], _I),
    "dpgo_problem_eval_terms_device_many": ([_I, _P, _P, _P, _P, _P], _I),
    "dpgo_spmm_device": ([_P, _P, _P, _I], _I),
    "dpgo_problem_eval_device": ([_P, _P, C.POINTER(_D), C.POINTER(_D)], _I),
    "dpgo_problem_eval_terms_device": ([_P, _P, C.POINTER(_D), C.POINTER(_D), C.POINTER(_D)], _I),
    "dpgo_bench_spmm": ([_P, _I, _I, C.POINTER(_D)], _I),
    "dpgo_problem_set_spmm_variant": ([_P, _I, C.POINTER(_I)], _I),
    "dpgo_problem_tcg_kernel_info": ([_P, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)], _I),
    "dpgo_bench_spmm_rotating": ([_P, _I, _I, _I, C.POINTER(_D), C.POINTER(_D)], _I),
    "dpgo_bench_hess_rotating": ([_P, _I, _I, _I, C.POINTER(_D)], _I),
    "dpgo_bench_hess": ([_P, _I, _I, C.POINTER(_D)], _I),
    "dpgo_bench_solve": ([_P, C.POINTER(RoptParamsC), _P, _I, _I, C.POINTER(_D), C.POINTER(_D), C.POINTER(_I)], _I),
    "dpgo_bench_iteration_kernels": ([_P, _I, _I, _P], _I),
    "dpgo_manifold_project": ([_I, _I, _I, _P, _P, _I], _I),
    "dpgo_manifold_tangent_project": ([_I, _I, _I, _P, _P, _P, _I], _I),
    "dpgo_manifold_retract": ([_I, _I, _I, _P, _P, _D, _P, _I], _I),
    "dpgo_manifold_project_device": ([_I, _I, _I, _P, _P, _P], _I),
    "dpgo_round_trajectory": ([_I, _I, _I, _P, _P, _P, _I], _I),
    "dpgo_round_trajectory_device": ([_I, _I, _I, _P, _P, _P, _P], _I),
    "dpgo_gather_tiles_device": ([_I, _I, _P, _P, _I, _P, _P], _I),
    "dpgo_locality_order": ([_I, _P, _P, _I, _I, _P], _I),
    "dpgo_locality_order_runs": ([_I, _P, _P, _I, _I, _I, _P], _I),
    "dpgo_permute_tiles_device": ([_I, _I, _I, _P, _P, _P, _I, _P], _I),
    "dpgo_exchange_plan_create": ([C.POINTER(_P), _I, _I, _I, _P, _P, _P, _P, _I], _I),
    "dpgo_exchange_plan_run": ([_P, _P], _I),
    "dpgo_exchange_plan_destroy": ([_P], _I),
    "dpgo_flags_write_device": ([_I, _P, _P, _P], _I),
    "dpgo_flags_wait_device": ([_I, _P, _P, _I, _P], _I),
    "dpgo_flags_wait_device_checked": ([_I, _P, _P, C.c_longlong, _P, _P], _I),
    "dpgo_max_translation_distance_device": ([_I, _I, _I, _P, _P, _P, C.POINTER(_D), _P], _I),
    "dpgo_axpby_project_device": ([_I, _I, _I, _D, _P, _D, _P, _D, _P, _I, _P, _P], _I),
    "dpgo_problem_set_persistent": ([_P, _I], _I),
    "dpgo_problem_persistent_info": ([_P, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)], _I),
    "dpgo_problem_persistent_phases": ([_P, _P, C.POINTER(_I)], _I),
    "dpgo_chordal_initialization": ([_I, _I, _I, _P, _P, _P, _P, _P, _P, _D, _I, _P, _P, _I], _I),
    "dpgo_odometry_initialization": ([_I, _I, _I, _P, _P, _P, _P, _P], _I),
    "dpgo_device_malloc": ([C.POINTER(_P), C.c_size_t, _I], _I),
    "dpgo_device_free": ([_P], _I),
    "dpgo_device_memcpy": ([_P, _P, C.c_size_t, _I, _P], _I),
    "dpgo_device_synchronize": ([_P], _I),
    "dpgo_comm_unique_id": ([_P], _I),
    "dpgo_comm_create": ([C.POINTER(_P), _I, _I, _P, _I], _I),
    "dpgo_comm_destroy": ([_P], _I),
    "dpgo_comm_info": ([_P, C.POINTER(_I), C.POINTER(_I)], _I),
    "dpgo_comm_exchange": ([_P, _I, _P, _P, _P, _I, _P, _P, _P, _P], _I),
    "dpgo_comm_allreduce": ([_P, _P, _I, _I, _P], _I),
    "dpgo_comm_broadcast": ([_P, _P, _I, _I, _P], _I),
    "dpgo_build_Q_bsr": ([_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _D, _D,
                          C.POINTER(_I), _P, _P, _P], _I),
    "dpgo_build_G_coupling": ([_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                               C.POINTER(_I), _P, _P, _P], _I),
    "dpgo_debug_reduction_primitives": ([_I, _I, _I, _P, _P, _P, _P, _P], _I),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libdpgo_hip.so (built in-tree by __graft_entry__.build() / `make -C dpgo_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "dpgo_amd: %s not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C dpgo_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64; if this library pulled in
    # the system runtime first, torch would later fail with "No HIP GPUs are available".  Import torch
    # first when it is installed so both bind to the same runtime (torch is plumbing here: device
    # memory, streams, torch.distributed).
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(LIB_PATH)
    for name, (argtypes, restype) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.argtypes = argtypes
        fn.restype = restype
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != OK:
        raise DpgoError(rc, load().dpgo_last_error().decode("utf-8", "replace"))


def ptr(a) -> Optional[int]:
    """Pointer of a numpy array (host), a torch tensor (device) or an int address."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    raise TypeError("cannot take the address of %r" % type(a))


def describe_options() -> str:
    """The library's DPGO_* switches as it read them (dpgo_describe_options)."""
    buf = C.create_string_buffer(16384)
    check(load().dpgo_describe_options(buf, len(buf)))
    return buf.value.decode("utf-8", "replace")


def device_count() -> int:
    c = C.c_int(0)
    rc = load().dpgo_device_count(C.byref(c))
    return c.value if rc == OK else 0


def f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)
