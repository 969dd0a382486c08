"""ctypes binding of piqp_amd/lib/libpiqp_amd.so (the C-ABI in include/piqp_amd.h).

There is no CPU fallback: if the HIP library is missing or no device is visible, every
constructor raises.  Nothing in this package imports the oracle.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpiqp_amd.so")
LIB_PATH = os.environ.get("PIQP_AMD_LIB", LIB_PATH)  # debugging aid: another build of the SAME library (tools/exp_ref_arith.py compares arithmetic variants)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
vp = C.c_void_p

VAR_NAMES = ("x", "y", "z_l", "z_u", "z_bl", "z_bu", "s_l", "s_u", "s_bl", "s_bu")


class DenseData(C.Structure):
    _fields_ = [("n", C.c_int), ("p", C.c_int), ("m", C.c_int),
                ("P_utri", vp), ("AT", vp), ("GT", vp),
                ("n_h_l", C.c_int), ("n_h_u", C.c_int), ("n_x_l", C.c_int), ("n_x_u", C.c_int),
                ("h_l_idx", vp), ("h_u_idx", vp), ("x_l_idx", vp), ("x_u_idx", vp),
                ("x_b_scaling", vp), ("mem", C.c_int)]


class SparseData(C.Structure):
    _fields_ = [("n", C.c_int), ("p", C.c_int), ("m", C.c_int),
                ("P_colptr", vp), ("P_rowind", vp), ("P_val", vp),
                ("AT_colptr", vp), ("AT_rowind", vp), ("AT_val", vp),
                ("GT_colptr", vp), ("GT_rowind", vp), ("GT_val", vp),
                ("n_h_l", C.c_int), ("n_h_u", C.c_int), ("n_x_l", C.c_int), ("n_x_u", C.c_int),
                ("h_l_idx", vp), ("h_u_idx", vp), ("x_l_idx", vp), ("x_u_idx", vp),
                ("x_b_scaling", vp), ("mem", C.c_int)]


class Vars(C.Structure):
    _fields_ = [(k, vp) for k in VAR_NAMES]


class Settings(C.Structure):
    _fields_ = [("rho_init", C.c_double), ("delta_init", C.c_double), ("eps_abs", C.c_double), ("eps_rel", C.c_double),
                ("check_duality_gap", C.c_int), ("eps_duality_gap_abs", C.c_double), ("eps_duality_gap_rel", C.c_double),
                ("infeasibility_threshold", C.c_double), ("reg_lower_limit", C.c_double),
                ("reg_finetune_lower_limit", C.c_double), ("reg_finetune_primal_update_threshold", C.c_int),
                ("reg_finetune_dual_update_threshold", C.c_int), ("max_iter", C.c_int), ("max_factor_retires", C.c_int),
                ("preconditioner_scale_cost", C.c_int), ("preconditioner_reuse_on_update", C.c_int),
                ("preconditioner_iter", C.c_int), ("tau", C.c_double), ("kkt_solver", C.c_int),
                ("iterative_refinement_always_enabled", C.c_int), ("iterative_refinement_eps_abs", C.c_double),
                ("iterative_refinement_eps_rel", C.c_double), ("iterative_refinement_max_iter", C.c_int),
                ("iterative_refinement_min_improvement_rate", C.c_double),
                ("iterative_refinement_static_regularization_eps", C.c_double),
                ("iterative_refinement_static_regularization_rel", C.c_double), ("verbose", C.c_int),
                ("compute_timings", C.c_int)]


class Info(C.Structure):
    _fields_ = [("status", C.c_int), ("iter", C.c_int)] + [(k, C.c_double) for k in (
        "rho", "delta", "mu", "sigma", "primal_step", "dual_step", "primal_res", "primal_res_rel", "dual_res",
        "dual_res_rel", "primal_res_reg", "primal_res_reg_rel", "dual_res_reg", "dual_res_reg_rel", "primal_prox_inf",
        "dual_prox_inf", "prev_primal_res", "prev_dual_res", "primal_obj", "dual_obj", "duality_gap",
        "duality_gap_rel")] + [("factor_retires", C.c_int), ("reg_limit", C.c_double), ("no_primal_update", C.c_int),
                               ("no_dual_update", C.c_int)] + [(k, C.c_double) for k in (
        "setup_time", "update_time", "solve_time", "kkt_factor_time", "kkt_solve_time", "run_time")] + [
        ("n_factor", C.c_int), ("n_solve", C.c_int), ("n_backend_solve", C.c_int)]


# every symbol include/piqp_amd.h declares (checked by tests/test_abi.py against the header text)
SYMBOLS = [
    "pq_settings_default", "pq_last_error_string", "pq_device_count", "pq_version",
    "pq_kkt_create_dense", "pq_kkt_create_sparse", "pq_kkt_clone", "pq_kkt_destroy", "pq_kkt_set_pointer_mode",
    "pq_kkt_update_data_dense", "pq_kkt_update_data_sparse", "pq_kkt_update_scalings_and_factor", "pq_kkt_solve",
    "pq_kkt_eval_P_x", "pq_kkt_eval_A_xn_and_AT_xt", "pq_kkt_eval_G_xn_and_GT_xt", "pq_kkt_print_info",
    "pq_kkt_synchronize", "pq_kkt_stream", "pq_kkt_internal_kkt_mat", "pq_kkt_internal_factor", "pq_kkt_dims", "pq_kkt_multistage_block_info", "pq_kkt_set_profiling", "pq_kkt_get_profile",
    "pq_kkt_sparse_stats", "pq_kkt_partition", "pq_kkt_set_exchange", "pq_kkt_partition_info", "pq_sparse_partition_plan",
    "pq_kktsys_create_dense", "pq_kktsys_create_sparse", "pq_kktsys_clone", "pq_kktsys_destroy",
    "pq_kktsys_set_pointer_mode", "pq_kktsys_backend", "pq_kktsys_update_data_dense", "pq_kktsys_update_data_sparse",
    "pq_kktsys_update_scalings_and_factor", "pq_kktsys_solve", "pq_kktsys_mul", "pq_kktsys_last_solve_stats",
    "pq_kktsys_condensed_residual", "pq_kktsys_synchronize",
    "pq_solver_create", "pq_solver_destroy", "pq_solver_clone", "pq_solver_settings", "pq_solver_setup_dense",
    "pq_solver_setup_sparse", "pq_solver_update_dense", "pq_solver_update_sparse", "pq_solver_solve", "pq_solver_info",
    "pq_solver_get_result", "pq_solver_dims", "pq_solver_set_trace", "pq_solver_trace_rows", "pq_solver_partition", "pq_solver_set_exchange",
    "pq_batch_create", "pq_batch_destroy", "pq_batch_settings", "pq_batch_setup_sparse", "pq_batch_update", "pq_batch_update_data", "pq_batch_solve", "pq_batch_info", "pq_batch_get_result",
    "pq_batch_dims", "pq_batch_block_info", "pq_batch_get_profile", "pq_batch_last_kernel_ms", "pq_batch_set_start_order",
    "pq_dense_factor_create", "pq_dense_factor_destroy", "pq_dense_factor_compute", "pq_dense_factor_info", "pq_dense_factor_solve_in_place", "pq_dense_factor_matrix", "pq_dense_factor_last_ms",
    "pq_debug_alloc_count", "pq_debug_chol_plan", "pq_kkt_set_exchange_norm", "pq_kkt_sharded_calls", "pq_kkt_sharded_solve_calls", "pq_solver_sharded_solve_calls", "pq_solver_set_exchange_norm", "pq_solver_sharded_calls", "pq_microbench_mfma_f64", "pq_microbench_hbm_copy", "pq_microbench_potrf_block", "pq_debug_potrf_block", "pq_rccl_unique_id", "pq_kkt_set_comm_rccl", "pq_solver_set_comm_rccl", "pq_kkt_native_exchange_calls", "pq_kkt_min_abs_pivot", "pq_solver_native_exchange_calls",
    "pq_sparse_amd_order", "pq_sparse_permute_sym_upper", "pq_sparse_kkt_symbolic", "pq_kkt_sparse_ordering", "pq_kkt_comm_info", "pq_solver_comm_info", "pq_kkt_exact_factor", "pq_sparse_uplooking_plan",
]

EXCHANGE_FN = C.CFUNCTYPE(C.c_int, vp, C.c_int)  # pq_exchange_fn

_lib = None


def load():
    """dlopen the HIP library; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # one HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64/libhsa-runtime; if it is going to be
        # used at all (device tensors, torch.distributed) it must be loaded before this library resolves libamdhip64.
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc, gfx950). piqp_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.pq_settings_default.argtypes = [C.POINTER(Settings)]
    L.pq_last_error_string.restype = C.c_char_p
    L.pq_version.restype = C.c_char_p
    L.pq_device_count.restype = C.c_int
    L.pq_kkt_create_dense.argtypes = [C.POINTER(vp), C.POINTER(DenseData), C.c_int, C.c_int]
    L.pq_kkt_create_sparse.argtypes = [C.POINTER(vp), C.POINTER(SparseData), C.c_int, C.c_int]
    L.pq_kkt_clone.argtypes = [vp, C.POINTER(vp)]
    L.pq_kkt_destroy.argtypes = [vp]
    L.pq_kkt_destroy.restype = None
    L.pq_kkt_set_pointer_mode.argtypes = [vp, C.c_int]
    L.pq_kkt_update_data_dense.argtypes = [vp, C.POINTER(DenseData), C.c_int]
    L.pq_kkt_update_data_sparse.argtypes = [vp, C.POINTER(SparseData), C.c_int]
    L.pq_kkt_update_scalings_and_factor.argtypes = [vp, C.c_double, vp, vp]
    L.pq_kkt_solve.argtypes = [vp] + [vp] * 6
    L.pq_kkt_eval_P_x.argtypes = [vp, C.c_double, vp, vp]
    L.pq_kkt_eval_A_xn_and_AT_xt.argtypes = [vp, C.c_double, C.c_double, vp, vp, vp, vp]
    L.pq_kkt_eval_G_xn_and_GT_xt.argtypes = [vp, C.c_double, C.c_double, vp, vp, vp, vp]
    L.pq_kkt_print_info.argtypes = [vp]
    L.pq_kkt_synchronize.argtypes = [vp]
    L.pq_kkt_stream.argtypes = [vp]
    L.pq_kkt_stream.restype = vp
    L.pq_kkt_internal_kkt_mat.argtypes = [vp, vp]
    L.pq_kkt_internal_factor.argtypes = [vp, vp]
    L.pq_kkt_dims.argtypes = [vp, _ip, _ip, _ip]
    L.pq_kkt_multistage_block_info.argtypes = [vp, vp, C.c_int]
    L.pq_sparse_amd_order.argtypes = [C.c_int, _ip, _ip, _ip]
    L.pq_sparse_permute_sym_upper.argtypes = [C.c_int, _ip, _ip, _ip, _ip, _ip, _ip]
    L.pq_sparse_kkt_symbolic.argtypes = [C.POINTER(SparseData), C.c_int, _ip, _ip, _ip, _ip, _ip, _ip, _ip]
    L.pq_kkt_sparse_ordering.argtypes = [vp, _ip, _ip]
    L.pq_batch_create.argtypes = [C.POINTER(vp), C.c_int]
    L.pq_batch_destroy.argtypes = [vp]
    L.pq_batch_settings.argtypes = [vp]
    L.pq_batch_settings.restype = C.POINTER(Settings)
    L.pq_batch_setup_sparse.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int] + [vp] * 15
    L.pq_batch_update.argtypes = [vp] + [vp] * 6
    L.pq_batch_update_data.argtypes = [vp] + [vp] * 9
    L.pq_batch_solve.argtypes = [vp]
    L.pq_batch_info.argtypes = [vp, C.c_int]
    L.pq_batch_info.restype = C.POINTER(Info)
    L.pq_batch_get_result.argtypes = [vp, C.c_int, vp]
    L.pq_batch_dims.argtypes = [vp, _ip, _ip, _ip, _ip]
    L.pq_batch_block_info.argtypes = [vp, vp, C.c_int]
    L.pq_batch_get_profile.argtypes = [vp, C.c_int, vp]
    L.pq_batch_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_double), _ip]
    L.pq_batch_set_start_order.argtypes = [vp, C.c_int]
    L.pq_dense_factor_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int]
    L.pq_dense_factor_destroy.argtypes = [vp]
    L.pq_dense_factor_destroy.restype = None
    L.pq_dense_factor_compute.argtypes = [vp, vp, C.c_int, C.c_int]
    L.pq_dense_factor_info.argtypes = [vp]
    L.pq_dense_factor_solve_in_place.argtypes = [vp, vp, C.c_int]
    L.pq_dense_factor_matrix.argtypes = [vp, vp, C.c_int]
    L.pq_dense_factor_last_ms.argtypes = [vp, vp]
    L.pq_kkt_set_profiling.argtypes = [vp, C.c_int]
    L.pq_kkt_get_profile.argtypes = [vp, C.c_int, _dp, _ip]
    L.pq_kktsys_create_dense.argtypes = [C.POINTER(vp), C.POINTER(DenseData), C.POINTER(Settings), C.c_int]
    L.pq_kktsys_create_sparse.argtypes = [C.POINTER(vp), C.POINTER(SparseData), C.POINTER(Settings), C.c_int]
    L.pq_kktsys_clone.argtypes = [vp, C.POINTER(vp)]
    L.pq_kktsys_destroy.argtypes = [vp]
    L.pq_kktsys_destroy.restype = None
    L.pq_kktsys_set_pointer_mode.argtypes = [vp, C.c_int]
    L.pq_kktsys_backend.argtypes = [vp]
    L.pq_kktsys_backend.restype = vp
    L.pq_kktsys_update_data_dense.argtypes = [vp, C.POINTER(DenseData), C.c_int]
    L.pq_kktsys_update_data_sparse.argtypes = [vp, C.POINTER(SparseData), C.c_int]
    L.pq_kktsys_update_scalings_and_factor.argtypes = [vp, C.c_int, C.c_double, C.c_double, C.POINTER(Vars)]
    L.pq_kktsys_solve.argtypes = [vp, C.POINTER(Vars), C.POINTER(Vars)]
    L.pq_kktsys_mul.argtypes = [vp, C.POINTER(Vars), C.POINTER(Vars)]
    L.pq_kktsys_last_solve_stats.argtypes = [vp, _ip, _ip, _dp, _dp]
    L.pq_kktsys_condensed_residual.argtypes = [vp, _dp, _dp]
    L.pq_kktsys_synchronize.argtypes = [vp]
    L.pq_solver_create.argtypes = [C.POINTER(vp), C.c_int]
    L.pq_solver_destroy.argtypes = [vp]
    L.pq_solver_destroy.restype = None
    L.pq_solver_clone.argtypes = [vp, C.POINTER(vp)]
    L.pq_solver_settings.argtypes = [vp]
    L.pq_solver_settings.restype = C.POINTER(Settings)
    L.pq_solver_setup_dense.argtypes = [vp, C.c_int, C.c_int, C.c_int] + [vp] * 9
    L.pq_solver_setup_sparse.argtypes = [vp, C.c_int, C.c_int, C.c_int] + [vp] * 15
    L.pq_solver_update_dense.argtypes = [vp] + [vp] * 9
    L.pq_solver_update_sparse.argtypes = [vp] + [vp] * 15
    L.pq_solver_solve.argtypes = [vp]
    L.pq_solver_info.argtypes = [vp]
    L.pq_solver_info.restype = C.POINTER(Info)
    L.pq_solver_get_result.argtypes = [vp, C.POINTER(Vars)]
    L.pq_solver_dims.argtypes = [vp, _ip, _ip, _ip]
    L.pq_solver_set_trace.argtypes = [vp, vp, C.c_int]
    L.pq_solver_trace_rows.argtypes = [vp]
    L.pq_kkt_sparse_stats.argtypes = [vp, _dp]
    L.pq_kkt_partition.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_longlong)]
    L.pq_kkt_set_exchange.argtypes = [vp, EXCHANGE_FN, vp, vp, vp, vp]
    L.pq_kkt_partition_info.argtypes = [vp, _ip]
    L.pq_rccl_unique_id.argtypes = [vp]
    L.pq_kkt_set_comm_rccl.argtypes = [vp, vp, C.c_int, C.c_int]
    L.pq_solver_set_comm_rccl.argtypes = [vp, vp, C.c_int, C.c_int]
    L.pq_kkt_native_exchange_calls.argtypes = [vp, _ip]
    L.pq_kkt_min_abs_pivot.argtypes = [vp, _dp]
    L.pq_kkt_exact_factor.argtypes = [vp, C.c_int, vp]
    L.pq_sparse_uplooking_plan.argtypes = [vp, C.c_int, vp, vp, vp]
    L.pq_kkt_exact_factor.restype = C.c_longlong
    L.pq_solver_native_exchange_calls.argtypes = [vp, _ip]
    L.pq_kkt_comm_info.argtypes = [vp, _ip]
    L.pq_solver_comm_info.argtypes = [vp, _ip]
    L.pq_sparse_partition_plan.argtypes = [C.POINTER(SparseData), C.c_int, C.c_int, vp, C.c_int, vp]
    L.pq_solver_partition.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_longlong)]
    L.pq_solver_set_exchange.argtypes = [vp, EXCHANGE_FN, vp, vp, vp, vp]
    L.pq_debug_alloc_count.restype = C.c_longlong
    L.pq_debug_alloc_count.argtypes = []
    L.pq_debug_chol_plan.argtypes = [C.c_int, C.c_int, vp, C.c_int]
    L.pq_kkt_set_exchange_norm.argtypes = [vp, vp]
    L.pq_solver_set_exchange_norm.argtypes = [vp, vp]
    L.pq_kkt_sharded_calls.argtypes = [vp, C.POINTER(C.c_int * 2)]
    L.pq_solver_sharded_calls.argtypes = [vp, C.POINTER(C.c_int * 2)]
    L.pq_kkt_sharded_solve_calls.argtypes = [vp, C.POINTER(C.c_int * 6)]
    L.pq_solver_sharded_solve_calls.argtypes = [vp, C.POINTER(C.c_int * 6)]
    L.pq_microbench_mfma_f64.argtypes = [C.c_int, C.c_int, _dp]
    L.pq_microbench_hbm_copy.argtypes = [C.c_int, C.c_size_t, C.c_int, _dp]
    L.pq_microbench_potrf_block.argtypes = [C.c_int, C.c_int, C.c_int, _dp, C.c_void_p]
    L.pq_debug_potrf_block.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _lib = L
    return L


def check(rc, what=""):
    if rc < 0:
        raise RuntimeError(f"piqp_amd {what} failed ({rc}): {load().pq_last_error_string().decode()}")
    return rc
