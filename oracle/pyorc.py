"""ctypes binding of the CPU oracle (oracle/_build/liborc.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (piqp_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

VAR_NAMES = ("x", "y", "z_l", "z_u", "z_bl", "z_bu", "s_l", "s_u", "s_bl", "s_bu")

DENSE_CHOLESKY, SPARSE_LDLT, SPARSE_LDLT_EQ_COND, SPARSE_LDLT_INEQ_COND, SPARSE_LDLT_COND, SPARSE_MULTISTAGE = range(6)
DENSE_LDLT_NO_PIVOT = 16
KKT_UPDATE_P, KKT_UPDATE_A, KKT_UPDATE_G = 1, 2, 4
SOLVED, MAX_ITER_REACHED, PRIMAL_INFEASIBLE, DUAL_INFEASIBLE, NUMERICS, UNSOLVED, INVALID_SETTINGS = 1, -1, -2, -3, -8, -9, -10


class CSC(C.Structure):
    _fields_ = [("rows", C.c_int), ("cols", C.c_int), ("colptr", _ip), ("rowind", _ip), ("val", _dp)]


class DataS(C.Structure):
    _fields_ = [("is_sparse", C.c_int), ("n", C.c_int), ("p", C.c_int), ("m", C.c_int),
                ("P_utri", _dp), ("AT", _dp), ("GT", _dp),
                ("sP_utri", CSC), ("sAT", CSC), ("sGT", CSC),
                ("c", _dp), ("b", _dp), ("h_l", _dp), ("h_u", _dp), ("x_l", _dp), ("x_u", _dp),
                ("n_h_l", C.c_int), ("n_h_u", C.c_int), ("n_x_l", C.c_int), ("n_x_u", C.c_int),
                ("h_l_idx", _ip), ("h_u_idx", _ip), ("x_l_idx", _ip), ("x_u_idx", _ip),
                ("x_b_scaling", _dp)]


class VarsS(C.Structure):
    _fields_ = [(k, _dp) for k in VAR_NAMES]


class SettingsS(C.Structure):
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


class InfoS(C.Structure):
    _fields_ = [("status", C.c_int), ("iter", C.c_int)] + [(k, C.c_double) for k in (
        "rho", "delta", "mu", "sigma", "primal_step", "dual_step", "primal_res", "primal_res_rel", "dual_res",
        "dual_res_rel", "primal_res_reg", "primal_res_reg_rel", "dual_res_reg", "dual_res_reg_rel", "primal_prox_inf",
        "dual_prox_inf", "prev_primal_res", "prev_dual_res", "primal_obj", "dual_obj", "duality_gap",
        "duality_gap_rel")] + [("factor_retires", C.c_int), ("reg_limit", C.c_double), ("no_primal_update", C.c_int),
                               ("no_dual_update", C.c_int)] + [(k, C.c_double) for k in (
        "setup_time", "update_time", "solve_time", "kkt_factor_time", "kkt_solve_time", "run_time")] + [
        ("n_factor", C.c_int), ("n_solve", C.c_int), ("n_backend_solve", C.c_int)]


def build(native=False, fma=False):
    """Compile the oracle with gcc (seconds)."""
    subprocess.check_call(["make", "-s", "-C", _HERE] + (["native"] if native else (["fma"] if fma else [])))


_lib_fma = None


def lib_fma():
    """the FMA-contracted build of the same sources (oracle/Makefile, target fma): a second legal build of the reference arithmetic"""
    global _lib_fma
    if _lib_fma is None:
        path = os.path.join(_HERE, "_build", "liborc_fma.so")
        if not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(os.path.join(_HERE, f)) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))):
            build(fma=True)
        _lib_fma = _bind(C.CDLL(path))
    return _lib_fma


_lib = None
_lib_native = None


def lib(native=False):
    global _lib, _lib_native
    if native:
        if _lib_native is None:
            path = os.path.join(_HERE, "_build", "liborc_native.so")
            if not os.path.exists(path):
                build(native=True)
            _lib_native = _bind(C.CDLL(path))
        return _lib_native
    if _lib is None:
        path = os.path.join(_HERE, "_build", "liborc.so")
        if not os.path.exists(path):
            build()
        _lib = _bind(C.CDLL(path))
    return _lib


def _bind(L):
    vp = C.c_void_p
    L.orc_settings_default.argtypes = [C.POINTER(SettingsS)]
    L.orc_data_create_dense.restype = C.POINTER(DataS)
    L.orc_data_create_dense.argtypes = [C.c_int] * 3 + [_dp] * 9
    L.orc_data_create_sparse.restype = C.POINTER(DataS)
    L.orc_data_create_sparse.argtypes = [C.c_int] * 3 + [_ip, _ip, _dp, _dp, _ip, _ip, _dp, _dp, _ip, _ip, _dp, _dp, _dp, _dp, _dp]
    L.orc_data_clone.restype = C.POINTER(DataS)
    L.orc_data_clone.argtypes = [C.POINTER(DataS)]
    L.orc_data_free.argtypes = [C.POINTER(DataS)]
    L.orc_dense_kkt_create.restype = vp
    L.orc_dense_kkt_create.argtypes = [C.POINTER(DataS), C.c_int]
    if hasattr(L, "orc_sparse_kkt_create"):
        L.orc_sparse_kkt_create.restype = vp
        L.orc_sparse_kkt_create.argtypes = [C.POINTER(DataS), C.c_int]
    if hasattr(L, "orc_multistage_kkt_create"):
        L.orc_multistage_kkt_create.restype = vp
        L.orc_multistage_kkt_create.argtypes = [C.POINTER(DataS)]
        L.orc_multistage_num_blocks.restype = C.c_int
        L.orc_multistage_num_blocks.argtypes = [vp]
        L.orc_multistage_block_info.argtypes = [vp, _ip]
        L.orc_multistage_row_perm.argtypes = [vp, C.c_int, _ip, _ip]
    L.orc_dense_kkt_internal_kkt_mat.restype = _dp
    L.orc_dense_kkt_internal_kkt_mat.argtypes = [vp]
    L.orc_dense_kkt_internal_factor.restype = _dp
    L.orc_dense_kkt_internal_factor.argtypes = [vp]
    L.orc_set_num_threads.argtypes = [C.c_int]
    L.orc_llt_compute.restype = C.c_int
    L.orc_llt_compute.argtypes = [_dp, C.c_int, C.c_int]
    L.orc_llt_solve_inplace.argtypes = [_dp, C.c_int, C.c_int, _dp]
    L.orc_ldlt_no_pivot_compute.restype = C.c_int
    L.orc_ldlt_no_pivot_compute.argtypes = [_dp, C.c_int, C.c_int, _dp]
    L.orc_ldlt_no_pivot_solve_inplace.argtypes = [_dp, C.c_int, C.c_int, _dp]
    L.orc_kkt_clone.restype = vp
    L.orc_kkt_clone.argtypes = [vp]
    L.orc_kkt_destroy.argtypes = [vp]
    L.orc_kkt_update_data.argtypes = [vp, C.POINTER(DataS), C.c_int]
    L.orc_kkt_update_scalings_and_factor.restype = C.c_int
    L.orc_kkt_update_scalings_and_factor.argtypes = [vp, C.POINTER(DataS), C.c_double, _dp, _dp]
    L.orc_kkt_solve.argtypes = [vp, C.POINTER(DataS)] + [_dp] * 6
    L.orc_kkt_eval_P_x.argtypes = [vp, C.POINTER(DataS), C.c_double, _dp, _dp]
    L.orc_kkt_eval_A_xn_and_AT_xt.argtypes = [vp, C.POINTER(DataS), C.c_double, C.c_double] + [_dp] * 4
    L.orc_kkt_eval_G_xn_and_GT_xt.argtypes = [vp, C.POINTER(DataS), C.c_double, C.c_double] + [_dp] * 4
    L.orc_kkt_system_create.restype = vp
    L.orc_kkt_system_create.argtypes = [C.POINTER(DataS), C.POINTER(SettingsS)]
    L.orc_kkt_system_clone.restype = vp
    L.orc_kkt_system_clone.argtypes = [vp]
    L.orc_kkt_system_free.argtypes = [vp]
    L.orc_kkt_system_backend.restype = vp
    L.orc_kkt_system_backend.argtypes = [vp]
    L.orc_kkt_system_update_data.argtypes = [vp, C.POINTER(DataS), C.c_int]
    L.orc_kkt_system_update_scalings_and_factor.restype = C.c_int
    L.orc_kkt_system_update_scalings_and_factor.argtypes = [vp, C.POINTER(DataS), C.POINTER(SettingsS), C.c_int,
                                                            C.c_double, C.c_double, C.POINTER(VarsS)]
    L.orc_kkt_system_solve_copy.restype = C.c_int
    L.orc_kkt_system_solve_copy.argtypes = [vp, C.POINTER(DataS), C.POINTER(SettingsS), C.POINTER(VarsS), C.POINTER(VarsS)]
    L.orc_kkt_system_mul.argtypes = [vp, C.POINTER(DataS), C.POINTER(VarsS), C.POINTER(VarsS)]
    L.orc_kkt_system_last_refine_steps.restype = C.c_int
    L.orc_kkt_system_last_refine_steps.argtypes = [vp]
    L.orc_kkt_system_backend_solves.restype = C.c_int
    L.orc_kkt_system_backend_solves.argtypes = [vp]
    for nm in ("x_reg", "z_reg", "rhs_x_bar", "rhs_z_bar"):
        f = getattr(L, "orc_kkt_system_" + nm)
        f.restype = _dp
        f.argtypes = [vp]
    L.orc_solver_create.restype = vp
    L.orc_solver_clone.restype = vp
    L.orc_solver_clone.argtypes = [vp]
    L.orc_solver_free.argtypes = [vp]
    L.orc_solver_settings.restype = C.POINTER(SettingsS)
    L.orc_solver_settings.argtypes = [vp]
    L.orc_solver_setup.restype = C.c_int
    L.orc_solver_setup.argtypes = [vp, C.POINTER(DataS)]
    L.orc_solver_update_dense.restype = C.c_int
    L.orc_solver_update_dense.argtypes = [vp] + [_dp] * 9
    L.orc_solver_update_sparse.restype = C.c_int
    L.orc_solver_update_sparse.argtypes = [vp, _ip, _ip, _dp, _dp, _ip, _ip, _dp, _dp, _ip, _ip, _dp, _dp, _dp, _dp, _dp]
    L.orc_solver_solve.restype = C.c_int
    L.orc_solver_solve.argtypes = [vp]
    L.orc_solver_info.restype = C.POINTER(InfoS)
    L.orc_solver_info.argtypes = [vp]
    L.orc_solver_result.restype = C.POINTER(VarsS)
    L.orc_solver_result.argtypes = [vp]
    L.orc_solver_data.restype = C.POINTER(DataS)
    L.orc_solver_data.argtypes = [vp]
    L.orc_solver_set_trace.argtypes = [vp, _dp, C.c_int]
    L.orc_solver_trace_rows.restype = C.c_int
    L.orc_solver_trace_rows.argtypes = [vp]
    L.ORC_STATE_CB = C.CFUNCTYPE(None, vp, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(VarsS))
    L.orc_solver_set_state_callback.argtypes = [vp, L.ORC_STATE_CB, vp]
    if hasattr(L, "orc_sparse_ldlt_create"):
        L.orc_sparse_ldlt_create.restype = vp
        L.orc_sparse_ldlt_free.argtypes = [vp]
        L.orc_sparse_ldlt_symbolic.argtypes = [vp, C.c_int, _ip, _ip]
        L.orc_sparse_ldlt_numeric.restype = C.c_int
        L.orc_sparse_ldlt_numeric.argtypes = [vp, C.c_int, _ip, _ip, _dp]
        L.orc_sparse_ldlt_solve_inplace.argtypes = [vp, _dp]
        L.orc_sparse_ldlt_nnz.restype = C.c_int
        L.orc_sparse_ldlt_nnz.argtypes = [vp]
        L.orc_amd_order.argtypes = [C.c_int, _ip, _ip, _ip]
        L.orc_permute_sym_upper.argtypes = [C.c_int, _ip, _ip, _dp, _ip, _ip, _ip, _dp, _ip]
        for nm, rt in (("dim", C.c_int), ("PKPt_colptr", _ip), ("PKPt_rowind", _ip), ("PKPt_val", _dp), ("perm", _ip), ("L_nnz", C.c_int), ("PKi", _ip),
                       ("nnz", C.c_int), ("L_cols", _ip), ("L_ind", _ip), ("L_vals", _dp), ("D", _dp), ("D_inv", _dp), ("etree", _ip)):
            f = getattr(L, "orc_sparse_kkt_" + nm)
            f.restype = rt
            f.argtypes = [vp]
        for nm, rt in (("dim", C.c_int), ("nnz", C.c_int), ("perm", _ip), ("PKPt_colptr", _ip), ("PKPt_rowind", _ip), ("PKi", _ip)):
            f = getattr(L, "orc_sparse_cond_kkt_" + nm)
            f.restype = rt
            f.argtypes = [vp]
    return L


def _f(a):
    """contiguous fp64 array or None -> (array kept alive, pointer)"""
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _fcol(a):
    if a is None:
        return None, None
    a = np.asfortranarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def _view(ptr, n, dtype=np.float64):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,))


def make_vars(n, p, m, fill=0.0):
    sizes = dict(x=n, y=p, z_l=m, z_u=m, z_bl=n, z_bu=n, s_l=m, s_u=m, s_bl=n, s_bu=n)
    return {k: np.full(sizes[k], fill, dtype=np.float64) for k in VAR_NAMES}


def _vars_struct(v):
    s = VarsS()
    for k in VAR_NAMES:
        a = v[k]
        assert a.dtype == np.float64 and a.flags.c_contiguous
        setattr(s, k, a.ctypes.data_as(_dp))
    return s


class Settings:
    def __init__(self, L=None, **kw):
        self.L = L or lib()
        self.s = SettingsS()
        self.L.orc_settings_default(C.byref(self.s))
        for k, v in kw.items():
            setattr(self.s, k, v)


class Data:
    """orc_data wrapper (dense/data.hpp / sparse/data.hpp restatement)."""

    def __init__(self, ptr, L, owned=True):
        self.ptr, self.L, self.owned = ptr, L, owned

    @classmethod
    def dense(cls, P, c, A=None, b=None, G=None, h_l=None, h_u=None, x_l=None, x_u=None, L=None):
        L = L or lib()
        n = P.shape[0]
        p = 0 if A is None else A.shape[0]
        m = 0 if G is None else G.shape[0]
        keep = [_fcol(P), _f(c), _fcol(A), _f(b), _fcol(G), _f(h_l), _f(h_u), _f(x_l), _f(x_u)]
        ptr = L.orc_data_create_dense(n, p, m, *[k[1] for k in keep])
        return cls(ptr, L)

    @classmethod
    def sparse(cls, P, c, A=None, b=None, G=None, h_l=None, h_u=None, x_l=None, x_u=None, L=None):
        import scipy.sparse as sp
        L = L or lib()
        n = P.shape[0]
        P = sp.csc_matrix(P); P.sort_indices()
        p = 0 if A is None else A.shape[0]
        m = 0 if G is None else G.shape[0]
        args = [_i(P.indptr), _i(P.indices), _f(P.data), _f(c)]
        if A is not None:
            A = sp.csc_matrix(A); A.sort_indices()
            args += [_i(A.indptr), _i(A.indices), _f(A.data)]
        else:
            args += [(None, None)] * 3
        args += [_f(b)]
        if G is not None:
            G = sp.csc_matrix(G); G.sort_indices()
            args += [_i(G.indptr), _i(G.indices), _f(G.data)]
        else:
            args += [(None, None)] * 3
        args += [_f(h_l), _f(h_u), _f(x_l), _f(x_u)]
        ptr = L.orc_data_create_sparse(n, p, m, *[a[1] for a in args])
        return cls(ptr, L)

    def clone(self):
        return Data(self.L.orc_data_clone(self.ptr), self.L)

    def __del__(self):
        if getattr(self, "owned", False) and self.ptr:
            self.L.orc_data_free(self.ptr)
            self.ptr = None

    # views into the C-owned arrays
    @property
    def n(self): return self.ptr.contents.n
    @property
    def p(self): return self.ptr.contents.p
    @property
    def m(self): return self.ptr.contents.m

    def mat(self, name):
        d = self.ptr.contents
        cols = {"P_utri": d.n, "AT": d.p, "GT": d.m}[name]
        if d.n * cols == 0:
            return np.zeros((d.n, cols), order="F")
        return np.ctypeslib.as_array(getattr(d, name), shape=(cols, d.n)).T  # column-major n x cols view

    def csc(self, name):
        import scipy.sparse as sp
        d = self.ptr.contents
        cs = getattr(d, "s" + name)
        nnz = cs.colptr[cs.cols]
        return sp.csc_matrix((_view(cs.val, nnz).copy(), _view(cs.rowind, nnz, np.int32).copy(),
                              _view(cs.colptr, cs.cols + 1, np.int32).copy()), shape=(cs.rows, cs.cols))

    def vec(self, name):
        d = self.ptr.contents
        size = dict(c=d.n, b=d.p, h_l=d.m, h_u=d.m, x_l=d.n, x_u=d.n, x_b_scaling=d.n)[name]
        return _view(getattr(d, name), size)

    def idx(self, name):
        d = self.ptr.contents
        cnt = getattr(d, "n_" + name)
        return _view(getattr(d, name + "_idx"), cnt, np.int32)[:cnt].copy() if cnt else np.zeros(0, np.int32)

    def counts(self):
        d = self.ptr.contents
        return d.n_h_l, d.n_h_u, d.n_x_l, d.n_x_u


class KKT:
    """KKTSolverBase restatement (dense::KKT / sparse::KKT)."""

    def __init__(self, data, kind="dense", use_ldlt=False, mode=0, _ptr=None):
        self.L, self.data = data.L, data
        if _ptr is not None:
            self.ptr = _ptr
        elif kind == "dense":
            self.ptr = self.L.orc_dense_kkt_create(data.ptr, int(use_ldlt))
        elif kind == "multistage":
            self.ptr = self.L.orc_multistage_kkt_create(data.ptr)
        else:
            self.ptr = self.L.orc_sparse_kkt_create(data.ptr, mode)
        self.owned = _ptr is None

    def __del__(self):
        if getattr(self, "owned", False) and self.ptr:
            self.L.orc_kkt_destroy(self.ptr)
            self.ptr = None

    def clone(self):
        k = KKT(self.data, _ptr=self.L.orc_kkt_clone(self.ptr))
        k.owned = True
        return k

    def update_data(self, options, data=None):
        self.L.orc_kkt_update_data(self.ptr, (data or self.data).ptr, options)

    def update_scalings_and_factor(self, delta, x_reg, z_reg, data=None):
        x, xp = _f(x_reg); z, zp = _f(z_reg)
        return bool(self.L.orc_kkt_update_scalings_and_factor(self.ptr, (data or self.data).ptr, delta, xp, zp))

    def solve(self, rhs_x, rhs_y, rhs_z, data=None):
        d = (data or self.data)
        a, ap = _f(rhs_x); b, bp = _f(rhs_y); c, cp = _f(rhs_z)
        lx, ly, lz = np.zeros(d.n), np.zeros(d.p), np.zeros(d.m)
        self.L.orc_kkt_solve(self.ptr, d.ptr, ap, bp, cp, lx.ctypes.data_as(_dp), ly.ctypes.data_as(_dp), lz.ctypes.data_as(_dp))
        return lx, ly, lz

    def eval_P_x(self, alpha, x, data=None):
        d = (data or self.data)
        a, ap = _f(x); z = np.zeros(d.n)
        self.L.orc_kkt_eval_P_x(self.ptr, d.ptr, alpha, ap, z.ctypes.data_as(_dp))
        return z

    def eval_A_xn_and_AT_xt(self, an, at, xn, xt, data=None):
        d = (data or self.data)
        a, ap = _f(xn); b, bp = _f(xt); zn, zt = np.zeros(d.p), np.zeros(d.n)
        self.L.orc_kkt_eval_A_xn_and_AT_xt(self.ptr, d.ptr, an, at, ap, bp, zn.ctypes.data_as(_dp), zt.ctypes.data_as(_dp))
        return zn, zt

    def eval_G_xn_and_GT_xt(self, an, at, xn, xt, data=None):
        d = (data or self.data)
        a, ap = _f(xn); b, bp = _f(xt); zn, zt = np.zeros(d.m), np.zeros(d.n)
        self.L.orc_kkt_eval_G_xn_and_GT_xt(self.ptr, d.ptr, an, at, ap, bp, zn.ctypes.data_as(_dp), zt.ctypes.data_as(_dp))
        return zn, zt

    def sparse_factor(self):
        """sparse_ldlt (KKT_FULL) only: the factor as sparse/ldlt.hpp:24-37 holds it + P K P' (numpy copies)"""
        L, k = self.L, self.ptr
        N = L.orc_sparse_kkt_dim(k)
        Lp = np.ctypeslib.as_array(L.orc_sparse_kkt_L_cols(k), (N + 1,)).copy()
        nz = int(Lp[N]); nk = L.orc_sparse_kkt_nnz(k)
        g = lambda f, n, : np.ctypeslib.as_array(f(k), (max(n, 1),))[:n].copy()
        return dict(N=N, L_cols=Lp, L_ind=g(L.orc_sparse_kkt_L_ind, nz), L_vals=g(L.orc_sparse_kkt_L_vals, nz), D=g(L.orc_sparse_kkt_D, N), D_inv=g(L.orc_sparse_kkt_D_inv, N),
                    perm=g(L.orc_sparse_kkt_perm, N), PKPt_colptr=g(L.orc_sparse_kkt_PKPt_colptr, N + 1), PKPt_rowind=g(L.orc_sparse_kkt_PKPt_rowind, nk),
                    PKPt_val=g(L.orc_sparse_kkt_PKPt_val, nk), etree=g(L.orc_sparse_kkt_etree, N))

    def block_info(self):
        """multistage only: rows of (start, diag_size, off_diag_size); the last row is the arrow corner block."""
        N = self.L.orc_multistage_num_blocks(self.ptr)
        out = np.zeros(3 * N, np.int32)
        self.L.orc_multistage_block_info(self.ptr, out.ctypes.data_as(_ip))
        return out.reshape(N, 3)

    def row_perm(self, which):
        """multistage only: (perm, block_row_sizes) of AT (which=0) / GT (which=1)."""
        N = self.L.orc_multistage_num_blocks(self.ptr)
        rows = self.data.p if which == 0 else self.data.m
        perm = np.zeros(max(rows, 1), np.int32); sizes = np.zeros(max(N - 1, 1), np.int32)
        self.L.orc_multistage_row_perm(self.ptr, which, perm.ctypes.data_as(_ip), sizes.ctypes.data_as(_ip))
        return perm[:rows], sizes[:N - 1]

    def internal_kkt_mat(self):
        n = self.data.n
        return np.ctypeslib.as_array(self.L.orc_dense_kkt_internal_kkt_mat(self.ptr), shape=(n, n)).T

    def internal_factor(self):
        n = self.data.n
        return np.ctypeslib.as_array(self.L.orc_dense_kkt_internal_factor(self.ptr), shape=(n, n)).T


class KKTSystem:
    def __init__(self, data, settings=None):
        self.L, self.data = data.L, data
        self.settings = settings or Settings(self.L)
        self.ptr = self.L.orc_kkt_system_create(data.ptr, C.byref(self.settings.s))
        if not self.ptr:
            raise RuntimeError("kkt solver not supported")

    def __del__(self):
        if getattr(self, "ptr", None):
            self.L.orc_kkt_system_free(self.ptr)
            self.ptr = None

    def backend(self):
        return KKT(self.data, _ptr=self.L.orc_kkt_system_backend(self.ptr))

    def update_data(self, options):
        self.L.orc_kkt_system_update_data(self.ptr, self.data.ptr, options)

    def update_scalings_and_factor(self, iterative_refinement, rho, delta, vars_):
        vs = _vars_struct(vars_)
        return bool(self.L.orc_kkt_system_update_scalings_and_factor(self.ptr, self.data.ptr, C.byref(self.settings.s),
                                                                     int(iterative_refinement), rho, delta, C.byref(vs)))

    def solve(self, rhs):
        d = self.data
        lhs = make_vars(d.n, d.p, d.m)
        rs, ls = _vars_struct(rhs), _vars_struct(lhs)
        ok = self.L.orc_kkt_system_solve_copy(self.ptr, d.ptr, C.byref(self.settings.s), C.byref(rs), C.byref(ls))
        return bool(ok), lhs

    def mul(self, lhs):
        d = self.data
        rhs = make_vars(d.n, d.p, d.m)
        ls, rs = _vars_struct(lhs), _vars_struct(rhs)
        self.L.orc_kkt_system_mul(self.ptr, d.ptr, C.byref(ls), C.byref(rs))
        return rhs

    def x_reg(self): return _view(self.L.orc_kkt_system_x_reg(self.ptr), self.data.n).copy()
    def z_reg(self): return _view(self.L.orc_kkt_system_z_reg(self.ptr), self.data.m).copy()
    def rhs_x_bar(self): return _view(self.L.orc_kkt_system_rhs_x_bar(self.ptr), self.data.n).copy()
    def rhs_z_bar(self): return _view(self.L.orc_kkt_system_rhs_z_bar(self.ptr), self.data.m).copy()
    def last_refine_steps(self): return self.L.orc_kkt_system_last_refine_steps(self.ptr)


class Solver:
    """DenseSolver / SparseSolver restatement (solver.hpp)."""

    def __init__(self, native=False, _ptr=None, _L=None):
        self.L = _L or lib(native)
        self.ptr = _ptr or self.L.orc_solver_create()
        self._trace = None
        self._cb = None

    def __del__(self):
        if getattr(self, "ptr", None):
            self.L.orc_solver_free(self.ptr)
            self.ptr = None

    @property
    def settings(self):
        return self.L.orc_solver_settings(self.ptr).contents

    def setup(self, P, c, A=None, b=None, G=None, h_l=None, h_u=None, x_l=None, x_u=None, sparse=False):
        mk = Data.sparse if sparse else Data.dense
        d = mk(P, c, A, b, G, h_l, h_u, x_l, x_u, L=self.L)
        d.owned = False  # solver takes ownership
        self.sparse = sparse
        return bool(self.L.orc_solver_setup(self.ptr, d.ptr))

    def update(self, P=None, c=None, A=None, b=None, G=None, h_l=None, h_u=None, x_l=None, x_u=None):
        if not getattr(self, "sparse", False):
            keep = [_fcol(P), _f(c), _fcol(A), _f(b), _fcol(G), _f(h_l), _f(h_u), _f(x_l), _f(x_u)]
            return bool(self.L.orc_solver_update_dense(self.ptr, *[k[1] for k in keep]))
        import scipy.sparse as sp
        args = []
        for M in (P, A, G):
            if M is None:
                args.append([(None, None)] * 3)
            else:
                M = sp.csc_matrix(M); M.sort_indices()
                args.append([_i(M.indptr), _i(M.indices), _f(M.data)])
        flat = args[0] + [_f(c)] + args[1] + [_f(b)] + args[2] + [_f(h_l), _f(h_u), _f(x_l), _f(x_u)]
        return bool(self.L.orc_solver_update_sparse(self.ptr, *[a[1] for a in flat]))

    def enable_trace(self, max_rows=512):
        self._trace = np.zeros((max_rows, 11))
        self.L.orc_solver_set_trace(self.ptr, self._trace.ctypes.data_as(_dp), max_rows)

    def trace(self):
        return self._trace[: self.L.orc_solver_trace_rows(self.ptr)].copy()

    def record_states(self):
        """capture the (kind, refine, rho, delta, vars) inputs of every KKTSystem factor/solve call"""
        states = []
        d = self.L.orc_solver_data(self.ptr).contents
        n, p, m = d.n, d.p, d.m
        sizes = dict(x=n, y=p, z_l=m, z_u=m, z_bl=n, z_bu=n, s_l=m, s_u=m, s_bl=n, s_bu=n)

        def cb(user, kind, refine, rho, delta, vptr):
            v = vptr.contents
            states.append(dict(kind=kind, refine=refine, rho=rho, delta=delta,
                               vars={k: _view(getattr(v, k), sizes[k]).copy() for k in VAR_NAMES}))
        self._cb = self.L.ORC_STATE_CB(cb)
        self.L.orc_solver_set_state_callback(self.ptr, self._cb, None)
        return states

    def solve(self):
        return self.L.orc_solver_solve(self.ptr)

    def clone(self):
        return Solver(_ptr=self.L.orc_solver_clone(self.ptr), _L=self.L)

    @property
    def info(self):
        return self.L.orc_solver_info(self.ptr).contents

    def data(self):
        return Data(self.L.orc_solver_data(self.ptr), self.L, owned=False)

    def result(self):
        d = self.L.orc_solver_data(self.ptr).contents
        r = self.L.orc_solver_result(self.ptr).contents
        sizes = dict(x=d.n, y=d.p, z_l=d.m, z_u=d.m, z_bl=d.n, z_bu=d.n, s_l=d.m, s_u=d.m, s_bl=d.n, s_bu=d.n)
        return {k: _view(getattr(r, k), sizes[k]).copy() for k in VAR_NAMES}

