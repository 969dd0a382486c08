"""Host-side mirror of the reference's KKT plugin surface over the C-ABI.

Class / method names and argument meaning follow PIQP v0.6.2:
  Data            dense::Data<T>                      include/piqp/dense/data.hpp:22-208
  DenseKKT        dense::KKT<T> : KKTSolverBase       include/piqp/dense/kkt.hpp, kkt_solver_base.hpp:20-44
  KKTSystem       piqp::KKTSystem<T,I,PIQP_DENSE>     include/piqp/kkt_system.hpp
  Variables       piqp::Variables<T>                  include/piqp/variables.hpp
Vectors may be numpy arrays (PQ_MEM_HOST: staged by the library) or torch CUDA tensors
(PQ_MEM_DEVICE: used in place in HBM, asynchronous on the handle's stream).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import VAR_NAMES, check

PIQP_INF = 1e30  # fwd.hpp:54
DENSE_CHOLESKY, SPARSE_LDLT, SPARSE_LDLT_EQ_COND, SPARSE_LDLT_INEQ_COND, SPARSE_LDLT_COND, SPARSE_MULTISTAGE = range(6)
SPARSE_LDLT_EXACT, SPARSE_LDLT_MULTIFRONTAL = 17, 18  # the two engines behind SPARSE_LDLT, selectable directly (include/piqp_amd.h)
DENSE_LDLT_NO_PIVOT = 16
KKT_UPDATE_NONE, KKT_UPDATE_P, KKT_UPDATE_A, KKT_UPDATE_G = 0, 1, 2, 4
MEM_HOST, MEM_DEVICE = 0, 1


def _is_torch(a):
    return type(a).__module__.startswith("torch")


def _ptr(a):
    """raw address of a numpy array / torch tensor (None -> NULL)"""
    if a is None:
        return None
    if _is_torch(a):
        return a.data_ptr()
    return a.ctypes.data


def _f64(a, order="C"):
    if a is None:
        return None
    if _is_torch(a):
        return a
    return np.require(a, dtype=np.float64, requirements=["F_CONTIGUOUS" if order == "F" else "C_CONTIGUOUS", "ALIGNED"])


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def var_sizes(n, p, m):
    return dict(x=n, y=p, z_l=m, z_u=m, z_bl=n, z_bu=n, s_l=m, s_u=m, s_bl=n, s_bu=n)


class Variables(dict):
    """piqp::Variables<T> as a dict of ten numpy vectors (variables.hpp:19-105)."""

    @classmethod
    def zeros(cls, n, p, m, fill=0.0):
        return cls({k: np.full(sz, fill, dtype=np.float64) for k, sz in var_sizes(n, p, m).items()})

    def to_struct(self):
        s = _lib.Vars()
        for k in VAR_NAMES:
            setattr(s, k, _ptr(self[k]))
        return s


class Data:
    """dense::Data<T>: stores the upper triangle of P and the TRANSPOSED A and G, plus the finite-bound
    index lists (dense/data.hpp:53-208).  All arithmetic here is setup-time host logic."""

    def __init__(self, P, c, A=None, b=None, G=None, h_l=None, h_u=None, x_l=None, x_u=None):
        P = np.asarray(P, dtype=np.float64)
        n = P.shape[0]
        self.n = n
        self.p = 0 if A is None else np.asarray(A).shape[0]
        self.m = 0 if G is None else np.asarray(G).shape[0]
        self.P_utri = np.asfortranarray(np.triu(P))
        self.AT = np.asfortranarray(np.asarray(A, dtype=np.float64).T) if self.p else np.zeros((n, 0), order="F")
        self.GT = np.asfortranarray(np.asarray(G, dtype=np.float64).T) if self.m else np.zeros((n, 0), order="F")
        self.c = np.array(c, dtype=np.float64)
        self.b = np.array(b, dtype=np.float64) if self.p else np.zeros(0)
        self.h_l = np.zeros(self.m); self.h_u = np.zeros(self.m)
        self.x_l = np.zeros(n); self.x_u = np.zeros(n)
        self.x_b_scaling = np.ones(n)
        self.set_h_l(h_l); self.set_h_u(h_u); self.disable_inf_constraints()
        self.set_x_l(x_l); self.set_x_u(x_u)

    # dense/data.hpp:98-142
    def set_h_l(self, h_l):
        if h_l is None:
            self.h_l[:] = -PIQP_INF
            self.h_l_idx = np.zeros(0, np.int32)
        else:
            h = np.asarray(h_l, dtype=np.float64)
            fin = h > -PIQP_INF
            self.h_l = np.where(fin, h, -PIQP_INF)
            self.h_l_idx = np.nonzero(fin)[0].astype(np.int32)
        self.n_h_l = len(self.h_l_idx)

    def set_h_u(self, h_u):
        if h_u is None:
            self.h_u[:] = PIQP_INF
            self.h_u_idx = np.zeros(0, np.int32)
        else:
            h = np.asarray(h_u, dtype=np.float64)
            fin = h < PIQP_INF
            self.h_u = np.where(fin, h, PIQP_INF)
            self.h_u_idx = np.nonzero(fin)[0].astype(np.int32)
        self.n_h_u = len(self.h_u_idx)

    # dense/data.hpp:144-169
    def disable_inf_constraints(self):
        both = (self.h_l <= -PIQP_INF) & (self.h_u >= PIQP_INF)
        if both.any():
            self.GT[:, both] = 0.0
            self.h_l[both] = -1.0
            self.h_u[both] = 1.0
            self.set_h_l(self.h_l.copy()); self.set_h_u(self.h_u.copy())

    # dense/data.hpp:171-207 (finite bounds compressed into the head of x_l / x_u)
    def set_x_l(self, x_l):
        self.x_l_idx = np.zeros(0, np.int32)
        if x_l is not None:
            x = np.asarray(x_l, dtype=np.float64)
            fin = x > -PIQP_INF
            self.x_l_idx = np.nonzero(fin)[0].astype(np.int32)
            self.x_l[: fin.sum()] = x[fin]
        self.n_x_l = len(self.x_l_idx)

    def set_x_u(self, x_u):
        self.x_u_idx = np.zeros(0, np.int32)
        if x_u is not None:
            x = np.asarray(x_u, dtype=np.float64)
            fin = x < PIQP_INF
            self.x_u_idx = np.nonzero(fin)[0].astype(np.int32)
            self.x_u[: fin.sum()] = x[fin]
        self.n_x_u = len(self.x_u_idx)

    def descriptor(self):
        """pq_dense_data view of this object (host memory); keeps the arrays alive on self"""
        d = _lib.DenseData()
        d.n, d.p, d.m = self.n, self.p, self.m
        self._keep = [np.asfortranarray(self.P_utri, dtype=np.float64), np.asfortranarray(self.AT, dtype=np.float64),
                      np.asfortranarray(self.GT, dtype=np.float64), _i32(self.h_l_idx), _i32(self.h_u_idx),
                      _i32(self.x_l_idx), _i32(self.x_u_idx), np.ascontiguousarray(self.x_b_scaling, dtype=np.float64)]
        d.P_utri, d.AT, d.GT = (a.ctypes.data for a in self._keep[:3])
        d.n_h_l, d.n_h_u, d.n_x_l, d.n_x_u = self.n_h_l, self.n_h_u, self.n_x_l, self.n_x_u
        d.h_l_idx, d.h_u_idx, d.x_l_idx, d.x_u_idx = (a.ctypes.data for a in self._keep[3:7])
        d.x_b_scaling = self._keep[7].ctypes.data
        d.mem = MEM_HOST
        return d


class SparseData(Data):
    """sparse::Data<T,I> (sparse/data.hpp:26-231): P_utri, AT, GT as CSC int32/fp64 + the same bound bookkeeping."""

    def __init__(self, P, c, A=None, b=None, G=None, h_l=None, h_u=None, x_l=None, x_u=None):
        import scipy.sparse as sp
        P = sp.csc_matrix(P)
        n = P.shape[0]
        self.n = n
        self.p = 0 if A is None else A.shape[0]
        self.m = 0 if G is None else G.shape[0]
        self.P_utri = sp.triu(P, format="csc"); self.P_utri.sort_indices()
        self.AT = sp.csc_matrix(A).T.tocsc() if self.p else sp.csc_matrix((n, 0)); self.AT.sort_indices()
        self.GT = sp.csc_matrix(G).T.tocsc() if self.m else sp.csc_matrix((n, 0)); self.GT.sort_indices()
        self.c = np.array(c, dtype=np.float64)
        self.b = np.array(b, dtype=np.float64) if self.p else np.zeros(0)
        self.h_l = np.zeros(self.m); self.h_u = np.zeros(self.m)
        self.x_l = np.zeros(n); self.x_u = np.zeros(n)
        self.x_b_scaling = np.ones(n)
        self.set_h_l(h_l); self.set_h_u(h_u); self.disable_inf_constraints()
        self.set_x_l(x_l); self.set_x_u(x_u)

    def disable_inf_constraints(self):
        both = (self.h_l <= -PIQP_INF) & (self.h_u >= PIQP_INF)
        if both.any():
            for i in np.nonzero(both)[0]:
                self.GT.data[self.GT.indptr[i]:self.GT.indptr[i + 1]] = 0.0
            self.h_l[both] = -1.0
            self.h_u[both] = 1.0
            self.set_h_l(self.h_l.copy()); self.set_h_u(self.h_u.copy())

    def descriptor(self):
        d = _lib.SparseData()
        d.n, d.p, d.m = self.n, self.p, self.m
        k = []
        for M in (self.P_utri, self.AT, self.GT):
            k += [_i32(M.indptr), _i32(M.indices), np.ascontiguousarray(M.data, dtype=np.float64)]
        k += [_i32(self.h_l_idx), _i32(self.h_u_idx), _i32(self.x_l_idx), _i32(self.x_u_idx), np.ascontiguousarray(self.x_b_scaling, dtype=np.float64)]
        self._keep = k
        (d.P_colptr, d.P_rowind, d.P_val, d.AT_colptr, d.AT_rowind, d.AT_val, d.GT_colptr, d.GT_rowind, d.GT_val) = (a.ctypes.data for a in k[:9])
        d.n_h_l, d.n_h_u, d.n_x_l, d.n_x_u = self.n_h_l, self.n_h_u, self.n_x_l, self.n_x_u
        d.h_l_idx, d.h_u_idx, d.x_l_idx, d.x_u_idx = (a.ctypes.data for a in k[9:13])
        d.x_b_scaling = k[13].ctypes.data
        d.mem = MEM_HOST
        return d


class _Handle:
    _destroy = None

    def __del__(self):
        h = getattr(self, "h", None)
        if h and getattr(self, "_owned", True):
            getattr(self.L, self._destroy)(h)
            self.h = None


class SparseKKTMixin:
    pass


class DenseKKT(_Handle):
    """dense::KKT<T> (dense/kkt.hpp) behind pq_kkt_*.  Same seven operations as KKTSolverBase."""
    _destroy = "pq_kkt_destroy"

    def __init__(self, data, kkt_solver=DENSE_CHOLESKY, device=0, _h=None, _owned=True):
        self.L = _lib.load()
        self._owned = _owned
        if _h is not None:
            self.h = _h
        else:
            h = C.c_void_p()
            desc = data.descriptor()
            if isinstance(data, SparseData):
                check(self.L.pq_kkt_create_sparse(C.byref(h), C.byref(desc), kkt_solver, device), "pq_kkt_create_sparse")
            else:
                check(self.L.pq_kkt_create_dense(C.byref(h), C.byref(desc), kkt_solver, device), "pq_kkt_create_dense")
            self.h = h
        n, p, m = C.c_int(), C.c_int(), C.c_int()
        self.L.pq_kkt_dims(self.h, C.byref(n), C.byref(p), C.byref(m))
        self.n, self.p, self.m = n.value, p.value, m.value
        self._mode = MEM_HOST

    def _set_mode(self, *arrays):
        mode = MEM_DEVICE if any(_is_torch(a) for a in arrays if a is not None) else MEM_HOST
        if mode != self._mode:
            check(self.L.pq_kkt_set_pointer_mode(self.h, mode))
            self._mode = mode
        return mode

    def _out(self, like, size):
        if _is_torch(like):
            import torch
            return torch.empty(size, dtype=torch.float64, device=like.device)
        return np.empty(size, dtype=np.float64)

    def clone(self):
        h = C.c_void_p()
        check(self.L.pq_kkt_clone(self.h, C.byref(h)), "pq_kkt_clone")
        return DenseKKT(None, _h=h)

    def update_data(self, data, options):
        desc = data.descriptor()
        if isinstance(data, SparseData):
            check(self.L.pq_kkt_update_data_sparse(self.h, C.byref(desc), options), "update_data")
        else:
            check(self.L.pq_kkt_update_data_dense(self.h, C.byref(desc), options), "update_data")

    def print_info(self):
        check(self.L.pq_kkt_print_info(self.h))

    def update_scalings_and_factor(self, delta, x_reg, z_reg):
        x_reg, z_reg = _f64(x_reg), _f64(z_reg)
        self._set_mode(x_reg, z_reg)
        return bool(check(self.L.pq_kkt_update_scalings_and_factor(self.h, float(delta), _ptr(x_reg), _ptr(z_reg)), "factor"))

    def solve(self, rhs_x, rhs_y, rhs_z):
        rhs_x, rhs_y, rhs_z = _f64(rhs_x), _f64(rhs_y), _f64(rhs_z)
        self._set_mode(rhs_x, rhs_y, rhs_z)
        lx, ly, lz = self._out(rhs_x, self.n), self._out(rhs_x, self.p), self._out(rhs_x, self.m)
        check(self.L.pq_kkt_solve(self.h, _ptr(rhs_x), _ptr(rhs_y), _ptr(rhs_z), _ptr(lx), _ptr(ly), _ptr(lz)), "solve")
        return lx, ly, lz

    def eval_P_x(self, alpha, x):
        x = _f64(x)
        self._set_mode(x)
        z = self._out(x, self.n)
        check(self.L.pq_kkt_eval_P_x(self.h, float(alpha), _ptr(x), _ptr(z)), "eval_P_x")
        return z

    def eval_A_xn_and_AT_xt(self, alpha_n, alpha_t, xn, xt):
        xn, xt = _f64(xn), _f64(xt)
        self._set_mode(xn, xt)
        zn, zt = self._out(xn, self.p), self._out(xn, self.n)
        check(self.L.pq_kkt_eval_A_xn_and_AT_xt(self.h, float(alpha_n), float(alpha_t), _ptr(xn), _ptr(xt), _ptr(zn), _ptr(zt)))
        return zn, zt

    def eval_G_xn_and_GT_xt(self, alpha_n, alpha_t, xn, xt):
        xn, xt = _f64(xn), _f64(xt)
        self._set_mode(xn, xt)
        zn, zt = self._out(xn, self.m), self._out(xn, self.n)
        check(self.L.pq_kkt_eval_G_xn_and_GT_xt(self.h, float(alpha_n), float(alpha_t), _ptr(xn), _ptr(xt), _ptr(zn), _ptr(zt)))
        return zn, zt

    def synchronize(self):
        check(self.L.pq_kkt_synchronize(self.h))

    def stream(self):
        return self.L.pq_kkt_stream(self.h)

    def set_profiling(self, on=True):
        check(self.L.pq_kkt_set_profiling(self.h, int(on)))

    def get_profile(self, stage):
        ms, cnt = C.c_double(), C.c_int()
        check(self.L.pq_kkt_get_profile(self.h, stage, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def sparse_stats(self):
        """dict of the symbolic-analysis figures (pq_kkt_sparse_stats)"""
        out = (C.c_double * 8)()
        check(self.L.pq_kkt_sparse_stats(self.h, out), "sparse_stats")
        keys = ("N", "nnz_K", "nnz_L", "supernodes", "tree_levels", "subtrees", "max_front", "flops_factor")
        return {k: (float(v) if k == "flops_factor" else int(v)) for k, v in zip(keys, out)}

    def exact_factor(self):
        """reference-order engine only (pq_kkt_exact_factor): dict with L_cols, L_ind, L_vals, D, D_inv, PKPt_val, perm as numpy arrays; raises for another engine"""
        nnz = self.L.pq_kkt_exact_factor(self.h, 0, None)
        check(int(min(nnz, 0)), "exact_factor")
        N = self.n + self.p + self.m
        out = {}
        for key, what, dt, ln in (("L_cols", 1, np.int32, N + 1), ("L_ind", 2, np.int32, nnz), ("L_vals", 3, np.float64, nnz), ("D", 4, np.float64, N), ("D_inv", 5, np.float64, N),
                                  ("perm", 7, np.int32, N)):
            a = np.zeros(max(int(ln), 1), dtype=dt)
            r = self.L.pq_kkt_exact_factor(self.h, what, a.ctypes.data)
            check(int(min(r, 0)), "exact_factor")
            out[key] = a[:int(ln)]
        nk = self.L.pq_kkt_exact_factor(self.h, 6, None)
        a = np.zeros(max(int(nk), 1))
        self.L.pq_kkt_exact_factor(self.h, 6, a.ctypes.data)
        out["PKPt_val"] = a[:int(nk)]
        return out

    def block_info(self):
        """sparse_multistage only: rows of (start, diag_size, off_diag_size); last row = arrow corner block."""
        N = check(self.L.pq_kkt_multistage_block_info(self.h, None, 0), "block_info")
        out = np.zeros((N, 3), dtype=np.int32)
        check(self.L.pq_kkt_multistage_block_info(self.h, out.ctypes.data, N), "block_info")
        return out

    def internal_kkt_mat(self):
        out = np.zeros((self.n, self.n), order="F")
        check(self.L.pq_kkt_internal_kkt_mat(self.h, out.ctypes.data))
        return out

    def internal_factor(self):
        out = np.zeros((self.n, self.n), order="F")
        check(self.L.pq_kkt_internal_factor(self.h, out.ctypes.data))
        return out


def default_settings(**kw):
    s = _lib.Settings()
    _lib.load().pq_settings_default(C.byref(s))
    for k, v in kw.items():
        setattr(s, k, v)
    return s


class KKTSystem(_Handle):
    """piqp::KKTSystem (kkt_system.hpp) behind pq_kktsys_*; Variables are dicts of numpy arrays or torch CUDA tensors."""
    _destroy = "pq_kktsys_destroy"

    def __init__(self, data, settings=None, device=0, _h=None):
        self.L = _lib.load()
        self.settings = settings or default_settings()
        if _h is not None:
            self.h = _h
        else:
            h = C.c_void_p()
            desc = data.descriptor()
            if isinstance(data, SparseData):
                check(self.L.pq_kktsys_create_sparse(C.byref(h), C.byref(desc), C.byref(self.settings), device), "pq_kktsys_create_sparse")
            else:
                check(self.L.pq_kktsys_create_dense(C.byref(h), C.byref(desc), C.byref(self.settings), device), "pq_kktsys_create_dense")
            self.h = h
        self.n, self.p, self.m = (data.n, data.p, data.m) if data is not None else (None, None, None)
        self._mode = MEM_HOST

    def clone(self):
        h = C.c_void_p()
        check(self.L.pq_kktsys_clone(self.h, C.byref(h)))
        k = KKTSystem(None, self.settings, _h=h)
        k.n, k.p, k.m = self.n, self.p, self.m
        return k

    def backend(self):
        return DenseKKT(None, _h=self.L.pq_kktsys_backend(self.h), _owned=False)

    def _set_mode(self, v):
        mode = MEM_DEVICE if any(_is_torch(a) for a in v.values()) else MEM_HOST
        if mode != self._mode:
            check(self.L.pq_kktsys_set_pointer_mode(self.h, mode))
            self._mode = mode

    def update_data(self, data, options):
        desc = data.descriptor()
        if isinstance(data, SparseData):
            check(self.L.pq_kktsys_update_data_sparse(self.h, C.byref(desc), options))
        else:
            check(self.L.pq_kktsys_update_data_dense(self.h, C.byref(desc), options))

    def update_scalings_and_factor(self, iterative_refinement, rho, delta, vars_):
        self._set_mode(vars_)
        vs = Variables.to_struct(vars_)
        return bool(check(self.L.pq_kktsys_update_scalings_and_factor(self.h, int(iterative_refinement), float(rho), float(delta), C.byref(vs))))

    def solve(self, rhs, lhs=None):
        self._set_mode(rhs)
        if lhs is None:
            if self._mode == MEM_DEVICE:
                import torch
                lhs = {k: torch.zeros(sz, dtype=torch.float64, device=rhs["x"].device) for k, sz in var_sizes(self.n, self.p, self.m).items()}
            else:
                lhs = Variables.zeros(self.n, self.p, self.m)
        rs, ls = Variables.to_struct(rhs), Variables.to_struct(lhs)
        ok = bool(check(self.L.pq_kktsys_solve(self.h, C.byref(rs), C.byref(ls))))
        return ok, lhs

    def mul(self, lhs):
        self._set_mode(lhs)
        if self._mode == MEM_DEVICE:
            import torch
            rhs = {k: torch.zeros(sz, dtype=torch.float64, device=lhs["x"].device) for k, sz in var_sizes(self.n, self.p, self.m).items()}
        else:
            rhs = Variables.zeros(self.n, self.p, self.m)
        ls, rs = Variables.to_struct(lhs), Variables.to_struct(rhs)
        check(self.L.pq_kktsys_mul(self.h, C.byref(ls), C.byref(rs)))
        return rhs

    def last_solve_stats(self):
        a, b, c, d = C.c_int(), C.c_int(), C.c_double(), C.c_double()
        self.L.pq_kktsys_last_solve_stats(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        return dict(refine_steps=a.value, backend_solves=b.value, refine_error=c.value, rhs_norm=d.value)

    def condensed_residual(self):
        r, q = C.c_double(), C.c_double()
        check(self.L.pq_kktsys_condensed_residual(self.h, C.byref(r), C.byref(q)))
        return r.value, q.value

    def synchronize(self):
        check(self.L.pq_kktsys_synchronize(self.h))


# results.hpp:18-27
PIQP_SOLVED, PIQP_MAX_ITER_REACHED, PIQP_PRIMAL_INFEASIBLE, PIQP_DUAL_INFEASIBLE = 1, -1, -2, -3
PIQP_NUMERICS, PIQP_UNSOLVED, PIQP_INVALID_SETTINGS = -8, -9, -10


class DenseSolver(_Handle):
    """piqp::DenseSolver<T> (solver.hpp:1262-1291): setup / update / solve / result over pq_solver_*.
    The interior-point loop runs on the host; every KKT factor, solve and mat-vec runs on the GPU."""
    _destroy = "pq_solver_destroy"
    _sparse = False

    def __init__(self, device=0, _h=None):
        self.L = _lib.load()
        if _h is not None:
            self.h = _h
        else:
            h = C.c_void_p()
            check(self.L.pq_solver_create(C.byref(h), device), "pq_solver_create")
            self.h = h
        self._trace = None

    @property
    def settings(self):
        return self.L.pq_solver_settings(self.h).contents

    def clone(self):
        h = C.c_void_p()
        check(self.L.pq_solver_clone(self.h, C.byref(h)))
        return type(self)(_h=h)

    @staticmethod
    def _col(a):
        return None if a is None else np.asfortranarray(a, dtype=np.float64)

    @staticmethod
    def _vec(a):
        return None if a is None else np.ascontiguousarray(a, dtype=np.float64)

    def setup(self, P, c, A=None, b=None, G=None, h_l=None, h_u=None, x_l=None, x_u=None):
        P = self._col(P)
        n = P.shape[0]
        p = 0 if A is None else np.asarray(A).shape[0]
        m = 0 if G is None else np.asarray(G).shape[0]
        keep = [P, self._vec(c), self._col(A), self._vec(b), self._col(G), self._vec(h_l), self._vec(h_u), self._vec(x_l), self._vec(x_u)]
        return bool(check(self.L.pq_solver_setup_dense(self.h, n, p, m, *[_ptr(a) for a in keep]), "setup"))

    def update(self, P=None, c=None, A=None, b=None, G=None, h_l=None, h_u=None, x_l=None, x_u=None):
        keep = [self._col(P), self._vec(c), self._col(A), self._vec(b), self._col(G), self._vec(h_l), self._vec(h_u), self._vec(x_l), self._vec(x_u)]
        return bool(check(self.L.pq_solver_update_dense(self.h, *[_ptr(a) for a in keep]), "update"))

    def enable_trace(self, max_rows=512):
        self._trace = np.zeros((max_rows, 11))
        check(self.L.pq_solver_set_trace(self.h, self._trace.ctypes.data, max_rows))

    def trace(self):
        return self._trace[: self.L.pq_solver_trace_rows(self.h)].copy()

    def solve(self):
        return self.L.pq_solver_solve(self.h)

    @property
    def info(self):
        return self.L.pq_solver_info(self.h).contents

    def result(self):
        n, p, m = C.c_int(), C.c_int(), C.c_int()
        check(self.L.pq_solver_dims(self.h, C.byref(n), C.byref(p), C.byref(m)))
        out = Variables.zeros(n.value, p.value, m.value)
        vs = Variables.to_struct(out)
        check(self.L.pq_solver_get_result(self.h, C.byref(vs)))
        return out


class SparseSolver(DenseSolver):
    """piqp::SparseSolver<T,I> (solver.hpp:1293-1322): CSC inputs."""
    _sparse = True

    @staticmethod
    def _csc(M):
        import scipy.sparse as sp
        if M is None:
            return [None, None, None]
        M = sp.csc_matrix(M)
        M.sort_indices()
        return [np.ascontiguousarray(M.indptr, dtype=np.int32), np.ascontiguousarray(M.indices, dtype=np.int32), np.ascontiguousarray(M.data, dtype=np.float64)]

    def setup(self, P, c, A=None, b=None, G=None, h_l=None, h_u=None, x_l=None, x_u=None):
        n = P.shape[0]
        p = 0 if A is None else A.shape[0]
        m = 0 if G is None else G.shape[0]
        keep = self._csc(P) + [self._vec(c)] + self._csc(A) + [self._vec(b)] + self._csc(G) + [self._vec(h_l), self._vec(h_u), self._vec(x_l), self._vec(x_u)]
        return bool(check(self.L.pq_solver_setup_sparse(self.h, n, p, m, *[_ptr(a) for a in keep]), "setup"))

    def update(self, P=None, c=None, A=None, b=None, G=None, h_l=None, h_u=None, x_l=None, x_u=None):
        keep = self._csc(P) + [self._vec(c)] + self._csc(A) + [self._vec(b)] + self._csc(G) + [self._vec(h_l), self._vec(h_u), self._vec(x_l), self._vec(x_u)]
        return bool(check(self.L.pq_solver_update_sparse(self.h, *[_ptr(a) for a in keep]), "update"))
