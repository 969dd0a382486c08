"""Batched sparse solver over pq_batch_*: many structurally identical QPs, one workgroup per QP, one kernel launch.

Equivalent of looping the reference's `SparseSolver::setup(...); solve();` (solver.hpp:1293-1322) over the instances with
kkt_solver = sparse_multistage.  Patterns are shared; values are stacked along a leading batch axis.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import VAR_NAMES, check
from .kkt import _Handle, _ptr


class BatchSparseSolver(_Handle):
    _destroy = "pq_batch_destroy"

    def __init__(self, device=0):
        self.L = _lib.load()
        h = C.c_void_p()
        check(self.L.pq_batch_create(C.byref(h), device), "pq_batch_create")
        self.h = h
        self.batch = self.n = self.p = self.m = 0

    @property
    def settings(self):
        return self.L.pq_batch_settings(self.h).contents

    @staticmethod
    def _pattern(M):
        import scipy.sparse as sp
        if M is None:
            return None, None, None
        M = sp.csc_matrix(M)
        M.sort_indices()
        return np.ascontiguousarray(M.indptr, dtype=np.int32), np.ascontiguousarray(M.indices, dtype=np.int32), M

    @staticmethod
    def _stack(a, batch, length):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.shape == (batch, length), (a.shape, (batch, length))
        return a

    def setup(self, P_pattern, P_values, c, A_pattern=None, A_values=None, b=None, G_pattern=None, G_values=None, h_l=None, h_u=None, x_l=None, x_u=None):
        """*_pattern: scipy sparse matrices giving the (sorted CSC) patterns of P (n x n), A (p x n), G (m x n);
        *_values: [batch, nnz] arrays in that CSC order; vectors: [batch, len]."""
        Pp, Pi, Pm = self._pattern(P_pattern)
        Ap, Ai, Am = self._pattern(A_pattern)
        Gp, Gi, Gm = self._pattern(G_pattern)
        n = Pm.shape[0]
        p = 0 if Am is None else Am.shape[0]
        m = 0 if Gm is None else Gm.shape[0]
        batch = np.asarray(c).shape[0]
        keep = [Pp, Pi, self._stack(P_values, batch, Pm.nnz), self._stack(c, batch, n),
                Ap, Ai, None if Am is None else self._stack(A_values, batch, Am.nnz), None if Am is None else self._stack(b, batch, p),
                Gp, Gi, None if Gm is None else self._stack(G_values, batch, Gm.nnz),
                None if Gm is None else self._stack(h_l, batch, m), None if Gm is None else self._stack(h_u, batch, m),
                self._stack(x_l, batch, n), self._stack(x_u, batch, n)]
        ok = bool(check(self.L.pq_batch_setup_sparse(self.h, batch, n, p, m, *[_ptr(a) for a in keep]), "pq_batch_setup_sparse"))
        self.batch, self.n, self.p, self.m = batch, n, p, m
        return ok

    def update(self, c=None, b=None, h_l=None, h_u=None, x_l=None, x_u=None):
        """new vectors for every instance ([batch, len] arrays, None = unchanged); matrices and the set of finite bounds stay"""
        keep = [self._stack(c, self.batch, self.n), self._stack(b, self.batch, self.p), self._stack(h_l, self.batch, self.m), self._stack(h_u, self.batch, self.m),
                self._stack(x_l, self.batch, self.n), self._stack(x_u, self.batch, self.n)]
        return bool(check(self.L.pq_batch_update(self.h, *[_ptr(a) for a in keep]), "pq_batch_update"))

    def update_data(self, P_values=None, A_values=None, G_values=None, c=None, b=None, h_l=None, h_u=None, x_l=None, x_u=None):
        """new matrix values ([batch, nnz] in the CSC order of the setup patterns) and / or vectors for every instance, None = unchanged:
        unscale -> assign -> (fresh) Ruiz equilibration on the device, solver.hpp:218-308"""
        st = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
        for a in (P_values, A_values, G_values):
            assert a is None or np.asarray(a).shape[0] == self.batch
        keep = [st(P_values), st(A_values), st(G_values), self._stack(c, self.batch, self.n), self._stack(b, self.batch, self.p), self._stack(h_l, self.batch, self.m),
                self._stack(h_u, self.batch, self.m), self._stack(x_l, self.batch, self.n), self._stack(x_u, self.batch, self.n)]
        return bool(check(self.L.pq_batch_update_data(self.h, *[_ptr(a) for a in keep]), "pq_batch_update_data"))

    def solve(self):
        """returns the number of instances that ended SOLVED"""
        return check(self.L.pq_batch_solve(self.h), "pq_batch_solve")

    def info(self, i):
        ptr = self.L.pq_batch_info(self.h, i)
        if not ptr:
            raise IndexError(i)
        return ptr.contents

    def statuses(self):
        return np.array([self.info(i).status for i in range(self.batch)])

    def iterations(self):
        return np.array([self.info(i).iter for i in range(self.batch)])

    def result(self, name):
        k = VAR_NAMES.index(name)
        length = {"x": self.n, "y": self.p, "z_bl": self.n, "z_bu": self.n, "s_bl": self.n, "s_bu": self.n}.get(name, self.m)
        out = np.zeros((self.batch, length))
        if length:
            check(self.L.pq_batch_get_result(self.h, k, out.ctypes.data), "pq_batch_get_result")
        return out

    def block_info(self):
        N = check(self.L.pq_batch_block_info(self.h, None, 0))
        out = np.zeros((N, 3), dtype=np.int32)
        check(self.L.pq_batch_block_info(self.h, out.ctypes.data, N))
        return out

    def profile(self, i):
        """device-clock seconds of instance i: dict(assemble, factor, chain_solve, kkt_solve, residuals, total)"""
        out = np.zeros(8)
        check(self.L.pq_batch_get_profile(self.h, i, out.ctypes.data))
        return dict(zip(("assemble", "factor", "chain_solve", "kkt_solve", "residuals", "total"), out[:6]))

    def set_start_order(self, longest_first=True):
        """True (default): the instances that needed most iterations in the previous solve start first in the next launch; False: index order"""
        check(self.L.pq_batch_set_start_order(self.h, 1 if longest_first else 0), "pq_batch_set_start_order")

    def last_kernel_ms(self):
        ms, nt = C.c_double(), C.c_int()
        check(self.L.pq_batch_last_kernel_ms(self.h, C.byref(ms), C.byref(nt)))
        return ms.value, nt.value
