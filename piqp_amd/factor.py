"""Host-side mirror of the reference's dense factorisation classes as stand-alone objects: piqp::dense::LDLTNoPivot<Mat, UpLo>
(/root/reference/include/piqp/dense/ldlt_no_pivot.hpp:87-262) and the Eigen::LLT<Mat, UpLo> that dense/kkt.hpp:82 uses -- same member names, same meaning of
info(); the work is done by the HIP kernels behind pq_dense_factor_* (include/piqp_amd.h).  What tests/src/dense/ldlt_test.cpp and
benchmarks/src/dense_cholesky_factorization_benchmark.cpp use."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check
from .kkt import DENSE_CHOLESKY, DENSE_LDLT_NO_PIVOT, MEM_DEVICE, MEM_HOST, _Handle, _is_torch, _ptr

LOWER, UPPER = 1, 2       # Eigen::Lower, Eigen::Upper
SUCCESS, NUMERICAL_ISSUE = 0, 1  # Eigen::ComputationInfo


class _DenseFactor(_Handle):
    _destroy = "pq_dense_factor_destroy"
    _kind = DENSE_LDLT_NO_PIVOT

    def __init__(self, n, uplo=LOWER, device=0):
        self.L = _lib.load()
        self.n, self.uplo = int(n), int(uplo)
        h = C.c_void_p()
        check(self.L.pq_dense_factor_create(C.byref(h), device, self.n, self._kind, self.uplo), "pq_dense_factor_create")
        self.h = h

    def compute(self, A):
        """A: n x n numpy array (any layout; only the `uplo` triangle is read).  Device-resident matrices: compute_colmajor"""
        Af = np.asfortranarray(A, dtype=np.float64)
        assert Af.shape == (self.n, self.n)
        self._info = check(self.L.pq_dense_factor_compute(self.h, Af.ctypes.data, self.n, MEM_HOST), "pq_dense_factor_compute")
        return self

    def compute_colmajor(self, ptr_or_tensor, lda=None, on_device=True):
        """the column-major matrix at a raw address / in a torch tensor's storage, leading dimension lda (default n); no copy through the host"""
        p = ptr_or_tensor if isinstance(ptr_or_tensor, int) else _ptr(ptr_or_tensor)
        self._keep = ptr_or_tensor
        self._info = check(self.L.pq_dense_factor_compute(self.h, p, self.n if lda is None else int(lda), MEM_DEVICE if on_device else MEM_HOST), "pq_dense_factor_compute")
        return self

    def info(self):
        return check(self.L.pq_dense_factor_info(self.h), "pq_dense_factor_info")

    def solveInPlace(self, x):
        if _is_torch(x):
            check(self.L.pq_dense_factor_solve_in_place(self.h, x.data_ptr(), MEM_DEVICE), "pq_dense_factor_solve_in_place")
            return x
        assert x.dtype == np.float64 and x.flags.c_contiguous and x.shape == (self.n,)
        check(self.L.pq_dense_factor_solve_in_place(self.h, x.ctypes.data, MEM_HOST), "pq_dense_factor_solve_in_place")
        return x

    def solve(self, b):
        x = np.array(b, dtype=np.float64)
        return self.solveInPlace(x)

    def _matrix(self):
        out = np.zeros((self.n, self.n), order="F")
        check(self.L.pq_dense_factor_matrix(self.h, out.ctypes.data, self.n), "pq_dense_factor_matrix")
        return out

    def last_ms(self):
        """(device time of the factorisation launches, wall time of the whole compute()) of the last compute(), ms"""
        o = np.zeros(2)
        check(self.L.pq_dense_factor_last_ms(self.h, o.ctypes.data), "pq_dense_factor_last_ms")
        return float(o[0]), float(o[1])


class LDLTNoPivot(_DenseFactor):
    """piqp::dense::LDLTNoPivot<Mat, UpLo>: A = L D L^T = U^T D U without pivoting"""
    _kind = DENSE_LDLT_NO_PIVOT

    def matrixLDLT(self):
        """ldlt_no_pivot.hpp:217: the `uplo` triangle holds the strictly triangular part of the unit factor and D on the diagonal (the other triangle: zeros here)"""
        return self._matrix()

    def vectorD(self):
        return np.diag(self._matrix()).copy()

    def matrixL(self):
        m = self._matrix()
        m = m if self.uplo == LOWER else m.T
        return np.tril(m, -1) + np.eye(self.n)

    def matrixU(self):
        return self.matrixL().T

    def reconstructedMatrix(self):
        Lm = self.matrixL()
        return (Lm * self.vectorD()[None, :]) @ Lm.T


class LLT(_DenseFactor):
    """Eigen::LLT<Mat, UpLo> as dense/kkt.hpp:82 uses it: A = L L^T = U^T U"""
    _kind = DENSE_CHOLESKY

    def matrixLLT(self):
        return self._matrix()

    def matrixL(self):
        m = self._matrix()
        return np.tril(m if self.uplo == LOWER else m.T)

    def matrixU(self):
        return self.matrixL().T

    def reconstructedMatrix(self):
        Lm = self.matrixL()
        return Lm @ Lm.T
