"""The reference's dense factorisation CLASSES as device objects: piqp::dense::LDLTNoPivot<Mat, Eigen::Lower / Eigen::Upper>
(/root/reference/include/piqp/dense/ldlt_no_pivot.hpp:87-262, Upper = the transposed view, :357-371) and the Eigen::LLT of dense/kkt.hpp:82, through
pq_dense_factor_* (include/piqp_amd.h) and its host mirror piqp_amd.LDLTNoPivot / piqp_amd.LLT.

Mirrors /root/reference/tests/src/dense/ldlt_test.cpp (SolveLower, SolveUpper: compute twice, info() == Success, b.isApprox(P_full * x, 1e-8)) and holds the factor
to the oracle's restatement of the same class (oracle/orc_dense.c: orc_ldlt_no_pivot_compute, orc_llt_compute), entry by entry, at the sizes of
benchmarks/src/dense_cholesky_factorization_benchmark.cpp:104-109 (4 ... 1024) and some that are not multiples of anything."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def upper_triangular_spd(n, seed, shift=1e-2):
    """a symmetric positive definite matrix of which only the UPPER triangle is kept (rand::dense_positive_definite_upper_triangular_rand, utils/random_utils.hpp)"""
    rng = np.random.default_rng(seed)
    U = np.triu(rng.standard_normal((n, n)), 1)
    S = U + U.T
    S += (shift + abs(np.linalg.eigvalsh(S).min())) * np.eye(n)
    return np.triu(S), S


def is_approx(a, b, prec):
    """Eigen's isApprox: |a - b| <= prec * min(|a|, |b|) in the Euclidean norm"""
    return np.linalg.norm(a - b) <= prec * min(np.linalg.norm(a), np.linalg.norm(b))


@pytest.mark.parametrize("uplo", ["lower", "upper"])
def test_ldlt_test_cpp_solve(hip, uplo):
    """ldlt_test.cpp:22-49 (SolveLower) / :51-77 (SolveUpper), dim = 50"""
    dim = 50
    P, P_full = upper_triangular_spd(dim, 7)
    if uplo == "lower":
        P = P.T.copy()  # P.transposeInPlace()
    ldlt = hip.LDLTNoPivot(dim, hip.LOWER if uplo == "lower" else hip.UPPER)
    ldlt.compute(P)
    assert ldlt.info() == 0  # Eigen::Success
    ldlt.compute(P)
    assert ldlt.info() == 0
    b = np.random.default_rng(8).standard_normal(dim)
    x = b.copy()
    ldlt.solveInPlace(x)
    assert is_approx(b, P_full @ x, 1e-8)


SIZES = [4, 8, 16, 31, 32, 50, 64, 127, 128, 129, 200, 256, 300, 384, 512, 1000, 1024]


@pytest.mark.parametrize("n", SIZES)
def test_ldlt_no_pivot_against_the_oracle_both_triangles(hip, orc, n):
    P_up, S = upper_triangular_spd(n, 100 + n)
    a = np.asfortranarray(S.copy()); w = np.zeros(n)
    assert orc.lib().orc_ldlt_no_pivot_compute(a.ctypes.data_as(orc._dp), n, n, w.ctypes.data_as(orc._dp)) == -1
    Lo, Do = np.tril(a, -1) + np.eye(n), np.diag(a).copy()
    lo = hip.LDLTNoPivot(n, hip.LOWER).compute(P_up.T.copy())
    up = hip.LDLTNoPivot(n, hip.UPPER).compute(P_up)
    assert lo.info() == 0 and up.info() == 0
    # the two triangles: the same numbers, transposed (ldlt_no_pivot.hpp:357-371 factors the transposed view with the Lower code)
    assert np.array_equal(lo.matrixLDLT(), up.matrixLDLT().T)
    assert np.array_equal(np.triu(lo.matrixLDLT(), 1), np.zeros((n, n))) and np.array_equal(np.tril(up.matrixLDLT(), -1), np.zeros((n, n)))
    Lh, Dh = lo.matrixL(), lo.vectorD()
    scale = np.abs(Lo).max()
    assert np.abs(Lh - Lo).max() <= 1e-10 * scale and np.abs(Dh - Do).max() <= 1e-10 * np.abs(Do).max()
    assert np.abs(lo.reconstructedMatrix() - S).max() <= 1e-12 * np.abs(S).max() * n
    assert np.array_equal(up.matrixU(), lo.matrixL().T)
    b = np.random.default_rng(n).standard_normal(n)
    xo = b.copy()
    orc.lib().orc_ldlt_no_pivot_solve_inplace(a.ctypes.data_as(orc._dp), n, n, xo.ctypes.data_as(orc._dp))
    xl, xu = lo.solve(b), up.solve(b)
    assert np.array_equal(xl, xu)
    assert is_approx(b, S @ xl, 1e-8) and np.abs(xl - xo).max() <= 1e-8 * np.abs(xo).max()


@pytest.mark.parametrize("n", SIZES)
def test_llt_against_the_oracle_both_triangles(hip, orc, n):
    P_up, S = upper_triangular_spd(n, 300 + n)
    a = np.asfortranarray(S.copy())
    assert orc.lib().orc_llt_compute(a.ctypes.data_as(orc._dp), n, n) == -1
    Lo = np.tril(a)
    lo = hip.LLT(n, hip.LOWER).compute(P_up.T.copy())
    up = hip.LLT(n, hip.UPPER).compute(P_up)
    assert lo.info() == 0 and up.info() == 0
    assert np.array_equal(lo.matrixLLT(), up.matrixLLT().T)
    assert np.abs(lo.matrixL() - Lo).max() <= 1e-10 * np.abs(Lo).max()
    b = np.random.default_rng(n).standard_normal(n)
    assert is_approx(b, S @ up.solve(b), 1e-8)


def test_only_the_named_triangle_is_read(hip):
    """compute() of a matrix whose other triangle holds garbage (Eigen reads m_matrix's UpLo part only, ldlt_no_pivot.hpp:408-411)"""
    n = 96
    P_up, S = upper_triangular_spd(n, 5)
    junk = np.random.default_rng(1).standard_normal((n, n)) * 1e6
    up_junk = P_up + np.tril(junk, -1)
    lo_junk = P_up.T + np.triu(junk, 1)
    clean = hip.LDLTNoPivot(n, hip.UPPER).compute(P_up).matrixLDLT()
    assert np.array_equal(hip.LDLTNoPivot(n, hip.UPPER).compute(up_junk).matrixLDLT(), clean)
    assert np.array_equal(hip.LDLTNoPivot(n, hip.LOWER).compute(lo_junk).matrixLDLT(), clean.T)


def test_info_like_the_classes(hip, orc):
    """LDLTNoPivot fails on an exact zero pivot only (ldlt_no_pivot.hpp:307) -- a quasi-definite matrix is factored, its negative pivots kept; Eigen::LLT fails on the
    first pivot that is not positive.  (The dense KKT BACKEND built on the same kernels gives up on a negative pivot in both cases: tests/test_dense_gpu.py.)"""
    S = np.array([[2.0, 1.0], [1.0, -3.0]])
    f = hip.LDLTNoPivot(2).compute(S)
    assert f.info() == 0 and f.vectorD()[1] < 0
    a = np.asfortranarray(S.copy()); w = np.zeros(2)
    assert orc.lib().orc_ldlt_no_pivot_compute(a.ctypes.data_as(orc._dp), 2, 2, w.ctypes.data_as(orc._dp)) == -1
    assert np.allclose(f.matrixLDLT(), np.tril(a), rtol=1e-15)
    Z = np.array([[0.0, 1.0], [1.0, 1.0]])
    assert hip.LDLTNoPivot(2).compute(Z).info() == 1  # Eigen::NumericalIssue
    assert hip.LLT(2).compute(S).info() == 1
    # a larger quasi-definite matrix (the KKT shape LDLTNoPivot exists for): [[H, A'], [A, -I]]
    n, p = 150, 60
    rng = np.random.default_rng(3)
    H = upper_triangular_spd(n, 9)[1]
    A = rng.standard_normal((p, n))
    K = np.block([[H, A.T], [A, -np.eye(p)]])
    f = hip.LDLTNoPivot(n + p).compute(K)
    assert f.info() == 0
    d = f.vectorD()
    assert (d[:n] > 0).all() and (d[n:] < 0).all()
    b = rng.standard_normal(n + p)
    assert is_approx(b, K @ f.solve(b), 1e-8)
    assert hip.LLT(n + p).compute(K).info() == 1
    # recompute on the same object after a failure
    ok = hip.LDLTNoPivot(2)
    assert ok.compute(Z).info() == 1 and ok.compute(S).info() == 0


def test_device_resident_input_with_a_leading_dimension(hip):
    import torch
    n, lda = 200, 256
    P_up, S = upper_triangular_spd(n, 11)
    buf = np.full((lda, n), 7.5, order="F")
    buf[:n, :] = P_up
    t = torch.from_numpy(np.ascontiguousarray(buf.T)).cuda()  # row-major [n][lda] = column-major lda x n
    f = hip.LDLTNoPivot(n, hip.UPPER).compute_colmajor(t, lda=lda)
    assert f.info() == 0
    assert np.array_equal(f.matrixLDLT(), hip.LDLTNoPivot(n, hip.UPPER).compute(P_up).matrixLDLT())
    x = torch.from_numpy(np.random.default_rng(0).standard_normal(n)).cuda()
    b = x.cpu().numpy().copy()
    f.solveInPlace(x)
    assert is_approx(b, S @ x.cpu().numpy(), 1e-8)
    dev_ms, wall_ms = f.last_ms()
    assert 0.0 < dev_ms <= wall_ms
