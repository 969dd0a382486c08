"""The 128 x 128 diagonal-block factorisation kernel (csrc/dense_kernels.hip potrf_block: the unblocked part of Eigen::LLT, dense/kkt.hpp:82, and of
dense/ldlt_no_pivot.hpp:278-311) on its own, through pq_debug_potrf_block: factor, reciprocal pivots, D and the operand pack of the panel solve against numpy in
extended precision, for full and short blocks, well and badly conditioned, definite and (LDLT) indefinite; failure reports; and bitwise repeatability -- the kernel is a
dataflow of eight waves without barriers, so every output is also compared bit for bit over many repetitions."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-12  # relative to the largest entry of the reference output (fp64, blocks with condition numbers up to 1e8: see _spd)


def _run(hip, A, ldlt, nb, reps=40):
    L = hip._lib.load()
    A128 = np.zeros((128, 128), order="F")
    A128[:nb, :nb] = A
    out = np.zeros((128, 128), order="F")
    rdiag = np.zeros(128); dvec = np.zeros(128); pack = np.zeros(36 * 256)
    info = C.c_int(0); differ = C.c_int(-1)
    hip._lib.check(L.pq_debug_potrf_block(0, int(ldlt), nb, reps, A128.ctypes.data, out.ctypes.data, rdiag.ctypes.data, dvec.ctypes.data, pack.ctypes.data,
                                          C.byref(info), C.byref(differ)))
    return out, rdiag, dvec, pack, info.value, differ.value


def _spd(nb, seed, cond=1e3):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((nb, nb)))
    ev = np.geomspace(1.0, cond, nb)
    A = (Q * ev) @ Q.T
    return 0.5 * (A + A.T)


def _ldl_ref(A):
    """unit-lower U and D with A = U D U^T, no pivoting, in extended precision"""
    n = A.shape[0]
    W = A.astype(np.longdouble).copy()
    U = np.eye(n, dtype=np.longdouble); d = np.zeros(n, dtype=np.longdouble)
    for c in range(n):
        d[c] = W[c, c]
        U[c + 1:, c] = W[c + 1:, c] / d[c]
        W[c + 1:, c + 1:] -= np.outer(U[c + 1:, c], W[c + 1:, c])
    return U, d


def _rel(a, b):
    return float(np.abs(np.asarray(a, dtype=np.longdouble) - b).max() / np.abs(b).max())


def _pack_blocks(pack):
    return pack.reshape(36, 16, 16).transpose(0, 2, 1)  # block b as a [row][column] array (the image is column-major)


def _check(hip, A, ldlt, nb, tol=TOL):
    out, rdiag, dvec, pack, info, differ = _run(hip, A, ldlt, nb)
    assert info == -1
    assert differ == 0, "the outputs of repeated factorisations differ"
    U, d = _ldl_ref(A)
    if ldlt:
        Lref = np.tril(U, -1) + np.diag(d)
        rref = 1.0 / d
        assert _rel(dvec[:nb], d) < tol
    else:
        Lref = U * np.sqrt(d)[None, :]
        rref = 1.0 / np.sqrt(d)
    assert _rel(np.tril(out[:nb, :nb]), np.tril(Lref)) < tol
    assert _rel(rdiag[:nb], rref) < tol
    if nb == 128:
        # the pack: -L(j, k) (LDLT: -U) below the diagonal, the inverse of the diagonal piece (LDLT: of its unit part) on it
        B = _pack_blocks(pack)
        Lp = np.tril(U, -1) + np.eye(nb) if ldlt else Lref
        scale = float(np.abs(Lp).max())
        for j in range(8):
            for k in range(j):
                assert np.abs(B[j * (j - 1) // 2 + k] + Lp[16 * j:16 * j + 16, 16 * k:16 * k + 16]).max() < tol * scale
            Wref = np.linalg.inv(Lp[16 * j:16 * j + 16, 16 * j:16 * j + 16].astype(np.float64))
            assert np.abs(B[28 + j] - Wref).max() < 1e-9 * np.abs(Wref).max()
    return out


@pytest.mark.parametrize("ldlt", [0, 1])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_full_block_against_numpy(hip, ldlt, seed):
    _check(hip, _spd(128, seed), ldlt, 128)


@pytest.mark.parametrize("ldlt", [0, 1])
@pytest.mark.parametrize("nb", [1, 2, 15, 16, 17, 33, 64, 100, 127])
def test_short_blocks_are_identity_padded(hip, ldlt, nb):
    _check(hip, _spd(nb, 100 + nb), ldlt, nb)


@pytest.mark.parametrize("ldlt", [0, 1])
def test_ill_conditioned_block(hip, ldlt):
    # pivots over eight orders of magnitude, like a KKT block late in the interior-point iteration: the bound scales with the conditioning
    _check(hip, _spd(128, 7, cond=1e8), ldlt, 128, tol=1e-9)


def test_indefinite_block_through_ldlt(hip):
    # quasi-definite like the sparse fronts' blocks (sparse/kkt.hpp: -delta on the dual part): LDLT without pivoting takes negative pivots as they come
    rng = np.random.default_rng(5)
    n, p = 128, 48
    H = _spd(n - p, 11, cond=50.0)
    B = 0.3 * rng.standard_normal((p, n - p))
    A = np.block([[H, B.T], [B, -0.7 * np.eye(p)]])
    out = _check(hip, A, 1, 128, tol=1e-11)
    assert (np.diag(out)[n - p:] < 0).all()


@pytest.mark.parametrize("col", [0, 5, 16, 77, 127])
def test_llt_reports_the_first_nonpositive_pivot(hip, col):
    A = _spd(128, 3)
    U, d = _ldl_ref(A)
    d = np.asarray(d, dtype=np.float64).copy(); d[col] = -abs(d[col])
    Uf = np.asarray(U, dtype=np.float64)
    A2 = (Uf * d) @ Uf.T
    A2 = 0.5 * (A2 + A2.T)
    _, _, _, _, info, _ = _run(hip, A2, 0, 128, reps=3)
    assert info == col


def test_ldlt_reports_an_exact_zero_pivot(hip):
    A = np.diag(np.arange(1.0, 129.0)); A[40, 40] = 0.0
    _, _, _, _, info, _ = _run(hip, A, 1, 128, reps=3)
    assert info == 40
