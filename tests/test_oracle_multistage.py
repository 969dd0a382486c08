"""Pins the oracle's `sparse_multistage` backend to the reference's own tests.  CPU only.

  docs/assets/robust_scenario_mpc.ipynb   print_info() output of the recorded run: block sizes 8,6 8,6 8,6 14,0 (x3),
                                          arrow width 8  -> extract_arrow_structure KAT (multistage_kkt.hpp:420-597)
  sparse/multistage_kkt_test.cpp:24-98    test_solve_multiply: multistage solve == sparse_ldlt solve, mul == mul (1e-8)
  sparse/multistage_kkt_test.cpp:100-172  UpdateData: update_data + refactor keeps agreeing
  sparse/multistage_kkt_test.cpp:174-211  FactorizeSolveSQP over the eight .mat fixtures
  sparse/solver_test.cpp                  same statuses / answers through the multistage backend
"""
import numpy as np
import pytest
import scipy.sparse as sp

from qp_gen import dense_strongly_convex_qp, random_vars
from qp_io import load_json, load_qp
from test_oracle_sparse import _sparse_args, _sparsify

FIXTURES = ["qp_small_sparse_dual_inf", "qp_small_dense", "qp_scenario_mpc_small", "qp_scenario_mpc", "qp_chain_mass_sqp",
            "qp_robot_arm_sqp", "qp_robot_arm_sqp_constr_perm", "qp_robot_arm_sqp_no_global"]


def _unit_scaling(orc, n, p, m):
    return orc.make_vars(n, p, m, fill=1.0)


def _check_solve_multiply(orc, d, k1, k2, seed, tol=1e-8):
    """multistage_kkt_test.cpp:24-98"""
    n, p, m = d.n, d.p, d.m
    rhs = random_vars(n, p, m, np.random.default_rng(seed))
    ok1, l1 = k1.solve(rhs)
    ok2, l2 = k2.solve(rhs)
    assert ok1 and ok2
    nhl, nhu, nxl, nxu = d.counts()
    hl, hu = d.idx("h_l"), d.idx("h_u")

    def cmp(a, b):
        scale = max(1.0, np.abs(b["x"]).max())
        assert np.allclose(a["x"], b["x"], rtol=tol, atol=tol * scale)
        assert np.allclose(a["y"], b["y"], rtol=tol, atol=tol * scale)
        for key, cnt in (("z_bl", nxl), ("z_bu", nxu), ("s_bl", nxl), ("s_bu", nxu)):
            assert np.allclose(a[key][:cnt], b[key][:cnt], rtol=tol, atol=tol * scale), key
        for key, idx in (("z_l", hl), ("s_l", hl), ("z_u", hu), ("s_u", hu)):
            assert np.allclose(a[key][idx], b[key][idx], rtol=tol, atol=tol * scale), key

    cmp(l1, l2)
    cmp(k1.mul(l1), k2.mul(l2))


def test_notebook_block_structure_known_answer(orc):
    """The only recorded output of extract_arrow_structure in the reference tree."""
    q = load_qp("qp_c0_scenario_mpc")
    tr = load_json("c0_trace.json")
    d = orc.Data.sparse(*_sparse_args(q))
    k = orc.KKT(d, kind="multistage")
    bi = k.block_info()
    assert [[int(r[1]), int(r[2])] for r in bi[:-1]] == tr["multistage_block_info"]
    assert int(bi[-1][1]) == tr["multistage_arrow_width"]
    # blocks tile the variables, arrow at the end
    assert int(bi[:, 1].sum()) == d.n and np.array_equal(bi[:, 0], np.concatenate([[0], np.cumsum(bi[:-1, 1])]))


@pytest.mark.parametrize("name", FIXTURES)
def test_factorize_solve_fixtures_match_sparse_ldlt(orc, name):
    """FactorizeSolveSQP, multistage_kkt_test.cpp:174-211 (rho = 0.9, delta = 1.2, unit scalings)"""
    q = load_qp(name)
    d = orc.Data.sparse(*_sparse_args(q))
    kms = orc.KKTSystem(d, orc.Settings(kkt_solver=orc.SPARSE_MULTISTAGE))
    ksp = orc.KKTSystem(d, orc.Settings(kkt_solver=orc.SPARSE_LDLT))
    sc = _unit_scaling(orc, d.n, d.p, d.m)
    assert kms.update_scalings_and_factor(False, 0.9, 1.2, sc) and ksp.update_scalings_and_factor(False, 0.9, 1.2, sc)
    _check_solve_multiply(orc, d, kms, ksp, seed=11)


def test_update_data_matches_sparse_ldlt(orc):
    """UpdateData, multistage_kkt_test.cpp:100-172"""
    n, p, m = 10, 8, 9
    q = _sparsify(dense_strongly_convex_qp(n, p, m, seed=21), 0.2, 4)
    q["x_l"] = np.full(n, -np.inf); q["x_u"] = np.full(n, np.inf)
    d = orc.Data.sparse(*_sparse_args(q))
    kms = orc.KKTSystem(d, orc.Settings(kkt_solver=orc.SPARSE_MULTISTAGE))
    ksp = orc.KKTSystem(d, orc.Settings(kkt_solver=orc.SPARSE_LDLT))
    sc = _unit_scaling(orc, n, p, m)
    assert kms.update_scalings_and_factor(False, 0.9, 1.2, sc) and ksp.update_scalings_and_factor(False, 0.9, 1.2, sc)
    _check_solve_multiply(orc, d, kms, ksp, seed=1)
    # new values on the same pattern; keep P positive semi-definite by shifting its diagonal
    rng = np.random.default_rng(9)
    dC = d.ptr.contents
    for cs in (dC.sP_utri, dC.sAT, dC.sGT):
        for i in range(cs.colptr[cs.cols]):
            cs.val[i] = rng.standard_normal()
    U = dC.sP_utri
    Pd = np.zeros((n, n))
    for j in range(n):
        for qq in range(U.colptr[j], U.colptr[j + 1]):
            Pd[U.rowind[qq], j] = U.val[qq]
    Pf = Pd + np.triu(Pd, 1).T
    mn = np.linalg.eigvalsh(Pf).min()
    if mn < 0:
        for j in range(n):
            for qq in range(U.colptr[j], U.colptr[j + 1]):
                if U.rowind[qq] == j:
                    U.val[qq] -= mn
    opts = orc.KKT_UPDATE_P | orc.KKT_UPDATE_A | orc.KKT_UPDATE_G
    kms.update_data(opts); ksp.update_data(opts)
    assert kms.update_scalings_and_factor(False, 0.9, 1.2, sc) and ksp.update_scalings_and_factor(False, 0.9, 1.2, sc)
    _check_solve_multiply(orc, d, kms, ksp, seed=2)


def test_row_permutation_and_empty_rows(orc):
    """transpose_to_block_mat (:672-818): rows are grouped by the block of their first column, empty rows go last."""
    n = 12
    rng = np.random.default_rng(0)
    P = sp.csc_matrix(np.triu(np.eye(n) + np.diag(0.1 * np.ones(n - 1), 1)))
    A = np.zeros((5, n))
    A[0, 7] = 1; A[0, 8] = 2      # late block
    A[1, 0] = 1; A[1, 1] = -1     # first block
    # row 2 stays empty
    A[3, 3] = 1; A[3, 4] = 1
    A[4, 0] = 3
    G = np.zeros((2, n)); G[0, 2] = 1; G[1, 10] = 1; G[1, 11] = 1
    d = orc.Data.sparse(P, rng.standard_normal(n), sp.csc_matrix(A), np.zeros(5), sp.csc_matrix(G), -np.ones(2), np.ones(2), None, None)
    k = orc.KKT(d, kind="multistage")
    bi = k.block_info()
    perm, sizes = k.row_perm(0)
    assert sorted(perm.tolist()) == list(range(5))
    assert perm[2] == 4                      # the empty row is last
    assert int(sizes.sum()) == 4
    # block of a row = block whose diagonal range holds its first column (capped at the last non-arrow block)
    starts, diags = bi[:-1, 0], bi[:-1, 1]
    acc = np.concatenate([[0], np.cumsum(sizes)])

    def blk(j):
        b = 0
        while starts[b] + diags[b] <= j and b + 1 < len(starts):
            b += 1
        return b
    for r, first in ((0, 7), (1, 0), (3, 3), (4, 0)):
        b = blk(first)
        assert acc[b] <= perm[r] < acc[b + 1]
    # and the backend still agrees with sparse_ldlt with an empty constraint row present
    kms = orc.KKTSystem(d, orc.Settings(kkt_solver=orc.SPARSE_MULTISTAGE))
    ksp = orc.KKTSystem(d, orc.Settings(kkt_solver=orc.SPARSE_LDLT))
    sc = _unit_scaling(orc, n, 5, 2)
    assert kms.update_scalings_and_factor(False, 0.9, 1.2, sc) and ksp.update_scalings_and_factor(False, 0.9, 1.2, sc)
    _check_solve_multiply(orc, d, kms, ksp, seed=3)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_random_structures_match_sparse_ldlt(orc, seed):
    """random banded + arrow structures: whatever block layout the heuristic picks, the algebra must agree"""
    rng = np.random.default_rng(100 + seed)
    n, p, m = 40 + 7 * seed, 20, 15
    bw = 2 + seed
    M = np.zeros((n, n))
    for i in range(n):
        for j in range(i, min(n, i + bw + 1)):
            M[i, j] = rng.standard_normal()
    if seed % 2:  # coupling to the last variables -> arrow
        M[:, n - 2:] += rng.standard_normal((n, 2)) * (rng.random((n, 2)) < 0.5)
    Pf = np.triu(M, 1); Pf = Pf + Pf.T
    Pf += (1e-2 + abs(np.linalg.eigvalsh(Pf).min())) * np.eye(n)
    A = np.zeros((p, n)); G = np.zeros((m, n))
    for r in range(p):
        j = rng.integers(0, n - bw); A[r, j:j + bw + 1] = rng.standard_normal(bw + 1)
    for r in range(m):
        j = rng.integers(0, n - bw); G[r, j:j + 2] = rng.standard_normal(2)
        if seed % 2 and r % 3 == 0:
            G[r, n - 1] = 1.0
    d = orc.Data.sparse(sp.csc_matrix(np.triu(Pf)), rng.standard_normal(n), sp.csc_matrix(A), rng.standard_normal(p),
                        sp.csc_matrix(G), -np.ones(m), np.ones(m), -np.ones(n), np.full(n, np.inf))
    kms = orc.KKTSystem(d, orc.Settings(kkt_solver=orc.SPARSE_MULTISTAGE))
    ksp = orc.KKTSystem(d, orc.Settings(kkt_solver=orc.SPARSE_LDLT))
    st = random_vars(n, p, m, rng, positive=True)
    assert kms.update_scalings_and_factor(False, 1e-3, 1e-2, st) and ksp.update_scalings_and_factor(False, 1e-3, 1e-2, st)
    _check_solve_multiply(orc, d, kms, ksp, seed=seed, tol=1e-7)
    # eval_* of the block containers against scipy
    k = kms.backend()
    x, y, z = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(m)
    assert np.allclose(k.eval_P_x(0.7, x), 0.7 * Pf @ x, atol=1e-12)
    zn, zt = k.eval_A_xn_and_AT_xt(1.3, -0.4, x, y)
    assert np.allclose(zn, 1.3 * A @ x, atol=1e-12) and np.allclose(zt, -0.4 * A.T @ y, atol=1e-12)
    zn, zt = k.eval_G_xn_and_GT_xt(-2.0, 0.5, x, z)
    assert np.allclose(zn, -2.0 * G @ x, atol=1e-12) and np.allclose(zt, 0.5 * G.T @ z, atol=1e-12)


def test_notebook_qp_solves_with_multistage(orc):
    """C0: the notebook reports the same 12 iterations / optimum for the multistage backend"""
    q = load_qp("qp_c0_scenario_mpc")
    tr = load_json("c0_trace.json")
    s = orc.Solver()
    s.settings.kkt_solver = orc.SPARSE_MULTISTAGE
    assert s.setup(*_sparse_args(q), sparse=True)
    assert s.solve() == orc.SOLVED
    assert s.info.iter == tr["iterations"]
    assert abs(s.info.primal_obj - tr["objective_scipy_trust_constr"]) < 1e-3


@pytest.mark.parametrize("name", ["qp_small_sparse_dual_inf", "qp_small_dense", "qp_scenario_mpc_small", "qp_scenario_mpc", "qp_chain_mass_sqp"])
def test_fixtures_solve_same_iterations_as_sparse_ldlt(orc, name):
    q = load_qp(name)
    res = []
    for ks in (orc.SPARSE_LDLT, orc.SPARSE_MULTISTAGE):
        s = orc.Solver()
        s.settings.kkt_solver = ks
        assert s.setup(*_sparse_args(q), sparse=True)
        res.append((s.solve(), s.info.iter, s.info.primal_obj))
    assert res[0][0] == res[1][0]
    assert res[0][1] == res[1][1]
    if res[0][0] == orc.SOLVED:
        assert abs(res[0][2] - res[1][2]) <= 1e-6 * max(1.0, abs(res[0][2]))
