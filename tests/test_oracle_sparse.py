"""Pins the oracle's sparse path (sparse_ldlt, KKT_FULL) to the reference's own tests.  CPU only.

  sparse/utils_test.cpp:55-92    AMD 4x4: ordering [1 2 0 3], exact permuted matrix and Ai_to_Ci = 3 0 2 1 5 4 6
  sparse/ldlt_test.cpp:22-79     LDLt numeric returns n == dim; b ~ P_full x (1e-8)
  sparse/kkt_test.cpp:40-162     update == fresh (PKPt upper-triangular), FactorizeSolve: K lhs ~ rhs (1e-8)
  sparse/solver_test.cpp         known answers / statuses, through the sparse backend
  docs notebook                  recorded sparse_ldlt trace (12 iterations) -- this is the backend it was recorded with
"""
import numpy as np
import pytest
import scipy.sparse as sp

from qp_gen import dense_strongly_convex_qp, random_vars
from qp_io import load_json, load_qp


def _sparse_args(q):
    return (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])


def _sparsify(q, density, seed):
    rng = np.random.default_rng(seed)
    out = dict(q)
    n = q["P"].shape[0]
    P = np.triu(q["P"], 1) * (rng.random((n, n)) < density)
    P = P + P.T
    P += (1e-2 + abs(np.linalg.eigvalsh(P).min())) * np.eye(n)
    out["P"] = sp.csc_matrix(np.triu(P))
    for k in ("A", "G"):
        if q[k] is not None:
            M = q[k] * (rng.random(q[k].shape) < density)
            M[np.arange(M.shape[0]), rng.integers(0, n, M.shape[0])] = 1.0  # no empty rows
            out[k] = sp.csc_matrix(M)
    return out


def test_amd_4x4_known_answer(orc):
    kat = load_json("kat_small.json")["amd_4x4"]
    L = orc.lib()
    A = sp.csc_matrix(np.array([[1, 0, 2, 3], [0, 4, 0, 5], [0, 0, 6, 0], [0, 0, 0, 7.0]]))
    A.sort_indices()
    ip, ii, ax = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    perm = np.zeros(4, np.int32)
    L.orc_amd_order(4, ip.ctypes.data_as(orc._ip), ii.ctypes.data_as(orc._ip), perm.ctypes.data_as(orc._ip))
    assert list(perm) == kat["ordering"]
    pinv = np.zeros(4, np.int32); pinv[perm] = np.arange(4)
    Cp, Ci, Cx, m = np.zeros(5, np.int32), np.zeros(7, np.int32), np.zeros(7), np.zeros(7, np.int32)
    L.orc_permute_sym_upper(4, ip.ctypes.data_as(orc._ip), ii.ctypes.data_as(orc._ip), ax.ctypes.data_as(orc._dp), pinv.ctypes.data_as(orc._ip),
                            Cp.ctypes.data_as(orc._ip), Ci.ctypes.data_as(orc._ip), Cx.ctypes.data_as(orc._dp), m.ctypes.data_as(orc._ip))
    assert list(m) == kat["Ai_to_Ci"]
    C = sp.csc_matrix((Cx, Ci, Cp), shape=(4, 4)).toarray()
    expect = np.zeros((4, 4))
    for (i, j, v) in [(0, 0, 4), (0, 3, 5), (1, 1, 6), (1, 2, 2), (2, 2, 1), (2, 3, 3), (3, 3, 7)]:
        expect[i, j] = v
    assert np.array_equal(C, expect)
    x = np.array([1.0, 2, 3, 4])
    assert list(x[perm]) == [2, 3, 1, 4]  # ordering.perm (utils_test.cpp:84-88)


@pytest.mark.parametrize("n,density", [(10, 0.5), (60, 0.1), (400, 0.02), (1500, 0.004)])
def test_amd_is_valid_and_reduces_fill(orc, n, density):
    L = orc.lib()
    rng = np.random.default_rng(n)
    M = sp.random(n, n, density=density, random_state=rng, format="csc")
    S = M + M.T + sp.eye(n) * (n + 1.0)
    U = sp.triu(S, format="csc"); U.sort_indices()
    ip, ii = U.indptr.astype(np.int32), U.indices.astype(np.int32)
    perm = np.zeros(n, np.int32)
    L.orc_amd_order(n, ip.ctypes.data_as(orc._ip), ii.ctypes.data_as(orc._ip), perm.ctypes.data_as(orc._ip))
    assert sorted(perm) == list(range(n))

    def fill(pm):
        f = L.orc_sparse_ldlt_create()
        Sp = sp.triu(S[pm][:, pm], format="csc"); Sp.sort_indices()
        a, b = Sp.indptr.astype(np.int32), Sp.indices.astype(np.int32)
        L.orc_sparse_ldlt_symbolic(f, n, a.ctypes.data_as(orc._ip), b.ctypes.data_as(orc._ip))
        nz = L.orc_sparse_ldlt_nnz(f)
        L.orc_sparse_ldlt_free(f)
        return nz
    nat, amd = fill(np.arange(n)), fill(perm)
    assert amd <= nat
    if n >= 400:
        assert amd < 0.7 * nat  # a fill-reducing ordering, not just a permutation


@pytest.mark.parametrize("n,density", [(10, 0.5), (80, 0.1), (300, 0.03)])
def test_sparse_ldlt_residual(orc, n, density):
    """sparse/ldlt_test.cpp:22-79 (quasi-definite matrices are fine: D may be negative)"""
    L = orc.lib()
    rng = np.random.default_rng(7 + n)
    M = sp.random(n, n, density=density, random_state=rng, format="csc")
    S = (M + M.T).toarray()
    S += (1e-2 + abs(np.linalg.eigvalsh(S).min())) * np.eye(n)
    S[n // 2:, n // 2:] *= -1.0  # make it quasi-definite-ish but still strongly factorisable
    S = np.triu(S) + np.triu(S, 1).T
    U = sp.csc_matrix(np.triu(S)); U.sort_indices()
    ip, ii, ax = U.indptr.astype(np.int32), U.indices.astype(np.int32), U.data.astype(np.float64)
    f = L.orc_sparse_ldlt_create()
    L.orc_sparse_ldlt_symbolic(f, n, ip.ctypes.data_as(orc._ip), ii.ctypes.data_as(orc._ip))
    assert L.orc_sparse_ldlt_numeric(f, n, ip.ctypes.data_as(orc._ip), ii.ctypes.data_as(orc._ip), ax.ctypes.data_as(orc._dp)) == n
    b = rng.standard_normal(n)
    x = b.copy()
    L.orc_sparse_ldlt_solve_inplace(f, x.ctypes.data_as(orc._dp))
    assert np.allclose(S @ x, b, rtol=1e-8, atol=1e-8)
    L.orc_sparse_ldlt_free(f)


def test_sparse_kkt_factorize_solve_and_dense_agreement(orc):
    """sparse/kkt_test.cpp:88-162 FactorizeSolve (KKT_FULL) + agreement with the dense backend on the same QP"""
    n, p, m = 20, 8, 9
    qd = dense_strongly_convex_qp(n, p, m, seed=3)
    qs = _sparsify(qd, 0.4, 1)
    dense_q = dict(qs); dense_q["P"] = qs["P"].toarray(); dense_q["A"] = qs["A"].toarray(); dense_q["G"] = qs["G"].toarray()
    ds = orc.Data.sparse(**qs)
    dd = orc.Data.dense(**dense_q)
    ks = orc.KKTSystem(ds, orc.Settings(kkt_solver=orc.SPARSE_LDLT))
    kd = orc.KKTSystem(dd)
    scaling = orc.make_vars(n, p, m, fill=1.0)
    assert ks.update_scalings_and_factor(False, 0.9, 1.2, scaling) and kd.update_scalings_and_factor(False, 0.9, 1.2, scaling)
    rhs = random_vars(n, p, m, np.random.default_rng(0))
    ok, lhs = ks.solve(rhs)
    okd, lhd = kd.solve(rhs)
    assert ok and okd
    back = ks.mul(lhs)
    nhl, nhu, nxl, nxu = ds.counts()
    assert np.allclose(rhs["x"], back["x"], atol=1e-8) and np.allclose(rhs["y"], back["y"], atol=1e-8)
    for key, cnt in (("z_bl", nxl), ("z_bu", nxu), ("s_bl", nxl), ("s_bu", nxu)):
        assert np.allclose(rhs[key][:cnt], back[key][:cnt], atol=1e-8)
    for key in lhs:
        cnt = {"z_bl": nxl, "s_bl": nxl, "z_bu": nxu, "s_bu": nxu}.get(key, len(lhs[key]))
        assert np.allclose(lhs[key][:cnt], lhd[key][:cnt], rtol=1e-9, atol=1e-9), key


def test_sparse_kkt_update_data_equals_fresh(orc):
    """sparse/kkt_test.cpp:40-86: update_data + refactor == fresh; PKPt stays upper triangular"""
    n, p, m = 10, 8, 9
    q1 = _sparsify(dense_strongly_convex_qp(n, p, m, seed=1), 0.5, 2)
    d = orc.Data.sparse(**q1)
    k = orc.KKT(d, kind="sparse", mode=0)
    x_reg, z_reg = np.full(n, 0.9), np.full(m, 2.2)
    assert k.update_scalings_and_factor(1.2, x_reg, z_reg)
    # new values, same pattern
    rng = np.random.default_rng(5)
    dC = d.ptr.contents
    for cs in (dC.sP_utri, dC.sAT, dC.sGT):
        nnz = cs.colptr[cs.cols]
        for i in range(nnz):
            cs.val[i] = cs.val[i] * (1.0 + 0.1 * rng.standard_normal())
    k.update_data(orc.KKT_UPDATE_P | orc.KKT_UPDATE_A | orc.KKT_UPDATE_G)
    assert k.update_scalings_and_factor(1.2, x_reg, z_reg)
    k2 = orc.KKT(d, kind="sparse", mode=0)
    assert k2.update_scalings_and_factor(1.2, x_reg, z_reg)
    L = orc.lib()
    N = L.orc_sparse_kkt_dim(k.ptr)
    cp = np.ctypeslib.as_array(L.orc_sparse_kkt_PKPt_colptr(k.ptr), shape=(N + 1,))
    nnz = cp[N]
    ri = np.ctypeslib.as_array(L.orc_sparse_kkt_PKPt_rowind(k.ptr), shape=(nnz,))
    v1 = np.ctypeslib.as_array(L.orc_sparse_kkt_PKPt_val(k.ptr), shape=(nnz,))
    v2 = np.ctypeslib.as_array(L.orc_sparse_kkt_PKPt_val(k2.ptr), shape=(nnz,))
    assert np.array_equal(v1, v2)
    for j in range(N):
        assert np.all(ri[cp[j]:cp[j + 1]] <= j) and ri[cp[j + 1] - 1] == j  # upper triangular, diagonal last


def test_recorded_notebook_trace_sparse_ldlt(orc):
    """the notebook trace was produced with sparse_ldlt: reproduce it with the oracle's sparse backend"""
    q = load_qp("qp_c0_scenario_mpc")
    tr = load_json("c0_trace.json")
    s = orc.Solver()
    s.settings.kkt_solver = orc.SPARSE_LDLT
    s.enable_trace()
    assert s.setup(*_sparse_args(q), sparse=True)
    assert s.solve() == orc.SOLVED
    assert s.info.iter == tr["iterations"]
    assert abs(s.info.primal_obj - tr["objective_scipy_trust_constr"]) < 1e-3
    t, ref = s.trace(), np.array(tr["rows"])
    for col, rtol in ((1, 2e-6), (2, 2e-5), (3, 5e-3), (4, 2e-5), (6, 1e-3), (7, 1e-3), (8, 1e-3)):
        assert np.allclose(t[:, col], ref[:, col], rtol=rtol, atol=1e-12), col
    assert np.allclose(t[:9, 5], ref[:9, 5], rtol=1e-4)


@pytest.mark.parametrize("name", ["qp_small_sparse_dual_inf", "qp_small_dense", "qp_scenario_mpc_small", "qp_scenario_mpc", "qp_chain_mass_sqp", "qp_robot_arm_sqp",
                                  "mm_HS21", "mm_HS35", "mm_DUAL1", "mm_QAFIRO", "mm_CVXQP1_S", "mm_AUG3D", "mm_LOTSCHD", "mm_PRIMALC1", "mm_QSCAGR7"])
def test_fixtures_sparse_backend(orc, name):
    q = load_qp(name)
    s = orc.Solver()
    s.settings.kkt_solver = orc.SPARSE_LDLT
    assert s.setup(*_sparse_args(q), sparse=True)
    st = s.solve()
    if name == "qp_small_sparse_dual_inf":
        assert st == orc.DUAL_INFEASIBLE  # tests/src/sparse/solver_test.cpp
    else:
        assert st == orc.SOLVED


MODES = [(0, "SPARSE_LDLT"), (1, "SPARSE_LDLT_EQ_COND"), (2, "SPARSE_LDLT_INEQ_COND"), (3, "SPARSE_LDLT_COND")]


@pytest.mark.parametrize("mode,ks", MODES)
def test_sparse_kkt_modes_factorize_solve(orc, mode, ks):
    """sparse/kkt_test.cpp:88-162 is typed over all four KKTModes: K lhs ~ rhs (1e-8) through KKTSystem, and every condensed
    mode must agree with KKT_FULL on the same QP"""
    n, p, m = 20, 8, 9
    qs = _sparsify(dense_strongly_convex_qp(n, p, m, seed=3), 0.4, 1)
    d = orc.Data.sparse(**qs)
    ksys = orc.KKTSystem(d, orc.Settings(kkt_solver=getattr(orc, ks)))
    kfull = orc.KKTSystem(d, orc.Settings(kkt_solver=orc.SPARSE_LDLT))
    scaling = orc.make_vars(n, p, m, fill=1.0)
    assert ksys.update_scalings_and_factor(False, 0.9, 1.2, scaling) and kfull.update_scalings_and_factor(False, 0.9, 1.2, scaling)
    rhs = random_vars(n, p, m, np.random.default_rng(0))
    ok, lhs = ksys.solve(rhs)
    okf, lhf = kfull.solve(rhs)
    assert ok and okf
    back = ksys.mul(lhs)
    nhl, nhu, nxl, nxu = d.counts()
    assert np.allclose(rhs["x"], back["x"], atol=1e-8) and np.allclose(rhs["y"], back["y"], atol=1e-8)
    for key, cnt in (("z_bl", nxl), ("z_bu", nxu), ("s_bl", nxl), ("s_bu", nxu)):
        assert np.allclose(rhs[key][:cnt], back[key][:cnt], atol=1e-8)
    for key in lhs:
        cnt = {"z_bl": nxl, "s_bl": nxl, "z_bu": nxu, "s_bu": nxu}.get(key, len(lhs[key]))
        assert np.allclose(lhs[key][:cnt], lhf[key][:cnt], rtol=1e-9, atol=1e-9), key


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_sparse_kkt_modes_update_data_equals_fresh(orc, mode):
    """sparse/kkt_test.cpp:40-86 for the condensed modes: new values on the same pattern, update + refactor == fresh"""
    n, p, m = 10, 8, 9
    q1 = _sparsify(dense_strongly_convex_qp(n, p, m, seed=1), 0.5, 2)
    d = orc.Data.sparse(**q1)
    k = orc.KKT(d, kind="sparse", mode=mode)
    x_reg, z_reg = np.full(n, 0.9), np.full(m, 2.2)
    assert k.update_scalings_and_factor(1.2, x_reg, z_reg)
    rng = np.random.default_rng(5)
    dC = d.ptr.contents
    for cs in (dC.sP_utri, dC.sAT, dC.sGT):
        for i in range(cs.colptr[cs.cols]):
            cs.val[i] = cs.val[i] * (1.0 + 0.1 * rng.standard_normal())
    k.update_data(orc.KKT_UPDATE_P | orc.KKT_UPDATE_A | orc.KKT_UPDATE_G)
    assert k.update_scalings_and_factor(1.2, x_reg, z_reg)
    k2 = orc.KKT(d, kind="sparse", mode=mode)
    assert k2.update_scalings_and_factor(1.2, x_reg, z_reg)
    r = [rng.standard_normal(s) for s in (n, p, m)]
    for a, b in zip(k.solve(*r), k2.solve(*r)):
        assert np.array_equal(a, b)
    L = orc.lib()
    N = L.orc_sparse_kkt_dim(k.ptr) if mode == 0 else None
    assert N is None


@pytest.mark.parametrize("name", ["qp_small_dense", "qp_scenario_mpc_small", "qp_scenario_mpc", "mm_HS21", "mm_DUAL1", "mm_QAFIRO", "mm_CVXQP1_S", "mm_LOTSCHD"])
@pytest.mark.parametrize("ks", ["SPARSE_LDLT_EQ_COND", "SPARSE_LDLT_INEQ_COND", "SPARSE_LDLT_COND"])
def test_fixtures_condensed_backends(orc, name, ks):
    q = load_qp(name)
    s = orc.Solver()
    s.settings.kkt_solver = getattr(orc, ks)
    assert s.setup(*_sparse_args(q), sparse=True)
    assert s.solve() == orc.SOLVED
    s0 = orc.Solver(); s0.settings.kkt_solver = orc.SPARSE_LDLT
    assert s0.setup(*_sparse_args(q), sparse=True)
    s0.solve()
    assert abs(s.info.primal_obj - s0.info.primal_obj) <= 1e-5 * (1 + abs(s0.info.primal_obj))


def _sweep_names(prefix, skip=()):
    import glob
    import os
    from qp_io import GOLDEN
    return sorted(n for n in (os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, prefix + "*.npz"))) if n not in skip)


# the reference's own sweeps as a pin of the ORACLE: tests/src/sparse/maros_meszaros_tests.cpp expects PIQP_SOLVED on every
# Maros-Meszaros file, netlib_lp_tests.cpp (infeasibility_threshold = 0.01) SOLVED on data/ and PRIMAL / DUAL INFEASIBLE on infeas/.
# mm_CONT-201 / mm_BOYD1 (10 s and 1 s of CPU) run in the GPU suite next to the device.
ORACLE_MISSES_REFERENCE = {"nl_bnl2", "nl_pilot-we", "nli_ceria3d", "nli_cplex2", "nli_qual"}  # MAX_ITER in the oracle: degenerate LPs, see test_mm_real_gpu.py


@pytest.mark.parametrize("name", _sweep_names("mm_", skip=("mm_CONT-201", "mm_BOYD1")))
def test_oracle_meets_the_maros_meszaros_contract(orc, name):
    from qp_io import load_qp
    q = load_qp(name)
    so = orc.Solver(); so.settings.kkt_solver = orc.SPARSE_LDLT
    assert so.setup(q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"], sparse=True)
    assert so.solve() == orc.SOLVED


@pytest.mark.parametrize("name", _sweep_names("nl_") + _sweep_names("nli_"))
def test_oracle_meets_the_netlib_contract(orc, name):
    from qp_io import load_qp
    q = load_qp(name)
    so = orc.Solver(); so.settings.kkt_solver = orc.SPARSE_LDLT; so.settings.infeasibility_threshold = 0.01
    assert so.setup(q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"], sparse=True)
    st = so.solve()
    expected = (orc.SOLVED,) if name.startswith("nl_") else (orc.PRIMAL_INFEASIBLE, orc.DUAL_INFEASIBLE)
    if name in ORACLE_MISSES_REFERENCE:
        assert st == orc.MAX_ITER_REACHED  # recorded deviation of the restatement from the reference's expectation (5 of 103)
    else:
        assert st in expected, (name, st)


# Which iteration counts are a property of the ALGORITHM and which of the rounding: the same oracle sources built with FMA contraction (oracle/Makefile target
# `fma`: gcc's default at -O3 -march=native, the flags the reference documents) against the build the tests pin (no contraction in the sparse LDLt, as
# the reference forces with its "force compiler to not use fma" temporaries, sparse/ldlt.hpp:151-159).  The device parity tests (tests/test_mm_real_gpu.py)
# hold a problem to the oracle's count exactly where the two builds agree, and to the range they span where they do not.
FMA_SENSITIVE = {"mm_QBEACONF": (17, 18), "mm_QCAPRI": (50, 35), "mm_QPILOTNO": (35, 62), "mm_QSHIP08S": (21, 19), "nl_fffff800": (43, 39)}
FMA_STABLE = ["mm_HS21", "mm_HS118", "mm_DUAL1", "mm_CVXQP1_S", "mm_LOTSCHD", "mm_QAFIRO", "mm_AUG3DCQP", "mm_CONT-050", "mm_QETAMACR", "mm_QGROW7", "mm_QSHIP08L", "nl_afiro",
              "nl_sc105", "nl_share2b", "nl_stocfor1", "qp_scenario_mpc", "qp_chain_mass_sqp"]


def _solve_with(orc, L, name):
    q = load_qp(name)
    so = orc.Solver(_L=L); so.settings.kkt_solver = orc.SPARSE_LDLT
    if name.startswith("nl"):
        so.settings.infeasibility_threshold = 0.01
    assert so.setup(q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"], sparse=True)
    return so.solve(), so.info.iter


@pytest.mark.parametrize("name", sorted(FMA_SENSITIVE))
def test_counts_decided_by_rounding_differ_between_the_two_oracle_builds(orc, name):
    a, b = _solve_with(orc, None, name), _solve_with(orc, orc.lib_fma(), name)
    assert a[0] == b[0] == orc.SOLVED
    assert (a[1], b[1]) == FMA_SENSITIVE[name], (name, a, b)


@pytest.mark.parametrize("name", FMA_STABLE)
def test_counts_of_well_posed_problems_do_not_depend_on_fma_contraction(orc, name):
    a, b = _solve_with(orc, None, name), _solve_with(orc, orc.lib_fma(), name)
    assert a == b, (name, a, b)
