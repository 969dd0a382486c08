"""GPU parity tests of the `sparse_multistage` KKT backend (device block-tridiagonal-arrow Cholesky chain) against the
CPU oracle's restatement of the reference's MultistageKKT.  Mirrors tests/src/sparse/multistage_kkt_test.cpp
(test_solve_multiply, UpdateData, FactorizeSolveSQP over the eight .mat fixtures) and the notebook's recorded
block structure.  Integer results (block layout) are compared exactly; floating-point results to the north-star
tolerance: relative KKT residual <= 1e-10 and identical IPM iteration counts."""
import numpy as np
import pytest
import scipy.sparse as sp

from qp_gen import dense_strongly_convex_qp, random_vars
from qp_io import load_json, load_qp
from test_sparse_gpu import _args, _rel, _sparsify

pytestmark = pytest.mark.gpu

FIXTURES = ["qp_small_sparse_dual_inf", "qp_small_dense", "qp_scenario_mpc_small", "qp_scenario_mpc", "qp_chain_mass_sqp",
            "qp_robot_arm_sqp", "qp_robot_arm_sqp_constr_perm", "qp_robot_arm_sqp_no_global"]


def _dense3(q, n, p, m):
    Pu = q["P"].toarray() if sp.issparse(q["P"]) else np.asarray(q["P"])
    Pu = np.triu(Pu)
    Pf = Pu + np.triu(Pu, 1).T
    A = q["A"].toarray() if p else np.zeros((0, n))
    G = q["G"].toarray() if m else np.zeros((0, n))
    return Pf, A, G


@pytest.mark.parametrize("name", ["qp_c0_scenario_mpc"] + FIXTURES)
def test_block_structure_identical_to_oracle(hip, orc, name):
    """extract_arrow_structure (multistage_kkt.hpp:420-597): integer output, must be bit-exact"""
    q = load_qp(name)
    k = hip.SparseKKT(hip.SparseData(*_args(q)), kkt_solver=hip.SPARSE_MULTISTAGE)
    ko = orc.KKT(orc.Data.sparse(*_args(q)), kind="multistage")
    assert np.array_equal(k.block_info(), ko.block_info())
    if name == "qp_c0_scenario_mpc":
        tr = load_json("c0_trace.json")
        bi = k.block_info()
        assert [[int(r[1]), int(r[2])] for r in bi[:-1]] == tr["multistage_block_info"] and int(bi[-1][1]) == tr["multistage_arrow_width"]


@pytest.mark.parametrize("name", FIXTURES)
def test_backend_factor_solve_fixtures(hip, orc, name):
    """FactorizeSolveSQP (multistage_kkt_test.cpp:174-211) at backend level: rho-like x_reg, delta = 1.2, z_reg from unit scalings"""
    q = load_qp(name)
    d = hip.SparseData(*_args(q)); od = orc.Data.sparse(*_args(q))
    n, p, m = od.n, od.p, od.m
    k = hip.SparseKKT(d, kkt_solver=hip.SPARSE_MULTISTAGE)
    ko = orc.KKT(od, kind="multistage")
    rng = np.random.default_rng(5)
    x_reg = rng.uniform(0.5, 2.0, n); z_reg = rng.uniform(0.1, 3.0, m); delta = 1.2
    assert k.update_scalings_and_factor(delta, x_reg, z_reg) and ko.update_scalings_and_factor(delta, x_reg, z_reg)
    rx, ry, rz = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(m)
    lx, ly, lz = k.solve(rx, ry, rz)
    ox, oy, oz = ko.solve(rx, ry, rz)
    assert _rel(lx, ox) < 1e-9 and _rel(ly, oy) < 1e-9 and _rel(lz, oz) < 1e-9
    # residual of the 3x3 system the backend solves (Ruiz-free data: the fixtures' own matrices)
    Pf, A, G = _dense3(q, n, p, m)
    r1 = rx - (Pf @ lx + x_reg * lx + A.T @ ly + G.T @ lz)
    r2 = ry - (A @ lx - delta * ly)
    r3 = rz - (G @ lx - z_reg * lz)
    nrm = max([np.abs(v).max() for v in (rx, ry, rz) if v.size])
    scale = max(1.0, np.abs(lx).max())
    assert max([np.abs(v).max() for v in (r1, r2, r3) if v.size]) <= 1e-10 * nrm * scale * max(1.0, np.abs(Pf).max())
    # eval_* (block_symv_l / block_t_gemv_*)
    x = rng.standard_normal(n); y = rng.standard_normal(p); z = rng.standard_normal(m)
    assert np.allclose(k.eval_P_x(-1.5, x), -1.5 * Pf @ x, rtol=1e-12, atol=1e-12 * max(1, np.abs(Pf).max()))
    zn, zt = k.eval_A_xn_and_AT_xt(-1.0, 2.0, x, y)
    assert np.allclose(zn, -A @ x, atol=1e-11 * max(1, np.abs(A).max() if p else 1)) and np.allclose(zt, 2.0 * A.T @ y, atol=1e-11 * max(1, np.abs(A).max() if p else 1))
    zn, zt = k.eval_G_xn_and_GT_xt(0.5, -3.0, x, z)
    assert np.allclose(zn, 0.5 * G @ x, atol=1e-11 * max(1, np.abs(G).max() if m else 1)) and np.allclose(zt, -3.0 * G.T @ z, atol=1e-11 * max(1, np.abs(G).max() if m else 1))


@pytest.mark.parametrize("name", FIXTURES)
def test_kkt_system_matches_sparse_ldlt_like_reference_test(hip, orc, name):
    """test_solve_multiply (multistage_kkt_test.cpp:24-98): multistage == sparse_ldlt through KKTSystem, both on the device,
    and the device multistage path against the oracle's; rho = 0.9, delta = 1.2, unit scalings"""
    q = load_qp(name)
    d = hip.SparseData(*_args(q)); od = orc.Data.sparse(*_args(q))
    n, p, m = od.n, od.p, od.m
    kms = hip.KKTSystem(d, hip.default_settings(kkt_solver=hip.SPARSE_MULTISTAGE))
    ksp = hip.KKTSystem(d, hip.default_settings(kkt_solver=hip.SPARSE_LDLT))
    kom = orc.KKTSystem(od, orc.Settings(kkt_solver=orc.SPARSE_MULTISTAGE))
    sc = hip.Variables.zeros(n, p, m, fill=1.0)
    assert kms.update_scalings_and_factor(False, 0.9, 1.2, sc) and ksp.update_scalings_and_factor(False, 0.9, 1.2, sc)
    assert kom.update_scalings_and_factor(False, 0.9, 1.2, sc)
    rhs = random_vars(n, p, m, np.random.default_rng(11))
    ok1, l1 = kms.solve(rhs); ok2, l2 = ksp.solve(rhs); ok3, l3 = kom.solve(rhs)
    assert ok1 and ok2 and ok3
    res, nrm = kms.condensed_residual()
    assert res <= 1e-10 * nrm
    hl, hu = od.idx("h_l"), od.idx("h_u")
    nxl, nxu = d.n_x_l, d.n_x_u
    scale = max(1.0, np.abs(l3["x"]).max())
    for other in (l2, l3):
        for key in ("x", "y"):
            assert np.allclose(l1[key], other[key], rtol=1e-8, atol=1e-8 * scale), key
        for key, cnt in (("z_bl", nxl), ("z_bu", nxu), ("s_bl", nxl), ("s_bu", nxu)):
            assert np.allclose(l1[key][:cnt], other[key][:cnt], rtol=1e-8, atol=1e-8 * scale), key
        for key, idx in (("z_l", hl), ("s_l", hl), ("z_u", hu), ("s_u", hu)):
            assert np.allclose(l1[key][idx], other[key][idx], rtol=1e-8, atol=1e-8 * scale), key
    b1, b3 = kms.mul(l1), kom.mul(l3)
    for key in ("x", "y"):
        assert np.allclose(b1[key], b3[key], rtol=1e-8, atol=1e-8 * scale), key


def test_update_data_matches_fresh(hip, orc):
    """UpdateData (multistage_kkt_test.cpp:100-172): new values on the same pattern; update == fresh bitwise, and == oracle"""
    n, p, m = 30, 12, 15
    q1 = _sparsify(dense_strongly_convex_qp(n, p, m, seed=1), 0.2, 2)
    d = hip.SparseData(*_args(q1))
    k = hip.SparseKKT(d, kkt_solver=hip.SPARSE_MULTISTAGE)
    x_reg, z_reg = np.full(n, 0.9), np.full(m, 2.2)
    assert k.update_scalings_and_factor(1.2, x_reg, z_reg)
    rng = np.random.default_rng(5)
    q2 = dict(q1)
    for key in ("P", "A", "G"):
        M = q1[key].copy(); M.data = M.data * (1.0 + 0.1 * rng.standard_normal(M.data.size)); q2[key] = M
    Pd = q2["P"].toarray(); Pd = Pd + np.triu(Pd, 1).T
    mn = np.linalg.eigvalsh(Pd).min()
    if mn < 0.01:
        q2["P"] = sp.csc_matrix(q2["P"] + sp.eye(n) * (0.01 - mn))
        assert q2["P"].nnz == q1["P"].nnz
    d2 = hip.SparseData(*_args(q2))
    k.update_data(d2, hip.KKT_UPDATE_P | hip.KKT_UPDATE_A | hip.KKT_UPDATE_G)
    assert k.update_scalings_and_factor(1.2, x_reg, z_reg)
    k2 = hip.SparseKKT(d2, kkt_solver=hip.SPARSE_MULTISTAGE)
    assert k2.update_scalings_and_factor(1.2, x_reg, z_reg)
    r = [rng.standard_normal(s) for s in (n, p, m)]
    a, b = k.solve(*r), k2.solve(*r)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
    ko = orc.KKT(orc.Data.sparse(*_args(q2)), kind="multistage")
    assert ko.update_scalings_and_factor(1.2, x_reg, z_reg)
    for u, v in zip(a, ko.solve(*r)):
        assert _rel(u, v) < 1e-9


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4])
def test_random_banded_arrow_structures(hip, orc, seed):
    """band + optional arrow + empty constraint rows: whatever layout the heuristic picks, device == oracle"""
    rng = np.random.default_rng(100 + seed)
    n, p, m = 40 + 7 * seed, 20, 15
    bw = 2 + seed
    M = np.zeros((n, n))
    for i in range(n):
        for j in range(i, min(n, i + bw + 1)):
            M[i, j] = rng.standard_normal()
    if seed % 2:
        M[:, n - 2:] += rng.standard_normal((n, 2)) * (rng.random((n, 2)) < 0.5)
    Pf = np.triu(M, 1); Pf = Pf + Pf.T
    Pf += (1e-2 + abs(np.linalg.eigvalsh(Pf).min())) * np.eye(n)
    A = np.zeros((p, n)); G = np.zeros((m, n))
    for r in range(p):
        if r == 3:
            continue  # an empty equality row
        j = rng.integers(0, n - bw); A[r, j:j + bw + 1] = rng.standard_normal(bw + 1)
    for r in range(m):
        j = rng.integers(0, n - bw); G[r, j:j + 2] = rng.standard_normal(2)
        if seed % 2 and r % 3 == 0:
            G[r, n - 1] = 1.0
    args = (sp.csc_matrix(np.triu(Pf)), rng.standard_normal(n), sp.csc_matrix(A), rng.standard_normal(p), sp.csc_matrix(G), -np.ones(m), np.ones(m),
            -np.ones(n), np.full(n, np.inf))
    k = hip.SparseKKT(hip.SparseData(*args), kkt_solver=hip.SPARSE_MULTISTAGE)
    ko = orc.KKT(orc.Data.sparse(*args), kind="multistage")
    assert np.array_equal(k.block_info(), ko.block_info())
    x_reg = rng.uniform(1e-3, 1.0, n); z_reg = rng.uniform(1e-2, 10.0, m); delta = 1e-2
    assert k.update_scalings_and_factor(delta, x_reg, z_reg) and ko.update_scalings_and_factor(delta, x_reg, z_reg)
    r = [rng.standard_normal(s) for s in (n, p, m)]
    for u, v in zip(k.solve(*r), ko.solve(*r)):
        assert _rel(u, v) < 1e-8
    lx, ly, lz = k.solve(*r)
    r1 = r[0] - (Pf @ lx + x_reg * lx + A.T @ ly + G.T @ lz)
    r2 = r[1] - (A @ lx - delta * ly)
    r3 = r[2] - (G @ lx - z_reg * lz)
    assert max(np.abs(r1).max(), np.abs(r2).max(), np.abs(r3).max()) <= 1e-10 * max(1.0, np.abs(lx).max()) * max(1.0, np.abs(Pf).max())


def test_c0_trace_multistage_backend(hip, orc):
    """C0: the notebook reports 12 iterations and the same optimum for the multistage backend"""
    q = load_qp("qp_c0_scenario_mpc"); tr = load_json("c0_trace.json")
    s = hip.SparseSolver()
    s.settings.kkt_solver = hip.SPARSE_MULTISTAGE
    assert s.setup(*_args(q))
    assert s.solve() == 1
    assert s.info.iter == tr["iterations"]
    assert abs(s.info.primal_obj - tr["objective_scipy_trust_constr"]) < 1e-3


@pytest.mark.parametrize("name", ["qp_small_sparse_dual_inf", "qp_small_dense", "qp_scenario_mpc_small", "qp_scenario_mpc", "qp_chain_mass_sqp"])
def test_fixture_iteration_parity_multistage(hip, orc, name):
    q = load_qp(name)
    sh = hip.SparseSolver(); sh.settings.kkt_solver = hip.SPARSE_MULTISTAGE
    so = orc.Solver(); so.settings.kkt_solver = orc.SPARSE_MULTISTAGE
    assert sh.setup(*_args(q)) and so.setup(*_args(q), sparse=True)
    st_h, st_o = sh.solve(), so.solve()
    assert st_h == st_o
    assert sh.info.iter == so.info.iter
    if st_o == 1:
        assert abs(sh.info.primal_obj - so.info.primal_obj) <= 1e-6 * (1 + abs(so.info.primal_obj))


def test_clone_bitwise_multistage(hip):
    q = load_qp("qp_scenario_mpc_small")
    s1 = hip.SparseSolver(); s1.settings.kkt_solver = hip.SPARSE_MULTISTAGE
    assert s1.setup(*_args(q))
    s2 = s1.clone()
    assert s1.solve() == 1 and s2.solve() == 1
    assert np.array_equal(s1.result()["x"], s2.result()["x"])


def test_long_chain_mpc_residual(hip, orc):
    """a C5-style chain scaled down (x_{k+1} = A x_k + B u_k, 400 stages of nx=6, nu=3, box bounds): property test at a size
    the oracle still factors in well under a second: relative KKT residual <= 1e-10, device == oracle"""
    rng = np.random.default_rng(45)
    nx, nu, T = 6, 3, 400
    nz = nx + nu
    n = T * nz + nx
    Ad = np.eye(nx) + 0.1 * rng.standard_normal((nx, nx)); Bd = rng.standard_normal((nx, nu))
    rows, cols, vals = [], [], []
    for t in range(T):
        for i in range(nx):
            for j in range(nx):
                rows.append(t * nx + i); cols.append(t * nz + j); vals.append(Ad[i, j])
            for j in range(nu):
                rows.append(t * nx + i); cols.append(t * nz + nx + j); vals.append(Bd[i, j])
            rows.append(t * nx + i); cols.append((t + 1) * nz + i); vals.append(-1.0)
    p = T * nx
    A = sp.csc_matrix((vals, (rows, cols)), shape=(p, n))
    P = sp.diags(rng.uniform(0.5, 2.0, n), format="csc")
    args = (P, rng.standard_normal(n), A, np.zeros(p), None, None, None, -np.ones(n), np.ones(n))
    d = hip.SparseData(*args); od = orc.Data.sparse(*args)
    k = hip.KKTSystem(d, hip.default_settings(kkt_solver=hip.SPARSE_MULTISTAGE))
    ko = orc.KKTSystem(od, orc.Settings(kkt_solver=orc.SPARSE_MULTISTAGE))
    assert np.array_equal(k.backend().block_info(), ko.backend().block_info())
    state = random_vars(n, p, 0, rng, positive=True)
    assert k.update_scalings_and_factor(False, 1e-6, 1e-4, state) and ko.update_scalings_and_factor(False, 1e-6, 1e-4, state)
    rhs = random_vars(n, p, 0, rng)
    ok, lhs = k.solve(rhs); oko, ref = ko.solve(rhs)
    assert ok and oko
    res, nrm = k.condensed_residual()
    assert res <= 1e-10 * nrm
    assert _rel(lhs["x"], ref["x"]) < 1e-7 and _rel(lhs["y"], ref["y"]) < 1e-7


def test_two_processes_bitwise_equal(tmp_path):
    """the chain/tree engine of `sparse_multistage` is chosen from the sparsity structure (symbolic cost model, multistage_kkt.hip), never
    from a timing: two fresh processes must return bitwise the same solutions, and both engines must actually occur in the sample"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for i in range(2):
        f = str(tmp_path / f"run{i}.npz")
        env = {k: v for k, v in os.environ.items() if k != "PIQP_AMD_MULTISTAGE"}
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "workers", "multistage_repro.py"), f], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(dict(np.load(f)))
    assert outs[0].keys() == outs[1].keys()
    for key in outs[0]:
        assert np.array_equal(outs[0][key], outs[1][key]), key
    engines = {int(v[0]) for k, v in outs[0].items() if k.endswith("_engine")}
    assert engines == {0, 1}, engines
