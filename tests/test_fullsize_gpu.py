"""BASELINE.json configs C2, C3 and C5 at their FULL sizes under `-m gpu`.

The reference's own tests assert these properties at n = 20 (tests/src/dense/kkt_test.cpp:67-139, sparse/kkt_test.cpp:88-162,
sparse/multistage_kkt_test.cpp:24-211); here they run at the sizes the bench is quoted on:
  * relative residual of the condensed KKT system <= 1e-10 (north_star tolerance, written below as TOL);
  * agreement with the CPU oracle (factor columns / solutions) on the same inputs;
  * agreement between independent device paths (two orderings, chain-free tree engine vs the oracle's serial recurrence).
The oracle legs use its OpenMP build (seconds on the GPU box's host cores)."""
import os

import numpy as np
import pytest

from qp_gen import c3_problem, dense_strongly_convex_qp, mpc_chain, random_vars

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return 0.0 if a.size == 0 else float(np.abs(a - b).max() / (1e-300 + np.abs(b).max()))


@pytest.fixture(scope="module")
def c2(hip, orc):
    n, p, m = 4096, 0, 4096
    q = dense_strongly_convex_qp(n, p, m, seed=43, double_sided=True, exact_shift=False)
    rng = np.random.default_rng(7)
    return dict(n=n, p=p, m=m, q=q, d=hip.Data(**q), od=orc.Data.dense(**q),
                x_reg=rng.uniform(1e-6, 2.0, n), z_reg=rng.uniform(1e-3, 3.0, m), delta=1e-4,
                rhs=(rng.standard_normal(n), np.zeros(0), rng.standard_normal(m)))


@pytest.mark.parametrize("kkt_solver", [0, 16])
def test_c2_dense_factor_columns_and_solve_vs_oracle(hip, orc, c2, kkt_solver):
    """C2 (n = 4096, m = 4096, p = 0): Eigen::LLT (dense/kkt.hpp:82) and LDLTNoPivot (dense/ldlt_no_pivot.hpp:313-354) at the size of the
    bench line.  Factor columns sampled across all 32 panels vs the oracle's factor; backend solve vs the oracle's; residual of the
    condensed system from a NumPy product with the assembled matrix."""
    n, m = c2["n"], c2["m"]
    k = hip.DenseKKT(c2["d"], kkt_solver=kkt_solver)
    ko = orc.KKT(c2["od"], use_ldlt=(kkt_solver == 16))
    assert k.update_scalings_and_factor(c2["delta"], c2["x_reg"], c2["z_reg"])
    assert ko.update_scalings_and_factor(c2["delta"], c2["x_reg"], c2["z_reg"])
    F, Fo = k.internal_factor(), ko.internal_factor()
    cols = sorted(set(list(range(0, n, 97)) + [0, 1, 127, 128, 129, 2047, 2048, 4000, n - 2, n - 1]))
    for c in cols:
        a, b = F[c:, c], Fo[c:, c]
        assert np.abs(a - b).max() <= 1e-9 * max(1.0, np.abs(b).max()), (kkt_solver, c)
    rx, ry, rz = c2["rhs"]
    lx, ly, lz = k.solve(rx, ry, rz)
    ox, oy, oz = ko.solve(rx, ry, rz)
    assert _rel(lx, ox) < 1e-8 and _rel(lz, oz) < 1e-8
    # residual of the reduced system K lx = rx + G^T (rz / z_reg) with the ORACLE's assembled K (independent of the device assembly)
    K = np.tril(ko.internal_kkt_mat()); K = K + np.tril(K, -1).T
    G = c2["od"].mat("GT").T
    b = rx + G.T @ (rz / c2["z_reg"])
    assert np.abs(K @ lx - b).max() <= TOL * np.abs(b).max()
    assert np.abs(G @ lx - c2["z_reg"] * lz - rz).max() <= TOL * max(1.0, np.abs(rz).max(), np.abs(G @ lx).max())


@pytest.mark.parametrize("kkt_solver", [0, 16])
@pytest.mark.parametrize("refine", [False, True])
def test_c2_kkt_system_residual(hip, c2, kkt_solver, refine):
    """C2 through KKTSystem (kkt_system.hpp:143-369) at an interior state with rho = 1e-6, delta = 1e-4, with and without the
    refinement loop (:256-301)"""
    n, p, m = c2["n"], c2["p"], c2["m"]
    k = hip.KKTSystem(c2["d"], hip.default_settings(kkt_solver=kkt_solver))
    rng = np.random.default_rng(4)
    state = random_vars(n, p, m, rng, positive=True)
    assert k.update_scalings_and_factor(refine, 1e-6, 1e-4, state)
    ok, lhs = k.solve(random_vars(n, p, m, rng))
    assert ok and all(np.isfinite(v).all() for v in lhs.values())
    res, nrm = k.condensed_residual()
    assert res <= TOL * nrm, (kkt_solver, refine, res, nrm)
    if refine:
        assert res <= 1e-12 + 1e-11 * nrm  # iterative_refinement_eps_{abs,rel} = 1e-12 (settings.hpp:74-75), one digit of slack


@pytest.fixture(scope="module")
def c3():
    return c3_problem()


def _sparse_dims(args):
    P, c, A, b, G = args[:5]
    return P.shape[0], (A.shape[0] if A is not None else 0), (G.shape[0] if G is not None else 0)


def test_c3_sparse_50k_vs_oracle(hip, orc, c3):
    """C3 (n = 50 000, p = 20 000, m = 30 000, N = 100 000): device multifrontal LDLt vs the oracle's up-looking LDLt
    (sparse/ldlt.hpp:101-169 restated) on one right-hand side, plus the residual of the 3x3 system in SciPy"""
    n, p, m = _sparse_dims(c3)
    d = hip.SparseData(*c3); od = orc.Data.sparse(*c3)
    k = hip.SparseKKT(d, kkt_solver=hip.SPARSE_LDLT)
    ko = orc.KKT(od, kind="sparse", mode=0)
    rng = np.random.default_rng(5)
    x_reg = rng.uniform(1e-6, 2.0, n); z_reg = rng.uniform(1e-3, 3.0, m); delta = 1e-4
    assert k.update_scalings_and_factor(delta, x_reg, z_reg) and ko.update_scalings_and_factor(delta, x_reg, z_reg)
    rx, ry, rz = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(m)
    lx, ly, lz = k.solve(rx, ry, rz)
    ox, oy, oz = ko.solve(rx, ry, rz)
    assert _rel(lx, ox) < 1e-8 and _rel(ly, oy) < 1e-8 and _rel(lz, oz) < 1e-8
    P, _, A, _, G = c3[:5]
    import scipy.sparse as sp
    Pf = P + P.T - sp.diags(P.diagonal())
    r1 = rx - (Pf @ lx + x_reg * lx + A.T @ ly + G.T @ lz)
    r2 = ry - (A @ lx - delta * ly)
    r3 = rz - (G @ lx - z_reg * lz)
    nrm = max(np.abs(rx).max(), np.abs(ry).max(), np.abs(rz).max())
    assert max(np.abs(r1).max(), np.abs(r2).max(), np.abs(r3).max()) <= TOL * nrm
    st = k.sparse_stats()
    assert st["N"] == n + p + m


def test_c3_orderings_agree(hip, c3):
    """the default (cost-model) ordering and forced AMD (sparse/ordering.hpp:67-84) eliminate in different orders; both must meet the
    tolerance and agree with each other.  Separate processes: the ordering switch is read once per process."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for ordering in ("nd", "amd"):
        env = dict(os.environ, PIQP_AMD_ORDERING=ordering)
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "workers", "fullsize_c3.py")], capture_output=True, text=True, timeout=900, env=env)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert r.returncode == 0 and lines, r.stderr[-3000:]
        outs.append(json.loads(lines[-1]))
    for o in outs:
        assert o["rel_kkt_residual"] <= TOL, o
    assert outs[0]["tree_levels"] != outs[1]["tree_levels"]  # really two different eliminations
    x0 = np.load(outs[0]["x_file"]); x1 = np.load(outs[1]["x_file"])
    assert _rel(x0, x1) < 1e-8
    for o in outs:
        os.unlink(o["x_file"])


@pytest.fixture(scope="module")
def c5():
    return mpc_chain(12, 8, 25000, seed=45)


@pytest.mark.parametrize("backend", ["SPARSE_MULTISTAGE", "SPARSE_LDLT"])
def test_c5_chain_500k_vs_oracle_multistage(hip, orc, c5, backend):
    """C5 (n = 500 012, p = 300 000: 25 000 stages of n_x = 12, n_u = 8): device backends (stage-parallel elimination) vs the oracle's
    serial block recurrence (multistage_kkt.hpp:1253-1352, :1709-1816 restated) through KKTSystem::solve, and the residual"""
    n, p, m = _sparse_dims(c5)
    d = hip.SparseData(*c5); od = orc.Data.sparse(*c5)
    k = hip.KKTSystem(d, hip.default_settings(kkt_solver=getattr(hip, backend)))
    ko = orc.KKTSystem(od, orc.Settings(kkt_solver=orc.SPARSE_MULTISTAGE))
    rng = np.random.default_rng(6)
    state = random_vars(n, p, m, rng, positive=True)
    rhs = random_vars(n, p, m, rng)
    assert k.update_scalings_and_factor(False, 1e-6, 1e-4, state) and ko.update_scalings_and_factor(False, 1e-6, 1e-4, state)
    ok, lhs = k.solve(rhs)
    oko, lo = ko.solve(rhs)
    assert ok and oko
    res, nrm = k.condensed_residual()
    assert res <= TOL * nrm, (res, nrm)
    for key in ("x", "y", "z_bl", "z_bu", "s_bl", "s_bu"):
        assert _rel(lhs[key], lo[key]) < 1e-8, key
