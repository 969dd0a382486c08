"""Real Maros-Meszaros problems through the sparse device path: the four larger problems SURVEY.md 8d names as
the cross-checks of the sparse configuration -- CONT-201, BOYD1, AUG3DCQP, LISWET1 (frozen by tests/golden/make_fixtures.py from
tests/data/maros_meszaros/*.mat) -- next to the banded synthetic C3, and the status contract of the reference's own sweep
(tests/src/sparse/maros_meszaros_tests.cpp: status == PIQP_SOLVED) over every frozen Maros-Meszaros fixture, device vs oracle.
Tolerances: relative KKT residual <= 1e-10 (north star), identical status, iteration counts equal (+-1 beyond 30 iterations)."""
import glob
import os

import numpy as np
import pytest

from qp_gen import random_vars
from qp_io import GOLDEN, load_qp

pytestmark = pytest.mark.gpu
TOL = 1e-10
BIG = ["mm_CONT-201", "mm_BOYD1", "mm_AUG3DCQP", "mm_LISWET1"]
ALL_MM = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "mm_*.npz")))


def _args(q):
    return (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])


def _rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return 0.0 if a.size == 0 else float(np.abs(a - b).max() / (1e-300 + np.abs(b).max()))


@pytest.mark.parametrize("name", BIG)
def test_kkt_factor_solve_on_real_problem(hip, orc, name):
    """KKTSystem factor + solve at an interior state (rho = 1e-6, delta = 1e-4): residual of the condensed system and agreement with the oracle's
    up-looking LDLt (sparse/ldlt.hpp:101-169 restated) on the same right-hand side; the symbolic figures are printed for DESIGN.md"""
    q = load_qp(name)
    d = hip.SparseData(*_args(q)); od = orc.Data.sparse(*_args(q))
    n, p, m = d.n, d.p, d.m
    k = hip.KKTSystem(d, hip.default_settings(kkt_solver=hip.SPARSE_LDLT))
    ko = orc.KKTSystem(od, orc.Settings(kkt_solver=orc.SPARSE_LDLT))
    rng = np.random.default_rng(11)
    state = random_vars(n, p, m, rng, positive=True)
    rhs = random_vars(n, p, m, rng)
    assert k.update_scalings_and_factor(False, 1e-6, 1e-4, state) and ko.update_scalings_and_factor(False, 1e-6, 1e-4, state)
    ok, lhs = k.solve(rhs)
    oko, lo = ko.solve(rhs)
    assert ok and oko
    res, nrm = k.condensed_residual()
    # residual of the condensed 3 x 3 system (kkt_system.hpp:507-519) for BOTH solutions, in extended precision, from the oracle's (Ruiz-free:
    # KKTSystem level) matrices and its x_reg / z_reg / reduced right-hand sides
    import scipy.sparse as sp
    L = np.longdouble
    Pu, AT, GT = od.csc("P_utri"), od.csc("AT"), od.csc("GT")
    Pf = (Pu + sp.triu(Pu, 1).T).tocsr()
    xr, zr, rx, rz, ry = ko.x_reg(), ko.z_reg(), ko.rhs_x_bar(), ko.rhs_z_bar(), rhs["y"]

    def resid(l):
        z = l["z_u"] - l["z_l"]
        r1 = rx.astype(L) - (Pf @ l["x"]).astype(L) - xr.astype(L) * l["x"] - (AT @ l["y"]).astype(L) - (GT @ z).astype(L)
        r2 = ry.astype(L) - (AT.T @ l["x"]).astype(L) + L(1e-4) * l["y"] if p else np.zeros(0, L)
        r3 = rz.astype(L) - (GT.T @ l["x"]).astype(L) + zr.astype(L) * z if m else np.zeros(0, L)
        scale = max(np.abs(rx).max(), np.abs(ry).max() if p else 0.0, np.abs(rz).max() if m else 0.0)
        return float(max(np.abs(r1).max(), np.abs(r2).max() if p else 0.0, np.abs(r3).max() if m else 0.0) / scale)
    rh, ro = resid(lhs), resid(lo)
    st = k.backend().sparse_stats()
    print(f"\n{name}: n={n} p={p} m={m} N={st['N']} nnz(K)={st['nnz_K']} nnz(L)={st['nnz_L']} supernodes={st['supernodes']} levels={st['tree_levels']} "
          f"max_front={st['max_front']} flops={st['flops_factor']:.3g} rel.residual device {rh:.2e} (own check {res / nrm:.2e}) oracle {ro:.2e}")
    # the bar is 1e-10 wherever the reference algorithm itself reaches it on this state; where it does not (BOYD1: P spans nine orders of
    # magnitude on its diagonal), the device must not be worse than the oracle by more than rounding scatter
    assert rh <= max(TOL, 4.0 * ro), (name, rh, ro)
    for key in ("x", "y"):
        assert _rel(lhs[key], lo[key]) < 1e-6, (name, key)


# Whole solves.  Round 5: kkt_solver = sparse_ldlt runs the reference's OWN elimination order on the device for KKT systems up to 8192 rows (sparse_exact.hip), the
# mat-vecs and the interior-point loop in the reference's order of operations, all built without FMA contraction like the oracle: a whole solve is then the same
# sequence of IEEE operations as the oracle's, and the contract is the strongest one there is -- the per-iteration table of solver.hpp:590-602 (objectives,
# residuals, rho, delta, mu, step lengths) BITWISE equal on every iteration, the same status, the same count, the same x.  That covers every fixture the earlier
# rounds had to list as trajectory sensitive (the degenerate LPs that decide their path on exact zero pivots: QBEACONF, fffff800, finnis, perold, forplan, ...):
# there is no allow-list any more.  Above 8192 rows the default engine is the supernodal multifrontal one (another summation order by construction, 10-60 x faster
# there: profiles/r06_big_engines.txt): same status, the optimum to 1e-6 and the oracle's iteration count -- no slack (round 6) -- except on the two of the eight
# such fixtures where the two summation orders end one iteration apart, pinned by name to the count the device takes (a change of either is a change of arithmetic).
# The reference-order engine does not stop at 8192 rows, though: kkt_solver = SPARSE_LDLT_EXACT runs it on any size, and all eight fixtures -- CONT-201, SURVEY 8(d)'s
# named cross-check, among them: 12 iterations like the oracle -- are held to the bitwise contract through it as well (test below; 0.1-4 s per whole solve).
EXACT_MAX_ROWS = 8192
MULTIFRONTAL_COUNTS = {"mm_CONT-201": (13, 12), "mm_STADAT3": (14, 15)}  # fixture -> (multifrontal engine, oracle)


def _rows(q):
    return q["P"].shape[0] + (0 if q["A"] is None else q["A"].shape[0]) + (0 if q["G"] is None else q["G"].shape[0])


def _solve_both(hip, orc, q, netlib=False):
    sh = hip.SparseSolver(); sh.settings.kkt_solver = hip.SPARSE_LDLT
    so = orc.Solver(); so.settings.kkt_solver = orc.SPARSE_LDLT
    if netlib:
        sh.settings.infeasibility_threshold = so.settings.infeasibility_threshold = 0.01
    sh.enable_trace(1024); so.enable_trace(1024)
    assert sh.setup(*_args(q)) and so.setup(*_args(q), sparse=True)
    return sh, so, sh.solve(), so.solve()


def _assert_same_solve(name, q, sh, so, st_h, st_o):
    assert st_h == st_o, (name, st_h, st_o)
    if _rows(q) <= EXACT_MAX_ROWS:
        th, to = sh.trace(), so.trace()
        assert sh.info.iter == so.info.iter, (name, sh.info.iter, so.info.iter)
        assert th.shape == to.shape, (name, th.shape, to.shape)
        same = (th == to) | ((th != th) & (to != to))
        bad = np.argwhere(~same)
        assert bad.size == 0, (name, "first differing (iteration, column)", bad[0].tolist(), th[tuple(bad[0])], to[tuple(bad[0])])
        assert np.array_equal(np.asarray(sh.result()["x"]), np.asarray(so.result()["x"])), name
    else:
        if name in MULTIFRONTAL_COUNTS:
            assert (sh.info.iter, so.info.iter) == MULTIFRONTAL_COUNTS[name], (name, sh.info.iter, so.info.iter)
        else:
            assert sh.info.iter == so.info.iter, (name, sh.info.iter, so.info.iter)
        if st_h == 1:
            assert abs(sh.info.primal_obj - so.info.primal_obj) <= 1e-6 * max(1.0, abs(so.info.primal_obj)) + 10 * so.settings.eps_abs


ABOVE_EXACT_MAX_ROWS = ["mm_BOYD1", "mm_CONT-101", "mm_CONT-201", "mm_LISWET1", "mm_POWELL20", "mm_STADAT3", "mm_UBH1", "nl_truss"]


@pytest.mark.parametrize("name", ABOVE_EXACT_MAX_ROWS)
def test_fixtures_above_8192_rows_bitwise_through_the_reference_order_engine(hip, orc, name):
    """the eight frozen fixtures whose KKT system has more than 8192 rows (9 806 .. 93 279) through kkt_solver = SPARSE_LDLT_EXACT: the per-iteration table, the status,
    the count and x BITWISE the oracle's, like the other 214 (sparse/ldlt.hpp:101-169 in the reference's own order of operations at any size; above 20 000 rows the
    work vector of the row pass lives in HBM).  Record: profiles/r06_exact_big.txt"""
    q = load_qp(name)
    netlib = name.startswith("nl")
    sh = hip.SparseSolver(); sh.settings.kkt_solver = hip.SPARSE_LDLT_EXACT
    so = orc.Solver(); so.settings.kkt_solver = orc.SPARSE_LDLT
    if netlib:
        sh.settings.infeasibility_threshold = so.settings.infeasibility_threshold = 0.01
    sh.enable_trace(1024); so.enable_trace(1024)
    assert sh.setup(*_args(q)) and so.setup(*_args(q), sparse=True)
    st_h, st_o = sh.solve(), so.solve()
    assert st_h == st_o == 1, (name, st_h, st_o)
    assert sh.info.iter == so.info.iter, (name, sh.info.iter, so.info.iter)
    th, to = sh.trace(), so.trace()
    assert th.shape == to.shape, (name, th.shape, to.shape)
    same = (th == to) | ((th != th) & (to != to))
    bad = np.argwhere(~same)
    assert bad.size == 0, (name, "first differing (iteration, column)", bad[0].tolist(), th[tuple(bad[0])], to[tuple(bad[0])])
    assert np.array_equal(np.asarray(sh.result()["x"]), np.asarray(so.result()["x"])), name


@pytest.mark.parametrize("name", ALL_MM)
def test_status_and_iterations_match_oracle(hip, orc, name):
    """maros_meszaros_tests.cpp contract through the device solver: SOLVED wherever the reference's sweep expects it, and the solve itself equal to the oracle's
    (bitwise below 8192 KKT rows, see above)"""
    q = load_qp(name)
    sh, so, st_h, st_o = _solve_both(hip, orc, q)
    assert st_o == 1, (name, st_o)  # the reference's sweep expects SOLVED on every file; the oracle meets it on all 110 frozen ones
    _assert_same_solve(name, q, sh, so, st_h, st_o)


NETLIB_FEAS = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "nl_*.npz")))
NETLIB_INFEAS = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "nli_*.npz")))
# where the ORACLE itself misses the reference test's expectation (MAX_ITER instead of SOLVED / INFEASIBLE): degenerate LPs whose trajectory depends on
# the fill-reducing ordering and on rounding; the reference's own run cannot be reproduced here (Eigen is absent).  The device is held to the oracle.
ORACLE_MISSES_REFERENCE = {"nl_bnl2", "nl_pilot-we", "nli_ceria3d", "nli_cplex2", "nli_qual"}


@pytest.mark.parametrize("name", NETLIB_FEAS + NETLIB_INFEAS)
def test_netlib_lp_status_matches_oracle(hip, orc, name):
    """tests/src/sparse/netlib_lp_tests.cpp through the device solver, with that test's setting (infeasibility_threshold = 0.01): SOLVED on the
    feasible set, PRIMAL or DUAL INFEASIBLE on the infeasible set -- asserted against the reference's expectation wherever the oracle meets it, and the whole
    solve against the oracle's everywhere (bitwise below 8192 KKT rows)"""
    q = load_qp(name)
    sh, so, st_h, st_o = _solve_both(hip, orc, q, netlib=True)
    expected = (1,) if name.startswith("nl_") else (-2, -3)
    if name in ORACLE_MISSES_REFERENCE:
        assert st_o not in expected  # keeps the list honest
    else:
        assert st_o in expected, (name, st_o)
    _assert_same_solve(name, q, sh, so, st_h, st_o)
