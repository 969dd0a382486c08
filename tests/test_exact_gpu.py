"""The reference-order sparse engine on the device (piqp_amd/csrc/sparse_exact.hip; kkt_solver = SPARSE_LDLT_EXACT, and SPARSE_LDLT up to 8192 KKT rows):
L_vals, D, D_inv of sparse/ldlt.hpp:101-169 and the solution of :171-218 BITWISE equal to the CPU oracle's restatement (built without FMA contraction, like the
reference forces its own loop), on frozen fixtures at an early and at a late interior-point scaling, through the C-ABI.  Tolerance: none (integer-like contract:
the same IEEE operations in the same order)."""
import glob
import os

import numpy as np
import pytest

from qp_gen import random_vars
from qp_io import GOLDEN, load_qp

pytestmark = pytest.mark.gpu

SMALL = ["qp_small_dense", "qp_scenario_mpc_small", "qp_scenario_mpc", "qp_small_sparse_dual_inf", "qp_chain_mass_sqp", "qp_robot_arm_sqp", "mm_HS21", "mm_DUAL1", "mm_QAFIRO",
         "mm_CVXQP1_S", "mm_LOTSCHD", "mm_QBEACONF", "mm_QCAPRI", "mm_QETAMACR", "mm_QGROW7", "mm_QGROW22", "mm_QSHARE1B", "mm_STADAT1", "mm_QPILOTNO", "mm_QSHIP08L",
         "mm_QSHIP08S", "mm_AUG3DCQP", "mm_CONT-050", "nl_afiro", "nl_fffff800", "nl_finnis", "nl_perold", "nl_forplan", "nl_truss"]


def _args(q):
    return (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])


def _scalings(n, m, rng, late):
    if not late:
        return 1e-4, np.full(n, 1e-6), np.abs(rng.standard_normal(m)) + 0.1
    # end-game: rho = delta at their floor, s / z spread over many orders of magnitude
    return 1e-10, np.full(n, 1e-10), np.exp(rng.uniform(-18.0, 12.0, m))


def _check(hip, orc, name, kkt_solver):
    q = load_qp(name)
    d = hip.SparseData(*_args(q)); od = orc.Data.sparse(*_args(q))
    k = hip.SparseKKT(d, kkt_solver=kkt_solver)
    ko = orc.KKT(od, kind="sparse", mode=0)
    n, p, m = d.n, d.p, d.m
    rng = np.random.default_rng(17)
    for late in (False, True):
        delta, x_reg, z_reg = _scalings(n, m, rng, late)
        ok_h = k.update_scalings_and_factor(delta, x_reg, z_reg)
        ok_o = ko.update_scalings_and_factor(delta, x_reg, z_reg)
        assert bool(ok_h) == bool(ok_o), (name, late, ok_h, ok_o)
        if not ok_o:
            continue
        fh, fo = k.exact_factor(), ko.sparse_factor()
        assert np.array_equal(fh["perm"], fo["perm"]) and np.array_equal(fh["L_cols"], fo["L_cols"]) and np.array_equal(fh["L_ind"], fo["L_ind"])
        assert np.array_equal(fh["PKPt_val"], fo["PKPt_val"]), (name, late)
        bad = np.nonzero(fh["L_vals"] != fo["L_vals"])[0]
        assert bad.size == 0, (name, late, "L_vals", bad[:5], fh["L_vals"][bad[:5]], fo["L_vals"][bad[:5]])
        assert np.array_equal(fh["D"], fo["D"]) and np.array_equal(fh["D_inv"], fo["D_inv"]), (name, late, "D")
        for _ in range(2):
            rx, ry, rz = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(m)
            lh = k.solve(rx, ry, rz)
            lo = ko.solve(rx, ry, rz)
            for a, b, nm in zip(lh, lo, "xyz"):
                assert np.array_equal(np.asarray(a), np.asarray(b)), (name, late, "solve", nm)


@pytest.mark.parametrize("name", SMALL)
def test_factor_and_solve_bitwise_equal_the_oracle(hip, orc, name):
    _check(hip, orc, name, hip.SPARSE_LDLT_EXACT)


def test_default_engine_below_the_threshold_is_the_exact_one(hip, orc):
    """kkt_solver = sparse_ldlt picks the reference-order engine up to 8192 KKT rows, the multifrontal one above"""
    q = load_qp("mm_QAFIRO")
    k = hip.SparseKKT(hip.SparseData(*_args(q)), kkt_solver=hip.SPARSE_LDLT)
    assert k.exact_factor()["L_cols"].size > 0
    q = load_qp("mm_LISWET1")
    k = hip.SparseKKT(hip.SparseData(*_args(q)), kkt_solver=hip.SPARSE_LDLT)
    with pytest.raises(Exception):
        k.exact_factor()


def test_exact_engine_with_the_work_vector_in_hbm(hip, orc):
    """N above what one workgroup's LDS holds (20 000+ rows): the same kernels with y / x in HBM"""
    _check(hip, orc, "mm_CONT-101", hip.SPARSE_LDLT_EXACT)


def test_zero_pivot_is_reported_like_the_reference(hip, orc):
    """D[k] == 0.0 -> failure (ldlt.hpp:163): an equality row tied to two cancelling variables"""
    import scipy.sparse as sp
    n = 6
    P = sp.csc_matrix((n, n))
    A = sp.csc_matrix(np.array([[1.0, -1.0, 0, 0, 0, 0], [0, 0, 1.0, 1.0, 0, 0]]))
    q = dict(P=P, c=np.zeros(n), A=A, b=np.zeros(2), G=None, h_l=None, h_u=None, x_l=None, x_u=None)
    d = hip.SparseData(*_args(q)); od = orc.Data.sparse(*_args(q))
    k = hip.SparseKKT(d, kkt_solver=hip.SPARSE_LDLT_EXACT); ko = orc.KKT(od, kind="sparse", mode=0)
    # x_reg = 0 on an LP: the first pivots are exact zeros
    ok_h = k.update_scalings_and_factor(1e-8, np.zeros(n), np.zeros(0)); ok_o = ko.update_scalings_and_factor(1e-8, np.zeros(n), np.zeros(0))
    assert not ok_o and not ok_h
    ok_h = k.update_scalings_and_factor(1e-8, np.full(n, 1e-8), np.zeros(0)); ok_o = ko.update_scalings_and_factor(1e-8, np.full(n, 1e-8), np.zeros(0))
    assert ok_o and ok_h
    assert np.array_equal(k.exact_factor()["D"], ko.sparse_factor()["D"])


@pytest.mark.parametrize("tokens", ["exact_one_queue", "exact_serial_path", "exact_no_lds", "exact_solve1", "exact_one_queue,exact_grid=7"])
def test_schedule_variants_are_bitwise_the_oracle_too(tokens):
    """the engine's other schedules (PIQP_AMD_DEBUG, read once per process: a worker each): one queue of tickets instead of the per-XCD queues, the path pass of a
    task on one wave, work vectors in HBM, the single-wave substitution, a handful of workgroups -- who computes an entry and when changes, the operations do not"""
    import subprocess
    import sys
    env = dict(os.environ)
    env["PIQP_AMD_DEBUG"] = tokens
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "workers", "exact_variant.py")
    r = subprocess.run([sys.executable, worker, "mm_QAFIRO", "mm_QBEACONF", "qp_chain_mass_sqp", "mm_STADAT1", "nl_finnis"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_two_factorisations_of_two_handles_at_the_same_time(hip, orc):
    """two handles of the reference-order engine factor and solve from two host threads at the same time (two streams): with the per-XCD queues a factorisation needs
    workgroups on every XCD, so such launches are ordered on the device (sparse_exact.hip XqLaunchOrder) -- every result equals the single-threaded one bit for bit"""
    import threading
    q = load_qp("mm_QPILOTNO")
    d = hip.SparseData(*_args(q))
    n, p, m = d.n, d.p, d.m
    rng = np.random.default_rng(3)
    x_reg, z_reg = np.full(n, 1e-6), np.abs(rng.standard_normal(m)) + 0.1
    rx, ry, rz = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(m)
    ks = [hip.SparseKKT(d, kkt_solver=hip.SPARSE_LDLT_EXACT) for _ in range(2)]
    out, err = {}, []

    def work(tag, k, reps):
        try:
            ref = None
            for it in range(reps):
                assert k.update_scalings_and_factor(1e-4, x_reg, z_reg)
                x = np.asarray(k.solve(rx, ry, rz)[0]).copy()
                if ref is None:
                    ref = x
                assert np.array_equal(x, ref), (tag, it)
            out[tag] = ref
        except Exception as e:  # noqa: BLE001
            err.append((tag, repr(e)))

    work("single", ks[0], 2)
    ts = [threading.Thread(target=work, args=(t, k, 30)) for t, k in zip("ab", ks)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    assert not err, err
    assert np.array_equal(out["a"], out["single"]) and np.array_equal(out["b"], out["single"])
