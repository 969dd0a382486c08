"""The reference-order sparse engine (piqp_amd/csrc/sparse_exact.hip), host part -- no GPU needed.

pq_sparse_uplooking_plan exports everything the device kernels replay (sparse_symbolic.cpp analyse_uplooking): the AMD ordering without a postorder, L's column
structure, every row of L in the order sparse/ldlt.hpp:121-143 visits it, and the chain tasks of the elimination tree.  Here
  * the structure is held to the oracle's restatement of LDLt::factorize_symbolic_upper_triangular (exact integer equality);
  * the numeric phase is replayed on the host in exactly the order the kernels use -- tasks in dependency order instead of row order, the pattern of a row
    in plan order, the scatter of a column as one vector operation (distinct targets), the quotients and the D[k] terms 64 entries at a time -- with rounded
    products and rounded differences, and must reproduce the oracle's L_vals, D and D_inv BITWISE: what the device computes is then fixed by the plan alone;
  * the schedule's invariants (a task's children are complete chains that end in a child of its first row; every row in exactly one task)."""
import ctypes as C

import numpy as np
import pytest

from qp_gen import random_vars
from qp_io import load_qp

_ip = C.POINTER(C.c_int)
FIXTURES = ["qp_small_dense", "qp_scenario_mpc_small", "qp_small_sparse_dual_inf", "mm_HS21", "mm_DUAL1", "mm_QAFIRO", "mm_CVXQP1_S", "mm_LOTSCHD", "mm_QBEACONF",
            "mm_QCAPRI", "nl_afiro", "nl_finnis", "nl_fffff800", "qp_chain_mass_sqp"]


def _p(a):
    return a.ctypes.data_as(_ip)


def plan(name_or_args):
    import piqp_amd
    L = piqp_amd._lib.load()
    args = name_or_args
    d = piqp_amd.SparseData(*args)
    desc = d.descriptor()
    sizes = (C.c_longlong * 6)()
    none = [None] * 14
    N = L.pq_sparse_uplooking_plan(C.byref(desc), sizes, *none)
    assert N > 0, N
    nnzL, ntask, height, crit, nnzK, nch = (int(v) for v in sizes)
    z = lambda n: np.zeros(max(n, 1), np.int32)
    a = dict(perm=z(N), Cp=z(N + 1), Ci=z(nnzK), diag_pos=z(N), etree=z(N), Lp=z(N + 1), Li=z(nnzL), Rp=z(N + 1), Rcol=z(nnzL), Rpos=z(nnzL), task_lo=z(ntask), task_hi=z(ntask),
             tchild_ptr=z(ntask + 1), tchild=z(nch))
    order = ("perm", "Cp", "Ci", "diag_pos", "etree", "Lp", "Li", "Rp", "Rcol", "Rpos", "task_lo", "task_hi", "tchild_ptr", "tchild")
    assert L.pq_sparse_uplooking_plan(C.byref(desc), sizes, *[_p(a[k]) for k in order]) == N
    a.update(N=N, nnzL=nnzL, ntask=ntask, height=height, crit=crit, nnzK=nnzK)
    for k, n in (("Ci", nnzK), ("Li", nnzL), ("Rcol", nnzL), ("Rpos", nnzL), ("tchild", nch), ("task_lo", ntask), ("task_hi", ntask)):
        a[k] = a[k][:n]
    return a, d


def _args(q):
    return (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])


def replay(pl, Cx):
    """the kernels' order of operations on the host (numpy float64: every product and every difference rounded on its own)"""
    N, Cp, Ci, Lp, Li, Rp, Rcol, Rpos = pl["N"], pl["Cp"], pl["Ci"], pl["Lp"], pl["Li"], pl["Rp"], pl["Rcol"], pl["Rpos"]
    Lx, D, Dinv = np.zeros(pl["nnzL"]), np.zeros(N), np.zeros(N)
    y = np.zeros(N)
    done = np.zeros(N, bool)
    info = N
    # tasks in an order a parallel run could produce: last task first among the ready ones (anything but row order)
    ntask = pl["ntask"]
    pending = list(range(ntask))
    while pending:
        ready = [t for t in pending if all(done[c] for c in pl["tchild"][pl["tchild_ptr"][t]:pl["tchild_ptr"][t + 1]])]
        assert ready, "the task graph has a cycle"
        t = ready[-1]
        pending.remove(t)
        for k in range(pl["task_lo"][t], pl["task_hi"][t] + 1):
            if info < N:
                break  # (a zero pivot poisons the rows above it; the oracle stops there as well)
            y[Ci[Cp[k]:Cp[k + 1]]] = Cx[Cp[k]:Cp[k + 1]]
            Dk = y[k]; y[k] = 0.0
            for base in range(Rp[k], Rp[k + 1], 64):
                hi = min(base + 64, Rp[k + 1])
                yis = np.zeros(hi - base)
                for e in range(base, hi):
                    i, pos = Rcol[e], Rpos[e]
                    yi = y[i]; y[i] = 0.0
                    yis[e - base] = yi
                    cs = Lp[i]
                    if pos > cs:
                        tg = Li[cs:pos]
                        y[tg] = y[tg] - Lx[cs:pos] * yi
                with np.errstate(divide="ignore", invalid="ignore"):
                    l = yis / D[Rcol[base:hi]]
                    Lx[Rpos[base:hi]] = l
                    tp = l * yis
                for v in tp:
                    Dk = Dk - v
            D[k] = Dk
            with np.errstate(divide="ignore"):
                Dinv[k] = np.float64(1.0) / Dk
            if Dk == 0.0:
                info = min(info, k)
            assert info < N or not y.any()
        done[pl["task_hi"][t]] = True
    return Lx, D, Dinv, info


@pytest.mark.parametrize("name", FIXTURES)
def test_plan_structure_equals_the_oracle(orc, name):
    q = load_qp(name)
    pl, _d = plan(_args(q))
    od = orc.Data.sparse(*_args(q))
    ko = orc.KKT(od, kind="sparse", mode=0)
    f = ko.sparse_factor()
    N = pl["N"]
    assert N == f["N"]
    assert np.array_equal(pl["perm"], f["perm"])
    assert np.array_equal(pl["Cp"], f["PKPt_colptr"]) and np.array_equal(pl["Ci"], f["PKPt_rowind"])
    assert np.array_equal(pl["etree"], f["etree"])
    assert np.array_equal(pl["Lp"], f["L_cols"])
    # the oracle's L_ind is filled by a numeric factorisation (ldlt.hpp:159): run one
    rng = np.random.default_rng(3)
    st = random_vars(od.n, od.p, od.m, rng, positive=True)
    assert ko.update_scalings_and_factor(1e-4, np.full(od.n, 1e-6), np.abs(rng.standard_normal(od.m)) + 0.1)
    f = ko.sparse_factor()
    assert np.array_equal(pl["Li"], f["L_ind"])
    # rows: Rcol / Rpos enumerate every entry exactly once, Li[Rpos[e]] == k
    for k in (0, N // 3, N // 2, N - 1):
        e0, e1 = pl["Rp"][k], pl["Rp"][k + 1]
        assert np.all(pl["Li"][pl["Rpos"][e0:e1]] == k)
        assert np.all(pl["Rcol"][e0:e1] < k)
    assert sorted(pl["Rpos"].tolist()) == list(range(pl["nnzL"]))
    # tasks partition the rows into chains; a task waits for chains that end in a child of its first row
    cover = np.zeros(N, int)
    nchild = np.bincount(pl["etree"][pl["etree"] >= 0], minlength=N)
    for t in range(pl["ntask"]):
        lo, hi = pl["task_lo"][t], pl["task_hi"][t]
        cover[lo:hi + 1] += 1
        for r in range(lo + 1, hi + 1):
            assert pl["etree"][r - 1] == r and nchild[r] == 1
        ch = pl["tchild"][pl["tchild_ptr"][t]:pl["tchild_ptr"][t + 1]]
        assert sorted(ch.tolist()) == sorted(np.nonzero(pl["etree"] == lo)[0].tolist())
        assert all(c in set(pl["task_hi"].tolist()) for c in ch)
    assert np.all(cover == 1)
    assert pl["crit"] <= pl["nnzL"] and pl["height"] <= N


@pytest.mark.parametrize("name", FIXTURES)
def test_replay_in_kernel_order_is_bitwise_the_oracle(orc, name):
    q = load_qp(name)
    pl, _d = plan(_args(q))
    od = orc.Data.sparse(*_args(q))
    ko = orc.KKT(od, kind="sparse", mode=0)
    rng = np.random.default_rng(5)
    for delta, scale in ((1e-4, 1.0), (1e-10, 1e-8)):
        x_reg = np.full(od.n, delta)
        z_reg = (np.abs(rng.standard_normal(od.m)) + 1e-3) * scale + 1e-12
        ok = ko.update_scalings_and_factor(delta, x_reg, z_reg)
        f = ko.sparse_factor()
        Lx, D, Dinv, info = replay(pl, f["PKPt_val"])
        assert (info == pl["N"]) == bool(ok)
        if ok:
            assert np.array_equal(Lx, f["L_vals"]), name
            assert np.array_equal(D, f["D"]) and np.array_equal(Dinv, f["D_inv"]), name
