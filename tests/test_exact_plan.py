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


ITEMS = ("stats", "perm", "Cp", "Ci", "diag_pos", "etree", "Lp", "Li", "Rp", "Rcol", "Rpos", "task_ptr", "task_rows", "row_task", "row_lane", "row_prev", "dep_ptr", "dep",
         "tk_kind", "tk_id", "Rcnt", "Rtab", "tab_ptr", "mask_ptr", "task_nU", "Tmask")


def plan(args):
    import piqp_amd
    L = piqp_amd._lib.load()
    d = piqp_amd.SparseData(*args)
    desc = d.descriptor()
    n = len(ITEMS)
    what = (C.c_int * n)(*range(n))
    lens = (C.c_longlong * n)()
    N = L.pq_sparse_uplooking_plan(C.byref(desc), n, what, None, lens)
    assert N > 0, N
    arrs = [np.zeros(max(int(lens[q]), 1), np.uint64 if ITEMS[q] == "Tmask" else np.int32) for q in range(n)]
    outs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
    assert L.pq_sparse_uplooking_plan(C.byref(desc), n, what, outs, lens) == N
    a = {ITEMS[q]: arrs[q][:int(lens[q])] for q in range(n)}
    nnzL, ntask, height, crit, nnzK, ntick = (int(v) for v in a["stats"])
    a.update(N=N, nnzL=nnzL, ntask=ntask, height=height, crit=crit, nnzK=nnzK)
    return a, d


def _args(q):
    return (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])


def replay(pl, Cx):
    """the kernels' order of operations on the host (numpy float64: every product and every difference rounded on its own): tickets in an order a parallel run
    could produce (the row passes of a task before its path pass, otherwise latest first), the row pass and the path pass as sparse_exact.hip runs them"""
    N, Cp, Ci, Lp, Li, Rp, Rcol, Rpos, Rcnt, Rtab = (pl[k] for k in ("N", "Cp", "Ci", "Lp", "Li", "Rp", "Rcol", "Rpos", "Rcnt", "Rtab"))
    Lx, D, Dinv = np.zeros(pl["nnzL"]), np.zeros(N), np.zeros(N)
    Ystash, Pstash, Dinit = np.zeros(pl["nnzL"]), np.zeros(pl["nnzL"]), np.zeros(N)
    Lblock = np.zeros(max(int(pl["tab_ptr"][-1]), 1))
    y = np.zeros(N)
    done = np.zeros(N, bool); p1done = np.zeros(N, bool); ready = np.zeros(N, bool)
    info = [N]
    tptr, trows, rtask, rlane, rprev = pl["task_ptr"], pl["task_rows"], pl["row_task"], pl["row_lane"], pl["row_prev"]

    def row_pass(k):
        t = rtask[k]
        W = tptr[t + 1] - tptr[t]
        multi = W > 1
        y[Ci[Cp[k]:Cp[k + 1]]] = Cx[Cp[k]:Cp[k + 1]]
        Dk = y[k]; y[k] = 0.0
        for base in range(Rp[k], Rp[k + 1], 64):
            hi = min(base + 64, Rp[k + 1])
            yis = np.zeros(hi - base)
            for e in range(base, hi):
                i = Rcol[e]
                yi = y[i]; y[i] = 0.0
                yis[e - base] = yi
                cnt = Rcnt[e]
                if cnt > 0:
                    cs = Lp[i]
                    tg = Li[cs:cs + cnt]
                    y[tg] = y[tg] - Lx[cs:cs + cnt] * yi
            ext = Rcnt[base:hi] >= 0
            with np.errstate(all="ignore"):
                l = yis / D[Rcol[base:hi]]
                tp = l * yis
            ee = np.arange(base, hi)
            Lx[Rpos[ee[ext]]] = l[ext]
            if multi:
                Ystash[base:hi] = yis
                Pstash[ee[ext]] = tp[ext]
                tb = pl["tab_ptr"][t]
                Lblock[tb + Rtab[ee[ext]] * W + rlane[k]] = l[ext]
            else:
                for v in tp:
                    Dk = Dk - v
        if multi:
            Dinit[k] = Dk
            p1done[k] = True
        else:
            finish(k, Dk)
        assert info[0] < N or not y.any()

    def finish(k, Dk):
        D[k] = Dk
        with np.errstate(all="ignore"):
            Dinv[k] = np.float64(1.0) / Dk
        if Dk == 0.0:
            info[0] = min(info[0], k)
        done[k] = True

    def path_pass(t):
        R = trows[tptr[t]:tptr[t + 1]]
        W = len(R)
        nU = pl["task_nU"][t]
        tb, mb = pl["tab_ptr"][t], pl["mask_ptr"][t]
        Dlane = np.zeros(W)
        for j, k in enumerate(R):
            assert p1done[k]
            acc = np.zeros(W); present = np.zeros(W, bool); pos = np.zeros(W, int)
            for e in range(Rp[k], Rp[k + 1]):
                if Rtab[e] >= nU:
                    c = Rtab[e] - nU
                    acc[c] = Ystash[e]; present[c] = True; pos[c] = Rpos[e]
            for e in range(Rp[k], Rp[k + 1]):
                u = Rtab[e]
                src = Ystash[e] if u < nU else acc[u - nU]
                mask = int(pl["Tmask"][mb + u])
                lanes = np.array([c for c in range(j) if (mask >> c) & 1], int)
                if lanes.size:
                    with np.errstate(all="ignore"):
                        acc[lanes] = acc[lanes] - Lblock[tb + u * W + lanes] * src
            with np.errstate(all="ignore"):
                l = acc / Dlane
                prodp = l * acc
            for c in range(j):
                if present[c]:
                    Lx[pos[c]] = l[c]
                    Lblock[tb + (nU + c) * W + j] = l[c]
            Dk = Dinit[k]
            for e in range(Rp[k], Rp[k + 1]):
                u = Rtab[e]
                with np.errstate(all="ignore"):
                    Dk = Dk - (Pstash[e] if u < nU else prodp[u - nU])
            finish(k, Dk)
            Dlane[j] = Dk

    # tickets: any order that respects the waits -- here: repeatedly the LAST ticket whose waits are satisfied
    kinds, ids = pl["tk_kind"], pl["tk_id"]
    pending = list(range(len(kinds)))
    while pending and info[0] == N:
        pick = None
        for q in reversed(pending):
            if kinds[q] == 0:
                k = ids[q]
                ok = all(done[c] for c in pl["dep"][pl["dep_ptr"][k]:pl["dep_ptr"][k + 1]])
            else:
                t = ids[q]
                ok = all(p1done[k] for k in trows[tptr[t]:tptr[t + 1]])
            if ok:
                pick = q
                break
        assert pick is not None, "the ticket graph has a cycle"
        pending.remove(pick)
        if kinds[pick] == 0:
            ready[ids[pick]] = True
            row_pass(ids[pick])
        else:
            path_pass(ids[pick])
    return Lx, D, Dinv, info[0]


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
    # tasks partition the rows into paths of the elimination tree of at most 64 rows; a row pass waits for the row's children outside its task
    et, tptr, trows = pl["etree"], pl["task_ptr"], pl["task_rows"]
    cover = np.zeros(N, int)
    for t in range(pl["ntask"]):
        R = trows[tptr[t]:tptr[t + 1]]
        assert 1 <= len(R) <= 64
        cover[R] += 1
        for a, b in zip(R[:-1], R[1:]):
            assert et[a] == b and pl["row_prev"][b] == a
        assert pl["row_prev"][R[0]] == -1
        assert np.array_equal(pl["row_lane"][R], np.arange(len(R))) and np.all(pl["row_task"][R] == t)
    assert np.all(cover == 1)
    outside = {k: sorted(c for c in np.nonzero(et == k)[0] if pl["row_task"][c] != pl["row_task"][k]) for k in range(N)}
    for k in range(N):
        R = trows[tptr[pl["row_task"][k]]:tptr[pl["row_task"][k] + 1]]
        want = sorted(c for r in R[:pl["row_lane"][k] + 1] for c in outside[r])  # the children outside the task of k and of the path rows below it
        assert sorted(pl["dep"][pl["dep_ptr"][k]:pl["dep_ptr"][k + 1]].tolist()) == want
        assert all(pl["task_rows"][pl["task_ptr"][pl["row_task"][c] + 1] - 1] == c for c in want)  # a waited-for row ends its own task
    # tickets: every row pass once, rows ascending; a path pass right behind the row pass of its last row
    rows_seen = [i for kd, i in zip(pl["tk_kind"], pl["tk_id"]) if kd == 0]
    assert rows_seen == list(range(N))
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
