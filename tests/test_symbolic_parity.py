"""S4 -- the integer work of sparse::KKT's constructor in the PRODUCT, pinned bit for bit (index work must be exact):
  * the reference's only golden for it, tests/src/sparse/utils_test.cpp:55-92 (4 x 4: AMD ordering [1 2 0 3], permuted matrix, Ai_to_Ci = 3 0 2 1 5 4 6),
    through the product's own pq_sparse_amd_order / pq_sparse_permute_sym_upper (sparse/ordering.hpp:67-124, sparse/utils.hpp:32-128);
  * K pattern (create_kkt_matrix, kkt_full.hpp:39-170 and the three eliminated variants), AMD ordering, PKPt pattern and the K -> PKPt value map PKi of
    the product against the oracle's restatement on frozen fixtures, all four KKTModes, exact integer equality;
  * (gpu) the ordering a handle actually eliminates in: under PIQP_AMD_ORDERING=amd its fill-reducing ordering IS that AMD ordering, and its elimination
    order is a relabelling that keeps the elimination tree's parent relation (a postorder of it: same fill, same nnz(L) as the oracle's symbolic phase).
These entry points are host-only: no GPU is needed for the first two groups."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

from qp_io import load_json, load_qp

_ip = C.POINTER(C.c_int)

FIXTURES = ["qp_small_dense", "qp_scenario_mpc_small", "qp_scenario_mpc", "qp_chain_mass_sqp", "qp_robot_arm_sqp", "qp_small_sparse_dual_inf",
            "mm_HS21", "mm_DUAL1", "mm_QAFIRO", "mm_CVXQP1_S", "mm_LOTSCHD", "mm_QBEACONF", "mm_QCAPRI", "mm_AUG3DCQP", "mm_LISWET1", "mm_CONT-050",
            "nl_fffff800", "nl_bnl2", "nl_afiro", "nl_ship08l"]


def _lib():
    import piqp_amd
    return piqp_amd._lib.load()


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(_ip)


def test_amd_4x4_known_answer_product():
    """utils_test.cpp:55-92 through the product's routines"""
    L = _lib()
    kat = load_json("kat_small.json")["amd_4x4"]
    A = sp.csc_matrix(np.array([[1, 0, 2, 3], [0, 4, 0, 5], [0, 0, 6, 0], [0, 0, 0, 7.0]]))
    A.sort_indices()
    ip, ii = _i32(A.indptr), _i32(A.indices)
    perm = np.zeros(4, np.int32)
    assert L.pq_sparse_amd_order(4, _p(ip), _p(ii), _p(perm)) == 0
    assert list(perm) == kat["ordering"] == [1, 2, 0, 3]
    pinv = np.zeros(4, np.int32); pinv[perm] = np.arange(4)
    Cp, Ci, m = np.zeros(5, np.int32), np.zeros(7, np.int32), np.zeros(7, np.int32)
    assert L.pq_sparse_permute_sym_upper(4, _p(ip), _p(ii), _p(pinv), _p(Cp), _p(Ci), _p(m)) == 0
    assert list(m) == kat["Ai_to_Ci"] == [3, 0, 2, 1, 5, 4, 6]
    Cx = np.zeros(7); Cx[m] = A.data
    Cm = sp.csc_matrix((Cx, Ci, Cp), shape=(4, 4)).toarray()
    expect = np.zeros((4, 4))
    for (i, j, v) in [(0, 0, 4), (0, 3, 5), (1, 1, 6), (1, 2, 2), (2, 2, 1), (2, 3, 3), (3, 3, 7)]:
        expect[i, j] = v
    assert np.array_equal(Cm, expect)
    x = np.array([1.0, 2, 3, 4])
    assert list(x[perm]) == [2, 3, 1, 4]  # ordering.perm (utils_test.cpp:84-88)


def _product_symbolic(L, d, mode):
    import piqp_amd
    desc = d.descriptor()
    nnz = C.c_int()
    N = L.pq_sparse_kkt_symbolic(C.byref(desc), mode, C.byref(nnz), None, None, None, None, None, None)
    assert N > 0, N
    Kp, perm, PKp = np.zeros(N + 1, np.int32), np.zeros(N, np.int32), np.zeros(N + 1, np.int32)
    Ki, PKr, PKi = np.zeros(nnz.value, np.int32), np.zeros(nnz.value, np.int32), np.zeros(nnz.value, np.int32)
    assert L.pq_sparse_kkt_symbolic(C.byref(desc), mode, C.byref(nnz), _p(Kp), _p(Ki), _p(perm), _p(PKp), _p(PKr), _p(PKi)) == N
    return N, nnz.value, Kp, Ki, perm, PKp, PKr, PKi


def _oracle_symbolic(orc, od, mode):
    k = orc.KKT(od, kind="sparse", mode=mode)
    Lo = orc.lib()
    pre = "orc_sparse_kkt_" if mode == 0 else "orc_sparse_cond_kkt_"
    N, nnz = getattr(Lo, pre + "dim")(k.ptr), getattr(Lo, pre + "nnz")(k.ptr)
    get = lambda nm, cnt: np.ctypeslib.as_array(getattr(Lo, pre + nm)(k.ptr), shape=(cnt,)).copy()
    return N, nnz, get("perm", N), get("PKPt_colptr", N + 1), get("PKPt_rowind", nnz), get("PKi", nnz), k


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("name", FIXTURES)
def test_kkt_pattern_ordering_and_value_map_equal_the_oracle(orc, name, mode):
    """exact equality of N, nnz(K), the AMD ordering, the PKPt pattern and PKi on frozen fixtures: the product's host code and the oracle are two
    independent restatements of sparse/kkt.hpp:51-70 (one C++, one C) held to the same integers"""
    import piqp_amd
    L = _lib()
    q = load_qp(name)
    args = (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
    d = piqp_amd.SparseData(*args)
    od = orc.Data.sparse(*args)
    N, nnz, Kp, Ki, perm, PKp, PKr, PKi = _product_symbolic(L, d, mode)
    No, nnzo, permo, PKpo, PKro, PKio, _keep = _oracle_symbolic(orc, od, mode)
    assert (N, nnz) == (No, nnzo)
    assert sorted(perm.tolist()) == list(range(N))
    assert np.array_equal(perm, permo), name
    assert np.array_equal(PKp, PKpo) and np.array_equal(PKr, PKro), name
    assert np.array_equal(PKi, PKio), name
    # K is upper triangular with its diagonal last in every column (kkt_full.hpp:46-87), PKPt likewise upper with sorted columns
    for j in range(N):
        col = Ki[Kp[j]:Kp[j + 1]]
        assert col.size and col[-1] == j and np.all(np.diff(col) > 0)
    for j in (0, N // 2, N - 1):
        col = PKr[PKp[j]:PKp[j + 1]]
        assert col.size and col[-1] == j and np.all(np.diff(col) > 0)


def _etree(N, Cp, Ci):
    parent = -np.ones(N, np.int64); anc = -np.ones(N, np.int64)
    for k in range(N):
        for q in range(Cp[k], Cp[k + 1]):
            i = Ci[q]
            while i != -1 and i < k:
                nxt = anc[i]
                anc[i] = k
                if nxt == -1:
                    parent[i] = k
                i = nxt
    return parent


_CHILD = r"""
import sys, json, ctypes as C
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import piqp_amd
from qp_io import load_qp
out = {}
for name in sys.argv[2:]:
    q = load_qp(name)
    d = piqp_amd.SparseData(q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
    k = piqp_amd.SparseKKT(d, kkt_solver=piqp_amd.SPARSE_LDLT)
    N = d.n + d.p + d.m
    fp, ep = np.zeros(N, np.int32), np.zeros(N, np.int32)
    kind = k.L.pq_kkt_sparse_ordering(k.h, fp.ctypes.data_as(C.POINTER(C.c_int)), ep.ctypes.data_as(C.POINTER(C.c_int)))
    out[name] = dict(kind=kind, fill=fp.tolist(), elim=ep.tolist(), nnz_L=k.sparse_stats()["nnz_L"])
print("RESULT " + json.dumps(out))
"""


@pytest.mark.gpu
def test_handle_eliminates_in_the_reference_ordering_when_amd_is_forced(hip, orc):
    """PIQP_AMD_ORDERING=amd: the handle's fill-reducing ordering equals the oracle's AMD exactly on 12 fixtures, and its elimination order is an
    etree-respecting relabelling of it (a vertex is eliminated after all its etree descendants) -- same fill pattern as the reference's symbolic phase
    (sparse/ldlt.hpp:42-99) up to the supernodes' explicit zeros, never less."""
    names = ["qp_scenario_mpc", "qp_chain_mass_sqp", "qp_robot_arm_sqp", "mm_DUAL1", "mm_QAFIRO", "mm_CVXQP1_S", "mm_LOTSCHD", "mm_QBEACONF", "mm_QCAPRI",
             "mm_AUG3DCQP", "nl_fffff800", "nl_afiro"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PIQP_AMD_ORDERING="amd")
    r = subprocess.run([sys.executable, "-c", _CHILD, root] + names, env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, r.stdout[-2000:] + r.stderr[-2000:]
    import json
    res = json.loads(line[0][7:])
    for name in names:
        q = load_qp(name)
        args = (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
        od = orc.Data.sparse(*args)
        N, nnz, permo, PKpo, PKro, _, ko = _oracle_symbolic(orc, od, 0)
        rec = res[name]
        assert rec["kind"] == 0, name
        assert rec["fill"] == permo.tolist(), name
        elim = np.array(rec["elim"])
        assert sorted(elim.tolist()) == list(range(N))
        # position of every AMD-numbered vertex in the handle's elimination order; parents (in the AMD-ordered etree) must come later
        pinv_amd = np.zeros(N, np.int64); pinv_amd[permo] = np.arange(N)
        pos = np.zeros(N, np.int64); pos[pinv_amd[elim]] = np.arange(N)
        parent = _etree(N, PKpo, PKro)
        has = parent >= 0
        assert np.all(pos[np.nonzero(has)[0]] < pos[parent[has]]), name
        nnz_L_ref = orc.lib().orc_sparse_kkt_L_nnz(ko.ptr)
        assert rec["nnz_L"] >= nnz_L_ref, (name, rec["nnz_L"], nnz_L_ref)  # relaxed supernodes / merged leaves only ADD explicit zeros
        assert rec["nnz_L"] <= 2.5 * nnz_L_ref + 64 * N, (name, rec["nnz_L"], nnz_L_ref)
