#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz / *.json.  Run in the BUILD container only (needs /root/reference).

What it freezes (data only -- inputs and expected outputs; no reference source text):
  * qp_*.npz          the .mat problem files held by the reference's own tests
                      (tests/data/*.mat, benchmarks/data/*.mat; schema P,c,A,b,G,h_l,h_u,x_l,x_u --
                      utils/io_utils.hpp:75-94) re-encoded as CSC triplets in .npz
  * mm_*.npz          a few small Maros-Meszaros problems (tests/data/maros_meszaros/*.mat)
  * qp_c0_scenario_mpc.npz   the n=122 scenario-MPC QP of docs/assets/robust_scenario_mpc.ipynb,
                      rebuilt by the independent construction below (own code; the legacy NumPy
                      seed-42 stream gives the same x0) and checked against the notebook's recorded
                      header (n, p, nnz) before it is written
  * c0_trace.json     the iteration table recorded in that notebook's stored output (the only IPM
                      trace in the reference tree) + its header facts
  * kat_small.json    hand-sized known answers quoted from the reference tests
"""
import json
import os
import sys

import numpy as np
import scipy.io as sio
import scipy.sparse as sp
from scipy.linalg import solve_discrete_are
from scipy.signal import cont2discrete

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def save_qp(name, P, c, A, b, G, h_l, h_u, x_l, x_u):
    P = sp.csc_matrix(P); P.sort_indices()
    n = P.shape[0]
    A = sp.csc_matrix(A if A is not None else np.zeros((0, n))); A.sort_indices()
    G = sp.csc_matrix(G if G is not None else np.zeros((0, n))); G.sort_indices()
    f64 = lambda v, k: np.full(k, np.nan) if v is None else np.asarray(v, dtype=np.float64).reshape(-1)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        n=n, p=A.shape[0], m=G.shape[0],
        P_indptr=P.indptr.astype(np.int32), P_indices=P.indices.astype(np.int32), P_data=P.data.astype(np.float64),
        A_indptr=A.indptr.astype(np.int32), A_indices=A.indices.astype(np.int32), A_data=A.data.astype(np.float64),
        G_indptr=G.indptr.astype(np.int32), G_indices=G.indices.astype(np.int32), G_data=G.data.astype(np.float64),
        c=f64(c, n), b=f64(b, A.shape[0]), h_l=f64(h_l, G.shape[0]), h_u=f64(h_u, G.shape[0]),
        x_l=f64(x_l, n), x_u=f64(x_u, n))


def convert_mat(path, name):
    d = sio.loadmat(path)
    g = lambda k: d[k] if k in d else None
    save_qp(name, g("P"), g("c"), g("A"), g("b"), g("G"), g("h_l"), g("h_u"), g("x_l"), g("x_u"))


# ----------------------------------------------------------------------------------------------
# C0: scenario-tree MPC for a chain of M masses, Ns scenarios that share (x0, u0), horizon N.
# Variable layout (own derivation; matches the notebook's so that the recorded trace applies):
#   per scenario s: [x_1,u_1, x_2,u_2, ..., x_{N-1},u_{N-1}, x_N]  then the shared [x_0, u_0] last.
def chain_mass(M, k, m=1.0, c=0.1, dt=0.5):
    nx, nu = 2 * M, M - 1
    T = -2.0 * k * np.eye(M) + k * np.eye(M, k=-1) + k * np.eye(M, k=1)
    Ac = np.zeros((nx, nx))
    Ac[:M, M:] = np.eye(M)
    Ac[M:, :M] = T / m
    Ac[M:, M:] = -2.0 * c * np.eye(M) / m
    Bc = np.zeros((nx, nu))
    Bc[nx - nu:, :] = np.eye(nu)
    Ad, Bd = cont2discrete((Ac, Bc, np.eye(nx), np.zeros((nx, nu))), dt, method="zoh")[:2]
    Q, R = 1e3 * np.eye(nx), 1e-1 * np.eye(nu)
    QN = solve_discrete_are(Ad, Bd, Q, R)
    return Ad, Bd, Q, R, QN


def scenario_mpc(M=3, N=5, Ns=3, seed=42, x_max=4.0, u_max=0.5):
    np.random.seed(seed)  # legacy global stream, as the notebook does
    nx, nu = 2 * M, M - 1
    systems = [chain_mass(M, k) for k in np.linspace(1.0, 2.0, Ns)]
    x0 = np.random.uniform(-1.0, 1.0, nx)
    blk = (N - 1) * (nx + nu) + nx
    n = Ns * blk + nx + nu
    p = Ns * N * nx
    # dense blocks are inserted with ALL their entries (explicit zeros included): the notebook's
    # slice-assignment into a csc_matrix does the same, which is what makes its header read
    # nnz(P upper)=375 and nnz(A)=1260 -- the sparsity PATTERN is part of the fixture.
    Pt, At = [], []

    def put(T, r0, c0, B):
        B = np.asarray(B, dtype=np.float64)
        rr, cc = np.meshgrid(np.arange(B.shape[0]) + r0, np.arange(B.shape[1]) + c0, indexing="ij")
        T.append((rr.ravel(), cc.ravel(), B.ravel()))

    x_l, x_u = np.zeros(n), np.zeros(n)
    root = n - (nx + nu)
    # shared first stage
    put(Pt, root, root, systems[0][2])
    put(Pt, root + nx, root + nx, systems[0][3])
    x_l[root:root + nx] = x0; x_u[root:root + nx] = x0
    x_l[root + nx:] = -u_max; x_u[root + nx:] = u_max
    for s, (Ad, Bd, Q, R, QN) in enumerate(systems):
        base = s * blk
        xs = lambda i: root if i == 0 else base + (i - 1) * (nx + nu)       # start of x_i
        us = lambda i: root + nx if i == 0 else base + (i - 1) * (nx + nu) + nx  # start of u_i
        for i in range(1, N):
            put(Pt, xs(i), xs(i), Q / Ns)
            put(Pt, us(i), us(i), R / Ns)
            x_l[us(i):us(i) + nu] = -u_max; x_u[us(i):us(i) + nu] = u_max
        put(Pt, xs(N), xs(N), QN / Ns)
        for i in range(N):
            r = s * N * nx + i * nx
            put(At, r, xs(i), Ad)
            put(At, r, us(i), Bd)
            put(At, r, xs(i + 1), -np.eye(nx))
            x_l[xs(i + 1):xs(i + 1) + nx] = -x_max; x_u[xs(i + 1):xs(i + 1) + nx] = x_max

    def assemble(T, shape):
        r = np.concatenate([t[0] for t in T]); c = np.concatenate([t[1] for t in T]); v = np.concatenate([t[2] for t in T])
        M = sp.coo_matrix((v, (r, c)), shape=shape).tocsc()  # keeps explicit zeros
        M.sort_indices()
        return M
    return assemble(Pt, (n, n)), np.zeros(n), assemble(At, (p, n)), np.zeros(p), x_l, x_u


# the iteration table stored in docs/assets/robust_scenario_mpc.ipynb (sparse_ldlt run), SURVEY.md A.6
C0_TRACE = """
0 3.25459e+02 -1.09791e+06 1.09824e+06 1.93609e-03 6.63672e+02 1.000e-06 1.000e-04 1.177e+04 0.0000 0.0000
1 7.50453e+02 -2.77013e+05 2.77764e+05 1.83182e-03 2.15106e+01 1.450e-07 1.450e-05 1.706e+03 0.8673 0.9900
2 3.18009e+03 -2.17709e+04 2.49510e+04 8.92483e-04 2.82825e+01 5.695e-08 1.278e-06 1.504e+02 0.8896 0.9398
3 3.56810e+03 2.22331e+03 1.34479e+03 3.99874e-04 1.98075e+02 2.444e-08 1.825e-07 2.148e+01 0.6191 0.9808
4 4.22500e+03 4.04344e+03 1.81561e+02 9.04618e-05 1.01304e+02 5.476e-09 4.090e-08 4.813e+00 0.7965 0.9681
5 4.46048e+03 4.38640e+03 7.40831e+01 1.69913e-06 4.08054e+00 4.367e-10 3.262e-09 3.838e-01 0.9784 0.9075
6 4.45291e+03 4.44882e+03 4.09229e+00 4.39455e-08 8.67742e-02 1.000e-10 1.625e-10 1.912e-02 0.9678 0.9702
7 4.45183e+03 4.45162e+03 2.14879e-01 8.47502e-10 1.09070e-01 1.000e-10 1.000e-10 9.685e-04 0.9793 0.9697
8 4.45174e+03 4.45173e+03 1.19518e-02 8.97259e-11 2.23268e-02 1.000e-10 1.000e-10 5.190e-05 0.9884 0.9834
9 4.45173e+03 4.45173e+03 1.33397e-03 3.31985e-11 2.23268e-04 1.000e-10 1.000e-10 5.501e-06 0.9900 0.9900
10 4.45173e+03 4.45173e+03 1.95468e-04 3.10567e-11 2.23263e-06 1.000e-10 1.000e-10 8.052e-07 0.9900 0.9900
11 4.45173e+03 4.45173e+03 2.80912e-05 1.25559e-11 2.23069e-08 1.000e-10 1.000e-10 1.157e-07 0.9900 0.9900
12 4.45173e+03 4.45173e+03 3.83423e-06 4.68808e-12 2.15834e-10 1.000e-10 1.000e-10 1.578e-08 0.9900 0.9900
"""


def main():
    td = os.path.join(REF, "tests", "data")
    for f in ("small_dense", "small_sparse_dual_inf", "scenario_mpc_small", "scenario_mpc", "chain_mass_sqp",
              "robot_arm_sqp", "robot_arm_sqp_constr_perm", "robot_arm_sqp_no_global"):
        convert_mat(os.path.join(td, f + ".mat"), "qp_" + f)
    # a handful of small Maros-Meszaros problems (status==SOLVED contract, maros_meszaros_tests.cpp)
    for f in ("HS21", "HS35", "HS53", "HS76", "HS118", "DUAL1", "DUALC1", "PRIMAL1", "QAFIRO", "LOTSCHD", "CVXQP1_S",
              "GENHS28", "TAME", "ZECEVIC2", "DUAL4", "QPTEST", "HS268", "QSCAGR7", "PRIMALC1", "CVXQP2_S", "AUG3D"):
        path = os.path.join(td, "maros_meszaros", f + ".mat")
        if os.path.exists(path):
            convert_mat(path, "mm_" + f)
    # the larger Maros-Meszaros problems SURVEY.md 8d names as the real cross-checks of the sparse configuration (C3): CONT-201 (n = 40 397,
    # p = 40 198: a PDE-constrained grid), BOYD1 (n = 93 261 with 18 dense equality rows), AUG3DCQP (3-D augmented system), LISWET1 (m = 10 000
    # banded inequalities)
    for f in ("CONT-201", "BOYD1", "AUG3DCQP", "LISWET1"):
        convert_mat(os.path.join(td, "maros_meszaros", f + ".mat"), "mm_" + f)
    # the problem sets of the reference's own sweeps (tests/src/sparse/maros_meszaros_tests.cpp: status == SOLVED on every file;
    # netlib_lp_tests.cpp: SOLVED on data/, PRIMAL or DUAL INFEASIBLE on infeas/), every file up to a size limit that keeps the fixtures small:
    # Maros-Meszaros <= 100 KB (107 of 137), netlib <= 50 KB (78 of 94 feasible, 25 of 29 infeasible)
    import glob
    for path in sorted(glob.glob(os.path.join(td, "maros_meszaros", "*.mat"))):
        if os.path.getsize(path) <= 100e3:
            convert_mat(path, "mm_" + os.path.basename(path)[:-4])
    for sub, pre in (("data", "nl_"), ("infeas", "nli_")):
        for path in sorted(glob.glob(os.path.join(td, "netlib", sub, "*.mat"))):
            if os.path.getsize(path) <= 50e3:
                convert_mat(path, pre + os.path.basename(path)[:-4])

    P, c, A, b, x_l, x_u = scenario_mpc()
    n, p = P.shape[0], A.shape[0]
    nnzP_u, nnzA = int(np.sum(P.tocoo().row <= P.tocoo().col)), A.nnz
    # header recorded in the notebook output: n=122, nnz(P utri)=375, p=90, nnz(A)=1260, n_x_l=n_x_u=122
    assert (n, p, nnzP_u, nnzA) == (122, 90, 375, 1260), (n, p, nnzP_u, nnzA)
    save_qp("qp_c0_scenario_mpc", P, c, A, b, None, None, None, x_l, x_u)
    rows = [[float(t) for t in line.split()] for line in C0_TRACE.strip().splitlines()]
    json.dump(dict(source="docs/assets/robust_scenario_mpc.ipynb stored output (sparse_ldlt, PIQP >=0.6.0,<0.6.2)",
                   columns=["iter", "prim_obj", "dual_obj", "duality_gap", "prim_res", "dual_res", "rho", "delta", "mu",
                            "p_step", "d_step"],
                   rows=rows, iterations=12, objective=4.45173e+03, objective_scipy_trust_constr=4451.7305,
                   n=122, p=90, nnz_P_utri=375, nnz_A=1260, n_x_l=122, n_x_u=122,
                   multistage_block_info=[[8, 6], [8, 6], [8, 6], [14, 0]] * 3, multistage_arrow_width=8),
              open(os.path.join(OUT, "c0_trace.json"), "w"), indent=1)

    json.dump({
        "dense_simple_qp": {  # tests/src/dense/solver_test.cpp:31-97
            "x": [0.4285714, 0.2142857], "y": [-1.5714286], "tol": 1e-6,
            "x_after_update": [0.2763157, 0.0921056], "y_after_update": [-1.2105263]},
        "infinity_bounds": {"x": [-0.5, -1.0, -0.5, -1.0]},  # dense/solver_test.cpp:347-377
        "amd_4x4": {  # tests/src/sparse/utils_test.cpp:55-92
            "ordering": [1, 2, 0, 3], "Ai_to_Ci": [3, 0, 2, 1, 5, 4, 6]},
    }, open(os.path.join(OUT, "kat_small.json"), "w"), indent=1)
    print("fixtures written to", OUT)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    main()
