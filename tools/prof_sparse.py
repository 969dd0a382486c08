#!/usr/bin/env python3
"""BASELINE configs[2] (C3): sparse QP n=50k, nnz(KKT) ~ 1M, sparse_ldlt (supernodal multifrontal LDLt) on the device next to
the CPU oracle (restatement of the reference's up-looking LDLt with the same AMD ordering).

  python tools/prof_sparse.py [--n 50000] [--reps 10] [--no-oracle]
"""
import argparse
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


from qp_gen import c3_problem as _c3  # noqa: E402


def c3_problem(n=50000, p=20000, m=30000, seed=44, spread=40, row_nnz=5):
    return _c3(n, p, m, seed, spread, row_nnz), (n, p, m)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=50000)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--spread", type=int, default=40)
    ap.add_argument("--row-nnz", type=int, default=5)
    ap.add_argument("--fixture", default=None, help="a frozen problem of tests/golden instead of the C3 generator, e.g. mm_BOYD1")
    args = ap.parse_args()
    import torch  # noqa: F401
    import piqp_amd as hip
    from qp_gen import random_vars
    scale = args.n / 50000.0
    if args.fixture:
        from qp_io import load_qp
        q = load_qp(args.fixture)
        a = (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
        n, p, m = q["P"].shape[0], (0 if q["A"] is None else q["A"].shape[0]), (0 if q["G"] is None else q["G"].shape[0])
    else:
        a, (n, p, m) = c3_problem(args.n, int(20000 * scale), int(30000 * scale), spread=args.spread, row_nnz=args.row_nnz)
    nnzK = sp.triu(a[0]).nnz + (a[2].nnz if a[2] is not None else 0) + p + (a[4].nnz if a[4] is not None else 0) + m
    t0 = time.perf_counter()
    d = hip.SparseData(*a)
    k = hip.KKTSystem(d, hip.default_settings(kkt_solver=hip.SPARSE_LDLT))
    t_setup = time.perf_counter() - t0
    be = k.backend()
    be.print_info()
    rng = np.random.default_rng(0)
    state = random_vars(n, p, m, rng, positive=True)
    rhs = random_vars(n, p, m, rng)
    for _ in range(2):
        assert k.update_scalings_and_factor(False, 1e-6, 1e-4, state)
        k.solve(rhs)
    be.set_profiling(True)
    k.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        k.update_scalings_and_factor(False, 1e-6, 1e-4, state)
    k.synchronize()
    t_fac = (time.perf_counter() - t0) / args.reps
    t0 = time.perf_counter()
    for _ in range(args.reps):
        ok, lhs = k.solve(rhs)
    k.synchronize()
    t_sol = (time.perf_counter() - t0) / args.reps
    be.set_profiling(False)
    prof = [be.get_profile(s) for s in range(3)]
    res, nrm = k.condensed_residual()
    print(f"C3 n={n} p={p} m={m} nnz(upper KKT)={nnzK}  setup {t_setup:.2f} s")
    print(f"device: factor call {t_fac * 1e3:.3f} ms (assembly {prof[0][0] / max(prof[0][1], 1):.3f} ms, numeric {prof[1][0] / max(prof[1][1], 1):.3f} ms)  "
          f"KKTSystem::solve {t_sol * 1e3:.3f} ms (backend solve {prof[2][0] / max(prof[2][1], 1):.3f} ms)  rel.res {res / nrm:.2e}")
    print(f"device: {1.0 / (t_fac + 2 * t_sol):.1f} IPM-iter KKT (1 factor + 2 solves)/s")
    if not args.no_oracle:
        from oracle import pyorc as orc
        od = orc.Data.sparse(*a)
        t0 = time.perf_counter()
        ko = orc.KKTSystem(od, orc.Settings(kkt_solver=orc.SPARSE_LDLT))
        o_setup = time.perf_counter() - t0
        ko.update_scalings_and_factor(False, 1e-6, 1e-4, state)
        t0 = time.perf_counter()
        for _ in range(3):
            ko.update_scalings_and_factor(False, 1e-6, 1e-4, state)
        o_fac = (time.perf_counter() - t0) / 3
        t0 = time.perf_counter()
        for _ in range(3):
            ko.solve(rhs)
        o_sol = (time.perf_counter() - t0) / 3
        L = orc.lib()
        print(f"oracle (1 core): setup {o_setup:.2f} s  factor {o_fac * 1e3:.3f} ms  solve {o_sol * 1e3:.3f} ms  nnz(L) {L.orc_sparse_kkt_L_nnz(ko.backend().ptr)}  "
              f"-> {1.0 / (o_fac + 2 * o_sol):.1f} IPM-iter KKT/s")


if __name__ == "__main__":
    main()
