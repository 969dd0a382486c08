#!/usr/bin/env python3
"""Why does the oracle end MAX_ITER on five netlib problems the reference's own test expects SOLVED / INFEASIBLE
(/root/reference/tests/src/sparse/netlib_lp_tests.cpp:37,54; tests/test_oracle_sparse.py ORACLE_MISSES_REFERENCE)?

The oracle is not touched.  What is varied is what the reference leaves to its third-party parts and to the build:
  * the fill-reducing ORDERING: the restated AMD (oracle/orc_sparse.c, after sparse/ordering.hpp:67-84 = Eigen::AMDOrdering, which is not in /root/reference) breaks
    ties by position, so the same problem with its variables and constraints renumbered (a random symmetric permutation of x, of the rows of A and of G -- the
    same LP) is factored in a different order, as it would be by any other AMD implementation;
  * the ROUNDING: the second legal build of the same sources (FMA contraction on, oracle/Makefile `fma`);
  * the equilibration: preconditioner_iter = 0 (a setting of the reference, solver.hpp:151-308);
  * the dense backend's arithmetic: the same LP through the oracle's DENSE solver (no ordering, no sparse LDLt at all) where it is small enough.
If the status flips under renumbering or rounding, the miss is a property of the trajectory on a degenerate LP and cannot be pinned without Eigen; if nothing
moves it, the restatement of the interior-point loop itself is wrong on these inputs.   python tools/exp_oracle_misses.py > profiles/r06_oracle_misses.txt"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyorc as orc  # noqa: E402
from qp_io import load_qp, dense_args  # noqa: E402

MISSES = ["nl_bnl2", "nl_pilot-we", "nli_ceria3d", "nli_cplex2", "nli_qual"]
CONTROLS = ["nl_afiro", "nl_share2b", "nl_fffff800", "nli_itest2", "nli_forest6"]  # problems the oracle gets right: do THEY move?
STATUS = {1: "SOLVED", -1: "MAX_ITER", -2: "PRIMAL_INF", -3: "DUAL_INF", -8: "NUMERICS", -9: "UNSOLVED", -10: "INVALID"}


def permuted(q, seed):
    rng = np.random.default_rng(seed)
    n = q["P"].shape[0]
    px = rng.permutation(n)
    out = dict(q)
    out["P"] = sp.csc_matrix(q["P"][px][:, px]); out["P"] = sp.csc_matrix(sp.triu(out["P"] + out["P"].T - sp.diags(out["P"].diagonal())))
    out["c"] = q["c"][px]
    for v in ("x_l", "x_u"):
        out[v] = None if q[v] is None else q[v][px]
    if q["A"] is not None:
        pa = rng.permutation(q["A"].shape[0])
        out["A"] = sp.csc_matrix(q["A"][pa][:, px]); out["b"] = q["b"][pa]
    if q["G"] is not None:
        pg = rng.permutation(q["G"].shape[0])
        out["G"] = sp.csc_matrix(q["G"][pg][:, px])
        for v in ("h_l", "h_u"):
            out[v] = None if q[v] is None else q[v][pg]
    return out


def run(q, L=None, dense=False, **settings):
    so = orc.Solver(_L=L)
    so.settings.infeasibility_threshold = 0.01
    if not dense:
        so.settings.kkt_solver = orc.SPARSE_LDLT
    for k, v in settings.items():
        setattr(so.settings, k, v)
    t = time.time()
    if dense:
        ok = so.setup(*dense_args(q), sparse=False)
    else:
        ok = so.setup(q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"], sparse=True)
    if not ok:
        return "setup failed"
    st = so.solve()
    return f"{STATUS.get(int(st), int(st))}/{int(so.info.iter)}"


def main():
    names = [a for a in sys.argv[1:] if not a.startswith("-")] or MISSES + CONTROLS
    Lf = orc.lib_fma()
    print("# status/iterations of the ORACLE (untouched) on the same LP under variations the reference leaves to Eigen / the compiler; expected: nl_* SOLVED, nli_* PRIMAL_INF or DUAL_INF")
    print(f"# {'problem':14s} {'n':>6s} {'p':>6s} {'m':>6s} | {'as frozen':>14s} {'FMA build':>14s} {'no Ruiz':>14s} {'dense solver':>14s} | renumbered (seeds 1..6)")
    for name in names:
        q = load_qp(name)
        n = q["P"].shape[0]; p = 0 if q["A"] is None else q["A"].shape[0]; m = 0 if q["G"] is None else q["G"].shape[0]
        base = run(q); fma = run(q, L=Lf); noruiz = run(q, preconditioner_iter=0)
        dn = run(q, dense=True) if n + p + m <= 6000 and n <= 3000 else "-"
        perms = [run(permuted(q, s)) for s in range(1, 7)]
        print(f"  {name:14s} {n:6d} {p:6d} {m:6d} | {base:>14s} {fma:>14s} {noruiz:>14s} {dn:>14s} | " + " ".join(f"{x:>13s}" for x in perms), flush=True)


if __name__ == "__main__":
    main()
