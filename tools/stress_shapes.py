#!/usr/bin/env python3
"""Unusual assembly-tree shapes through the sparse backends (round 4: wider classification of multi-workgroup fronts, merged spines, huge-front substitution):
one factor + solve each, KKT residual by the library's own mat-vecs, solution against the oracle where it is small enough.
   python tools/stress_shapes.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scipy.sparse as sp
import torch  # noqa
import piqp_amd as hip
from oracle import pyorc as orc
from qp_gen import c3_problem, random_vars, dense_strongly_convex_qp


def arrow(n, dense_rows, seed):
    rng = np.random.default_rng(seed)
    P = sp.diags([rng.uniform(1, 2, n)], [0], format="csc")
    A = sp.csc_matrix(rng.standard_normal((dense_rows, n)))
    G = sp.eye(n, format="csc")[: n // 2]
    return (P, rng.standard_normal(n), A, np.zeros(dense_rows), G, -np.ones(n // 2), np.ones(n // 2), None, None)


def dense_as_sparse(n, p, m, seed):
    q = dense_strongly_convex_qp(n, p, m, seed=seed, double_sided=True, exact_shift=False)
    A = sp.csc_matrix(q["A"]) if p else None
    return (sp.csc_matrix(np.triu(q["P"])), q["c"], A, q["b"] if p else None, sp.csc_matrix(q["G"]), q["h_l"], q["h_u"], None, None)


# (host memory: run under `ulimit -v`; the condensed modes refuse constraint blocks whose product pattern needs more than 2e8 terms)
CASES = [("arrow n=8000, 300 dense rows", arrow(8000, 300, 1), False), ("arrow n=2000, 900 dense rows", arrow(2000, 900, 2), True),
         ("dense as sparse 600/200/400", dense_as_sparse(600, 200, 400, 3), True), ("dense as sparse 1500/0/1500", dense_as_sparse(1500, 0, 1500, 4), False),
         ("window 2900 of n=3000", c3_problem(3000, 1200, 1800, 5, 2900, 8), False), ("window 700 of n=20000", c3_problem(20000, 8000, 12000, 6, 700, 6), False)]
bad = 0
for ks in (hip.SPARSE_LDLT, 4):
    for name, a, small in CASES:
        d = hip.SparseData(*a)
        n, p, m = d.n, d.p, d.m
        t0 = time.perf_counter()
        try:
            k = hip.KKTSystem(d, hip.default_settings(kkt_solver=ks))
        except Exception as e:  # noqa: BLE001
            print(f"ks={ks} {name:32s} refused at setup: {str(e)[:120]}", flush=True)
            continue
        st = k.backend().sparse_stats()
        rng = np.random.default_rng(9)
        state = random_vars(n, p, m, rng, positive=True); rhs = random_vars(n, p, m, rng)
        ok = k.update_scalings_and_factor(False, 1e-6, 1e-4, state)
        ok2, lhs = k.solve(rhs)
        res, nrm = k.condensed_residual()
        line = f"ks={ks} {name:32s} N={st['N']} levels={st['tree_levels']} max_front={st['max_front']} factor ok {ok} solve ok {ok2} rel.res {res / nrm:.2e}"
        if small:
            ko = orc.KKTSystem(orc.Data.sparse(*a), orc.Settings(kkt_solver=ks))
            ko.update_scalings_and_factor(False, 1e-6, 1e-4, state)
            oko, lo = ko.solve(rhs)
            if oko:
                err = max(np.abs(np.asarray(lhs[key]) - np.asarray(lo[key])).max() / max(1.0, np.abs(np.asarray(lo[key])).max()) for key in ("x",))
                line += f"  |x - x_oracle| {err:.2e}"
        print(line + f"  ({time.perf_counter() - t0:.1f} s)", flush=True)
        if not (ok and ok2 and res <= 1e-10 * nrm):
            bad += 1
print("FAILED" if bad else "all shapes ok", bad)
sys.exit(1 if bad else 0)
