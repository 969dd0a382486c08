#!/usr/bin/env python3
"""Soak of the sparse backend on a frozen fixture: `reps` x (factor + two solves) on the same state must reproduce the first solution bit for bit
(two streams, events, batched multi-workgroup fronts, flag-ordered sweeps: a race would show as a differing bit sooner or later).
usage: python tools/soak_sparse.py [fixture=mm_CONT-201] [reps=300]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
import piqp_amd as hip  # noqa: E402
from qp_gen import random_vars  # noqa: E402
from qp_io import load_qp  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "mm_CONT-201"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
q = load_qp(name)
d = hip.SparseData(q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
k = hip.KKTSystem(d, hip.default_settings(kkt_solver=hip.SPARSE_LDLT))
rng = np.random.default_rng(5)
states = [random_vars(d.n, d.p, d.m, rng, positive=True) for _ in range(2)]
rhs = [random_vars(d.n, d.p, d.m, rng) for _ in range(2)]
ref = {}
bad = 0
for it in range(reps):
    s = it & 1
    assert k.update_scalings_and_factor(False, 1e-6 if s == 0 else 1e-10, 1e-4 if s == 0 else 1e-10, states[s])
    for r in range(2):
        ok, lhs = k.solve(rhs[r])
        assert ok
        key = (s, r)
        cur = np.concatenate([np.asarray(lhs["x"]), np.asarray(lhs["y"])])
        if key not in ref:
            ref[key] = cur.copy()
        elif not np.array_equal(cur, ref[key]):
            bad += 1
            print("iteration", it, key, "differs: max abs", np.abs(cur - ref[key]).max())
print(f"{name}: {reps} x (factor + 2 solves), two alternating states: {'bitwise reproducible' if bad == 0 else str(bad) + ' MISMATCHES'}")
sys.exit(1 if bad else 0)
