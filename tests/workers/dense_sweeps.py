#!/usr/bin/env python3
"""Worker of tests/test_dense_gpu.py::test_sweeps_with_block_inverses_against_the_substitution_sweeps: one dense factorisation, five backend solves on the same
handle with the sweep schedule PIQP_AMD_DEBUG selects; residual of each against the device's own factor in extended precision.
   python tests/workers/dense_sweeps.py n kkt_solver tag"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import piqp_amd as hip
from qp_gen import dense_strongly_convex_qp

n, ks, tag = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
q = dense_strongly_convex_qp(n, 0, n // 2, seed=7 + n, double_sided=True, exact_shift=False)
k = hip.DenseKKT(hip.Data(**q), kkt_solver=ks)
rng = np.random.default_rng(n)
assert k.update_scalings_and_factor(1e-4, np.full(n, 1e-6), rng.uniform(0.5, 2.0, n // 2))
F = np.tril(k.internal_factor()).astype(np.longdouble)
res, last = [], None
for i in range(5):
    rhs = rng.standard_normal(n)
    lx, _, _ = k.solve(rhs, np.zeros(0), np.zeros(n // 2))
    xl = lx.astype(np.longdouble)
    if ks == 16:
        D = np.diag(F).copy(); Lm = F.copy(); np.fill_diagonal(Lm, 1.0)
        Kx = Lm @ (D * (Lm.T @ xl))
    else:
        Kx = F @ (F.T @ xl)
    res.append(float(np.abs(Kx - rhs).max() / np.abs(rhs).max()))
    last = (rhs, lx)
again, _, _ = k.solve(last[0], np.zeros(0), np.zeros(n // 2))
path = os.path.join(tempfile.gettempdir(), f"dense_sweeps_{tag}_{n}_{ks}.npy")
np.save(path, np.asarray(last[1]))
print("RESULT " + json.dumps(dict(res=res, x=path, repeat_bitwise=bool(np.array_equal(again, last[1])))), flush=True)
