#!/usr/bin/env python3
"""Soak test of the dense path's flag / counter / dataflow kernels: thousands of factor + solve steps at several sizes and both factorisations,
every result checked against the first one of its configuration (bitwise) -- a lost hand-off would show as a timeout (NaN) or a changed bit.
  timeout 600 python tools/soak_dense.py [rounds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401

import piqp_amd
from qp_gen import dense_strongly_convex_qp, random_vars

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
t_all = time.perf_counter()
for n, m, p, solver, reps in ((4096, 4096, 0, 0, rounds), (4096, 4096, 0, 16, rounds // 3), (1000, 700, 100, 0, rounds), (300, 200, 50, 16, rounds), (129, 64, 0, 0, rounds),
                              (2050, 1000, 0, 0, rounds // 2)):
    q = dense_strongly_convex_qp(n, p, m, seed=3, double_sided=True, exact_shift=False)
    k = piqp_amd.KKTSystem(piqp_amd.Data(**q), piqp_amd.default_settings(kkt_solver=solver))
    rng = np.random.default_rng(0)
    state = {kk: torch.from_numpy(v).cuda() for kk, v in random_vars(n, p, m, rng, positive=True).items()}
    rhs = {kk: torch.from_numpy(v).cuda() for kk, v in random_vars(n, p, m, rng).items()}
    lhs = {kk: torch.zeros_like(v) for kk, v in rhs.items()}
    ref = None
    t0 = time.perf_counter()
    for it in range(reps):
        assert k.update_scalings_and_factor(False, 1e-6, 1e-4, state)
        k.solve(rhs, lhs)
        if it % 50 == 0 or it == reps - 1:
            x = lhs["x"].cpu().numpy().copy()
            assert np.isfinite(x).all(), (n, solver, it)
            if ref is None:
                ref = x
            assert np.array_equal(x, ref), (n, solver, it, float(np.abs(x - ref).max()))
    k.synchronize()
    print(f"n={n} m={m} p={p} kkt_solver={solver}: {reps} steps ok, {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per factor + solve", flush=True)
print(f"soak ok in {time.perf_counter() - t_all:.1f} s")
