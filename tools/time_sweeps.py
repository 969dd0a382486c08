#!/usr/bin/env python3
"""Time of the triangular sweeps of one dense backend solve (stage 5 of the backend's profiler), n = 4096 by default; no result check (for schedule experiments).
   PIQP_AMD_DEBUG=... python3 tools/time_sweeps.py [n] [kkt_solver]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import piqp_amd as hip
from qp_gen import dense_strongly_convex_qp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ks = int(sys.argv[2]) if len(sys.argv) > 2 else 0
q = dense_strongly_convex_qp(n, 0, n, seed=7 + n, double_sided=True, exact_shift=False)
k = hip.DenseKKT(hip.Data(**q), kkt_solver=ks)
rng = np.random.default_rng(n)
assert k.update_scalings_and_factor(1e-4, np.full(n, 1e-6), rng.uniform(0.5, 2.0, n))
rhs = rng.standard_normal(n)
for _ in range(3):
    k.solve(rhs, np.zeros(0), np.zeros(n))
k.set_profiling(2)
for _ in range(20):
    k.solve(rhs, np.zeros(0), np.zeros(n))
k.synchronize()
ms, cnt = k.get_profile(5)
print(f"sweeps of one solve: {ms / max(cnt, 1) * 1e3:.1f} us ({cnt} solves)  PIQP_AMD_DEBUG={os.environ.get('PIQP_AMD_DEBUG', '')}")
