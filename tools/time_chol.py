#!/usr/bin/env python3
"""hipEvent time of the persistent factorisation launch (default: C2, n = 4096) for the schedule variant in PIQP_AMD_DEBUG:  python tools/time_chol.py [steps] [n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401

import piqp_amd
from qp_gen import dense_strongly_convex_qp, random_vars

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
q = dense_strongly_convex_qp(n, 0, n, seed=3, double_sided=True, exact_shift=False)
k = piqp_amd.KKTSystem(piqp_amd.Data(**q), piqp_amd.default_settings(kkt_solver=0))
rng = np.random.default_rng(0)
state = {kk: torch.from_numpy(v).cuda() for kk, v in random_vars(n, 0, n, rng, positive=True).items()}
rhs = {kk: torch.from_numpy(v).cuda() for kk, v in random_vars(n, 0, n, rng).items()}
lhs = {kk: torch.zeros_like(v) for kk, v in rhs.items()}
b = k.backend()
for _ in range(5):
    assert k.update_scalings_and_factor(False, 1e-6, 1e-4, state)
b.set_profiling(2)
for _ in range(steps):
    assert k.update_scalings_and_factor(False, 1e-6, 1e-4, state)
k.solve(rhs, lhs)
k.synchronize()
b.set_profiling(0)
p = [b.get_profile(s) for s in range(6)]
print(f"PIQP_AMD_DEBUG={os.environ.get('PIQP_AMD_DEBUG', '')!r}: n = {n}: persistent launch {p[3][0] / max(p[3][1], 1):.4f} ms ({p[3][1]} launches), factorisation stage {p[1][0] / max(p[1][1], 1):.4f} ms, x[0] = {float(lhs['x'][0].cpu()):.17g}")
