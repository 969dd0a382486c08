#!/usr/bin/env python3
"""Small driver for rocprofv3: a few factor+solve steps of the dense path, nothing else on the GPU.
usage: rocprofv3 --kernel-trace --stats -d OUT -- python3 tools/prof_dense.py [n] [m] [p] [steps] [kkt_solver]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401  (one HIP runtime per process)

import piqp_amd
from qp_gen import dense_strongly_convex_qp, random_vars

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
p = int(sys.argv[3]) if len(sys.argv) > 3 else 0
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
solver = int(sys.argv[5]) if len(sys.argv) > 5 else 0
q = dense_strongly_convex_qp(n, p, m, seed=43, double_sided=True, exact_shift=False)
d = piqp_amd.Data(**q)
k = piqp_amd.KKTSystem(d, piqp_amd.default_settings(kkt_solver=solver))
rng = np.random.default_rng(0)
state = {kk: torch.from_numpy(v).cuda() for kk, v in random_vars(n, p, m, rng, positive=True).items()}
rhs = {kk: torch.from_numpy(v).cuda() for kk, v in random_vars(n, p, m, rng).items()}
lhs = {kk: torch.zeros_like(v) for kk, v in rhs.items()}
for i in range(steps):
    assert k.update_scalings_and_factor(False, 1e-6, 1e-4, state)
    k.solve(rhs, lhs)
    k.solve(rhs, lhs)
res, nrm = k.condensed_residual()
print("rel residual", res / nrm)
