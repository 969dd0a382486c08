#!/usr/bin/env python3
"""In-kernel device-clock split of the batched interior-point kernel (k_batch_ipm): mean over the instances of a batch of the time one instance spends in
assembly / chain factorisation / chain substitution / KKTSystem::solve / residuals / whole solve.   python tools/prof_batch_split.py [batch ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import piqp_amd as hip
from qp_gen import mpc_batch

for B in [int(a) for a in sys.argv[1:]] or [1024, 8192]:
    mb = mpc_batch(B, seed=1000)
    bs = hip.BatchSparseSolver()
    assert bs.setup(mb["P_pattern"], mb["P_values"], mb["c"], mb["A_pattern"], mb["A_values"], mb["b"], x_l=mb["x_l"], x_u=mb["x_u"])
    bs.solve(); bs.solve()
    P = np.array([list(bs.profile(i).values()) for i in range(0, B, max(B // 256, 1))])
    m = P.mean(axis=0) * 1e3
    print(f"B={B:6d} kernel {bs.last_kernel_ms()[0]:.3f} ms | per instance (ms): assembly {m[0]:.3f} chain factor {m[1]:.3f} chain solve {m[2]:.3f} KKTSystem::solve total {m[3]:.3f} "
          f"residuals {m[4]:.3f} whole {m[5]:.3f}", flush=True)
