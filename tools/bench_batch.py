#!/usr/bin/env python3
"""Batch-size sweep of the batched solver on the C4 workload (kernel time by hipEvents)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import piqp_amd as hip
from qp_gen import mpc_batch
sizes = [int(a) for a in sys.argv[1:]] or [64, 256, 1024, 2048, 4096, 8192, 16384]
mb = mpc_batch(max(sizes), seed=1000)
for B in sizes:
    sub = {k: (v[:B] if isinstance(v, np.ndarray) and v.ndim == 2 else v) for k, v in mb.items()}
    bs = hip.BatchSparseSolver()
    bs.setup(sub["P_pattern"], sub["P_values"], sub["c"], sub["A_pattern"], sub["A_values"], sub["b"], x_l=sub["x_l"], x_u=sub["x_u"])
    bs.solve()
    ms = []
    for _ in range(3):
        solved = bs.solve(); ms.append(bs.last_kernel_ms()[0])
    its = bs.iterations()
    if os.environ.get("CONC"):
        pr_all = np.zeros((B, 8))
        for i in range(B):
            pr_all[i] = list(bs.profile(i).values()) + [0, 0] if False else np.zeros(8)
        import ctypes as C
        for i in range(B):
            out = np.zeros(8); bs.L.pq_batch_get_profile(bs.h, i, out.ctypes.data); pr_all[i] = out
        st = pr_all[:, 6]; en = st + pr_all[:, 5]
        t0 = st.min(); probe = t0 + 0.5 * (en.max() - t0)
        print("      concurrent WGs at mid-kernel:", int(((st <= probe) & (en > probe)).sum()), " at 10%:", int(((st <= t0 + 0.1 * (en.max() - t0)) & (en > t0 + 0.1 * (en.max() - t0))).sum()),
              " mean instance ms", pr_all[:, 5].mean() * 1e3)
    pr = bs.profile(0)
    print("      instance 0 (us): " + "  ".join(f"{k} {v * 1e6:.0f}" for k, v in pr.items()), f" iters {its[0]}")
    print(f"B={B:6d} solved={solved} kernel {min(ms):8.3f} ms  -> {B / min(ms) * 1e3:10.0f} QP/s   iters mean {its.mean():.2f}  us per QP-iteration-slot {min(ms) * 1e3 / its.max():.1f}", flush=True)
