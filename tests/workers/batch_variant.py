"""Worker of tests/test_batch_variants_gpu.py: solves a seeded MPC batch and stores every instance's iteration count and solution (PIQP_AMD_DEBUG of the
parent selects the kernel variant; the library parses it once per process)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import piqp_amd as hip  # noqa: E402
from qp_gen import mpc_batch  # noqa: E402

out = sys.argv[1]
res = {}
for tag, kw in (("c4", dict(B=96, seed=4321)), ("wide", dict(B=12, T=12, nx=3, nu=2, seed=77))):
    B = kw.pop("B")
    mb = mpc_batch(B, **kw)
    bs = hip.BatchSparseSolver()
    assert bs.setup(mb["P_pattern"], mb["P_values"], mb["c"], mb["A_pattern"], mb["A_values"], mb["b"], x_l=mb["x_l"], x_u=mb["x_u"])
    assert bs.solve() == B
    res[tag + "_iter"] = np.asarray(bs.iterations())
    for f in ("x", "y", "z_bl", "z_bu"):
        res[tag + "_" + f] = np.asarray(bs.result(f))
np.savez(out, **res)
