import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import piqp_amd as hip
from qp_gen import mpc_batch
for B in (8192, 1500, 300):
    mb = mpc_batch(B, seed=2024 + B)
    bs = hip.BatchSparseSolver()
    assert bs.setup(mb["P_pattern"], mb["P_values"], mb["c"], mb["A_pattern"], mb["A_values"], mb["b"], x_l=mb["x_l"], x_u=mb["x_u"])
    assert bs.solve() == B
    x0 = bs.result("x").copy(); it0 = np.asarray(bs.iterations()).copy()
    for rep in range(60):
        if rep % 7 == 3: bs.set_start_order(rep % 2 == 0)
        assert bs.solve() == B
        assert np.array_equal(np.asarray(bs.iterations()), it0)
    assert np.array_equal(bs.result("x"), x0)
    print("batch", B, "61 solves bitwise stable, kernel ms", round(bs.last_kernel_ms()[0], 3), flush=True)
print("batch soak ok")
