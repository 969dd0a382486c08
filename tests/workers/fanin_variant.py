#!/usr/bin/env python3
"""Worker of tests/test_sparse_fanin_gpu.py: KKT factor + solve of an "arrow" QP -- a diagonal P, a handful of dense equality rows: thousands of
single-column leaves under one small root of the assembly tree -- with whatever PIQP_AMD_DEBUG the parent set (`no_accumulators` = the
tree as the symbolic factorisation gives it, default = fan-in bounded by accumulator supernodes).  Solution, residual and the symbolic figures go to an .npz.

  python tests/workers/fanin_variant.py out.npz
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch  # noqa: F401
    import piqp_amd as hip
    from qp_gen import random_vars
    rng = np.random.default_rng(3)
    n, p = 6000, 5
    P = sp.diags(1.0 + rng.random(n)).tocsc()
    A = sp.csc_matrix(rng.standard_normal((p, n)))
    c = rng.standard_normal(n)
    b = rng.standard_normal(p)
    x_l = np.full(n, -2.0); x_u = np.full(n, 2.0)
    d = hip.SparseData(P, c, A, b, None, None, None, x_l, x_u)
    k = hip.KKTSystem(d, hip.default_settings(kkt_solver=hip.SPARSE_LDLT_MULTIFRONTAL))
    state = random_vars(n, p, 0, rng, positive=True)
    rhs = random_vars(n, p, 0, rng)
    assert k.update_scalings_and_factor(False, 1e-6, 1e-4, state)
    ok, lhs = k.solve(rhs)
    assert ok
    k.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        k.update_scalings_and_factor(False, 1e-6, 1e-4, state)
        k.solve(rhs)
    k.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    res, nrm = k.condensed_residual()
    st = k.backend().sparse_stats()
    out = {"x": np.asarray(lhs["x"]), "y": np.asarray(lhs["y"]), "rel_res": np.array([res / nrm]), "ms": np.array([ms]),
           "supernodes": np.array([st["supernodes"]]), "levels": np.array([st["tree_levels"]])}
    np.savez(sys.argv[1], **out)
    print("ok", st, ms)


if __name__ == "__main__":
    main()
