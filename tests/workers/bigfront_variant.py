#!/usr/bin/env python3
"""Worker of tests/test_sparse_bigfront_gpu.py: KKT factor + solve (sparse_ldlt) of a synthetic QP whose assembly tree has fronts of several hundred rows,
with whatever PIQP_AMD_DEBUG the parent set (`no_big` = every front through one workgroup's pivot loop; default = big fronts on the batched dense
kernels, panel fronts staged in LDS, wide fronts through the blocked substitution).  Solution, residual and symbolic figures go to an .npz.

  python tests/workers/bigfront_variant.py {grid|dense_rows|dense_rows_big} out.npz
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def problem(kind):
    rng = np.random.default_rng(17)
    if kind == "grid":
        # PDE-constrained: state y on an nx x ny grid, control u, 5-point Laplacian y - coupling u = b (CONT-xxx shape); box on u
        nx, ny = 136, 120
        N = nx * ny
        ex, ey = np.ones(nx), np.ones(ny)
        T = lambda e: sp.diags([-e[:-1], 2.0 * e, -e[:-1]], [-1, 0, 1])  # noqa: E731
        Lap = (sp.kron(sp.eye(ny), T(ex)) + sp.kron(T(ey), sp.eye(nx))).tocsc()
        n = 2 * N
        P = sp.diags(np.concatenate([np.full(N, 1.0), np.full(N, 1e-2)])).tocsc()
        A = sp.hstack([Lap, -sp.eye(N)]).tocsc()
        c = rng.standard_normal(n); b = rng.standard_normal(N)
        x_l = np.concatenate([np.full(N, -1e30), np.full(N, -1.0)]); x_u = np.concatenate([np.full(N, 1e30), np.full(N, 1.0)])
        return (P, c, A, b, None, None, None, x_l, x_u), n, N, 0
    # a few hundred dense equality rows: the root front has that many pivots (several 128-column panels, the last one ragged) over a wide child level;
    # dense_rows_big: 1150 of them -- a front beyond the 1040 / 768 rows the wide substitution kernels keep in registers per block (their in-line tails)
    n, p = (4000, 1150) if kind == "dense_rows_big" else (3000, 333)
    P = sp.diags(0.5 + rng.random(n)).tocsc()
    A = sp.csc_matrix(rng.standard_normal((p, n)) * (rng.random((p, n)) < 0.6))
    G = sp.random(200, n, density=0.01, random_state=5, format="csc")
    c = rng.standard_normal(n); b = rng.standard_normal(p)
    h_u = np.abs(rng.standard_normal(200)) + 1.0
    return (P, c, A, b, G, np.full(200, -1e30), h_u, None, None), n, p, 200


def main():
    import torch  # noqa: F401
    import piqp_amd as hip
    from qp_gen import random_vars
    kind, outp = sys.argv[1], sys.argv[2]
    a, n, p, m = problem(kind)
    d = hip.SparseData(*a)
    k = hip.KKTSystem(d, hip.default_settings(kkt_solver=hip.SPARSE_LDLT_MULTIFRONTAL))
    rng = np.random.default_rng(23)
    state = random_vars(d.n, d.p, d.m, rng, positive=True)
    rhs = random_vars(d.n, d.p, d.m, rng)
    out = {}
    for tag, (rho, delta) in (("a", (1e-6, 1e-4)), ("b", (1e-10, 1e-10))):  # an early state and the hardest regularisation of the interior-point loop
        assert k.update_scalings_and_factor(False, rho, delta, state)
        ok, lhs = k.solve(rhs)
        assert ok
        res, nrm = k.condensed_residual()
        out["x_" + tag] = np.asarray(lhs["x"]); out["y_" + tag] = np.asarray(lhs["y"]); out["res_" + tag] = np.array([res / nrm])
    st = k.backend().sparse_stats()
    out["max_front"] = np.array([st["max_front"]]); out["levels"] = np.array([st["tree_levels"]])
    np.savez(outp, **out)
    print("ok", kind, st, out["res_a"], out["res_b"])


if __name__ == "__main__":
    main()
