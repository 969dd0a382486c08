#!/usr/bin/env python3
"""Worker of tests/test_sparse_variants_gpu.py: factor + solve of one seeded sparse KKT system (and a multistage chain) with whatever
PIQP_AMD_* schedule toggles the parent put into the environment; the solutions go to an .npz for a bitwise comparison.

  python tests/workers/sparse_variant.py out.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch  # noqa: F401
    import piqp_amd as hip
    from qp_gen import c3_problem, mpc_chain, random_vars
    out = {}
    if len(sys.argv) > 2 and sys.argv[2] == "huge":
        # C3 recipe with rows of 10 nonzeros in 1500-variable windows at n = 8000: a chain of fronts of 2600-3700 rows with 130-380 pivots (and accumulator
        # supernodes between them) -- multi-workgroup fronts in the factorisation, huge fronts (k_front_fwd_rows / k_front_bwd_cols) in the substitution
        n, p, m = 8000, 3200, 4800
        a = c3_problem(n, p, m, 44, 1500, 10)
        k = hip.KKTSystem(hip.SparseData(*a), hip.default_settings(kkt_solver=hip.SPARSE_LDLT_MULTIFRONTAL))
        st = k.backend().sparse_stats()
        assert st["max_front"] >= 2048, st
        rng = np.random.default_rng(0)
        state = random_vars(n, p, m, rng, positive=True)
        rhs = random_vars(n, p, m, rng)
        assert k.update_scalings_and_factor(False, 1e-6, 1e-4, state)
        ok, lhs = k.solve(rhs)
        assert ok
        res, nrm = k.condensed_residual()
        out["rel_residual"] = np.array([res / nrm])
        for key, v in lhs.items():
            out["wide_" + key] = np.asarray(v)
        np.savez(sys.argv[1], **out)
        print("ok", sorted(out))
        return
    # (a) C3 recipe at n = 6000: fronts wide enough for every schedule variant to have work
    n, p, m = 6000, 2400, 3600
    a = c3_problem(n, p, m, 44, 40)
    k = hip.KKTSystem(hip.SparseData(*a), hip.default_settings(kkt_solver=hip.SPARSE_LDLT_MULTIFRONTAL))
    rng = np.random.default_rng(0)
    state = random_vars(n, p, m, rng, positive=True)
    rhs = random_vars(n, p, m, rng)
    assert k.update_scalings_and_factor(False, 1e-6, 1e-4, state)
    ok, lhs = k.solve(rhs)
    assert ok
    for key, v in lhs.items():
        out["c3_" + key] = np.asarray(v)
    # (b) a block-tridiagonal chain (the shape of BASELINE configs[4]) through the same multifrontal backend
    q = mpc_chain(6, 3, 400, 5)
    k2 = hip.KKTSystem(hip.SparseData(*q), hip.default_settings(kkt_solver=hip.SPARSE_LDLT_MULTIFRONTAL))
    n2, p2, m2 = q[0].shape[0], q[2].shape[0], (q[4].shape[0] if q[4] is not None else 0)
    state2 = random_vars(n2, p2, m2, rng, positive=True)
    rhs2 = random_vars(n2, p2, m2, rng)
    assert k2.update_scalings_and_factor(False, 1e-6, 1e-4, state2)
    ok, lhs2 = k2.solve(rhs2)
    assert ok
    for key, v in lhs2.items():
        out["chain_" + key] = np.asarray(v)
    # (c) a tree with real separators (frozen Maros-Meszaros CONT-101): big fronts on the batched dense kernels, panel fronts, wide-front substitution
    from qp_io import load_qp
    q3 = load_qp("mm_CONT-101")
    a3 = (q3["P"], q3["c"], q3["A"], q3["b"], q3["G"], q3["h_l"], q3["h_u"], q3["x_l"], q3["x_u"])
    d3 = hip.SparseData(*a3)
    k3 = hip.KKTSystem(d3, hip.default_settings(kkt_solver=hip.SPARSE_LDLT_MULTIFRONTAL))
    state3 = random_vars(d3.n, d3.p, d3.m, rng, positive=True)
    rhs3 = random_vars(d3.n, d3.p, d3.m, rng)
    assert k3.update_scalings_and_factor(False, 1e-6, 1e-4, state3)
    ok, lhs3 = k3.solve(rhs3)
    assert ok
    for key, v in lhs3.items():
        out["cont_" + key] = np.asarray(v)
    np.savez(sys.argv[1], **out)
    print("ok", sorted(out))


if __name__ == "__main__":
    main()
