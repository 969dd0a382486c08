#!/usr/bin/env python3
"""Worker of tests/test_ruiz_gpu.py: whole solves (setup, solve, update of matrices + bounds, solve, matrix-free update that disables a
row of G, solve) of dense and sparse fixtures with the equilibration wherever the parent's PIQP_AMD_DEBUG puts it (device kernels by
default, the host routine under `host_ruiz`).  Solutions, iteration counts and the scaled-back data land in an .npz for a bitwise comparison.

  python tests/workers/ruiz_variant.py out.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def record(out, tag, s, status):
    r = s.result()
    out[tag + "_status"] = np.array([status, s.info.iter])
    for k in ("x", "y", "z_l", "z_u", "z_bl", "z_bu", "s_l", "s_u"):
        out[tag + "_" + k] = np.asarray(r[k])
    i = s.info
    out[tag + "_res"] = np.array([i.primal_res, i.dual_res, i.primal_obj, i.dual_obj])


def run_case(out, tag, hip, q, sparse, kkt_solver, scale_cost):
    from qp_io import dense_args
    args = [q[k] for k in ("P", "c", "A", "b", "G", "h_l", "h_u", "x_l", "x_u")] if sparse else list(dense_args(q))
    s = (hip.SparseSolver if sparse else hip.DenseSolver)()
    s.settings.kkt_solver = kkt_solver
    s.settings.preconditioner_scale_cost = scale_cost
    assert s.setup(*args)
    record(out, tag + "_setup", s, s.solve())
    # update(): new P and G (same pattern), new bounds -> unscale, replace, fresh equilibration (solver.hpp:218-308)
    P, c, A, b, G, h_l, h_u, x_l, x_u = args
    rng = np.random.default_rng(5)
    if sparse:
        P2 = P.copy(); P2.data = P2.data * 1.5
        G2 = None
        if G is not None:
            G2 = G.copy(); G2.data = G2.data * (1.0 + 0.25 * rng.random(G2.data.size))
    else:
        P2 = P * 1.5
        G2 = None if G is None else G * (1.0 + 0.25 * rng.random(G.shape))
    c2 = c * 0.5
    assert s.update(P=P2, c=c2, G=G2)
    record(out, tag + "_update", s, s.solve())
    # the same with the scaling kept (preconditioner_reuse_on_update)
    s.settings.preconditioner_reuse_on_update = 1
    assert s.update(P=P, c=c)
    record(out, tag + "_reuse", s, s.solve())
    if G is not None and h_l is not None and h_u is not None:
        # a matrix-free update that leaves row 0 of G without a finite bound: the row is zeroed where the scaled matrix lives
        hl3, hu3 = np.array(h_l, dtype=float), np.array(h_u, dtype=float)
        hl3[0], hu3[0] = -np.inf, np.inf
        assert s.update(h_l=hl3, h_u=hu3)
        record(out, tag + "_rowoff", s, s.solve())
    c3 = s.clone()
    record(out, tag + "_clone", c3, c3.solve())


def main():
    import torch  # noqa: F401
    import piqp_amd as hip
    from qp_io import load_qp
    out = {}
    for name, scale_cost in (("qp_small_dense", 0), ("qp_small_dense", 1), ("mm_QAFIRO", 1), ("mm_HS21", 0), ("mm_DUAL1", 1)):
        q = load_qp(name)
        run_case(out, f"dense_{name}_{scale_cost}", hip, q, False, hip.DENSE_CHOLESKY, scale_cost)
    for name, scale_cost in (("qp_chain_mass_sqp", 0), ("qp_scenario_mpc", 1), ("mm_QAFIRO", 1), ("mm_CVXQP1_M", 0), ("mm_LISWET1", 1), ("mm_AUG3DCQP", 0)):
        q = load_qp(name)
        run_case(out, f"sparse_{name}_{scale_cost}", hip, q, True, hip.SPARSE_LDLT, scale_cost)
    # a generated dense problem wide enough for several 256 x 32 tiles per matrix, with one-sided and two-sided rows
    from qp_gen import dense_strongly_convex_qp
    import scipy.sparse as sp
    q = dense_strongly_convex_qp(600, 90, 700, seed=3, double_sided=True, exact_shift=False)
    for k in ("P", "A", "G"):
        q[k] = sp.csc_matrix(q[k])
    run_case(out, "dense_generated_1", hip, q, False, hip.DENSE_LDLT_NO_PIVOT, 1)
    np.savez(sys.argv[1], **out)
    print("ok", len(out))


if __name__ == "__main__":
    main()
