#!/usr/bin/env python3
"""Worker of tests/test_multistage_gpu.py::test_two_processes_bitwise_equal: one factor + solve through `sparse_multistage` on structures on
both sides of the chain/tree engine threshold, in a FRESH process; solutions go to an .npz for an `array_equal` comparison.  The engine is
chosen by a symbolic cost model (multistage_kkt.hip), so the arithmetic must not depend on the process or on timing.

  python tests/workers/multistage_repro.py out.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch  # noqa: F401
    import piqp_amd as hip
    from qp_gen import mpc_chain, random_vars
    from qp_io import load_qp
    out = {}
    cases = [("T16", mpc_chain(12, 8, 16, 3)), ("T24", mpc_chain(6, 3, 24, 4)), ("T40", mpc_chain(6, 3, 40, 5)), ("T100", mpc_chain(2, 1, 100, 6))]
    for nm in ("qp_robot_arm_sqp", "qp_robot_arm_sqp_no_global", "qp_chain_mass_sqp"):
        q = load_qp(nm)
        cases.append((nm, (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])))
    rng = np.random.default_rng(0)
    for name, a in cases:
        d = hip.SparseData(*a)
        k = hip.KKTSystem(d, hip.default_settings(kkt_solver=hip.SPARSE_MULTISTAGE))
        state = random_vars(d.n, d.p, d.m, rng, positive=True)
        rhs = random_vars(d.n, d.p, d.m, rng)
        assert k.update_scalings_and_factor(False, 1e-6, 1e-4, state)
        ok, lhs = k.solve(rhs)
        assert ok
        for key, v in lhs.items():
            out[name + "_" + key] = np.asarray(v)
        try:
            k.backend().sparse_stats(); out[name + "_engine"] = np.array([1])
        except Exception:  # noqa: BLE001  (chain engine: no tree statistics)
            out[name + "_engine"] = np.array([0])
    np.savez(sys.argv[1], **out)
    print("ok", len(out))


if __name__ == "__main__":
    main()
