#!/usr/bin/env python3
"""One small QP, whole solve: the host-driven SparseSolver (one launch per operation), the batched whole-IPM kernel with a batch of ONE (one launch per
solve) and the oracle on one host core, on the frozen fixtures of the reference's own small benchmarks (benchmarks/src/sqp_benchmarks.cpp:16-118).
   python tools/small_qp_one_launch.py [fixture ...]"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def best(f, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t0)
    return min(ts), r


def main():
    import torch  # noqa: F401
    import piqp_amd as hip
    from oracle import pyorc as orc
    from qp_io import load_qp
    names = sys.argv[1:] or ["qp_chain_mass_sqp", "qp_robot_arm_sqp", "qp_scenario_mpc", "qp_scenario_mpc_small", "qp_small_dense"]
    for name in names:
        q = load_qp(name)
        a = (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
        n = q["P"].shape[0]; p = 0 if q["A"] is None else q["A"].shape[0]; m = 0 if q["G"] is None else q["G"].shape[0]
        line = f"{name:28s} n={n:5d} p={p:5d} m={m:5d}"
        for label, ks_o, ks_h in (("multistage", orc.SPARSE_MULTISTAGE, hip.SPARSE_MULTISTAGE), ("sparse_ldlt", orc.SPARSE_LDLT, hip.SPARSE_LDLT)):
            so = orc.Solver(); so.settings.kkt_solver = ks_o
            so.setup(*a, sparse=True)
            t_o, st_o = best(so.solve)
            sh = hip.SparseSolver(); sh.settings.kkt_solver = ks_h
            sh.setup(*a)
            t_h, st_h = best(sh.solve)
            line += f" | {label}: oracle {t_o*1e3:7.2f} ms/{so.info.iter:3d} it, host-driven device {t_h*1e3:7.2f} ms/{sh.info.iter:3d} it (st {st_h})"
        try:
            bs = hip.BatchSparseSolver()
            P = sp.csc_matrix(sp.triu(q["P"])); P.sort_indices()
            A = sp.csc_matrix(q["A"]) if p else None
            if p: A.sort_indices()
            G = sp.csc_matrix(q["G"]) if m else None
            if m: G.sort_indices()
            one = lambda v: None if v is None else np.asarray(v, dtype=np.float64)[None, :]
            ok = bs.setup(P, P.data[None, :], one(q["c"]), A if p else None, A.data[None, :] if p else None, one(q["b"]) if p else None,
                          G if m else None, G.data[None, :] if m else None, one(q["h_l"]) if m else None, one(q["h_u"]) if m else None, one(q["x_l"]), one(q["x_u"]))
            t_b, solved = best(bs.solve)
            info = bs.info(0)
            line += f" | batch of one: {t_b*1e3:7.2f} ms/{info.iter:3d} it (st {info.status}, setup ok {ok})"
        except Exception as e:  # noqa: BLE001
            line += f" | batch of one: {type(e).__name__}: {str(e)[:160]}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
