#!/usr/bin/env python3
"""The reference's own small benchmarks (benchmarks/src/sqp_benchmarks.cpp:16-118) on the frozen copies of their two models: per repetition
`update(all data) + solve()`, with the settings the benchmark sets (robot arm: reg_lower_limit = reg_finetune_lower_limit = 1e-8), for the three KKT
solvers it times -- device next to the oracle on one host core: status, iterations, milliseconds per repetition.
   python tools/sqp_benchmarks.py > gpurun_out/sqp_benchmarks.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = [("qp_chain_mass_sqp", None), ("qp_robot_arm_sqp", 1e-8)]
SOLVERS = [("sparse_ldlt (KKT_FULL; reference-order engine)", 1), ("sparse_ldlt, multifrontal engine", 18), ("sparse_ldlt_cond (KKT_ALL_ELIMINATED)", 4), ("sparse_multistage", 5)]


def run(mod, q, ks, reg, sparse_kw, reps):
    a = (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
    s = mod.SparseSolver() if hasattr(mod, "SparseSolver") else mod.Solver()
    s.settings.kkt_solver = ks
    if reg is not None:
        s.settings.reg_lower_limit = reg
        s.settings.reg_finetune_lower_limit = reg
    s.setup(*a, **sparse_kw)
    ts = []
    st = None
    for _ in range(reps):
        t0 = time.perf_counter()
        s.update(*a)
        st = s.solve()
        ts.append(time.perf_counter() - t0)
    return st, s.info.iter, min(ts) * 1e3, s.info.primal_obj


def main():
    import torch  # noqa: F401
    import piqp_amd as hip
    from oracle import pyorc as orc
    from qp_io import load_qp
    for name, reg in CASES:
        q = load_qp(name)
        for label, ks in SOLVERS:
            so = run(orc, q, 1 if ks == 18 else ks, reg, {"sparse": True}, 5)
            sh = run(hip, q, ks, reg, {}, 5)
            print(f"{name:20s} {label:40s} oracle (1 core): status {so[0]:2d} {so[1]:3d} it {so[2]:8.2f} ms   device: status {sh[0]:2d} {sh[1]:3d} it {sh[2]:8.2f} ms   "
                  f"obj {so[3]:.9e} / {sh[3]:.9e}", flush=True)


        # round 5: the same QP as a batch of ONE through the batched solver: the whole interior-point method in one workgroup, one launch per solve, no host round
        # trip per iteration (kkt_solver = sparse_multistage; update_data = unscale -> assign -> equilibrate on the device, like update())
        try:
            import numpy as np
            import scipy.sparse as sp
            bs = hip.BatchSparseSolver()
            if reg is not None:
                bs.settings.reg_lower_limit = reg; bs.settings.reg_finetune_lower_limit = reg
            rep = lambda v: None if v is None else np.asarray(v, dtype=np.float64)[None, :]
            Pm = sp.csc_matrix(sp.triu(q["P"])); Am = sp.csc_matrix(q["A"]) if q["A"] is not None else None; Gm = sp.csc_matrix(q["G"]) if q["G"] is not None else None
            for M in (Pm, Am, Gm):
                if M is not None:
                    M.sort_indices()
            kw = dict(x_l=rep(q["x_l"]), x_u=rep(q["x_u"]))
            ok = bs.setup(Pm, rep(Pm.data), rep(q["c"]), Am, None if Am is None else rep(Am.data), rep(q["b"]), Gm, None if Gm is None else rep(Gm.data), rep(q["h_l"]), rep(q["h_u"]), **kw)
            assert ok
            bs.solve()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                bs.update_data(P_values=rep(Pm.data), A_values=None if Am is None else rep(Am.data), G_values=None if Gm is None else rep(Gm.data), c=rep(q["c"]), b=rep(q["b"]),
                               h_l=rep(q["h_l"]), h_u=rep(q["h_u"]), x_l=rep(q["x_l"]), x_u=rep(q["x_u"]))
                nsolved = bs.solve()
                ts.append(time.perf_counter() - t0)
            inf = bs.info(0)
            print(f"{name:20s} {'batched solver, batch of 1 (one launch per solve)':40s} device: solved {nsolved}/1 status {inf.status:2d} {inf.iter:3d} it {min(ts) * 1e3:8.2f} ms   obj {inf.primal_obj:.9e}", flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"{name:20s} batched solver, batch of 1: {type(e).__name__}: {str(e)[:160]}", flush=True)


if __name__ == "__main__":
    main()
