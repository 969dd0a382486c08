#!/usr/bin/env python3
"""Stage timings of the sparse_multistage backend on fixtures / synthetic chains (device) next to the oracle (CPU).

  python tools/prof_multistage.py [--reps 50]
"""
import argparse
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


from qp_gen import mpc_chain  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--cond", action="store_true", help="time sparse_ldlt_cond (4) instead of sparse_ldlt (1) next to multistage")
    ap.add_argument("--c5", action="store_true", help="only BASELINE configs[4]: one block-tridiagonal QP, n = 500k (25000 stages of n_x=12, n_u=8)")
    args = ap.parse_args()
    import torch  # noqa: F401
    import piqp_amd as hip
    from oracle import pyorc as orc
    from qp_gen import random_vars
    from qp_io import load_qp

    def fixture(name):
        q = load_qp(name)
        return (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])

    cases = [("c0_scenario_mpc", fixture("qp_c0_scenario_mpc")), ("scenario_mpc", fixture("qp_scenario_mpc")), ("chain_mass_sqp", fixture("qp_chain_mass_sqp")),
             ("robot_arm_sqp", fixture("qp_robot_arm_sqp")), ("mpc nx=2 nu=1 T=40", mpc_chain(2, 1, 40, 1)), ("mpc nx=12 nu=8 T=1000", mpc_chain(12, 8, 1000, 2)),
             ("mpc nx=12 nu=8 T=5000", mpc_chain(12, 8, 5000, 3))]
    if args.c5:
        cases = [("C5 mpc nx=12 nu=8 T=25000", mpc_chain(12, 8, 25000, 5))]
    for name, a in cases:
        d = hip.SparseData(*a); od = orc.Data.sparse(*a)
        n, p, m = od.n, od.p, od.m
        rng = np.random.default_rng(0)
        for ks, kname in (((hip.SPARSE_LDLT, "sparse_ldlt"),) if args.c5 else ((hip.SPARSE_MULTISTAGE, "multistage"), ((4, "ldlt_cond") if args.cond else (hip.SPARSE_LDLT, "sparse_ldlt")))):
            t_s = time.perf_counter()
            k = hip.KKTSystem(d, hip.default_settings(kkt_solver=ks))
            t_setup = time.perf_counter() - t_s
            ko = orc.KKTSystem(od, orc.Settings(kkt_solver=ks if not args.c5 else orc.SPARSE_MULTISTAGE))
            if args.c5:
                k.backend().print_info(); print(f"    device setup {t_setup:.2f} s (oracle column = CPU multistage backend)")
            state = random_vars(n, p, m, rng, positive=True)
            rhs = random_vars(n, p, m, rng)
            be = k.backend()
            for _ in range(3):
                k.update_scalings_and_factor(False, 1e-6, 1e-4, state); k.solve(rhs)
            be.set_profiling(True)
            k.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                k.update_scalings_and_factor(False, 1e-6, 1e-4, state)
            k.synchronize()
            t_fac = (time.perf_counter() - t0) / args.reps
            t0 = time.perf_counter()
            for _ in range(args.reps):
                k.solve(rhs)
            k.synchronize()
            t_sol = (time.perf_counter() - t0) / args.reps
            be.set_profiling(False)
            prof = [be.get_profile(s) for s in range(3)]
            res, nrm = k.condensed_residual()
            t0 = time.perf_counter()
            reps_o = max(3, args.reps // 5)
            for _ in range(reps_o):
                ko.update_scalings_and_factor(False, 1e-6, 1e-4, state)
            o_fac = (time.perf_counter() - t0) / reps_o
            t0 = time.perf_counter()
            for _ in range(reps_o):
                ko.solve(rhs)
            o_sol = (time.perf_counter() - t0) / reps_o
            bi = be.block_info() if ks == hip.SPARSE_MULTISTAGE else None
            extra = f" stages={len(bi) - 1} maxw={bi[:-1, 1].max()} arrow={bi[-1, 1]}" if bi is not None else ""
            print(f"{name:24s} {kname:11s} n={n} p={p} m={m}{extra}\n"
                  f"    device: factor call {t_fac * 1e6:8.1f} us (assemble {prof[0][0] / max(prof[0][1], 1) * 1e3:7.1f} us, chain {prof[1][0] / max(prof[1][1], 1) * 1e3:7.1f} us)"
                  f"  solve call {t_sol * 1e6:8.1f} us (backend {prof[2][0] / max(prof[2][1], 1) * 1e3:7.1f} us x{prof[2][1] / args.reps:.1f})  rel.res {res / nrm:.1e}\n"
                  f"    oracle: factor {o_fac * 1e6:8.1f} us  solve {o_sol * 1e6:8.1f} us", flush=True)


if __name__ == "__main__":
    main()
