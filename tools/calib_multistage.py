#!/usr/bin/env python3
"""Calibration data for the deterministic engine choice of `sparse_multistage` (chain recurrence vs nested-dissection tree):
times one factorisation and one backend solve with each engine forced, next to the structural figures the cost model may use.

  python tools/calib_multistage.py > gpurun_out/calib_multistage.jsonl
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch  # noqa: F401
    import piqp_amd as hip
    from qp_gen import mpc_chain, random_vars
    from qp_io import load_qp

    def fixture(name):
        q = load_qp(name)
        return (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])

    cases = [(nm, fixture(nm)) for nm in ("qp_c0_scenario_mpc", "qp_scenario_mpc", "qp_chain_mass_sqp", "qp_robot_arm_sqp", "qp_robot_arm_sqp_no_global")]
    for nx, nu, Ts in ((2, 1, (16, 24, 40, 64, 100, 400)), (6, 3, (16, 24, 40, 64, 100, 400)), (12, 8, (16, 24, 40, 64, 100, 1000)), (30, 10, (16, 32, 64, 200))):
        for T in Ts:
            cases.append((f"mpc nx={nx} nu={nu} T={T}", mpc_chain(nx, nu, T, 7)))
    reps = 20
    for name, a in cases:
        d = hip.SparseData(*a)
        n, p, m = d.n, d.p, d.m
        rng = np.random.default_rng(0)
        state = random_vars(n, p, m, rng, positive=True)
        rhs = random_vars(n, p, m, rng)
        row = dict(name=name, n=n, p=p, m=m)
        for eng in ("chain", "tree"):
            os.environ["PIQP_AMD_MULTISTAGE"] = eng
            k = hip.KKTSystem(d, hip.default_settings(kkt_solver=hip.SPARSE_MULTISTAGE))
            be = k.backend()
            for _ in range(3):
                k.update_scalings_and_factor(False, 1e-6, 1e-4, state); k.solve(rhs)
            k.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                k.update_scalings_and_factor(False, 1e-6, 1e-4, state)
            k.synchronize()
            row[eng + "_factor_us"] = (time.perf_counter() - t0) / reps * 1e6
            t0 = time.perf_counter()
            for _ in range(reps):
                k.solve(rhs)
            k.synchronize()
            row[eng + "_solve_us"] = (time.perf_counter() - t0) / reps * 1e6
            if eng == "chain":
                bi = be.block_info()
                w = bi[:-1, 1].astype(float); o = bi[:-1, 2].astype(float); arrow = float(bi[-1, 1]); h = w + o + arrow
                row.update(stages=len(w), arrow=arrow, sum_w=w.sum(), sum_hw=(h * w).sum(), sum_h2w=(h * h * w).sum(), max_h=h.max())
            else:
                try:
                    row["tree_stats"] = be.sparse_stats()
                except Exception as e:  # noqa: BLE001
                    row["tree_stats"] = str(e)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
