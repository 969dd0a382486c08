#!/usr/bin/env python3
"""Iteration counts of the dense solver on the fixtures of tests/test_solver_gpu.py, device loop and host loop, next to the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import piqp_amd as hip
from oracle import pyorc as orc
from qp_io import load_qp
names = sys.argv[1:] or ["qp_small_dense", "qp_scenario_mpc_small", "qp_chain_mass_sqp", "qp_robot_arm_sqp", "mm_HS21", "mm_HS118", "mm_DUAL1", "mm_CVXQP1_S", "mm_QAFIRO"]
def dense_args(q):
    d = lambda M: None if M is None else np.asarray(M.todense()) if hasattr(M, "todense") else np.asarray(M)
    return (d(q["P"]), q["c"], d(q["A"]), q["b"], d(q["G"]), q["h_l"], q["h_u"], q["x_l"], q["x_u"])
for nm in names:
    q = load_qp(nm); a = dense_args(q)
    sh = hip.DenseSolver(); sh.setup(*a); st = sh.solve()
    so = orc.Solver(); so.setup(*a); sto = so.solve()
    print(f"{nm:26s} device {st} it {sh.info.iter:3d}   oracle {sto} it {so.info.iter:3d}   n={a[0].shape[0]}")
