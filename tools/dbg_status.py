#!/usr/bin/env python3
"""Side-by-side verbose tables (solver.hpp:590-602) of the device and the oracle on one fixture, every iteration: where the two
trajectories part and whether rho / delta show a recovery (solver.hpp:688-708) on one side only.
   python tools/dbg_status.py nl_finnis [kkt_solver] [rows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import piqp_amd as hip
from oracle import pyorc as orc
from qp_io import load_qp
name = sys.argv[1]; ks = int(sys.argv[2]) if len(sys.argv) > 2 else 1; rows = int(sys.argv[3]) if len(sys.argv) > 3 else 60
q = load_qp(name)
args = (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
sh = hip.SparseSolver(); sh.settings.kkt_solver = ks; sh.enable_trace()
so = orc.Solver(); so.settings.kkt_solver = ks; so.enable_trace()
if name.startswith("nl"):
    sh.settings.infeasibility_threshold = 0.01; so.settings.infeasibility_threshold = 0.01
sh.setup(*args); so.setup(*args, sparse=True)
print(name, "status dev/orc", sh.solve(), so.solve(), "iters", sh.info.iter, so.info.iter, "factorisations", sh.info.n_factor, so.info.n_factor)
th, to = sh.trace(), so.trace()
cols = [3, 4, 5, 6, 7, 8, 9, 10]
print("it | device: gap prim_res dual_res rho delta mu p_step d_step | oracle: same")
for i in range(min(max(len(th), len(to)), rows)):
    a = " ".join("%9.2e" % v for v in th[i][cols]) if i < len(th) else " " * 79
    b = " ".join("%9.2e" % v for v in to[i][cols]) if i < len(to) else ""
    print("%3d | %s | %s" % (i, a, b))
