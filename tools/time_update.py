#!/usr/bin/env python3
"""update() + solve() of one sparse QP, timed apart (the device equilibration runs inside update):  python tools/time_update.py [c3|<fixture>]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import piqp_amd as hip
from qp_gen import c3_problem
from qp_io import load_qp
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
if name == "c3":
    a = c3_problem()
else:
    q = load_qp(name); a = (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
s = hip.SparseSolver(); s.settings.kkt_solver = hip.SPARSE_LDLT
t0 = time.perf_counter(); s.setup(*a); t1 = time.perf_counter(); st = s.solve(); t2 = time.perf_counter()
print(name, "setup %.1f ms  solve %.1f ms  status %d iters %d" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, st, s.info.iter))
for _ in range(3):
    t0 = time.perf_counter(); s.update(*a); t1 = time.perf_counter(); st = s.solve(); t2 = time.perf_counter()
    print(name, "update %.2f ms  solve %.2f ms  status %d iters %d obj %.12e" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, st, s.info.iter, s.info.primal_obj))
