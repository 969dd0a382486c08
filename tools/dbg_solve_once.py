#!/usr/bin/env python3
"""One whole interior-point solve of a frozen fixture through the device sparse solver (for rocprofv3 timelines).  usage: dbg_solve_once.py fixture [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import piqp_amd as hip
from qp_io import load_qp
q = load_qp(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a = (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
for r in range(reps):
    sh = hip.SparseSolver(); sh.settings.kkt_solver = hip.SPARSE_LDLT
    sh.setup(*a)
    t0 = time.perf_counter(); st = sh.solve(); t = time.perf_counter() - t0
    print(f"{sys.argv[1]} status {st} iter {sh.info.iter} solve {t*1e3:.2f} ms  ({t*1e3/max(sh.info.iter,1):.3f} ms/it)")
