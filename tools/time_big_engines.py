#!/usr/bin/env python3
"""Whole solves of the fixtures above 8192 KKT rows through the two engines behind sparse_ldlt (reference-order / multifrontal): status, iterations, wall time of solve().
    python tools/time_big_engines.py [name ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import piqp_amd as hip  # noqa: E402
from qp_io import load_qp  # noqa: E402

names = [a for a in sys.argv[1:]] or ["nl_truss", "mm_STADAT3", "mm_CONT-101", "mm_LISWET1", "mm_POWELL20", "mm_UBH1", "mm_CONT-201", "mm_BOYD1"]
print(f"# {'fixture':14s} {'rows':>7s} | reference-order engine: status iter setup ms solve ms | multifrontal engine: status iter setup ms solve ms")
for name in names:
    q = load_qp(name)
    a = (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
    rows = q["P"].shape[0] + (0 if q["A"] is None else q["A"].shape[0]) + (0 if q["G"] is None else q["G"].shape[0])
    out = []
    for ks in (hip.SPARSE_LDLT_EXACT, hip.SPARSE_LDLT_MULTIFRONTAL):
        try:
            s = hip.SparseSolver(); s.settings.kkt_solver = ks
            if name.startswith("nl"):
                s.settings.infeasibility_threshold = 0.01
            t0 = time.time(); ok = s.setup(*a); t1 = time.time(); st = s.solve(); t2 = time.time()
            out.append(f"{int(st):3d} {s.info.iter:4d} {1e3 * (t1 - t0):9.1f} {1e3 * (t2 - t1):9.1f}")
        except Exception as e:  # noqa: BLE001
            out.append(f"ERROR {str(e)[:60]}")
    print(f"  {name:14s} {rows:7d} | {out[0]} | {out[1]}", flush=True)
