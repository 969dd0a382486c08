#!/usr/bin/env python3
"""status / iterations of the device sparse solver vs the oracle on named fixtures: python tools/dbg_sweep.py name [name ...] (PIQP_AMD_ORDERING honoured)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
import piqp_amd as hip  # noqa: E402
from oracle import pyorc as orc  # noqa: E402
from qp_io import load_qp  # noqa: E402

for name in sys.argv[1:]:
    q = load_qp(name)
    a = (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
    thr = 0.01 if name.startswith("nl") else None
    sh = hip.SparseSolver(); sh.settings.kkt_solver = hip.SPARSE_LDLT
    so = orc.Solver(); so.settings.kkt_solver = orc.SPARSE_LDLT
    if thr:
        sh.settings.infeasibility_threshold = thr; so.settings.infeasibility_threshold = thr
    sh.setup(*a); so.setup(*a, sparse=True)
    st_h, st_o = sh.solve(), so.solve()
    print(f"{name:16s} ordering={os.environ.get('PIQP_AMD_ORDERING', 'default'):8s} device status {st_h:3d} iter {sh.info.iter:3d} obj {sh.info.primal_obj:.8e} | oracle status {st_o:3d} iter {so.info.iter:3d} obj {so.info.primal_obj:.8e}")
