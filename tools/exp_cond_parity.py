#!/usr/bin/env python3
"""Whole solves of the frozen Maros-Meszaros / netlib fixtures through a condensed sparse backend (default sparse_ldlt_cond = KKT_ALL_ELIMINATED), device next to
the oracle: status and iteration count per fixture, mismatches listed.   python tools/exp_cond_parity.py [kkt_solver] > gpurun_out/r04_cond_parity.txt"""
import glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import piqp_amd as hip
from oracle import pyorc as orc
from qp_io import load_qp
ks = int(sys.argv[1]) if len(sys.argv) > 1 else 4
names = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(ROOT, "tests", "golden", "mm_*.npz")) + glob.glob(os.path.join(ROOT, "tests", "golden", "nl*_*.npz")))
bad = []
t0 = time.time()
for name in names:
    q = load_qp(name)
    a = (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
    try:
        sh = hip.SparseSolver(); sh.settings.kkt_solver = ks
        so = orc.Solver(); so.settings.kkt_solver = ks
        if name.startswith("nl"):
            sh.settings.infeasibility_threshold = 0.01; so.settings.infeasibility_threshold = 0.01
        ok_h = sh.setup(*a); ok_o = so.setup(*a, sparse=True)
        st_h, st_o = sh.solve(), so.solve()
        same = st_h == st_o and abs(sh.info.iter - so.info.iter) <= (0 if so.info.iter < 30 else 1)
        print(f"{name:20s} oracle {st_o:3d}/{so.info.iter:3d}  device {st_h:3d}/{sh.info.iter:3d}  {'' if same else '<-- differs'}", flush=True)
        if not same:
            bad.append(name)
    except Exception as e:  # noqa: BLE001
        print(f"{name:20s} refused: {str(e)[:100]}", flush=True)
print(f"{len(names)} fixtures, {len(bad)} differ: {bad}  ({time.time() - t0:.0f} s)")
