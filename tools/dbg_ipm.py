#!/usr/bin/env python3
"""Per-iteration trace of the device-resident IPM next to the host IPM (PIQP_AMD_HOST_IPM=1) and the oracle on one fixture."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
import piqp_amd as hip  # noqa: E402
from oracle import pyorc as orc  # noqa: E402
from qp_io import dense_args, load_qp  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "qp_robot_arm_sqp"
ks = int(sys.argv[2]) if len(sys.argv) > 2 else 0
q = load_qp(name)
args = dense_args(q) if ks == 0 else tuple(q[k] for k in ("P", "c", "A", "b", "G", "h_l", "h_u", "x_l", "x_u"))
tr = {}
for kind in ("device", "host"):
    os.environ["PIQP_AMD_HOST_IPM"] = "1" if kind == "host" else "0"
    s = hip.DenseSolver() if ks == 0 else hip.SparseSolver(); s.settings.kkt_solver = ks; s.enable_trace(); assert s.setup(*args); st = s.solve()
    tr[kind] = s.trace(); print(kind, "status", st, "iter", s.info.iter)
so = orc.Solver(); so.settings.kkt_solver = ks; so.enable_trace(); so.setup(*args, sparse=ks != 0); st = so.solve(); tr["oracle"] = so.trace(); print("oracle status", st, "iter", so.info.iter)
np.set_printoptions(linewidth=250, precision=6)
cols = ["it", "pobj", "dobj", "gap", "pres", "dres", "rho", "delta", "mu", "ps", "ds"]
for i in range(max(len(t) for t in tr.values())):
    for k, t in tr.items():
        if i < len(t):
            print(f"{k:7s}", " ".join(f"{c}={v:.9e}" for c, v in zip(cols[1:], t[i][1:])))
    print()
