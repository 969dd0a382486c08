#!/usr/bin/env python3
"""Accuracy of the sparse device factorisation on the recorded interior-point states of a fixture: relative residual of the condensed KKT system
(extended precision, the oracle's Ruiz-scaled matrices) for the device backend and for the oracle's up-looking LDLt on the same states.
usage: python tools/dbg_sparse_accuracy.py fixture [infeasibility_threshold]"""
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
import piqp_amd as hip  # noqa: E402
from oracle import pyorc as orc  # noqa: E402
from qp_io import load_qp  # noqa: E402

name = sys.argv[1]
q = load_qp(name)
a = (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
so = orc.Solver(); so.settings.kkt_solver = orc.SPARSE_LDLT
if len(sys.argv) > 2:
    so.settings.infeasibility_threshold = float(sys.argv[2])
assert so.setup(*a, sparse=True)
states = so.record_states()
so.solve()
od = so.data()
Pu, AT, GT = od.csc("P_utri"), od.csc("AT"), od.csc("GT")
n, p, m = od.n, od.p, od.m


class Scaled(hip.SparseData):
    def __init__(self):
        self.n, self.p, self.m = n, p, m
        self.P_utri, self.AT, self.GT = Pu, AT, GT
        self.h_l_idx, self.h_u_idx, self.x_l_idx, self.x_u_idx = od.idx("h_l"), od.idx("h_u"), od.idx("x_l"), od.idx("x_u")
        self.n_h_l, self.n_h_u, self.n_x_l, self.n_x_u = od.counts()
        self.x_b_scaling = od.vec("x_b_scaling").copy()


kh = hip.KKTSystem(Scaled(), hip.default_settings(kkt_solver=hip.SPARSE_LDLT))
ko = orc.KKTSystem(od, orc.Settings(kkt_solver=orc.SPARSE_LDLT))
fs = [s for s in states if s["kind"] == 0]
ss = [s for s in states if s["kind"] == 1]
Pf = (Pu + sp.triu(Pu, 1).T).tocsr(); A = AT.T.tocsr(); G = GT.T.tocsr()
L = np.longdouble
for it in range(len(fs)):
    st = fs[it]; rhs = ss[min(2 * it + 1, len(ss) - 1)]["vars"]
    okh = kh.update_scalings_and_factor(False, st["rho"], st["delta"], st["vars"])
    oko = ko.update_scalings_and_factor(False, st["rho"], st["delta"], st["vars"])
    _, lh = kh.solve(rhs); _, lo = ko.solve(rhs)
    xr, zr, rx, rz, ry = ko.x_reg(), ko.z_reg(), ko.rhs_x_bar(), ko.rhs_z_bar(), rhs["y"]

    def resid(l):
        z = l["z_u"] - l["z_l"]
        r1 = rx.astype(L) - (Pf @ l["x"]).astype(L) - xr.astype(L) * l["x"] - (AT @ l["y"]).astype(L) - (GT @ z).astype(L)
        r2 = ry.astype(L) - (A @ l["x"]).astype(L) + L(st["delta"]) * l["y"] if p else np.zeros(1, L)
        r3 = rz.astype(L) - (G @ l["x"]).astype(L) + zr.astype(L) * z if m else np.zeros(1, L)
        nrm = max(np.abs(rx).max(), np.abs(ry).max() if p else 0.0, np.abs(rz).max() if m else 0.0)
        return float(max(np.abs(r1).max(), np.abs(r2).max(), np.abs(r3).max()) / nrm)
    rh, ro = resid(lh), resid(lo)
    import ctypes as C
    mp = C.c_double(); hip._lib.load().pq_kkt_min_abs_pivot(kh.backend().h, C.byref(mp))
    print(f"state {it:2d} rho={st['rho']:.1e} delta={st['delta']:.1e} ok {int(okh)}/{int(oko)} rel.residual device {rh:.2e} oracle {ro:.2e} ratio {rh / max(ro, 1e-300):.2f}  device min|pivot| {mp.value:.2e}")
