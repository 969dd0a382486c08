"""Replay of the recorded interior-point states of an oracle solve through the dense device backend and the oracle's own backend
(same Ruiz-scaled matrices), with the residual of the condensed KKT system evaluated in extended precision.  Used by
tests/test_dense_gpu.py and tools/dbg_dense_accuracy.py."""
import numpy as np

import piqp_amd as hip
from oracle import pyorc as orc
from qp_io import dense_args, load_qp


def replay(name, ks):
    q = load_qp(name)
    so = orc.Solver(); so.settings.kkt_solver = ks
    assert so.setup(*dense_args(q))
    states = so.record_states()
    so.solve()
    od = so.data()
    n, p, m = od.n, od.p, od.m
    Pu, AT, GT = od.mat("P_utri").copy(), od.mat("AT").copy(), od.mat("GT").copy()

    class Scaled(hip.Data):  # identical (Ruiz-scaled) matrices for both backends
        def __init__(self):
            self.n, self.p, self.m = n, p, m
            self.P_utri, self.AT, self.GT = np.asfortranarray(Pu), np.asfortranarray(AT), np.asfortranarray(GT)
            self.h_l_idx, self.h_u_idx, self.x_l_idx, self.x_u_idx = od.idx("h_l"), od.idx("h_u"), od.idx("x_l"), od.idx("x_u")
            self.n_h_l, self.n_h_u, self.n_x_l, self.n_x_u = od.counts()
            self.x_b_scaling = od.vec("x_b_scaling").copy()
    kh = hip.KKTSystem(Scaled(), hip.default_settings(kkt_solver=ks))
    ko = orc.KKTSystem(od, orc.Settings(kkt_solver=ks))
    fs = [s for s in states if s["kind"] == 0]
    ss = [s for s in states if s["kind"] == 1]
    L = np.longdouble
    Pf = (np.triu(Pu) + np.triu(Pu, 1).T).astype(L); A = AT.T.astype(L); G = GT.T.astype(L)
    out = []
    for it in range(len(fs)):
        st = fs[it]; rhs = ss[min(2 * it + 1, len(ss) - 1)]["vars"]
        okh = kh.update_scalings_and_factor(False, st["rho"], st["delta"], st["vars"])
        oko = ko.update_scalings_and_factor(False, st["rho"], st["delta"], st["vars"])
        # (a failed factorisation is reported, never solved with: its NaN / Inf factor would only put RuntimeWarnings into the test log)
        lh = lo = None
        if okh:
            _, lh = kh.solve(rhs)
        if oko:
            _, lo = ko.solve(rhs)
        if not oko:
            out.append((it, st["rho"], st["delta"], okh, oko, float("nan"), float("nan")))
            continue
        xr, zr, rx, rz, ry = ko.x_reg(), ko.z_reg(), ko.rhs_x_bar(), ko.rhs_z_bar(), rhs["y"]

        def resid(l):
            z = (l["z_u"] - l["z_l"]).astype(L); x = l["x"].astype(L); y = l["y"].astype(L)
            r1 = rx.astype(L) - Pf @ x - xr.astype(L) * x - A.T @ y - G.T @ z
            r2 = ry.astype(L) - A @ x + L(st["delta"]) * y
            r3 = rz.astype(L) - G @ x + zr.astype(L) * z
            nrm = max(np.abs(rx).max(), np.abs(ry).max() if p else 0.0, np.abs(rz).max() if m else 0.0)
            return float(max(np.abs(r1).max(), np.abs(r2).max() if p else 0.0, np.abs(r3).max() if m else 0.0) / nrm)
        out.append((it, st["rho"], st["delta"], okh, oko, resid(lh) if okh else float("nan"), resid(lo)))
    return out
