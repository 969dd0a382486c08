"""step-by-step check of the reference-order engine on one fixture (debugging aid): factor, compare, solve, compare; prints and flushes after every step"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import piqp_amd as hip  # noqa: E402
from oracle import pyorc as orc  # noqa: E402
from qp_io import load_qp  # noqa: E402


def say(*a):
    print(*a, flush=True)


nm = sys.argv[1]
q = load_qp(nm)
a = (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
d = hip.SparseData(*a); od = orc.Data.sparse(*a)
n, p, m = d.n, d.p, d.m
say(nm, "n p m", n, p, m)
k = hip.SparseKKT(d, kkt_solver=hip.SPARSE_LDLT_EXACT)
say("created", k.sparse_stats())
ko = orc.KKT(od, kind="sparse", mode=0)
rng = np.random.default_rng(1)
x_reg, z_reg = np.full(n, 1e-6), np.abs(rng.standard_normal(m)) + 0.1
ok = k.update_scalings_and_factor(1e-4, x_reg, z_reg)
say("device factor ok", ok)
oko = ko.update_scalings_and_factor(1e-4, x_reg, z_reg)
fh, fo = k.exact_factor(), ko.sparse_factor()
say("PKPt equal", np.array_equal(fh["PKPt_val"], fo["PKPt_val"]), "L_ind equal", np.array_equal(fh["L_ind"], fo["L_ind"]))
bad = np.nonzero(fh["L_vals"] != fo["L_vals"])[0]
say("L_vals mismatches", bad.size, "of", fo["L_vals"].size, "first", bad[:5], "D mismatches", int((fh["D"] != fo["D"]).sum()), "Dinv", int((fh["D_inv"] != fo["D_inv"]).sum()))
if bad.size:
    rows = fo["L_ind"][bad]
    say("  first bad rows", rows[:10], "max abs diff", np.abs(fh["L_vals"][bad] - fo["L_vals"][bad]).max())
rx, ry, rz = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(m)
lh = k.solve(rx, ry, rz)
say("device solve done")
lo = ko.solve(rx, ry, rz)
for u, v, t in zip(lh, lo, "xyz"):
    u, v = np.asarray(u), np.asarray(v)
    say("  solve", t, "equal", np.array_equal(u, v), "max diff", float(np.abs(u - v).max()) if u.size else 0.0)
for _ in range(3):
    k.update_scalings_and_factor(1e-4, x_reg, z_reg); k.solve(rx, ry, rz)
say("repeat ok; D equal again", np.array_equal(k.exact_factor()["D"], fo["D"]))
