"""Factor / solve time of the reference-order sparse engine (sparse_exact.hip) next to the multifrontal engine and the CPU oracle (one core), per fixture.
usage: python tools/time_exact.py [fixture ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import piqp_amd  # noqa: E402
from oracle import pyorc as orc  # noqa: E402
from qp_io import load_qp  # noqa: E402

names = sys.argv[1:] or ["mm_QAFIRO", "mm_QSHARE1B", "mm_QBEACONF", "nl_finnis", "nl_fffff800", "mm_QGROW22", "nl_perold", "mm_QPILOTNO", "mm_QSHIP08L", "mm_STADAT1", "qp_robot_arm_sqp",
                         "qp_chain_mass_sqp", "mm_AUG3DCQP", "mm_CONT-050", "nl_truss"]
print(f"{'fixture':18s} {'N':>6s} {'nnzL':>8s} {'height':>6s} {'crit':>8s} {'tasks':>6s} {'waves':>5s} | {'exact fac':>9s} {'sol':>7s} | {'mf fac':>7s} {'sol':>7s} | {'cpu fac':>7s} {'sol':>7s}  (ms)  ns/crit-step")
for nm in names:
    q = load_qp(nm)
    a = (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
    d = piqp_amd.SparseData(*a); od = orc.Data.sparse(*a)
    n, p, m = d.n, d.p, d.m
    rng = np.random.default_rng(1)
    x_reg, z_reg = np.full(n, 1e-6), np.abs(rng.standard_normal(m)) + 0.1
    rx, ry, rz = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(m)
    res = []
    st = None
    for ks in (piqp_amd.SPARSE_LDLT_EXACT, piqp_amd.SPARSE_LDLT_MULTIFRONTAL):
        k = piqp_amd.SparseKKT(d, kkt_solver=ks)
        if ks == piqp_amd.SPARSE_LDLT_EXACT:
            st = k.sparse_stats()
        for _ in range(3):
            k.update_scalings_and_factor(1e-4, x_reg, z_reg); k.solve(rx, ry, rz)
        k.set_profiling(True)
        reps = 10
        for _ in range(reps):
            k.update_scalings_and_factor(1e-4, x_reg, z_reg); k.solve(rx, ry, rz)
        k.synchronize()
        k.set_profiling(False)
        pr = [k.get_profile(s) for s in range(3)]
        res += [(pr[0][0] + pr[1][0]) / reps, pr[2][0] / reps]
    ko = orc.KKT(od, kind="sparse", mode=0)
    ko.update_scalings_and_factor(1e-4, x_reg, z_reg)
    t0 = time.perf_counter()
    for _ in range(5):
        ko.update_scalings_and_factor(1e-4, x_reg, z_reg)
    t1 = time.perf_counter()
    for _ in range(5):
        ko.solve(rx, ry, rz)
    t2 = time.perf_counter()
    res += [(t1 - t0) / 5 * 1e3, (t2 - t1) / 5 * 1e3]
    crit = st["max_front"]
    print(f"{nm:18s} {st['N']:6d} {st['nnz_L']:8d} {st['tree_levels']:6d} {crit:8d} {st['supernodes']:6d} {st['subtrees']:5d} | {res[0]:9.3f} {res[1]:7.3f} | {res[2]:7.3f} {res[3]:7.3f} | {res[4]:7.3f} {res[5]:7.3f}"
          f"        {res[0] * 1e6 / max(crit, 1):6.0f}", flush=True)
