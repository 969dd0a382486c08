import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import piqp_amd as hip
from oracle import pyorc as orc
from qp_gen import c3_problem, mpc_chain
for name, a, ks in (("C3", c3_problem(), 1), ("chain T=5000", mpc_chain(12, 8, 5000, 3), 1)):
    s = hip.SparseSolver(); s.settings.kkt_solver = ks
    t0 = time.perf_counter(); assert s.setup(*a); ts = time.perf_counter() - t0
    t0 = time.perf_counter(); st = s.solve(); tsol = time.perf_counter() - t0
    i = s.info
    print(name, "device solver: status", st, "iter", i.iter, f"setup {ts:.2f}s solve {tsol*1e3:.1f} ms  kkt_factor {i.kkt_factor_time*1e3:.1f} ms kkt_solve {i.kkt_solve_time*1e3:.1f} ms  -> other {(tsol - i.kkt_factor_time - i.kkt_solve_time)*1e3:.1f} ms")
    so = orc.Solver(); so.settings.kkt_solver = ks
    t0 = time.perf_counter(); so.setup(*a, sparse=True); ts = time.perf_counter() - t0
    t0 = time.perf_counter(); st = so.solve(); tsol = time.perf_counter() - t0
    print(name, "oracle: status", st, "iter", so.info.iter, f"setup {ts:.2f}s solve {tsol*1e3:.1f} ms")
