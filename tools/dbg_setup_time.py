import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import piqp_amd as hip
from qp_gen import dense_strongly_convex_qp
q = dense_strongly_convex_qp(4096, 0, 4096, seed=1, exact_shift=False)
P = q["P"]; P = np.triu(P) + np.triu(P, 1).T
for rep in range(2):
    s = hip.DenseSolver()
    t0 = time.perf_counter()
    s.setup(P, q["c"], None, None, q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
    t1 = time.perf_counter()
    print("setup wall", t1 - t0, "info.setup_time", s.info.setup_time, flush=True)
    t0 = time.perf_counter(); s.update(P=P * 1.1); t1 = time.perf_counter()
    print("update(P) wall", t1 - t0, flush=True)
    st = s.solve(); print("status", st, "iter", s.info.iter, flush=True)
