import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import piqp_amd as hip
from qp_gen import dense_strongly_convex_qp
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 16
q = dense_strongly_convex_qp(dim, dim // 2, dim // 2, seed=100 + dim)
Pf = np.triu(q["P"]) + np.triu(q["P"], 1).T
a = (Pf, q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
s = hip.DenseSolver(); s.settings.kkt_solver = 0
s.setup(*a); s.solve()
t0 = time.perf_counter(); st = s.solve(); t = time.perf_counter() - t0
print("MARK dim", dim, "status", st, "iters", s.info.iter, "ms", t * 1e3)
