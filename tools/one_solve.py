import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import piqp_amd as hip
from qp_io import load_qp
name = sys.argv[1]; ks = int(sys.argv[2])
q = load_qp(name)
a = (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
s = hip.SparseSolver(); s.settings.kkt_solver = ks
if "robot" in name:
    s.settings.reg_lower_limit = 1e-8; s.settings.reg_finetune_lower_limit = 1e-8
s.setup(*a); s.solve()
s.update(*a)
t0 = time.perf_counter(); st = s.solve(); t = time.perf_counter() - t0
print("MARK status", st, "iters", s.info.iter, "ms", t * 1e3)
