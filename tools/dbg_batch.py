import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import piqp_amd as hip
from oracle import pyorc as orc
from qp_gen import mpc_batch, mpc_instance
B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
mb = mpc_batch(B, seed=1000)
bs = hip.BatchSparseSolver()
assert bs.setup(mb["P_pattern"], mb["P_values"], mb["c"], mb["A_pattern"], mb["A_values"], mb["b"], x_l=mb["x_l"], x_u=mb["x_u"])
print("solved", bs.solve(), "kernel ms", bs.last_kernel_ms())
x = bs.result("x")
import time
t0 = time.perf_counter(); bs.solve(); print("second solve wall ms", (time.perf_counter() - t0) * 1e3, "kernel ms", bs.last_kernel_ms())
its = bs.iterations(); print("iters: min", its.min(), "max", its.max(), "mean", its.mean(), "statuses", np.unique(bs.statuses(), return_counts=True))
NCHK = int(sys.argv[2]) if len(sys.argv) > 2 else B
mism = 0
for i in range(NCHK):
    s = orc.Solver(); s.settings.kkt_solver = orc.SPARSE_MULTISTAGE
    s.setup(*mpc_instance(mb, i), sparse=True); st = s.solve()
    inf = bs.info(i)
    flag = "" if (inf.status == st and inf.iter == s.info.iter) else "  <<<<"
    mism += bool(flag)
    if flag or i < 4: print(i, "dev", inf.status, inf.iter, f"{inf.primal_obj:.9e} pres {inf.primal_res:.2e} dres {inf.dual_res:.2e} mu {inf.mu:.2e} rho {inf.rho:.1e} delta {inf.delta:.1e} nfac {inf.n_factor} nsol {inf.n_solve} nbs {inf.n_backend_solve}",
          "| orc", st, s.info.iter, f"{s.info.primal_obj:.9e}", "dx", f"{np.abs(x[i]-s.result()['x']).max():.1e}", flag)
    if flag and mism < 3:
        h = hip.SparseSolver(); h.settings.kkt_solver = hip.SPARSE_MULTISTAGE; h.enable_trace()
        h.setup(*mpc_instance(mb, i)); sth = h.solve()
        print("   host-driven device solver:", sth, h.info.iter, f"{h.info.primal_obj:.9e}")
        s2 = orc.Solver(); s2.settings.kkt_solver = orc.SPARSE_MULTISTAGE; s2.enable_trace(); s2.setup(*mpc_instance(mb, i), sparse=True); s2.solve()
        s3 = orc.Solver(); s3.settings.kkt_solver = orc.SPARSE_LDLT; s3.enable_trace(); s3.setup(*mpc_instance(mb, i), sparse=True); s3.solve()
        th, to, tl = h.trace(), s2.trace(), s3.trace()
        print("   sparse_ldlt oracle iters", s3.info.iter)
        for r in range(max(len(th), len(to))):
            a = th[r] if r < len(th) else None; b = to[r] if r < len(to) else None; c = tl[r] if r < len(tl) else None
            f = lambda t: "-" if t is None else f"pres {t[4]:.6e} dres {t[5]:.6e} mu {t[8]:.3e}"
            print("   ", r, "| hip", f(a), "| orc", f(b), "| orc-ldlt", f(c))
print("mismatches", mism, "of", NCHK)
