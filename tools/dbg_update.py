import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, scipy.sparse as sp, torch
import piqp_amd as hip
from oracle import pyorc as orc
from qp_gen import dense_strongly_convex_qp
inf = np.inf
q = dense_strongly_convex_qp(30, 8, 14, seed=11)
P, c, A, b, G, h_l, h_u, x_l, x_u = (q[k] for k in ("P", "c", "A", "b", "G", "h_l", "h_u", "x_l", "x_u"))
args = (sp.csc_matrix(np.triu(P)), c, sp.csc_matrix(A), b, sp.csc_matrix(G), h_l, h_u, x_l, x_u)
sh, so = hip.SparseSolver(), orc.Solver()
sh.settings.kkt_solver = so.settings.kkt_solver = 1
assert sh.setup(*args) and so.setup(*args, sparse=True)
def both(tag):
    a, b2 = sh.solve(), so.solve()
    print(tag, "device", a, sh.info.iter, sh.info.primal_obj, "oracle", b2, so.info.iter, so.info.primal_obj, "dx", np.abs(sh.result()["x"] - so.result()["x"]).max())
both("initial")
rng = np.random.default_rng(3)
c2 = c + 0.1 * rng.standard_normal(c.size)
sh.update(c=c2); so.update(c=c2); both("c")
h_u2 = h_u.copy(); h_l2 = h_l.copy(); h_u2[np.isfinite(h_u2) & (h_u2 < 1e29)] += 0.05
x_u2 = x_u.copy(); x_u2[0] = 5.0 if x_u2[0] > 1e29 else inf
x_l2 = x_l.copy(); x_l2[1] = -5.0
sh.update(b=b * 1.0, h_l=h_l2, h_u=h_u2, x_l=x_l2, x_u=x_u2); so.update(b=b * 1.0, h_l=h_l2, h_u=h_u2, x_l=x_l2, x_u=x_u2); both("bounds")
h_l3 = h_l2.copy(); h_u3 = h_u2.copy(); h_l3[2] = -inf; h_u3[2] = inf
sh.update(h_l=h_l3, h_u=h_u3); so.update(h_l=h_l3, h_u=h_u3); both("row zeroed")
sh.update(c=c); so.update(c=c); both("c again")
# fresh solvers on the final data for reference
s2 = hip.SparseSolver(); s2.settings.kkt_solver = 1
s2.setup(sp.csc_matrix(np.triu(P)), c, sp.csc_matrix(A), b, sp.csc_matrix(G), h_l3, h_u3, x_l2, x_u2); print("fresh device", s2.solve(), s2.info.iter, s2.info.primal_obj)
o2 = orc.Solver(); o2.settings.kkt_solver = 1
o2.setup(sp.csc_matrix(np.triu(P)), c, sp.csc_matrix(A), b, sp.csc_matrix(G), h_l3, h_u3, x_l2, x_u2, sparse=True); print("fresh oracle", o2.solve(), o2.info.iter, o2.info.primal_obj)
