"""whole solves of one small QP on the default engine (for a kernel profile): python tools/one_small_solve.py [fixture] [repetitions]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import piqp_amd as hip
from qp_io import load_qp
nm = sys.argv[1] if len(sys.argv) > 1 else "qp_chain_mass_sqp"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
q = load_qp(nm)
a = (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
s = hip.SparseSolver()
s.settings.kkt_solver = hip.SPARSE_LDLT
s.setup(*a)
for _ in range(reps):
    s.update(*a)
    st = s.solve()
print(nm, "status", st, "iterations", s.info.iter)
