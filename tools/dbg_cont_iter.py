import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import piqp_amd as hip
from qp_io import load_qp
q = load_qp(sys.argv[1] if len(sys.argv) > 1 else "mm_CONT-201")
def args(q): return (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
sh = hip.SparseSolver(); sh.settings.kkt_solver = hip.SPARSE_LDLT; sh.settings.verbose = True
sh.setup(*args(q)); st = sh.solve()
print("status", st, "iter", sh.info.iter, "obj", sh.info.primal_obj, "pres", sh.info.primal_res, "dres", sh.info.dual_res, "gap", sh.info.duality_gap)
