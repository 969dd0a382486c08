#!/usr/bin/env python3
"""setup / solve / update wall times of the full dense solver at the BASELINE configs[1] shape"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402
import piqp_amd as hip  # noqa: E402
from qp_gen import dense_strongly_convex_qp  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
q = dense_strongly_convex_qp(n, 0, m, seed=43, double_sided=True, exact_shift=False)
s = hip.DenseSolver()
t0 = time.perf_counter(); assert s.setup(q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"]); t_setup = time.perf_counter() - t0
t0 = time.perf_counter(); st = s.solve(); t_solve = time.perf_counter() - t0
i = s.info
print(f"n={n} m={m}: setup {t_setup:.3f} s (info.setup_time {i.setup_time:.3f}), solve {t_solve * 1e3:.1f} ms status {st} iter {i.iter} kkt_factor {i.kkt_factor_time * 1e3:.1f} ms kkt_solve {i.kkt_solve_time * 1e3:.1f} ms")
t0 = time.perf_counter(); s.update(c=q["c"] * 1.01); t_upd_c = time.perf_counter() - t0
t0 = time.perf_counter(); s.update(P=q["P"], G=q["G"]); t_upd_m = time.perf_counter() - t0; upd_info = s.info.update_time
t0 = time.perf_counter(); st = s.solve(); t_solve2 = time.perf_counter() - t0
print(f"update(c) {t_upd_c * 1e3:.1f} ms, update(P, G) {t_upd_m * 1e3:.1f} ms (info.update_time {upd_info * 1e3:.1f} ms), second solve {t_solve2 * 1e3:.1f} ms status {st} iter {s.info.iter}")
