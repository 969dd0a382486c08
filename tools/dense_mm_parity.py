#!/usr/bin/env python3
"""The reference's DENSE Maros-Meszaros sweep (/root/reference/tests/src/dense/maros_meszaros_tests.cpp:21-51: every problem with n <= 1000 and p + m <= 1000
through DenseSolver, status SOLVED) through the dense device solver and through the oracle's dense solver: status, iteration count, objective, time.
One line per frozen fixture and backend (dense_cholesky = Eigen::LLT, dense_ldlt_no_pivot); `=` where the device's count is the oracle's.
    python tools/dense_mm_parity.py > profiles/r06_dense_mm_parity.txt          (PIQP_AMD_HOST_IPM=1 for the host-side loop)"""
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import piqp_amd as hip  # noqa: E402
from oracle import pyorc as orc  # noqa: E402
from qp_io import GOLDEN, dense_args, load_qp  # noqa: E402


def dense_sweep_names():
    out = []
    for f in sorted(glob.glob(os.path.join(GOLDEN, "mm_*.npz"))):
        name = os.path.basename(f)[:-4]
        q = load_qp(name)
        n = q["P"].shape[0]; p = 0 if q["A"] is None else q["A"].shape[0]; m = 0 if q["G"] is None else q["G"].shape[0]
        if n <= 1000 and p + m <= 1000:
            out.append((name, n, p, m))
    return out


def main():
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    loop = "host loop" if os.environ.get("PIQP_AMD_HOST_IPM") == "1" else "device loop"
    print(f"# dense Maros-Meszaros sweep, {loop}: status / iterations / objective, device next to oracle; kkt_solver 0 = dense_cholesky, 16 = dense_ldlt_no_pivot")
    print(f"# {'problem':12s} {'n':>5s} {'p':>5s} {'m':>5s} | ks | {'device':>22s} {'ms':>8s} | {'oracle':>22s} {'ms':>8s} | count")
    nsame = ntot = 0
    for name, n, p, m in dense_sweep_names():
        if only and name not in only:
            continue
        args = dense_args(load_qp(name))
        for ks in (0, 16):
            sh, so = hip.DenseSolver(), orc.Solver()
            sh.settings.kkt_solver = so.settings.kkt_solver = ks
            assert sh.setup(*args) and so.setup(*args)
            t0 = time.time(); st_h = sh.solve(); t1 = time.time(); st_o = so.solve(); t2 = time.time()
            same = int(st_h) == int(st_o) and sh.info.iter == so.info.iter
            nsame += same; ntot += 1
            print(f"  {name:12s} {n:5d} {p:5d} {m:5d} | {ks:2d} | {int(st_h):3d} {sh.info.iter:4d} {sh.info.primal_obj:14.7e} {1e3 * (t1 - t0):8.1f} | "
                  f"{int(st_o):3d} {so.info.iter:4d} {so.info.primal_obj:14.7e} {1e3 * (t2 - t1):8.1f} | {'=' if same else 'DIFFERENT'}", flush=True)
    print(f"# {nsame} of {ntot} solves with the oracle's status and iteration count")


if __name__ == "__main__":
    main()
