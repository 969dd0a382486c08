"""Where does the device's interior-point trajectory leave the oracle's?  Runs one frozen fixture through the device solver (sparse_ldlt: the reference-order
engine up to 8192 KKT rows) and through the CPU oracle (built without FMA contraction) with the per-iteration trace on, and reports the first iteration and
column of the verbose table (solver.hpp:590-602) whose doubles differ bitwise, the statuses and the counts.
usage: python tools/trace_diff.py fixture [fixture ...]   (--all: every frozen mm_* / nl_* / nli_* / qp_* fixture, one line each)"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import piqp_amd as hip  # noqa: E402
from oracle import pyorc as orc  # noqa: E402
from qp_io import GOLDEN, load_qp  # noqa: E402

COLS = ("iter", "prim_obj", "dual_obj", "duality_gap", "prim_res", "dual_res", "rho", "delta", "mu", "p_step", "d_step")


def run(name, verbose=True, ks=1):
    q = load_qp(name)
    if "P" not in q:
        return None
    a = (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
    sh = hip.SparseSolver(); sh.settings.kkt_solver = ks; sh.enable_trace(1024)
    so = orc.Solver(); so.settings.kkt_solver = ks if ks in (1, 2, 3, 4) else orc.SPARSE_LDLT; so.enable_trace(1024)
    if name.startswith("nl"):
        sh.settings.infeasibility_threshold = so.settings.infeasibility_threshold = 0.01
    assert sh.setup(*a) and so.setup(*a, sparse=True)
    st_h, st_o = sh.solve(), so.solve()
    th, to = sh.trace(), so.trace()
    rows = min(len(th), len(to))
    first = None
    for r in range(rows):
        bad = [c for c in range(th.shape[1]) if not (th[r, c] == to[r, c] or (th[r, c] != th[r, c] and to[r, c] != to[r, c]))]
        if bad:
            first = (r, bad)
            break
    same = first is None and len(th) == len(to) and st_h == st_o
    rh = sh.result(); ro = so.result()
    xeq = np.array_equal(np.asarray(rh["x"]), np.asarray(ro["x"])) if same else False
    line = f"{name:22s} status {st_h:3d}/{st_o:3d} iter {sh.info.iter:4d}/{so.info.iter:4d} " + ("TRACE BITWISE EQUAL" + (" + x" if xeq else " (x differs)") if same else
                                                                                                  f"first difference: row {first[0] if first else rows} cols {[COLS[c] for c in first[1]] if first else 'length'}")
    print(line, flush=True)
    if verbose and first:
        r = first[0]
        for c in first[1][:4]:
            print(f"    {COLS[c]:12s} device {th[r, c]!r:26} oracle {to[r, c]!r:26} rel {abs(th[r, c] - to[r, c]) / max(abs(to[r, c]), 1e-300):.2e}")
    return same, st_h == st_o, sh.info.iter == so.info.iter


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--all" in sys.argv:
        args = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "*.npz")))
    ks = 18 if "--multifrontal" in sys.argv else 1
    for a_ in sys.argv:
        if a_.startswith("--ks="):
            ks = int(a_[5:])
    tot = [0, 0, 0, 0]
    for nm in args:
        try:
            r = run(nm, verbose="--all" not in sys.argv, ks=ks)
        except Exception as e:  # noqa: BLE001
            print(f"{nm:22s} ERROR {type(e).__name__}: {e}", flush=True)
            continue
        if r is None:
            continue
        tot[0] += 1; tot[1] += r[0]; tot[2] += r[1]; tot[3] += r[2]
    print(f"fixtures {tot[0]}: trace bitwise equal {tot[1]}, same status {tot[2]}, same iteration count {tot[3]}")
