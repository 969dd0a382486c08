#!/usr/bin/env python3
"""Results of the batched solver on the C4 recipe, every field and counter, to an .npz -- or compared bit for bit with one written before (a change of the kernel
that must not change any arithmetic: reductions, layouts, prefetching).
    python tools/dump_batch_results.py dump <file.npz> [batch ...]      python tools/dump_batch_results.py compare <file.npz>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401

import piqp_amd as hip  # noqa: E402
from qp_gen import mpc_batch  # noqa: E402

FIELDS = ("x", "y", "z_bl", "z_bu", "s_bl", "s_bu")


def run(B):
    mb = mpc_batch(B, seed=1000)
    bs = hip.BatchSparseSolver()
    assert bs.setup(mb["P_pattern"], mb["P_values"], mb["c"], mb["A_pattern"], mb["A_values"], mb["b"], x_l=mb["x_l"], x_u=mb["x_u"])
    bs.set_start_order(False)
    solved = bs.solve()
    ms = []
    for _ in range(5):
        bs.solve(); ms.append(bs.last_kernel_ms()[0])
    out = {f"{f}_{B}": bs.result(f) for f in FIELDS}
    out[f"info_{B}"] = np.array([[bs.info(i).status, bs.info(i).iter, bs.info(i).n_factor, bs.info(i).n_solve] for i in range(B)])
    out[f"obj_{B}"] = np.array([[bs.info(i).primal_obj, bs.info(i).dual_obj, bs.info(i).primal_res, bs.info(i).dual_res, bs.info(i).mu] for i in range(B)])
    print(f"B = {B}: solved {solved}, kernel ms median {sorted(ms)[2]:.3f} min {min(ms):.3f}", flush=True)
    return out


mode, path = sys.argv[1], sys.argv[2]
if mode == "dump":
    sizes = [int(a) for a in sys.argv[3:]] or [1, 64, 1024, 8192]
    out = {"sizes": np.array(sizes)}
    for B in sizes:
        out.update(run(B))
    np.savez(path, **out)
else:
    ref = np.load(path)
    bad = []
    for B in ref["sizes"]:
        cur = run(int(B))
        for k, v in cur.items():
            same = np.array_equal(v.view(np.uint64) if v.dtype == np.float64 else v, ref[k].view(np.uint64) if ref[k].dtype == np.float64 else ref[k])
            if not same:
                bad.append(k)
    print("arrays that differ in any bit:", bad or "none")
    sys.exit(1 if bad else 0)
