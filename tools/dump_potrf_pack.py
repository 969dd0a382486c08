#!/usr/bin/env python3
"""Outputs of the diagonal-block kernel (pq_debug_potrf_block) on fixed inputs, saved / compared bit for bit: a change of the kernel that claims to be bitwise neutral is run
against the file a build before the change wrote.   python tools/dump_potrf_pack.py save|compare <file.npz>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import piqp_amd  # noqa: E402
import test_potrf_block_gpu as T  # noqa: E402

mode, path = sys.argv[1], sys.argv[2]
out = {}
for ldlt in (0, 1):
    for seed, nb, cond in ((1, 128, 1e3), (7, 128, 1e8), (3, 100, 1e3), (5, 17, 10.0)):
        A = T._spd(nb, seed, cond=cond)
        L, rdiag, dvec, pack, info, differ = T._run(piqp_amd, A, ldlt, nb, reps=3)
        key = f"{ldlt}_{seed}_{nb}"
        out[key + "_L"] = np.tril(L); out[key + "_r"] = rdiag; out[key + "_d"] = dvec; out[key + "_p"] = pack
if mode == "save":
    np.savez(path, **out)
    print("saved", len(out), "arrays to", path)
else:
    ref = np.load(path)
    bad = [k for k in out if not np.array_equal(out[k].view(np.uint64), ref[k].view(np.uint64))]
    print("arrays that differ in any bit:", bad if bad else "none", f"({len(out)} compared)")
    for k in bad[:6]:
        d = np.abs(out[k] - ref[k]); print("  ", k, "max abs diff", float(np.nanmax(d)), "entries", int((out[k].view(np.uint64) != ref[k].view(np.uint64)).sum()))
