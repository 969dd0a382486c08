#!/usr/bin/env python3
"""Where does the diagonal-block kernel differ from numpy?  Per 16 x 16 tile of the factor: max abs error; repeatability over `reps` launches.
usage: python tools/dbg_potrf_check.py [ldlt] [nb] [reps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import piqp_amd  # noqa: E402
import test_potrf_block_gpu as T  # noqa: E402

ldlt = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
A = T._spd(nb, 1)
out, rdiag, dvec, pack, info, differ = T._run(piqp_amd, A, ldlt, nb, reps)
U, d = T._ldl_ref(A)
Lref = (np.tril(U, -1) + np.diag(d)) if ldlt else U * np.sqrt(d)[None, :]
E = np.abs(np.tril(out[:nb, :nb]) - np.tril(Lref)).astype(np.float64)
print(f"ldlt={ldlt} nb={nb}: info {info}, repetitions that differ from the first: {differ} of {reps - 1}")
nt = (nb + 15) // 16
print("max abs error per tile (rows = block row):")
for i in range(nt):
    print(" ".join(f"{E[16*i:16*i+16, 16*j:16*j+16].max():8.1e}" if j <= i else "        " for j in range(nt)))
print("rdiag rel err", float(np.abs(rdiag[:nb] - (1 / d if ldlt else 1 / np.sqrt(d))).max() / np.abs(1 / d).max()))
