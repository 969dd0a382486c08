#!/usr/bin/env python3
"""In-kernel timeline of the 128 x 128 diagonal-block factorisation (potrf_block, csrc/dense_kernels.hip): shader-clock stamps per
16-column step.  usage: python tools/dbg_potrf.py [ldlt]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import piqp_amd  # noqa: E402

L = piqp_amd._lib.load()
ldlt = int(sys.argv[1]) if len(sys.argv) > 1 else 0
us = C.c_double()
ts = np.zeros(64, dtype=np.int64)
piqp_amd._lib.check(L.pq_microbench_potrf_block(0, ldlt, 20, C.byref(us), ts.ctypes.data))
print(f"k_potrf_diag<ldlt={ldlt}>: {us.value:.2f} us per launch (hipEvents)")
t = ts.reshape(8, 8)
t0 = t[0, 0]
print("step  start  factor16 | wait  subst | wait  update | (inverse done)   [shader-clock cycles, 2.4 GHz max: 1000 cycles >= 0.42 us]")
for k in range(8):
    r = t[k]
    f = lambda a: f"{int(a - t0):7d}" if a else "      -"
    print(f"{k}  {f(r[0])} {int(r[1]-r[0]):6d} | {f(r[2])} {int(r[3]-r[2]) if r[3] else 0:6d} | {f(r[4])} {int(r[5]-r[4]) if r[5] else 0:6d} | {f(r[6])}")
print("total cycles first stamp -> last stamp:", int(t.max() - t0))
