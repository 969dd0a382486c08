#!/usr/bin/env python3
"""Host-only: the assembly tree the symbolic analysis builds for a generated C3-style problem (no GPU needed).
   PIQP_AMD_DEBUG=tree_profile,sn_stats python tools/tree_profile.py [spread] [row_nnz] [n]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import piqp_amd as hip
from piqp_amd import _lib
from qp_gen import c3_problem
spread = int(sys.argv[1]) if len(sys.argv) > 1 else 300
row_nnz = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = int(sys.argv[3]) if len(sys.argv) > 3 else 50000
a = c3_problem(n=n, p=n * 2 // 5, m=n * 3 // 5, seed=44, spread=spread, row_nnz=row_nnz)
d = hip.SparseData(*a); desc = d.descriptor()
L = _lib.load()
t0 = time.perf_counter()
N = L.pq_sparse_partition_plan(C.byref(desc), 0, 1, None, 0, None)
print("N", N, "analysis %.2f s" % (time.perf_counter() - t0))
