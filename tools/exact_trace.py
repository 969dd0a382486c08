"""timeline of one factorisation of the reference-order engine (PIQP_AMD_DEBUG=exact_trace): per ticket when it was drawn, when its waits were over, when it was done"""
import os, sys
import numpy as np
os.environ["PIQP_AMD_DEBUG"] = (os.environ.get("PIQP_AMD_DEBUG", "") + ",exact_trace").strip(",")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import piqp_amd as hip
from qp_io import load_qp
nm = sys.argv[1]
q = load_qp(nm)
a = (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
d = hip.SparseData(*a)
k = hip.SparseKKT(d, kkt_solver=hip.SPARSE_LDLT_EXACT)
rng = np.random.default_rng(1)
x_reg, z_reg = np.full(d.n, 1e-6), np.abs(rng.standard_normal(d.m)) + 0.1
for _ in range(3):
    k.update_scalings_and_factor(1e-4, x_reg, z_reg)
L = k.L
def item(what, dt, n=None):
    n = L.pq_kkt_exact_factor(k.h, what, None) if n is None else n
    out = np.zeros(max(int(n), 1), dt)
    L.pq_kkt_exact_factor(k.h, what, out.ctypes.data)
    return out[:int(n)]
tr = item(8, np.int64).reshape(-1, 4)
kind, ids, rlen, tptr, trows = item(9, np.int32), item(10, np.int32), item(11, np.int32), item(12, np.int32), item(13, np.int32)
t0 = tr[:, 0].min()
us = lambda c: (c - t0) / 100.0
tot = us(tr[:, 2].max())
print(f"{nm}: tickets {len(kind)}, total {tot:.1f} us, workgroups seen {len(set(tr[:, 3]))}")
for kd, name in ((0, "row pass"), (1, "path pass")):
    m = kind == kd
    if not m.any(): continue
    wait = (tr[m, 1] - tr[m, 0]) / 100.0; work = (tr[m, 2] - tr[m, 1]) / 100.0
    if kd == 0:
        steps = rlen[ids[m]]
    else:
        steps = np.array([rlen[trows[tptr[t]:tptr[t + 1]]].sum() for t in ids[m]])
    print(f"  {name}: {m.sum()} tickets, wait sum {wait.sum():.0f} us (mean {wait.mean():.2f}), work sum {work.sum():.0f} us (mean {work.mean():.2f}, max {work.max():.1f}), steps {steps.sum()}, "
          f"work per step {1e3 * work.sum() / max(steps.sum(), 1):.0f} ns; fit work = a + b steps: ", end="")
    A = np.vstack([np.ones(m.sum()), steps]).T
    coef = np.linalg.lstsq(A, work, rcond=None)[0]
    print(f"a = {coef[0]:.2f} us, b = {1e3 * coef[1]:.0f} ns/step")
# the last tickets to finish: the critical tail
order = np.argsort(tr[:, 2])[-12:]
for tk in order:
    print(f"   ticket {tk:5d} kind {kind[tk]} id {ids[tk]:5d} drawn {us(tr[tk, 0]):8.1f} ready {us(tr[tk, 1]):8.1f} done {us(tr[tk, 2]):8.1f}  steps {rlen[ids[tk]] if kind[tk] == 0 else rlen[trows[tptr[ids[tk]]:tptr[ids[tk] + 1]]].sum()}")
