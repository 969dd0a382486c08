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
tr = tr[:len(kind)]  # (per-XCD queues: one slot per row, in the order of the queues)
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

# ---- the substitution (k_ul_solve2): tickets 0 .. ntask-1 forward passes in tsort order, ntask .. 2 ntask-1 backward passes in the opposite order
rx, ry, rz = rng.standard_normal(d.n), rng.standard_normal(d.p), rng.standard_normal(d.m)
for _ in range(3):
    k.solve(rx, ry, rz)
st = item(14, np.int64).reshape(-1, 4)
tsort, nU, tpar, Lp = item(15, np.int32), item(16, np.int32), item(17, np.int32), item(1, np.int32)
nt = len(tsort)
W = np.diff(tptr)
colent = np.diff(Lp)
bent = np.array([colent[trows[tptr[t]:tptr[t + 1]]].sum() for t in range(nt)])
s0 = st[:, 0].min()
su = lambda c: (c - s0) / 100.0
print(f"substitution: {2 * nt} tickets, total {su(st[:, 2].max()):.1f} us; forward done at {su(st[:nt, 2].max()):.1f} us, first backward ready at {su(st[nt:, 1].min()):.1f} us")
for name, sl, task_of, size in (("forward", slice(0, nt), lambda tk: tsort[tk], lambda t: nU[t] + W[t]), ("backward", slice(nt, 2 * nt), lambda tk: tsort[2 * nt - 1 - tk], lambda t: bent[t])):
    tks = np.arange(2 * nt)[sl]
    tasks = np.array([task_of(tk) for tk in tks])
    wait = (st[sl, 1] - st[sl, 0]) / 100.0; work = (st[sl, 2] - st[sl, 1]) / 100.0
    sz = np.array([size(t) for t in tasks])
    A = np.vstack([np.ones(len(tks)), W[tasks], sz]).T
    coef = np.linalg.lstsq(A, work, rcond=None)[0]
    print(f"  {name}: work sum {work.sum():.0f} us, mean {work.mean():.2f}, max {work.max():.1f}; fit work = {coef[0]:.2f} us + {1e3 * coef[1]:.0f} ns x rows + {1e3 * coef[2]:.1f} ns x {'table rows' if name == 'forward' else 'entries'}")
    big = np.argsort(work)[-6:]
    for i in big:
        t = tasks[i]
        print(f"     task {t:5d} W {W[t]:2d} size {sz[i]:6d}: drawn {su(st[tks[i], 0]):8.1f} ready {su(st[tks[i], 1]):8.1f} done {su(st[tks[i], 2]):8.1f}  work {work[i]:7.1f} us = {1e3 * work[i] / max(sz[i], 1):.0f} ns per unit")
    # critical chain: follow the latest-finishing ticket back through what it waited for
order = np.argsort(st[:, 2])[-8:]
for tk in order:
    t = tsort[tk] if tk < nt else tsort[2 * nt - 1 - tk]
    print(f"   last: ticket {tk:5d} ({'fwd' if tk < nt else 'bwd'}) task {t:5d} W {W[t]:2d} drawn {su(st[tk, 0]):8.1f} ready {su(st[tk, 1]):8.1f} done {su(st[tk, 2]):8.1f}")
