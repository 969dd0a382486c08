#!/usr/bin/env python3
"""Where do the multi-millisecond steps of the dense path come from (round-5 review: n = 1024 runs of 0.75 / 0.79 / 14.5 ms per step in the bench's size sweep)?
Per-STEP times (1 update_scalings_and_factor + 2 KKTSystem::solve + a stream synchronisation) of thousands of steps at n = 1024 / 1100 / 1536, the host time of every
call of a step kept apart (a call that blocks on the host shows in that call, work that is slow on the device shows in the synchronisation), in two regimes:
  one handle   : the steady state of a solver;
  fresh handles: a new KKTSystem every 7 steps (2 untimed + 5), the regime of the size sweep (bench.py dense_size_sweep) -- what a new handle's first steps pay.
    python tools/soak_stalls.py [steps] > profiles/r06_soak_dense.txt"""
import gc
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401

import piqp_amd  # noqa: E402
from qp_gen import dense_strongly_convex_qp, random_vars  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 2000


def make(n, seed=3):
    q = dense_strongly_convex_qp(n, 0, n, seed=seed, double_sided=True, exact_shift=False)
    k = piqp_amd.KKTSystem(piqp_amd.Data(**q), piqp_amd.default_settings(kkt_solver=0))
    rng = np.random.default_rng(0)
    state = {kk: torch.from_numpy(v).cuda() for kk, v in random_vars(n, 0, n, rng, positive=True).items()}
    rhs = {kk: torch.from_numpy(v).cuda() for kk, v in random_vars(n, 0, n, rng).items()}
    lhs = {kk: torch.zeros_like(v) for kk, v in rhs.items()}
    return k, state, rhs, lhs


def step(k, state, rhs, lhs):
    t0 = time.perf_counter()
    ok = k.update_scalings_and_factor(False, 1e-6, 1e-4, state)
    t1 = time.perf_counter()
    k.solve(rhs, lhs)
    t2 = time.perf_counter()
    k.solve(rhs, lhs)
    t3 = time.perf_counter()
    k.synchronize()
    t4 = time.perf_counter()
    assert ok
    return (t4 - t0, t1 - t0, t2 - t1, t3 - t2, t4 - t3)


def report(tag, rows):
    a = np.array(rows) * 1e3
    tot = a[:, 0]
    med = float(np.median(tot))
    print(f"{tag}: {len(tot)} steps, median {med:.3f} ms, p90 {np.percentile(tot, 90):.3f}, p99 {np.percentile(tot, 99):.3f}, max {tot.max():.3f}, p99 / median {np.percentile(tot, 99) / med:.2f}")
    slow = np.where(tot > 3.0 * med)[0]
    print(f"    steps above 3 x median: {len(slow)}" + (" at " + " ".join(str(int(i)) for i in slow[:20]) if len(slow) else ""))
    for i in slow[:8]:
        print(f"      step {int(i):5d}: total {a[i, 0]:8.3f} ms = factor call {a[i, 1]:7.3f} + solve call {a[i, 2]:7.3f} + solve call {a[i, 3]:7.3f} + synchronise {a[i, 4]:7.3f}   "
              f"(median step: {np.median(a[:, 1]):.3f} + {np.median(a[:, 2]):.3f} + {np.median(a[:, 3]):.3f} + {np.median(a[:, 4]):.3f})")
    sys.stdout.flush()


def main():
    gc_was = gc.isenabled()
    for n in (1024, 1100, 1536):
        k, state, rhs, lhs = make(n)
        for _ in range(20):
            step(k, state, rhs, lhs)
        rows = [step(k, state, rhs, lhs) for _ in range(STEPS)]
        report(f"n = {n}, one handle, Python's collector on ", rows)
        gc.disable()
        rows = [step(k, state, rhs, lhs) for _ in range(STEPS)]
        report(f"n = {n}, one handle, Python's collector OFF", rows)
        if gc_was:
            gc.enable()
        del k
        # the size sweep's regime: a new handle, two untimed steps, five timed ones
        rows, first = [], []
        for rep in range(max(20, STEPS // 50)):
            k, state, rhs, lhs = make(n, seed=900 + n)
            w = [step(k, state, rhs, lhs) for _ in range(2)]
            first.append(w[0])
            rows += [step(k, state, rhs, lhs) for _ in range(5)]
            del k
        report(f"n = {n}, fresh handle every 7 steps (steps 3-7 of each)", rows)
        report(f"n = {n}, fresh handle: the FIRST step of each handle      ", first)


if __name__ == "__main__":
    main()
