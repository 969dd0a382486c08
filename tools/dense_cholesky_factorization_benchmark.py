#!/usr/bin/env python3
"""The reference's factor-only benchmark (/root/reference/benchmarks/src/dense_cholesky_factorization_benchmark.cpp:16-109: compute() of a dense positive definite
matrix, n = 4, 8, ..., 1024, for Eigen::LLT and LDLTNoPivot, Lower and Upper) through the device objects (piqp_amd.LLT / piqp_amd.LDLTNoPivot, pq_dense_factor_*),
the matrix resident in device memory, next to the oracle's restatement of the same classes on one host core.
    python tools/dense_cholesky_factorization_benchmark.py [max_n] > profiles/r06_dense_cholesky_factorization_benchmark.txt
device columns: median over the repetitions of (hipEvent time of the factorisation launches) / (wall time of the whole compute(): symmetric completion of the
named triangle straight into the factor buffer, factorisation, one 4-byte status read-back).  Eigen's LDLT with pivoting (BM_EIGEN_LDLT_*) has no counterpart here:
it is not on PIQP's path."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import piqp_amd as hip  # noqa: E402
from oracle import pyorc as orc  # noqa: E402  (the CPU side of the table)


def spd_upper(n, seed):
    rng = np.random.default_rng(seed)
    U = np.triu(rng.standard_normal((n, n)), 1)
    S = U + U.T
    S += (1.0 + abs(np.linalg.eigvalsh(S).min())) * np.eye(n)
    return np.triu(S), S


def oracle_us(S, ldlt, budget_s=0.3):
    n = S.shape[0]
    L = orc.lib()
    w = np.zeros(n)
    reps, t_all = 0, 0.0
    best = []
    while t_all < budget_s or reps < 3:
        a = np.asfortranarray(S.copy())
        t0 = time.perf_counter()
        ret = L.orc_ldlt_no_pivot_compute(a.ctypes.data_as(orc._dp), n, n, w.ctypes.data_as(orc._dp)) if ldlt else L.orc_llt_compute(a.ctypes.data_as(orc._dp), n, n)
        dt = time.perf_counter() - t0
        assert ret == -1
        best.append(dt); t_all += dt; reps += 1
    return float(np.median(best)) * 1e6


def device_us(cls, n, uplo, P_up, reps=30):
    A = P_up if uplo == hip.UPPER else P_up.T
    t = torch.from_numpy(np.ascontiguousarray(A.T)).cuda()  # row-major transpose = the column-major matrix
    f = cls(n, uplo)
    for _ in range(3):
        f.compute_colmajor(t)
    assert f.info() == 0
    dev, wall = [], []
    for _ in range(reps):
        f.compute_colmajor(t)
        d, w = f.last_ms()
        dev.append(d); wall.append(w)
    return float(np.median(dev)) * 1e3, float(np.median(wall)) * 1e3


def main():
    max_n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    print("# compute() of a dense positive definite matrix, microseconds; device = factorisation launches / whole compute() call")
    print(f"{'n':>6} | {'oracle LLT':>10} {'oracle LDLT':>11} | {'LLT Lower':>19} {'LLT Upper':>19} | {'LDLTNoPivot Lower':>19} {'LDLTNoPivot Upper':>19} | device compute() / oracle (LLT, LDLT)")
    n = 4
    sizes = []
    while n <= max_n:
        sizes.append(n); n *= 2
    for n in sizes + [s for s in (2048, 4096) if s <= max_n * 4 and max_n >= 1024]:
        P_up, S = spd_upper(n, n)
        o_llt = oracle_us(S, False) if n <= 2048 else float("nan")
        o_ldlt = oracle_us(S, True) if n <= 2048 else float("nan")
        cells = []
        for cls in (hip.LLT, hip.LDLTNoPivot):
            for uplo in (hip.LOWER, hip.UPPER):
                cells.append(device_us(cls, n, uplo, P_up, reps=30 if n <= 1024 else 10))
        fmt = lambda c: f"{c[0]:8.1f} / {c[1]:8.1f}"
        print(f"{n:6d} | {o_llt:10.1f} {o_ldlt:11.1f} | {fmt(cells[0])} {fmt(cells[1])} | {fmt(cells[2])} {fmt(cells[3])} | {cells[0][1] / o_llt:7.2f}x {cells[2][1] / o_ldlt:7.2f}x", flush=True)


if __name__ == "__main__":
    main()
