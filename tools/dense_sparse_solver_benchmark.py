#!/usr/bin/env python3
"""The shapes of the reference's benchmarks/src/dense_sparse_solver_benchmark.cpp:16-56 -- DenseSolver on rand::dense_strongly_convex_qp(dim, dim/2, dim/2) and
SparseSolver on the same recipe at 10 % density, dim = 4 ... 1024 (x 2), timed per solve() after one setup (the benchmark re-solves the same problem) --
device next to the oracle on one host core: status, iterations, milliseconds per solve.
   python tools/dense_sparse_solver_benchmark.py > gpurun_out/r04_dense_sparse_solver_benchmark.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scipy.sparse as sp


def sparse_variant(q, density, seed):
    """the dense recipe thinned to `density` (P keeps its diagonal shifted to stay strongly convex; every constraint row keeps one entry)"""
    rng = np.random.default_rng(seed)
    n = q["P"].shape[0]
    P = np.triu(q["P"], 1) * (rng.random((n, n)) < density)
    P = P + P.T
    P += (1e-2 + abs(np.linalg.eigvalsh(P).min())) * np.eye(n)
    out = dict(q); out["P"] = sp.csc_matrix(np.triu(P))
    x_sol = rng.standard_normal(n)
    for k in ("A", "G"):
        M = q[k] * (rng.random(q[k].shape) < density)
        M[np.arange(M.shape[0]), rng.integers(0, n, M.shape[0])] = 1.0
        out[k] = sp.csc_matrix(M)
    out["b"] = out["A"] @ x_sol
    Gx = out["G"] @ x_sol
    out["h_l"] = np.where(np.isfinite(q["h_l"]), Gx - 0.1, -np.inf); out["h_u"] = np.where(np.isfinite(q["h_u"]), Gx + 0.1, np.inf)
    out["x_l"] = np.where(np.isfinite(q["x_l"]), x_sol - 0.1, -np.inf); out["x_u"] = np.where(np.isfinite(q["x_u"]), x_sol + 0.1, np.inf)
    return out


def run(solver, a, kw, reps):
    assert solver.setup(*a, **kw)
    st = solver.solve()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); st = solver.solve(); ts.append(time.perf_counter() - t0)
    return st, solver.info.iter, min(ts) * 1e3


def main():
    import torch  # noqa: F401
    import piqp_amd as hip
    from oracle import pyorc as orc
    from qp_gen import dense_strongly_convex_qp
    print("kind    dim | oracle (1 core): status iters ms/solve | device: status iters ms/solve | device / oracle")
    for dim in (4, 8, 16, 32, 64, 128, 256, 512, 1024):
        q = dense_strongly_convex_qp(dim, dim // 2, dim // 2, seed=100 + dim)
        reps = 5 if dim <= 256 else 3
        so = orc.Solver(); so.settings.kkt_solver = 0
        sh = hip.DenseSolver(); sh.settings.kkt_solver = 0
        Pf = np.triu(q["P"]) + np.triu(q["P"], 1).T
        a = (Pf, q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
        ro = run(so, a, {}, reps); rh = run(sh, a, {}, reps)
        print(f"dense  {dim:5d} | {ro[0]:2d} {ro[1]:3d} {ro[2]:9.3f} | {rh[0]:2d} {rh[1]:3d} {rh[2]:9.3f} | {rh[2] / ro[2]:6.2f}x", flush=True)
        qs = sparse_variant(q, 0.1, 200 + dim)
        a = (qs["P"], qs["c"], qs["A"], qs["b"], qs["G"], qs["h_l"], qs["h_u"], qs["x_l"], qs["x_u"])
        so = orc.Solver(); so.settings.kkt_solver = orc.SPARSE_LDLT
        sh = hip.SparseSolver(); sh.settings.kkt_solver = hip.SPARSE_LDLT
        ro = run(so, a, {"sparse": True}, reps); rh = run(sh, a, {}, reps)
        print(f"sparse {dim:5d} | {ro[0]:2d} {ro[1]:3d} {ro[2]:9.3f} | {rh[0]:2d} {rh[1]:3d} {rh[2]:9.3f} | {rh[2] / ro[2]:6.2f}x", flush=True)
        if dim <= 128:
            # round 5: the same sparse QP as a batch of ONE through the batched solver (kkt_solver = sparse_multistage; the whole interior-point method of an
            # instance in one workgroup, one launch per solve, no host round trip per iteration) -- and as a batch of 256 copies (time per QP)
            for batch in (1, 256):
                try:
                    bs = hip.BatchSparseSolver()
                    rep = lambda v: None if v is None else np.repeat(np.asarray(v, dtype=np.float64)[None, :], batch, axis=0)
                    Pm, Am, Gm = sp.csc_matrix(qs["P"]), sp.csc_matrix(qs["A"]), sp.csc_matrix(qs["G"])
                    for M in (Pm, Am, Gm):
                        M.sort_indices()
                    ok = bs.setup(Pm, rep(Pm.data), rep(qs["c"]), Am, rep(Am.data), rep(qs["b"]), Gm, rep(Gm.data), rep(qs["h_l"]), rep(qs["h_u"]), x_l=rep(qs["x_l"]), x_u=rep(qs["x_u"]))
                    assert ok
                    bs.solve()
                    ts = []
                    for _ in range(reps):
                        t0 = time.perf_counter(); nsolved = bs.solve(); ts.append(time.perf_counter() - t0)
                    print(f"  batched solver, batch {batch:3d}: solved {nsolved}/{batch}, iterations {bs.info(0).iter}, {min(ts) * 1e3:9.3f} ms per launch = {min(ts) * 1e3 / batch:9.4f} ms per QP "
                          f"({min(ts) * 1e3 / batch / ro[2]:6.2f}x the oracle's time)", flush=True)
                except Exception as e:  # noqa: BLE001
                    print(f"  batched solver, batch {batch}: {type(e).__name__}: {str(e)[:120]}", flush=True)


if __name__ == "__main__":
    main()
