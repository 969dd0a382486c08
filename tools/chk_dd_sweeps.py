#!/usr/bin/env python3
"""Sweeps with the inverted 128-row diagonal blocks (PIQP_AMD_DEBUG=inv_sweeps=1; the default from eight block rows on, round 5) against the substitution form
(inv_sweeps=0): backend solve of the condensed system on random right-hand sides -- residual of both in extended precision against the device's own factor L L^T
(or L D L^T), agreement of the two solutions, time.  With PIQP_AMD_DEBUG=...,trsv_ts the --child form prints the forward sweep's timeline (100 MHz clock).
   python tools/chk_dd_sweeps.py [n ...]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(sizes):
    import numpy as np
    import piqp_amd as hip
    from qp_gen import dense_strongly_convex_qp
    out = {}
    for n in sizes:
        for ks in (0, 16):
            q = dense_strongly_convex_qp(n, 0, n, seed=7 + n, double_sided=True, exact_shift=False)
            k = hip.DenseKKT(hip.Data(**q), kkt_solver=ks)
            rng = np.random.default_rng(n)
            x_reg = np.full(n, 1e-6); z_reg = rng.uniform(0.5, 2.0, n)
            assert k.update_scalings_and_factor(1e-4, x_reg, z_reg)
            F = np.tril(k.internal_factor()).astype(np.longdouble)
            rhs = rng.standard_normal(n)
            lx, _, _ = k.solve(rhs, np.zeros(0), np.zeros(n))   # condensed: K lx = rhs + GT (zinv o 0)
            xl = lx.astype(np.longdouble)
            if ks == 16:
                D = np.diag(F).copy(); Lm = F.copy(); np.fill_diagonal(Lm, 1.0)
                Kx = Lm @ (D * (Lm.T @ xl))   # (matrix-vector products only: a longdouble matrix-matrix product of this size takes minutes)
            else:
                Kx = F @ (F.T @ xl)
            res = float(np.abs(Kx - rhs).max() / np.abs(rhs).max())
            k.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                k.solve(rhs, np.zeros(0), np.zeros(n))
            k.synchronize()
            ms = (time.perf_counter() - t0) / 20 * 1e3
            np.save(f"/tmp/chk_dd_{os.environ.get('CHK_TAG', 'x')}_{n}_{ks}.npy", np.asarray(lx))
            out[f"{n}/{ks}"] = dict(res=res, ms=ms)
    print("RESULT " + json.dumps(out), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child([int(a) for a in sys.argv[2:]])
    import numpy as np
    sizes = sys.argv[1:] or ["200", "1000", "2048", "4096"]
    res = {}
    variants = [("inv", "inv_sweeps=1"), ("subst", "inv_sweeps=0")]
    if os.environ.get("CHK_VARIANTS"):  # e.g. CHK_VARIANTS="inv1:inv_sweeps=1,inv2:inv_sweeps=2" -- each compared with the substitution
        variants = [tuple(v.split(":")) for v in os.environ["CHK_VARIANTS"].split(",")] + [("subst", "inv_sweeps=0")]
    for name, tok in variants:
        e = dict(os.environ); e["PIQP_AMD_DEBUG"] = tok; e["CHK_TAG"] = name
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + sizes, env=e, capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(name, "FAILED", r.stdout[-1500:], r.stderr[-3000:]); return 1
        res[name] = json.loads(line[0][7:])
    bad = 0
    for vname, _ in variants[:-1]:
      for key in res[vname]:
        n, ks = key.split("/")
        xa = np.load(f"/tmp/chk_dd_{vname}_{n}_{ks}.npy"); xb = np.load(f"/tmp/chk_dd_subst_{n}_{ks}.npy")
        dx = float(np.abs(xa - xb).max() / np.abs(xb).max())
        a, b = res[vname][key], res["subst"][key]
        print(f"n/kkt_solver {key:8s} residual vs own factor: {vname} {a['res']:.2e}  substitution {b['res']:.2e}   |x_{vname} - x_subst| / |x| = {dx:.2e}   solve wall ms (host pointers): {vname} {a['ms']:.3f}  substitution {b['ms']:.3f}")
        bad += (a["res"] > 2.0 * b["res"] + 1e-15) or dx > 1e-6
    print("OK" if not bad else f"{bad} PROBLEMS")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
