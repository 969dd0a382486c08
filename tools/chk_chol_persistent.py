#!/usr/bin/env python3
"""Persistent factorisation (k_chol_persistent) against the launch-per-panel path (PIQP_AMD_DEBUG=chol_launches): the factor, the reciprocal pivots
(through a solve) and the failure report must agree bit for bit; one child process per variant (the library reads the environment once).
   python tools/chk_chol_persistent.py [n ...]"""
import hashlib
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
            d = hip.Data(**q)
            k = hip.DenseKKT(d, kkt_solver=ks)
            rng = np.random.default_rng(n)
            x_reg = np.full(n, 1e-6); z_reg = rng.uniform(0.5, 2.0, n)
            reps = []
            for rep in range(3):
                t0 = time.perf_counter()
                ok = k.update_scalings_and_factor(1e-4, x_reg, z_reg)
                k.synchronize()
                dt = time.perf_counter() - t0
                F = np.tril(k.internal_factor())
                rhs = rng.standard_normal(n)
                lx, ly, lz = k.solve(rhs, np.zeros(0), np.zeros(n))
                reps.append(dict(ok=bool(ok), fac=hashlib.sha256(F.tobytes()).hexdigest(), x=hashlib.sha256(np.ascontiguousarray(lx).tobytes()).hexdigest(), ms=dt * 1e3,
                                 finite=bool(np.isfinite(F).all())))
            out[f"{n}/{ks}"] = reps
    print("RESULT " + json.dumps(out), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child([int(a) for a in sys.argv[2:]])
    sizes = [a for a in sys.argv[1:]] or ["384", "512", "1024", "2048", "4096"]
    res = {}
    for name, env in (("persistent", {}), ("launches", {"PIQP_AMD_DEBUG": "chol_launches"})):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + sizes, env=e, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(name, "FAILED", r.stdout[-1500:], r.stderr[-3000:])
            return 1
        res[name] = json.loads(line[0][7:])
    bad = 0
    for key in res["persistent"]:
        a, b = res["persistent"][key], res["launches"][key]
        same = all(x["fac"] == y["fac"] and x["x"] == y["x"] and x["ok"] == y["ok"] for x, y in zip(a, b))
        stable = all(x["fac"] == a[0]["fac"] for x in a)
        print(f"n/kkt_solver {key:8s} bitwise equal to the launch-per-panel path: {same}   repeatable: {stable}   ok: {[x['ok'] for x in a]}   "
              f"wall ms persistent {[round(x['ms'], 2) for x in a]} launches {[round(x['ms'], 2) for x in b]}")
        bad += (not same) or (not stable) or not all(x["ok"] and x["finite"] for x in a)
    print("ALL EQUAL" if not bad else f"{bad} MISMATCHES")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
