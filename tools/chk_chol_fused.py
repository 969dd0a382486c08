#!/usr/bin/env python3
"""Fused assembly + factorisation (k_chol_persistent with assembly tasks, round 4) against the round-3 sequence (PIQP_AMD_DEBUG=chol_unfused: assembly launch, then
the persistent factorisation): the factor must agree to rounding (the K-sliced tiles of the fused path add their partial sums in slice order, the assembly launch
splits only its tail tiles), be repeatable bit for bit, report the same status, and solve to the same residual; one child process per variant.
   python tools/chk_chol_fused.py [n ...]   (m = n, p = 0; "n:m:p" for other shapes)"""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(shapes):
    import numpy as np
    import piqp_amd as hip
    from qp_gen import dense_strongly_convex_qp
    out = {}
    for sh in shapes:
        n, m, p = sh
        for ks in (0, 16):
            q = dense_strongly_convex_qp(n, p, m, seed=7 + n, double_sided=True, exact_shift=False)
            d = hip.Data(**q)
            k = hip.DenseKKT(d, kkt_solver=ks)
            rng = np.random.default_rng(n)
            x_reg = np.full(n, 1e-6); z_reg = rng.uniform(0.5, 2.0, m)
            reps = []
            for rep in range(4):
                k.synchronize()
                t0 = time.perf_counter()
                ok = k.update_scalings_and_factor(1e-4, x_reg, z_reg)
                k.synchronize()
                dt = time.perf_counter() - t0
                F = np.tril(k.internal_factor())
                rhs = rng.standard_normal(n) if rep == 0 else rhs
                lx, ly, lz = k.solve(rhs, np.zeros(p), np.zeros(m))
                reps.append(dict(ok=bool(ok), fac=hashlib.sha256(F.tobytes()).hexdigest(), ms=dt * 1e3, finite=bool(np.isfinite(F).all())))
            np.save(f"/tmp/chk_fused_{os.environ.get('CHK_TAG', 'x')}_{n}_{m}_{p}_{ks}.npy", F)
            np.save(f"/tmp/chk_fused_{os.environ.get('CHK_TAG', 'x')}_{n}_{m}_{p}_{ks}_x.npy", np.asarray(lx))
            out[f"{n}:{m}:{p}/{ks}"] = reps
    print("RESULT " + json.dumps(out), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child([tuple(int(v) for v in a.split(":")) for a in sys.argv[2:]])
    import numpy as np
    shapes = []
    for a in (sys.argv[1:] or ["512", "1024", "2048", "4096", "1024:2048:64", "1536:512:0"]):
        v = [int(x) for x in a.split(":")]
        shapes.append((v[0], v[1] if len(v) > 1 else v[0], v[2] if len(v) > 2 else 0))
    args = [":".join(str(x) for x in sh) for sh in shapes]
    res = {}
    for name, env in (("fused", {"PIQP_AMD_DEBUG": "chol_fused"}), ("unfused", {})):
        e = dict(os.environ); e.update(env); e["CHK_TAG"] = name
        if os.environ.get("PIQP_AMD_DEBUG") and env.get("PIQP_AMD_DEBUG"):
            e["PIQP_AMD_DEBUG"] = os.environ["PIQP_AMD_DEBUG"] + "," + env["PIQP_AMD_DEBUG"]
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + args, env=e, capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if r.stderr.strip():
            print(f"--- {name} stderr (tail)\n" + r.stderr[-6000:])
        if not line:
            print(name, "FAILED", r.stdout[-1500:])
            return 1
        res[name] = json.loads(line[0][7:])
    bad = 0
    for key in res["fused"]:
        a, b = res["fused"][key], res["unfused"][key]
        shape, ks = key.split("/")
        n, m, p = (int(v) for v in shape.split(":"))
        Fa = np.load(f"/tmp/chk_fused_fused_{n}_{m}_{p}_{ks}.npy"); Fb = np.load(f"/tmp/chk_fused_unfused_{n}_{m}_{p}_{ks}.npy")
        xa = np.load(f"/tmp/chk_fused_fused_{n}_{m}_{p}_{ks}_x.npy"); xb = np.load(f"/tmp/chk_fused_unfused_{n}_{m}_{p}_{ks}_x.npy")
        rel = float(np.abs(Fa - Fb).max() / np.abs(Fb).max())
        relx = float(np.abs(xa - xb).max() / np.abs(xb).max())
        stable = all(x["fac"] == a[0]["fac"] for x in a)
        okeq = all(x["ok"] == y["ok"] for x, y in zip(a, b))
        print(f"n:m:p/kkt_solver {key:18s} max |dF| / max |F| = {rel:.2e}  |dx| = {relx:.2e}  repeatable: {stable}  ok: {[x['ok'] for x in a]} / {[x['ok'] for x in b]}   "
              f"wall ms fused {[round(x['ms'], 2) for x in a]} unfused {[round(x['ms'], 2) for x in b]}")
        bad += (rel > 1e-11) or (relx > 1e-8) or (not stable) or (not okeq) or not all(x["finite"] for x in a)
    print("ALL OK" if not bad else f"{bad} PROBLEMS")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
