#!/usr/bin/env python3
"""Two builds of the library on the same dense systems (PIQP_AMD_LIB selects the shared library): factor + backend solve of random condensed systems, LL^T and
LDL^T, sizes with full and ragged last blocks -- the solutions are compared bit for bit and the solve is timed.
   python tools/chk_lib_variants.py piqp_amd/lib/exp/libpiqp_amd_base.so [piqp_amd/lib/libpiqp_amd.so]
a library may carry a debugging token: path@token (PIQP_AMD_DEBUG of that run), e.g.  piqp_amd/lib/libpiqp_amd.so piqp_amd/lib/libpiqp_amd.so@trsv_mfma"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
SIZES = (200, 512, 1000, 2048, 4096)


def child(out):
    import numpy as np
    import torch  # noqa
    import piqp_amd as hip
    from qp_gen import dense_strongly_convex_qp
    res = {}
    for n in SIZES:
        for ks in (0, 16):
            q = dense_strongly_convex_qp(n, 0, n, seed=7 + n, double_sided=True, exact_shift=False)
            k = hip.DenseKKT(hip.Data(**q), kkt_solver=ks)
            rng = np.random.default_rng(n)
            assert k.update_scalings_and_factor(1e-4, np.full(n, 1e-6), rng.uniform(0.5, 2.0, n))
            rhs = rng.standard_normal(n)
            lx, _, lz = k.solve(rhs, np.zeros(0), rng.standard_normal(n))
            k.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                k.solve(rhs, np.zeros(0), np.zeros(n))
            k.synchronize()
            res[f"x_{n}_{ks}"] = lx; res[f"z_{n}_{ks}"] = lz; res[f"t_{n}_{ks}"] = np.array([(time.perf_counter() - t0) / 20 * 1e3])
    np.savez(out, **res)


def main():
    import numpy as np
    libs = sys.argv[1:3] if len(sys.argv) > 2 else [sys.argv[1], os.path.join(ROOT, "piqp_amd", "lib", "libpiqp_amd.so")]
    outs = []
    for i, lib in enumerate(libs):
        out = f"/tmp/chk_lib_{i}.npz"
        lib, _, tok = lib.partition("@")
        env = dict(os.environ); env["PIQP_AMD_LIB"] = os.path.abspath(lib)
        env.pop("PIQP_AMD_DEBUG", None)
        if tok:
            env["PIQP_AMD_DEBUG"] = tok
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", out], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(dict(np.load(out)))
    a, b = outs
    bad = 0
    for key in sorted(a):
        if key.startswith("t_"):
            print(f"{key[2:]:12s} solve incl. host copies: {a[key][0]:.3f} ms -> {b[key][0]:.3f} ms")
        elif not np.array_equal(a[key], b[key]):
            bad += 1; print("DIFFERS", key, "max |d| %.3e of max |x| %.3e" % (float(np.abs(a[key] - b[key]).max()), float(np.abs(a[key]).max())))
    print("bitwise equal" if bad == 0 else f"{bad} arrays differ")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        main()
