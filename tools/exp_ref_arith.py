#!/usr/bin/env python3
"""Round 4: which part of the sparse fronts' arithmetic has to be the reference's (sparse/ldlt.hpp:151-158) for the device to end with the oracle's STATUS, and
what each choice does to the iteration counts of the other fixtures.  Variants = builds of sparse_kkt.hip with -DPQ_REF_MODE=m (piqp_amd/lib/exp/libpiqp_amd_m<m>.so;
mode 9 = the shipped library), one child process each; whole interior-point solves (sparse_ldlt) of every frozen mm_ / nl_ / nli_ / qp_ fixture.
    python tools/exp_ref_arith.py [--modes 0,1,3,4,5] > gpurun_out/r04_ref_arith.txt"""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
MODE_TEXT = {0: "rounds 1-3 arithmetic (fused, (y_r/d) y_c, summed Schur terms)", 1: "pivot loops per term", 3: "pivot loops + Schur complement per term",
             4: "diagonal entries per term", 5: "pivot loops + trailing diagonal per term",
             9: "the shipped library: mode 3 where every front is a one-workgroup front, mode 0 where the tree has fronts on the matrix cores"}


def _args(q):
    return (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])


def names():
    from qp_io import GOLDEN
    out = sorted(os.path.basename(f)[:-4] for pre in ("mm_", "nl_", "nli_", "qp_") for f in glob.glob(os.path.join(GOLDEN, pre + "*.npz")))
    return [n for n in out if n not in ("mm_BOYD1", "qp_c0_scenario_mpc")]


def child(kind):
    from qp_io import load_qp
    out = {}
    if kind == "oracle":
        from oracle import pyorc as orc
        for fma in (0, 1):
            for name in names():
                q = load_qp(name)
                so = orc.Solver(_L=orc.lib_fma() if fma else None); so.settings.kkt_solver = orc.SPARSE_LDLT
                if name.startswith("nl"):
                    so.settings.infeasibility_threshold = 0.01
                assert so.setup(*_args(q), sparse=True)
                st = int(so.solve())
                out.setdefault(name, []).append([st, int(so.info.iter)])
    else:
        import piqp_amd as hip
        for name in names():
            q = load_qp(name)
            sh = hip.SparseSolver(); sh.settings.kkt_solver = hip.SPARSE_LDLT
            if name.startswith("nl"):
                sh.settings.infeasibility_threshold = 0.01
            assert sh.setup(*_args(q))
            st = int(sh.solve())
            out[name] = [st, int(sh.info.iter)]
    print("RESULT " + json.dumps(out), flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    modes = [0, 1, 3, 4, 5, 9]
    if "--modes" in sys.argv:
        modes = [int(x) for x in sys.argv[sys.argv.index("--modes") + 1].split(",")]
    res = {}
    for v in ["oracle"] + modes:
        e = dict(os.environ)
        if v != "oracle":
            lib = os.path.join(ROOT, "piqp_amd", "lib", "exp", f"libpiqp_amd_m{v}.so")
            if v == 9:
                pass
            elif os.path.exists(lib):
                e["PIQP_AMD_LIB"] = lib
            else:
                print(f"(no build of mode {v})"); continue
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(v)], env=e, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(f"variant {v} failed:\n{r.stdout[-1500:]}\n{r.stderr[-1500:]}"); continue
        res[v] = json.loads(line[0][7:])
    orc = res["oracle"]
    print("whole solves, sparse_ldlt; entries status/iterations (1 solved, -1 max iter, -2 / -3 infeasible); oracle: no-FMA build | FMA-contracted build")
    for m in modes:
        if m in res:
            print(f"mode {m}: {MODE_TEXT.get(m, '')}")
    print(f"{'fixture':22s} {'oracle':>15s} " + " ".join(f"{'mode ' + str(m):>9s}" for m in modes if m in res))
    summ = {m: [0, 0, 0] for m in modes if m in res}
    for name in sorted(orc):
        (sa, ia), (sb, ib) = orc[name]
        cells, moved = [], False
        for m in modes:
            if m not in res:
                continue
            st, it = res[m][name]
            st_ok = st == sa or st == sb
            it_ok = st_ok and min(ia, ib) <= it <= max(ia, ib)
            summ[m][0] += 1; summ[m][1] += st_ok; summ[m][2] += it_ok
            moved |= not it_ok
            cells.append(f"{st:2d}/{it:3d}" + ("=" if it_ok else ("~" if st_ok else "!")))
        if moved:
            print(f"{name:22s} {sa:2d}/{ia:3d} |{sb:2d}/{ib:3d}  " + " ".join(f"{c:>9s}" for c in cells))
    print("(only fixtures where some variant leaves the oracle's range are listed; '=' within the range of the two oracle builds, '~' same status, '!' other status)")
    for m, (n, s_ok, i_ok) in summ.items():
        print(f"mode {m}: {n} fixtures, status equal to an oracle build on {s_ok}, iteration count within the oracle builds' range on {i_ok}")


if __name__ == "__main__":
    main()
