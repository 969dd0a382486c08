#!/usr/bin/env python3
"""Counter-experiment for the trajectory-sensitive fixtures: whole interior-point solves of the listed problems on the
device with the fill-reducing ordering forced to the reference's (PIQP_AMD_ORDERING=amd: Eigen-style AMD, sparse/ordering.hpp:67-84), to nested
dissection, and with the default cost model, for the device-resident loop and the host-side loop, next to the oracle.  One child process per
variant (the library reads the environment once).   python tools/exp_ordering_parity.py [--all] [name ...] > gpurun_out/r03_ordering_parity.txt"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SENSITIVE = ["mm_QBEACONF", "mm_QCAPRI", "mm_QETAMACR", "mm_QGROW7", "mm_QPILOTNO", "mm_QSHIP08L", "mm_QSHIP08S", "nl_fffff800"]
ORACLE_MISSES = ["nl_bnl2", "nl_pilot-we", "nli_ceria3d", "nli_cplex2", "nli_qual"]


def _args(q):
    return (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])


def child(names, want_oracle):
    import piqp_amd as hip
    from qp_io import load_qp
    out = {}
    for name in names:
        q = load_qp(name)
        netlib = name.startswith("nl")
        sh = hip.SparseSolver(); sh.settings.kkt_solver = hip.SPARSE_LDLT
        if netlib:
            sh.settings.infeasibility_threshold = 0.01
        assert sh.setup(*_args(q))
        st = sh.solve()
        rec = {"st": int(st), "it": int(sh.info.iter), "obj": float(sh.info.primal_obj)}
        try:
            rec["ordering"] = sh.kkt_stats().get("ordering", "?")
        except Exception:
            pass
        if want_oracle:
            from oracle import pyorc as orc
            so = orc.Solver(); so.settings.kkt_solver = orc.SPARSE_LDLT
            if netlib:
                so.settings.infeasibility_threshold = 0.01
            assert so.setup(*_args(q), sparse=True)
            rec["o_st"] = int(so.solve()); rec["o_it"] = int(so.info.iter); rec["o_obj"] = float(so.info.primal_obj)
        out[name] = rec
    print("RESULT " + json.dumps(out), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(sys.argv[3:], sys.argv[2] == "1")
    names = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--all" in sys.argv:
        import glob
        from qp_io import GOLDEN
        names = sorted(os.path.basename(f)[:-4] for pre in ("mm_", "nl_", "nli_") for f in glob.glob(os.path.join(GOLDEN, pre + "*.npz")))
        names = [n for n in names if n not in ("mm_CONT-201", "mm_BOYD1")]
    if not names:
        names = SENSITIVE + ORACLE_MISSES
    variants = [("default", {}), ("amd", {"PIQP_AMD_ORDERING": "amd"}), ("nd", {"PIQP_AMD_ORDERING": "nd"}),
                ("amd+hostloop", {"PIQP_AMD_ORDERING": "amd", "PIQP_AMD_HOST_IPM": "1"}), ("default+hostloop", {"PIQP_AMD_HOST_IPM": "1"})]
    extra = os.environ.get("EXP_VARIANTS")
    if extra:
        variants = [v for v in variants if v[0] in extra.split(",")]
    res = {}
    for i, (vn, env) in enumerate(variants):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "1" if i == 0 else "0"] + names, env=e, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(f"variant {vn} failed:\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}")
            continue
        res[vn] = json.loads(line[0][7:])
    base = res.get(variants[0][0], {})
    print("status: 1 solved, -1 max iter, -2 primal infeasible, -3 dual infeasible, -8 numerics; entries are status/iterations")
    print(f"{'problem':16s} {'oracle':>9s} " + " ".join(f"{vn:>17s}" for vn, _ in variants))
    for name in names:
        o = base.get(name, {})
        cells = []
        for vn, _ in variants:
            r = res.get(vn, {}).get(name)
            cells.append(f"{r['st']:2d}/{r['it']:3d}" + ("=" if r and o and r["st"] == o.get("o_st") and r["it"] == o.get("o_it") else " ") if r else "   -   ")
        print(f"{name:16s} {o.get('o_st', 0):2d}/{o.get('o_it', 0):3d}   " + " ".join(f"{c:>17s}" for c in cells))
    print("('=' marks status and iteration count equal to the oracle's)")


if __name__ == "__main__":
    main()
