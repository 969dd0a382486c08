"""Experiment (round 4): what a cancellation-aware pivot failure signal does to the reference's trajectories.  The oracle's up-looking LDLt
(sparse/ldlt.hpp:101-169 restated) is run with the hook ORC_EXP_CANCEL_TOL = t: pivot k additionally fails when |D[k]| <= t * max(|a_kk|, |l_ki y_i|).
For every frozen Maros-Meszaros / netlib / qp fixture: status / iterations without the hook and with t in TOLS.  CPU only.
NEEDS an oracle built with the hook of tools/exp_cancel_pivot_hook.c.txt pasted in (round 5 removed it from oracle/orc_sparse.c).
usage: python tools/exp_cancel_pivot.py [workers]"""
import glob, os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOLS = ["0", "1.1102230246251565e-16", "2.220446049250313e-16"]

def one(name, tol, fma):
    os.environ["ORC_EXP_CANCEL_TOL"] = tol
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
    from qp_io import load_qp
    from oracle import pyorc as orc
    q = load_qp(name)
    so = orc.Solver(_L=orc.lib_fma() if fma else None); so.settings.kkt_solver = orc.SPARSE_LDLT
    if name.startswith("nl"):
        so.settings.infeasibility_threshold = 0.01
    assert so.setup(q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"], sparse=True)
    st = so.solve()
    return st, so.info.iter

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--one":
        name = sys.argv[2]
        out = {}
        for fma in (0, 1):
            for t in TOLS:
                # the hook reads its environment variable once per process
                r = subprocess.run([sys.executable, __file__, "--leaf", name, t, str(fma)], capture_output=True, text=True)
                out[f"{'fma' if fma else 'std'}:{t}"] = r.stdout.strip()
        print(json.dumps({name: out}))
    elif len(sys.argv) > 2 and sys.argv[1] == "--leaf":
        st, it = one(sys.argv[2], sys.argv[3], int(sys.argv[4]))
        print(f"{st}/{it}")
    else:
        from concurrent.futures import ThreadPoolExecutor
        workers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
        names = sorted(os.path.basename(f)[:-4] for pat in ("mm_*.npz", "nl_*.npz", "nli_*.npz", "qp_*.npz") for f in glob.glob(os.path.join(ROOT, "tests", "golden", pat)))
        def run(n):
            r = subprocess.run([sys.executable, __file__, "--one", n], capture_output=True, text=True)
            return r.stdout.strip()
        with ThreadPoolExecutor(workers) as ex:
            for line in ex.map(run, names):
                print(line, flush=True)
