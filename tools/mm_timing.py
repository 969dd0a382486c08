#!/usr/bin/env python3
"""Per-problem time of a whole interior-point solve over the frozen Maros-Meszaros / netlib fixtures, device (sparse_ldlt) next to the oracle on one
host core, sorted by the device's time per iteration: a way to find structures the device schedule handles badly (that is how the unbounded
fan-in of BOYD1's assembly tree was found).   python tools/mm_timing.py [prefix ...] > gpurun_out/mm_timing.txt"""
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch  # noqa: F401
    import piqp_amd as hip
    from oracle import pyorc as orc
    from qp_io import GOLDEN, load_qp
    prefixes = sys.argv[1:] or ["mm_", "nl_"]
    names = sorted(os.path.basename(f)[:-4] for pre in prefixes for f in glob.glob(os.path.join(GOLDEN, pre + "*.npz")))
    rows = []
    for name in names:
        q = load_qp(name)
        a = (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
        sh = hip.SparseSolver(); sh.settings.kkt_solver = hip.SPARSE_LDLT
        so = orc.Solver(); so.settings.kkt_solver = orc.SPARSE_LDLT
        t0 = time.perf_counter(); sh.setup(*a); t_set = time.perf_counter() - t0
        t0 = time.perf_counter(); st_h = sh.solve(); t_h = time.perf_counter() - t0
        so.setup(*a, sparse=True)
        t0 = time.perf_counter(); st_o = so.solve(); t_o = time.perf_counter() - t0
        st = sh._kkt_stats() if hasattr(sh, "_kkt_stats") else {}
        rows.append((t_h / max(sh.info.iter, 1), name, q["P"].shape[0], t_set, t_h, sh.info.iter, st_h, t_o, so.info.iter, st_o))
        print(f"{name:16s} n={q['P'].shape[0]:6d} setup {t_set*1e3:8.1f} ms  device {t_h*1e3:9.1f} ms / {sh.info.iter:3d} it (status {st_h:2d})   oracle {t_o*1e3:9.1f} ms / {so.info.iter:3d} it"
              f"   device/oracle {t_h / max(t_o, 1e-9):7.2f}", flush=True)
    rows.sort(reverse=True)
    print("\nslowest per iteration on the device:")
    for r in rows[:15]:
        print(f"  {r[1]:16s} n={r[2]:6d}  {r[0]*1e3:8.2f} ms/it  (oracle {r[7] / max(r[8], 1) * 1e3:8.2f} ms/it)")


if __name__ == "__main__":
    main()
