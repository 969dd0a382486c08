#!/usr/bin/env python3
"""What a stage-partitioned run would leave REPLICATED on every rank, measured on one GPU (sharded assembly / mat-vec / residual):
BASELINE configs[4], one block-tridiagonal QP n = 500 012 (25 000 stages of n_x = 12, n_u = 8), sparse_multistage on the tree engine, KKT steps of
1 update_scalings_and_factor + 2 KKTSystem::solve, without and with the iterative-refinement loop (which adds the mat-vecs of the residual).
Run under rocprofv3 --kernel-trace --stats; tools/rocprof_summary.py + the classification below turn the kernel table into two sums:
the assembly-tree work that pq_kkt_partition divides over the ranks, and everything else (value refresh, vector kernels, mat-vecs), which every rank repeats.
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c5rep -- python3 tools/prof_replicated_c5.py [--refine] [--steps 20]
  python tools/prof_replicated_c5.py --classify gpurun_out/prof_c5rep"""
import argparse
import glob
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

TREE = ("k_subtree_", "k_top_", "k_front_", "k_big_", "k_potrf_diag_fronts", "k_trsm_panel_fronts", "k_syrk_lower_fronts", "k_level_", "k_chain_")


def classify(path, steps):
    dbs = glob.glob(path + "/**/*_results.db", recursive=True)
    tot = {"tree": 0.0, "replicated": 0.0}
    rows_out = []
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        for name, calls, total, avg, pct in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
            kind = "tree" if any(t in name for t in TREE) else "replicated"
            tot[kind] += total
            rows_out.append((total, kind, name[:90], calls, avg))
    rows_out.sort(reverse=True)
    print(f"{'ms/step':>9s} {'class':10s} {'calls/step':>10s} {'avg ms':>8s}  kernel   (rocprofv3 top_kernels durations, microseconds / 1000)")
    for total, kind, name, calls, avg in rows_out[:40]:
        print(f"{total / 1e3 / steps:9.2f} {kind:10s} {calls / steps:10.1f} {avg / 1e3:8.2f}  {name}")
    print(f"\nper KKT step ({steps} steps): assembly-tree kernels (divided by pq_kkt_partition) {tot['tree'] / 1e3 / steps:.2f} ms, "
          f"everything else (replicated on every rank) {tot['replicated'] / 1e3 / steps:.2f} ms")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--refine", action="store_true")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--stages", type=int, default=25000)
    ap.add_argument("--classify", default=None)
    args = ap.parse_args()
    if args.classify:
        return classify(args.classify, args.steps)
    os.environ["PIQP_AMD_MULTISTAGE"] = "tree"
    import numpy as np
    import torch
    import piqp_amd as hip
    from qp_gen import mpc_chain, random_vars
    a = mpc_chain(12, 8, args.stages, 5)
    d = hip.SparseData(*a)
    n, p, m = d.n, d.p, d.m
    rng = np.random.default_rng(0)
    dev = torch.device("cuda", 0)
    to_dev = lambda v: {k: torch.from_numpy(np.ascontiguousarray(x)).to(dev) for k, x in v.items()}  # noqa: E731
    state = to_dev(random_vars(n, p, m, rng, positive=True))
    rhs = [to_dev(random_vars(n, p, m, rng)) for _ in range(2)]
    k = hip.KKTSystem(d, hip.default_settings(kkt_solver=hip.SPARSE_MULTISTAGE))
    for _ in range(args.steps):
        assert k.update_scalings_and_factor(args.refine, 1e-6, 1e-4, state)
        k.solve(rhs[0]); k.solve(rhs[1])
    k.synchronize()
    print("refine steps of the last solve:", k.last_solve_stats())


if __name__ == "__main__":
    main()
