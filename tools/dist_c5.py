#!/usr/bin/env python3
"""BASELINE configs[4]: ONE block-tridiagonal multistage QP, stage-partitioned over the ranks of a process group.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/dist_c5.py [--stages 25000]

One rank per GPU over RCCL ("nccl"); when the node has fewer GPUs than ranks (the 1-GPU test boxes) the ranks share cuda:0 and
the collectives are host-staged through "gloo" -- same library code path, same partition, only the transport differs.
Checks on every rank that the partitioned factor + solve reproduces the single-GPU result of the same backend bit for bit,
times 1 update_scalings_and_factor + 2 KKTSystem::solve per step, optionally runs the full interior-point solve, and rank 0
prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stages", type=int, default=25000)
    ap.add_argument("--nx", type=int, default=12)
    ap.add_argument("--nu", type=int, default=8)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--backend", default="multistage", choices=["multistage", "ldlt_cond", "ldlt"])
    ap.add_argument("--problem", default="chain", choices=["chain", "c3", "cont"],
                    help="c3: BASELINE configs[2], a general sparse QP; cont: the frozen Maros-Meszaros CONT-101 (a PDE grid: big fronts on the batched dense kernels, "
                         "panel fronts, wide-front substitution -- in the owned and in the shared part of the tree)")
    ap.add_argument("--transport", default="auto", choices=["auto", "callback", "native"],
                    help="callback: torch.distributed's RCCL collectives on the registered device buffers (default); native: the library's own communicator")
    ap.add_argument("--full-solve", action="store_true")
    ap.add_argument("--refine", action="store_true", help="iterative refinement on in every KKTSystem::solve / the whole solve (exercises the sharded refinement residual, SURVEY 8(e) row 2)")
    ap.add_argument("--no-reference", action="store_true", help="skip the unpartitioned run on every rank (bench mode)")
    ap.add_argument("--wait-stdin", action="store_true", help="block on stdin before touching the GPU (spawned by bench.py, piqp_amd.dist.spawn_waiting)")
    args = ap.parse_args()
    if args.wait_stdin:
        sys.stdin.readline()

    os.environ["PIQP_AMD_MULTISTAGE"] = "tree"  # the engine choice must not depend on a per-rank timing probe
    import torch
    import torch.distributed as dist
    from piqp_amd import dist as pd
    rank, world, dev_index = pd.init()
    shared_gpu = world > 1 and dist.get_backend() != "nccl"
    import piqp_amd as hip
    from qp_gen import c3_problem, mpc_chain, random_vars

    if args.problem == "cont":
        from qp_io import load_qp
        q = load_qp("mm_CONT-101")
        a = (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
    else:
        a = mpc_chain(args.nx, args.nu, args.stages, 5) if args.problem == "chain" else c3_problem()
    d = hip.SparseData(*a)
    n, p, m = d.n, d.p, d.m
    ks = {"multistage": hip.SPARSE_MULTISTAGE, "ldlt_cond": 4, "ldlt": hip.SPARSE_LDLT}[args.backend]
    rng = np.random.default_rng(0)
    dev = torch.device("cuda", dev_index)
    to_dev = lambda v: {k: torch.from_numpy(np.ascontiguousarray(x)).to(dev) for k, x in v.items()}  # noqa: E731
    state = to_dev(random_vars(n, p, m, rng, positive=True))
    rhs = [to_dev(random_vars(n, p, m, rng)) for _ in range(2)]
    out = {"workload": (f"multistage chain n_x={args.nx} n_u={args.nu} stages={args.stages}" if args.problem == "chain" else ("C3 sparse QP" if args.problem == "c3" else "Maros-Meszaros CONT-101")) + f": n={n} p={p} m={m}", "backend": args.backend, "world": world,
           "transport": "gloo (host-staged, ranks share one GPU)" if shared_gpu else ("rccl" if (dist.is_available() and dist.is_initialized()) else "none")}

    refine = bool(args.refine)

    def run_steps(k, steps):
        for i in range(steps):
            assert k.update_scalings_and_factor(refine, 1e-6, 1e-4, state)
            k.solve(rhs[0]); k.solve(rhs[1])
        k.synchronize()

    ref = None
    if not args.no_reference:
        k0 = hip.KKTSystem(d, hip.default_settings(kkt_solver=ks), device=dev_index)
        assert k0.update_scalings_and_factor(refine, 1e-6, 1e-4, state)
        _, ref = k0.solve(rhs[0])
        out["single_gpu_refine_steps"] = k0.last_solve_stats()
        ref = {k: v.clone() for k, v in ref.items()}
        run_steps(k0, args.warmup)
        t0 = time.perf_counter(); run_steps(k0, args.steps); t_single = (time.perf_counter() - t0) / args.steps
        out["single_gpu_ms_per_step"] = t_single * 1e3
        # one evaluation of the refinement residual (kkt_system.hpp:507-536: five mat-vecs, three error kernels, the norm) on every row, as a single GPU does it
        k0.condensed_residual(); k0.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            k0.condensed_residual()
        k0.synchronize()
        out["residual_eval_ms_single_gpu"] = (time.perf_counter() - t0) / 20 * 1e3
        if args.backend != "ldlt":  # all source entries of the value assembly: what a partition of one rank selects
            out["sharded_assembly_entries_single_gpu"] = int(pd.StagePartition(k0, rank=0, world=1, native=False).sharded_calls()[1])
        del k0

    k1 = hip.KKTSystem(d, hip.default_settings(kkt_solver=ks), device=dev_index)
    sp = pd.StagePartition(k1, native=(args.transport == "native") if args.transport != "auto" else None)
    info = sp.info()
    assert k1.update_scalings_and_factor(refine, 1e-6, 1e-4, state)
    _, got = k1.solve(rhs[0])
    out["refine_steps"] = k1.last_solve_stats()
    if sp.error is not None:
        raise sp.error
    if ref is not None:
        same = all(torch.equal(got[k], ref[k]) for k in ("x", "y", "z_l", "z_u", "z_bl", "z_bu", "s_l", "s_u", "s_bl", "s_bu"))
        err = max(float((got[k] - ref[k]).abs().max()) if got[k].numel() else 0.0 for k in ref)
        out["bitwise_equal_to_single_gpu"] = bool(same); out["max_abs_diff"] = err
        if not same and os.environ.get("PIQP_AMD_DIST_DIAG"):
            for k in ref:
                if got[k].numel() and not torch.equal(got[k], ref[k]):
                    dd = (got[k] - ref[k]).abs()
                    idx = torch.nonzero(dd > 0).flatten()
                    print(f"[rank {rank}] {k}: {idx.numel()} of {dd.numel()} entries differ, first {idx[:6].tolist()} last {idx[-3:].tolist()} max {float(dd.max()):.3e}; span {info['span']}", file=sys.stderr, flush=True)
    res, nrm = k1.condensed_residual()
    out["rel_kkt_residual"] = res / nrm
    run_steps(k1, args.warmup)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); run_steps(k1, args.steps); el = time.perf_counter() - t0
    el = pd.max_over_ranks(el)
    out["ms_per_step"] = el / args.steps * 1e3
    out["steps_per_s"] = args.steps / el
    # ... and on this rank's rows only + the all-reduce(max) of the norm (KKT_FULL backends, otherwise the same as above)
    k1.condensed_residual(); k1.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(20):
        k1.condensed_residual()
    k1.synchronize()
    out["residual_eval_ms_partitioned"] = pd.max_over_ranks((time.perf_counter() - t0) / 20) * 1e3
    out["exchange_calls"] = sp.exchange_calls()
    sc = sp.sharded_calls()
    shr = pd.gather_stats([[float(sc[0]), float(sc[1])]])
    out["sharded_residual"] = {"evaluations_per_rank": [int(r[0]) for r in shr], "rows_per_rank": [int(r[1]) for r in shr], "rows_total": n + p + m,
                               "norm_all_reduces": int(sp.calls[3]) if not sp.native else None}
    if args.backend != "ldlt":
        # condensed modes (sparse_ldlt_cond, the multistage tree engine): sharded_calls = (value assemblies done on the rank's own fronts only, source entries it evaluates)
        out["sharded_assembly"] = {"assemblies_per_rank": [int(r[0]) for r in shr], "entries_per_rank": [int(r[1]) for r in shr]}
        del out["sharded_residual"]
    # the solve side (round 5): sharded residual evaluations, partial backend solves (fold / recovery on this rank's rows), gathers of the eliminated multipliers
    ss = sp.sharded_solve_calls()
    ssr = pd.gather_stats([[float(v) for v in ss]])
    out["sharded_solve"] = {"residual_evaluations_per_rank": [int(r[0]) for r in ssr], "residual_rows_per_rank": [int(r[1]) for r in ssr], "rows_total": n + p + m,
                            "partial_backend_solves_per_rank": [int(r[2]) for r in ssr], "multiplier_gathers_per_rank": [int(r[3]) for r in ssr],
                            "x_rows_folded_per_rank": [int(r[4]) for r in ssr], "constraint_rows_recovered_per_rank": [int(r[5]) for r in ssr], "x_rows_total": n,
                            "constraint_rows_total": p + m, "norm_all_reduces": int(sp.calls[3]) if not sp.native else None}
    out["native_rccl"] = bool(sp.native)
    # self-proving multi-GPU record: who ran the collectives and what they saw, per rank
    ci = sp.comm_info()
    out["collective_backend"] = ("rccl (library's own communicator, ncclAllReduce / ncclAllGather on the handle's stream)" if ci["transport"] == "native"
                                 else f"torch.distributed '{ci['process_group_backend']}' through the exchange callback" if ci["transport"] == "callback" else "none")
    crow = pd.gather_stats([[float(ci["process_group_size"]), float(ci["library_comm_count"]), float(ci["library_comm_rank"]), float(ci["library_comm_device"]), float(ci["device"]),
                             float(torch.cuda.device_count())]])
    out["ranks_seen"] = int(crow[0][1]) if ci["transport"] == "native" else int(crow[0][0])
    out["ranks_seen_source"] = "ncclCommCount of the library's communicator" if ci["transport"] == "native" else "torch.distributed.get_world_size"
    out["per_rank"] = [dict(rank=r, device=int(x[4]), visible_devices=int(x[5]), library_comm_rank=int(x[2]), library_comm_device=int(x[3])) for r, x in enumerate(crow)]
    out["exchange_bytes"] = dict(factor_all_reduce=ci["exchange_bytes"][0], forward_all_reduce=ci["exchange_bytes"][1], solution_all_gather=ci["exchange_bytes"][2])
    rows = pd.gather_stats([[float(info["owned_supernodes"]), float(info["shared_supernodes"]), float(info["boundary_roots"]), float(info["span"][0]), float(info["span"][1]),
                             float(info["work_permille"]), float(info["shared_work_permille"])]])
    out["partition"] = [dict(rank=r, owned_supernodes=int(x[0]), span=[int(x[3]), int(x[4])], work_permille=int(x[5])) for r, x in enumerate(rows)]
    out["shared_supernodes"] = int(rows[0][1]); out["boundary_roots"] = int(rows[0][2]); out["shared_work_permille"] = int(rows[0][6])
    out["exchange_doubles"] = info["exchange_doubles"]
    if ref is not None:
        flags = pd.gather_stats([[1.0 if out["bitwise_equal_to_single_gpu"] else 0.0, out["rel_kkt_residual"]]])
        out["bitwise_equal_all_ranks"] = all(f[0] == 1.0 for f in flags)
        out["rel_kkt_residual"] = max(f[1] for f in flags)

    if args.full_solve:
        s0 = hip.SparseSolver(device=dev_index)
        s0.settings.kkt_solver = ks
        s0.settings.iterative_refinement_always_enabled = bool(refine)
        assert s0.setup(*a)
        sp2 = pd.StagePartition(s0, native=(args.transport == "native") if args.transport != "auto" else None)
        t0 = time.perf_counter(); st = s0.solve(); tsol = time.perf_counter() - t0
        if sp2.error is not None:
            raise sp2.error
        x = s0.result()["x"]
        out["full_solve"] = {"status": int(st), "iter": int(s0.info.iter), "solve_ms": pd.max_over_ranks(tsol) * 1e3, "kkt_factor_ms": s0.info.kkt_factor_time * 1e3,
                             "kkt_solve_ms": s0.info.kkt_solve_time * 1e3, "primal_obj": float(s0.info.primal_obj)}
        xs = pd.gather_stats([[float(np.sum(x)), float(np.abs(x).max()), float(s0.info.iter)]])
        out["full_solve"]["identical_on_all_ranks"] = all(r == xs[0] for r in xs)
        if not args.no_reference:
            s1 = hip.SparseSolver(device=dev_index); s1.settings.kkt_solver = ks; s1.settings.iterative_refinement_always_enabled = bool(refine)
            assert s1.setup(*a)
            t0 = time.perf_counter(); st1 = s1.solve(); t1 = time.perf_counter() - t0
            out["full_solve"]["single_gpu"] = {"status": int(st1), "iter": int(s1.info.iter), "solve_ms": t1 * 1e3}
            out["full_solve"]["x_equal_to_single_gpu"] = bool(np.array_equal(x, s1.result()["x"]))
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps(out))
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
