#!/usr/bin/env python3
"""bench.py -- KKT factor+solve throughput of the dense hot path on MI355X (BASELINE.json configs[1]).

A "step" is the KKT work of ONE interior-point iteration exactly as the reference's own timers count it
(include/piqp/solver.hpp:683-714,728-737,761-769): one KKTSystem::update_scalings_and_factor
(scalings + KKT assembly + factorisation) followed by the predictor and the corrector
KKTSystem::solve.  Inputs (problem matrices, the interior (s,z) state, the right-hand sides) are resident
in HBM before the timed region starts; calls go through the C-ABI in PQ_MEM_DEVICE pointer mode.

  python bench.py --gpus N --steps K --warmup W        (N>1: under torch.distributed.run, or alone -- it then starts the N ranks itself)

Multi-GPU: the dense single-QP path does not shard (SURVEY.md 8e "replicas only"); with N ranks every
rank factors+solves its own independent QP instance of the same shape (weak scaling, no data-path
collective; the barrier and the max-over-ranks reduction are the only communication).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PEAK_FP64_MFMA_TFLOPS = 78.6  # MI355X fp64 matrix peak (vendor sheet; measured value reported alongside)
PEAK_HBM_GBS = 8000.0


def self_launch(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks ourselves (one per GPU, RCCL) as a child process
    BEFORE anything here touches the GPU -- a process that has initialised HIP must never exec -- and leave with its exit code."""
    if args.gpus <= 1 or "RANK" in os.environ or int(os.environ.get("WORLD_SIZE", "1")) > 1:
        return
    import subprocess
    port = 29500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.run(cmd, env=env).returncode)


def panel_update_flops(n, nb=128):
    """algorithmic flops of the fused trailing-update launches of one factorisation (sum_k rs_k (rs_k + 1) nb, the
    "panel update" of the north star) plus what the same launches do besides: the next diagonal block (nb^3 / 3) and the substitution of the
    next panel below it (rows x nb^2)"""
    tot, launches = 0.0, 0
    k = 0
    while k + nb < n:
        rs = n - k - nb
        nbn = min(nb, rs)
        tot += float(rs) * (rs + 1) * nb + nbn ** 3 / 3.0 + float(max(0, rs - nbn)) * nbn * nbn
        launches += 1
        k += nb
    return tot, launches


def pmc_traffic(kernel_key, n, m, p):
    """HBM traffic of one launch of `kernel_key` from the committed rocprofv3 PMC passes of this workload (FETCH_SIZE / WRITE_SIZE in
    separate passes, gfx950 correction applied, see the file); None when the file does not cover this shape.  Not measured in this run:
    the source file and its commit are reported next to the number."""
    for fname in ("r06_pmc_dense_c2.json", "r05_pmc_dense_c2.json", "r04_pmc_dense_c2.json", "r03_pmc_dense_c2.json", "r02_pmc_dense_c2.json"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", fname)))
            if (pmc["n"], pmc["m"], pmc["p"]) != (n, m, p):
                continue
            return pmc["per_launch"][kernel_key]["traffic_bytes"], f"profiles/{fname} ({pmc.get('collected', 'rocprofv3 --pmc passes of tools/prof_dense.py')})"
        except Exception:  # noqa: BLE001
            continue
    return None, None


def dense_leg(piqp_amd, pd, torch, np, q, n, p, m, kkt_solver, refine, steps, warmup, rank, world, local_rank, dev, kernel_pass=0):
    """times `steps` KKT steps (1 update_scalings_and_factor + 2 KKTSystem::solve) with every input resident in HBM; returns the raw figures"""
    from qp_gen import random_vars
    ksys = piqp_amd.KKTSystem(piqp_amd.Data(**q), piqp_amd.default_settings(kkt_solver=kkt_solver), device=local_rank)
    backend = ksys.backend()
    rng = np.random.default_rng(1000 + rank)
    # two interior IPM states and rhs sets, alternated so no step re-reads its predecessor's vectors
    states = [{k: torch.from_numpy(v).to(dev) for k, v in random_vars(n, p, m, rng, positive=True).items()} for _ in range(2)]
    rhss = [{k: torch.from_numpy(v).to(dev) for k, v in random_vars(n, p, m, rng).items()} for _ in range(4)]
    lhs = {k: torch.zeros_like(v) for k, v in rhss[0].items()}
    rho, delta = 1e-6, 1e-4

    def step(i):
        ok = ksys.update_scalings_and_factor(refine, rho, delta, states[i & 1])
        ok1, _ = ksys.solve(rhss[(2 * i) & 3], lhs)       # predictor
        ok2, _ = ksys.solve(rhss[(2 * i + 1) & 3], lhs)   # corrector
        return ok and ok1 and ok2

    def barrier():
        pd.barrier()
        ksys.synchronize()
        torch.cuda.synchronize()

    for i in range(warmup):
        assert step(i), "factorisation failed in warmup"
    res, nrm = ksys.condensed_residual()  # parity gate: relative KKT residual of the last solve
    rel_res = res / nrm
    assert rel_res <= 1e-10, f"KKT residual {rel_res:.3e} above 1e-10"
    backend.set_profiling(1)
    # Python's cyclic collector stays out of the timed region: a generation-2 collection of a process that has torch imported takes 35-75 ms and lands inside
    # whichever call is running when the allocation counter trips -- that, not the library or the device, was every multi-millisecond step of the round-5 size
    # sweep (tools/soak_stalls.py, profiles/r06_soak_dense.txt: 2000 steps without one with the collector off; with it on, the whole excess sits in ONE host
    # call and the stream synchronisation behind it takes its usual microsecond)
    import gc
    gc.collect()
    gc_was = gc.isenabled()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        ok = step(i)
    barrier()
    t1 = time.perf_counter()
    if gc_was:
        gc.enable()
    assert ok
    backend.set_profiling(0)
    elapsed = pd.max_over_ranks(t1 - t0, device=dev if world > 1 else None)
    prof = [backend.get_profile(s) for s in range(3)]
    out = {"elapsed": elapsed, "rel_res": rel_res, "asm_ms": prof[0][0] / max(prof[0][1], 1), "fac_ms": prof[1][0] / max(prof[1][1], 1),
           "sol_ms": prof[2][0] / max(prof[2][1], 1), "backend_solves_per_step": prof[2][1] / steps, "refine": ksys.last_solve_stats()}
    if kernel_pass > 0:
        # kernel-level brackets (hipEvents around every fused-update / panel-solve / sweep launch on the backend's stream) in a SEPARATE
        # pass of the same steps right after the timed region: 62 extra event markers between dependent launches would cost the timed
        # region ~0.1 ms per step
        backend.set_profiling(2)
        for i in range(kernel_pass):
            step(i)
        ksys.synchronize()
        backend.set_profiling(0)
        kp = [backend.get_profile(s) for s in range(6)]
        out["kernels"] = {"fused_ms_per_step": kp[3][0] / kernel_pass, "fused_launches_per_step": kp[3][1] / kernel_pass,
                          "trsm_ms_per_step": kp[4][0] / kernel_pass, "trsm_launches_per_step": kp[4][1] / kernel_pass,
                          "sweeps_ms_per_solve": kp[5][0] / max(kp[5][1], 1), "assembly_ms_per_step": kp[0][0] / max(kp[0][1], 1)}
    del ksys
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--p", type=int, default=0)
    ap.add_argument("--kkt-solver", type=int, default=0, help="0 = dense_cholesky (reference default), 16 = pivot-free LDLt")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch-total", type=int, default=8192, help="BASELINE configs[3]: number of MPC QPs in the batched leg (0 = skip)")
    ap.add_argument("--no-sparse-legs", action="store_true", help="skip the sparse C3 / C5-size KKT legs (BASELINE configs[2], configs[4])")
    ap.add_argument("--no-extra-dense-legs", action="store_true", help="skip the dense_ldlt_no_pivot and refinement-on legs of configs[1]")
    ap.add_argument("--cpu-steps", type=int, default=0, help="0 = auto (about 10-30 s of CPU work)")
    ap.add_argument("--no-size-sweep", action="store_true", help="skip the dense size sweep n = 64 .. 4096 (device vs CPU oracle, the crossover)")
    ap.add_argument("--no-dist-c5", action="store_true", help="N > 1 only: skip the stage-partitioned single-QP leg (BASELINE configs[4])")
    ap.add_argument("--cpu-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_child:
        return cpu_child(args)
    self_launch(args)

    # The dense CPU-baseline legs (the oracle on this box's host cores: the configs[1] baseline and the CPU half of the size sweep) run FIRST, in a child
    # process with OpenMP's default wait policy, on an otherwise idle machine: this process, which times the device, then sets OMP_WAIT_POLICY=passive for
    # itself so that the oracle threads of the later single-thread sparse checks never spin behind a device leg (round-4 advice: the passive policy and the cold
    # calibration call biased the CPU numbers downwards).
    cpu_dense = None
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 and args.gpus <= 1 and not args.no_cpu_baseline:
        import subprocess
        env = {k: v for k, v in os.environ.items() if k != "OMP_WAIT_POLICY"}
        try:
            cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-child"] + sys.argv[1:], env=env, capture_output=True, text=True, timeout=600)
            lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
            cpu_dense = json.loads(lines[-1]) if cp.returncode == 0 and lines else {"error": f"cpu child rc {cp.returncode}: {cp.stderr[-300:]}"}
        except Exception as e:  # noqa: BLE001
            cpu_dense = {"error": f"{type(e).__name__}: {e}"}
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")

    # BASELINE configs[4] at N > 1: ONE n = 500k multistage QP, stage-partitioned over the ranks (tools/dist_c5.py).  It runs in child
    # processes with their own process group so that nothing in there can cost this run its JSON line; the children are started here,
    # before this process touches the GPU, and wait on stdin until the other legs are done.
    c5_child = c5_native_child = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not args.no_dist_c5 and not args.no_sparse_legs:
        import importlib.util
        spec = importlib.util.spec_from_file_location("_pq_dist_spawn", os.path.join(ROOT, "piqp_amd", "dist.py"))
        spawn_mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(spawn_mod)
        c5_child = spawn_mod.spawn_waiting([os.path.join(ROOT, "tools", "dist_c5.py"), "--wait-stdin", "--steps", "10", "--full-solve", "--transport", "callback"])
        # the same leg once more with the library's own RCCL communicator (pq_kkt_set_comm_rccl): never run on more than one rank so far, hence a
        # second, separate group of children that can hang or fail without touching the first result
        c5_native_child = spawn_mod.spawn_waiting([os.path.join(ROOT, "tools", "dist_c5.py"), "--wait-stdin", "--steps", "10", "--transport", "native"], port_offset=61)

    import numpy as np
    import torch

    import piqp_amd
    from piqp_amd import dist as pd

    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    rank, world, local_rank = pd.init()  # RCCL ("nccl") process group when WORLD_SIZE > 1; local_rank = device index of this rank
    if world > 1:
        args.no_cpu_baseline = True  # the CPU baseline is a rank-0, N = 1 figure (torch.distributed.run also pins OMP_NUM_THREADS=1)
    dev = torch.device("cuda", local_rank)
    from qp_gen import dense_strongly_convex_qp

    n, p, m = args.n, args.p, args.m
    # synthetic QP of the BASELINE shape; one independent instance per rank (seed 43 + rank)
    # (exact_shift=False: the diagonal shift that makes P positive definite comes from the semicircle law instead of an eigensolver run --
    # round 1 ran torch.linalg.eigvalsh here and its rocSOLVER kernels polluted the rocprofv3 summaries of this command)
    q = dense_strongly_convex_qp(n, p, m, seed=43 + rank, double_sided=True, exact_shift=False)
    solver_name = {0: "dense_cholesky", 16: "dense_ldlt_no_pivot"}
    main_leg = dense_leg(piqp_amd, pd, torch, np, q, n, p, m, args.kkt_solver, False, args.steps, args.warmup, rank, world, local_rank, dev, kernel_pass=5)
    extra = {}
    if not args.no_extra_dense_legs:
        other = 16 if args.kkt_solver == 0 else 0
        for key, ks, refine in ((solver_name[other], other, False), (solver_name[args.kkt_solver] + "+iterative_refinement", args.kkt_solver, True)):
            extra[key] = dense_leg(piqp_amd, pd, torch, np, q, n, p, m, ks, refine, max(5, args.steps // 2), 2, rank, world, local_rank, dev, kernel_pass=3)

    # who is in this run, as the process group itself reports it (every rank contributes its device): lets the reader of an N > 1 line check that
    # N ranks on N distinct GPUs took part
    ranks_rows = pd.gather_stats([[float(rank), float(local_rank), float(torch.cuda.device_count())]], device=dev if world > 1 else None)
    if rank == 0:
        elapsed = main_leg["elapsed"]
        ms_per_step = elapsed / args.steps * 1e3
        value = world * args.steps / elapsed
        kk = main_leg["kernels"]
        flops_asm = float(n) * (n + 1) * m                    # k_syrk_lower<EPI_ASSEMBLE>, SURVEY.md 8d (C2)
        flops_upd, upd_launches = panel_update_flops(n)       # all fused trailing-update launches of one factorisation
        flops_llt = n ** 3 / 3.0
        asm_s = main_leg["asm_ms"] * 1e-3
        upd_s = kk["fused_ms_per_step"] * 1e-3

        def roof(kernel, flops_per_step, secs_per_step, launches_per_step, key):
            traffic, src = pmc_traffic(key, n, m, p)
            ach = flops_per_step / secs_per_step / 1e12 if secs_per_step > 0 else 0.0
            return {"bound": "mfma", "kernel": kernel, "achieved": ach, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP64_MFMA_TFLOPS,
                    "traffic": traffic, "traffic_source": src, "alg_flops_per_launch": flops_per_step / max(launches_per_step, 1),
                    "avg_launch_ms": secs_per_step * 1e3 / max(launches_per_step, 1), "launches_per_step": launches_per_step, "ms_per_step": secs_per_step * 1e3}
        # the assembly stage is two k_syrk_lower<EPI_ASSEMBLE,2,2> launches (main tiles; split-K tail tiles) + k_syrk_tail_reduce (launch_syrk_t in dense_kernels.hip)
        r_asm = roof("k_syrk_lower<EPI_ASSEMBLE> (dense/kkt.hpp:140-160 update_kkt): 2 launches per factorisation (main tiles, split-K tail) + k_syrk_tail_reduce; "
                     "stage hipEvent-bracketed in the timed region, per-launch figures = stage / 2", flops_asm, asm_s, 2, "assembly")
        persistent = kk["fused_launches_per_step"] < 1.5  # round 3: every round of the factorisation after the first diagonal block / panel in ONE launch
        if persistent:
            r_upd = roof("k_chol_persistent = the whole blocked factorisation after its first diagonal block and panel in ONE persistent launch: for every panel the "
                         "trailing (panel) update, the factorisation of the next diagonal block and the substitution of the next panel behind it, as a ticket-ordered "
                         "task list with look-ahead (dense/ldlt_no_pivot.hpp:313-354, Eigen::LLT at dense/kkt.hpp:82); hipEvent-bracketed in a separate pass of "
                         "the same steps", flops_upd, upd_s, 1, "panel_update")
        else:
            r_upd = roof("k_syrk_lower<EPI_SUBTRACT_POTRF> = one launch per panel: trailing (panel) update of the factorisation + factorisation of the next diagonal "
                         "block + substitution of the next panel behind it (dense/ldlt_no_pivot.hpp:313-354, Eigen::LLT at dense/kkt.hpp:82); hipEvent-bracketed per "
                         "launch in a separate pass of the same steps", flops_upd, upd_s, upd_launches, "panel_update")
        # `roofline` is the factorisation launch: the kernel the north star names ("MFMA panel update"), the longest single launch of the step and the one
        # furthest from its bound.  The assembly STAGE (three unequal launches: main tiles, split-K tail, tail reduce -- bracketed as one) takes about as long
        # per step; it is reported beside it with its own fraction, and `longest_stage` says which of the two stages took more of this run's step (round-5
        # advice: the choice must not hide that)
        dominant, secondary = r_upd, r_asm
        longest = {"stage": "assembly" if asm_s > upd_s else "factorisation", "assembly_ms_per_step": asm_s * 1e3, "factorisation_launch_ms_per_step": upd_s * 1e3,
                   "assembly_frac_of_peak": r_asm["frac"], "factorisation_frac_of_peak": r_upd["frac"]}
        out = {
            "metric": "KKT factor+solve/sec (per IPM iter)",
            "value": value,
            "unit": "IPM-iter KKT (1 factor + 2 solves)/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"dense QP n={n} p={p} m_ineq={m} (BASELINE configs[1]), kkt_solver={solver_name[args.kkt_solver]}, "
                                   "1 update_scalings_and_factor + 2 KKTSystem::solve per step, inputs resident in HBM",
                       "n": n, "p": p, "m": m, "parallelism": f"independent QP replicas x{world}"},
            "roofline": dominant,            # the factorisation launch (see above)
            "roofline_secondary": secondary,
            "longest_stage": longest,
            "stages": {"assembly_ms": main_leg["asm_ms"], "factorisation_ms": main_leg["fac_ms"],
                       "factorisation_tflops": flops_llt / (main_leg["fac_ms"] * 1e-3) / 1e12 if main_leg["fac_ms"] > 0 else 0.0,
                       "backend_solve_ms": main_leg["sol_ms"], "panel_update_ms": kk["fused_ms_per_step"], "panel_solve_ms": kk["trsm_ms_per_step"],
                       "triangular_sweeps_ms_per_solve": kk["sweeps_ms_per_solve"],
                       "step_tflops": (flops_asm + flops_llt) / (ms_per_step * 1e-3) / 1e12},
            "parity": {"rel_kkt_residual": main_leg["rel_res"], "tolerance": 1e-10},
            "process_group": {"collective_backend": (torch.distributed.get_backend() if (world > 1 and torch.distributed.is_initialized()) else "none (single process)"),
                              "ranks_seen": (torch.distributed.get_world_size() if (world > 1 and torch.distributed.is_initialized()) else 1),
                              "per_rank": [dict(rank=int(r[0]), device=int(r[1]), visible_devices=int(r[2])) for r in ranks_rows],
                              "data_path_collectives": "none in this leg (independent replicas / shards); barrier + MAX of the elapsed time + final gather only",
                              "note": "no multi-GPU figure was measured by the build itself (1-GPU boxes only): every N > 1 number is the driver's"},
        }
        for key, leg in extra.items():
            st = max(5, args.steps // 2)
            out.setdefault("dense_legs", {})[key] = {
                "value": world * st / leg["elapsed"], "unit": out["unit"], "ms_per_step": leg["elapsed"] / st * 1e3, "assembly_ms": leg["asm_ms"],
                "factorisation_ms": leg["fac_ms"], "backend_solve_ms": leg["sol_ms"], "backend_solves_per_step": leg["backend_solves_per_step"],
                "rel_kkt_residual": leg["rel_res"], "last_solve": leg["refine"], "panel_update_ms": leg["kernels"]["fused_ms_per_step"],
                "panel_solve_ms": leg["kernels"]["trsm_ms_per_step"]}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = (cpu_dense or {}).get("baseline") or {"error": (cpu_dense or {}).get("error", "not run")}
            if out["cpu_baseline"].get("value"):
                out["speedup_vs_cpu_baseline"] = value / world / out["cpu_baseline"]["value"]
    else:
        out = None
    # second half of BASELINE.json's metric ("QP solves/sec at 1/2/4/8 GPUs"): the batched sparse_multistage leg
    # (the secondary legs must never cost the run its JSON line: a failure is reported in place of the leg)
    if args.batch_total > 0:
        try:
            bq = batched_qp(args, rank, world, local_rank, dev, pd)
        except Exception as e:  # noqa: BLE001
            bq = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0:
            out["batched_qp"] = bq
    if world == 1 and not args.no_size_sweep:
        try:
            sw = dense_size_sweep(piqp_amd, pd, torch, np, args, rank, world, local_rank, dev, (cpu_dense or {}).get("sweep"))
        except Exception as e:  # noqa: BLE001
            sw = {"error": f"{type(e).__name__}: {e}"}
        out["dense_size_sweep"] = sw
    if not args.no_sparse_legs:
        try:
            sl = sparse_legs(args, rank, world, local_rank, dev, pd)
        except Exception as e:  # noqa: BLE001
            sl = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0:
            out["sparse_kkt"] = sl
    if world == 1 and not args.no_sparse_legs:
        try:
            sq = small_qp_legs(args)
        except Exception as e:  # noqa: BLE001
            sq = {"error": f"{type(e).__name__}: {e}"}
        out["small_qp"] = sq
    if c5_child is not None:
        pd.barrier()
        rc, c5_out, c5_err = pd.release_and_collect(c5_child, timeout=300)
        if rank == 0:
            leg = {"config": "BASELINE configs[4]: one block-tridiagonal multistage QP, n = 500k, stage-partitioned over the ranks", "scaling": "strong"}
            line = [ln for ln in (c5_out or "").splitlines() if ln.startswith("{")]
            if rc == 0 and line:
                leg.update(json.loads(line[-1]))
                if leg.get("single_gpu_ms_per_step"):
                    leg["speedup_vs_single_gpu"] = leg["single_gpu_ms_per_step"] / leg["ms_per_step"]  # same backend, same data, unpartitioned, timed by every rank first
            else:
                leg["error"] = f"child returncode {rc} (None = timed out after 300 s)"; leg["stderr_tail"] = c5_err
            out["stage_partitioned_c5"] = leg
        pd.barrier()
        rc, n_out, n_err = pd.release_and_collect(c5_native_child, timeout=180)
        if rank == 0:
            line = [ln for ln in (n_out or "").splitlines() if ln.startswith("{")]
            if rc == 0 and line:
                nat = json.loads(line[-1])
                keep = ("ms_per_step", "steps_per_s", "bitwise_equal_all_ranks", "rel_kkt_residual", "collective_backend", "ranks_seen", "ranks_seen_source", "per_rank", "exchange_calls",
                        "exchange_bytes", "single_gpu_ms_per_step")
                out["stage_partitioned_c5"]["native_rccl_transport"] = {k: nat[k] for k in keep if k in nat}
            else:
                out["stage_partitioned_c5"]["native_rccl_transport"] = {"error": f"child returncode {rc} (None = timed out after 180 s)", "stderr_tail": n_err}
        pd.barrier()
    if rank == 0:
        emit(out)
    pd.finalize()


def emit(out, limit=6144):
    """Everything measured goes to bench_details.json (next to this file, and to gpurun_out/ when that exists) and to an earlier stdout line
    prefixed '#details '; the LAST stdout line is the contract line: the required keys + roofline + cpu_baseline + parity + one scalar per
    secondary leg, at most `limit` bytes (round 4's 24 KB line was not parsed by the driver)."""
    full = json.dumps(out)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            if os.path.isdir(d):
                open(os.path.join(d, "bench_details.json"), "w").write(full + "\n")
        except OSError:
            pass
    print("#details " + full, flush=True)

    def short(v, n=200):
        return v if not isinstance(v, str) or len(v) <= n else v[:n - 3] + "..."

    def roof_short(r):
        if not isinstance(r, dict):
            return r
        keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "alg_flops_per_launch", "alg_bytes_per_launch", "avg_launch_ms",
                "launches_per_step", "ms_per_step")
        return {k: (short(r[k], 160) if k in ("kernel", "traffic_source") else r[k]) for k in keep if k in r}

    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data") if k in out}
    line["config"] = {k: short(v, 240) for k, v in out.get("config", {}).items()}
    line["roofline"] = roof_short(out.get("roofline"))
    line["roofline_secondary"] = roof_short(out.get("roofline_secondary"))
    if "longest_stage" in out:
        line["longest_stage"] = out["longest_stage"]
    if "cpu_baseline" in out:
        cb = out["cpu_baseline"]
        line["cpu_baseline"] = {k: short(cb[k], 240) for k in ("value", "unit", "cores", "kind", "sample", "seconds", "factor_gflops") if k in cb}
        if "speedup_vs_cpu_baseline" in out:
            line["speedup_vs_cpu_baseline"] = out["speedup_vs_cpu_baseline"]
    line["parity"] = out.get("parity")
    line["stages"] = out.get("stages")
    pg = out.get("process_group", {})
    line["process_group"] = {"collective_backend": pg.get("collective_backend"), "ranks_seen": pg.get("ranks_seen"),
                             "devices": [r.get("device") for r in pg.get("per_rank", [])]}
    legs = {}
    for k, v in out.get("dense_legs", {}).items():
        legs["dense:" + k] = v.get("value")
    bq = out.get("batched_qp") or {}
    if "error" in bq:
        legs["batched_qp:error"] = short(bq["error"], 120)
    for mode in ("strong", "weak"):
        if isinstance(bq.get(mode), dict):
            legs[f"batched_qp:{mode}:qp_per_s"] = bq[mode].get("qp_per_s")
            legs[f"batched_qp:{mode}:ms"] = bq[mode].get("ms")
    if isinstance(bq.get("roofline"), dict):
        legs["batched_qp:hbm_frac"] = bq["roofline"].get("frac")
    if "predicted_strong_scaling_8gpu" in bq:
        legs["batched_qp:predicted_strong_scaling_8gpu"] = bq["predicted_strong_scaling_8gpu"]
    sk = out.get("sparse_kkt") or {}
    if "error" in sk:
        legs["sparse_kkt:error"] = short(sk["error"], 120)
    for k, v in sk.items():
        if isinstance(v, dict) and "value" in v:
            legs["sparse:" + k] = v["value"]
            if isinstance(v.get("roofline"), dict):
                legs["sparse:" + k + ":frac"] = v["roofline"].get("frac")
                legs["sparse:" + k + ":bound"] = v["roofline"].get("bound")
    c5 = out.get("stage_partitioned_c5")
    if isinstance(c5, dict):
        for k in ("ms_per_step", "speedup_vs_single_gpu", "bitwise_equal_all_ranks", "ranks_seen", "error"):
            if k in c5:
                legs["stage_partitioned_c5:" + k] = short(c5[k], 120)
        nat = c5.get("native_rccl_transport")
        if isinstance(nat, dict):
            for k in ("ms_per_step", "bitwise_equal_all_ranks", "ranks_seen", "error"):
                if k in nat:
                    legs["stage_partitioned_c5:native_rccl:" + k] = short(nat[k], 120)
    sw = out.get("dense_size_sweep")
    if isinstance(sw, dict):
        for k in ("smallest_n_where_device_beats_one_host_thread", "smallest_n_where_device_beats_all_host_threads", "error"):
            if k in sw:
                legs["dense_size_sweep:" + k] = short(sw[k], 120)
    sq = out.get("small_qp")
    if isinstance(sq, dict):
        for k, v in sq.items():
            if isinstance(v, (int, float)):
                legs["small_qp:" + k] = v
    line["legs"] = legs
    line["details"] = "bench_details.json (and the '#details' stdout line before this one)"
    txt = json.dumps(line)
    if len(txt) > limit:  # never lose the line to its size: drop the optional parts, largest first
        for k in ("legs", "roofline_secondary", "stages", "process_group"):
            line.pop(k, None)
            txt = json.dumps(line)
            if len(txt) <= limit:
                break
    print(txt, flush=True)


def sparse_legs(args, rank, world, local_rank, dev, pd):
    """KKT factor + 2 solves per second on the sparse configurations (one independent instance per rank, like the dense leg):
    configs[2] C3 (n=50k, N=100k, sparse_ldlt) and a configs[4]-size chain (n=500k block-tridiagonal, sparse_ldlt with the
    nested-dissection tree = stage-partitioned elimination on ONE GPU).  CPU baseline: the oracle on one core."""
    import numpy as np
    import torch
    import piqp_amd
    from qp_gen import c3_problem, mpc_chain, random_vars
    res = {}
    cases = [("C3", "sparse QP n=50000 p=20000 m=30000, nnz(upper KKT)=4.9e5, kkt_solver=sparse_ldlt (BASELINE configs[2])", c3_problem(seed=44 + rank), piqp_amd.SPARSE_LDLT, 1),
             ("C5_single_gpu", "one block-tridiagonal QP n=500012 p=300000 (25000 stages of n_x=12,n_u=8), kkt_solver=sparse_ldlt, ONE GPU (BASELINE configs[4] size)",
              mpc_chain(12, 8, 25000, 5 + rank), piqp_amd.SPARSE_LDLT, 5)]
    # the real Maros-Meszaros cross-checks SURVEY.md 8d names next to the synthetic C3 (frozen fixtures, tests/golden/make_fixtures.py)
    from qp_io import load_qp
    # round 4: the banded C3 recipe is the benign half of "Maros-Meszaros-style" (every row inside a 40-variable window, fronts <= 92); a wider variant beside it --
    # rows of 10 nonzeros inside 300-variable windows, nnz(upper KKT) = 7.4e5: under AMD nnz(L) = 1.9e7, fronts up to 620 and an assembly tree ~1800 levels deep (the
    # symbolic analysis picks nested dissection for it since round 4: 17 levels of merged fronts, nnz(L) = 4.6e7); and the same rows in 1500-variable windows:
    # nnz(L) = 1.2e8, a chain of ~150 fronts of 3000-4000 rows, 3.6e11 flops per factorisation (no oracle leg: minutes per step on one core)
    cases.append(("C3_wide", "sparse QP n=50000 p=20000 m=30000, rows of 10 nonzeros in 300-variable windows, nnz(upper KKT)=7.4e5, kkt_solver=sparse_ldlt (harder variant of BASELINE configs[2])",
                  c3_problem(seed=44 + rank, spread=300, row_nnz=10), piqp_amd.SPARSE_LDLT, 1))
    cases.append(("C3_window1500", "sparse QP n=50000 p=20000 m=30000, rows of 10 nonzeros in 1500-variable windows, nnz(upper KKT)=7.5e5, kkt_solver=sparse_ldlt (hardest variant of BASELINE configs[2])",
                  c3_problem(seed=44 + rank, spread=1500, row_nnz=10), piqp_amd.SPARSE_LDLT, None))
    for nm, what in (("CONT-201", "PDE-constrained grid, n=40397 p=40198"), ("BOYD1", "n=93261 with 18 dense equality rows")):
        q = load_qp("mm_" + nm)
        cases.append(("MM_" + nm, f"Maros-Meszaros {nm} ({what}), kkt_solver=sparse_ldlt", (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"]),
                      piqp_amd.SPARSE_LDLT, 1))
    for key, desc, a, ks, oracle_ks in cases:
        d = piqp_amd.SparseData(*a)
        n, p, m = d.n, d.p, d.m
        t0 = time.perf_counter()
        k = piqp_amd.KKTSystem(d, piqp_amd.default_settings(kkt_solver=ks), device=local_rank)
        t_setup = time.perf_counter() - t0
        rng = np.random.default_rng(7 + rank)
        state_h = random_vars(n, p, m, rng, positive=True)
        rhs_h = [random_vars(n, p, m, rng) for _ in range(2)]
        # resident in HBM: device tensors, PQ_MEM_DEVICE pointer mode (no PCIe inside the timed region)
        state = {kk: torch.from_numpy(v).to(dev) for kk, v in state_h.items()}
        rhs = [{kk: torch.from_numpy(v).to(dev) for kk, v in r.items()} for r in rhs_h]
        lhs = {kk: torch.zeros_like(v) for kk, v in rhs[0].items()}
        for _ in range(2):
            assert k.update_scalings_and_factor(False, 1e-6, 1e-4, state)
            k.solve(rhs[0], lhs)
        res_inf, nrm = k.condensed_residual()
        steps = 3 if key in ("C3_wide", "C3_window1500") else 10
        be = k.backend(); be.set_profiling(True)
        pd.barrier(); k.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            k.update_scalings_and_factor(False, 1e-6, 1e-4, state)
            k.solve(rhs[0], lhs); k.solve(rhs[1], lhs)
        k.synchronize(); torch.cuda.synchronize(); pd.barrier()
        el = pd.max_over_ranks(time.perf_counter() - t0, device=dev if world > 1 else None)
        be.set_profiling(False)
        prof = [be.get_profile(s) for s in range(3)]
        if rank == 0:
            r = {"workload": desc, "value": world * steps / el, "unit": "IPM-iter KKT (1 factor + 2 solves)/s", "ms_per_step": el / steps * 1e3,
                 "factor_ms": (prof[0][0] + prof[1][0]) / max(prof[1][1], 1), "backend_solve_ms": prof[2][0] / max(prof[2][1], 1), "setup_s": t_setup,
                 "rel_kkt_residual": res_inf / nrm}
            if key == "MM_BOYD1":  # the one leg above 1e-10: the reference algorithm itself does not reach it on this matrix
                r["residual_note"] = ("P spans nine orders of magnitude on its diagonal; tests/test_mm_real_gpu.py::test_kkt_factor_solve_on_real_problem evaluates both "
                                      "solutions in extended precision on the same state: device 2.2e-9, oracle (the reference's up-looking LDLt restated) 1.7e-8")
            # roofline of the two dominant phases against HBM with the ALGORITHMIC bytes of SURVEY.md 8d (C3 row): factor reads PKPt once and
            # writes L once (12 B per entry: value + index) plus D / D_inv / diag (24 N); one solve reads L twice plus six vector passes.
            try:
                stt = be.sparse_stats()
                N_, nK, nL = stt["N"], stt["nnz_K"], stt["nnz_L"]
                bytes_factor = 12.0 * nK + 12.0 * nL + 24.0 * N_
                bytes_solve = 24.0 * nL + 48.0 * N_
                fac_s = r["factor_ms"] * 1e-3; sol_s = r["backend_solve_ms"] * 1e-3
                r["symbolic"] = stt
                traffic_f = traffic_s = None
                try:  # rocprofv3 PMC passes of the C3 and CONT-201 workloads (profiles/r02_pmc_sparse_batch.json, r03_pmc_sparse_cont201.json); other workloads: not measured
                    pmc = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_sparse_batch.json")))["sparse_c3"]
                    if key == "C3" and abs(pmc["factor_per_launch"]["algorithmic_bytes"] - bytes_factor) < 0.05 * bytes_factor:
                        traffic_f = pmc["factor_per_launch"]["traffic_bytes"]; traffic_s = pmc["solve_per_launch"]["traffic_bytes"]
                    f3 = next((f for f in (os.path.join(ROOT, "profiles", nm) for nm in ("r06_pmc_sparse_cont201.json", "r04_pmc_sparse_cont201.json", "r03_pmc_sparse_cont201.json", "r02_pmc_sparse_batch.json")) if os.path.exists(f)))
                    pmc2 = json.load(open(f3)).get("sparse_cont201")  # the newest committed passes of the CONT-201 workload
                    if key == "MM_CONT-201" and pmc2 and abs(pmc2["factor_per_launch"]["algorithmic_bytes"] - bytes_factor) < 0.05 * bytes_factor:
                        traffic_f = pmc2["factor_per_launch"]["traffic_bytes"]; traffic_s = pmc2["solve_per_launch"]["traffic_bytes"]
                    fw = os.path.join(ROOT, "profiles", "r04_pmc_sparse_c3_wide.json")
                    if key == "C3_wide" and os.path.exists(fw):
                        pmc3 = json.load(open(fw))["sparse_c3_wide"]
                        traffic_f = pmc3["factor_per_launch"]["traffic_bytes"]; traffic_s = pmc3["solve_per_launch"]["traffic_bytes"]
                except Exception:  # noqa: BLE001
                    pass
                # the bound of a sparse phase is max(flops / fp64 MFMA peak, algorithmic bytes / HBM peak) -- the wide-front
                # trees are flop-bound, the banded ones byte-bound; `frac` = that bound's time / the measured time
                t_flop = stt["flops_factor"] / (PEAK_FP64_MFMA_TFLOPS * 1e12); t_byte = bytes_factor / (PEAK_HBM_GBS * 1e9)
                fkern = ("multifrontal factorisation (k_subtree_factor_lds / _pk + k_top_factor / k_front_factor levels; big fronts: k_potrf_trsm_fronts + k_syrk_lower_fronts / "
                         "k_syrk_half_fronts per level, k_front_panel_step at the top of the tree), hipEvent-bracketed on the backend stream")
                if t_flop >= t_byte:
                    ach = stt["flops_factor"] / fac_s / 1e12
                    r["roofline"] = {"bound": "mfma", "kernel": fkern, "achieved": ach, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP64_MFMA_TFLOPS,
                                     "traffic": traffic_f, "alg_flops_per_launch": stt["flops_factor"], "alg_bytes_per_launch": bytes_factor, "avg_launch_ms": r["factor_ms"]}
                else:
                    r["roofline"] = {"bound": "hbm", "kernel": fkern, "achieved": bytes_factor / fac_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                     "frac": bytes_factor / fac_s / 1e9 / PEAK_HBM_GBS, "traffic": traffic_f, "alg_flops_per_launch": stt["flops_factor"],
                                     "alg_bytes_per_launch": bytes_factor, "avg_launch_ms": r["factor_ms"],
                                     "note": "dependent-latency bound (tree of small fronts), not bandwidth bound: see DESIGN.md section 6"}
                r["roofline_solve"] = {"bound": "hbm", "kernel": "backend solve (k_subtree_fwd/bwd_wave + k_front_fwd/bwd_wide levels, k_level_fwd/bwd_mixed where a level holds both kinds)", "achieved": bytes_solve / sol_s / 1e9,
                                       "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": bytes_solve / sol_s / 1e9 / PEAK_HBM_GBS, "traffic": traffic_s,
                                       "alg_bytes_per_launch": bytes_solve, "avg_launch_ms": r["backend_solve_ms"]}
                r["factor_gflops"] = stt["flops_factor"] / fac_s / 1e9
            except Exception as e:  # noqa: BLE001
                r["roofline_error"] = str(e)
            if not args.no_cpu_baseline and oracle_ks is not None:
                from oracle import pyorc
                od = pyorc.Data.sparse(*a)
                ko = pyorc.KKTSystem(od, pyorc.Settings(kkt_solver=oracle_ks))
                ko.update_scalings_and_factor(False, 1e-6, 1e-4, state_h)
                t0 = time.perf_counter()
                cs = 1 if key == "C3_wide" else 3
                for _ in range(cs):
                    ko.update_scalings_and_factor(False, 1e-6, 1e-4, state_h); ko.solve(rhs_h[0]); ko.solve(rhs_h[1])
                elc = time.perf_counter() - t0
                r["cpu_baseline"] = {"value": cs / elc, "unit": r["unit"], "cores": 1, "kind": "port",
                                     "sample": f"{cs} steps of the same workload, oracle kkt_solver={'sparse_ldlt' if oracle_ks == 1 else 'sparse_multistage'} (gcc -O3), 1 thread"}
                r["speedup_vs_cpu_core"] = r["value"] / world / (cs / elc)
            res[key] = r
        del k
    return res if rank == 0 else None


def small_qp_legs(args):
    """Whole interior-point solves of small sparse QPs (the reference's own benchmark shapes, benchmarks/src/sqp_benchmarks.cpp, and a degenerate Maros-Meszaros LP-like
    QP): kkt_solver = sparse_ldlt runs the reference-order engine up to 8192 KKT rows (sparse_exact.hip: the solve is the oracle's bit for bit), sparse_ldlt_multifrontal
    forces the supernodal engine; next to the CPU oracle on one core.  Times are solve() wall clock of a warm solver (second solve), host pointers."""
    import numpy as np
    import piqp_amd as hip
    from oracle import pyorc as orc
    from qp_io import load_qp
    res = {}
    for name in ("qp_chain_mass_sqp", "mm_QPILOTNO"):
        q = load_qp(name)
        a = (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
        so = orc.Solver(); so.settings.kkt_solver = orc.SPARSE_LDLT; so.enable_trace(1024)
        so.setup(*a, sparse=True)
        so.solve()
        so.enable_trace(1024)
        t0 = time.perf_counter(); st_o = so.solve(); t_o = time.perf_counter() - t0
        tro = so.trace()
        for tag, ks in (("reference_order", hip.SPARSE_LDLT), ("multifrontal", hip.SPARSE_LDLT_MULTIFRONTAL)):
            sh = hip.SparseSolver(); sh.settings.kkt_solver = ks; sh.enable_trace(1024)
            sh.setup(*a)
            sh.solve()
            sh.enable_trace(1024)
            t0 = time.perf_counter(); st_h = sh.solve(); t_h = time.perf_counter() - t0
            trh = sh.trace()
            same = bool(trh.shape == tro.shape and np.array_equal(trh, tro))
            res[f"{name}:{tag}:solve_ms"] = t_h * 1e3
            res[f"{name}:{tag}:iterations"] = int(sh.info.iter)
            res[f"{name}:{tag}:status_equal_oracle"] = int(st_h == st_o)
            res[f"{name}:{tag}:trace_bitwise_equal_oracle"] = int(same)
        res[f"{name}:oracle_1_core:solve_ms"] = t_o * 1e3
        res[f"{name}:oracle_1_core:iterations"] = int(so.info.iter)
    return res


def batched_qp(args, rank, world, local_rank, dev, pd):
    """BASELINE configs[3]: --batch-total independent MPC QPs (n = 120, 40 stages of n_x = 2, n_u = 1, p = 80, box bounds),
    solved by the batched kernel (one workgroup = one whole interior-point solve).  Instances are independent, so the
    batch is sharded contiguously over the ranks with no data-path collective ("strong": the fixed batch is split;
    "weak": every rank solves a full batch of its own).  Timed region = solve() of all instances, inputs resident."""
    import numpy as np
    import torch
    import piqp_amd
    from qp_gen import mpc_batch, mpc_instance

    def run(mb, reps=3):
        bs = piqp_amd.BatchSparseSolver(device=local_rank)
        assert bs.setup(mb["P_pattern"], mb["P_values"], mb["c"], mb["A_pattern"], mb["A_values"], mb["b"], x_l=mb["x_l"], x_u=mb["x_u"])
        bs.set_start_order(False)  # index order: what a fresh batch gets (every figure of this leg unless it says longest_first)
        bs.solve()  # warm-up
        pd.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            solved = bs.solve()
        torch.cuda.synchronize(); pd.barrier()
        el = (time.perf_counter() - t0) / reps
        return bs, solved, pd.max_over_ranks(el, device=dev if world > 1 else None)

    total = args.batch_total
    full = mpc_batch(total, seed=1000)
    sub = lambda lo, hi: {k: (v[lo:hi] if isinstance(v, np.ndarray) and v.ndim == 2 else v) for k, v in full.items()}
    res = {}
    lo, hi = pd.shard_range(total, rank, world)
    bs, solved, el = run(sub(lo, hi))
    rows = pd.gather_stats([[float(solved), float(bs.iterations().sum()), float(hi - lo)]], device=dev if world > 1 else None)
    if rank == 0:
        tot_solved = sum(r[0] for r in rows); tot_it = sum(r[1] for r in rows)
        res["strong"] = {"qps_total": total, "qp_per_s": total / el, "ms": el * 1e3, "solved": int(tot_solved), "iters_mean": tot_it / total,
                         "kernel_ms_rank0": bs.last_kernel_ms()[0], "threads_per_qp": bs.last_kernel_ms()[1],
                         "start_order": "qp_per_s / ms = the batch started in INDEX order, what the first solve of a fresh batch gets (the headline since round 4); "
                                        "*_longest_first = started longest first by the PREVIOUS solve's iteration counts (the library's default on a re-solve, as a "
                                        "receding-horizon controller re-solves its batch) -- a perfect prediction here, since the timed solves repeat the warm-up's batch"}
    # the same batch started longest first (the library's default when a batch is solved again)
    try:
        bs.set_start_order(True)
        bs.solve()
        pd.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            bs.solve()
        torch.cuda.synchronize(); pd.barrier()
        el_lf = pd.max_over_ranks((time.perf_counter() - t0) / 3, device=dev if world > 1 else None)
        bs.set_start_order(False)
        if rank == 0:
            res["strong"]["ms_longest_first"] = el_lf * 1e3
            res["strong"]["qp_per_s_longest_first"] = total / el_lf
    except Exception as e:  # noqa: BLE001
        if rank == 0:
            res["strong"]["longest_first_error"] = f"{type(e).__name__}: {e}"
    if world > 1:
        bsw, solved_w, el_w = run(full)
        rows = pd.gather_stats([[float(solved_w)]], device=dev)
        if rank == 0:
            res["weak"] = {"qps_total": total * world, "qp_per_s": total * world / el_w, "ms": el_w * 1e3, "solved": int(sum(r[0] for r in rows))}
    # one-GPU batch-size sweep (rank 0, N = 1 runs only): what a fixed batch split over G GPUs can gain is bounded by t(B) / t(B / G) of ONE GPU -- a
    # shard of 8192 / 8 = 1024 QPs is four per CU, i.e. one round of workgroups whose duration is one QP's latency
    if world == 1 and total >= 2048:
        sweep = {}
        for bsz in sorted({total // 8, total // 4, total // 2, total, 2 * total}):
            try:
                mbz = mpc_batch(bsz, seed=1000) if bsz > total else sub(0, bsz)
                _b, _solved, el_z = run(mbz, reps=3)
                sweep[str(bsz)] = {"ms": el_z * 1e3, "qp_per_s": bsz / el_z, "kernel_ms": _b.last_kernel_ms()[0], "solved": int(_solved)}
                del _b
            except Exception as e:  # noqa: BLE001
                sweep[str(bsz)] = {"error": f"{type(e).__name__}: {e}"}
        res["batch_size_sweep_one_gpu"] = sweep
        try:
            t_full, t_8th = sweep[str(total)]["ms"], sweep[str(total // 8)]["ms"]
            res["predicted_strong_scaling_8gpu"] = t_full / t_8th
            res["predicted_strong_scaling_note"] = (f"t({total}) / t({total // 8}) on ONE GPU = upper bound of the 8-GPU speed-up of the FIXED {total}-QP batch (no multi-GPU node was "
                                                    "available to the build); the >= 6x of the north star is reachable as weak scaling (a full batch per GPU, no data-path "
                                                    "collective) or with batches of >= 8 x 8192 QPs")
        except Exception:  # noqa: BLE001
            pass
    if rank != 0:
        return None
    res["workload"] = f"{total} linear-MPC QPs, n=120 (40 stages x (n_x=2,n_u=1)), p=80, box bounds on all variables, kkt_solver=sparse_multistage (BASELINE configs[3])"
    res["unit"] = "QP solves/s (whole interior-point solve, inputs resident in HBM)"
    res["sharding"] = f"contiguous shards of independent QPs over {world} rank(s), no data-path collective"
    pr = bs.profile(0)
    res["in_kernel_us_instance0"] = {k: v * 1e6 for k, v in pr.items()}
    # roofline with the ALGORITHMIC bytes of SURVEY.md 8d (C4 row), per QP and IPM iteration:
    #   8 (sum |D_i| + |B_i|) (1 write + 1 read in the factorisation + 2 reads per backend solve) + 8 nnz(AT, GT blocks) (1 + 2 n_solves)
    #   + 8 (n + p + m) * 12 vector passes, with n_solves = 2 (predictor + corrector, no refinement step needed on this recipe)
    try:
        bi = bs.block_info().astype(np.int64)
        blocks = int((bi[:, 1] * bi[:, 1] + bi[:, 2] * bi[:, 1]).sum())
        nnz_c = int(full["A_pattern"].nnz)
        nvec = int(full["n"]) + int(full["p"])
        n_solves = 2
        bytes_iter = 8.0 * blocks * (2 + 2 * n_solves) + 8.0 * nnz_c * (1 + 2 * n_solves) + 8.0 * nvec * 12
        kernel_s = bs.last_kernel_ms()[0] * 1e-3
        its = float(bs.iterations().sum())
        traffic_b = None
        try:
            if total == 8192 and world == 1:
                for fname in ("r06_pmc_batch_c4.json", "r03_pmc_batch_c4.json", "r02_pmc_sparse_batch.json"):  # the newest committed PMC passes of this kernel
                    if os.path.exists(os.path.join(ROOT, "profiles", fname)):
                        traffic_b = json.load(open(os.path.join(ROOT, "profiles", fname)))["batch_c4"]["per_launch"]["traffic_bytes"]
                        break
        except Exception:  # noqa: BLE001
            pass
        res["roofline"] = {"bound": "hbm", "kernel": "k_batch_ipm (one workgroup = one whole interior-point solve), hipEvent-bracketed",
                           "achieved": bytes_iter * its / kernel_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": bytes_iter * its / kernel_s / 1e9 / PEAK_HBM_GBS,
                           "traffic": traffic_b, "alg_bytes_per_qp_iteration": bytes_iter, "qp_iterations_in_launch": its, "avg_launch_ms": kernel_s * 1e3,
                           "note": "chain fronts and panels stay in LDS / registers; the measured traffic (PMC) is the per-instance vector arena streaming through L2 / "
                                   "Infinity Cache in every vector phase plus the register save areas of the out-of-line calls -- a miss-latency bound, see profiles/r06_pmc_batch_c4.json"}
    except Exception as e:  # noqa: BLE001
        res["roofline_error"] = str(e)
    if not args.no_cpu_baseline:
        # CPU baseline: the oracle (restatement of the reference's SparseSolver + sparse_multistage) on a bounded sample, 1 thread
        from oracle import pyorc
        sample = min(256, total)
        solvers = []
        for i in range(sample):
            so = pyorc.Solver(); so.settings.kkt_solver = pyorc.SPARSE_MULTISTAGE
            so.setup(*mpc_instance(full, i), sparse=True)
            solvers.append(so)
        t0 = time.perf_counter()
        its = [(so.solve(), so.info.iter) for so in solvers]
        el_cpu = time.perf_counter() - t0
        dev_it = bs.iterations()[:sample] if lo == 0 else None
        same = None if dev_it is None else float(np.mean([its[i][1] == dev_it[i] for i in range(min(sample, len(dev_it)))]))
        res["cpu_baseline"] = {"value": sample / el_cpu, "unit": "QP solves/s", "cores": 1, "kind": "port",
                               "sample": f"solve() of the first {sample} instances (setup excluded), oracle built with gcc -O3, 1 thread", "seconds": el_cpu}
        res["iteration_count_parity_on_sample"] = same
        res["speedup_vs_cpu_core"] = res["strong"]["qp_per_s"] / (sample / el_cpu)
        # ... and on ALL host cores: one process per core, each with its own solver objects (how a PIQP user parallelises independent QPs:
        # the reference has no batch API).  Processes, not threads: spawned interpreters that never touch the GPU.
        try:
            import multiprocessing as mp
            cores = len(os.sched_getaffinity(0))
            per = 64
            ctx = mp.get_context("spawn")
            with ctx.Manager() as mgr, ctx.Pool(cores) as pool:
                barrier = mgr.Barrier(cores)
                rows = pool.map(_cpu_batch_worker, [(1000 + 7919 * w, per, barrier) for w in range(cores)], chunksize=1)
            el_all = max(r[0] for r in rows)
            nsolved = sum(r[1] for r in rows)
            res["cpu_baseline_all_cores"] = {"value": cores * per / el_all, "unit": "QP solves/s", "cores": cores, "kind": "port",
                                             "sample": f"{cores} processes x {per} QPs of the same recipe (own seeds), solve() only, slowest process's time; {nsolved} solved",
                                             "seconds": el_all}
            res["speedup_vs_cpu_all_cores"] = res["strong"]["qp_per_s"] / (cores * per / el_all)
        except Exception as e:  # noqa: BLE001
            res["cpu_baseline_all_cores"] = {"error": str(e)}
    return res


def _cpu_batch_worker(job):
    """one host process of the all-cores CPU baseline of the batched leg: `count` MPC QPs of the C4 recipe through the oracle (no GPU, no torch)"""
    seed, count, barrier = job
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    from oracle import pyorc
    from qp_gen import mpc_batch, mpc_instance
    mb = mpc_batch(count, seed=seed)
    solvers = []
    for i in range(count):
        so = pyorc.Solver(); so.settings.kkt_solver = pyorc.SPARSE_MULTISTAGE
        so.setup(*mpc_instance(mb, i), sparse=True)
        solvers.append(so)
    [so.solve() for so in solvers]  # warm
    try:
        barrier.wait(300)  # every process times the same interval: all cores loaded with solves, none still setting up
    except Exception:  # noqa: BLE001  (a broken barrier only makes the figure pessimistic)
        pass
    t0 = time.perf_counter()
    st = [so.solve() for so in solvers]
    return time.perf_counter() - t0, sum(1 for v in st if v == 1)


def dense_strongly_convex_qp_small():
    from qp_gen import dense_strongly_convex_qp
    return dense_strongly_convex_qp(1024, 0, 1024, seed=7, double_sided=True, exact_shift=False)


# (4 .. 1024 x 2 are the reference's own factorisation-benchmark sizes, dense_cholesky_factorization_benchmark.cpp:97-102; `device_factorisation_ms` of a row is the
# factorisation alone, hipEvent-bracketed on the backend's stream)
SWEEP_SIZES = (4, 8, 16, 32, 64, 128, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096)


def dense_size_sweep(piqp_amd, pd, torch, np, args, rank, world, local_rank, dev, cpu_rows):
    """The reference's own factorisation-benchmark sizes (benchmarks/src/dense_cholesky_factorization_benchmark.cpp:97-102: n = 4 .. 1024, x2) carried on to the
    BASELINE size: the same step (1 factor + 2 solves, m = n, p = 0) on the device and on the CPU oracle (one thread, and up to 32 threads; measured by the CPU
    child process, cpu_size_sweep), so that the crossover below which the host wins is a measured number.  n < 384 or n % 128 != 0 runs the launch-per-panel path."""
    from qp_gen import dense_strongly_convex_qp
    rows = {}
    for n in SWEEP_SIZES:
        q = dense_strongly_convex_qp(n, 0, n, seed=900 + n, double_sided=True, exact_shift=False)
        # three runs of 5 steps, the MEDIAN run reported and the slowest beside it (round-5 review: a minimum hides stalls; their cause -- Python's collector inside
        # the timed region -- is gone, see dense_leg)
        legs = sorted((dense_leg(piqp_amd, pd, torch, np, q, n, 0, n, args.kkt_solver, False, 5, 2, rank, world, local_rank, dev, kernel_pass=0) for _ in range(3)), key=lambda g: g["elapsed"])
        leg = legs[1]
        r = {"device_ms_per_step": leg["elapsed"] / 5 * 1e3, "device_ms_per_step_max": legs[2]["elapsed"] / 5 * 1e3, "device_ms_per_step_runs": [g["elapsed"] / 5 * 1e3 for g in legs],
             "device_assembly_ms": leg["asm_ms"], "device_factorisation_ms": leg["fac_ms"], "device_backend_solve_ms": leg["sol_ms"]}
        r.update((cpu_rows or {}).get(str(n), {}))
        # factor-only, as the reference's benchmark measures it (compute() of a positive definite matrix resident in device memory) through the class objects
        # (pq_dense_factor_*): [device time of the factorisation launches, wall time of the whole compute() call], microseconds, median of 15
        try:
            S = q["P"] + q["P"].T - np.diag(np.diag(q["P"])) + float(n) * np.eye(n)
            fo = {}
            for name, cls, uplo in (("LLT_Lower", piqp_amd.LLT, piqp_amd.LOWER), ("LDLTNoPivot_Lower", piqp_amd.LDLTNoPivot, piqp_amd.LOWER), ("LDLTNoPivot_Upper", piqp_amd.LDLTNoPivot, piqp_amd.UPPER)):
                t = torch.from_numpy(np.ascontiguousarray(S)).to(dev)  # (symmetric: either triangle of either storage order is the same matrix)
                f = cls(n, uplo, device=local_rank)
                for _ in range(2):
                    f.compute_colmajor(t)
                assert f.info() == 0
                ms = []
                for _ in range(15):
                    f.compute_colmajor(t)
                    ms.append(f.last_ms())
                fo[name] = [float(np.median([a for a, _ in ms])) * 1e3, float(np.median([b for _, b in ms])) * 1e3]
                del f
            r["factor_only_us"] = fo
        except Exception as e:  # noqa: BLE001
            r["factor_only_error"] = f"{type(e).__name__}: {e}"
        rows[str(n)] = r
    cross1 = [int(n) for n, r in rows.items() if "cpu_1_thread_ms_per_step" in r and r["device_ms_per_step"] < r["cpu_1_thread_ms_per_step"]]
    crossa = [int(n) for n, r in rows.items() if "cpu_all_threads_ms_per_step" in r and r["device_ms_per_step"] < r["cpu_all_threads_ms_per_step"]]
    return {"workload": "dense QP, m = n, p = 0; step = 1 update_scalings_and_factor + 2 KKTSystem::solve, inputs resident (device) / in host memory (oracle)",
            "sizes": rows, "smallest_n_where_device_beats_one_host_thread": min(cross1) if cross1 else None,
            "smallest_n_where_device_beats_all_host_threads": min(crossa) if crossa else None,
            "note": "device times include the per-call host synchronisations of the C-ABI (factor status read-back, solve finiteness); below the crossover a "
                    "host-side Cholesky is the faster backend and the reference's dense_cholesky should be kept.  CPU rows: child process, default OpenMP wait policy, idle GPU"}


def cpu_size_sweep(args):
    """CPU half of the size sweep (runs in the CPU child): the oracle on one thread and on min(available, 32) threads, one untimed step first"""
    import numpy as np
    from oracle import pyorc
    from qp_gen import dense_strongly_convex_qp, random_vars
    try:
        L = pyorc.lib(native=True)
    except Exception:
        L = pyorc.lib()
    avail = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        pass
    rows = {}
    for n in SWEEP_SIZES:
        if n > 2048:
            continue
        q = dense_strongly_convex_qp(n, 0, n, seed=900 + n, double_sided=True, exact_shift=False)
        od = pyorc.Data.dense(**q, L=L)
        rng = np.random.default_rng(1000)
        state = random_vars(n, 0, n, rng, positive=True)
        rhs = [random_vars(n, 0, n, rng) for _ in range(2)]
        r = {}
        for label, threads in (("cpu_1_thread_ms_per_step", 1), ("cpu_all_threads_ms_per_step", min(avail, 32))):
            L.orc_set_num_threads(threads)
            ks = pyorc.KKTSystem(od, pyorc.Settings(L, kkt_solver=args.kkt_solver))
            ks.update_scalings_and_factor(False, 1e-6, 1e-4, state); ks.solve(rhs[0])
            flops = float(n) * (n + 1) * n + n ** 3 / 3.0
            reps = max(1, min(50, int(2e9 * (1 if threads == 1 else 4) / flops)))
            t0 = time.perf_counter()
            for _ in range(reps):
                ks.update_scalings_and_factor(False, 1e-6, 1e-4, state); ks.solve(rhs[0]); ks.solve(rhs[1])
            r[label] = (time.perf_counter() - t0) / reps * 1e3
        r["cpu_threads_all"] = min(avail, 32)
        rows[str(n)] = r
    return rows


def cpu_child(args):
    """`bench.py --cpu-child`: the dense CPU-baseline legs, no GPU, no torch; prints one JSON line"""
    from qp_gen import dense_strongly_convex_qp
    n, p, m = args.n, args.p, args.m
    q = dense_strongly_convex_qp(n, p, m, seed=43, double_sided=True, exact_shift=False)
    out = {"baseline": cpu_baseline(q, n, p, m, args)}
    if not args.no_size_sweep:
        try:
            out["sweep"] = cpu_size_sweep(args)
        except Exception as e:  # noqa: BLE001
            out["sweep_error"] = f"{type(e).__name__}: {e}"
    print(json.dumps(out), flush=True)


def cpu_baseline(q, n, p, m, args):
    """The oracle (CPU restatement of the reference algorithms) timed on this box's host cores on a bounded
    sample of the same workload: the same step (1 factor + 2 KKTSystem::solve).  kind = "port"."""
    import numpy as np
    from oracle import pyorc
    from qp_gen import random_vars
    try:
        L = pyorc.lib(native=True)  # -march=native build made on this machine
        build = "gcc -O3 -march=native -fopenmp"
    except Exception:
        L = pyorc.lib()
        build = "gcc -O3 -march=x86-64-v3 -fopenmp"
    avail = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # pick the thread count that is actually fastest on this box (cgroup quotas / SMT make "all logical CPUs" a bad default):
    # one factorisation of a small instance per candidate
    od = pyorc.Data.dense(**q, L=L)
    cal_state = random_vars(n, p, m, np.random.default_rng(1), positive=True)
    best, cores = None, 1
    sweep = {}
    for t in [c for c in (1, 8, 16, 32, 64, 96, 128, 192, 256) if c <= avail]:
        if t == 1 and float(n) * n * m > 2e10:
            continue  # (one thread at the full size would take the whole CPU budget; its rate is printed from the size sweep's n = 2048 row)
        L.orc_set_num_threads(t)
        kc = pyorc.KKTSystem(od, pyorc.Settings(L, kkt_solver=args.kkt_solver))
        kc.update_scalings_and_factor(False, 1e-6, 1e-4, cal_state)  # untimed: first touch of the workspaces, thread team start-up
        t0 = time.perf_counter()
        kc.update_scalings_and_factor(False, 1e-6, 1e-4, cal_state)
        dt = time.perf_counter() - t0
        sweep[str(t)] = (float(n) * (n + 1) * m + n ** 3 / 3.0) / dt / 1e9
        if best is None or dt < best:
            best, cores = dt, t
    L.orc_set_num_threads(cores)
    ks = pyorc.KKTSystem(od, pyorc.Settings(L, kkt_solver=args.kkt_solver))
    rng = np.random.default_rng(1000)
    state = random_vars(n, p, m, rng, positive=True)
    rhs = [random_vars(n, p, m, rng) for _ in range(2)]

    def step():
        ok = ks.update_scalings_and_factor(False, 1e-6, 1e-4, state)
        ks.solve(rhs[0])
        ks.solve(rhs[1])
        return ok

    t0 = time.perf_counter()
    assert step()
    first = time.perf_counter() - t0
    steps = args.cpu_steps or max(1, min(20, int(15.0 / max(first, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    el = time.perf_counter() - t0
    flops = float(n) * (n + 1) * m + n ** 3 / 3.0
    return {"value": steps / el, "unit": "IPM-iter KKT (1 factor + 2 solves)/s", "cores": cores, "kind": "port",
            "sample": f"{steps} steps of the same n={n} p={p} m={m} workload (after 1 untimed step), oracle built with {build}, {cores} OpenMP threads (fastest of a 1..{avail} sweep)",
            "seconds": el, "factor_gflops": flops * steps / el / 1e9, "factor_gflops_per_thread": flops * steps / el / 1e9 / cores,
            "thread_sweep_factor_gflops": sweep,
            "note": "GotoBLAS-structure SYRK / rank update with a register-blocked AVX2 8 x 6 (AVX-512: 16 x 12) micro-kernel on packed panels, 2-D task decomposition "
                    "(oracle/orc_dense.c syrk_like_lower): 22-31 GFLOP/s on one 2.1 GHz Xeon thread of the build container"}


if __name__ == "__main__":
    main()
