"""Stage partition of one KKT system over several processes (include/piqp_amd.h, pq_kkt_partition).

CPU part: the subtree-to-rank plan (host code only, no device).  GPU part: two ranks that share the test box's single GPU over
gloo reproduce the single-GPU factor / solve / full interior-point solve bit for bit (tools/dist_c5.py)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _plan(args, mode, world):
    import piqp_amd as hip
    from piqp_amd import _lib
    L = _lib.load()
    d = hip.SparseData(*args)
    desc = d.descriptor()
    N = L.pq_sparse_partition_plan(C.byref(desc), mode, world, None, 0, None)
    assert N > 0
    owner = np.full(N, -7, dtype=np.int32)
    work = np.zeros(world + 1)
    assert L.pq_sparse_partition_plan(C.byref(desc), mode, world, owner.ctypes.data, N, work.ctypes.data) == N
    return owner, work


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_plan_chain(world):
    """a multistage chain: every rank gets one contiguous range of the elimination order (= of the stages), balanced, and the
    replicated top is a negligible share of the work"""
    from qp_gen import mpc_chain
    owner, work = _plan(mpc_chain(4, 2, 1500, 3), 3, world)
    assert owner.min() >= -1 and owner.max() == world - 1
    if world == 1:
        assert (owner == 0).all() and work[1] == 0.0
        return
    per_rank, shared = work[:world], work[world]
    assert per_rank.min() > 0
    assert per_rank.max() <= 1.25 * per_rank.mean()
    assert shared <= 0.10 * per_rank.sum()  # this 1500-stage chain is small for 8 ranks (6.5 % replicated); the n = 500k chain of configs[4] is < 0.1 %
    # contiguity: dropping the shared columns, the owner sequence is non-decreasing
    o = owner[owner >= 0]
    assert (np.diff(o) >= 0).all()
    assert (owner < 0).sum() < 0.08 * owner.size


def test_plan_chain_at_full_c5_size():
    """BASELINE configs[4] at its real size -- n = 500 012, 25 000 stages of n_x = 12, n_u = 8 -- planned for 8 ranks (host-only analysis, about 100 s
    on 8 cores): every rank owns one contiguous range of stages with 12-13 % of the factorisation work, the replicated top is 0.3 %."""
    from qp_gen import mpc_chain
    world = 8
    owner, work = _plan(mpc_chain(12, 8, 25000, 5), 3, world)
    assert owner.size == 500012 and owner.min() == -1 and owner.max() == world - 1
    per_rank, shared = work[:world], work[world]
    share = per_rank / per_rank.sum()
    assert share.min() >= 0.115 and share.max() <= 0.135, share
    assert shared <= 0.005 * per_rank.sum()
    o = owner[owner >= 0]
    assert (np.diff(o) >= 0).all()              # contiguous, in stage order
    assert (owner < 0).sum() <= 2000            # the shared separators: 1488 of 500 012 columns


@pytest.mark.parametrize("mode", [0, 3])
def test_plan_general_sparse(mode):
    """a general sparse QP: same invariants except balance / shared share, which depend on the top fronts"""
    from qp_gen import c3_problem
    args = c3_problem(n=2000, p=700, m=1100, seed=5, spread=40)
    for world in (2, 4):
        owner, work = _plan(args, mode, world)
        assert owner.min() >= -1 and owner.max() <= world - 1
        o = owner[owner >= 0]
        assert (np.diff(o) >= 0).all()
        assert abs(work.sum() - _plan(args, mode, 1)[1].sum()) <= 1e-9 * work.sum()


def _run_ranks(nranks, extra, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tools", "dist_c5.py")] + extra
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_USE_AGENT_STORE"):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and lines, r.stderr[-3000:]
    return json.loads(lines[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("backend,problem,nranks", [("multistage", "chain", 2), ("ldlt", "chain", 3), ("ldlt_cond", "c3", 2), ("ldlt", "cont", 2)])
def test_partitioned_equals_single_gpu(backend, problem, nranks):
    out = _run_ranks(nranks, ["--stages", "800", "--steps", "2", "--warmup", "1", "--backend", backend, "--problem", problem, "--full-solve"], 29650 + nranks)
    assert out["world"] == nranks
    assert out["bitwise_equal_all_ranks"] and out["max_abs_diff"] == 0.0
    assert out["rel_kkt_residual"] <= 1e-10
    assert out["exchange_calls"][0] >= 1 and out["exchange_calls"][1] == out["exchange_calls"][2] >= 1
    fs = out["full_solve"]
    assert fs["status"] == 1 and fs["identical_on_all_ranks"] and fs["x_equal_to_single_gpu"] and fs["iter"] == fs["single_gpu"]["iter"]
    assert sum(p["owned_supernodes"] for p in out["partition"]) > 0 and out["shared_supernodes"] >= 1


@pytest.mark.gpu
@pytest.mark.parametrize("problem,nranks", [("chain", 2), ("chain", 4), ("c3", 2)])
def test_sharded_refinement_residual_and_assembly_equal_the_replicated_ones(problem, nranks):
    """SURVEY 8(e) row 2 (round 4): with iterative refinement on, every rank of a stage-partitioned sparse_ldlt (KKT_FULL) backend evaluates the refinement residual
    on its own rows only (its subtrees + the shared top) and the norm crosses the ranks in one all-reduce(max) per refinement step; the per-factorisation value
    assembly touches only the diagonal entries of the fronts it factors.  The partitioned solve -- KKTSystem::solve with its refinement loop, and the whole
    interior-point solve with refinement always on -- must be bitwise the single-GPU one (which evaluates everything on every row), on every rank."""
    out = _run_ranks(nranks, ["--stages", "800", "--steps", "2", "--warmup", "1", "--backend", "ldlt", "--problem", problem, "--full-solve", "--refine"], 29690 + nranks)
    assert out["world"] == nranks
    assert out["bitwise_equal_all_ranks"] and out["max_abs_diff"] == 0.0
    assert out["rel_kkt_residual"] <= 1e-10
    sr = out["sharded_residual"]
    assert all(e >= 2 for e in sr["evaluations_per_rank"]), sr          # the residual really was evaluated in its sharded form ...
    assert sr["norm_all_reduces"] >= 2                                     # ... with its all-reduce(max)
    assert max(sr["rows_per_rank"]) < 0.75 * sr["rows_total"], sr         # ... on a share of the rows (own subtrees + shared top)
    assert sum(sr["rows_per_rank"]) >= sr["rows_total"]                   # every row is somebody's
    assert out["refine_steps"] == out["single_gpu_refine_steps"]
    fs = out["full_solve"]
    assert fs["status"] == 1 and fs["identical_on_all_ranks"] and fs["x_equal_to_single_gpu"] and fs["iter"] == fs["single_gpu"]["iter"]


@pytest.mark.gpu
@pytest.mark.parametrize("backend,nranks", [("multistage", 2), ("ldlt_cond", 4)])
def test_sharded_value_assembly_of_the_condensed_modes(backend, nranks):
    """SURVEY 8(e) row 2, condensed modes (round 4): a stage-partitioned sparse_ldlt_cond / sparse_multistage (tree engine) backend re-evaluates, per factorisation, only
    the values of the fronts it factors -- P entries, delta^-1 A^T A entries, G^T W G product terms and diagonal shifts selected by destination front.  The other
    fronts' values are never read on that rank: factor + solve stay bitwise the single-GPU ones on every rank, and every rank evaluates a share of the entries."""
    out = _run_ranks(nranks, ["--stages", "800", "--steps", "2", "--warmup", "1", "--backend", backend, "--full-solve"], 29720 + nranks)
    assert out["world"] == nranks
    assert out["bitwise_equal_all_ranks"] and out["max_abs_diff"] == 0.0
    assert out["rel_kkt_residual"] <= 1e-10
    sa = out["sharded_assembly"]
    assert all(a >= 3 for a in sa["assemblies_per_rank"]), sa                      # every factorisation of the partitioned handle assembled its share only
    total = out["sharded_assembly_entries_single_gpu"]
    assert max(sa["entries_per_rank"]) < 0.8 * total and sum(sa["entries_per_rank"]) >= total, (sa, total)
    fs = out["full_solve"]
    assert fs["status"] == 1 and fs["identical_on_all_ranks"] and fs["x_equal_to_single_gpu"] and fs["iter"] == fs["single_gpu"]["iter"]


@pytest.mark.gpu
@pytest.mark.parametrize("backend,problem,nranks", [("multistage", "chain", 2), ("multistage", "chain", 4), ("ldlt_cond", "chain", 2), ("ldlt_cond", "c3", 4)])
def test_sharded_solve_side_of_the_condensed_backends(backend, problem, nranks):
    """SURVEY 8(e) row 2, the solve side of the condensed modes and of sparse_multistage (round 5): with iterative refinement on, a stage-partitioned condensed backend
    evaluates the refinement residual on its own rows (x rows of its fronts + the constraint rows that touch them), folds that residual into ITS x rows only, recovers
    the eliminated multipliers on THOSE constraint rows only, and the refined multipliers cross the ranks once per KKTSystem::solve -- each row from its owner rank.
    KKTSystem::solve with its refinement loop and the whole interior-point solve must be bitwise the single-GPU ones (everything on every row) on every rank."""
    out = _run_ranks(nranks, ["--stages", "800", "--steps", "2", "--warmup", "1", "--backend", backend, "--problem", problem, "--full-solve", "--refine"], 29760 + nranks)
    assert out["world"] == nranks
    assert out["bitwise_equal_all_ranks"] and out["max_abs_diff"] == 0.0
    assert out["rel_kkt_residual"] <= 1e-10
    ss = out["sharded_solve"]
    assert all(e >= 2 for e in ss["residual_evaluations_per_rank"]), ss
    assert ss["norm_all_reduces"] >= 2
    assert all(e >= 1 for e in ss["partial_backend_solves_per_rank"]), ss     # refinement steps really solved on a partial right-hand side ...
    assert all(e >= 1 for e in ss["multiplier_gathers_per_rank"]), ss         # ... and the multipliers crossed the ranks afterwards
    assert max(ss["residual_rows_per_rank"]) < 0.8 * ss["rows_total"] and sum(ss["residual_rows_per_rank"]) >= ss["rows_total"], ss
    assert max(ss["x_rows_folded_per_rank"]) < 0.8 * ss["x_rows_total"] and sum(ss["x_rows_folded_per_rank"]) >= ss["x_rows_total"], ss
    assert max(ss["constraint_rows_recovered_per_rank"]) < 0.8 * ss["constraint_rows_total"], ss
    assert sum(ss["constraint_rows_recovered_per_rank"]) >= ss["constraint_rows_total"], ss
    assert out["refine_steps"] == out["single_gpu_refine_steps"]
    fs = out["full_solve"]
    assert fs["status"] == 1 and fs["identical_on_all_ranks"] and fs["x_equal_to_single_gpu"] and fs["iter"] == fs["single_gpu"]["iter"]


@pytest.mark.gpu
@pytest.mark.parametrize("ks", [1, 4])
def test_partition_of_a_system_small_enough_for_the_reference_order_engine(ks):
    """round-5 advice: kkt_solver = sparse_ldlt (and the condensed modes) on a KKT system of at most 8192 rows builds the reference-order engine, which has no stage
    partition.  pq_kkt_partition on such a handle used to fail ("not supported by this backend"); now the handle switches to the multifrontal engine, built from its
    own copy of the data, and partitions that: the partitioned (world = 1) factor + solve equals the multifrontal engine's, bit for bit."""
    import numpy as np
    import piqp_amd as hip
    from piqp_amd.dist import StagePartition
    from qp_gen import c3_problem
    args = c3_problem(n=900, p=300, m=500, seed=11, spread=40)
    d = hip.SparseData(*args)
    n, p, m = d.n, d.p, d.m
    rng = np.random.default_rng(3)
    x_reg = rng.uniform(0.5, 2.0, n); z_reg = rng.uniform(0.1, 3.0, m)
    r = [rng.standard_normal(k) for k in (n, p, m)]
    k = hip.SparseKKT(d, kkt_solver=ks)
    assert k.update_scalings_and_factor(0.7, x_reg, z_reg)
    ref_order = k.solve(*r)                                  # (the reference-order engine's answer, before the switch)
    part = StagePartition(k, rank=0, world=1)
    assert part.sizes[2] >= 0
    assert k.update_scalings_and_factor(0.7, x_reg, z_reg)
    got = k.solve(*r)
    env_ks = {1: hip.SPARSE_LDLT_MULTIFRONTAL, 4: 4}[ks]
    import os
    os.environ["PIQP_AMD_SPARSE_LDLT"] = "multifrontal"
    try:
        k2 = hip.SparseKKT(d, kkt_solver=env_ks)
    finally:
        del os.environ["PIQP_AMD_SPARSE_LDLT"]
    assert k2.update_scalings_and_factor(0.7, x_reg, z_reg)
    want = k2.solve(*r)
    for a, b, c in zip(got, want, ref_order):
        assert np.array_equal(a, b)
        assert np.allclose(a, c, rtol=1e-8, atol=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("native", [True, False])
def test_rccl_transport_with_a_one_rank_group(native):
    """all a 1-GPU box can say about the RCCL transport: a one-rank "nccl" process group with the exchanges forced on
    (PIQP_AMD_EXCHANGE_WORLD1).  native=True: the library's own communicator (pq_kkt_set_comm_rccl: ncclCommInitRank from the broadcast
    unique id, ncclAllReduce / ncclAllGather enqueued on the handle's stream, no callback); native=False: the callback transport
    (C library -> callback -> torch.distributed -> RCCL on the registered device buffers).  Plus the bookkeeping collectives on device
    tensors (tools/nccl_world1_check.py)."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_USE_AGENT_STORE", "PIQP_AMD_CALLBACK_EXCHANGE"):
        env.pop(k, None)
    env.update(PIQP_AMD_FORCE_PG="1", PIQP_AMD_EXCHANGE_WORLD1="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29671" if native else "29673")
    env.pop("PIQP_AMD_NATIVE_RCCL", None)
    if native:
        env["PIQP_AMD_NATIVE_RCCL"] = "1"  # opt-in since round 3 (piqp_amd/dist.py): no multi-rank run of the native transport exists yet
    else:
        env["PIQP_AMD_CALLBACK_EXCHANGE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dist_c5.py"), "--stages", "600", "--steps", "2", "--warmup", "1", "--full-solve"], capture_output=True, text=True,
                       timeout=600, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and lines, r.stderr[-3000:]
    out = json.loads(lines[-1])
    assert out["transport"] == "rccl" and out["world"] == 1 and out["native_rccl"] == native
    assert out["bitwise_equal_all_ranks"] and out["exchange_calls"][2] >= 3  # (one rank owns every subtree: no boundary, so only the gather runs)
    assert out["full_solve"]["status"] == 1 and out["full_solve"]["x_equal_to_single_gpu"]
    if native:
        return
    env["MASTER_PORT"] = "29672"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nccl_world1_check.py")], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


def test_condensed_mode_refuses_dense_constraint_rows_instead_of_exhausting_memory():
    """a block of dense equality rows makes A^T A dense; the term lists of the condensed modes would need billions of entries (the reference's own A^T A would be a
    dense n x n matrix): the analysis refuses with an error instead of taking the host down (round 4), the KKT_FULL mode takes the same data"""
    import scipy.sparse as sp
    import piqp_amd as hip
    from piqp_amd import _lib
    rng = np.random.default_rng(0)
    n, rows = 4000, 400
    P = sp.diags([rng.uniform(1, 2, n)], [0], format="csc")
    A = sp.csc_matrix(rng.standard_normal((rows, n)))
    d = hip.SparseData(P, np.zeros(n), A, np.zeros(rows), None, None, None, None, None)
    L = _lib.load()
    desc = d.descriptor()
    assert L.pq_sparse_partition_plan(C.byref(desc), 3, 1, None, 0, None) < 0
    assert "2e8 terms" in L.pq_last_error_string().decode()
    assert L.pq_sparse_partition_plan(C.byref(desc), 0, 1, None, 0, None) == n + rows


@pytest.mark.parametrize("spread,row_nnz,n", [(300, 10, 6000), (1000, 8, 3000)])
def test_symbolic_analysis_of_wide_window_problems_on_the_host(spread, row_nnz, n):
    """the round-4 symbolic paths -- spines of wide fronts merged into their parents, nested dissection against AMD on two host threads, the cost model for trees with
    many levels of big fronts -- on the host alone (no GPU): the analysis completes (every K entry falls inside its front, every child's update rows inside its
    parent's front: analyse_with_order throws otherwise) and a two-rank plan covers every column"""
    from qp_gen import c3_problem
    a = c3_problem(n, n * 2 // 5, n * 3 // 5, 45, spread, row_nnz)
    for mode in (0, 3):
        owner, work = _plan(a, mode, 2)
        assert owner.min() >= -1 and owner.max() == 1
        assert work[:2].min() > 0
