"""GPU parity tests of the dense KKT hot path: HIP kernels (through the C-ABI) vs the CPU oracle on the
same seeded inputs.  Tolerances: fp64; north_star asks <= 1e-10 relative KKT residual; element-wise
comparisons against the oracle use bounds scaled by the conditioning of the test matrices.

Mirrors the reference's tests/src/dense/kkt_test.cpp (UpdateData, FactorizeSolve) and ldlt_test.cpp,
run against both implementations.
"""
import os

import numpy as np
import pytest

from qp_gen import dense_strongly_convex_qp, random_vars

pytestmark = pytest.mark.gpu

DIMS = [(20, 8, 9), (10, 8, 9), (64, 0, 30), (40, 12, 0), (33, 5, 0), (130, 20, 70), (200, 50, 100), (257, 3, 129), (384, 64, 256), (512, 0, 512)]


def _mk(hip, orc, n, p, m, seed):
    q = dense_strongly_convex_qp(n, p, m, seed=seed)
    return q, hip.Data(**q), orc.Data.dense(**q)


def _rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / (1e-300 + np.abs(b).max()))


@pytest.mark.parametrize("dims", DIMS)
def test_mfma_f64_assembly_matches_oracle(hip, orc, dims):
    """dense/kkt.hpp:140-160 update_kkt: K_lower = P^T + diag(x_reg) + AT_A/delta + GT diag(1/z_reg) GT^T"""
    n, p, m = dims
    q, d, od = _mk(hip, orc, n, p, m, seed=n + m)
    rng = np.random.default_rng(1)
    x_reg = rng.uniform(0.5, 2.0, n); z_reg = rng.uniform(0.1, 3.0, m); delta = 1.2
    k = hip.DenseKKT(d)
    ko = orc.KKT(od)
    assert k.update_scalings_and_factor(delta, x_reg, z_reg)
    assert ko.update_scalings_and_factor(delta, x_reg, z_reg)
    K, Ko = np.tril(k.internal_kkt_mat()), np.tril(ko.internal_kkt_mat())
    assert _rel(K, Ko) < 1e-13
    # asymmetric-operand check of the MFMA fragment mapping: every entry individually close
    assert np.allclose(K, Ko, rtol=1e-12, atol=1e-12 * np.abs(Ko).max())


@pytest.mark.parametrize("kkt_solver", [0, 16])
@pytest.mark.parametrize("dims", DIMS)
def test_factor_matches_oracle(hip, orc, kkt_solver, dims):
    """Eigen::LLT (dense/kkt.hpp:82) / LDLTNoPivot (dense/ldlt_no_pivot.hpp:313-354): factor entries vs the oracle's"""
    n, p, m = dims
    q, d, od = _mk(hip, orc, n, p, m, seed=2 * n + p)
    rng = np.random.default_rng(2)
    x_reg = rng.uniform(0.5, 2.0, n); z_reg = rng.uniform(0.1, 3.0, m); delta = 0.7
    k = hip.DenseKKT(d, kkt_solver=kkt_solver)
    ko = orc.KKT(od, use_ldlt=(kkt_solver == 16))
    assert k.update_scalings_and_factor(delta, x_reg, z_reg)
    assert ko.update_scalings_and_factor(delta, x_reg, z_reg)
    F, Fo = np.tril(k.internal_factor()), np.tril(ko.internal_factor())
    assert _rel(F, Fo) < 1e-10
    # reconstruct K from the device factor
    K = np.tril(ko.internal_kkt_mat()); K = K + np.tril(K, -1).T
    if kkt_solver == 0:
        R = F @ F.T
    else:
        L = np.tril(F, -1) + np.eye(n)
        R = L @ np.diag(np.diag(F)) @ L.T
    assert _rel(R, K) < 1e-12


@pytest.mark.parametrize("kkt_solver", [0, 16])
@pytest.mark.parametrize("dims", DIMS)
def test_backend_solve_and_evals_match_oracle(hip, orc, kkt_solver, dims):
    """dense/kkt.hpp:86-132 solve + eval_P_x / eval_A.. / eval_G.. against the oracle"""
    n, p, m = dims
    q, d, od = _mk(hip, orc, n, p, m, seed=3 * n + m)
    rng = np.random.default_rng(3)
    x_reg = rng.uniform(0.5, 2.0, n); z_reg = rng.uniform(0.1, 3.0, m); delta = 1.2
    k = hip.DenseKKT(d, kkt_solver=kkt_solver)
    ko = orc.KKT(od, use_ldlt=(kkt_solver == 16))
    assert k.update_scalings_and_factor(delta, x_reg, z_reg) and ko.update_scalings_and_factor(delta, x_reg, z_reg)
    rx, ry, rz = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(m)
    lx, ly, lz = k.solve(rx, ry, rz)
    ox, oy, oz = ko.solve(rx, ry, rz)
    assert _rel(lx, ox) < 1e-9 and _rel(ly, oy) < 1e-9 and _rel(lz, oz) < 1e-9
    # residual of the 3x3 condensed system (kkt_system.hpp:507-519), computed in numpy
    P = np.triu(q["P"]); Pf = P + np.triu(P, 1).T
    A = q["A"] if p else np.zeros((0, n)); G = q["G"] if m else np.zeros((0, n))
    r1 = rx - (Pf @ lx + x_reg * lx + A.T @ ly + G.T @ lz)
    r2 = ry - (A @ lx - delta * ly)
    r3 = rz - (G @ lx - z_reg * lz)
    nrm = max([np.abs(v).max() for v in (rx, ry, rz) if v.size])
    assert max([np.abs(v).max() for v in (r1, r2, r3) if v.size]) <= 1e-10 * nrm
    x = rng.standard_normal(n); y = rng.standard_normal(p); z = rng.standard_normal(m)
    assert _rel(k.eval_P_x(-1.5, x), ko.eval_P_x(-1.5, x)) < 1e-13
    for (a, b_) in zip(k.eval_A_xn_and_AT_xt(-1.0, 2.0, x, y), ko.eval_A_xn_and_AT_xt(-1.0, 2.0, x, y)):
        assert _rel(a, b_) < 1e-13 or np.abs(np.asarray(a) - b_).max() < 1e-13
    for (a, b_) in zip(k.eval_G_xn_and_GT_xt(0.5, -3.0, x, z), ko.eval_G_xn_and_GT_xt(0.5, -3.0, x, z)):
        assert _rel(a, b_) < 1e-13 or np.abs(np.asarray(a) - b_).max() < 1e-13


def test_update_data_equals_fresh_bitwise(hip, orc):
    """dense/kkt_test.cpp:24-65: update_data + refactor == fresh backend, lower triangle bit-equal"""
    q = dense_strongly_convex_qp(10, 8, 9, seed=1)
    d = hip.Data(**q)
    d.P_utri[1, 1] = 0.0
    rho, delta = 0.9, 1.2
    x_reg = np.full(10, rho); z_reg = np.full(9, 1 + delta)
    k = hip.DenseKKT(d)
    assert k.update_scalings_and_factor(delta, x_reg, z_reg)
    q2 = dense_strongly_convex_qp(10, 8, 9, seed=2)
    d2 = hip.Data(**q2)
    k.update_data(d2, hip.KKT_UPDATE_P | hip.KKT_UPDATE_A | hip.KKT_UPDATE_G)
    assert k.update_scalings_and_factor(delta, x_reg, z_reg)
    k2 = hip.DenseKKT(d2)
    assert k2.update_scalings_and_factor(delta, x_reg, z_reg)
    assert np.array_equal(np.tril(k.internal_kkt_mat()), np.tril(k2.internal_kkt_mat()))
    assert np.array_equal(np.tril(k.internal_factor()), np.tril(k2.internal_factor()))


def test_factor_failure_semantics(hip, orc):
    """LLT fails iff a pivot <= 0 (dense/kkt.hpp:83); LDLTNoPivot, the class, only on an exact zero pivot (ldlt_no_pivot.hpp:307: tests/test_potrf_block_gpu.py holds the
    kernel to that, indefinite blocks included).  The dense BACKEND with kkt_solver = dense_ldlt_no_pivot -- this library's extension: dense/kkt.hpp itself only ever calls
    Eigen::LLT -- reports a pivot that is not positive like the LLT backend does (round 6): its matrix is positive definite by construction, a negative pivot is a
    breakdown, and carrying it on cost three Maros-Meszaros problems their convergence (profiles/r06_dense_mm_parity.txt).  The oracle's restated class keeps the
    reference's test."""
    n = 40
    P = -np.eye(n)  # negative definite, no constraints
    d = hip.Data(P, np.zeros(n))
    od = orc.Data.dense(P, np.zeros(n))
    x_reg = np.full(n, 0.5); z_reg = np.zeros(0)
    assert hip.DenseKKT(d).update_scalings_and_factor(1.0, x_reg, z_reg) is False
    assert orc.KKT(od).update_scalings_and_factor(1.0, x_reg, z_reg) is False
    assert hip.DenseKKT(d, kkt_solver=16).update_scalings_and_factor(1.0, x_reg, z_reg) is False
    assert orc.KKT(od, use_ldlt=True).update_scalings_and_factor(1.0, x_reg, z_reg) is True
    x_reg0 = np.full(n, 1.0)  # P + I = 0 -> exact zero pivot
    assert hip.DenseKKT(d, kkt_solver=16).update_scalings_and_factor(1.0, x_reg0, z_reg) is False
    assert orc.KKT(od, use_ldlt=True).update_scalings_and_factor(1.0, x_reg0, z_reg) is False
    # failure deep inside a large matrix (second panel)
    n = 300
    q = dense_strongly_convex_qp(n, 0, 0, seed=9)
    Pn = q["P"].copy(); Pn[200, 200] = -1e6
    d = hip.Data(Pn, q["c"]); od = orc.Data.dense(Pn, q["c"])
    assert hip.DenseKKT(d).update_scalings_and_factor(1.0, np.full(n, 1e-3), z_reg) is False
    assert orc.KKT(od).update_scalings_and_factor(1.0, np.full(n, 1e-3), z_reg) is False


@pytest.mark.parametrize("kkt_solver", [0, 16])
@pytest.mark.parametrize("dims", [(20, 8, 9), (64, 0, 30), (40, 12, 0), (130, 20, 70), (200, 50, 100)])
def test_kkt_system_factorize_solve(hip, orc, kkt_solver, dims):
    """dense/kkt_test.cpp:67-139 FactorizeSolve through pq_kktsys_*: mul(solve(rhs)) ~ rhs (1e-8) and == oracle"""
    n, p, m = dims
    q, d, od = _mk(hip, orc, n, p, m, seed=7 + n)
    st = hip.default_settings(kkt_solver=kkt_solver)
    k = hip.KKTSystem(d, st)
    ko = orc.KKTSystem(od, orc.Settings(kkt_solver=kkt_solver))
    scaling = hip.Variables.zeros(n, p, m, fill=1.0)
    assert k.update_scalings_and_factor(False, 0.9, 1.2, scaling)
    assert ko.update_scalings_and_factor(False, 0.9, 1.2, scaling)
    rng = np.random.default_rng(0)
    rhs = random_vars(n, p, m, rng)
    ok, lhs = k.solve(rhs)
    oko, ref = ko.solve(rhs)
    assert ok and oko
    back = k.mul(lhs)
    oback = ko.mul(lhs)
    nxl, nxu = d.n_x_l, d.n_x_u
    for key in ("x", "y"):
        assert np.allclose(rhs[key], back[key], rtol=1e-8, atol=1e-8)
    for key, cnt in (("z_bl", nxl), ("z_bu", nxu), ("s_bl", nxl), ("s_bu", nxu)):
        assert np.allclose(rhs[key][:cnt], back[key][:cnt], rtol=1e-8, atol=1e-8)
        assert np.allclose(back[key][:cnt], oback[key][:cnt], rtol=1e-12, atol=1e-12)
    for key, idx in (("z_l", d.h_l_idx), ("s_l", d.h_l_idx), ("z_u", d.h_u_idx), ("s_u", d.h_u_idx)):
        assert np.allclose(rhs[key][idx], back[key][idx], rtol=0, atol=1e-8)
    for key in lhs:
        cnt = {"z_bl": nxl, "s_bl": nxl, "z_bu": nxu, "s_bu": nxu}.get(key, len(ref[key]))
        assert _rel(lhs[key][:cnt], ref[key][:cnt]) < 1e-9, key
    res, nrm = k.condensed_residual()
    assert res <= 1e-10 * nrm


@pytest.mark.parametrize("dims", [(60, 10, 40), (200, 50, 100), (300, 0, 200)])
def test_kkt_system_iterative_refinement(hip, orc, dims):
    """kkt_system.hpp:195-207,256-301: static regularisation + refinement loop on an interior IPM state;
    same number of refinement steps as the oracle and the same solution."""
    n, p, m = dims
    q, d, od = _mk(hip, orc, n, p, m, seed=11 + n)
    k = hip.KKTSystem(d)
    ko = orc.KKTSystem(od)
    rng = np.random.default_rng(5)
    state = random_vars(n, p, m, rng, positive=True)
    assert k.update_scalings_and_factor(True, 1e-6, 1e-4, state)
    assert ko.update_scalings_and_factor(True, 1e-6, 1e-4, state)
    rhs = random_vars(n, p, m, rng)
    ok, lhs = k.solve(rhs)
    oko, ref = ko.solve(rhs)
    assert ok and oko
    stats = k.last_solve_stats()
    assert abs(stats["refine_steps"] - ko.last_refine_steps()) <= 1
    res, nrm = k.condensed_residual()
    assert res <= 1e-10 * nrm
    nxl, nxu = d.n_x_l, d.n_x_u
    for key in lhs:
        cnt = {"z_bl": nxl, "s_bl": nxl, "z_bu": nxu, "s_bu": nxu}.get(key, len(ref[key]))
        assert _rel(lhs[key][:cnt], ref[key][:cnt]) < 1e-8, key


def test_clone_is_bitwise_deterministic(hip):
    """kkt_system.hpp:70-95 clone + tests/src/dense/solver_test.cpp:379-401: a copy gives bit-identical results"""
    n, p, m = 200, 50, 100
    q = dense_strongly_convex_qp(n, p, m, seed=21)
    d = hip.Data(**q)
    k = hip.KKTSystem(d)
    rng = np.random.default_rng(8)
    state = random_vars(n, p, m, rng, positive=True)
    rhs = random_vars(n, p, m, rng)
    assert k.update_scalings_and_factor(False, 1e-6, 1e-4, state)
    k2 = k.clone()
    ok1, l1 = k.solve(rhs)
    ok2, l2 = k2.solve(rhs)
    assert ok1 and ok2
    for key in l1:
        assert np.array_equal(l1[key], l2[key]), key
    # refactor in the clone from the same state: still bit-identical
    assert k2.update_scalings_and_factor(False, 1e-6, 1e-4, state)
    ok3, l3 = k2.solve(rhs)
    for key in l1:
        assert np.array_equal(l1[key], l3[key]), key


def test_device_pointer_mode_matches_host_mode(hip):
    """PQ_MEM_DEVICE: vectors resident in HBM (torch CUDA tensors) give the same bits as the staged host path"""
    import torch
    n, p, m = 130, 20, 70
    q = dense_strongly_convex_qp(n, p, m, seed=31)
    d = hip.Data(**q)
    k = hip.KKTSystem(d)
    rng = np.random.default_rng(9)
    state = random_vars(n, p, m, rng, positive=True)
    rhs = random_vars(n, p, m, rng)
    assert k.update_scalings_and_factor(False, 1e-6, 1e-4, state)
    ok, lhs = k.solve(rhs)
    dstate = {kk: torch.from_numpy(v).cuda() for kk, v in state.items()}
    drhs = {kk: torch.from_numpy(v).cuda() for kk, v in rhs.items()}
    assert k.update_scalings_and_factor(False, 1e-6, 1e-4, dstate)
    ok2, dl = k.solve(drhs)
    k.synchronize()
    assert ok and ok2
    for key in lhs:
        cnt = {"z_bl": d.n_x_l, "s_bl": d.n_x_l, "z_bu": d.n_x_u, "s_bu": d.n_x_u}.get(key, len(lhs[key]))
        assert np.array_equal(lhs[key][:cnt], dl[key].cpu().numpy()[:cnt]), key


@pytest.mark.parametrize("kkt_solver", [0, 16])
@pytest.mark.parametrize("n", [1024, 1100, 1930])
def test_sweeps_with_block_inverses_against_the_substitution_sweeps(hip, kkt_solver, n):
    """round 5: from eight block rows on the triangular sweeps multiply by the inverses of the 128-row diagonal blocks (computed in double-double after every
    factorisation) and helper workgroups stream the block rows; PIQP_AMD_DEBUG=inv_sweeps=0 keeps the substitution form (tests/workers/dense_sweeps.py runs one
    of the two).  Same factor, five solves in a row on one handle (the hand-over buffers are re-armed by the sweeps themselves), a ragged last block at
    n = 1100 / 1930: each solution solves L L^T x = b (L D L^T x = b) to the substitution's residual, and the two agree to rounding."""
    import json
    import subprocess
    import sys
    out = {}
    for tag, tok in (("inv", "inv_sweeps=1"), ("subst", "inv_sweeps=0")):
        env = dict(os.environ); env["PIQP_AMD_DEBUG"] = tok
        r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "workers", "dense_sweeps.py"), str(n), str(kkt_solver), tag], env=env,
                           capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        assert line, (r.stdout[-2000:], r.stderr[-2000:])
        out[tag] = json.loads(line[0][7:])
    for i, (ri, rs) in enumerate(zip(out["inv"]["res"], out["subst"]["res"])):
        assert ri <= 2.0 * rs + 1e-15, (i, ri, rs)
    xa = np.load(out["inv"]["x"]); xb = np.load(out["subst"]["x"])
    assert np.abs(xa - xb).max() <= 1e-9 * np.abs(xb).max()
    assert out["inv"]["repeat_bitwise"] and out["subst"]["repeat_bitwise"]


@pytest.mark.parametrize("n,m", [(1024, 1024), (2048, 1536)])
def test_large_factor_solve_residual(hip, n, m):
    """size-independent property at larger sizes: ||rhs - K lhs|| / ||rhs|| <= 1e-10 (north_star tolerance)"""
    q = dense_strongly_convex_qp(n, 0, m, seed=n, double_sided=True)
    d = hip.Data(**q)
    for solver in (0, 16):
        k = hip.KKTSystem(d, hip.default_settings(kkt_solver=solver))
        rng = np.random.default_rng(4)
        state = random_vars(n, 0, m, rng, positive=True)
        assert k.update_scalings_and_factor(False, 1e-6, 1e-4, state)
        ok, lhs = k.solve(random_vars(n, 0, m, rng))
        assert ok
        res, nrm = k.condensed_residual()
        assert res <= 1e-10 * nrm, (solver, res, nrm)


def test_assembly_split_k_tail_matches_numpy(hip):
    """n = 4096 gives 528 lower tiles on 512 workgroup slots: the last 16 tiles take the split-K path
    (k_syrk_lower partial tiles + k_syrk_tail_reduce).  Checked against a NumPy fp64 GEMM."""
    n, m = 4096, 512
    rng = np.random.default_rng(12)
    G = rng.standard_normal((m, n))
    P = np.triu(rng.standard_normal((n, n)) * 0.01) + np.diag(np.full(n, 5.0))
    d = hip.Data(P, np.zeros(n), None, None, G, -np.ones(m), np.ones(m))
    k = hip.DenseKKT(d)
    x_reg = rng.uniform(0.5, 2.0, n); z_reg = rng.uniform(0.1, 3.0, m)
    assert k.update_scalings_and_factor(1.0, x_reg, z_reg)
    K = np.tril(k.internal_kkt_mat())
    Pf = np.triu(P) + np.triu(P, 1).T
    ref = np.tril(Pf + np.diag(x_reg) + (G.T * (1.0 / z_reg)) @ G)
    assert np.abs(K - ref).max() <= 1e-12 * np.abs(ref).max()



KNOWN_DEVICE_ONLY_FAILURES = {0: [], 16: [12, 13, 16, 18, 19, 20]}  # replay of qp_robot_arm_sqp: kkt_solver -> states the oracle factorises and the device reports as failed (round 5: none -- the scalings x_reg / z_reg are formed without FMA contraction now, bitwise the oracle's, and state 12 of the LL^T replay no longer differs in the sign of its last pivot; rounds 3-4 had [12])


@pytest.mark.parametrize("kkt_solver", [0, 16])
def test_accuracy_on_recorded_ipm_states_of_the_hardest_fixture(hip, orc, kkt_solver):
    """qp_robot_arm_sqp ends at rho = delta = 1e-10 (the regularisation floor): the condensed matrix loses ~8 digits in ANY factorisation
    (the oracle's relative residual is ~1e-8 there, far above the 1e-10 bar, which no implementation can meet on these states).  The device
    panel solve multiplies by explicitly inverted 16 x 16 diagonal pieces (k_trsm_panel) instead of substituting, so the parity statement
    that matters is checked on every recorded state of the oracle's solve: wherever both factorisations succeed, the device residual
    (extended precision) is within one decimal digit of the oracle's on every state (the ratios scatter between 0.1 and 6 once both sit at
    1e-7: rounding noise), within a factor 2 in the median, and meets the bar wherever the oracle does."""
    from dense_replay import replay
    rows = replay("qp_robot_arm_sqp", kkt_solver)
    both = [(it, rh, ro) for it, rho, delta, okh, oko, rh, ro in rows if okh and oko]
    dev_only_fail = [it for it, rho, delta, okh, oko, rh, ro in rows if oko and not okh]
    orc_only_fail = [it for it, rho, delta, okh, oko, rh, ro in rows if okh and not oko]
    print(f"\nreplay kkt_solver={kkt_solver}: {len(rows)} states, both factorise on {len(both)}, device-only failures {dev_only_fail}, oracle-only failures {orc_only_fail}")
    # a state the oracle factorises and the device does not would silently drop out of the comparison below, so such states are counted, not skipped:
    # at most one, and only at the regularisation floor (rho = delta <= 1e-9), where the smallest pivot of the condensed matrix is of the size of the
    # assembly's rounding error and its sign -- the failure criterion of Eigen::LLT, dense/kkt.hpp:83 -- is decided by the summation order of G' W G
    # (measured in round 3: state 12 of 20 for LL^T, none for LDL^T)
    floor = {it for it, rho, delta, okh, oko, rh, ro in rows if rho <= 1e-9 and delta <= 1e-9}
    # (kkt_solver = 16, round 6: the device's dense L D L^T backend reports a pivot that is not positive -- like LLT -- where the oracle's restated class only tests for an
    # exact zero and factors on with a negative pivot: six states of this replay, all at the floor)
    assert (kkt_solver == 16 or len(dev_only_fail) <= 1) and set(dev_only_fail) <= floor, dev_only_fail
    # pinned (round-3 advice): exactly the known state for LL^T, none for LDL^T -- a second one, or another one, is a change of the assembly's arithmetic
    assert dev_only_fail == KNOWN_DEVICE_ONLY_FAILURES[kkt_solver], (kkt_solver, dev_only_fail)
    assert len(both) >= 10
    for it, rh, ro in both:
        assert rh <= max(10.0 * ro, 1e-12), (it, rh, ro)
        if ro <= 2.5e-11:
            assert rh <= 1e-10, (it, rh, ro)
    assert np.median([rh / ro for _, rh, ro in both]) <= 2.0


def test_persistent_factorisation_is_bitwise_the_launch_per_panel_one(hip):
    """round 3: k_chol_persistent (every round of the blocked factorisation in ONE launch: ticket-ordered task list, look-ahead, the first panel row handed
    over slice by slice, two workgroups per panel tile) against the launch-per-panel path (PIQP_AMD_DEBUG=chol_launches): every tile receives the same
    products in the same order, so factor, reciprocal pivots (through a solve) and success flag agree bit for bit, for LL^T and LDL^T, run after run"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "chk_chol_persistent.py"), "384", "640", "1024", "2048", "4096"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ALL EQUAL" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("kkt_solver", [0, 16])
@pytest.mark.parametrize("n,bad", [(512, 300), (1024, 700), (1024, 100)])
def test_failure_inside_the_persistent_launch_then_refactor_on_the_same_handle(hip, orc, n, bad, kkt_solver):
    """round-3 advice: the non-positive pivot (LL^T, dense/kkt.hpp:83) / exact zero pivot (LDL^T, ldlt_no_pivot.hpp:307) INSIDE k_chol_persistent -- a later
    diagonal block, reached through the launch's own rounds -- is reported like the launch-per-panel path and the oracle report it, and the handle's cumulative
    flag counters survive it: a good matrix factored next on the same handle gives bitwise the factor of a fresh handle"""
    q = dense_strongly_convex_qp(n, 0, 0, seed=3 + n)
    Pg = q["P"].copy()
    Pb = Pg.copy()
    if kkt_solver == 0:
        Pb[bad, bad] = -1e6  # indefinite: LL^T must fail, LDL^T would not
    else:
        Pb[bad, :] = 0.0; Pb[:, bad] = 0.0  # a zero row / column: with x_reg = 0 there the pivot is exactly zero
    x_reg = np.full(n, 1e-3); z_reg = np.zeros(0)
    if kkt_solver == 16:
        x_reg = x_reg.copy(); x_reg[bad] = 0.0
    d = hip.Data(Pb, q["c"])
    k = hip.DenseKKT(d, kkt_solver=kkt_solver)
    assert k.update_scalings_and_factor(1.0, x_reg, z_reg) is False
    assert orc.KKT(orc.Data.dense(Pb, q["c"]), use_ldlt=kkt_solver == 16).update_scalings_and_factor(1.0, x_reg, z_reg) is False
    assert k.update_scalings_and_factor(1.0, x_reg, z_reg) is False  # (and again: the counters of the failed launch are consistent)
    # the same handle, good data
    k.update_data(hip.Data(Pg, q["c"]), 7)  # KKT_UPDATE_P | A | G
    xg = np.full(n, 1e-3)
    assert k.update_scalings_and_factor(1.0, xg, z_reg) is True
    fresh = hip.DenseKKT(hip.Data(Pg, q["c"]), kkt_solver=kkt_solver)
    assert fresh.update_scalings_and_factor(1.0, xg, z_reg) is True
    assert np.array_equal(np.tril(k.internal_factor()), np.tril(fresh.internal_factor()))


@pytest.mark.parametrize("n", [2048, 640])
def test_two_persistent_factorisations_on_two_streams(hip, n):
    """two handles (one a clone of the other) factor and solve from two host threads at the same time: the persistent launches of both compete for the
    CUs, neither is ever fully resident -- the ticket order guarantees progress with any number of resident workgroups -- and every result equals the
    single-threaded one bit for bit.  n = 2048: sweeps with block inverses and helper workgroups, the inverses on a side stream; n = 640: substitution sweeps
    on one XCD (both handles draw their block rows by ticket among the workgroups that land there)"""
    import threading
    import torch
    m = n
    q = dense_strongly_convex_qp(n, 0, m, seed=5, double_sided=True, exact_shift=False)
    k1 = hip.KKTSystem(hip.Data(**q), hip.default_settings(kkt_solver=0))
    k2 = k1.clone()
    rng = np.random.default_rng(0)
    sv = random_vars(n, 0, m, rng, positive=True); rv = random_vars(n, 0, m, rng)
    out, err = {}, []

    def work(tag, k, reps):
        try:
            state = {kk: torch.from_numpy(v).cuda() for kk, v in sv.items()}
            rhs = {kk: torch.from_numpy(v).cuda() for kk, v in rv.items()}
            lhs = {kk: torch.zeros_like(v) for kk, v in rhs.items()}
            ref = None
            for it in range(reps):
                assert k.update_scalings_and_factor(False, 1e-6, 1e-4, state)
                k.solve(rhs, lhs)
                x = lhs["x"].cpu().numpy().copy()
                if ref is None:
                    ref = x
                assert np.array_equal(x, ref), (tag, it)
            out[tag] = ref
        except Exception as e:  # noqa: BLE001
            err.append((tag, repr(e)))

    work("single", k1, 3)
    t1 = threading.Thread(target=work, args=("a", k1, 40)); t2 = threading.Thread(target=work, args=("b", k2, 40))
    t1.start(); t2.start(); t1.join(300); t2.join(300)
    assert not err, err
    assert np.array_equal(out["a"], out["single"]) and np.array_equal(out["b"], out["single"])
