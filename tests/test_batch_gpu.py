"""GPU parity tests of the batched solver (pq_batch_*): every instance of a batch must end with the status, the iteration
count and the solution the CPU oracle (restatement of the reference's SparseSolver with kkt_solver = sparse_multistage)
reaches on that instance alone.  BASELINE configs[3]: linear-MPC QPs n = 120 (40 stages of n_x = 2, n_u = 1)."""
import numpy as np
import pytest

from qp_gen import mpc_batch, mpc_instance

pytestmark = pytest.mark.gpu


def _oracle_solve(orc, args):
    s = orc.Solver()
    s.settings.kkt_solver = orc.SPARSE_MULTISTAGE
    assert s.setup(*args, sparse=True)
    st = s.solve()
    return st, s.info.iter, s.info.primal_obj, s.result()


def _run_batch(hip, mb):
    bs = hip.BatchSparseSolver()
    assert bs.setup(mb["P_pattern"], mb["P_values"], mb["c"], mb["A_pattern"], mb["A_values"], mb["b"], x_l=mb["x_l"], x_u=mb["x_u"])
    solved = bs.solve()
    return bs, solved


@pytest.mark.parametrize("shuffle", [False, True])
def test_mpc_batch_matches_oracle_per_instance(hip, orc, shuffle):
    B = 24
    mb = mpc_batch(B, seed=1000, shuffle_rows=shuffle)
    bs, solved = _run_batch(hip, mb)
    assert solved == B
    x, y, zbl, zbu = bs.result("x"), bs.result("y"), bs.result("z_bl"), bs.result("z_bu")
    k = orc.KKT(orc.Data.sparse(*mpc_instance(mb, 0)), kind="multistage")
    assert np.array_equal(bs.block_info(), k.block_info())
    for i in range(B):
        st, it, obj, ref = _oracle_solve(orc, mpc_instance(mb, i))
        info = bs.info(i)
        assert info.status == st == 1
        assert info.iter == it, (i, info.iter, it)
        assert abs(info.primal_obj - obj) <= 1e-8 * (1 + abs(obj))
        scale = 1 + np.abs(ref["x"]).max()
        assert np.abs(x[i] - ref["x"]).max() <= 1e-7 * scale
        assert np.abs(y[i] - ref["y"]).max() <= 1e-6 * (1 + np.abs(ref["y"]).max())
        assert np.abs(zbl[i] - ref["z_bl"]).max() <= 1e-6 * (1 + np.abs(ref["z_bl"]).max())
        assert np.abs(zbu[i] - ref["z_bu"]).max() <= 1e-6 * (1 + np.abs(ref["z_bu"]).max())


def test_batch_equals_single_qp_device_solver(hip):
    """the batched kernel and the host-driven single-QP solver run the same algorithm on the same backend"""
    B = 6
    mb = mpc_batch(B, T=12, nx=3, nu=2, seed=77)
    bs, solved = _run_batch(hip, mb)
    assert solved == B
    x = bs.result("x")
    for i in range(B):
        s = hip.SparseSolver(); s.settings.kkt_solver = hip.SPARSE_MULTISTAGE
        assert s.setup(*mpc_instance(mb, i))
        assert s.solve() == 1
        assert s.info.iter == bs.info(i).iter
        assert np.abs(s.result()["x"] - x[i]).max() <= 1e-8 * (1 + np.abs(x[i]).max())


def test_batch_properties_at_full_size(hip):
    """BASELINE configs[3] at full size (8192 QPs): size-independent properties -- every instance solved, primal feasibility
    and bound satisfaction of the returned points, and instance i of the big batch == instance i of a small batch (bitwise:
    instances never interact)"""
    B = 8192
    mb = mpc_batch(B, seed=1000)
    bs, solved = _run_batch(hip, mb)
    assert solved == B
    x = bs.result("x")
    A = mb["A_pattern"].copy()
    worst = 0.0
    for i in range(0, B, 257):
        A.data = mb["A_values"][i]
        worst = max(worst, np.abs(A @ x[i] - mb["b"][i]).max())
        assert np.all(x[i] >= mb["x_l"][i] - 1e-6) and np.all(x[i] <= mb["x_u"][i] + 1e-6)
    assert worst <= 1e-6
    small = {k: (v[:16] if isinstance(v, np.ndarray) and v.ndim == 2 else v) for k, v in mb.items()}
    bs2, solved2 = _run_batch(hip, small)
    assert solved2 == 16
    assert np.array_equal(bs2.result("x"), x[:16])
    assert np.array_equal(bs2.iterations(), bs.iterations()[:16])


def test_c4_full_size_every_instance_vs_oracle(hip, orc):
    """BASELINE configs[3] at full size against the oracle, instance by instance (all 8192; the oracle needs 0.75 ms per instance): status and optimum
    equal everywhere; the iteration count equal except on instances whose termination test sits on its threshold.  This recipe stops on the duality-gap
    test with mu * (n_x_l + n_x_u) ~ 1.0e-8 = eps_duality_gap_abs at iteration 8 by construction, and the condensed multistage system (delta -> 1e-10)
    leaves ~1e-5 relative solve noise in both implementations, so which side of the threshold such an instance lands on is decided by rounding.  The
    mismatch set is asserted: at most 0.5 % of the batch, each off by exactly one iteration (the set itself, with both final gaps, is printed)."""
    B = 8192
    mb = mpc_batch(B, seed=1000)
    bs, solved = _run_batch(hip, mb)
    assert solved == B
    its = bs.iterations()
    mism = []
    for i in range(B):
        s = orc.Solver(); s.settings.kkt_solver = orc.SPARSE_MULTISTAGE
        assert s.setup(*mpc_instance(mb, i), sparse=True)
        st = s.solve()
        info = bs.info(i)
        assert info.status == st == 1, (i, info.status, st)
        assert abs(info.primal_obj - s.info.primal_obj) <= 1e-8 * (1 + abs(s.info.primal_obj)), i
        if int(its[i]) != s.info.iter:
            mism.append((i, int(its[i]), int(s.info.iter), float(info.duality_gap), float(s.info.duality_gap)))
    print(f"\nC4 full size: {B - len(mism)} of {B} instances with the oracle's iteration count; mismatches (instance, device iter, oracle iter, device gap, oracle gap): {mism}")
    assert len(mism) <= B // 200, len(mism)
    for (i, a, b, gh, go) in mism:
        assert abs(a - b) == 1, (i, a, b)


def test_batch_with_general_inequalities_and_one_sided_bounds(hip, orc):
    """m > 0 with one-sided rows and partially bounded variables exercises every branch of the in-kernel KKTSystem"""
    import scipy.sparse as sp
    B = 8
    mb = mpc_batch(B, T=10, nx=2, nu=2, seed=5)
    n, p = mb["n"], mb["p"]
    rng = np.random.default_rng(3)
    m = 12
    rows = np.repeat(np.arange(m), 2)
    cols = np.concatenate([[4 * (r % 10) + 0, 4 * (r % 10) + 2] for r in range(m)])
    G_pattern = sp.csc_matrix((np.ones(2 * m), (rows, cols)), shape=(m, n)); G_pattern.sort_indices()
    Gv = rng.standard_normal((B, G_pattern.nnz))
    h_l = np.tile(np.where(np.arange(m) % 3 == 0, -np.inf, -1.0), (B, 1)) * np.ones((B, m))
    h_u = np.tile(np.where(np.arange(m) % 3 == 1, np.inf, 1.5), (B, 1)) * np.ones((B, m))
    x_l = mb["x_l"].copy(); x_l[:, ::3] = -np.inf
    x_u = mb["x_u"].copy(); x_u[:, 1::4] = np.inf
    bs = hip.BatchSparseSolver()
    assert bs.setup(mb["P_pattern"], mb["P_values"], mb["c"], mb["A_pattern"], mb["A_values"], mb["b"], G_pattern, Gv, h_l, h_u, x_l, x_u)
    assert bs.solve() == B
    x, zl, zu = bs.result("x"), bs.result("z_l"), bs.result("z_u")
    for i in range(B):
        P, c, A, b, _, _, _, _, _ = mpc_instance(mb, i)
        G = G_pattern.copy(); G.data = Gv[i].copy()
        st, it, obj, ref = _oracle_solve(orc, (P, c, A, b, sp.csc_matrix(G), h_l[i], h_u[i], x_l[i], x_u[i]))
        assert st == 1 and bs.info(i).status == 1
        assert bs.info(i).iter == it
        assert np.abs(x[i] - ref["x"]).max() <= 1e-7 * (1 + np.abs(ref["x"]).max())
        assert np.abs(zl[i] - ref["z_l"]).max() <= 1e-6 * (1 + np.abs(ref["z_l"]).max())
        assert np.abs(zu[i] - ref["z_u"]).max() <= 1e-6 * (1 + np.abs(ref["z_u"]).max())


def test_batch_update_vectors_matches_oracle_update(hip, orc):
    """pq_batch_update: new c / b / box bounds for every instance, Ruiz scaling of the setup reused -- the reference's update() without
    a matrix argument (solver.hpp:218-308), per instance"""
    B = 12
    mb = mpc_batch(B, seed=2000)
    bs, solved = _run_batch(hip, mb)
    assert solved == B
    rng = np.random.default_rng(5)
    c2 = mb["c"] + 0.05 * rng.standard_normal(mb["c"].shape)
    b2 = mb["b"].copy(); b2[:, :2] += 0.1 * rng.standard_normal((B, 2))   # the rows that carry x0 (and whatever the shuffle put there)
    xl2 = mb["x_l"] * 1.1; xu2 = mb["x_u"] * 1.1                            # same finite pattern, wider box
    assert bs.update(c=c2, b=b2, x_l=xl2, x_u=xu2)
    assert bs.solve() == B
    x = bs.result("x")
    for i in range(B):
        so = orc.Solver(); so.settings.kkt_solver = orc.SPARSE_MULTISTAGE
        assert so.setup(*mpc_instance(mb, i), sparse=True)
        so.solve()
        assert so.update(c=c2[i], b=b2[i], x_l=xl2[i], x_u=xu2[i])
        assert so.solve() == 1
        info = bs.info(i)
        assert info.status == 1 and abs(info.iter - so.info.iter) <= 1, (i, info.iter, so.info.iter)
        assert abs(info.primal_obj - so.info.primal_obj) <= 1e-7 * (1 + abs(so.info.primal_obj))
        assert np.abs(x[i] - so.result()["x"]).max() <= 1e-6 * (1 + np.abs(so.result()["x"]).max())
    # a changed set of finite bounds is refused
    bad = xu2.copy(); bad[0, 0] = np.inf
    with pytest.raises(RuntimeError):
        bs.update(x_u=bad)


@pytest.mark.parametrize("reuse", [0, 1])
def test_batch_update_matrices_equals_single_qp_update(hip, reuse):
    """pq_batch_update_data: new P / A values (same patterns) and vectors for every instance -- unscale, assign and a fresh (or reused) Ruiz
    equilibration per instance on the device (solver.hpp:218-308, sparse/preconditioner.hpp:65-258).  The single-QP SparseSolver::update runs
    the same kernels on one instance, so iteration counts must agree exactly and the solutions to rounding."""
    B = 6
    mb = mpc_batch(B, T=12, nx=3, nu=2, seed=91)
    bs = hip.BatchSparseSolver()
    bs.settings.preconditioner_reuse_on_update = reuse
    assert bs.setup(mb["P_pattern"], mb["P_values"], mb["c"], mb["A_pattern"], mb["A_values"], mb["b"], x_l=mb["x_l"], x_u=mb["x_u"])
    assert bs.solve() == B
    rng = np.random.default_rng(17)
    P2 = mb["P_values"] * (1.0 + 0.5 * rng.random(mb["P_values"].shape))      # diagonal-dominant cost stays convex
    A2 = mb["A_values"] * (1.0 + 0.02 * rng.standard_normal(mb["A_values"].shape))
    c2 = mb["c"] + 0.1 * rng.standard_normal(mb["c"].shape)
    xu2 = mb["x_u"] * 1.2
    assert bs.update_data(P_values=P2, A_values=A2, c=c2, x_u=xu2)
    assert bs.solve() == B
    x, y = bs.result("x"), bs.result("y")
    Pp, Ap = mb["P_pattern"].tocsc(), mb["A_pattern"].tocsc()
    Pp.sort_indices(); Ap.sort_indices()
    import scipy.sparse as sp
    for i in range(B):
        s = hip.SparseSolver(); s.settings.kkt_solver = hip.SPARSE_MULTISTAGE
        s.settings.preconditioner_reuse_on_update = reuse
        assert s.setup(*mpc_instance(mb, i))
        assert s.solve() == 1
        Pi = sp.csc_matrix((P2[i], Pp.indices, Pp.indptr), shape=Pp.shape)
        Ai = sp.csc_matrix((A2[i], Ap.indices, Ap.indptr), shape=Ap.shape)
        assert s.update(P=Pi, A=Ai, c=c2[i], x_u=xu2[i])
        assert s.solve() == 1
        assert s.info.iter == bs.info(i).iter, (i, s.info.iter, bs.info(i).iter)
        assert np.abs(s.result()["x"] - x[i]).max() <= 1e-8 * (1 + np.abs(x[i]).max())
        assert np.abs(s.result()["y"] - y[i]).max() <= 1e-7 * (1 + np.abs(y[i]).max())
    # and the matrices really changed the problem
    bs0 = hip.BatchSparseSolver()
    assert bs0.setup(mb["P_pattern"], mb["P_values"], mb["c"], mb["A_pattern"], mb["A_values"], mb["b"], x_l=mb["x_l"], x_u=mb["x_u"])
    assert bs0.solve() == B
    assert np.abs(bs0.result("x") - x).max() > 1e-3


def test_start_order_does_not_change_results(hip):
    """pq_batch_set_start_order: the second solve of a handle starts the instances that took longest first; every instance's result is what it is in index order"""
    B = 300
    mb = mpc_batch(B, seed=555)
    bs, solved = _run_batch(hip, mb)
    assert solved == B
    x1, it1 = bs.result("x").copy(), np.asarray(bs.iterations()).copy()
    assert bs.solve() == B  # started longest first
    assert np.array_equal(bs.result("x"), x1) and np.array_equal(np.asarray(bs.iterations()), it1)
    bs.set_start_order(False)
    assert bs.solve() == B  # index order again
    assert np.array_equal(bs.result("x"), x1) and np.array_equal(np.asarray(bs.iterations()), it1)


# (nx, nu, T): chains whose uniform run eliminates w = nx + nu columns with u = nx coupling rows -- w = 2 .. 6, u = 1 .. 4: every width of the register-carried
# substitution (msdev::solve_chain_wave_reg) and of the one-lane-per-row factorisation of the small-batch kernel variants (msdev::factor_chain_rows, round 6; needs
# (2 w)^2 <= n doubles of staging: T is chosen so that it holds)
CHAIN_SHAPES = [(1, 1, 24), (1, 2, 20), (2, 1, 40), (2, 2, 20), (3, 1, 20), (2, 3, 24), (3, 2, 24), (2, 4, 26),
                # (fronts of more than 64 entries: the resident / staged modes of the kernel, no register-carried chain)
                (3, 3, 26), (4, 2, 26)]


@pytest.mark.parametrize("shape", CHAIN_SHAPES)
@pytest.mark.parametrize("B", [5, 3000])
def test_chain_shapes_match_the_oracle_in_both_kernel_variants(hip, orc, shape, B):
    """B = 5: the variant for a few instances per compute unit (one lane per row of a front in the uniform run, two substitution stages per loop trip); B = 3000: a
    variant for full compute units (one lane per entry).  Every 97th instance against the oracle alone; all instances solved and the two batches' common instances
    bitwise equal (an instance does not depend on the batch it is in, nor on the kernel variant: the arithmetic is the same)."""
    nx, nu, T = shape
    mb = mpc_batch(B, T=T, nx=nx, nu=nu, seed=300 + 10 * nx + nu)
    bs, solved = _run_batch(hip, mb)
    assert solved == B
    x, y = bs.result("x"), bs.result("y")
    for i in range(0, B, 97):
        st, it, obj, ref = _oracle_solve(orc, mpc_instance(mb, i))
        info = bs.info(i)
        assert info.status == st == 1 and info.iter == it, (shape, i, info.status, st, info.iter, it)
        assert np.abs(x[i] - ref["x"]).max() <= 1e-7 * (1 + np.abs(ref["x"]).max())
        assert np.abs(y[i] - ref["y"]).max() <= 1e-6 * (1 + np.abs(ref["y"]).max())
    if B > 5:
        sub = {k: (v[:5] if isinstance(v, np.ndarray) and v.ndim == 2 else v) for k, v in mb.items()}
        bs5, solved5 = _run_batch(hip, sub)
        assert solved5 == 5
        for f in ("x", "y", "z_bl", "z_bu", "s_bl", "s_bu"):
            assert np.array_equal(bs5.result(f), bs.result(f)[:5]), (shape, f)
        assert [bs5.info(i).iter for i in range(5)] == [bs.info(i).iter for i in range(5)]
