"""The reference's own small benchmarks as parity cases (benchmarks/src/sqp_benchmarks.cpp:16-118): the chain-mass and robot-arm SQP subproblems, per
repetition `update(all data) + solve()`, the three KKT solvers the benchmark times, and the settings it sets (robot arm: reg_lower_limit =
reg_finetune_lower_limit = 1e-8).  Device vs the CPU oracle: same status, same optimum, same iteration count -- within one iteration on the robot arm, whose
count the oracle's own three backends already disagree on (17 / 16 / 17: its last iterations run at the regularisation floor, profiles/r03_sqp_benchmarks.txt)."""
import numpy as np
import pytest

from qp_io import load_qp

pytestmark = pytest.mark.gpu

SOLVERS = [("sparse_ldlt", 1), ("sparse_ldlt_cond", 4), ("sparse_multistage", 5)]


def _args(q):
    return (q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])


def _bench_loop(solver, a, reg, reps=2, **kw):
    if reg is not None:
        solver.settings.reg_lower_limit = reg
        solver.settings.reg_finetune_lower_limit = reg
    assert solver.setup(*a, **kw)
    out = []
    for _ in range(reps):
        solver.update(*a)
        st = solver.solve()
        out.append((st, solver.info.iter, solver.info.primal_obj, np.array(solver.result()["x"])))
    return out


@pytest.mark.parametrize("label,ks", SOLVERS)
@pytest.mark.parametrize("name,reg,slack", [("qp_chain_mass_sqp", None, 0), ("qp_robot_arm_sqp", 1e-8, 1)])
def test_reference_sqp_benchmark_loop(hip, orc, name, reg, slack, label, ks):
    q = load_qp(name)
    sh = hip.SparseSolver(); sh.settings.kkt_solver = ks
    so = orc.Solver(); so.settings.kkt_solver = ks
    dev = _bench_loop(sh, _args(q), reg)
    ref = _bench_loop(so, _args(q), reg, sparse=True)
    for (st_h, it_h, obj_h, x_h), (st_o, it_o, obj_o, x_o) in zip(dev, ref):
        assert st_h == st_o == 1
        assert abs(it_h - it_o) <= slack, (it_h, it_o)
        assert abs(obj_h - obj_o) <= 1e-6 * (1 + abs(obj_o))
        # (the robot arm's optimum is flat -- objective -6.5e-7, the oracle's own backends differ by several percent in single entries of x --: objective only)
        if slack == 0:
            assert np.abs(x_h - x_o).max() <= 1e-5 * (1 + np.abs(x_o).max())
