"""The reference's DENSE Maros-Meszaros sweep (/root/reference/tests/src/dense/maros_meszaros_tests.cpp:21-51) through the dense device solver: every frozen
Maros-Meszaros problem with n <= 1000 and p + m <= 1000 (the reference's own filter, :47-51) through DenseSolver with default settings -- kkt_solver =
dense_cholesky, the only factorisation dense/kkt.hpp has (Eigen::LLT, :82) -- must end PIQP_SOLVED (:36).  72 of the 110 frozen problems qualify.

On top of the reference's contract the device is held to the oracle's solve: same status, the optimum to 1e-6, and the SAME ITERATION COUNT -- on 57 of the 72.
The other 15 are the degenerate LPs-with-a-QP-term of the set (the netlib-derived Q* problems, GOULDQP2, DUALC8): they reach the regularisation floor rho = delta =
1e-10 long before they converge, the condensed matrix P + rho I + A'A / delta + G'WG then has a condition number beyond 1e16, and EVERY factorisation of it -- the
oracle's included -- leaves relative residuals of 1e-8 .. 1e-4 in the Newton step (profiles/r06_dense_accuracy.txt: the device's residual is 0.2-5 x the oracle's on
the recorded states of these very solves).  From there on the step lengths are decided by rounding (profiles/r06_dense_mm_parity.txt, r06_ipm_gouldqp2.txt: the three
loops agree to three digits until iteration 7 of GOULDQP2 and then take steps of 0.95 / 0.13 / 0.12), and there is no reference-order dense factorisation to run instead
-- Eigen::LLT's own blocked summation order is not in the tree ("parity unpinned at the bit level").  For those 15 the device's count is pinned
to what it is, per fixture, next to the oracle's: a change of either is a change of arithmetic and must be looked at, not absorbed by a slack."""
import glob
import os

import pytest

from qp_io import GOLDEN, dense_args, load_qp

pytestmark = pytest.mark.gpu


def dense_sweep_names():
    out = []
    for f in sorted(glob.glob(os.path.join(GOLDEN, "mm_*.npz"))):
        name = os.path.basename(f)[:-4]
        q = load_qp(name)
        n = q["P"].shape[0]; p = 0 if q["A"] is None else q["A"].shape[0]; m = 0 if q["G"] is None else q["G"].shape[0]
        if n <= 1000 and p + m <= 1000:
            out.append(name)
    return out


NAMES = dense_sweep_names()
# fixture -> (device iterations, oracle iterations), device-resident interior-point loop, kkt_solver = dense_cholesky; record: profiles/r06_dense_mm_parity.txt
ROUNDING_DECIDED = {
    "mm_DUALC8": (11, 10), "mm_GOULDQP2": (27, 14), "mm_QBEACONF": (23, 21), "mm_QBORE3D": (27, 18), "mm_QBRANDY": (18, 16), "mm_QETAMACR": (28, 31),
    "mm_QFFFFF80": (66, 114), "mm_QFORPLAN": (43, 38), "mm_QGROW22": (36, 30), "mm_QRECIPE": (19, 20), "mm_QSCAGR7": (16, 15), "mm_QSCFXM2": (29, 28),
    "mm_QSCTAP1": (17, 16), "mm_QSHARE1B": (26, 28), "mm_QSHARE2B": (18, 19),
}


def test_the_sweep_is_the_reference_s():
    assert len(NAMES) == 72 and set(ROUNDING_DECIDED) <= set(NAMES)


@pytest.mark.parametrize("name", NAMES)
def test_dense_solver_meets_the_reference_sweep_and_the_oracle(hip, orc, name):
    args = dense_args(load_qp(name))
    sh, so = hip.DenseSolver(), orc.Solver()
    assert sh.setup(*args) and so.setup(*args)
    st_h, st_o = sh.solve(), so.solve()
    assert st_o == orc.SOLVED, (name, st_o)   # pins the oracle to the reference's expectation (maros_meszaros_tests.cpp:36)
    assert st_h == 1, (name, st_h)            # the reference's contract, through the device
    assert abs(sh.info.primal_obj - so.info.primal_obj) <= 1e-6 * max(1.0, abs(so.info.primal_obj)) + 10 * so.settings.eps_abs, (name, sh.info.primal_obj, so.info.primal_obj)
    if name in ROUNDING_DECIDED:
        assert (sh.info.iter, so.info.iter) == ROUNDING_DECIDED[name], (name, sh.info.iter, so.info.iter)
    else:
        assert sh.info.iter == so.info.iter, (name, sh.info.iter, so.info.iter)


@pytest.mark.parametrize("name", NAMES)
def test_dense_ldlt_backend_on_the_same_sweep(hip, orc, name):
    """kkt_solver = dense_ldlt_no_pivot (this library's extension of the dense backend; dense/kkt.hpp only calls Eigen::LLT) on the same 72 problems: SOLVED, at the optimum the
    reference's backend finds.  Round 6: the backend reports a pivot that is not positive like LLT does; before, QBEACONF, QGROW15 and QGROW22 ended MAX_ITER on
    factorisations that had carried a negative pivot on (profiles/r06_dense_mm_parity.txt)."""
    args = dense_args(load_qp(name))
    sh, so = hip.DenseSolver(), orc.Solver()
    sh.settings.kkt_solver = 16
    assert sh.setup(*args) and so.setup(*args)
    assert sh.solve() == 1 and so.solve() == orc.SOLVED
    assert abs(sh.info.primal_obj - so.info.primal_obj) <= 1e-6 * max(1.0, abs(so.info.primal_obj)) + 10 * so.settings.eps_abs, (name, sh.info.primal_obj, so.info.primal_obj)
