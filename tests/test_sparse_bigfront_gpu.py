"""Fronts of several hundred rows (sparse_kkt.hip: BigLevels / k_front_panel path / k_front_fwd_wide, dense_kernels.hip: the *_fronts kernels) on two
synthetic trees -- a PDE grid (CONT-xxx shape: nested-dissection separators, big + panel + wide fronts on many levels) and a QP with 333 dense equality
rows (a root with three 128-column panels, the last one ragged).  The multi-workgroup path must agree with the path that sends every front through one
workgroup's pivot loop (PIQP_AMD_DEBUG=no_big: the arithmetic of the small fronts, no matrix-core kernels) to rounding, and with the oracle's up-looking
LDLt (sparse/ldlt.hpp:101-169 restated), at an early interior-point state and at rho = delta = 1e-10."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "workers", "bigfront_variant.py")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _run(tmp_path, kind, name, env_extra):
    out = str(tmp_path / (kind + "_" + name + ".npz"))
    env = {k: v for k, v in os.environ.items() if not k.startswith("PIQP_AMD_")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, WORKER, kind, out], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, name + ": " + r.stderr[-3000:]
    return dict(np.load(out))


def _rel(a, b):
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


@pytest.mark.parametrize("kind", ["grid", "dense_rows", "dense_rows_big"])
def test_big_fronts_agree_with_one_workgroup_fronts_and_oracle(tmp_path, kind):
    big = _run(tmp_path, kind, "big", {})
    one = _run(tmp_path, kind, "one", {"PIQP_AMD_DEBUG": "no_big"})
    assert big["max_front"][0] >= 250, big["max_front"]  # the case is what it claims to be
    assert big["res_a"][0] <= 1e-10 and one["res_a"][0] <= 1e-10, (big["res_a"], one["res_a"])
    for key in ("x_a", "y_a"):
        assert _rel(big[key], one[key]) < 1e-9, (kind, key, _rel(big[key], one[key]))
    # rho = delta = 1e-10: the bar is the one-workgroup path's own residual (same matrix, same elimination tree; rounding scatter allowed)
    assert big["res_b"][0] <= max(1e-10, 4.0 * one["res_b"][0]), (big["res_b"], one["res_b"])
    # the oracle on the early state
    from oracle import pyorc as orc
    from qp_gen import random_vars
    sys.path.insert(0, os.path.join(ROOT, "tests", "workers"))
    import bigfront_variant as bv
    a, n, p, m = bv.problem(kind)
    od = orc.Data.sparse(*a)
    ko = orc.KKTSystem(od, orc.Settings(kkt_solver=orc.SPARSE_LDLT))
    rng = np.random.default_rng(23)
    state = random_vars(od.n, od.p, od.m, rng, positive=True)
    rhs = random_vars(od.n, od.p, od.m, rng)
    assert ko.update_scalings_and_factor(False, 1e-6, 1e-4, state)
    ok, lo = ko.solve(rhs)
    assert ok
    assert _rel(big["x_a"], lo["x"]) < 1e-7 and _rel(big["y_a"], lo["y"]) < 1e-7, (_rel(big["x_a"], lo["x"]), _rel(big["y_a"], lo["y"]))
