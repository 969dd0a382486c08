"""Ruiz equilibration on the device (piqp_amd/csrc/ruiz_kernels.hip) against the host restatement of dense/preconditioner.hpp:62-258 and
sparse/preconditioner.hpp:65-290 kept in solver.cpp.  Inf-norms are max-reductions, every product keeps the host's order and the one sum is
accumulated sequentially, so the two must agree bit for bit: whole solves (setup, update with new matrices and a fresh equilibration,
update that reuses the scaling, a bounds update that disables a row of G, a clone) of dense and sparse fixtures are run once per variant in
separate processes (PIQP_AMD_DEBUG is parsed once per process) and every result array is compared with array_equal."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "workers", "ruiz_variant.py")


def _run(tmp_path, name, env_extra):
    out = str(tmp_path / (name + ".npz"))
    env = {k: v for k, v in os.environ.items() if not k.startswith("PIQP_AMD_")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, WORKER, out], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, name + ": " + r.stderr[-3000:]
    return dict(np.load(out))


def test_device_equilibration_is_bitwise_the_host_routine(tmp_path):
    dev = _run(tmp_path, "device", {})
    host = _run(tmp_path, "host", {"PIQP_AMD_DEBUG": "host_ruiz"})
    assert sorted(dev) == sorted(host) and len(dev) > 100
    solved = [k for k in dev if k.endswith("_status") and dev[k][0] == 1]
    assert len(solved) >= 0.8 * len([k for k in dev if k.endswith("_status")])
    for key in host:
        assert np.array_equal(dev[key], host[key], equal_nan=True), f"{key}: device {dev[key][:4]} vs host {host[key][:4]}"
