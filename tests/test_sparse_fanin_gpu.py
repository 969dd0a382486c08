"""Bounded fan-in of the assembly tree (sparse_symbolic.cpp::insert_accumulators).  An "arrow" QP -- diagonal P, five dense equality rows -- has
thousands of single-column leaves under one small root: without a bound the root's extend-add and the forward substitution's gather are serial
loops over all of them (Maros-Meszaros BOYD1: 67 ms per factorisation).  With accumulator supernodes (no pivot columns, 64 children each) the same
system must give the same solution to rounding -- the partial sums only change the order in which the root receives its children --, a tree
with more supernodes and more levels, and it must not be slower."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "workers", "fanin_variant.py")


def _run(tmp_path, name, env_extra):
    out = str(tmp_path / (name + ".npz"))
    env = {k: v for k, v in os.environ.items() if not k.startswith("PIQP_AMD_")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, WORKER, out], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, name + ": " + r.stderr[-3000:]
    return dict(np.load(out))


def test_accumulator_supernodes_bound_the_fan_in(tmp_path):
    acc = _run(tmp_path, "accumulators", {})
    raw = _run(tmp_path, "raw", {"PIQP_AMD_DEBUG": "no_accumulators"})
    assert acc["rel_res"][0] <= 1e-10 and raw["rel_res"][0] <= 1e-10
    for key in ("x", "y"):
        assert np.abs(acc[key] - raw[key]).max() <= 1e-9 * (1.0 + np.abs(raw[key]).max()), key
    assert acc["supernodes"][0] > raw["supernodes"][0]      # the accumulators are supernodes of their own ...
    assert acc["levels"][0] > raw["levels"][0]              # ... one or two levels between the leaves and the root
    assert acc["ms"][0] <= 1.2 * raw["ms"][0] + 0.2         # and never the slower tree (here: several times faster)
