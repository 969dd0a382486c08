"""Kernel variants of the batched interior-point solver that must not change a single bit: the register budget the kernel is compiled for (chosen from the
batch size), and the chain substitution carried in registers (uniform gap-free chains) against the one that goes through LDS stage by stage.  One process per
variant (PIQP_AMD_DEBUG is parsed once per process), same seeded batches, bitwise comparison of iteration counts and solutions."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "workers", "batch_variant.py")

VARIANTS = {
    "default": {},
    "chain_substitution_through_lds": {"PIQP_AMD_DEBUG": "batch_no_chain_reg"},
    "five_waves_per_simd": {"PIQP_AMD_DEBUG": "batch_wpe=5"},
    "three_waves_per_simd_lds_chain": {"PIQP_AMD_DEBUG": "batch_wpe=3,batch_no_chain_reg"},
}


def _run(tmp_path, name, env_extra):
    out = str(tmp_path / (name + ".npz"))
    env = dict(os.environ)
    for k in list(env):
        if k.startswith("PIQP_AMD_"):
            env.pop(k)
    env.update(env_extra)
    r = subprocess.run([sys.executable, WORKER, out], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, name + ": " + r.stderr[-2000:]
    return dict(np.load(out))


def test_batch_kernel_variants_are_bitwise_identical(tmp_path):
    ref = _run(tmp_path, "default", VARIANTS["default"])
    assert all(np.isfinite(v).all() for v in ref.values())
    for name, env in VARIANTS.items():
        if name == "default":
            continue
        got = _run(tmp_path, name, env)
        assert sorted(got) == sorted(ref)
        for key in ref:
            assert np.array_equal(got[key], ref[key]), f"{name}: {key} differs from the default kernel (max |d| = {np.abs(got[key].astype(float) - ref[key].astype(float)).max():.3e})"
