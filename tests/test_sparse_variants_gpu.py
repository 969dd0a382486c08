"""The sparse multifrontal backend picks its device schedule from the shape of the assembly tree (full or packed fronts in the LDS subtree
walk, chain walks or one supernode per workgroup in the persistent top, a finer subtree partition and chain walks for the substitution).
None of that may change a single bit of the result: the order of the floating-point operations of every front is fixed by the tree.  Each
variant runs in its own process (the one debugging variable PIQP_AMD_DEBUG is parsed once per process) on the same seeded systems and the solutions are compared
bitwise."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "workers", "sparse_variant.py")

VARIANTS = {
    "default": {},
    "packed_fronts_forced": {"PIQP_AMD_DEBUG": "subtree_packed=1"},
    "full_fronts_forced": {"PIQP_AMD_DEBUG": "subtree_packed=0"},
    "top_one_supernode_per_workgroup": {"PIQP_AMD_DEBUG": "top_no_walks"},
    "top_level_launches": {"PIQP_AMD_DEBUG": "top_levels"},
    "solve_on_factor_partition": {"PIQP_AMD_DEBUG": "solve_sub_cols=0"},
    "solve_without_chain_walks": {"PIQP_AMD_DEBUG": "solve_no_chains"},
    "solve_level_launches": {"PIQP_AMD_DEBUG": "top_levels_solve"},
    "small_subtrees": {"PIQP_AMD_DEBUG": "sub_cols=48,solve_sub_cols=8"},
    "one_launch_per_level": {"PIQP_AMD_DEBUG": "no_level_runs"},  # no merged runs of chain levels in the level-scheduled substitution
    "front_updates_whole_tiles": {"PIQP_AMD_DEBUG": "front_updates_whole_tiles"},  # the generic tile kernel also where the top of the tree takes half tiles
    "solve_level_two_launches": {"PIQP_AMD_DEBUG": "solve_level_two_launches"},  # substitution: single-wave and wide fronts of a mixed level in two launches instead of one
    "front_factor_256": {"PIQP_AMD_DEBUG": "front_factor_256"},  # the one-workgroup launch of the levels with panel fronts on four waves instead of eight
    "front_joint_updates": {"PIQP_AMD_DEBUG": "front_joint_updates"},  # big fronts' and panel fronts' trailing updates in one launch behind the join of the two streams
    "front_no_step": {"PIQP_AMD_DEBUG": "front_no_step"},  # trailing updates of the top levels in their own launch instead of inside the panel step's
    "front_no_follow": {"PIQP_AMD_DEBUG": "front_no_follow"},  # diagonal block and panel rows of the big fronts in two launches instead of one with in-launch hand-over
    "extend_add_grid64": {"PIQP_AMD_DEBUG": "extend_add_grid64"},  # the children's update matrices merged on 64 workgroups per front at every level
    "zero_fill_then_first_child": {"PIQP_AMD_DEBUG": "no_fused_first_child"},  # multi-workgroup fronts zero-filled in advance instead of written together with their first child
    "update_matrices_in_one_pass": {"PIQP_AMD_DEBUG": "multi_update"},  # the update matrix of a multi-panel front updated in one pass after the last panel instead of once per panel (opt-in: slower)
    "one_stream": {"PIQP_AMD_DEBUG": "no_fork"},  # the factorisation's second stream off (big fronts' diagonal blocks next to the one-workgroup fronts)
}


def _run(tmp_path, name, env_extra, mode=None):
    out = str(tmp_path / (name + ".npz"))
    env = dict(os.environ)
    for k in list(env):
        if k.startswith("PIQP_AMD_"):
            env.pop(k)
    env.update(env_extra)
    r = subprocess.run([sys.executable, WORKER, out] + ([mode] if mode else []), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, name + ": " + r.stderr[-2000:]
    return dict(np.load(out))


def test_schedule_variants_are_bitwise_identical(tmp_path):
    ref = _run(tmp_path, "default", VARIANTS["default"])
    assert all(np.isfinite(v).all() for v in ref.values())
    assert any(np.abs(v).max() > 0 for v in ref.values())
    for name, env in VARIANTS.items():
        if name == "default":
            continue
        got = _run(tmp_path, name, env)
        assert sorted(got) == sorted(ref)
        for key in ref:
            assert np.array_equal(got[key], ref[key]), f"{name}: {key} differs from the default schedule (max |d| = {np.abs(got[key] - ref[key]).max():.3e})"


def test_huge_fronts_substitution_agrees_with_the_per_pivot_routines(tmp_path):
    """fronts of 1024 rows and more hand their update rows to k_front_fwd_rows / k_front_bwd_cols (several workgroups per front); the solution meets the
    KKT residual bar of every other path and agrees with the per-pivot routines (PIQP_AMD_DEBUG=no_wide_solve: one workgroup per front, front_fwd / front_bwd)
    -- not bitwise: the backward column sums are formed chunk by chunk"""
    ref = _run(tmp_path, "huge_default", {}, "huge")
    assert float(ref["rel_residual"][0]) <= 1e-10, ref["rel_residual"]
    zf = _run(tmp_path, "huge_zero_fill", {"PIQP_AMD_DEBUG": "no_fused_first_child"}, "huge")  # the factorisation's assembly variant: bitwise
    for key in ref:
        assert np.array_equal(zf[key], ref[key]), key
    pp = _run(tmp_path, "huge_one_pass", {"PIQP_AMD_DEBUG": "multi_update"}, "huge")  # update matrices in one pass after the last panel (opt-in): bitwise
    for key in ref:
        assert np.array_equal(pp[key], ref[key]), key
    got = _run(tmp_path, "huge_per_pivot", {"PIQP_AMD_DEBUG": "no_wide_solve"}, "huge")
    assert float(got["rel_residual"][0]) <= 1e-10, got["rel_residual"]
    for key in ref:
        if key.startswith("wide_"):
            scale = max(1.0, float(np.abs(ref[key]).max()))
            assert np.abs(got[key] - ref[key]).max() <= 1e-8 * scale, (key, float(np.abs(got[key] - ref[key]).max()), scale)
