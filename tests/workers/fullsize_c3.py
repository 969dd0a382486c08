#!/usr/bin/env python3
"""Worker of tests/test_fullsize_gpu.py::test_c3_orderings_agree: factor + solve of the full-size C3 system (n = 50 000) with the
fill-reducing ordering selected by PIQP_AMD_ORDERING in the environment (read once per process by the library).
Prints one JSON line: residual, symbolic figures, and the file holding the solution x."""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch  # noqa: F401
    import piqp_amd as hip
    from qp_gen import c3_problem, random_vars
    a = c3_problem()
    n, p, m = a[0].shape[0], a[2].shape[0], a[4].shape[0]
    k = hip.KKTSystem(hip.SparseData(*a), hip.default_settings(kkt_solver=hip.SPARSE_LDLT_MULTIFRONTAL))
    rng = np.random.default_rng(0)
    state = random_vars(n, p, m, rng, positive=True)
    rhs = random_vars(n, p, m, rng)
    assert k.update_scalings_and_factor(False, 1e-6, 1e-4, state)
    ok, lhs = k.solve(rhs)
    assert ok
    res, nrm = k.condensed_residual()
    st = k.backend().sparse_stats()
    fd, path = tempfile.mkstemp(suffix=".npy")
    os.close(fd)
    np.save(path, np.asarray(lhs["x"]))
    print(json.dumps(dict(rel_kkt_residual=res / nrm, tree_levels=st["tree_levels"], nnz_L=st["nnz_L"], max_front=st["max_front"],
                          ordering=os.environ.get("PIQP_AMD_ORDERING", "default"), x_file=path)))


if __name__ == "__main__":
    main()
