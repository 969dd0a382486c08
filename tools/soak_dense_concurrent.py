#!/usr/bin/env python3
"""Two KKTSystem handles (one a clone of the other) driven from two host threads on their own streams: the fused factorisation launches and the
persistent sweeps of both interleave on the device.  Every result must equal the single-threaded one bit for bit; a deadlock between the
workgroups of two launches would show as a timeout.   timeout 300 python tools/soak_dense_concurrent.py"""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401

import piqp_amd
from qp_gen import dense_strongly_convex_qp, random_vars

for n, m, reps in ((4096, 4096, 200), (1500, 1200, 600)):
    q = dense_strongly_convex_qp(n, 0, m, seed=5, double_sided=True, exact_shift=False)
    k1 = piqp_amd.KKTSystem(piqp_amd.Data(**q), piqp_amd.default_settings(kkt_solver=0))
    k2 = k1.clone()
    rng = np.random.default_rng(0)
    sv = random_vars(n, 0, m, rng, positive=True); rv = random_vars(n, 0, m, rng)
    out = {}

    def work(tag, k):
        state = {kk: torch.from_numpy(v).cuda() for kk, v in sv.items()}
        rhs = {kk: torch.from_numpy(v).cuda() for kk, v in rv.items()}
        lhs = {kk: torch.zeros_like(v) for kk, v in rhs.items()}
        ref = None
        for it in range(reps):
            assert k.update_scalings_and_factor(False, 1e-6, 1e-4, state)
            k.solve(rhs, lhs)
            if it % 20 == 0 or it == reps - 1:
                x = lhs["x"].cpu().numpy().copy()
                if ref is None:
                    ref = x
                assert np.array_equal(x, ref), (tag, it)
        out[tag] = ref

    work("single", k1)
    t1 = threading.Thread(target=work, args=("a", k1)); t2 = threading.Thread(target=work, args=("b", k2))
    t1.start(); t2.start(); t1.join(); t2.join()
    assert np.array_equal(out["a"], out["single"]) and np.array_equal(out["b"], out["single"])
    print(f"n={n}: 2 x {reps} concurrent steps ok", flush=True)
print("concurrent soak ok")
