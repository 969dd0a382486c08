#!/usr/bin/env python3
"""Accuracy of the dense device factorisation on the recorded interior-point states of a fixture (default qp_robot_arm_sqp, whose
rho = delta = 1e-10 end game is the hardest conditioning in the test set): relative residual of the condensed KKT system (extended
precision, Ruiz-scaled matrices of the oracle) for the device backend and for the oracle on the same states.
usage: python tools/dbg_dense_accuracy.py [fixture] [kkt_solver]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
from dense_replay import replay  # noqa: E402


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "qp_robot_arm_sqp"
    ks = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    for it, rho, delta, okh, oko, rh, ro in replay(name, ks):
        print(f"state {it:2d} rho={rho:.1e} delta={delta:.1e} factor ok {int(okh)}/{int(oko)}  rel.residual device {rh:.2e}  oracle {ro:.2e}  ratio {rh / max(ro, 1e-300):.2f}")
