"""Loaders for the frozen QP fixtures in tests/golden (written by tests/golden/make_fixtures.py)."""
import json
import os

import numpy as np
import scipy.sparse as sp

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_qp(name):
    """returns dict(P, c, A, b, G, h_l, h_u, x_l, x_u) with scipy CSC matrices; absent vectors are None"""
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    n, p, m = int(d["n"]), int(d["p"]), int(d["m"])
    csc = lambda k, shape: sp.csc_matrix((d[k + "_data"], d[k + "_indices"], d[k + "_indptr"]), shape=shape)
    vec = lambda k: None if (d[k].size and np.all(np.isnan(d[k]))) else d[k].astype(np.float64)
    q = dict(P=csc("P", (n, n)), c=d["c"].astype(np.float64), A=csc("A", (p, n)) if p else None,
             b=d["b"].astype(np.float64) if p else None, G=csc("G", (m, n)) if m else None,
             h_l=vec("h_l") if m else None, h_u=vec("h_u") if m else None, x_l=vec("x_l"), x_u=vec("x_u"))
    return q


def dense_args(q):
    """(P, c, A, b, G, h_l, h_u, x_l, x_u) as dense numpy arrays / None"""
    td = lambda M: None if M is None else np.asarray(M.todense(), dtype=np.float64)
    P = td(q["P"])
    P = np.triu(P) + np.triu(P, 1).T  # the solver only reads the upper triangle (solver.hpp:182)
    return P, q["c"], td(q["A"]), q["b"], td(q["G"]), q["h_l"], q["h_u"], q["x_l"], q["x_u"]


def load_json(name):
    return json.load(open(os.path.join(GOLDEN, name)))
