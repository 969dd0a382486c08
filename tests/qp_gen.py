"""Seeded synthetic QPs following the recipe of the reference's rand::dense_strongly_convex_qp
(include/piqp/utils/random_utils.hpp:131-208) with NumPy's PCG64 instead of libstdc++'s mt19937 stream
(the recipe defines configs C1/C2, the stream does not)."""
import numpy as np


def _min_eig_sym_from_upper(U):
    n = U.shape[0]
    if n > 1500:
        try:
            import torch
            if torch.cuda.is_available():
                t = torch.from_numpy(U).cuda()
                S = t + t.T - torch.diag(torch.diagonal(t))
                return float(torch.linalg.eigvalsh(S).min().item())
        except Exception:
            pass
    S = U + U.T - np.diag(np.diag(U))
    return float(np.linalg.eigvalsh(S).min())


def dense_strongly_convex_qp(dim, n_eq, n_ineq, seed=42, bounds_perc=0.5, strong_convexity_factor=1e-2, double_sided=False, exact_shift=True):
    """returns dict(P, c, A, b, G, h_l, h_u, x_l, x_u); P holds the upper triangle only (like the reference's Model)"""
    rng = np.random.default_rng(seed)
    P = np.triu(rng.standard_normal((dim, dim)), 1)
    # |lambda_min| of a symmetric N(0,1) matrix -> 2 sqrt(n) (semicircle law); exact_shift=False avoids the O(n^3) eig
    lam_min = _min_eig_sym_from_upper(P) if exact_shift else -2.2 * np.sqrt(dim)
    P[np.arange(dim), np.arange(dim)] += strong_convexity_factor + abs(lam_min)
    A = rng.standard_normal((n_eq, dim))
    G = rng.standard_normal((n_ineq, dim))
    x_sol = rng.standard_normal(dim)
    c = rng.standard_normal(dim)
    b = A @ x_sol if n_eq > 0 else np.zeros(0)
    delta_u = np.where(rng.random(n_ineq) < 0.3, rng.random(n_ineq), 0.0)
    delta_l = np.where(rng.random(n_ineq) < 0.3, rng.random(n_ineq), 0.0)
    Gx = G @ x_sol if n_ineq > 0 else np.zeros(0)
    h_l, h_u = Gx - delta_l, Gx + delta_u
    if not double_sided:
        r = rng.random(n_ineq)
        h_l = np.where(r < 0.33, -np.inf, h_l)
        h_u = np.where((r >= 0.33) & (r < 0.66), np.inf, h_u)
    x_l = np.full(dim, -np.inf)
    x_u = np.full(dim, np.inf)
    r = rng.random(dim)
    coin = rng.random(dim) < 0.5
    mag = rng.random(dim)
    lo_only = r < bounds_perc / 3
    up_only = (r >= bounds_perc / 3) & (r < bounds_perc * 2 / 3)
    both = (r >= bounds_perc * 2 / 3) & (r < bounds_perc)
    x_l[lo_only] = x_sol[lo_only] - np.where(coin[lo_only], mag[lo_only], 0.0)
    x_u[up_only] = x_sol[up_only] + np.where(coin[up_only], mag[up_only], 0.0)
    x_l[both] = x_sol[both] - np.where(coin[both], mag[both], 0.0)
    x_u[both] = x_sol[both] + np.where(~coin[both], mag[both], 0.0)
    return dict(P=P, c=c, A=A if n_eq > 0 else None, b=b if n_eq > 0 else None, G=G if n_ineq > 0 else None,
                h_l=h_l if n_ineq > 0 else None, h_u=h_u if n_ineq > 0 else None, x_l=x_l, x_u=x_u)


def random_vars(n, p, m, rng, positive=False):
    """ten random vectors; positive=True gives s, z > 0 (an interior IPM state)"""
    sizes = dict(x=n, y=p, z_l=m, z_u=m, z_bl=n, z_bu=n, s_l=m, s_u=m, s_bl=n, s_bu=n)
    v = {}
    for k, sz in sizes.items():
        if positive and k != "x" and k != "y":
            v[k] = rng.uniform(0.1, 10.0, sz) * 10.0 ** rng.uniform(-3, 1, sz)
        else:
            v[k] = rng.standard_normal(sz)
    return v
