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


def mpc_batch(batch, T=40, nx=2, nu=1, seed=1000, shuffle_rows=False):
    """BASELINE configs[3] (C4): `batch` independent linear-MPC QPs of identical structure.
    Variables [x_0,u_0,...,x_{T-1},u_{T-1}] (n = T(nx+nu)); p = T*nx equality rows (x_0 = x_init and
    x_{k+1} = A_d x_k + B_d u_k); box bounds on every variable; Q, R diagonal positive definite; no general
    inequalities.  Per instance: own (A_d, B_d) perturbation, Q, R, x_init (seed + instance).
    Returns shared scipy patterns and stacked value arrays in the patterns' sorted-CSC order."""
    import scipy.sparse as sp
    nz = nx + nu
    n, p = T * nz, T * nx
    rows, cols = [], []
    for i in range(nx):                       # x_0 = x_init
        rows.append(i); cols.append(i)
    for k in range(T - 1):
        for i in range(nx):
            r = nx + k * nx + i
            for j in range(nz):
                rows.append(r); cols.append(k * nz + j)
            rows.append(r); cols.append((k + 1) * nz + i)
    rows, cols = np.array(rows), np.array(cols)
    perm = np.arange(p)
    if shuffle_rows:
        perm = np.random.default_rng(seed - 1).permutation(p)
        rows = perm[rows]
    order = np.lexsort((rows, cols))          # CSC order: by column, then row
    A_pattern = sp.csc_matrix((np.ones(len(rows)), (rows, cols)), shape=(p, n))
    A_pattern.sort_indices()
    P_pattern = sp.identity(n, format="csc")
    A_base = np.eye(nx) + 0.1 * np.random.default_rng(seed - 2).standard_normal((nx, nx))
    A_base *= 0.95 / max(1.0, np.abs(np.linalg.eigvals(A_base)).max())   # (marginally) stable: every instance stays feasible
    B_base = np.random.default_rng(seed - 3).standard_normal((nx, nu))
    Pv = np.zeros((batch, n)); Av = np.zeros((batch, len(rows))); c = np.zeros((batch, n)); b = np.zeros((batch, p))
    xl = np.zeros((batch, n)); xu = np.zeros((batch, n))
    for inst in range(batch):
        rng = np.random.default_rng(seed + inst)
        Ad = A_base + 0.02 * rng.standard_normal((nx, nx)); Bd = B_base + 0.05 * rng.standard_normal((nx, nu))
        q = rng.uniform(0.5, 2.0, nx); r = rng.uniform(0.05, 0.5, nu)
        Pv[inst] = np.tile(np.concatenate([q, r]), T)
        c[inst] = 0.1 * rng.standard_normal(n)
        vals = [1.0] * nx
        for k in range(T - 1):
            for i in range(nx):
                vals.extend(list(Ad[i]) + list(Bd[i]) + [-1.0])
        Av[inst] = np.array(vals)[order]
        bi = np.zeros(p); bi[:nx] = rng.uniform(-0.8, 0.8, nx)
        b[inst] = bi if not shuffle_rows else _scatter(bi, perm)
        lim = np.tile(np.concatenate([np.full(nx, 3.0), np.full(nu, 0.3)]), T)
        xl[inst], xu[inst] = -lim, lim
    return dict(P_pattern=P_pattern, P_values=Pv, c=c, A_pattern=A_pattern, A_values=Av, b=b, x_l=xl, x_u=xu, n=n, p=p)


def _scatter(v, perm):
    out = np.zeros_like(v)
    out[perm] = v
    return out


def mpc_instance(mb, i):
    """instance i of mpc_batch as scipy matrices / vectors (for the single-QP solvers)"""
    import scipy.sparse as sp
    P = mb["P_pattern"].copy(); P.data = mb["P_values"][i].copy()
    A = mb["A_pattern"].copy(); A.data = mb["A_values"][i].copy()
    return (sp.csc_matrix(P), mb["c"][i], sp.csc_matrix(A), mb["b"][i], None, None, None, mb["x_l"][i], mb["x_u"][i])


def mpc_chain(nx, nu, T, seed):
    """one linear-MPC QP as a long block-tridiagonal chain (BASELINE configs[4] recipe): variables [x_0,u_0,...,x_{T-1},u_{T-1},x_T],
    p = T*nx dynamics rows x_{k+1} = A_d x_k + B_d u_k, diagonal P, box bounds on every variable; returns the nine setup arguments"""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    nz = nx + nu
    n = T * nz + nx
    Ad = np.eye(nx) + 0.1 * rng.standard_normal((nx, nx)); Bd = rng.standard_normal((nx, nu))
    t = np.arange(T)
    i = np.arange(nx)
    rows_blk = (t[:, None, None] * nx + i[None, :, None])                       # T x nx x 1
    cols_dyn = t[:, None, None] * nz + np.arange(nz)[None, None, :]              # T x 1 x nz
    rows = np.concatenate([np.broadcast_to(rows_blk, (T, nx, nz)).ravel(), (t[:, None] * nx + i[None, :]).ravel()])
    cols = np.concatenate([np.broadcast_to(cols_dyn, (T, nx, nz)).ravel(), ((t[:, None] + 1) * nz + i[None, :]).ravel()])
    vals = np.concatenate([np.broadcast_to(np.hstack([Ad, Bd])[None, :, :], (T, nx, nz)).ravel(), -np.ones(T * nx)])
    p = T * nx
    A = sp.csc_matrix((vals, (rows, cols)), shape=(p, n))
    P = sp.diags(rng.uniform(0.5, 2.0, n), format="csc")
    return (P, rng.standard_normal(n), A, np.zeros(p), None, None, None, -np.ones(n), np.ones(n))


def c3_problem(n=50000, p=20000, m=30000, seed=44, spread=40, row_nnz=5):
    """BASELINE configs[2]: sparse QP, P banded upper-tri (~3 nnz/col + diagonal), A and G rows with `row_nnz` nnz each
    inside a window of `spread` variables; N = n + p + m = 100 000.  Defaults: the banded recipe of rounds 1-3 (nnz(upper K) = 4.9e5, fronts <= 92);
    spread = 1500, row_nnz = 10: the wider variant of round 4 (nnz(upper K) ~ 0.95e6, the BASELINE figure; constraint rows couple variables 1500 apart)"""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    P = sp.diags([rng.uniform(1, 2, n), rng.uniform(-0.3, 0.3, n - 1), rng.uniform(-0.2, 0.2, n - 2), rng.uniform(-0.1, 0.1, n - 3)], [0, 1, 2, 3], format="csc")

    def rows(k):
        cols = (rng.integers(0, n - spread, k)[:, None] + rng.choice(spread, (k, row_nnz), replace=True)).ravel()
        M = sp.csc_matrix((rng.standard_normal(row_nnz * k), (np.repeat(np.arange(k), row_nnz), cols)), shape=(k, n))
        M.sum_duplicates()
        return M
    A, G = rows(p), rows(m)
    return (P, rng.standard_normal(n), A, np.zeros(p), G, -np.ones(m), np.ones(m), None, None)
