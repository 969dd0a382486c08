// piqp_amd/csrc/solver.cpp -- host front end (reference include/piqp/solver.hpp, dense|sparse/data.hpp,
// dense|sparse/preconditioner.hpp) driving the device KKTSystem.  See solver.hpp.
#include "trace.hpp"
#include "solver.hpp"

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <thread>

namespace pq {

namespace {
constexpr double PIQP_INF = 1e30;  // fwd.hpp:54
// host-side data preparation (transposes, Ruiz passes over n x (n + p + m) dense data) runs on a few threads: chunks of columns are
// independent and every reduction below is a max (exact, order-free), so the results do not depend on the thread count
template <class F>
void parallel_for(int count, long long work_per_item, F&& f)
{
    int T = (int)std::min<long long>(16, std::max(1u, std::thread::hardware_concurrency()));
    if ((long long)count * work_per_item < (1LL << 18) || count < 2 * T) T = 1;
    if (T <= 1) { f(0, count, 0); return; }
    std::vector<std::thread> th;
    const int chunk = (count + T - 1) / T;
    for (int t = 0; t < T; ++t) {
        const int lo = t * chunk, hi = std::min(count, lo + chunk);
        if (lo >= hi) break;
        th.emplace_back([&f, lo, hi, t] { f(lo, hi, t); });
    }
    for (auto& x : th) x.join();
}
constexpr int PF_MAX_THREADS = 16;

inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline double dot(const Vec& a, const Vec& b, int n) { double s = 0.0; for (int i = 0; i < n; ++i) s += a[i] * b[i]; return s; }
inline double min_coeff(const Vec& a, int n) { double r = a[0]; for (int i = 1; i < n; ++i) r = std::min(r, a[i]); return r; }
}  // namespace

// ------------------------------------------------------------------ HostData
void HostData::resize_vectors()
{
    c.assign(n, 0.0); b.assign(p, 0.0); h_l.assign(m, 0.0); h_u.assign(m, 0.0); x_l.assign(n, 0.0); x_u.assign(n, 0.0);
    x_b_scaling.assign(n, 1.0);
    h_l_idx.assign(m, 0); h_u_idx.assign(m, 0); x_l_idx.assign(n, 0); x_u_idx.assign(n, 0);
}
// dense/data.hpp:98-119
void HostData::set_h_l(const double* v)
{
    n_h_l = 0;
    for (int i = 0; i < m; ++i) {
        if (v && v[i] > -PIQP_INF) { h_l[i] = v[i]; h_l_idx[n_h_l++] = i; }
        else h_l[i] = -PIQP_INF;
    }
}
// dense/data.hpp:121-142
void HostData::set_h_u(const double* v)
{
    n_h_u = 0;
    for (int i = 0; i < m; ++i) {
        if (v && v[i] < PIQP_INF) { h_u[i] = v[i]; h_u_idx[n_h_u++] = i; }
        else h_u[i] = PIQP_INF;
    }
}
// dense/data.hpp:144-169 (rows with no finite side: zero the row of G, pretend h = (-1, 1))
bool HostData::disable_inf_constraints(IVec* rows)
{
    bool any = false;
    for (int i = 0; i < m; ++i) {
        if (h_l[i] <= -PIQP_INF && h_u[i] >= PIQP_INF) {
            if (sparse) { for (int k = sGT.colptr[i]; k < sGT.colptr[i + 1]; ++k) sGT.val[k] = 0.0; }
            else if (!GT.empty()) std::fill(GT.begin() + (size_t)i * n, GT.begin() + (size_t)(i + 1) * n, 0.0);  // (empty: the matrix lives on the device, see `rows`)
            if (rows) rows->push_back(i);
            h_l[i] = -1.0; h_u[i] = 1.0;
            any = true;
        }
    }
    if (any) { Vec hl = h_l, hu = h_u; set_h_l(hl.data()); set_h_u(hu.data()); }
    return any;
}
// dense/data.hpp:171-207
void HostData::set_x_l(const double* v)
{
    n_x_l = 0;
    if (v) for (int i = 0; i < n; ++i) if (v[i] > -PIQP_INF) { x_l[n_x_l] = v[i]; x_l_idx[n_x_l] = i; ++n_x_l; }
}
void HostData::set_x_u(const double* v)
{
    n_x_u = 0;
    if (v) for (int i = 0; i < n; ++i) if (v[i] < PIQP_INF) { x_u[n_x_u] = v[i]; x_u_idx[n_x_u] = i; ++n_x_u; }
}

pq_dense_data HostData::dense_descriptor() const
{
    pq_dense_data d{};
    d.n = n; d.p = p; d.m = m;
    d.P_utri = P_utri.data(); d.AT = AT.data(); d.GT = GT.data();
    d.n_h_l = n_h_l; d.n_h_u = n_h_u; d.n_x_l = n_x_l; d.n_x_u = n_x_u;
    d.h_l_idx = h_l_idx.data(); d.h_u_idx = h_u_idx.data(); d.x_l_idx = x_l_idx.data(); d.x_u_idx = x_u_idx.data();
    d.x_b_scaling = x_b_scaling.data();
    d.mem = PQ_MEM_HOST;
    return d;
}
pq_sparse_data HostData::sparse_descriptor() const
{
    pq_sparse_data d{};
    d.n = n; d.p = p; d.m = m;
    d.P_colptr = sP_utri.colptr.data(); d.P_rowind = sP_utri.rowind.data(); d.P_val = sP_utri.val.data();
    d.AT_colptr = sAT.colptr.data(); d.AT_rowind = sAT.rowind.data(); d.AT_val = sAT.val.data();
    d.GT_colptr = sGT.colptr.data(); d.GT_rowind = sGT.rowind.data(); d.GT_val = sGT.val.data();
    d.n_h_l = n_h_l; d.n_h_u = n_h_u; d.n_x_l = n_x_l; d.n_x_u = n_x_u;
    d.h_l_idx = h_l_idx.data(); d.h_u_idx = h_u_idx.data(); d.x_l_idx = x_l_idx.data(); d.x_u_idx = x_u_idx.data();
    d.x_b_scaling = x_b_scaling.data();
    d.mem = PQ_MEM_HOST;
    return d;
}

static void finish_data(HostData& d, const double* c, const double* b, const double* h_l, const double* h_u, const double* x_l, const double* x_u)
{
    std::copy(c, c + d.n, d.c.begin());
    if (b && d.p) std::copy(b, b + d.p, d.b.begin());
    d.set_h_l(h_l); d.set_h_u(h_u); d.disable_inf_constraints(); d.set_x_l(x_l); d.set_x_u(x_u);
}

// MT(i, k) = M(k, i) for a rows x n column-major M: 32 x 32 tiles keep both sides of the transpose in cache
static void transpose_into(int n, const double* M, int rows, Vec& MT)
{
    MT.resize((size_t)n * rows);
    parallel_for((rows + 31) / 32, 32LL * n, [&](int lo, int hi, int) {
        for (int kb = lo * 32; kb < std::min(rows, hi * 32); kb += 32)
            for (int ib = 0; ib < n; ib += 32)
                for (int k = kb; k < std::min(rows, kb + 32); ++k)
                    for (int i = ib; i < std::min(n, ib + 32); ++i) MT[i + (size_t)k * n] = M[k + (size_t)i * rows];
    });
}

// solver.hpp:169-192 (DenseSolver): P_utri = upper(P), AT = A^T, GT = G^T
std::unique_ptr<HostData> make_dense_host_data(int n, int p, int m, const double* P, const double* c, const double* A, const double* b, const double* G, const double* h_l,
                                               const double* h_u, const double* x_l, const double* x_u)
{
    auto d = std::make_unique<HostData>();
    d->sparse = false; d->n = n; d->p = A ? p : 0; d->m = G ? m : 0;
    p = d->p; m = d->m;
    d->P_utri.assign((size_t)n * n, 0.0);
    parallel_for(n, n, [&](int lo, int hi, int) { for (int j = lo; j < hi; ++j) for (int i = 0; i <= j; ++i) d->P_utri[i + (size_t)j * n] = P[i + (size_t)j * n]; });
    if (p > 0) transpose_into(n, A, p, d->AT); else d->AT.clear();
    if (m > 0) transpose_into(n, G, m, d->GT); else d->GT.clear();
    d->resize_vectors();
    finish_data(*d, c, b, h_l, h_u, x_l, x_u);
    return d;
}

static void csc_transpose(int rows, int cols, const int* Ap, const int* Ai, const double* Ax, Csc& T)
{
    const int nnz = Ap ? Ap[cols] : 0;
    T.rows = cols; T.cols = rows;
    T.colptr.assign(rows + 1, 0); T.rowind.assign(nnz, 0); T.val.assign(nnz, 0.0);
    for (int k = 0; k < nnz; ++k) T.colptr[Ai[k] + 1]++;
    for (int i = 0; i < rows; ++i) T.colptr[i + 1] += T.colptr[i];
    IVec next(T.colptr.begin(), T.colptr.end() - 1);
    for (int j = 0; j < cols; ++j) for (int k = Ap[j]; k < Ap[j + 1]; ++k) { const int q = next[Ai[k]]++; T.rowind[q] = j; T.val[q] = Ax[k]; }
}

// solver.hpp:169-192 (SparseSolver)
std::unique_ptr<HostData> make_sparse_host_data(int n, int p, int m, const int* Pp, const int* Pi, const double* Px, const double* c, const int* Ap, const int* Ai,
                                                const double* Ax, const double* b, const int* Gp, const int* Gi, const double* Gx, const double* h_l, const double* h_u,
                                                const double* x_l, const double* x_u)
{
    auto d = std::make_unique<HostData>();
    d->sparse = true; d->n = n; d->p = Ap ? p : 0; d->m = Gp ? m : 0;
    p = d->p; m = d->m;
    Csc& U = d->sP_utri;
    U.rows = U.cols = n; U.colptr.assign(n + 1, 0);
    for (int j = 0; j < n; ++j) {
        std::vector<std::pair<int, double>> col;
        for (int k = Pp[j]; k < Pp[j + 1]; ++k) if (Pi[k] <= j) col.emplace_back(Pi[k], Px[k]);
        std::sort(col.begin(), col.end(), [](const auto& a, const auto& b2) { return a.first < b2.first; });
        for (auto& e : col) { U.rowind.push_back(e.first); U.val.push_back(e.second); }
        U.colptr[j + 1] = (int)U.rowind.size();
    }
    if (p > 0) csc_transpose(p, n, Ap, Ai, Ax, d->sAT); else { d->sAT.rows = n; d->sAT.cols = 0; d->sAT.colptr.assign(1, 0); }
    if (m > 0) csc_transpose(m, n, Gp, Gi, Gx, d->sGT); else { d->sGT.rows = n; d->sGT.cols = 0; d->sGT.colptr.assign(1, 0); }
    d->resize_vectors();
    finish_data(*d, c, b, h_l, h_u, x_l, x_u);
    return d;
}

// ------------------------------------------------------------------ Ruiz
static inline double limit_scaling(double d) { return d < 1e-4 ? 1.0 : (d > 1e4 ? 1e4 : d); }  // dense/preconditioner.hpp:513-523

void Ruiz::init(const HostData& d)
{
    n = d.n; p = d.p; m = d.m;
    c = c_inv = 1.0;
    delta.assign(n + p + m, 1.0); delta_inv.assign(n + p + m, 1.0); delta_b.assign(n, 1.0); delta_b_inv.assign(n, 1.0);
}

namespace {
// P_ij *= s_i s_j on the stored upper triangle (diagonal twice) -- dense/preconditioner.hpp:119-124, sparse/utils.hpp:172-199
void scale_P(HostData& d, const double* s)
{
    const int n = d.n;
    if (!d.sparse) {
        // column scaling first, then row scaling, per element (the order of the two products of the reference), one pass over the data
        parallel_for(n, n, [&](int lo, int hi, int) {
            for (int j = lo; j < hi; ++j) { double* col = d.P_utri.data() + (size_t)j * n; const double sj = s[j]; for (int i = 0; i <= j; ++i) col[i] = (col[i] * sj) * s[i]; }
        });
    } else {
        Csc& U = d.sP_utri;
        for (int j = 0; j < n; ++j) for (int q = U.colptr[j]; q < U.colptr[j + 1]; ++q) U.val[q] *= s[U.rowind[q]];
        for (int j = 0; j < n; ++j) for (int q = U.colptr[j]; q < U.colptr[j + 1]; ++q) U.val[q] *= s[j];
    }
}
void scale_P_scalar(HostData& d, double g)
{
    if (!d.sparse) { const int n = d.n; parallel_for(n, n, [&](int lo, int hi, int) { for (size_t q = (size_t)lo * n; q < (size_t)hi * n; ++q) d.P_utri[q] *= g; }); }
    else for (double& v : d.sP_utri.val) v *= g;
}
void scale_T(HostData& d, bool isG, const double* srow, const double* scol)
{
    const int n = d.n, cols = isG ? d.m : d.p;
    if (!d.sparse) {
        Vec& M = isG ? d.GT : d.AT;
        parallel_for(cols, n, [&](int lo, int hi, int) {
            for (int j = lo; j < hi; ++j) { double* col = M.data() + (size_t)j * n; const double sc = scol[j]; for (int i = 0; i < n; ++i) col[i] = (srow[i] * col[i]) * sc; }
        });
    } else {
        Csc& M = isG ? d.sGT : d.sAT;
        for (int j = 0; j < cols; ++j) for (int q = M.colptr[j]; q < M.colptr[j + 1]; ++q) M.val[q] *= srow[M.rowind[q]];
        for (int j = 0; j < cols; ++j) for (int q = M.colptr[j]; q < M.colptr[j + 1]; ++q) M.val[q] *= scol[j];
    }
}
// out[k] = inf-norm of column k of the symmetric matrix stored as the upper triangle of d.P_utri (dense)
void sym_col_inf_norms(const HostData& d, double* out)
{
    const int n = d.n;
    Vec part((size_t)PF_MAX_THREADS * n, 0.0);
    parallel_for(n, n, [&](int lo, int hi, int t) {
        double* loc = part.data() + (size_t)t * n;
        for (int j = lo; j < hi; ++j) {
            const double* col = d.P_utri.data() + (size_t)j * n;
            double cm = 0.0;
            for (int i = 0; i < j; ++i) { const double a = std::fabs(col[i]); cm = std::max(cm, a); loc[i] = std::max(loc[i], a); }
            loc[j] = std::max(loc[j], std::max(cm, std::fabs(col[j])));
        }
    });
    for (int k = 0; k < n; ++k) out[k] = 0.0;
    for (int t = 0; t < PF_MAX_THREADS; ++t) { const double* loc = part.data() + (size_t)t * n; for (int k = 0; k < n; ++k) out[k] = std::max(out[k], loc[k]); }
}
}  // namespace

// dense/preconditioner.hpp:62-222, sparse/preconditioner.hpp:65-250
void Ruiz::scale_data(HostData& d, bool reuse_prev_scaling, bool scale_cost, int max_iter, double epsilon)
{
    const int N = n + p + m;
    if (!reuse_prev_scaling) {
        c = 1.0;
        std::fill(delta.begin(), delta.end(), 1.0);
        std::fill(delta_b.begin(), delta_b.end(), 1.0);
        Vec& di = delta_inv;     // scratch: this iteration's scaling
        Vec& dib = delta_b_inv;
        std::fill(di.begin(), di.end(), 0.0);
        std::fill(dib.begin(), dib.end(), 0.0);
        for (int it = 0; it < max_iter; ++it) {
            double dev = 0.0;
            for (int i = 0; i < N; ++i) dev = std::max(dev, std::fabs(1.0 - di[i]));
            for (int i = 0; i < n; ++i) dev = std::max(dev, std::fabs(1.0 - dib[i]));
            if (!(dev > epsilon)) break;
            if (!d.sparse) {
                // inf-norms of the columns of [P; A; G] (P symmetric from its upper triangle) and of the rows of A, G: every matrix is walked
                // once in storage order, per-thread partial maxima are combined afterwards (max is exact, so this equals the column-by-column
                // evaluation of dense/preconditioner.hpp:88-111)
                sym_col_inf_norms(d, di.data());
                Vec part((size_t)PF_MAX_THREADS * n, 0.0);
                auto rows_and_cols = [&](const Vec& MT, int cols, double* colnorm) {
                    parallel_for(cols, n, [&](int lo, int hi, int t) {
                        double* loc = part.data() + (size_t)t * n;
                        for (int j = lo; j < hi; ++j) {
                            const double* col = MT.data() + (size_t)j * n;
                            double cm = 0.0;
                            for (int i = 0; i < n; ++i) { const double a = std::fabs(col[i]); cm = std::max(cm, a); loc[i] = std::max(loc[i], a); }
                            colnorm[j] = cm;
                        }
                    });
                };
                if (p > 0) rows_and_cols(d.AT, p, di.data() + n);
                if (m > 0) rows_and_cols(d.GT, m, di.data() + n + p);
                for (int t = 0; t < PF_MAX_THREADS; ++t) { const double* loc = part.data() + (size_t)t * n; for (int k = 0; k < n; ++k) di[k] = std::max(di[k], loc[k]); }
                for (int k = 0; k < n; ++k) di[k] = std::max(di[k], d.x_b_scaling[k]);
            } else {
                std::fill(di.begin(), di.end(), 0.0);
                const Csc& U = d.sP_utri;
                for (int j = 0; j < n; ++j) {
                    for (int q = U.colptr[j]; q < U.colptr[j + 1]; ++q) {
                        const int r = U.rowind[q]; const double a = std::fabs(U.val[q]);
                        di[j] = std::max(di[j], a);
                        if (r != j) di[r] = std::max(di[r], a);
                    }
                    di[j] = std::max(di[j], d.x_b_scaling[j]);
                }
                for (int j = 0; j < p; ++j) for (int q = d.sAT.colptr[j]; q < d.sAT.colptr[j + 1]; ++q) {
                    const int r = d.sAT.rowind[q]; const double a = std::fabs(d.sAT.val[q]);
                    di[r] = std::max(di[r], a); di[n + j] = std::max(di[n + j], a);
                }
                for (int j = 0; j < m; ++j) for (int q = d.sGT.colptr[j]; q < d.sGT.colptr[j + 1]; ++q) {
                    const int r = d.sGT.rowind[q]; const double a = std::fabs(d.sGT.val[q]);
                    di[r] = std::max(di[r], a); di[n + p + j] = std::max(di[n + p + j], a);
                }
            }
            for (int i = 0; i < n; ++i) dib[i] = d.x_b_scaling[i];
            for (int i = 0; i < N; ++i) di[i] = 1.0 / std::sqrt(limit_scaling(di[i]));
            for (int i = 0; i < n; ++i) dib[i] = 1.0 / std::sqrt(limit_scaling(dib[i]));
            scale_P(d, di.data());
            for (int i = 0; i < n; ++i) d.c[i] *= di[i];
            scale_T(d, false, di.data(), di.data() + n);
            scale_T(d, true, di.data(), di.data() + n + p);
            for (int i = 0; i < n; ++i) d.x_b_scaling[i] *= dib[i] * di[i];
            for (int i = 0; i < N; ++i) delta[i] *= di[i];
            for (int i = 0; i < n; ++i) delta_b[i] *= dib[i];
            if (scale_cost) {
                double gamma = 0.0;
                if (!d.sparse) {
                    Vec tmp(n, 0.0);
                    sym_col_inf_norms(d, tmp.data());
                    for (int k = 0; k < n; ++k) gamma += tmp[k];
                } else {
                    Vec tmp(n, 0.0);
                    const Csc& U = d.sP_utri;
                    for (int j = 0; j < n; ++j) for (int q = U.colptr[j]; q < U.colptr[j + 1]; ++q) {
                        const int r = U.rowind[q]; const double a = std::fabs(U.val[q]);
                        tmp[j] = std::max(tmp[j], a); if (r != j) tmp[r] = std::max(tmp[r], a);
                    }
                    for (int j = 0; j < n; ++j) gamma += tmp[j];
                }
                gamma /= (double)n;
                gamma = limit_scaling(gamma);
                double cinf = 0.0; for (int i = 0; i < n; ++i) cinf = std::max(cinf, std::fabs(d.c[i]));
                gamma = limit_scaling(std::max(gamma, cinf));
                gamma = 1.0 / gamma;
                scale_P_scalar(d, gamma);
                for (int i = 0; i < n; ++i) d.c[i] *= gamma;
                c *= gamma;
            }
        }
        c_inv = 1.0 / c;
        for (int i = 0; i < N; ++i) delta_inv[i] = 1.0 / delta[i];
        for (int i = 0; i < n; ++i) delta_b_inv[i] = 1.0 / delta_b[i];
    } else {
        scale_P_scalar(d, c);
        scale_P(d, delta.data());
        for (int i = 0; i < n; ++i) d.c[i] *= c * delta[i];
        scale_T(d, false, delta.data(), delta.data() + n);
        scale_T(d, true, delta.data(), delta.data() + n + p);
        for (int i = 0; i < n; ++i) d.x_b_scaling[i] *= delta_b[i] * delta[i];
    }
    scale_bounds(d);
}

// dense/preconditioner.hpp:208-221 / :247-257: b, h_l, h_u, x_l, x_u
void Ruiz::scale_bounds(HostData& d) const
{
    for (int i = 0; i < p; ++i) d.b[i] *= delta[n + i];
    for (int i = 0; i < m; ++i) { d.h_l[i] *= delta[n + p + i]; d.h_u[i] *= delta[n + p + i]; }
    for (int i = 0; i < d.n_x_l; ++i) d.x_l[i] *= delta_b[d.x_l_idx[i]];
    for (int i = 0; i < d.n_x_u; ++i) d.x_u[i] *= delta_b[d.x_u_idx[i]];
}
void Ruiz::unscale_bounds(HostData& d) const
{
    for (int i = 0; i < p; ++i) d.b[i] *= delta_inv[n + i];
    for (int i = 0; i < m; ++i) { d.h_l[i] *= delta_inv[n + p + i]; d.h_u[i] *= delta_inv[n + p + i]; }
    for (int i = 0; i < d.n_x_l; ++i) d.x_l[i] *= delta_b_inv[d.x_l_idx[i]];
    for (int i = 0; i < d.n_x_u; ++i) d.x_u[i] *= delta_b_inv[d.x_u_idx[i]];
}

// dense/preconditioner.hpp:224-258
void Ruiz::unscale_data(HostData& d)
{
    scale_P_scalar(d, c_inv);
    scale_P(d, delta_inv.data());
    for (int i = 0; i < n; ++i) d.c[i] *= c_inv * delta_inv[i];
    scale_T(d, false, delta_inv.data(), delta_inv.data() + n);
    scale_T(d, true, delta_inv.data(), delta_inv.data() + n + p);
    for (int i = 0; i < n; ++i) d.x_b_scaling[i] *= delta_b_inv[i] * delta_inv[i];
    unscale_bounds(d);
}

void Ruiz::unscale_vectors(HostData& d) const
{
    for (int i = 0; i < n; ++i) d.c[i] *= c_inv * delta_inv[i];
    unscale_bounds(d);
}
void Ruiz::scale_vectors(HostData& d) const
{
    for (int i = 0; i < n; ++i) d.c[i] *= c * delta[i];
    scale_bounds(d);
}

// ------------------------------------------------------------------ HostVars
void HostVars::resize(int n, int p, int m)
{
    x.assign(n, 0.0); y.assign(p, 0.0); z_l.assign(m, 0.0); z_u.assign(m, 0.0); z_bl.assign(n, 0.0); z_bu.assign(n, 0.0);
    s_l.assign(m, 0.0); s_u.assign(m, 0.0); s_bl.assign(n, 0.0); s_bu.assign(n, 0.0);
}
Vec& HostVars::field(int k)
{
    switch (k) { case 0: return x; case 1: return y; case 2: return z_l; case 3: return z_u; case 4: return z_bl; case 5: return z_bu;
                 case 6: return s_l; case 7: return s_u; case 8: return s_bl; default: return s_bu; }
}
static double** vars_field(pq_vars& v, int k)
{
    switch (k) { case 0: return &v.x; case 1: return &v.y; case 2: return &v.z_l; case 3: return &v.z_u; case 4: return &v.z_bl; case 5: return &v.z_bu;
                 case 6: return &v.s_l; case 7: return &v.s_u; case 8: return &v.s_bl; default: return &v.s_bu; }
}

// ------------------------------------------------------------------ Solver
Solver::Solver(int device) : device_(device) { pq_settings_default(&m_settings); }
Solver::~Solver() = default;

void Solver::stage_alloc()
{
    const int n = m_data->n, p = m_data->p, m = m_data->m;
    PQ_HIP(hipSetDevice(device_));
    dev_in_.clear(); dev_out_.clear();
    dev_in_.resize(10); dev_out_.resize(10);
    HostVars tmp; tmp.resize(n, p, m);
    for (int k = 0; k < 10; ++k) {
        const size_t sz = std::max<size_t>(tmp.field(k).size(), 1);
        dev_in_[k].alloc(sz); dev_out_[k].alloc(sz);
        dev_in_[k].zero(m_kkt_system->stream()); dev_out_[k].zero(m_kkt_system->stream());
        *vars_field(din_, k) = dev_in_[k].p;
        *vars_field(dout_, k) = dev_out_[k].p;
    }
    dxa_.alloc(std::max(n, 1)); dxb_.alloc(std::max(n, 1)); dxc_.alloc(std::max(n, 1));
    dya_.alloc(std::max(p, 1)); dyb_.alloc(std::max(p, 1)); dza_.alloc(std::max(m, 1)); dzb_.alloc(std::max(m, 1));
}

void Solver::to_device(const HostVars& h, pq_vars& d)
{
    hipStream_t st = m_kkt_system->stream();
    const int nxl = m_data->n_x_l, nxu = m_data->n_x_u;
    for (int k = 0; k < 10; ++k) {
        const Vec& v = h.field(k);
        size_t cnt = v.size();
        if (k == 4 || k == 8) cnt = nxl;  // box vectors: only the compressed head is meaningful
        if (k == 5 || k == 9) cnt = nxu;
        if (cnt) PQ_HIP(hipMemcpyAsync(*vars_field(d, k), v.data(), cnt * sizeof(double), hipMemcpyHostToDevice, st));
    }
}
void Solver::from_device(const pq_vars& d, HostVars& h)
{
    hipStream_t st = m_kkt_system->stream();
    const int nxl = m_data->n_x_l, nxu = m_data->n_x_u;
    for (int k = 0; k < 10; ++k) {
        Vec& v = h.field(k);
        size_t cnt = v.size();
        if (k == 4 || k == 8) cnt = nxl;
        if (k == 5 || k == 9) cnt = nxu;
        if (cnt) PQ_HIP(hipMemcpyAsync(v.data(), *vars_field(const_cast<pq_vars&>(d), k), cnt * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    stream_wait(st);
}

// KKTSystem::init + init_kkt_solver (kkt_system.hpp:97-132,455-497)
void Solver::make_kkt()
{
    m_kkt_system.reset();
    KKTSolverBase* backend = nullptr;
    if (!m_data->sparse) {
        if (m_settings.kkt_solver != PQ_DENSE_CHOLESKY && m_settings.kkt_solver != PQ_DENSE_LDLT_NO_PIVOT) { std::fprintf(stderr, "kkt solver not supported\n"); return; }
        pq_dense_data desc = druiz_ ? druiz_->dense_descriptor(*m_data) : m_data->dense_descriptor();
        backend = make_dense_kkt(&desc, m_settings.kkt_solver, device_);
    } else {
        pq_sparse_data desc = m_data->sparse_descriptor();
        backend = make_sparse_kkt(&desc, m_settings.kkt_solver, device_);
        if (!backend) { std::fprintf(stderr, "kkt solver not supported\n"); return; }
    }
    m_kkt_system = std::make_unique<KKTSystem>(backend, m_settings);
    const HostData& d = *m_data;
    m_kkt_system->set_bounds(d.n_h_l, d.n_h_u, d.n_x_l, d.n_x_u, d.h_l_idx.data(), d.h_u_idx.data(), d.x_l_idx.data(), d.x_u_idx.data(), d.x_b_scaling.data(), PQ_MEM_HOST);
}

// solver.hpp:151-216
bool Solver::setup(std::unique_ptr<HostData> data)
{
    const double t0 = now_s();
    m_data = std::move(data);
    const int n = m_data->n, p = m_data->p, m = m_data->m;
    // init_workspace :361-377
    m_result.resize(n, p, m); res_nr.resize(n, p, m); res.resize(n, p, m); step.resize(n, p, m); prox_vars.resize(n, p, m);
    m_info = pq_info{};
    m_info.rho = m_settings.rho_init; m_info.delta = m_settings.delta_init;
    m_preconditioner.init(*m_data);
    // dense/preconditioner.hpp:62-222 on the device (ruiz_kernels.hip); the host routine stays reachable for the bitwise comparison test
    druiz_.reset();
    if (!debug_token("host_ruiz")) {
        druiz_ = std::make_unique<DeviceRuiz>(device_, *m_data);
        if (!m_data->sparse) { druiz_->upload_dense(*m_data, PQ_KKT_UPDATE_P | PQ_KKT_UPDATE_A | PQ_KKT_UPDATE_G); release_dense_staging(); }
    }
    scale_problem(false);
    make_kkt();
    if (!m_kkt_system) { m_setup_done = false; return false; }
    stage_alloc();
    dipm_.reset();
    // the interior-point loop: vectors resident in HBM (device_ipm.hip; behind the reference-order sparse engine its dot products are summed left to right like the
    // reference's, k_seq_dots) -- or the host loop below (PIQP_AMD_HOST_IPM=1).  Behind the reference-order engine BOTH are the oracle's sequence of IEEE operations
    // (tools/trace_diff.py: every frozen fixture up to 8192 KKT rows bitwise, with either loop).
    const char* host_ipm = std::getenv("PIQP_AMD_HOST_IPM");
    const bool use_host = host_ipm && host_ipm[0] == '1';
    if (!use_host) { dipm_ = std::make_unique<DeviceIpm>(); dipm_->init(*m_data, m_preconditioner, m_kkt_system->stream()); }
    m_first_run = true; m_setup_done = true;
    m_info.setup_time = now_s() - t0;
    return true;
}

// the unscaled dense matrices are staging only once they are in HBM (the scaled truth is DeviceRuiz's)
void Solver::release_dense_staging()
{
    Vec().swap(m_data->P_utri); Vec().swap(m_data->AT); Vec().swap(m_data->GT);
}
// preconditioner.scale_data / unscale_data of solver.hpp:163,260,287
void Solver::scale_problem(bool reuse)
{
    HostData& d = *m_data;
    const bool cost = m_settings.preconditioner_scale_cost != 0;
    if (!druiz_) { m_preconditioner.scale_data(d, reuse, cost, m_settings.preconditioner_iter); return; }
    druiz_->scale(d, m_preconditioner, reuse, cost, m_settings.preconditioner_iter);
    m_preconditioner.scale_bounds(d);
}
void Solver::unscale_problem()
{
    HostData& d = *m_data;
    if (!druiz_) { m_preconditioner.unscale_data(d); return; }
    druiz_->unscale(d, m_preconditioner);
    m_preconditioner.unscale_bounds(d);
}

Solver* Solver::clone() const
{
    std::unique_ptr<Solver> s(new Solver(device_));
    s->m_settings = m_settings; s->m_info = m_info;
    s->m_first_run = m_first_run; s->m_setup_done = m_setup_done; s->m_enable_iterative_refinement = m_enable_iterative_refinement;
    if (m_data) {
        s->m_data = std::make_unique<HostData>(*m_data);
        s->m_preconditioner = m_preconditioner;
        if (druiz_) s->druiz_ = m_data->sparse ? std::make_unique<DeviceRuiz>(device_, *m_data) : druiz_->clone();
        s->m_result = m_result; s->res_nr = res_nr; s->res = res; s->step = step; s->prox_vars = prox_vars;
        if (m_kkt_system) {
            s->m_kkt_system.reset(m_kkt_system->clone()); s->stage_alloc();
            if (dipm_) { s->dipm_ = std::make_unique<DeviceIpm>(); s->dipm_->init(*s->m_data, s->m_preconditioner, s->m_kkt_system->stream()); }
        }
    }
    return s.release();
}

static void refresh_kkt(const DeviceRuiz* dr, KKTSystem& k, const HostData& d, int options)
{
    // KKTSystem::update_data, kkt_system.hpp:134-141 (+ bound lists / x_b_scaling, which live in `data` in the reference)
    if (!d.sparse) { pq_dense_data desc = dr ? dr->dense_descriptor(d) : d.dense_descriptor(); k.backend()->update_data_dense(&desc, options); }
    else { pq_sparse_data desc = d.sparse_descriptor(); k.backend()->update_data_sparse(&desc, options); }
    k.set_bounds(d.n_h_l, d.n_h_u, d.n_x_l, d.n_x_u, d.h_l_idx.data(), d.h_u_idx.data(), d.x_l_idx.data(), d.x_u_idx.data(), d.x_b_scaling.data(), PQ_MEM_HOST);
}

// update() without a matrix argument (solver.hpp:218-308 with every optional matrix empty): the reference still unscales and rescales
// P, A, G with the unchanged scaling (reuse_prev_scaling is forced, :283-285) -- the identity up to rounding, O(n (n + p + m)) host work
// and a full re-upload here.  Only the vectors go through unscale -> assign -> scale; the matrices stay as they are on host and device.
bool Solver::update_vectors_only(const double* c, const double* b, const double* h_l, const double* h_u, const double* x_l, const double* x_u, double t0)
{
    HostData& d = *m_data;
    m_preconditioner.unscale_vectors(d);
    if (c) std::copy(c, c + d.n, d.c.begin());
    if (b) std::copy(b, b + d.p, d.b.begin());
    if (h_l) d.set_h_l(h_l);
    if (h_u) d.set_h_u(h_u);
    IVec zeroed;
    const bool row_zeroed = (h_l || h_u) && d.disable_inf_constraints(&zeroed);  // a row of G without any finite bound is zeroed: that IS a matrix change
    if (row_zeroed && druiz_) druiz_->zero_G_rows(zeroed);
    if (x_l) d.set_x_l(x_l);
    if (x_u) d.set_x_u(x_u);
    m_preconditioner.scale_vectors(d);
    if (row_zeroed) { refresh_kkt(druiz_.get(), *m_kkt_system, d, PQ_KKT_UPDATE_G); }
    else m_kkt_system->set_bounds(d.n_h_l, d.n_h_u, d.n_x_l, d.n_x_u, d.h_l_idx.data(), d.h_u_idx.data(), d.x_l_idx.data(), d.x_u_idx.data(), d.x_b_scaling.data(), PQ_MEM_HOST);
    if (dipm_) dipm_->refresh_data(d, m_preconditioner);
    m_info.update_time = now_s() - t0;
    return true;
}

// solver.hpp:218-308 with the dense update_P/A/G of :311-351
bool Solver::update_dense(const double* P, const double* c, const double* A, const double* b, const double* G, const double* h_l, const double* h_u, const double* x_l,
                          const double* x_u)
{
    if (!m_setup_done) { std::fprintf(stderr, "Solver not setup yet\n"); return false; }
    const double t0 = now_s();
    HostData& d = *m_data;
    const int n = d.n, p = d.p, m = d.m;
    if (!P && !A && !G) return update_vectors_only(c, b, h_l, h_u, x_l, x_u, t0);
    unscale_problem();
    int opt = PQ_KKT_UPDATE_NONE;
    if (P) {
        d.P_utri.resize((size_t)n * n);
        parallel_for(n, n, [&](int lo, int hi, int) {
            for (int j = lo; j < hi; ++j) {
                double* col = d.P_utri.data() + (size_t)j * n;
                for (int i = 0; i <= j; ++i) col[i] = P[i + (size_t)j * n];
                for (int i = j + 1; i < n; ++i) col[i] = 0.0;
            }
        });
        opt |= PQ_KKT_UPDATE_P;
    }
    if (A) { transpose_into(n, A, p, d.AT); opt |= PQ_KKT_UPDATE_A; }
    if (G) { transpose_into(n, G, m, d.GT); opt |= PQ_KKT_UPDATE_G; }
    if (c) std::copy(c, c + n, d.c.begin());
    if (b) std::copy(b, b + p, d.b.begin());
    if (h_l) d.set_h_l(h_l);
    if (h_u) d.set_h_u(h_u);
    IVec zeroed;
    if (h_l || h_u) d.disable_inf_constraints(&zeroed);
    if (x_l) d.set_x_l(x_l);
    if (x_u) d.set_x_u(x_u);
    if (druiz_) { druiz_->upload_dense(d, opt); druiz_->zero_G_rows(zeroed); release_dense_staging(); }
    bool reuse = m_settings.preconditioner_reuse_on_update != 0;
    if (opt == PQ_KKT_UPDATE_NONE) reuse = true;
    scale_problem(reuse);
    refresh_kkt(druiz_.get(), *m_kkt_system, d, opt);
    if (dipm_) dipm_->refresh_data(d, m_preconditioner);
    m_info.update_time = now_s() - t0;
    return true;
}

// solver.hpp:218-308 with the sparse update_P/A/G of :317-358 (identical sparsity required)
bool Solver::update_sparse(const int* Pp, const int* Pi, const double* Px, const double* c, const int* Ap, const int* Ai, const double* Ax, const double* b, const int* Gp,
                           const int* Gi, const double* Gx, const double* h_l, const double* h_u, const double* x_l, const double* x_u)
{
    if (!m_setup_done) { std::fprintf(stderr, "Solver not setup yet\n"); return false; }
    const double t0 = now_s();
    HostData& d = *m_data;
    const int n = d.n, p = d.p, m = d.m;
    if (!Px && !Ax && !Gx) return update_vectors_only(c, b, h_l, h_u, x_l, x_u, t0);
    unscale_problem();
    int opt = PQ_KKT_UPDATE_NONE;
    (void)Pi;
    if (Px) {
        for (int j = 0; j < n; ++j) {
            const int have = Pp[j + 1] - Pp[j], need = d.sP_utri.colptr[j + 1] - d.sP_utri.colptr[j];
            if (have < need) { std::fprintf(stderr, "P nonzeros missmatch\n"); return false; }
            std::copy(Px + Pp[j], Px + Pp[j] + need, d.sP_utri.val.begin() + d.sP_utri.colptr[j]);
        }
        opt |= PQ_KKT_UPDATE_P;
    }
    auto retranspose = [&](const int* Mp, const int* Mi, const double* Mx, Csc& T, int rows) {
        IVec next(T.colptr.begin(), T.colptr.end() - 1);
        for (int j = 0; j < n; ++j) for (int q = Mp[j]; q < Mp[j + 1]; ++q) { const int t = next[Mi[q]]++; T.rowind[t] = j; T.val[t] = Mx[q]; }
        (void)rows;
    };
    if (Ax) { if (Ap[n] != d.sAT.nnz()) { std::fprintf(stderr, "A nonzeros missmatch\n"); return false; } retranspose(Ap, Ai, Ax, d.sAT, p); opt |= PQ_KKT_UPDATE_A; }
    if (Gx) { if (Gp[n] != d.sGT.nnz()) { std::fprintf(stderr, "G nonzeros missmatch\n"); return false; } retranspose(Gp, Gi, Gx, d.sGT, m); opt |= PQ_KKT_UPDATE_G; }
    if (c) std::copy(c, c + n, d.c.begin());
    if (b) std::copy(b, b + p, d.b.begin());
    if (h_l) d.set_h_l(h_l);
    if (h_u) d.set_h_u(h_u);
    if (h_l || h_u) d.disable_inf_constraints();
    if (x_l) d.set_x_l(x_l);
    if (x_u) d.set_x_u(x_u);
    bool reuse = m_settings.preconditioner_reuse_on_update != 0;
    if (opt == PQ_KKT_UPDATE_NONE) reuse = true;
    scale_problem(reuse);
    refresh_kkt(druiz_.get(), *m_kkt_system, d, opt);
    if (dipm_) dipm_->refresh_data(d, m_preconditioner);
    m_info.update_time = now_s() - t0;
    return true;
}

// ---- calls into the device KKTSystem ------------------------------------------------------------
bool Solver::kkt_factor()
{
    m_info.n_factor++;
    to_device(m_result, din_);
    return m_kkt_system->update_scalings_and_factor(m_enable_iterative_refinement, m_info.rho, m_info.delta, din_);
}
void Solver::kkt_solve(const HostVars& rhs, HostVars& lhs)
{
    m_info.n_solve++;
    to_device(rhs, din_);
    m_kkt_system->solve(din_, dout_);  // return value ignored by the reference too (solver.hpp:487,732,765,838)
    m_info.n_backend_solve += m_kkt_system->last_backend_solves;
    from_device(dout_, lhs);
}
void Solver::eval_P_x(double alpha, const Vec& x, Vec& z)
{
    hipStream_t st = m_kkt_system->stream();
    const int n = m_data->n;
    PQ_HIP(hipMemcpyAsync(dxa_.p, x.data(), n * sizeof(double), hipMemcpyHostToDevice, st));
    m_kkt_system->backend()->eval_P_x(alpha, dxa_.p, dxb_.p);
    PQ_HIP(hipMemcpyAsync(z.data(), dxb_.p, n * sizeof(double), hipMemcpyDeviceToHost, st));
    stream_wait(st);
}
void Solver::eval_A(double an, double at, const Vec& xn, const Vec& xt, Vec& zn, Vec& zt)
{
    hipStream_t st = m_kkt_system->stream();
    const int n = m_data->n, p = m_data->p;
    PQ_HIP(hipMemcpyAsync(dxa_.p, xn.data(), n * sizeof(double), hipMemcpyHostToDevice, st));
    if (p) PQ_HIP(hipMemcpyAsync(dya_.p, xt.data(), p * sizeof(double), hipMemcpyHostToDevice, st));
    m_kkt_system->backend()->eval_A_xn_and_AT_xt(an, at, dxa_.p, dya_.p, dyb_.p, dxb_.p);
    if (p) PQ_HIP(hipMemcpyAsync(zn.data(), dyb_.p, p * sizeof(double), hipMemcpyDeviceToHost, st));
    PQ_HIP(hipMemcpyAsync(zt.data(), dxb_.p, n * sizeof(double), hipMemcpyDeviceToHost, st));
    stream_wait(st);
}
void Solver::eval_G(double an, double at, const Vec& xn, const Vec& xt, Vec& zn, Vec& zt)
{
    hipStream_t st = m_kkt_system->stream();
    const int n = m_data->n, m = m_data->m;
    PQ_HIP(hipMemcpyAsync(dxa_.p, xn.data(), n * sizeof(double), hipMemcpyHostToDevice, st));
    if (m) PQ_HIP(hipMemcpyAsync(dza_.p, xt.data(), m * sizeof(double), hipMemcpyHostToDevice, st));
    m_kkt_system->backend()->eval_G_xn_and_GT_xt(an, at, dxa_.p, dza_.p, dzb_.p, dxb_.p);
    if (m) PQ_HIP(hipMemcpyAsync(zn.data(), dzb_.p, m * sizeof(double), hipMemcpyDeviceToHost, st));
    PQ_HIP(hipMemcpyAsync(zt.data(), dxb_.p, n * sizeof(double), hipMemcpyDeviceToHost, st));
    stream_wait(st);
}

// solver.hpp:884-891
double Solver::calculate_mu() const
{
    const HostData& d = *m_data; const HostVars& r = m_result;
    return (dot(r.s_l, r.z_l, d.m) + dot(r.s_u, r.z_u, d.m) + dot(r.s_bl, r.z_bl, d.n_x_l) + dot(r.s_bu, r.z_bu, d.n_x_u)) / (double)(d.n_h_l + d.n_h_u + d.n_x_l + d.n_x_u);
}

// solver.hpp:893-958
void Solver::calculate_step(double& alpha_s, double& alpha_z) const
{
    const HostData& d = *m_data; const HostVars& r = m_result; const HostVars& st = step;
    double as = 1.0, az = 1.0;
    for (int i = 0; i < d.m; ++i) {
        if (st.s_l[i] < 0) as = std::min(as, -r.s_l[i] / st.s_l[i]);
        if (st.s_u[i] < 0) as = std::min(as, -r.s_u[i] / st.s_u[i]);
        if (st.z_l[i] < 0) az = std::min(az, -r.z_l[i] / st.z_l[i]);
        if (st.z_u[i] < 0) az = std::min(az, -r.z_u[i] / st.z_u[i]);
    }
    for (int i = 0; i < d.n_x_l; ++i) {
        if (st.s_bl[i] < 0) as = std::min(as, -r.s_bl[i] / st.s_bl[i]);
        if (st.z_bl[i] < 0) az = std::min(az, -r.z_bl[i] / st.z_bl[i]);
    }
    for (int i = 0; i < d.n_x_u; ++i) {
        if (st.s_bu[i] < 0) as = std::min(as, -r.s_bu[i] / st.s_bu[i]);
        if (st.z_bu[i] < 0) az = std::min(az, -r.z_bu[i] / st.z_bu[i]);
    }
    alpha_s = as; alpha_z = az;
}

static double inf_scaled(const Vec& v, const double* sc, double c, int n)
{
    double r = 0.0;
    for (int i = 0; i < n; ++i) { const double a = std::fabs(v[i] * c * sc[i]); if (a > r || a != a) r = a; }
    return r;
}

// solver.hpp:1130-1164 (general rows by |.|_inf; box rows by the signed scaled value, as the reference does)
double Solver::primal_res_of(const HostVars& v) const
{
    const HostData& d = *m_data; const int n = d.n, p = d.p, m = d.m;
    const double* dinv = m_preconditioner.delta_inv.data();
    double inf = inf_scaled(v.y, dinv + n, 1.0, p);
    inf = std::max(inf, inf_scaled(v.z_l, dinv + n + p, 1.0, m));
    inf = std::max(inf, inf_scaled(v.z_u, dinv + n + p, 1.0, m));
    for (int i = 0; i < d.n_x_l; ++i) inf = std::max(inf, v.z_bl[i] * m_preconditioner.delta_b_inv[d.x_l_idx[i]]);
    for (int i = 0; i < d.n_x_u; ++i) inf = std::max(inf, v.z_bu[i] * m_preconditioner.delta_b_inv[d.x_u_idx[i]]);
    return inf;
}
// solver.hpp:1184-1196
double Solver::dual_res_of(const Vec& x) const { return inf_scaled(x, m_preconditioner.delta_inv.data(), m_preconditioner.c_inv, m_data->n); }
// solver.hpp:1166-1182
double Solver::primal_prox_inf() const
{
    const HostData& d = *m_data; const int n = d.n, p = d.p, m = d.m;
    const double ci = m_preconditioner.c_inv; const double* dl = m_preconditioner.delta.data();
    double inf = 0.0;
    for (int i = 0; i < p; ++i) inf = std::max(inf, std::fabs((prox_vars.y[i] - m_result.y[i]) * ci * dl[n + i]));
    for (int i = 0; i < m; ++i) inf = std::max(inf, std::fabs((prox_vars.z_l[i] - m_result.z_l[i]) * ci * dl[n + p + i]));
    for (int i = 0; i < m; ++i) inf = std::max(inf, std::fabs((prox_vars.z_u[i] - m_result.z_u[i]) * ci * dl[n + p + i]));
    for (int i = 0; i < d.n_x_l; ++i) inf = std::max(inf, (prox_vars.z_bl[i] - m_result.z_bl[i]) * ci * m_preconditioner.delta_b[d.x_l_idx[i]]);
    for (int i = 0; i < d.n_x_u; ++i) inf = std::max(inf, (prox_vars.z_bu[i] - m_result.z_bu[i]) * ci * m_preconditioner.delta_b[d.x_u_idx[i]]);
    return inf;
}
// solver.hpp:1198-1203
double Solver::dual_prox_inf() const
{
    double inf = 0.0;
    for (int i = 0; i < m_data->n; ++i) inf = std::max(inf, std::fabs((m_result.x[i] - prox_vars.x[i]) * m_preconditioner.delta[i]));
    return inf;
}

// solver.hpp:960-1105
void Solver::update_residuals_nr()
{
    const HostData& d = *m_data; const int n = d.n, p = d.p, m = d.m;
    HostVars& r = m_result; HostVars& nr = res_nr;
    Vec& work_x = step.x; Vec& work_z = step.z_l;
    const double ci = m_preconditioner.c_inv;
    const double* dinv = m_preconditioner.delta_inv.data();

    eval_A(-1.0, 1.0, r.x, r.y, nr.y, work_x);
    for (int i = 0; i < m; ++i) work_z[i] = r.z_u[i] - r.z_l[i];
    Vec& work_x_2 = nr.x;
    eval_G(1.0, 1.0, r.x, work_z, nr.z_l, work_x_2);
    for (int i = 0; i < m; ++i) nr.z_u[i] = -nr.z_l[i];
    for (int i = 0; i < n; ++i) work_x[i] += work_x_2[i];

    eval_P_x(-1.0, r.x, nr.x);
    double dual_rel_norm = inf_scaled(nr.x, dinv, ci, n);

    double tmp = -dot(r.x, nr.x, n);
    m_info.primal_obj = 0.5 * tmp;
    m_info.dual_obj = -0.5 * tmp;
    double dg_rel = ci * std::fabs(tmp);
    tmp = dot(d.c, r.x, n); m_info.primal_obj += tmp; dg_rel = std::max(dg_rel, ci * std::fabs(tmp));
    tmp = dot(d.b, r.y, p); m_info.dual_obj -= tmp; dg_rel = std::max(dg_rel, ci * std::fabs(tmp));
    tmp = -dot(d.h_l, r.z_l, m); m_info.dual_obj -= tmp; dg_rel = std::max(dg_rel, ci * std::fabs(tmp));
    tmp = dot(d.h_u, r.z_u, m); m_info.dual_obj -= tmp; dg_rel = std::max(dg_rel, ci * std::fabs(tmp));
    tmp = -dot(d.x_l, r.z_bl, d.n_x_l); m_info.dual_obj -= tmp; dg_rel = std::max(dg_rel, ci * std::fabs(tmp));
    tmp = dot(d.x_u, r.z_bu, d.n_x_u); m_info.dual_obj -= tmp; dg_rel = std::max(dg_rel, ci * std::fabs(tmp));

    m_info.duality_gap = std::fabs(m_info.primal_obj - m_info.dual_obj);
    m_info.primal_obj *= ci; m_info.dual_obj *= ci; m_info.duality_gap *= ci;
    m_info.duality_gap_rel = m_info.duality_gap / std::max(1.0, dg_rel);

    for (int i = 0; i < n; ++i) nr.x[i] -= d.c[i];
    dual_rel_norm = std::max(dual_rel_norm, inf_scaled(d.c, dinv, ci, n));
    for (int i = 0; i < d.n_x_l; ++i) { const int idx = d.x_l_idx[i]; work_x[idx] -= d.x_b_scaling[idx] * r.z_bl[i]; }
    for (int i = 0; i < d.n_x_u; ++i) { const int idx = d.x_u_idx[i]; work_x[idx] += d.x_b_scaling[idx] * r.z_bu[i]; }
    dual_rel_norm = std::max(dual_rel_norm, inf_scaled(work_x, dinv, ci, n));
    for (int i = 0; i < n; ++i) nr.x[i] -= work_x[i];

    double primal_rel_norm = inf_scaled(nr.y, dinv + n, 1.0, p);
    for (int i = 0; i < p; ++i) nr.y[i] += d.b[i];
    primal_rel_norm = std::max(primal_rel_norm, inf_scaled(d.b, dinv + n, 1.0, p));

    const double* dz = dinv + n + p;
    int i = 0;
    for (int ii = 0; ii < d.n_h_l; ++ii) {
        const int idx = d.h_l_idx[ii];
        while (i < idx) nr.z_l[i++] = 0.0;
        primal_rel_norm = std::max(primal_rel_norm, nr.z_l[i] * dz[i]);  // signed, like the reference (:1047)
        nr.z_l[i] += -d.h_l[i] - r.s_l[i];
        primal_rel_norm = std::max(primal_rel_norm, d.h_l[i] * dz[i]);
        primal_rel_norm = std::max(primal_rel_norm, r.s_l[i] * dz[i]);
        ++i;
    }
    while (i < m) nr.z_l[i++] = 0.0;
    i = 0;
    for (int ii = 0; ii < d.n_h_u; ++ii) {
        const int idx = d.h_u_idx[ii];
        while (i < idx) nr.z_u[i++] = 0.0;
        primal_rel_norm = std::max(primal_rel_norm, nr.z_u[i] * dz[i]);
        nr.z_u[i] += d.h_u[i] - r.s_u[i];
        primal_rel_norm = std::max(primal_rel_norm, d.h_u[i] * dz[i]);
        primal_rel_norm = std::max(primal_rel_norm, r.s_u[i] * dz[i]);
        ++i;
    }
    while (i < m) nr.z_u[i++] = 0.0;

    const double* dbi = m_preconditioner.delta_b_inv.data();
    for (i = 0; i < d.n_x_l; ++i) {
        const int idx = d.x_l_idx[i];
        nr.z_bl[i] = d.x_b_scaling[idx] * r.x[idx];
        primal_rel_norm = std::max(primal_rel_norm, nr.z_bl[i] * dbi[idx]);
        primal_rel_norm = std::max(primal_rel_norm, d.x_l[i] * dbi[idx]);
        primal_rel_norm = std::max(primal_rel_norm, r.s_bl[i] * dbi[idx]);
    }
    for (i = 0; i < d.n_x_l; ++i) nr.z_bl[i] += -d.x_l[i] - r.s_bl[i];
    for (i = 0; i < d.n_x_u; ++i) {
        const int idx = d.x_u_idx[i];
        nr.z_bu[i] = -d.x_b_scaling[idx] * r.x[idx];
        primal_rel_norm = std::max(primal_rel_norm, nr.z_bu[i] * dbi[idx]);
        primal_rel_norm = std::max(primal_rel_norm, d.x_u[i] * dbi[idx]);
        primal_rel_norm = std::max(primal_rel_norm, r.s_bu[i] * dbi[idx]);
    }
    for (i = 0; i < d.n_x_u; ++i) nr.z_bu[i] += d.x_u[i] - r.s_bu[i];

    m_info.prev_primal_res = m_info.primal_res;
    m_info.prev_dual_res = m_info.dual_res;
    m_info.primal_res = primal_res_of(nr);
    m_info.primal_res_rel = m_info.primal_res / std::max(1.0, primal_rel_norm);
    m_info.dual_res = dual_res_of(nr.x);
    m_info.dual_res_rel = m_info.dual_res / std::max(1.0, dual_rel_norm);
}

// solver.hpp:1107-1128
void Solver::update_residuals_r()
{
    const HostData& d = *m_data; const int n = d.n, p = d.p, m = d.m;
    const HostVars& r = m_result; const HostVars& nr = res_nr; const HostVars& px = prox_vars;
    const double rho = m_info.rho, delta = m_info.delta;
    for (int i = 0; i < n; ++i) res.x[i] = nr.x[i] - rho * (r.x[i] - px.x[i]);
    for (int i = 0; i < p; ++i) res.y[i] = nr.y[i] - delta * (px.y[i] - r.y[i]);
    for (int i = 0; i < m; ++i) res.z_l[i] = nr.z_l[i] - delta * (px.z_l[i] - r.z_l[i]);
    for (int i = 0; i < m; ++i) res.z_u[i] = nr.z_u[i] - delta * (px.z_u[i] - r.z_u[i]);
    for (int i = 0; i < d.n_x_l; ++i) res.z_bl[i] = nr.z_bl[i] - delta * (px.z_bl[i] - r.z_bl[i]);
    for (int i = 0; i < d.n_x_u; ++i) res.z_bu[i] = nr.z_bu[i] - delta * (px.z_bu[i] - r.z_bu[i]);
    const double primal_rel_scaling = m_info.primal_res_rel > 0 ? m_info.primal_res / m_info.primal_res_rel : 1.0;
    const double dual_rel_scaling = m_info.dual_res_rel > 0 ? m_info.dual_res / m_info.dual_res_rel : 1.0;
    m_info.primal_res_reg = primal_res_of(res);
    m_info.primal_res_reg_rel = m_info.primal_res_reg / primal_rel_scaling;
    m_info.dual_res_reg = dual_res_of(res.x);
    m_info.dual_res_reg_rel = m_info.dual_res_reg / dual_rel_scaling;
    m_info.primal_prox_inf = primal_prox_inf() * m_info.delta;
    m_info.dual_prox_inf = dual_prox_inf() * m_info.rho;
}

bool verify_settings(const pq_settings& s)
{
    // settings.hpp:84-106
    return s.rho_init > 0 && s.delta_init > 0 && s.eps_abs > 0 && s.eps_rel >= 0 && s.eps_duality_gap_abs > 0 && s.eps_duality_gap_rel >= 0 &&
           s.infeasibility_threshold >= 0 && s.reg_lower_limit > 0 && s.reg_finetune_primal_update_threshold >= 0 && s.reg_finetune_dual_update_threshold >= 0 &&
           s.max_iter > 0 && s.max_factor_retires > 0 && s.preconditioner_iter >= 0 && s.tau > 0 && s.tau <= 1 && s.iterative_refinement_eps_abs > 0 &&
           s.iterative_refinement_eps_rel >= 0 && s.iterative_refinement_max_iter >= 0 && s.iterative_refinement_min_improvement_rate >= 1.0 &&
           s.iterative_refinement_static_regularization_eps > 0 && s.iterative_refinement_static_regularization_rel >= 0;
}

// solver.hpp:379-882
int Solver::solve_impl()
{
    PQ_ZONE("piqp_amd::Solver::solve_impl");
    pq_info& info = m_info;
    const pq_settings& set = m_settings;
    if (!m_setup_done) { std::fprintf(stderr, "Solver not setup yet\n"); info.status = PQ_UNSOLVED; return info.status; }
    if (!verify_settings(set)) { info.status = PQ_INVALID_SETTINGS; return info.status; }
    m_kkt_system->set_settings(set);
    if (dipm_) return dipm_->solve(*m_kkt_system, set, m_preconditioner, info, trace_, trace_max_, &trace_rows_);
    const HostData& d = *m_data;
    const int n = d.n, p = d.p, m = d.m;
    HostVars& r = m_result; HostVars& px = prox_vars;
    double t0;

    info.kkt_factor_time = 0; info.kkt_solve_time = 0;
    info.n_factor = info.n_solve = info.n_backend_solve = 0;
    info.status = PQ_UNSOLVED;
    info.iter = 0;
    info.reg_limit = set.reg_lower_limit;
    info.factor_retires = 0; info.no_primal_update = 0; info.no_dual_update = 0;
    info.mu = 0; info.primal_step = 0; info.dual_step = 0;
    info.rho = set.rho_init; info.delta = set.delta_init;

    // :416-437 start from s = z = 1 on the finite bounds
    std::fill(r.s_l.begin(), r.s_l.end(), 0.0); std::fill(r.s_u.begin(), r.s_u.end(), 0.0);
    std::fill(r.z_l.begin(), r.z_l.end(), 0.0); std::fill(r.z_u.begin(), r.z_u.end(), 0.0);
    for (int i = 0; i < d.n_h_l; ++i) { r.s_l[d.h_l_idx[i]] = 1.0; r.z_l[d.h_l_idx[i]] = 1.0; }
    for (int i = 0; i < d.n_h_u; ++i) { r.s_u[d.h_u_idx[i]] = 1.0; r.z_u[d.h_u_idx[i]] = 1.0; }
    for (int i = 0; i < d.n_x_l; ++i) { r.s_bl[i] = 1.0; r.z_bl[i] = 1.0; }
    for (int i = 0; i < d.n_x_u; ++i) { r.s_bu[i] = 1.0; r.z_bu[i] = 1.0; }

    m_enable_iterative_refinement = set.iterative_refinement_always_enabled != 0;

    t0 = now_s();
    while (!kkt_factor()) {
        if (!m_enable_iterative_refinement) m_enable_iterative_refinement = true;
        else if (info.factor_retires < set.max_factor_retires) {
            info.delta *= 100; info.rho *= 100; info.factor_retires++;
            info.reg_limit = std::min(10 * info.reg_limit, set.eps_abs);
        } else { info.status = PQ_NUMERICS; return info.status; }
    }
    info.factor_retires = 0;
    info.kkt_factor_time += now_s() - t0;

    for (int i = 0; i < n; ++i) res.x[i] = -d.c[i];
    for (int i = 0; i < p; ++i) res.y[i] = d.b[i];
    for (int i = 0; i < m; ++i) { res.z_l[i] = -d.h_l[i]; res.z_u[i] = d.h_u[i]; }
    for (int i = 0; i < n; ++i) { res.z_bl[i] = -d.x_l[i]; res.z_bu[i] = d.x_u[i]; }
    std::fill(res.s_l.begin(), res.s_l.end(), 0.0); std::fill(res.s_u.begin(), res.s_u.end(), 0.0);
    std::fill(res.s_bl.begin(), res.s_bl.end(), 0.0); std::fill(res.s_bu.begin(), res.s_bu.end(), 0.0);

    t0 = now_s();
    kkt_solve(res, r);
    info.kkt_solve_time += now_s() - t0;

    if (m + d.n_x_l + d.n_x_u > 0) {
        // :504-570 shift into the interior, then centre
        double delta_s = 0.0, delta_z = 0.0;
        if (m > 0) { delta_s = std::max(delta_s, -min_coeff(r.s_l, m)); delta_s = std::max(delta_s, -min_coeff(r.s_u, m)); }
        if (d.n_x_l > 0) delta_s = std::max(delta_s, -min_coeff(r.s_bl, d.n_x_l));
        if (d.n_x_u > 0) delta_s = std::max(delta_s, -min_coeff(r.s_bu, d.n_x_u));
        if (m > 0) { delta_z = std::max(delta_z, -min_coeff(r.z_l, m)); delta_z = std::max(delta_z, -min_coeff(r.z_u, m)); }
        if (d.n_x_l > 0) delta_z = std::max(delta_z, -min_coeff(r.z_bl, d.n_x_l));
        if (d.n_x_u > 0) delta_z = std::max(delta_z, -min_coeff(r.z_bu, d.n_x_u));
        for (int i = 0; i < d.n_h_l; ++i) { const int idx = d.h_l_idx[i]; r.s_l[idx] += delta_s; r.z_l[idx] += delta_z; }
        for (int i = 0; i < d.n_h_u; ++i) { const int idx = d.h_u_idx[i]; r.s_u[idx] += delta_s; r.z_u[idx] += delta_z; }
        for (int i = 0; i < d.n_x_l; ++i) { r.s_bl[i] += delta_s; r.z_bl[i] += delta_z; }
        for (int i = 0; i < d.n_x_u; ++i) { r.s_bu[i] += delta_s; r.z_bu[i] += delta_z; }
        info.mu = std::max(calculate_mu(), 1e-10);
        auto centre = [&](double& z, double& s) { const double cc = z - delta_z; z = (cc + std::sqrt(cc * cc + 4 * info.mu)) / 2; s = z - cc; };
        for (int i = 0; i < d.n_h_l; ++i) centre(r.z_l[d.h_l_idx[i]], r.s_l[d.h_l_idx[i]]);
        for (int i = 0; i < d.n_h_u; ++i) centre(r.z_u[d.h_u_idx[i]], r.s_u[d.h_u_idx[i]]);
        for (int i = 0; i < d.n_x_l; ++i) centre(r.z_bl[i], r.s_bl[i]);
        for (int i = 0; i < d.n_x_u; ++i) centre(r.z_bu[i], r.s_bu[i]);
        info.mu = calculate_mu();
    }

    px.x = r.x; px.y = r.y; px.z_l = r.z_l; px.z_u = r.z_u;
    std::copy(r.z_bl.begin(), r.z_bl.begin() + d.n_x_l, px.z_bl.begin());
    std::copy(r.z_bu.begin(), r.z_bu.begin() + d.n_x_u, px.z_bu.begin());

    while (info.iter < set.max_iter) {
        if (info.iter == 0) {
            update_residuals_nr();
            info.prev_primal_res = info.primal_res;
            info.prev_dual_res = info.dual_res;
        }
        if (trace_ && trace_rows_ < trace_max_) {
            double* row = trace_ + (size_t)trace_rows_ * 11;
            row[0] = info.iter; row[1] = info.primal_obj; row[2] = info.dual_obj; row[3] = info.duality_gap; row[4] = info.primal_res; row[5] = info.dual_res;
            row[6] = info.rho; row[7] = info.delta; row[8] = info.mu; row[9] = info.primal_step; row[10] = info.dual_step;
            trace_rows_++;
        }
        if (set.verbose) {
            std::printf("%3d   % .5e   % .5e   %.5e   %.5e   %.5e   %.3e   %.3e   %.3e   %.4f   %.4f\n", info.iter, info.primal_obj, info.dual_obj, info.duality_gap,
                        info.primal_res, info.dual_res, info.rho, info.delta, info.mu, info.primal_step, info.dual_step);
            std::fflush(stdout);
        }

        if ((info.primal_res < set.eps_abs || info.primal_res_rel < set.eps_rel) && (info.dual_res < set.eps_abs || info.dual_res_rel < set.eps_rel) &&
            (!set.check_duality_gap || info.duality_gap < set.eps_duality_gap_abs || info.duality_gap_rel < set.eps_duality_gap_rel)) {
            info.status = PQ_SOLVED;
            return info.status;
        }

        update_residuals_r();

        if (info.no_dual_update > std::min(5, set.reg_finetune_dual_update_threshold) && info.primal_prox_inf > set.infeasibility_threshold &&
            (info.primal_res_reg < set.eps_abs || info.primal_res_reg_rel < set.eps_rel)) {
            info.status = PQ_PRIMAL_INFEASIBLE;
            return info.status;
        }
        if (info.no_primal_update > std::min(5, set.reg_finetune_primal_update_threshold) && info.dual_prox_inf > set.infeasibility_threshold &&
            (info.dual_res_reg < set.eps_abs || info.dual_res_reg_rel < set.eps_rel)) {
            info.status = PQ_DUAL_INFEASIBLE;
            return info.status;
        }

        info.iter++;

        // :634-666 keep z off the boundary
        bool boundary_shifted = false;
        const double epsilon = DBL_EPSILON;
        for (int i = 0; i < d.n_h_l; ++i) { double& z = r.z_l[d.h_l_idx[i]]; if (z < epsilon) { z += epsilon; boundary_shifted = true; } }
        for (int i = 0; i < d.n_h_u; ++i) { double& z = r.z_u[d.h_u_idx[i]]; if (z < epsilon) { z += epsilon; boundary_shifted = true; } }
        if (d.n_x_l > 0 && min_coeff(r.z_bl, d.n_x_l) < epsilon) { for (int i = 0; i < d.n_x_l; ++i) r.z_bl[i] += epsilon; boundary_shifted = true; }
        if (d.n_x_u > 0 && min_coeff(r.z_bu, d.n_x_u) < epsilon) { for (int i = 0; i < d.n_x_u; ++i) r.z_bu[i] += epsilon; boundary_shifted = true; }
        if (boundary_shifted) info.mu = calculate_mu();

        // :668-681
        if ((info.no_primal_update > set.reg_finetune_primal_update_threshold && info.rho == info.reg_limit && info.reg_limit != set.reg_finetune_lower_limit) ||
            (info.no_dual_update > set.reg_finetune_dual_update_threshold && info.delta == info.reg_limit && info.reg_limit != set.reg_finetune_lower_limit)) {
            if (info.dual_prox_inf < set.infeasibility_threshold && info.primal_prox_inf < set.infeasibility_threshold) {
                info.reg_limit = set.reg_finetune_lower_limit;
                info.no_primal_update = 0;
                info.no_dual_update = 0;
            }
        }

        t0 = now_s();
        bool regularization_changed = false;
        while (!kkt_factor()) {
            if (!m_enable_iterative_refinement) { m_enable_iterative_refinement = true; continue; }
            if (info.factor_retires < set.max_factor_retires) {
                info.delta *= 100; info.rho *= 100; info.factor_retires++;
                info.reg_limit = std::min(10 * info.reg_limit, set.eps_abs);
                regularization_changed = true;
                continue;
            }
            info.status = PQ_NUMERICS;
            return info.status;
        }
        info.factor_retires = 0;
        info.kkt_factor_time += now_s() - t0;
        if (regularization_changed) update_residuals_r();

        if (m + d.n_x_l + d.n_x_u > 0) {
            // predictor
            for (int i = 0; i < m; ++i) { res.s_l[i] = -r.s_l[i] * r.z_l[i]; res.s_u[i] = -r.s_u[i] * r.z_u[i]; }
            for (int i = 0; i < d.n_x_l; ++i) res.s_bl[i] = -r.s_bl[i] * r.z_bl[i];
            for (int i = 0; i < d.n_x_u; ++i) res.s_bu[i] = -r.s_bu[i] * r.z_bu[i];
            t0 = now_s();
            kkt_solve(res, step);
            info.kkt_solve_time += now_s() - t0;

            double alpha_s, alpha_z;
            calculate_step(alpha_s, alpha_z);
            alpha_s *= set.tau; alpha_z *= set.tau;

            double sigma = 0.0, acc = 0.0;
            for (int i = 0; i < m; ++i) acc += (r.s_l[i] + alpha_s * step.s_l[i]) * (r.z_l[i] + alpha_z * step.z_l[i]);
            sigma = acc; acc = 0.0;
            for (int i = 0; i < m; ++i) acc += (r.s_u[i] + alpha_s * step.s_u[i]) * (r.z_u[i] + alpha_z * step.z_u[i]);
            sigma += acc; acc = 0.0;
            for (int i = 0; i < d.n_x_l; ++i) acc += (r.s_bl[i] + alpha_s * step.s_bl[i]) * (r.z_bl[i] + alpha_z * step.z_bl[i]);
            sigma += acc; acc = 0.0;
            for (int i = 0; i < d.n_x_u; ++i) acc += (r.s_bu[i] + alpha_s * step.s_bu[i]) * (r.z_bu[i] + alpha_z * step.z_bu[i]);
            sigma += acc;
            sigma /= (info.mu * (double)(d.n_h_l + d.n_h_u + d.n_x_l + d.n_x_u));
            sigma = std::max(0.0, std::min(1.0, sigma));
            info.sigma = sigma * sigma * sigma;

            // corrector
            const double sm = info.sigma * info.mu;
            for (int i = 0; i < m; ++i) { res.s_l[i] += -step.s_l[i] * step.z_l[i] + sm; res.s_u[i] += -step.s_u[i] * step.z_u[i] + sm; }
            for (int i = 0; i < d.n_x_l; ++i) res.s_bl[i] += -step.s_bl[i] * step.z_bl[i] + sm;
            for (int i = 0; i < d.n_x_u; ++i) res.s_bu[i] += -step.s_bu[i] * step.z_bu[i] + sm;
            t0 = now_s();
            kkt_solve(res, step);
            info.kkt_solve_time += now_s() - t0;

            calculate_step(alpha_s, alpha_z);
            info.primal_step = alpha_s * set.tau;
            info.dual_step = alpha_z * set.tau;

            for (int i = 0; i < n; ++i) r.x[i] += info.primal_step * step.x[i];
            for (int i = 0; i < p; ++i) r.y[i] += info.dual_step * step.y[i];
            for (int i = 0; i < m; ++i) { r.z_l[i] += info.dual_step * step.z_l[i]; r.z_u[i] += info.dual_step * step.z_u[i]; }
            for (int i = 0; i < d.n_x_l; ++i) r.z_bl[i] += info.dual_step * step.z_bl[i];
            for (int i = 0; i < d.n_x_u; ++i) r.z_bu[i] += info.dual_step * step.z_bu[i];
            for (int i = 0; i < m; ++i) { r.s_l[i] += info.primal_step * step.s_l[i]; r.s_u[i] += info.primal_step * step.s_u[i]; }
            for (int i = 0; i < d.n_x_l; ++i) r.s_bl[i] += info.primal_step * step.s_bl[i];
            for (int i = 0; i < d.n_x_u; ++i) r.s_bu[i] += info.primal_step * step.s_bu[i];

            const double mu_prev = info.mu;
            info.mu = calculate_mu();
            const double mu_rate = std::max(0.0, (mu_prev - info.mu) / mu_prev);

            update_residuals_nr();

            if (info.dual_res < 0.95 * info.prev_dual_res || (info.dual_res < set.eps_abs || info.dual_res_rel < set.eps_rel) ||
                (info.rho == set.reg_finetune_lower_limit && info.dual_prox_inf < set.infeasibility_threshold)) {
                px.x = r.x;
                info.rho = std::max(info.reg_limit, (1.0 - mu_rate) * info.rho);
            } else {
                info.no_primal_update++;
                if (info.iter < 5 || info.dual_prox_inf < set.infeasibility_threshold) info.rho = std::max(info.reg_limit, (1.0 - 0.666 * mu_rate) * info.rho);
            }
            if (info.primal_res < 0.95 * info.prev_primal_res || (info.primal_res < set.eps_abs || info.primal_res_rel < set.eps_rel) ||
                (info.delta == set.reg_finetune_lower_limit && info.primal_prox_inf < set.infeasibility_threshold)) {
                px.y = r.y; px.z_l = r.z_l; px.z_u = r.z_u;
                std::copy(r.z_bl.begin(), r.z_bl.begin() + d.n_x_l, px.z_bl.begin());
                std::copy(r.z_bu.begin(), r.z_bu.begin() + d.n_x_u, px.z_bu.begin());
                info.delta = std::max(info.reg_limit, (1.0 - mu_rate) * info.delta);
            } else {
                info.no_dual_update++;
                if (info.iter < 5 || info.primal_prox_inf < set.infeasibility_threshold) info.delta = std::max(info.reg_limit, (1.0 - 0.666 * mu_rate) * info.delta);
            }
        } else {
            // :831-877 no inequalities: one solve, full step
            t0 = now_s();
            kkt_solve(res, step);
            info.kkt_solve_time += now_s() - t0;
            info.primal_step = 1.0; info.dual_step = 1.0;
            for (int i = 0; i < n; ++i) r.x[i] += step.x[i];
            for (int i = 0; i < p; ++i) r.y[i] += step.y[i];
            update_residuals_nr();
            if (info.dual_res < 0.95 * info.prev_dual_res || (info.dual_res < set.eps_abs || info.dual_res_rel < set.eps_rel)) {
                px.x = r.x;
                info.rho = std::max(info.reg_limit, 0.1 * info.rho);
            } else {
                info.no_primal_update++;
                if (info.iter < 5 || info.dual_prox_inf < set.infeasibility_threshold) info.rho = std::max(info.reg_limit, 0.5 * info.rho);
            }
            if (info.primal_res < 0.95 * info.prev_primal_res || (info.primal_res < set.eps_abs || info.primal_res_rel < set.eps_rel)) {
                px.y = r.y;
                info.delta = std::max(info.reg_limit, 0.1 * info.delta);
            } else {
                info.no_dual_update++;
                if (info.iter < 5 || info.primal_prox_inf < set.infeasibility_threshold) info.delta = std::max(info.reg_limit, 0.5 * info.delta);
            }
        }
    }
    info.status = PQ_MAX_ITER_REACHED;
    return info.status;
}

// solver.hpp:1205-1227
void Solver::unscale_results()
{
    const HostData& d = *m_data; const int n = d.n, p = d.p, m = d.m;
    HostVars& r = m_result; const Ruiz& pc = m_preconditioner;
    for (int i = 0; i < n; ++i) r.x[i] *= pc.delta[i];
    for (int i = 0; i < p; ++i) r.y[i] = r.y[i] * pc.c_inv * pc.delta[n + i];
    for (int i = 0; i < m; ++i) { r.z_l[i] = r.z_l[i] * pc.c_inv * pc.delta[n + p + i]; r.z_u[i] = r.z_u[i] * pc.c_inv * pc.delta[n + p + i]; }
    for (int i = 0; i < m; ++i) { r.s_l[i] *= pc.delta_inv[n + p + i]; r.s_u[i] *= pc.delta_inv[n + p + i]; }
    for (int i = 0; i < d.n_x_l; ++i) { const int idx = d.x_l_idx[i]; r.z_bl[i] = r.z_bl[i] * pc.c_inv * pc.delta_b[idx]; r.s_bl[i] *= pc.delta_b_inv[idx]; }
    for (int i = 0; i < d.n_x_u; ++i) { const int idx = d.x_u_idx[i]; r.z_bu[i] = r.z_bu[i] * pc.c_inv * pc.delta_b[idx]; r.s_bu[i] *= pc.delta_b_inv[idx]; }
}

// solver.hpp:1229-1259
void Solver::restore_dual()
{
    const HostData& d = *m_data; const int n = d.n, m = d.m;
    HostVars& r = m_result;
    for (int i = 0; i < m; ++i) { if (r.z_l[i] == 0) r.s_l[i] = PIQP_INF; if (r.z_u[i] == 0) r.s_u[i] = PIQP_INF; }
    for (int i = d.n_x_l; i < n; ++i) { r.z_bl[i] = 0.0; r.s_bl[i] = PIQP_INF; }
    for (int i = d.n_x_u; i < n; ++i) { r.z_bu[i] = 0.0; r.s_bu[i] = PIQP_INF; }
    for (int i = d.n_x_l - 1; i >= 0; --i) { const int idx = d.x_l_idx[i]; std::swap(r.z_bl[i], r.z_bl[idx]); std::swap(r.s_bl[i], r.s_bl[idx]); }
    for (int i = d.n_x_u - 1; i >= 0; --i) { const int idx = d.x_u_idx[i]; std::swap(r.z_bu[i], r.z_bu[idx]); std::swap(r.s_bu[i], r.s_bu[idx]); }
}

// solver.hpp:69-148
int Solver::solve()
{
    PQ_ZONE("piqp_amd::Solver::solve");
    const double t0 = now_s();
    if (m_setup_done) PQ_HIP(hipSetDevice(device_));
    const int status = solve_impl();
    if (m_setup_done && status != PQ_INVALID_SETTINGS) {
        if (dipm_) dipm_->download_result(m_result);
        else { unscale_results(); restore_dual(); }
    }
    m_info.solve_time = now_s() - t0;
    m_info.run_time = (m_first_run ? m_info.setup_time : m_info.update_time) + m_info.solve_time;
    if (m_settings.verbose) std::printf("\nstatus:               %d\nnumber of iterations: %d\nobjective:            %.5e\n", status, m_info.iter, m_info.primal_obj);
    m_first_run = false;
    return status;
}

}  // namespace pq

// ------------------------------------------------------------------ C-ABI (include/piqp_amd.h, "Solver" section)
using namespace pq;

struct pq_solver {
    std::unique_ptr<Solver> impl;
};

extern "C" {

int pq_solver_create(pq_solver** out, int device)
{
    if (!out) return fail(PQ_ERR_INVALID, "null argument");
    *out = nullptr;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return fail(PQ_ERR_HIP, "no HIP device visible: this library has no CPU fallback");
    if (device < 0 || device >= cnt) return fail(PQ_ERR_INVALID, "device %d out of range", device);
    return guarded([&] { auto* h = new pq_solver; h->impl.reset(new Solver(device)); *out = h; return (int)PQ_OK; });
}
void pq_solver_destroy(pq_solver* s) { delete s; }
int pq_solver_clone(const pq_solver* s, pq_solver** out)
{
    if (!s || !out) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { auto* h = new pq_solver; h->impl.reset(s->impl->clone()); *out = h; return (int)PQ_OK; });
}
pq_settings* pq_solver_settings(pq_solver* s) { return s ? &s->impl->settings() : nullptr; }

int pq_solver_setup_dense(pq_solver* s, int n, int p, int m, const double* P, const double* c, const double* A, const double* b, const double* G, const double* h_l,
                          const double* h_u, const double* x_l, const double* x_u)
{
    if (!s || !P || !c || n <= 0) return fail(PQ_ERR_INVALID, "bad argument");
    // solver.hpp:175-178 argument checks
    if ((A && !b && p > 0) || (!h_l && !h_u && G && m > 0)) return fail(PQ_ERR_INVALID, "b / h_l or h_u must be provided");
    return guarded([&] { return s->impl->setup(make_dense_host_data(n, p, m, P, c, A, b, G, h_l, h_u, x_l, x_u)) ? 1 : 0; });
}
int pq_solver_setup_sparse(pq_solver* s, int n, int p, int m, const int* Pp, const int* Pi, const double* Px, const double* c, const int* Ap, const int* Ai, const double* Ax,
                           const double* b, const int* Gp, const int* Gi, const double* Gx, const double* h_l, const double* h_u, const double* x_l, const double* x_u)
{
    if (!s || !Pp || !c || n <= 0) return fail(PQ_ERR_INVALID, "bad argument");
    if ((Ap && !b && p > 0) || (!h_l && !h_u && Gp && m > 0)) return fail(PQ_ERR_INVALID, "b / h_l or h_u must be provided");
    return guarded([&] { return s->impl->setup(make_sparse_host_data(n, p, m, Pp, Pi, Px, c, Ap, Ai, Ax, b, Gp, Gi, Gx, h_l, h_u, x_l, x_u)) ? 1 : 0; });
}
int pq_solver_update_dense(pq_solver* s, const double* P, const double* c, const double* A, const double* b, const double* G, const double* h_l, const double* h_u,
                           const double* x_l, const double* x_u)
{
    if (!s) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { return s->impl->update_dense(P, c, A, b, G, h_l, h_u, x_l, x_u) ? 1 : 0; });
}
int pq_solver_update_sparse(pq_solver* s, const int* Pp, const int* Pi, const double* Px, const double* c, const int* Ap, const int* Ai, const double* Ax, const double* b,
                            const int* Gp, const int* Gi, const double* Gx, const double* h_l, const double* h_u, const double* x_l, const double* x_u)
{
    if (!s) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { return s->impl->update_sparse(Pp, Pi, Px, c, Ap, Ai, Ax, b, Gp, Gi, Gx, h_l, h_u, x_l, x_u) ? 1 : 0; });
}
int pq_solver_solve(pq_solver* s)
{
    if (!s) return PQ_UNSOLVED;
    int status = PQ_UNSOLVED;
    int rc = guarded([&] { status = s->impl->solve(); return (int)PQ_OK; });
    return rc < 0 ? PQ_NUMERICS : status;
}
const pq_info* pq_solver_info(const pq_solver* s) { return s ? &s->impl->info() : nullptr; }
int pq_solver_get_result(const pq_solver* s, pq_vars* out)
{
    if (!s || !out) return fail(PQ_ERR_INVALID, "null argument");
    const HostVars& r = s->impl->result();
    double* dst[10] = {out->x, out->y, out->z_l, out->z_u, out->z_bl, out->z_bu, out->s_l, out->s_u, out->s_bl, out->s_bu};
    for (int k = 0; k < 10; ++k) if (dst[k]) std::copy(r.field(k).begin(), r.field(k).end(), dst[k]);
    return PQ_OK;
}
int pq_solver_dims(const pq_solver* s, int* n, int* p, int* m)
{
    if (!s || !s->impl->data()) return fail(PQ_ERR_INVALID, "solver not set up");
    if (n) *n = s->impl->data()->n;
    if (p) *p = s->impl->data()->p;
    if (m) *m = s->impl->data()->m;
    return PQ_OK;
}
int pq_solver_set_trace(pq_solver* s, double* buf_host, int max_rows)
{
    if (!s) return fail(PQ_ERR_INVALID, "null argument");
    s->impl->set_trace(buf_host, max_rows);
    return PQ_OK;
}
int pq_solver_partition(pq_solver* s, int rank, int world, long long sizes_out[3])
{
    if (!s || !sizes_out || !s->impl->backend()) return fail(PQ_ERR_INVALID, "solver not set up");
    return guarded([&] { s->impl->backend()->partition(rank, world, sizes_out); return (int)PQ_OK; });
}
int pq_solver_set_exchange_norm(pq_solver* s, double* buf_norm)
{
    if (!s || !s->impl) return fail(PQ_ERR_INVALID, "null handle");
    return guarded([&] { s->impl->backend()->set_exchange_norm(buf_norm); return (int)PQ_OK; });
}
int pq_solver_sharded_calls(pq_solver* s, int out[2])
{
    if (!s || !s->impl || !out) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { s->impl->backend()->sharded_calls(out); return (int)PQ_OK; });
}
int pq_solver_sharded_solve_calls(pq_solver* s, int out[6])
{
    if (!s || !s->impl || !out || !s->impl->backend()) return fail(PQ_ERR_INVALID, "null argument / solver not set up");
    return guarded([&] { s->impl->backend()->sharded_solve_calls(out); return (int)PQ_OK; });
}
int pq_solver_native_exchange_calls(pq_solver* s, int out[3])
{
    if (!s || !out) return fail(PQ_ERR_INVALID, "null argument");
    if (!s->impl->backend()) return fail(PQ_ERR_INVALID, "solver not set up");
    return guarded([&] { s->impl->backend()->native_exchange_calls(out); return (int)PQ_OK; });
}
int pq_solver_set_comm_rccl(pq_solver* s, const unsigned char id[128], int rank, int world)
{
    if (!s || !id) return fail(PQ_ERR_INVALID, "null argument");
    if (!s->impl->backend()) return fail(PQ_ERR_INVALID, "solver not set up");
    return guarded([&] { s->impl->backend()->set_comm_rccl(id, rank, world); return (int)PQ_OK; });
}
int pq_solver_comm_info(pq_solver* s, int out[4])
{
    if (!s || !out) return fail(PQ_ERR_INVALID, "null argument");
    if (!s->impl->backend()) return fail(PQ_ERR_INVALID, "solver not set up");
    return guarded([&] { s->impl->backend()->comm_info(out); return (int)PQ_OK; });
}
int pq_solver_set_exchange(pq_solver* s, pq_exchange_fn exchange, void* user, double* buf_factor, double* buf_forward, double* buf_gather)
{
    if (!s || !s->impl->backend()) return fail(PQ_ERR_INVALID, "solver not set up");
    return guarded([&] { s->impl->backend()->set_exchange(exchange, user, buf_factor, buf_forward, buf_gather); return (int)PQ_OK; });
}
int pq_solver_trace_rows(const pq_solver* s) { return s ? s->impl->trace_rows() : 0; }

}  // extern "C"
