// piqp_amd/csrc/device_ipm.hip -- the interior-point iteration of piqp::SolverBase::solve_impl (reference solver.hpp:379-882)
// with every Variables-sized vector resident in HBM: residuals (:960-1128), step lengths (:893-958),
// mu / sigma dots (:884-891, :747-753), the shift into the interior (:504-570), the variable updates (:779-790) and the
// proximal-point bookkeeping run as elementwise / reduction kernels on the KKTSystem's stream.  The host keeps only the
// scalar control flow of solve_impl (termination tests, rho / delta / reg_limit updates, retries): it reads a handful of
// reduced scalars per iteration from pinned memory, never a vector.  Every KKT factor / solve / mat-vec goes to the
// device KKTSystem / backend with device pointers, so there is no PCIe traffic inside the loop.
//
// Reductions are two-stage with a fixed order (block partials, then one workgroup), i.e. deterministic: a cloned solver
// reproduces its results bit for bit.
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>

#include "trace.hpp"
#include "solver.hpp"

namespace pq {

namespace {

constexpr int NT = 256;
constexpr int MAXB = 512;   // blocks of a reduction launch
constexpr int NS = 20;      // scalar slots of one phase

enum Op : int { OP_SUM = 0, OP_MAX = 1, OP_MIN = 2, OP_AMAXNAN = 3 };  // OP_MAX / OP_MIN = std::max / std::min; OP_AMAXNAN = Eigen lpNorm<Infinity> (NaN propagates)

__device__ __forceinline__ double op_identity(int op) { return op == OP_MIN ? DBL_MAX : (op == OP_MAX ? -DBL_MAX : 0.0); }
__device__ __forceinline__ double op_apply(int op, double a, double b)
{
    switch (op) {
    case OP_SUM: return a + b;
    case OP_MAX: return a < b ? b : a;
    case OP_MIN: return b < a ? b : a;
    default: return (b > a || b != b) ? b : a;
    }
}

struct Ops { int op[NS]; };

// block-level reduction of K per-thread accumulators into part[blockIdx.x * NS + k]
template <int K>
__device__ void block_reduce_store(double (&v)[K], const Ops& ops, double* __restrict__ part)
{
    __shared__ double sh[NT / 64][NS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double x = v[k];
        for (int o = 32; o > 0; o >>= 1) x = op_apply(ops.op[k], x, __shfl_xor(x, o));
        if (lane == 0) sh[wave][k] = x;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        double x = sh[0][threadIdx.x];
        for (int w = 1; w < NT / 64; ++w) x = op_apply(ops.op[threadIdx.x], x, sh[w][threadIdx.x]);
        part[(size_t)blockIdx.x * NS + threadIdx.x] = x;
    }
}
// second stage: one wave per slot; lane l folds the block partials l, l + 64, ... in order, then the 64 lane values are folded by a
// fixed butterfly.  Same order on every launch -> bitwise reproducible.  (The first version walked all <= 512 partials with one
// thread per slot: 512 dependent L2 reads, 88 us per reduction and nine reductions per iteration -- a fifth of the solve.)
// `host` : the same slot in the host's pinned (device-visible) copy: the scalars are there when the stream has drained, without a copy in between
__global__ __launch_bounds__(1024) void k_final_reduce(int nblocks, int K, Ops ops, const double* __restrict__ part, double* __restrict__ out, double* __restrict__ host)
{
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (k >= K) return;
    const int op = ops.op[k];
    double x = op_identity(op);
    for (int b = lane; b < nblocks; b += 64) x = op_apply(op, x, part[(size_t)b * NS + k]);
    for (int o = 32; o > 0; o >>= 1) x = op_apply(op, x, __shfl_xor(x, o));
    if (lane == 0) { out[k] = x; host[k] = x; }
}

// ---- reference-order sums (behind the reference-order sparse engine, sparse_exact.hip).  The reference's dot products are left-to-right sums of rounded
// products (Eigen's expressions as the CPU oracle restates them: oracle/orc_solver.c dot()); a tree reduction gives the same number only up to rounding, and on the
// degenerate LPs that decide their trajectory by rounding "up to rounding" is another iteration count.  One wave per dot product: 64 products at a time across the
// lanes, added to the running sum strictly in index order (lane 0 first).  term = a[i] b[i], or (a[i] + as da[i]) (b[i] + az db[i]) when da is set (solver.hpp:747-750).
constexpr int MAXDOT = 12;
struct DotJob { const double *a, *b, *da, *db; int n; };
struct DotJobs { DotJob j[MAXDOT]; double as, az; };
__global__ __launch_bounds__(64) void k_seq_dots(DotJobs J, double* __restrict__ out, double* __restrict__ host)
{
    const DotJob jb = J.j[blockIdx.x];
    const int lane = threadIdx.x;
    double s = 0.0;
    for (int base = 0; base < jb.n; base += 64) {
        const int i = base + lane;
        double t = 0.0;
        if (i < jb.n) t = jb.da ? (jb.a[i] + J.as * jb.da[i]) * (jb.b[i] + J.az * jb.db[i]) : jb.a[i] * jb.b[i];
        const int cnt = min(64, jb.n - base);
        for (int l = 0; l < cnt; ++l) {
            const int lo = __builtin_amdgcn_readlane(__double2loint(t), l), hi = __builtin_amdgcn_readlane(__double2hiint(t), l);
            s = s + __hiloint2double(hi, lo);
        }
    }
    if (lane == 0) { out[blockIdx.x] = s; host[blockIdx.x] = s; }
}

struct Dims {
    int n, p, m, nxl, nxu;
};
struct Masks {  // finite-bound structure (shared by every vector kernel)
    const int *has_l, *has_u, *pos_l, *pos_u, *x_l_idx, *x_u_idx;
};
struct DataV {  // scaled problem vectors + Ruiz scalings
    const double *c, *b, *h_l, *h_u, *x_l, *x_u, *xbs, *dl, *dli, *db, *dbi;
};

#define GRID_STRIDE(i, cnt) for (int i = blockIdx.x * NT + threadIdx.x; i < (cnt); i += gridDim.x * NT)

// :416-437
__global__ void k_init_sz(Dims d, Masks M, pq_vars r)
{
    GRID_STRIDE(i, d.m) {
        const double l = M.has_l[i] ? 1.0 : 0.0, u = M.has_u[i] ? 1.0 : 0.0;
        r.s_l[i] = l; r.z_l[i] = l; r.s_u[i] = u; r.z_u[i] = u;
    }
    GRID_STRIDE(i, d.n) {
        const double l = i < d.nxl ? 1.0 : 0.0, u = i < d.nxu ? 1.0 : 0.0;
        r.s_bl[i] = l; r.z_bl[i] = l; r.s_bu[i] = u; r.z_bu[i] = u;
    }
}
// :473-483
__global__ void k_init_rhs(Dims d, DataV D, pq_vars rs)
{
    GRID_STRIDE(i, d.n) { rs.x[i] = -D.c[i]; rs.z_bl[i] = -D.x_l[i]; rs.z_bu[i] = D.x_u[i]; rs.s_bl[i] = 0.0; rs.s_bu[i] = 0.0; }
    GRID_STRIDE(i, d.p) rs.y[i] = D.b[i];
    GRID_STRIDE(i, d.m) { rs.z_l[i] = -D.h_l[i]; rs.z_u[i] = D.h_u[i]; rs.s_l[i] = 0.0; rs.s_u[i] = 0.0; }
}
// :506-521 slots: 0 = min over the s vectors, 1 = min over the z vectors
__global__ __launch_bounds__(NT) void k_min_sz(Dims d, pq_vars r, Ops ops, double* __restrict__ part)
{
    double v[2] = {DBL_MAX, DBL_MAX};
    GRID_STRIDE(i, d.m) { v[0] = fmin(v[0], fmin(r.s_l[i], r.s_u[i])); v[1] = fmin(v[1], fmin(r.z_l[i], r.z_u[i])); }
    GRID_STRIDE(i, d.nxl) { v[0] = fmin(v[0], r.s_bl[i]); v[1] = fmin(v[1], r.z_bl[i]); }
    GRID_STRIDE(i, d.nxu) { v[0] = fmin(v[0], r.s_bu[i]); v[1] = fmin(v[1], r.z_bu[i]); }
    block_reduce_store<2>(v, ops, part);
}
// :523-541 shift, then slot 0 = sum s.z
__global__ __launch_bounds__(NT) void k_shift_mu(Dims d, Masks M, pq_vars r, double delta_s, double delta_z, Ops ops, double* __restrict__ part)
{
    double v[1] = {0.0};
    GRID_STRIDE(i, d.m) {
        if (M.has_l[i]) { r.s_l[i] += delta_s; r.z_l[i] += delta_z; }
        if (M.has_u[i]) { r.s_u[i] += delta_s; r.z_u[i] += delta_z; }
        v[0] += r.s_l[i] * r.z_l[i] + r.s_u[i] * r.z_u[i];
    }
    GRID_STRIDE(i, d.nxl) { r.s_bl[i] += delta_s; r.z_bl[i] += delta_z; v[0] += r.s_bl[i] * r.z_bl[i]; }
    GRID_STRIDE(i, d.nxu) { r.s_bu[i] += delta_s; r.z_bu[i] += delta_z; v[0] += r.s_bu[i] * r.z_bu[i]; }
    block_reduce_store<1>(v, ops, part);
}
// :545-568 centre, then slot 0 = sum s.z
__global__ __launch_bounds__(NT) void k_centre_mu(Dims d, Masks M, pq_vars r, double mu, double delta_z, Ops ops, double* __restrict__ part)
{
    double v[1] = {0.0};
    auto centre = [&](double& z, double& s) { const double cc = z - delta_z; z = (cc + sqrt(cc * cc + 4 * mu)) / 2; s = z - cc; };
    GRID_STRIDE(i, d.m) {
        if (M.has_l[i]) centre(r.z_l[i], r.s_l[i]);
        if (M.has_u[i]) centre(r.z_u[i], r.s_u[i]);
        v[0] += r.s_l[i] * r.z_l[i] + r.s_u[i] * r.z_u[i];
    }
    GRID_STRIDE(i, d.nxl) { centre(r.z_bl[i], r.s_bl[i]); v[0] += r.s_bl[i] * r.z_bl[i]; }
    GRID_STRIDE(i, d.nxu) { centre(r.z_bu[i], r.s_bu[i]); v[0] += r.s_bu[i] * r.z_bu[i]; }
    block_reduce_store<1>(v, ops, part);
}
// prox <- result: which = 1 x, 2 duals, 3 both (:572-579, :799-828)
__global__ void k_copy_prox(Dims d, pq_vars r, pq_vars px, int which)
{
    if (which & 1) GRID_STRIDE(i, d.n) px.x[i] = r.x[i];
    if (which & 2) {
        GRID_STRIDE(i, d.p) px.y[i] = r.y[i];
        GRID_STRIDE(i, d.m) { px.z_l[i] = r.z_l[i]; px.z_u[i] = r.z_u[i]; }
        GRID_STRIDE(i, d.nxl) px.z_bl[i] = r.z_bl[i];
        GRID_STRIDE(i, d.nxu) px.z_bu[i] = r.z_bu[i];
    }
}

// ---- update_residuals_nr (:960-1105); the mat-vecs are issued by the host between these kernels --------------------
__global__ void k_nr_wz(Dims d, pq_vars r, double* __restrict__ work_z) { GRID_STRIDE(i, d.m) work_z[i] = r.z_u[i] - r.z_l[i]; }
__global__ void k_nr_after_G(Dims d, pq_vars nr, double* __restrict__ work_x, const double* __restrict__ work_x2)
{
    GRID_STRIDE(i, d.m) nr.z_u[i] = -nr.z_l[i];
    GRID_STRIDE(i, d.n) work_x[i] += work_x2[i];
}
// slots: 0 amax(nr.x scaled) [before -c], 1 sum x.nrx, 2 sum c.x, 3 amax(c scaled), 4 amax(work_x scaled) [after the box terms]
__global__ __launch_bounds__(NT) void k_nr_x(Dims d, Masks M, DataV D, double ci, pq_vars r, pq_vars nr, double* __restrict__ work_x, Ops ops, double* __restrict__ part)
{
    double v[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    GRID_STRIDE(j, d.n) {
        const double nx = nr.x[j], x = r.x[j], c = D.c[j], sc = ci * D.dli[j];
        double t = fabs(nx * ci * D.dli[j]); if (t > v[0] || t != t) v[0] = t;
        v[1] += x * nx;
        v[2] += c * x;
        t = fabs(c * ci * D.dli[j]); if (t > v[3] || t != t) v[3] = t;
        double wx = work_x[j];
        const int il = M.pos_l[j], iu = M.pos_u[j];
        if (il >= 0) wx -= D.xbs[j] * r.z_bl[il];
        if (iu >= 0) wx += D.xbs[j] * r.z_bu[iu];
        t = fabs(wx * ci * D.dli[j]); if (t > v[4] || t != t) v[4] = t;
        nr.x[j] = (nx - c) - wx;
        (void)sc;
    }
    block_reduce_store<5>(v, ops, part);
}
// slots: 0 amax(nr.y scaled) [before +b], 1 sum b.y, 2 amax(b scaled), 3 sum h_l.z_l, 4 sum h_u.z_u, 5 max of the signed general-row terms,
//        6 sum x_l.z_bl, 7 sum x_u.z_bu, 8 max of the signed box terms
__global__ __launch_bounds__(NT) void k_nr_yz(Dims d, Masks M, DataV D, pq_vars r, pq_vars nr, Ops ops, double* __restrict__ part)
{
    double v[9] = {0.0, 0.0, 0.0, 0.0, 0.0, -DBL_MAX, 0.0, 0.0, -DBL_MAX};
    const double* dy = D.dli + d.n;
    const double* dz = D.dli + d.n + d.p;
    GRID_STRIDE(i, d.p) {
        const double ny = nr.y[i];
        double t = fabs(ny * dy[i]); if (t > v[0] || t != t) v[0] = t;
        v[1] += D.b[i] * r.y[i];
        t = fabs(D.b[i] * dy[i]); if (t > v[2] || t != t) v[2] = t;
        nr.y[i] = ny + D.b[i];
    }
    GRID_STRIDE(i, d.m) {
        v[3] += D.h_l[i] * r.z_l[i];
        v[4] += D.h_u[i] * r.z_u[i];
        if (M.has_l[i]) {
            const double z = nr.z_l[i];
            v[5] = fmax(v[5], fmax(z * dz[i], fmax(D.h_l[i] * dz[i], r.s_l[i] * dz[i])));
            nr.z_l[i] = z + (-D.h_l[i] - r.s_l[i]);
        } else nr.z_l[i] = 0.0;
        if (M.has_u[i]) {
            const double z = nr.z_u[i];
            v[5] = fmax(v[5], fmax(z * dz[i], fmax(D.h_u[i] * dz[i], r.s_u[i] * dz[i])));
            nr.z_u[i] = z + (D.h_u[i] - r.s_u[i]);
        } else nr.z_u[i] = 0.0;
    }
    GRID_STRIDE(i, d.nxl) {
        const int idx = M.x_l_idx[i];
        const double t = D.xbs[idx] * r.x[idx];
        v[6] += D.x_l[i] * r.z_bl[i];
        v[8] = fmax(v[8], fmax(t * D.dbi[idx], fmax(D.x_l[i] * D.dbi[idx], r.s_bl[i] * D.dbi[idx])));
        nr.z_bl[i] = t + (-D.x_l[i] - r.s_bl[i]);
    }
    GRID_STRIDE(i, d.nxu) {
        const int idx = M.x_u_idx[i];
        const double t = -D.xbs[idx] * r.x[idx];
        v[7] += D.x_u[i] * r.z_bu[i];
        v[8] = fmax(v[8], fmax(t * D.dbi[idx], fmax(D.x_u[i] * D.dbi[idx], r.s_bu[i] * D.dbi[idx])));
        nr.z_bu[i] = t + (D.x_u[i] - r.s_bu[i]);
    }
    block_reduce_store<9>(v, ops, part);
}
// primal_res_of / dual_res_of (:1130-1196) of one Variables set.  slots: 0 amax(y), 1 amax(z_l), 2 amax(z_u), 3 signed max box, 4 amax(x);
// plus (with_mu) 5 = sum s.z of `r`
__global__ __launch_bounds__(NT) void k_res_norms(Dims d, Masks M, DataV D, double ci, pq_vars v_, pq_vars r, int with_mu, Ops ops, double* __restrict__ part)
{
    double v[6] = {0.0, 0.0, 0.0, -DBL_MAX, 0.0, 0.0};
    const double* dy = D.dli + d.n;
    const double* dz = D.dli + d.n + d.p;
    GRID_STRIDE(i, d.p) { const double t = fabs(v_.y[i] * dy[i]); if (t > v[0] || t != t) v[0] = t; }
    GRID_STRIDE(i, d.m) {
        double t = fabs(v_.z_l[i] * dz[i]); if (t > v[1] || t != t) v[1] = t;
        t = fabs(v_.z_u[i] * dz[i]); if (t > v[2] || t != t) v[2] = t;
        if (with_mu) v[5] += r.s_l[i] * r.z_l[i] + r.s_u[i] * r.z_u[i];
    }
    GRID_STRIDE(i, d.nxl) { v[3] = fmax(v[3], v_.z_bl[i] * D.dbi[M.x_l_idx[i]]); if (with_mu) v[5] += r.s_bl[i] * r.z_bl[i]; }
    GRID_STRIDE(i, d.nxu) { v[3] = fmax(v[3], v_.z_bu[i] * D.dbi[M.x_u_idx[i]]); if (with_mu) v[5] += r.s_bu[i] * r.z_bu[i]; }
    GRID_STRIDE(j, d.n) { const double t = fabs(v_.x[j] * ci * D.dli[j]); if (t > v[4] || t != t) v[4] = t; }
    block_reduce_store<6>(v, ops, part);
}
// update_residuals_r (:1107-1128): rs = nr - rho (x - xi) ..., slots 0..4 as k_res_norms on rs, 5 primal_prox_inf, 6 dual_prox_inf
__global__ __launch_bounds__(NT) void k_res_r(Dims d, Masks M, DataV D, double ci, double rho, double delta, pq_vars r, pq_vars nr, pq_vars px, pq_vars rs, Ops ops,
                                              double* __restrict__ part)
{
    double v[7] = {0.0, 0.0, 0.0, -DBL_MAX, 0.0, 0.0, 0.0};
    const double* dy = D.dli + d.n;
    const double* dz = D.dli + d.n + d.p;
    const double* ly = D.dl + d.n;
    const double* lz = D.dl + d.n + d.p;
    GRID_STRIDE(j, d.n) {
        const double e = nr.x[j] - rho * (r.x[j] - px.x[j]);
        rs.x[j] = e;
        double t = fabs(e * ci * D.dli[j]); if (t > v[4] || t != t) v[4] = t;
        t = fabs((r.x[j] - px.x[j]) * D.dl[j]); if (v[6] < t) v[6] = t;
    }
    GRID_STRIDE(i, d.p) {
        const double e = nr.y[i] - delta * (px.y[i] - r.y[i]);
        rs.y[i] = e;
        double t = fabs(e * dy[i]); if (t > v[0] || t != t) v[0] = t;
        t = fabs((px.y[i] - r.y[i]) * ci * ly[i]); if (v[5] < t) v[5] = t;
    }
    GRID_STRIDE(i, d.m) {
        const double el = nr.z_l[i] - delta * (px.z_l[i] - r.z_l[i]), eu = nr.z_u[i] - delta * (px.z_u[i] - r.z_u[i]);
        rs.z_l[i] = el; rs.z_u[i] = eu;
        double t = fabs(el * dz[i]); if (t > v[1] || t != t) v[1] = t;
        t = fabs(eu * dz[i]); if (t > v[2] || t != t) v[2] = t;
        t = fabs((px.z_l[i] - r.z_l[i]) * ci * lz[i]); if (v[5] < t) v[5] = t;
        t = fabs((px.z_u[i] - r.z_u[i]) * ci * lz[i]); if (v[5] < t) v[5] = t;
    }
    GRID_STRIDE(i, d.nxl) {
        const int idx = M.x_l_idx[i];
        const double e = nr.z_bl[i] - delta * (px.z_bl[i] - r.z_bl[i]);
        rs.z_bl[i] = e;
        v[3] = fmax(v[3], e * D.dbi[idx]);
        const double t = (px.z_bl[i] - r.z_bl[i]) * ci * D.db[idx]; if (v[5] < t) v[5] = t;
    }
    GRID_STRIDE(i, d.nxu) {
        const int idx = M.x_u_idx[i];
        const double e = nr.z_bu[i] - delta * (px.z_bu[i] - r.z_bu[i]);
        rs.z_bu[i] = e;
        v[3] = fmax(v[3], e * D.dbi[idx]);
        const double t = (px.z_bu[i] - r.z_bu[i]) * ci * D.db[idx]; if (v[5] < t) v[5] = t;
    }
    block_reduce_store<7>(v, ops, part);
}
// :634-666 keep z off the boundary.  The box vectors are shifted as a whole when their minimum is below eps: the minimum is
// reduced first (slots 0, 1), the shift is applied by k_boundary_apply from the device scalars (no host round trip)
__global__ __launch_bounds__(NT) void k_boundary_min(Dims d, Masks M, pq_vars r, double eps, Ops ops, double* __restrict__ part)
{
    double v[2] = {DBL_MAX, DBL_MAX};
    GRID_STRIDE(i, d.m) {
        if (M.has_l[i] && r.z_l[i] < eps) r.z_l[i] += eps;
        if (M.has_u[i] && r.z_u[i] < eps) r.z_u[i] += eps;
    }
    GRID_STRIDE(i, d.nxl) v[0] = fmin(v[0], r.z_bl[i]);
    GRID_STRIDE(i, d.nxu) v[1] = fmin(v[1], r.z_bu[i]);
    block_reduce_store<2>(v, ops, part);
}
// applies the box shifts, builds the predictor right-hand side (:719-723) and reduces slot 0 = sum s.z (mu after the shift)
__global__ __launch_bounds__(NT) void k_boundary_apply_pred(Dims d, pq_vars r, pq_vars rs, double eps, const double* __restrict__ mins, Ops ops, double* __restrict__ part)
{
    double v[1] = {0.0};
    const bool sl = d.nxl > 0 && mins[0] < eps, su = d.nxu > 0 && mins[1] < eps;
    GRID_STRIDE(i, d.m) {
        rs.s_l[i] = -r.s_l[i] * r.z_l[i]; rs.s_u[i] = -r.s_u[i] * r.z_u[i];
        v[0] += r.s_l[i] * r.z_l[i] + r.s_u[i] * r.z_u[i];
    }
    GRID_STRIDE(i, d.nxl) { if (sl) r.z_bl[i] += eps; rs.s_bl[i] = -r.s_bl[i] * r.z_bl[i]; v[0] += r.s_bl[i] * r.z_bl[i]; }
    GRID_STRIDE(i, d.nxu) { if (su) r.z_bu[i] += eps; rs.s_bu[i] = -r.s_bu[i] * r.z_bu[i]; v[0] += r.s_bu[i] * r.z_bu[i]; }
    block_reduce_store<1>(v, ops, part);
}
// calculate_step (:893-958): slots 0 alpha_s, 1 alpha_z
__global__ __launch_bounds__(NT) void k_step(Dims d, pq_vars r, pq_vars st, Ops ops, double* __restrict__ part)
{
    double v[2] = {1.0, 1.0};
    auto upd = [](double& a, double rr, double ss) { if (ss < 0) { const double c = -rr / ss; if (c < a) a = c; } };
    GRID_STRIDE(i, d.m) { upd(v[0], r.s_l[i], st.s_l[i]); upd(v[0], r.s_u[i], st.s_u[i]); upd(v[1], r.z_l[i], st.z_l[i]); upd(v[1], r.z_u[i], st.z_u[i]); }
    GRID_STRIDE(i, d.nxl) { upd(v[0], r.s_bl[i], st.s_bl[i]); upd(v[1], r.z_bl[i], st.z_bl[i]); }
    GRID_STRIDE(i, d.nxu) { upd(v[0], r.s_bu[i], st.s_bu[i]); upd(v[1], r.z_bu[i], st.z_bu[i]); }
    block_reduce_store<2>(v, ops, part);
}
// :747-750 slot 0 = sum (s + as ds)(z + az dz)
__global__ __launch_bounds__(NT) void k_sigma(Dims d, pq_vars r, pq_vars st, double as, double az, Ops ops, double* __restrict__ part)
{
    double v[1] = {0.0};
    GRID_STRIDE(i, d.m) v[0] += (r.s_l[i] + as * st.s_l[i]) * (r.z_l[i] + az * st.z_l[i]) + (r.s_u[i] + as * st.s_u[i]) * (r.z_u[i] + az * st.z_u[i]);
    GRID_STRIDE(i, d.nxl) v[0] += (r.s_bl[i] + as * st.s_bl[i]) * (r.z_bl[i] + az * st.z_bl[i]);
    GRID_STRIDE(i, d.nxu) v[0] += (r.s_bu[i] + as * st.s_bu[i]) * (r.z_bu[i] + az * st.z_bu[i]);
    block_reduce_store<1>(v, ops, part);
}
// :756-759
__global__ void k_corrector_rhs(Dims d, pq_vars st, pq_vars rs, double smu)
{
    GRID_STRIDE(i, d.m) { rs.s_l[i] += -st.s_l[i] * st.z_l[i] + smu; rs.s_u[i] += -st.s_u[i] * st.z_u[i] + smu; }
    GRID_STRIDE(i, d.nxl) rs.s_bl[i] += -st.s_bl[i] * st.z_bl[i] + smu;
    GRID_STRIDE(i, d.nxu) rs.s_bu[i] += -st.s_bu[i] * st.z_bu[i] + smu;
}
// :779-790
__global__ void k_update(Dims d, pq_vars r, pq_vars st, double ps, double ds)
{
    GRID_STRIDE(i, d.n) r.x[i] += ps * st.x[i];
    GRID_STRIDE(i, d.p) r.y[i] += ds * st.y[i];
    GRID_STRIDE(i, d.m) { r.z_l[i] += ds * st.z_l[i]; r.z_u[i] += ds * st.z_u[i]; r.s_l[i] += ps * st.s_l[i]; r.s_u[i] += ps * st.s_u[i]; }
    GRID_STRIDE(i, d.nxl) { r.z_bl[i] += ds * st.z_bl[i]; r.s_bl[i] += ps * st.s_bl[i]; }
    GRID_STRIDE(i, d.nxu) { r.z_bu[i] += ds * st.z_bu[i]; r.s_bu[i] += ps * st.s_bu[i]; }
}
// unscale_results + restore_dual (:1205-1259): out = expanded, unscaled result
__global__ void k_finish(Dims d, Masks M, DataV D, double ci, pq_vars r, pq_vars out)
{
    GRID_STRIDE(i, d.n) out.x[i] = r.x[i] * D.dl[i];
    GRID_STRIDE(i, d.p) out.y[i] = r.y[i] * ci * D.dl[d.n + i];
    GRID_STRIDE(i, d.m) {
        const double zl = r.z_l[i] * ci * D.dl[d.n + d.p + i], zu = r.z_u[i] * ci * D.dl[d.n + d.p + i];
        out.z_l[i] = zl; out.z_u[i] = zu;
        out.s_l[i] = zl == 0 ? 1e30 : r.s_l[i] * D.dli[d.n + d.p + i];
        out.s_u[i] = zu == 0 ? 1e30 : r.s_u[i] * D.dli[d.n + d.p + i];
    }
    GRID_STRIDE(j, d.n) {
        const int il = M.pos_l[j], iu = M.pos_u[j];
        out.z_bl[j] = il >= 0 ? r.z_bl[il] * ci * D.db[j] : 0.0;
        out.s_bl[j] = il >= 0 ? r.s_bl[il] * D.dbi[j] : 1e30;
        out.z_bu[j] = iu >= 0 ? r.z_bu[iu] * ci * D.db[j] : 0.0;
        out.s_bu[j] = iu >= 0 ? r.s_bu[iu] * D.dbi[j] : 1e30;
    }
}

inline Ops make_ops(std::initializer_list<int> l)
{
    Ops o{};
    int k = 0;
    for (int v : l) o.op[k++] = v;
    return o;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ DeviceIpm
struct DeviceIpm::Impl {
    int n = 0, p = 0, m = 0, nxl = 0, nxu = 0, nhl = 0, nhu = 0;
    hipStream_t st = nullptr;
    std::vector<DBuf<double>> bufs;  // 6 Variables sets x 10 fields
    pq_vars R{}, NR{}, RS{}, ST{}, PX{}, OUT{};
    DBuf<double> c, b, h_l, h_u, x_l, x_u, xbs, dl, dli, db, dbi, work_x, work_z, part, scal;
    DBuf<int> has_l, has_u, pos_l, pos_u, x_l_idx, x_u_idx;
    HBuf<double> scal_h;
    Dims dims() const { return Dims{n, p, m, nxl, nxu}; }
    Masks masks() const { return Masks{has_l.p, has_u.p, pos_l.p, pos_u.p, x_l_idx.p, x_u_idx.p}; }
    DataV data() const { return DataV{c.p, b.p, h_l.p, h_u.p, x_l.p, x_u.p, xbs.p, dl.p, dli.p, db.p, dbi.p}; }
    int grid(int cnt) const { return std::max(1, std::min(MAXB, (cnt + NT - 1) / NT)); }
    int gmax() const { return grid(std::max(n, std::max(p, m))); }
    // finish a reduction phase: combine the block partials and bring K scalars to the host (one synchronisation)
    const double* reduce(int nblocks, int K, const Ops& ops)
    {
        reduce_on_device(nblocks, K, ops, 0);
        return fetch(K);
    }
    // phases that reduce in several kernels before the host needs anything use one slot each and fetch once
    double* part_slot(int slot) { return part.p + (size_t)slot * MAXB * NS; }
    void reduce_on_device(int nblocks, int K, const Ops& ops, int slot) { hipLaunchKernelGGL(k_final_reduce, dim3(1), dim3(64 * K), 0, st, nblocks, K, ops, part_slot(slot), scal.p + slot * NS, scal_h.p + slot * NS); }
    // reference-order mode: the listed dot products, each summed left to right by one wave; results at scal_h[3 * NS + q] once the stream has drained
    bool exact = false;
    void seq_dots(std::initializer_list<DotJob> jobs, double as = 0.0, double az = 0.0)
    {
        DotJobs J{};
        int k = 0;
        for (const DotJob& j : jobs) J.j[k++] = j;
        J.as = as; J.az = az;
        hipLaunchKernelGGL(k_seq_dots, dim3(k), dim3(64), 0, st, J, scal.p + 3 * NS, scal_h.p + 3 * NS);
    }
    const double* seq() const { return scal_h.p + 3 * NS; }
    const double* fetch(int count)
    {
        (void)count;  // k_final_reduce wrote the scalars into the pinned buffer itself (one device-to-host copy less per synchronisation, ~9 per iteration)
        stream_wait(st);
        return scal_h.p;
    }
};

DeviceIpm::DeviceIpm() : I(new Impl) {}
DeviceIpm::~DeviceIpm() = default;

void DeviceIpm::init(const HostData& d, const Ruiz& rz, hipStream_t st)
{
    Impl& s = *I;
    s.n = d.n; s.p = d.p; s.m = d.m; s.nxl = d.n_x_l; s.nxu = d.n_x_u; s.nhl = d.n_h_l; s.nhu = d.n_h_u; s.st = st;
    const int n = s.n, p = s.p, m = s.m;
    const size_t len[10] = {(size_t)n, (size_t)p, (size_t)m, (size_t)m, (size_t)n, (size_t)n, (size_t)m, (size_t)m, (size_t)n, (size_t)n};
    s.bufs.clear(); s.bufs.resize(60);
    pq_vars* sets[6] = {&s.R, &s.NR, &s.RS, &s.ST, &s.PX, &s.OUT};
    for (int q = 0; q < 6; ++q) {
        double** f[10] = {&sets[q]->x, &sets[q]->y, &sets[q]->z_l, &sets[q]->z_u, &sets[q]->z_bl, &sets[q]->z_bu, &sets[q]->s_l, &sets[q]->s_u, &sets[q]->s_bl, &sets[q]->s_bu};
        for (int k = 0; k < 10; ++k) { DBuf<double>& b = s.bufs[q * 10 + k]; b.alloc(std::max<size_t>(len[k], 1)); b.zero(st); *f[k] = b.p; }
    }
    s.work_x.alloc(std::max(n, 1)); s.work_z.alloc(std::max(m, 1));
    s.part.alloc((size_t)3 * MAXB * NS); s.scal.alloc(4 * NS); s.scal_h.alloc(4 * NS);
    s.has_l.alloc(std::max(m, 1)); s.has_u.alloc(std::max(m, 1)); s.pos_l.alloc(std::max(n, 1)); s.pos_u.alloc(std::max(n, 1)); s.x_l_idx.alloc(std::max(n, 1)); s.x_u_idx.alloc(std::max(n, 1));
    s.c.alloc(std::max(n, 1)); s.b.alloc(std::max(p, 1)); s.h_l.alloc(std::max(m, 1)); s.h_u.alloc(std::max(m, 1)); s.x_l.alloc(std::max(n, 1)); s.x_u.alloc(std::max(n, 1));
    s.xbs.alloc(std::max(n, 1)); s.dl.alloc(std::max(n + p + m, 1)); s.dli.alloc(std::max(n + p + m, 1)); s.db.alloc(std::max(n, 1)); s.dbi.alloc(std::max(n, 1));
    refresh_data(d, rz);
}

void DeviceIpm::refresh_data(const HostData& d, const Ruiz& rz)
{
    Impl& s = *I;
    const int n = s.n, m = s.m;
    s.nxl = d.n_x_l; s.nxu = d.n_x_u; s.nhl = d.n_h_l; s.nhu = d.n_h_u;
    std::vector<int> has_l(std::max(m, 1), 0), has_u(std::max(m, 1), 0), pos_l(std::max(n, 1), -1), pos_u(std::max(n, 1), -1);
    for (int i = 0; i < d.n_h_l; ++i) has_l[d.h_l_idx[i]] = 1;
    for (int i = 0; i < d.n_h_u; ++i) has_u[d.h_u_idx[i]] = 1;
    for (int i = 0; i < d.n_x_l; ++i) pos_l[d.x_l_idx[i]] = i;
    for (int i = 0; i < d.n_x_u; ++i) pos_u[d.x_u_idx[i]] = i;
    auto upi = [&](DBuf<int>& b, const int* v, size_t cnt) { if (cnt) PQ_HIP(hipMemcpy(b.p, v, cnt * sizeof(int), hipMemcpyHostToDevice)); };
    upi(s.has_l, has_l.data(), m); upi(s.has_u, has_u.data(), m); upi(s.pos_l, pos_l.data(), n); upi(s.pos_u, pos_u.data(), n);
    upi(s.x_l_idx, d.x_l_idx.data(), d.n_x_l); upi(s.x_u_idx, d.x_u_idx.data(), d.n_x_u);
    auto up = [&](DBuf<double>& b, const Vec& v, size_t cnt) { if (cnt) PQ_HIP(hipMemcpy(b.p, v.data(), cnt * sizeof(double), hipMemcpyHostToDevice)); };
    up(s.c, d.c, s.n); up(s.b, d.b, s.p); up(s.h_l, d.h_l, s.m); up(s.h_u, d.h_u, s.m); up(s.x_l, d.x_l, s.n); up(s.x_u, d.x_u, s.n); up(s.xbs, d.x_b_scaling, s.n);
    up(s.dl, rz.delta, (size_t)s.n + s.p + s.m); up(s.dli, rz.delta_inv, (size_t)s.n + s.p + s.m); up(s.db, rz.delta_b, s.n); up(s.dbi, rz.delta_b_inv, s.n);
}

void DeviceIpm::download_result(HostVars& out)
{
    Impl& s = *I;
    const pq_vars& o = s.OUT;
    const double* src[10] = {o.x, o.y, o.z_l, o.z_u, o.z_bl, o.z_bu, o.s_l, o.s_u, o.s_bl, o.s_bu};
    for (int k = 0; k < 10; ++k) { Vec& v = out.field(k); if (!v.empty()) PQ_HIP(hipMemcpyAsync(v.data(), src[k], v.size() * sizeof(double), hipMemcpyDeviceToHost, s.st)); }
    stream_wait(s.st);
}

// solve_impl (solver.hpp:379-882) with device-resident vectors
int DeviceIpm::solve(KKTSystem& kkt, const pq_settings& set, const Ruiz& rz, pq_info& info, double* trace, int trace_max, int* trace_rows)
{
    PQ_ZONE("piqp_amd::DeviceIpm::solve");  // solver.hpp:379 piqp::Solver::solve_impl
    Impl& s = *I;
    const Dims d = s.dims();
    const Masks M = s.masks();
    const DataV D = s.data();
    const int n = s.n, p = s.p, m = s.m;
    hipStream_t st = s.st;
    KKTSolverBase* be = kkt.backend();
    const double ci = rz.c_inv;
    const int G = s.gmax();
    const double ntot = (double)(s.nhl + s.nhu + s.nxl + s.nxu);
    const bool has_ineq = m + s.nxl + s.nxu > 0;
    bool refine = set.iterative_refinement_always_enabled != 0;
    const Ops ops_mu = make_ops({OP_SUM});
    s.exact = be->reference_order();
    // calculate_mu (solver.hpp:884-891) in the reference's order: four dot products, added in this order
    auto mu_jobs = [&]() { s.seq_dots({DotJob{s.R.s_l, s.R.z_l, nullptr, nullptr, m}, DotJob{s.R.s_u, s.R.z_u, nullptr, nullptr, m}, DotJob{s.R.s_bl, s.R.z_bl, nullptr, nullptr, s.nxl},
                                       DotJob{s.R.s_bu, s.R.z_bu, nullptr, nullptr, s.nxu}}); };
    auto sum4 = [](const double* q) { return ((q[0] + q[1]) + q[2]) + q[3]; };

    info.kkt_factor_time = 0; info.kkt_solve_time = 0;
    info.n_factor = info.n_solve = info.n_backend_solve = 0;
    info.status = PQ_UNSOLVED;
    info.iter = 0;
    info.reg_limit = set.reg_lower_limit;
    info.factor_retires = 0; info.no_primal_update = 0; info.no_dual_update = 0;
    info.mu = 0; info.primal_step = 0; info.dual_step = 0;
    info.rho = set.rho_init; info.delta = set.delta_init;

    auto now = []() -> double { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto factor = [&]() { info.n_factor++; return kkt.update_scalings_and_factor(refine, info.rho, info.delta, s.R); };
    auto ksolve = [&](const pq_vars& rhs, pq_vars& lhs) {
        const double t0 = now();
        info.n_solve++;
        kkt.solve(rhs, lhs);
        info.n_backend_solve += kkt.last_backend_solves;
        stream_wait(st);
        info.kkt_solve_time += now() - t0;
    };

    hipLaunchKernelGGL(k_init_sz, dim3(G), dim3(NT), 0, st, d, M, s.R);
    double t0 = now();
    while (!factor()) {
        if (!refine) refine = true;
        else if (info.factor_retires < set.max_factor_retires) {
            info.delta *= 100; info.rho *= 100; info.factor_retires++;
            info.reg_limit = std::min(10 * info.reg_limit, set.eps_abs);
        } else {
            info.status = PQ_NUMERICS;
            hipLaunchKernelGGL(k_finish, dim3(G), dim3(NT), 0, st, d, M, D, ci, s.R, s.OUT);
            return info.status;
        }
    }
    info.factor_retires = 0;
    info.kkt_factor_time += now() - t0;

    hipLaunchKernelGGL(k_init_rhs, dim3(G), dim3(NT), 0, st, d, D, s.RS);
    ksolve(s.RS, s.R);

    if (has_ineq) {
        const Ops o2 = make_ops({OP_MIN, OP_MIN});
        hipLaunchKernelGGL(k_min_sz, dim3(G), dim3(NT), 0, st, d, s.R, o2, s.part.p);
        const double* r2 = s.reduce(G, 2, o2);
        const double delta_s = std::max(0.0, -r2[0]), delta_z = std::max(0.0, -r2[1]);
        hipLaunchKernelGGL(k_shift_mu, dim3(G), dim3(NT), 0, st, d, M, s.R, delta_s, delta_z, ops_mu, s.part.p);
        if (s.exact) mu_jobs();
        info.mu = s.reduce(G, 1, ops_mu)[0] / ntot;
        if (s.exact) info.mu = sum4(s.seq()) / ntot;
        info.mu = std::max(info.mu, 1e-10);
        hipLaunchKernelGGL(k_centre_mu, dim3(G), dim3(NT), 0, st, d, M, s.R, info.mu, delta_z, ops_mu, s.part.p);
        if (s.exact) mu_jobs();
        info.mu = s.reduce(G, 1, ops_mu)[0] / ntot;
        if (s.exact) info.mu = sum4(s.seq()) / ntot;
    }
    hipLaunchKernelGGL(k_copy_prox, dim3(G), dim3(NT), 0, st, d, s.R, s.PX, 3);

    // update_residuals_nr; with_mu: also reduce sum s.z of the current iterate
    double mu_sum = 0.0;
    auto residuals_nr = [&](bool with_mu) {
        be->eval_A_xn_and_AT_xt(-1.0, 1.0, s.R.x, s.R.y, s.NR.y, s.work_x.p);
        if (m > 0) hipLaunchKernelGGL(k_nr_wz, dim3(G), dim3(NT), 0, st, d, s.R, s.work_z.p);
        be->eval_G_xn_and_GT_xt(1.0, 1.0, s.R.x, s.work_z.p, s.NR.z_l, s.NR.x);
        hipLaunchKernelGGL(k_nr_after_G, dim3(G), dim3(NT), 0, st, d, s.NR, s.work_x.p, s.NR.x);
        be->eval_P_x(-1.0, s.R.x, s.NR.x);
        if (s.exact)  // (before k_nr_x turns nr.x = -P x into the residual: solver.hpp:975 takes x . (P x) first)
            s.seq_dots({DotJob{s.R.x, s.NR.x, nullptr, nullptr, n}, DotJob{D.c, s.R.x, nullptr, nullptr, n}, DotJob{D.b, s.R.y, nullptr, nullptr, p}, DotJob{D.h_l, s.R.z_l, nullptr, nullptr, m},
                        DotJob{D.h_u, s.R.z_u, nullptr, nullptr, m}, DotJob{D.x_l, s.R.z_bl, nullptr, nullptr, s.nxl}, DotJob{D.x_u, s.R.z_bu, nullptr, nullptr, s.nxu},
                        DotJob{s.R.s_l, s.R.z_l, nullptr, nullptr, m}, DotJob{s.R.s_u, s.R.z_u, nullptr, nullptr, m}, DotJob{s.R.s_bl, s.R.z_bl, nullptr, nullptr, s.nxl},
                        DotJob{s.R.s_bu, s.R.z_bu, nullptr, nullptr, s.nxu}});
        const Ops ox = make_ops({OP_AMAXNAN, OP_SUM, OP_SUM, OP_AMAXNAN, OP_AMAXNAN});
        hipLaunchKernelGGL(k_nr_x, dim3(G), dim3(NT), 0, st, d, M, D, ci, s.R, s.NR, s.work_x.p, ox, s.part_slot(0));
        s.reduce_on_device(G, 5, ox, 0);
        const Ops oy = make_ops({OP_AMAXNAN, OP_SUM, OP_AMAXNAN, OP_SUM, OP_SUM, OP_MAX, OP_SUM, OP_SUM, OP_MAX});
        hipLaunchKernelGGL(k_nr_yz, dim3(G), dim3(NT), 0, st, d, M, D, s.R, s.NR, oy, s.part_slot(1));
        s.reduce_on_device(G, 9, oy, 1);
        const Ops orr = make_ops({OP_AMAXNAN, OP_AMAXNAN, OP_AMAXNAN, OP_MAX, OP_AMAXNAN, OP_SUM});
        hipLaunchKernelGGL(k_res_norms, dim3(G), dim3(NT), 0, st, d, M, D, ci, s.NR, s.R, with_mu ? 1 : 0, orr, s.part_slot(2));
        s.reduce_on_device(G, 6, orr, 2);
        const double* a = s.fetch(3 * NS);
        const double* bq = a + NS;
        const double* q = a + 2 * NS;
        double dots[8] = {a[1], a[2], bq[1], bq[3], bq[4], bq[6], bq[7], q[5]};  // x.Px', c.x, b.y, h_l.z_l, h_u.z_u, x_l.z_bl, x_u.z_bu, sum s.z
        if (s.exact) { const double* e = s.seq(); for (int t = 0; t < 7; ++t) dots[t] = e[t]; dots[7] = sum4(e + 7); }
        // objective / gap (:975-1013)
        double tmp = -dots[0];
        info.primal_obj = 0.5 * tmp;
        info.dual_obj = -0.5 * tmp;
        double dg_rel = ci * std::fabs(tmp);
        tmp = dots[1]; info.primal_obj += tmp; dg_rel = std::max(dg_rel, ci * std::fabs(tmp));
        tmp = dots[2]; info.dual_obj -= tmp; dg_rel = std::max(dg_rel, ci * std::fabs(tmp));
        tmp = -dots[3]; info.dual_obj -= tmp; dg_rel = std::max(dg_rel, ci * std::fabs(tmp));
        tmp = dots[4]; info.dual_obj -= tmp; dg_rel = std::max(dg_rel, ci * std::fabs(tmp));
        tmp = -dots[5]; info.dual_obj -= tmp; dg_rel = std::max(dg_rel, ci * std::fabs(tmp));
        tmp = dots[6]; info.dual_obj -= tmp; dg_rel = std::max(dg_rel, ci * std::fabs(tmp));
        info.duality_gap = std::fabs(info.primal_obj - info.dual_obj);
        info.primal_obj *= ci; info.dual_obj *= ci; info.duality_gap *= ci;
        info.duality_gap_rel = info.duality_gap / std::max(1.0, dg_rel);
        const double dual_rel_norm = std::max(a[0], std::max(a[3], a[4]));
        double primal_rel_norm = std::max(bq[0], bq[2]);
        if (s.nhl + s.nhu > 0) primal_rel_norm = std::max(primal_rel_norm, bq[5]);
        if (s.nxl + s.nxu > 0) primal_rel_norm = std::max(primal_rel_norm, bq[8]);
        info.prev_primal_res = info.primal_res;
        info.prev_dual_res = info.dual_res;
        double pr = std::max(q[0], std::max(q[1], q[2]));
        if (s.nxl + s.nxu > 0) pr = std::max(pr, q[3]);
        info.primal_res = pr;
        info.primal_res_rel = info.primal_res / std::max(1.0, primal_rel_norm);
        info.dual_res = q[4];
        info.dual_res_rel = info.dual_res / std::max(1.0, dual_rel_norm);
        mu_sum = dots[7];
    };
    auto residuals_r = [&]() {
        const Ops orr = make_ops({OP_AMAXNAN, OP_AMAXNAN, OP_AMAXNAN, OP_MAX, OP_AMAXNAN, OP_MAX, OP_MAX});
        hipLaunchKernelGGL(k_res_r, dim3(G), dim3(NT), 0, st, d, M, D, ci, info.rho, info.delta, s.R, s.NR, s.PX, s.RS, orr, s.part.p);
        const double* q = s.reduce(G, 7, orr);
        const double primal_rel_scaling = info.primal_res_rel > 0 ? info.primal_res / info.primal_res_rel : 1.0;
        const double dual_rel_scaling = info.dual_res_rel > 0 ? info.dual_res / info.dual_res_rel : 1.0;
        double pr = std::max(q[0], std::max(q[1], q[2]));
        if (s.nxl + s.nxu > 0) pr = std::max(pr, q[3]);
        info.primal_res_reg = pr;
        info.primal_res_reg_rel = info.primal_res_reg / primal_rel_scaling;
        info.dual_res_reg = q[4];
        info.dual_res_reg_rel = info.dual_res_reg / dual_rel_scaling;
        info.primal_prox_inf = std::max(0.0, q[5]) * info.delta;
        info.dual_prox_inf = std::max(0.0, q[6]) * info.rho;
    };

    while (info.iter < set.max_iter) {
        if (info.iter == 0) {
            residuals_nr(false);
            info.prev_primal_res = info.primal_res;
            info.prev_dual_res = info.dual_res;
        }
        if (trace && *trace_rows < trace_max) {
            double* row = trace + (size_t)(*trace_rows) * 11;
            row[0] = info.iter; row[1] = info.primal_obj; row[2] = info.dual_obj; row[3] = info.duality_gap; row[4] = info.primal_res; row[5] = info.dual_res;
            row[6] = info.rho; row[7] = info.delta; row[8] = info.mu; row[9] = info.primal_step; row[10] = info.dual_step;
            (*trace_rows)++;
        }
        if (set.verbose) {
            std::printf("%3d   % .5e   % .5e   %.5e   %.5e   %.5e   %.3e   %.3e   %.3e   %.4f   %.4f\n", info.iter, info.primal_obj, info.dual_obj, info.duality_gap,
                        info.primal_res, info.dual_res, info.rho, info.delta, info.mu, info.primal_step, info.dual_step);
            std::fflush(stdout);
        }
        if ((info.primal_res < set.eps_abs || info.primal_res_rel < set.eps_rel) && (info.dual_res < set.eps_abs || info.dual_res_rel < set.eps_rel) &&
            (!set.check_duality_gap || info.duality_gap < set.eps_duality_gap_abs || info.duality_gap_rel < set.eps_duality_gap_rel)) {
            info.status = PQ_SOLVED;
            break;
        }
        residuals_r();
        if (info.no_dual_update > std::min(5, set.reg_finetune_dual_update_threshold) && info.primal_prox_inf > set.infeasibility_threshold &&
            (info.primal_res_reg < set.eps_abs || info.primal_res_reg_rel < set.eps_rel)) {
            info.status = PQ_PRIMAL_INFEASIBLE;
            break;
        }
        if (info.no_primal_update > std::min(5, set.reg_finetune_primal_update_threshold) && info.dual_prox_inf > set.infeasibility_threshold &&
            (info.dual_res_reg < set.eps_abs || info.dual_res_reg_rel < set.eps_rel)) {
            info.status = PQ_DUAL_INFEASIBLE;
            break;
        }
        info.iter++;

        // :634-666 boundary shift (general rows in place; box vectors as a whole from their device-side minimum); the predictor rhs and
        // mu of the shifted iterate come out of the same pass
        if (has_ineq) {
            const Ops ob = make_ops({OP_MIN, OP_MIN});
            hipLaunchKernelGGL(k_boundary_min, dim3(G), dim3(NT), 0, st, d, M, s.R, DBL_EPSILON, ob, s.part_slot(1));
            s.reduce_on_device(G, 2, ob, 1);
            hipLaunchKernelGGL(k_boundary_apply_pred, dim3(G), dim3(NT), 0, st, d, s.R, s.RS, DBL_EPSILON, s.scal.p + NS, ops_mu, s.part_slot(0));
            if (s.exact) mu_jobs();
            info.mu = s.reduce(G, 1, ops_mu)[0] / ntot;  // equals the previous value bit for bit when nothing was shifted
            if (s.exact) info.mu = sum4(s.seq()) / ntot;
        }
        // :668-681
        if ((info.no_primal_update > set.reg_finetune_primal_update_threshold && info.rho == info.reg_limit && info.reg_limit != set.reg_finetune_lower_limit) ||
            (info.no_dual_update > set.reg_finetune_dual_update_threshold && info.delta == info.reg_limit && info.reg_limit != set.reg_finetune_lower_limit)) {
            if (info.dual_prox_inf < set.infeasibility_threshold && info.primal_prox_inf < set.infeasibility_threshold) {
                info.reg_limit = set.reg_finetune_lower_limit;
                info.no_primal_update = 0;
                info.no_dual_update = 0;
            }
        }
        t0 = now();
        bool regularization_changed = false;
        bool numerics = false;
        while (!factor()) {
            if (!refine) { refine = true; continue; }
            if (info.factor_retires < set.max_factor_retires) {
                info.delta *= 100; info.rho *= 100; info.factor_retires++;
                info.reg_limit = std::min(10 * info.reg_limit, set.eps_abs);
                regularization_changed = true;
                continue;
            }
            numerics = true;
            break;
        }
        if (numerics) { info.status = PQ_NUMERICS; break; }
        info.factor_retires = 0;
        info.kkt_factor_time += now() - t0;
        if (regularization_changed) residuals_r();  // rewrites the x / y / z rows of the rhs only; the predictor s rows stay

        if (has_ineq) {
            ksolve(s.RS, s.ST);
            const Ops os = make_ops({OP_MIN, OP_MIN});
            hipLaunchKernelGGL(k_step, dim3(G), dim3(NT), 0, st, d, s.R, s.ST, os, s.part.p);
            const double* a2 = s.reduce(G, 2, os);
            double alpha_s = a2[0] * set.tau, alpha_z = a2[1] * set.tau;
            hipLaunchKernelGGL(k_sigma, dim3(G), dim3(NT), 0, st, d, s.R, s.ST, alpha_s, alpha_z, ops_mu, s.part.p);
            if (s.exact)
                s.seq_dots({DotJob{s.R.s_l, s.R.z_l, s.ST.s_l, s.ST.z_l, m}, DotJob{s.R.s_u, s.R.z_u, s.ST.s_u, s.ST.z_u, m}, DotJob{s.R.s_bl, s.R.z_bl, s.ST.s_bl, s.ST.z_bl, s.nxl},
                            DotJob{s.R.s_bu, s.R.z_bu, s.ST.s_bu, s.ST.z_bu, s.nxu}}, alpha_s, alpha_z);
            double sigma = s.reduce(G, 1, ops_mu)[0];
            if (s.exact) sigma = sum4(s.seq());
            sigma /= info.mu * ntot;
            sigma = std::max(0.0, std::min(1.0, sigma));
            info.sigma = sigma * sigma * sigma;
            hipLaunchKernelGGL(k_corrector_rhs, dim3(G), dim3(NT), 0, st, d, s.ST, s.RS, info.sigma * info.mu);
            ksolve(s.RS, s.ST);
            hipLaunchKernelGGL(k_step, dim3(G), dim3(NT), 0, st, d, s.R, s.ST, os, s.part.p);
            a2 = s.reduce(G, 2, os);
            info.primal_step = a2[0] * set.tau;
            info.dual_step = a2[1] * set.tau;
            hipLaunchKernelGGL(k_update, dim3(G), dim3(NT), 0, st, d, s.R, s.ST, info.primal_step, info.dual_step);
            const double mu_prev = info.mu;
            residuals_nr(true);
            info.mu = mu_sum / ntot;
            const double mu_rate = std::max(0.0, (mu_prev - info.mu) / mu_prev);
            if (info.dual_res < 0.95 * info.prev_dual_res || (info.dual_res < set.eps_abs || info.dual_res_rel < set.eps_rel) ||
                (info.rho == set.reg_finetune_lower_limit && info.dual_prox_inf < set.infeasibility_threshold)) {
                hipLaunchKernelGGL(k_copy_prox, dim3(G), dim3(NT), 0, st, d, s.R, s.PX, 1);
                info.rho = std::max(info.reg_limit, (1.0 - mu_rate) * info.rho);
            } else {
                info.no_primal_update++;
                if (info.iter < 5 || info.dual_prox_inf < set.infeasibility_threshold) info.rho = std::max(info.reg_limit, (1.0 - 0.666 * mu_rate) * info.rho);
            }
            if (info.primal_res < 0.95 * info.prev_primal_res || (info.primal_res < set.eps_abs || info.primal_res_rel < set.eps_rel) ||
                (info.delta == set.reg_finetune_lower_limit && info.primal_prox_inf < set.infeasibility_threshold)) {
                hipLaunchKernelGGL(k_copy_prox, dim3(G), dim3(NT), 0, st, d, s.R, s.PX, 2);
                info.delta = std::max(info.reg_limit, (1.0 - mu_rate) * info.delta);
            } else {
                info.no_dual_update++;
                if (info.iter < 5 || info.primal_prox_inf < set.infeasibility_threshold) info.delta = std::max(info.reg_limit, (1.0 - 0.666 * mu_rate) * info.delta);
            }
        } else {
            // :831-877 no inequalities: one solve, full step
            ksolve(s.RS, s.ST);
            info.primal_step = 1.0; info.dual_step = 1.0;
            hipLaunchKernelGGL(k_update, dim3(G), dim3(NT), 0, st, d, s.R, s.ST, 1.0, 1.0);
            residuals_nr(false);
            if (info.dual_res < 0.95 * info.prev_dual_res || (info.dual_res < set.eps_abs || info.dual_res_rel < set.eps_rel)) {
                hipLaunchKernelGGL(k_copy_prox, dim3(G), dim3(NT), 0, st, d, s.R, s.PX, 1);
                info.rho = std::max(info.reg_limit, 0.1 * info.rho);
            } else {
                info.no_primal_update++;
                if (info.iter < 5 || info.dual_prox_inf < set.infeasibility_threshold) info.rho = std::max(info.reg_limit, 0.5 * info.rho);
            }
            if (info.primal_res < 0.95 * info.prev_primal_res || (info.primal_res < set.eps_abs || info.primal_res_rel < set.eps_rel)) {
                hipLaunchKernelGGL(k_copy_prox, dim3(G), dim3(NT), 0, st, d, s.R, s.PX, 2);
                info.delta = std::max(info.reg_limit, 0.1 * info.delta);
            } else {
                info.no_dual_update++;
                if (info.iter < 5 || info.primal_prox_inf < set.infeasibility_threshold) info.delta = std::max(info.reg_limit, 0.5 * info.delta);
            }
        }
    }
    if (info.status == PQ_UNSOLVED && info.iter >= set.max_iter) info.status = PQ_MAX_ITER_REACHED;
    hipLaunchKernelGGL(k_finish, dim3(G), dim3(NT), 0, st, d, M, D, ci, s.R, s.OUT);
    PQ_HIP(hipGetLastError());
    (void)n; (void)p;
    return info.status;
}

}  // namespace pq
