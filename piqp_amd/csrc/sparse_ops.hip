// piqp_amd/csrc/sparse_ops.hip -- see sparse_ops.hpp.
#include "sparse_ops.hpp"

#include <stdexcept>

namespace pq {

namespace {

inline dim3 g1(int n) { return dim3(n > 0 ? (n + 255) / 256 : 1); }

__global__ void k_remap_values(int nnz, const int* __restrict__ dst_idx, const double* __restrict__ src, double* __restrict__ dst)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nnz) dst[dst_idx[q]] = src[q];
}
__global__ void k_remap_values64(int nnz, const long long* __restrict__ dst_idx, const double* __restrict__ src, double* __restrict__ dst)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nnz) dst[dst_idx[q]] = src[q];
}
__global__ void k_gather_values(int nnz, const int* __restrict__ src_idx, const double* __restrict__ src, double* __restrict__ dst)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nnz) dst[q] = src[src_idx[q]];
}

// ---- column dots of a CSC matrix, one column per lane, with COALESCED traffic (round 3).
// A thread that walks its own column reads val[q], rowind[q] at addresses a whole column apart from its neighbours': every load instruction of the wave
// touches 64 cache lines (measured on the n = 500k chain: k_recover_duals 190 us and k_fold_rhs 110 us per call for 76 MB of matrix = 0.4 TB/s, a third of
// a KKT step).  Here the wave owns 64 consecutive columns, i.e. ONE contiguous range of entries: it streams that range through LDS in chunks -- lane e
// loads entries e, e + 64, ... (coalesced) together with the gathered operand -- and every lane then sums the part of ITS column that lies in the chunk,
// left to right, with the same multiply-add expression as before: bitwise the results of the thread-per-column loops (tests/test_sparse_gpu.py), at
// streaming bandwidth.  OP(q, i) = the operand of entry q in row i (x[i], or zinv[i] * rhs_z[i] for the folds).
constexpr int DOT_CHUNK = 512;               // entries per wave and chunk: 2 x 4 KB of LDS per wave
constexpr int DOT_LDS_DOUBLES = 2 * DOT_CHUNK * 4;  // per 256-thread workgroup
__device__ __forceinline__ void dot_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// returns sum_{q in [colptr[j], colptr[j + 1])} val[q] * op(rowind[q]) for this lane's column j (0 for j >= ncols or when skip says so); sm = the
// workgroup's DOT_LDS_DOUBLES doubles.  All 64 lanes of a wave must call it together.
template <class Op>
__device__ __forceinline__ double wave_col_dot(int j, int ncols, const int* __restrict__ colptr, const int* __restrict__ rowind, const double* __restrict__ val, Op op,
                                               double* __restrict__ sm, int max_len)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* sv = sm + wave * 2 * DOT_CHUNK;
    double* sx = sv + DOT_CHUNK;
    const bool in = j < ncols;
    int lo = in ? colptr[j] : 0, hi = in ? colptr[j + 1] : 0;
    if (max_len > 0 && hi - lo > max_len) hi = lo;  // (a column left to k_spmv_long_cols)
    const int j0 = j - lane;                                    // first column of the wave
    const int wlo = colptr[min(j0, ncols)], whi = colptr[min(j0 + 64, ncols)];
    double s = 0.0;
    for (int base = wlo; base < whi; base += DOT_CHUNK) {
        const int cnt = min(DOT_CHUNK, whi - base);
#pragma unroll 2
        for (int e = lane; e < cnt; e += 64) { const int q = base + e; sv[e] = val[q]; sx[e] = op(rowind[q]); }
        dot_wave_sync();
        const int a = max(lo, base), b = min(hi, base + cnt);
        for (int q = a; q < b; ++q) s += sv[q - base] * sx[q - base];
        dot_wave_sync();
    }
    return s;
}

// y[j] = (ACC ? y[j] : 0) + alpha * sum_q val[q] * x[row[q]] over column j  (CSC column dot)
constexpr int SPMV_LONG_COL = 2048;  // columns with more entries than this are summed by a whole workgroup (k_spmv_long_cols)
// ALPHA_FIRST: every term is val * (alpha * x[row]) and the sum is stored as it is -- the form of the reference's scatter products (sparse/kkt.hpp:188,199:
// z.noalias() += alpha * M * x walks the columns of M and adds val * (alpha x_j) to the target row, i.e. row by row a left-to-right sum of such terms);
// otherwise alpha * (sum of val * x[row]), the form of its transposed (column-dot) products.  max_len: columns with more entries are left to k_spmv_long_cols
// (0: none are -- the reference-order mode sums every column left to right).
template <bool ACC, bool ALPHA_FIRST>
__global__ __launch_bounds__(256) void k_spmv_cols(int ncols, const int* __restrict__ colptr, const int* __restrict__ rowind, const double* __restrict__ val, const double* __restrict__ x,
                                                   double alpha, double* __restrict__ y, int max_len)
{
    __shared__ double sm[DOT_LDS_DOUBLES];
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    double s;
    if (ALPHA_FIRST) s = wave_col_dot(j, ncols, colptr, rowind, val, [&](int i) { return alpha * x[i]; }, sm, max_len);
    else s = wave_col_dot(j, ncols, colptr, rowind, val, [&](int i) { return x[i]; }, sm, max_len);
    if (j >= ncols) return;
    if (max_len > 0 && colptr[j + 1] - colptr[j] > max_len) return;
    const double t = ALPHA_FIRST ? s : alpha * s;
    y[j] = ACC ? y[j] + t : t;
}
// eval_P_x in the reference's order (sparse/kkt.hpp:179-185: z = alpha * P_utri.selfadjointView<Upper>() * x as the CPU oracle restates it, oracle/orc_sparse.c
// sparse_eval_P_x): row r first collects val * (alpha x_j) over the stored entries (r, j), j >= r ascending -- the scatter pass over the upper triangle --
// and then receives alpha * (sum over the strictly upper part of column r of val * x_i).  One thread per row on the symmetrised copy, whose column r lists the
// rows i < r (upper part of column r) before r and the mirrored entries j > r.
__global__ __launch_bounds__(256) void k_sym_spmv_ref(int n, const int* __restrict__ colptr, const int* __restrict__ rowind, const double* __restrict__ val, const double* __restrict__ x,
                                                      double alpha, double* __restrict__ z)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    double acc = 0.0, s = 0.0;
    for (int q = colptr[r]; q < colptr[r + 1]; ++q) {
        const int i = rowind[q];
        if (i < r) s += val[q] * x[i];
        else acc += val[q] * (alpha * x[i]);
    }
    z[r] = acc + alpha * s;
}
// A column with very many entries (a dense row of A seen from its transpose copy: MM BOYD1 has 18 of 93 261 entries each) costs a thread-per-
// column kernel a serial loop -- 1.15 ms per mat-vec there.  One workgroup per such column: thread t sums the entries t, t + 256, ... in
// order, the 256 partial sums are added pairwise in a fixed tree.  Reproducible; not the left-to-right sum of the short columns.
template <bool ACC>
__global__ __launch_bounds__(256) void k_spmv_long_cols(const int* __restrict__ cols, const int* __restrict__ colptr, const int* __restrict__ rowind,
                                                        const double* __restrict__ val, const double* __restrict__ x, double alpha, double* __restrict__ y)
{
    __shared__ double part[256];
    const int j = cols[blockIdx.x], t = threadIdx.x;
    const int lo = colptr[j], hi = colptr[j + 1];
    double s = 0.0;
    for (int q = lo + t; q < hi; q += 256) s += val[q] * x[rowind[q]];
    part[t] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) part[t] += part[t + w];
        __syncthreads();
    }
    if (t == 0) y[j] = ACC ? y[j] + alpha * part[0] : alpha * part[0];
}

// out[j] = rhs_x[j] + sum_{G rows i of column j} G(i,j) zinv[i] rhs_z[i] + delta_inv * sum_{A rows i} A(i,j) rhs_y[i]
__global__ __launch_bounds__(256) void k_fold_rhs(int n, const int* __restrict__ Gp, const int* __restrict__ Gi, const double* __restrict__ Gx, const int* __restrict__ Ap, const int* __restrict__ Ai,
                                                  const double* __restrict__ Ax, const double* __restrict__ rhs_x, const double* __restrict__ rhs_y, const double* __restrict__ rhs_z,
                                                  const double* __restrict__ zinv, double delta_inv, double* __restrict__ out, int with_A, int with_G)
{
    __shared__ double sm[DOT_LDS_DOUBLES];
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    double sg = 0.0, sa = 0.0;
    if (with_G) sg = wave_col_dot(j, n, Gp, Gi, Gx, [&](int i) { return zinv[i] * rhs_z[i]; }, sm, 0);
    if (with_A) sa = wave_col_dot(j, n, Ap, Ai, Ax, [&](int i) { return rhs_y[i]; }, sm, 0);
    if (j >= n) return;
    out[j] = (rhs_x[j] + sg) + delta_inv * sa;
}
// the same fold in the reference's order (sparse/kkt.hpp:113-136 as the CPU oracle restates it, orc_sparse_cond.c cond_solve): the terms are added INTO rhs_x[j] one
// after the other -- first G(i, j) (zinv_i rhs_z_i) over the constraints i of column j, then A(i, j) (delta_inv rhs_y_i) -- one thread per row
__global__ __launch_bounds__(256) void k_fold_rhs_ref(int n, const int* __restrict__ Gp, const int* __restrict__ Gi, const double* __restrict__ Gx, const int* __restrict__ Ap,
                                                      const int* __restrict__ Ai, const double* __restrict__ Ax, const double* __restrict__ rhs_x, const double* __restrict__ rhs_y,
                                                      const double* __restrict__ rhs_z, const double* __restrict__ zinv, double delta_inv, double* __restrict__ out, int with_A, int with_G)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double s = rhs_x[j];
    if (with_G) for (int q = Gp[j]; q < Gp[j + 1]; ++q) { const int i = Gi[q]; s += Gx[q] * (zinv[i] * rhs_z[i]); }
    if (with_A) for (int q = Ap[j]; q < Ap[j + 1]; ++q) { const int i = Ai[q]; s += Ax[q] * (delta_inv * rhs_y[i]); }
    out[j] = s;
}
// rows k < p: lhs_y ; rows p <= k < p + m: lhs_z.  Two grids in one launch: blocks [0, ceil(p / 256)) serve the equality rows, the rest the inequality rows
// (a wave never straddles the two matrices).
__global__ __launch_bounds__(256) void k_recover_duals(int p, int m, const int* __restrict__ ATp, const int* __restrict__ ATi, const double* __restrict__ ATx, const int* __restrict__ GTp,
                                                       const int* __restrict__ GTi, const double* __restrict__ GTx, const double* __restrict__ x, const double* __restrict__ rhs_y,
                                                       const double* __restrict__ rhs_z, const double* __restrict__ zinv, double delta_inv, double* __restrict__ lhs_y, double* __restrict__ lhs_z,
                                                       int with_A, int with_G)
{
    __shared__ double sm[DOT_LDS_DOUBLES];
    const int pblocks = (p + 255) / 256;
    if ((int)blockIdx.x < pblocks) {
        if (!with_A) return;
        const int k = blockIdx.x * 256 + threadIdx.x;
        const double s = wave_col_dot(k, p, ATp, ATi, ATx, [&](int i) { return x[i]; }, sm, 0);
        if (k < p) lhs_y[k] = delta_inv * s - delta_inv * rhs_y[k];
    } else {
        if (!with_G) return;
        const int i = ((int)blockIdx.x - pblocks) * 256 + threadIdx.x;
        const double s = wave_col_dot(i, m, GTp, GTi, GTx, [&](int r) { return x[r]; }, sm, 0);
        if (i < m) lhs_z[i] = (s - rhs_z[i]) * zinv[i];
    }
}

// ---- refinement residual on listed rows (CscOperators::residual_rows) ----
__device__ __forceinline__ double col_dot_seq(int j, const int* __restrict__ colptr, const int* __restrict__ rowind, const double* __restrict__ val, const double* __restrict__ x)
{
    double s = 0.0;
    for (int q = colptr[j]; q < colptr[j + 1]; ++q) s += val[q] * x[rowind[q]];  // (the expression of wave_col_dot: same contraction, same order)
    return s;
}
__global__ __launch_bounds__(256) void k_residual_rows(const int* __restrict__ rows_x, int nx, const int* __restrict__ rows_y, int ny, const int* __restrict__ rows_z, int nz,
                                                       const int* __restrict__ Pp, const int* __restrict__ Pi, const double* __restrict__ Px, const int* __restrict__ ATp,
                                                       const int* __restrict__ ATi, const double* __restrict__ ATx, const int* __restrict__ Ap, const int* __restrict__ Ai,
                                                       const double* __restrict__ Ax, const int* __restrict__ GTp, const int* __restrict__ GTi, const double* __restrict__ GTx,
                                                       const int* __restrict__ Gp, const int* __restrict__ Gi, const double* __restrict__ Gx, const double* __restrict__ lhs_x,
                                                       const double* __restrict__ lhs_y, const double* __restrict__ lhs_z, const double* __restrict__ rhs_x,
                                                       const double* __restrict__ rhs_y, const double* __restrict__ rhs_z, const double* __restrict__ x_reg, double delta,
                                                       const double* __restrict__ z_reg, double* __restrict__ err_x, double* __restrict__ err_y, double* __restrict__ err_z,
                                                       unsigned long long* __restrict__ absmax_bits)
{
    __shared__ double red[4];
    const int t = blockIdx.x * 256 + threadIdx.x;
    double e = 0.0;
    if (t < nx) {
        const int i = rows_x[t];
        const double pxv = 1.0 * col_dot_seq(i, Pp, Pi, Px, lhs_x);
        const double aty = Ap ? 1.0 * col_dot_seq(i, Ap, Ai, Ax, lhs_y) : 0.0;
        const double gtz = Gp ? 1.0 * col_dot_seq(i, Gp, Gi, Gx, lhs_z) : 0.0;
        double v = pxv;             // k_err_x
        v += x_reg[i] * lhs_x[i];
        v += aty;
        v += gtz;
        e = rhs_x[i] - v;
        err_x[i] = e;
    } else if (t < nx + ny) {
        const int j = rows_y[t - nx];
        double v = 1.0 * col_dot_seq(j, ATp, ATi, ATx, lhs_x);  // k_err_yz with the scalar delta
        v -= delta * lhs_y[j];
        e = rhs_y[j] - v;
        err_y[j] = e;
    } else if (t < nx + ny + nz) {
        const int k = rows_z[t - nx - ny];
        double v = 1.0 * col_dot_seq(k, GTp, GTi, GTx, lhs_x);
        v -= z_reg[k] * lhs_z[k];
        e = rhs_z[k] - v;
        err_z[k] = e;
    }
    double a = fabs(e);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(a, off, 64);
        a = (a != a || o != o) ? __builtin_nan("") : (a > o ? a : o);
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = red[0];
        for (int k = 1; k < 4; ++k) r = (r != r || red[k] != red[k]) ? __builtin_nan("") : (red[k] > r ? red[k] : r);
        atomicMax(absmax_bits, (unsigned long long)__double_as_longlong(r != r ? __builtin_nan("") : r) & 0x7fffffffffffffffull);
    }
}

// ---- the condensed right-hand side and the dual recovery on listed rows (CscOperators::fold_rhs_rows / recover_duals_rows) ----
__global__ __launch_bounds__(256) void k_fold_rhs_rows(const int* __restrict__ rows, int nrows, const int* __restrict__ Gp, const int* __restrict__ Gi, const double* __restrict__ Gx,
                                                       const int* __restrict__ Ap, const int* __restrict__ Ai, const double* __restrict__ Ax, const double* __restrict__ rhs_x,
                                                       const double* __restrict__ rhs_y, const double* __restrict__ rhs_z, const double* __restrict__ zinv, double delta_inv,
                                                       double* __restrict__ out, int with_A, int with_G, int ref_order)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nrows) return;
    const int j = rows[t];
    if (ref_order) {  // k_fold_rhs_ref
        double s = rhs_x[j];
        if (with_G) for (int q = Gp[j]; q < Gp[j + 1]; ++q) { const int i = Gi[q]; s += Gx[q] * (zinv[i] * rhs_z[i]); }
        if (with_A) for (int q = Ap[j]; q < Ap[j + 1]; ++q) { const int i = Ai[q]; s += Ax[q] * (delta_inv * rhs_y[i]); }
        out[j] = s;
        return;
    }
    double sg = 0.0, sa = 0.0;  // k_fold_rhs
    if (with_G) for (int q = Gp[j]; q < Gp[j + 1]; ++q) { const int i = Gi[q]; sg += Gx[q] * (zinv[i] * rhs_z[i]); }
    if (with_A) for (int q = Ap[j]; q < Ap[j + 1]; ++q) sa += Ax[q] * rhs_y[Ai[q]];
    out[j] = (rhs_x[j] + sg) + delta_inv * sa;
}
__global__ __launch_bounds__(256) void k_recover_duals_rows(const int* __restrict__ rows_y, int ny, const int* __restrict__ rows_z, int nz, const int* __restrict__ ATp,
                                                            const int* __restrict__ ATi, const double* __restrict__ ATx, const int* __restrict__ GTp, const int* __restrict__ GTi,
                                                            const double* __restrict__ GTx, const double* __restrict__ x, const double* __restrict__ rhs_y,
                                                            const double* __restrict__ rhs_z, const double* __restrict__ zinv, double delta_inv, double* __restrict__ lhs_y,
                                                            double* __restrict__ lhs_z)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ny) {
        const int k = rows_y[t];
        const double s = col_dot_seq(k, ATp, ATi, ATx, x);
        lhs_y[k] = delta_inv * s - delta_inv * rhs_y[k];
    } else if (t < ny + nz) {
        const int i = rows_z[t - ny];
        const double s = col_dot_seq(i, GTp, GTi, GTx, x);
        lhs_z[i] = (s - rhs_z[i]) * zinv[i];
    }
}

// host-side CSC transpose with a value map: T = M^T, tmap[q_in_T] = q_in_M
void transpose_with_map(int rows, int cols, const int* Mp, const int* Mi, std::vector<int>& Tp, std::vector<int>& Ti, std::vector<int>& tmap)
{
    const int nnz = Mp[cols];
    Tp.assign(rows + 1, 0); Ti.assign(nnz, 0); tmap.assign(nnz, 0);
    for (int q = 0; q < nnz; ++q) Tp[Mi[q] + 1]++;
    for (int i = 0; i < rows; ++i) Tp[i + 1] += Tp[i];
    std::vector<int> nx(Tp.begin(), Tp.end() - 1);
    for (int j = 0; j < cols; ++j) for (int q = Mp[j]; q < Mp[j + 1]; ++q) { const int t = nx[Mi[q]]++; Ti[t] = j; tmap[t] = q; }
}

}  // namespace

void launch_remap_values(int nnz, const int* dst_idx, const double* src, double* dst, hipStream_t st)
{
    if (nnz > 0) hipLaunchKernelGGL(k_remap_values, g1(nnz), dim3(256), 0, st, nnz, dst_idx, src, dst);
}
void launch_remap_values64(int nnz, const long long* dst_idx, const double* src, double* dst, hipStream_t st)
{
    if (nnz > 0) hipLaunchKernelGGL(k_remap_values64, g1(nnz), dim3(256), 0, st, nnz, dst_idx, src, dst);
}
void launch_gather_values(int nnz, const int* src_idx, const double* src, double* dst, hipStream_t st)
{
    if (nnz > 0) hipLaunchKernelGGL(k_gather_values, g1(nnz), dim3(256), 0, st, nnz, src_idx, src, dst);
}

void CscOperators::init(const pq_sparse_data* d, hipStream_t st)
{
    n_ = d->n; p_ = d->p; m_ = d->m;
    nzP_ = d->P_colptr[n_]; nzA_ = p_ ? d->AT_colptr[p_] : 0; nzG_ = m_ ? d->GT_colptr[m_] : 0;
    // symmetric completion of P: pattern + source index of every entry
    {
        std::vector<int> cnt(n_ + 1, 0);
        for (int j = 0; j < n_; ++j) for (int q = d->P_colptr[j]; q < d->P_colptr[j + 1]; ++q) { const int i = d->P_rowind[q]; cnt[j + 1]++; if (i != j) cnt[i + 1]++; }
        std::vector<int> fp(n_ + 1, 0);
        for (int j = 0; j < n_; ++j) fp[j + 1] = fp[j] + cnt[j + 1];
        std::vector<int> fi(fp[n_]), src(fp[n_]), nx(fp.begin(), fp.end() - 1);
        // rows ascending in every column: first the upper entries of column j (rows <= j), later the mirrored ones (rows > j)
        for (int j = 0; j < n_; ++j) for (int q = d->P_colptr[j]; q < d->P_colptr[j + 1]; ++q) { const int t = nx[j]++; fi[t] = d->P_rowind[q]; src[t] = q; }
        for (int j = 0; j < n_; ++j) for (int q = d->P_colptr[j]; q < d->P_colptr[j + 1]; ++q) { const int i = d->P_rowind[q]; if (i != j) { const int t = nx[i]++; fi[t] = j; src[t] = q; } }
        upload_vec(Pf_p_, fp, st); upload_vec(Pf_i_, fi, st); upload_vec(Pf_src_, src, st);
        nzPf_ = fp[n_];
        Pf_x_.alloc(nzPf_ ? nzPf_ : 1);
    }
    P_x_.alloc(nzP_ ? nzP_ : 1);
    Pdiag_.alloc(n_ ? n_ : 1);
    // AT (n x p) and its transpose A (p x n); GT (n x m) and G
    {
        std::vector<int> atp(d->AT_colptr, d->AT_colptr + p_ + 1), ati(d->AT_rowind, d->AT_rowind + nzA_);
        if (!p_) atp.assign(1, 0);
        upload_vec(AT_p_, atp, st); upload_vec(AT_i_, ati, st); AT_x_.alloc(nzA_ ? nzA_ : 1);
        std::vector<int> tp, ti, tm;
        transpose_with_map(n_, p_, atp.data(), ati.data(), tp, ti, tm);
        upload_vec(A_p_, tp, st); upload_vec(A_i_, ti, st); upload_vec(A_src_, tm, st); A_x_.alloc(nzA_ ? nzA_ : 1);
        std::vector<int> gtp(d->GT_colptr, d->GT_colptr + m_ + 1), gti(d->GT_rowind, d->GT_rowind + nzG_);
        if (!m_) gtp.assign(1, 0);
        upload_vec(GT_p_, gtp, st); upload_vec(GT_i_, gti, st); GT_x_.alloc(nzG_ ? nzG_ : 1);
        transpose_with_map(n_, m_, gtp.data(), gti.data(), tp, ti, tm);
        upload_vec(G_p_, tp, st); upload_vec(G_i_, ti, st); upload_vec(G_src_, tm, st); G_x_.alloc(nzG_ ? nzG_ : 1);
    }
    // columns a single thread should not sum alone (k_spmv_long_cols)
    {
        auto longs = [&](const DBuf<int>& cp, int ncols, DBuf<int>& out, int& cnt) {
            std::vector<int> h(ncols + 1, 0), l;
            if (ncols > 0) PQ_HIP(hipMemcpy(h.data(), cp.p, sizeof(int) * (ncols + 1), hipMemcpyDeviceToHost));
            for (int j = 0; j < ncols; ++j) if (h[j + 1] - h[j] > SPMV_LONG_COL) l.push_back(j);
            cnt = (int)l.size();
            if (l.empty()) l.push_back(0);
            upload_vec(out, l, st);
        };
        stream_wait(st);
        longs(Pf_p_, n_, long_Pf_, nlong_[0]); longs(AT_p_, p_, long_AT_, nlong_[1]); longs(A_p_, n_, long_A_, nlong_[2]);
        longs(GT_p_, m_, long_GT_, nlong_[3]); longs(G_p_, n_, long_G_, nlong_[4]);
    }
    upload_values(d, st);
}

void CscOperators::upload_values(const pq_sparse_data* d, hipStream_t st)
{
    if (d->n != n_ || d->p != p_ || d->m != m_) throw std::runtime_error("update_data: dimension mismatch");
    if (nzP_) {
        PQ_HIP(hipMemcpyAsync(P_x_.p, d->P_val, sizeof(double) * nzP_, hipMemcpyHostToDevice, st));
        launch_gather_values(nzPf_, Pf_src_.p, P_x_.p, Pf_x_.p, st);
    }
    std::vector<double> pd(n_, 0.0);
    for (int j = 0; j < n_; ++j) for (int q = d->P_colptr[j]; q < d->P_colptr[j + 1]; ++q) if (d->P_rowind[q] == j) pd[j] = d->P_val[q];
    if (n_) PQ_HIP(hipMemcpyAsync(Pdiag_.p, pd.data(), sizeof(double) * n_, hipMemcpyHostToDevice, st));
    if (nzA_) {
        PQ_HIP(hipMemcpyAsync(AT_x_.p, d->AT_val, sizeof(double) * nzA_, hipMemcpyHostToDevice, st));
        launch_gather_values(nzA_, A_src_.p, AT_x_.p, A_x_.p, st);
    }
    if (nzG_) {
        PQ_HIP(hipMemcpyAsync(GT_x_.p, d->GT_val, sizeof(double) * nzG_, hipMemcpyHostToDevice, st));
        launch_gather_values(nzG_, G_src_.p, GT_x_.p, G_x_.p, st);
    }
    PQ_HIP(hipGetLastError());
    stream_wait(st);  // `pd` and the caller's arrays must outlive the copies
}

void CscOperators::clone_from(const CscOperators& o, hipStream_t st)
{
    n_ = o.n_; p_ = o.p_; m_ = o.m_; nzP_ = o.nzP_; nzA_ = o.nzA_; nzG_ = o.nzG_; nzPf_ = o.nzPf_;
    auto cpd = [&](DBuf<double>& d, const DBuf<double>& s) { d.alloc(s.n ? s.n : 1); if (s.n) PQ_HIP(hipMemcpyAsync(d.p, s.p, s.bytes(), hipMemcpyDeviceToDevice, st)); };
    auto cpi = [&](DBuf<int>& d, const DBuf<int>& s) { d.alloc(s.n ? s.n : 1); if (s.n) PQ_HIP(hipMemcpyAsync(d.p, s.p, s.bytes(), hipMemcpyDeviceToDevice, st)); };
    cpd(P_x_, o.P_x_); cpd(Pf_x_, o.Pf_x_); cpd(AT_x_, o.AT_x_); cpd(A_x_, o.A_x_); cpd(GT_x_, o.GT_x_); cpd(G_x_, o.G_x_); cpd(Pdiag_, o.Pdiag_);
    cpi(Pf_p_, o.Pf_p_); cpi(Pf_i_, o.Pf_i_); cpi(Pf_src_, o.Pf_src_); cpi(AT_p_, o.AT_p_); cpi(AT_i_, o.AT_i_); cpi(A_p_, o.A_p_); cpi(A_i_, o.A_i_);
    cpi(A_src_, o.A_src_); cpi(GT_p_, o.GT_p_); cpi(GT_i_, o.GT_i_); cpi(G_p_, o.G_p_); cpi(G_i_, o.G_i_); cpi(G_src_, o.G_src_);
    cpi(long_Pf_, o.long_Pf_); cpi(long_AT_, o.long_AT_); cpi(long_A_, o.long_A_); cpi(long_GT_, o.long_GT_); cpi(long_G_, o.long_G_);
    for (int q = 0; q < 5; ++q) nlong_[q] = o.nlong_[q];
    ref_order_ = o.ref_order_;
}

// scatter: the product the reference forms by walking columns and adding into the target (M * x for a column-major M); ref: reference-order mode
template <bool ACC>
static void spmv(int ncols, const DBuf<int>& cp, const DBuf<int>& ri, const DBuf<double>& v, const DBuf<int>& long_cols, int nlong, const double* x, double alpha, double* y,
                 hipStream_t st, bool ref = false, bool scatter = false)
{
    if (ncols <= 0) return;
    if (ref) {
        if (scatter) hipLaunchKernelGGL((k_spmv_cols<ACC, true>), g1(ncols), dim3(256), 0, st, ncols, cp.p, ri.p, v.p, x, alpha, y, 0);
        else hipLaunchKernelGGL((k_spmv_cols<ACC, false>), g1(ncols), dim3(256), 0, st, ncols, cp.p, ri.p, v.p, x, alpha, y, 0);
        return;
    }
    hipLaunchKernelGGL((k_spmv_cols<ACC, false>), g1(ncols), dim3(256), 0, st, ncols, cp.p, ri.p, v.p, x, alpha, y, SPMV_LONG_COL);
    if (nlong > 0) hipLaunchKernelGGL(k_spmv_long_cols<ACC>, dim3(nlong), dim3(256), 0, st, long_cols.p, cp.p, ri.p, v.p, x, alpha, y);
}

void CscOperators::eval_P_x(double alpha, const double* x, double* z, hipStream_t st) const
{
    if (ref_order_) { if (n_ > 0) hipLaunchKernelGGL(k_sym_spmv_ref, g1(n_), dim3(256), 0, st, n_, Pf_p_.p, Pf_i_.p, Pf_x_.p, x, alpha, z); return; }
    spmv<false>(n_, Pf_p_, Pf_i_, Pf_x_, long_Pf_, nlong_[0], x, alpha, z, st);
}
void CscOperators::eval_A_xn_and_AT_xt(double an, double at, const double* xn, const double* xt, double* zn, double* zt, hipStream_t st) const
{
    spmv<false>(p_, AT_p_, AT_i_, AT_x_, long_AT_, nlong_[1], xn, an, zn, st, ref_order_, false);  // A x = (AT)^T x
    spmv<false>(n_, A_p_, A_i_, A_x_, long_A_, nlong_[2], xt, at, zt, st, ref_order_, true);       // AT y = (A)^T y
}
void CscOperators::eval_G_xn_and_GT_xt(double an, double at, const double* xn, const double* xt, double* zn, double* zt, hipStream_t st) const
{
    spmv<false>(m_, GT_p_, GT_i_, GT_x_, long_GT_, nlong_[3], xn, an, zn, st, ref_order_, false);
    spmv<false>(n_, G_p_, G_i_, G_x_, long_G_, nlong_[4], xt, at, zt, st, ref_order_, true);
}
bool CscOperators::residual_rows(const int* rows_x, int nx, const int* rows_y, int ny, const int* rows_z, int nz, const double* lhs_x, const double* lhs_y, const double* lhs_z,
                                 const double* rhs_x, const double* rhs_y, const double* rhs_z, const double* x_reg, double delta, const double* z_reg, double* err_x, double* err_y,
                                 double* err_z, unsigned long long* absmax_bits, hipStream_t st) const
{
    if (has_long_columns()) return false;
    const int tot = nx + ny + nz;
    if (tot <= 0) return true;
    hipLaunchKernelGGL(k_residual_rows, g1(tot), dim3(256), 0, st, rows_x, nx, rows_y, ny, rows_z, nz, Pf_p_.p, Pf_i_.p, Pf_x_.p, AT_p_.p, AT_i_.p, AT_x_.p, p_ > 0 ? A_p_.p : (const int*)nullptr,
                       A_i_.p, A_x_.p, GT_p_.p, GT_i_.p, GT_x_.p, m_ > 0 ? G_p_.p : (const int*)nullptr, G_i_.p, G_x_.p, lhs_x, lhs_y, lhs_z, rhs_x, rhs_y, rhs_z, x_reg, delta, z_reg,
                       err_x, err_y, err_z, absmax_bits);
    PQ_HIP(hipGetLastError());
    return true;
}
void CscOperators::fold_rhs(const double* rhs_x, const double* rhs_y, const double* rhs_z, const double* zinv, double delta_inv, double* out, hipStream_t st, bool with_A,
                            bool with_G) const
{
    if (ref_order_) {
        hipLaunchKernelGGL(k_fold_rhs_ref, g1(n_), dim3(256), 0, st, n_, G_p_.p, G_i_.p, G_x_.p, A_p_.p, A_i_.p, A_x_.p, rhs_x, rhs_y, rhs_z, zinv, delta_inv, out, with_A ? 1 : 0,
                           with_G ? 1 : 0);
        return;
    }
    hipLaunchKernelGGL(k_fold_rhs, g1(n_), dim3(256), 0, st, n_, G_p_.p, G_i_.p, G_x_.p, A_p_.p, A_i_.p, A_x_.p, rhs_x, rhs_y, rhs_z, zinv, delta_inv, out, with_A ? 1 : 0,
                       with_G ? 1 : 0);
}
void CscOperators::recover_duals(const double* x, const double* rhs_y, const double* rhs_z, const double* zinv, double delta_inv, double* lhs_y, double* lhs_z, hipStream_t st,
                                 bool with_A, bool with_G) const
{
    if (p_ + m_ > 0)
        hipLaunchKernelGGL(k_recover_duals, dim3((p_ + 255) / 256 + (m_ + 255) / 256 > 0 ? (p_ + 255) / 256 + (m_ + 255) / 256 : 1), dim3(256), 0, st, p_, m_, AT_p_.p, AT_i_.p, AT_x_.p, GT_p_.p, GT_i_.p, GT_x_.p, x, rhs_y, rhs_z, zinv, delta_inv, lhs_y, lhs_z,
                           with_A ? 1 : 0, with_G ? 1 : 0);
}
void CscOperators::fold_rhs_rows(const int* rows_x, int nx, const double* rhs_x, const double* rhs_y, const double* rhs_z, const double* zinv, double delta_inv, double* out,
                                 hipStream_t st, bool with_A, bool with_G) const
{
    if (nx > 0)
        hipLaunchKernelGGL(k_fold_rhs_rows, g1(nx), dim3(256), 0, st, rows_x, nx, G_p_.p, G_i_.p, G_x_.p, A_p_.p, A_i_.p, A_x_.p, rhs_x, rhs_y, rhs_z, zinv, delta_inv, out, with_A ? 1 : 0,
                           with_G ? 1 : 0, ref_order_ ? 1 : 0);
}
void CscOperators::recover_duals_rows(const int* rows_y, int ny, const int* rows_z, int nz, const double* x, const double* rhs_y, const double* rhs_z, const double* zinv,
                                      double delta_inv, double* lhs_y, double* lhs_z, hipStream_t st) const
{
    if (ny + nz > 0)
        hipLaunchKernelGGL(k_recover_duals_rows, g1(ny + nz), dim3(256), 0, st, rows_y, ny, rows_z, nz, AT_p_.p, AT_i_.p, AT_x_.p, GT_p_.p, GT_i_.p, GT_x_.p, x, rhs_y, rhs_z, zinv,
                           delta_inv, lhs_y, lhs_z);
}
void CscOperators::add_AT_y(double alpha, const double* y, double* z, hipStream_t st) const
{
    if (p_ > 0) spmv<true>(n_, A_p_, A_i_, A_x_, long_A_, nlong_[2], y, alpha, z, st);
}
void CscOperators::add_GT_y(double alpha, const double* y, double* z, hipStream_t st) const
{
    if (m_ > 0) spmv<true>(n_, G_p_, G_i_, G_x_, long_G_, nlong_[4], y, alpha, z, st);
}

}  // namespace pq
