// piqp_amd/csrc/multistage_device.hpp -- workgroup-collective device routines of the sparse_multistage
// backend (reference sparse/multistage_kkt.hpp).  Every routine is executed by ALL threads of one workgroup of
// NT threads on the data of ONE QP; the single-QP kernels of multistage_kkt.hip and the batched
// interior-point kernel of batch_solver.hip are thin shells around them.
//
//   gram_stage / assemble_stage   block_syrk_ln_calc + populate_kkt_fac (:832-994, :1008-1219): lower triangle of a stage's frontal
//                                 matrix  P + diag(x_reg) + delta^-1 X_A X_A^T + X_G diag(1/z_reg) X_G^T
//   factor_chain                  factor_kkt (:1253-1352): panel Cholesky per stage, Schur complement carried to the next stage
//                                 and the arrow corner as a multifrontal update matrix, explicit inverse of every L_ii
//   solve_chain                   solve_llt_in_place (:1709-1816): forward and backward block substitution
#pragma once

#include <hip/hip_runtime.h>

namespace pq {
namespace msdev {

struct MsMeta {  // device views of multistage::Symbolic
    int N, arrow, n;
    const int* w;
    const int* off;
    const int* h;
    const int* start;
    const long long* front_off;
    const long long* pan_off;
    __device__ __forceinline__ int W(int b) const { return w[b]; }
    __device__ __forceinline__ int Off(int b) const { return off[b]; }
    __device__ __forceinline__ int H(int b) const { return h[b]; }
    __device__ __forceinline__ int Start(int b) const { return start[b]; }
    __device__ __forceinline__ long long FrontOff(int b) const { return front_off[b]; }
    __device__ __forceinline__ long long PanOff(int b) const { return pan_off[b]; }
};
// the same tables as one packed copy (w | off | h | start as int, then front_off | pan_off as int64); the batched kernel
// keeps it in LDS and builds this view from the `extern __shared__` symbol so the compiler emits ds_read, not flat loads
struct PackedMeta {
    int N, arrow, n;
    const int* mi;
    const long long* ml;
    __device__ __forceinline__ int W(int b) const { return mi[b]; }
    __device__ __forceinline__ int Off(int b) const { return mi[N + b]; }
    __device__ __forceinline__ int H(int b) const { return mi[2 * N + b]; }
    __device__ __forceinline__ int Start(int b) const { return mi[3 * N + b]; }
    __device__ __forceinline__ long long FrontOff(int b) const { return ml[b]; }
    __device__ __forceinline__ long long PanOff(int b) const { return ml[N + b]; }
};
struct GroupMeta {
    const int* row_ptr;  // N entries
    const int* rows;     // grouped position -> caller's constraint index
    const long long* x_off;
};

// entries [lo, hi) of the lower triangle of X_b X_b^T
template <int NT, class Meta>
__device__ __forceinline__ void gram_stage(const Meta& M, const GroupMeta& Gm, const double* __restrict__ X, double* __restrict__ out, int b, int lo, int hi)
{
    const int h = M.H(b);
    const int rows = Gm.row_ptr[b + 1] - Gm.row_ptr[b];
    const double* Xb = X + Gm.x_off[b];
    double* O = out + M.FrontOff(b);
    for (int idx = lo + (int)threadIdx.x; idx < hi; idx += NT) {
        const int r = idx % h, c = idx / h;
        if (r < c) continue;
        double s = 0.0;
        for (int k = 0; k < rows; ++k) s += Xb[r + (long long)k * h] * Xb[c + (long long)k * h];
        O[idx] = s;
    }
}

// entries [lo, hi) of stage b's frontal matrix (lower triangle); block N-1 is the arrow corner (P + x_reg only: every
// product that lands there is carried by the stage fronts)
template <int NT, class Meta>
__device__ __forceinline__ void assemble_stage(const Meta& M, const GroupMeta& Gm, const double* __restrict__ XG, const double* __restrict__ Pf,
                                               const double* __restrict__ AtAf, const double* __restrict__ zinv, const double* __restrict__ x_reg, double delta_inv,
                                               double* __restrict__ F, int b, int lo, int hi)
{
    const int h = M.H(b), w = M.W(b);
    const bool corner = b == M.N - 1;
    const int rows = corner ? 0 : Gm.row_ptr[b + 1] - Gm.row_ptr[b];
    const double* Xb = XG + (corner ? 0 : Gm.x_off[b]);
    const int* rid = Gm.rows + (corner ? 0 : Gm.row_ptr[b]);
    const long long fo = M.FrontOff(b);
    const int start = M.Start(b);
    for (int idx = lo + (int)threadIdx.x; idx < hi; idx += NT) {
        const int r = idx % h, c = idx / h;
        if (r < c) continue;
        double s = 0.0;
        for (int k = 0; k < rows; ++k) s += Xb[r + (long long)k * h] * zinv[rid[k]] * Xb[c + (long long)k * h];
        double v = Pf[fo + idx] + delta_inv * AtAf[fo + idx] + s;
        if (r == c && c < w) v += x_reg[start + c];
        F[fo + idx] = v;
    }
}

// position inside front b of the t-th row of the update matrix carried from stage b-1 ([off_{b-1} | arrow])
__device__ __forceinline__ int carry_row(int t, int off_prev, int w, int offb, bool corner)
{
    if (t < off_prev) return t;
    return corner ? t - off_prev : w + offb + (t - off_prev);
}

// One workgroup walks the block-tridiagonal-arrow chain.  Per stage b:
//   front += carried update;  [L_b; C_b; F_b] = panel Cholesky of the first w_b columns (a non-positive pivot zeroes its column);
//   carried update = trailing block - [C_b; F_b][C_b; F_b]^T;  inverse of L_b for the solves.
// LDS = true: front (sm[0..fcap)), carried update (sm[fcap..lofs)) and inverse (sm[lofs..)) live in LDS;
// false: the front is factored in place in HBM/L2 and only the inverse is staged in LDS (when li_in_lds).
template <int NT, bool LDS, class Meta>
__device__ __forceinline__ void factor_chain(const Meta& M, double* __restrict__ fronts, double* __restrict__ pan, double* sm, int fcap, int lofs, int li_in_lds)
{
    const int tid = threadIdx.x;
    const int N = M.N;
    int u_prev = 0, off_prev = 0, ldu = 0;
    double* Usrc = nullptr;
    for (int b = 0; b < N; ++b) {
        const int h = M.H(b), w = M.W(b);
        if (h == 0) break;  // no arrow corner
        const bool corner = b == N - 1;
        const int offb = M.Off(b);
        const int u = h - w;
        double* Fg = fronts + M.FrontOff(b);
        double* P = pan + M.PanOff(b);
        double* Li = P + (long long)h * w;
        double* F;
        if constexpr (LDS) {
            F = sm;
            for (int idx = tid; idx < h * h; idx += NT) F[idx] = Fg[idx];
            __syncthreads();
        } else {
            F = Fg;
        }
        const int ld = h;
        if (u_prev > 0) {  // extend-add of the carried update matrix (distinct targets -> no conflicts)
            for (int idx = tid; idx < u_prev * u_prev; idx += NT) {
                const int i = idx % u_prev, j = idx / u_prev;
                if (i < j) continue;
                F[carry_row(i, off_prev, w, offb, corner) + carry_row(j, off_prev, w, offb, corner) * ld] += Usrc[i + j * ldu];
            }
            __syncthreads();
        }
        // right-looking Cholesky of the h x w column panel
        for (int j = 0; j < w; ++j) {
            const double d = F[j + j * ld];
            const double inv = d > 0.0 ? 1.0 / sqrt(d) : 0.0;
            for (int r = j + 1 + tid; r < h; r += NT) F[r + j * ld] *= inv;
            __syncthreads();
            if (tid == 0) F[j + j * ld] = d * inv;
            const int nc = w - j - 1, nr = h - j - 1;
            for (int idx = tid; idx < nc * nr; idx += NT) {
                const int c = j + 1 + idx / nr, r = j + 1 + idx % nr;
                if (r >= c) F[r + c * ld] -= F[r + j * ld] * F[c + j * ld];
            }
            __syncthreads();
        }
        double* Udst;
        int ldud;
        if constexpr (LDS) { Udst = sm + fcap; ldud = u; } else { Udst = F + w + w * ld; ldud = ld; }
        double* Xb = li_in_lds ? sm + lofs : Li;
        // explicit inverse of L_b: lane c solves L X = e_c by forward substitution
        auto invert = [&](int lane, int lanes) {
            for (int c = lane; c < w; c += lanes) {
                double* X = Xb + c * w;
                for (int r = 0; r < c; ++r) X[r] = 0.0;
                const double dc = F[c + c * ld];
                X[c] = dc != 0.0 ? 1.0 / dc : 0.0;
                for (int r = c + 1; r < w; ++r) {
                    double s = 0.0;
                    for (int k = c; k < r; ++k) s += F[r + k * ld] * X[k];
                    const double dr = F[r + r * ld];
                    X[r] = dr != 0.0 ? -s / dr : 0.0;
                }
            }
        };
        // Schur complement of the panel + copy of the factor panel to HBM for the solves
        auto schur = [&](int lane, int lanes) {
            for (int idx = lane; idx < u * u; idx += lanes) {
                const int i = idx % u, j = idx / u;
                if (i < j) continue;
                double s = F[(w + i) + (w + j) * ld];
                for (int k = 0; k < w; ++k) s -= F[(w + i) + k * ld] * F[(w + j) + k * ld];
                Udst[i + j * ldud] = s;
            }
            for (int idx = lane; idx < h * w; idx += lanes) P[idx] = F[idx];
        };
        if constexpr (NT > 64) {  // wave 0 inverts while the other waves update
            if (tid < 64) invert(tid, 64);
            else schur(tid - 64, NT - 64);
        } else {
            invert(tid, NT);
            schur(tid, NT);
        }
        __syncthreads();
        if (li_in_lds) {
            for (int idx = tid; idx < w * w; idx += NT) Li[idx] = Xb[idx];
            __syncthreads();
        }
        u_prev = u; off_prev = offb; Usrc = Udst; ldu = ldud;
    }
}

// Forward and backward block substitution; x (n entries) is overwritten.  sm: xs[hcap], ys[hcap], panel[...] (LDS variant).
template <int NT, bool LDS, class Meta>
__device__ __forceinline__ void solve_chain(const Meta& M, const double* __restrict__ pan, double* __restrict__ x, double* sm, int hcap)
{
    double* xs = sm;
    double* ys = sm + hcap;
    double* Pl = sm + 2 * hcap;
    const int tid = threadIdx.x;
    const int N = M.N, n = M.n, arrow = M.arrow;
    for (int pass = 0; pass < 2; ++pass) {
        for (int bb = 0; bb < N; ++bb) {
            const int b = pass == 0 ? bb : N - 1 - bb;
            const int h = M.H(b), w = M.W(b);
            if (h == 0) continue;
            const int u = h - w, offb = M.Off(b), start = M.Start(b);
            const double* Pg = pan + M.PanOff(b);
            const double* P;
            if constexpr (LDS) {
                const int cnt = h * w + w * w;
                for (int idx = tid; idx < cnt; idx += NT) Pl[idx] = Pg[idx];
                P = Pl;
            } else {
                P = Pg;
            }
            const double* Li = P + (long long)h * w;
            if (pass == 0) {
                for (int r = tid; r < w; r += NT) xs[r] = x[start + r];
                __syncthreads();
                for (int r = tid; r < w; r += NT) {  // y_b = L_b^{-1} x_b
                    double s = 0.0;
                    for (int k = 0; k <= r; ++k) s += Li[r + k * w] * xs[k];
                    ys[r] = s;
                }
                __syncthreads();
                for (int r = tid; r < w; r += NT) x[start + r] = ys[r];
                for (int t = tid; t < u; t += NT) {  // x_{b+1}[0:off] -= C_b y_b ;  x_N -= F_b y_b
                    double s = 0.0;
                    for (int k = 0; k < w; ++k) s += P[(w + t) + k * h] * ys[k];
                    const int tgt = t < offb ? start + w + t : n - arrow + (t - offb);
                    x[tgt] -= s;
                }
                __syncthreads();
            } else {
                for (int r = tid; r < w; r += NT) xs[r] = x[start + r];
                for (int t = tid; t < u; t += NT) xs[w + t] = x[t < offb ? start + w + t : n - arrow + (t - offb)];
                __syncthreads();
                for (int k = tid; k < w; k += NT) {  // z = x_b - C_b^T x_{b+1}[0:off] - F_b^T x_N
                    double s = xs[k];
                    for (int t = 0; t < u; ++t) s -= P[(w + t) + k * h] * xs[w + t];
                    ys[k] = s;
                }
                __syncthreads();
                for (int r = tid; r < w; r += NT) {  // x_b = L_b^{-T} z
                    double s = 0.0;
                    for (int k = r; k < w; ++k) s += Li[k + r * w] * ys[k];
                    x[start + r] = s;
                }
                __syncthreads();
            }
        }
    }
}

}  // namespace msdev
}  // namespace pq
