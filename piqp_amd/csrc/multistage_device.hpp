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

// all fronts of one QP in one flat loop over the lower-triangular entries (ent_b = stage, ent_rc = row | col << 16): full
// lane utilisation and one round of overlapped HBM loads instead of one short, dependent loop per stage
template <int NT, class Meta, class P1, class P2, class P3, class P4, class P5, class FP, class IP>
__device__ __forceinline__ void assemble_flat(const Meta& M, const GroupMeta& Gm, P1 XG, P2 Pf, P3 AtAf, P4 zinv, P5 x_reg, double delta_inv, FP F, IP ent_b, IP ent_rc, int n_ent)
{
    // (P1 .. P5 / FP / IP: pointers to double / double / const int in whatever address space the caller holds them -- the batched kernel passes device-memory
    // pointers so that the loads are global_load, not FLAT)
    typedef __attribute__((address_space(1))) const int gcint;
    typedef __attribute__((address_space(1))) const long long gcll;
    gcint* row_ptr = (gcint*)Gm.row_ptr;
    gcint* rows_i = (gcint*)Gm.rows;
    gcll* x_off = (gcll*)Gm.x_off;
    for (int e = threadIdx.x; e < n_ent; e += NT) {
        const int b = ent_b[e], rc = ent_rc[e];
        const int r = rc & 0xffff, c = rc >> 16;
        const int h = M.H(b);
        const long long at = M.FrontOff(b) + r + (long long)c * h;
        double s = 0.0;
        if (b < M.N - 1) {
            const int k0 = row_ptr[b], rows = row_ptr[b + 1] - k0;
            const P1 Xb = XG + x_off[b];
            for (int k = 0; k < rows; ++k) s += Xb[r + (long long)k * h] * zinv[rows_i[k0 + k]] * Xb[c + (long long)k * h];
        }
        double v = Pf[at] + delta_inv * AtAf[at] + s;
        if (r == c && c < M.W(b)) v += x_reg[M.Start(b) + c];
        F[at] = v;
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

// ---- single-wave variants for tiny stages (every front has at most 64 entries: one entry per lane) --------------------
// Used by the batched kernel with 64-thread workgroups.  The front of the current stage lives in REGISTERS (lane = row +
// col * h holds one entry); pivots and panel columns are broadcast with v_readlane, the carried update matrix moves
// between stages with one cross-lane permute, and nothing waits on a barrier or on HBM (the next stage's assembled
// front is prefetched into a register while this stage is factored).  What the solves need is not L itself but
//   Linv_b = L_bb^{-1} (w x w)   and   Q_b = [C_b; F_b] * Linv_b (u x w),
// so that a forward step is  y_b = Linv_b x_b,  x_next -= Q_b x_b  (both straight from x_b: one LDS round trip per stage)
// and a backward step is  x_b = Linv_b^T x_b - Q_b^T x_next.  They are written to `pan` (LDS) per stage: Linv | Q.
constexpr int WAVE_WMAX = 8;

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0); vmcnt / expcnt untouched
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 1/sqrt(d) to ~1 ulp without the IEEE sqrt + divide chains (two Newton steps on v_rsq_f64)
__device__ __forceinline__ double rsqrt_newton(double d)
{
    double y = __builtin_amdgcn_rsq(d);
    y = y * (1.5 - 0.5 * d * y * y);
    y = y * (1.5 - 0.5 * d * y * y);
    return y;
}

// value of `v` in lane `src` (uniform) broadcast to every lane
__device__ __forceinline__ double lane_bcast(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// wave-uniform 64-bit value -> scalar registers
__device__ __forceinline__ long long uni(long long v)
{
    const int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffffLL));
    const int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
    return ((long long)hi << 32) | (unsigned int)lo;
}

typedef const double __attribute__((address_space(1))) global_cdouble;  // HBM pointer (global_load: does not touch lgkmcnt)

// cross-lane read of a double by byte address (lane << 2): the index arithmetic of __shfl leaves the stage loops
__device__ __forceinline__ double bperm_d(int addr, double v)
{
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// One stage of factor_chain_wave with the panel width W known at compile time: every loop over the panel columns unrolls, the reciprocal pivots and the
// column of Linv live in registers (with a run-time width they were indexed arrays in scratch), and cross-lane reads that do not depend on each other are
// requested together -- one LDS round trip for the Schur complement instead of one per panel column (the wave's own LDS latencies, not instruction
// issue, are most of a stage: rocprofv3 SQ_ACTIVE_INST_VALU 42 % of the SIMD cycles at 4.5 waves per SIMD, profiles/r03_pmc_batch_c4.txt).
// The arithmetic is the run-time-width loop's, operation by operation (same expressions, same order): the factor is bitwise the same.
//   f: this lane's entry (r, c) of the assembled front (+ carried update); on return lanes (r >= W, c >= W, r >= c) hold the update matrix to carry
template <int W>
__device__ __forceinline__ void factor_stage_wave(double& f, const int h, const int u, const int lane, const int r, const int c, double* __restrict__ P)
{
    int ar[W], ac[W];
#pragma unroll
    for (int j = 0; j < W; ++j) { ar[j] = (r + j * h) << 2; ac[j] = (c + j * h) << 2; }
    // ---- right-looking Cholesky of the h x W column panel, pivots by v_readlane, columns by two cross-lane reads ----
    double invs[W];
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const double d = lane_bcast(f, j + j * h);
        const double inv = d > 0.0 ? rsqrt_newton(d) : 0.0;
        invs[j] = inv;
        if (c == j) f = r == j ? d * inv : (r > j ? f * inv : f);
        const double a = bperm_d(ar[j], f), bb = bperm_d(ac[j], f);
        if (c > j && c < W && r >= c) f -= a * bb;
    }
    // ---- Schur complement of the panel: lanes (r >= W, c >= W, r >= c); the panel columns they read are final, so all reads go out first ----
    {
        double a[W], bb[W];
#pragma unroll
        for (int k = 0; k < W; ++k) { a[k] = bperm_d(ar[k], f); bb[k] = bperm_d(ac[k], f); }
        const bool mine = r >= W && c >= W && r >= c;
#pragma unroll
        for (int k = 0; k < W; ++k)
            if (mine) f -= a[k] * bb[k];
    }
    // ---- Linv: lane j < W builds column j of L^{-1} in registers (1/L_kk = invs[k]: no divisions) ----
    double X[W];
#pragma unroll
    for (int rr = 0; rr < W; ++rr) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < rr; ++k) s += lane_bcast(f, rr + k * h) * X[k];  // X[k] is 0 for k < lane
        X[rr] = lane == rr ? invs[rr] : (lane < rr ? -invs[rr] * s : 0.0);
    }
    // ---- Q = [C; F] * Linv and the stores for the solves ----
#pragma unroll
    for (int k = 0; k < W; ++k)
        if (lane < W) P[k + lane * W] = X[k];
    for (int t = 0; t < u; ++t) {
        double q = 0.0;
#pragma unroll
        for (int k = 0; k < W; ++k) q += lane_bcast(f, (W + t) + k * h) * X[k];
        if (lane < W) P[W * W + t + lane * u] = q;
    }
}

// value of the lane SH further up in the same row of 16 lanes (0.0 past the row's end)
template <int SH>
__device__ __forceinline__ double dpp_row_shl(double v)
{
    static_assert(SH >= 1 && SH <= 15, "row shift");
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x100 + SH, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x100 + SH, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// The leading run of a uniform chain -- K stages of ONE shape (W columns eliminated, u <= W rows below them, gap-free, no arrow: the run the register-carried
// substitution uses, checked at setup) -- with ONE LANE PER ROW of the front: lane r holds row r, its h = W + u <= 2 W entries in registers.  factor_stage_wave keeps
// one ENTRY per lane, and everything a pivot needs from another entry is a trip through the LDS crossbar (two per pivot and two per column of the Schur complement,
// each ~130 cycles of latency: ~1 900 cycles a stage, 0.39 of the 1.25 ms a lone instance takes).  With a row per lane the pivot, and the entry L(c, j) that row r's
// update of column c needs, are wave-uniform lane reads, and the update itself stays in the lane's registers.  Every entry receives the same updates in the same
// order with the same operands as in factor_stage_wave (panel columns right-looking, the Schur complement by ascending pivot; products commute): the panels are
// bitwise the same.  Returns, in the one-entry-per-lane layout of stage K - 1, what factor_chain_wave carries into stage K.
template <int W, class Meta>
__device__ __forceinline__ double factor_chain_rows(const Meta& M, global_cdouble* fronts_g, double* __restrict__ pan, const int K, double* __restrict__ stage)
{
    // stage: (2 W)^2 doubles of LDS nobody else uses during the factorisation.  A front arrives from memory as before -- ONE load per lane, entry l of the front in lane
    // l, requested two stages ahead -- and is turned into rows through `stage` (one write, then this lane's row read back) behind the arithmetic of the stage before:
    // a lane loading its row itself is h loads per stage instead of one, and with every compute unit full (8192 instances) the extra memory instructions cost
    // more than the lane reads save.
    constexpr int HM = 2 * W;
    const int lane = threadIdx.x;
    const int u = __builtin_amdgcn_readfirstlane(M.Off(0));
    const int h = W + u, PS = W * W + u * W, hh = h * h;
    const long long f0 = uni(M.FrontOff(0)), p0 = uni(M.PanOff(0));
    const int r = lane;  // this lane's row (lanes >= h idle along)
    const bool ent = lane < hh;
    double* const wr = stage + (ent ? (lane % h) * HM + lane / h : 0);  // where this lane's ENTRY goes
    const double* const rd = stage + (r < h ? r : 0) * HM;            // this lane's ROW
    // (requests without uniform branches, from clamped addresses: see solve_chain_wave_reg)
    auto load_entry = [&](int b) -> double { return ent ? fronts_g[f0 + (long long)min(b, K - 1) * hh + lane] : 0.0; };
    double rows[HM], fr[HM];
    auto to_rows = [&](double e) {
        if (ent) *wr = e;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int c = 0; c < HM; ++c) { const double rv = rd[c < h ? c : 0]; rows[c] = c < h ? rv : 0.0; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    to_rows(load_entry(0));
    double gA = load_entry(1), gB = load_entry(2);
#pragma unroll
    for (int c = 0; c < HM; ++c) fr[c] = 0.0;
    auto stage_b = [&](const int b, double& g) __attribute__((always_inline)) {
        // ---- this stage's front + what the stage before left for its first u rows and columns ----
        double carried[W];
#pragma unroll
        for (int c = 0; c < W; ++c) carried[c] = (b > 0 && c < u) ? dpp_row_shl<W>(fr[W + c]) : 0.0;
#pragma unroll
        for (int c = 0; c < HM; ++c) {
            double add = 0.0;
            if (c < W) add = (b > 0 && r < u && c < u && r >= c) ? carried[c] : 0.0;
            fr[c] = rows[c] + add;
        }
        // ---- panel columns and Schur complement, right-looking ----
        double invs[W];
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const double d = lane_bcast(fr[j], j);
            const double inv = d > 0.0 ? rsqrt_newton(d) : 0.0;
            invs[j] = inv;
            fr[j] = r == j ? d * inv : (r > j ? fr[j] * inv : fr[j]);
            const double lr = fr[j];
            // (all 2 W columns, no branch on the run's h: a column beyond the front belongs to idle lanes that hold zeros -- the lane reads of a pivot then go out
            // together instead of one per branch)
            double lc[HM];
#pragma unroll
            for (int c = j + 1; c < HM; ++c) lc[c] = lane_bcast(fr[j], c);
#pragma unroll
            for (int c = j + 1; c < HM; ++c)
                if (r >= c) fr[c] -= lr * lc[c];
        }
        // ---- the next front: entries (requested two stages ago) -> rows; its register then takes the request for the front three stages on ----
        to_rows(g);  // (at the last stage of the run: a front nobody uses)
        g = load_entry(b + 3);
        // ---- Linv: lane j < W builds column j of L^{-1} (1 / L_kk = invs[k]) ----
        double X[W];
#pragma unroll
        for (int rr = 0; rr < W; ++rr) {
            double sacc = 0.0;
#pragma unroll
            for (int k = 0; k < rr; ++k) sacc += lane_bcast(fr[k], rr) * X[k];  // X[k] is 0 for k < lane
            X[rr] = lane == rr ? invs[rr] : (lane < rr ? -invs[rr] * sacc : 0.0);
        }
        double* P = pan + p0 + (long long)b * PS;
#pragma unroll
        for (int k = 0; k < W; ++k)
            if (lane < W) P[k + lane * W] = X[k];
        // ---- Q = [C; F] Linv ----
#pragma unroll
        for (int t = 0; t < W; ++t) {
            double lq[W];
#pragma unroll
            for (int k = 0; k < W; ++k) lq[k] = lane_bcast(fr[k], W + t);  // (t >= u: an idle lane's zeros, nothing stored)
            double q = 0.0;
#pragma unroll
            for (int k = 0; k < W; ++k) q += lq[k] * X[k];
            if (lane < W && t < u) P[W * W + t + lane * u] = q;
        }
    };
    for (int b = 0; b < K; b += 2) {
        stage_b(b, gA);
        if (b + 1 < K) stage_b(b + 1, gB);
    }
    // hand-over: entry (rr, cc) of stage K - 1's front to lane rr + cc h
    double out = 0.0;
    const int rr = lane % h, cc = lane / h;
#pragma unroll
    for (int c = 0; c < HM; ++c) {
        const double t = bperm_d(rr << 2, fr[c]);
        if (cc == c) out = t;
    }
    return ent ? out : 0.0;
}

// fronts_g: assembled fronts in HBM; pan: LDS, per stage Linv (w x w) then Q (u x w) at PanOff(b)
// ROWS: the kernel variant may factor the uniform run with one lane per row (factor_chain_rows; rows_stage != nullptr says that this chain has such a run and where
// the LDS staging is).  A compile-time switch because the mere presence of that code in the variants built for full compute units (four waves per SIMD, 128
// registers) cost them 3-8 % (measured at 8192 instances: 5.74 ms without it, 5.9-6.0 with it, 6.2 on the one-entry-per-lane path next to it); the variants for
// two and three waves per SIMD (up to 12 instances per compute unit) have it: 2304 / 3072 instances 2.08 / 2.33 -> 1.88 / 2.10 ms.
template <bool ROWS = false, class Meta>
__device__ __forceinline__ void factor_chain_wave(const Meta& M, global_cdouble* fronts_g, double* __restrict__ pan, const int reg_k = 0, double* __restrict__ rows_stage = nullptr)
{
    // reg_k > 0: the stages 0 .. reg_k - 1 have ONE shape and follow each other without gaps (the run the register-carried substitution uses, checked at
    // setup): their widths and offsets are affine in the stage number and are not looked up in the table (five LDS reads + waits per stage)
    const int lane = threadIdx.x;
    const int N = M.N;
    const int rw = reg_k > 0 ? __builtin_amdgcn_readfirstlane(M.W(0)) : 0, ro = reg_k > 0 ? __builtin_amdgcn_readfirstlane(M.Off(0)) : 0;
    const int rh = rw + ro;
    const long long rf0 = reg_k > 0 ? uni(M.FrontOff(0)) : 0, rp0 = reg_k > 0 ? uni(M.PanOff(0)) : 0;
    int h_prev = 0, w_prev = 0, off_prev = 0;
    double f = 0.0;                  // after a stage: lanes (r >= w, c >= w, r >= c) hold the carried update matrix
    int b_first = 0;
    if constexpr (ROWS)
    if (rows_stage != nullptr && reg_k > 0 && reg_k < N && rw <= 6 && ro >= 1 && ro <= rw) {
        // the uniform run with one lane per row (factor_chain_rows); the stages behind it -- usually one terminal stage -- below
        switch (rw) {
        case 1: f = factor_chain_rows<1>(M, fronts_g, pan, reg_k, rows_stage); break;
        case 2: f = factor_chain_rows<2>(M, fronts_g, pan, reg_k, rows_stage); break;
        case 3: f = factor_chain_rows<3>(M, fronts_g, pan, reg_k, rows_stage); break;
        case 4: f = factor_chain_rows<4>(M, fronts_g, pan, reg_k, rows_stage); break;
        case 5: f = factor_chain_rows<5>(M, fronts_g, pan, reg_k, rows_stage); break;
        default: f = factor_chain_rows<6>(M, fronts_g, pan, reg_k, rows_stage); break;
        }
        b_first = reg_k; h_prev = rh; w_prev = rw; off_prev = ro;
    }
    int h = __builtin_amdgcn_readfirstlane(M.H(b_first));
    double nxt = (h > 0 && lane < h * h) ? fronts_g[uni(M.FrontOff(b_first)) + lane] : 0.0;
    int r = h > 0 ? lane % h : 0, c = h > 0 ? lane / h : 0;  // lane -> (row, col) of the current front; recomputed only when h changes
    // the carried update is a fixed cross-lane permutation per stage SHAPE: source lane and mask are recomputed only when the shape changes
    int c_sig[7] = {-1, -1, -1, -1, -1, -1, -1};
    int c_addr = 0;
    bool c_has = false;
    for (int b = b_first; b < N; ++b) {
        if (h == 0) break;  // no arrow corner
        const bool affine = b < reg_k;
        const int w = affine ? rw : __builtin_amdgcn_readfirstlane(M.W(b));
        const bool corner = b == N - 1;
        const int offb = affine ? ro : __builtin_amdgcn_readfirstlane(M.Off(b));
        const int u = h - w;
        // ---- carried update: new entry (r, c) <- old trailing entry (i, j) ----
        double carried = 0.0;
        if (h_prev > 0) {
            const int sig[7] = {h, w, offb, h_prev, w_prev, off_prev, (int)corner};
            bool same = true;
#pragma unroll
            for (int q = 0; q < 7; ++q) same = same && sig[q] == c_sig[q];
            if (!same) {
#pragma unroll
                for (int q = 0; q < 7; ++q) c_sig[q] = sig[q];
                auto inv_carry = [&](int t) {  // inverse of carry_row: -1 = nothing lands on row t
                    if (t < off_prev) return t;
                    const int base = corner ? 0 : w + offb;
                    if (t >= base && !corner) return off_prev + (t - base);
                    if (corner) return off_prev + t;
                    return -1;
                };
                const int i = inv_carry(r), j = inv_carry(c);
                const int u_prev = h_prev - w_prev;
                c_has = lane < h * h && i >= 0 && j >= 0 && i < u_prev && j < u_prev && i >= j;
                c_addr = (c_has ? (w_prev + i) + (w_prev + j) * h_prev : lane) << 2;
            }
            const double got = bperm_d(c_addr, f);
            carried = c_has ? got : 0.0;
        }
        f = nxt + carried;
        const bool affine_n = b + 1 < reg_k;
        const int hn = affine_n ? rh : (b + 1 < N ? __builtin_amdgcn_readfirstlane(M.H(b + 1)) : 0);
        if (hn > 0) nxt = lane < hn * hn ? fronts_g[(affine_n ? rf0 + (long long)(b + 1) * (rh * rh) : uni(M.FrontOff(b + 1))) + lane] : 0.0;  // prefetch, consumed next iteration
        double* P = pan + (affine ? rp0 + (long long)b * (rw * rw + ro * rw) : uni(M.PanOff(b)));
        switch (w) {
        case 1: factor_stage_wave<1>(f, h, u, lane, r, c, P); break;
        case 2: factor_stage_wave<2>(f, h, u, lane, r, c, P); break;
        case 3: factor_stage_wave<3>(f, h, u, lane, r, c, P); break;
        case 4: factor_stage_wave<4>(f, h, u, lane, r, c, P); break;
        case 5: factor_stage_wave<5>(f, h, u, lane, r, c, P); break;
        case 6: factor_stage_wave<6>(f, h, u, lane, r, c, P); break;
        case 7: factor_stage_wave<7>(f, h, u, lane, r, c, P); break;
        case 8: factor_stage_wave<8>(f, h, u, lane, r, c, P); break;
        default: break;  // w == 0: nothing to eliminate (widths above WAVE_WMAX never reach this mode)
        }
        h_prev = h; w_prev = w; off_prev = offb;
        if (hn != h && hn > 0) { r = lane % hn; c = lane / hn; }
        h = hn;
    }
    wave_lds_sync();
}

// one stage of the forward / backward substitution, panel width W at compile time (all LDS reads of a stage in one round trip)
template <int W>
__device__ __forceinline__ void solve_stage_fwd_wave(const double* __restrict__ Li, double* __restrict__ x, const int h, const int u, const int offb, const int start, const int tail, const int lane)
{
    const double* Q = Li + W * W;
    // lanes < W: y_b = Linv x_b; lanes W .. h - 1: x_next -= Q x_b -- ONE loop for both (the stored Linv has exact zeros above its diagonal, so the
    // row sum may run over all W columns)
    double acc = 0.0;
    int tgt = -1;
    const bool top = lane < W;
    const int t = lane - W;
    const double* cf = top ? Li + lane : Q + (lane < h ? t : 0);
    const int cs = top ? W : u;
    if (top) tgt = start + lane;
    else if (lane < h) { tgt = t < offb ? start + W + t : tail + (t - offb); acc = x[tgt]; }
    if (lane < h) {
        double cv[W], xv[W];
#pragma unroll
        for (int k = 0; k < W; ++k) { cv[k] = cf[k * cs]; xv[k] = x[start + k]; }
#pragma unroll
        for (int k = 0; k < W; ++k) acc += (top ? cv[k] : -cv[k]) * xv[k];
    }
    wave_lds_sync();  // every read of x_b above has returned before any lane overwrites it
    if (tgt >= 0) x[tgt] = acc;
    wave_lds_sync();
}
template <int W>
__device__ __forceinline__ void solve_stage_bwd_wave(const double* __restrict__ Li, double* __restrict__ x, const int u, const int offb, const int start, const int tail, const int lane)
{
    const double* Q = Li + W * W;
    double acc = 0.0;
    if (lane < W) {  // x_b = Linv^T x_b - Q^T x_next
        double lv[W], xv[W];
#pragma unroll
        for (int k = 0; k < W; ++k) { lv[k] = Li[k + lane * W]; xv[k] = x[start + k]; }
#pragma unroll
        for (int k = 0; k < W; ++k)
            if (k >= lane) acc += lv[k] * xv[k];
        for (int t = 0; t < u; ++t) acc -= Q[t + lane * u] * x[t < offb ? start + W + t : tail + (t - offb)];
    }
    wave_lds_sync();
    if (lane < W) x[start + lane] = acc;
    wave_lds_sync();
}

// The same two sweeps for a chain WITHOUT arrow whose stages all eliminate W columns and follow each other without gaps (start_{b+1} = start_b + W: the
// usual optimal-control chain), with nothing of one stage's result going through LDS to the next: stage b's lanes W .. h_b - 1 hold the entries of x_{b+1}
// they have just updated, a row shift (DPP) moves them to lanes 0 .. u_b - 1 where stage b + 1 expects them, and the entries stage b does not touch come from
// the right-hand side, requested a stage ahead together with the coefficients.  The stage table is read once (lane b holds stage b; a v_readlane per use).
// What a stage waits for is its own arithmetic -- W lane reads and W multiply-adds -- instead of two to three LDS round trips and two wave syncs
// (the chain substitution was 0.66 ms of an instance's 2.8 ms).  Same products in the same order: bitwise the same solution.
// FWD = true: the forward sweep over stages 0 .. K - 1 (the state stage K expects is left in x); false: the backward sweep over stages K - 1 .. 0 (stage K's
// solution is read from x).  K < nst: the stages K .. nst - 1 -- a terminal stage of another width, say -- go through solve_stage_*_wave in between.
template <int W, bool FWD, class Meta>
__device__ __forceinline__ void solve_chain_wave_reg(const Meta& M, const double* __restrict__ pan, double* __restrict__ x, const int K, const int nst)
{
    // the K stages have ONE shape (W columns, u rows below them, checked at setup): stage b starts at s0 + b W and keeps its panel at p0 + b (W W + u W) --
    // no table look-ups in the loops, the per-lane operand pointers just advance
    const int lane = threadIdx.x;
    const int u = __builtin_amdgcn_readfirstlane(M.Off(0)), s0 = __builtin_amdgcn_readfirstlane(M.Start(0)), p0 = __builtin_amdgcn_readfirstlane((int)M.PanOff(0));
    const int h = W + u, PS = W * W + u * W;
    const bool top = lane < W;
    if constexpr (FWD) {
        // ---- forward: y_b = Linv_b x_b,  x_{b+1}[0 : u] -= Q_b x_b ----
        const double* cfp = pan + p0 + (top ? lane : W * W + (lane < h ? lane - W : 0));
        const int cs = top ? W : u;
        double* xp = x + s0 + lane;
        // (two copies of the stage per loop trip, the operands of one requested by the other: handing the registers of the next stage's operands on at the end of
        // a trip -- cf = cfn -- made every stage wait for the loads it had just issued, an LDS latency per stage that nothing hid)
        double cfA[W], cfB[W];
#pragma unroll
        for (int k = 0; k < W; ++k) cfA[k] = cfp[k * cs];
        double cur = lane < h ? xp[0] : 0.0;
        // (the entries of x a stage is the first to touch -- still the right-hand side -- are requested two stages ahead for the same reason)
        double freshA = (K > 1 && lane >= u && lane < h) ? xp[W] : 0.0, freshB = 0.0;
        auto stage_fwd = [&](const int b, double (&cf)[W], double (&cfn)[W], double& fresh_in, double& fresh_out) __attribute__((always_inline)) {
            // (unconditional requests from clamped addresses: a load under a uniform branch makes the compiler's wait counts at the join conservative -- it
            // waited for ALL LDS reads, the ones just issued included, in front of every stage's lane reads)
            cfp += (b + 1 < K) ? PS : 0;
#pragma unroll
            for (int k = 0; k < W; ++k) cfn[k] = cfp[k * cs];
            fresh_out = (lane >= u && lane < h) ? xp[(b + 2 < K) ? 2 * W : 0] : 0.0;
            double xk[W];
#pragma unroll
            for (int k = 0; k < W; ++k) xk[k] = lane_bcast(cur, k);
            double acc = top ? 0.0 : cur;
#pragma unroll
            for (int k = 0; k < W; ++k) acc += (top ? cf[k] : -cf[k]) * xk[k];
            if (top) xp[0] = acc;  // y_b (read again by the backward sweep only)
            const double down = dpp_row_shl<W>(acc);
            cur = lane < u ? down : fresh_in;
            xp += W;
        };
        for (int b = 0; b < K; b += 2) {
            stage_fwd(b, cfA, cfB, freshA, freshB);
            if (b + 1 < K) stage_fwd(b + 1, cfB, cfA, freshB, freshA);
        }
        if (K < nst && lane < u) xp[0] = cur;  // hand-over: what stage K finds in x
    } else {
        // ---- backward: x_b = Linv_b^T y_b - Q_b^T x_{b+1}[0 : u] ----
        const double* lp = pan + p0 + (K - 1) * PS + (top ? lane : 0) * W;  // Linv_b[k + lane W]
        const double* qp = pan + p0 + (K - 1) * PS + W * W + (top ? lane : 0) * u;  // Q_b[t + lane u]
        double* xb = x + s0 + (K - 1) * W;
        double lvA[W], yvA[W], lvB[W], yvB[W];
#pragma unroll
        for (int k = 0; k < W; ++k) { lvA[k] = lp[k]; yvA[k] = xb[k]; }
        double sol = 0.0;  // lanes < u: the solution of the stage above, as far as this stage needs it
        if (K < nst && lane < u) sol = xb[W + lane];
        double qvA[W], qvB[W];  // (u <= W in such a run)
#pragma unroll
        for (int t = 0; t < W; ++t) { const double qv_t = qp[t < u ? t : 0]; qvA[t] = t < u ? qv_t : 0.0; }
        auto stage_bwd = [&](const int b, double (&lv)[W], double (&yv)[W], double (&qv)[W], double (&lvn)[W], double (&yvn)[W], double (&qvn)[W]) __attribute__((always_inline)) {
            {   // (unconditional, clamped: see the forward sweep)
                const int bp = b > 0 ? PS : 0, bw = b > 0 ? W : 0;
#pragma unroll
                for (int k = 0; k < W; ++k) { lvn[k] = lp[k - bp]; yvn[k] = xb[k - bw]; }
#pragma unroll
                for (int t = 0; t < W; ++t) { const double qv_t = qp[(t < u ? t : 0) - bp]; qvn[t] = t < u ? qv_t : 0.0; }
            }
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < W; ++k)
                if (k >= lane) acc += lv[k] * yv[k];
#pragma unroll
            for (int t = 0; t < W; ++t)
                if (t < u) acc -= qv[t] * lane_bcast(sol, t);
            if (top) xb[lane] = acc;
            sol = acc;
            lp -= PS; qp -= PS; xb -= W;
        };
        for (int b = K - 1; b >= 0; b -= 2) {
            stage_bwd(b, lvA, yvA, qvA, lvB, yvB, qvB);
            if (b - 1 >= 0) stage_bwd(b - 1, lvB, yvB, qvB, lvA, yvA, qvA);
        }
    }
}

// pan, x: LDS.  Forward and backward block substitution by one wave, one LDS round trip per stage.
template <class Meta>
__device__ __forceinline__ void solve_chain_wave(const Meta& M, const double* __restrict__ pan, double* __restrict__ x, const int reg_w = 0, const int reg_k = 0, const int reg_nst = 0)
{
    // reg_w > 0: the stages 0 .. reg_k - 1 all eliminate reg_w columns of a gap-free chain without arrow (checked at setup): their part of both sweeps is
    // carried in registers (solve_chain_wave_reg); the stages reg_k .. -- none, or a terminal stage of another width -- take the step functions below
    const int lane = threadIdx.x;
    const int N = M.N, tail = M.n - M.arrow;
    const int b0 = reg_w > 0 ? reg_k : 0;
    switch (reg_w) {
    case 1: solve_chain_wave_reg<1, true>(M, pan, x, reg_k, reg_nst); break;
    case 2: solve_chain_wave_reg<2, true>(M, pan, x, reg_k, reg_nst); break;
    case 3: solve_chain_wave_reg<3, true>(M, pan, x, reg_k, reg_nst); break;
    case 4: solve_chain_wave_reg<4, true>(M, pan, x, reg_k, reg_nst); break;
    case 5: solve_chain_wave_reg<5, true>(M, pan, x, reg_k, reg_nst); break;
    case 6: solve_chain_wave_reg<6, true>(M, pan, x, reg_k, reg_nst); break;
    default: break;
    }
    if (reg_w > 0) wave_lds_sync();
    for (int b = b0; b < N; ++b) {
        const int h = __builtin_amdgcn_readfirstlane(M.H(b)), w = __builtin_amdgcn_readfirstlane(M.W(b));
        if (h == 0) continue;
        const int u = h - w, offb = __builtin_amdgcn_readfirstlane(M.Off(b)), start = __builtin_amdgcn_readfirstlane(M.Start(b));
        const double* Li = pan + uni(M.PanOff(b));
        switch (w) {
        case 1: solve_stage_fwd_wave<1>(Li, x, h, u, offb, start, tail, lane); break;
        case 2: solve_stage_fwd_wave<2>(Li, x, h, u, offb, start, tail, lane); break;
        case 3: solve_stage_fwd_wave<3>(Li, x, h, u, offb, start, tail, lane); break;
        case 4: solve_stage_fwd_wave<4>(Li, x, h, u, offb, start, tail, lane); break;
        case 5: solve_stage_fwd_wave<5>(Li, x, h, u, offb, start, tail, lane); break;
        case 6: solve_stage_fwd_wave<6>(Li, x, h, u, offb, start, tail, lane); break;
        case 7: solve_stage_fwd_wave<7>(Li, x, h, u, offb, start, tail, lane); break;
        case 8: solve_stage_fwd_wave<8>(Li, x, h, u, offb, start, tail, lane); break;
        default: break;
        }
    }
    for (int b = N - 1; b >= b0; --b) {
        const int h = __builtin_amdgcn_readfirstlane(M.H(b)), w = __builtin_amdgcn_readfirstlane(M.W(b));
        if (h == 0) continue;
        const int u = h - w, offb = __builtin_amdgcn_readfirstlane(M.Off(b)), start = __builtin_amdgcn_readfirstlane(M.Start(b));
        const double* Li = pan + uni(M.PanOff(b));
        switch (w) {
        case 1: solve_stage_bwd_wave<1>(Li, x, u, offb, start, tail, lane); break;
        case 2: solve_stage_bwd_wave<2>(Li, x, u, offb, start, tail, lane); break;
        case 3: solve_stage_bwd_wave<3>(Li, x, u, offb, start, tail, lane); break;
        case 4: solve_stage_bwd_wave<4>(Li, x, u, offb, start, tail, lane); break;
        case 5: solve_stage_bwd_wave<5>(Li, x, u, offb, start, tail, lane); break;
        case 6: solve_stage_bwd_wave<6>(Li, x, u, offb, start, tail, lane); break;
        case 7: solve_stage_bwd_wave<7>(Li, x, u, offb, start, tail, lane); break;
        case 8: solve_stage_bwd_wave<8>(Li, x, u, offb, start, tail, lane); break;
        default: break;
        }
    }
    if (reg_w > 0) wave_lds_sync();
    switch (reg_w) {
    case 1: solve_chain_wave_reg<1, false>(M, pan, x, reg_k, reg_nst); break;
    case 2: solve_chain_wave_reg<2, false>(M, pan, x, reg_k, reg_nst); break;
    case 3: solve_chain_wave_reg<3, false>(M, pan, x, reg_k, reg_nst); break;
    case 4: solve_chain_wave_reg<4, false>(M, pan, x, reg_k, reg_nst); break;
    case 5: solve_chain_wave_reg<5, false>(M, pan, x, reg_k, reg_nst); break;
    case 6: solve_chain_wave_reg<6, false>(M, pan, x, reg_k, reg_nst); break;
    default: break;
    }
    if (reg_w > 0) wave_lds_sync();
}

}  // namespace msdev
}  // namespace pq
