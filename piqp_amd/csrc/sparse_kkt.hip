// piqp_amd/csrc/sparse_kkt.hip -- device-resident replacement of piqp::sparse::KKT<T,I,KKT_FULL>
// (reference include/piqp/sparse/kkt.hpp + kkt_full.hpp + ldlt.hpp).
//
//   reference                                        here
//   PKPt values + diagonal refresh (kkt_full:172-210) k_set_diag on the device copy of the PKPt values
//   LDLt::factorize_numeric (ldlt.hpp:101-169,         supernodal multifrontal LDLt: fronts assembled from the PKPt values
//     up-looking, serial over rows)                    per tree level k_front_factor (extend-add + partial dense LDLt,
//                                                      one workgroup per front, LDS-resident when it fits); fronts
//                                                      that big_front() selects go through the dense MFMA panel kernels, a tree level at a time
//   lsolve/dsolve/ltsolve + perm/permt (ldlt:171-218)  k_perm_gather, per level k_subtree_fwd_wave / k_front_fwd_wide, k_scale, k_front_bwd_wide / k_subtree_bwd_wave, k_perm_scatter
//   eval_P_x / eval_A.. / eval_G.. (kkt.hpp:179-203)   k_spmv_cols on CSC copies (P symmetrised, A and G kept in both
//                                                      orientations so every product is a conflict-free column dot)
//   update_data_impl (kkt_full:212-251)                k_remap_values through the composed index maps
// No atomics anywhere: children are merged into their parent in a fixed order, so results are bitwise
// reproducible (the reference's clone test needs that).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <type_traits>

#include "trace.hpp"
#include "dense_kernels.hpp"
#include "kkt_solver_base.hpp"
#include "sparse_ops.hpp"
#include "rccl_transport.hpp"
#include "sparse_symbolic.hpp"

namespace pq {

namespace {

// Fronts that go through the dense multi-workgroup MFMA kernels (all such fronts of a tree level together: ~80 us per level and panel whatever
// their number) instead of one workgroup's pivot loop (1.5 - 2.5 us per pivot).  Fronts under 96 rows never do: C3 / C5-type trees (fronts <= 92
// rows) keep their persistent top launch.
// (round 4: also the wide fronts of fewer pivots whose panel does not fit one workgroup's LDS -- left to one workgroup they are factored in HBM, and their
// Schur complement, f^2 w flops, by that one workgroup: 9.8 ms per level on the C3 variant with 1500-variable windows, fronts of 1000-4000 rows)
// and every front of 768 rows or more: whatever its pivots, one workgroup zero-fills and extend-adds its f x f square -- 8.4 ms per level there)
// (an accumulator supernode -- w = 0, the extend-add alone -- included)
using sparse::big_front;  // (sparse_symbolic.hpp: the ordering cost model and the spine merging price the same classification)
constexpr int SUB_SOLVE_THREADS = 64;   // substitution inside a subtree is a chain of short vector operations: one wave per subtree
constexpr int SUBTREE_LDS_BYTES = 150 * 1024;  // two fronts of the LDS-native subtree walker
constexpr int SUB_THREADS = 256;    // small subtrees: fronts reach ~100 rows near the subtree root, so a full workgroup (64 threads measured 2x slower)
constexpr int IND_SCRATCH = 128;     // doubles of LDS behind a front for the reciprocals of its independent leading pivots
constexpr int LDS_FRONT_DOUBLES = 12288;  // 96 KiB: fronts up to 110 x 110 are factored inside LDS
// doubles of LDS a one-workgroup front works in: the whole front when it fits, else its f x w pivot panel when that fits (the pivot loop then runs
// at LDS latency, 0.2 - 0.3 us per pivot, and only the one-pass Schur complement touches the trailing block in HBM), else nothing (all in HBM:
// a global-memory round trip per pivot, 5 - 10 us)
__host__ __device__ inline long long front_lds_doubles(long long f, long long w)
{
    if (f * f <= LDS_FRONT_DOUBLES) return f * f;
    if (w > 0 && f * w <= LDS_FRONT_DOUBLES) return f * w;
    return 0;
}
// A level-scheduled front that stages its panel: zero-fill, own entries and the children's update matrices are handled by the multi-workgroup kernels
// of the big fronts (a single workgroup pays a memory round trip per handful of entries there), one workgroup factors the panel in LDS
// (k_front_panel: the same rank-1 pivot loop as everywhere, no inverted blocks), and the Schur complement T -= L D L^T runs on the matrix cores
// with the trailing updates of the level's big fronts (k_syrk_lower_fronts)
__host__ __device__ inline bool panel_front(long long f, long long w) { return !big_front((int)f, (int)w) && f * f > LDS_FRONT_DOUBLES && w > 0 && f * w <= LDS_FRONT_DOUBLES; }

struct SnRec {  // everything the numeric kernels need about one supernode, in one 32-byte record (one load instead of a
                 // chain of dependent loads through six index arrays)
    int first;      // first (permuted) column
    int w;          // pivots
    int f;          // front order
    int rows_ptr;   // offset into front_rows / fvec
    int child_lo, child_hi;  // range in `child`
    int rel_ptr;    // offset into `rel` of this supernode's update rows inside its parent
    int parent;     // parent supernode (-1: root)
    long long front_off;
    int nind;       // leading pivots that are mutually independent (merged sibling leaves): eliminated in one pass
    int fe_lo, fe_hi;  // assembly entries of this supernode: values vals[fe_lo .. fe_hi) (the value array is stored in this order), front offsets fe_off[..]
};
struct ChildRec {  // what a parent needs to know about one child, stored parallel to the child lists (one load instead of child -> record)
    int c;        // the child supernode
    int vc_off;   // offset of its update vector in fvec (rows_ptr + w)
    int uc;       // update rows (f - w)
    int rel_ptr;  // offset into `rel`
};
struct FrontMeta {  // device-side views of the symbolic analysis
    const SnRec* sn;
    const int* front_rows;
    const int* child;
    const int* rel;
    // assembly: the PKPt entries owned by supernode s are fe_q[fe_ptr[s] .. fe_ptr[s+1]), each at offset fe_off inside the front
    const int* fe_ptr;
    const int* fe_q;
    const int* fe_off;
    const double* vals;
    const ChildRec* crec;
};

__global__ void k_set_diag(int n, int p, int m, const int* __restrict__ diag_pos, const double* __restrict__ Pdiag, const double* __restrict__ x_reg, double delta,
                           const double* __restrict__ z_reg, double* __restrict__ vals)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = n + p + m;
    if (col >= N) return;
    double v;
    if (col < n) v = Pdiag[col] + x_reg[col];        // kkt_full.hpp:181
    else if (col < n + p) v = -delta;                // :194
    else v = -z_reg[col - n - p];                    // :207
    vals[diag_pos[col]] = v;
}

// condensed modes (kkt_{eq,ineq,all}_eliminated.hpp update_kkt_*): diagonal of the top-left block += x_reg, of a kept
// constraint block = -delta / -z_reg; col indexes the columns of K in the caller's order (n, then kept p, then kept m)
// ... on a list of columns only (stage partition: the columns of the fronts this rank factors -- SURVEY 8(e) row 2, value assembly)
__global__ void k_set_diag_list(int ncols, const int* __restrict__ cols, int n, int p, const int* __restrict__ diag_pos, const double* __restrict__ Pdiag, const double* __restrict__ x_reg,
                                double delta, const double* __restrict__ z_reg, double* __restrict__ vals)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ncols) return;
    const int col = cols[t];
    double v;
    if (col < n) v = Pdiag[col] + x_reg[col];
    else if (col < n + p) v = -delta;
    else v = -z_reg[col - n - p];
    vals[diag_pos[col]] = v;
}
__global__ void k_norm_to_buf(const unsigned long long* __restrict__ bits, double* __restrict__ buf)  // |err|_inf as ordered bits -> double, NaN -> +inf (it must survive a max)
{
    const double v = __longlong_as_double((long long)bits[0]);
    buf[0] = v != v ? __builtin_huge_val() : v;
}
// (sel != nullptr, here and in the two kernels below: only the listed columns / entries -- the value assembly of a stage-partitioned handle)
__global__ void k_cond_diag(int n, int np, int nm, const int* __restrict__ diag_pos, const double* __restrict__ x_reg, double delta, const double* __restrict__ z_reg,
                            double* __restrict__ vals, const int* __restrict__ sel = nullptr, int nsel = 0)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (sel ? nsel : n + np + nm)) return;
    const int col = sel ? sel[idx] : idx;
    if (col < n) vals[diag_pos[col]] += x_reg[col];
    else if (col < n + np) vals[diag_pos[col]] = -delta;
    else vals[diag_pos[col]] = -z_reg[col - n - np];
}
// value of every entry of upper(MT diag(1/w) MT^T) from its product-term list (constraints ascending, the reference's order);
// w == nullptr: unit weights.  out[e] (ACC: += alpha * sum, through the index map dst) 
template <bool MAPPED>
__global__ void k_gram_values(int nent, const int* __restrict__ ptr, const int* __restrict__ q1, const int* __restrict__ q2, const int* __restrict__ kk,
                              const double* __restrict__ x, const double* __restrict__ w, const int* __restrict__ dst, double* __restrict__ out,
                              const int* __restrict__ sel = nullptr)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nent) return;
    const int e = sel ? sel[idx] : idx;
    double s = 0.0;
    if (w) for (int t = ptr[e]; t < ptr[e + 1]; ++t) s += x[q2[t]] * x[q1[t]] / w[kk[t]];
    else for (int t = ptr[e]; t < ptr[e + 1]; ++t) s += x[q2[t]] * x[q1[t]];
    if (MAPPED) out[dst[e]] += s; else out[e] = s;
}
__global__ void k_axpy_mapped(int nent, const int* __restrict__ dst, double alpha, const double* __restrict__ src, double* __restrict__ out, const int* __restrict__ sel = nullptr)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nent) return;
    const int e = sel ? sel[idx] : idx;
    out[dst[e]] += alpha * src[e];
}
__global__ void k_remap_values_sel(int nsel, const int* __restrict__ sel, const int* __restrict__ dst_idx, const double* __restrict__ src, double* __restrict__ dst)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < nsel) { const int q = sel[idx]; dst[dst_idx[q]] = src[q]; }
}
__global__ void k_reciprocal(int m, const double* __restrict__ z, double* __restrict__ zinv)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) zinv[i] = 1.0 / z[i];
}

__global__ void k_scatter_fronts(int nnz, const long long* __restrict__ a_dst, const double* __restrict__ vals, double* __restrict__ fronts)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nnz) fronts[a_dst[q]] = vals[q];
}

__global__ void k_perm_gather(int N, const int* __restrict__ P, const double* __restrict__ a, int na, const double* __restrict__ b, int nb, const double* __restrict__ c,
                              double* __restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    const int o = P[j];  // ordering.perm: x[j] = rhs[P[j]] with rhs = [a; b; c]
    out[j] = o < na ? a[o] : (o < na + nb ? b[o - na] : c[o - na - nb]);
}
__global__ void k_perm_scatter(int N, const int* __restrict__ P, const double* __restrict__ in, double* __restrict__ a, int na, double* __restrict__ b, int nb,
                               double* __restrict__ c, const int* __restrict__ err, int epoch)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    const int o = P[j];  // ordering.permt: rhs[P[j]] = x[j]
    double v = in[j];
    if (j == 0 && err && *err == epoch) v = __builtin_nan("");  // a flag wait of the sweeps gave up: the solution is poisoned, the caller's finite check reports it
    if (o < na) a[o] = v; else if (o < na + nb) b[o - na] = v; else c[o - na - nb] = v;
}
__global__ void k_scale(int N, const double* __restrict__ d, double* __restrict__ x)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < N) x[j] *= d[j];
}

// 1 / d for a pivot: v_rcp_f64 + two Newton steps + one residual correction (an IEEE division is ~250 dependent cycles on the critical
// path of every pivot; this is ~80 and agrees with it to the last bit except in rare half-ulp cases)
__device__ __forceinline__ double pivot_rcp(double d)
{
    double y = __builtin_amdgcn_rcp(d);
    y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
    y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
    return __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
}

// ---- the arithmetic of one elimination step (round 4).  The reference's up-looking LDLt forms every entry of row k as
//        y[j] -= L(j,i) * y_i   (product rounded, then the difference rounded: "force compiler to not use fma instruction", ldlt.hpp:151-153)
//        l_ki = y_i / D[i];  D[k] -= l_ki * y_i                                                       (ldlt.hpp:155-158)
// i.e. entry (r, c), r >= c, loses  fl( fl(y_ci / d_i) * y_ri ):  the QUOTIENT of the column index's value times the unscaled value of the row index.
// The one-workgroup fronts below do exactly that per term (before round 4: (y_ri * (1/d_i)) * y_ci with the subtraction fused), so that a pivot which
// cancels to an EXACT zero in the reference -- two variables tied by an equality row whose pivot is -delta: 1e10 + 8e-7 - 1e10, the signal that sends
// solver.hpp:688-708 into its recovery path on QBEACONF, fffff800, robot_arm_sqp -- does so here too whenever the entry receives its terms in the same
// order (always for entries with one or two terms; extend-add groups the terms of a child, the up-looking loop does not).
// y / d from the correctly rounded reciprocal: q = y dinv, one correction with the exact residual (Markstein) gives the IEEE quotient (2e8 random
// pairs: no mismatch) in three dependent operations instead of the ~250 cycles of a division.
__device__ __forceinline__ double ref_quot(double y, double d, double dinv)
{
    const double q = y * dinv;
    return __builtin_fma(__builtin_fma(-q, d, y), dinv, q);
}
__device__ __forceinline__ double ref_msub(double a, double x, double y) { return __dsub_rn(a, __dmul_rn(x, y)); }  // fl(a - fl(x y)): never contracted
// Which parts use that arithmetic (PQ_REF_MODE, bit mask; tools/exp_ref_arith.md records the variants measured in round 4):
//   1  REF_PIVOT  every multiply-subtract of the pivot loops and the scaling of the finished columns
//   2  REF_SCHUR  the one-pass Schur complement of the trailing block: term by term from T instead of "sum the w terms, subtract once"
//   4  REF_DIAG   the DIAGONAL entries only (panel and trailing block), everything else as in rounds 1-3 (fused, (y_r / d) y_c, summed Schur terms)
// A handle whose tree has no multi-workgroup front (they run on the matrix cores and cannot follow this arithmetic) uses PQ_REF_MODE, every other handle
// PQ_REF_MODE_BIG: the kernels below are instantiated for both (template parameter RM), SparseKKT::ref_mode_ picks one per handle.
#ifndef PQ_REF_MODE
#define PQ_REF_MODE 3
#endif
#ifndef PQ_REF_MODE_BIG
#define PQ_REF_MODE_BIG 0
#endif
#define PQ_LAUNCH_RM(kern, ...)                                                            \
    do {                                                                                   \
        if (ref_mode_ == PQ_REF_MODE) hipLaunchKernelGGL((kern<PQ_REF_MODE>), __VA_ARGS__); \
        else hipLaunchKernelGGL((kern<PQ_REF_MODE_BIG>), __VA_ARGS__);                      \
    } while (0)
#define PQ_ATTR_RM(kern, bytes)                                                                                                                     \
    do {                                                                                                                                            \
        PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern<PQ_REF_MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes)));        \
        PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern<PQ_REF_MODE_BIG>), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes)));    \
    } while (0)
#define PQ_REF_FLAGS constexpr bool REF_PIVOT = (RM & 1) != 0, REF_SCHUR = (RM & 2) != 0, REF_DIAG = (RM & 4) != 0; (void)REF_PIVOT; (void)REF_SCHUR; (void)REF_DIAG; static_assert(!(REF_SCHUR && REF_DIAG), "REF_SCHUR already treats the diagonal term by term")
constexpr int TD_REGS = 4;  // trailing diagonal entries a thread carries through the pivot loop (REF_DIAG); rows beyond TD_REGS x threads go through memory
// one panel entry of the pivot loop: W(i,j) loses the term of pivot k.  yi, yj: unscaled entries of column k in rows i and j; lj = yj / d (REF_PIVOT only)
template <int RM>
__device__ __forceinline__ double pivot_term(double wij, double yi, double yj, double d, double dinv, double lj, bool diag)
{
    PQ_REF_FLAGS;
    if constexpr (REF_PIVOT) return ref_msub(wij, yi, lj);
    else {
        if (REF_DIAG && diag) return ref_msub(wij, yj, ref_quot(yj, d, dinv));
        return wij - (yi * dinv) * yj;
    }
}
template <int RM>
__device__ __forceinline__ double scale_entry(double y, double d, double dinv) { return (RM & 1) ? ref_quot(y, d, dinv) : y * dinv; }
// REF_DIAG: the diagonal entries of the TRAILING rows i >= w (those of the update matrix) receive the reference's term of every pivot, one rounded product and one
// rounded difference each, in pivot order -- carried in registers through the pivot loop (thread t owns rows w + t + q nt), stored before the Schur
// complement, which then leaves the diagonal alone.  ptr(i): address of T(i,i); ycol(i): unscaled entry of the pivot column in row i.
template <class P>
__device__ __forceinline__ void td_load(double (&td)[TD_REGS], int w, int f, int tid, int nt, P ptr)
{
#pragma unroll
    for (int q = 0; q < TD_REGS; ++q) { const int i = w + tid + q * nt; td[q] = i < f ? *ptr(i) : 0.0; }
}
template <class P, class Y>
__device__ __forceinline__ void td_step(double (&td)[TD_REGS], int w, int f, int tid, int nt, P ptr, Y ycol, double d, double dinv)
{
#pragma unroll
    for (int q = 0; q < TD_REGS; ++q) {
        const int i = w + tid + q * nt;
        if (i < f) { const double y = ycol(i); td[q] = ref_msub(td[q], y, ref_quot(y, d, dinv)); }
    }
    for (int i = w + tid + TD_REGS * nt; i < f; i += nt) { double* p = ptr(i); const double y = ycol(i); *p = ref_msub(*p, y, ref_quot(y, d, dinv)); }
}
template <class P>
__device__ __forceinline__ void td_store(const double (&td)[TD_REGS], int w, int f, int tid, int nt, P ptr)
{
#pragma unroll
    for (int q = 0; q < TD_REGS; ++q) { const int i = w + tid + q * nt; if (i < f) *ptr(i) = td[q]; }
}

// Schur complement of the trailing u x u block in one pass (lower triangle), 2 x 2 entries per thread: T[i,j] loses fl(y_ik * l_jk) for k ascending,
// one rounded product and one rounded difference per pivot like the reference's row loop.  y: the UNSCALED panel entries of the trailing rows (still in
// place), l = y / d: written by scale_columns() into the strictly upper part of the front, entry (k, w + r) -- nothing else ever lives there.
template <bool SPLIT>  // SPLIT: the panel (LDS) and the trailing block T + upper part (HBM) are different arrays with the same f x f indexing
__device__ __forceinline__ void schur_2x2(const double* __restrict__ Lp, double* __restrict__ Tp, int f, int w, int u, int tid, int nt)
{
    const double* W = Lp;
    const double* Up = SPLIT ? Tp : Lp;
    const int tx = tid & 15, ty = tid >> 4, tys = nt >> 4;
    const int nb = (u + 1) >> 1;
    for (int bj = ty; bj < nb; bj += tys) {
        for (int bi = bj + tx; bi < nb; bi += 16) {
            const int i0 = 2 * bi, j0 = 2 * bj;
            const bool i1ok = i0 + 1 < u, j1ok = j0 + 1 < u;
            const double* Yi0 = W + (w + i0);
            const double* Yi1 = W + (w + (i1ok ? i0 + 1 : i0));
            const double* Lj0 = Up + (long long)(w + j0) * f;
            const double* Lj1 = Up + (long long)(w + (j1ok ? j0 + 1 : j0)) * f;
            double* T = Tp + (w + i0) + (long long)(w + j0) * f;
            const bool ok10 = i1ok, ok01 = j1ok && i0 >= j0 + 1, ok11 = i1ok && j1ok;  // (i0, j0 + 1): below / on the diagonal only
            double a00 = T[0], a10 = ok10 ? T[1] : 0.0, a01 = ok01 ? T[f] : 0.0, a11 = ok11 ? T[f + 1] : 0.0;
            for (int k = 0; k < w; ++k) {
                const long long ck = (long long)k * f;
                const double y0 = Yi0[ck], y1 = Yi1[ck], l0 = Lj0[k], l1 = Lj1[k];
                a00 = ref_msub(a00, y0, l0); a01 = ref_msub(a01, y0, l1); a10 = ref_msub(a10, y1, l0); a11 = ref_msub(a11, y1, l1);
            }
            T[0] = a00;                                    // i0 >= j0 always
            if (ok10) T[1] = a10;                          // (i0 + 1, j0)
            if (ok01) T[f] = a01;
            if (ok11) T[f + 1] = a11;                      // (i0 + 1, j0 + 1)
        }
    }
}
// rounds 1-3 (and the off-diagonal entries under REF_DIAG): T[i,j] -= sum_k (L[i,k] d_k) L[j,k], the w terms summed first; the columns are SCALED here
template <bool SPLIT, int RM>  // SPLIT: the panel L (LDS) and the trailing block T (HBM) are different arrays with the same f x f indexing
__device__ __forceinline__ void schur_2x2_sum(const double* __restrict__ Lp, double* __restrict__ Tp, int f, int w, int u, int tid, int nt)
{
    PQ_REF_FLAGS;
    const double* W = Lp;
    const int tx = tid & 15, ty = tid >> 4, tys = nt >> 4;
    const int nb = (u + 1) >> 1;
    for (int bj = ty; bj < nb; bj += tys) {
        for (int bi = bj + tx; bi < nb; bi += 16) {
            const int i0 = 2 * bi, j0 = 2 * bj;
            const bool i1ok = i0 + 1 < u, j1ok = j0 + 1 < u;
            const double* Li0 = W + (w + i0);
            const double* Li1 = W + (w + (i1ok ? i0 + 1 : i0));
            const double* Lj0 = W + (w + j0);
            const double* Lj1 = W + (w + (j1ok ? j0 + 1 : j0));
            double a00 = 0.0, a01 = 0.0, a10 = 0.0, a11 = 0.0;
            for (int k = 0; k < w; ++k) {
                const long long ck = (long long)k * f;
                const double dk = W[k + ck];
                const double x0 = Li0[ck] * dk, x1 = Li1[ck] * dk, y0 = Lj0[ck], y1 = Lj1[ck];
                a00 += x0 * y0; a01 += x0 * y1; a10 += x1 * y0; a11 += x1 * y1;
            }
            double* T = Tp + (w + i0) + (long long)(w + j0) * f;
            const bool dg = REF_DIAG && i0 == j0;          // (REF_DIAG: the diagonal entries were updated pivot by pivot, front_factor_body)
            if (!dg) T[0] -= a00;                          // i0 >= j0 always
            if (i1ok) T[1] -= a10;                         // (i0 + 1, j0)
            if (j1ok && i0 >= j0 + 1) T[f] -= a01;         // (i0, j0 + 1): below / on the diagonal only
            if (i1ok && j1ok && !dg) T[f + 1] -= a11;      // (i0 + 1, j0 + 1)
        }
    }
}
// After the pivot loop the columns 0 .. w-1 hold UNSCALED entries below the diagonal.  Every one becomes l = y / d_k (the reference's L(i,k), ldlt.hpp:155):
// the panel rows in place; the trailing rows i >= w into the strictly upper part Up[k + i f] (the Schur complement needs their unscaled values as well),
// copied into place by place_trailing() afterwards.
__device__ __forceinline__ void scale_columns(double* __restrict__ W, double* __restrict__ Up, int f, int w, int tid, int nt)
{
    const int tx = tid & 15, ty = tid >> 4, tys = nt >> 4;
    for (int k = ty; k < w; k += tys) {
        double d = W[k + (long long)k * f];
        if (d == 0.0) d = 1.0;
        const double dinv = pivot_rcp(d);
        for (int i = k + 1 + tx; i < f; i += 16) {
            const double l = ref_quot(W[i + (long long)k * f], d, dinv);
            if (i < w) W[i + (long long)k * f] = l;
            else Up[k + (long long)i * f] = l;
        }
    }
}
__device__ __forceinline__ void place_trailing(double* __restrict__ W, const double* __restrict__ Up, int f, int w, int tid, int nt)
{
    const int tx = tid & 15, ty = tid >> 4, tys = nt >> 4;
    for (int k = ty; k < w; k += tys)
        for (int i = w + tx; i < f; i += 16) W[i + (long long)k * f] = Up[k + (long long)i * f];
}

// extend-add of every child's update matrix into front s (fixed child order)
__device__ __forceinline__ void extend_add(const FrontMeta& M, double* __restrict__ fronts, int s, double* __restrict__ F, int f)
{
    const SnRec me = M.sn[s];
    for (int ci = me.child_lo; ci < me.child_hi; ++ci) {
        const SnRec ch = M.sn[M.child[ci]];
        const int wc = ch.w, fc = ch.f;
        const int uc = fc - wc;
        const double* U = fronts + ch.front_off + wc + (long long)wc * fc;
        const int* rel = M.rel + ch.rel_ptr;
        {   // 16 x (threads / 16) thread grid over the lower triangle: no integer division in the index arithmetic
            const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4, tys = blockDim.x >> 4;
            for (int j = ty; j < uc; j += tys) {
                const long long cj = (long long)rel[j] * f;
                const double* __restrict__ Uj = U + (long long)j * fc;
                // four entries per step, all loads before the first store: the child's update matrix and this front never overlap, but both hang off
                // `fronts`, and a load the compiler must keep behind the previous store costs a memory round trip per entry (front in HBM)
                for (int i = j + tx; i < uc; i += 64) {
                    double uv[4], fv[4];
                    long long at[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int iq = i + 16 * q;
                        const bool ok = iq < uc;
                        at[q] = ok ? rel[iq] + cj : -1;
                        uv[q] = ok ? Uj[iq] : 0.0;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) fv[q] = at[q] >= 0 ? F[at[q]] : 0.0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (at[q] >= 0) F[at[q]] = fv[q] + uv[q];
                }
            }
        }
        __syncthreads();
    }
}

// One workgroup per front: extend-add, then right-looking LDLt of the first w columns (unit L below the
// diagonal, D on it), Schur complement left in the trailing (f-w) x (f-w) block for the parent.
// Fails (info = first failing global column) iff a pivot is exactly zero, like ldlt.hpp:163.
// zero + the K entries the front owns: needs nothing from the children, so the persistent top kernel runs it while they are still busy
__device__ __forceinline__ void front_assemble_own(const FrontMeta& M, double* __restrict__ fronts, const SnRec& me, double* __restrict__ lds)
{
    // (two copies of the loops instead of one pointer chosen at run time: a pointer that may be LDS or HBM compiles to FLAT accesses)
    const int f = me.f;
    if ((long long)f * f <= LDS_FRONT_DOUBLES) {
        for (int idx = threadIdx.x; idx < f * f; idx += blockDim.x) lds[idx] = 0.0;
        __syncthreads();
        for (int e = me.fe_lo + threadIdx.x; e < me.fe_hi; e += blockDim.x) lds[M.fe_off[e]] = M.vals[e];
    } else {
        double* W = fronts + me.front_off;
        for (int idx = threadIdx.x; idx < f * f; idx += blockDim.x) W[idx] = 0.0;
        __syncthreads();
        for (int e = me.fe_lo + threadIdx.x; e < me.fe_hi; e += blockDim.x) W[M.fe_off[e]] = M.vals[e];
    }
    __syncthreads();
}
// Where the front is worked on -- FRONT_LDS: all of it in LDS (W == lds), copied out at the end; FRONT_HBM: in place (W == F); FRONT_PANEL: assembled in
// place, then its f x w pivot panel is staged in LDS for the pivot loop (W == lds, same column stride f) and only the Schur complement goes back to the
// trailing block in HBM.  Same operations in the same order in all three.  Instantiations instead of one pointer chosen at run time -- such a
// pointer compiles to FLAT loads / stores (k_front_factor had 20 + 10 of them in its pivot loops).
enum { FRONT_HBM = 0, FRONT_LDS = 1, FRONT_PANEL = 2, FRONT_PANEL_ONLY = 3 };  // PANEL_ONLY: assembled by other kernels, Schur complement by another kernel (D left in `dvec`)
template <int WHERE, int RM>
__device__ __forceinline__ void front_factor_body(const FrontMeta& M, double* __restrict__ fronts, int s, const SnRec& me, double* __restrict__ rdiag,
                                                  int* __restrict__ info, double* __restrict__ lds, double* __restrict__ W, double* __restrict__ dvec = nullptr)
{
    PQ_REF_FLAGS;
    const int first = me.first, w = me.w, f = me.f;
    double* F = fronts + me.front_off;
    constexpr bool in_lds = WHERE == FRONT_LDS;
    constexpr bool panel = WHERE == FRONT_PANEL || WHERE == FRONT_PANEL_ONLY;
    if constexpr (panel) {
        if constexpr (WHERE == FRONT_PANEL) extend_add(M, fronts, s, F, f);
        {   // stage the panel, eight loads in flight per thread
            const int n = f * w, nt8 = 8 * blockDim.x;
            for (int base = 0; base < n; base += nt8) {
                double v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { const int idx = base + q * blockDim.x + threadIdx.x; v[q] = idx < n ? F[idx] : 0.0; }
#pragma unroll
                for (int q = 0; q < 8; ++q) { const int idx = base + q * blockDim.x + threadIdx.x; if (idx < n) lds[idx] = v[q]; }
            }
        }
        __syncthreads();
    } else {
        extend_add(M, fronts, s, W, f);
    }
    // ---- panel: right-looking LDLt of the first w columns, updates confined to the panel (rows k+1..f-1, columns k+1..w-1)
    const int tid = threadIdx.x, nt = blockDim.x;
    int k0 = 0;
    constexpr bool trail_diag = REF_DIAG && WHERE != FRONT_PANEL_ONLY;  // (the trailing update of a PANEL_ONLY front runs on the matrix cores)
    double* Tt = (WHERE == FRONT_PANEL) ? F : W;  // where the trailing block lives
    auto tptr = [&](int i) -> double* { return Tt + i + (long long)i * f; };
    double td[TD_REGS];
    if constexpr (trail_diag) td_load(td, w, f, tid, nt, tptr);
    if (me.nind >= 2) {
        // independent leading pivots (merged sibling leaves): one pass instead of nind barriers, see independent_pivots_pk
        const int ni = min(me.nind, IND_SCRATCH);
        double* scratch = in_lds ? lds + f * f : (panel ? lds + f * w : lds);
        const int tx = tid & 15, ty = tid >> 4, tys = nt >> 4;
        for (int k = tid; k < ni; k += nt) {
            double d = W[k + (long long)k * f];
            if (d == 0.0) { if (*info < 0) *info = first + k; d = 1.0; }
            const double dinv = pivot_rcp(d);
            scratch[k] = dinv;
            rdiag[first + k] = dinv;
        }
        __syncthreads();
        for (int j = ni + ty; j < w; j += tys) {
            double* Wj = W + (long long)j * f;
            for (int i = j + tx; i < f; i += 16) {
                double a = Wj[i];
                for (int k = 0; k < ni; ++k) {
                    const double* Ck = W + (long long)k * f;
                    const double cj = Ck[j];
                    if (cj != 0.0) {
                        double dk = 1.0;
                        if constexpr (REF_PIVOT || REF_DIAG) { dk = Ck[k]; if (dk == 0.0) dk = 1.0; }
                        a = pivot_term<RM>(a, Ck[i], cj, dk, scratch[k], REF_PIVOT ? ref_quot(cj, dk, scratch[k]) : 0.0, i == j);
                    }
                }
                Wj[i] = a;
            }
        }
        if constexpr (trail_diag) {
            for (int k = 0; k < ni; ++k) {
                const double* Ck = W + (long long)k * f;
                double dk = Ck[k];
                if (dk == 0.0) dk = 1.0;
                td_step(td, w, f, tid, nt, tptr, [&](int i) { return Ck[i]; }, dk, scratch[k]);
            }
        }
        __syncthreads();
        k0 = ni;
    }
    for (int k = k0; k < w; ++k) {
        double d = W[k + (long long)k * f];  // every thread reads the same word (broadcast)
        if (d == 0.0) { if (tid == 0 && *info < 0) *info = first + k; d = 1.0; }
        const double dinv = pivot_rcp(d);
        if (tid == 0) rdiag[first + k] = dinv;
        const int r = f - k - 1, pc = w - k - 1;
        const double* colk = W + (k + 1) + (long long)k * f;
        // W[i,j] -= (a_i / d) * a_j for k < j < w, i >= j, with the UNSCALED column k (column k is final after this step: its
        // scaling by 1/d is deferred to one pass after the loop -> one barrier per pivot instead of two); the arithmetic of a term: pivot_term<RM>()
        // (Round 3 tried two columns of a thread's stride per pass, the scaled entry a_i / d read once for both -- five LDS accesses per two multiply-adds
        // instead of six, bitwise the same: CONT-201 factorisation 2.29 instead of 2.21 ms, C3 unchanged.  Not kept.)
        {
            const int tx = tid & 15, ty = tid >> 4, tys = nt >> 4;
            for (int j = ty; j < pc; j += tys) {
                const double cj = colk[j];
                const double lj = REF_PIVOT ? ref_quot(cj, d, dinv) : 0.0;
                double* Wj = W + (k + 1) + (long long)(k + 1 + j) * f;
                for (int i = j + tx; i < r; i += 16) Wj[i] = pivot_term<RM>(Wj[i], colk[i], cj, d, dinv, lj, i == j);
            }
        }
        if constexpr (trail_diag) td_step(td, w, f, tid, nt, tptr, [&](int i) { return W[i + (long long)k * f]; }, d, dinv);
        __syncthreads();
    }
    if constexpr (trail_diag) td_store(td, w, f, tid, nt, tptr);
    if constexpr (WHERE == FRONT_PANEL_ONLY || !REF_SCHUR) {
        // every entry below the diagonal is scaled in place (the trailing update of a PANEL_ONLY front runs on the matrix cores, k_syrk_lower_fronts)
        const int tx = tid & 15, ty = tid >> 4, tys = nt >> 4;
        for (int k = ty; k < w; k += tys) {
            double d = W[k + (long long)k * f];
            if (d == 0.0) d = 1.0;
            const double dinv = pivot_rcp(d);
            for (int i = k + 1 + tx; i < f; i += 16) W[i + (long long)k * f] = scale_entry<RM>(W[i + (long long)k * f], d, dinv);
        }
        __syncthreads();
    }
    if constexpr (WHERE == FRONT_PANEL_ONLY) {
        for (int k = tid; k < w; k += nt) { const double d = lds[k + k * f]; dvec[k] = d == 0.0 ? 1.0 : d; }
        for (int idx = tid; idx < f * w; idx += nt) F[idx] = lds[idx];
        return;
    }
    const int u = f - w;
    if constexpr (REF_SCHUR) {
        // ---- l = y / d (trailing rows: into the strictly upper part), then the Schur complement of the trailing block in one pass from the unscaled
        //      trailing rows and those quotients (no barriers, long chains), then the trailing rows' quotients into place
        double* Up = (WHERE == FRONT_PANEL) ? F : W;
        scale_columns(W, Up, f, w, tid, nt);
        __syncthreads();
        if (u > 0) {
            if constexpr (WHERE == FRONT_PANEL) schur_2x2<true>(lds, F, f, w, u, tid, nt);
            else schur_2x2<false>(W, W, f, w, u, tid, nt);
            __syncthreads();
            place_trailing(W, Up, f, w, tid, nt);
            __syncthreads();
        }
    } else if (u > 0) {
        // ---- Schur complement of the trailing block in one pass: T[i,j] -= sum_k L[i,k] d_k L[j,k]  (no barriers, long dot products)
        if constexpr (WHERE == FRONT_PANEL) schur_2x2_sum<true, RM>(lds, F, f, w, u, tid, nt);
        else schur_2x2_sum<false, RM>(W, W, f, w, u, tid, nt);
        __syncthreads();
    }
    if constexpr (WHERE == FRONT_PANEL) {
        for (int idx = threadIdx.x; idx < f * w; idx += blockDim.x) F[idx] = lds[idx];
    }
    if (in_lds) {
        // the factor panel (contiguous) and the lower triangle of the update matrix: nobody reads the rest
        for (int idx = threadIdx.x; idx < f * w; idx += blockDim.x) F[idx] = lds[idx];
        const int tx = tid & 15, ty = tid >> 4, tys = nt >> 4;
        for (int j = w + ty; j < f; j += tys)
            for (int i = j + tx; i < f; i += 16) F[i + (long long)j * f] = lds[i + j * f];
    }
}

template <int RM>
__device__ __forceinline__ void front_factor(const FrontMeta& M, double* __restrict__ fronts, int s, double* __restrict__ rdiag,
                                             int* __restrict__ info, double* __restrict__ lds, bool own_assembled = false)
{
    const SnRec me = M.sn[s];
    const int w = me.w, f = me.f;
    // assembly: zero, own K entries, then the children's update matrices (fixed order)
    if (!own_assembled) front_assemble_own(M, fronts, me, lds);
    if ((long long)f * f <= LDS_FRONT_DOUBLES) front_factor_body<FRONT_LDS, RM>(M, fronts, s, me, rdiag, info, lds, lds);
    else if (front_lds_doubles(f, w) > 0) front_factor_body<FRONT_PANEL, RM>(M, fronts, s, me, rdiag, info, lds, lds);
    else front_factor_body<FRONT_HBM, RM>(M, fronts, s, me, rdiag, info, lds, fronts + me.front_off);
}

// One workgroup per front of an assembly-tree level.  Fronts on the multi-workgroup path (job_of[s] >= 0) were assembled by its kernels: a big
// front is left to the dense kernels, a panel_front() has its panel factored here (D left in the job's dvec for the trailing update) -- in the
// same launch as the level's one-workgroup fronts, which are independent of it.
template <int RM>
__global__ __launch_bounds__(1024) void k_front_factor(FrontMeta M, double* __restrict__ fronts, const int* __restrict__ list, const int* __restrict__ job_of,
                                                      const dense::FrontJob* __restrict__ jobs, double* __restrict__ rdiag, int* __restrict__ info)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int s = list[blockIdx.x];
    const int jid = job_of ? job_of[s] : -1;
    if (jid < 0) { front_factor<RM>(M, fronts, s, rdiag, info, lds); return; }
    const dense::FrontJob j = jobs[jid];
    if (j.kind != 1) return;
    const SnRec me = M.sn[s];
    front_factor_body<FRONT_PANEL_ONLY, RM>(M, fronts, s, me, rdiag, info, lds, lds, j.dvec);
}

// One workgroup per small subtree, fronts never leave LDS: the front of supernode s is zeroed and assembled in LDS from the
// K entries it owns (fe lists), children are merged from LDS when the child is the previous supernode of the walk (the usual
// case: a chain) and from HBM otherwise, the panel + Schur complement run in LDS, and only the factor panel (f x w) is written
// to HBM; the update matrix goes to HBM only when the parent is not the next supernode of this walk.  Two LDS buffers
// alternate between "front being factored" and "previous front, holding the update matrix".
// Packed variant of the LDS subtree walk: both front buffers hold only the lower triangle (column j starts at j (2f - j + 1) / 2),
// which halves the LDS of a walk -- a subtree with a 96 x 96 front drops from 147 KB to 74.5 KB and two workgroups share a CU.
// pk_base(j, f) + i is the slot of entry (i, j), i >= j.
__device__ __forceinline__ int pk_base(int j, int f) { return (j * (2 * f - j - 1)) >> 1; }
// (packed fronts have no upper part: the quotients l = y / d of the trailing rows go to `Lb`, entry (k, r) at Lb[k + r w] -- the walk's other LDS buffer,
// whose update matrix the extend-add has consumed by then)
__device__ __forceinline__ void schur_2x2_pk(double* __restrict__ W, const double* __restrict__ Lb, int f, int w, int u, int tid, int nt)
{
    const int tx = tid & 15, ty = tid >> 4, tys = nt >> 4;
    const int nb = (u + 1) >> 1;
    for (int bj = ty; bj < nb; bj += tys) {
        for (int bi = bj + tx; bi < nb; bi += 16) {
            const int i0 = 2 * bi, j0 = 2 * bj;
            const bool i1ok = i0 + 1 < u, j1ok = j0 + 1 < u;
            const int ri0 = w + i0, ri1 = w + (i1ok ? i0 + 1 : i0);
            const double* Lj0 = Lb + j0 * w;
            const double* Lj1 = Lb + (j1ok ? j0 + 1 : j0) * w;
            double* T0 = W + pk_base(w + j0, f) + (w + i0);
            double* T1 = W + pk_base(w + (j1ok ? j0 + 1 : j0), f) + (w + i0);
            const bool ok10 = i1ok, ok01 = j1ok && i0 >= j0 + 1, ok11 = i1ok && j1ok;
            double a00 = T0[0], a10 = ok10 ? T0[1] : 0.0, a01 = ok01 ? T1[0] : 0.0, a11 = ok11 ? T1[1] : 0.0;
            int ck = 0;  // pk_base(k, f)
            for (int k = 0; k < w; ++k) {
                const double* Ck = W + ck;
                const double y0 = Ck[ri0], y1 = Ck[ri1], l0 = Lj0[k], l1 = Lj1[k];
                a00 = ref_msub(a00, y0, l0); a01 = ref_msub(a01, y0, l1); a10 = ref_msub(a10, y1, l0); a11 = ref_msub(a11, y1, l1);
                ck += f - k - 1;
            }
            T0[0] = a00;
            if (ok10) T0[1] = a10;
            if (ok01) T1[0] = a01;
            if (ok11) T1[1] = a11;
        }
    }
}
template <int RM>
__device__ __forceinline__ void schur_2x2_pk_sum(double* __restrict__ W, int f, int w, int u, int tid, int nt)
{
    PQ_REF_FLAGS;
    const int tx = tid & 15, ty = tid >> 4, tys = nt >> 4;
    const int nb = (u + 1) >> 1;
    for (int bj = ty; bj < nb; bj += tys) {
        for (int bi = bj + tx; bi < nb; bi += 16) {
            const int i0 = 2 * bi, j0 = 2 * bj;
            const bool i1ok = i0 + 1 < u, j1ok = j0 + 1 < u;
            const int ri0 = w + i0, ri1 = w + (i1ok ? i0 + 1 : i0), rj0 = w + j0, rj1 = w + (j1ok ? j0 + 1 : j0);
            double a00 = 0.0, a01 = 0.0, a10 = 0.0, a11 = 0.0;
            int ck = 0;  // pk_base(k, f)
            for (int k = 0; k < w; ++k) {
                const double* Ck = W + ck;
                const double dk = Ck[k];
                const double x0 = Ck[ri0] * dk, x1 = Ck[ri1] * dk, y0 = Ck[rj0], y1 = Ck[rj1];
                a00 += x0 * y0; a01 += x0 * y1; a10 += x1 * y0; a11 += x1 * y1;
                ck += f - k - 1;
            }
            double* T0 = W + pk_base(w + j0, f) + (w + i0);
            const bool dg = REF_DIAG && i0 == j0;
            if (!dg) T0[0] -= a00;
            if (i1ok) T0[1] -= a10;
            if (j1ok) {
                double* T1 = W + pk_base(w + j0 + 1, f) + (w + i0);
                if (i0 >= j0 + 1) T1[0] -= a01;
                if (i1ok && !dg) T1[1] -= a11;
            }
        }
    }
}
__device__ __forceinline__ void scale_columns_pk(double* __restrict__ W, double* __restrict__ Lb, int f, int w, int tid, int nt)
{
    const int tx = tid & 15, ty = tid >> 4, tys = nt >> 4;
    for (int k = ty; k < w; k += tys) {
        double* Ck = W + pk_base(k, f);
        double d = Ck[k];
        if (d == 0.0) d = 1.0;
        const double dinv = pivot_rcp(d);
        for (int i = k + 1 + tx; i < f; i += 16) {
            const double l = ref_quot(Ck[i], d, dinv);
            if (i < w) Ck[i] = l;
            else Lb[k + (i - w) * w] = l;
        }
    }
}
__device__ __forceinline__ void place_trailing_pk(double* __restrict__ W, const double* __restrict__ Lb, int f, int w, int tid, int nt)
{
    const int tx = tid & 15, ty = tid >> 4, tys = nt >> 4;
    for (int k = ty; k < w; k += tys) {
        double* Ck = W + pk_base(k, f);
        for (int i = w + tx; i < f; i += 16) Ck[i] = Lb[k + (i - w) * w];
    }
}

// The first `nind` pivots of a front are mutually independent (merged sibling leaves: their mutual block is structurally zero), so they
// need no pivot-by-pivot loop: their reciprocals go to `scratch` (LDS) and every entry of the remaining panel columns receives its nind
// updates in one pass, in the same pivot order and with the same operands as the per-pivot loop (bitwise the same result).  Returns the
// first pivot the per-pivot loop still has to do.
template <int RM>
__device__ __forceinline__ int independent_pivots_pk(double* __restrict__ W, int f, int w, int nind, int first, double* __restrict__ scratch, double* __restrict__ rdiag,
                                                     int* __restrict__ info, int tid, int nt, double (&td)[TD_REGS])
{
    PQ_REF_FLAGS;
    if (nind < 2) return 0;
    const int tx = tid & 15, ty = tid >> 4, tys = nt >> 4;
    for (int k = tid; k < nind; k += nt) {
        double d = W[pk_base(k, f) + k];
        if (d == 0.0) { if (*info < 0) *info = first + k; d = 1.0; }
        const double dinv = pivot_rcp(d);
        scratch[k] = dinv;
        rdiag[first + k] = dinv;
    }
    __syncthreads();
    for (int j = nind + ty; j < w; j += tys) {
        double* Wj = W + pk_base(j, f);
        for (int i = j + tx; i < f; i += 16) {
            double a = Wj[i];
            int ck = 0;  // pk_base(k, f)
            for (int k = 0; k < nind; ++k) {
                const double* Ck = W + ck;
                const double cj = Ck[j];
                if (cj != 0.0) {
                    double dk = 1.0;
                    if constexpr (REF_PIVOT || REF_DIAG) { dk = Ck[k]; if (dk == 0.0) dk = 1.0; }
                    a = pivot_term<RM>(a, Ck[i], cj, dk, scratch[k], REF_PIVOT ? ref_quot(cj, dk, scratch[k]) : 0.0, i == j);
                }
                ck += f - k - 1;
            }
            Wj[i] = a;
        }
    }
    if constexpr (REF_DIAG) {
        int ck = 0;
        for (int k = 0; k < nind; ++k) {
            const double* Ck = W + ck;
            double dk = Ck[k];
            if (dk == 0.0) dk = 1.0;
            td_step(td, w, f, tid, nt, [&](int i) -> double* { return W + pk_base(i, f) + i; }, [&](int i) { return Ck[i]; }, dk, scratch[k]);
            ck += f - k - 1;
        }
    }
    __syncthreads();
    return nind;
}
template <int RM>
__global__ __launch_bounds__(1024) void k_subtree_factor_pk(FrontMeta M, double* __restrict__ fronts, const double* __restrict__ vals, const int* __restrict__ fe_offp,
                                                           const int* __restrict__ sub_lo, const int* __restrict__ sub_hi, int cap, double* __restrict__ rdiag,
                                                           int* __restrict__ info)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    PQ_REF_FLAGS;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int tx = tid & 15, ty = tid >> 4, tys = nt >> 4;
    const int lo = sub_lo[blockIdx.x], hi = sub_hi[blockIdx.x];
    double* cur = lds;
    double* prev = lds + cap;
    bool prev_valid = false;
    for (int s = lo; s <= hi; ++s) {
        const SnRec me = M.sn[s];
        const int first = me.first, w = me.w, f = me.f;
        double* W = cur;
        const int npk = (f * (f + 1)) >> 1;
        for (int idx = tid; idx < npk; idx += nt) W[idx] = 0.0;
        __syncthreads();
        for (int e = me.fe_lo + tid; e < me.fe_hi; e += nt) W[fe_offp[e]] = vals[e];
        __syncthreads();
        for (int ci = me.child_lo; ci < me.child_hi; ++ci) {
            const int c = M.child[ci];
            const SnRec ch = M.sn[c];
            const int wc = ch.w, fc = ch.f, uc = fc - wc;
            const bool from_lds = prev_valid && c == s - 1;
            const int* rel = M.rel + ch.rel_ptr;
            if (from_lds) {
                for (int j = ty; j < uc; j += tys) {
                    const int cj = pk_base(rel[j], f);
                    const double* Uj = prev + pk_base(wc + j, fc) + wc;
                    for (int i = j + tx; i < uc; i += 16) W[cj + rel[i]] += Uj[i];
                }
            } else {
                const double* U = fronts + ch.front_off + wc + (long long)wc * fc;
                for (int j = ty; j < uc; j += tys) {
                    const int cj = pk_base(rel[j], f);
                    for (int i = j + tx; i < uc; i += 16) W[cj + rel[i]] += U[i + (long long)j * fc];
                }
            }
            __syncthreads();
        }
        // ---- panel
        auto tptr = [&](int i) -> double* { return W + pk_base(i, f) + i; };
        double td[TD_REGS];
        if constexpr (REF_DIAG) td_load(td, w, f, tid, nt, tptr);
        for (int k = independent_pivots_pk<RM>(W, f, w, me.nind, first, prev, rdiag, info, tid, nt, td); k < w; ++k) {
            const int ck = pk_base(k, f);
            double d = W[ck + k];
            if (d == 0.0) { if (tid == 0 && *info < 0) *info = first + k; d = 1.0; }
            const double dinv = pivot_rcp(d);
            if (tid == 0) rdiag[first + k] = dinv;
            const int r = f - k - 1, pc = w - k - 1;
            const double* colk = W + ck + (k + 1);
            for (int j = ty; j < pc; j += tys) {
                const double cj = colk[j];
                const double lj = REF_PIVOT ? ref_quot(cj, d, dinv) : 0.0;
                double* Wj = W + pk_base(k + 1 + j, f) + (k + 1);
                for (int i = j + tx; i < r; i += 16) Wj[i] = pivot_term<RM>(Wj[i], colk[i], cj, d, dinv, lj, i == j);
            }
            if constexpr (REF_DIAG) td_step(td, w, f, tid, nt, tptr, [&](int i) { return W[ck + i]; }, d, dinv);
            __syncthreads();
        }
        if constexpr (REF_DIAG) td_store(td, w, f, tid, nt, tptr);
        const int u = f - w;
        if constexpr (REF_SCHUR) {
            // deferred division of the finished columns, Schur complement, trailing rows' quotients into place (see front_factor_body)
            scale_columns_pk(W, prev, f, w, tid, nt);
            __syncthreads();
            if (u > 0) {
                schur_2x2_pk(W, prev, f, w, u, tid, nt);
                __syncthreads();
                place_trailing_pk(W, prev, f, w, tid, nt);
                __syncthreads();
            }
        } else {
            for (int k = ty; k < w; k += tys) {   // deferred scaling of the finished columns (see front_factor)
                double* Ck = W + pk_base(k, f);
                double d = Ck[k];
                if (d == 0.0) d = 1.0;
                const double dinv = pivot_rcp(d);
                for (int i = k + 1 + tx; i < f; i += 16) Ck[i] = scale_entry<RM>(Ck[i], d, dinv);
            }
            __syncthreads();
            if (u > 0) {
                schur_2x2_pk_sum<RM>(W, f, w, u, tid, nt);
                __syncthreads();
            }
        }
        // ---- factor panel to HBM (full column-major layout there: the solves and the parents outside the subtree read it)
        double* F = fronts + me.front_off;
        for (int j = ty; j < w; j += tys) {
            const double* Cj = W + pk_base(j, f);
            double* Fj = F + (long long)j * f;
            for (int i = j + tx; i < f; i += 16) Fj[i] = Cj[i];
        }
        const bool keep = me.parent == s + 1 && s + 1 <= hi;
        if (!keep && u > 0) {
            for (int j = ty; j < u; j += tys) {
                const double* Cj = W + pk_base(w + j, f) + w;
                for (int i = j + tx; i < u; i += 16) F[(w + i) + (long long)(w + j) * f] = Cj[i];
            }
        }
        __syncthreads();
        double* t = cur; cur = prev; prev = t;
        prev_valid = keep;
    }
}

template <int RM>
__global__ __launch_bounds__(1024) void k_subtree_factor_lds(FrontMeta M, double* __restrict__ fronts, const double* __restrict__ vals, const int* __restrict__ fe_ptr,
                                                            const int* __restrict__ fe_q, const int* __restrict__ fe_off, const int* __restrict__ sub_lo,
                                                            const int* __restrict__ sub_hi, int cap, double* __restrict__ rdiag, int* __restrict__ info)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    PQ_REF_FLAGS;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lo = sub_lo[blockIdx.x], hi = sub_hi[blockIdx.x];
    double* cur = lds;
    double* prev = lds + cap;
    bool prev_valid = false;  // prev holds the factored front of supernode s-1 (its trailing block = update matrix)
    int prev_f = 0, prev_w = 0;
    for (int s = lo; s <= hi; ++s) {
        const SnRec me = M.sn[s];
        const int first = me.first, w = me.w, f = me.f;
        double* W = cur;
        for (int idx = tid; idx < f * f; idx += nt) W[idx] = 0.0;
        __syncthreads();
        for (int e = me.fe_lo + tid; e < me.fe_hi; e += nt) W[fe_off[e]] = vals[e];
        __syncthreads();
        for (int ci = me.child_lo; ci < me.child_hi; ++ci) {
            const int c = M.child[ci];
            const SnRec ch = M.sn[c];
            const int wc = ch.w, fc = ch.f, uc = fc - wc;
            const bool from_lds = prev_valid && c == s - 1;
            const double* U = from_lds ? prev + wc + wc * fc : fronts + ch.front_off + wc + (long long)wc * fc;
            const int* rel = M.rel + ch.rel_ptr;
            {
                const int tx = tid & 15, ty = tid >> 4, tys = nt >> 4;
                for (int j = ty; j < uc; j += tys) {
                    const int cj = rel[j] * f;
                    for (int i = j + tx; i < uc; i += 16) W[rel[i] + cj] += U[i + (long long)j * fc];
                }
            }
            __syncthreads();
        }
        (void)prev_f; (void)prev_w;
        // ---- panel
        auto tptr = [&](int i) -> double* { return W + i + i * f; };
        double td[TD_REGS];
        if constexpr (REF_DIAG) td_load(td, w, f, tid, nt, tptr);
        for (int k = 0; k < w; ++k) {
            double d = W[k + k * f];
            if (d == 0.0) { if (tid == 0 && *info < 0) *info = first + k; d = 1.0; }
            const double dinv = pivot_rcp(d);
            if (tid == 0) rdiag[first + k] = dinv;
            const int r = f - k - 1, pc = w - k - 1;
            const double* colk = W + (k + 1) + k * f;
            {
                const int tx = tid & 15, ty = tid >> 4, tys = nt >> 4;
                for (int j = ty; j < pc; j += tys) {
                    const double cj = colk[j];
                    const double lj = REF_PIVOT ? ref_quot(cj, d, dinv) : 0.0;
                    double* Wj = W + (k + 1) + (k + 1 + j) * f;
                    for (int i = j + tx; i < r; i += 16) Wj[i] = pivot_term<RM>(Wj[i], colk[i], cj, d, dinv, lj, i == j);
                }
            }
            if constexpr (REF_DIAG) td_step(td, w, f, tid, nt, tptr, [&](int i) { return W[i + k * f]; }, d, dinv);
            __syncthreads();
        }
        if constexpr (REF_DIAG) td_store(td, w, f, tid, nt, tptr);
        const int u = f - w;
        if constexpr (REF_SCHUR) {
            // deferred division of the finished columns, Schur complement, trailing rows' quotients into place (see front_factor_body)
            scale_columns(W, W, f, w, tid, nt);
            __syncthreads();
            if (u > 0) {
                schur_2x2<false>(W, W, f, w, u, tid, nt);
                __syncthreads();
                place_trailing(W, W, f, w, tid, nt);
                __syncthreads();
            }
        } else {
            {   // deferred scaling of the finished columns (see front_factor)
                const int tx = tid & 15, ty = tid >> 4, tys = nt >> 4;
                for (int k = ty; k < w; k += tys) {
                    double d = W[k + k * f];
                    if (d == 0.0) d = 1.0;
                    const double dinv = pivot_rcp(d);
                    for (int i = k + 1 + tx; i < f; i += 16) W[i + k * f] = scale_entry<RM>(W[i + k * f], d, dinv);
                }
            }
            __syncthreads();
            // ---- Schur complement
            if (u > 0) {
                schur_2x2_sum<false, RM>(W, W, f, w, u, tid, nt);
                __syncthreads();
            }
        }
        // ---- factor panel to HBM (the first w columns of the front are contiguous); update matrix only if nobody reads it from LDS
        double* F = fronts + me.front_off;
        for (int idx = tid; idx < f * w; idx += nt) F[idx] = W[idx];
        const bool keep = me.parent == s + 1 && s + 1 <= hi;
        if (!keep && u > 0) {
            const int tx = tid & 15, ty = tid >> 4, tys = nt >> 4;
            for (int j = ty; j < u; j += tys)
                for (int i = j + tx; i < u; i += 16) F[(w + i) + (long long)(w + j) * f] = W[(w + i) + (w + j) * f];
        }
        __syncthreads();
        double* t = cur; cur = prev; prev = t;
        prev_valid = keep; prev_f = f; prev_w = w;
    }
}

struct SubClass {  // subtrees launched together: same dynamic LDS size
    int nsub = 0, cap = 0, fmax = 0, bytes = 0, threads = 256;
    bool lds_walk = true, packed = false;
    DBuf<int> lo, hi;
};
struct SubSchedule {
    std::vector<SubClass> cls;
};
// one workgroup per small subtree: its supernodes lo..hi (a postorder range, children before parents) are factored one
// after the other by the same workgroup -- no launch and no inter-workgroup dependency inside the subtree
template <int RM>
__global__ __launch_bounds__(SUB_THREADS) void k_subtree_factor(FrontMeta M, double* __restrict__ fronts, const int* __restrict__ sub_lo, const int* __restrict__ sub_hi,
                                                        double* __restrict__ rdiag, int* __restrict__ info)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lo = sub_lo[blockIdx.x], hi = sub_hi[blockIdx.x];
    for (int s = lo; s <= hi; ++s) {
        front_factor<RM>(M, fronts, s, rdiag, info, lds);
        __syncthreads();
    }
}

// extend-add of ONE child into a front that is then factored by the dense kernels (one launch per child: stream
// order = fixed merge order, entries of one child never collide)
// own K entries of ONE front that the dense multi-workgroup path factors (the front was zeroed by a memset on the stream)
// ---- big fronts (big_front() above): the dense MFMA panel kernels factor them, ALL big fronts of a tree level per launch
// (blockIdx.y = front of the level's list).  Zero-fill and the front's own K entries need nothing from the children: one launch each for every
// big front of the tree at the start of the factorisation.  The children are merged by one launch per level: column c of a front belongs to
// workgroup c mod G of that front, which walks the children in order -- an entry receives its contributions in child order whatever the
// grid (fixed summation order), and no two workgroups touch the same entry.
__global__ __launch_bounds__(256) void k_big_zero(FrontMeta M, double* __restrict__ fronts, const int* __restrict__ list, int fuse)
{
    const SnRec me = M.sn[list[blockIdx.y]];
    if (fuse && me.child_hi > me.child_lo) return;  // written once by k_big_extend_add together with its first child (round 4)
    double* F = fronts + me.front_off;
    const int f = me.f;
    // the lower triangle only: nothing reads the strict upper triangle of a multi-workgroup front as a number (half the zero-fill traffic, which
    // was the largest single item of the factorisation's HBM traffic on CONT-201: profiles/r02_pmc_sparse_batch.json)
    for (int j = blockIdx.x; j < f; j += gridDim.x) {
        double* Fj = F + (long long)j * f;
        for (int i = j + threadIdx.x; i < f; i += 256) Fj[i] = 0.0;
    }
}
__global__ __launch_bounds__(256) void k_big_assemble(FrontMeta M, double* __restrict__ fronts, const int* __restrict__ list, int fuse)
{
    const SnRec me = M.sn[list[blockIdx.y]];
    if (fuse && me.child_hi > me.child_lo) return;
    double* F = fronts + me.front_off;
    for (int e = me.fe_lo + blockIdx.x * 256 + threadIdx.x; e < me.fe_hi; e += gridDim.x * 256) F[M.fe_off[e]] = M.vals[e];
}
// fuse (round 4): a front with children is not zero-filled in advance; this kernel WRITES every entry of its lower triangle once -- 0 + the first child's
// contribution where the child has one, 0 elsewhere (inverse row map of the first child in LDS) --, adds the front's own K entries to the columns it owns and
// goes on with the second child.  K + U1 and U1 + K are the same number, so the fronts are bitwise those of the zero-fill path; what goes away is the
// zero-fill's write and the read half of the first child's read-modify-write (CONT-201: 50 MB + ~40 MB of 512 MB per factorisation).
__global__ __launch_bounds__(256) void k_big_extend_add(FrontMeta M, double* __restrict__ fronts, const int* __restrict__ list, int fuse)
{
    constexpr int OWN_CAP = 2048;
    __shared__ int own[OWN_CAP];
    __shared__ int nown;
    extern __shared__ int inv[];  // fuse: front row -> row of the first child's update matrix (or -1)
    const SnRec me = M.sn[list[blockIdx.y]];
    const int f = me.f;
    double* F = fronts + me.front_off;
    const int G = gridDim.x, mine = blockIdx.x;
    int cfirst = me.child_lo;
    if (fuse && me.child_hi > me.child_lo) {
        const SnRec ch = M.sn[M.child[me.child_lo]];
        const int wc = ch.w, fc = ch.f, uc = fc - wc;
        const double* U = fronts + ch.front_off + wc + (long long)wc * fc;
        const int* rel = M.rel + ch.rel_ptr;
        for (int i = threadIdx.x; i < f; i += 256) inv[i] = -1;
        __syncthreads();
        for (int j = threadIdx.x; j < uc; j += 256) inv[rel[j]] = j;
        __syncthreads();
        const int cnt = mine < f ? (f - mine + G - 1) / G : 0;  // columns mine, mine + G, ...
        const int TX = cnt <= 4 ? 64 : 16, TY = 256 / TX;
        const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
        for (int q = ty; q < cnt; q += TY) {
            const int c = mine + q * G;
            const int jc = inv[c];
            double* __restrict__ Fc = F + (long long)c * f;
            const double* __restrict__ Uj = U + (long long)(jc >= 0 ? jc : 0) * fc;
            for (int i = c + tx; i < f; i += 4 * TX) {
                double uv[4];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int iq = i + TX * q4;
                    const int ic = (iq < f && jc >= 0) ? inv[iq] : -1;
                    uv[q4] = ic >= 0 ? Uj[ic] : 0.0;
                }
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) { const int iq = i + TX * q4; if (iq < f) Fc[iq] = 0.0 + uv[q4]; }
            }
        }
        __syncthreads();
        for (int e = me.fe_lo + threadIdx.x; e < me.fe_hi; e += 256) {
            const int off = M.fe_off[e];
            if ((off / f) % G == mine) F[off] += M.vals[e];
        }
        __syncthreads();
        cfirst = me.child_lo + 1;
    }
    for (int ci = cfirst; ci < me.child_hi; ++ci) {
        const SnRec ch = M.sn[M.child[ci]];
        const int wc = ch.w, fc = ch.f, uc = fc - wc;
        const double* U = fronts + ch.front_off + wc + (long long)wc * fc;
        const int* rel = M.rel + ch.rel_ptr;
        for (int j0 = 0; j0 < uc; j0 += OWN_CAP) {
            // the child's columns that land in this workgroup's columns of the front (column c of the front belongs to workgroup c mod G)
            if (threadIdx.x == 0) nown = 0;
            __syncthreads();
            for (int j = j0 + threadIdx.x; j < min(uc, j0 + OWN_CAP); j += 256)
                if (rel[j] % G == mine) own[atomicAdd(&nown, 1)] = j;
            __syncthreads();
            const int cnt = nown;
            // 16 rows x 16 columns of the child's update matrix per step, or 64 rows x 4 columns where this workgroup owns only a few columns (the top of the
            // tree: one to four fronts per level on a wider grid -- a column of 500 rows was eight dependent rounds of index, load, store on 16 lanes)
            const int TX = cnt <= 4 ? 64 : 16, TY = 256 / TX;
            const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
            for (int q = ty; q < cnt; q += TY) {
                const int j = own[q];
                const long long cj = (long long)rel[j] * f;
                const double* __restrict__ Uj = U + (long long)j * fc;
                for (int i = j + tx; i < uc; i += 4 * TX) {  // four entries per step, loads before stores (see extend_add)
                    double uv[4], fv[4];
                    long long at[4];
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int iq = i + TX * q4;
                        const bool ok = iq < uc;
                        at[q4] = ok ? rel[iq] + cj : -1;
                        uv[q4] = ok ? Uj[iq] : 0.0;
                    }
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) fv[q4] = at[q4] >= 0 ? F[at[q4]] : 0.0;
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) if (at[q4] >= 0) F[at[q4]] = fv[q4] + uv[q4];
                }
            }
            __syncthreads();  // the next child (or chunk) may add to the same entries: this workgroup's stores first
        }
    }
}

// forward substitution on one front: v = [x(pivots) + children; children], y = L11^-1 v1, v2 -= L21 y
__device__ void front_fwd(const FrontMeta& M, const double* __restrict__ fronts, int s, double* __restrict__ x, double* __restrict__ fvec)
{
    const SnRec me = M.sn[s];
    const int first = me.first, w = me.w, f = me.f;
    const double* F = fronts + me.front_off;
    double* v = fvec + me.rows_ptr;
    for (int i = threadIdx.x; i < f; i += blockDim.x) v[i] = (i < w) ? x[first + i] : 0.0;
    __syncthreads();
    for (int ci = me.child_lo; ci < me.child_hi; ++ci) {
        const SnRec ch = M.sn[M.child[ci]];
        const int wc = ch.w, fc = ch.f;
        const double* vc = fvec + ch.rows_ptr + wc;
        const int* rel = M.rel + ch.rel_ptr;
        for (int i = threadIdx.x; i < fc - wc; i += blockDim.x) v[rel[i]] += vc[i];
        __syncthreads();
    }
    for (int k = 0; k < w; ++k) {
        const double yk = v[k];
        const double* col = F + (long long)k * f;
        for (int i = k + 1 + threadIdx.x; i < f; i += blockDim.x) v[i] -= col[i] * yk;
        __syncthreads();
    }
    for (int i = threadIdx.x; i < w; i += blockDim.x) x[first + i] = v[i];
}
__global__ __launch_bounds__(SUB_SOLVE_THREADS) void k_subtree_fwd(FrontMeta M, const double* __restrict__ fronts, const int* __restrict__ sub_lo, const int* __restrict__ sub_hi,
                                                     double* __restrict__ x, double* __restrict__ fvec)
{
    const int lo = sub_lo[blockIdx.x], hi = sub_hi[blockIdx.x];
    for (int s = lo; s <= hi; ++s) { front_fwd(M, fronts, s, x, fvec); __syncthreads(); }
}

// backward substitution on one front: x1 = L11^-T (y1 - L21^T x2), x2 gathered from the already-final ancestors
__device__ void front_bwd(const FrontMeta& M, const double* __restrict__ fronts, int s, double* __restrict__ x, double* __restrict__ fvec)
{
    const SnRec me = M.sn[s];
    const int first = me.first, w = me.w, f = me.f;
    const double* F = fronts + me.front_off;
    const int* rows = M.front_rows + me.rows_ptr;
    double* v = fvec + me.rows_ptr;
    for (int i = threadIdx.x; i < f; i += blockDim.x) v[i] = x[rows[i]];
    __syncthreads();
    // y1[k] -= sum_{i >= w} L[i,k] * x2[i]   (thread per pivot column, contiguous reads down the column)
    for (int k = threadIdx.x; k < w; k += blockDim.x) {
        const double* col = F + (long long)k * f;
        double sacc = 0.0;
        for (int i = w; i < f; ++i) sacc += col[i] * v[i];
        v[k] -= sacc;
    }
    __syncthreads();
    for (int i = w - 1; i > 0; --i) {
        const double xi = v[i];
        for (int k = threadIdx.x; k < i; k += blockDim.x) v[k] -= F[i + (long long)k * f] * xi;
        __syncthreads();
    }
    for (int i = threadIdx.x; i < w; i += blockDim.x) x[first + i] = v[i];
}
__global__ __launch_bounds__(SUB_SOLVE_THREADS) void k_subtree_bwd(FrontMeta M, const double* __restrict__ fronts, const int* __restrict__ sub_lo, const int* __restrict__ sub_hi,
                                                     double* __restrict__ x, double* __restrict__ fvec)
{
    const int lo = sub_lo[blockIdx.x], hi = sub_hi[blockIdx.x];
    for (int s = hi; s >= lo; --s) { front_bwd(M, fronts, s, x, fvec); __syncthreads(); }
}

// ---- wide fronts (more than 128 rows) of the level-scheduled top: blocked substitution, the front vector in LDS.
// front_fwd / front_bwd above pay one global-memory round trip and one barrier PER PIVOT (CONT-201's 663-row root: 382 us forward, 279 us
// backward at 5 GB/s).  Here 16 pivots go at a time: their 16 x 16 triangle is solved by one wave in registers (v_readlane broadcasts), the
// rest of the block is a rank-16 update (forward: rows over the threads, coalesced column reads) or 16 column dot products (backward:
// left-looking, two columns per wave, lanes along the contiguous column, fixed-order wave reduction); two barriers per block, and the operands
// of block b + 1 are loaded into a second register buffer BEFORE block b is worked on, so that no step waits for memory (a 32-pivot, single-
// buffered version of the same kernels waited one memory latency per block: 5 us per block on the 663-row root).  Which routine a front
// takes depends on the front alone (f > 128), never on the schedule, so every schedule variant still produces the same bits; the forward
// sweep also keeps the per-entry operation order of front_fwd.
constexpr int WIDE_NT = 512, WIDE_B = 16;
constexpr int WIDE_FCAP = 7000;  // rows of a front this path keeps in LDS (56 KB); wider ones take front_fwd / front_bwd
// HUGE fronts (round 4; the C3 variant with 1500-variable windows has a chain of ~150 fronts of 3000-4000 rows and 70-470 pivots): one workgroup streams the
// f x w panel of such a front at ~15 GB/s, 540 us per front and sweep.  Their rows below the pivot block are therefore taken off the front's workgroup:
// forward, the front's workgroup solves the w x w pivot block only and k_front_fwd_rows (one wave per 64 rows, any number of workgroups) applies the
// pivots to the update rows; backward, k_front_bwd_cols forms the column sums over the update rows in chunks of HUGE_ROWS rows (partial sums, added in
// chunk order by the front's workgroup) before the pivot block is solved.  The forward arithmetic is the one-workgroup kernel's, entry by entry.
constexpr int HUGE_F = 1024, HUGE_ROWS = 256, HUGE_COLS = 64;
__host__ __device__ inline bool huge_front(int f, int w) { return f >= HUGE_F && w > 0 && f - w >= HUGE_ROWS; }
constexpr int WIDE_BCH = 12;     // backward: 64-row chunks of a column held in registers (768 rows below the block; further rows are loaded in line)

__device__ __forceinline__ double wide_bcast(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// workgroup barrier that waits for LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would wait for the operands just requested for the NEXT
// block -- the whole memory latency, every block.  Everything the waves exchange inside the block loops goes through LDS.
__device__ __forceinline__ void wide_lds_barrier()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0); vmcnt / expcnt untouched
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// DPP exchange of a double (both halves)
template <int CTRL>
__device__ __forceinline__ double wide_dpp(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// The xor butterfly t += t[lane ^ o], o = 32, 16, 8, 4, 2, 1 -- the summation tree every parity test was pinned with (another tree moves the
// threshold-sitting iteration counts of CONT-201 and nl_perold) --, its four inner steps as DPP exchanges with the SAME partners instead of trips through
// the LDS crossbar: lane ^ 8 = row_ror:8, lane ^ 4 = row_shl:4 on the even banks + row_shr:4 on the odd ones, lane ^ 2 / ^ 1 = quad permutes.
__device__ __forceinline__ double wide_xor4(double v)
{
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x104, 0xf, 0x5, false);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x104, 0xf, 0x5, false);
    lo = __builtin_amdgcn_update_dpp(lo, __double2loint(v), 0x114, 0xf, 0xa, false);
    hi = __builtin_amdgcn_update_dpp(hi, __double2hiint(v), 0x114, 0xf, 0xa, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wide_wave_sum(double v)
{
    v += __shfl_xor(v, 32);
    v += __shfl_xor(v, 16);
    v += wide_dpp<0x128>(v);  // row_ror:8
    v += wide_xor4(v);
    v += wide_dpp<0x4E>(v);   // quad_perm [2, 3, 0, 1]
    v += wide_dpp<0xB1>(v);   // quad_perm [1, 0, 3, 2]
    return v;
}

struct WideFwdOps { double Lt[WIDE_B], Lr0[WIDE_B], Lr1[WIDE_B]; };  // triangle row (wave 0, lane = row), two rows below the block per thread

__device__ __forceinline__ void front_fwd_wide_body(const FrontMeta& M, const double* __restrict__ fronts, const int* __restrict__ list, double* __restrict__ x,
                                                    double* __restrict__ fvec, int fcap, const int bid, const int* __restrict__ hoff)
{
    extern __shared__ __attribute__((aligned(16))) double vs[];
    const int s = list[bid];
    const SnRec me = M.sn[s];
    const int first = me.first, w = me.w, f = me.f;
    if (f > fcap) { front_fwd(M, fronts, s, x, fvec); return; }
    const double* __restrict__ F = fronts + me.front_off;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // rows this workgroup applies the pivots to: a huge front's pivot block only WHEN the caller runs k_front_fwd_rows behind this launch -- which it says by passing
    // the schedule's offsets of the huge fronts, as for the backward kernels (round-4 advice: without them the whole front is eliminated here, slower and correct)
    const int flim = (hoff != nullptr && huge_front(f, w)) ? w : f;
    for (int i = tid; i < f; i += WIDE_NT) vs[i] = (i < w) ? x[first + i] : 0.0;
    __syncthreads();
    for (int ci = me.child_lo; ci < me.child_hi; ++ci) {
        const SnRec ch = M.sn[M.child[ci]];
        const int wc = ch.w, fc = ch.f;
        const double* vc = fvec + ch.rows_ptr + wc;
        const int* rel = M.rel + ch.rel_ptr;
        for (int i = tid; i < fc - wc; i += WIDE_NT) vs[rel[i]] += vc[i];
        __syncthreads();
    }
    // FULL: a whole block of WIDE_B pivots (every block but possibly the last): nbk is a compile-time constant there, which removes the clamps, the
    // selects and most of the scalar address arithmetic (the generic instantiation spent ~400 scalar instructions per block on them)
    auto prefetch_tri = [&](auto full_tag, WideFwdOps& o, int kb) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int nbk = FULL ? WIDE_B : min(WIDE_B, w - kb);
        if (wave == 0) {
            // (a load under a per-lane condition compiles to a branch of its own -- 48 of them per block --, and a select on the loaded value waits for
            //  it, which turns the prefetch into a synchronous load: here every load is unconditional from an address clamped INTO the block, i.e. into cache
            //  lines the neighbours fetch anyway, and the masks are applied where the values are USED)
            const int lc = min(lane, nbk - 1);
            const double* pt = F + (kb + lc) + (long long)kb * f;  // column pointers advance by f and stop at the block's last column: two scalar adds per load
#pragma unroll
            for (int k = 0; k < WIDE_B; ++k) { o.Lt[k] = *pt; pt += (FULL || k + 1 < nbk) ? f : 0; }  // masked where it is used (step)
        }
    };
    auto prefetch_rows = [&](auto full_tag, WideFwdOps& o, int kb) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int nbk = FULL ? WIDE_B : min(WIDE_B, w - kb);
        const int i0 = kb + nbk + tid, i1 = i0 + WIDE_NT;
        const int wbase0 = i0 - lane, wbase1 = i1 - lane;  // first row of this wave's 64 rows: a wave entirely below the front loads nothing (uniform branch)
        if (wbase0 < flim) {
            const bool ok0 = i0 < flim;
            const double* p0 = F + (ok0 ? i0 : flim - 1) + (long long)kb * f;
#pragma unroll
            for (int k = 0; k < WIDE_B; ++k) { o.Lr0[k] = *p0; p0 += (FULL || k + 1 < nbk) ? f : 0; }  // rows below the front are never stored, columns beyond nbk meet y = 0
        } else {
#pragma unroll
            for (int k = 0; k < WIDE_B; ++k) o.Lr0[k] = 0.0;
        }
        if (wbase1 < flim) {
            const bool ok1 = i1 < flim;
            const double* p1 = F + (ok1 ? i1 : flim - 1) + (long long)kb * f;
#pragma unroll
            for (int k = 0; k < WIDE_B; ++k) { o.Lr1[k] = *p1; p1 += (FULL || k + 1 < nbk) ? f : 0; }
        } else {
#pragma unroll
            for (int k = 0; k < WIDE_B; ++k) o.Lr1[k] = 0.0;
        }
    };
    // Two register buffers, each refilled RIGHT AFTER its use with the operands of the block two ahead (kb2): a request then has a whole block of the
    // other buffer to arrive in -- requested one block ahead at the top of the block, it had ~0.6 us of work to hide ~2 us of memory latency behind
    auto step_t = [&](auto full_tag, WideFwdOps& o, int kb, int kb2) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int nbk = FULL ? WIDE_B : min(WIDE_B, w - kb);
        if (wave == 0) {
            double vv = lane < nbk ? vs[kb + lane] : 0.0;
#pragma unroll
            for (int k = 0; k < WIDE_B; ++k) {
                const double yk = wide_bcast(vv, k);
                const double l = (k < lane && lane < nbk) ? o.Lt[k] : 0.0;
                vv = __builtin_fma(-l, yk, vv);
            }
            if (lane < nbk) vs[kb + lane] = vv;
        }
        if (kb2 < w) prefetch_tri(full_tag, o, kb2);
        wide_lds_barrier();
        double y[WIDE_B];
#pragma unroll
        for (int k = 0; k < WIDE_B; ++k) { const double t = vs[kb + min(k, nbk - 1)]; y[k] = k < nbk ? t : 0.0; }
        const int i0 = kb + nbk + tid, i1 = i0 + WIDE_NT;
        if (i0 < flim) {
            double vi = vs[i0];
#pragma unroll
            for (int k = 0; k < WIDE_B; ++k) vi = __builtin_fma(-o.Lr0[k], y[k], vi);
            vs[i0] = vi;
        }
        if (i1 < flim) {
            double vi = vs[i1];
#pragma unroll
            for (int k = 0; k < WIDE_B; ++k) vi = __builtin_fma(-o.Lr1[k], y[k], vi);
            vs[i1] = vi;
        }
        for (int i2 = i1 + WIDE_NT; i2 < flim; i2 += WIDE_NT) {
            double vi = vs[i2];
            for (int k = 0; k < nbk; ++k) vi = __builtin_fma(-F[i2 + (long long)(kb + k) * f], y[k], vi);
            vs[i2] = vi;
        }
        if (kb2 < w) prefetch_rows(full_tag, o, kb2);
        wide_lds_barrier();
    };
    {   // all blocks through the generic instantiation, double-buffered (forward: a compile-time-full instantiation for the whole blocks, as in the backward
        // kernel, measured slower -- 1.24 -> 1.53 ms over CONT-201's wide fronts, a few spills)
        const std::false_type gen{};
        WideFwdOps A, B;
        if (w > 0) { prefetch_tri(gen, A, 0); prefetch_rows(gen, A, 0); }
        if (w > WIDE_B) { prefetch_tri(gen, B, WIDE_B); prefetch_rows(gen, B, WIDE_B); }
        for (int kb = 0; kb < w; kb += 2 * WIDE_B) {
            step_t(gen, A, kb, kb + 2 * WIDE_B);
            if (kb + WIDE_B >= w) break;
            step_t(gen, B, kb + WIDE_B, kb + 3 * WIDE_B);
        }
    }
    double* v = fvec + me.rows_ptr;
    for (int i = tid; i < f; i += WIDE_NT) {
        const double t = vs[i];
        if (i < w) x[first + i] = t;
        v[i] = t;
    }
}
__global__ __launch_bounds__(WIDE_NT) void k_front_fwd_wide(FrontMeta M, const double* __restrict__ fronts, const int* __restrict__ list, double* __restrict__ x,
                                                            double* __restrict__ fvec, int fcap, const int* __restrict__ hoff)
{
    front_fwd_wide_body(M, fronts, list, x, fvec, fcap, (int)blockIdx.x, hoff);
}

// update rows of the huge fronts of a level: v[r] -= sum_k L[r, k] y[k], k ascending in one fma chain (the order of the one-workgroup kernel); one wave per 64 rows
__global__ __launch_bounds__(64) void k_front_fwd_rows(FrontMeta M, const double* __restrict__ fronts, const int* __restrict__ list, double* __restrict__ fvec)
{
    extern __shared__ __attribute__((aligned(16))) double ys[];
    const SnRec me = M.sn[list[blockIdx.y]];
    const int w = me.w, f = me.f;
    const int r0 = w + (int)blockIdx.x * 64;
    if (r0 >= f) return;
    double* __restrict__ v = fvec + me.rows_ptr;
    const double* __restrict__ F = fronts + me.front_off;
    for (int k = threadIdx.x; k < w; k += 64) ys[k] = v[k];
    __syncthreads();
    const int r = r0 + (int)threadIdx.x;
    const double* __restrict__ pr = F + min(r, f - 1);
    double vi = r < f ? v[r] : 0.0;
    int k = 0;
    for (; k + 16 <= w; k += 16) {  // sixteen loads in flight per lane (one wave per workgroup: the latency of a trip is all there is to hide)
        double l[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) l[q] = pr[(long long)(k + q) * f];
#pragma unroll
        for (int q = 0; q < 16; ++q) vi = __builtin_fma(-l[q], ys[k + q], vi);
    }
    for (; k < w; ++k) vi = __builtin_fma(-pr[(long long)k * f], ys[k], vi);
    if (r < f) v[r] = vi;
}
// column sums of the huge fronts of a level over their update rows, HUGE_ROWS rows per workgroup: part[hoff[s] + chunk * w + k] = sum_{i in chunk} L[i, k] x[rows[i]]
__global__ __launch_bounds__(HUGE_ROWS) void k_front_bwd_cols(FrontMeta M, const double* __restrict__ fronts, const int* __restrict__ list, const double* __restrict__ x,
                                                              const int* __restrict__ hoff, double* __restrict__ part)
{
    __shared__ double red[HUGE_ROWS / 64][16];
    const int s = list[blockIdx.y];
    const SnRec me = M.sn[s];
    const int w = me.w, f = me.f;
    const int r0 = w + (int)blockIdx.x * HUGE_ROWS;
    if (r0 >= f) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = r0 + tid;
    const double xi = i < f ? x[M.front_rows[me.rows_ptr + i]] : 0.0;  // rows below the front meet x = 0
    const double* __restrict__ pr = fronts + me.front_off + min(i, f - 1);
    double* __restrict__ out = part + hoff[s] + (long long)blockIdx.x * w;
    // blockIdx.z: groups of HUGE_COLS columns (every workgroup writes its own entries of `out`)
    const int kend = min(w, ((int)blockIdx.z + 1) * HUGE_COLS);
    for (int k = (int)blockIdx.z * HUGE_COLS; k < kend; k += 16) {
        double a[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = pr[(long long)min(k + q, w - 1) * f] * xi;
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = wide_wave_sum(a[q]);
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) red[wave][q] = a[q];
        }
        __syncthreads();
        if (tid < 16 && k + tid < w) {
            double t = red[0][tid];
#pragma unroll
            for (int u = 1; u < HUGE_ROWS / 64; ++u) t += red[u][tid];
            out[k + tid] = t;
        }
        __syncthreads();
    }
}

constexpr int WIDE_CPW = WIDE_B / (WIDE_NT / 64);  // backward: columns per wave
struct WideBwdOps { double Lt[WIDE_B], col[WIDE_CPW][WIDE_BCH]; };  // triangle column (wave 0, lane = column), this wave's columns below the block

__device__ __forceinline__ void front_bwd_wide_body(const FrontMeta& M, const double* __restrict__ fronts, const int* __restrict__ list, double* __restrict__ x,
                                                    double* __restrict__ fvec, int fcap, const int bid, const int* __restrict__ hoff, const double* __restrict__ hpart)
{
    extern __shared__ __attribute__((aligned(16))) double vs[];
    const int s = list[bid];
    const SnRec me = M.sn[s];
    const int first = me.first, w = me.w;
    if (me.f > fcap) { front_bwd(M, fronts, s, x, fvec); return; }
    const double* __restrict__ F = fronts + me.front_off;
    const int* __restrict__ rows = M.front_rows + me.rows_ptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool huge = huge_front(me.f, w) && hoff != nullptr;
    const long long ldf = me.f;       // column stride of the front
    const int f = huge ? w : me.f;    // rows this workgroup sums over (a huge front: the pivot block only, the update rows arrive as partial sums of k_front_bwd_cols)
    double* ss = vs + ((f + 1) & ~1);  // column sums of the current block
    for (int i = tid; i < f; i += WIDE_NT) {
        double t = x[rows[i]];
        if (huge) {
            const double* pp = hpart + hoff[s] + i;
            const int nch = (me.f - w + HUGE_ROWS - 1) / HUGE_ROWS;
            double ps = pp[0];
            for (int c = 1; c < nch; ++c) ps += pp[(long long)c * w];
            t -= ps;
        }
        vs[i] = t;
    }
    __syncthreads();
    auto prefetch_tri = [&](auto full_tag, WideBwdOps& o, int kb) {
        constexpr bool FULL = decltype(full_tag)::value;  // (see k_front_fwd_wide)
        const int nbk = FULL ? WIDE_B : min(WIDE_B, w - kb);
        if (wave == 0) {
            const double* col = F + kb + (long long)(kb + min(lane, nbk - 1)) * ldf;
#pragma unroll
            for (int i = 0; i < WIDE_B; ++i) o.Lt[i] = col[min(i, nbk - 1)];  // (clamped into the block; masked where it is used)
        }
    };
    auto prefetch_cols = [&](auto full_tag, WideBwdOps& o, int kb) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int nbk = FULL ? WIDE_B : min(WIDE_B, w - kb), r0 = kb + nbk;
#pragma unroll
        for (int c = 0; c < WIDE_CPW; ++c) {
            const int k = wave * WIDE_CPW + c;  // wave-uniform
            const double* col = F + (long long)(kb + min(k, nbk - 1)) * ldf;
#pragma unroll
            for (int ch = 0; ch < WIDE_BCH; ++ch) {
                const int i = r0 + lane + 64 * ch;
                if (k < nbk && r0 + 64 * ch < f) o.col[c][ch] = col[min(i, f - 1)];  // uniform condition; rows below the front meet x = 0 in the dot product
                else o.col[c][ch] = 0.0;
            }
        }
    };
    // (as in the forward kernel: a buffer is refilled right after its use with the operands of the block two further on, kb2 < 0: none)
    auto step_t = [&](auto full_tag, WideBwdOps& o, int kb, int kb2) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int nbk = FULL ? WIDE_B : min(WIDE_B, w - kb), r0 = kb + nbk;
        // s_k = sum_{i >= r0} L[i, k] x[i]
        double acc[WIDE_CPW];
#pragma unroll
        for (int c = 0; c < WIDE_CPW; ++c) acc[c] = 0.0;
#pragma unroll
        for (int ch = 0; ch < WIDE_BCH; ++ch) {
            const int i = r0 + lane + 64 * ch;
            const double xi = i < f ? vs[i] : 0.0;
#pragma unroll
            for (int c = 0; c < WIDE_CPW; ++c) acc[c] = __builtin_fma(o.col[c][ch], xi, acc[c]);
        }
        for (int i = r0 + lane + 64 * WIDE_BCH; i < f; i += 64) {
            const double xi = vs[i];
#pragma unroll
            for (int c = 0; c < WIDE_CPW; ++c) {
                const int k = wave * WIDE_CPW + c;
                if (k < nbk) acc[c] = __builtin_fma(F[i + (long long)(kb + k) * ldf], xi, acc[c]);
            }
        }
        if (kb2 >= 0) prefetch_cols(std::true_type{}, o, kb2);  // (every block after the first one processed is a full one)
#pragma unroll
        for (int c = 0; c < WIDE_CPW; ++c) {
            const double t = wide_wave_sum(acc[c]);
            if (lane == 0) ss[wave * WIDE_CPW + c] = t;
        }
        wide_lds_barrier();
        if (wave == 0) {
            double vk = lane < nbk ? vs[kb + lane] - ss[lane] : 0.0;
#pragma unroll
            for (int i = WIDE_B - 1; i >= 1; --i) {
                const double xi = wide_bcast(vk, i);
                const double l = (lane < i && i < nbk) ? o.Lt[i] : 0.0;
                vk = __builtin_fma(-l, xi, vk);
            }
            if (lane < nbk) vs[kb + lane] = vk;
        }
        if (kb2 >= 0) prefetch_tri(std::true_type{}, o, kb2);
        wide_lds_barrier();
    };
    {   // the ragged last block first, on its own; then the full blocks on two buffers
        const int wfull = w - w % WIDE_B;
        if (wfull < w) {
            WideBwdOps T;
            prefetch_tri(std::false_type{}, T, wfull); prefetch_cols(std::false_type{}, T, wfull);
            step_t(std::false_type{}, T, wfull, -1);
        }
        const std::true_type full{};
        WideBwdOps A, B;
        const int klast = wfull - WIDE_B;
        if (klast >= 0) { prefetch_tri(full, A, klast); prefetch_cols(full, A, klast); }
        if (klast - WIDE_B >= 0) { prefetch_tri(full, B, klast - WIDE_B); prefetch_cols(full, B, klast - WIDE_B); }
        for (int kb = klast; kb >= 0; kb -= 2 * WIDE_B) {
            step_t(full, A, kb, kb - 2 * WIDE_B);
            if (kb - WIDE_B < 0) break;
            step_t(full, B, kb - WIDE_B, kb - 3 * WIDE_B);
        }
    }
    for (int i = tid; i < w; i += WIDE_NT) x[first + i] = vs[i];
}
__global__ __launch_bounds__(WIDE_NT) void k_front_bwd_wide(FrontMeta M, const double* __restrict__ fronts, const int* __restrict__ list, double* __restrict__ x,
                                                            double* __restrict__ fvec, int fcap, const int* __restrict__ hoff, const double* __restrict__ hpart)
{
    front_bwd_wide_body(M, fronts, list, x, fvec, fcap, (int)blockIdx.x, hoff, hpart);
}

// ---- single-wave subtree substitution: the front vector lives in registers (rows lane and lane + 64, fronts of a workgroup subtree have at
// most SUB_FMAX <= 128 rows), pivots are broadcast with v_readlane, the panel columns are read straight from HBM with the loads of four
// steps in flight, and a parent that directly follows its child in the walk takes the child's update vector from LDS.  Same arithmetic,
// same order as front_fwd / front_bwd (bitwise the same results): what changes is that a dependent step costs an FMA + a readlane
// instead of a global-memory round trip + a barrier.
// agent-scope (L1-bypassing, write-through) accesses for values handed from one workgroup to another inside ONE launch
template <bool AGENT>
__device__ __forceinline__ double ldx(const double* p)
{
    if (!AGENT) return *p;
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
template <bool AGENT>
__device__ __forceinline__ void stx(double* p, double v)
{
    if (!AGENT) { *p = v; return; }
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// single-wave workgroup: wait until *flag != 0 (bounded; a timeout raises err and lets the launch drain)
// debug: per-workgroup start / end / wait-done timestamps of the flag-ordered sweeps (PIQP_AMD_DEBUG=dbg_ts=<file prefix>)
__device__ long long* g_dbg_ts = nullptr;
__device__ __forceinline__ void wave_wait_flag(const int* flag, int* err, int epoch)
{
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
            __builtin_amdgcn_s_sleep(1);  // (an exponential poll back-off was measured neutral and removed)
            if (++spins > 8000000u || ((spins & 15) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch)) {
                __hip_atomic_store(err, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}
__device__ __forceinline__ void wave_publish_flag(int* flag, int epoch)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double bcast_row(double v0, double v1, int row)
{
    const int r = __builtin_amdgcn_readfirstlane(row);
    const double v = r < 64 ? v0 : v1;
    const int src = r & 63;
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
// TOP = true: the level-sorted top of the tree in ONE launch, one single-wave workgroup per supernode (sub_lo = the list, in dependency
// order): a workgroup waits for the flags of its children, takes their update vectors with agent-scope loads (they were stored write-through
// by another CU a microsecond earlier: no fence needed on either side) and publishes its own flag.  Dependencies always have a smaller
// block index, so they were dispatched earlier: no deadlock whatever the residency.
// Pulls the factor panel of a front towards this CU while the wave still waits for its neighbours in the tree: one load per 128-byte
// line (up to four per lane = 32 KB), summed and consumed by an empty asm only after the wait.
struct PanelTouch { double t[4]; };
__device__ __forceinline__ PanelTouch touch_panel(const double* __restrict__ F, int n, int lane)
{
    PanelTouch p;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = (lane + 64 * u) * 16;
        p.t[u] = i < n ? F[i] : 0.0;
    }
    return p;
}
__device__ __forceinline__ void touch_done(const PanelTouch& p)
{
    const double acc = (p.t[0] + p.t[1]) + (p.t[2] + p.t[3]);
    asm volatile("" ::"v"(acc));
}
template <bool TOP>
__device__ __forceinline__ void subtree_fwd_wave_body(const FrontMeta& M, const double* __restrict__ fronts, const int* __restrict__ sub_lo, const int* __restrict__ sub_hi,
                                                      double* __restrict__ x, double* __restrict__ fvec, const int* __restrict__ top_pos, int* __restrict__ flags,
                                                      int* __restrict__ err, const int* __restrict__ child_tp, int epoch, const int bid)
{
    __shared__ double sv[2][128];
    const int lane = threadIdx.x, r0 = lane, r1 = lane + 64;
    // TOP: the walk is a chain of top supernodes (parent[t] == t + 1); children outside the walk are waited for by flag
    const int lo = sub_lo[bid], hi = sub_hi[bid];
    long long* const dbg = TOP ? g_dbg_ts : nullptr;
    if (dbg && lane == 0) dbg[3 * bid] = wall_clock64();
    int cur = 0;
    bool prev_valid = false;  // sv[cur ^ 1][0 .. u) = update vector of supernode s - 1
    SnRec me = M.sn[lo];
    for (int s = lo; s <= hi; ++s) {
        const SnRec nxt = M.sn[s < hi ? s + 1 : s];  // the next link's record travels while this one is processed
        const int first = me.first, w = me.w, f = me.f;
        const double* F = fronts + me.front_off;
        double* a = sv[cur];
        int ci0 = me.child_lo;  // children from here on go through the generic gather below
        if (TOP) {
            // everything that does not depend on the children is loaded BEFORE the wave waits for their flags: the panel (touched), the
            // right-hand side, and for the first four children the record, the scatter indices and the flag position -- after the wait
            // only the update vectors themselves are one round trip away, all children in flight together
            constexpr int MAXC = 4;
            const PanelTouch pt = touch_panel(F, f * w, lane);
            const double xa0 = r0 < w ? x[first + r0] : 0.0, xa1 = r1 < w ? x[first + r1] : 0.0;
            const int nch = me.child_hi - me.child_lo;
            const double* cvc[MAXC];
            int cr0[MAXC], cr1[MAXC], ctp[MAXC];
#pragma unroll
            for (int q = 0; q < MAXC; ++q) {
                cr0[q] = -1; cr1[q] = -1; ctp[q] = -1; cvc[q] = fvec;
                if (q < nch) {
                    const ChildRec cr = M.crec[me.child_lo + q];
                    const int tpq = child_tp[me.child_lo + q];
                    const bool inwalk = prev_valid && cr.c == s - 1;
                    const int* rel = M.rel + cr.rel_ptr;
                    cvc[q] = fvec + cr.vc_off;
                    if (r0 < cr.uc) cr0[q] = rel[r0];
                    if (r1 < cr.uc) cr1[q] = rel[r1];
                    ctp[q] = inwalk ? -2 : tpq;
                }
            }
#pragma unroll
            for (int q = 0; q < MAXC; ++q) if (ctp[q] >= 0) wave_wait_flag(flags + ctp[q], err, epoch);
            for (int ci = me.child_lo + MAXC; ci < me.child_hi; ++ci) {
                if (prev_valid && M.child[ci] == s - 1) continue;  // the previous link of this walk: no flag, its vector is in LDS
                const int tp = child_tp[ci];
                if (tp >= 0) wave_wait_flag(flags + tp, err, epoch);
            }
            touch_done(pt);
            if (dbg && lane == 0 && s == lo) dbg[3 * bid + 1] = wall_clock64();
            if (r0 < f) a[r0] = xa0;
            if (r1 < f) a[r1] = xa1;
            __syncthreads();
            double g0[MAXC], g1[MAXC];
#pragma unroll
            for (int q = 0; q < MAXC; ++q) {
                g0[q] = 0.0; g1[q] = 0.0;
                if (cr0[q] >= 0) g0[q] = ctp[q] == -2 ? sv[cur ^ 1][r0] : ldx<true>(cvc[q] + r0);
                if (cr1[q] >= 0) g1[q] = ctp[q] == -2 ? sv[cur ^ 1][r1] : ldx<true>(cvc[q] + r1);
            }
#pragma unroll
            for (int q = 0; q < MAXC; ++q) {
                if (q < nch) {  // uniform
                    if (cr0[q] >= 0) a[cr0[q]] += g0[q];
                    if (cr1[q] >= 0) a[cr1[q]] += g1[q];
                    __syncthreads();
                }
            }
            ci0 = me.child_lo + MAXC;
        } else {
            if (r0 < f) a[r0] = r0 < w ? x[first + r0] : 0.0;
            if (r1 < f) a[r1] = r1 < w ? x[first + r1] : 0.0;
            __syncthreads();
        }
        for (int ci = ci0; ci < me.child_hi; ++ci) {
            const ChildRec cr = M.crec[ci];
            const int uc = cr.uc;
            const double* vc = (prev_valid && cr.c == s - 1) ? sv[cur ^ 1] : fvec + cr.vc_off;
            const int* rel = M.rel + cr.rel_ptr;
            for (int i = lane; i < uc; i += 64) a[rel[i]] += (TOP ? ldx<true>(vc + i) : vc[i]);
            __syncthreads();
        }
        double v0 = r0 < f ? a[r0] : 0.0, v1 = r1 < f ? a[r1] : 0.0;
        auto panel_sweep = [&](const double* __restrict__ Fp) {
            int k = 0;
            const int rc0 = min(r0, f - 1), rc1 = min(r1, f - 1);
            for (; k + 4 <= w; k += 4) {
                double c0[4], c1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {  // unconditional loads from rows clamped into the front (a load under a lane condition is a branch of its own);
                    const double* col = Fp + (k + u) * f;  // the masks are applied where the values are used (C3 / C5 backend solve -4 %; the same change in
                    c0[u] = col[rc0];                      // the backward kernel measured neutral and was not kept)
                    c1[u] = col[rc1];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double yk = bcast_row(v0, v1, k + u);
                    v0 -= ((r0 > k + u && r0 < f) ? c0[u] : 0.0) * yk;
                    if (r1 > k + u) v1 -= (r1 < f ? c1[u] : 0.0) * yk;
                }
            }
            for (; k < w; ++k) {
                const double* col = Fp + k * f;
                const double c0 = (r0 > k && r0 < f) ? col[r0] : 0.0, c1 = (r1 > k && r1 < f) ? col[r1] : 0.0;
                const double yk = bcast_row(v0, v1, k);
                v0 -= c0 * yk;
                v1 -= c1 * yk;
            }
        };
        panel_sweep(F);
        if (r0 < w) x[first + r0] = v0;
        if (r1 < w) x[first + r1] = v1;
        __syncthreads();  // every lane has taken its entries of `a`
        const bool keep = me.parent == s + 1 && s + 1 <= hi;
        if (r0 >= w && r0 < f) { a[r0 - w] = v0; if (!keep) stx<TOP>(fvec + me.rows_ptr + r0, v0); }
        if (r1 >= w && r1 < f) { a[r1 - w] = v1; if (!keep) stx<TOP>(fvec + me.rows_ptr + r1, v1); }
        __syncthreads();
        cur ^= 1;
        prev_valid = keep;
        me = nxt;
    }
    if (TOP) wave_publish_flag(flags + top_pos[hi], epoch);
    if (dbg && lane == 0) dbg[3 * bid + 2] = wall_clock64();
}
template <bool TOP>
__global__ __launch_bounds__(64) void k_subtree_fwd_wave(FrontMeta M, const double* __restrict__ fronts, const int* __restrict__ sub_lo, const int* __restrict__ sub_hi,
                                                         double* __restrict__ x, double* __restrict__ fvec, const int* __restrict__ top_pos, int* __restrict__ flags,
                                                         int* __restrict__ err, const int* __restrict__ child_tp, int epoch)
{
    subtree_fwd_wave_body<TOP>(M, fronts, sub_lo, sub_hi, x, fvec, top_pos, flags, err, child_tp, epoch, (int)blockIdx.x);
}
// TOP = true: as above for the backward sweep; sub_lo = the level-sorted list, block b takes entry ntop - 1 - b (parents first) and waits
// for its parent's flag; the ancestors' solution entries are read with agent-scope loads, its own are stored write-through
template <bool TOP>
__device__ __forceinline__ void subtree_bwd_wave_body(const FrontMeta& M, const double* __restrict__ fronts, const int* __restrict__ sub_lo, const int* __restrict__ sub_hi,
                                                      double* __restrict__ x, int red_thr, int ntop, const int* __restrict__ top_pos, int* __restrict__ flags,
                                                      int* __restrict__ err, const int* __restrict__ pub, int epoch, const double* __restrict__ rdiag, const int bid)
{
    __shared__ double sv[2][128];
    const int lane = threadIdx.x, r0 = lane, r1 = lane + 64;
    // TOP: walks in reverse order (ntop = number of walks); the parent of the walk's last supernode is waited for by flag, every
    // supernode of the walk publishes its own flag (children outside the walk hang off any of them)
    const int wi = TOP ? ntop - 1 - bid : bid;
    const int lo = sub_lo[wi], hi = sub_hi[wi];
    if (TOP) {
        const SnRec me = M.sn[hi];
        const PanelTouch pt = touch_panel(fronts + me.front_off, me.f * me.w, lane);
        const int par = me.parent;
        if (par >= 0) wave_wait_flag(flags + top_pos[par], err, epoch);
        touch_done(pt);
    }
    int cur = 0;
    bool prev_valid = false;  // sv[cur ^ 1][0 .. f_parent) = final vector of supernode s + 1
    SnRec me = M.sn[hi];
    for (int s = hi; s >= lo; --s) {
        const SnRec nxt = M.sn[s > lo ? s - 1 : s];  // the next link's record travels while this one is processed
        const int first = me.first, w = me.w, f = me.f;
        const double* F = fronts + me.front_off;
        const int* rows = M.front_rows + me.rows_ptr;
        const bool from_lds = prev_valid && me.parent == s + 1;
        const double* pv = sv[cur ^ 1];
        const int* rel = M.rel + me.rel_ptr;
        double v0 = 0.0, v1 = 0.0;
        // rdiag != nullptr: the diagonal solve x <- D^-1 x rides along (every pivot is loaded exactly once in the backward sweep)
        if (r0 < f) v0 = r0 < w ? (rdiag ? x[first + r0] * rdiag[first + r0] : x[first + r0]) : (from_lds ? pv[rel[r0 - w]] : ldx<TOP>(x + rows[r0]));
        if (r1 < f) v1 = r1 < w ? (rdiag ? x[first + r1] * rdiag[first + r1] : x[first + r1]) : (from_lds ? pv[rel[r1 - w]] : ldx<TOP>(x + rows[r1]));
        // y1[j] -= sum_{i >= w} L[i,j] x2[i].  Few pivots under many update rows (the usual shape inside a subtree): lanes over the rows i,
        // one coalesced column load and one wave reduction per pivot.  Otherwise lane j = pivot column j, ascending i, like front_bwd.
        if (f - w > red_thr * w) {
            int k = 0;
            for (; k + 2 <= w; k += 2) {
                const double* ca = F + (long long)k * f;
                const double* cb = ca + f;
                double pa = 0.0, pb = 0.0;
                if (r0 >= w && r0 < f) { pa = ca[r0] * v0; pb = cb[r0] * v0; }
                if (r1 >= w && r1 < f) { pa += ca[r1] * v1; pb += cb[r1] * v1; }
                for (int o = 32; o > 0; o >>= 1) { pa += __shfl_xor(pa, o); pb += __shfl_xor(pb, o); }
                if (r0 == k) v0 -= pa;
                if (r0 == k + 1) v0 -= pb;
                if (r1 == k) v1 -= pa;
                if (r1 == k + 1) v1 -= pb;
            }
            for (; k < w; ++k) {
                const double* ca = F + (long long)k * f;
                double pa = 0.0;
                if (r0 >= w && r0 < f) pa = ca[r0] * v0;
                if (r1 >= w && r1 < f) pa += ca[r1] * v1;
                for (int o = 32; o > 0; o >>= 1) pa += __shfl_xor(pa, o);
                if (r0 == k) v0 -= pa;
                if (r1 == k) v1 -= pa;
            }
        } else if (w <= 32 && f - w >= 16) {
            // at most 32 pivot columns: the wave splits into G = 64 / P groups (P = 8, 16 or 32 lanes, one per column) and every group takes
            // 1 / G of the update rows, so the sweep is (f - w) / (4 G) dependent round trips to L2 instead of (f - w) / 4; the groups'
            // partial sums are combined by shuffles.  x2 is read from LDS (the lanes of a batch need different rows).
            double* xs = sv[cur];
            if (r0 < f) xs[r0] = v0;
            if (r1 < f) xs[r1] = v1;
            __syncthreads();
            const int P = w <= 8 ? 8 : (w <= 16 ? 16 : 32), G = 64 / P;
            const int j = lane & (P - 1), h = lane / P;
            const int u = f - w, chunk = (u + G - 1) / G;
            const int ibeg = w + h * chunk, iend = min(f, ibeg + chunk);
            const double* cj = F + (long long)j * f;
            double sp = 0.0;
            for (int i = ibeg; i < iend; i += 4) {
                double a[4], xv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { const bool ok = j < w && i + q < iend; a[q] = ok ? cj[i + q] : 0.0; xv[q] = ok ? xs[i + q] : 0.0; }
#pragma unroll
                for (int q = 0; q < 4; ++q) sp += a[q] * xv[q];
            }
            for (int o = P; o < 64; o <<= 1) sp += __shfl_xor(sp, o);
            if (lane < w) v0 -= sp;  // lane = column = pivot row (w <= 32)
            __syncthreads();         // xs is reused for the final vector below
        } else {
            double s0 = 0.0, s1 = 0.0;
            const double* cj0 = F + (long long)r0 * f;
            const double* cj1 = F + (long long)r1 * f;
            int i = w;
            for (; i + 4 <= f; i += 4) {
                double a0[4], a1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { a0[u] = r0 < w ? cj0[i + u] : 0.0; a1[u] = r1 < w ? cj1[i + u] : 0.0; }
#pragma unroll
                for (int u = 0; u < 4; ++u) { const double xi = bcast_row(v0, v1, i + u); s0 += a0[u] * xi; s1 += a1[u] * xi; }
            }
            for (; i < f; ++i) {
                const double a0 = r0 < w ? cj0[i] : 0.0, a1 = r1 < w ? cj1[i] : 0.0;
                const double xi = bcast_row(v0, v1, i);
                s0 += a0 * xi; s1 += a1 * xi;
            }
            if (r0 < w) v0 -= s0;
            if (r1 < w) v1 -= s1;
        }
        // x1 = L11^-T y1: rows w-1 .. 1, lane j < i
        {
            const double* cj0 = F + (long long)r0 * f;
            const double* cj1 = F + (long long)r1 * f;
            int i = w - 1;
            for (; i - 3 >= 1; i -= 4) {
                double a0[4], a1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { a0[u] = r0 < i - u ? cj0[i - u] : 0.0; a1[u] = r1 < i - u ? cj1[i - u] : 0.0; }
#pragma unroll
                for (int u = 0; u < 4; ++u) { const double xi = bcast_row(v0, v1, i - u); v0 -= a0[u] * xi; v1 -= a1[u] * xi; }
            }
            for (; i >= 1; --i) {
                const double a0 = r0 < i ? cj0[i] : 0.0, a1 = r1 < i ? cj1[i] : 0.0;
                const double xi = bcast_row(v0, v1, i);
                v0 -= a0 * xi; v1 -= a1 * xi;
            }
        }
        if (r0 < w) stx<TOP>(x + first + r0, v0);
        if (r1 < w) stx<TOP>(x + first + r1, v1);
        double* a = sv[cur];
        if (r0 < f) a[r0] = v0;
        if (r1 < f) a[r1] = v1;
        __syncthreads();
        cur ^= 1;
        prev_valid = true;
        if (TOP) { const int tp = top_pos[s]; if (pub[tp]) wave_publish_flag(flags + tp, epoch); }  // only where a child in another walk waits for it
        me = nxt;
    }
}
template <bool TOP>
__global__ __launch_bounds__(64) void k_subtree_bwd_wave(FrontMeta M, const double* __restrict__ fronts, const int* __restrict__ sub_lo, const int* __restrict__ sub_hi,
                                                         double* __restrict__ x, int red_thr, int ntop, const int* __restrict__ top_pos, int* __restrict__ flags,
                                                         int* __restrict__ err, const int* __restrict__ pub, int epoch, const double* __restrict__ rdiag)
{
    subtree_bwd_wave_body<TOP>(M, fronts, sub_lo, sub_hi, x, red_thr, ntop, top_pos, flags, err, pub, epoch, rdiag, (int)blockIdx.x);
}
// A level that holds single-wave fronts AND wide fronts: both kinds in ONE launch (the first nn workgroups run a single-wave front on their wave 0, the
// other waves leave at once -- a barrier does not wait for finished waves --, the rest are the wide fronts).  The two launches of such a level ran one after
// the other, 6-13 us each whatever they did.
__global__ __launch_bounds__(WIDE_NT) void k_level_fwd_mixed(FrontMeta M, const double* __restrict__ fronts, const int* __restrict__ list, int nn, double* __restrict__ x,
                                                             double* __restrict__ fvec, int fcap, const int* __restrict__ hoff)
{
    if ((int)blockIdx.x < nn) {
        if (threadIdx.x >= 64) return;
        subtree_fwd_wave_body<false>(M, fronts, list, list, x, fvec, nullptr, nullptr, nullptr, nullptr, 0, (int)blockIdx.x);
        return;
    }
    front_fwd_wide_body(M, fronts, list + nn, x, fvec, fcap, (int)blockIdx.x - nn, hoff);
}
__global__ __launch_bounds__(WIDE_NT) void k_level_bwd_mixed(FrontMeta M, const double* __restrict__ fronts, const int* __restrict__ list, int nn, double* __restrict__ x,
                                                             double* __restrict__ fvec, int fcap, int red_thr, const int* __restrict__ hoff, const double* __restrict__ hpart)
{
    if ((int)blockIdx.x < nn) {
        if (threadIdx.x >= 64) return;
        subtree_bwd_wave_body<false>(M, fronts, list, list, x, red_thr, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, (int)blockIdx.x);
        return;
    }
    front_bwd_wide_body(M, fronts, list + nn, x, fvec, fcap, (int)blockIdx.x - nn, hoff, hpart);
}

// ---- the top of the assembly tree in ONE launch: workgroup b takes the top supernodes b, b + G, ... of the level-sorted (=
// topological) list and waits on per-supernode completion flags instead of on kernel boundaries.  Every dependency has a
// smaller list index and all G workgroups are resident (G <= number of CUs, one workgroup per CU), so the smallest unfinished
// index is always being worked on: no deadlock.  Flags are released / acquired at agent scope (the data crosses XCD L2s).
// every spin is bounded: if the workgroups are not all resident (another stream occupies the device) the wait gives up, raises
// `err` and the launch drains instead of hanging; the host then reports a failed factorisation / a non-finite solve
__device__ __forceinline__ void top_done(int* flag, int epoch)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// `list` / `ntop` may be a SUFFIX of the level-sorted top list (start = its first position): children in earlier levels or in
// subtrees were finished by earlier launches on the stream
template <int RM>
__global__ __launch_bounds__(512) void k_top_factor(FrontMeta M, double* __restrict__ fronts, const int* __restrict__ list, int ntop, const int* __restrict__ top_pos, int start,
                                                    int* __restrict__ flags, int* __restrict__ err, double* __restrict__ rdiag, int* __restrict__ info, int epoch)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    for (int b = blockIdx.x; b < ntop; b += gridDim.x) {
        const int s = list[b];
        const SnRec me = M.sn[s];
        front_assemble_own(M, fronts, me, lds);
        // one lane polls every child's flag (relaxed), then ONE agent-scope acquire by that lane invalidates this CU's L1 for the whole
        // workgroup (a fence per child executed by all 256 threads cost several microseconds per front)
        if (threadIdx.x == 0) {
            bool any = false;
            for (int ci = me.child_lo; ci < me.child_hi; ++ci) {
                const int tp = top_pos[M.child[ci]] - start;
                if (tp < 0) continue;
                any = true;
                unsigned spins = 0;
                while (__hip_atomic_load(flags + tp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 8000000u || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) {
                        __hip_atomic_store(err, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (*info < 0) *info = 0;  // the factorisation is reported as failed (what k_top_check did after the launch)
                        break;
                    }
                }
            }
            if (any) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        front_factor<RM>(M, fronts, s, rdiag, info, lds, true);
        top_done(flags + b, epoch);
    }
}

// The persistent top launch over WALKS: a walk is a chain lo..hi of top supernodes with parent[t] == t + 1.  One workgroup factors the
// chain with two packed lower-triangle fronts in LDS (like k_subtree_factor_pk): the update matrix of a link stays in LDS for the
// next one, flags + HBM round trips only where the tree branches.  flags are per supernode (position in the list - start); walks are
// sorted by the position of their last supernode, so every dependency has a smaller walk index.
template <int RM>
__global__ __launch_bounds__(512) void k_top_factor_walk(FrontMeta M, double* __restrict__ fronts, const double* __restrict__ vals, const int* __restrict__ fe_offp,
                                                         const int* __restrict__ walk_lo, const int* __restrict__ walk_hi, int nwalk, const int* __restrict__ top_pos, int start,
                                                         int cap, int* __restrict__ flags, int* __restrict__ err, double* __restrict__ rdiag, int* __restrict__ info, int epoch)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    PQ_REF_FLAGS;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int tx = tid & 15, ty = tid >> 4, tys = nt >> 4;
    for (int b = blockIdx.x; b < nwalk; b += gridDim.x) {
        const int lo = walk_lo[b], hi = walk_hi[b];
        double* cur = lds;
        double* prev = lds + cap;
        bool prev_valid = false;
        for (int s = lo; s <= hi; ++s) {
            const SnRec me = M.sn[s];
            const int first = me.first, w = me.w, f = me.f;
            double* W = cur;
            const int npk = (f * (f + 1)) >> 1;
            // what needs nothing from the children comes first: they may still be busy
            for (int idx = tid; idx < npk; idx += nt) W[idx] = 0.0;
            __syncthreads();
            for (int e = me.fe_lo + tid; e < me.fe_hi; e += nt) W[fe_offp[e]] = vals[e];
            if (tid == 0) {
                bool any = false;
                for (int ci = me.child_lo; ci < me.child_hi; ++ci) {
                    const int c = M.child[ci];
                    if (prev_valid && c == s - 1) continue;
                    const int tp = top_pos[c] - start;
                    if (tp < 0) continue;
                    any = true;
                    unsigned spins = 0;
                    while (__hip_atomic_load(flags + tp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > 8000000u || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) {
                            __hip_atomic_store(err, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (*info < 0) *info = 0;  // the factorisation is reported as failed
                            break;
                        }
                    }
                }
                if (any) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            for (int ci = me.child_lo; ci < me.child_hi; ++ci) {
                const int c = M.child[ci];
                const SnRec ch = M.sn[c];
                const int wc = ch.w, fc = ch.f, uc = fc - wc;
                const int* rel = M.rel + ch.rel_ptr;
                if (prev_valid && c == s - 1) {
                    for (int j = ty; j < uc; j += tys) {
                        const int cj = pk_base(rel[j], f);
                        const double* Uj = prev + pk_base(wc + j, fc) + wc;
                        for (int i = j + tx; i < uc; i += 16) W[cj + rel[i]] += Uj[i];
                    }
                } else {
                    const double* U = fronts + ch.front_off + wc + (long long)wc * fc;
                    for (int j = ty; j < uc; j += tys) {
                        const int cj = pk_base(rel[j], f);
                        for (int i = j + tx; i < uc; i += 16) W[cj + rel[i]] += U[i + (long long)j * fc];
                    }
                }
                __syncthreads();
            }
            auto tptr = [&](int i) -> double* { return W + pk_base(i, f) + i; };
            double td[TD_REGS];
            if constexpr (REF_DIAG) td_load(td, w, f, tid, nt, tptr);
            for (int k = independent_pivots_pk<RM>(W, f, w, me.nind, first, prev, rdiag, info, tid, nt, td); k < w; ++k) {
                const int ck = pk_base(k, f);
                double d = W[ck + k];
                if (d == 0.0) { if (tid == 0 && *info < 0) *info = first + k; d = 1.0; }
                const double dinv = pivot_rcp(d);
                if (tid == 0) rdiag[first + k] = dinv;
                const int r = f - k - 1, pc = w - k - 1;
                const double* colk = W + ck + (k + 1);
                for (int j = ty; j < pc; j += tys) {
                    const double cj = colk[j];
                    const double lj = REF_PIVOT ? ref_quot(cj, d, dinv) : 0.0;
                    double* Wj = W + pk_base(k + 1 + j, f) + (k + 1);
                    for (int i = j + tx; i < r; i += 16) Wj[i] = pivot_term<RM>(Wj[i], colk[i], cj, d, dinv, lj, i == j);
                }
                if constexpr (REF_DIAG) td_step(td, w, f, tid, nt, tptr, [&](int i) { return W[ck + i]; }, d, dinv);
                __syncthreads();
            }
            if constexpr (REF_DIAG) td_store(td, w, f, tid, nt, tptr);
            const int u = f - w;
            if constexpr (REF_SCHUR) {
                // deferred division of the finished columns, Schur complement, trailing rows' quotients into place (see front_factor_body)
                scale_columns_pk(W, prev, f, w, tid, nt);
                __syncthreads();
                if (u > 0) {
                    schur_2x2_pk(W, prev, f, w, u, tid, nt);
                    __syncthreads();
                    place_trailing_pk(W, prev, f, w, tid, nt);
                    __syncthreads();
                }
            } else {
                for (int k = ty; k < w; k += tys) {   // deferred scaling of the finished columns (see front_factor)
                    double* Ck = W + pk_base(k, f);
                    double d = Ck[k];
                    if (d == 0.0) d = 1.0;
                    const double dinv = pivot_rcp(d);
                    for (int i = k + 1 + tx; i < f; i += 16) Ck[i] = scale_entry<RM>(Ck[i], d, dinv);
                }
                __syncthreads();
                if (u > 0) {
                    schur_2x2_pk_sum<RM>(W, f, w, u, tid, nt);
                    __syncthreads();
                }
            }
            double* F = fronts + me.front_off;
            for (int j = ty; j < w; j += tys) {
                const double* Cj = W + pk_base(j, f);
                double* Fj = F + (long long)j * f;
                for (int i = j + tx; i < f; i += 16) Fj[i] = Cj[i];
            }
            const bool keep = me.parent == s + 1 && s + 1 <= hi;
            if (!keep && u > 0) {
                for (int j = ty; j < u; j += tys) {
                    const double* Cj = W + pk_base(w + j, f) + w;
                    for (int i = j + tx; i < u; i += 16) F[(w + i) + (long long)(w + j) * f] = Cj[i];
                }
            }
            if (!keep) top_done(flags + (top_pos[s] - start), epoch);  // barrier + release + flag
            else __syncthreads();
            double* t = cur; cur = prev; prev = t;
            prev_valid = keep;
        }
    }
}
// a timed-out wait surfaces as a failed factorisation (info) or a NaN in the solution (caught by KKTSystem's finite check)
__global__ void k_top_check(const int* __restrict__ err, int* __restrict__ info, double* __restrict__ x)
{
    if (*err == 0) return;
    if (info && *info < 0) *info = 0;
    if (x) x[0] = __builtin_nan("");
}

// ---- stage partition (pq_kkt_partition): data that crosses ranks -----------------------------------------------------------
// one workgroup per boundary subtree root: its u x u update block (lower triangle) -> dense slot of the exchange buffer; slots of
// other ranks' subtrees are zeroed so that an all-reduce(sum) delivers every block to every rank.  The last double carries
// "a pivot of my subtrees was zero".
__global__ __launch_bounds__(256) void k_pack_updates(FrontMeta M, const double* __restrict__ fronts, const int* __restrict__ bsn, const int* __restrict__ bowner, int rank,
                                                      const long long* __restrict__ boff, long long flag_off, const int* __restrict__ info, double* __restrict__ buf)
{
    const int b = blockIdx.x;
    const SnRec ch = M.sn[bsn[b]];
    const int wc = ch.w, fc = ch.f, uc = fc - wc;
    double* out = buf + boff[b];
    if (bowner[b] == rank) {
        const double* U = fronts + ch.front_off + wc + (long long)wc * fc;
        for (int idx = threadIdx.x; idx < uc * uc; idx += blockDim.x) {
            const int i = idx % uc, j = idx / uc;
            out[idx] = i >= j ? U[i + (long long)j * fc] : 0.0;
        }
    } else {
        for (int idx = threadIdx.x; idx < uc * uc; idx += blockDim.x) out[idx] = 0.0;
    }
    if (b == 0 && threadIdx.x == 0) buf[flag_off] = *info >= 0 ? 1.0 : 0.0;
}
__global__ __launch_bounds__(256) void k_unpack_updates(FrontMeta M, double* __restrict__ fronts, const int* __restrict__ bsn, const int* __restrict__ bowner, int rank,
                                                        const long long* __restrict__ boff, long long flag_off, int* __restrict__ info, const double* __restrict__ buf)
{
    const int b = blockIdx.x;
    if (bowner[b] != rank) {
        const SnRec ch = M.sn[bsn[b]];
        const int wc = ch.w, fc = ch.f, uc = fc - wc;
        const double* in = buf + boff[b];
        double* U = fronts + ch.front_off + wc + (long long)wc * fc;
        for (int idx = threadIdx.x; idx < uc * uc; idx += blockDim.x) {
            const int i = idx % uc, j = idx / uc;
            if (i >= j) U[i + (long long)j * fc] = in[idx];
        }
    }
    if (b == 0 && threadIdx.x == 0 && buf[flag_off] > 0.0 && *info < 0) *info = 0;
}
__global__ __launch_bounds__(64) void k_pack_fvec(FrontMeta M, const double* __restrict__ fvec, const int* __restrict__ bsn, const int* __restrict__ bowner, int rank,
                                                  const int* __restrict__ boff, double* __restrict__ buf)
{
    const int b = blockIdx.x;
    const SnRec ch = M.sn[bsn[b]];
    const int uc = ch.f - ch.w;
    const double* v = fvec + ch.rows_ptr + ch.w;
    const bool mine = bowner[b] == rank;
    for (int i = threadIdx.x; i < uc; i += blockDim.x) buf[boff[b] + i] = mine ? v[i] : 0.0;
}
__global__ __launch_bounds__(64) void k_unpack_fvec(FrontMeta M, double* __restrict__ fvec, const int* __restrict__ bsn, const int* __restrict__ bowner, int rank,
                                                    const int* __restrict__ boff, const double* __restrict__ buf)
{
    const int b = blockIdx.x;
    if (bowner[b] == rank) return;
    const SnRec ch = M.sn[bsn[b]];
    const int uc = ch.f - ch.w;
    double* v = fvec + ch.rows_ptr + ch.w;
    for (int i = threadIdx.x; i < uc; i += blockDim.x) v[i] = buf[boff[b] + i];
}
// solution columns: this rank's span -> its chunk of the gather buffer; after the all-gather every other rank's span -> x
__global__ void k_pack_span(int lo, int hi, const double* __restrict__ x, double* __restrict__ chunk)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (lo + i < hi) chunk[i] = x[lo + i];
}
__global__ void k_unpack_spans(int world, int rank, int max_span, const int* __restrict__ span_lo, const int* __restrict__ span_hi, const double* __restrict__ buf,
                               double* __restrict__ x)
{
    const int q = blockIdx.y;
    if (q == rank) return;
    const int lo = span_lo[q], hi = span_hi[q];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; lo + i < hi; i += gridDim.x * blockDim.x) x[lo + i] = buf[(long long)q * max_span + i];
}

// eliminated multipliers (condensed modes): the rows this rank owns -> its chunk of the gather buffer; afterwards every other rank's rows -> lhs_y / lhs_z.
// rows[] holds y row j as j and z row k as p + k, rank by rank (ptr[])
__global__ void k_pack_duals(int cnt, const int* __restrict__ rows, int p, const double* __restrict__ lhs_y, const double* __restrict__ lhs_z, double* __restrict__ chunk)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    const int r = rows[i];
    chunk[i] = r < p ? lhs_y[r] : lhs_z[r - p];
}
__global__ void k_unpack_duals(int rank, int slot, const int* __restrict__ ptr, const int* __restrict__ rows, int p, const double* __restrict__ buf, double* __restrict__ lhs_y,
                               double* __restrict__ lhs_z)
{
    const int q = blockIdx.y;
    if (q == rank) return;
    const int lo = ptr[q], cnt = ptr[q + 1] - lo;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += gridDim.x * blockDim.x) {
        const int r = rows[lo + i];
        const double v = buf[(long long)q * slot + i];
        if (r < p) lhs_y[r] = v; else lhs_z[r - p] = v;
    }
}

inline dim3 g1(int n) { return dim3(n > 0 ? (n + 255) / 256 : 1); }

class SparseKKT final : public KKTSolverBase {
public:
    SparseKKT(const pq_sparse_data* d, int mode, int device) : dev_(device), mode_(mode)
    {
        if (d->mem != PQ_MEM_HOST) throw std::runtime_error("sparse data must be host-resident");
        PQ_HIP(hipSetDevice(dev_));
        PQ_HIP(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
        PQ_HIP(hipStreamCreateWithFlags(&st2_, hipStreamNonBlocking));
        PQ_HIP(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming)); PQ_HIP(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
        sparse::analyse_kkt(d, mode, S_);
        n_ = S_.n; p_ = S_.p; m_ = S_.m; N_ = S_.N;
        compute_level_lds();
        build_device(d);
    }
    ~SparseKKT() override
    {
        (void)hipSetDevice(dev_);
        if (st_) { (void)hipStreamSynchronize(st_); }
        if (comm_) rccl::comm_destroy(comm_);
        if (ev_fork_) (void)hipEventDestroy(ev_fork_);
        if (ev_join_) (void)hipEventDestroy(ev_join_);
        if (st2_) (void)hipStreamDestroy(st2_);
        if (st_) (void)hipStreamDestroy(st_);
    }

    KKTSolverBase* clone() const override
    {
        PQ_HIP(hipSetDevice(dev_));
        stream_wait(st_);
        return new SparseKKT(*this, 0);
    }

    // sparse/kkt_full.hpp:212-251
    void update_data_sparse(const pq_sparse_data* d, int options) override
    {
        if (d->n != n_ || d->p != p_ || d->m != m_) throw std::runtime_error("update_data: dimension mismatch");
        PQ_HIP(hipSetDevice(dev_));
        // values only (identical sparsity is a precondition, solver.hpp:325,341,356); the flags say which matrices changed,
        // but Solver::update rewrites all of them through unscale -> rescale, so refresh everything that is stored
        (void)options;
        upload_values(d);
    }

    // sparse/kkt.hpp:83-105
    bool update_scalings_and_factor(double delta, const double* x_reg, const double* z_reg) override
    {
        PQ_ZONE("piqp_amd::SparseKKT::update_scalings_and_factor");
        PQ_HIP(hipSetDevice(dev_));
        delta_ = delta;
        const int t0 = prof_.begin(0, st_);
        static const bool full_diag = debug_token("no_diag_list") != nullptr;  // debugging aid
        if (mode_ == 0 && part_on_ && world_ > 1 && need_all_n_ > 0 && !full_diag) {
            // stage partition: only the diagonal entries of the fronts this rank factors (its own subtrees and the shared top)
            hipLaunchKernelGGL(k_set_diag_list, g1(need_all_n_), dim3(256), 0, st_, need_all_n_, need_all_.p, n_, p_, diag_pos_.p, ops_.P_diag(), x_reg, delta, z_reg, vals_.p);
        } else if (mode_ == 0) {
            hipLaunchKernelGGL(k_set_diag, g1(N_), dim3(256), 0, st_, n_, p_, m_, diag_pos_.p, ops_.P_diag(), x_reg, delta, z_reg, vals_.p);
        } else if (part_on_ && world_ > 1 && sel_ready_ && !full_diag) {
            // stage partition, condensed modes: the values of the fronts this rank factors (its own subtrees and the shared top) only -- the same
            // kernels over the lists of source entries whose destination lies in such a front; the other fronts' values are never read on this rank
            const bool eq = mode_ & 1, ineq = mode_ & 2;
            if (m_ > 0) hipLaunchKernelGGL(k_reciprocal, g1(m_), dim3(256), 0, st_, m_, z_reg, zinv_.p);
            for (const auto& r : own_val_ranges_) PQ_HIP(hipMemsetAsync(vals_.p + r.first, 0, sizeof(double) * (r.second - r.first), st_));
            if (selP_n_) hipLaunchKernelGGL(k_remap_values_sel, g1(selP_n_), dim3(256), 0, st_, selP_n_, selP_.p, mapP_.p, ops_.P_x(), vals_.p);
            if (selD_n_) hipLaunchKernelGGL(k_cond_diag, g1(selD_n_), dim3(256), 0, st_, n_, eq ? 0 : p_, ineq ? 0 : m_, diag_pos_.p, x_reg, delta, z_reg, vals_.p, selD_.p, selD_n_);
            if (eq) { if (selAA_n_) hipLaunchKernelGGL(k_axpy_mapped, g1(selAA_n_), dim3(256), 0, st_, selAA_n_, mapAA_.p, 1.0 / delta, ata_vals_.p, vals_.p, selAA_.p); }
            else if (selA_n_) hipLaunchKernelGGL(k_remap_values_sel, g1(selA_n_), dim3(256), 0, st_, selA_n_, selA_.p, mapA_.p, ops_.AT_x(), vals_.p);
            if (ineq) {
                if (selGG_n_) hipLaunchKernelGGL(k_gram_values<true>, g1(selGG_n_), dim3(256), 0, st_, selGG_n_, gg_ptr_.p, gg_q1_.p, gg_q2_.p, gg_k_.p, ops_.GT_x(), z_reg, mapGG_.p, vals_.p, selGG_.p);
            } else if (selG_n_) hipLaunchKernelGGL(k_remap_values_sel, g1(selG_n_), dim3(256), 0, st_, selG_n_, selG_.p, mapG_.p, ops_.GT_x(), vals_.p);
            ++sharded_asm_;
        } else {
            // update_kkt_cost_scalings / _equality_scalings / _inequality_scaling of the mode, in the reference's accumulation order
            const bool eq = mode_ & 1, ineq = mode_ & 2;
            if (m_ > 0) hipLaunchKernelGGL(k_reciprocal, g1(m_), dim3(256), 0, st_, m_, z_reg, zinv_.p);
            PQ_HIP(hipMemsetAsync(vals_.p, 0, sizeof(double) * (size_t)nnzK_, st_));
            launch_remap_values(ops_.nzP(), mapP_.p, ops_.P_x(), vals_.p, st_);
            hipLaunchKernelGGL(k_cond_diag, g1(N_), dim3(256), 0, st_, n_, eq ? 0 : p_, ineq ? 0 : m_, diag_pos_.p, x_reg, delta, z_reg, vals_.p);
            if (eq) { if (nzAA_) hipLaunchKernelGGL(k_axpy_mapped, g1(nzAA_), dim3(256), 0, st_, nzAA_, mapAA_.p, 1.0 / delta, ata_vals_.p, vals_.p); }
            else launch_remap_values(ops_.nzA(), mapA_.p, ops_.AT_x(), vals_.p, st_);
            if (ineq) {
                if (nzGG_) hipLaunchKernelGGL(k_gram_values<true>, g1(nzGG_), dim3(256), 0, st_, nzGG_, gg_ptr_.p, gg_q1_.p, gg_q2_.p, gg_k_.p, ops_.GT_x(), z_reg, mapGG_.p, vals_.p);
            } else launch_remap_values(ops_.nzG(), mapG_.p, ops_.GT_x(), vals_.p, st_);
        }
        prof_.end(0, t0, st_);  // the fronts are assembled from `vals` inside the factor kernels (no HBM zero-fill / scatter pass)
        const int t1 = prof_.begin(1, st_);
        PQ_HIP(hipMemsetAsync(info_.p, 0xFF, sizeof(int), st_));
        FrontMeta M = meta();
        if (part_on_) {
            if (world_ > 1 && transport_ == Transport::None) throw std::runtime_error("partitioned backend used before pq_kkt_set_exchange / pq_kkt_set_comm_rccl");
            factor_subtrees(M, part_sched_);
            factor_levels(M, own_ptr_, own_sn_d_.p, own_lds_, own_big_);
            const int nb = (int)PT_.boundary.size();
            if (nb > 0) {
                hipLaunchKernelGGL(k_pack_updates, dim3(nb), dim3(256), 0, st_, M, fronts_.p, b_sn_.p, b_owner_.p, rank_, b_mat_off_.p, PT_.bmat_off.back(), info_.p, xbuf_factor_);
                exchange(0);
                hipLaunchKernelGGL(k_unpack_updates, dim3(nb), dim3(256), 0, st_, M, fronts_.p, b_sn_.p, b_owner_.p, rank_, b_mat_off_.p, PT_.bmat_off.back(), info_.p, xbuf_factor_);
            }
            factor_levels(M, sh_ptr_, sh_sn_d_.p, sh_lds_, sh_big_);
        } else {
            factor_numeric(M);
        }
        PQ_HIP(hipGetLastError());
        prof_.end(1, t1, st_);
        PQ_HIP(hipMemcpyAsync(info_h_.p, info_.p, sizeof(int), hipMemcpyDeviceToHost, st_));
        stream_wait(st_);
        return info_h_.p[0] == -1;  // n == cols (sparse/kkt.hpp:104)
    }

    // sparse/kkt.hpp:107-176, KKT_FULL branch
    void solve(const double* rhs_x, const double* rhs_y, const double* rhs_z, double* lhs_x, double* lhs_y, double* lhs_z) override
    {
        PQ_ZONE("piqp_amd::SparseKKT::solve");
        PQ_HIP(hipSetDevice(dev_));
        const int tk = prof_.begin(2, st_);
        FrontMeta M = meta();
        const bool eq = mode_ & 1, ineq = mode_ & 2;
        const double delta_inv = 1.0 / delta_;
        bool partial = false;
        if (mode_ == 0) {
            hipLaunchKernelGGL(k_perm_gather, g1(N_), dim3(256), 0, st_, N_, P_.p, rhs_x, n_, rhs_y, p_, rhs_z, xp_.p);
        } else {
            // sparse/kkt.hpp:113-136: fold the eliminated blocks into the x part of the right-hand side
            // (a right-hand side left behind by refine_error_sharded is valid on this rank's rows only: fold those, recover those -- finish_sharded_solve completes
            // the multipliers once per KKTSystem::solve)
            partial = part_on_ && world_ > 1 && sharded_agreed_ == 1 && rhs_x == partial_rhs_;
            if (partial) ops_.fold_rhs_rows(need_x_.p, fold_x_n_, rhs_x, rhs_y, rhs_z, zinv_.p, delta_inv, rhs_top_.p, st_, eq, ineq);
            else ops_.fold_rhs(rhs_x, rhs_y, rhs_z, zinv_.p, delta_inv, rhs_top_.p, st_, eq, ineq);
            const double* tail = mode_ == 1 ? rhs_z : rhs_y;
            const int ntail = mode_ == 1 ? m_ : (mode_ == 2 ? p_ : 0);
            hipLaunchKernelGGL(k_perm_gather, g1(N_), dim3(256), 0, st_, N_, P_.p, rhs_top_.p, n_, tail, ntail, (const double*)nullptr, xp_.p);
        }
        if (part_on_) {
            if (world_ > 1 && transport_ == Transport::None) throw std::runtime_error("partitioned backend used before pq_kkt_set_exchange / pq_kkt_set_comm_rccl");
            subtree_fwd(M, part_sched_);
            fwd_levels(M, own_ll_);
            const int nb = (int)PT_.boundary.size();
            if (nb > 0) {
                hipLaunchKernelGGL(k_pack_fvec, dim3(nb), dim3(64), 0, st_, M, fvec_.p, b_sn_.p, b_owner_.p, rank_, b_vec_off_.p, xbuf_forward_);
                exchange(1);
                hipLaunchKernelGGL(k_unpack_fvec, dim3(nb), dim3(64), 0, st_, M, fvec_.p, b_sn_.p, b_owner_.p, rank_, b_vec_off_.p, xbuf_forward_);
            }
            fwd_levels(M, sh_ll_);
            hipLaunchKernelGGL(k_scale, g1(N_), dim3(256), 0, st_, N_, rdiag_.p, xp_.p);
            bwd_levels(M, sh_ll_);
            bwd_levels(M, own_ll_);
            subtree_bwd(M, part_sched_);
            if (world_ > 1 || (std::getenv("PIQP_AMD_EXCHANGE_WORLD1") && transport_ != Transport::None)) {
                const int lo = PT_.span_lo[rank_], hi = PT_.span_hi[rank_];
                if (hi > lo) hipLaunchKernelGGL(k_pack_span, g1(hi - lo), dim3(256), 0, st_, lo, hi, xp_.p, xbuf_gather_ + (size_t)rank_ * gather_slot_);
                exchange(2);
                hipLaunchKernelGGL(k_unpack_spans, dim3(std::max(1, std::min(256, (PT_.max_span + 255) / 256)), world_), dim3(256), 0, st_, world_, rank_, gather_slot_, span_lo_d_.p,
                                   span_hi_d_.p, xbuf_gather_, xp_.p);
            }
        } else {
            solve_numeric(M);
        }
        if (mode_ == 0) {
            hipLaunchKernelGGL(k_perm_scatter, g1(N_), dim3(256), 0, st_, N_, P_.p, xp_.p, lhs_x, n_, lhs_y, p_, lhs_z, solve_err_ptr_, solve_epoch_used_);
        } else {
            double* tail = mode_ == 1 ? lhs_z : lhs_y;
            const int ntail = mode_ == 1 ? m_ : (mode_ == 2 ? p_ : 0);
            hipLaunchKernelGGL(k_perm_scatter, g1(N_), dim3(256), 0, st_, N_, P_.p, xp_.p, lhs_x, n_, tail, ntail, (double*)nullptr, solve_err_ptr_, solve_epoch_used_);
            if (partial) {
                ops_.recover_duals_rows(need_y_.p, eq ? need_y_n_ : 0, need_z_.p, ineq ? need_z_n_ : 0, lhs_x, rhs_y, rhs_z, zinv_.p, delta_inv, lhs_y, lhs_z, st_);
                ++sharded_solves_;
            } else ops_.recover_duals(lhs_x, rhs_y, rhs_z, zinv_.p, delta_inv, lhs_y, lhs_z, st_, eq, ineq);  // sparse/kkt.hpp:147-175
        }
        PQ_HIP(hipGetLastError());
        prof_.end(2, tk, st_);
    }

    // sparse/kkt.hpp:179-203
    void eval_P_x(double alpha, const double* x, double* z) override
    {
        PQ_ZONE("piqp_amd::SparseKKT::eval_P_x");
        PQ_HIP(hipSetDevice(dev_));
        ops_.eval_P_x(alpha, x, z, st_);
    }
    void eval_A_xn_and_AT_xt(double alpha_n, double alpha_t, const double* xn, const double* xt, double* zn, double* zt) override
    {
        PQ_HIP(hipSetDevice(dev_));
        ops_.eval_A_xn_and_AT_xt(alpha_n, alpha_t, xn, xt, zn, zt, st_);
    }
    void eval_G_xn_and_GT_xt(double alpha_n, double alpha_t, const double* xn, const double* xt, double* zn, double* zt) override
    {
        PQ_HIP(hipSetDevice(dev_));
        ops_.eval_G_xn_and_GT_xt(alpha_n, alpha_t, xn, xt, zn, zt, st_);
    }

    void sparse_stats(double out[8]) const override
    {
        out[0] = N_; out[1] = nnzK_; out[2] = (double)S_.nnzL; out[3] = S_.nsuper; out[4] = S_.nlevels; out[5] = S_.nsub; out[6] = S_.max_front; out[7] = S_.flops;
    }
    int sparse_ordering(int* fill_perm, int* elim_perm) const override
    {
        if (fill_perm) std::copy(S_.fill_perm.begin(), S_.fill_perm.end(), fill_perm);
        if (elim_perm) std::copy(S_.P.begin(), S_.P.end(), elim_perm);
        return std::string(S_.ordering) == "amd" ? 0 : 1;
    }
    // ---- pq_kkt_partition / pq_kkt_set_exchange (include/piqp_amd.h)
    void partition(int rank, int world, long long sizes[3]) override
    {
        if (world < 1 || rank < 0 || rank >= world) throw std::runtime_error("partition: bad rank / world");
        PQ_HIP(hipSetDevice(dev_));
        sparse::partition_tree(S_, world, PT_);
        rank_ = rank; world_ = world;
        // schedules of this rank: the workgroup subtrees and level lists it owns, and the shared level lists
        std::vector<int> mine;
        for (int k = 0; k < S_.nsub; ++k) if (PT_.owner[S_.sub_hi[k]] == rank) mine.push_back(k);
        build_sub_schedule(mine, part_sched_);
        auto filter = [&](int want, std::vector<int>& ptr, std::vector<int>& sn, DBuf<int>& dev, std::vector<int>& lds) {
            ptr.assign(1, 0); sn.clear(); lds.clear();
            for (int l = 0; l < S_.top_nlevels; ++l) {
                long long mx = 0;
                const size_t before = sn.size();
                for (int q = S_.top_level_ptr[l]; q < S_.top_level_ptr[l + 1]; ++q) {
                    const int s = S_.top_level_sn[q];
                    if (PT_.owner[s] != want) continue;
                    sn.push_back(s);
                    const long long f = S_.front_rows_ptr[s + 1] - S_.front_rows_ptr[s];
                    mx = std::max(mx, front_lds_doubles(f, S_.sn_first[s + 1] - S_.sn_first[s]));
                }
                if (sn.size() > before) { ptr.push_back((int)sn.size()); lds.push_back(((int)mx + IND_SCRATCH) * (int)sizeof(double)); }
            }
            upload_vec(dev, sn, st_);
        };
        filter(rank, own_ptr_, own_sn_, own_sn_d_, own_lds_);
        filter(-1, sh_ptr_, sh_sn_, sh_sn_d_, sh_lds_);
        build_level_lists(own_ptr_, own_sn_, own_ll_); build_level_lists(sh_ptr_, sh_sn_, sh_ll_);
        build_big_levels(own_ptr_, own_sn_, own_big_); build_big_levels(sh_ptr_, sh_sn_, sh_big_);
        std::vector<int> bo(PT_.boundary.size());
        for (size_t b = 0; b < bo.size(); ++b) bo[b] = PT_.owner[PT_.boundary[b]];
        upload_vec(b_sn_, PT_.boundary, st_); upload_vec(b_owner_, bo, st_); upload_vec(b_mat_off_, PT_.bmat_off, st_); upload_vec(b_vec_off_, PT_.bvec_off, st_);
        upload_vec(span_lo_d_, PT_.span_lo, st_); upload_vec(span_hi_d_, PT_.span_hi, st_);
        {   // the rows of the KKT system whose columns this rank eliminates (owner == rank) or every rank does (shared top), in the caller's
            // numbering, split into the x / y / z blocks -- the rows its part of a solve reads and the diagonal entries its fronts hold (KKT_FULL only)
            std::vector<int> nx, ny, nz, nall;
            const bool elim_y = mode_ & 1, elim_z = mode_ & 2;
            for (int sn = 0; sn + 1 < (int)S_.sn_first.size(); ++sn) {
                if (PT_.owner[sn] != rank && PT_.owner[sn] >= 0) continue;
                for (int c = S_.sn_first[sn]; c < S_.sn_first[sn + 1]; ++c) {
                    const int v = S_.P[c];
                    if (mode_ == 0) nall.push_back(v);
                    if (v < n_) nx.push_back(v);
                    else if (mode_ == 0) { if (v < n_ + p_) ny.push_back(v - n_); else nz.push_back(v - n_ - p_); }
                    else if (mode_ == 1) nz.push_back(v - n_);  // (the block that stays in the system: z when the equalities are eliminated, y otherwise)
                    else ny.push_back(v - n_);
                }
            }
            dual_own_ptr_.assign((size_t)world + 1, 0);
            std::vector<int> dual_rows;
            fold_x_n_ = 0;
            if (mode_ != 0) {
                // Condensed modes (round 5): an eliminated constraint row matters to the ranks that eliminate an x column it touches -- their folded right-hand side
                // reads its residual and their x rows of the next residual read its multiplier.  Its OWNER (the rank whose value the others receive at the end of a
                // KKTSystem::solve) is the rank of the first such column outside the shared top, rank 0 when there is none.
                std::vector<int> Ap, Ai, Gp, Gi;
                ops_.download_column_patterns(Ap, Ai, Gp, Gi, st_);
                std::vector<int> col_owner((size_t)n_, -1);
                for (int sn = 0; sn + 1 < (int)S_.sn_first.size(); ++sn)
                    for (int c = S_.sn_first[sn]; c < S_.sn_first[sn + 1]; ++c) if (S_.P[c] < n_) col_owner[S_.P[c]] = PT_.owner[sn];
                auto block = [&](bool elim, int rows, const std::vector<int>& Mp, const std::vector<int>& Mi, std::vector<int>& need, int shift) {
                    if (!elim || rows == 0) return;
                    std::vector<int> owner((size_t)rows, -1);
                    std::vector<char> mine((size_t)rows, 0);
                    for (int j = 0; j < n_; ++j) {
                        const int o = col_owner[j];
                        for (int q = Mp[j]; q < Mp[j + 1]; ++q) {
                            const int i = Mi[q];
                            if (o < 0 || o == rank) mine[i] = 1;
                            if (o >= 0 && owner[i] < 0) owner[i] = o;
                        }
                    }
                    for (int i = 0; i < rows; ++i) {
                        if (owner[i] < 0) { owner[i] = 0; if (rank == 0) mine[i] = 1; }  // touches the shared top only, or nothing at all
                        if (mine[i]) need.push_back(i);
                    }
                    for (int r = 0; r < world; ++r) for (int i = 0; i < rows; ++i) if (owner[i] == r) dual_own_ptr_[r + 1]++;
                    dual_owner_tmp_.push_back({shift, std::move(owner)});
                };
                dual_owner_tmp_.clear();
                block(elim_y, p_, Ap, Ai, ny, 0);
                block(elim_z, m_, Gp, Gi, nz, p_);
                for (int r = 0; r < world; ++r) dual_own_ptr_[r + 1] += dual_own_ptr_[r];
                dual_rows.assign((size_t)dual_own_ptr_[world], 0);
                std::vector<int> fill(dual_own_ptr_.begin(), dual_own_ptr_.end() - 1);
                for (const auto& b : dual_owner_tmp_) for (int i = 0; i < (int)b.second.size(); ++i) dual_rows[fill[b.second[i]]++] = b.first + i;
                dual_owner_tmp_.clear();
                fold_x_n_ = (int)nx.size();
            }
            std::sort(nx.begin(), nx.end()); std::sort(ny.begin(), ny.end()); std::sort(nz.begin(), nz.end()); std::sort(nall.begin(), nall.end());
            upload_vec(dual_rows_, dual_rows, st_); upload_vec(dual_own_ptr_d_, dual_own_ptr_, st_);
            int most = 0;
            for (int r = 0; r < world; ++r) most = std::max(most, dual_own_ptr_[r + 1] - dual_own_ptr_[r]);
            gather_slot_ = std::max(std::max(1, PT_.max_span), most);
            partial_rhs_ = nullptr; sharded_solves_ = 0; dual_gathers_ = 0;
            upload_vec(need_x_, nx, st_); upload_vec(need_y_, ny, st_); upload_vec(need_z_, nz, st_); upload_vec(need_all_, nall, st_);
            need_x_n_ = (int)nx.size(); need_y_n_ = (int)ny.size(); need_z_n_ = (int)nz.size(); need_all_n_ = (int)nall.size();
            sel_ready_ = false; sharded_asm_ = 0; own_val_ranges_.clear();
            selP_n_ = selA_n_ = selG_n_ = selAA_n_ = selGG_n_ = selD_n_ = 0;
            if (mode_ != 0) {
                // condensed modes: the value array is stored front by front (position e holds PKPt entry fe_q[e]); a position is needed here when its front is
                const bool eq = mode_ & 1, ineq = mode_ & 2;
                std::vector<char> needed((size_t)std::max(nnzK_, 1), 0);
                for (int sn = 0; sn + 1 < (int)S_.sn_first.size(); ++sn) {
                    if (PT_.owner[sn] != rank && PT_.owner[sn] >= 0) continue;
                    const int lo = S_.fe_ptr[sn], hi = S_.fe_ptr[sn + 1];
                    for (int e = lo; e < hi; ++e) needed[e] = 1;
                    if (hi > lo) {
                        if (!own_val_ranges_.empty() && own_val_ranges_.back().second == (size_t)lo) own_val_ranges_.back().second = (size_t)hi;
                        else own_val_ranges_.push_back({(size_t)lo, (size_t)hi});
                    }
                }
                std::vector<int> pos((size_t)std::max(nnzK_, 1), 0);
                for (int e = 0; e < nnzK_; ++e) pos[S_.fe_q[e]] = e;
                auto pick = [&](size_t count, auto&& dest_of, DBuf<int>& buf, int& cnt) {
                    std::vector<int> sel;
                    for (size_t q = 0; q < count; ++q) if (needed[pos[dest_of(q)]]) sel.push_back((int)q);
                    cnt = (int)sel.size();
                    upload_vec(buf, sel, st_);
                };
                pick((size_t)ops_.nzP(), [&](size_t q) { return S_.PKi[S_.P_utri_to_Ki[q]]; }, selP_, selP_n_);
                if (!eq) pick((size_t)ops_.nzA(), [&](size_t q) { return S_.PKi[S_.AT_to_Ki[q]]; }, selA_, selA_n_);
                if (!ineq) pick((size_t)ops_.nzG(), [&](size_t q) { return S_.PKi[S_.GT_to_Ki[q]]; }, selG_, selG_n_);
                if (eq) pick((size_t)nzAA_, [&](size_t e) { return S_.PKi[S_.gramA_to_Ki[e]]; }, selAA_, selAA_n_);
                if (ineq) pick((size_t)nzGG_, [&](size_t e) { return S_.PKi[S_.gramG_to_Ki[e]]; }, selGG_, selGG_n_);
                pick(S_.diag_pos.size(), [&](size_t c) { return S_.diag_pos[c]; }, selD_, selD_n_);
                sel_ready_ = true;
            }
            norm_bits_.alloc(4);
            sharded_evals_ = 0;
        }
        PQ_HIP(hipMemsetAsync(rdiag_.p, 0, sizeof(double) * (size_t)N_, st_));
        stream_wait(st_);
        part_on_ = true;
        drop_transport();  // a new partition invalidates both transports: their buffer sizes and the communicator's world belong to the old one
        sizes[0] = PT_.bmat_off.back() + 1; sizes[1] = std::max(1, PT_.bvec_off.back()); sizes[2] = gather_slot_;
    }
    void set_exchange(pq_exchange_fn fn, void* user, double* buf_factor, double* buf_forward, double* buf_gather) override
    {
        if (!part_on_) throw std::runtime_error("set_exchange: call pq_kkt_partition first");
        if (world_ > 1 && (!fn || !buf_factor || !buf_forward || !buf_gather)) throw std::runtime_error("set_exchange: null argument");
        drop_transport();  // the callback transport replaces a native one (its communicator and the library's own buffers go)
        xfn_ = fn; xuser_ = user; xbuf_factor_ = buf_factor; xbuf_forward_ = buf_forward; xbuf_gather_ = buf_gather;
        transport_ = fn ? Transport::Callback : Transport::None;
    }
    void set_exchange_norm(double* buf_norm) override
    {
        if (!part_on_) throw std::runtime_error("set_exchange_norm: call pq_kkt_partition first");
        if (transport_ != Transport::Callback) throw std::runtime_error("set_exchange_norm: the callback transport only (the native transport owns its buffer)");
        xbuf_norm_ = buf_norm;
        sharded_agreed_ = -1;
    }
    bool refine_error_sharded(const double* lhs_x, const double* lhs_y, const double* lhs_z, const double* rhs_x, const double* rhs_y, const double* rhs_z, const double* x_reg,
                              double delta, const double* z_reg, double* err_x, double* err_y, double* err_z, double* norm) override
    {
        static const bool off = debug_token("replicated_residual") != nullptr;  // debugging aid: PIQP_AMD_DEBUG=replicated_residual
        if (!part_on_ || world_ < 2 || transport_ == Transport::None) return false;  // (structure and transport: the same on every rank by construction)
        PQ_HIP(hipSetDevice(dev_));
        if (sharded_agreed_ < 0) {
            // Whether THIS rank can take the sharded path also depends on per-process state (a registered norm buffer, the debugging switch, a column too long for the
            // row kernels): the ranks agree ONCE, through the factor exchange every rank has -- slot 0 carries "I can", the sum must be world -- so that no rank waits
            // in the all-reduce(max) below for one that went the replicated way (round-4 advice).  Without a boundary there is nothing to send and nobody to wait for.
            const bool can = !off && xbuf_norm_ != nullptr && !ops_.has_long_columns();
            if (!xbuf_factor_) sharded_agreed_ = can ? 1 : 0;
            else {
                // (the factor exchange buffer: always at least one word, free between factorisations, and tests count the forward / gather exchanges of the solves)
                PQ_HIP(hipMemsetAsync(xbuf_factor_, 0, sizeof(double) * ((size_t)PT_.bmat_off.back() + 1), st_));
                const double mine = can ? 1.0 : 0.0;
                PQ_HIP(hipMemcpyAsync(xbuf_factor_, &mine, sizeof(double), hipMemcpyHostToDevice, st_));
                stream_wait(st_);
                exchange(0);
                double sum = 0.0;
                PQ_HIP(hipMemcpyAsync(&sum, xbuf_factor_, sizeof(double), hipMemcpyDeviceToHost, st_));
                stream_wait(st_);
                sharded_agreed_ = sum == (double)world_ ? 1 : 0;
                if (can && !sharded_agreed_) std::fprintf(stderr, "piqp_amd: sharded refinement residual switched off -- %d of %d ranks can take it (norm buffer / debug switch / long columns differ)\n", (int)sum, world_);
            }
        }
        if (!sharded_agreed_) return false;
        PQ_HIP(hipMemsetAsync(norm_bits_.p, 0, sizeof(unsigned long long), st_));
        if (!ops_.residual_rows(need_x_.p, need_x_n_, need_y_.p, need_y_n_, need_z_.p, need_z_n_, lhs_x, lhs_y, lhs_z, rhs_x, rhs_y, rhs_z, x_reg, delta, z_reg, err_x, err_y, err_z,
                                reinterpret_cast<unsigned long long*>(norm_bits_.p), st_))
            return false;
        hipLaunchKernelGGL(k_norm_to_buf, dim3(1), dim3(1), 0, st_, reinterpret_cast<const unsigned long long*>(norm_bits_.p), xbuf_norm_);
        exchange(3);  // ONE all-reduce(max) per refinement step
        PQ_HIP(hipMemcpyAsync(norm_h_.p, xbuf_norm_, sizeof(double), hipMemcpyDeviceToHost, st_));
        stream_wait(st_);
        *norm = norm_h_.p[0];
        ++sharded_evals_;
        partial_rhs_ = err_x;  // (condensed modes: the next backend solve on this residual folds and recovers on this rank's rows only)
        return true;
    }
    // kkt_solver_base.hpp: the eliminated multipliers of a refined solve, each from its owner rank, in one all-gather
    void finish_sharded_solve(double* lhs_y, double* lhs_z) override
    {
        if (mode_ == 0 || !part_on_ || world_ < 2 || sharded_agreed_ != 1 || dual_own_ptr_.empty() || dual_own_ptr_.back() == 0) return;
        PQ_HIP(hipSetDevice(dev_));
        const int lo = dual_own_ptr_[rank_], cnt = dual_own_ptr_[rank_ + 1] - lo;
        if (cnt > 0) hipLaunchKernelGGL(k_pack_duals, g1(cnt), dim3(256), 0, st_, cnt, dual_rows_.p + lo, p_, lhs_y, lhs_z, xbuf_gather_ + (size_t)rank_ * gather_slot_);
        exchange(2);
        hipLaunchKernelGGL(k_unpack_duals, dim3(std::max(1, std::min(256, (gather_slot_ + 255) / 256)), world_), dim3(256), 0, st_, rank_, gather_slot_, dual_own_ptr_d_.p, dual_rows_.p, p_,
                           xbuf_gather_, lhs_y, lhs_z);
        PQ_HIP(hipGetLastError());
        ++dual_gathers_;
    }
    void sharded_solve_calls(int out[6]) const override
    {
        out[0] = sharded_evals_; out[1] = need_x_n_ + need_y_n_ + need_z_n_; out[2] = sharded_solves_; out[3] = dual_gathers_; out[4] = fold_x_n_;
        out[5] = ((mode_ & 1) ? need_y_n_ : 0) + ((mode_ & 2) ? need_z_n_ : 0);
    }
    // out[1]: KKT_FULL: rows of this rank's share of the residual; condensed modes: source entries of the value assembly this rank evaluates (of all: the same sum at world 1)
    void sharded_calls(int out[2]) const override
    {
        out[0] = mode_ == 0 ? sharded_evals_ : sharded_asm_;
        out[1] = mode_ == 0 ? need_x_n_ + need_y_n_ + need_z_n_ : selP_n_ + selA_n_ + selG_n_ + selAA_n_ + selGG_n_ + selD_n_;
    }
    void drop_transport()
    {
        xbuf_norm_ = nullptr; own_norm_.release(); sharded_agreed_ = -1;
        if (comm_) { stream_wait(st_); rccl::comm_destroy(comm_); comm_ = nullptr; }
        own_factor_.release(); own_forward_.release(); own_gather_.release();
        xfn_ = nullptr; xuser_ = nullptr; xbuf_factor_ = xbuf_forward_ = xbuf_gather_ = nullptr;
        transport_ = Transport::None;
    }
    // pq_kkt_set_comm_rccl: from here on the three exchanges are ncclAllReduce / ncclAllGather calls enqueued on st_ behind the kernels that
    // fill the library's own exchange buffers -- no stream drain, no host callback
    void set_comm_rccl(const unsigned char* id128, int rank, int world) override
    {
        if (!part_on_) throw std::runtime_error("set_comm_rccl: call pq_kkt_partition first");
        if (rank != rank_ || world != world_) throw std::runtime_error("set_comm_rccl: rank / world differ from pq_kkt_partition");
        drop_transport();
        comm_ = rccl::comm_create(id128, rank, world, dev_);
        own_factor_.alloc((size_t)PT_.bmat_off.back() + 1); own_forward_.alloc((size_t)std::max(1, PT_.bvec_off.back())); own_gather_.alloc((size_t)world_ * gather_slot_);
        own_factor_.zero(st_); own_forward_.zero(st_); own_gather_.zero(st_);
        stream_wait(st_);
        xfn_ = nullptr; xbuf_factor_ = own_factor_.p; xbuf_forward_ = own_forward_.p; xbuf_gather_ = own_gather_.p;
        own_norm_.alloc(2); own_norm_.zero(st_); stream_wait(st_); xbuf_norm_ = own_norm_.p; sharded_agreed_ = -1;
        transport_ = Transport::Native;
    }
    double min_abs_pivot() override
    {
        PQ_HIP(hipSetDevice(dev_));
        std::vector<double> h((size_t)N_);
        PQ_HIP(hipMemcpyAsync(h.data(), rdiag_.p, sizeof(double) * (size_t)N_, hipMemcpyDeviceToHost, st_));
        stream_wait(st_);
        double mx = 0.0;
        for (double r : h) mx = std::max(mx, std::fabs(r));
        return mx > 0.0 ? 1.0 / mx : 0.0;
    }
    void native_exchange_calls(int out[3]) const override { for (int q = 0; q < 3; ++q) out[q] = native_calls_[q]; }
    void comm_info(int out[4]) const override
    {
        out[0] = transport_ == Transport::Native ? 2 : (transport_ == Transport::Callback ? 1 : 0);
        out[1] = out[2] = out[3] = -1;
        if (transport_ == Transport::Native && comm_) rccl::comm_info(comm_, out + 1);
    }
    void partition_info(int out[8]) const override
    {
        if (!part_on_) throw std::runtime_error("partition_info: not partitioned");
        int own = 0, sh = 0;
        for (int o : PT_.owner) { own += o == rank_; sh += o < 0; }
        out[0] = own; out[1] = sh; out[2] = (int)PT_.boundary.size(); out[3] = PT_.span_lo[rank_]; out[4] = PT_.span_hi[rank_];
        out[5] = (int)(1000.0 * PT_.work[rank_] / std::max(PT_.total_work, 1.0)); out[6] = (int)(1000.0 * PT_.shared_work / std::max(PT_.total_work, 1.0)); out[7] = world_;
    }

    void print_info() override
    {
        std::printf("substitution schedule: %d single-wave walks, %d supernodes in %d flag-ordered levels above them\n", (int)S_.solve_sub_lo.size(), ntop_solve_, S_.solve_top_nlevels);
        std::printf("substitution top: %d chain walks; factor top: %d chain walks over %d supernodes\n", nwalk_solve_, ntopwalk_, top_nper_);
        std::printf("top of the tree: %d supernodes in %d levels: %d level launches, then %d supernodes in one persistent launch\n", ntop_, S_.top_nlevels, top_l0_, top_nper_);
        if (debug_token("print_levels")) {
            for (int l = 0; l < S_.top_nlevels; ++l) {
                int mf = 0, mw = 0; double fl = 0.0;
                for (int q = S_.top_level_ptr[l]; q < S_.top_level_ptr[l + 1]; ++q) {
                    const int s2 = S_.top_level_sn[q];
                    const int w = S_.sn_first[s2 + 1] - S_.sn_first[s2], f = S_.front_rows_ptr[s2 + 1] - S_.front_rows_ptr[s2];
                    mf = std::max(mf, f); mw = std::max(mw, w); fl += (double)w * f * f;
                }
                std::printf("  top level %2d: %5d fronts, max front %4d, max pivots %3d, sum w f^2 = %.2e;  one-workgroup fronts (f x w) by pivots:", l, S_.top_level_ptr[l + 1] - S_.top_level_ptr[l], mf, mw, fl);
                std::vector<std::pair<int, int>> small;
                for (int q = S_.top_level_ptr[l]; q < S_.top_level_ptr[l + 1]; ++q) if (!is_big(S_.top_level_sn[q])) { const int s2 = S_.top_level_sn[q]; small.push_back({S_.sn_first[s2 + 1] - S_.sn_first[s2], S_.front_rows_ptr[s2 + 1] - S_.front_rows_ptr[s2]}); }
                std::sort(small.rbegin(), small.rend());
                for (size_t q = 0; q < small.size() && q < 4; ++q) std::printf(" %dx%d", small[q].second, small[q].first);
                std::printf("\n");
            }
        }
        for (const SubClass& c : sched_.cls)
            std::printf("subtree walk class: %d subtrees, front capacity %d doubles, %d threads, %d bytes of LDS per workgroup\n", c.nsub, c.cap, c.threads, c.bytes);
        std::printf("sparse multifrontal LDLt (%s ordering): N = %d, nnz(K) = %d, nnz(L) = %lld, supernodes = %d, tree levels = %d (%d subtrees walked by one workgroup each + %d level launches), max front = %d, front storage = %.1f MB\n",
                    S_.ordering, N_, nnzK_, S_.nnzL, S_.nsuper, S_.nlevels, S_.nsub, S_.top_nlevels, S_.max_front, S_.front_doubles * 8.0 / 1e6);
    }

    const double* P_diag_device() const override { return ops_.P_diag(); }
    int n() const override { return n_; }
    int p() const override { return p_; }
    int m() const override { return m_; }
    hipStream_t stream() const override { return st_; }
    int device() const override { return dev_; }
    void set_profiling(int level) override { prof_.enabled = level != 0; }
    void get_profile(int stage, double* total_ms, int* count) override
    {
        if (stage < 0 || stage >= StageProfiler::NSTAGE) throw std::runtime_error("bad stage");
        PQ_HIP(hipSetDevice(dev_));
        prof_.collect(stage, st_, total_ms, count);
    }
    const sparse::Symbolic& symbolic() const { return S_; }

private:
    SparseKKT(const SparseKKT& o, int) : dev_(o.dev_), mode_(o.mode_), nzAA_(o.nzAA_), nzGG_(o.nzGG_), n_(o.n_), p_(o.p_), m_(o.m_), N_(o.N_), nnzK_(o.nnzK_), delta_(o.delta_), S_(o.S_), level_lds_(o.level_lds_), sub_lds_(o.sub_lds_), ntop_(o.ntop_), top_grid_(o.top_grid_), top_lds_(o.top_lds_), top_l0_(o.top_l0_), top_start_(o.top_start_), top_nper_(o.top_nper_), top_persistent_(o.top_persistent_)
    {
        ref_mode_ = o.ref_mode_;
        PQ_HIP(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
        PQ_HIP(hipStreamCreateWithFlags(&st2_, hipStreamNonBlocking));
        PQ_HIP(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming)); PQ_HIP(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
        auto cpd = [&](DBuf<double>& d, const DBuf<double>& s) { d.alloc(s.n ? s.n : 1); if (s.n) PQ_HIP(hipMemcpyAsync(d.p, s.p, s.bytes(), hipMemcpyDeviceToDevice, st_)); };
        auto cpi = [&](DBuf<int>& d, const DBuf<int>& s) { d.alloc(s.n ? s.n : 1); if (s.n) PQ_HIP(hipMemcpyAsync(d.p, s.p, s.bytes(), hipMemcpyDeviceToDevice, st_)); };
        auto cpl = [&](DBuf<long long>& d, const DBuf<long long>& s) { d.alloc(s.n ? s.n : 1); if (s.n) PQ_HIP(hipMemcpyAsync(d.p, s.p, s.bytes(), hipMemcpyDeviceToDevice, st_)); };
        ops_.clone_from(o.ops_, st_);
        build_level_lists(S_.solve_top_level_ptr, S_.solve_top_level_sn, solve_ll_);
        cpd(vals_, o.vals_); cpd(fronts_, o.fronts_); cpd(rdiag_, o.rdiag_);
        xp_.alloc(o.xp_.n); fvec_.alloc(o.fvec_.n);
        build_big_levels(S_.top_level_ptr, S_.top_level_sn, top_big_);
        cpi(diag_pos_, o.diag_pos_); cpi(P_, o.P_); cpi(level_sn_, o.level_sn_); cpi(top_pos_, o.top_pos_); top_flags_.alloc(o.top_flags_.n ? o.top_flags_.n : 1); top_flags_.zero(st_); cpi(solve_level_sn_, o.solve_level_sn_); cpi(solve_top_pos_, o.solve_top_pos_); solve_flags_.alloc(o.solve_flags_.n ? o.solve_flags_.n : 1); solve_flags_.zero(st_); ntop_solve_ = o.ntop_solve_; cpi(solve_pub_, o.solve_pub_); cpi(solve_walk_lo_, o.solve_walk_lo_); cpi(solve_walk_hi_, o.solve_walk_hi_); nwalk_solve_ = o.nwalk_solve_; cpi(solve_child_tp_, o.solve_child_tp_); crec_.alloc(o.crec_.n ? o.crec_.n : 1); if (o.crec_.n) PQ_HIP(hipMemcpyAsync(crec_.p, o.crec_.p, o.crec_.bytes(), hipMemcpyDeviceToDevice, st_)); cpi(top_walk_lo_, o.top_walk_lo_); cpi(top_walk_hi_, o.top_walk_hi_); ntopwalk_ = o.ntopwalk_; top_walk_cap_ = o.top_walk_cap_; cpi(fe_ptr_, o.fe_ptr_); cpi(fe_q_, o.fe_q_); cpi(fe_off_, o.fe_off_); cpi(fe_offp_, o.fe_offp_);
        snrec_.alloc(o.snrec_.n ? o.snrec_.n : 1); if (o.snrec_.n) PQ_HIP(hipMemcpyAsync(snrec_.p, o.snrec_.p, o.snrec_.bytes(), hipMemcpyDeviceToDevice, st_)); cpi(sn_first_, o.sn_first_); cpi(front_rows_ptr_, o.front_rows_ptr_); cpi(front_rows_, o.front_rows_);
        cpi(child_ptr_, o.child_ptr_); cpi(child_, o.child_); cpi(rel_ptr_, o.rel_ptr_); cpi(rel_, o.rel_);
        cpi(mapP_, o.mapP_); cpi(mapA_, o.mapA_); cpi(mapG_, o.mapG_);
        cpi(mapAA_, o.mapAA_); cpi(mapGG_, o.mapGG_); cpi(aa_ptr_, o.aa_ptr_); cpi(aa_q1_, o.aa_q1_); cpi(aa_q2_, o.aa_q2_); cpi(aa_k_, o.aa_k_);
        cpi(gg_ptr_, o.gg_ptr_); cpi(gg_q1_, o.gg_q1_); cpi(gg_q2_, o.gg_q2_); cpi(gg_k_, o.gg_k_);
        cpd(ata_vals_, o.ata_vals_); cpd(zinv_, o.zinv_); rhs_top_.alloc(o.rhs_top_.n ? o.rhs_top_.n : 1);
        cpl(front_off_, o.front_off_);
        info_.alloc(1); info_h_.alloc(1);
        build_full_schedule();
        build_solve_schedule();
        stream_wait(st_);
    }

    // dynamic LDS of the level kernel = largest front of the level that is factored inside LDS
    void compute_level_lds()
    {
        static PerDeviceOnce attr_set;
        attr_set([&] {
            PQ_ATTR_RM(k_front_factor, (LDS_FRONT_DOUBLES + IND_SCRATCH) * (int)sizeof(double));
            PQ_ATTR_RM(k_subtree_factor, (LDS_FRONT_DOUBLES + IND_SCRATCH) * (int)sizeof(double));
            PQ_ATTR_RM(k_subtree_factor_lds, SUBTREE_LDS_BYTES);
            PQ_ATTR_RM(k_subtree_factor_pk, SUBTREE_LDS_BYTES);
            PQ_ATTR_RM(k_top_factor, (LDS_FRONT_DOUBLES + IND_SCRATCH) * (int)sizeof(double));
        });
        level_lds_.assign(S_.top_nlevels, 0);
        for (int l = 0; l < S_.top_nlevels; ++l) {
            long long mx = 0;
            for (int q = S_.top_level_ptr[l]; q < S_.top_level_ptr[l + 1]; ++q) {
                const int s = S_.top_level_sn[q];
                const long long f = S_.front_rows_ptr[s + 1] - S_.front_rows_ptr[s];
                mx = std::max(mx, front_lds_doubles(f, S_.sn_first[s + 1] - S_.sn_first[s]));
            }
            level_lds_[l] = ((int)mx + IND_SCRATCH) * (int)sizeof(double);
        }
        {
            const long long f = S_.sub_max_front;
            sub_lds_ = (int)((std::min<long long>(f * f, LDS_FRONT_DOUBLES) + IND_SCRATCH) * (long long)sizeof(double));  // (a front that does not fit may still stage its panel: the cap covers it)
        }
        // single-launch top of the tree: possible when no top front needs the multi-launch dense path
        ntop_ = (int)S_.top_level_sn.size();
        bool any_big = false;
        long long top_mx = 0;
        for (int s : S_.top_level_sn) {
            const int w = S_.sn_first[s + 1] - S_.sn_first[s];
            const long long f = S_.front_rows_ptr[s + 1] - S_.front_rows_ptr[s];
            if (is_big(s)) any_big = true;
            top_mx = std::max(top_mx, front_lds_doubles(f, w));
        }
        top_lds_ = ((int)top_mx + IND_SCRATCH) * (int)sizeof(double);
        top_grid_ = std::min(ntop_, 224);
        {   // arithmetic of the one-workgroup fronts: the reference's, term by term, unless some front of the tree runs on the matrix cores (see ref_quot)
            bool multi = false;
            for (int s = 0; s + 1 < (int)S_.sn_first.size() && !multi; ++s) multi = is_big(s);
            ref_mode_ = multi ? PQ_REF_MODE_BIG : PQ_REF_MODE;
            if (const char* e = debug_token("ref_mode")) ref_mode_ = std::atoi(e) == PQ_REF_MODE ? PQ_REF_MODE : PQ_REF_MODE_BIG;  // debugging aid: PIQP_AMD_DEBUG=ref_mode=<m>
        }
        // measured: worth it for the factorisation when every top supernode gets its own workgroup; the substitution fronts are too
        // cheap to pay for agent-scope release / acquire per supernode, they stay on level launches
        top_persistent_ = ntop_ > 0 && ntop_ <= 1024 && !any_big && !debug_token("top_levels");
        // the levels from top_l0_ on (at most 1024 supernodes, none on the dense multi-launch path) go into the persistent launch
        top_l0_ = S_.top_nlevels;
        if (!debug_token("top_levels")) {
            for (int l = S_.top_nlevels - 1; l >= 0; --l) {
                bool big = false;
                for (int q = S_.top_level_ptr[l]; q < S_.top_level_ptr[l + 1]; ++q) {
                    const int s = S_.top_level_sn[q];
                    if (is_big(s)) big = true;
                }
                if (big || ntop_ - S_.top_level_ptr[l] > 1024) break;
                top_l0_ = l;
            }
        }
        top_start_ = top_l0_ < S_.top_nlevels ? S_.top_level_ptr[top_l0_] : ntop_;
        top_nper_ = ntop_ - top_start_;
        if (top_nper_ < 2) { top_l0_ = S_.top_nlevels; top_start_ = ntop_; top_nper_ = 0; }  // a single supernode gains nothing
        // chains of the persistent part -> walks (k_top_factor_walk) when two packed fronts of the largest one fit the LDS
        ntopwalk_ = 0; top_walk_cap_ = 0;
        if (top_nper_ > 0 && !debug_token("top_no_walks")) {
            std::vector<int> pos(S_.nsuper, -1);
            for (int q = top_start_; q < ntop_; ++q) pos[S_.top_level_sn[q]] = q;
            std::vector<std::pair<int, std::pair<int, int>>> walks;
            long long fm = 0;
            for (int s = 0; s < S_.nsuper;) {
                if (pos[s] < 0) { ++s; continue; }
                int hi = s;
                while (hi + 1 < S_.nsuper && pos[hi + 1] >= 0 && S_.sn_parent[hi] == hi + 1) ++hi;
                for (int t = s; t <= hi; ++t) fm = std::max<long long>(fm, S_.front_rows_ptr[t + 1] - S_.front_rows_ptr[t]);
                walks.push_back({pos[hi], {s, hi}});
                s = hi + 1;
            }
            const long long capP = fm * (fm + 1) / 2;
            if (2 * capP * 8 <= SUBTREE_LDS_BYTES && (int)walks.size() < top_nper_) {
                std::sort(walks.begin(), walks.end());
                std::vector<int> lo, hi;
                for (const auto& wk : walks) { lo.push_back(wk.second.first); hi.push_back(wk.second.second); }
                upload_vec(top_walk_lo_, lo, st_); upload_vec(top_walk_hi_, hi, st_);
                ntopwalk_ = (int)walks.size(); top_walk_cap_ = (int)capP;
                static PerDeviceOnce attr_set;
                attr_set([&] { PQ_ATTR_RM(k_top_factor_walk, SUBTREE_LDS_BYTES); });
            }
        }
    }

    FrontMeta meta() const { return FrontMeta{snrec_.p, front_rows_.p, child_.p, rel_.p, fe_ptr_.p, fe_q_.p, fe_off_.p, vals_.p, crec_.p}; }

    // numeric phase of the factorisation / substitution on handle-owned buffers only (so that they can be recorded as graphs)
    void factor_numeric(const FrontMeta& M)
    {
        PQ_ZONE("piqp_amd::SparseKKT::factor_numeric");  // sparse/ldlt.hpp:109 piqp::LDLt::factorize_numeric
        // zero-fill, own entries and step counters of the multi-workgroup fronts need nothing from the subtrees: on the second stream, next to the subtree walks
        const bool pre = top_big_.total > 0 && !no_fork_;
        if (pre) {
            PQ_HIP(hipEventRecord(ev_fork_, st_)); PQ_HIP(hipStreamWaitEvent(st2_, ev_fork_, 0));
            big_prepare(M, top_big_, st2_);
            PQ_HIP(hipEventRecord(ev_join_, st2_));
        }
        factor_subtrees(M, sched_);
        if (pre) PQ_HIP(hipStreamWaitEvent(st_, ev_join_, 0));
        // wide lower levels: one launch per level; the narrow levels near the root (<= 1024 supernodes in total): one persistent launch
        factor_levels(M, S_.top_level_ptr, level_sn_.p, level_lds_, top_big_, top_l0_, pre);
        if (top_nper_ > 0) {
            // flags carry the number of the factorisation that set them (no memset in between; a recorded graph replays fixed arguments,
            // so there they are zeroed and the epoch stays 1)
            int epoch = 1;
            if (factor_epoch_ >= 2000000000) { PQ_HIP(hipMemsetAsync(top_flags_.p, 0, sizeof(int) * (2 * (size_t)ntop_ + 1), st_)); factor_epoch_ = 0; }
            epoch = ++factor_epoch_;
            if (ntopwalk_ > 0)
                PQ_LAUNCH_RM(k_top_factor_walk, dim3(std::min(ntopwalk_, 224)), dim3(top_threads()), 2 * (size_t)top_walk_cap_ * sizeof(double), st_, M, fronts_.p, vals_.p, fe_offp_.p,
                                   top_walk_lo_.p, top_walk_hi_.p, ntopwalk_, top_pos_.p, top_start_, top_walk_cap_, top_flags_.p, top_flags_.p + 2 * ntop_, rdiag_.p, info_.p, epoch);
            else
                PQ_LAUNCH_RM(k_top_factor, dim3(std::min(top_nper_, 224)), dim3(top_threads()), top_lds_, st_, M, fronts_.p, level_sn_.p + top_start_, top_nper_, top_pos_.p, top_start_,
                                   top_flags_.p, top_flags_.p + 2 * ntop_, rdiag_.p, info_.p, epoch);
        }
    }
    void solve_numeric(const FrontMeta& M)
    {
        // the sweeps run on their own (finer) partition of the tree: S_.solve_*
        const int nt = ntop_solve_;
        subtree_fwd(M, solve_sched_);
        // the top of the tree: one launch per sweep when every top front fits the single-wave kernels, else one launch per level
        bool wave_top = nt > 1 && !debug_token("top_levels_solve");
        if (wave_top) for (int s2 : S_.solve_top_level_sn) if (S_.front_rows_ptr[s2 + 1] - S_.front_rows_ptr[s2] > 128) { wave_top = false; break; }
        // flags carry the number of the solve that set them: no memset between solves (a recorded graph replays fixed arguments, so there
        // the flags are zeroed and the epoch stays 1)
        int epoch = 1;
        solve_err_ptr_ = nullptr; solve_epoch_used_ = 0;
        if (wave_top) {
            if (solve_epoch_ >= 2000000000) { PQ_HIP(hipMemsetAsync(solve_flags_.p, 0, sizeof(int) * (2 * (size_t)nt + 1), st_)); solve_epoch_ = 0; }
            epoch = ++solve_epoch_;
            solve_err_ptr_ = solve_flags_.p + 2 * nt; solve_epoch_used_ = epoch;
            static const char* dbg_ts = debug_token("dbg_ts");
            DBuf<long long> ts;
            if (dbg_ts) { ts.alloc(3 * (size_t)nwalk_solve_); long long* pp = ts.p; PQ_HIP(hipMemsetAsync(ts.p, 0, ts.bytes(), st_)); PQ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_ts), &pp, sizeof(pp))); }
            hipLaunchKernelGGL(k_subtree_fwd_wave<true>, dim3(nwalk_solve_), dim3(64), 0, st_, M, fronts_.p, solve_walk_lo_.p, solve_walk_hi_.p, xp_.p, fvec_.p, solve_top_pos_.p, solve_flags_.p,
                               solve_flags_.p + 2 * nt, solve_child_tp_.p, epoch);
            if (dbg_ts) {
                stream_wait(st_);
                std::vector<long long> h(3 * (size_t)nwalk_solve_);
                PQ_HIP(hipMemcpy(h.data(), ts.p, ts.bytes(), hipMemcpyDeviceToHost));
                long long* pp = nullptr; PQ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_ts), &pp, sizeof(pp)));
                if (FILE* fo = std::fopen(dbg_ts, "wb")) {
                    const int nw = nwalk_solve_;
                    std::fwrite(&nw, sizeof(int), 1, fo);
                    std::fwrite(h.data(), sizeof(long long), h.size(), fo);
                    std::fwrite(S_.solve_walk_lo.data(), sizeof(int), nw, fo); std::fwrite(S_.solve_walk_hi.data(), sizeof(int), nw, fo);
                    std::fclose(fo);
                }
            }
        } else fwd_levels(M, solve_ll_);
        bool fuse_scale = wave_top;
        for (const SubClass& c : solve_sched_.cls) if (c.fmax > 128) fuse_scale = false;
        const double* rd = fuse_scale ? rdiag_.p : nullptr;  // the diagonal solve rides in the backward kernels when they are all single-wave
        if (!fuse_scale) hipLaunchKernelGGL(k_scale, g1(N_), dim3(256), 0, st_, N_, rdiag_.p, xp_.p);
        if (wave_top) {
            hipLaunchKernelGGL(k_subtree_bwd_wave<true>, dim3(nwalk_solve_), dim3(64), 0, st_, M, fronts_.p, solve_walk_lo_.p, solve_walk_hi_.p, xp_.p, bwd_red_thr(), nwalk_solve_, solve_top_pos_.p,
                               solve_flags_.p + nt, solve_flags_.p + 2 * nt, solve_pub_.p, epoch, rd);
            // a wait that gave up left the epoch in the error slot: k_perm_scatter poisons the solution (solve_err_ptr_)
        } else bwd_levels(M, solve_ll_);
        subtree_bwd(M, solve_sched_, rd);
    }
    // data crosses ranks: the stream is drained, the caller's collective runs (pq_exchange_fn), then the stream continues
    void exchange(int which)
    {
        // a one-rank group has nothing to exchange; PIQP_AMD_EXCHANGE_WORLD1=1 still goes through the callback (a 1-GPU box can then
        // exercise the caller's RCCL transport end to end)
        static const bool force1 = std::getenv("PIQP_AMD_EXCHANGE_WORLD1") != nullptr;
        if (world_ == 1 && !(force1 && transport_ != Transport::None)) return;
        if (transport_ == Transport::Native) {
            // native transport: stream-ordered behind the pack kernel, the unpack kernel follows on the same stream
            if (which == 0) rccl::all_reduce_sum(comm_, xbuf_factor_, (size_t)PT_.bmat_off.back() + 1, st_);
            else if (which == 1) rccl::all_reduce_sum(comm_, xbuf_forward_, (size_t)std::max(1, PT_.bvec_off.back()), st_);
            else if (which == 3) rccl::all_reduce_max(comm_, xbuf_norm_, 1, st_);
            else rccl::all_gather(comm_, xbuf_gather_, (size_t)gather_slot_, rank_, st_);
            if (which < 3) ++native_calls_[which];
            return;
        }
        if (transport_ != Transport::Callback || !xfn_) throw std::runtime_error("partitioned backend used before pq_kkt_set_exchange / pq_kkt_set_comm_rccl");
        stream_wait(st_);
        if (xfn_(xuser_, which) != 0) throw std::runtime_error("exchange callback failed");
    }
    // ---- subtree schedule: one launch of all subtree walks with the dynamic LDS of the largest walk (two fronts of its largest front order)
    void build_sub_schedule(const std::vector<int>& subs, SubSchedule& out)
    {
        out.cls.clear();
        // ONE class sized for the largest subtree.  Measured and removed (round 1): several LDS size classes -- their launches run back to back
        // and the tails add up (C5 backend solve 1.4 -> 2.1 ms with four classes) -- and staging the per-subtree metadata in LDS, which costs
        // more occupancy than the latency it removes (C5 factor 2.1 -> 3.9 ms at 54 KB per workgroup).  Round 2 re-measured two classes (packed fronts <= 40 KB:
        // four walks per CU; the rest) launched side by side on the factorisation's two streams: C3 0.51 -> 0.56 ms, CONT-201 unchanged -- still one class
        static const int limits[] = {SUBTREE_LDS_BYTES};
        const int ncls = 1;
        std::vector<std::vector<int>> members(ncls);
        struct Need { int fm, ent, rel, sn; };
        std::vector<Need> need(subs.size());
        auto bytes_of = [&](long long cap) -> long long { return 2 * cap * 8; };
        for (size_t i = 0; i < subs.size(); ++i) {
            const int lo = S_.sub_lo[subs[i]], hi = S_.sub_hi[subs[i]];
            Need q{0, S_.fe_ptr[hi + 1] - S_.fe_ptr[lo], S_.rel_ptr[hi + 1] - S_.rel_ptr[lo], hi - lo + 1};
            for (int t = lo; t <= hi; ++t) q.fm = std::max(q.fm, S_.front_rows_ptr[t + 1] - S_.front_rows_ptr[t]);
            need[i] = q;
            const long long b = bytes_of((long long)q.fm * q.fm);
            int c = 0;
            while (c + 1 < ncls && b > limits[c]) ++c;
            members[c].push_back((int)i);
        }
        for (int c = 0; c < ncls; ++c) {
            if (members[c].empty()) continue;
            SubClass k;
            Need mx{0, 0, 0, 0};
            std::vector<int> lo, hi;
            // longest subtrees first: the launch is a handful of rounds of workgroups, so the last round should hold the short walks
            {
                std::vector<double> work(subs.size(), 0.0);
                for (int i : members[c]) {
                    for (int t = S_.sub_lo[subs[i]]; t <= S_.sub_hi[subs[i]]; ++t) {
                        const double w = S_.sn_first[t + 1] - S_.sn_first[t], f = S_.front_rows_ptr[t + 1] - S_.front_rows_ptr[t], u = f - w;
                        work[i] += w * w * w / 3.0 + w * w * u + w * u * u + 2.0 * f * f + 3000.0;
                    }
                }
                std::stable_sort(members[c].begin(), members[c].end(), [&](int a, int b) { return work[a] > work[b]; });
            }
            for (int i : members[c]) {
                mx.fm = std::max(mx.fm, need[i].fm); mx.ent = std::max(mx.ent, need[i].ent); mx.rel = std::max(mx.rel, need[i].rel); mx.sn = std::max(mx.sn, need[i].sn);
                lo.push_back(S_.sub_lo[subs[i]]); hi.push_back(S_.sub_hi[subs[i]]);
            }
            k.nsub = (int)lo.size();
            k.cap = mx.fm * mx.fm;
            k.fmax = mx.fm;
            long long b = bytes_of(k.cap);
            k.bytes = (int)b;
            k.lds_walk = b <= SUBTREE_LDS_BYTES;
            // a walk keeps only the lower triangles of its two fronts when that raises the occupancy: half the LDS (C3: two workgroups per CU instead of one)
            // (PIQP_AMD_DEBUG=subtree_packed=0 / 1 forces it off / on: bitwise-consistency test)
            {
                static const char* pe = debug_token("subtree_packed");
                // default: packed whenever that lets more walks share a CU (at most eight 256-thread workgroups fit by waves)
                const long long bp = 2LL * ((long long)mx.fm * (mx.fm + 1) / 2) * 8;
                const auto per_cu = [](long long bytes) { return std::min<long long>(8, (160 * 1024) / std::max<long long>(bytes, 1)); };
                k.packed = k.lds_walk && (pe ? pe[0] == '1' : per_cu(bp) > per_cu(b));
                if (k.packed) { k.cap = (mx.fm * (mx.fm + 1)) / 2; b = 2LL * k.cap * 8; k.bytes = (int)b; }
            }
            // a workgroup that has the CU to itself (LDS) gets eight waves instead of four: measured 1.08 -> 1.04 ms on C3; with several
            // workgroups per CU more threads only add barrier cost (C5: 1.45 -> 2.0 ms)
            k.threads = b > 80 * 1024 ? 512 : (k.packed && b > 52 * 1024 ? 384 : SUB_THREADS);  // two packed walks per CU: 384 (C3 0.82 -> 0.79 ms)
            upload_vec(k.lo, lo, st_); upload_vec(k.hi, hi, st_);
            out.cls.push_back(std::move(k));
        }
    }
    static int top_threads() { return 512; }  // measured: 512 beats 256 (C3 1.04 -> 0.98 ms)
    static int bwd_red_thr() { return 6; }    // 3 before the split-wave product existed; 6: C3 solve 0.31 -> 0.29 ms
    void build_full_schedule()
    {
        std::vector<int> all(S_.nsub);
        for (int k = 0; k < S_.nsub; ++k) all[k] = k;
        build_sub_schedule(all, sched_);
    }
    void build_solve_schedule()
    {
        solve_sched_.cls.clear();
        if (S_.solve_sub_lo.empty()) return;
        SubClass k;
        k.nsub = (int)S_.solve_sub_lo.size();
        k.fmax = S_.solve_sub_max_front;
        upload_vec(k.lo, S_.solve_sub_lo, st_); upload_vec(k.hi, S_.solve_sub_hi, st_);
        solve_sched_.cls.push_back(std::move(k));
    }
    void subtree_fwd(const FrontMeta& M, const SubSchedule& sc)
    {
        for (const SubClass& c : sc.cls) {
            if (c.fmax <= 128) hipLaunchKernelGGL(k_subtree_fwd_wave<false>, dim3(c.nsub), dim3(64), 0, st_, M, fronts_.p, c.lo.p, c.hi.p, xp_.p, fvec_.p, (const int*)nullptr, (int*)nullptr, (int*)nullptr, (const int*)nullptr, 0);
            else hipLaunchKernelGGL(k_subtree_fwd, dim3(c.nsub), dim3(SUB_SOLVE_THREADS), 0, st_, M, fronts_.p, c.lo.p, c.hi.p, xp_.p, fvec_.p);
        }
    }
    void subtree_bwd(const FrontMeta& M, const SubSchedule& sc, const double* rd = nullptr)
    {
        for (const SubClass& c : sc.cls) {
            if (c.fmax <= 128) hipLaunchKernelGGL(k_subtree_bwd_wave<false>, dim3(c.nsub), dim3(64), 0, st_, M, fronts_.p, c.lo.p, c.hi.p, xp_.p, bwd_red_thr(), 0, (const int*)nullptr, (int*)nullptr, (int*)nullptr, (const int*)nullptr, 0, rd);
            else hipLaunchKernelGGL(k_subtree_bwd, dim3(c.nsub), dim3(SUB_SOLVE_THREADS), 0, st_, M, fronts_.p, c.lo.p, c.hi.p, xp_.p, fvec_.p);
        }
    }
    void factor_subtrees(const FrontMeta& M, const SubSchedule& sc)
    {
        for (const SubClass& c : sc.cls) {
            if (c.lds_walk && c.packed)
                PQ_LAUNCH_RM(k_subtree_factor_pk, dim3(c.nsub), dim3(c.threads), c.bytes, st_, M, fronts_.p, vals_.p, fe_offp_.p, c.lo.p, c.hi.p, c.cap, rdiag_.p, info_.p);
            else if (c.lds_walk)
                PQ_LAUNCH_RM(k_subtree_factor_lds, dim3(c.nsub), dim3(c.threads), c.bytes, st_, M, fronts_.p, vals_.p, fe_ptr_.p, fe_q_.p, fe_off_.p, c.lo.p, c.hi.p, c.cap, rdiag_.p,
                                   info_.p);
            else
                PQ_LAUNCH_RM(k_subtree_factor, dim3(c.nsub), dim3(SUB_THREADS), sub_lds_, st_, M, fronts_.p, c.lo.p, c.hi.p, rdiag_.p, info_.p);
        }
    }
    // ---- big fronts of a level schedule, grouped by level for the batched dense path
    bool is_big(int s) const
    {
        const int f = S_.front_rows_ptr[s + 1] - S_.front_rows_ptr[s], w = S_.sn_first[s + 1] - S_.sn_first[s];
        if (no_big_) return false;
        if (big_front(f, w)) return true;
        // a panel front alone does not pull a small top out of its single persistent launch (k_top_factor stages the panel itself): measured on AUG3DCQP
        // (41 top supernodes, two panel fronts, none big): factorisation 0.45 ms persistent, 0.61 ms with eight level launches around the two fronts
        if (tree_has_big_ < 0) {
            tree_has_big_ = (int)S_.top_level_sn.size() > 1024 ? 1 : 0;
            for (int t : S_.top_level_sn) if (big_front(S_.front_rows_ptr[t + 1] - S_.front_rows_ptr[t], S_.sn_first[t + 1] - S_.sn_first[t])) { tree_has_big_ = 1; break; }
        }
        return tree_has_big_ == 1 && panel_front(f, w);
    }
    struct BigLevels {
        std::vector<int> ptr, rounds, ndense, npanel, panel_lds, lmaxf, multi_rows;  // multi_rows[l] > 0: the level's multi-panel fronts take their update matrices in one pass (most update rows)  // level l of the schedule -> jobs [ptr[l], ptr[l + 1]); children of its widest fan-in; jobs by kind; LDS of k_front_panel
        std::vector<std::vector<int>> rows_below;  // per level, per panel: most rows below the diagonal block over the level's fronts (0 = none)
        int total = 0, max_f = 0, max_own = 0;
        DBuf<int> list, job_of;  // job_of[s] = index into jobs, -1 for the fronts one workgroup handles alone
        DBuf<dense::FrontJob> jobs;
        DBuf<double> scratch;  // pack + D of every front of the widest level
        DBuf<int> cnt;         // per job 8 x FRONT_CNT_PANELS step counters (dense::FrontJob::cnt), zeroed at the start of every factorisation
    };
    void build_big_levels(const std::vector<int>& ptr, const std::vector<int>& sn, BigLevels& B)
    {
        B.ptr.assign(1, 0); B.rounds.clear(); B.rows_below.clear(); B.ndense.clear(); B.npanel.clear(); B.panel_lds.clear(); B.lmaxf.clear(); B.multi_rows.clear(); B.total = 0; B.max_f = 0; B.max_own = 0;
        std::vector<int> list;
        int widest = 0;
        for (int l = 0; l + 1 < (int)ptr.size(); ++l) {
            int rounds = 0, nd = 0, np = 0, lf = 0;
            long long plds = 0;
            std::vector<int> rb;
            for (int q = ptr[l]; q < ptr[l + 1]; ++q) {
                const int s = sn[q];
                if (!is_big(s)) continue;
                list.push_back(s);
                const int w = S_.sn_first[s + 1] - S_.sn_first[s], f = S_.front_rows_ptr[s + 1] - S_.front_rows_ptr[s];
                if (panel_front(f, w)) { ++np; plds = std::max(plds, (long long)f * w); } else ++nd;
                rounds = std::max(rounds, S_.child_ptr[s + 1] - S_.child_ptr[s]);
                B.max_f = std::max(B.max_f, f); B.max_own = std::max(B.max_own, S_.fe_ptr[s + 1] - S_.fe_ptr[s]); lf = std::max(lf, f);
                for (int k = 0, pn = 0; k < w; k += dense::FACTOR_NB, ++pn) {
                    if ((int)rb.size() <= pn) rb.push_back(0);
                    rb[pn] = std::max(rb[pn], f - k - std::min(dense::FACTOR_NB, w - k));
                }
            }
            B.ptr.push_back((int)list.size()); B.rounds.push_back(rounds); B.rows_below.push_back(rb); B.lmaxf.push_back(lf);
            {   // one pass over the update matrices of the level's multi-panel fronts (dense::launch_front_updates_multi) where the level is too wide for the
                // one-launch panel step (that path keeps its per-panel tiles: it is bound by the chain of its few fronts, not by traffic)
                // MEASURED SLOWER and therefore opt-in (PIQP_AMD_DEBUG=multi_update; bitwise equal, variant test): wide C3 variant 6.56 -> 6.97 ms, 1500-window
                // variant 53.3 -> 57.3 ms per factorisation -- the trailing updates of these levels are not bound by HBM traffic (the tiles of a panel launch
                // come from the L2 / Infinity Cache the previous launch left them in), and a tile's panels one after the other in one workgroup expose the
                // read-modify-write of every panel
                static const bool off = debug_token("multi_update") == nullptr;
                const int nb0 = (int)list.size() - B.ptr[l];
                const int T0 = rb.empty() ? 0 : (rb[0] + 127) / 128;
                const bool step_ok = nb0 > 0 && T0 > 0 && T0 <= 16 && (long long)nb0 * T0 * (T0 + 1) <= 256;
                int mr = 0;
                if (!off && !step_ok && rb.size() >= 2)
                    for (int q = B.ptr[l]; q < (int)list.size(); ++q) {
                        const int s2 = list[q], w2 = S_.sn_first[s2 + 1] - S_.sn_first[s2], f2 = S_.front_rows_ptr[s2 + 1] - S_.front_rows_ptr[s2];
                        if (!panel_front(f2, w2) && w2 > dense::FACTOR_NB && f2 > w2) mr = std::max(mr, f2 - w2);
                    }
                B.multi_rows.push_back(mr);
            }
            B.ndense.push_back(nd); B.npanel.push_back(np); B.panel_lds.push_back((int)((plds + IND_SCRATCH) * (long long)sizeof(double)));
            widest = std::max(widest, B.ptr[l + 1] - B.ptr[l]);
            if (B.ptr[l + 1] - B.ptr[l] > 65535) throw std::runtime_error("sparse backend: more than 65535 multi-workgroup fronts on one level of the assembly tree");
        }
        B.total = (int)list.size();
        if (B.total == 0) return;
        // scratch of a level: D (FACTOR_NB doubles) for every front, the operand pack of the panel solve for the dense-path fronts only
        // (D of EVERY panel of a front is kept until the level is done: w doubles rounded up to the panel width)
        auto dpad = [&](int s2) { const int w2 = S_.sn_first[s2 + 1] - S_.sn_first[s2]; return (size_t)std::max(1, (w2 + dense::FACTOR_NB - 1) / dense::FACTOR_NB) * dense::FACTOR_NB; };
        size_t need = 0;
        std::vector<size_t> dtot(B.ndense.size(), 0);
        for (size_t l = 0; l < B.ndense.size(); ++l) {
            for (int q = B.ptr[l]; q < B.ptr[l + 1]; ++q) dtot[l] += dpad(list[q]);
            need = std::max(need, dtot[l] + (size_t)B.ndense[l] * dense::FACTOR_PACK_DOUBLES);
        }
        B.scratch.alloc(need);
        (void)widest;
        std::vector<dense::FrontJob> jobs(B.total);
        B.cnt.alloc((size_t)B.total * dense::FRONT_CNT_INTS * dense::FRONT_CNT_PANELS);
        for (int l = 0; l + 1 < (int)B.ptr.size(); ++l) {
            double* packs = B.scratch.p + dtot[l];
            int nd = 0;
            size_t doff = 0;
            for (int q = B.ptr[l]; q < B.ptr[l + 1]; ++q) {
                const int s = list[q];
                dense::FrontJob& j = jobs[q];
                j.F = fronts_.p + S_.front_off[s];
                j.f = S_.front_rows_ptr[s + 1] - S_.front_rows_ptr[s]; j.w = S_.sn_first[s + 1] - S_.sn_first[s]; j.first = S_.sn_first[s];
                j.kind = panel_front(j.f, j.w) ? 1 : 0;
                j.multi = (B.multi_rows[l] > 0 && j.kind == 0 && j.w > dense::FACTOR_NB && j.f > j.w) ? 1 : 0;
                j.dvec = B.scratch.p + doff; doff += dpad(s);
                j.pack = j.kind == 0 ? packs + (size_t)(nd++) * dense::FACTOR_PACK_DOUBLES : nullptr;
                j.cnt = B.cnt.p + (size_t)q * dense::FRONT_CNT_INTS * dense::FRONT_CNT_PANELS;
            }
        }
        upload_vec(B.list, list, st_);
        {
            std::vector<int> jo(S_.nsuper ? S_.nsuper : 1, -1);
            for (int q = 0; q < B.total; ++q) jo[list[q]] = q;
            upload_vec(B.job_of, jo, st_);
        }
        B.jobs.alloc(jobs.size());
        PQ_HIP(hipMemcpyAsync(B.jobs.p, jobs.data(), jobs.size() * sizeof(dense::FrontJob), hipMemcpyHostToDevice, st_));
        stream_wait(st_);
    }
    // one launch per level of a (possibly filtered) level schedule for the fronts one workgroup factors; the level's big fronts then go through
    // the dense multi-workgroup kernels together: children merged (fixed order), then the blocked partial LDLt panel by panel
    // zero-fill, own K entries and step counters of the multi-workgroup fronts of a level schedule (nothing here depends on the children)
    // (the inverse row map of the first child needs 4 f bytes of LDS next to the 8 KB column list)
    static int fuse_first_child(const BigLevels& B)
    {
        static const bool off = debug_token("no_fused_first_child") != nullptr;  // bitwise variant test
        return (!off && B.max_f <= 12000) ? 1 : 0;
    }
    void big_prepare(const FrontMeta& M, const BigLevels& B, hipStream_t s)
    {
        if (B.total <= 0) return;
        PQ_HIP(hipMemsetAsync(B.cnt.p, 0, (size_t)B.total * dense::FRONT_CNT_INTS * dense::FRONT_CNT_PANELS * sizeof(int), s));
        for (int q0 = 0; q0 < B.total; q0 += 65535) {  // (grid.y limit)
            const int nq = std::min(65535, B.total - q0);
            hipLaunchKernelGGL(k_big_zero, dim3((unsigned)std::min<long long>(256, ((long long)B.max_f * B.max_f + 255) / 256), nq), dim3(256), 0, s, M, fronts_.p, B.list.p + q0, fuse_first_child(B));
            if (B.max_own > 0) hipLaunchKernelGGL(k_big_assemble, dim3(std::min(64, (B.max_own + 255) / 256), nq), dim3(256), 0, s, M, fronts_.p, B.list.p + q0, fuse_first_child(B));
        }
    }
    void factor_levels(const FrontMeta& M, const std::vector<int>& ptr, const int* sn_dev, const std::vector<int>& lds, const BigLevels& B, int lend = 1 << 30, bool prepared = false)
    {
        if (!prepared) big_prepare(M, B, st_);
        for (int l = 0; l + 1 < (int)ptr.size() && l < lend; ++l) {
            const int cnt = ptr[l + 1] - ptr[l];
            if (cnt <= 0) continue;
            const int nbig = B.total > 0 ? B.ptr[l + 1] - B.ptr[l] : 0;
            // (column c of a front belongs to workgroup c mod G, an entry receives its contributions in child order whatever G is: the levels with a handful
            // of fronts take a wider grid)
            // (one or two fronts of 1500+ rows: 216 MB per level through 256 workgroups of four waves ran at 1.1 TB/s)
            static const char* ea_wide = debug_token("extend_add_wide");
            const int ea_big = ea_wide ? std::atoi(ea_wide) : 1024;
            const int ea_grid = debug_token("extend_add_grid64") ? 64 : (nbig <= 2 && B.total > 0 && B.lmaxf[l] >= 1536 && ea_big > 0 ? ea_big : (nbig <= 4 ? 256 : (nbig <= 16 ? 128 : 64)));
            if (nbig > 0 && B.rounds[l] > 0) hipLaunchKernelGGL(k_big_extend_add, dim3(ea_grid, nbig), dim3(256), fuse_first_child(B) ? B.lmaxf[l] * (int)sizeof(int) : 0, st_, M, fronts_.p, B.list.p + B.ptr[l], fuse_first_child(B));
            const bool small = cnt > nbig || (nbig > 0 && B.npanel[l] > 0);
            // the fronts one workgroup handles and the big fronts' first diagonal blocks + panels are independent: side by side on two streams,
            // joined before the trailing updates (which also carry the panel fronts' Schur complements)
            const bool fork = small && nbig > 0 && B.ndense[l] > 0 && !no_fork_;
            hipStream_t ss = st_;
            if (fork) { PQ_HIP(hipEventRecord(ev_fork_, st_)); PQ_HIP(hipStreamWaitEvent(st2_, ev_fork_, 0)); ss = st2_; }
            // (a level that holds panel fronts -- 130-220 rows, up to 128 pivots at 0.3 us each in one workgroup's LDS pivot loop, the longest item of ten levels
            // of CONT-201 -- runs its one-workgroup launch with eight waves: two per SIMD hide each other's LDS round trips; the work of an entry does not
            // depend on the thread that does it)
            static const bool ff256 = debug_token("front_factor_256") != nullptr;
            // (and so does every level with at most one front per CU: BOYD1 factorisation 0.355 -> 0.343 ms, C3 unchanged)
            const int ff_threads = (!ff256 && ((nbig > 0 && B.npanel[l] > 0) || cnt <= 256)) ? 1024 : 256;
            if (small)
                PQ_LAUNCH_RM(k_front_factor, dim3(cnt), dim3(ff_threads), std::max(lds[l], nbig > 0 ? B.panel_lds[l] : 0), ss, M, fronts_.p, sn_dev + ptr[l], B.total > 0 ? B.job_of.p : (const int*)nullptr,
                                   B.jobs.p, rdiag_.p, info_.p);
            if (fork) PQ_HIP(hipEventRecord(ev_join_, st2_));
            if (nbig <= 0) continue;
            static const bool joint_updates = debug_token("front_joint_updates") != nullptr;
            for (int pn = 0; pn < (int)B.rows_below[l].size(); ++pn) {
                // The big fronts' own panel step needs nothing from the second stream: diagonal block, panel rows and trailing update -- in one launch where the
                // level is few enough fronts for half / quarter tiles -- run before the join; only the Schur complements of the panel fronts (factored by the
                // one-workgroup launch next door, single panel) wait for it.  (One update launch for both kinds behind the join left the big fronts' update
                // waiting for the 45-65 us pivot loops of the panel fronts at ten levels of CONT-201.)
                const bool split = fork && pn == 0 && B.npanel[l] > 0 && B.ndense[l] > 0 && !joint_updates;
                bool done_big = false;
                if (B.ndense[l] > 0 && (B.npanel[l] == 0 || split || pn > 0) && B.multi_rows[l] == 0)  // (a panel front has one panel: nothing of its kind beyond pn = 0)
                    done_big = dense::launch_front_panel_step(B.jobs.p + B.ptr[l], nbig, pn, B.rows_below[l][pn], info_.p, rdiag_.p, st_);
                if (!done_big) {
                    if (B.ndense[l] > 0) dense::launch_front_diag_panels(B.jobs.p + B.ptr[l], nbig, pn, B.rows_below[l][pn], info_.p, rdiag_.p, st_, true);
                    if (split) dense::launch_front_updates(B.jobs.p + B.ptr[l], nbig, pn, B.rows_below[l][pn], st_, 0);
                }
                if (fork && pn == 0) PQ_HIP(hipStreamWaitEvent(st_, ev_join_, 0));
                if (split) dense::launch_front_updates(B.jobs.p + B.ptr[l], nbig, pn, B.rows_below[l][pn], st_, 1);
                else if (!done_big) dense::launch_front_updates(B.jobs.p + B.ptr[l], nbig, pn, B.rows_below[l][pn], st_);
            }
            if (B.multi_rows[l] > 0) dense::launch_front_updates_multi(B.jobs.p + B.ptr[l], nbig, B.multi_rows[l], st_);
        }
    }
    // Substitution by levels: the fronts of a level with at most 128 rows go through the single-wave kernels (one wave per front, vector in
    // registers: the same kernels as the subtree walk with lo = hi = the supernode), the wider ones through the blocked LDS kernels -- two
    // launches for a mixed level, from a per-level list sorted narrow first.
    struct LevelLists {
        std::vector<int> ptr, narrow, lds;  // level l = list[ptr[l] .. ptr[l+1]): `narrow[l]` single-wave fronts first; LDS bytes of its widest front
        DBuf<int> dev;
        // runs of consecutive levels that hold single-wave fronts only and continue each other as chains (front t's one child inside the run is t - 1):
        // ONE launch of chain walks per run instead of one launch per level (CONT-201: ten levels of four fronts each -> one launch of four walks)
        struct Run { int a, b, off, nwalk; };
        std::vector<Run> runs;
        std::vector<int> run_of;  // level -> run (or -1)
        DBuf<int> walk_lo, walk_hi;
        // huge fronts (huge_front()): per level the list [hptr[l], hptr[l + 1]) of hdev, the grids of k_front_fwd_rows / k_front_bwd_cols and the LDS of the former;
        // hoff[s] = offset of front s's partial column sums in huge_part_ (the buffer is reused level after level)
        std::vector<int> hptr, hfwd_grid, hbwd_grid, hlds;
        DBuf<int> hdev, hoff;
    };
    void build_level_lists(const std::vector<int>& ptr, const std::vector<int>& sn, LevelLists& L)
    {
        L.ptr = ptr; L.narrow.clear(); L.lds.clear(); L.runs.clear();
        const int nl = (int)ptr.size() - 1;
        std::vector<int> order;
        for (int l = 0; l < nl; ++l) {
            int fmax = 0;
            for (int q = ptr[l]; q < ptr[l + 1]; ++q) if (S_.front_rows_ptr[sn[q] + 1] - S_.front_rows_ptr[sn[q]] <= 128) order.push_back(sn[q]);
            L.narrow.push_back((int)order.size() - ptr[l]);
            for (int q = ptr[l]; q < ptr[l + 1]; ++q) {
                const int f = S_.front_rows_ptr[sn[q] + 1] - S_.front_rows_ptr[sn[q]];
                if (f > 128) { order.push_back(sn[q]); if (f <= wide_fcap_) fmax = std::max(fmax, f); }
            }
            L.lds.push_back((((fmax + 1) & ~1) + WIDE_B) * (int)sizeof(double));
        }
        upload_vec(L.dev, order, st_);
        {
            std::vector<int> hl, ho(S_.nsuper ? S_.nsuper : 1, 0);
            L.hptr.assign(1, 0); L.hfwd_grid.clear(); L.hbwd_grid.clear(); L.hlds.clear();
            size_t need = 0;
            for (int l = 0; l < nl; ++l) {
                int gf = 0, gb = 0, wmax = 0;
                size_t off = 0;
                for (int q = ptr[l]; q < ptr[l + 1]; ++q) {
                    const int s2 = sn[q], f = S_.front_rows_ptr[s2 + 1] - S_.front_rows_ptr[s2], w = S_.sn_first[s2 + 1] - S_.sn_first[s2];
                    if (!huge_front(f, w) || f > wide_fcap_) continue;
                    hl.push_back(s2);
                    const int nch = (f - w + HUGE_ROWS - 1) / HUGE_ROWS;
                    gf = std::max(gf, (f - w + 63) / 64); gb = std::max(gb, nch); wmax = std::max(wmax, w);
                    ho[s2] = (int)off; off += (size_t)nch * w;
                }
                if (off > 0x7fffffffull) throw std::runtime_error("sparse backend: partial sums of a level's huge fronts exceed 2^31 doubles");
                need = std::max(need, off);
                L.hptr.push_back((int)hl.size()); L.hfwd_grid.push_back(gf); L.hbwd_grid.push_back(gb); L.hlds.push_back(wmax * (int)sizeof(double));
            }
            if (!hl.empty()) {
                upload_vec(L.hdev, hl, st_); upload_vec(L.hoff, ho, st_);
                if (huge_part_.n < need) { stream_wait(st_); huge_part_.alloc(need); }
            }
        }
        L.run_of.assign(std::max(nl, 1), -1);
        if (no_runs_) return;
        std::vector<int> lvl(S_.nsuper ? S_.nsuper : 1, -1), walk_end(S_.nsuper ? S_.nsuper : 1, -1), wlo, whi;
        for (int l = 0; l < nl; ++l) for (int q = ptr[l]; q < ptr[l + 1]; ++q) lvl[sn[q]] = l;
        auto narrow_only = [&](int l) { return ptr[l + 1] > ptr[l] && L.narrow[l] == ptr[l + 1] - ptr[l]; };
        for (int l = 0; l < nl;) {
            if (!narrow_only(l)) { ++l; continue; }
            const int a = l, off = (int)wlo.size();
            for (int q = ptr[a]; q < ptr[a + 1]; ++q) { walk_end[sn[q]] = (int)wlo.size(); wlo.push_back(sn[q]); whi.push_back(sn[q]); }
            int b2 = a;
            while (b2 + 1 < nl && narrow_only(b2 + 1)) {
                bool ok = true;
                for (int q = ptr[b2 + 1]; q < ptr[b2 + 2] && ok; ++q) {
                    const int t = sn[q];
                    int nin = 0, cin = -1;
                    for (int ci = S_.child_ptr[t]; ci < S_.child_ptr[t + 1]; ++ci) { const int c = S_.child[ci]; if (lvl[c] >= a && lvl[c] <= b2) { ++nin; cin = c; } }
                    if (nin == 0) continue;
                    if (!(nin == 1 && cin == t - 1 && lvl[cin] == b2 && walk_end[cin] >= 0 && S_.sn_parent[cin] == t)) ok = false;
                }
                if (!ok) break;
                for (int q = ptr[b2 + 1]; q < ptr[b2 + 2]; ++q) {
                    const int t = sn[q];
                    bool ext = false;
                    for (int ci = S_.child_ptr[t]; ci < S_.child_ptr[t + 1]; ++ci) { const int c = S_.child[ci]; if (lvl[c] >= a && lvl[c] <= b2) ext = true; }
                    if (ext) { const int wk = walk_end[t - 1]; whi[wk] = t; walk_end[t - 1] = -1; walk_end[t] = wk; }
                    else { walk_end[t] = (int)wlo.size(); wlo.push_back(t); whi.push_back(t); }
                }
                ++b2;
            }
            if (b2 > a) {
                for (int r = a; r <= b2; ++r) L.run_of[r] = (int)L.runs.size();
                L.runs.push_back({a, b2, off, (int)wlo.size() - off});
            } else { wlo.resize(off); whi.resize(off); }
            l = b2 + 1;
        }
        if (!wlo.empty()) { upload_vec(L.walk_lo, wlo, st_); upload_vec(L.walk_hi, whi, st_); }
    }
    void fwd_levels(const FrontMeta& M, const LevelLists& L)
    {
        for (int l = 0; l + 1 < (int)L.ptr.size(); ++l) {
            const int cnt = L.ptr[l + 1] - L.ptr[l], nn = cnt > 0 ? L.narrow[l] : 0;
            const int* list = L.dev.p + L.ptr[l];
            if (L.run_of[l] >= 0) {
                const LevelLists::Run& r = L.runs[L.run_of[l]];
                if (l == r.a) hipLaunchKernelGGL(k_subtree_fwd_wave<false>, dim3(r.nwalk), dim3(64), 0, st_, M, fronts_.p, L.walk_lo.p + r.off, L.walk_hi.p + r.off, xp_.p, fvec_.p, (const int*)nullptr, (int*)nullptr, (int*)nullptr, (const int*)nullptr, 0);
                continue;
            }
            static const bool two_launches = debug_token("solve_level_two_launches") != nullptr;
            const int nh = L.hptr.empty() ? 0 : L.hptr[l + 1] - L.hptr[l];
            auto huge_rows = [&] {  // the update rows of the level's huge fronts, behind their pivot blocks
                if (nh > 0) hipLaunchKernelGGL(k_front_fwd_rows, dim3(L.hfwd_grid[l], nh), dim3(64), L.hlds[l], st_, M, fronts_.p, L.hdev.p + L.hptr[l], fvec_.p);
            };
            if (nn > 0 && cnt > nn && !two_launches) { hipLaunchKernelGGL(k_level_fwd_mixed, dim3(cnt), dim3(WIDE_NT), L.lds[l], st_, M, fronts_.p, list, nn, xp_.p, fvec_.p, wide_fcap_, (const int*)L.hoff.p); huge_rows(); continue; }
            if (nn > 0) hipLaunchKernelGGL(k_subtree_fwd_wave<false>, dim3(nn), dim3(64), 0, st_, M, fronts_.p, list, list, xp_.p, fvec_.p, (const int*)nullptr, (int*)nullptr, (int*)nullptr, (const int*)nullptr, 0);
            if (cnt > nn) hipLaunchKernelGGL(k_front_fwd_wide, dim3(cnt - nn), dim3(WIDE_NT), L.lds[l], st_, M, fronts_.p, list + nn, xp_.p, fvec_.p, wide_fcap_, (const int*)L.hoff.p);
            huge_rows();
        }
    }
    void bwd_levels(const FrontMeta& M, const LevelLists& L)
    {
        for (int l = (int)L.ptr.size() - 2; l >= 0; --l) {
            const int cnt = L.ptr[l + 1] - L.ptr[l], nn = cnt > 0 ? L.narrow[l] : 0;
            const int* list = L.dev.p + L.ptr[l];
            if (L.run_of[l] >= 0) {
                const LevelLists::Run& r = L.runs[L.run_of[l]];
                if (l == r.b) hipLaunchKernelGGL(k_subtree_bwd_wave<false>, dim3(r.nwalk), dim3(64), 0, st_, M, fronts_.p, L.walk_lo.p + r.off, L.walk_hi.p + r.off, xp_.p, bwd_red_thr(), 0, (const int*)nullptr, (int*)nullptr, (int*)nullptr, (const int*)nullptr, 0, (const double*)nullptr);
                continue;
            }
            static const bool two_launches = debug_token("solve_level_two_launches") != nullptr;
            const int nh = L.hptr.empty() ? 0 : L.hptr[l + 1] - L.hptr[l];
            if (nh > 0) hipLaunchKernelGGL(k_front_bwd_cols, dim3(L.hbwd_grid[l], nh, (L.hlds[l] / (int)sizeof(double) + HUGE_COLS - 1) / HUGE_COLS), dim3(HUGE_ROWS), 0, st_, M, fronts_.p, L.hdev.p + L.hptr[l], xp_.p, L.hoff.p, huge_part_.p);
            const int* hoff = L.hoff.p;  // (nullptr when the schedule holds no huge front)
            if (nn > 0 && cnt > nn && !two_launches) { hipLaunchKernelGGL(k_level_bwd_mixed, dim3(cnt), dim3(WIDE_NT), L.lds[l], st_, M, fronts_.p, list, nn, xp_.p, fvec_.p, wide_fcap_, bwd_red_thr(), hoff, huge_part_.p); continue; }
            if (cnt > nn) hipLaunchKernelGGL(k_front_bwd_wide, dim3(cnt - nn), dim3(WIDE_NT), L.lds[l], st_, M, fronts_.p, list + nn, xp_.p, fvec_.p, wide_fcap_, hoff, huge_part_.p);
            if (nn > 0) hipLaunchKernelGGL(k_subtree_bwd_wave<false>, dim3(nn), dim3(64), 0, st_, M, fronts_.p, list, list, xp_.p, bwd_red_thr(), 0, (const int*)nullptr, (int*)nullptr, (int*)nullptr, (const int*)nullptr, 0, (const double*)nullptr);
        }
    }

    void build_device(const pq_sparse_data* d)
    {
        nnzK_ = S_.Cp[N_];
        upload_vec(P_, S_.P, st_); upload_vec(level_sn_, S_.top_level_sn, st_); {
            std::vector<int> tp(S_.nsuper ? S_.nsuper : 1, -1);
            for (size_t q = 0; q < S_.top_level_sn.size(); ++q) tp[S_.top_level_sn[q]] = (int)q;
            upload_vec(top_pos_, tp, st_);
            top_flags_.alloc(2 * S_.top_level_sn.size() + 2); top_flags_.zero(st_);
        }
        {   // the substitution's own schedule
            upload_vec(solve_level_sn_, S_.solve_top_level_sn, st_);
            build_level_lists(S_.solve_top_level_ptr, S_.solve_top_level_sn, solve_ll_);
            ntop_solve_ = (int)S_.solve_top_level_sn.size();
            std::vector<int> tp(S_.nsuper ? S_.nsuper : 1, -1);
            for (size_t q = 0; q < S_.solve_top_level_sn.size(); ++q) tp[S_.solve_top_level_sn[q]] = (int)q;
            upload_vec(solve_top_pos_, tp, st_);
            solve_flags_.alloc(2 * S_.solve_top_level_sn.size() + 2); solve_flags_.zero(st_);  // flags hold the epoch of the solve that set them: they must start at zero
            upload_vec(solve_walk_lo_, S_.solve_walk_lo, st_); upload_vec(solve_walk_hi_, S_.solve_walk_hi, st_);
            nwalk_solve_ = (int)S_.solve_walk_lo.size();
            {   // backward sweep: a supernode publishes its flag only if a top child in ANOTHER walk waits for it
                std::vector<int> walk_of(S_.nsuper ? S_.nsuper : 1, -1), pub(tp.size() ? S_.solve_top_level_sn.size() + 1 : 1, 0);
                for (int wk = 0; wk < nwalk_solve_; ++wk) for (int t = S_.solve_walk_lo[wk]; t <= S_.solve_walk_hi[wk]; ++t) walk_of[t] = wk;
                for (size_t q = 0; q < S_.solve_top_level_sn.size(); ++q) {
                    const int c = S_.solve_top_level_sn[q], ps = S_.sn_parent[c];
                    if (ps >= 0 && tp[ps] >= 0 && walk_of[ps] != walk_of[c]) pub[tp[ps]] = 1;
                }
                upload_vec(solve_pub_, pub, st_);
            }
            build_solve_schedule();
        }
        upload_vec(fe_ptr_, S_.fe_ptr, st_); upload_vec(fe_q_, S_.fe_q, st_); upload_vec(fe_off_, S_.fe_off, st_);
        {   // the same offsets for a front stored as its packed lower triangle (k_subtree_factor_pk)
            std::vector<int> offp(S_.fe_off.size());
            for (int t = 0; t < S_.nsuper; ++t) {
                const int f = S_.front_rows_ptr[t + 1] - S_.front_rows_ptr[t];
                for (int e = S_.fe_ptr[t]; e < S_.fe_ptr[t + 1]; ++e) {
                    const int i = S_.fe_off[e] % f, j = S_.fe_off[e] / f;
                    if (i < j) throw std::runtime_error("front entry above the diagonal");
                    offp[e] = (j * (2 * f - j - 1)) / 2 + i;
                }
            }
            upload_vec(fe_offp_, offp, st_);
        } upload_vec(sn_first_, S_.sn_first, st_);
        upload_vec(front_rows_ptr_, S_.front_rows_ptr, st_); upload_vec(front_rows_, S_.front_rows, st_); upload_vec(child_ptr_, S_.child_ptr, st_); upload_vec(child_, S_.child, st_);
        {
            std::vector<ChildRec> cr(S_.child.size() ? S_.child.size() : 1);
            std::vector<int> ctp(S_.child.size() ? S_.child.size() : 1, -1);
            std::vector<int> pos(S_.nsuper ? S_.nsuper : 1, -1);
            for (size_t q = 0; q < S_.solve_top_level_sn.size(); ++q) pos[S_.solve_top_level_sn[q]] = (int)q;
            for (size_t ci = 0; ci < S_.child.size(); ++ci) {
                const int c = S_.child[ci];
                const int wc = S_.sn_first[c + 1] - S_.sn_first[c], fc = S_.front_rows_ptr[c + 1] - S_.front_rows_ptr[c];
                cr[ci] = ChildRec{c, S_.front_rows_ptr[c] + wc, fc - wc, S_.rel_ptr[c]};
                ctp[ci] = pos[c];
            }
            upload_vec(crec_, cr, st_); upload_vec(solve_child_tp_, ctp, st_);
        }
        upload_vec(rel_ptr_, S_.rel_ptr, st_); upload_vec(rel_, S_.rel, st_); upload_vec(front_off_, S_.front_off, st_);
        {
            std::vector<SnRec> rec(S_.nsuper ? S_.nsuper : 1);
            for (int q = 0; q < S_.nsuper; ++q) {
                SnRec& r = rec[q];
                r.first = S_.sn_first[q]; r.w = S_.sn_first[q + 1] - S_.sn_first[q]; r.f = S_.front_rows_ptr[q + 1] - S_.front_rows_ptr[q];
                r.rows_ptr = S_.front_rows_ptr[q]; r.child_lo = S_.child_ptr[q]; r.child_hi = S_.child_ptr[q + 1]; r.rel_ptr = S_.rel_ptr[q]; r.parent = S_.sn_parent[q];
                r.front_off = S_.front_off[q]; r.fe_lo = S_.fe_ptr[q]; r.fe_hi = S_.fe_ptr[q + 1]; r.nind = S_.sn_nind[q];
            }
            upload_vec(snrec_, rec, st_);
        }
        vals_.alloc(nnzK_ ? nnzK_ : 1); vals_.zero(st_);
        fronts_.alloc(S_.front_doubles ? (size_t)S_.front_doubles : 1);
        rdiag_.alloc(N_); xp_.alloc(N_); fvec_.alloc(S_.front_rows.size() ? S_.front_rows.size() : 1);
        build_big_levels(S_.top_level_ptr, S_.top_level_sn, top_big_);
        info_.alloc(1); info_h_.alloc(1);
        // value maps K-index -> PKPt-index composed with the per-matrix maps (kkt_full.hpp:219-249)
        const int nzP = d->P_colptr[n_], nzA = p_ ? d->AT_colptr[p_] : 0, nzG = m_ ? d->GT_colptr[m_] : 0;
        const bool eq = mode_ & 1, ineq = mode_ & 2;
        std::vector<int> mp(nzP), ma(eq ? 0 : nzA), mg(ineq ? 0 : nzG);
        // the device value array is stored in FRONT order (position e of the assembly lists holds PKPt entry fe_q[e]): a front reads
        // its own entries as one contiguous run instead of gathering them through an index list
        std::vector<int> pos(nnzK_ ? nnzK_ : 1, 0);
        for (int e = 0; e < nnzK_; ++e) pos[S_.fe_q[e]] = e;
        {
            std::vector<int> dp(S_.diag_pos.size());
            for (size_t c = 0; c < dp.size(); ++c) dp[c] = pos[S_.diag_pos[c]];
            upload_vec(diag_pos_, dp, st_);
        }
        for (int q = 0; q < nzP; ++q) mp[q] = pos[S_.PKi[S_.P_utri_to_Ki[q]]];
        for (size_t q = 0; q < ma.size(); ++q) ma[q] = pos[S_.PKi[S_.AT_to_Ki[q]]];
        for (size_t q = 0; q < mg.size(); ++q) mg[q] = pos[S_.PKi[S_.GT_to_Ki[q]]];
        upload_vec(mapP_, mp, st_); upload_vec(mapA_, ma, st_); upload_vec(mapG_, mg, st_);
        // eliminated blocks: entry -> PKPt index and product-term lists
        nzAA_ = (int)S_.gramA.rowind.size(); nzGG_ = (int)S_.gramG.rowind.size();
        std::vector<int> maa(nzAA_), mgg(nzGG_);
        for (int e = 0; e < nzAA_; ++e) maa[e] = pos[S_.PKi[S_.gramA_to_Ki[e]]];
        for (int e = 0; e < nzGG_; ++e) mgg[e] = pos[S_.PKi[S_.gramG_to_Ki[e]]];
        upload_vec(mapAA_, maa, st_); upload_vec(mapGG_, mgg, st_);
        upload_vec(aa_ptr_, S_.gramA.ptr, st_); upload_vec(aa_q1_, S_.gramA.q1, st_); upload_vec(aa_q2_, S_.gramA.q2, st_); upload_vec(aa_k_, S_.gramA.k, st_);
        upload_vec(gg_ptr_, S_.gramG.ptr, st_); upload_vec(gg_q1_, S_.gramG.q1, st_); upload_vec(gg_q2_, S_.gramG.q2, st_); upload_vec(gg_k_, S_.gramG.k, st_);
        ata_vals_.alloc(nzAA_ ? nzAA_ : 1); zinv_.alloc(m_ ? m_ : 1); rhs_top_.alloc(n_ ? n_ : 1);
        build_full_schedule();
        ops_.init(d, st_);  // CSC copies for the mat-vecs (uploads the values once)
        remap_values();
    }

    void upload_values(const pq_sparse_data* d)
    {
        ops_.upload_values(d, st_);
        remap_values();
    }

    // device copies of the caller's values -> PKPt value array
    void remap_values()
    {
        if (mode_ == 0) {
            launch_remap_values(ops_.nzP(), mapP_.p, ops_.P_x(), vals_.p, st_);
            launch_remap_values(ops_.nzA(), mapA_.p, ops_.AT_x(), vals_.p, st_);
            launch_remap_values(ops_.nzG(), mapG_.p, ops_.GT_x(), vals_.p, st_);
        } else if ((mode_ & 1) && nzAA_) {
            // update_AT_A (kkt_all_eliminated.hpp:184-202): the values change only with the data; every factorisation rebuilds PKPt
            hipLaunchKernelGGL(k_gram_values<false>, g1(nzAA_), dim3(256), 0, st_, nzAA_, aa_ptr_.p, aa_q1_.p, aa_q2_.p, aa_k_.p, ops_.AT_x(), (const double*)nullptr,
                               (const int*)nullptr, ata_vals_.p);
        }
        PQ_HIP(hipGetLastError());
        stream_wait(st_);
    }

    int dev_, mode_ = 0, nzAA_ = 0, nzGG_ = 0, n_ = 0, p_ = 0, m_ = 0, N_ = 0, nnzK_ = 0;
    double delta_ = 1.0;
    hipStream_t st_ = nullptr, st2_ = nullptr;  // st2_: the one-workgroup fronts of a level next to its big fronts' diagonal blocks (factor_levels)
    hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;
    sparse::Symbolic S_;
    std::vector<int> level_lds_;
    int sub_lds_ = 0, ntop_ = 0, top_grid_ = 0, top_lds_ = 0, top_l0_ = 0, top_start_ = 0, top_nper_ = 0;
    SubSchedule sched_, part_sched_;
    bool top_persistent_ = false;
    CscOperators ops_;
    DBuf<double> vals_, fronts_, rdiag_, xp_, fvec_;
    DBuf<int> fe_ptr_, fe_q_, fe_off_, fe_offp_, top_pos_, top_flags_, solve_level_sn_, solve_top_pos_, solve_flags_, solve_walk_lo_, solve_walk_hi_, solve_pub_;
    SubSchedule solve_sched_;
    LevelLists solve_ll_, own_ll_, sh_ll_;
    bool no_runs_ = debug_token("no_level_runs") != nullptr;  // debugging aid: one substitution launch per level, no merged runs of chain levels
    bool no_fork_ = debug_token("no_fork") != nullptr;  // debugging aid: everything on one stream
    mutable int tree_has_big_ = -1;  // lazily: does the top of the tree hold a front for the dense kernels (or is it too large for one persistent launch)
    // rows (caller's numbering) of the x / y / z blocks and of the whole KKT system that belong to the fronts this rank factors
    DBuf<int> need_x_, need_y_, need_z_, need_all_;
    DBuf<int> selP_, selA_, selG_, selAA_, selGG_, selD_;  // condensed modes, partitioned: source entries / diagonal columns whose values land in a front this rank factors
    int selP_n_ = 0, selA_n_ = 0, selG_n_ = 0, selAA_n_ = 0, selGG_n_ = 0, selD_n_ = 0, sharded_asm_ = 0;
    bool sel_ready_ = false;
    std::vector<std::pair<size_t, size_t>> own_val_ranges_;  // positions of the value array those fronts own (zeroed per factorisation)
    int need_x_n_ = 0, need_y_n_ = 0, need_z_n_ = 0, need_all_n_ = 0, sharded_evals_ = 0;
    // condensed modes (round 5): rows of the eliminated blocks by owner rank (dual_rows_[dual_own_ptr_[r] .. dual_own_ptr_[r + 1]) ; y row j as j, z row k as p + k), the
    // gather slot that holds a span of solution columns or one rank's multipliers, and the residual buffer whose next backend solve is a partial one
    std::vector<int> dual_own_ptr_;
    std::vector<std::pair<int, std::vector<int>>> dual_owner_tmp_;
    DBuf<int> dual_rows_, dual_own_ptr_d_;
    int gather_slot_ = 1, fold_x_n_ = 0, sharded_solves_ = 0, dual_gathers_ = 0;
    const double* partial_rhs_ = nullptr;
    DBuf<double> norm_bits_, own_norm_;
    HBuf<double> norm_h_{2};
    double* xbuf_norm_ = nullptr;
    int sharded_agreed_ = -1;  // -1: the ranks have not agreed yet on the sharded residual (reset whenever the partition or the transport changes)
    int ref_mode_ = PQ_REF_MODE;  // arithmetic of the one-workgroup fronts (PQ_REF_MODE: the reference's, term by term; PQ_REF_MODE_BIG where the tree has multi-workgroup fronts)
    bool no_big_ = debug_token("no_big") != nullptr;    // debugging aid: every front through one workgroup's pivot loop (accuracy comparisons)
    DBuf<double> huge_part_;  // partial column sums of the huge fronts of one level (k_front_bwd_cols)
    int wide_fcap_ = debug_token("no_wide_solve") ? 0 : WIDE_FCAP;  // debugging aid: wide fronts through the per-pivot routines
    BigLevels top_big_, own_big_, sh_big_;
    int ntop_solve_ = 0, nwalk_solve_ = 0, solve_epoch_ = 0, solve_epoch_used_ = 0, factor_epoch_ = 0;
    const int* solve_err_ptr_ = nullptr;
    DBuf<int> top_walk_lo_, top_walk_hi_;
    int ntopwalk_ = 0, top_walk_cap_ = 0;
    DBuf<SnRec> snrec_;
    DBuf<ChildRec> crec_;
    DBuf<int> solve_child_tp_;
    DBuf<int> diag_pos_, P_, level_sn_, sn_first_, front_rows_ptr_, front_rows_, child_ptr_, child_, rel_ptr_, rel_;
    DBuf<int> mapP_, mapA_, mapG_, mapAA_, mapGG_, aa_ptr_, aa_q1_, aa_q2_, aa_k_, gg_ptr_, gg_q1_, gg_q2_, gg_k_;
    DBuf<double> ata_vals_, zinv_, rhs_top_;
    DBuf<long long> front_off_;
    DBuf<int> info_;
    HBuf<int> info_h_;
    StageProfiler prof_;
    // stage partition (pq_kkt_partition)
    bool part_on_ = false;
    int rank_ = 0, world_ = 1;
    sparse::Partition PT_;
    std::vector<int> own_ptr_, own_sn_, own_lds_, sh_ptr_, sh_sn_, sh_lds_;
    DBuf<int> own_sn_d_, sh_sn_d_, b_sn_, b_owner_, b_vec_off_, span_lo_d_, span_hi_d_;
    DBuf<long long> b_mat_off_;
    enum class Transport { None, Callback, Native };  // who performs the three exchanges: nobody yet / the caller's callback / the library's own RCCL calls
    Transport transport_ = Transport::None;
    pq_exchange_fn xfn_ = nullptr;
    rccl::Comm* comm_ = nullptr;  // native RCCL transport (pq_kkt_set_comm_rccl); the exchange buffers below are then the library's own
    DBuf<double> own_factor_, own_forward_, own_gather_;
    int native_calls_[3] = {0, 0, 0};
    void* xuser_ = nullptr;
    double *xbuf_factor_ = nullptr, *xbuf_forward_ = nullptr, *xbuf_gather_ = nullptr;
};

}  // namespace

KKTSolverBase* make_multifrontal_kkt(const pq_sparse_data* data, int mode, int device) { return new SparseKKT(data, mode, device); }

namespace {
// A host copy of the matrices a sparse backend is built from (the C-ABI lends them for the duration of a call only).
struct SparseDataCopy {
    std::vector<int> pc, pr, ac, ar, gc, gr, hl, hu, xl, xu;
    std::vector<double> pv, av, gv, xb;
    pq_sparse_data d{};
    bool has_xb = false;
    void take(const pq_sparse_data* s)
    {
        auto csc = [](int cols, const int* cp, const int* ri, const double* v, std::vector<int>& c, std::vector<int>& r, std::vector<double>& x) {
            c.assign(cp, cp + cols + 1);
            r.assign(ri, ri + c[cols]);
            x.assign(v, v + c[cols]);
        };
        csc(s->n, s->P_colptr, s->P_rowind, s->P_val, pc, pr, pv);
        csc(s->p, s->AT_colptr, s->AT_rowind, s->AT_val, ac, ar, av);
        csc(s->m, s->GT_colptr, s->GT_rowind, s->GT_val, gc, gr, gv);
        auto idx = [](const int* p, int k, std::vector<int>& o) { if (p && k > 0) o.assign(p, p + k); else o.clear(); };
        idx(s->h_l_idx, s->n_h_l, hl); idx(s->h_u_idx, s->n_h_u, hu); idx(s->x_l_idx, s->n_x_l, xl); idx(s->x_u_idx, s->n_x_u, xu);
        has_xb = s->x_b_scaling != nullptr;
        if (has_xb) xb.assign(s->x_b_scaling, s->x_b_scaling + s->n);
        d = *s;
        d.P_colptr = pc.data(); d.P_rowind = pr.data(); d.P_val = pv.data();
        d.AT_colptr = ac.data(); d.AT_rowind = ar.data(); d.AT_val = av.data();
        d.GT_colptr = gc.data(); d.GT_rowind = gr.data(); d.GT_val = gv.data();
        d.h_l_idx = hl.data(); d.h_u_idx = hu.data(); d.x_l_idx = xl.data(); d.x_u_idx = xu.data();
        d.x_b_scaling = has_xb ? xb.data() : nullptr;
        d.mem = PQ_MEM_HOST;
    }
};

// kkt_solver = sparse_ldlt (or a condensed mode) on a system small enough for the reference-order engine -- which has no stage partition: a caller that asks for one
// (pq_kkt_partition) gets the multifrontal engine from there on, built from this wrapper's copy of the data, instead of "not supported" (round-5 advice).  Everything
// else is forwarded to the engine in use.
class EngineSwitchKKT final : public KKTSolverBase {
public:
    EngineSwitchKKT(KKTSolverBase* exact, const pq_sparse_data* data, int mode, int device) : cur_(exact), mode_(mode), dev_(device) { copy_.take(data); }
    ~EngineSwitchKKT() override { delete cur_; }
    KKTSolverBase* clone() const override
    {
        auto* c = new EngineSwitchKKT(cur_->clone(), &copy_.d, mode_, dev_);
        c->switched_ = switched_;
        return c;
    }
    void update_data_sparse(const pq_sparse_data* data, int options) override { copy_.take(data); cur_->update_data_sparse(data, options); }
    bool update_scalings_and_factor(double delta, const double* x_reg, const double* z_reg) override { return cur_->update_scalings_and_factor(delta, x_reg, z_reg); }
    void solve(const double* rx, const double* ry, const double* rz, double* lx, double* ly, double* lz) override { cur_->solve(rx, ry, rz, lx, ly, lz); }
    void eval_P_x(double alpha, const double* x, double* z) override { cur_->eval_P_x(alpha, x, z); }
    void eval_A_xn_and_AT_xt(double an, double at, const double* xn, const double* xt, double* zn, double* zt) override { cur_->eval_A_xn_and_AT_xt(an, at, xn, xt, zn, zt); }
    void eval_G_xn_and_GT_xt(double an, double at, const double* xn, const double* xt, double* zn, double* zt) override { cur_->eval_G_xn_and_GT_xt(an, at, xn, xt, zn, zt); }
    void print_info() override { cur_->print_info(); }
    const double* P_diag_device() const override { return cur_->P_diag_device(); }
    int n() const override { return cur_->n(); }
    int p() const override { return cur_->p(); }
    int m() const override { return cur_->m(); }
    hipStream_t stream() const override { return cur_->stream(); }
    int device() const override { return cur_->device(); }
    void sparse_stats(double out[8]) const override { cur_->sparse_stats(out); }
    int sparse_ordering(int* fill_perm, int* elim_perm) const override { return cur_->sparse_ordering(fill_perm, elim_perm); }
    void partition(int rank, int world, long long sizes[3]) override
    {
        if (!switched_) {
            // (the streams of the two engines are their own: whatever the caller queued on the old one is finished before it goes)
            stream_wait(cur_->stream());
            KKTSolverBase* mf = make_multifrontal_kkt(&copy_.d, mode_, dev_);
            delete cur_;
            cur_ = mf;
            switched_ = true;
        }
        cur_->partition(rank, world, sizes);
    }
    void set_exchange(pq_exchange_fn fn, void* user, double* bf, double* bw, double* bg) override { cur_->set_exchange(fn, user, bf, bw, bg); }
    void set_comm_rccl(const unsigned char* id128, int rank, int world) override { cur_->set_comm_rccl(id128, rank, world); }
    bool reference_order() const override { return cur_->reference_order(); }
    long long exact_factor(int what, void* out_host) override { return cur_->exact_factor(what, out_host); }
    double min_abs_pivot() override { return cur_->min_abs_pivot(); }
    void native_exchange_calls(int out[3]) const override { cur_->native_exchange_calls(out); }
    void set_exchange_norm(double* buf_norm) override { cur_->set_exchange_norm(buf_norm); }
    bool refine_error_sharded(const double* lx, const double* ly, const double* lz, const double* rx, const double* ry, const double* rz, const double* x_reg, double delta,
                              const double* z_reg, double* ex, double* ey, double* ez, double* norm) override
    {
        return cur_->refine_error_sharded(lx, ly, lz, rx, ry, rz, x_reg, delta, z_reg, ex, ey, ez, norm);
    }
    void sharded_calls(int out[2]) const override { cur_->sharded_calls(out); }
    void finish_sharded_solve(double* ly, double* lz) override { cur_->finish_sharded_solve(ly, lz); }
    void sharded_solve_calls(int out[6]) const override { cur_->sharded_solve_calls(out); }
    void comm_info(int out[4]) const override { cur_->comm_info(out); }
    void partition_info(int out[8]) const override { cur_->partition_info(out); }
    void set_profiling(int level) override { cur_->set_profiling(level); }
    void get_profile(int stage, double* total_ms, int* count) override { cur_->get_profile(stage, total_ms, count); }

private:
    KKTSolverBase* cur_;
    SparseDataCopy copy_;
    int mode_, dev_;
    bool switched_ = false;
};
}  // namespace

// KKTSystem::init_kkt_solver<PIQP_SPARSE> (kkt_system.hpp:470-497): all six sparse backends of the reference
KKTSolverBase* make_sparse_kkt(const pq_sparse_data* data, int kkt_solver, int device)
{
    switch (kkt_solver) {
    case PQ_SPARSE_MULTISTAGE: return make_multistage_kkt(data, device);
    case PQ_SPARSE_LDLT: case PQ_SPARSE_LDLT_EQ_COND: case PQ_SPARSE_LDLT_INEQ_COND: case PQ_SPARSE_LDLT_COND: {
        // Two engines behind the reference's sparse_ldlt family: the reference-order up-looking LDLt (sparse_exact.hip: L, D and the solves bitwise
        // the reference's, so that rounding-decided trajectories are the reference's too) for KKT systems up to PIQP_AMD_EXACT_MAX_N rows (default 8192) whose
        // factorisation takes up to PIQP_AMD_EXACT_MAX_FLOPS flops (default 4e7) -- every netlib / Maros-Meszaros problem of that size (the largest, STCQP2, takes
        // 3.3e7); that engine is a chain of ordered operations, and on denser systems (the reference's dense-vs-sparse benchmark at dim >= 256: 4.5e7 flops and up) the
        // supernodal multifrontal engine is several times to 20 x faster.  PIQP_AMD_SPARSE_LDLT=exact|multifrontal forces one.
        const int mode = kkt_solver - PQ_SPARSE_LDLT;  // KKTMode bits: 1 = equalities eliminated, 2 = inequalities eliminated
        const char* eng = std::getenv("PIQP_AMD_SPARSE_LDLT");
        const char* mx = std::getenv("PIQP_AMD_EXACT_MAX_N");
        const long long max_n = mx ? std::atoll(mx) : 8192;
        const long long N = (long long)data->n + ((mode & 1) ? 0 : data->p) + ((mode & 2) ? 0 : data->m);
        const bool exact = eng ? std::string(eng) == "exact" : N <= max_n;
        if (exact) {
            const char* mf = std::getenv("PIQP_AMD_EXACT_MAX_FLOPS");
            // (the condensed modes' systems are denser -- nl_czprob 2.1e9 flops, eleven fixtures above 2e8 -- and keep their bitwise contract too: limit 3e9 there)
            const double max_flops = eng ? 0.0 : (mf ? std::atof(mf) : (mode == 0 ? 4e7 : 3e9));
            if (KKTSolverBase* k = make_exact_sparse_kkt(data, mode, device, max_flops)) return new EngineSwitchKKT(k, data, mode, device);
        }
        return new SparseKKT(data, mode, device);
    }
    case PQ_SPARSE_LDLT_EXACT: return make_exact_sparse_kkt(data, 0, device);
    case PQ_SPARSE_LDLT_MULTIFRONTAL: return new SparseKKT(data, 0, device);
    default: return nullptr;
    }
}

}  // namespace pq
