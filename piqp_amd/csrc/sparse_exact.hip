// piqp_amd/csrc/sparse_exact.hip -- the reference's OWN sparse elimination on the device (round 5):
// piqp::sparse::KKT<T,I,KKT_FULL> with AMDOrdering and LDLt (reference include/piqp/sparse/kkt.hpp:51-176, kkt_full.hpp:172-251,
// ldlt.hpp:101-218), every floating-point operation in the reference's order, so that L, D and every solve are BITWISE the values of the
// reference's CPU path (its restatement oracle/orc_sparse.c compiled without FMA contraction, as ldlt.hpp:151-158 forces).
//
// Why it exists: the supernodal multifrontal engine (sparse_kkt.hip) groups the terms of an entry by child front.  Degenerate LPs decide
// their interior-point trajectory on exact zeros of cancelling pivots (solver.hpp:688-708: D[k] == 0.0 -> regularisation x 100), and no
// per-term variant of a multifrontal sum reproduces those (profiles/r04_ref_arith.txt).  This engine does, by construction:
//
//   reference (serial)                                        here
//   for k: pattern of row k by etree walks (:121-143)          fixed once on the host (sparse_symbolic.cpp analyse_uplooking): Rcol / Rpos
//   for i in pattern (topological order):                      one WAVE per row: the i loop stays sequential (that order is the rounding
//     for p in column i: y[L_ind[p]] -= fl(L_vals[p] * y_i)      order of every entry), the p loop runs across the 64 lanes -- distinct
//     l_ki = y_i / D[i]; D[k] -= fl(l_ki * y_i)                  targets, so lane order is immaterial; y lives in LDS (dense, N doubles);
//   D[k] == 0.0 -> return k                                      quotients of 64 entries at a time, D[k] -= ... strictly in pattern order
//   rows one after the other                                   rows of disjoint elimination subtrees concurrently: a chain of the tree
//                                                              (k -> k+1 the only child) is one task, tasks are handed out in row order
//                                                              to a persistent grid, a task waits for the tasks that end in its children
//   lsolve / dsolve / ltsolve (:171-218)                       k_ul_solve: x in LDS, columns in the reference's order, the products of a
//                                                              column across the lanes, the sums in the reference's order
//
// Algorithmic work: the same sum_j (c_j^2 + 3 c_j) flops as any LDLt of this pattern; the dependent chain is one LDS round trip per entry of L
// on the longest root path of the elimination tree (UpLooking::crit_steps), not a memory round trip: the columns a step reads are prefetched.
#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

#include "trace.hpp"
#include "kkt_solver_base.hpp"
#include "sparse_ops.hpp"
#include "sparse_symbolic.hpp"

namespace pq {

namespace {

inline dim3 g1(int n) { return dim3(n > 0 ? (n + 255) / 256 : 1); }

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double readlane_d(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int readfirst(int v) { return __builtin_amdgcn_readfirstlane(v); }
// fl(a - fl(x y)): product rounded, then the difference rounded ("force compiler to not use fma instruction", ldlt.hpp:151-153)
__device__ __forceinline__ double msub(double a, double x, double y) { return __dsub_rn(a, __dmul_rn(x, y)); }

// kkt_full.hpp:172-210 update_kkt_*: the diagonal of P K P' from the current scalings
__global__ void k_ul_set_diag(int n, int p, int m, const int* __restrict__ diag_pos, const double* __restrict__ Pdiag, const double* __restrict__ x_reg, double delta,
                              const double* __restrict__ z_reg, double* __restrict__ vals)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= n + p + m) return;
    double v;
    if (col < n) v = __dadd_rn(Pdiag[col], x_reg[col]);  // kkt_full.hpp:181
    else if (col < n + p) v = -delta;                     // :194
    else v = -z_reg[col - n - p];                         // :207
    vals[diag_pos[col]] = v;
}

struct UlFactorArgs {
    int N, nticket, epoch;
    const int *Cp, *Ci;
    const double* Cx;
    const int *tk_kind, *tk_id, *task_ptr, *task_rows, *row_task, *row_lane, *row_prev, *dep_ptr, *dep;
    const int *Rp, *Rcol, *Rpos, *Rcnt, *Rtab, *Lp, *Li;
    const int *tab_ptr, *mask_ptr, *task_nU;
    const unsigned long long* Tmask;
    double *Lx, *D, *Dinv, *Ystash, *Pstash, *Dinit, *Lblock;
    int *done, *p1done, *ready, *ticket, *info;
    double* yglob;  // N doubles per workgroup when y does not fit LDS
};

constexpr int UL_PF = 8;  // columns prefetched ahead of the dependent chain (their first 64 entries)

__device__ __forceinline__ bool spin_until(const int* flag, int epoch)
{
    long long spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1ll << 26)) return false;  // (seconds: a scheduling bug must not take the device with it)
    }
    return true;
}

// ROW PASS of row k (ldlt.hpp:121-163): the entries of row k in the columns outside the row's task, in the reference's order.  A row that is a task of its own is
// finished here; for a row on a longer path the updates into the path's rows are left to the path pass (Rcnt counts only the entries of a column above them), the
// values y_i, the products l_ki y_i and the initial values of the path columns go to Ystash / Pstash / Dinit, the quotients also into the task's table.
// y: this wave's dense work vector, all zero on entry and on exit.
__device__ __forceinline__ void ul_row(const UlFactorArgs& a, double* __restrict__ y, const int k, const int lane)
{
    const int t = a.row_task[k];
    const int W = a.task_ptr[t + 1] - a.task_ptr[t];
    const bool multi = W > 1;
    const int tb = a.tab_ptr[t], lanek = a.row_lane[k];
    // scatter A(0:k, k) into y (:127-131)
    const int p0 = a.Cp[k], p1 = a.Cp[k + 1];
    for (int q = p0 + lane; q < p1; q += 64) y[a.Ci[q]] = a.Cx[q];
    wave_sync();
    double Dk = y[k];  // :145  D[k] = y[k]
    wave_sync();
    if (lane == 0) y[k] = 0.0;
    const int r0 = a.Rp[k], r1 = a.Rp[k + 1];
    for (int base = r0; base < r1; base += 64) {
        const int e = base + lane;
        const bool in = e < r1;
        const int i = in ? a.Rcol[e] : 0, pos = in ? a.Rpos[e] : 0;
        const int cs = in ? a.Lp[i] : 0;
        const int rc = in ? a.Rcnt[e] : 0;   // entries of column i this pass scatters: all above row k (= L_nnz[i] at this moment, :149) or those above the task's path
        const int cnt = rc > 0 ? rc : 0;
        const bool ext = in && rc >= 0;      // (a column of the task's own path otherwise: its value in this row comes out of the path pass)
        const double Di = ext ? a.D[i] : 1.0;
        const int tabu = in ? a.Rtab[e] : 0;
        const int ns = min(64, r1 - base);
        double my_yi = 0.0;
        // the first 64 entries of the columns of the next UL_PF steps travel ahead of the chain.  Every load is unconditional (lanes past the end of their
        // column re-read its first entry, steps past the end of the row read entry 0 of the arrays): a load under a branch would be waited for at the branch
        int pf_i[UL_PF];
        double pf_v[UL_PF];
#pragma unroll
        for (int d = 0; d < UL_PF; ++d) {
            const int csu = __builtin_amdgcn_readlane(cs, d), cntu = __builtin_amdgcn_readlane(cnt, d);
            const int q = csu + (lane < cntu ? lane : 0);
            pf_i[d] = a.Li[q]; pf_v[d] = a.Lx[q];
        }
        for (int sb = 0; sb < ns; sb += UL_PF) {
#pragma unroll
            for (int d = 0; d < UL_PF; ++d) {
                const int s = sb + d;  // (steps ns .. of the last block are empty: cnt = 0 there, and nothing is cleared)
                const int iu = __builtin_amdgcn_readlane(i, s & 63), csu = __builtin_amdgcn_readlane(cs, s & 63), cntu = __builtin_amdgcn_readlane(cnt, s & 63);
                const int t0 = pf_i[d];
                const double v0 = pf_v[d];
                {   // refill this slot with the column of step s + UL_PF
                    const int s2 = (s + UL_PF) & 63;
                    const int cs2 = __builtin_amdgcn_readlane(cs, s2), cnt2 = __builtin_amdgcn_readlane(cnt, s2);
                    const int q = cs2 + (lane < cnt2 ? lane : 0);
                    pf_i[d] = a.Li[q]; pf_v[d] = a.Lx[q];
                }
                const double yi = y[iu];  // :147 (every lane reads the same word)
                if (lane == s) my_yi = yi;
                if (lane < cntu) y[t0] = msub(y[t0], v0, yi);  // :150-154, distinct targets
                if (lane == 0 && s < ns) y[iu] = 0.0;          // :148
                for (int q = 64 + lane; q < cntu; q += 64) { const int tt = a.Li[csu + q]; y[tt] = msub(y[tt], a.Lx[csu + q], yi); }
                wave_sync();
            }
        }
        // :155-161 for the 64 entries at once; D[k] loses its terms strictly in pattern order
        const double l = __ddiv_rn(my_yi, Di);
        const double tp = __dmul_rn(l, my_yi);
        if (ext) a.Lx[pos] = l;
        if (multi) {
            if (in) a.Ystash[e] = my_yi;
            if (ext) { a.Pstash[e] = tp; a.Lblock[tb + tabu * W + lanek] = l; }
        } else {
            for (int s = 0; s < ns; ++s) Dk = __dsub_rn(Dk, readlane_d(tp, s));
        }
    }
    if (lane == 0) {
        if (multi) a.Dinit[k] = Dk;
        else {
            a.D[k] = Dk;
            a.Dinv[k] = __ddiv_rn(1.0, Dk);  // :166
            if (Dk == 0.0) atomicMin(a.info, k);  // :163 (the smallest such k is the row the serial loop stops at)
        }
    }
}

// PATH PASS of task t: its rows one after the other, the path rows as lanes.  For row k the pattern is walked once more in the reference's order; an entry in an
// outside column i sends y_i (from the row pass) to the path rows c < k that hold an entry L(c, i) -- acc_c -= fl(L(c, i) y_i) -- and an entry in a path column c0
// first has its value final (acc of lane c0: every term it receives comes from a column the order visits earlier) and then does the same.  The quotients and the
// terms of D[k] follow, the latter again in pattern order.  L(c, .) is read from the task's table, where the passes that produced it left it.
__device__ __forceinline__ bool ul_path(const UlFactorArgs& a, const int t, const int lane, double* s_acc, int* s_pos)
{
    const int rb = a.task_ptr[t], W = a.task_ptr[t + 1] - rb;
    const int nU = a.task_nU[t], tb = a.tab_ptr[t], mb = a.mask_ptr[t];
    const int lw = lane < W ? lane : W - 1;
    double Dlane = 1.0;  // lane c: D of path row c once it is known
    for (int j = 0; j < W; ++j) {
        const int k = a.task_rows[rb + j];
        if (!spin_until(a.p1done + k, a.epoch)) return false;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const int r0 = a.Rp[k], r1 = a.Rp[k + 1];
        s_acc[lane] = 0.0; s_pos[lane] = -1;
        wave_sync();
        for (int e = r0 + lane; e < r1; e += 64) {
            const int u = a.Rtab[e];
            if (u >= nU) { s_acc[u - nU] = a.Ystash[e]; s_pos[u - nU] = a.Rpos[e]; }
        }
        wave_sync();
        double acc = s_acc[lane];
        const int mypos = s_pos[lane];
        wave_sync();
        const unsigned long long below = j >= 64 ? ~0ull : ((1ull << j) - 1ull);  // path rows under row k
        for (int base = r0; base < r1; base += 64) {
            const int e = base + lane;
            const bool in = e < r1;
            const int ue = in ? a.Rtab[e] : 0;
            const double yse = in ? a.Ystash[e] : 0.0;
            const unsigned long long me = in ? a.Tmask[mb + ue] : 0ull;
            const int mlo = (int)(unsigned)(me & 0xffffffffull), mhi = (int)(unsigned)(me >> 32);
            const int ns = min(64, r1 - base);
            double pf_v[UL_PF];
#pragma unroll
            for (int d = 0; d < UL_PF; ++d) pf_v[d] = a.Lblock[tb + __builtin_amdgcn_readlane(ue, d) * W + lw];
            for (int sb = 0; sb < ns; sb += UL_PF) {
#pragma unroll
                for (int d = 0; d < UL_PF; ++d) {
                    const int s = sb + d;
                    const int u = __builtin_amdgcn_readlane(ue, s & 63);
                    const unsigned long long m = (((unsigned long long)(unsigned)__builtin_amdgcn_readlane(mhi, s & 63) << 32) | (unsigned)__builtin_amdgcn_readlane(mlo, s & 63)) & below;
                    const double v = pf_v[d];
                    pf_v[d] = a.Lblock[tb + __builtin_amdgcn_readlane(ue, (s + UL_PF) & 63) * W + lw];
                    if (s < ns) {
                        const double src = u < nU ? readlane_d(yse, s & 63) : readlane_d(acc, (u - nU) & 63);
                        if ((m >> lane) & 1ull) acc = msub(acc, v, src);
                    }
                }
            }
        }
        // quotients of the path columns (:155), their places in L and in the table, their terms of D[k]
        const bool have = lane < j && mypos >= 0;
        const double l = __ddiv_rn(acc, Dlane);
        const double prodp = __dmul_rn(l, acc);
        if (have) { a.Lx[mypos] = l; a.Lblock[tb + (nU + lane) * W + j] = l; }
        double Dk = a.Dinit[k];
        for (int base = r0; base < r1; base += 64) {
            const int e = base + lane;
            const bool in = e < r1;
            const int ue = in ? a.Rtab[e] : 0;
            const double pse = in ? a.Pstash[e] : 0.0;
            const int ns = min(64, r1 - base);
            for (int s = 0; s < ns; ++s) {
                const int u = __builtin_amdgcn_readlane(ue, s);
                const double term = u < nU ? readlane_d(pse, s) : readlane_d(prodp, (u - nU) & 63);
                Dk = __dsub_rn(Dk, term);
            }
        }
        if (lane == 0) {
            a.D[k] = Dk;
            a.Dinv[k] = __ddiv_rn(1.0, Dk);
            if (Dk == 0.0) atomicMin(a.info, k);
        }
        if (lane == j) Dlane = Dk;
    }
    return true;
}

template <bool LDSY>
__global__ __launch_bounds__(64) void k_ul_factor(UlFactorArgs a)
{
    extern __shared__ double ul_sm[];
    __shared__ int s_task;
    __shared__ double s_acc[64];
    __shared__ int s_pos[64];
    double* __restrict__ y = LDSY ? ul_sm : a.yglob + (size_t)blockIdx.x * a.N;
    const int lane = threadIdx.x;
    for (int tt = lane; tt < a.N; tt += 64) y[tt] = 0.0;
    __syncthreads();
    // (the whole workgroup is one wave: __syncthreads() costs nothing and keeps the control flow around the ticket uniform for the compiler)
    for (int guard = 0; guard <= a.nticket; ++guard) {
        if (lane == 0) s_task = atomicAdd(a.ticket, 1);
        __syncthreads();
        const int tk = readfirst(s_task);
        __syncthreads();
        if (tk >= a.nticket) break;
        const int kind = a.tk_kind[tk], id = a.tk_id[tk];
        bool ok = true;
        if (kind == 0) {
            // row pass: the row's children outside its task must be complete rows; the rows below it on its own path are not waited for -- what THEY waited for is
            // inherited through the `ready` word of the row before
            const int k = id;
            const int c0 = a.dep_ptr[k], c1 = a.dep_ptr[k + 1];
            for (int c = c0 + lane; c < c1; c += 64) ok &= spin_until(a.done + a.dep[c], a.epoch);
            const int prev = a.row_prev[k];
            if (prev >= 0) ok &= spin_until(a.ready + prev, a.epoch);
            ok = __ballot(!ok) == 0;
            if (ok) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
                __hip_atomic_store(a.ready + k, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ul_row(a, y, k, lane);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                const int t = a.row_task[k];
                int* flag = (a.task_ptr[t + 1] - a.task_ptr[t] > 1 ? a.p1done : a.done) + k;
                __hip_atomic_store(flag, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (every lane stores the same word: no divergence at the loop's end)
            }
        } else {
            ok = ul_path(a, id, lane, s_acc, s_pos);
            if (ok) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_store(a.done + a.task_rows[a.task_ptr[id + 1] - 1], a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (!ok) { if (lane == 0) atomicMin(a.info, -2); break; }
    }
}

struct UlSolveArgs {
    int N, n, p, m;
    const int *perm, *Lp, *Li, *Lcol;
    const double *Lx, *Dinv;
    const int4* bgroup;  // backward sweep: groups of whole columns, last columns first: {qlo, qhi, -, -}
    int nbgroup;
    const double *rx, *ry, *rz;
    double *lx, *ly, *lz;
    double* xglob;
    int* err;  // set when the result holds a non-finite value
    int epoch;
};

// ordering.perm, lsolve, dsolve, ltsolve, ordering.permt (sparse/kkt.hpp:107-145 KKT_FULL, ldlt.hpp:171-218) by ONE wave; x in LDS
template <bool LDSX>
__global__ __launch_bounds__(64) void k_ul_solve(UlSolveArgs a)
{
    extern __shared__ double ul_sm[];
    double* __restrict__ x = LDSX ? ul_sm : a.xglob;
    const int lane = threadIdx.x;
    const int N = a.N;
    for (int j = lane; j < N; j += 64) {
        const int o = a.perm[j];
        x[j] = o < a.n ? a.rx[o] : (o < a.n + a.p ? a.ry[o - a.n] : a.rz[o - a.n - a.p]);
    }
    wave_sync();
    // lsolve: for j ascending: x[L_ind[p]] -= fl(L_vals[p] * x[j]).  The CSC arrays are streamed 64 entries at a time; inside a chunk the columns
    // are taken one after the other (a target receives its terms in ascending column order), the entries of one column across the lanes
    const int nnz = a.Lp[N];
    {
        int col = INT_MAX, row = 0;
        double v = 0.0;
        if (lane < nnz) { col = a.Lcol[lane]; row = a.Li[lane]; v = a.Lx[lane]; }
        for (int base = 0; base < nnz; base += 64) {
            int ncol = INT_MAX, nrow = 0;
            double nv = 0.0;
            const int q2 = base + 64 + lane;
            if (q2 < nnz) { ncol = a.Lcol[q2]; nrow = a.Li[q2]; nv = a.Lx[q2]; }  // the next chunk travels while this one is consumed
            int jcur = readfirst(col);
            unsigned long long mk = 1;
            while (mk != 0) {
                const double xj = x[jcur];
                if (col == jcur) x[row] = msub(x[row], v, xj);
                wave_sync();
                mk = __ballot(col > jcur && col != INT_MAX);
                if (mk != 0) jcur = __builtin_amdgcn_readlane(col, __builtin_ctzll(mk));
            }
            col = ncol; row = nrow; v = nv;
        }
    }
    // dsolve
    for (int j = lane; j < N; j += 64) x[j] = __dmul_rn(x[j], a.Dinv[j]);
    wave_sync();
    // ltsolve: for j descending: x[j] -= fl(L_vals[p] * x[L_ind[p]]) for p ascending.  Groups of whole columns (at most 64 entries, or one long column)
    {
        int4 g = a.nbgroup > 0 ? a.bgroup[0] : make_int4(0, 0, 0, 0);
        int col = -1, row = 0;
        double v = 0.0;
        if (a.nbgroup > 0 && g.x + lane < g.y && g.y - g.x <= 64) { col = a.Lcol[g.x + lane]; row = a.Li[g.x + lane]; v = a.Lx[g.x + lane]; }
        for (int gi = 0; gi < a.nbgroup; ++gi) {
            int4 g2 = make_int4(0, 0, 0, 0);
            int ncol = -1, nrow = 0;
            double nv = 0.0;
            if (gi + 1 < a.nbgroup) {
                g2 = a.bgroup[gi + 1];
                if (g2.x + lane < g2.y && g2.y - g2.x <= 64) { ncol = a.Lcol[g2.x + lane]; nrow = a.Li[g2.x + lane]; nv = a.Lx[g2.x + lane]; }
            }
            if (g.y - g.x <= 64) {
                const int cnt = g.y - g.x;
                int jcur = __builtin_amdgcn_readlane(col, cnt - 1);
                unsigned long long nm = 1;
                while (nm != 0) {
                    const bool mine = col == jcur;
                    const unsigned long long mk = __ballot(mine);
                    const int la = __builtin_ctzll(mk), lb = 64 - __builtin_clzll(mk);
                    double s = x[jcur];
                    const double pr = mine ? __dmul_rn(v, x[row]) : 0.0;
                    for (int l = la; l < lb; ++l) s = __dsub_rn(s, readlane_d(pr, l));
                    x[jcur] = s;  // (every lane writes the same word)
                    wave_sync();
                    nm = __ballot(col >= 0 && col < jcur);
                    if (nm != 0) jcur = __builtin_amdgcn_readlane(col, 63 - __builtin_clzll(nm));
                }
            } else {  // one long column: its entries in ascending order, 64 products at a time
                const int j = a.Lcol[g.x];
                double s = x[j];
                for (int q0 = g.x; q0 < g.y; q0 += 64) {
                    const int q = q0 + lane;
                    const double pr = q < g.y ? __dmul_rn(a.Lx[q], x[a.Li[q]]) : 0.0;
                    const int c = min(64, g.y - q0);
                    for (int l = 0; l < c; ++l) s = __dsub_rn(s, readlane_d(pr, l));
                }
                x[j] = s;
                wave_sync();
            }
            g = g2; col = ncol; row = nrow; v = nv;
        }
    }
    bool bad = false;
    for (int j = lane; j < N; j += 64) {
        const int o = a.perm[j];
        const double xv = x[j];
        bad |= !(fabs(xv) <= 1.7976931348623157e308);
        if (o < a.n) a.lx[o] = xv;
        else if (o < a.n + a.p) a.ly[o - a.n] = xv;
        else a.lz[o - a.n - a.p] = xv;
    }
    if (bad && a.err) *a.err = a.epoch;
}

class ExactSparseKKT final : public KKTSolverBase {
public:
    ExactSparseKKT(const pq_sparse_data* d, int device) : dev_(device)
    {
        if (d->mem != PQ_MEM_HOST) throw std::runtime_error("sparse data must be host-resident");
        PQ_HIP(hipSetDevice(dev_));
        PQ_HIP(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
        sparse::Symbolic S;
        sparse::analyse_kkt_pattern(d, 0, S);
        sparse::analyse_uplooking(S, d, U_);
        n_ = U_.n; p_ = U_.p; m_ = U_.m; N_ = U_.N;
        nnzK_ = U_.Cp[N_];
        build_device();
        ops_.init(d, st_);
        ops_.set_reference_order(true);
        remap_values();
    }
    ~ExactSparseKKT() override
    {
        (void)hipSetDevice(dev_);
        if (st_) { (void)hipStreamSynchronize(st_); (void)hipStreamDestroy(st_); }
    }
    KKTSolverBase* clone() const override
    {
        PQ_HIP(hipSetDevice(dev_));
        stream_wait(st_);
        return new ExactSparseKKT(*this, 0);
    }
    void update_data_sparse(const pq_sparse_data* d, int options) override
    {
        if (d->n != n_ || d->p != p_ || d->m != m_) throw std::runtime_error("update_data: dimension mismatch");
        PQ_HIP(hipSetDevice(dev_));
        (void)options;
        ops_.upload_values(d, st_);
        remap_values();
    }
    // sparse/kkt.hpp:83-105
    bool update_scalings_and_factor(double delta, const double* x_reg, const double* z_reg) override
    {
        PQ_ZONE("piqp_amd::ExactSparseKKT::update_scalings_and_factor");
        PQ_HIP(hipSetDevice(dev_));
        const int t0 = prof_.begin(0, st_);
        hipLaunchKernelGGL(k_ul_set_diag, g1(N_), dim3(256), 0, st_, n_, p_, m_, diag_pos_.p, ops_.P_diag(), x_reg, delta, z_reg, vals_.p);
        prof_.end(0, t0, st_);
        const int t1 = prof_.begin(1, st_);
        ++epoch_;
        ctl_h_.p[0] = 0; ctl_h_.p[1] = INT_MAX;
        PQ_HIP(hipMemcpyAsync(ctl_.p, ctl_h_.p, 2 * sizeof(int), hipMemcpyHostToDevice, st_));
        UlFactorArgs a;
        a.N = N_; a.nticket = nticket_; a.epoch = epoch_;
        a.Cp = Cp_.p; a.Ci = Ci_.p; a.Cx = vals_.p;
        a.tk_kind = tk_kind_.p; a.tk_id = tk_id_.p; a.task_ptr = task_ptr_.p; a.task_rows = task_rows_.p; a.row_task = row_task_.p; a.row_lane = row_lane_.p;
        a.row_prev = row_prev_.p; a.dep_ptr = dep_ptr_.p; a.dep = dep_.p;
        a.Rp = Rp_.p; a.Rcol = Rcol_.p; a.Rpos = Rpos_.p; a.Rcnt = Rcnt_.p; a.Rtab = Rtab_.p; a.Lp = Lp_.p; a.Li = Li_.p;
        a.tab_ptr = tab_ptr_.p; a.mask_ptr = mask_ptr_.p; a.task_nU = task_nU_.p; a.Tmask = Tmask_.p;
        a.Lx = Lx_.p; a.D = D_.p; a.Dinv = Dinv_.p; a.Ystash = Ystash_.p; a.Pstash = Pstash_.p; a.Dinit = Dinit_.p; a.Lblock = Lblock_.p;
        a.done = done_.p; a.p1done = p1done_.p; a.ready = ready_.p; a.ticket = ctl_.p; a.info = ctl_.p + 1;
        a.yglob = yglob_.p;
        if (N_ > 0) {
            if (lds_y_) hipLaunchKernelGGL(k_ul_factor<true>, dim3(grid_), dim3(64), (size_t)N_ * sizeof(double), st_, a);
            else hipLaunchKernelGGL(k_ul_factor<false>, dim3(grid_), dim3(64), 0, st_, a);
        }
        PQ_HIP(hipGetLastError());
        prof_.end(1, t1, st_);
        PQ_HIP(hipMemcpyAsync(ctl_h_.p + 2, ctl_.p + 1, sizeof(int), hipMemcpyDeviceToHost, st_));
        stream_wait(st_);
        if (ctl_h_.p[2] == -2) throw std::runtime_error("reference-order factorisation: a task waited for its children without end (scheduling error)");
        return ctl_h_.p[2] == INT_MAX;  // n == cols (sparse/kkt.hpp:104)
    }
    // sparse/kkt.hpp:107-145, KKT_FULL
    void solve(const double* rhs_x, const double* rhs_y, const double* rhs_z, double* lhs_x, double* lhs_y, double* lhs_z) override
    {
        PQ_ZONE("piqp_amd::ExactSparseKKT::solve");
        PQ_HIP(hipSetDevice(dev_));
        const int tk = prof_.begin(2, st_);
        UlSolveArgs a;
        a.N = N_; a.n = n_; a.p = p_; a.m = m_;
        a.perm = perm_.p; a.Lp = Lp_.p; a.Li = Li_.p; a.Lcol = Lcol_.p; a.Lx = Lx_.p; a.Dinv = Dinv_.p;
        a.bgroup = reinterpret_cast<const int4*>(bgroup_.p); a.nbgroup = nbgroup_;
        a.rx = rhs_x; a.ry = rhs_y; a.rz = rhs_z; a.lx = lhs_x; a.ly = lhs_y; a.lz = lhs_z;
        a.xglob = xglob_.p; a.err = nullptr; a.epoch = 0;
        if (N_ > 0) {
            if (lds_x_) hipLaunchKernelGGL(k_ul_solve<true>, dim3(1), dim3(64), (size_t)N_ * sizeof(double), st_, a);
            else hipLaunchKernelGGL(k_ul_solve<false>, dim3(1), dim3(64), 0, st_, a);
        }
        PQ_HIP(hipGetLastError());
        prof_.end(2, tk, st_);
    }
    void eval_P_x(double alpha, const double* x, double* z) override { PQ_HIP(hipSetDevice(dev_)); ops_.eval_P_x(alpha, x, z, st_); }
    void eval_A_xn_and_AT_xt(double an, double at, const double* xn, const double* xt, double* zn, double* zt) override
    {
        PQ_HIP(hipSetDevice(dev_));
        ops_.eval_A_xn_and_AT_xt(an, at, xn, xt, zn, zt, st_);
    }
    void eval_G_xn_and_GT_xt(double an, double at, const double* xn, const double* xt, double* zn, double* zt) override
    {
        PQ_HIP(hipSetDevice(dev_));
        ops_.eval_G_xn_and_GT_xt(an, at, xn, xt, zn, zt, st_);
    }
    void sparse_stats(double out[8]) const override
    {
        out[0] = N_; out[1] = nnzK_; out[2] = (double)U_.nnzL; out[3] = ntask_; out[4] = U_.height; out[5] = grid_; out[6] = (double)U_.crit_steps; out[7] = U_.flops;
    }
    int sparse_ordering(int* fill_perm, int* elim_perm) const override
    {
        if (fill_perm) std::copy(U_.perm.begin(), U_.perm.end(), fill_perm);
        if (elim_perm) std::copy(U_.perm.begin(), U_.perm.end(), elim_perm);  // the reference eliminates in AMD's own order
        return 0;
    }
    bool reference_order() const override { return true; }
    long long exact_factor(int what, void* out_host) override
    {
        PQ_HIP(hipSetDevice(dev_));
        stream_wait(st_);
        const long long nnzL = U_.nnzL;
        switch (what) {
        case 0: return nnzL;
        case 1: if (out_host) std::copy(U_.Lp.begin(), U_.Lp.end(), (int*)out_host); return N_ + 1;
        case 2: if (out_host) std::copy(U_.Li.begin(), U_.Li.end(), (int*)out_host); return nnzL;
        case 3: if (out_host && nnzL) PQ_HIP(hipMemcpy(out_host, Lx_.p, sizeof(double) * (size_t)nnzL, hipMemcpyDeviceToHost)); return nnzL;
        case 4: if (out_host && N_) PQ_HIP(hipMemcpy(out_host, D_.p, sizeof(double) * (size_t)N_, hipMemcpyDeviceToHost)); return N_;
        case 5: if (out_host && N_) PQ_HIP(hipMemcpy(out_host, Dinv_.p, sizeof(double) * (size_t)N_, hipMemcpyDeviceToHost)); return N_;
        case 6: if (out_host && nnzK_) PQ_HIP(hipMemcpy(out_host, vals_.p, sizeof(double) * (size_t)nnzK_, hipMemcpyDeviceToHost)); return nnzK_;
        case 7: if (out_host) std::copy(U_.perm.begin(), U_.perm.end(), (int*)out_host); return N_;
        default: throw std::runtime_error("exact_factor: unknown item");
        }
    }
    double min_abs_pivot() override
    {
        PQ_HIP(hipSetDevice(dev_));
        std::vector<double> h((size_t)N_);
        if (N_) PQ_HIP(hipMemcpyAsync(h.data(), D_.p, sizeof(double) * (size_t)N_, hipMemcpyDeviceToHost, st_));
        stream_wait(st_);
        double mn = N_ ? std::fabs(h[0]) : 0.0;
        for (double r : h) mn = std::min(mn, std::fabs(r));
        return mn;
    }
    void print_info() override
    {
        std::printf("sparse up-looking LDLt in the reference's order (AMD ordering, no postorder): N = %d, nnz(K) = %d, nnz(L) = %lld, elimination tree height = %d, %d chain tasks on %d waves, "
                    "dependent steps on the longest root path = %lld, work vector in %s\n", N_, nnzK_, U_.nnzL, U_.height, ntask_, grid_, U_.crit_steps, lds_y_ ? "LDS" : "HBM");
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

private:
    ExactSparseKKT(const ExactSparseKKT& o, int) : dev_(o.dev_), n_(o.n_), p_(o.p_), m_(o.m_), N_(o.N_), nnzK_(o.nnzK_), U_(o.U_)
    {
        PQ_HIP(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
        build_device();
        ops_.clone_from(o.ops_, st_);
        if (nnzK_) PQ_HIP(hipMemcpyAsync(vals_.p, o.vals_.p, sizeof(double) * (size_t)nnzK_, hipMemcpyDeviceToDevice, st_));
        if (U_.nnzL) PQ_HIP(hipMemcpyAsync(Lx_.p, o.Lx_.p, sizeof(double) * (size_t)U_.nnzL, hipMemcpyDeviceToDevice, st_));
        if (N_) {
            PQ_HIP(hipMemcpyAsync(D_.p, o.D_.p, sizeof(double) * (size_t)N_, hipMemcpyDeviceToDevice, st_));
            PQ_HIP(hipMemcpyAsync(Dinv_.p, o.Dinv_.p, sizeof(double) * (size_t)N_, hipMemcpyDeviceToDevice, st_));
        }
        stream_wait(st_);
    }

    void build_device()
    {
        upload_vec(perm_, U_.perm, st_); upload_vec(Cp_, U_.Cp, st_); upload_vec(Ci_, U_.Ci, st_); upload_vec(diag_pos_, U_.diag_pos, st_);
        upload_vec(mapP_, U_.mapP, st_); upload_vec(mapA_, U_.mapA, st_); upload_vec(mapG_, U_.mapG, st_);
        upload_vec(Lp_, U_.Lp, st_); upload_vec(Li_, U_.Li, st_); upload_vec(Lcol_, U_.Lcol, st_);
        upload_vec(Rp_, U_.Rp, st_); upload_vec(Rcol_, U_.Rcol, st_); upload_vec(Rpos_, U_.Rpos, st_);
        upload_vec(tk_kind_, U_.tk_kind, st_); upload_vec(tk_id_, U_.tk_id, st_); upload_vec(task_ptr_, U_.task_ptr, st_); upload_vec(task_rows_, U_.task_rows, st_);
        upload_vec(row_task_, U_.row_task, st_); upload_vec(row_lane_, U_.row_lane, st_); upload_vec(row_prev_, U_.row_prev, st_); upload_vec(dep_ptr_, U_.dep_ptr, st_);
        upload_vec(dep_, U_.dep, st_); upload_vec(Rcnt_, U_.Rcnt, st_); upload_vec(Rtab_, U_.Rtab, st_); upload_vec(tab_ptr_, U_.tab_ptr, st_); upload_vec(mask_ptr_, U_.mask_ptr, st_);
        upload_vec(task_nU_, U_.task_nU, st_); upload_vec(Tmask_, U_.Tmask, st_);
        ntask_ = (int)U_.task_ptr.size() - 1; nticket_ = (int)U_.tk_kind.size();
        Ystash_.alloc(U_.nnzL ? (size_t)U_.nnzL : 1); Pstash_.alloc(U_.nnzL ? (size_t)U_.nnzL : 1); Dinit_.alloc(N_ ? N_ : 1);
        Lblock_.alloc(U_.tab_ptr.back() ? (size_t)U_.tab_ptr.back() : 1); Lblock_.zero(st_);
        p1done_.alloc(N_ ? N_ : 1); p1done_.zero(st_); ready_.alloc(N_ ? N_ : 1); ready_.zero(st_);
        vals_.alloc(nnzK_ ? nnzK_ : 1); vals_.zero(st_);
        Lx_.alloc(U_.nnzL ? (size_t)U_.nnzL : 1); Lx_.zero(st_);
        D_.alloc(N_ ? N_ : 1); Dinv_.alloc(N_ ? N_ : 1); D_.zero(st_); Dinv_.zero(st_);
        done_.alloc(N_ ? N_ : 1); done_.zero(st_);
        ctl_.alloc(4); ctl_h_.alloc(4);
        {   // backward sweep groups: whole columns, last first, at most 64 entries each (a longer column alone)
            std::vector<int> g;
            int j = N_ - 1;
            while (j >= 0) {
                const int hi = U_.Lp[j + 1];
                int lo = U_.Lp[j];
                if (hi == lo) { --j; continue; }
                int jl = j;
                if (hi - lo <= 64) while (jl > 0 && hi - U_.Lp[jl - 1] <= 64) { --jl; lo = U_.Lp[jl]; }
                g.push_back(lo); g.push_back(hi); g.push_back(jl); g.push_back(j);
                j = jl - 1;
            }
            nbgroup_ = (int)g.size() / 4;
            if (g.empty()) g.assign(4, 0);
            upload_vec(bgroup_, g, st_);
        }
        // the work vectors live in LDS when they fit (one wave = one workgroup; the whole 160 KB of a CU minus a margin)
        int dev_lds = 0, ncu = 0;
        PQ_HIP(hipDeviceGetAttribute(&dev_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev_));
        PQ_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev_));
        const size_t need = (size_t)N_ * sizeof(double);
        dev_lds -= 1024;  // (the kernels' few static words come out of the same budget)
        lds_y_ = lds_x_ = need <= (size_t)dev_lds && !debug_token("exact_no_lds");
        int per_cu = 1;
        if (lds_y_) {
            if (need > 48 * 1024) {
                static PerDeviceOnce once;  // (the limit is per device and monotone: set it to the device maximum once)
                once([&] {
                    PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ul_factor<true>), hipFuncAttributeMaxDynamicSharedMemorySize, dev_lds));
                    PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ul_solve<true>), hipFuncAttributeMaxDynamicSharedMemorySize, dev_lds));
                });
            }
            PQ_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_ul_factor<true>, 64, need));
        } else {
            PQ_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_ul_factor<false>, 64, 0));
            per_cu = std::min(per_cu, 4);
        }
        per_cu = std::max(1, per_cu);
        grid_ = std::max(1, std::min(nticket_, per_cu * ncu));  // every workgroup of the launch is resident: tasks are taken in row order and wait only for earlier ones
        if (const char* t = debug_token("exact_grid")) grid_ = std::max(1, std::min(grid_, std::atoi(t)));
        yglob_.alloc(lds_y_ ? 1 : (size_t)grid_ * (size_t)N_);
        xglob_.alloc(lds_x_ ? 1 : (size_t)std::max(N_, 1));
        stream_wait(st_);
    }
    void remap_values()
    {
        launch_remap_values(ops_.nzP(), mapP_.p, ops_.P_x(), vals_.p, st_);
        launch_remap_values(ops_.nzA(), mapA_.p, ops_.AT_x(), vals_.p, st_);
        launch_remap_values(ops_.nzG(), mapG_.p, ops_.GT_x(), vals_.p, st_);
        PQ_HIP(hipGetLastError());
        stream_wait(st_);
    }

    int dev_, n_ = 0, p_ = 0, m_ = 0, N_ = 0, nnzK_ = 0, ntask_ = 0, nticket_ = 0, grid_ = 1, nbgroup_ = 0, epoch_ = 0;
    bool lds_y_ = true, lds_x_ = true;
    hipStream_t st_ = nullptr;
    sparse::UpLooking U_;
    CscOperators ops_;
    DBuf<int> perm_, Cp_, Ci_, diag_pos_, mapP_, mapA_, mapG_, Lp_, Li_, Lcol_, Rp_, Rcol_, Rpos_, Rcnt_, Rtab_, tk_kind_, tk_id_, task_ptr_, task_rows_, row_task_, row_lane_, row_prev_, dep_ptr_, dep_, tab_ptr_, mask_ptr_, task_nU_, done_,
        p1done_, ready_, ctl_, bgroup_;
    DBuf<unsigned long long> Tmask_;
    DBuf<double> vals_, Lx_, D_, Dinv_, Ystash_, Pstash_, Dinit_, Lblock_, yglob_, xglob_;
    HBuf<int> ctl_h_;
    StageProfiler prof_;
};

}  // namespace

KKTSolverBase* make_exact_sparse_kkt(const pq_sparse_data* data, int device) { return new ExactSparseKKT(data, device); }

}  // namespace pq
