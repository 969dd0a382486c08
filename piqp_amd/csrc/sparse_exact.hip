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
#include <map>
#include <mutex>
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
// s - pr[0] - pr[1] - ... - pr[cnt - 1], one after the other (the lanes of pr in order): the ordered chains of the reference's loops (D[k] -= ..., x[j] -= ...).
// Unrolled by eight with a scalar trip count: two lane reads and one subtraction per term.
__device__ __forceinline__ double chain_sub(double s, const double pr, int cnt)
{
    cnt = __builtin_amdgcn_readfirstlane(cnt);
    int l = 0;
    for (; l + 8 <= cnt; l += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) s = __dsub_rn(s, readlane_d(pr, l + q));
    }
    for (; l < cnt; ++l) s = __dsub_rn(s, readlane_d(pr, l));
    return s;
}
// Hand-over between waves WITHOUT fences (MI355X_MICROARCH.md, workgroup dispatch / inter-workgroup visibility: an agent-scope acquire costs 1.7 us and a release
// 1.7 - 6.5 us per workgroup, several times that with more workgroups per CU -- more than a whole task of this engine): every word one wave writes for another is
// written and read with agent-scope 8-byte / 4-byte atomics (write-through `sc1` stores, `sc1` loads that bypass the reader's L1), the flag follows the payload after
// s_waitcnt vmcnt(0), and the consumer's loads follow its successful poll.  Nothing else in these kernels is written by one wave and read by another.
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// A word whose readers all run on the writer's XCD (round 5, per-XCD row queues): a plain store leaves the line in that XCD's L2, where the readers' L1-bypassing
// loads (ldw) find it after an L2 round trip; a write-through (sc1) store drops the line and the same readers go out to the fabric (MI355X_MICROARCH.md,
// "stores of each flavour").  Nothing on another XCD may read such a word before the launch ends.
__device__ __forceinline__ void stl(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int xcc_id()
{
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
    return x & 7;
}
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

// Device layout of the schedule (built once in build_device from sparse::UpLooking; a row costs a handful of dependent memory round trips, so everything a pass
// needs about a row sits in ONE record and everything about its entries in arrays that are contiguous per task):
//   rowrec[16 k ..]   es, en (the row's entries in the task-ordered entry space), cp0, cpn (its column of P K P'), task, W, lane, table offset, previous path row,
//                     dep0, depn (children outside the task), nU
//   taskrec[8 t ..]   first index in task_rows, W, nU, table offset
//   E4[e]             column i, first CSC entry of column i, entries of column i the row pass scatters (-1: a column of the task's own path), CSC position of L(k, i)
//   Etab[e], Emask[e] the entry's row in the task's table and that row's presence bits
// Ystash / Pstash are indexed in the same entry space.
// ---- condensed KKT modes (kkt_{eq,ineq,all}_eliminated.hpp update_kkt_*): the same kernels as sparse_kkt.hip keeps for the multifrontal engine, compiled here
// without FMA contraction; every sum in the reference's order (the CPU oracle's orc_sparse_cond.c restates them)
__global__ void k_ul_cond_diag(int n, int np, int nm, const int* __restrict__ diag_pos, const double* __restrict__ x_reg, double delta, const double* __restrict__ z_reg,
                               double* __restrict__ vals)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= n + np + nm) return;
    if (col < n) vals[diag_pos[col]] += x_reg[col];
    else if (col < n + np) vals[diag_pos[col]] = -delta;
    else vals[diag_pos[col]] = -z_reg[col - n - np];
}
// value of every entry of upper(MT diag(1/w) MT^T) from its product-term list (constraints ascending, the reference's order); w == nullptr: unit weights
template <bool MAPPED>
__global__ void k_ul_gram_values(int nent, const int* __restrict__ ptr, const int* __restrict__ q1, const int* __restrict__ q2, const int* __restrict__ kk,
                                 const double* __restrict__ x, const double* __restrict__ w, const int* __restrict__ dst, double* __restrict__ out)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nent) return;
    double s = 0.0;
    if (w) for (int t = ptr[e]; t < ptr[e + 1]; ++t) s += x[q2[t]] * x[q1[t]] / w[kk[t]];
    else for (int t = ptr[e]; t < ptr[e + 1]; ++t) s += x[q2[t]] * x[q1[t]];
    if (MAPPED) out[dst[e]] += s; else out[e] = s;
}
__global__ void k_ul_axpy_mapped(int nent, const int* __restrict__ dst, double alpha, const double* __restrict__ src, double* __restrict__ out)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < nent) out[dst[e]] += alpha * src[e];
}
__global__ void k_ul_reciprocal(int m, const double* __restrict__ z, double* __restrict__ zinv)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) zinv[i] = 1.0 / z[i];
}

struct UlFactorArgs {
    int N, nticket, epoch;
    const int *Cp, *Ci;
    const double* Cx;
    const int *tk_kind, *tk_id, *task_rows, *rowrec, *taskrec, *dep;
    const int4* E4;
    const int *Etab, *Li;
    const unsigned long long* Emask;
    double *Lx, *D, *Dinv, *Ystash, *Pstash, *Dinit, *Lblock;
    int *done, *p1done, *ticket, *info;
    const int *xq_ptr, *xq_rows;  // per-XCD row queues (nullable: one queue of tickets): XCD q works rows xq_rows[xq_ptr[q] .. xq_ptr[q + 1]), ascending; a.ticket + 16 q counts
    double* Dloc;                 // with the queues: D once more, stored plainly for the rows of the same task (same XCD) to poll
    int rowpar;     // 1: the path pass of a row follows its row pass on the same wave (ul_path_row); 0: one path pass per task (ul_path; PIQP_AMD_DEBUG=exact_serial_path)
    double* yglob;  // N doubles per workgroup when y does not fit LDS
    long long* trace;  // debugging aid (PIQP_AMD_DEBUG=exact_trace), nullable: per ticket 4 x wall_clock64 (100 MHz): drawn, waits over, done; [3] = workgroup
};

constexpr int UL_PF = 8;    // row pass: columns prefetched ahead of the dependent chain (their first 64 entries); a load of a word another wave wrote comes from memory,
                            // not from L2 (write-through stores drop the line): ~2 us, i.e. the pace is latency / depth until the depth covers it
constexpr int UL_PFP = 8;   // path pass / substitution: table values fetched ahead

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


struct UlSolve2Args {
    int N, n, p, m, ntask, epoch;
    const int *perm, *taskrec, *task_rows, *tsort, *tdep, *fs_u, *fs_col, *Lp, *Li, *Lsrc;
    const int4* fs4;  // forward pass, per table row of a task in ascending column order: {table row, column, mask of the task's rows with an entry (lo, hi)}
    const int* fs_task;  // ... and the task that column belongs to (its forward pass must be complete before the column's x is read)
    const unsigned long long* Tmask;
    const int* mask_ptr;
    const double *Lblock, *Lx, *Dinv;
    const double *rx, *ry, *rz;
    double *lx, *ly, *lz;
    double *xf, *xz, *xb;
    int *fdone, *bdone, *ticket, *info;
    int fwd_only;
    const int *ta_ptr, *ta_rows, *Lsrc2;  // backward pass: per task the rows above it that its columns touch; per entry of L its operand (lane of the task, or 64 + list index)
    int xa_cap;                           // doubles of LDS for the longest such list (a multiple of 64)
    long long* trace;                     // debugging aid (PIQP_AMD_DEBUG=exact_trace), nullable: per ticket 4 x wall_clock64: drawn, waits over, done; [3] = workgroup
};

__device__ __forceinline__ double ldw(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stw(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ldf(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stf(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// bound of every wait of the engine: ~0.7 us a poll, so a few seconds -- the longest legitimate wait is a fraction of one factorisation (hundreds of milliseconds on the
// largest fixtures); what ends here is reported (info = -2), retried once on the single ticket queue by the factorisation, and thrown if it happens again
constexpr long long UL_SPIN_LIMIT = 1ll << 22;
__device__ __forceinline__ bool spin_until(const int* flag, int epoch)
{
    long long spins = 0;
    while (ldf(flag) != epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > UL_SPIN_LIMIT) return false;  // (a few seconds: a scheduling bug must not take the device with it)
    }
    return true;
}

// a word that has not been written in this factorisation: a quiet NaN no computation produces (hardware NaNs are canonical, propagated ones carry their
// operand's payload); k_ul_prepare puts it into D and into the tasks' tables before every factorisation
constexpr long long UL_SENT = 0x7ff8dead5eed0001ll;
// ... in ONE launch with the reset of the ticket counters and of the result word (five small operations on the stream before: ~20 us of a 60 us factorisation
// of a 59-row system)
__global__ void k_ul_prepare(size_t nd, double* __restrict__ D, size_t nl, double* __restrict__ Dloc, size_t nt, double* __restrict__ table, int* __restrict__ ctl,
                             int* __restrict__ xtick)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const double sent = __longlong_as_double(UL_SENT);
    if (i < nd) D[i] = sent;
    if (i < nl) Dloc[i] = sent;
    for (size_t j = i; j < nt; j += (size_t)gridDim.x * blockDim.x) table[j] = sent;
    if (i < 128 && xtick) xtick[i] = 0;
    if (i == 0) { ctl[0] = 0; ctl[1] = INT_MAX; }
}
__global__ void k_ul_poison_if_failed(const int* __restrict__ info, double* __restrict__ x, int n)
{
    if (*info <= -2 && (int)threadIdx.x < n) x[threadIdx.x] = __longlong_as_double(0x7ff8000000000000ll);
}
// which XCD does a workgroup of a launch of this shape land on?  (the per-XCD queues are only used when every one of the eight gets workgroups)
__global__ void k_ul_xcd_probe(int* __restrict__ count)
{
    if (threadIdx.x == 0) atomicAdd(count + xcc_id(), 1);
}
__device__ __forceinline__ bool poll_value(const double* p, double& out)
{
    long long spins = 0;
    for (;;) {
        const double v = ldw(p);
        if (__double_as_longlong(v) != UL_SENT) { out = v; return true; }
        __builtin_amdgcn_s_sleep(1);
        if (++spins > UL_SPIN_LIMIT) return false;
    }
}

// ROW PASS of row k (ldlt.hpp:121-163): the entries of row k in the columns outside the row's task, in the reference's order.  A row that is a task of its own is
// finished here; for a row on a longer path the updates into the path's rows are left to the path pass (E4.z counts only the entries of a column above them), the
// values y_i, the products l_ki y_i and the initial values of the path columns go to Ystash / Pstash / Dinit, the quotients also into the task's table.
// y: this wave's dense work vector, all zero on entry and on exit.
// wait(): the row's dependencies (rows below it outside its task) -- called after the loads that do not depend on them were issued: the row's own entries of K, the
// first 64 records of its pattern.
template <class Wait>
__device__ __forceinline__ double ul_row(const UlFactorArgs& a, double* __restrict__ y, const int k, const int lane, const int es, const int en, const int p0, const int pn,
                                         const int W, const int lanek, const int tb, Wait wait, bool& ok)
{
    const bool multi = W > 1;
    ok = true;
    bool waited = false;
    // scatter A(0:k, k) into y (:127-131)
    for (int q = lane; q < pn; q += 64) y[a.Ci[p0 + q]] = a.Cx[p0 + q];
    wave_sync();
    double Dk = y[k];  // :145  D[k] = y[k]
    wave_sync();
    if (lane == 0) y[k] = 0.0;
    for (int base = 0; base < en; base += 64) {
        const int e = es + base + lane;
        const bool in = base + lane < en;
        const int4 ev = a.E4[in ? e : es];
        const int i = in ? ev.x : 0, pos = ev.w;
        const int cs = in ? ev.y : 0;
        const int rc = in ? ev.z : 0;        // entries of column i this pass scatters: all above row k (= L_nnz[i] at this moment, :149) or those above the task's path
        const int cnt = rc > 0 ? rc : 0;
        const bool ext = in && rc >= 0;      // (a column of the task's own path otherwise: its value in this row comes out of the path pass)
        const int tabu = a.Etab[in ? e : es];
        if (!waited) { waited = true; if (!wait()) { ok = false; return 0.0; } }
        const double Dld = ldw(a.D + i);           // (unconditional: issued together with the column prefetch below)
        const int ns = min(64, en - base);
        double my_yi = 0.0;
        // the first 64 entries of the columns of the next UL_PF steps travel ahead of the chain.  Every load is unconditional (lanes past the end of their
        // column re-read its first entry, steps past the end of the row read entry 0 of the arrays): a load under a branch would be waited for at the branch
        int pf_i[UL_PF];
        double pf_v[UL_PF];
#pragma unroll
        for (int d = 0; d < UL_PF; ++d) {
            const int csu = __builtin_amdgcn_readlane(cs, d), cntu = __builtin_amdgcn_readlane(cnt, d);
            const int q = csu + (lane < cntu ? lane : 0);
            pf_i[d] = a.Li[q]; pf_v[d] = ldw(a.Lx + q);
        }
        const double Di = ext ? Dld : 1.0;
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
                    pf_i[d] = a.Li[q]; pf_v[d] = ldw(a.Lx + q);
                }
                const double yi = y[iu];  // :147 (every lane reads the same word)
                if (lane == s) my_yi = yi;
                if (lane < cntu) y[t0] = msub(y[t0], v0, yi);  // :150-154, distinct targets
                if (lane == 0 && s < ns) y[iu] = 0.0;          // :148
                for (int q = 64 + lane; q < cntu; q += 64) { const int tt = a.Li[csu + q]; y[tt] = msub(y[tt], ldw(a.Lx + csu + q), yi); }
                wave_sync();
            }
        }
        // :155-161 for the 64 entries at once; D[k] loses its terms strictly in pattern order
        const double l = __ddiv_rn(my_yi, Di);
        const double tp = __dmul_rn(l, my_yi);
        if (ext) stw(a.Lx + pos, l);
        if (ext) { if (a.xq_ptr) stl(a.Lblock + tb + tabu * W + lanek, l); else stw(a.Lblock + tb + tabu * W + lanek, l); }  // (the task's table holds every entry of its rows: the substitution reads L from there; its readers inside this launch are rows of the same task)
        if (multi) {
            if (in) stw(a.Ystash + e, my_yi);
            if (ext) stw(a.Pstash + e, tp);
        } else {
            Dk = chain_sub(Dk, tp, ns);
        }
    }
    if (!waited && !wait()) { ok = false; return 0.0; }
    if (lane == 0) {
        if (multi) stw(a.Dinit + k, Dk);
        else {
            stw(a.D + k, Dk);
            a.Dinv[k] = __ddiv_rn(1.0, Dk);  // :166
            if (Dk == 0.0) atomicMin(a.info, k);  // :163 (the smallest such k is the row the serial loop stops at)
        }
    }
    return Dk;
}

// PATH PASS of task t: its rows one after the other, the path rows as lanes.  For row k the pattern is walked once more in the reference's order; an entry in an
// outside column i sends y_i (from the row pass) to the path rows c < k that hold an entry L(c, i) -- acc_c -= fl(L(c, i) y_i) -- and an entry in a path column c0
// first has its value final (acc of lane c0: every term it receives comes from a column the order visits earlier) and then does the same.  The quotients and the
// terms of D[k] follow, the latter again in pattern order.  L(c, .) is read from the task's table, where the passes that produced it left it.
// The pass starts when ALL row passes of the task are complete: from then on nothing it reads changes except what it writes itself, and the entries of row j + 2
// and the table values of the first steps of row j + 1 are fetched while row j is computed (rows of three entries are common: memory round trips, not steps,
// would otherwise set the pace).
struct UlChunk { int tab, mlo, mhi, pos; double ys, ps; };

__device__ __forceinline__ UlChunk ul_load_chunk(const UlFactorArgs& a, int e0, int cnt, int fallback, int lane)
{
    const bool in = lane < cnt;
    const int e = in ? e0 + lane : fallback;
    UlChunk c;
    const int tab = a.Etab[e];
    const unsigned long long m = a.Emask[e];
    const double ys = ldw(a.Ystash + e), ps = ldw(a.Pstash + e);
    const int pos = a.E4[e].w;
    c.tab = in ? tab : 0; c.mlo = in ? (int)(unsigned)(m & 0xffffffffull) : 0; c.mhi = in ? (int)(unsigned)(m >> 32) : 0; c.pos = pos;
    c.ys = in ? ys : 0.0; c.ps = in ? ps : 0.0;
    return c;
}

__device__ __forceinline__ bool ul_path(const UlFactorArgs& a, const int t, const int lane, double* s_acc, int* s_pos)
{
    const int rb = a.taskrec[8 * t], W = a.taskrec[8 * t + 1], nU = a.taskrec[8 * t + 2], tb = a.taskrec[8 * t + 3];
    const int lw = lane < W ? lane : W - 1;
    const int myk = a.task_rows[rb + lw];
    bool ok = spin_until(a.p1done + myk, a.epoch);
    if (__ballot(!ok)) return false;
    // lane j <-> row j of the task
    const int es_l = a.rowrec[16 * myk], en_l = a.rowrec[16 * myk + 1];
    const double dinit_l = ldw(a.Dinit + myk);
    const int E0 = __builtin_amdgcn_readlane(es_l, 0);
    double Dlane = 1.0;  // lane c: D of path row c once it is known
    // pipeline registers: the first 64 entries of rows j (cur), j + 1 (nx1), j + 2 (nx2); the table values of the first UL_PFP steps of rows j (pfc) and j + 1 (pfn)
    UlChunk cur = ul_load_chunk(a, E0, min(64, __builtin_amdgcn_readlane(en_l, 0)), E0, lane);
    UlChunk nx1 = W > 1 ? ul_load_chunk(a, __builtin_amdgcn_readlane(es_l, 1), min(64, __builtin_amdgcn_readlane(en_l, 1)), E0, lane) : cur;
    double pfc[UL_PFP], pfn[UL_PFP];
#pragma unroll
    for (int d = 0; d < UL_PFP; ++d) pfc[d] = ldw(a.Lblock + tb + __builtin_amdgcn_readlane(cur.tab, d) * W + lw);
    for (int j = 0; j < W; ++j) {
        const int k = __builtin_amdgcn_readlane(myk, j), es = __builtin_amdgcn_readlane(es_l, j), en = __builtin_amdgcn_readlane(en_l, j);
        // fetch ahead: entries of row j + 2, first table values of row j + 1
        UlChunk nx2 = nx1;
        if (j + 2 < W) nx2 = ul_load_chunk(a, __builtin_amdgcn_readlane(es_l, j + 2), min(64, __builtin_amdgcn_readlane(en_l, j + 2)), E0, lane);
#pragma unroll
        for (int d = 0; d < UL_PFP; ++d) pfn[d] = ldw(a.Lblock + tb + __builtin_amdgcn_readlane(nx1.tab, d) * W + lw);
        // the initial values of the path columns (A(c, k), left in Ystash by the row pass) go to their lanes through LDS
        s_acc[lane] = 0.0; s_pos[lane] = -1;
        wave_sync();
        if (lane < en && cur.tab >= nU) { s_acc[cur.tab - nU] = cur.ys; s_pos[cur.tab - nU] = cur.pos; }
        for (int e = es + 64 + lane; e < es + en; e += 64) {
            const int u = a.Etab[e];
            if (u >= nU) { s_acc[u - nU] = ldw(a.Ystash + e); s_pos[u - nU] = a.E4[e].w; }
        }
        wave_sync();
        double acc = s_acc[lane];
        const int mypos = s_pos[lane];
        wave_sync();
        const unsigned long long below = j >= 64 ? ~0ull : ((1ull << j) - 1ull);  // path rows under row k
        UlChunk ch = cur;
        for (int base = 0; base < en; base += 64) {
            const int ns = min(64, en - base);
            double pf_v[UL_PFP];
            if (base == 0) {
#pragma unroll
                for (int d = 0; d < UL_PFP; ++d) pf_v[d] = pfc[d];
            } else {
                ch = ul_load_chunk(a, es + base, ns, E0, lane);
#pragma unroll
                for (int d = 0; d < UL_PFP; ++d) pf_v[d] = ldw(a.Lblock + tb + __builtin_amdgcn_readlane(ch.tab, d) * W + lw);
            }
            for (int sb = 0; sb < ns; sb += UL_PFP) {
#pragma unroll
                for (int d = 0; d < UL_PFP; ++d) {
                    const int s = sb + d;
                    if (s < ns) {  // (a row of three entries runs three steps, not eight)
                        const int u = __builtin_amdgcn_readlane(ch.tab, s & 63);
                        const unsigned long long m = (((unsigned long long)(unsigned)__builtin_amdgcn_readlane(ch.mhi, s & 63) << 32) | (unsigned)__builtin_amdgcn_readlane(ch.mlo, s & 63)) & below;
                        const double v = pf_v[d];
                        pf_v[d] = ldw(a.Lblock + tb + __builtin_amdgcn_readlane(ch.tab, (s + UL_PFP) & 63) * W + lw);
                        const double src = u < nU ? readlane_d(ch.ys, s & 63) : readlane_d(acc, (u - nU) & 63);
                        if (__builtin_amdgcn_inverse_ballot_w64(m)) acc = msub(acc, v, src);
                    }
                }
            }
        }
        // quotients of the path columns (:155), their places in L and in the table, their terms of D[k]
        const bool have = lane < j && mypos >= 0;
        const double l = __ddiv_rn(acc, Dlane);
        const double prodp = __dmul_rn(l, acc);
        if (have) { stw(a.Lx + mypos, l); stw(a.Lblock + tb + (nU + lane) * W + j, l); }
        double Dk = readlane_d(dinit_l, j);
        ch = cur;
        for (int base = 0; base < en; base += 64) {
            const int ns = min(64, en - base);
            if (base > 0) ch = ul_load_chunk(a, es + base, ns, E0, lane);
            for (int s = 0; s < ns; ++s) {
                const int u = __builtin_amdgcn_readlane(ch.tab, s);
                const double term = u < nU ? readlane_d(ch.ps, s) : readlane_d(prodp, (u - nU) & 63);
                Dk = __dsub_rn(Dk, term);
            }
        }
        if (lane == 0) {
            stw(a.D + k, Dk);
            a.Dinv[k] = __ddiv_rn(1.0, Dk);
            if (Dk == 0.0) atomicMin(a.info, k);
        }
        if (lane == j) Dlane = Dk;
        // the table values of row j + 1's first steps were fetched before this row wrote its own quotients L(path row j, path column c) into the table: lane j
        // (the newest path row under row j + 1) takes them from the registers they were computed in
        const int en_next = j + 1 < W ? __builtin_amdgcn_readlane(en_l, j + 1) : 0;
#pragma unroll
        for (int d = 0; d < UL_PFP; ++d) {
            if (d < en_next) {
                const int un = __builtin_amdgcn_readlane(nx1.tab, d);
                if (un >= nU) { const double lv = readlane_d(l, (un - nU) & 63); if (lane == j) pfn[d] = lv; }
            }
        }
        cur = nx1; nx1 = nx2;
#pragma unroll
        for (int d = 0; d < UL_PFP; ++d) pfc[d] = pfn[d];
    }
    return true;
}

// PATH PASS of ONE row, run by the wave that has just finished the row's row pass (round 5, second form: the rows of a task advance side by side instead of one
// after the other).  Same arithmetic as ul_path, entry for entry; what a row needs from the rows below it on the path arrives through memory:
//   * before the first step: the row passes of the rows below are complete (their quotients L(c, i) in outside columns i are in the table);
//   * at a step whose source is path column c0: D of row c0 (that row is complete), and for every path row c between c0 and this row that holds an entry
//     L(c, c0): row c has published it (prog[c] counts the path columns a row has published; a complete row counts 127);
//   * this row publishes its own quotient L(k, c0) and bumps its counter at once, so that the rows above can take their c0 step.
// A row's pattern in front of its c0 entry is (about) the pattern of row c0, so rows that run side by side reach their waits about when the word arrives.
__device__ __forceinline__ bool ul_path_row(const UlFactorArgs& a, const int k, const int lane, const int es, const int en, const int W, const int j, const int tb,
                                            const int nU, const int rb, double Dk, double* s_acc, int* s_pos)
{
    const int lw = lane < W ? lane : W - 1;
    const int rowl = a.task_rows[rb + lw];        // lane c: the path row c
    bool ok = lane < j ? spin_until(a.p1done + rowl, a.epoch) : true;
    if (__ballot(!ok)) return false;
    s_acc[lane] = 0.0;
    wave_sync();
    for (int e = es + lane; e < es + en; e += 64) {
        const int u = a.Etab[e];
        if (u >= nU) s_acc[u - nU] = ldw(a.Ystash + e);   // the initial values of the path columns, A(c, k), left by the row pass
    }
    wave_sync();
    double acc = s_acc[lane];
    wave_sync();
    (void)s_pos;
    const unsigned long long below = j >= 64 ? ~0ull : ((1ull << j) - 1ull);  // path rows under row k
    for (int base = 0; base < en; base += 64) {
        const int ns = min(64, en - base);
        const UlChunk ch = ul_load_chunk(a, es + base, ns, es, lane);
        double pf_v[UL_PFP];
#pragma unroll
        for (int d = 0; d < UL_PFP; ++d) pf_v[d] = ldw(a.Lblock + tb + __builtin_amdgcn_readlane(ch.tab, d) * W + lw);
        for (int sb = 0; sb < ns; sb += UL_PFP) {
#pragma unroll
            for (int d = 0; d < UL_PFP; ++d) {
                const int s = sb + d;
                if (s < ns) {  // (a row of three entries runs three steps, not eight: what follows its last step is on the chain of hand-overs)
                    const int u = __builtin_amdgcn_readlane(ch.tab, s & 63);
                    const unsigned long long m = (((unsigned long long)(unsigned)__builtin_amdgcn_readlane(ch.mhi, s & 63) << 32) | (unsigned)__builtin_amdgcn_readlane(ch.mlo, s & 63)) & below;
                    double v = pf_v[d];
                    pf_v[d] = ldw(a.Lblock + tb + __builtin_amdgcn_readlane(ch.tab, (s + UL_PFP) & 63) * W + lw);
                    const bool bit = __builtin_amdgcn_inverse_ballot_w64(m);
                    double src, term;
                    if (u < nU) {
                        src = readlane_d(ch.ys, s & 63);
                        term = readlane_d(ch.ps, s & 63);
                    } else {
                        // A path column c0.  What this row needs from the rows below arrives AS VALUES (round 5): D of row c0 and, for the rows c between c0 and
                        // this one that hold an entry in column c0, their quotient L(c, c0) -- words that hold UL_SENT until their row stores them (k_ul_prepare
                        // before every factorisation), polled with the same loads that fetch them: one round trip per hand-over instead of flag, value, drained
                        // store, flag, value.  This row's own quotient goes out the same way, at once.
                        const int c0 = u - nU;
                        src = readlane_d(acc, c0 & 63);
                        const int row0 = __builtin_amdgcn_readlane(rowl, c0 & 63);
                        double D0;
                        if (!poll_value((a.xq_ptr ? a.Dloc : a.D) + row0, D0)) return false;
                        const double l = __ddiv_rn(src, D0);
                        term = __dmul_rn(l, src);
                        if (a.xq_ptr) stl(a.Lblock + tb + u * W + j, l); else stw(a.Lblock + tb + u * W + j, l);
                        stw(a.Lx + __builtin_amdgcn_readlane(ch.pos, s & 63), l);
                        // (measured and not kept: D and the quotients requested eight steps ahead with the table row and polled only while they still hold the
                        // sentinel -- the extra loads cost more than the saved round trips: chain-mass 2.32 -> 2.48 ms, STADAT1 6.7 -> 7.3)
                        double vv = 0.0;
                        const bool okw = bit ? poll_value(a.Lblock + tb + u * W + lw, vv) : true;
                        if (__ballot(!okw)) return false;
                        v = vv;
                    }
                    if (bit) acc = msub(acc, v, src);
                    Dk = __dsub_rn(Dk, term);
                }
            }
        }
    }
    if (lane == 0) {
        if (a.xq_ptr) stl(a.Dloc + k, Dk);   // (the rows above poll this word)
        stw(a.D + k, Dk);
        a.Dinv[k] = __ddiv_rn(1.0, Dk);
        if (Dk == 0.0) atomicMin(a.info, k);
    }
    drain_stores();  // (the task's last row is followed by its `done` word, which other tasks take as "everything of this row is there")
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
    // Per-XCD queues (round 5): the rows of a task are all in the queue of ONE XCD and a workgroup only draws from the queue of the XCD it runs on, so the words the
    // rows of a task hand to one another stay in that XCD's L2.  Every queue is in ascending row order: the smallest unfinished row of the whole matrix has been
    // drawn in its queue (everything before it there is finished) and waits for nothing -- progress never depends on another queue's workgroups.
    const int xq = a.xq_ptr ? xcc_id() : 0;
    const int qbeg = a.xq_ptr ? a.xq_ptr[xq] : 0, qcnt = a.xq_ptr ? a.xq_ptr[xq + 1] - qbeg : a.nticket;
    int* const tcount = a.ticket + (a.xq_ptr ? 16 * xq : 0);
    if (lane == 0) s_task = atomicAdd(tcount, 1);
    __syncthreads();
    // (the whole workgroup is one wave: __syncthreads() costs nothing and keeps the control flow around the ticket uniform for the compiler)
    for (int guard = 0; guard <= a.nticket; ++guard) {
        const int tq = readfirst(s_task);
        __syncthreads();
        if (tq >= qcnt) break;
        const int tk = qbeg + tq;
        // (the next ticket is drawn when this one is done: a ticket drawn ahead sits parked for as long as the current task takes -- with rows that wait for
        // one another side by side that was 11 ms on CONT-050, the rows above spinning for a row whose ticket nobody worked on)
        const int kind = a.xq_ptr ? 0 : a.tk_kind[tk], id = a.xq_ptr ? a.xq_rows[tk] : a.tk_id[tk];
        bool ok = true;
        if (a.trace && lane == 0) { a.trace[4 * (size_t)tk] = wall_clock64(); a.trace[4 * (size_t)tk + 3] = blockIdx.x; }
        if (kind == 0) {
            // row pass: the row's children outside its task must be complete rows; the rows below it on its own path are not waited for -- what THEY waited for is
            // inherited through the `ready` word of the row before
            const int k = id;
            const int4 r0 = *reinterpret_cast<const int4*>(a.rowrec + 16 * k), r1 = *reinterpret_cast<const int4*>(a.rowrec + 16 * k + 4),
                       r2 = *reinterpret_cast<const int4*>(a.rowrec + 16 * k + 8);
            const int es = r0.x, en = r0.y, cp0 = r0.z, cpn = r0.w, W = r1.y, lanek = r1.z, tb = r1.w, prev = r2.x, c0 = r2.y, cn = r2.z, nU = r2.w;
            const int serial_task = a.rowrec[16 * k + 12];
            (void)prev;
            const double Dk0 = ul_row(a, y, k, lane, es, en, cp0, cpn, W, lanek, tb, [&]() {
                bool w = true;
                for (int c = lane; c < cn; c += 64) w &= spin_until(a.done + a.dep[c0 + c], a.epoch);
                if (a.trace && lane == 0) a.trace[4 * (size_t)tk + 1] = wall_clock64();
                return __ballot(!w) == 0;
            }, ok);
            if (ok) {
                drain_stores();
                int* flag = (W > 1 ? a.p1done : a.done) + k;
                stf(flag, a.epoch);  // (every lane stores the same word: no divergence at the loop's end)
                if (W > 1 && !serial_task) {
                    const int rb = a.taskrec[8 * r1.x];
                    ok = ul_path_row(a, k, lane, es, en, W, lanek, tb, nU, rb, readlane_d(Dk0, 0), s_acc, s_pos);
                    if (ok && lanek == W - 1) stf(a.done + k, a.epoch);  // (ul_path_row has drained its stores)
                }
            }
        } else if (!a.rowrec[16 * a.task_rows[a.taskrec[8 * id]] + 12]) {
            // (the rows of this task did their path pass themselves, side by side)
        } else {
            if (a.trace && lane == 0) a.trace[4 * (size_t)tk + 1] = wall_clock64();
            ok = ul_path(a, id, lane, s_acc, s_pos);
            if (ok) {
                drain_stores();
                stf(a.done + a.task_rows[a.taskrec[8 * id] + a.taskrec[8 * id + 1] - 1], a.epoch);
            }
        }
        if (!ok) { if (lane == 0) atomicMin(a.info, -2); break; }
        if (a.trace && lane == 0) a.trace[4 * (size_t)tk + 2] = wall_clock64();
        if (lane == 0) s_task = atomicAdd(tcount, 1);
        __syncthreads();
    }
}

// ---- substitution on the factorisation's tasks (lsolve, dsolve, ltsolve of ldlt.hpp:171-218 with ordering.perm / permt of sparse/kkt.hpp:139-144 folded in).
// One persistent launch: tickets 0 .. ntask-1 are the forward passes of the tasks (in the order of their last rows), ntask .. 2 ntask-1 the backward passes in the
// opposite order; a wave never waits for a later ticket.  The vectors live in HBM: xf (forward result), xz (xf scaled by D_inv), xb (final).
//   forward pass of a task: its rows as lanes, acc = permuted right-hand side; the task's table rows (the columns that reach it, outside ones and its own) in
//     ascending column order: acc_t -= fl(L(t, j) x_j) -- the order in which the reference's column loop subtracts from x_t;
//   backward pass: its columns from the last to the first; the entries of column j in ascending row order: 64 products fl(L(t, j) x_t) across the lanes, subtracted
//     from x_j one after the other (the reference's inner loop); x_t of a row on the same path comes from its lane, of a row above from xb.

// (everything that does not depend on other tasks -- the task record, its rows, the right-hand side, the first chunk of the table -- is requested BEFORE the wait
// for the tasks below: a hop of the dependency chain then costs the flag, one load of x and the arithmetic, not five dependent loads on top)
// Chunks of 64 table rows; the records of the chunk AFTER the current one ({table row, column, mask of the task rows that hold an entry}: one 16-byte load per
// lane), its x values and -- from the middle of the current chunk on -- its table values are requested while the current chunk is computed: after the first
// chunk no memory round trip is left on the chain of steps.
template <int PFS, class Wait>
__device__ __forceinline__ bool ul_fwd_task(const UlSolve2Args& a, const int t, const int lane, Wait wait)
{
    static_assert(PFS == 32, "the refill below assumes half a chunk");
    const int4 r0 = *reinterpret_cast<const int4*>(a.taskrec + 8 * t);
    const int rb = r0.x, W = r0.y, nU = r0.z, tb = r0.w, fs0 = a.taskrec[8 * t + 4];
    const int nsrc = nU + W;
    const int lw = lane < W ? lane : W - 1;
    const int row = a.task_rows[rb + lw];
    const int o = a.perm[row];
    double acc = o < a.n ? a.rx[o] : (o < a.n + a.p ? a.ry[o - a.n] : a.rz[o - a.n - a.p]);
    const double dinv = a.Dinv[row];
    int4 rec = a.fs4[fs0 + (lane < nsrc ? lane : 0)];
    int rt = a.fs_task[fs0 + (lane < nsrc ? lane : 0)];
    if (lane >= nsrc) { rec.x = 0; rec.z = 0; rec.w = 0; rt = t; }  // (steps past the end: empty masks -- they run, and change nothing)
    const double* __restrict__ tbase = a.Lblock + tb;        // row u of the task's table: tbase + u W, this lane's value at + lw
    // one step: acc_t -= fl(L(t, j) x_j) for the task rows t of the mask.  No branch: both candidate operands are read (the x of an outside column from its
    // lane of xc, the value of a path column from its lane of acc) and one is selected.
    auto step = [&](const int ue, const int mlo, const int mhi, const double xc, const int sl, const double v) {
        const int u = __builtin_amdgcn_readlane(ue, sl);
        const unsigned long long m = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(mhi, sl) << 32) | (unsigned)__builtin_amdgcn_readlane(mlo, sl);
        const double xo = readlane_d(xc, sl), xa = readlane_d(acc, (u - nU) & 63);
        const double src = u < nU ? xo : xa;
        const double r = msub(acc, v, src);
        acc = __builtin_amdgcn_inverse_ballot_w64(m) ? r : acc;
    };
    // A column's x is read when the forward pass of ITS task is complete -- waited for column by column, chunk by chunk, not for all the tasks below at the start:
    // most of a big task's columns belong to tasks that finished long ago, and the task right below it on the tree (the one it would wait for longest) holds the
    // LAST outside columns of the list.  The top of the tree then overlaps: a task works through its early chunks while the task below is still running.
    auto wait_sources = [&](const int src_task) { const bool w = src_task == t ? true : spin_until(a.fdone + src_task, a.epoch); return __ballot(!w) == 0; };
    if (nsrc <= 8) {
        // most tasks are a row or two with a handful of columns: eight steps, nothing in flight behind them
        double tv[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) tv[d] = (tbase + (size_t)__builtin_amdgcn_readlane(rec.x, d) * W)[lw];
        if (!wait() || !wait_sources(rt)) return false;
        const double xs8 = ldw(a.xf + rec.y);
#pragma unroll
        for (int d = 0; d < 8; ++d) step(rec.x, rec.z, rec.w, xs8, d, tv[d]);
        if (lane < W) { stw(a.xf + row, acc); stw(a.xz + row, __dmul_rn(acc, dinv)); }
        return true;
    }
    auto fetch = [&](const int base, int4& r, int& rtask) {
        const int q = base + lane;
        const bool live = q < nsrc;
        r = a.fs4[fs0 + (live ? q : 0)];
        rtask = a.fs_task[fs0 + (live ? q : 0)];
        if (!live) { r.x = 0; r.z = 0; r.w = 0; rtask = t; }
    };
    // Pipeline over the chunks of 64 columns: records three chunks ahead, the "done" words of a chunk's tasks two chunks ahead, its x values one chunk ahead -- and
    // the x values ONLY after every one of those words has been SEEN set (a load of x issued behind a load of the word that has not returned yet may be served
    // before it: the word would say "complete" about an x read too early; found by two netlib fixtures whose solves differed in the last bits).  A chunk with a
    // task still running is waited for after the current chunk's steps.
    int4 nrec, nnrec; int nrt, nnrt;
    fetch(64, nrec, nrt);
    fetch(128, nnrec, nnrt);
    double pf_v[PFS];
#pragma unroll
    for (int d = 0; d < PFS; ++d) pf_v[d] = (tbase + (size_t)__builtin_amdgcn_readlane(rec.x, d) * W)[lw];
    int nflag = ldf(a.fdone + nrt);
    if (!wait() || !wait_sources(rt)) return false;
    double xs = ldw(a.xf + rec.y);   // (a path column's value is taken from its lane instead)
    for (int base = 0; base < nsrc; base += 64) {
        const int ns = min(64, nsrc - base);
        const int ue = rec.x, mlo = rec.z, mhi = rec.w;
        const double xcur = xs;
        const bool spec = __ballot(nrt != t && nflag != a.epoch) == 0;  // (nflag was requested a whole chunk ago: it is here)
        double nxs = 0.0;
        if (spec) nxs = ldw(a.xf + nrec.y);
        const int nnflag = ldf(a.fdone + nnrt);
        int4 n3rec; int n3rt;
        fetch(base + 192, n3rec, n3rt);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half * 32 < ns) {
#pragma unroll
                for (int d = 0; d < PFS; ++d) {
                    const int sl = half * 32 + d;
                    const double v = pf_v[d];
                    // refill: the table row 32 steps ahead -- of this chunk (first half) or of the next one (second half)
                    pf_v[d] = (tbase + (size_t)__builtin_amdgcn_readlane(half == 0 ? ue : nrec.x, (sl + 32) & 63) * W)[lw];
                    step(ue, mlo, mhi, xcur, sl, v);
                }
            }
        }
        if (!spec && base + 64 < nsrc) {
            if (!wait_sources(nrt)) return false;
            nxs = ldw(a.xf + nrec.y);
        }
        rec = nrec; rt = nrt; xs = nxs; nrec = nnrec; nrt = nnrt; nflag = nnflag; nnrec = n3rec; nnrt = n3rt;
    }
    if (lane < W) { stw(a.xf + row, acc); stw(a.xz + row, __dmul_rn(acc, dinv)); }
    return true;
}

// s - p[0] - ... - p[NT - 1] with the terms in LDS (every lane reads the same words: broadcasts): ALL the 16-byte reads are requested first, then the NT subtractions
// run one after the other.  Inside this kernel an LDS read comes back after ~300 cycles, not the ~100 of an idle compute unit: a loop that reads eight terms per round
// and waits for them spent five times the subtractions' own time waiting (43 cycles per term measured against 12 in isolation, tools/ub/chain.hip).
template <int NT>
__device__ __forceinline__ double lds_chain_straight(double s, const double* __restrict__ p)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 t[NT / 2];
#pragma unroll
    for (int q = 0; q < NT / 2; ++q) t[q] = *reinterpret_cast<const d2*>(p + 2 * q);
#pragma unroll
    for (int q = 0; q < NT / 2; ++q) { s = __dsub_rn(s, t[q].x); s = __dsub_rn(s, t[q].y); }
    return s;
}
// s - p[0] - ... - p[cnt - 1]; the caller has stored +0.0 in p[cnt .. 63] (x - 0.0 == x for every x, signed zeros and NaNs included), so the chain may run over the
// next multiple of 16 terms
__device__ __forceinline__ double lds_chain_sub(double s, const double* __restrict__ p, int cnt)
{
    if (cnt <= 16) return lds_chain_straight<16>(s, p);
    if (cnt <= 32) return lds_chain_straight<32>(s, p);
    if (cnt <= 48) return lds_chain_straight<48>(s, p);
    return lds_chain_straight<64>(s, p);
}

// Backward pass of a task (round 5, second form).  What it reads from OTHER tasks are the final x of the rows above it that its columns touch -- a short list per
// task (the structure of its top row): fetched ONCE, into LDS (s_xa), so that a column's operands are lane shuffles and LDS reads and the only loads per column are
// its own entries (Lsrc2: 0 .. 63 = a row of this task by lane, 64 + i = entry i of the task's list), requested UL_PB columns ahead.
constexpr int UL_LANE_CHAIN = 12;  // ordered chains up to this length run on lane reads (15-21 cycles a term), longer ones through LDS (~300 cycles + 7 a term)
constexpr int UL_PB = 4;   // columns whose first 64 entries are in flight
constexpr int UL_AR = 5;   // list entries per lane fetched before the wait (lists of up to 320 rows; longer ones finish in a loop)
template <class Wait>
__device__ __forceinline__ bool ul_bwd_task(const UlSolve2Args& a, const int t, const int lane, Wait wait, double* __restrict__ s_xa, double* __restrict__ s_pr)
{
    const int rb = a.taskrec[8 * t], W = readfirst(a.taskrec[8 * t + 1]);
    const int lw = lane < W ? lane : W - 1;
    const int row = a.task_rows[rb + lw];
    const int lp0 = a.Lp[row], lp1 = a.Lp[row + 1];
    const int o = a.perm[row];
    const int ta0 = readfirst(a.ta_ptr[t]), nA = readfirst(a.ta_ptr[t + 1]) - ta0;
    int r_sl[UL_PB];
    double r_v[UL_PB];
#pragma unroll
    for (int d = 0; d < UL_PB; ++d) {
        const int c = W - 1 - d, cc = c >= 0 ? c : 0;
        const int p0 = __builtin_amdgcn_readlane(lp0, cc), p1 = __builtin_amdgcn_readlane(lp1, cc);
        const int qq = p0 + lane < p1 ? p0 + lane : (p1 > p0 ? p0 : 0);
        r_sl[d] = a.Lsrc2[qq]; r_v[d] = a.Lx[qq];
    }
    int arow[UL_AR];
#pragma unroll
    for (int k = 0; k < UL_AR; ++k) arow[k] = a.ta_rows[ta0 + (lane + 64 * k < nA ? lane + 64 * k : 0)];  // (one word of padding behind the last list)
    if (!wait()) return false;  // (the loads above do not depend on other tasks; the values below do)
    double xfin = ldw(a.xz + row);
#pragma unroll
    for (int k = 0; k < UL_AR; ++k) {
        const double xv = ldw(a.xb + arow[k]);
        if (lane + 64 * k < nA) s_xa[lane + 64 * k] = xv;
    }
    for (int idx = lane + 64 * UL_AR; idx < nA; idx += 64) s_xa[idx] = ldw(a.xb + a.ta_rows[ta0 + idx]);
    wave_sync();
    for (int cb = W - 1; cb >= 0; cb -= UL_PB) {
#pragma unroll
        for (int d = 0; d < UL_PB; ++d) {
            const int c = cb - d;
            if (c >= 0) {
                const int q0 = __builtin_amdgcn_readlane(lp0, c), q1 = __builtin_amdgcn_readlane(lp1, c);
                int sl = r_sl[d];
                double v = r_v[d];
                {   // this slot: the column UL_PB below
                    const int c2 = c - UL_PB, cc = c2 >= 0 ? c2 : 0;
                    const int p0 = __builtin_amdgcn_readlane(lp0, cc), p1 = __builtin_amdgcn_readlane(lp1, cc);
                    const int qq = p0 + lane < p1 ? p0 + lane : (p1 > p0 ? p0 : 0);
                    r_sl[d] = a.Lsrc2[qq]; r_v[d] = a.Lx[qq];
                }
                double s = readlane_d(xfin, c);
                for (int qb = q0; qb < q1; qb += 64) {
                    const int cnt = min(64, q1 - qb);
                    int nsl = 0;
                    double nv = 0.0;
                    if (qb + 64 < q1) { const int qq = qb + 64 + lane < q1 ? qb + 64 + lane : qb + 64; nsl = a.Lsrc2[qq]; nv = a.Lx[qq]; }  // the column's next 64 entries
                    const double xin = __shfl(xfin, sl < 64 ? sl : 0, 64);
                    const double xo = s_xa[sl >= 64 ? sl - 64 : 0];
                    const double pr = __dmul_rn(v, sl < 64 ? xin : xo);
                    if (cnt <= UL_LANE_CHAIN) s = chain_sub(s, pr, cnt);  // (short chains by lane reads: an LDS round trip costs more than a dozen of them)
                    else {
                        s_pr[lane] = lane < cnt ? pr : 0.0;
                        wave_sync();
                        s = lds_chain_sub(s, s_pr, cnt);
                        wave_sync();
                    }
                    sl = nsl; v = nv;
                }
                if (lane == c) xfin = s;
            }
        }
    }
    if (lane < W) {
        stw(a.xb + row, xfin);
        if (o < a.n) a.lx[o] = xfin;
        else if (o < a.n + a.p) a.ly[o - a.n] = xfin;
        else a.lz[o - a.n - a.p] = xfin;
    }
    return true;
}

template <int PFS>
__global__ __launch_bounds__(64) void k_ul_solve2(UlSolve2Args a)
{
    extern __shared__ double ul_sx[];  // the backward pass's x of the rows above the task (xa_cap doubles), then 64 products
    double* const s_xa = ul_sx;
    double* const s_pr = ul_sx + a.xa_cap;
    __shared__ int s_task;
    const int lane = threadIdx.x;
    if (lane == 0) s_task = atomicAdd(a.ticket, 1);
    __syncthreads();
    for (int guard = 0; guard <= 2 * a.ntask; ++guard) {
        const int tk = readfirst(s_task);
        __syncthreads();
        if (tk >= 2 * a.ntask) break;
        bool ok = true;
        if (a.trace && lane == 0) { a.trace[4 * (size_t)tk] = wall_clock64(); a.trace[4 * (size_t)tk + 3] = blockIdx.x; }
        if (tk < a.ntask) {
            const int t = a.tsort[tk];
            ok = ul_fwd_task<PFS>(a, t, lane, [&]() {  // (the task waits for the tasks of its columns as it reaches them: ul_fwd_task)
                if (a.trace && lane == 0) a.trace[4 * (size_t)tk + 1] = wall_clock64();
                return true;
            });
            if (ok) {
                drain_stores();
                stf(a.fdone + t, a.epoch);
            }
        } else {
            const int t = a.tsort[2 * a.ntask - 1 - tk];
            const int parent = a.taskrec[8 * t + 7];
            if (a.fwd_only) ok = __ballot(!spin_until(parent >= 0 ? a.bdone + parent : a.fdone + t, a.epoch)) == 0;  // (debugging aid: the forward pass alone, for timing)
            else ok = ul_bwd_task(a, t, lane, [&]() {
                const bool w = spin_until(parent >= 0 ? a.bdone + parent : a.fdone + t, a.epoch);
                if (a.trace && lane == 0) a.trace[4 * (size_t)tk + 1] = wall_clock64();
                return __ballot(!w) == 0;
            }, s_xa, s_pr);
            if (ok) {
                drain_stores();
                stf(a.bdone + t, a.epoch);
            }
        }
        if (!ok) { if (lane == 0) atomicMin(a.info, -2); break; }
        if (a.trace && lane == 0) a.trace[4 * (size_t)tk + 2] = wall_clock64();
        if (lane == 0) s_task = atomicAdd(a.ticket, 1);
        __syncthreads();
    }
}


// With the per-XCD queues a factorisation needs workgroups on every XCD: two persistent launches from different handles / streams that each hold a part of the chip could
// wait for one another's unscheduled workgroups (seen once in a while with two host threads before this ordering: a wait without end).  The engine's persistent launches of
// one device -- factorisations and substitutions -- are therefore ordered on the device: a launch waits (hipStreamWaitEvent, nothing on the host) for the previous one.
struct XqLaunchOrder {
    std::mutex mu;
    struct Last { hipStream_t st = nullptr; hipEvent_t ev = nullptr; };
    std::map<int, Last> last;
    // wait for the previous launch of this device IF it went to another stream, launch, record on the handle's own event -- under ONE lock: two threads must not
    // both find "nothing to wait for".  A single handle pays an event record per launch and no wait.
    template <class Launch>
    void run(int dev, hipStream_t st, hipEvent_t own, Launch&& launch)
    {
        static const bool off = debug_token("exact_no_order") != nullptr;  // (measurement aid)
        if (off) { launch(); return; }
        std::lock_guard<std::mutex> lk(mu);
        Last& l = last[dev];
        if (l.ev != nullptr && l.st != st) PQ_HIP(hipStreamWaitEvent(st, l.ev, 0));
        launch();
        PQ_HIP(hipEventRecord(own, st));
        l.st = st; l.ev = own;
    }
    void forget(int dev, hipEvent_t own)  // (a handle is going away)
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = last.find(dev);
        if (it != last.end() && it->second.ev == own) last.erase(it);
    }
};
inline XqLaunchOrder& xq_launch_order() { static XqLaunchOrder* o = new XqLaunchOrder; return *o; }  // (never destroyed: handles may outlive the statics at process exit)

class ExactSparseKKT final : public KKTSolverBase {
public:
    struct TooCostly {};  // thrown by the constructor before anything is allocated: the factorisation has more flops than the caller's limit
    ExactSparseKKT(const pq_sparse_data* d, int mode, int device, double max_flops) : dev_(device), mode_(mode)
    {
        if (d->mem != PQ_MEM_HOST) throw std::runtime_error("sparse data must be host-resident");
        sparse::Symbolic S;
        sparse::analyse_kkt_pattern(d, mode_, S);
        sparse::analyse_uplooking(S, d, U_);
        if (max_flops > 0.0 && U_.flops > max_flops) throw TooCostly{};
        PQ_HIP(hipSetDevice(dev_));
        PQ_HIP(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
        n_ = U_.n; p_ = U_.p; m_ = U_.m; N_ = U_.N;
        if (mode_ != 0) {
            // eliminated blocks: entry -> value index of P K P' and the product-term lists (kkt_all_eliminated.hpp:184-223)
            nzAA_ = (int)S.gramA.rowind.size(); nzGG_ = (int)S.gramG.rowind.size();
            std::vector<int> maa(nzAA_), mgg(nzGG_);
            for (int e = 0; e < nzAA_; ++e) maa[e] = U_.PKi[S.gramA_to_Ki[e]];
            for (int e = 0; e < nzGG_; ++e) mgg[e] = U_.PKi[S.gramG_to_Ki[e]];
            upload_vec(mapAA_, maa, st_); upload_vec(mapGG_, mgg, st_);
            upload_vec(aa_ptr_, S.gramA.ptr, st_); upload_vec(aa_q1_, S.gramA.q1, st_); upload_vec(aa_q2_, S.gramA.q2, st_); upload_vec(aa_k_, S.gramA.k, st_);
            upload_vec(gg_ptr_, S.gramG.ptr, st_); upload_vec(gg_q1_, S.gramG.q1, st_); upload_vec(gg_q2_, S.gramG.q2, st_); upload_vec(gg_k_, S.gramG.k, st_);
            ata_vals_.alloc(nzAA_ ? nzAA_ : 1);
        }
        zinv_.alloc(m_ ? m_ : 1); rhs_top_.alloc(n_ ? n_ : 1);
        nnzK_ = U_.Cp[N_];
        build_device();
        ops_.init(d, st_);
        ops_.set_reference_order(true);
        remap_values();
    }
    ~ExactSparseKKT() override
    {
        (void)hipSetDevice(dev_);
        if (st_) { (void)hipStreamSynchronize(st_); }
        if (xq_event_) { xq_launch_order().forget(dev_, xq_event_); (void)hipEventDestroy(xq_event_); }
        if (st_) (void)hipStreamDestroy(st_);
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
        delta_ = delta;
        if (mode_ == 0) {
            hipLaunchKernelGGL(k_ul_set_diag, g1(N_), dim3(256), 0, st_, n_, p_, m_, diag_pos_.p, ops_.P_diag(), x_reg, delta, z_reg, vals_.p);
        } else {
            // update_kkt_cost_scalings / _equality_scalings / _inequality_scaling of the mode, every entry's terms in the reference's order
            const bool eq = mode_ & 1, ineq = mode_ & 2;
            if (m_ > 0) hipLaunchKernelGGL(k_ul_reciprocal, g1(m_), dim3(256), 0, st_, m_, z_reg, zinv_.p);
            PQ_HIP(hipMemsetAsync(vals_.p, 0, sizeof(double) * (size_t)nnzK_, st_));
            launch_remap_values(ops_.nzP(), mapP_.p, ops_.P_x(), vals_.p, st_);
            hipLaunchKernelGGL(k_ul_cond_diag, g1(N_), dim3(256), 0, st_, n_, eq ? 0 : p_, ineq ? 0 : m_, diag_pos_.p, x_reg, delta, z_reg, vals_.p);
            if (eq) { if (nzAA_) hipLaunchKernelGGL(k_ul_axpy_mapped, g1(nzAA_), dim3(256), 0, st_, nzAA_, mapAA_.p, 1.0 / delta, ata_vals_.p, vals_.p); }
            else launch_remap_values(ops_.nzA(), mapA_.p, ops_.AT_x(), vals_.p, st_);
            if (ineq) {
                if (nzGG_) hipLaunchKernelGGL(k_ul_gram_values<true>, g1(nzGG_), dim3(256), 0, st_, nzGG_, gg_ptr_.p, gg_q1_.p, gg_q2_.p, gg_k_.p, ops_.GT_x(), z_reg, mapGG_.p, vals_.p);
            } else launch_remap_values(ops_.nzG(), mapG_.p, ops_.GT_x(), vals_.p, st_);
        }
        prof_.end(0, t0, st_);
        const int t1 = prof_.begin(1, st_);
        for (int attempt = 0;; ++attempt) {
        ++epoch_;
        UlFactorArgs a;
        a.N = N_; a.nticket = nticket_; a.epoch = epoch_;
        a.Cp = Cp_.p; a.Ci = Ci_.p; a.Cx = vals_.p;
        a.tk_kind = tk_kind_.p; a.tk_id = tk_id_.p; a.task_rows = task_rows_.p; a.rowrec = rowrec_.p; a.taskrec = taskrec_.p; a.dep = dep_.p;
        a.E4 = reinterpret_cast<const int4*>(E4_.p); a.Etab = Etab_.p; a.Li = Li_.p; a.Emask = Emask_.p;
        a.Lx = Lx_.p; a.D = D_.p; a.Dinv = Dinv_.p; a.Ystash = Ystash_.p; a.Pstash = Pstash_.p; a.Dinit = Dinit_.p; a.Lblock = Lblock_.p;
        a.done = done_.p; a.p1done = p1done_.p; a.ticket = xq_on_ ? xtick_.p : ctl_.p; a.info = ctl_.p + 1;
        a.xq_ptr = xq_on_ ? xq_ptr_.p : nullptr; a.xq_rows = xq_rows_.p; a.Dloc = Dloc_.p;
        a.rowpar = serial_path_ ? 0 : 1;
        a.yglob = yglob_.p;
        a.trace = trace_.n > 1 ? trace_.p : nullptr;
        if (N_ > 0) {
            {
                const size_t most = std::max(std::max(D_.n, Lblock_.n / 8 + 1), (size_t)128);  // (the table: up to eight words per thread)
                hipLaunchKernelGGL(k_ul_prepare, dim3((unsigned)((most + 255) / 256)), dim3(256), 0, st_, D_.n, D_.p, xq_on_ ? Dloc_.n : (size_t)0, Dloc_.p, Lblock_.n, Lblock_.p, ctl_.p,
                                   xq_on_ ? xtick_.p : (int*)nullptr);
            }
            auto launch = [&] {
                if (lds_y_) hipLaunchKernelGGL(k_ul_factor<true>, dim3(grid_), dim3(64), (size_t)N_ * sizeof(double), st_, a);
                else hipLaunchKernelGGL(k_ul_factor<false>, dim3(grid_), dim3(64), 0, st_, a);
            };
            if (xq_on_) xq_launch_order().run(dev_, st_, xq_event_, launch);
            else launch();
        }
        PQ_HIP(hipGetLastError());
        if (N_ == 0) { prof_.end(1, t1, st_); stream_wait(st_); return true; }  // (nothing to factor)
        PQ_HIP(hipMemcpyAsync(ctl_h_.p + 2, ctl_.p + 1, sizeof(int), hipMemcpyDeviceToHost, st_));
        stream_wait(st_);
        if (ctl_h_.p[2] <= -2) {
            // A wait ran into its bound.  With the per-XCD queues that is what happens when an XCD has no resident workgroup of this launch -- the residency was
            // probed once, at build time, and another process on the device (test workers, ranks sharing one GPU) or another long persistent kernel can take
            // it away: the queue of that XCD then never moves.  The single ticket queue has no such requirement (tickets are drawn by whoever runs, waits only
            // target earlier tickets), so the factorisation is run again on it, once, and this handle keeps it; only a second failure is an error.
            if (xq_on_ && attempt == 0) { xq_on_ = false; ++xq_fallbacks_; continue; }
            prof_.end(1, t1, st_);
            throw std::runtime_error("reference-order factorisation: a task waited for its children without end (scheduling error)");
        }
        prof_.end(1, t1, st_);
        return ctl_h_.p[2] == INT_MAX;  // n == cols (sparse/kkt.hpp:104)
        }
    }
    // sparse/kkt.hpp:107-145, KKT_FULL
    void solve(const double* rhs_x, const double* rhs_y, const double* rhs_z, double* lhs_x, double* lhs_y, double* lhs_z) override
    {
        PQ_ZONE("piqp_amd::ExactSparseKKT::solve");
        PQ_HIP(hipSetDevice(dev_));
        const int tk = prof_.begin(2, st_);
        // sparse/kkt.hpp:113-136: the eliminated blocks are folded into the x part of the right-hand side; the KKT vector of the mode is [x; kept y; kept z]
        const bool eq = mode_ & 1, ineq = mode_ & 2;
        const double delta_inv = 1.0 / delta_;
        const double *in_x = rhs_x, *in_y = rhs_y, *in_z = rhs_z;
        double *out_y = lhs_y, *out_z = lhs_z;
        int kp = p_, km = m_;
        if (mode_ != 0) {
            ops_.fold_rhs(rhs_x, rhs_y, rhs_z, zinv_.p, delta_inv, rhs_top_.p, st_, eq, ineq);
            in_x = rhs_top_.p;
            if (eq) { kp = 0; in_y = nullptr; out_y = nullptr; }
            if (ineq) { km = 0; in_z = nullptr; out_z = nullptr; }
        }
        if (N_ > 0 && !one_wave_solve_) {
            ++sepoch_;
            PQ_HIP(hipMemsetAsync(ctl_.p + 2, 0, 2 * sizeof(int), st_));  // the substitution's ticket and its result word
            UlSolve2Args b;
            b.N = N_; b.n = n_; b.p = kp; b.m = km; b.ntask = ntask_; b.epoch = sepoch_;
            b.perm = perm_.p; b.taskrec = taskrec_.p; b.task_rows = task_rows_.p; b.tsort = tsort_.p; b.tdep = tdep_.p; b.fs_u = fs_u_.p; b.fs_col = fs_col_.p; b.fs4 = reinterpret_cast<const int4*>(fs4_.p); b.fs_task = fs_task_.p;
            b.Lp = Lp_.p; b.Li = Li_.p; b.Lsrc = Lsrc_.p; b.Tmask = Tmask_.p; b.mask_ptr = mask_ptr_.p; b.Lblock = Lblock_.p; b.Lx = Lx_.p; b.Dinv = Dinv_.p;
            b.rx = in_x; b.ry = in_y; b.rz = in_z; b.lx = lhs_x; b.ly = out_y; b.lz = out_z;
            b.xf = xf_.p; b.xz = xz_.p; b.xb = xb_.p; b.fdone = fdone_.p; b.bdone = bdone_.p; b.ticket = ctl_.p + 2; b.info = ctl_.p + 3;
            b.fwd_only = fwd_only_ ? 1 : 0;
            b.ta_ptr = ta_ptr_.p; b.ta_rows = ta_rows_.p; b.Lsrc2 = Lsrc2_.p; b.xa_cap = xa_cap_;
            b.trace = strace_.n > 1 ? strace_.p : nullptr;
            // (with the per-XCD queues the engine's persistent launches of one device run one after the other, the substitutions included: a factorisation then only
            // ever shares the chip with short launches that leave on their own -- see XqLaunchOrder)
            auto launch = [&] { hipLaunchKernelGGL(k_ul_solve2<32>, dim3(sgrid_), dim3(64), (size_t)(xa_cap_ + 64) * sizeof(double), st_, b); };
            if (xq_on_) xq_launch_order().run(dev_, st_, xq_event_, launch);
            else launch();
            // a wait of the substitution that ran into its bound leaves tasks undone and stale finite values in the solution: poison it (the dense sweeps do the same),
            // so that KKTSystem's refinement / the solver's finiteness checks fail the step instead of taking it
            hipLaunchKernelGGL(k_ul_poison_if_failed, dim3(1), dim3(64), 0, st_, ctl_.p + 3, lhs_x, n_);
        } else if (N_ > 0) {
        UlSolveArgs a;
        a.N = N_; a.n = n_; a.p = kp; a.m = km;
        a.perm = perm_.p; a.Lp = Lp_.p; a.Li = Li_.p; a.Lcol = Lcol_.p; a.Lx = Lx_.p; a.Dinv = Dinv_.p;
        a.bgroup = reinterpret_cast<const int4*>(bgroup_.p); a.nbgroup = nbgroup_;
        a.rx = in_x; a.ry = in_y; a.rz = in_z; a.lx = lhs_x; a.ly = out_y; a.lz = out_z;
        a.xglob = xglob_.p; a.err = nullptr; a.epoch = 0;
        {
            if (lds_x_) hipLaunchKernelGGL(k_ul_solve<true>, dim3(1), dim3(64), (size_t)N_ * sizeof(double), st_, a);
            else hipLaunchKernelGGL(k_ul_solve<false>, dim3(1), dim3(64), 0, st_, a);
        }
        }
        if (mode_ != 0) ops_.recover_duals(lhs_x, rhs_y, rhs_z, zinv_.p, delta_inv, lhs_y, lhs_z, st_, eq, ineq);  // sparse/kkt.hpp:147-175
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
        case 8: if (out_host && trace_.n > 1) PQ_HIP(hipMemcpy(out_host, trace_.p, sizeof(long long) * trace_.n, hipMemcpyDeviceToHost)); return trace_.n > 1 ? (long long)trace_.n : 0;  // PIQP_AMD_DEBUG=exact_trace
        case 9: if (out_host) { if (xq_on_) std::fill((int*)out_host, (int*)out_host + N_, 0); else std::copy(U_.tk_kind.begin(), U_.tk_kind.end(), (int*)out_host); } return xq_on_ ? N_ : nticket_;  // (per-XCD queues: trace slot = position in xq_rows)
        case 10: if (out_host) { if (xq_on_) std::copy(xq_rows_h_.begin(), xq_rows_h_.begin() + N_, (int*)out_host); else std::copy(U_.tk_id.begin(), U_.tk_id.end(), (int*)out_host); } return xq_on_ ? N_ : nticket_;
        case 11: if (out_host) { for (int k = 0; k < N_; ++k) ((int*)out_host)[k] = U_.Rp[k + 1] - U_.Rp[k]; } return N_;
        case 12: if (out_host) std::copy(U_.task_ptr.begin(), U_.task_ptr.end(), (int*)out_host); return ntask_ + 1;
        case 13: if (out_host) std::copy(U_.task_rows.begin(), U_.task_rows.end(), (int*)out_host); return N_;
        case 14: if (out_host && strace_.n > 1) PQ_HIP(hipMemcpy(out_host, strace_.p, sizeof(long long) * strace_.n, hipMemcpyDeviceToHost)); return strace_.n > 1 ? (long long)strace_.n : 0;  // the last solve's timeline
        case 15: if (out_host) std::copy(U_.tsort.begin(), U_.tsort.end(), (int*)out_host); return ntask_;
        case 16: if (out_host) std::copy(U_.task_nU.begin(), U_.task_nU.end(), (int*)out_host); return ntask_;
        case 17: if (out_host) std::copy(U_.tparent.begin(), U_.tparent.end(), (int*)out_host); return ntask_;
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
    ExactSparseKKT(const ExactSparseKKT& o, int) : dev_(o.dev_), mode_(o.mode_), nzAA_(o.nzAA_), nzGG_(o.nzGG_), n_(o.n_), p_(o.p_), m_(o.m_), N_(o.N_), nnzK_(o.nnzK_), delta_(o.delta_), U_(o.U_)
    {
        PQ_HIP(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
        build_device();
        auto cpi = [&](DBuf<int>& d, const DBuf<int>& sc) { d.alloc(sc.n ? sc.n : 1); if (sc.n) PQ_HIP(hipMemcpyAsync(d.p, sc.p, sc.bytes(), hipMemcpyDeviceToDevice, st_)); };
        auto cpd = [&](DBuf<double>& d, const DBuf<double>& sc) { d.alloc(sc.n ? sc.n : 1); if (sc.n) PQ_HIP(hipMemcpyAsync(d.p, sc.p, sc.bytes(), hipMemcpyDeviceToDevice, st_)); };
        cpi(mapAA_, o.mapAA_); cpi(mapGG_, o.mapGG_); cpi(aa_ptr_, o.aa_ptr_); cpi(aa_q1_, o.aa_q1_); cpi(aa_q2_, o.aa_q2_); cpi(aa_k_, o.aa_k_);
        cpi(gg_ptr_, o.gg_ptr_); cpi(gg_q1_, o.gg_q1_); cpi(gg_q2_, o.gg_q2_); cpi(gg_k_, o.gg_k_);
        cpd(ata_vals_, o.ata_vals_); cpd(zinv_, o.zinv_); rhs_top_.alloc(n_ ? n_ : 1);
        ops_.clone_from(o.ops_, st_);
        if (nnzK_) PQ_HIP(hipMemcpyAsync(vals_.p, o.vals_.p, sizeof(double) * (size_t)nnzK_, hipMemcpyDeviceToDevice, st_));
        if (U_.nnzL) PQ_HIP(hipMemcpyAsync(Lx_.p, o.Lx_.p, sizeof(double) * (size_t)U_.nnzL, hipMemcpyDeviceToDevice, st_));
        if (o.Lblock_.n) PQ_HIP(hipMemcpyAsync(Lblock_.p, o.Lblock_.p, o.Lblock_.bytes(), hipMemcpyDeviceToDevice, st_));  // (the substitution reads L from the tasks' tables)
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
        upload_vec(tk_kind_, U_.tk_kind, st_); upload_vec(tk_id_, U_.tk_id, st_); upload_vec(task_rows_, U_.task_rows, st_); upload_vec(dep_, U_.dep, st_);
        ntask_ = (int)U_.task_ptr.size() - 1; nticket_ = (int)U_.tk_kind.size();
        upload_vec(tsort_, U_.tsort, st_); upload_vec(tdep_, U_.tdep, st_); upload_vec(fs_u_, U_.fs_u, st_); upload_vec(fs_col_, U_.fs_col, st_); upload_vec(Lsrc_, U_.Lsrc, st_);
        {   // the forward pass's records in step order: no second, dependent load for the mask
            std::vector<int> f4((size_t)std::max<size_t>(U_.fs_u.size(), 1) * 4, 0);
            for (int t = 0; t < ntask_; ++t)
                for (int q = U_.fs_ptr[t]; q < U_.fs_ptr[t + 1]; ++q) {
                    const unsigned long long mk = U_.Tmask[(size_t)U_.mask_ptr[t] + U_.fs_u[q]];
                    f4[4 * (size_t)q] = U_.fs_u[q]; f4[4 * (size_t)q + 1] = U_.fs_col[q]; f4[4 * (size_t)q + 2] = (int)(unsigned)(mk & 0xffffffffull); f4[4 * (size_t)q + 3] = (int)(unsigned)(mk >> 32);
                }
            upload_vec(fs4_, f4, st_);
            std::vector<int> ft(std::max<size_t>(U_.fs_col.size(), 1), 0);
            for (size_t q = 0; q < U_.fs_col.size(); ++q) ft[q] = U_.row_task[U_.fs_col[q]];
            upload_vec(fs_task_, ft, st_);
        }
        {   // the backward pass's lists: per task the rows above it that its columns touch (first seen first), per entry of L where its operand comes from
            std::vector<int> tap((size_t)ntask_ + 1, 0), tar, src2(std::max<size_t>(U_.Lsrc.size(), 1), 0), seen((size_t)std::max(N_, 1), -1), slot((size_t)std::max(N_, 1), 0);
            int longest = 0;
            for (int t = 0; t < ntask_; ++t) {
                const int base = (int)tar.size();
                for (int g = U_.task_ptr[t]; g < U_.task_ptr[t + 1]; ++g) {
                    const int j = U_.task_rows[g];
                    for (int q = U_.Lp[j]; q < U_.Lp[j + 1]; ++q) {
                        if (U_.Lsrc[q] >= 0) { src2[q] = U_.Lsrc[q]; continue; }
                        const int r = U_.Li[q];
                        if (seen[r] != t) { seen[r] = t; slot[r] = (int)tar.size() - base; tar.push_back(r); }
                        src2[q] = 64 + slot[r];
                    }
                }
                tap[t + 1] = (int)tar.size();
                longest = std::max(longest, (int)tar.size() - base);
            }
            tar.push_back(0);  // (padding: a task without such rows still forms the address of its first one)
            xa_cap_ = std::max(64, (longest + 63) / 64 * 64);
            upload_vec(ta_ptr_, tap, st_); upload_vec(ta_rows_, tar, st_); upload_vec(Lsrc2_, src2, st_);
        }
        upload_vec(Tmask_, U_.Tmask, st_); upload_vec(mask_ptr_, U_.mask_ptr, st_);
        xf_.alloc(N_ ? N_ : 1); xz_.alloc(N_ ? N_ : 1); xb_.alloc(N_ ? N_ : 1); xb_.zero(st_); xf_.zero(st_);
        fdone_.alloc(ntask_ ? ntask_ : 1); fdone_.zero(st_); bdone_.alloc(ntask_ ? ntask_ : 1); bdone_.zero(st_);
        {   // the packed records and the task-ordered entry space (see UlFactorArgs)
            const long long nnzL = U_.nnzL;
            std::vector<int> rowrec((size_t)std::max(N_, 1) * 16, 0), taskrec((size_t)std::max(ntask_, 1) * 8, 0), e4((size_t)std::max<long long>(nnzL, 1) * 4, 0), etab(std::max<long long>(nnzL, 1), 0);
            std::vector<unsigned long long> emask(std::max<long long>(nnzL, 1), 0ull);
            int w = 0;
            for (int t = 0; t < ntask_; ++t) {
                const int rb = U_.task_ptr[t], W = U_.task_ptr[t + 1] - rb;
                // rows side by side (ul_path_row) cost about one memory round trip per path column of a row, ~3 us each, whatever the rows hold; one wave walking the
                // task's rows one after the other (ul_path) costs ~0.18 us per entry: short rows (chains of a few entries, STADAT-like) go to the latter
                long long task_entries = 0;
                for (int q = 0; q < W; ++q) { const int kk = U_.task_rows[rb + q]; task_entries += U_.Rp[kk + 1] - U_.Rp[kk]; }
                const bool serial_task = serial_path_ || task_entries < (long long)serial_below_ * W;
                taskrec[8 * t] = rb; taskrec[8 * t + 1] = W; taskrec[8 * t + 2] = U_.task_nU[t]; taskrec[8 * t + 3] = U_.tab_ptr[t];
                taskrec[8 * t + 4] = U_.fs_ptr[t]; taskrec[8 * t + 5] = U_.tdep_ptr[t]; taskrec[8 * t + 6] = U_.tdep_ptr[t + 1] - U_.tdep_ptr[t]; taskrec[8 * t + 7] = U_.tparent[t];
                for (int q = 0; q < W; ++q) {
                    const int k = U_.task_rows[rb + q];
                    int* r = rowrec.data() + 16 * (size_t)k;
                    r[0] = w; r[1] = U_.Rp[k + 1] - U_.Rp[k]; r[2] = U_.Cp[k]; r[3] = U_.Cp[k + 1] - U_.Cp[k]; r[4] = t; r[5] = W; r[6] = q; r[7] = U_.tab_ptr[t];
                    r[8] = U_.row_prev[k]; r[9] = U_.dep_ptr[k]; r[10] = U_.dep_ptr[k + 1] - U_.dep_ptr[k]; r[11] = U_.task_nU[t]; r[12] = serial_task ? 1 : 0;
                    for (int e = U_.Rp[k]; e < U_.Rp[k + 1]; ++e, ++w) {
                        const int i = U_.Rcol[e];
                        e4[4 * (size_t)w] = i; e4[4 * (size_t)w + 1] = U_.Lp[i]; e4[4 * (size_t)w + 2] = U_.Rcnt[e]; e4[4 * (size_t)w + 3] = U_.Rpos[e];
                        etab[w] = U_.Rtab[e];
                        emask[w] = U_.Tmask[U_.mask_ptr[t] + U_.Rtab[e]];
                    }
                }
            }
            if (w != nnzL) throw std::runtime_error("reference-order engine: entry space");
            upload_vec(rowrec_, rowrec, st_); upload_vec(taskrec_, taskrec, st_); upload_vec(E4_, e4, st_); upload_vec(Etab_, etab, st_); upload_vec(Emask_, emask, st_);
        }
        Ystash_.alloc(U_.nnzL ? (size_t)U_.nnzL : 1); Pstash_.alloc(U_.nnzL ? (size_t)U_.nnzL : 1); Dinit_.alloc(N_ ? N_ : 1);
        Lblock_.alloc(U_.tab_ptr.back() ? (size_t)U_.tab_ptr.back() : 1); Lblock_.zero(st_);
        p1done_.alloc(N_ ? N_ : 1); p1done_.zero(st_);
        vals_.alloc(nnzK_ ? nnzK_ : 1); vals_.zero(st_);
        Lx_.alloc(U_.nnzL ? (size_t)U_.nnzL : 1); Lx_.zero(st_);
        D_.alloc(N_ ? N_ : 1); Dinv_.alloc(N_ ? N_ : 1); D_.zero(st_); Dinv_.zero(st_);
        done_.alloc(N_ ? N_ : 1); done_.zero(st_);
        ctl_.alloc(4); ctl_h_.alloc(4);
        {   // per-XCD row queues: tasks in the order of their last rows, each to the queue with the least work so far (entries of its rows); rows ascending per queue
            xq_on_ = !serial_path_ && debug_token("exact_one_queue") == nullptr;
            if (!xq_event_) PQ_HIP(hipEventCreateWithFlags(&xq_event_, hipEventDisableTiming));
            std::vector<int> order((size_t)ntask_), qof((size_t)std::max(ntask_, 1), 0);
            for (int t = 0; t < ntask_; ++t) order[t] = t;
            std::sort(order.begin(), order.end(), [&](int x, int y) { return U_.task_rows[U_.task_ptr[x + 1] - 1] < U_.task_rows[U_.task_ptr[y + 1] - 1]; });
            long long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int t : order) {
                long long w = 0;
                for (int g = U_.task_ptr[t]; g < U_.task_ptr[t + 1]; ++g) { const int k = U_.task_rows[g]; w += 4 + (U_.Rp[k + 1] - U_.Rp[k]); }
                int best = 0;
                for (int q = 1; q < 8; ++q) if (load[q] < load[best]) best = q;
                qof[t] = best; load[best] += w;
            }
            std::vector<int> qp(9, 0), qr((size_t)std::max(N_, 1), 0);
            for (int k = 0; k < N_; ++k) qp[qof[U_.row_task[k]] + 1]++;
            for (int q = 0; q < 8; ++q) qp[q + 1] += qp[q];
            std::vector<int> fill(qp.begin(), qp.end() - 1);
            for (int k = 0; k < N_; ++k) qr[fill[qof[U_.row_task[k]]]++] = k;  // (k ascending: every queue ascending)
            upload_vec(xq_ptr_, qp, st_); upload_vec(xq_rows_, qr, st_); xq_rows_h_ = qr;
            xtick_.alloc(8 * 16 + 16); xtick_.zero(st_);
            Dloc_.alloc(N_ ? N_ : 1); Dloc_.zero(st_);
        }
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
        if (xq_on_) {
            // every XCD's queue needs workgroups of its own: at least 64 workgroups, and a probe launch of the same shape must find all eight XCDs served
            grid_ = std::max(grid_, std::min(64, per_cu * ncu));
            DBuf<int> cnt; cnt.alloc(8); cnt.zero(st_);
            hipLaunchKernelGGL(k_ul_xcd_probe, dim3(grid_), dim3(64), 0, st_, cnt.p);
            int h[8];
            PQ_HIP(hipMemcpyAsync(h, cnt.p, sizeof(h), hipMemcpyDeviceToHost, st_));
            stream_wait(st_);
            for (int q = 0; q < 8; ++q) if (h[q] < 1) xq_on_ = false;  // (another partition mode, fewer XCDs: one queue of tickets as before)
        }
        {
            int per_cu_s = 1;
            const size_t sneed = (size_t)(xa_cap_ + 64) * sizeof(double);
            if (sneed > (size_t)dev_lds) one_wave_solve_ = true;  // (a task whose columns reach more rows above it than LDS holds: the single-wave substitution)
            else if (sneed > 48 * 1024) {
                static PerDeviceOnce once_s;
                once_s([&] { PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ul_solve2<32>), hipFuncAttributeMaxDynamicSharedMemorySize, dev_lds)); });
            }
            PQ_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_s, k_ul_solve2<32>, 64, one_wave_solve_ ? 0 : sneed));
            sgrid_ = std::max(1, std::min(2 * ntask_, std::max(1, per_cu_s) * ncu));
            if (const char* tg = debug_token("exact_grid")) sgrid_ = std::max(1, std::min(sgrid_, std::atoi(tg)));
        }
        trace_.alloc(debug_token("exact_trace") ? (size_t)4 * std::max(nticket_, 1) : 1);
        strace_.alloc(debug_token("exact_trace") ? (size_t)8 * std::max(ntask_, 1) : 1);
        yglob_.alloc(lds_y_ ? 1 : (size_t)grid_ * (size_t)N_);
        xglob_.alloc(lds_x_ ? 1 : (size_t)std::max(N_, 1));
        stream_wait(st_);
    }
    void remap_values()
    {
        if (mode_ == 0) {
            launch_remap_values(ops_.nzP(), mapP_.p, ops_.P_x(), vals_.p, st_);
            launch_remap_values(ops_.nzA(), mapA_.p, ops_.AT_x(), vals_.p, st_);
            launch_remap_values(ops_.nzG(), mapG_.p, ops_.GT_x(), vals_.p, st_);
        } else if ((mode_ & 1) && nzAA_) {
            // update_AT_A (kkt_all_eliminated.hpp:184-202): the values change only with the data; every factorisation rebuilds P K P'
            hipLaunchKernelGGL(k_ul_gram_values<false>, g1(nzAA_), dim3(256), 0, st_, nzAA_, aa_ptr_.p, aa_q1_.p, aa_q2_.p, aa_k_.p, ops_.AT_x(), (const double*)nullptr,
                               (const int*)nullptr, ata_vals_.p);
        }
        PQ_HIP(hipGetLastError());
        stream_wait(st_);
    }

    int dev_, mode_ = 0, nzAA_ = 0, nzGG_ = 0, n_ = 0, p_ = 0, m_ = 0, N_ = 0, nnzK_ = 0, ntask_ = 0, nticket_ = 0, grid_ = 1, nbgroup_ = 0, epoch_ = 0;
    double delta_ = 1.0;
    DBuf<int> mapAA_, mapGG_, aa_ptr_, aa_q1_, aa_q2_, aa_k_, gg_ptr_, gg_q1_, gg_q2_, gg_k_;
    DBuf<double> ata_vals_, zinv_, rhs_top_;
    bool lds_y_ = true, lds_x_ = true;
    hipStream_t st_ = nullptr;
    sparse::UpLooking U_;
    CscOperators ops_;
    DBuf<int> perm_, Cp_, Ci_, diag_pos_, mapP_, mapA_, mapG_, Lp_, Li_, Lcol_, Rp_, Rcol_, Rpos_, tk_kind_, tk_id_, task_rows_, dep_, rowrec_, taskrec_, E4_, Etab_, done_, p1done_, ctl_, bgroup_;
    DBuf<int> tsort_, tdep_, fs_u_, fs_col_, fs4_, fs_task_, Lsrc_, mask_ptr_, fdone_, bdone_, ta_ptr_, ta_rows_, Lsrc2_;
    int xa_cap_ = 64;
    DBuf<unsigned long long> Emask_, Tmask_;
    DBuf<double> xf_, xz_, xb_;
    DBuf<long long> trace_, strace_;
    DBuf<int> xq_ptr_, xq_rows_, xtick_;
    DBuf<double> Dloc_;
    bool xq_on_ = false;
    int xq_fallbacks_ = 0;  // factorisations that were run again on the single ticket queue
    hipEvent_t xq_event_ = nullptr;
    std::vector<int> xq_rows_h_;
    // tasks with fewer entries per row than this run their path pass on ONE wave (ul_path).  0 since the rows of a task hand over by values (one round trip per
    // row): side by side wins at every row length measured (QAFIRO 0.09 -> 0.05 ms, finnis 0.39 -> 0.28, STADAT1 7.9 -> 6.7; 17 before, when a hand-over cost five)
    int serial_below_ = debug_token("exact_serial_below") ? std::atoi(debug_token("exact_serial_below")) : 0;
    bool serial_path_ = debug_token("exact_serial_path") != nullptr;  // debugging aid: one path pass per task (the first form) instead of one per row
    int sepoch_ = 0, sgrid_ = 1;
    bool fwd_only_ = debug_token("exact_fwd_only") != nullptr;  // debugging aid (timing): the backward pass of the substitution skipped
    bool one_wave_solve_ = debug_token("exact_solve1") != nullptr;  // debugging aid: the single-wave substitution of the first version
    DBuf<double> vals_, Lx_, D_, Dinv_, Ystash_, Pstash_, Dinit_, Lblock_, yglob_, xglob_;
    HBuf<int> ctl_h_;
    StageProfiler prof_;
};

}  // namespace

// max_flops > 0: nullptr when the factorisation would take more flops than that (the caller then builds the multifrontal engine)
KKTSolverBase* make_exact_sparse_kkt(const pq_sparse_data* data, int mode, int device, double max_flops)
{
    try {
        return new ExactSparseKKT(data, mode, device, max_flops);
    } catch (const ExactSparseKKT::TooCostly&) {
        return nullptr;
    }
}

}  // namespace pq
