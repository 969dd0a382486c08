// piqp_amd/csrc/dense_kernels.hip -- hand-written gfx950 kernels of the dense KKT path.
//
// Reference operations replaced (PIQP v0.6.2 include/piqp/):
//   dense/kkt.hpp:140-160  update_kkt       -> k_syrk_lower<EPI_ASSEMBLE>  (fp64 MFMA 16x16x4, LDS double-buffered)
//   dense/kkt.hpp:53,68    AT_A = AT*AT^T   -> k_syrk_lower<EPI_STORE>
//   Eigen::LLT::compute (dense/kkt.hpp:82) / LDLTNoPivot (dense/ldlt_no_pivot.hpp:313-354)
//                                           -> ONE launch per panel, k_syrk_lower<EPI_SUBTRACT_POTRF>: trailing update + potrf_block of the next
//                                              diagonal block (fused_next_diag) + panel_follow of the next panel; k_potrf_diag + k_trsm_panel
//                                              only for the first panel
//   llt.solveInPlace (dense/kkt.hpp:170) / ldlt_no_pivot.hpp:432-450 -> k_trsv_persistent (k_trsv_fwd_step / k_trsv_bwd_step: fallback)
//   dense/kkt.hpp:94-104,112-131 GEMVs      -> k_gemv_n_partial + k_reduce_partials, k_gemv_t
//
// Layout: everything column-major fp64.  The product matrices GT (n x m) / AT (n x p) are stored as in
// dense::Data (dense/data.hpp:29-31): column k of GT is row k of G, so both SYRK operands are read
// along contiguous columns (1 KiB per wave-instruction).
#include "dense_kernels.hpp"

#include <algorithm>
#include <map>
#include <mutex>
#include <stdexcept>
#include <utility>
#include <vector>
#include <cstdlib>
#include <cstdio>
#include <cstring>

namespace pq {
namespace dense {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double readlane_d(double v, int srclane)
{
    long long bits = __double_as_longlong(v);
    int lo = (int)(bits & 0xffffffffll), hi = (int)(bits >> 32);
    lo = __builtin_amdgcn_readlane(lo, srclane);
    hi = __builtin_amdgcn_readlane(hi, srclane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// ------------------------------------------------------------------------------------------------
// P_full = symmetric completion of P_utri (used by the assembly epilogue, coalesced along rows, and
// by eval_P_x as a plain column-dot GEMV).  dense/kkt.hpp:112-113 reads both triangles of P_utri.
// FROM_LOWER: the source holds the LOWER triangle (LDLTNoPivot<.., Eigen::Lower>::compute of a caller's matrix, ldlt_no_pivot.hpp:408-411): the element (r, c) of the
// upper triangle this kernel works from is then the source's (c, r) -- read along the source's columns all the same.
template <bool FROM_LOWER>
__global__ __launch_bounds__(256) void k_symmetrize_upper(const double* __restrict__ Pu, int ldu, int n, double* __restrict__ Pf, double* __restrict__ pdiag)
{
    __shared__ double tile[32][33];
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const bool lower_tile = r0 > c0;
    const int sr0 = lower_tile ? c0 : r0, sc0 = lower_tile ? r0 : c0;  // source tile in the upper triangle
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;            // 32 x 8
    for (int cc = ty; cc < 32; cc += 8) {
        if constexpr (FROM_LOWER) {
            int c = sc0 + tx, r = sr0 + cc;
            tile[tx][cc] = (r < n && c < n) ? Pu[c + (size_t)r * ldu] : 0.0;
        } else {
            int r = sr0 + tx, c = sc0 + cc;
            tile[cc][tx] = (r < n && c < n) ? Pu[r + (size_t)c * ldu] : 0.0;  // tile[col][row] of the source
        }
    }
    __syncthreads();
    for (int cc = ty; cc < 32; cc += 8) {
        int i = r0 + tx, j = c0 + cc;  // output element (i, j)
        if (i < n && j < n) {
            double v;
            if (lower_tile) v = tile[tx][cc];                       // Pu[j, i]: source row = j - sr0 = cc, col = i - sc0 = tx
            else if (r0 == c0) v = (tx <= cc) ? tile[cc][tx] : tile[tx][cc];
            else v = tile[cc][tx];
            Pf[i + (size_t)j * n] = v;
            if (i == j && pdiag) pdiag[i] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Lower-triangular rank-k product on the fp64 matrix cores.
//   acc(i,j) = sum_k A[i,k] * w[k] * B[j,k]      (A, B: n x kdim column-major; w optional)
//   EPI_ASSEMBLE: C(i,j) = Pfull(i,j) + [i==j] x_reg(i) + dinv * ATA(i,j) + acc     (dense/kkt.hpp:144-158)
//   EPI_SUBTRACT: C(i,j) -= acc                                                     (LLT rankUpdate / ldlt_no_pivot.hpp:350)
//   EPI_STORE   : C(i,j) = acc                                                      (dense/kkt.hpp:53)
// One 128x128 output tile per 256-thread workgroup (2x2 waves, each 64x64 = 4x4 MFMA 16x16 tiles,
// 64 fp64 accumulators per lane).  K is consumed in steps of 16 through two LDS stages; global loads
// of stage t+1 are issued before the MFMAs of stage t.  LDS rows are padded to 144 doubles so the
// 16-lane groups of a ds_read_b64 fall on disjoint banks.  The MFMA is issued with the COLUMN
// operand as A and the ROW operand as B so that each lane's four results are consecutive ROWS of C:
// stores (and the Pfull / ATA epilogue loads) are 128 B contiguous per 16 lanes.
#ifndef PQ_FUSE_TS
#define PQ_FUSE_TS 0
#endif
// Operand panels of a persistent launch.  Reading them IN PLACE with plain loads is not safe: the same addresses held a trailing tile earlier in the launch,
// and although that tile was only ever touched with agent-scope loads and write-through stores, stale copies do survive somewhere (measured: the factor of
// n = 4096 differed from the launch-per-panel path in one run of three).  Reading them in place with 8-byte agent-scope loads is safe and slow (every tile
// fetches its 256 KB of operands from memory: 30 us per tile against 23).  So a solved panel is written twice: in place (the factor), and into a SIDE
// buffer whose addresses nobody has read before in this launch -- no stale copy can exist anywhere -- from which the trailing updates read it with plain
// 16-byte loads, L1 / L2 cached like in the launch-per-panel kernels.  (Panel 0 is solved before the launch and read in place.)
constexpr bool AGENT_OPERANDS = false;
constexpr bool FUSE_TS_ON = PQ_FUSE_TS != 0;  // in-kernel clock stamps of the fused launch (PIQP_AMD_DEBUG=fused_ts=<panel>): only in builds with -DPQ_FUSE_TS=1
constexpr int TS = 128;
constexpr int BK = 16;
constexpr int LDS_LD = TS + 16;
constexpr int SYRK_LDS_BYTES = 2 * 2 * BK * LDS_LD * (int)sizeof(double);

__device__ __forceinline__ double ld_agent(const double* p);    // agent-scope relaxed load / store (sc1: L1-bypassing, written through), defined with the sweeps
__device__ __forceinline__ void st_agent(double* p, double v);
typedef __attribute__((address_space(1))) int gint;
__device__ __forceinline__ int ldi_agent(const int* p) { return __hip_atomic_load((const gint*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sti_agent(int* p, int v) { __hip_atomic_store((gint*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int addi_agent(int* p, int v) { return __hip_atomic_fetch_add((gint*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// NT threads cooperate on a 128 x 16 operand stage (1024 double2): ITERS = 1024 / NT loads per thread
template <bool CHECK, int NT>
__device__ __forceinline__ void load_tile(const double* __restrict__ M, int ld, int r0, int k0, int nrows, int kdim, int tid, d2 (&v)[1024 / NT])
{
    const int r = r0 + 2 * (tid & 63);
#pragma unroll
    for (int it = 0; it < 1024 / NT; ++it) {
        const int k = k0 + it * (NT / 64) + (tid >> 6);
        // (explicit global address space: inside the out-of-line roles of k_chol_persistent the operand pointer comes out of a struct in memory and is a
        // GENERIC pointer to the compiler -- FLAT loads, which also count on the LDS counter, so the first LDS reads of the K loop waited for every operand
        // stage still in flight)
        const __attribute__((address_space(1))) double* Mg = (const __attribute__((address_space(1))) double*)M;
        if (!CHECK) {
            v[it] = *reinterpret_cast<const __attribute__((address_space(1))) d2*>(Mg + r + (size_t)k * ld);
        } else {
            d2 t = {0.0, 0.0};
            if (k < kdim) {
                if (r < nrows) t.x = Mg[r + (size_t)k * ld];
                if (r + 1 < nrows) t.y = Mg[r + 1 + (size_t)k * ld];
            }
            v[it] = t;
        }
    }
}

// the same 128 x 16 stage read with agent-scope (sc1, L1-bypassing) loads: data another workgroup of the SAME launch wrote through (no edge handling)
template <int NT>
__device__ __forceinline__ void load_tile_agent(const double* __restrict__ M, int ld, int r0, int k0, int tid, d2 (&v)[1024 / NT])
{
    const int r = r0 + 2 * (tid & 63);
#pragma unroll
    for (int it = 0; it < 1024 / NT; ++it) {
        const int k = k0 + it * (NT / 64) + (tid >> 6);
        const double* p = M + r + (size_t)k * ld;
        v[it].x = ld_agent(p);
        v[it].y = ld_agent(p + 1);
    }
}

// ... with the edge handling of load_tile<true> (the fronts of the sparse backend: k_front_panel_step)
template <int NT>
__device__ __forceinline__ void load_tile_agent_chk(const double* __restrict__ M, int ld, int r0, int k0, int nrows, int kdim, int tid, d2 (&v)[1024 / NT])
{
    const int r = r0 + 2 * (tid & 63);
#pragma unroll
    for (int it = 0; it < 1024 / NT; ++it) {
        const int k = k0 + it * (NT / 64) + (tid >> 6);
        d2 t = {0.0, 0.0};
        if (k < kdim) {
            const double* p = M + r + (size_t)k * ld;
            if (r < nrows) t.x = ld_agent(p);
            if (r + 1 < nrows) t.y = ld_agent(p + 1);
        }
        v[it] = t;
    }
}

template <int NT>
__device__ __forceinline__ void store_tile(double* __restrict__ S, int tid, const d2 (&v)[1024 / NT])
{
#pragma unroll
    for (int it = 0; it < 1024 / NT; ++it) {
        const int k = it * (NT / 64) + (tid >> 6);
        *reinterpret_cast<d2*>(S + k * LDS_LD + 2 * (tid & 63)) = v[it];
    }
}

// NEG: the operand is staged negated, so that an accumulator initialised with C ends as C - A diag(w) B^T
template <bool CHECK, int NT, bool NEG = false, bool AGENT = false>
__device__ __forceinline__ void scale_tile(const double* __restrict__ w, int k0, int kdim, int tid, d2 (&v)[1024 / NT])
{
    if (!w && !NEG) return;
#pragma unroll
    for (int it = 0; it < 1024 / NT; ++it) {
        const int k = k0 + it * (NT / 64) + (tid >> 6);
        double s = w ? ((!CHECK || k < kdim) ? (AGENT ? ld_agent(w + k) : ((const __attribute__((address_space(1))) double*)w)[k]) : 0.0) : 1.0;
        if (NEG) s = -s;
        v[it].x *= s;
        v[it].y *= s;
    }
}

// diagonal-block factorisation on an LDS-resident block (defined with k_potrf_diag below)
template <bool LDLT, int NWAVES>
__device__ __forceinline__ void potrf_block(double* __restrict__ Tb, double* __restrict__ rds, int nb, int kglobal, int* __restrict__ info, double* __restrict__ rdiag,
                                            double* __restrict__ dvec, double* __restrict__ Aout, int lda, double* __restrict__ pack, double* __restrict__ w16,
                                            long long* __restrict__ ts = nullptr, int* __restrict__ cnt = nullptr, int nactive = 8,
                                            const double* __restrict__ fetch_scratch = nullptr, const int* __restrict__ fetch_flags = nullptr, int fetch_token = 0,
                                            const int* __restrict__ fetch_abort = nullptr, bool pack_lkk = false);
__device__ __forceinline__ int tb_index(int bi, int bj);
__device__ __forceinline__ void tile_store(double* __restrict__ blk, int lane, d4 t);
constexpr int FUSE_ROLES = 9, FUSE_OWN = 4;  // workgroups that share the next diagonal block of a fused trailing update (owner + 8 helpers), blocks per workgroup
template <int NT, bool PERSIST>
__device__ __forceinline__ bool fused_next_diag(const SyrkArgs& a, double* __restrict__ smem, int role);
template <int NT, int MTC, int MTR, bool PERSIST, int NSTRIP>
__device__ __forceinline__ bool panel_follow(const SyrkArgs& a, double* __restrict__ smem, const d4 (&acc)[MTC][MTR], int row0, int wr, int wc, bool first_row);
// the fused next-panel factorisation: its workgroups stage a whole 128 x 128 operand panel (147 KB: one workgroup per CU, which the kernel's
// 147 VGPRs impose anyway); the 36 tile blocks + 64 doubles of pivots reuse that LDS afterwards
constexpr int FUSED_LDS_BYTES = (128 * (128 + 16) + 128) * 8;  // whole operand panel of the next diagonal block + D; the tile blocks reuse it

// One tile of a fused trailing update (K <= 128: one panel): accumulators start from C, the column operand is staged negated (pure-store epilogue),
// ALL operand stages are requested at once into registers (with one 147 KB-LDS workgroup per CU nothing else hides a stage's load latency: the
// one-stage-ahead loop spent 2.9 us per 16-column stage, 23 us per tile, for 4 us of matrix-core work), then stream through two LDS buffers.
// Tiles of the first column (tj == 0) are the NEXT panel: their workgroup keeps the updated rows and solves them behind the factorisation of the
// diagonal block (panel_follow).
// PERSIST (k_chol_persistent: every round of the factorisation in ONE launch, no kernel boundary between producer and consumer): operands and C were
// written by other workgroups of the same launch, so they are read with agent-scope loads and written through (sc1) -- MI355X_MICROARCH.md,
// inter-workgroup visibility -- and the caller publishes a flag afterwards.  n is a multiple of 128 there (no edge tiles).  Returns false when a
// bounded wait inside gave up.
// HALF (persistent launch, panel tiles only): the workgroup takes rows [64 h, 64 h + 64) of the tile -- the eight waves as 2 x 4 (32 x 32 each) for the
// update, four strip waves for the substitution.  A panel row's task of round k needs that row's output of round k - 1, so update + substitution of ONE
// workgroup (28 + 13 us for a whole tile) bound the round from below whatever the diagonal block does; two workgroups per row halve both.  Every element
// receives the same products in the same order as in the whole-tile form.
__device__ __forceinline__ bool chol_wait3(const int* p0, int w0, const int* p1, int w1, const int* p2, int w2, int* abort_w, int sleep);
// SUMSUB (the trailing updates of the sparse fronts, launch_front_updates): the products are summed from zero and subtracted from C at the end, the arithmetic of
// k_syrk_lower<EPI_SUBTRACT> -- the quasi-definite fronts lose accuracy when the accumulators start from C -- with this function's
// operand staging (every stage requested at once) and, with HALF, two workgroups per tile.
// QUARTER (with HALF, front updates only): rows [32 h, 32 h + 32), the waves as 1 x 8 (32 x 16 each) -- four workgroups per tile where even halves leave the chip empty.
// FOLLOW (k_front_panel_step, with HALF and SUMSUB): the operand rows are being solved by other workgroups of the same launch, 16 columns at a time -- K stage kt
// is requested when both row strips have counted step kt (late_p[q][kt] reaching late_w[q]), so the tile's K loop runs one stage behind the panel solve instead
// of starting after it.  Same products in the same order.
template <int NT, bool PERSIST, bool HALF = false, bool SUMSUB = false, bool QUARTER = false, bool FOLLOW = false>
__device__ __forceinline__ bool fused_tile(const SyrkArgs& a, const int ti, const int tj, double* __restrict__ smem, const int h = 0)
{
    static_assert(!QUARTER || HALF, "QUARTER refines HALF");
    static_assert(!FOLLOW || (HALF && SUMSUB && !PERSIST), "FOLLOW: the front tiles");
    constexpr int SPLIT = QUARTER ? 4 : (HALF ? 2 : 1);
    constexpr int WR = 4 / SPLIT, WC = 8 / WR;
    static_assert(NT == 64 * WR * WC, "eight waves");
    constexpr int SUBR = 32, SUBC = TS / WC;
    constexpr int MTR = 2, MTC = SUBC / 16;
    constexpr int ROWS = TS / SPLIT;  // rows of the tile this workgroup works on
    const int row0 = ti * TS + (HALF ? ROWS * h : 0), col0 = tj * TS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const bool edge = !PERSIST && ((row0 + TS > a.n) || (col0 + TS > a.n) || a.unaligned);
    const bool skip_wave = (ti == tj) && ((HALF ? ROWS * h : 0) + (wr + 1) * SUBR <= wc * SUBC);  // sub-tile strictly above the diagonal

    d4 acc[MTC][MTR];
#pragma unroll
    for (int x = 0; x < MTC; ++x)
#pragma unroll
        for (int y = 0; y < MTR; ++y) acc[x][y] = (d4){0.0, 0.0, 0.0, 0.0};
    d4 cfetch[SUMSUB ? MTC : 1][SUMSUB ? MTR : 1];  // SUMSUB: C waits here for the sum
    if (!skip_wave) {
        // C is fetched BEFORE the K loop (its latency hides behind the first operand stages) instead of read-modify-written after it
#pragma unroll
        for (int x = 0; x < MTC; ++x)
#pragma unroll
            for (int y = 0; y < MTR; ++y) {
                const int gi = row0 + wr * SUBR + y * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gj = col0 + wc * SUBC + x * 16 + (lane >> 4) + 4 * r;
                    const bool ok = gi < a.n && gj < a.n && gi >= gj;
                    const double* cp = a.C + (ok ? (size_t)gi + (size_t)gj * a.ldc : 0);
                    const double cv = PERSIST ? ld_agent(cp) : *cp;
                    if constexpr (SUMSUB) cfetch[x][y][r] = ok ? cv : 0.0;
                    else acc[x][y][r] = ok ? cv : 0.0;
                }
            }
    }
    const bool tr_row = PERSIST && HALF && a.fuse_tr2 && threadIdx.x == 0 && ti == 1 && tj == 0 && h == 0;  // PIQP_AMD_DEBUG=chol_trace: the first panel row's own timeline
    if (tr_row) a.fuse_tr2[48] = wall_clock64();
    // (persistent launch, panel tiles: the operand rows come from the panel tasks of the round before, later than the tile itself -- wait for them with the
    // tile's loads in flight; front panel steps in one launch, k_front_panel_step: the operand rows are solved by other workgroups of the same launch)
    if constexpr (!FOLLOW) {
        if (a.late_p[0] && !chol_wait3(nullptr, 0, a.late_p[0], a.late_w[0], a.late_p[1], a.late_w[1], a.fuse_abort, 0)) return false;
    }
    if (tr_row) a.fuse_tr2[49] = wall_clock64();
    const bool dbg_tile = FUSE_TS_ON && !PERSIST && a.fuse_ts && threadIdx.x == 0 && ti == 2 && tj == 1;  // debugging aid: an ordinary tile's timeline
    if (dbg_tile) a.fuse_ts[84] = clock64();
    const bool tr_tile = PERSIST && !HALF && a.fuse_tr2 && threadIdx.x == 0 && ti == 5 && tj == 3;  // PIQP_AMD_DEBUG=chol_trace: one bulk tile per round
    if (tr_tile) a.fuse_tr2[40] = wall_clock64();
    const int nkt = (a.kdim + BK - 1) / BK;  // <= 8
    if constexpr (FOLLOW) {
        constexpr int PER = 1024 / NT;
        double* A2 = smem;                     // [2][BK][LDS_LD]: stage kt in slot kt & 1
        double* B2 = smem + 2 * BK * LDS_LD;
#pragma unroll 1
        for (int kt = 0; kt < nkt; ++kt) {
            if (!chol_wait3(nullptr, 0, a.late_p[0] + kt, a.late_w[0], a.late_p[1] + kt, a.late_w[1], a.fuse_abort, 1)) return false;  // (polls 0.4 us apart: a grid of tight pollers slows what it waits for)
            const int k0 = kt * BK;
            d2 pa0[PER], pb0[PER];
            // (agent-scope loads: a 128-byte line at a strip boundary may sit in this XCD's L2 with the old values of the neighbouring strip)
            if ((tid & 63) < ROWS / 2) load_tile_agent_chk<NT>(a.A, a.lda, row0, k0, a.n, a.kdim, tid, pa0);
            else { for (int it = 0; it < PER; ++it) pa0[it] = (d2){0.0, 0.0}; }
            load_tile_agent_chk<NT>(a.B, a.ldb, col0, k0, a.n, a.kdim, tid, pb0);
            scale_tile<true, NT, !SUMSUB, true>(a.w, k0, a.kdim, tid, pb0);
            const int q = kt & 1;
            store_tile<NT>(A2 + q * BK * LDS_LD, tid, pa0);
            store_tile<NT>(B2 + q * BK * LDS_LD, tid, pb0);
            __syncthreads();  // (also: every wave is done with the other slot's previous stage before anybody writes it again)
            if (!skip_wave) {
                const double* Asb = A2 + q * BK * LDS_LD + wr * SUBR + (lane & 15);
                const double* Bsb = B2 + q * BK * LDS_LD + wc * SUBC + (lane & 15);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int kk = ks * 4 + (lane >> 4);
                    double af[MTR], bf[MTC];
#pragma unroll
                    for (int u = 0; u < MTR; ++u) af[u] = Asb[kk * LDS_LD + u * 16];
#pragma unroll
                    for (int u = 0; u < MTC; ++u) bf[u] = Bsb[kk * LDS_LD + u * 16];
#pragma unroll
                    for (int x = 0; x < MTC; ++x)
#pragma unroll
                        for (int y = 0; y < MTR; ++y)
                            acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[x], af[y], acc[x][y], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    } else {
        constexpr int PER = 1024 / NT;
        d2 pa[8][PER], pb[8][PER];
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
            if (kt < nkt) {
                const int k0 = kt * BK;
                if constexpr (HALF) {
                    // ROWS rows of the row operand (the rest of the stage stays unused), the whole column operand
                    const bool chk = !PERSIST && (edge || (k0 + BK > a.kdim));
                    // (operand rows other workgroups of THIS launch solved and wrote through: a 128-byte line at a strip boundary may sit in this XCD's L2 with
                    // the old values of the neighbouring strip -- agent-scope loads)
                    const bool agent = !PERSIST && a.late_p[0] != nullptr;
                    if ((tid & 63) < ROWS / 2) {
                        if (agent) load_tile_agent_chk<NT>(a.A, a.lda, row0, k0, a.n, a.kdim, tid, pa[kt]);
                        else if (chk) load_tile<true, NT>(a.A, a.lda, row0, k0, a.n, a.kdim, tid, pa[kt]);
                        else load_tile<false, NT>(a.A, a.lda, row0, k0, a.n, a.kdim, tid, pa[kt]);
                    } else { for (int it = 0; it < PER; ++it) pa[kt][it] = (d2){0.0, 0.0}; }
                    if (agent) load_tile_agent_chk<NT>(a.B, a.ldb, col0, k0, a.n, a.kdim, tid, pb[kt]);
                    else if (chk) load_tile<true, NT>(a.B, a.ldb, col0, k0, a.n, a.kdim, tid, pb[kt]);
                    else load_tile<false, NT>(a.B, a.ldb, col0, k0, a.n, a.kdim, tid, pb[kt]);
                } else if constexpr (PERSIST && AGENT_OPERANDS) {
                    load_tile_agent<NT>(a.A, a.lda, row0, k0, tid, pa[kt]); load_tile_agent<NT>(a.B, a.ldb, col0, k0, tid, pb[kt]);
                } else {
                    if (edge || (k0 + BK > a.kdim)) { load_tile<true, NT>(a.A, a.lda, row0, k0, a.n, a.kdim, tid, pa[kt]); load_tile<true, NT>(a.B, a.ldb, col0, k0, a.n, a.kdim, tid, pb[kt]); }
                    else { load_tile<false, NT>(a.A, a.lda, row0, k0, a.n, a.kdim, tid, pa[kt]); load_tile<false, NT>(a.B, a.ldb, col0, k0, a.n, a.kdim, tid, pb[kt]); }
                }
            }
        }
        // The stages go through LDS four at a time (8 x 18 KB: the workgroup has the 145 KB of the diagonal-block role anyway): two groups, three barriers,
        // and sixteen k-slices back to back whose LDS reads overlap -- with one stage per barrier the K loop of a HALF tile took 12 us for 3.4 us of matrix-core
        // work per wave, and that loop sits in every panel row's round (update, substitution, publication: the cycle that bounds the late rounds).
        // Same k order as before: bitwise the same tile.
        constexpr int GRP = 4;
        double* A4 = smem;                       // [GRP][BK][LDS_LD]
        double* B4 = smem + GRP * BK * LDS_LD;   // [GRP][BK][LDS_LD]
        static_assert(2 * GRP * BK * LDS_LD * (int)sizeof(double) <= FUSED_LDS_BYTES, "four stages per operand must fit");
#pragma unroll
        for (int g0 = 0; g0 < 8; g0 += GRP) {
            if (g0 < nkt) {
                if (g0 > 0) __syncthreads();  // every wave has read the previous group
#pragma unroll
                for (int q = 0; q < GRP; ++q) {
                    const int kt = g0 + q;
                    if (kt < nkt) {
                        scale_tile<true, NT, !SUMSUB, PERSIST>(a.w, kt * BK, a.kdim, tid, pb[kt]);
                        store_tile<NT>(A4 + q * BK * LDS_LD, tid, pa[kt]);
                        store_tile<NT>(B4 + q * BK * LDS_LD, tid, pb[kt]);
                    }
                }
                __syncthreads();
                if (g0 == 0) {
                    if (dbg_tile) a.fuse_ts[85] = clock64();  // first operand stages in LDS
                    if (tr_tile) a.fuse_tr2[41] = wall_clock64();
                    if (tr_row) a.fuse_tr2[50] = wall_clock64();
                }
                if (!skip_wave) {
#pragma unroll
                    for (int q = 0; q < GRP; ++q) {
                        const int kt = g0 + q;
                        if (kt < nkt) {
                            const double* Asb = A4 + q * BK * LDS_LD + wr * SUBR + (lane & 15);
                            const double* Bsb = B4 + q * BK * LDS_LD + wc * SUBC + (lane & 15);
#pragma unroll
                            for (int ks = 0; ks < 4; ++ks) {
                                const int kk = ks * 4 + (lane >> 4);
                                double af[MTR], bf[MTC];
#pragma unroll
                                for (int u = 0; u < MTR; ++u) af[u] = Asb[kk * LDS_LD + u * 16];
#pragma unroll
                                for (int u = 0; u < MTC; ++u) bf[u] = Bsb[kk * LDS_LD + u * 16];
#pragma unroll
                                for (int x = 0; x < MTC; ++x)
#pragma unroll
                                    for (int y = 0; y < MTR; ++y)
                                        acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[x], af[y], acc[x][y], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();  // the K loop is done with the LDS (the epilogue / panel_follow reuse it)
    }
    if (dbg_tile) a.fuse_ts[86] = clock64();  // K loop done (the accumulators' first use waited for C)
    if (tr_tile) a.fuse_tr2[42] = wall_clock64();
    if (tr_row) a.fuse_tr2[51] = wall_clock64();
    if constexpr (!QUARTER) {
        if (a.fuse_cnt && tj == 0 && a.fuse_pack) return panel_follow<NT, MTC, MTR, PERSIST, ROWS / 16>(a, smem, acc, row0, wr, wc, ti == 1);
    }
    if (skip_wave) return true;
    // epilogue: lane holds rows gi (consecutive over lane&15) and columns gj = base + (lane>>4) + 4*r; the accumulator started from C
#pragma unroll
    for (int x = 0; x < MTC; ++x) {
#pragma unroll
        for (int y = 0; y < MTR; ++y) {
            const int gi = row0 + wr * SUBR + y * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gj = col0 + wc * SUBC + x * 16 + (lane >> 4) + 4 * r;
                if (gi < a.n && gj < a.n && gi >= gj && gj < a.ncol) {
                    double* cp = a.C + (size_t)gi + (size_t)gj * a.ldc;
                    if (PERSIST) st_agent(cp, acc[x][y][r]);
                    else if constexpr (SUMSUB) *(__attribute__((address_space(1))) double*)cp = cfetch[x][y][r] - acc[x][y][r];
                    else *cp = acc[x][y][r];
                }
            }
        }
    }
    if (dbg_tile) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); a.fuse_ts[87] = clock64(); }  // stores drained
    if (tr_tile) { a.fuse_tr2[43] = wall_clock64(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); a.fuse_tr2[44] = wall_clock64(); }
    return true;
}

// WR x WC = waves per tile (rows x columns):
//   2 x 2 -> 256 threads, 64 x 64 per wave (throughput shape, 2 workgroups per CU);
//   4 x 4 -> 1024 threads, 32 x 32 per wave (low-latency shape for short K: a quarter of the MFMA chain per wave, used for the unfused
//            trailing updates);
//   4 x 2 -> 512 threads, 32 x 64 per wave: the fused trailing update + next diagonal block (fused_tile / fused_next_diag), whose in-register
//            block factorisation needs more than the 128 VGPRs a 1024-thread workgroup can have.
template <int EPI, int WR, int WC>
__device__ __forceinline__ void syrk_lower_body(const SyrkArgs& a, const int block_x)
{
    constexpr int NT = 64 * WR * WC;
    constexpr int MTR = 8 / WR, MTC = 8 / WC;     // MFMA tiles per wave: rows, columns
    constexpr int SUBR = TS / WR, SUBC = TS / WC;  // rows / columns of C per wave
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* As = smem;                    // [2][BK][LDS_LD]
    double* Bs = smem + 2 * BK * LDS_LD;  // [2][BK][LDS_LD]

    // linear block id -> lower-triangular tile (ti >= tj); split launches map several K-slices onto one tile
    int bid = block_x;
    if constexpr (EPI == EPI_SUBTRACT_POTRF) {
        // tile (0, 0) of the trailing matrix = the next diagonal block: updated AND factored by the first FUSE_ROLES workgroups together (owner +
        // helpers); the other tiles follow, the first tile column (= the next panel below its diagonal block) first: its workgroups also solve
        // that panel (panel_follow)
        if (bid < FUSE_ROLES) { fused_next_diag<NT, false>(a, smem, bid); return; }
        bid -= FUSE_ROLES - 1;
        const int T = (a.n + TS - 1) / TS;
        int ti, tj;
        if (bid <= T - 1) { ti = bid; tj = 0; }
        else {
            const int bb = bid - T;  // lower triangle of the (T - 1) x (T - 1) rest
            int t2 = (int)((sqrt(8.0 * (double)bb + 1.0) - 1.0) * 0.5);
            while ((t2 + 1) * (t2 + 2) / 2 <= bb) ++t2;
            while (t2 * (t2 + 1) / 2 > bb) --t2;
            ti = t2 + 1; tj = bb - t2 * (t2 + 1) / 2 + 1;
        }
        (void)fused_tile<NT, false>(a, ti, tj, smem);
        return;
    } else {
    const int b = a.tile_begin + bid / a.k_split;
    const int kslice = bid % a.k_split;
    int ti, tj;
    if (a.tile_order) {
        const int pk = a.tile_order[b];
        ti = pk >> 16; tj = pk & 0xffff;
    } else {
        ti = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
        while ((ti + 1) * (ti + 2) / 2 <= b) ++ti;
        while (ti * (ti + 1) / 2 > b) --ti;
        tj = b - ti * (ti + 1) / 2;
    }
    if (a.first_col_only) { ti = b; tj = 0; }
    const int row0 = ti * TS, col0 = tj * TS;
    if (col0 >= a.ncol) return;  // (a tile column beyond the written columns)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const bool edge = (row0 + TS > a.n) || (col0 + TS > a.n) || a.unaligned;
    const bool skip_wave = (ti == tj) && ((wr + 1) * SUBR <= wc * SUBC);  // sub-tile strictly above the diagonal

    d4 acc[MTC][MTR];
#pragma unroll
    for (int x = 0; x < MTC; ++x)
#pragma unroll
        for (int y = 0; y < MTR; ++y) acc[x][y] = (d4){0.0, 0.0, 0.0, 0.0};

    const int nkt_all = (a.kdim + BK - 1) / BK;
    const int kt_per = (nkt_all + a.k_split - 1) / a.k_split;
    const int kt_begin = kslice * kt_per;
    const int nkt = max(0, min(nkt_all, kt_begin + kt_per) - kt_begin);
    d2 va[1024 / NT], vb[1024 / NT];
    if (nkt > 0) {
        const int k0 = kt_begin * BK;
        const bool chk = edge || (k0 + BK > a.kdim);
        if (chk) { load_tile<true, NT>(a.A, a.lda, row0, k0, a.n, a.kdim, tid, va); load_tile<true, NT>(a.B, a.ldb, col0, k0, a.n, a.kdim, tid, vb); scale_tile<true, NT>(a.w, k0, a.kdim, tid, vb); }
        else { load_tile<false, NT>(a.A, a.lda, row0, k0, a.n, a.kdim, tid, va); load_tile<false, NT>(a.B, a.ldb, col0, k0, a.n, a.kdim, tid, vb); scale_tile<false, NT>(a.w, k0, a.kdim, tid, vb); }
        store_tile<NT>(As, tid, va);
        store_tile<NT>(Bs, tid, vb);
    }
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1 < nkt);
        if (more) {
            const int k0 = (kt_begin + kt + 1) * BK;
            const bool chk = edge || (k0 + BK > a.kdim);
            if (chk) { load_tile<true, NT>(a.A, a.lda, row0, k0, a.n, a.kdim, tid, va); load_tile<true, NT>(a.B, a.ldb, col0, k0, a.n, a.kdim, tid, vb); scale_tile<true, NT>(a.w, k0, a.kdim, tid, vb); }
            else { load_tile<false, NT>(a.A, a.lda, row0, k0, a.n, a.kdim, tid, va); load_tile<false, NT>(a.B, a.ldb, col0, k0, a.n, a.kdim, tid, vb); scale_tile<false, NT>(a.w, k0, a.kdim, tid, vb); }
        }
        if (!skip_wave) {
            const double* Asb = As + cur * BK * LDS_LD + wr * SUBR + (lane & 15);
            const double* Bsb = Bs + cur * BK * LDS_LD + wc * SUBC + (lane & 15);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int kk = ks * 4 + (lane >> 4);
                double af[MTR], bf[MTC];
#pragma unroll
                for (int q = 0; q < MTR; ++q) af[q] = Asb[kk * LDS_LD + q * 16];
#pragma unroll
                for (int q = 0; q < MTC; ++q) bf[q] = Bsb[kk * LDS_LD + q * 16];
#pragma unroll
                for (int x = 0; x < MTC; ++x)
#pragma unroll
                    for (int y = 0; y < MTR; ++y)
                        acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[x], af[y], acc[x][y], 0, 0, 0);
            }
        }
        if (more) {
            store_tile<NT>(As + (cur ^ 1) * BK * LDS_LD, tid, va);
            store_tile<NT>(Bs + (cur ^ 1) * BK * LDS_LD, tid, vb);
        }
        __syncthreads();
    }

    if (skip_wave) return;
    if (a.part) {
        // split launch: raw partial tile (column-major 128 x 128) for k_syrk_tail_reduce; fixed slot per (tile, slice)
        double* P = a.part + (size_t)block_x * TS * TS;
#pragma unroll
        for (int x = 0; x < MTC; ++x)
#pragma unroll
            for (int y = 0; y < MTR; ++y)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    P[(wr * SUBR + y * 16 + (lane & 15)) + (size_t)(wc * SUBC + x * 16 + (lane >> 4) + 4 * r) * TS] = acc[x][y][r];
        return;
    }
    // epilogue: lane holds rows gi (consecutive over lane&15) and columns gj = base + (lane>>4) + 4*r
#pragma unroll
    for (int x = 0; x < MTC; ++x) {
#pragma unroll
        for (int y = 0; y < MTR; ++y) {
            const int gi = row0 + wr * SUBR + y * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gj = col0 + wc * SUBC + x * 16 + (lane >> 4) + 4 * r;
                if (gi < a.n && gj < a.n && gi >= gj && gj < a.ncol) {
                    const size_t ci = (size_t)gi + (size_t)gj * a.ldc;
                    const double v = acc[x][y][r];
                    if (EPI == EPI_ASSEMBLE) {
                        double base = a.Pfull[(size_t)gi + (size_t)gj * a.ldp];
                        if (gi == gj) base += a.x_reg[gi];
                        if (a.ATA) base += a.dinv * a.ATA[(size_t)gi + (size_t)gj * a.ldata];
                        ((__attribute__((address_space(1))) double*)a.C)[ci] = base + v;
                    } else if (EPI == EPI_SUBTRACT) {
                        ((__attribute__((address_space(1))) double*)a.C)[ci] -= v;  // (explicit global address space: the front kernels build their arguments from a job record in memory)
                    } else {
                        ((__attribute__((address_space(1))) double*)a.C)[ci] = v;
                    }
                }
            }
        }
    }
    }
}

template <int EPI, int WR, int WC>
__global__ __launch_bounds__(64 * WR * WC, (WR * WC == 4) ? 2 : (WR * WC == 8 ? 2 : 4)) void k_syrk_lower(SyrkArgs a)
{
    syrk_lower_body<EPI, WR, WC>(a, (int)blockIdx.x);
}

// The trailing update of panel `panel` of MANY independent fronts (the big fronts of one level of a sparse assembly tree) in one launch:
// blockIdx.y = front, blockIdx.x = tile of its trailing matrix (fronts with fewer tiles or fewer panels leave).
__device__ __forceinline__ bool front_panel(const FrontJob& j, int panel, int& k, int& nb, int& rs)
{
    k = panel * FACTOR_NB;
    if (k >= j.w) return false;
    nb = min(FACTOR_NB, j.w - k);
    rs = j.f - k - nb;
    return true;
}
template <int WR, int WC>
__global__ __launch_bounds__(64 * WR * WC, (WR * WC == 4) ? 2 : (WR * WC == 8 ? 2 : 4)) void k_syrk_lower_fronts(const FrontJob* __restrict__ jobs, int panel, int kind)
{
    const FrontJob j = jobs[blockIdx.y];
    if (kind >= 0 && j.kind != kind) return;  // (kind: only the fronts of that kind, -1: all)
    int k, nb, rs;
    if (!front_panel(j, panel, k, nb, rs) || rs <= 0) return;
    const int T = (rs + TS - 1) / TS;
    if ((int)blockIdx.x >= T * (T + 1) / 2) return;
    SyrkArgs a;
    a.n = rs; a.kdim = nb;
    a.A = j.F + (k + nb) + (size_t)k * j.f; a.lda = j.f; a.B = a.A; a.ldb = j.f; a.w = j.dvec + k;
    a.C = j.F + (k + nb) + (size_t)(k + nb) * j.f; a.ldc = j.f;
    a.unaligned = ((j.f & 1) || (reinterpret_cast<uintptr_t>(a.A) & 15)) ? 1 : 0;
    if (j.multi) {  // the update-matrix region [w, f)^2 receives all panels in ONE pass after the last panel (k_syrk_multi_fronts): here only the pivot columns still to come
        a.ncol = j.w - (k + nb);
        if (a.ncol <= 0) return;
    }
    syrk_lower_body<EPI_SUBTRACT, WR, WC>(a, (int)blockIdx.x);
}
// round 4: the update-matrix region of the fronts with several panels (FrontJob::multi), every panel's products tile by tile in panel order -- the tile is read from
// and written to HBM once instead of once per panel (it stays in the L2 between the panels); the sums are those of the per-panel launches: each panel's
// products summed from zero in k order and subtracted, panel after panel
template <int WR, int WC>
__global__ __launch_bounds__(64 * WR * WC, (WR * WC == 4) ? 2 : (WR * WC == 8 ? 2 : 4)) void k_syrk_multi_fronts(const FrontJob* __restrict__ jobs)
{
    const FrontJob j = jobs[blockIdx.y];
    if (!j.multi || j.kind != 0) return;
    const int u = j.f - j.w;
    const int T = (u + TS - 1) / TS;
    if (u <= 0 || (int)blockIdx.x >= T * (T + 1) / 2) return;
    for (int k = 0; k < j.w; k += FACTOR_NB) {
        SyrkArgs a;
        a.n = u; a.kdim = min(FACTOR_NB, j.w - k);
        a.A = j.F + j.w + (size_t)k * j.f; a.lda = j.f; a.B = a.A; a.ldb = j.f; a.w = j.dvec + k;
        a.C = j.F + j.w + (size_t)j.w * j.f; a.ldc = j.f;
        a.unaligned = ((j.f & 1) || (reinterpret_cast<uintptr_t>(a.A) & 15)) ? 1 : 0;
        syrk_lower_body<EPI_SUBTRACT, WR, WC>(a, (int)blockIdx.x);
        __threadfence_block();
        __syncthreads();  // this workgroup's stores of panel k before its loads of the next panel; the LDS stages are free again
    }
}

// The same update for the levels near the root of an assembly tree, where a handful of tiles is all there is and 250 CUs idle: two workgroups per tile (rows
// [64 h, 64 h + 64)), every operand stage requested at once (fused_tile<HALF, SUMSUB>): a 128 x 128 x 128 tile is 13.6 us of matrix-core work on ONE CU whatever
// the wave shape, and the generic kernel adds a load -> LDS -> barrier round trip per 16-column stage (30 us per launch measured at the top of CONT-201).
// Same products in the same order, summed from zero and subtracted from C: bitwise the tile of k_syrk_lower_fronts.
template <int SPLIT>
__global__ __launch_bounds__(512) void k_syrk_half_fronts(const FrontJob* __restrict__ jobs, int panel, int kind)
{
    const FrontJob j = jobs[blockIdx.y];
    if (kind >= 0 && j.kind != kind) return;
    int k, nb, rs;
    if (!front_panel(j, panel, k, nb, rs) || rs <= 0) return;
    const int T = (rs + TS - 1) / TS;
    const int b = (int)blockIdx.x / SPLIT, h = (int)blockIdx.x % SPLIT;
    if (b >= T * (T + 1) / 2) return;
    int ti = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= b) ++ti;
    while (ti * (ti + 1) / 2 > b) --ti;
    const int tj = b - ti * (ti + 1) / 2;
    if (ti * TS + (TS / SPLIT) * h >= rs) return;  // (nothing in this part)
    SyrkArgs a;
    a.n = rs; a.kdim = nb;
    a.A = j.F + (k + nb) + (size_t)k * j.f; a.lda = j.f; a.B = a.A; a.ldb = j.f; a.w = j.dvec + k;
    a.C = j.F + (k + nb) + (size_t)(k + nb) * j.f; a.ldc = j.f;
    a.unaligned = ((j.f & 1) || (reinterpret_cast<uintptr_t>(a.A) & 15)) ? 1 : 0;
    if (j.multi) {  // (see k_syrk_lower_fronts)
        a.ncol = j.w - (k + nb);
        if (a.ncol <= 0 || tj * TS >= a.ncol) return;
    }
    extern __shared__ __attribute__((aligned(16))) double smem[];
    (void)fused_tile<512, false, true, true, SPLIT == 4>(a, ti, tj, smem, h);
}

// sums the K-slices of a split tile in slice order and applies the epilogue (one workgroup per tile)
template <int EPI>
__global__ __launch_bounds__(256) void k_syrk_tail_reduce(SyrkArgs a)
{
    const int b = a.tile_begin + (int)blockIdx.x;
    int ti, tj;
    if (a.tile_order) {
        const int pk = a.tile_order[b];
        ti = pk >> 16; tj = pk & 0xffff;
    } else {
        ti = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
        while ((ti + 1) * (ti + 2) / 2 <= b) ++ti;
        while (ti * (ti + 1) / 2 > b) --ti;
        tj = b - ti * (ti + 1) / 2;
    }
    if (a.first_col_only) { ti = b; tj = 0; }
    const int row0 = ti * TS, col0 = tj * TS;
    const double* P0 = a.part + (size_t)blockIdx.x * a.k_split * TS * TS;
    {
        const int idx = (int)blockIdx.y * 256 + (int)threadIdx.x;  // one element per thread
        const int li = idx & (TS - 1), lj = idx >> 7;
        const int gi = row0 + li, gj = col0 + lj;
        if (gi >= a.n || gj >= a.n || gi < gj) return;
        double v = 0.0;
#pragma unroll 8
        for (int sl = 0; sl < a.k_split; ++sl) v += P0[(size_t)sl * TS * TS + idx];
        const size_t ci = (size_t)gi + (size_t)gj * a.ldc;
        if (EPI == EPI_ASSEMBLE) {
            double base = a.Pfull[(size_t)gi + (size_t)gj * a.ldp];
            if (gi == gj) base += a.x_reg[gi];
            if (a.ATA) base += a.dinv * a.ATA[(size_t)gi + (size_t)gj * a.ldata];
            a.C[ci] = base + v;
        } else if (EPI == EPI_SUBTRACT) {
            a.C[ci] -= v;
        } else {
            a.C[ci] = v;
        }
    }
}

static int g_slots = 0;  // resident workgroup slots of the SYRK kernel (2 per CU)
static int syrk_slots()
{
    if (g_slots == 0) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        g_slots = 2 * (cus > 0 ? cus : 256);
    }
    return g_slots;
}
// tail plan: the last `rem` tiles of a launch that would occupy < 60 % of one more round are split along K
static void syrk_tail_plan(int n, int kdim, int& rem, int& ksplit)
{
    const int T = div_up(n, TS);
    const int ntiles = T * (T + 1) / 2;
    const int slots = syrk_slots();
    const int nkt = div_up(kdim, BK);
    rem = ntiles % slots;
    ksplit = 1;
    if (ntiles > slots && rem > 0 && rem * 10 < slots * 6) {
        int s = slots / rem;
        s = std::min(s, nkt / 4);
        if (s >= 2) ksplit = s;
    }
    if (ksplit == 1) rem = 0;
}
size_t syrk_split_workspace_doubles(int n, int kdim)
{
    int rem, ks;
    syrk_tail_plan(n, kdim, rem, ks);
    return (size_t)rem * ks * TS * TS;
}

// device copy of the XCD-aware tile order for a T x T lower-triangular tile grid (built once per device and T, never freed: a few KB)
static const int* syrk_tile_order(int T)
{
    static std::mutex mu;
    static std::map<std::pair<int, int>, int*> cache;
    if (T < 16) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find({dev, T});
    if (it != cache.end()) return it->second;
    constexpr int S = 8, NX = 8;  // patch of S x S tiles (64 = the workgroups one XCD runs at a time), NX XCDs
    std::vector<int> blocked;
    for (int SI = 0; SI * S < T; ++SI)
        for (int SJ = 0; SJ <= SI; ++SJ)
            for (int ti = SI * S; ti < std::min(T, SI * S + S); ++ti)
                for (int tj = SJ * S; tj < SJ * S + S && tj <= ti; ++tj) blocked.push_back(ti << 16 | tj);
    const int nt = (int)blocked.size();
    std::vector<int> start(NX + 1, 0);
    for (int x = 0; x < NX; ++x) start[x + 1] = start[x] + nt / NX + (x < nt % NX ? 1 : 0);
    std::vector<int> order(nt);
    for (int b = 0; b < nt; ++b) order[b] = blocked[start[b % NX] + b / NX];
    int* d = nullptr;
    if (hipMalloc(&d, sizeof(int) * nt) != hipSuccess) { (void)hipGetLastError(); cache[{dev, T}] = nullptr; return nullptr; }
    ++alloc_counter();
    if (hipMemcpy(d, order.data(), sizeof(int) * nt, hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(d); d = nullptr; }
    cache[{dev, T}] = d;
    return d;
}

void syrk_prepare(int n) { (void)syrk_tile_order(div_up(n, TS)); }  // the per-device tile-order table, built at create time

template <int EPI>
static void launch_syrk_t(SyrkArgs a, hipStream_t s, double* ws, size_t ws_doubles)
{
    const int T = div_up(a.n, TS);
    const int ntiles = a.first_col_only ? T : T * (T + 1) / 2;
    int rem = 0, ks = 1;
    if (ws && !a.first_col_only) syrk_tail_plan(a.n, a.kdim, rem, ks);
    if (rem > 0 && (size_t)rem * ks * TS * TS > ws_doubles) { rem = 0; ks = 1; }
    const int main_tiles = ntiles - rem;
    a.tile_begin = 0; a.k_split = 1; a.part = nullptr;
    // short inner dimension (factorisation trailing updates, K = 128): latency-bound per tile -> 16-wave shape
    const bool low_latency = a.kdim <= 256;
    a.tile_order = (!low_latency && !a.first_col_only) ? syrk_tile_order(T) : nullptr;
    if (low_latency) hipLaunchKernelGGL((k_syrk_lower<EPI, 4, 4>), dim3(main_tiles), dim3(1024), SYRK_LDS_BYTES, s, a);
    else hipLaunchKernelGGL((k_syrk_lower<EPI, 2, 2>), dim3(main_tiles), dim3(256), SYRK_LDS_BYTES, s, a);
    if (rem > 0) {
        a.tile_begin = main_tiles; a.k_split = ks; a.part = ws;
        hipLaunchKernelGGL((k_syrk_lower<EPI, 2, 2>), dim3(rem * ks), dim3(256), SYRK_LDS_BYTES, s, a);
        hipLaunchKernelGGL(k_syrk_tail_reduce<EPI>, dim3(rem, TS * TS / 256), dim3(256), 0, s, a);
    }
}

void launch_syrk(int epi, const SyrkArgs& args_in, hipStream_t s, double* split_ws, size_t split_ws_doubles)
{
    SyrkArgs a = args_in;
    if (a.n <= 0) return;
    a.unaligned = ((a.lda & 1) || (a.ldb & 1) || (reinterpret_cast<uintptr_t>(a.A) & 15) || (reinterpret_cast<uintptr_t>(a.B) & 15)) ? 1 : 0;
    static PerDeviceOnce attr_set;
    attr_set([&] {
        PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_syrk_lower<EPI_ASSEMBLE, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, SYRK_LDS_BYTES));
        PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_syrk_lower<EPI_SUBTRACT, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, SYRK_LDS_BYTES));
        PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_syrk_lower<EPI_STORE, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, SYRK_LDS_BYTES));
        PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_syrk_lower<EPI_ASSEMBLE, 4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, SYRK_LDS_BYTES));
        PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_syrk_lower<EPI_SUBTRACT, 4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, SYRK_LDS_BYTES));
        PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_syrk_lower<EPI_STORE, 4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, SYRK_LDS_BYTES));
    });
    if (epi == EPI_SUBTRACT_POTRF) {
        static PerDeviceOnce fused_attr;
        fused_attr([&] {
            PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_syrk_lower<EPI_SUBTRACT_POTRF, 4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS_BYTES));
        });
        const int T = div_up(a.n, TS);
        a.tile_begin = 0; a.k_split = 1; a.part = nullptr; a.first_col_only = 0;
        if (!a.fuse_scratch || !a.fuse_flags || a.fuse_nb <= 0 || a.kdim > TS) throw std::runtime_error("fused trailing update: scratch / flags / panel width");
        hipLaunchKernelGGL((k_syrk_lower<EPI_SUBTRACT_POTRF, 4, 2>), dim3(T * (T + 1) / 2 + FUSE_ROLES - 1), dim3(512), FUSED_LDS_BYTES, s, a);
        PQ_HIP(hipGetLastError());
        return;
    }
    switch (epi) {
    case EPI_ASSEMBLE: launch_syrk_t<EPI_ASSEMBLE>(a, s, split_ws, split_ws_doubles); break;
    case EPI_SUBTRACT: launch_syrk_t<EPI_SUBTRACT>(a, s, split_ws, split_ws_doubles); break;
    default: launch_syrk_t<EPI_STORE>(a, s, split_ws, split_ws_doubles); break;
    }
    PQ_HIP(hipGetLastError());
}

// GT (n x m column-major, ld = n) -> row panels of 128 rows stored one after the other, each 128 x m column-major with ld = 128: a 128 x 16 operand stage of
// the fused assembly is then 16 KB of consecutive memory instead of sixteen 1 KB pieces 8 n bytes apart (n = 4096: 32 KB -- outside L2 the persistent launch's
// assembly tasks, which do not run in lockstep like the tiles of k_syrk_lower, got 2 TB/s out of that pattern: 31 us per 128-column chunk)
__global__ __launch_bounds__(256) void k_pack_row_panels(const double* __restrict__ G, int n, int m, double* __restrict__ Gp)
{
    const int k = blockIdx.x, i = blockIdx.y;  // column, row panel
    const int r = threadIdx.x;
    if (r < 128) Gp[((size_t)i * m + k) * 128 + r] = G[(size_t)i * 128 + r + (size_t)k * n];
}
void launch_pack_row_panels(const double* G, int n, int m, double* Gp, hipStream_t s)
{
    if (n <= 0 || m <= 0) return;
    hipLaunchKernelGGL(k_pack_row_panels, dim3(m, n / 128), dim3(128), 0, s, G, n, m, Gp);
    PQ_HIP(hipGetLastError());
}
void launch_syrk_first_col(const SyrkArgs& args_in, int ks, double* ws, hipStream_t s)
{
    SyrkArgs a = args_in;
    if (a.n <= 0) return;
    a.unaligned = ((a.lda & 1) || (a.ldb & 1) || (reinterpret_cast<uintptr_t>(a.A) & 15) || (reinterpret_cast<uintptr_t>(a.B) & 15)) ? 1 : 0;
    static PerDeviceOnce attr_set;
    attr_set([&] { PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_syrk_lower<EPI_ASSEMBLE, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, SYRK_LDS_BYTES)); });
    const int T = div_up(a.n, TS);
    a.first_col_only = 1; a.tile_order = nullptr; a.tile_begin = 0;
    if (ks <= 1 || !ws) {
        a.k_split = 1; a.part = nullptr;
        hipLaunchKernelGGL((k_syrk_lower<EPI_ASSEMBLE, 2, 2>), dim3(T), dim3(256), SYRK_LDS_BYTES, s, a);
    } else {
        a.k_split = ks; a.part = ws;
        hipLaunchKernelGGL((k_syrk_lower<EPI_ASSEMBLE, 2, 2>), dim3(T * ks), dim3(256), SYRK_LDS_BYTES, s, a);
        hipLaunchKernelGGL(k_syrk_tail_reduce<EPI_ASSEMBLE>, dim3(T, TS * TS / 256), dim3(256), 0, s, a);
    }
    PQ_HIP(hipGetLastError());
}

// assembly without inequality rows (m == 0): C_lower = Pfull + diag(x_reg) + dinv*ATA
__global__ __launch_bounds__(256) void k_assemble_no_g(int n, const double* __restrict__ Pf, const double* __restrict__ x_reg, const double* __restrict__ ATA, double dinv, double* __restrict__ C)
{
    const int j = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && i >= j) {
        double v = Pf[i + (size_t)j * n];
        if (i == j) v += x_reg[i];
        if (ATA) v += dinv * ATA[i + (size_t)j * n];
        C[i + (size_t)j * n] = v;
    }
}
void launch_assemble_no_g(int n, const double* Pf, const double* x_reg, const double* ATA, double dinv, double* C, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_assemble_no_g, dim3(div_up(n, 256), n), dim3(256), 0, s, n, Pf, x_reg, ATA, dinv, C);
    PQ_HIP(hipGetLastError());
}


// Stage the lower triangle of an nr x nr block (nr <= NBLK) into LDS as Ls[c * LD + r], zeros above the diagonal,
// identity padding beyond nr.  All loads are unconditional (clamped address + select) and issued 16 at a time per
// thread, so a thread has 16 L2 round trips in flight instead of one per loop iteration.
template <int NBLK, int LD, int NT = 256, bool TRANSPOSE = false>
__device__ __forceinline__ void stage_lower_block(const double* __restrict__ A, int lda, int nr, double* __restrict__ Ls, int tid)
{
    constexpr int PER_THREAD = NBLK * NBLK / NT;
    constexpr int BATCH = 16;
#pragma unroll 1
    for (int b0 = 0; b0 < PER_THREAD; b0 += BATCH) {
        double v[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int idx = (b0 + u) * NT + tid;
            const int r = idx % NBLK, c = idx / NBLK;
            const bool ok = (r < nr) && (c < nr) && (r >= c);
            const double* p = ok ? (A + r + (size_t)c * lda) : A;
            const double t = *p;
            v[u] = ok ? t : ((r == c) ? 1.0 : 0.0);
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int idx = (b0 + u) * NT + tid;
            if (TRANSPOSE) Ls[(idx % NBLK) * LD + (idx / NBLK)] = v[u];  // Ls[r * LD + c]: the backward sweep reads rows of L with consecutive lanes
            else Ls[(idx / NBLK) * LD + (idx % NBLK)] = v[u];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Serial part of the blocked factorisation: diagonal block (<= 128 x 128) and the panel below it.
//
// 16 x 16 tile toolkit.  A diagonal block lives in LDS as 36 column-major 16 x 16 blocks (its lower block triangle; block (bi, bj) at
// tb_index(bi, bj) * 256, element (r, c) of a block at c * 16 + r).  A wave touches a block in one of two register forms:
//   tile form: lane (i = lane & 15, g = lane >> 4), register r  <->  element [i][g + 4 r].  This is the C/D layout of v_mfma_f64_16x16x4
//              AND the layout of its A / B operands: register ks of a tile is the operand of k-step ks.  For X, Y, T in tile form
//                  T += X Y^T   is   T = mfma(Y[ks], X[ks], T), ks = 0..3
//              so a product's result feeds the next product without any data movement, and the LDS reads (lanes of a group on 16
//              consecutive doubles, groups 32 doubles apart) are conflict-free;
//   row form:  every lane holds row (lane & 15) of the block in 16 registers (the four lane groups are replicas); the scalar recurrences
//              (16 x 16 factorisation, substitution, inversion) run in this form with v_readlane broadcasts and no LDS traffic.
__device__ __forceinline__ int tb_index(int bi, int bj) { return bi * (bi + 1) / 2 + bj; }
constexpr int TB_BLOCKS = 36;
constexpr int TB_DOUBLES = TB_BLOCKS * 256;
constexpr int POTRF_LDS_BYTES = (TB_DOUBLES + 8 * 32) * (int)sizeof(double);  // tile blocks + the pivots of the eight steps

__device__ __forceinline__ d4 tile_load(const double* __restrict__ blk, int lane)
{
    const double* p = blk + (lane >> 4) * 16 + (lane & 15);
    return (d4){p[0], p[64], p[128], p[192]};
}
__device__ __forceinline__ void tile_store(double* __restrict__ blk, int lane, d4 t)
{
    double* p = blk + (lane >> 4) * 16 + (lane & 15);
    p[0] = t[0]; p[64] = t[1]; p[128] = t[2]; p[192] = t[3];
}

// LDS hand-over between the lanes of ONE wave (its LDS instructions execute in order; this keeps the compiler from reordering them)
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 1/sqrt(d) to ~1 ulp without the IEEE sqrt + divide chains (two Newton steps on v_rsq_f64)
// PIVOT_NEWTON Newton steps on the hardware's reciprocal (square root): tools/ub/rsq_accuracy.hip measures what one and two steps leave
constexpr int PIVOT_NEWTON = 2;  // (one step: 37 / 20 ulp and NO change in the cycles of the 128-block -- 55 793 against 56 100 -- the loop is bound by instruction issue: round 5)
__device__ __forceinline__ double rsqrt_newton(double d)
{
    double y = __builtin_amdgcn_rsq(d);
    y = y * (1.5 - 0.5 * d * y * y);
    if (PIVOT_NEWTON > 1) y = y * (1.5 - 0.5 * d * y * y);
    return y;
}
__device__ __forceinline__ double rcp_newton(double d)
{
    double y = __builtin_amdgcn_rcp(d);
    y = y * (2.0 - d * y);
    if (PIVOT_NEWTON > 1) y = y * (2.0 - d * y);
    return y;
}

// 16 x 16 diagonal piece in tile form, one rank-1 update on the matrix cores per pivot: column c lives in lane group c & 3, register c >> 2;
// its part below the diagonal, zero everywhere else, IS both operands of  T -= l l^T  (k-slot = the lane group), so a pivot step is a
// handful of VALU instructions and one v_mfma_f64_16x16x4 instead of (15 - c) broadcast + FMA pairs.  Measured on MI355X (tools/dbg_potrf.py):
// a single wave issues ~7 cycles per instruction and a DEPENDENT fp64 MFMA costs ~180 cycles, so a pivot is ~300 cycles either way
// (row-form VALU version 5500 cycles per piece, this one 4500-5000; a variant that computes the next reciprocal pivot one step ahead by a
// scalar recurrence was slower, 5600-6500: its broadcasts still wait for the MFMA in flight and the extra instructions are not free).
//   LDLT == false: Eigen LLT unblocked semantics per SURVEY.md A.4: pivot x <= 0 -> fail, l = sqrt(x), column / l
//   LDLT == true : dense/ldlt_no_pivot.hpp:278-311: unit L below the diagonal, D on it, fail iff pivot == 0
// On return t holds the factor (zeros above the diagonal); lane c < 16 holds the reciprocal pivot of column c in rd (LLT: 1 / l_cc,
// LDLT: 1 / d_c) and d_c in dd.  Returns the first failing column or -1.
// Dpub / rpub / prog (LDS): every FOURTH pivot the four columns just finished -- exactly one tile register -- are published: one unmasked store per
// lane into the block's own LDS tile, their reciprocal pivots into rpub, then *prog = base + c + 1, so that the waves solving the panel below
// (tile_trsm_rt_follow) run four columns behind this factorisation instead of starting after it.  (Publishing every column -- masked stores and a
// fence per pivot, seven waves polling -- doubled the time of this routine: 4800 -> 10 500 cycles.)
// The progress word and the published columns live in LDS.  Through generic `volatile` pointers the compiler emitted FLAT stores / loads with
// sc0 sc1 and a vmcnt(0) drain per publication (~250 cycles per pivot, measured); explicit LDS pointers make them ds_write / ds_read.
typedef __attribute__((address_space(3))) volatile double lds_vdouble;
typedef __attribute__((address_space(3))) volatile int lds_vint;
// Index permutation of the in-register factorisation: the 4 x 4 transpose of a 4-bit index (an involution).  The matrix core fixes which four
// column indices a lane holds (g, g + 4, g + 8, g + 12); relabelling rows AND columns by pi makes those the CONSECUTIVE logical columns
// 4 g .. 4 g + 3, so that the next pivot column usually lives in the same lanes as the current one.
__device__ __forceinline__ constexpr int pi16(int x) { return (x >> 2) + 4 * (x & 3); }
// 16 x 16 block image in LDS (column-major) <-> the permuted tile: lane (ip, g) register r holds logical element [pi(ip)][4 g + r]
__device__ __forceinline__ d4 tile_load_perm(const double* __restrict__ blk, int lane)
{
    const double* p = blk + (lane >> 4) * 64 + pi16(lane & 15);
    return (d4){p[0], p[16], p[32], p[48]};
}
__device__ __forceinline__ void tile_store_perm(double* __restrict__ blk, int lane, d4 t)
{
    double* p = blk + (lane >> 4) * 64 + pi16(lane & 15);
    p[0] = t[0]; p[16] = t[1]; p[32] = t[2]; p[48] = t[3];
}

// In-register factorisation of one 16 x 16 diagonal piece by one wave: t in the PERMUTED tile form above, replaced by the factor (LDLT: D on
// the diagonal).  Published to LDS for the waves that follow (tile_trsm_rt_follow): after every fourth column the four finished columns (natural
// block image at Dpub), their reciprocal pivots rpub[0, 16) (and the pivots D at rpub[16, 32), LDLT), then the progress word.
// One wave, in-order issue: every instruction of the pivot loop is on the clock (tools/ub/factor16.hip: 289 cycles per pivot for the first
// tile-form version, ~100 of them waiting for the matrix core).  What keeps the loop short:
//  * the NEGATED tile s = -T is carried and columns are scaled by +1 / l, so the rank-1 update is s += (-l)(-l)^T without operand negation;
//  * the factor is collected in a second tile (Lo -= column, zero elsewhere) instead of being selected into the working tile: with the diagonal
//    row inside the update operand the matrix core overwrites only entries nobody reads again;
//  * the NEXT pivot column is updated on the vector ALU (one FMA with a lane-read scalar) from the matrix core's result of the PREVIOUS update,
//    so the dependent chain column -> reciprocal square root -> scaled column -> next column never waits for the update in flight; only at the
//    three crossings into another lane group (columns 4, 8, 12) is the column taken from the matrix core.  The FMA reproduces the matrix core's
//    value bit for bit (one fused product per k-slice, the other three slices exact zeros: checked by tools/ub/factor16.hip);
//  * pivots / reciprocals stay in uniform registers and go to LDS once per four columns.
// A non-positive (LDLT: zero) pivot is recorded off the chain and NOT replaced: the rest of a failed block is NaN / Inf, which nobody reads
// (dense/kkt.hpp:83 reports the failure, the caller regularises and factors again).
template <bool LDLT>
__device__ __forceinline__ int factor16_tile(d4& t, int lane, volatile double* Dpub_g, volatile double* rpub_g, volatile int* prog_g, int base)
{
    lds_vdouble* Dpub = (lds_vdouble*)Dpub_g;
    lds_vdouble* rpub = (lds_vdouble*)rpub_g;
    lds_vint* prog = (lds_vint*)prog_g;
    const int ip = lane & 15, g = lane >> 4, il = pi16(ip);  // physical lane row, lane group, logical row
    int failed = -1;
    d4 s = {-t[0], -t[1], -t[2], -t[3]};
    d4 Lo = {0.0, 0.0, 0.0, 0.0};
    double col = s[0];                      // (negated) current pivot column in the lanes of its group, all updates applied
    double dk = -readlane_d(col, 0);
    double rq[4], dq[4];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int gc = c >> 2, rc = c & 3;
        const bool incol = (g == gc);
        const int c1 = (c + 1) & 15, gn = c1 >> 2, rn = c1 & 3;
        const int src1 = 16 * gc + pi16(c1);  // lane of this column's entry in row c + 1
        double r, lneg, yneg = 0.0;
        if (!LDLT) {
            if (!(dk > 0.0) && failed < 0) failed = c;
            r = rsqrt_newton(dk);  // 1 / l
            lneg = (incol && il >= c) ? col * r : 0.0;  // -(column c of L), diagonal included: s_cc r = -d r = -l_cc
            Lo[rc] -= lneg;
        } else {
            if (dk == 0.0 && failed < 0) failed = c;
            r = rcp_newton(dk);    // 1 / d
            yneg = (incol && il > c) ? col : 0.0;       // -(column c of L D)
            lneg = yneg * r;                            // -(column c of L)
            Lo[rc] = (incol && il == c) ? dk : Lo[rc] - lneg;
        }
        rq[rc] = r; dq[rc] = dk;
        if (c < 15) {
            const double l1 = readlane_d(lneg, src1);
            if (rn != 0) {
                // column c + 1 sits in the same lanes: s[rn] carries the updates < c (matrix core, issued a whole pivot ago), update c by FMA
                col = LDLT ? __builtin_fma(yneg, l1, s[rn]) : __builtin_fma(lneg, l1, s[rn]);
                s = LDLT ? __builtin_amdgcn_mfma_f64_16x16x4f64(lneg, yneg, s, 0, 0, 0) : __builtin_amdgcn_mfma_f64_16x16x4f64(lneg, lneg, s, 0, 0, 0);
            } else {
                s = LDLT ? __builtin_amdgcn_mfma_f64_16x16x4f64(lneg, yneg, s, 0, 0, 0) : __builtin_amdgcn_mfma_f64_16x16x4f64(lneg, lneg, s, 0, 0, 0);
                col = s[0];  // next lane group: the one place the chain waits for the matrix core
            }
            dk = -readlane_d(col, 16 * gn + pi16(c1));
        }
        if (rc == 3) {
            // logical columns 4 gc .. 4 gc + 3 are final: the lanes of group gc hold them in their four registers.  No s_waitcnt between the
            // data and the progress word: the LDS executes one wave's instructions in order (a release fence here costs ~350 cycles per
            // publication, measured); the asm statements only pin the compiler's order.
            if (incol) {
#pragma unroll
                for (int q = 0; q < 4; ++q) Dpub[(4 * gc + q) * 16 + il] = Lo[q];
            }
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { rpub[4 * gc + q] = rq[q]; if (LDLT) rpub[16 + 4 * gc + q] = dq[q]; }
            }
            asm volatile("" ::: "memory");
            if (lane == 0) *prog = base + c + 1;
            asm volatile("" ::: "memory");
        }
    }
    t = Lo;
    return failed;
}

// X <- X L^-T for a 16 x 16 lower-triangular L, both in tile form (LDLT: unit diagonal, then X <- X D^-1 when SCALE): column c of X is
// final after c rank-1 updates  X[:, j] -= X[:, c] L[j][c]  (j > c), each ONE MFMA.  Its column-side operand -- column c of -L below the
// diagonal, pre-scaled by 1 / l_cc -- does not depend on X, and its row-side operand is register c >> 2 of the accumulator as it is (the
// other lane groups meet zeros), so the dependent chain is MFMA -> MFMA; the columns are scaled by their reciprocal pivots at the end.
// rdc[r] = reciprocal pivot of column g + 4 r (this lane's columns).
template <bool LDLT, bool SCALE>
__device__ __forceinline__ void tile_trsm_rt(d4& x, const d4& L, const d4& rdc, int lane)
{
    const int i = lane & 15, g = lane >> 4;
    d4 nls;
#pragma unroll
    for (int r = 0; r < 4; ++r) nls[r] = LDLT ? -L[r] : -L[r] * rdc[r];
#pragma unroll
    for (int c = 0; c < 15; ++c) {
        const int g0 = c & 3, r0 = c >> 2;
        const double aop = (g == g0 && i > c) ? nls[r0] : 0.0;
        x = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, x[r0], x, 0, 0, 0);
    }
    if (!LDLT || SCALE) x *= rdc;
}

// The same substitution run four columns BEHIND the factorisation of L by another wave of the workgroup (factor16_tile publishes one tile register
// = four columns of L and their reciprocal pivots in LDS, then *prog): the rank-1 updates of those columns are issued as soon as they exist, so the
// panel is solved about one register's worth of work after the diagonal piece is factored instead of ~3000 cycles.
template <bool LDLT>
__device__ __forceinline__ void tile_trsm_rt_follow(d4& x, const volatile double* Lpub_g, const volatile double* rpub_g, const volatile int* prog_g, int base, int lane)
{
    const lds_vdouble* Lpub = (const lds_vdouble*)Lpub_g;
    const lds_vdouble* rpub = (const lds_vdouble*)rpub_g;
    const lds_vint* prog = (const lds_vint*)prog_g;
    const int i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int r0 = 0; r0 < 4; ++r0) {
        while (*prog < base + 4 * r0 + 4) __builtin_amdgcn_s_sleep(2);  // ~128 cycles between polls
        const double l = Lpub[(g + 4 * r0) * 16 + i];   // register r0 of L in tile form: column 4 r0 + g
        const double rc = LDLT ? 1.0 : rpub[g + 4 * r0];
        const double nls = -(l * rc);
#pragma unroll
        for (int g0 = 0; g0 < 4; ++g0) {
            const int c = 4 * r0 + g0;
            if (c < 15) {
                const double aop = (g == g0 && i > c) ? nls : 0.0;
                x = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, x[r0], x, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] *= rpub[g + 4 * r];
}

// Factorisation of a diagonal block of order nb <= 128 held in LDS (tile blocks Tb, identity-padded beyond nb).  Waves 0..7 own one 16-row
// block row each and run it as a dataflow program -- there is NO workgroup barrier between the steps; the waves meet through two kinds of
// monotonic LDS words (one wave's LDS instructions execute in order, so a word written after the data is seen after the data):
//   prog      columns of the block factored so far (factor16_tile, every fourth column);
//   xdone[w]  steps whose panel piece X_w = tile (w, k) wave w has solved and stored.
// Wave w, for k = 0 .. w - 1:
//   solves its piece of panel k against L_kk four columns behind wave k's factorisation (tile_trsm_rt_follow), stores it, bumps xdone[w];
//   updates the tiles (w, c), k < c < w, of its block row as X_c arrives: T -= X_w (D) X_c^T;
//   updates its own diagonal tile (w, w), which lives in REGISTERS in the permuted form factor16_tile wants (operands re-read from the
//   stored X_w with permuted rows), from its first update to its factorisation: the critical wave k + 1 goes from the last column of L_kk
//   through one 16 x 16 substitution tail and four matrix-core products straight into the factorisation of step k + 1;
// then, as step w: factors its diagonal piece, publishing as it goes; writes pivots, the inverse of L_ww (for the panel kernel and the
// triangular sweeps) and block row w of the factor while the later waves carry on.
// The wave that shares a SIMD with the factoring wave (w = k + 4, waves are dealt round-robin to the four SIMDs) does not follow column by
// column: every matrix-core instruction of a SIMD partner sits between the pivots of the critical wave (measured: 5 700 instead of 4 400
// cycles per 16 columns); it solves its piece when the whole of L_kk is there.  (Pausing the partner for the whole step, updates included,
// makes the factorisation steps uniform -- 4 550 cycles -- but the paused waves then arrive late at their own steps: 63 500 cycles per
// block instead of 57 000.)
// (Round 3, measured with tools/dbg_potrf.py on the 56 100-cycle block and all bitwise neutral -- none kept:
//  * the pivot recurrence of factor16_tile kept uniform in vector registers (entry of row c + 1 and diagonal entry of column c + 1 read off the chain, no
//    lane read between two reciprocal square roots; tools/ub/factor16.hip "UNI"): 255 instead of 241 ticks per pivot -- the loop is bound by the number of
//    instructions a single wave issues, not by the latency of the chain;
//  * the update of the wave's own diagonal tile pipelined into the substitution (slice ks behind the published register ks, so that the wave the next
//    factorisation waits for is left with three rank-1 updates and ONE product behind the last publication): 55 800 cycles, the launch unchanged (1.240-1.249 ms);
//    with a branch between the last product of a matrix-core chain and the vector instruction that reads its result the compiler's wait-state count came out
//    short on the taken path and the results changed from run to run -- keep such code straight-line;
//  * the tiles of a block row updated two at a time with interleaved product chains: unchanged -- a tile update is 660 cycles of which the four products
//    are the smaller part (three tile loads, the wait for the operand piece, the store);
//  * all tile updates but the first deferred into the waits of the NEXT substitution (the bottom block rows run a whole step behind the factorisation --
//    wave 7 enters step k when factor16 of step k has finished -- and the last step starts 3 300 cycles after the seventh has ended): 64 200 cycles, the
//    deferred products then sit between the pivots of whichever wave shares the SIMD.)
// Outputs: the factor (lower triangle) to Aout, reciprocal pivots to rdiag[kglobal..], D to dvec (LDLT, nullable), and `pack`
// (nullable): the operand pack of k_trsm_panel -- 28 strictly-lower blocks of -L at j (j - 1) / 2 + k, then the 8 inverted
// diagonal pieces W_jj, all as column-major 16 x 16 blocks; `w16` (nullable): the same eight W_jj once more, into the array that holds them
// for ALL diagonal pieces of the factor (the sweeps of launch_trsv multiply by them).  rds: 8 x 32 doubles of LDS (pivots of every step).
// The pack goes to HBM AS THE STEPS COMPLETE (write-through stores): -L(w, k) by wave w right after its substitution of step k, W_kk and the
// pivots by wave k after its inversion.  cnt (nullable): eight agent-scope counters, cnt[k] += 1 per wave once its part of step k is
// in L2 (8 - k increments per call) -- the panel workgroups of the same launch (panel_follow) run step k of THEIR substitution when it
// reaches call-count x (8 - k).  The increments trail the stores by a step so that no wave waits for its own write-through.
constexpr int PACK_BLOCKS = 36;
constexpr int POTRF_RDS_DOUBLES = 8 * 32;
// block image (column-major 16 x 16) -> tile registers with the ROWS permuted: lane (ip, g) register r holds [pi(ip)][g + 4 r]
__device__ __forceinline__ d4 tile_load_rowperm(const double* __restrict__ blk, int lane)
{
    const double* p = blk + (lane >> 4) * 16 + pi16(lane & 15);
    return (d4){p[0], p[64], p[128], p[192]};
}
template <bool LDLT, int NWAVES>
__device__ __forceinline__ void potrf_block(double* __restrict__ Tb, double* __restrict__ rds, int nb, int kglobal, int* __restrict__ info, double* __restrict__ rdiag,
                            double* __restrict__ dvec, double* __restrict__ Aout, int lda, double* __restrict__ pack, double* __restrict__ w16,
                            long long* __restrict__ ts, int* __restrict__ cnt, int nactive,
                            const double* __restrict__ fetch_scratch, const int* __restrict__ fetch_flags, int fetch_token, const int* __restrict__ fetch_abort, bool pack_lkk)
{
    // pack_lkk: slot 28 + k of the pack takes the factored piece L_kk itself instead of its inverse (the consumers that solve by substitution:
    // front_trsm_follow)
    // nactive < 8 (short blocks of the sparse fronts: nb <= 16 nactive): block rows nactive .. 7 are identity padding and their waves sit out -- nothing
    // of theirs is stored (the pack blocks and inverted pieces of those rows stay unwritten: only a consumer that solves by substitution and masks the
    // padded columns, trsm_panel_body<SUBST>, may follow; the dense backend always passes 8)
    // ts (debugging aid, nullptr in production): shader-clock stamps of step k at ts[8 k + q] -- q = 0 / 1 wave k before / after its 16 x 16
    // factorisation, 2 / 3 wave k + 1 before / after its substitution, 4 / 5 wave k + 1 before / after its tile updates, 6 wave k after the inversion
    auto stamp = [&](int k, int q) { if (ts && (threadIdx.x & 63) == 0) ts[8 * k + q] = clock64(); };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    __shared__ int prog;      // columns of the block factored so far
    __shared__ int xdone[8];  // per wave: panel pieces solved and stored so far
    if (tid == 0) prog = 0;
    if (tid < 8) xdone[tid] = 0;
    __syncthreads();
    lds_vint* progp = (lds_vint*)&prog;
    lds_vint* xd = (lds_vint*)xdone;
    if (wave < nactive) {
        const int w = wave;
        double* Dww = Tb + tb_index(w, w) * 256;
        if (fetch_scratch) {
            // fused next diagonal block (fused_next_diag): the blocks the helper workgroups computed are pulled in HERE, by the wave whose block row they
            // belong to, when it starts -- waves 0 and 1 own theirs and factor at once, the later rows have until the chain reaches them (the
            // owner used to poll all eight helpers and copy 64 KB before anybody started: 4.5 us on the critical path of every panel).  Nobody reads a
            // tile of block row w before wave w has said so (xdone), so no barrier is needed.
            int seen_role = 0;
            bool bad = false;
            for (int b = max(tb_index(w, 0), FUSE_OWN); b <= tb_index(w, w); ++b) {
                const int role = b / FUSE_OWN;
                if (role != seen_role) {
                    unsigned spins = 0;
                    while (ldi_agent(fetch_flags + role) != fetch_token) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > 20000000u || (fetch_abort && (spins & 1023u) == 0 && ldi_agent(fetch_abort) != 0)) { bad = true; break; }
                    }
                    seen_role = role;
                }
                double v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = ld_agent(fetch_scratch + (size_t)b * 256 + lane + 64 * q);
#pragma unroll
                for (int q = 0; q < 4; ++q) Tb[b * 256 + lane + 64 * q] = v[q];
            }
            if (bad && lane == 0 && *info < 0) *info = kglobal;  // a helper never arrived (cannot happen with a healthy device): reported, not waited for
            wave_lds_sync();
        }
        d4 dperm = tile_load_perm(Dww, lane);  // own diagonal tile: in registers until it is factored
        int pending = -1;  // step whose pack stores this wave has not signalled yet
        auto signal_pending = [&]() {
            if (cnt && pending >= 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) addi_agent(cnt + pending, 1);
            }
            pending = -1;
        };
#pragma unroll 1
        for (int k = 0; k < w; ++k) {
            signal_pending();  // (step k - 1: stored a whole step ago)
            double* Dkk = Tb + tb_index(k, k) * 256;
            double* rbuf = rds + k * 32;  // reciprocal pivots [0, 16) and D [16, 32) of step k
            const double* dvs = rbuf + 16;
            double* Xwk = Tb + tb_index(w, k) * 256;
            if (w == k + 1) stamp(k, 2);
            d4 x = tile_load(Xwk, lane);
            if (w == k + 4) { while (*progp < 16 * k + 16) __builtin_amdgcn_s_sleep(8); }
            tile_trsm_rt_follow<LDLT>(x, Dkk, rbuf, &prog, 16 * k, lane);
            tile_store(Xwk, lane, x);
            asm volatile("" ::: "memory");
            if (lane == 0) xd[w] = k + 1;
            asm volatile("" ::: "memory");
            if (w == k + 1) { stamp(k, 3); stamp(k, 4); }
            const d4 nxw = {-x[0], -x[1], -x[2], -x[3]};
            if (pack) {
                double* pb = pack + (size_t)(w * (w - 1) / 2 + k) * 256 + g * 16 + i;
#pragma unroll
                for (int r = 0; r < 4; ++r) st_agent(pb + 64 * r, nxw[r]);
            }
            pending = k;
            d4 dsc = {1.0, 1.0, 1.0, 1.0};  // LDLT: D of the operand columns this lane feeds to the update products
            if (LDLT) dsc = (d4){dvs[g], dvs[g + 4], dvs[g + 8], dvs[g + 12]};
            // own diagonal tile first (for wave k + 1 it is the only one): operands = the stored X_w with permuted rows
            {
                const d4 xp = tile_load_rowperm(Xwk, lane);
                const d4 nxp = {-xp[0], -xp[1], -xp[2], -xp[3]};
                d4 xcp = xp;
                if (LDLT) xcp *= dsc;
                // one accumulator, k ascending: the summation order every parity test was pinned with (four independent products summed afterwards
                // save ~200 cycles per step and moved two threshold-sitting iteration counts by one: mm_QAFIRO 13 -> 12, mm_CONT-201 12 -> 13)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) dperm = __builtin_amdgcn_mfma_f64_16x16x4f64(xcp[ks], nxp[ks], dperm, 0, 0, 0);
            }
            for (int c = k + 1; c < w; ++c) {
                while (xd[c] < k + 1) __builtin_amdgcn_s_sleep(2);
                d4 xc = tile_load(Tb + tb_index(c, k) * 256, lane);
                if (LDLT) xc *= dsc;
                double* Tw = Tb + tb_index(w, c) * 256;
                d4 t = tile_load(Tw, lane);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) t = __builtin_amdgcn_mfma_f64_16x16x4f64(xc[ks], nxw[ks], t, 0, 0, 0);
                tile_store(Tw, lane, t);
            }
            if (w == k + 1) stamp(k, 5);
        }
        // the last wave lets the panel workgroups have its part of step 6 before it factors step 7 (nothing is left to hide the wait behind)
        if (w == 7) signal_pending();
        // ---- step w: the own diagonal piece ----
        {
            const int k = w;
            double* rbuf = rds + k * 32;
            const double* dvs = rbuf + 16;
            stamp(k, 0);
            __builtin_amdgcn_s_setprio(3);  // the critical wave
            const int failed = factor16_tile<LDLT>(dperm, lane, Dww, rbuf, &prog, 16 * k);  // (fills Dww, rbuf and, LDLT, dvs)
            __builtin_amdgcn_s_setprio(0);
            if (failed >= 0 && lane == 0 && 16 * k + failed < nb) { if (*info < 0) *info = kglobal + 16 * k + failed; }
            stamp(k, 1);
            asm volatile("" ::: "memory");
            // block row k is finished: pivots out, and the inverse of L_kk for the panel kernel / the sweeps -- nobody waits for it.
            // W^T = I L^-T by the same rank-1 substitution, then one product with the identity transposes it: W = I (W^T)^T.
            if (lane < 16 && 16 * k + lane < nb) {
                st_agent(rdiag + kglobal + 16 * k + lane, rbuf[lane]);
                if (LDLT && dvec) st_agent(dvec + 16 * k + lane, dvs[lane]);  // (written through: the next trailing update of a persistent launch reads it from other CUs)
            }
            if (pack && pack_lkk) {
                const d4 lkk = tile_load(Dww, lane);
                double* pb = pack + (size_t)(28 + k) * 256 + g * 16 + i;
#pragma unroll
                for (int r = 0; r < 4; ++r) st_agent(pb + 64 * r, lkk[r]);
            } else if (pack || w16) {
                const d4 lkk = tile_load(Dww, lane);  // the factored piece in the natural tile form
                d4 eye;
#pragma unroll
                for (int r = 0; r < 4; ++r) eye[r] = (i == g + 4 * r) ? 1.0 : 0.0;
                d4 z = eye;
                const d4 rdc = {rbuf[g], rbuf[g + 4], rbuf[g + 8], rbuf[g + 12]};
                tile_trsm_rt<LDLT, false>(z, lkk, rdc, lane);
                d4 wv = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) wv = __builtin_amdgcn_mfma_f64_16x16x4f64(z[ks], eye[ks], wv, 0, 0, 0);
                if (pack) {
                    double* pb = pack + (size_t)(28 + k) * 256 + g * 16 + i;
#pragma unroll
                    for (int r = 0; r < 4; ++r) st_agent(pb + 64 * r, wv[r]);
                }
                if (w16) {  // kept for the triangular sweeps (launch_trsv)
                    __attribute__((address_space(1))) double* wp = (__attribute__((address_space(1))) double*)(w16 + k * 256) + (lane >> 4) * 16 + (lane & 15);
                    wp[0] = wv[0]; wp[64] = wv[1]; wp[128] = wv[2]; wp[192] = wv[3];
                }
            }
            if (cnt) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) {
                    addi_agent(cnt + k, 1);
                    if (pending >= 0) addi_agent(cnt + pending, 1);
                }
                pending = -1;
            }
            // ... and block row k of the factor goes to HBM now (its blocks (k, 0..k) are final and only READ from here on), so that no
            // store tail is left at the end: 128-byte segments
            for (int bj = 0; bj <= k; ++bj) {
                const double* blk = Tb + tb_index(k, bj) * 256;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = lane + 64 * q, r = 16 * k + (e & 15), c = 16 * bj + (e >> 4);
                    const double v = blk[e];
                    if (r < nb && c < nb && r >= c) ((__attribute__((address_space(1))) double*)Aout)[(size_t)r + (size_t)c * lda] = v;  // (always device memory)
                }
            }
            stamp(k, 6);
        }
    }
    __syncthreads();
}

constexpr int POTRF_THREADS = 512;
template <bool LDLT>
__device__ __forceinline__ void potrf_diag_body(double* __restrict__ A, int lda, int nb, int kglobal, int* __restrict__ info, double* __restrict__ rdiag,
                                                double* __restrict__ dvec, double* __restrict__ pack, double* __restrict__ w16, long long* __restrict__ ts, int nactive = 8,
                                                int* __restrict__ cnt = nullptr, bool pack_lkk = false)
{
    extern __shared__ __attribute__((aligned(16))) double Tb[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int b = wave; b < TB_BLOCKS; b += POTRF_THREADS / 64) {
        int bi = 0;
        while ((bi + 1) * (bi + 2) / 2 <= b) ++bi;
        const int bj = b - bi * (bi + 1) / 2;
        double v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = lane + 64 * q, r = 16 * bi + (e & 15), c = 16 * bj + (e >> 4);
            const bool in = r < nb && c < nb;
            const bool ok = in && r >= c;
            const double* p = ok ? (A + r + (size_t)c * lda) : A;
            const double t = *p;
            v[q] = ok ? t : ((!in && r == c) ? 1.0 : 0.0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) Tb[b * 256 + lane + 64 * q] = v[q];
    }
    __syncthreads();
    potrf_block<LDLT, POTRF_THREADS / 64>(Tb, Tb + TB_DOUBLES, nb, kglobal, info, rdiag, dvec, A, lda, pack, w16, ts, cnt, nactive, nullptr, nullptr, 0, nullptr, pack_lkk);
}
template <bool LDLT>
__global__ __launch_bounds__(POTRF_THREADS) void k_potrf_diag(double* __restrict__ A, int lda, int nb, int kglobal, int* __restrict__ info, double* __restrict__ rdiag,
                                                              double* __restrict__ dvec, double* __restrict__ pack, double* __restrict__ w16, long long* __restrict__ ts)
{
    potrf_diag_body<LDLT>(A, lda, nb, kglobal, info, rdiag, dvec, pack, w16, ts);
}
// diagonal block of panel `panel` of every front of the list (LDLt; blockIdx.x = front)
__global__ __launch_bounds__(POTRF_THREADS) void k_potrf_diag_fronts(const FrontJob* __restrict__ jobs, int panel, int* __restrict__ info, double* __restrict__ rdiag)
{
    const FrontJob j = jobs[blockIdx.x];
    int k, nb, rs;
    if (j.kind != 0 || !front_panel(j, panel, k, nb, rs)) return;
    potrf_diag_body<true>(j.F + k + (size_t)k * j.f, j.f, nb, j.first + k, info, rdiag, j.dvec + k, rs > 0 ? j.pack : nullptr, nullptr, nullptr, min(8, (nb + 15) >> 4));
}

void launch_potrf_diag(bool ldlt, double* A, int lda, int nb, int kglobal, int* info, double* rdiag, double* dvec, double* pack, double* w16, hipStream_t s, long long* ts)
{
    static PerDeviceOnce attr_set;
    attr_set([&] {
        PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_diag<false>), hipFuncAttributeMaxDynamicSharedMemorySize, POTRF_LDS_BYTES));
        PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_diag<true>), hipFuncAttributeMaxDynamicSharedMemorySize, POTRF_LDS_BYTES));
    });
    if (ldlt) hipLaunchKernelGGL(k_potrf_diag<true>, dim3(1), dim3(POTRF_THREADS), POTRF_LDS_BYTES, s, A, lda, nb, kglobal, info, rdiag, dvec, pack, w16, ts);
    else hipLaunchKernelGGL(k_potrf_diag<false>, dim3(1), dim3(POTRF_THREADS), POTRF_LDS_BYTES, s, A, lda, nb, kglobal, info, rdiag, dvec, pack, w16, ts);
    PQ_HIP(hipGetLastError());
}

// Tile (0, 0) of a fused trailing update = the next diagonal block.  Only its lower block triangle is computed -- 36 tile blocks, one wave
// and one MFMA accumulator each --, the accumulators start from C, ONE operand panel is staged (rows 0..127 of the panel serve as row and,
// times -D, as column operand), and the result is factored in LDS (potrf_block).  This tile bounds the whole launch, and on one CU its update
// is pure matrix-core throughput (measured: 37 000 cycles for the 36 blocks), so it is spread over FUSE_ROLES = 9 workgroups on nine CUs, four
// blocks each (one product wave per SIMD); a helper writes its blocks through to a.fuse_scratch (sc1 stores), drains, and publishes
// a.fuse_flags[role] = a.fuse_token (unique per launch: no reset between launches); the owner polls the eight flags (bounded) and pulls the
// blocks into its LDS with L1-bypassing loads -- the write-through recipe of MI355X_MICROARCH.md (no release / acquire fences).
// K <= 128 (one panel): the WHOLE operand panel goes to LDS in one round of loads -- the double-buffered 16-column stages of the generic
// kernel cost a load -> LDS -> barrier round trip each (measured 2700 cycles per stage) -- and the 32 MFMA k-slices run back to back on four
// interleaved accumulator chains (a single chain is a dependent MFMA every ~180 cycles plus the LDS read in front of it: 18 000 cycles measured).
template <int NT, bool PERSIST>
__device__ __forceinline__ bool fused_next_diag(const SyrkArgs& a, double* __restrict__ smem, int role)
{
    constexpr int NW = NT / 64;
    static_assert(FUSE_OWN * FUSE_ROLES >= TB_BLOCKS && FUSE_OWN <= NW, "every tile block needs a wave");
    double* As = smem;                    // [kdim <= 128][LDS_LD]: the whole operand panel
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const bool dbg_ts = FUSE_TS_ON && !PERSIST && a.fuse_ts && tid == 0 && role == 0;
    if (dbg_ts) a.fuse_ts[0] = clock64();
    const int nbn = a.fuse_nb;
    const bool edge = !PERSIST && ((TS > a.n) || a.unaligned);
    // the tile block of this wave
    const int t = role * FUSE_OWN + wave;  // four blocks per workgroup: one wave per SIMD does the products, the other waves only help staging
    const bool on = wave < FUSE_OWN && t < TB_BLOCKS;
    int bi = 0;
    {
        const int tt = on ? t : 0;
        while ((bi + 1) * (bi + 2) / 2 <= tt) ++bi;
    }
    const int bj = (on ? t : 0) - bi * (bi + 1) / 2;
    const int nkt = (a.kdim + BK - 1) / BK;  // <= 8
    double* ws = smem + TS * LDS_LD;         // -D of the panel (LLT: -1), after the operand panel
    d4 acc;
    d4 p1 = {0.0, 0.0, 0.0, 0.0}, p2 = p1, p3 = p1;
    bool progressive = PERSIST && a.fuse_xcnt != nullptr;
    if (progressive) {
        // (a crew that arrives when the whole row is already published -- its own tile came late -- takes the one-shot path below: one round of loads
        // instead of eight; same products, same order)
        __shared__ int all_s;
        if (tid == 0) all_s = (ldi_agent(a.fuse_xcnt + 7) - a.fuse_xwant >= 0) ? 1 : 0;
        __syncthreads();
        if (all_s) progressive = false;
    }
    if (progressive) {
        // Persistent launch, rounds >= 1: the operand (block row k + 1 of panel k) is being solved RIGHT NOW by the first panel workgroup of the round
        // before, behind the factorisation of that round's diagonal block; it publishes every 16-column slice as it becomes final (panel_follow: write-through
        // stores, then fuse_xpub += 1 per wave).  The crew takes each slice as it arrives -- its four k-steps go to the four accumulator chains exactly as
        // in the one-shot loop below, in the same order: bitwise the same block -- so that only the LAST slice's products, not the operand fetch and all 32,
        // stand between the end of one diagonal block and the start of the next.
        const int li = bi * 16 + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int lj = bj * 16 + g + 4 * r;
            const bool ok = on && li >= lj;
            const double cv = ld_agent(a.C + (ok ? (size_t)li + (size_t)lj * a.ldc : 0));
            acc[r] = ok ? cv : 0.0;
        }
        __shared__ int pok_s;
        const double* Ar = As + bi * 16 + i;
        const double* Ac = As + bj * 16 + i;
#pragma unroll 1
        for (int kt = 0; kt < 8; ++kt) {
            if (tid == 0) {
                int ok = 1;
                unsigned spins = 0;
                while (ldi_agent(a.fuse_xcnt + kt) - a.fuse_xwant < 0) {  // (one counter per slice: the strip waves of the row run independently of each other)
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > 20000000u || ((spins & 1023u) == 0 && a.fuse_abort && ldi_agent(a.fuse_abort) != 0)) { ok = 0; break; }
                }
                pok_s = ok;
                if (a.fuse_tr2n && role == 0) a.fuse_tr2n[8 + kt] = wall_clock64();
            }
            __syncthreads();
            if (!pok_s) {
                if (tid == 0 && *a.fuse_info < 0) *a.fuse_info = a.fuse_kglobal;
                return false;
            }
            d2 va[1024 / NT];
            load_tile_agent<NT>(a.A, a.lda, 0, kt * BK, tid, va);
            if (tid < BK) ws[kt * BK + tid] = -(a.w ? ld_agent(a.w + kt * BK + tid) : 1.0);
            store_tile<NT>(As + kt * BK * LDS_LD, tid, va);
            __syncthreads();
            if (on) {
                double af[4], bf[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kk = (4 * kt + u) * 4 + g;
                    af[u] = Ar[kk * LDS_LD];
                    bf[u] = Ac[kk * LDS_LD] * ws[kk];
                }
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[0], af[0], acc, 0, 0, 0);
                p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[1], af[1], p1, 0, 0, 0);
                p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[2], af[2], p2, 0, 0, 0);
                p3 = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[3], af[3], p3, 0, 0, 0);
            }
        }
        if (on) acc = ((acc + p1) + p2) + p3;  // fixed order
        if (a.fuse_tr2n && role == 0 && tid == 0) a.fuse_tr2n[16] = wall_clock64();
    } else {
    {
        d2 va[8][1024 / NT];
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
            if (kt < nkt) {
                const int k0 = kt * BK;
                if constexpr (PERSIST && AGENT_OPERANDS) load_tile_agent<NT>(a.A, a.lda, 0, k0, tid, va[kt]);
                else if (edge || k0 + BK > a.kdim) load_tile<true, NT>(a.A, a.lda, 0, k0, a.n, a.kdim, tid, va[kt]);
                else load_tile<false, NT>(a.A, a.lda, 0, k0, a.n, a.kdim, tid, va[kt]);
            }
        }
        const double wv = (tid < TS) ? -((a.w && tid < a.kdim) ? (PERSIST ? ld_agent(a.w + tid) : a.w[tid]) : 1.0) : 0.0;  // the sign of C - A D A^T rides on D
        // C of this wave's block (identity padding beyond the order of a last, partial panel)
        const int li = bi * 16 + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int lj = bj * 16 + g + 4 * r;
            const bool in = li < nbn && lj < nbn;
            const bool ok = on && in && li >= lj;
            const double* cp = a.C + (ok ? (size_t)li + (size_t)lj * a.ldc : 0);
            const double cv = PERSIST ? ld_agent(cp) : *cp;
            acc[r] = ok ? cv : ((!in && li == lj) ? 1.0 : 0.0);
        }
        if (tid < TS) ws[tid] = wv;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
            if (kt < nkt) store_tile<NT>(As + kt * BK * LDS_LD, tid, va[kt]);
    }
    __syncthreads();
    if (dbg_ts) a.fuse_ts[4] = clock64();
    if (on) {
        const double* Ar = As + bi * 16 + i;
        const double* Ac = As + bj * 16 + i;
#pragma unroll 2
        for (int ks = 0; ks < 4 * nkt; ks += 4) {
            double af[4], bf[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kk = (ks + u) * 4 + g;
                af[u] = Ar[kk * LDS_LD];
                bf[u] = Ac[kk * LDS_LD] * ws[kk];
            }
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[0], af[0], acc, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[1], af[1], p1, 0, 0, 0);
            p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[2], af[2], p2, 0, 0, 0);
            p3 = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[3], af[3], p3, 0, 0, 0);
        }
        acc = ((acc + p1) + p2) + p3;  // fixed order
    }
    }
    __syncthreads();  // every wave is done with the operand panel (the owner reuses its LDS for the tile blocks)
    if (role > 0) {
        // helper: block -> scratch, written through; drained; then the token
        if (on) {
            double* blk = a.fuse_scratch + (size_t)t * 256 + g * 16 + i;
#pragma unroll
            for (int r = 0; r < 4; ++r) st_agent(blk + 64 * r, acc[r]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) sti_agent(a.fuse_flags + role, a.fuse_token);
        if (PERSIST && a.fuse_tr2n && tid == 0 && role == 1) a.fuse_tr2n[19] = wall_clock64();
        return true;
    }
    if (dbg_ts) a.fuse_ts[1] = clock64();
    double* Tb = smem;  // the operand panel is dead now
    if (on) tile_store(Tb + t * 256, lane, acc);
    // (the helpers' blocks are pulled inside potrf_block, row by row, by the waves that need them)
    __syncthreads();
    if (dbg_ts) a.fuse_ts[2] = clock64();
    if (PERSIST && a.fuse_tr2n && tid == 0) a.fuse_tr2n[17] = wall_clock64();
    long long* pts = (FUSE_TS_ON && !PERSIST && a.fuse_ts) ? a.fuse_ts + 8 : nullptr;
    if (a.fuse_ldlt) potrf_block<true, NW>(Tb, Tb + TB_DOUBLES, nbn, a.fuse_kglobal, a.fuse_info, a.fuse_rdiag, a.fuse_dvec, a.C, a.ldc, a.fuse_pack, a.fuse_w16, pts, a.fuse_cnt, 8,
                                           a.fuse_scratch, a.fuse_flags, a.fuse_token, PERSIST ? a.fuse_abort : nullptr);
    else potrf_block<false, NW>(Tb, Tb + TB_DOUBLES, nbn, a.fuse_kglobal, a.fuse_info, a.fuse_rdiag, a.fuse_dvec, a.C, a.ldc, a.fuse_pack, a.fuse_w16, pts, a.fuse_cnt, 8,
                                a.fuse_scratch, a.fuse_flags, a.fuse_token, PERSIST ? a.fuse_abort : nullptr);
    if (dbg_ts) a.fuse_ts[3] = clock64();
    if (PERSIST && a.fuse_tr2n && tid == 0) a.fuse_tr2n[18] = wall_clock64();
    return true;
}

// The panel solve of the NEXT panel inside the fused trailing update: the workgroup of tile (ti, 0) has just updated the 128 rows of that panel
// it owns.  Instead of writing them out for a k_trsm_panel launch it keeps them -- tiles through LDS into the row-strip form of k_trsm_panel, one
// wave per 16 rows, eight 16 x 16 tiles in registers -- and runs the block substitution  X_k = T_k W_kk^T,  T_j -= X_k L_jk^T (j > k)  step by
// step BEHIND the factorisation of the diagonal block, which workgroup 0 of this launch publishes step by step (potrf_block: pack + cnt).  The
// panel is finished about one step after the diagonal block instead of one launch later (k_trsm_panel: 9.6 us + two launch boundaries).  Same
// products in the same order as k_trsm_panel: bitwise the same panel.  Waits only target workgroups with lower block indices; bounded spins.
template <int NT, int MTC, int MTR, bool PERSIST, int NSTRIP>
__device__ __forceinline__ bool panel_follow(const SyrkArgs& a, double* __restrict__ smem, const d4 (&acc)[MTC][MTR], int row0, int wr, int wc, bool first_row)
{
    static_assert(NT == 512 && (NSTRIP == 8 || NSTRIP == 4), "eight waves; one 16-row strip per wave (NSTRIP = 4: a half tile, waves 4 .. 7 only help fetching)");
    const bool strip = (threadIdx.x >> 6) < NSTRIP;  // this wave holds a strip
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    double* Ts = smem;             // the tile as 8 x 8 block images (row block, column block)
#pragma unroll
    for (int x = 0; x < MTC; ++x)
#pragma unroll
        for (int y = 0; y < MTR; ++y) tile_store(Ts + ((wr * MTR + y) * 8 + (wc * MTC + x)) * 256, lane, acc[x][y]);
    __syncthreads();
    d4 T[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) T[j] = strip ? tile_load(Ts + (wave * 8 + j) * 256, lane) : (d4){0.0, 0.0, 0.0, 0.0};
    // persistent launch: the first block row below the diagonal block is the operand of the NEXT diagonal block's update -- its slices go out as they
    // become final (see fused_next_diag); every wave counts in fuse_xpub once per slice
    const bool publish = PERSIST && a.fuse_xpub != nullptr && first_row;
    const bool dbg = FUSE_TS_ON && !PERSIST && a.fuse_ts && tid == 0 && row0 == TS;  // debugging aid: stamps of the first panel workgroup at fuse_ts[72..]
    if (dbg) a.fuse_ts[80] = clock64();
    // The operand blocks of a step are fetched when the factorisation has published them -- and with them those of every FURTHER step it has published by
    // then, in one round of loads (all of a thread's loads in flight together): a workgroup that starts behind the factorisation (persistent launch: its own
    // inputs come from the round before) catches up at the cost of its products instead of paying an L2 / HBM round trip and two barriers per step (3.7 us
    // per early step, measured, against the 3.4 us the factorisation needs for one).  The blocks live at their pack index in the LDS the tile just left.
    __syncthreads();  // every wave holds its strip in registers: Ts is free
    double* Pk = Ts;
    double* rds = Ts + PACK_BLOCKS * 256;  // [128] reciprocal pivots of the published steps (LDLt, published rows only)
    __shared__ int have_s, nblk_s, blist[PACK_BLOCKS];
    int have = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (k >= have) {
            if (tid == 0) {
                int ok = 1, nxt = k;
                unsigned spins = 0;
                while (ldi_agent(a.fuse_cnt + k) - a.fuse_token * (8 - k) < 0) {
                    __builtin_amdgcn_s_sleep(4);
                    if (++spins > 20000000u || (PERSIST && a.fuse_abort && (spins & 1023u) == 0 && ldi_agent(a.fuse_abort) != 0)) { ok = 0; break; }
                }
                if (ok) { nxt = k + 1; while (nxt < 8 && ldi_agent(a.fuse_cnt + nxt) - a.fuse_token * (8 - nxt) >= 0) ++nxt; }
                int nb = 0;
                for (int q = k; q < nxt; ++q) { blist[nb++] = 28 + q; for (int j = q + 1; j < 8; ++j) blist[nb++] = j * (j - 1) / 2 + q; }
                have_s = ok ? nxt : -1; nblk_s = nb;
                if (PERSIST && a.fuse_tr2 && first_row && row0 == TS) for (int q = k; q < nxt; ++q) a.fuse_tr2[32 + q] = wall_clock64();
                if (dbg) a.fuse_ts[72 + k] = clock64();
            }
            __syncthreads();
            if (have_s < 0) {  // the diagonal block never arrived (cannot happen with a healthy device): report instead of hanging
                if (tid == 0 && *a.fuse_info < 0) *a.fuse_info = a.fuse_kglobal;
                return false;
            }
            // (LDLt, published rows: the reciprocal pivots of the newly published steps go to LDS with the blocks -- a load inside the step loop would have the
            // loop wait for its own write-through stores, which share the counter)
            if (PERSIST && a.fuse_ldlt && tid < 128 && tid >= 16 * k && tid < 16 * have_s) rds[tid] = ld_agent(a.fuse_rdiag + a.fuse_kglobal + tid);
            have = have_s;
            const int nb = nblk_s;
            if (nb <= 8) {
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int e = u * NT + tid, bi = e >> 8; v[u] = ld_agent(a.fuse_pack + (size_t)blist[bi < nb ? bi : 0] * 256 + (e & 255)); }
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int e = u * NT + tid, bi = e >> 8; if (bi < nb) Pk[blist[bi] * 256 + (e & 255)] = v[u]; }
            } else {
                double v[PACK_BLOCKS * 256 / NT];
#pragma unroll
                for (int u = 0; u < PACK_BLOCKS * 256 / NT; ++u) { const int e = u * NT + tid, bi = e >> 8; v[u] = ld_agent(a.fuse_pack + (size_t)blist[bi < nb ? bi : 0] * 256 + (e & 255)); }
#pragma unroll
                for (int u = 0; u < PACK_BLOCKS * 256 / NT; ++u) { const int e = u * NT + tid, bi = e >> 8; if (bi < nb) Pk[blist[bi] * 256 + (e & 255)] = v[u]; }
            }
            __syncthreads();
        }
        if (!strip) continue;
        const d4 w = tile_load(Pk + (28 + k) * 256, lane);
        d4 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) x = __builtin_amdgcn_mfma_f64_16x16x4f64(w[ks], T[k][ks], x, 0, 0, 0);
        T[k] = x;
        if (PERSIST) {
            // (every row of the persistent launch sends its slices as they become final -- a row that stored its whole strip at the end spent 3-5 us draining
            // 64 write-through stores per lane before it could report, and the next round's first row waits for exactly that report; only the FIRST row
            // announces its slices, to the next crew)
            // This slice goes out now; the slice of TWO steps ago is announced.  A write-through store is acknowledged ~2 us after its issue and a step is
            // ~0.5 us of products: draining before every announcement made the acknowledgement the length of a step (8 x 2.7 us behind a finished
            // diagonal block -- the longest stretch of a late round, longer than the diagonal block itself).  Vector-memory operations leave the counter in
            // issue order, so "at most 16 outstanding" = everything older than the two youngest slices (8 stores each) has arrived.
            double* Cr = a.C + (row0 + wave * 16 + i);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * k + g + 4 * r;
                double v = x[r];
                if (a.fuse_ldlt) v *= rds[c];
                st_agent(Cr + (size_t)c * a.ldc, v);
                st_agent(a.fuse_side + (row0 + wave * 16 + i) + (size_t)c * a.ldc, v);
            }
            if (publish && k >= 2) { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); if (lane == 0) addi_agent(a.fuse_xpub + (k - 2), 1); }
            if (a.fuse_tr2 && tid == 0 && row0 == TS) a.fuse_tr2[k] = wall_clock64();
        }
#pragma unroll
        for (int j = k + 1; j < 8; ++j) {
            const d4 nl = tile_load(Pk + (j * (j - 1) / 2 + k) * 256, lane);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) T[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(nl[ks], x[ks], T[j], 0, 0, 0);
        }
    }
    if (PERSIST && !publish) return true;  // (the caller drains the last slices before it reports the row)
    if (publish) {
        if (!strip) return true;
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        if (lane == 0) addi_agent(a.fuse_xpub + 6, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) addi_agent(a.fuse_xpub + 7, 1);
        if (a.fuse_tr2 && tid == 0 && row0 == TS) a.fuse_tr2[27] = wall_clock64();
        if (dbg) a.fuse_ts[81] = clock64();
        return true;
    }
    const int row = row0 + wave * 16 + i;
    if (strip && row < a.n) {
        double* Cr = a.C + row;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * j + g + 4 * r;
                if (c < a.fuse_nb) {
                    double v = T[j][r];
                    if (a.fuse_ldlt) v *= ld_agent(a.fuse_rdiag + a.fuse_kglobal + c);
                    if (PERSIST) { st_agent(Cr + (size_t)c * a.ldc, v); st_agent(a.fuse_side + row + (size_t)c * a.ldc, v); }
                    else Cr[(size_t)c * a.ldc] = v;
                }
            }
    }
    if (dbg) a.fuse_ts[81] = clock64();
    return true;
}

// ------------------------------------------------------------------------------------------------
// The whole blocked factorisation after its first diagonal block and panel in ONE persistent launch (round 3).
//
// Round k = what the fused launch of panel k did: trailing update U_k of every tile, factorisation of the next diagonal block D_{k+1}
// (owner + eight helpers), substitution of the next panel behind it.  Here the rounds are not separated by kernel boundaries: every piece of
// work is a TASK of a fixed, topologically ordered list, workgroups draw tickets from one counter and wait -- bounded -- only for results of
// EARLIER tickets, so progress never depends on how many workgroups are resident (two factorisations on two streams, a busy device).  The
// order puts the next round's critical tasks right behind what they need:
//   crew(0) panel(0) bulk(0, col 1) | crew(1) panel(1) bulk(0, cols >= 2) bulk(1, col 1) | crew(2) panel(2) bulk(1, cols >= 2) bulk(2, col 1) | ...
// so the diagonal-block chain of round k + 1 starts as soon as ITS inputs exist (look-ahead) while the bulk tiles of round k are still running.  A
// ticket is drawn only when its `gate` is open -- that many rounds have all their panel tasks finished, one word every idle workgroup polls with a
// long sleep --: crew and panel tasks are let in a round early (the crew follows the slices of its operand as they are published, a panel task
// waits for its own two rows), everything else when the panel it reads is complete.
// What crosses workgroups inside the launch is written through (sc1) and read with agent-scope loads, and published by one word per unit:
//   lready[k T + i]  panel tasks of block row i of panel k that have finished (split(k - 1) per row; panel 0 is solved before the launch);
//   tver[i T + j]    number of trailing updates tile (i, j) has received;
//   pdone[k]         panel tasks of round k that have finished (the owner of round k + 2 reuses their operand-pack buffer);
// as launch-unique or cumulative values, so nothing is reset between factorisations.  Every tile receives the same products in the same
// order as in the launch-per-panel path: bitwise the same factor (tools/chk_chol_persistent.py, tests/test_dense_gpu.py).
struct CholTask {
    short kind, round, a, b;  // kind 0: helper (role a) / 1: owner / 2: panel tile (a, 0), half b (-1: the whole tile) / 3: bulk tile (a, b) of round `round`
                              // (4 / 5: the halves of the diagonal tile of the first bulk column); fused assembly only -- 6: K-slice `round` of the assembly of the
                              // ABSOLUTE tile (a, b); 7: the updates of panels [round, gate + 1) on the absolute tile (a, b) in one visit
    int gate;                 // panel rounds that must be complete before the ticket is drawn
    int aux;                  // kind 6: first 128 x 128 slot of the tile's partial sums in CholArgs::part
};
// K-slices of an assembly tile of block column j (mchunks = m / 128 operand chunks): the early columns are wanted first and are cut finest, every slice a
// task of its own whose partial sum goes to a slot of `part`; the last slice to finish adds them up in slice order (chol_role_reduce).  One slice: no partial.
__host__ __device__ inline int chol_asm_slices(int j, int mchunks)
{
    // block columns 1-2 (the chain starts on them): 8 slices; 3-8: 4; 9-16: 2; beyond: 1 (a task costs ~20 us besides its chunks)
    const int s = j <= 2 ? 8 : (j <= 8 ? 4 : (j <= 16 ? 2 : 1));
    return s < mchunks ? s : (mchunks > 0 ? mchunks : 1);
}
// (the threshold of chol_bulk_bound: rounds with at least that many trailing tile rows keep whole panel tiles.  22 was the best of a sweep in round 3; PIQP_AMD_DEBUG=chol_whole=<rows>
// overrides it -- one value per process: the cumulative counters of a handle assume the same split in every factorisation)
constexpr int CHOL_WHOLE_ROWS = 22;
static int chol_split_thr()
{
    static const int v = [] { const char* e = debug_token("chol_whole"); return e ? std::atoi(e) : CHOL_WHOLE_ROWS; }();
    return v;
}
// bands of block columns with one slice count, assembled slice-major (see chol_build_tasks_fused): [lo, hi)
__host__ __device__ inline int chol_asm_band_end(int lo) { return lo <= 2 ? 3 : (lo <= 8 ? 9 : lo + 8); }
// workgroups per panel tile: a row's task of round k needs that row's output of round k - 1, so one workgroup's update + substitution (28 + 13 us for a
// whole tile) bound a round from below; halves (fused_tile<HALF>) keep that under the diagonal block's 30 us.  Early rounds are bound by the bulk
// tiles anyway and keep whole tiles (half as many workgroups parked on the chain).
// A round with >= 22 trailing tile rows (> 230 bulk tiles) is bound by the bulk tiles, not by the chain: there whole tiles park half as many workgroups on
// the chain and leave them to the bulk.
// (threshold swept at T = 32, factorisation ms: never 1.28-1.32, 30: 1.28, 26: 1.26, 22: 1.245, 18: 1.27, 14: 1.29, 10: 1.32)
__host__ __device__ inline bool chol_bulk_bound(int T, int k, int thr) { return T - k - 1 >= thr; }
__host__ __device__ inline int chol_split(int T, int k, int thr) { return chol_bulk_bound(T, k, thr) ? 1 : 2; }
// ... except the first CHOL_FAST_ROWS block rows below the diagonal block, which are always halves: the first row's slices feed the next crew and the second
// row's completion starts the next round's first row, and a whole tile's update (31 us) arrives at the diagonal block when that is finished (the rounds with
// whole first rows ran at 48 us, the others at 36)
constexpr int CHOL_FAST_ROWS = 2;
__host__ __device__ inline int chol_split_row(int T, int k, int ti, int thr) { return ti <= CHOL_FAST_ROWS ? 2 : chol_split(T, k, thr); }
__host__ __device__ inline int chol_panel_tasks(int T, int k, int thr)  // panel tasks of round k (rows ti = 1 .. T - k - 2)
{
    const int rows = T - k - 2, fast = rows < CHOL_FAST_ROWS ? rows : CHOL_FAST_ROWS;
    return rows <= 0 ? 0 : 2 * fast + chol_split(T, k, thr) * (rows - fast);
}
struct CholArgs {
    double* A; double* side; int lda, n, T, ldlt;
    int* info; double* rdiag; double* dvec; double* pack2; double* w16;
    double* scratch; int* fuse_flags; int* fuse_cnt; int token_base;
    const CholTask* tasks; int ntasks;
    int* ticket;   // [0] next ticket, [1] abort
    int* lready; int* tver; int* pdone;
    int* xcnt;            // per round and slice: strip waves of the first panel row that have published the slice, cumulative over the factorisations of this handle
    int* progress;        // gen + number of rounds whose panel tasks have ALL finished (they finish in round order)
    int* dhalf;           // per round: halves of the split diagonal tile of the first bulk column that have finished (cumulative over the factorisations)
    int gen;       // launch-unique base of the flag values
    int fcount;    // persistent factorisations this handle has run before this one (pdone counters are cumulative)
    int split_thr; // chol_split_thr(): rounds with at least this many trailing tile rows keep whole panel tiles
    // fused assembly (dense/kkt.hpp:140-160 as tasks of this launch, GT != nullptr): K = Pfull + diag(x_reg) + dinv ATA + GT diag(zinv) GT^T for the tiles of the
    // block columns >= 1 (column 0 is assembled, factored and solved before the launch); m % 128 == 0
    const double* GT; int ldg, m; const double* zinv;
    const double* Pfull; int ldp; const double* x_reg; const double* ATA; int ldata; double dinv;
    double* part;  // partial sums of the K-sliced tiles, 128 x 128 column-major slots
    int* acnt;     // per tile: slices that have stored their partial sum, cumulative over the factorisations
    // The assembly tasks (kind 6) are NOT in the ticket list: they sit in eight queues, tile (i, j) in queue i mod 8, and the list holds one TOKEN (kind 8) per
    // task.  A workgroup that draws a token takes the next task of the queue of ITS XCD (workgroups go to the XCDs round-robin), so the ~32 tiles an XCD works on at
    // a time come from four block rows and a few block columns and share their operand panels in that XCD's L2 -- drawn from one list in ticket order every
    // tile's 1 MB of operands per slice came from the fabric (25 us per 128-column chunk instead of the 15 of a tile whose operands are in L2); an empty queue
    // steals from the next one.  A task that finds its tile not assembled yet runs assembly tasks itself until it is (help_assembled): whatever the queues'
    // progress, nobody ever sleeps on a tile whose assembly has not been drawn.
    const CholTask* aq; int aq_ptr[9]; int* aq_head;
    long long* trace;  // debugging aid (PIQP_AMD_DEBUG=chol_trace), nullable: per ticket 4 x wall_clock64 (100 MHz): drawn, inputs ready, done; [3] = workgroup id
};

// waits until *p - want >= 0 for up to three words; false on abort / timeout (sets the abort word).  Threads 0 .. 2 take one word each: the words come from L2
// (agent-scope loads, 0.7-1 us apiece under load) and one thread asking for them in turn put three of those round trips in front of every task.
__device__ __forceinline__ bool chol_wait3(const int* p0, int w0, const int* p1, int w1, const int* p2, int w2, int* abort_w, int sleep)
{
    __shared__ int ok_s[3];
    if (threadIdx.x < 3) {
        const int* p = threadIdx.x == 0 ? p0 : (threadIdx.x == 1 ? p1 : p2);
        const int w = threadIdx.x == 0 ? w0 : (threadIdx.x == 1 ? w1 : w2);
        int ok = 1;
        if (p) {
            unsigned spins = 0;
            while (ldi_agent(p) - w < 0) {
                __builtin_amdgcn_s_sleep(4);
                if (sleep) __builtin_amdgcn_s_sleep(12);
                ++spins;
                if ((spins & 255u) == 0 && ldi_agent(abort_w) != 0) { ok = 0; break; }
                if (spins > 8000000u) { ok = 0; sti_agent(abort_w, 1); break; }
            }
        }
        ok_s[threadIdx.x] = ok;
    }
    __syncthreads();
    const bool ok = (ok_s[0] & ok_s[1] & ok_s[2]) != 0;
    __syncthreads();
    return ok;
}

// ---- fused assembly / multi-panel updates (round 4): ONE accumulate routine for both.  A visit adds `nchunk` operand chunks of 128 columns to a 128 x 128 tile:
//   FAR = false  assembly, chunk c = columns [128 (c0 + c), +128) of GT, column operand scaled by zinv: acc += G_i diag(zinv) G_j^T, accumulators from zero;
//                the epilogue stores the raw partial sum (part != nullptr) or C = base + acc (dense/kkt.hpp:144-158, the arithmetic of EPI_ASSEMBLE)
//   FAR = true   trailing updates of panels [c0, c0 + nchunk): accumulators from C, column operand negated (and scaled by D for LDLt): the k-slices of the
//                panels in ascending order, i.e. exactly the products the one-panel visits add -- bitwise the same tile, with one C fetch and one store
// Operand stages: four (64 columns) per group through LDS, the next group's loads issued behind the LDS stores of the current one, so they fly during its products.
struct AccumArgs {
    const double* A0; const double* A1;  // row operand, first column of the visit: chunk 0 reads A0, chunks >= 1 read A1 (panel 0 lives in place, the others in the side copy)
    const double* B0; const double* B1;  // column operand, likewise
    int ld; int nchunk;
    const double* w;                     // per-column scale of the column operand, indexed from the visit's first column (nullable)
    double* C; int ldc; int diag;        // the tile; diag: on the block diagonal (sub-tiles above it are skipped)
    double* part;                        // !FAR: raw partial sum instead of the epilogue
    const double* Pt; int ldp; const double* xr; const double* At; int ldat; double dinv;  // !FAR epilogue: tile origins inside Pfull / ATA, x_reg at the tile's first row
};
template <int NT, bool FAR>
__device__ __forceinline__ void accum_tile(const AccumArgs& g, double* __restrict__ smem)
{
    constexpr int WR = 4, WC = 2, SUBR = 32, SUBC = 64, MTR = 2, MTC = 4, PER = 1024 / NT, GRP = 4;
    static_assert(NT == 512, "eight waves");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const bool skip_wave = g.diag && ((wr + 1) * SUBR <= wc * SUBC);
    d4 acc[MTC][MTR];
#pragma unroll
    for (int x = 0; x < MTC; ++x)
#pragma unroll
        for (int y = 0; y < MTR; ++y) acc[x][y] = (d4){0.0, 0.0, 0.0, 0.0};
    if (FAR && !skip_wave) {
#pragma unroll
        for (int x = 0; x < MTC; ++x)
#pragma unroll
            for (int y = 0; y < MTR; ++y) {
                const int li = wr * SUBR + y * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lj = wc * SUBC + x * 16 + (lane >> 4) + 4 * r;
                    const bool ok = !g.diag || li >= lj;
                    const double cv = ld_agent(g.C + (ok ? (size_t)li + (size_t)lj * g.ldc : 0));
                    acc[x][y][r] = ok ? cv : 0.0;
                }
            }
    }
    // Software pipeline over groups of TWO stages (32 operand columns): LDS holds two groups (the one being multiplied and the next one), the registers two more
    // (groups n + 2 and n + 3 in flight from memory) -- one barrier per group, and the LDS stores of group n + 1 and the loads of group n + 3 are issued in front of
    // the products of group n instead of between two barriers.  (History: a whole chunk in registers, requested behind the second group's LDS stores, spilled --
    // 220-240 bytes of scratch per lane, 26-31 us per 128-column chunk; four-stage groups through ONE LDS buffer with two barriers each: 21 us; the matrix cores
    // need 13.7 us.)
    constexpr int GS = 2;                                   // stages per group
    constexpr int GROUP_DOUBLES = 2 * GS * BK * LDS_LD;     // A stages, then B stages
    static_assert(2 * GROUP_DOUBLES * (int)sizeof(double) <= FUSED_LDS_BYTES, "two groups in LDS");
    d2 ra[2][GS][PER], rb[2][GS][PER];
    auto request = [&](int gi, d2 (&pa)[GS][PER], d2 (&pb)[GS][PER]) {  // group gi = stages [GS (gi & 3), +GS) of chunk gi >> 2
        const int c = gi >> 2, k0 = (gi & 3) * GS * BK;
        const double* A = (c == 0 ? g.A0 : g.A1) + (size_t)c * 128 * g.ld;
        const double* B = (c == 0 ? g.B0 : g.B1) + (size_t)c * 128 * g.ld;
#pragma unroll
        for (int q = 0; q < GS; ++q) { load_tile<false, NT>(A, g.ld, 0, k0 + q * BK, 0, 0, tid, pa[q]); load_tile<false, NT>(B, g.ld, 0, k0 + q * BK, 0, 0, tid, pb[q]); }
    };
    auto to_lds = [&](int gi, d2 (&pa)[GS][PER], d2 (&pb)[GS][PER]) {
        double* buf = smem + (gi & 1) * GROUP_DOUBLES;
        const double* wg = g.w ? g.w + (size_t)gi * GS * BK : nullptr;
#pragma unroll
        for (int q = 0; q < GS; ++q) {
            scale_tile<false, NT, FAR, FAR>(wg, q * BK, 0, tid, pb[q]);
            store_tile<NT>(buf + q * BK * LDS_LD, tid, pa[q]);
            store_tile<NT>(buf + (GS + q) * BK * LDS_LD, tid, pb[q]);
        }
    };
    auto products = [&](int gi) {
        if (skip_wave) return;
        const double* buf = smem + (gi & 1) * GROUP_DOUBLES;
#pragma unroll
        for (int q = 0; q < GS; ++q) {
            const double* Asb = buf + q * BK * LDS_LD + wr * SUBR + (lane & 15);
            const double* Bsb = buf + (GS + q) * BK * LDS_LD + wc * SUBC + (lane & 15);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int kk = ks * 4 + (lane >> 4);
                double af[MTR], bf[MTC];
#pragma unroll
                for (int u = 0; u < MTR; ++u) af[u] = Asb[kk * LDS_LD + u * 16];
#pragma unroll
                for (int u = 0; u < MTC; ++u) bf[u] = Bsb[kk * LDS_LD + u * 16];
#pragma unroll
                for (int x = 0; x < MTC; ++x)
#pragma unroll
                    for (int y = 0; y < MTR; ++y)
                        acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[x], af[y], acc[x][y], 0, 0, 0);
            }
        }
    };
    const int ngroups = 4 * g.nchunk;  // (even)
    request(0, ra[0], rb[0]);
    request(1, ra[1], rb[1]);
    __syncthreads();                   // (the LDS of the previous task is free)
    to_lds(0, ra[0], rb[0]);
    if (2 < ngroups) request(2, ra[0], rb[0]);
    __syncthreads();
#pragma unroll 1
    for (int gi = 0; gi < ngroups; gi += 2) {
        // even group gi: group gi + 1 waits in ra[1], group gi + 2 is in flight into ra[0]
        to_lds(gi + 1, ra[1], rb[1]);
        if (gi + 3 < ngroups) request(gi + 3, ra[1], rb[1]);
        products(gi);
        __syncthreads();
        // odd group gi + 1
        if (gi + 2 < ngroups) to_lds(gi + 2, ra[0], rb[0]);
        if (gi + 4 < ngroups) request(gi + 4, ra[0], rb[0]);
        products(gi + 1);
        __syncthreads();
    }
    if (skip_wave) return;
#pragma unroll
    for (int x = 0; x < MTC; ++x) {
#pragma unroll
        for (int y = 0; y < MTR; ++y) {
            const int li = wr * SUBR + y * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lj = wc * SUBC + x * 16 + (lane >> 4) + 4 * r;
                if (g.diag && li < lj) continue;
                const double v = acc[x][y][r];
                if constexpr (FAR) st_agent(g.C + (size_t)li + (size_t)lj * g.ldc, v);
                else if (g.part) st_agent(g.part + li + lj * TS, v);
                else {
                    double base = ((const __attribute__((address_space(1))) double*)g.Pt)[(size_t)li + (size_t)lj * g.ldp];
                    if (g.diag && li == lj) base += ((const __attribute__((address_space(1))) double*)g.xr)[li];
                    if (g.At) base += g.dinv * ((const __attribute__((address_space(1))) double*)g.At)[(size_t)li + (size_t)lj * g.ldat];
                    st_agent(g.C + (size_t)li + (size_t)lj * g.ldc, base + v);
                }
            }
        }
    }
}
// the last slice of a K-sliced assembly tile to finish adds the partial sums in slice order and applies the epilogue (k_syrk_tail_reduce's arithmetic)
__device__ __forceinline__ void accum_reduce(const AccumArgs& g, int nslice)
{
    for (int idx = threadIdx.x; idx < TS * TS; idx += blockDim.x) {
        const int li = idx & (TS - 1), lj = idx >> 7;
        if (g.diag && li < lj) continue;
        double v = 0.0;
        for (int sl = 0; sl < nslice; ++sl) v += ld_agent(g.part + (size_t)sl * TS * TS + idx);
        double base = ((const __attribute__((address_space(1))) double*)g.Pt)[(size_t)li + (size_t)lj * g.ldp];
        if (g.diag && li == lj) base += ((const __attribute__((address_space(1))) double*)g.xr)[li];
        if (g.At) base += g.dinv * ((const __attribute__((address_space(1))) double*)g.At)[(size_t)li + (size_t)lj * g.ldat];
        st_agent(g.C + (size_t)li + (size_t)lj * g.ldc, base + v);
    }
}

constexpr int CHOL_THREADS = 512;
// The two roles as out-of-line functions: inlined into the ticket loop they drove the kernel to 256 VGPRs + 224 spilled (892 B of scratch per lane) and
// the diagonal block took twice as long as in the launch-per-panel kernel (208 VGPRs, no spills).  The LDS view is rebuilt from the extern symbol inside
// so that the compiler still emits ds_read / ds_write (a pointer passed in would be a generic one: FLAT accesses), and the functions carry the kernel's
// workgroup size / waves-per-SIMD attributes: a device function without them is compiled for 1024-thread workgroups, i.e. 128 VGPRs (the tile role
// then spilled: 44 us per tile instead of 23).
// (a pointer read out of a struct in memory is a GENERIC pointer to the compiler: 530 FLAT loads in the tile role, which also tie the LDS waits to the
// global traffic -- 44 us per tile; ld_agent / st_agent and the flag helpers therefore cast to the global address space themselves)
__device__ __noinline__ bool chol_role_crew(const SyrkArgs& a, int role)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const SyrkArgs b = a;  // a register copy: through the reference every field is re-read from memory (with a full drain) behind each store of the loop
    return fused_next_diag<CHOL_THREADS, true>(b, smem, role);
}
__device__ __noinline__ bool chol_role_tile(const SyrkArgs& a, int ti, int tj)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const SyrkArgs b = a;
    return fused_tile<CHOL_THREADS, true>(b, ti, tj, smem);
}
__device__ __noinline__ bool chol_role_half(const SyrkArgs& a, int ti, int tj, int h)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const SyrkArgs b = a;
    return fused_tile<CHOL_THREADS, true, true>(b, ti, tj, smem, h);
}
__device__ __noinline__ void chol_role_accum_far(const AccumArgs& a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const AccumArgs b = a;
    accum_tile<CHOL_THREADS, true>(b, smem);
}
__device__ __noinline__ void chol_role_accum_asm(const AccumArgs& a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const AccumArgs b = a;
    accum_tile<CHOL_THREADS, false>(b, smem);
}
__global__ __launch_bounds__(CHOL_THREADS, 2) void k_chol_persistent(CholArgs c)
{
    __shared__ int s_ticket;
    __shared__ CholTask s_task;
    __shared__ CholTask s_asm;
    __shared__ int s_help, s_last;
    const int tid = threadIdx.x;
    int* abort_w = c.ticket + 1;
    const int T = c.T, NB = FACTOR_NB, ntasks = c.ntasks;
    for (;;) {
        if (tid == 0) {
            // Thread 0 draws the next ticket -- by fetch_add (a compare-and-swap on the head serialises a grid of drawers: 18 ms per factorisation,
            // measured) and only while the head's gate is open: an idle workgroup sleeps on `progress` instead of parking on a task a round away and
            // polling its words.  A ticket drawn past the gate in a race is still served: inside a task a workgroup only ever waits for EARLIER tickets.
            int t = -1;
            unsigned spins = 0;
            // (Every step of this is a round trip to L2, 0.7-1 us under load, and a bulk-bound round is thousands of tasks: the two words are requested
            // together, the head's task record is kept -- it is the drawn task unless another workgroup drew in between -- and handed to the workgroup
            // through LDS instead of being read again by everybody.)
            CholTask tk = {0, 0, 0, 0, 0};
            for (;;) {
                const int praw = ldi_agent(c.progress);
                const int cur = ldi_agent(c.ticket);
                const int P = max(praw - c.gen, 0);
                if (cur >= ntasks) { t = ntasks; break; }
                tk = c.tasks[cur];
                if (tk.gate <= P) {
                    t = addi_agent(c.ticket, 1);
                    if (t != cur && t < ntasks) tk = c.tasks[t];
                    break;
                }
                // the head's gate is closed.  Fused assembly: an assembly task needs nothing -- take one from the queues instead of sleeping (its token becomes a
                // no-op when the list reaches it).  Without this, 170 of 256 workgroups slept at the gate for 200 us at a time while 900 assembly tasks waited
                // further down the list (tools/chol_trace_report.py on the first version: 31 % of the launch between tasks).
                if (c.GT != nullptr && c.aq_ptr[8] > 0) {
                    bool got = false;
                    const int xcd = (int)blockIdx.x & 7;
                    for (int q8 = 0; q8 < 8 && !got; ++q8) {
                        const int y = (xcd + q8) & 7, len = c.aq_ptr[y + 1] - c.aq_ptr[y];
                        if (len <= 0 || ldi_agent(c.aq_head + y) >= len) continue;
                        const int hq = addi_agent(c.aq_head + y, 1);
                        if (hq < len) { s_asm = c.aq[c.aq_ptr[y] + hq]; got = true; }
                    }
                    if (got) { t = -2; break; }
                }
                __builtin_amdgcn_s_sleep(127);  // (an idle workgroup polls rarely: a grid of pollers slows the write-through traffic of the chain)
                if (((++spins & 15u) == 0 && ldi_agent(abort_w) != 0) || spins > 8000000u) { t = ntasks; break; }
            }
            s_task = tk;
            s_ticket = t;
        }
        __syncthreads();
        const int t = s_ticket;
        const CholTask tk = s_task;
        __syncthreads();
        if (t >= ntasks) return;
        const bool gate_help = t == -2;  // an assembly task taken at a closed gate (s_asm), no ticket
        if (!gate_help && c.trace && tid == 0) { c.trace[4 * (size_t)t] = wall_clock64(); c.trace[4 * (size_t)t + 3] = blockIdx.x; }
        const bool asm_in = c.GT != nullptr;  // fused assembly: a tile's first update waits for tver == gen like every later one waits for gen + k
        // thread 0: the next assembly task -- own XCD's queue first, then the others'
        auto draw_asm = [&]() -> bool {
            const int xcd = (int)blockIdx.x & 7;
            for (int q = 0; q < 8; ++q) {
                const int y = (xcd + q) & 7, len = c.aq_ptr[y + 1] - c.aq_ptr[y];
                if (len <= 0 || ldi_agent(c.aq_head + y) >= len) continue;
                const int hq = addi_agent(c.aq_head + y, 1);
                if (hq < len) { s_asm = c.aq[c.aq_ptr[y] + hq]; return true; }
            }
            return false;
        };
        // one K-slice of the assembly of the absolute tile (q.a, q.b); the last slice of a tile to finish adds the partial sums up and announces the tile
        auto run_asm = [&](const CholTask& q, long long* tr) {
            const int i = q.a, j = q.b, sl = q.round, mch = c.m / NB, S = chol_asm_slices(j, mch);
            const int c0 = (int)((long long)sl * mch / S), c1 = (int)((long long)(sl + 1) * mch / S);
            AccumArgs g;
            g.C = c.A + (size_t)i * NB + (size_t)j * NB * c.lda; g.ldc = c.lda; g.diag = i == j ? 1 : 0;
            // (GT as row panels, launch_pack_row_panels: panel i = 128 x m, ld = 128)
            g.A0 = g.A1 = c.GT + ((size_t)i * c.m + (size_t)c0 * NB) * NB;
            g.B0 = g.B1 = c.GT + ((size_t)j * c.m + (size_t)c0 * NB) * NB;
            g.ld = NB; g.nchunk = c1 - c0; g.w = c.zinv + (size_t)c0 * NB;
            g.part = S > 1 ? c.part + ((size_t)q.aux + sl) * TS * TS : nullptr;
            g.Pt = c.Pfull + (size_t)i * NB + (size_t)j * NB * c.ldp; g.ldp = c.ldp; g.xr = c.x_reg + (size_t)i * NB;
            g.At = c.ATA ? c.ATA + (size_t)i * NB + (size_t)j * NB * c.ldata : nullptr; g.ldat = c.ldata; g.dinv = c.dinv;
            if (tr && tid == 0) { tr[1] = wall_clock64(); tr[3] = (long long)blockIdx.x + 1000ll * j; }
            chol_role_accum_asm(g);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            bool last = S == 1;
            if (S > 1) {
                if (tid == 0) s_last = (addi_agent(c.acnt + (size_t)i * T + j, 1) + 1 - (c.fcount + 1) * S == 0) ? 1 : 0;
                __syncthreads();
                last = s_last != 0;
                __syncthreads();
                if (last) {
                    g.part = c.part + (size_t)q.aux * TS * TS;
                    accum_reduce(g, S);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                }
            }
            if (last && tid == 0) __hip_atomic_store(c.tver + (size_t)i * T + j, c.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tr && tid == 0) tr[2] = wall_clock64();
        };
        // before a task sleeps on its tile: if the tile is not assembled yet, assemble (anything) until it is or the queues are empty
        auto help_assembled = [&](int i, int j) {
            if (!asm_in) return;
            for (;;) {
                if (tid == 0) s_help = (ldi_agent(c.tver + (size_t)i * T + j) - c.gen >= 0) ? 2 : (draw_asm() ? 1 : 0);
                __syncthreads();
                const int st = s_help;
                const CholTask q = s_asm;
                __syncthreads();
                if (st != 1) return;
                run_asm(q, nullptr);
            }
        };
        if (gate_help) {
            const CholTask q = s_asm;
            __syncthreads();
            run_asm(q, nullptr);
            continue;
        }
        if (tk.kind == 8) {
            // ---- assembly token: the next task of this XCD's queue
            if (tid == 0) s_help = draw_asm() ? 1 : 0;
            __syncthreads();
            const int st = s_help;
            const CholTask q = s_asm;
            __syncthreads();
            if (st == 1) run_asm(q, c.trace ? c.trace + 4 * (size_t)t : nullptr);
            else if (c.trace && tid == 0) { c.trace[4 * (size_t)t + 1] = c.trace[4 * (size_t)t + 2] = wall_clock64(); c.trace[4 * (size_t)t + 3] = blockIdx.x; }
            continue;
        }
        if (tk.kind == 7) {
            // ---- several panels' updates of a far tile in one visit, absolute tile (i, j)
            const int i = tk.a, j = tk.b;
            AccumArgs g;
            g.C = c.A + (size_t)i * NB + (size_t)j * NB * c.lda; g.ldc = c.lda; g.diag = i == j ? 1 : 0;
            g.part = nullptr; g.Pt = nullptr; g.ldp = 0; g.xr = nullptr; g.At = nullptr; g.ldat = 0; g.dinv = 0.0;
            const int klo = tk.round, khi = tk.gate + 1, kp = khi - 1;
            auto ready = [&](int row) { return (c.fcount + 1) * (kp > 0 ? chol_split_row(T, kp - 1, row - kp, c.split_thr) : 1); };
            help_assembled(i, j);
            // (without the fused assembly a tile that has received no update carries no mark of this launch: its first visit does not wait for one)
            bool ok = chol_wait3((klo > 0 || asm_in) ? c.tver + (size_t)i * T + j : nullptr, c.gen + klo, kp > 0 ? c.lready + (size_t)kp * T + i : nullptr, ready(i), kp > 0 ? c.lready + (size_t)kp * T + j : nullptr, ready(j), abort_w, 1);
            if (c.trace && tid == 0) c.trace[4 * (size_t)t + 1] = wall_clock64();
            if (ok) {
                const size_t col = (size_t)klo * NB * c.lda;
                g.A0 = (klo == 0 ? c.A : c.side) + (size_t)i * NB + col; g.A1 = c.side + (size_t)i * NB + col;
                g.B0 = (klo == 0 ? c.A : c.side) + (size_t)j * NB + col; g.B1 = c.side + (size_t)j * NB + col;
                g.ld = c.lda; g.nchunk = khi - klo; g.w = c.ldlt ? c.dvec + (size_t)klo * NB : nullptr;
                chol_role_accum_far(g);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_store(c.tver + (size_t)i * T + j, c.gen + khi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (c.trace && tid == 0) c.trace[4 * (size_t)t + 2] = wall_clock64();
            if (!ok) {
                if (tid == 0) { __hip_atomic_store(abort_w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (*c.info < 0) *c.info = 0; }
                return;
            }
            continue;
        }
        const int k = tk.round, kk = k * NB, rs = c.n - kk - NB;  // rs = order of the trailing matrix of round k (a multiple of 128, >= 128)
        SyrkArgs a;
        a.n = rs; a.kdim = NB;
        a.A = (k == 0 ? c.A : c.side) + (kk + NB) + (size_t)kk * c.lda; a.lda = c.lda;  // operand panel k: from the side copy (panel 0: in place, solved before the launch)
        a.B = a.A; a.ldb = c.lda;
        a.fuse_side = c.side + (kk + NB) + (size_t)(kk + NB) * c.lda;  // where this round's panel tasks leave the copy of panel k + 1 (same offsets as C)
        a.w = c.ldlt ? c.dvec + kk : nullptr;
        a.C = c.A + (kk + NB) + (size_t)(kk + NB) * c.lda; a.ldc = c.lda;
        a.fuse_nb = NB; a.fuse_kglobal = kk + NB; a.fuse_ldlt = c.ldlt; a.fuse_info = c.info; a.fuse_rdiag = c.rdiag; a.fuse_dvec = c.dvec + kk + NB;
        a.fuse_pack = (rs - NB > 0) ? c.pack2 + (size_t)(k & 1) * FACTOR_PACK_DOUBLES : nullptr;
        a.fuse_w16 = c.w16 + (size_t)((kk + NB) / 16) * 256;
        a.fuse_token = c.token_base + k + 1; a.fuse_flags = c.fuse_flags; a.fuse_scratch = c.scratch; a.fuse_cnt = c.fuse_cnt;
        a.fuse_abort = abort_w;
        // progressive hand-over of the first panel row: the strip waves of round k's panel tasks (1, 0) count every published slice s in xcnt[8 k + s]
        // (8 waves per slice and factorisation, cumulative); the crew of round k + 1 consumes them
        a.fuse_xpub = c.xcnt + 8 * (size_t)k;
        a.fuse_xcnt = k > 0 ? c.xcnt + 8 * (size_t)(k - 1) : nullptr;
        a.fuse_xwant = (c.fcount + 1) * 8;
        a.fuse_tr2 = c.trace ? c.trace + 4 * (size_t)ntasks + 64 * (size_t)k : nullptr;         // stamps of round k's first panel row ...
        a.fuse_tr2n = c.trace ? c.trace + 4 * (size_t)ntasks + 64 * (size_t)(k > 0 ? k - 1 : T) : nullptr;  // ... and of the crew that consumes it (round k's crew reads round k - 1's row)
        // absolute block coordinates of what this task touches: relative tile (ti, tj) of round k = absolute (k + 1 + ti, k + 1 + tj)
        const int* lr = c.lready + (size_t)k * T;   // panel k
        // (cumulative counters: block row i of panel k was solved by the split_row(k - 1, i - k) tasks of round k - 1; the first row of panel k, i = k + 1, by two)
        auto ready_of = [&](int i) { return (c.fcount + 1) * (k > 0 ? chol_split_row(T, k - 1, i - k, c.split_thr) : 1); };
        bool ok = true;
        if (tk.kind <= 1) {
            // crew of the next diagonal block: its tile (k + 1, k + 1) must have received U_0 .. U_{k-1}; operand = block row k + 1 of panel k.  The
            // owner also waits until every panel task of round k - 2 is done with the pack buffer this round writes.
            const int d = k + 1;
            // (its operand, block row d of panel k, arrives slice by slice inside fused_next_diag)
            help_assembled(d, d);
            ok = chol_wait3((k > 0 || asm_in) ? c.tver + (size_t)d * T + d : nullptr, c.gen + k, nullptr, 0,
                            (tk.kind == 1 && k >= 2) ? c.pdone + (k - 2) : nullptr, (c.fcount + 1) * chol_panel_tasks(T, k - 2, c.split_thr), abort_w, 0);
            if (c.trace && tid == 0) c.trace[4 * (size_t)t + 1] = wall_clock64();
            if (ok) ok = chol_role_crew(a, tk.kind == 1 ? 0 : tk.a);
        } else {
            // a tile (ti, tj) of the trailing matrix = absolute (i, j); tj == 0: the next panel (solved behind the diagonal block by the same workgroup)
            const bool panel = tk.kind == 2;
            const int dh = tk.kind >= 4 ? tk.kind - 4 : -1;  // half of a split bulk tile (the diagonal tile of the first bulk column: it feeds the next crew)
            const int ti = tk.a, tj = panel ? 0 : tk.b;
            const int i = k + 1 + ti, j = k + 1 + tj;
            help_assembled(i, j);
            if (panel && k > 0) {
                ok = chol_wait3(c.tver + (size_t)i * T + j, c.gen + k, nullptr, 0, nullptr, 0, abort_w, 0);
                a.late_p[0] = lr + i; a.late_w[0] = ready_of(i); a.late_p[1] = lr + j; a.late_w[1] = ready_of(j);
            } else {
                ok = chol_wait3((k > 0 || asm_in) ? c.tver + (size_t)i * T + j : nullptr, c.gen + k, k > 0 ? lr + i : nullptr, ready_of(i), k > 0 ? lr + j : nullptr, ready_of(j), abort_w, panel ? 0 : 1);
            }
            if (c.trace && tid == 0) c.trace[4 * (size_t)t + 1] = wall_clock64();
            if (ok) ok = (panel && tk.b >= 0) ? chol_role_half(a, ti, 0, tk.b) : (dh >= 0 ? chol_role_half(a, ti, tj, dh) : chol_role_tile(a, ti, tj));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (ok && tid == 0) {
                if (panel) {
                    addi_agent(c.lready + (size_t)(k + 1) * T + i, 1);
                    const int before = addi_agent(c.pdone + k, 1);
                    // the last panel task of the round: rounds complete in order (row i of panel k + 1 needs row i of panel k)
                    if (before + 1 - (c.fcount + 1) * chol_panel_tasks(T, k, c.split_thr) == 0) sti_agent(c.progress, c.gen + k + 1);
                } else if (dh < 0 || addi_agent(c.dhalf + k, 1) + 1 == 2 * (c.fcount + 1)) {  // (a split tile: the second half to finish announces it)
                    __hip_atomic_store(c.tver + (size_t)i * T + j, c.gen + k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        if (c.trace && tid == 0) c.trace[4 * (size_t)t + 2] = wall_clock64();
        if (!ok) {
            // a bounded wait gave up (cannot happen with a healthy device): everybody leaves, the factorisation is reported as failed
            if (tid == 0) { __hip_atomic_store(abort_w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (*c.info < 0) *c.info = kk + NB; }
            return;
        }
    }
}

// host side: the task list of a T x T tile grid (T >= 3), in ticket order.
// (Tried and measured at n = 4096, factorisation ms in bench.py: this list 1.35-1.45; two queues -- crew, panel and first-column tiles in one served
// first, the other tiles in a second, panel tasks drawn only when the round before is complete, a drawn task whose conditions do not hold yet held back --
// 1.50: the rows start their update later than with the early draw, and that costs more than the parked workgroups it saves; drawing with a
// compare-and-swap on the head, so that nobody overshoots a gate: 18 ms.
// Round 3 repeated the two queues WITH the early draw kept (crew, panel rows and the first two tile columns in a chain queue served first, the other tiles in a
// bulk queue, every task carrying the number of tickets of the other queue that must be drawn before it may start, a workgroup holding such a task serving the
// other queue meanwhile -- deadlock-free under any residency, bitwise equal results): 1.35-1.41 ms against 1.27 for this list on the same box, whatever the
// queue membership (0 / 1 / 2 / 4 tile columns) and with or without a late draw of the panel rows >= 3.  The timeline shows why: in the bulk-bound rounds the
// chain queue's ~130 tasks per round are drawn the moment their gate opens and wait inside, which takes a third of the chip away from the trailing tiles --
// the chain reaches round 9 at 600 us instead of 512.  The single list meters the chain tasks out between the tiles; that is worth more than the bypass.)
constexpr double CHOL_DEFER = 0.05;
constexpr bool CHOL_SPLIT_DIAG = true;
static int chol_far_g()  // PIQP_AMD_DEBUG=chol_far=<g>: panels per visit of the far tiles (1: one panel per visit as in round 3; 4: rounds 4-5)
{
    // round 6, persistent launch at n = 4096 (tools/time_chol.py): g = 1 1.238 ms, 2 1.158, 3 1.17, 4 1.203, 6 1.32, 8 1.37 with the visits in phase; out of phase by
    // the tile's column (below) 2: 1.131, 3: 1.147, 4: 1.181
    static const int g = [] { const char* e = debug_token("chol_far"); const int v = e ? std::atoi(e) : 2; return v < 1 ? 1 : (v > 16 ? 16 : v); }();
    return g;
}
static void chol_build_tasks(int T, std::vector<CholTask>& H)
{
    static const int far_slack = [] { const char* e = debug_token("chol_slack"); return e ? std::atoi(e) : 3; }();
    static const double defer = [] { const char* e = debug_token("chol_defer"); return e ? std::atof(e) : CHOL_DEFER; }();
    const int far_g = chol_far_g();
    static const int far_phase = [] { const char* e = debug_token("chol_phase"); return e ? std::atoi(e) : 1; }();  // 0: every tile's visits end with the same panels (g - 1, 2 g - 1, ...: rounds 4-5); 1: shifted by the tile's column (default); 2: by row + column
    // One queue, sorted by a key in units of chain rounds.  The crew and panel tasks of round k have key k - 1 (drawn a round early: the crew follows the
    // slices of its operand, a panel task waits for its own rows); the first tile column of round k, which feeds them, k - 0.5; tile column tj >= 2 of
    // round k,  k + CHOL_DEFER (tj - 1):  with CHOL_DEFER = 0 that is the plain order  crew(k+1) panel(k+1) bulk(k) ...;  a small slope pushes the far
    // columns of the early rounds (465 tiles against a 36 us chain round: more than the chip does) a round or two back, behind the next chain tasks.
    // Measured at T = 32 (factorisation ms): slope 0: 1.333, 0.02: 1.326, 0.04: 1.312, 0.07: 1.310, 0.1: 1.329, and 1.42 for the full deadline order
    // (0.2 (j - 1) + 0.8 k: every workgroup busy with deferred tiles when a chain task comes up).  A tile's updates keep their order (the key grows with k),
    // every task only waits for tasks with a smaller key, and a gate is never later than the tasks around it need (gate = round of the panel read).
    struct Keyed { double key; int cls; CholTask t; };
    std::vector<Keyed> all;
    for (int k = 0; k + 1 < T; ++k) {
        const int Tk = T - k - 1, gate = std::max(k - 1, 0);
        const double kc = (double)k - 1.0;
        for (int r = 1; r < FUSE_ROLES; ++r) all.push_back({kc, 0, {0, (short)k, (short)r, 0, gate}});
        all.push_back({kc, 1, {1, (short)k, 0, 0, gate}});
        // (Tried: drawing the panel tasks of the rows >= 3 of the bulk-bound rounds only when their inputs exist -- gate k, key k - 0.3 ... k - 0.7 -- so that they do
        // not park a workgroup for a round: 1.24-1.27 ms against 1.26, inside the run-to-run spread.  Not kept.)
        for (int ti = 1; ti < Tk; ++ti) {
            const int spr = chol_split_row(T, k, ti, chol_split_thr());
            if (spr == 1) all.push_back({kc, 2, {2, (short)k, (short)ti, -1, gate}});
            else for (int h = 0; h < spr; ++h) all.push_back({kc, 2, {2, (short)k, (short)ti, (short)h, gate}});
        }
        for (int tj = 1; tj < Tk; ++tj) {
            // (the second tile column as well: its diagonal tile is the diagonal block two rounds on, its other tiles the panel after next -- behind the deferred far
            // columns of earlier rounds their updates arrived late, and the crew of round k + 2 waited for them)
            const double key = tj == 1 ? (double)k - 0.5 : (tj == 2 ? (double)k - 0.25 : (double)k + defer * (tj - 1));
            // round 4: the updates of a FAR tile column (absolute column j, updates k <= j - 4 - far_slack) are taken far_g panels per visit (kind 7: one C
            // fetch, one store, one ticket for far_g x 128 operand columns; same products in the same order -- bitwise the same tile)
            const int j = k + 1 + tj, nfar = j - 3;
            for (int ti = tj; ti < Tk; ++ti) {
                // (round 6: the visits of different tiles are out of phase -- with every far tile visited in the rounds 3, 7, 11, ... those rounds carried a burst of
                // ~150 four-panel tasks and the chain's own tasks queued behind it: rounds of 55-65 us among rounds of 37-42, profiles/r04_chol_timeline.txt)
                const int i = k + 1 + ti;
                const int o = far_g > 1 ? (far_phase == 1 ? j % far_g : (far_phase == 2 ? (i + j) % far_g : 0)) : 0;  // the first o panels of the tile one by one
                const int nb = (far_g > 1 && nfar - far_slack - o > 0) ? (nfar - far_slack - o) / far_g : 0;
                const bool in_block = k >= o && k < o + nb * far_g;
                if (in_block && ((k - o) % far_g) != far_g - 1) continue;  // (the visit is listed with its last panel)
                // tile (1, 1) is the next round's diagonal block: its update stands between this round's second panel row and the next crew -- two workgroups
                if (in_block) all.push_back({key, 3, {7, (short)(k - far_g + 1), (short)(k + 1 + ti), (short)j, k}});
                else if (ti == 1 && tj == 1 && CHOL_SPLIT_DIAG) { all.push_back({key, 3, {4, (short)k, 1, 1, k}}); all.push_back({key, 3, {5, (short)k, 1, 1, k}}); }
                else all.push_back({key, 3, {3, (short)k, (short)ti, (short)tj, k}});
            }
        }
    }
    std::stable_sort(all.begin(), all.end(), [](const Keyed& a, const Keyed& b) { return a.key != b.key ? a.key < b.key : a.cls < b.cls; });
    H.clear();
    for (const Keyed& q : all) H.push_back(q.t);
}
// The task list WITH the assembly inside the launch (round 4).  What changes against chol_build_tasks:
//  * every tile (i, j), j >= 1, is assembled by chol_asm_slices(j) tasks of kind 6 (column-major: the columns the chain needs first come first); block column 0
//    is assembled, factored and solved by launches before this one;
//  * the updates k <= j - 4 of a tile of column j ("far": three or more tile columns ahead of the chain) are taken FAR_G panels per visit (kind 7) as long as the
//    visit ends FAR_SLACK columns before the chain needs the tile; the rest stay one panel per visit (kind 3), the near ones untouched;
//  * the order comes from a time model instead of round keys (the chain cannot run ahead of the assembly of its columns any more): assembly tasks at the time the
//    chip reaches them (FUSE_WEFF workgroups of FUSE_CHUNK_US per operand chunk), the chain's round k at tau_c(k) = max(tau_c(k - 1) + FUSE_ROUND_US, column k + 1
//    assembled), its tasks relative to tau_c(k) like their keys relative to k before;
//  * one relaxation pass in creation order (which is a topological order) pushes every task behind everything it waits for, so "a workgroup only ever waits for
//    earlier tickets" -- the property that makes the single list deadlock-free under any residency -- holds by construction whatever the model says.
constexpr int FAR_G = 4, FAR_SLACK = 3;
constexpr double FUSE_ROUND_US = 43.0, FUSE_CHUNK_US = 19.5, FUSE_VISIT_US = 20.0, FUSE_WEFF = 230.0;  // (measured, PIQP_AMD_DEBUG=chol_trace: 19.5 us per 128-column chunk, 20 us per task)
static void chol_build_tasks_fused(int T, int mchunks, std::vector<CholTask>& H, int& part_slots, std::vector<CholTask>& Q, int (&qptr)[9])
{
    struct Node { double tau, dur; int cls; CholTask t; std::vector<int> deps; };
    std::vector<Node> N;
    auto add = [&](double tau, double dur, int cls, CholTask t, std::vector<int> deps) { N.push_back({tau, dur, cls, t, std::move(deps)}); return (int)N.size() - 1; };
    std::vector<std::vector<int>> lastw((size_t)T * T);     // tasks whose completion the next update of tile (i, j) waits for
    std::vector<std::vector<int>> solved((size_t)T * T);    // [k T + i]: panel tasks that solve block row i of panel k
    std::vector<std::vector<int>> panel_of_round((size_t)T);
    std::vector<int> owner((size_t)T, -1);
    auto cat = [](std::vector<int> a, const std::vector<int>& b) { a.insert(a.end(), b.begin(), b.end()); return a; };
    // ---- assembly: eight queues (tile (i, j) in queue i mod 8), each column-major and slice-major inside a column; one token per task in the ticket list
    part_slots = 0;
    std::vector<int> aux((size_t)T * T, 0);
    for (int j = 1; j < T; ++j) {
        const int S = chol_asm_slices(j, mchunks);
        for (int i = j; i < T; ++i) { aux[(size_t)i * T + j] = part_slots; if (S > 1) part_slots += S; }
    }
    Q.clear();
    std::vector<int> qpos((size_t)T * T, 0);  // position inside its queue of the LAST slice of tile (i, j)
    std::vector<double> qdur;
    for (int x = 0; x < 8; ++x) {
        qptr[x] = (int)Q.size();
        // column-major (the chain needs block column k + 1 in round k), slice-major inside a column.  (Tried: bands of eight columns slice-major, so that what an
        // XCD runs together reads one K range of a dozen operand panels -- the chunks took the same 20 us, which is 13.7 us of matrix-core work at the clock the
        // whole chip sustains under this load, not memory; and the band's columns all finished together, which stalled the chain for 400 us behind round 1.)
        for (int jb = 1; jb < T; ++jb) {
            const int S = chol_asm_slices(jb, mchunks);
            for (int sl = 0; sl < S; ++sl)
                for (int j = jb; j < jb + 1; ++j)
                    for (int i = j; i < T; ++i) {
                        if ((i & 7) != x) continue;
                        qpos[(size_t)i * T + j] = (int)Q.size() - qptr[x];
                        Q.push_back({6, (short)sl, (short)i, (short)j, x, aux[(size_t)i * T + j]});
                        const int c0 = (int)((long long)sl * mchunks / S), c1 = (int)((long long)(sl + 1) * mchunks / S);
                        qdur.push_back(FUSE_VISIT_US + FUSE_CHUNK_US * (c1 - c0));
                    }
        }
    }
    qptr[8] = (int)Q.size();
    // tokens: the queues advance together (one token in eight goes to each), so token number t stands for position t / 8 of every queue
    const int ntok = (int)Q.size();
    std::vector<int> tok((size_t)ntok);
    {
        double cum = 0.0;
        int maxlen = 0;
        for (int x = 0; x < 8; ++x) maxlen = std::max(maxlen, qptr[x + 1] - qptr[x]);
        int made = 0;
        for (int pos = 0; pos < maxlen; ++pos)
            for (int x = 0; x < 8; ++x) {
                if (pos >= qptr[x + 1] - qptr[x]) continue;
                const double dur = qdur[(size_t)qptr[x] + pos];
                tok[(size_t)made++] = add(cum / FUSE_WEFF, dur, 3, {8, 0, 0, 0, 0, 0}, {});
                cum += dur;
            }
    }
    std::vector<double> col_done((size_t)T, 0.0);
    auto tile_token = [&](int i, int j) {  // the token by which tile (i, j) is expected to be drawn completely
        const int x = i & 7;
        int before = 0;  // tokens made before position qpos of queue x: all queues' entries at smaller positions + the queues < x at this one
        const int pos = qpos[(size_t)i * T + j];
        for (int y = 0; y < 8; ++y) before += std::min(qptr[y + 1] - qptr[y], pos + (y <= x ? 1 : 0));
        return tok[(size_t)std::min(std::max(before - 1, 0), ntok - 1)];
    };
    for (int j = 1; j < T; ++j)
        for (int i = j; i < T; ++i) {
            const int id = tile_token(i, j);
            lastw[(size_t)i * T + j].push_back(id);
            col_done[(size_t)j] = std::max(col_done[(size_t)j], N[(size_t)id].tau + N[(size_t)id].dur + 15.0);
        }
    // ---- the chain's clock
    std::vector<double> tc((size_t)T, 0.0);
    for (int k = 0; k + 1 < T; ++k) tc[(size_t)k] = std::max(k > 0 ? tc[(size_t)k - 1] + FUSE_ROUND_US : 0.0, col_done[(size_t)k + 1] + 5.0);
    const double R = FUSE_ROUND_US;
    for (int k = 0; k + 1 < T; ++k) {
        const int Tk = T - k - 1, gate = std::max(k - 1, 0), d = k + 1;
        const double t0 = tc[(size_t)k];
        // crew of the next diagonal block (tile (d, d)): its previous update, block row d of panel k; the owner also the panel tasks of round k - 2 (pack buffer)
        {
            std::vector<int> deps = lastw[(size_t)d * T + d];
            if (k > 0) deps = cat(deps, solved[(size_t)k * T + d]);
            for (int r = 1; r < FUSE_ROLES; ++r) add(t0 - R, 20.0, 0, {0, (short)k, (short)r, 0, gate, 0}, deps);
            if (k >= 2) deps = cat(deps, panel_of_round[(size_t)k - 2]);
            owner[(size_t)k] = add(t0 - R, 5.0, 1, {1, (short)k, 0, 0, gate, 0}, deps);  // (the panel rows FOLLOW the owner: a short lead)
        }
        for (int ti = 1; ti < Tk; ++ti) {
            const int i = k + 1 + ti;
            std::vector<int> deps = lastw[(size_t)i * T + d];
            if (k > 0) deps = cat(cat(deps, solved[(size_t)k * T + i]), solved[(size_t)k * T + d]);
            deps.push_back(owner[(size_t)k]);
            const int spr = chol_split_row(T, k, ti, chol_split_thr());
            std::vector<int> ids;
            for (int h = 0; h < spr; ++h) ids.push_back(add(t0 - R, 30.0, 2, {2, (short)k, (short)ti, (short)(spr == 1 ? -1 : h), gate, 0}, deps));
            solved[(size_t)(k + 1) * T + i] = ids;
            lastw[(size_t)i * T + d] = ids;
            panel_of_round[(size_t)k] = cat(panel_of_round[(size_t)k], ids);
        }
        for (int tj = 1; tj < Tk; ++tj) {
            const int j = k + 1 + tj;
            // is update k of column j part of a FAR_G visit?  (far: k <= j - 4; visits cover [0, nb FAR_G), nb = (j - 3 - FAR_SLACK) / FAR_G)
            const int nfar = j - 3, nb = nfar - FAR_SLACK > 0 ? (nfar - FAR_SLACK) / FAR_G : 0;
            const bool in_block = k < nb * FAR_G;
            if (in_block && (k % FAR_G) != FAR_G - 1) continue;  // (the visit is created with its last panel)
            const double tau = tj == 1 ? t0 - 0.5 * R : (tj == 2 ? t0 - 0.25 * R : t0 + CHOL_DEFER * R * (tj - 1));
            for (int ti = tj; ti < Tk; ++ti) {
                const int i = k + 1 + ti;
                std::vector<int> deps = lastw[(size_t)i * T + j];
                if (k > 0) deps = cat(cat(deps, solved[(size_t)k * T + i]), solved[(size_t)k * T + j]);
                std::vector<int> ids;
                if (in_block) ids.push_back(add(tau, FUSE_VISIT_US + FUSE_CHUNK_US * FAR_G, 3, {7, (short)(k - FAR_G + 1), (short)i, (short)j, k, 0}, deps));
                else if (ti == 1 && tj == 1 && CHOL_SPLIT_DIAG) { ids.push_back(add(tau, 18.0, 3, {4, (short)k, 1, 1, k, 0}, deps)); ids.push_back(add(tau, 18.0, 3, {5, (short)k, 1, 1, k, 0}, deps)); }
                else ids.push_back(add(tau, 25.0, 3, {3, (short)k, (short)ti, (short)tj, k, 0}, deps));
                lastw[(size_t)i * T + j] = ids;
            }
        }
    }
    // a task is drawn when what it waits for is expected to be DONE (drawn earlier it would park its workgroup for the rest of the producer's run: with the
    // producers' start times only, the tiles' first updates slept 250-600 us each behind 130-390 us assembly tasks and the launch took 7 ms)
    // -- except along the chain itself (crew and panel tasks waiting for chain or tile tasks), which keeps its early draw: a crew follows the slices of its operand.
    // The gate of a task (drawn only when that many rounds have ALL their panel tasks done) must refer to earlier tickets as well, or the head of the list waits
    // for a ticket behind it: every panel task of round gate - 1 (tests/test_chol_plan.py found exactly this in the first version of this list).
    for (Node& q : N) {
        for (int dep : q.deps) {
            const Node& d = N[(size_t)dep];
            // crew and panel tasks keep their early draw against chain and tile tasks (they FOLLOW their producers), and must: drawn at their inputs' finish
            // times they sat BEHIND bulk tasks of their own round whose gate (the round before complete) was still closed -- every round then started only when the
            // one before had completely finished: 150 us per round (tools/chol_trace_report.py)
            const bool early = q.cls <= 2 && d.t.kind != 8;
            q.tau = std::max(q.tau, d.tau + (early ? 1e-3 : d.dur));
        }
        if (q.t.gate >= 1 && q.t.kind != 8)
            for (int dep : panel_of_round[(size_t)q.t.gate - 1]) q.tau = std::max(q.tau, N[(size_t)dep].tau + 1e-3);
    }
    std::vector<int> order(N.size());
    for (size_t q = 0; q < N.size(); ++q) order[q] = (int)q;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return N[(size_t)a].tau != N[(size_t)b].tau ? N[(size_t)a].tau < N[(size_t)b].tau : N[(size_t)a].cls < N[(size_t)b].cls; });
    H.clear();
    for (int q : order) H.push_back(N[(size_t)q].t);
}
size_t chol_task_count(int T)
{
    size_t n = 0;
    for (int k = 0; k + 1 < T; ++k) { const int Tk = T - k - 1; n += FUSE_ROLES + (size_t)chol_panel_tasks(T, k, chol_split_thr()) + (size_t)Tk * (Tk - 1) / 2 + ((CHOL_SPLIT_DIAG && Tk >= 2) ? 1 : 0); }
    return n;
}

struct CholPlan {
    int T = 0;
    CholTask* tasks = nullptr;  // device
    int ntasks = 0;
    int grid = 0;
    int part_slots = 0;         // fused assembly: 128 x 128 slots of partial sums
    CholTask* aq = nullptr;     // fused assembly: the eight queues of assembly tasks (device), queue x = [aq_ptr[x], aq_ptr[x + 1])
    int aq_ptr[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
};
// device-resident task list per (device, T, operand chunks of the fused assembly or 0); built at create time (chol_prepare), never freed: a few tens of KB
static const CholPlan* chol_plan(int T, int mchunks = 0)
{
    static std::mutex mu;
    static std::map<std::pair<int, std::pair<int, int>>, CholPlan> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    const auto key = std::make_pair(dev, std::make_pair(T, mchunks));
    auto it = cache.find(key);
    if (it != cache.end()) return it->second.tasks ? &it->second : nullptr;
    CholPlan& P = cache[key];
    P.T = T;
    std::vector<CholTask> h, q;
    if (mchunks > 0) chol_build_tasks_fused(T, mchunks, h, P.part_slots, q, P.aq_ptr);
    else {
        chol_build_tasks(T, h);
    }
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_chol_persistent), hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS_BYTES) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k_chol_persistent), CHOL_THREADS, FUSED_LDS_BYTES) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); return nullptr; }
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) { (void)hipGetLastError(); return nullptr; }
    P.grid = std::min<int>(cus * per_cu, (int)h.size());
    if (hipMalloc(&P.tasks, sizeof(CholTask) * h.size()) != hipSuccess) { (void)hipGetLastError(); P.tasks = nullptr; return nullptr; }
    ++alloc_counter();
    if (hipMemcpy(P.tasks, h.data(), sizeof(CholTask) * h.size(), hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(P.tasks); P.tasks = nullptr; return nullptr; }
    if (!q.empty()) {
        if (hipMalloc(&P.aq, sizeof(CholTask) * q.size()) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(P.tasks); P.tasks = nullptr; return nullptr; }
        ++alloc_counter();
        if (hipMemcpy(P.aq, q.data(), sizeof(CholTask) * q.size(), hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(P.tasks); (void)hipFree(P.aq); P.tasks = nullptr; return nullptr; }
    }
    P.ntasks = (int)h.size();
    return &P;
}
// host-only: the ticket-ordered task list as (kind, round, a, b, gate, aux) -- tests/test_chol_plan.py checks on the CPU that every task only waits for earlier tickets
int chol_debug_plan(int T, int mchunks, int* out6, int capacity_tasks)
{
    std::vector<CholTask> h, q;
    int slots = 0, qptr[9];
    if (mchunks > 0) chol_build_tasks_fused(T, mchunks, h, slots, q, qptr);
    else chol_build_tasks(T, h);
    h.insert(h.end(), q.begin(), q.end());  // (the assembly queues behind the ticket list: kind 6, gate = queue)
    if (out6)
        for (size_t q = 0; q < h.size() && (int)q < capacity_tasks; ++q) {
            int* o = out6 + 6 * q;
            o[0] = h[q].kind; o[1] = h[q].round; o[2] = h[q].a; o[3] = h[q].b; o[4] = h[q].gate; o[5] = h[q].aux;
        }
    return (int)h.size();
}
bool chol_persistent_supported(int n) { return n % FACTOR_NB == 0 && n / FACTOR_NB >= 3 && n / FACTOR_NB <= 1024; }
bool chol_prepare(int n) { return chol_persistent_supported(n) && chol_plan(n / FACTOR_NB) != nullptr; }
// the fused assembly needs whole operand chunks and enough of them to be worth the partial sums
bool chol_fused_supported(int n, int m) { return chol_persistent_supported(n) && m >= 4 * FACTOR_NB && m % FACTOR_NB == 0; }
size_t chol_prepare_fused(int n, int m)  // 0: not available; else the doubles of partial-sum workspace (>= 1)
{
    if (!chol_fused_supported(n, m)) return 0;
    const CholPlan* P = chol_plan(n / FACTOR_NB, m / FACTOR_NB);
    return P ? (size_t)std::max(P->part_slots, 1) * TS * TS : 0;
}
size_t chol_flag_ints(int n) { const size_t T = (size_t)(n / FACTOR_NB); return 12 + 3 * T * T + 10 * T + 1; }
// Rounds 0 .. T - 2 of the blocked factorisation of the n x n lower triangle at A (the first diagonal block and the first panel are already
// factored / solved: launch_potrf_diag + launch_trsm_panel).  flags: chol_flag_ints(n) ints zeroed at allocation; gen: launch-unique, advancing by
// at least T + 2 per call; fcount: calls made before on this flag array; token_base: fused-launch tokens consumed so far (advances by T - 1).
bool launch_chol_persistent(bool ldlt, double* A, double* side, int lda, int n, int* info, double* rdiag, double* dvec, double* pack2, double* w16, double* scratch, int* fuse_flags, int* fuse_cnt,
                            int token_base, int* flags, int gen, int fcount, hipStream_t s, const CholAssembly* fa)
{
    const int T = n / FACTOR_NB;
    const CholPlan* P = chol_plan(T, fa ? fa->m / FACTOR_NB : 0);
    if (!P) return false;
    static long long* trace_d = nullptr;  // PIQP_AMD_DEBUG=chol_trace: timeline of the LAST launch, summarised to stderr (synchronises: a debugging aid)
    static size_t trace_n = 0;
    const bool want_trace = debug_token("chol_trace") != nullptr;
    if (want_trace && trace_n < 4 * (size_t)P->ntasks + 64 * (size_t)(T + 1)) {
        if (trace_d) (void)hipFree(trace_d);
        trace_n = 4 * (size_t)P->ntasks + 64 * (size_t)(T + 1);
        PQ_HIP(hipMalloc(&trace_d, trace_n * sizeof(long long)));
    }
    CholArgs c;
    c.A = A; c.side = side; c.lda = lda; c.n = n; c.T = T; c.ldlt = ldlt ? 1 : 0;
    c.info = info; c.rdiag = rdiag; c.dvec = dvec; c.pack2 = pack2; c.w16 = w16;
    c.scratch = scratch; c.fuse_flags = fuse_flags; c.fuse_cnt = fuse_cnt; c.token_base = token_base;
    c.tasks = P->tasks; c.ntasks = P->ntasks;
    c.ticket = flags; c.aq_head = flags + 4; c.lready = flags + 12; c.tver = c.lready + (size_t)T * T; c.pdone = c.tver + (size_t)T * T; c.xcnt = c.pdone + T; c.progress = c.xcnt + 8 * (size_t)T; c.dhalf = c.progress + 1;
    c.gen = gen; c.fcount = fcount; c.split_thr = chol_split_thr();
    c.acnt = c.dhalf + T;
    c.GT = nullptr; c.ldg = 0; c.m = 0; c.zinv = nullptr; c.Pfull = nullptr; c.ldp = 0; c.x_reg = nullptr; c.ATA = nullptr; c.ldata = 0; c.dinv = 0.0; c.part = nullptr;
    c.aq = P->aq; for (int x = 0; x < 9; ++x) c.aq_ptr[x] = P->aq_ptr[x];
    if (fa) { c.GT = fa->GT; c.ldg = fa->ldg; c.m = fa->m; c.zinv = fa->zinv; c.Pfull = fa->Pfull; c.ldp = fa->ldp; c.x_reg = fa->x_reg; c.ATA = fa->ATA; c.ldata = fa->ldata; c.dinv = fa->dinv; c.part = fa->part; }
    c.trace = want_trace ? trace_d : nullptr;
    if (want_trace) PQ_HIP(hipMemsetAsync(trace_d, 0, trace_n * sizeof(long long), s));
    PQ_HIP(hipMemsetAsync(flags, 0, 12 * sizeof(int), s));  // the ticket counter, the abort word, the heads of the assembly queues
    hipLaunchKernelGGL(k_chol_persistent, dim3(P->grid), dim3(CHOL_THREADS), FUSED_LDS_BYTES, s, c);
    PQ_HIP(hipGetLastError());
    if (want_trace) {
        std::vector<long long> h(4 * (size_t)P->ntasks + 64 * (size_t)(T + 1));
        std::vector<CholTask> tk((size_t)P->ntasks);
        PQ_HIP(hipMemcpyAsync(h.data(), trace_d, h.size() * sizeof(long long), hipMemcpyDeviceToHost, s));
        PQ_HIP(hipMemcpyAsync(tk.data(), P->tasks, tk.size() * sizeof(CholTask), hipMemcpyDeviceToHost, s));
        stream_wait(s);
        long long t0 = h[0];
        for (int t = 0; t < P->ntasks; ++t) t0 = std::min(t0, h[4 * (size_t)t]);
        auto us = [&](long long v) { return (double)(v - t0) * 0.01; };
        if (const char* fn = std::getenv("PIQP_AMD_CHOL_TRACE_FILE")) {
            // raw timeline of this launch, one line per ticket: ticket kind round a b gate drawn inputs done workgroup(+1000 j for an assembly token)  [us]
            if (FILE* fp = std::fopen(fn, "w")) {
                for (int t = 0; t < P->ntasks; ++t)
                    std::fprintf(fp, "%d %d %d %d %d %d %.2f %.2f %.2f %lld\n", t, tk[(size_t)t].kind, tk[(size_t)t].round, tk[(size_t)t].a, tk[(size_t)t].b, tk[(size_t)t].gate, us(h[4 * (size_t)t]),
                                 us(h[4 * (size_t)t + 1]), us(h[4 * (size_t)t + 2]), h[4 * (size_t)t + 3]);
                std::fclose(fp);
            }
        }
        std::fprintf(stderr, "[piqp_amd] k_chol_persistent timeline (us since the first ticket), T = %d, %d tasks, grid %d\n", T, P->ntasks, P->grid);
        if (fa) {
            // fused assembly: per block column the end of its last assembly task; the several-panel visits of the far tiles
            std::vector<double> aend((size_t)T, 0.0), awork((size_t)T, 0.0); std::vector<int> an((size_t)T, 0);
            double fwork = 0, fwait = 0; int fn = 0, fpan = 0;
            for (int t = 0; t < P->ntasks; ++t) {
                const CholTask& q = tk[(size_t)t];
                const double d0 = us(h[4 * (size_t)t]), d1 = us(h[4 * (size_t)t + 1]), d2 = us(h[4 * (size_t)t + 2]);
                if (q.kind == 8 && d2 > d1) { const int jc = (int)(h[4 * (size_t)t + 3] / 1000); if (jc > 0 && jc < T) { aend[(size_t)jc] = std::max(aend[(size_t)jc], d2); awork[(size_t)jc] += d2 - d1; ++an[(size_t)jc]; } }
                if (q.kind == 7) { fwork += d2 - d1; fwait += d1 - d0; ++fn; fpan += q.gate + 1 - q.round; }
            }
            std::fprintf(stderr, "[piqp_amd]  assembly tasks, block column: tasks / avg us / last end:");
            for (int j = 1; j < T; ++j) std::fprintf(stderr, " %d: %d / %.0f / %.0f |", j, an[(size_t)j], an[(size_t)j] ? awork[(size_t)j] / an[(size_t)j] : 0.0, aend[(size_t)j]);
            std::fprintf(stderr, "\n[piqp_amd]  far visits: %d (%d panels), avg work %.1f us, avg wait %.1f us\n", fn, fpan, fn ? fwork / fn : 0.0, fn ? fwait / fn : 0.0);
        }
        for (int k = 0; k + 1 < T; ++k) {
            double own_ready = 0, own_done = 0, help_done = 0, pan_ready = 0, pan_done = 0, bulk_first = 1e30, bulk_last = 0, bulk_work = 0, bulk_wait = 0;
            double r1[3] = {0, 0, 0}, r2[3] = {0, 0, 0}, dg[3] = {0, 0, 0};  // first / second panel row, diagonal tile of the first bulk column: drawn, inputs, done (latest half)
            int nb = 0;
            for (int t = 0; t < P->ntasks; ++t) {
                if (tk[(size_t)t].kind >= 6 || tk[(size_t)t].round != k) continue;
                const double d0 = us(h[4 * (size_t)t]), d1 = us(h[4 * (size_t)t + 1]), d2 = us(h[4 * (size_t)t + 2]);
                switch (tk[(size_t)t].kind) {
                case 0: help_done = std::max(help_done, d2); break;
                case 1: own_ready = d1; own_done = d2; break;
                case 2:
                    pan_ready = std::max(pan_ready, d1); pan_done = std::max(pan_done, d2);
                    if (tk[(size_t)t].a == 1) { r1[0] = std::max(r1[0], d0); r1[1] = std::max(r1[1], d1); r1[2] = std::max(r1[2], d2); }
                    if (tk[(size_t)t].a == 2) { r2[0] = std::max(r2[0], d0); r2[1] = std::max(r2[1], d1); r2[2] = std::max(r2[2], d2); }
                    break;
                default:
                    if (tk[(size_t)t].a == 1 && tk[(size_t)t].b == 1) { dg[0] = std::max(dg[0], d0); dg[1] = std::max(dg[1], d1); dg[2] = std::max(dg[2], d2); }
                    bulk_first = std::min(bulk_first, d1); bulk_last = std::max(bulk_last, d2); bulk_work += d2 - d1; bulk_wait += d1 - d0; ++nb; break;
                }
            }
            std::fprintf(stderr, "[piqp_amd]  round %2d: owner inputs %7.1f done %7.1f (helpers %7.1f) | panel inputs %7.1f done %7.1f | bulk %4d tiles: first start %7.1f last end %7.1f, avg work %5.1f avg wait %6.1f\n",
                         k, own_ready, own_done, help_done, pan_ready, pan_done, nb, nb ? bulk_first : 0.0, bulk_last, nb ? bulk_work / nb : 0.0, nb ? bulk_wait / nb : 0.0);
            std::fprintf(stderr, "[piqp_amd]    first panel row drawn %7.1f inputs %7.1f done %7.1f | second row %7.1f %7.1f %7.1f | diagonal tile of the first bulk column %7.1f %7.1f %7.1f\n", r1[0], r1[1],
                         r1[2], r2[0], r2[1], r2[2], dg[0], dg[1], dg[2]);
            if (k % 6 == 2) {
                const long long* q = h.data() + 4 * (size_t)P->ntasks + 64 * (size_t)k;
                std::fprintf(stderr, "[piqp_amd]    hand-over of round %d's first panel row to the crew of round %d: potrf steps seen by the row", k, k + 1);
                for (int u = 0; u < 8; ++u) std::fprintf(stderr, " %.1f", us(q[32 + u]));
                std::fprintf(stderr, " | slices sent");
                for (int u = 0; u < 8; ++u) std::fprintf(stderr, " %.1f", us(q[u]));
                std::fprintf(stderr, " last counted %.1f | slices seen by the next owner", us(q[27]));
                for (int u = 0; u < 8; ++u) std::fprintf(stderr, " %.1f", us(q[8 + u]));
                std::fprintf(stderr, " | its products done %.1f, potrf start %.1f end %.1f, helper 1 published %.1f\n", us(q[16]), us(q[17]), us(q[18]), us(q[19]));
                if (q[48]) std::fprintf(stderr, "[piqp_amd]    first panel row (half 0) of round %d: tile requested %.1f, operand rows there +%.1f, first stage in LDS +%.1f, K loop +%.1f, first look at the diagonal block +%.1f us\n", k,
                                        us(q[48]), (q[49] - q[48]) * 0.01, (q[50] - q[49]) * 0.01, (q[51] - q[50]) * 0.01, (q[32] - q[51]) * 0.01);
                if (q[40]) std::fprintf(stderr, "[piqp_amd]    bulk tile (5, 3) of round %d: first stage in LDS +%.1f, K loop +%.1f, stores issued +%.1f, drained +%.1f us\n", k, (q[41] - q[40]) * 0.01,
                                        (q[42] - q[41]) * 0.01, (q[43] - q[42]) * 0.01, (q[44] - q[43]) * 0.01);
            }
        }
    }
    return true;
}

// ------------------------------------------------------------------------------------------------
// Panel solve below the diagonal block:  A21 <- A21 * L11^-T            (Eigen LLT: solveInPlace<OnTheRight>)
//                                        A21 <- A21 * L11^-T(unit) D^-1 (dense/ldlt_no_pivot.hpp:345-346)
// One wave per 16 rows, no barriers after the operand pack is staged: the 16 x 128 row block stays in registers as eight tiles and the
// block substitution  X_k = T_k W_kk^T,  T_j -= X_k L_jk^T (j > k)  runs entirely on the matrix cores -- X_k comes out of the first
// product in exactly the layout the second one consumes (tile form above).  W_kk are the inverted 16 x 16 diagonal pieces, -L_jk the
// negated off-diagonal blocks, both prepared by potrf_block.
constexpr int TRSM_ROWS = 64;  // rows per workgroup (4 waves)
constexpr int TRSM_LDS_BYTES = PACK_BLOCKS * 256 * (int)sizeof(double);
// SUBST: the 16-column steps solve against the diagonal piece L_kk itself (tile_trsm_rt: 15 dependent rank-1 updates on the matrix cores, the
// arithmetic of a plain substitution) instead of multiplying by its explicit inverse W_kk (4 products).  The inverse is what the dense backend
// uses -- its H is positive definite and the rho = delta = 1e-10 gates of tests/dense_replay.py hold --; the fronts of the sparse backend are
// quasi-definite (pivots of +rho and -delta next to O(1) entries: unit-lower-triangular pieces with entries of 1e10), where the product with the
// inverse cost an order of magnitude of KKT residual on the last interior-point states of CONT-201 (tools/dbg_sparse_accuracy.py).
template <bool LDLT, bool SUBST = false>
__device__ __forceinline__ void trsm_panel_body(double* __restrict__ A_, int lda, int k0, int nb, int n, const double* __restrict__ pack_, const double* __restrict__ rdiag_, const int block_x)
{
    // (explicit global address space: the front kernels pass pointers read from a job record in memory -- generic pointers, FLAT accesses otherwise)
    typedef __attribute__((address_space(1))) double gd;
    gd* A = (gd*)A_;
    const gd* pack = (const gd*)pack_;
    const gd* rdiag = (const gd*)rdiag_;
    extern __shared__ __attribute__((aligned(16))) double Ps[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    {
        const __attribute__((address_space(1))) d2* src = reinterpret_cast<const __attribute__((address_space(1))) d2*>(pack);
        d2* dst = reinterpret_cast<d2*>(Ps);
        d2 v[PACK_BLOCKS * 128 / 256];
#pragma unroll
        for (int u = 0; u < PACK_BLOCKS * 128 / 256; ++u) v[u] = src[u * 256 + tid];
#pragma unroll
        for (int u = 0; u < PACK_BLOCKS * 128 / 256; ++u) dst[u * 256 + tid] = v[u];
    }
    const int row = k0 + nb + block_x * TRSM_ROWS + wave * 16 + i;
    const bool row_ok = row < n;
    gd* Ar = A + (row_ok ? row : 0) + (size_t)k0 * lda;
    d4 T[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * j + g + 4 * r;
            const bool ok = row_ok && c < nb;
            const double t = Ar[ok ? (size_t)c * lda : 0];
            T[j][r] = ok ? t : 0.0;
        }
    // SUBST: the eight factored diagonal pieces come from the front itself (strictly lower parts; beyond nb: nothing to solve) -- all requested here,
    // before the steps: loaded inside the step loop (behind its early exit) each piece was a memory round trip in front of its 15 dependent products
    d4 lkk_all[SUBST ? 8 : 1];
    if constexpr (SUBST) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = 16 * k + i, cc = 16 * k + g + 4 * r;
                const bool ok = rr < nb && cc < nb && rr > cc;
                const double t = A[ok ? (size_t)(k0 + rr) + (size_t)(k0 + cc) * lda : 0];
                lkk_all[k][r] = ok ? t : 0.0;
            }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (SUBST && 16 * k >= nb) break;  // (padding beyond a short panel: nothing to solve, nothing stored)
        d4 x = {0.0, 0.0, 0.0, 0.0};
        if constexpr (SUBST) {
            const d4 lkk = lkk_all[k];
            x = T[k];
            const d4 one = {1.0, 1.0, 1.0, 1.0};
            tile_trsm_rt<LDLT, false>(x, lkk, one, lane);
        } else {
            const d4 w = tile_load(Ps + (28 + k) * 256, lane);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) x = __builtin_amdgcn_mfma_f64_16x16x4f64(w[ks], T[k][ks], x, 0, 0, 0);
        }
        T[k] = x;
#pragma unroll
        for (int j = k + 1; j < 8; ++j) {
            const d4 nl = tile_load(Ps + (j * (j - 1) / 2 + k) * 256, lane);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) T[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(nl[ks], x[ks], T[j], 0, 0, 0);
        }
    }
    if (row_ok) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * j + g + 4 * r;
                if (c < nb) {
                    double v = T[j][r];
                    if (LDLT) v *= rdiag[k0 + c];
                    Ar[(size_t)c * lda] = v;
                }
            }
    }
}

template <bool LDLT>
__global__ __launch_bounds__(256) void k_trsm_panel(double* __restrict__ A, int lda, int k0, int nb, int n, const double* __restrict__ pack, const double* __restrict__ rdiag)
{
    trsm_panel_body<LDLT>(A, lda, k0, nb, n, pack, rdiag, (int)blockIdx.x);
}
// rows below the diagonal block of panel `panel` of every front of the list (blockIdx.y = front, blockIdx.x = strip of TRSM_ROWS rows)
__global__ __launch_bounds__(256) void k_trsm_panel_fronts(const FrontJob* __restrict__ jobs, int panel, const double* __restrict__ rdiag)
{
    const FrontJob j = jobs[blockIdx.y];
    int k, nb, rs;
    if (j.kind != 0 || !front_panel(j, panel, k, nb, rs) || (int)blockIdx.x * TRSM_ROWS >= rs) return;
    trsm_panel_body<true, true>(j.F, j.f, k, nb, j.f, j.pack, rdiag + j.first, (int)blockIdx.x);
}

// The same rows solved step by step BEHIND the factorisation of the diagonal block, which workgroup 0 of the SAME launch runs (k_potrf_trsm_fronts): potrf_block
// writes -L(j, k) and the factored piece L_kk through to L2 as its 16-column steps complete and counts them in cnt[k] (fresh counters per front and panel:
// nact - k increments say step k is there); every wave here keeps its 16 rows in registers and runs step k when it is -- the arithmetic of
// trsm_panel_body<LDLT, true>, product for product.  After the block's last step a wave is left with one 16 x 16 substitution instead of the whole panel
// (the substitution launch took 23 us behind the 32 us of the diagonal block, at every 128-column panel of the top of the tree).
// SIGNAL (k_front_panel_step: the trailing update runs in the same launch): the solved rows are written through and every wave counts itself in done[strip]
// when its rows have landed -- eight waves say the 128-row strip is there.
template <bool LDLT, bool SIGNAL = false>
__device__ __forceinline__ void front_trsm_follow(double* __restrict__ A_, int lda, int k0, int nb, int n, const double* __restrict__ pack, const double* __restrict__ rdiag,
                                                  const int* __restrict__ cnt, int strip, int* __restrict__ info, int kglobal, int* __restrict__ done = nullptr)
{
    typedef __attribute__((address_space(1))) double gd;
    gd* A = (gd*)A_;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int row = k0 + nb + strip * (POTRF_THREADS / 64) * 16 + wave * 16 + i;
    const bool row_ok = row < n;
    gd* Ar = A + (row_ok ? row : 0) + (size_t)k0 * lda;
    d4 T[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * j + g + 4 * r;
            const bool ok = row_ok && c < nb;
            const double t = Ar[ok ? (size_t)c * lda : 0];
            T[j][r] = ok ? t : 0.0;
        }
    const int nact = min(8, (nb + 15) >> 4);
    bool alive = true;
    int last = -1;  // SIGNAL: the step whose rows this wave has stored but not counted yet
    d4 rd[8];  // reciprocal pivots of this lane's columns, requested with the pieces of their step (asked for at the end they were a round trip behind the last step)
#pragma unroll
    for (int k = 0; k < 8; ++k) rd[k] = (d4){1.0, 1.0, 1.0, 1.0};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (16 * k >= nb) break;  // (padding beyond a short panel: nothing to solve, nothing stored)
        if constexpr (SIGNAL) {
            // the 16 solved columns of the step before have landed by now or shortly: count them (the trailing tiles of this launch run one K stage behind)
            if (last >= 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) addi_agent(done + 8 * strip + last, 1);
            }
        }
        if (alive) {
            unsigned spins = 0;
            while (__builtin_amdgcn_readfirstlane(ldi_agent(cnt + k)) < nact - k) {
                __builtin_amdgcn_s_sleep(4);  // (a grid of tight pollers slows the write-through traffic of the block they wait for)
                if (++spins > 20000000u) { alive = false; break; }  // (cannot happen with a healthy device; reported below)
            }
        }
        d4 lkk;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = 16 * k + i, cc = 16 * k + g + 4 * r;
            const double t = ld_agent(pack + (size_t)(28 + k) * 256 + (g + 4 * r) * 16 + i);
            lkk[r] = (rr < nb && cc < nb && rr > cc) ? t : 0.0;
        }
        d4 nl[8];
#pragma unroll
        for (int j = k + 1; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) nl[j][r] = ld_agent(pack + (size_t)(j * (j - 1) / 2 + k) * 256 + (g + 4 * r) * 16 + i);
        if (LDLT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int c = 16 * k + g + 4 * r; rd[k][r] = ld_agent(rdiag + k0 + (c < nb ? c : 0)); }
        }
        d4 x = T[k];
        const d4 one = {1.0, 1.0, 1.0, 1.0};
        tile_trsm_rt<LDLT, false>(x, lkk, one, lane);
        T[k] = x;
        if constexpr (SIGNAL) {
            if (row_ok) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * k + g + 4 * r;
                    if (c < nb) st_agent((double*)(Ar + (size_t)c * lda), LDLT ? x[r] * rd[k][r] : x[r]);
                }
            }
            last = k;
        }
#pragma unroll
        for (int j = k + 1; j < 8; ++j) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) T[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(nl[j][ks], x[ks], T[j], 0, 0, 0);
        }
    }
    if (!alive && lane == 0 && *info < 0) *info = kglobal;
    if constexpr (SIGNAL) {
        if (last >= 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) addi_agent(done + 8 * strip + last, 1);
        }
        return;
    }
    if (row_ok) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * j + g + 4 * r;
                if (c < nb) {
                    double v = T[j][r];
                    if (LDLT) v *= rd[j][r];
                    Ar[(size_t)c * lda] = v;
                }
            }
    }
}
// diagonal block (workgroup 0) and the rows below it (workgroups 1 ..: 128 rows each) of panel `panel` of every front of the list, one launch
__global__ __launch_bounds__(POTRF_THREADS) void k_potrf_trsm_fronts(const FrontJob* __restrict__ jobs, int panel, int* __restrict__ info, double* __restrict__ rdiag)
{
    const FrontJob j = jobs[blockIdx.y];
    int k, nb, rs;
    if (j.kind != 0 || !front_panel(j, panel, k, nb, rs)) return;
    int* cnt = j.cnt + FRONT_CNT_INTS * panel;
    if (blockIdx.x == 0) {
        potrf_diag_body<true>(j.F + k + (size_t)k * j.f, j.f, nb, j.first + k, info, rdiag, j.dvec + k, rs > 0 ? j.pack : nullptr, nullptr, nullptr, min(8, (nb + 15) >> 4), rs > 0 ? cnt : nullptr, true);
        return;
    }
    const int strip = (int)blockIdx.x - 1;
    if (strip * (POTRF_THREADS / 64) * 16 >= rs) return;
    front_trsm_follow<true>(j.F, j.f, k, nb, j.f, j.pack, rdiag + j.first, cnt, strip, info, j.first + k);
}

// A whole panel step of the fronts of a level in ONE launch (the top of an assembly tree: a handful of fronts, no one-workgroup panel fronts among them):
// workgroup 0 of a front factors the diagonal block, workgroups 1 .. nstrips solve the rows below it behind that factorisation (front_trsm_follow<SIGNAL>),
// the others are the half / quarter tiles of the trailing update (the body of k_syrk_half_fronts): a tile requests C, waits until the two 128-row strips it
// multiplies are there (done[ti], done[tj] at eight waves each) and then runs as before -- its launch, its C fetch and the update's prologue no longer sit
// behind the panel solve.  Producers always have the lower block index within their front: dispatched first, no deadlock whatever part of the grid is resident.
template <int SPLIT>
__global__ __launch_bounds__(POTRF_THREADS) void k_front_panel_step(const FrontJob* __restrict__ jobs, int panel, int nstrips, int* __restrict__ info, double* __restrict__ rdiag)
{
    const FrontJob j = jobs[blockIdx.y];
    int k, nb, rs;
    if (j.kind != 0 || !front_panel(j, panel, k, nb, rs)) return;
    int* cnt = j.cnt + FRONT_CNT_INTS * panel;
    int* done = cnt + 32;  // [strip][step]: waves of the strip whose 16 columns of that step have landed
    const int bx = (int)blockIdx.x;
    if (bx == 0) {
        potrf_diag_body<true>(j.F + k + (size_t)k * j.f, j.f, nb, j.first + k, info, rdiag, j.dvec + k, rs > 0 ? j.pack : nullptr, nullptr, nullptr, min(8, (nb + 15) >> 4), rs > 0 ? cnt : nullptr, true);
        return;
    }
    if (bx <= nstrips) {
        const int strip = bx - 1;
        if (strip * TS >= rs) return;
        front_trsm_follow<true, true>(j.F, j.f, k, nb, j.f, j.pack, rdiag + j.first, cnt, strip, info, j.first + k, done);
        return;
    }
    if (rs <= 0) return;
    const int T = (rs + TS - 1) / TS;
    const int b = (bx - 1 - nstrips) / SPLIT, h = (bx - 1 - nstrips) % SPLIT;
    if (b >= T * (T + 1) / 2) return;
    int ti = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= b) ++ti;
    while (ti * (ti + 1) / 2 > b) --ti;
    const int tj = b - ti * (ti + 1) / 2;
    if (ti * TS + (TS / SPLIT) * h >= rs) return;  // (nothing in this part)
    SyrkArgs a;
    a.n = rs; a.kdim = nb;
    a.A = j.F + (k + nb) + (size_t)k * j.f; a.lda = j.f; a.B = a.A; a.ldb = j.f; a.w = j.dvec + k;
    a.C = j.F + (k + nb) + (size_t)(k + nb) * j.f; a.ldc = j.f;
    a.unaligned = ((j.f & 1) || (reinterpret_cast<uintptr_t>(a.A) & 15)) ? 1 : 0;
    a.late_p[0] = done + 8 * ti; a.late_w[0] = POTRF_THREADS / 64; a.late_p[1] = done + 8 * tj; a.late_w[1] = POTRF_THREADS / 64;
    a.fuse_abort = cnt + 8;  // (only a wait that gave up sets it; zeroed with the counters before every factorisation)
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // a strip never arrived (bounded wait; cannot happen with a healthy device): the tile is left as it was and the factorisation is reported as failed
    if (!fused_tile<512, false, true, true, SPLIT == 4, true>(a, ti, tj, smem, h) && threadIdx.x == 0 && *info < 0) *info = j.first + k;
}

// one panel step of the partial LDLt of many fronts: diagonal blocks and panels (two launches whatever the number of fronts), then the trailing updates
static void front_attrs()
{
    static PerDeviceOnce attr_set;
    attr_set([&] {
    PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_diag_fronts), hipFuncAttributeMaxDynamicSharedMemorySize, POTRF_LDS_BYTES));
    PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_trsm_panel_fronts), hipFuncAttributeMaxDynamicSharedMemorySize, TRSM_LDS_BYTES));
    PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_trsm_fronts), hipFuncAttributeMaxDynamicSharedMemorySize, POTRF_LDS_BYTES));
    PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_front_panel_step<2>), hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS_BYTES));
    PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_front_panel_step<4>), hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS_BYTES));
    PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_syrk_lower_fronts<4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, SYRK_LDS_BYTES));
    PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_syrk_half_fronts<2>), hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS_BYTES));
    PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_syrk_half_fronts<4>), hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS_BYTES));
    });
}
void launch_front_diag_panels(const FrontJob* jobs, int njobs, int panel, int max_rows_below, int* info, double* rdiag, hipStream_t s, bool follow)
{
    if (njobs <= 0) return;
    front_attrs();
    static const bool no_follow = debug_token("front_no_follow") != nullptr;
    if (follow && !no_follow && panel < FRONT_CNT_PANELS) {
        // (workgroup 0 of a front is dispatched before its followers, which wait for nobody else: no deadlock whatever part of the grid is resident)
        hipLaunchKernelGGL(k_potrf_trsm_fronts, dim3(1 + div_up(std::max(max_rows_below, 0), (POTRF_THREADS / 64) * 16), njobs), dim3(POTRF_THREADS), POTRF_LDS_BYTES, s, jobs, panel, info, rdiag);
        PQ_HIP(hipGetLastError());
        return;
    }
    hipLaunchKernelGGL(k_potrf_diag_fronts, dim3(njobs), dim3(POTRF_THREADS), POTRF_LDS_BYTES, s, jobs, panel, info, rdiag);
    if (max_rows_below > 0) hipLaunchKernelGGL(k_trsm_panel_fronts, dim3(div_up(max_rows_below, TRSM_ROWS), njobs), dim3(256), TRSM_LDS_BYTES, s, jobs, panel, rdiag);
    PQ_HIP(hipGetLastError());
}
bool launch_front_panel_step(const FrontJob* jobs, int njobs, int panel, int max_rows_below, int* info, double* rdiag, hipStream_t s)
{
    static const bool off = debug_token("front_no_follow") != nullptr || debug_token("front_no_step") != nullptr || debug_token("front_updates_whole_tiles") != nullptr;
    if (off || njobs <= 0 || max_rows_below <= 0 || panel >= FRONT_CNT_PANELS) return false;
    const int T = div_up(max_rows_below, TS);
    if (T > 16 || (long long)njobs * T * (T + 1) > 256) return false;  // (the levels whose updates take half or quarter tiles anyway: launch_front_updates)
    front_attrs();
    const int nstrips = T;
    if ((long long)njobs * T * (T + 1) * 2 <= 256) hipLaunchKernelGGL(k_front_panel_step<4>, dim3(1 + nstrips + 2 * T * (T + 1), njobs), dim3(POTRF_THREADS), FUSED_LDS_BYTES, s, jobs, panel, nstrips, info, rdiag);
    else hipLaunchKernelGGL(k_front_panel_step<2>, dim3(1 + nstrips + T * (T + 1), njobs), dim3(POTRF_THREADS), FUSED_LDS_BYTES, s, jobs, panel, nstrips, info, rdiag);
    PQ_HIP(hipGetLastError());
    return true;
}
void launch_front_updates(const FrontJob* jobs, int njobs, int panel, int max_rows_below, hipStream_t s, int kind)
{
    if (njobs <= 0 || max_rows_below <= 0) return;
    front_attrs();
    const int T = div_up(max_rows_below, TS);
    static const bool no_half = debug_token("front_updates_whole_tiles") != nullptr;
    if (!no_half && (long long)njobs * T * (T + 1) <= 256) {  // few tiles (the top of the tree): two or four workgroups per tile, every workgroup on its own CU
        if ((long long)njobs * T * (T + 1) * 2 <= 256) hipLaunchKernelGGL(k_syrk_half_fronts<4>, dim3(2 * T * (T + 1), njobs), dim3(512), FUSED_LDS_BYTES, s, jobs, panel, kind);
        else hipLaunchKernelGGL(k_syrk_half_fronts<2>, dim3(T * (T + 1), njobs), dim3(512), FUSED_LDS_BYTES, s, jobs, panel, kind);
        PQ_HIP(hipGetLastError());
        return;
    }
    hipLaunchKernelGGL((k_syrk_lower_fronts<4, 4>), dim3(T * (T + 1) / 2, njobs), dim3(1024), SYRK_LDS_BYTES, s, jobs, panel, kind);
    PQ_HIP(hipGetLastError());
}

void launch_front_updates_multi(const FrontJob* jobs, int njobs, int max_update_rows, hipStream_t s)
{
    if (njobs <= 0 || max_update_rows <= 0) return;
    front_attrs();
    const int T = div_up(max_update_rows, TS);
    hipLaunchKernelGGL((k_syrk_multi_fronts<4, 4>), dim3(T * (T + 1) / 2, njobs), dim3(1024), SYRK_LDS_BYTES, s, jobs);
    PQ_HIP(hipGetLastError());
}

void launch_trsm_panel(bool ldlt, double* A, int lda, int k0, int nb, int n, const double* pack, const double* rdiag, hipStream_t s)
{
    const int rs = n - k0 - nb;
    if (rs <= 0) return;
    static PerDeviceOnce attr_set;
    attr_set([&] {
        PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_trsm_panel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, TRSM_LDS_BYTES));
        PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_trsm_panel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, TRSM_LDS_BYTES));
    });
    if (ldlt) hipLaunchKernelGGL(k_trsm_panel<true>, dim3(div_up(rs, TRSM_ROWS)), dim3(256), TRSM_LDS_BYTES, s, A, lda, k0, nb, n, pack, rdiag);
    else hipLaunchKernelGGL(k_trsm_panel<false>, dim3(div_up(rs, TRSM_ROWS)), dim3(256), TRSM_LDS_BYTES, s, A, lda, k0, nb, n, pack, rdiag);
    PQ_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// Triangular solves with one right-hand side, blocked by 128 columns.  Step j of the forward sweep:
// every row block r >= j first applies the contribution of block column j-1 (whose x is final), then
// block r == j solves its diagonal block.  The diagonal solve is a single-wave substitution (two rows
// per lane, v_readlane broadcast of the pivot).  unit_diag/div_d implement dense/ldlt_no_pivot.hpp:446-448.

// Single-wave substitution on a 128x128 triangular block held in LDS (Ls[c*(TB+1)+r] = L[r,c], zeros above the
// diagonal, identity padding).  Lane l owns rows l and l+64.  The block is consumed 16 columns at a time: the 32
// entries a lane needs are pulled into registers first, then 16 dependent steps run without touching LDS.
__device__ __forceinline__ void diag_solve_fwd(const double* __restrict__ Ls, const double* __restrict__ rd, int lane, double& b0, double& b1)
{
    constexpr int LD = 128 + 1;
#pragma unroll 1
    for (int kb = 0; kb < 8; ++kb) {
        double l0[16], l1[16], rr[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int k = kb * 16 + q;
            l0[q] = Ls[k * LD + lane];
            l1[q] = Ls[k * LD + lane + 64];
            rr[q] = rd[k];
        }
        if (kb < 4) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int k = kb * 16 + q;
                const double piv = readlane_d(b0, k) * rr[q];
                b0 = (lane == k) ? piv : b0 - l0[q] * piv;
                b1 -= l1[q] * piv;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int k = kb * 16 + q;
                const double piv = readlane_d(b1, k - 64) * rr[q];
                b1 = (lane + 64 == k) ? piv : b1 - l1[q] * piv;
            }
        }
    }
}
// L^T y = b on the block staged TRANSPOSED (Ls[k*(TB+1)+i] = L[k,i]): k descending, b_i -= L[k,i] y_k for i < k.  (Reading the
// untransposed staging with a lane stride of 129 doubles put 64 lanes on 16 bank pairs: the backward sweep took twice the forward one.)
__device__ __forceinline__ void diag_solve_bwd(const double* __restrict__ Ls, const double* __restrict__ rd, int lane, double& b0, double& b1)
{
    constexpr int LD = 128 + 1;
#pragma unroll 1
    for (int kb = 7; kb >= 0; --kb) {
        double l0[16], l1[16], rr[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int k = kb * 16 + q;
            l0[q] = Ls[k * LD + lane];       // unconditional: the staging holds zeros above the diagonal, and a predicated LDS read
            l1[q] = Ls[k * LD + lane + 64];  // per entry serialised 32 round trips per 16 columns (6 us per block instead of 1 us)
            rr[q] = rd[k];
        }
        if (kb >= 4) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {  // ascending trip count: a descending loop was left rolled and the register arrays went to scratch
                const int q = 15 - u;
                const int k = kb * 16 + q;
                const double piv = readlane_d(b1, k - 64) * rr[q];
                b1 = (lane + 64 == k) ? piv : b1 - l1[q] * piv;
                b0 -= l0[q] * piv;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int q = 15 - u;
                const int k = kb * 16 + q;
                const double piv = readlane_d(b0, k) * rr[q];
                b0 = (lane == k) ? piv : b0 - l0[q] * piv;
            }
        }
    }
}

constexpr int TB = 128;
constexpr int TRSV_LDS_BYTES = (TB * (TB + 1) + 3 * TB) * (int)sizeof(double);
constexpr int TRSV_P_LDS_BYTES = (TB * (TB + 1) + 3 * TB + 8 * 256 + 64 + TB) * (int)sizeof(double);  // + the eight inverted diagonal pieces, scratch of the diagonal step

// forward step j: row blocks r >= j subtract L[r, j-1] * x_{j-1}; block r == j then solves L_jj y = b.
// rdiag = reciprocal diagonal of L (nullptr: unit diagonal).  The diagonal workgroup issues the loads of
// L_jj before it touches x, so they overlap with the update.
__global__ __launch_bounds__(256) void k_trsv_fwd_step(const double* __restrict__ L, int ld, int n, double* __restrict__ x, const double* __restrict__ rdiag, int j)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* Ls = sm;                    // Ls[c * (TB+1) + r]
    double* xs = sm + TB * (TB + 1);    // x of block j-1
    double* bs = xs + TB;               // partial sums / rhs of block j
    double* rd = bs + TB;               // reciprocal pivots of block j
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = j + blockIdx.x;
    const int row0 = r * TB, nrows = min(TB, n - row0);
    const bool diag = (r == j);
    if (diag) {
        stage_lower_block<TB, TB + 1>(L + row0 + (size_t)row0 * ld, ld, nrows, Ls, tid);
        if (tid < TB) rd[tid] = (rdiag && tid < nrows) ? rdiag[row0 + tid] : 1.0;
    }
    const int row = tid & 127, half = tid >> 7;
    double mine = (half == 0 && row < nrows) ? x[row0 + row] : 0.0;
    if (j > 0) {
        const int c0 = (j - 1) * TB;
        if (tid < TB) xs[tid] = x[c0 + tid];
        __syncthreads();
        double sacc = 0.0;
        if (row < nrows) {
            const double* Lp = L + (row0 + row) + (size_t)(c0 + half * 64) * ld;
            double lv[64];
#pragma unroll
            for (int c = 0; c < 64; ++c) lv[c] = Lp[(size_t)c * ld];  // all 64 loads in flight before the first FMA
#pragma unroll
            for (int c = 0; c < 64; ++c) sacc += lv[c] * xs[half * 64 + c];
        }
        if (half == 1) bs[row] = sacc;
        __syncthreads();
        if (half == 0) mine -= (sacc + bs[row]);
        __syncthreads();
    }
    if (!diag) {
        if (half == 0 && row < nrows && j > 0) x[row0 + row] = mine;
        return;
    }
    if (half == 0) bs[row] = mine;
    __syncthreads();
    if (wave == 0) {
        double b0 = bs[lane], b1 = bs[lane + 64];
        diag_solve_fwd(Ls, rd, lane, b0, b1);
        if (lane < nrows) x[row0 + lane] = b0;
        if (lane + 64 < nrows) x[row0 + lane + 64] = b1;
    }
}

// backward step j (descending): row blocks r <= j subtract L[j+1 block, r block]^T x_{j+1}; block r == j then
// solves L_jj^T y = b.  The off-diagonal block is transposed through LDS (coalesced loads along the rows
// of L, conflict-free reads along its columns).
__global__ __launch_bounds__(256) void k_trsv_bwd_step(const double* __restrict__ L, int ld, int n, double* __restrict__ x, const double* __restrict__ rdiag, int j, int nblk)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* Ls = sm;
    double* xs = sm + TB * (TB + 1);
    double* bs = xs + TB;
    double* rd = bs + TB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = blockIdx.x;  // 0..j
    const int row0 = r * TB, nrows = min(TB, n - row0);
    const bool diag = (r == j);
    double mine = (tid < nrows) ? x[row0 + tid] : 0.0;
    if (j + 1 < nblk) {
        const int c0 = (j + 1) * TB, nc = min(TB, n - c0);
        if (tid < TB) xs[tid] = (tid < nc) ? x[c0 + tid] : 0.0;
        // tile[i * (TB+1) + c] = L[c0 + c, row0 + i]
        for (int i = wave; i < TB; i += 4) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = lane + 64 * h;
                Ls[i * (TB + 1) + c] = (i < nrows && c < nc) ? L[(c0 + c) + (size_t)(row0 + i) * ld] : 0.0;
            }
        }
        __syncthreads();
        if (tid < TB) {
            double sacc = 0.0;
#pragma unroll 16
            for (int c = 0; c < TB; ++c) sacc += Ls[tid * (TB + 1) + c] * xs[c];
            mine -= sacc;
        }
        __syncthreads();
    }
    if (!diag) {
        if (tid < nrows && j + 1 < nblk) x[row0 + tid] = mine;
        return;
    }
    if (tid < TB) { bs[tid] = mine; rd[tid] = (rdiag && tid < nrows) ? rdiag[row0 + tid] : 1.0; }
    stage_lower_block<TB, TB + 1, 256, true>(L + row0 + (size_t)row0 * ld, ld, nrows, Ls, tid);
    __syncthreads();
    if (wave == 0) {
        double b0 = bs[lane], b1 = bs[lane + 64];
        diag_solve_bwd(Ls, rd, lane, b0, b1);
        if (lane < nrows) x[row0 + lane] = b0;
        if (lane + 64 < nrows) x[row0 + lane + 64] = b1;
    }
}

// ------------------------------------------------------------------------------------------------
// Persistent triangular sweep: ONE launch per sweep, one workgroup per 128-row block, blocks chained through
// the solution values themselves instead of kernel boundaries (round 5; per-block flags until round 4 -- see below).  Block r waits only for the x_j it consumes next; the 128x128
// block of L it will multiply by x_j is already in registers when x_j arrives (its loads are issued before the
// wait), and its own diagonal block is staged in LDS at kernel start.  Hand-off follows the write-through recipe
// (cdna_hip_programming.md, Guideline 16 R1): x_r is stored with agent-scope relaxed atomics (sc1, L2
// write-through), every storing wave drains vmcnt, then one lane publishes flag[r]; consumers poll the flag
// relaxed and read x_j with agent-scope loads (sc1), so no acquire fence is needed.  Waits only ever target
// workgroups with a LOWER block index (dispatched earlier), and every spin is bounded (err flag on timeout).
typedef unsigned long long u64;
__device__ __forceinline__ void st_agent(double* p, double v)
{
    // (explicit global address space: these helpers only ever touch device memory, and inside the out-of-line roles of k_chol_persistent a pointer read from
    // the argument struct is a generic one to the compiler -- FLAT instructions, which tie the LDS waits to the global traffic)
    __hip_atomic_store((__attribute__((address_space(1))) u64*)reinterpret_cast<u64*>(p), (u64)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// round 5: the hand-over is the VALUE.  The solution of a sweep is written into a buffer of its own (ypoll: one per direction) that holds TRSV_SENT -- a NaN with a
// payload no arithmetic produces -- in every word not yet solved; a consumer polls the 128 words it needs (one per lane of its first two waves) until they are no
// sentinels.  One trip to the memory side instead of two (flag, then values), and the producer neither drains its stores nor publishes.  A block puts the
// sentinels back into the OTHER direction's buffer when it is done (stream order separates the sweeps), so the buffers are ready for the next solve.
constexpr u64 TRSV_SENT = 0x7ff8dead5eed0002ull;
__device__ __forceinline__ void st_agent_bits(double* p, u64 v)
{
    __hip_atomic_store((__attribute__((address_space(1))) u64*)reinterpret_cast<u64*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ONE XCD (round 5): a sweep of at most 32 block rows runs on the 32 compute units of XCD 0 alone -- the launch has eight times the workgroups, those that land
// elsewhere leave at once, the others take their block rows by ticket.  Producer and consumer then share an L2: the values are stored plainly (they stay in that
// L2) and polled there, instead of travelling to the memory side and back between two XCDs (1.8 us of the 3.5 us a block row took).
__device__ __forceinline__ void st_wg(double* p, double v)
{
    __hip_atomic_store((__attribute__((address_space(1))) u64*)reinterpret_cast<u64*>(p), (u64)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int xcc_id_of_wave()
{
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
    return v & 7;
}
__global__ void k_xcd_probe(int* __restrict__ count)
{
    if (threadIdx.x == 0) atomicAdd(count + xcc_id_of_wave(), 1);
}
// true when a launch of 8 x 32 workgroups puts exactly 32 on every XCD (eight XCDs served round robin: the mode this device is normally in)
bool probe_one_xcd_sweeps(hipStream_t s)
{
    int* cnt = nullptr;
    PQ_HIP(hipMalloc(&cnt, 8 * sizeof(int)));
    PQ_HIP(hipMemsetAsync(cnt, 0, 8 * sizeof(int), s));
    hipLaunchKernelGGL(k_xcd_probe, dim3(256), dim3(256), 0, s, cnt);
    int h[8];
    PQ_HIP(hipMemcpyAsync(h, cnt, sizeof(h), hipMemcpyDeviceToHost, s));
    PQ_HIP(hipStreamSynchronize(s));
    PQ_HIP(hipFree(cnt));
    for (int q = 0; q < 8; ++q) if (h[q] != 32) return false;
    return true;
}
__device__ __forceinline__ double ld_agent(const double* p)
{
    return __longlong_as_double((long long)__hip_atomic_load((__attribute__((address_space(1))) const u64*)reinterpret_cast<const u64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// ---- the diagonal step of the sweeps as ONE product with the inverse of the 128 x 128 diagonal block (rounds 4 / 5) -----------------------------------------
// x_r = L_rr^-1 b is what bounded a sweep: eight dependent 16-column groups, 3.5 of the 5.3 us a block row cost (profiles/r03: 170 us per sweep = 5 % of HBM).
// A product with the explicit inverse takes the dependence away, but with an inverse FORMED in double it costs accuracy exactly where it matters -- round 2
// measured 2x the residual in the median and 11x at worst on the rho = delta = 1e-10 states, and two iteration counts moved.  Here V is computed in double-double
// (106 bits) where the conditioning of the whole block enters and rounded to double once per factorisation: fl(V b) then carries the rounding of one product of
// 128 terms, and the residuals against the factor are 0.5 - 1.0 x the substitution's (profiles/r05_dense_sweeps.txt).  (Round 4 applied the double-double V with
// a compensated product in a 512-thread kernel that, it turned out, spilled 300 - 1200 registers; removed.)
struct ddn { double h, l; };
__device__ __forceinline__ ddn dd_two_sum(double a, double b) { const double s = a + b, bb = s - a; return {s, (a - (s - bb)) + (b - bb)}; }
__device__ __forceinline__ ddn dd_quick(double a, double b) { const double s = a + b; return {s, b - (s - a)}; }
__device__ __forceinline__ ddn dd_add(ddn a, ddn b) { ddn s = dd_two_sum(a.h, b.h); s.l += a.l + b.l; return dd_quick(s.h, s.l); }
__device__ __forceinline__ ddn dd_neg(ddn a) { return {-a.h, -a.l}; }
__device__ __forceinline__ ddn dd_mul_d(ddn a, double b) { const double p = a.h * b; const double e = __builtin_fma(a.h, b, -p); return dd_quick(p, __builtin_fma(a.l, b, e)); }
__device__ __forceinline__ ddn dd_mul(ddn a, ddn b) { const double p = a.h * b.h; double e = __builtin_fma(a.h, b.h, -p); e += a.h * b.l + a.l * b.h; return dd_quick(p, e); }
// V = L_rr^-1 of every 128-row diagonal block (identity beyond the matrix; UNIT: unit diagonal, the stored one is D), rounded to double, 8 columns of V per
// workgroup: block forward substitution in groups of 16 rows, X_g = W_g (E_g - sum_{h < g} L_gh X_h).  The sums over the earlier groups -- where the conditioning
// of the whole block enters -- run in double-double; W_g = L_gg^-1, the inverse of one 16 x 16 piece, in double (its error is eps times the condition of that piece).
// Thread = (row i of the group, column jq, half ks of the 16 terms of a product); the two halves sit 32 lanes apart in one wave.
constexpr int DDI_LDS_BYTES = 128 * 8 * (int)sizeof(ddn) + (8 * 256 + 36 * 256) * (int)sizeof(double);
__device__ __forceinline__ ddn dd_lane_sum32(ddn a)  // a + the value of the lane 32 further (both lanes get the sum)
{
    const ddn o = {__shfl_xor(a.h, 32), __shfl_xor(a.l, 32)};
    return dd_add(a, o);
}
template <bool UNIT>
__global__ __launch_bounds__(256) void k_block_inverse_dd(const double* __restrict__ L, int ld, int n, double* __restrict__ Vsq)
{
    extern __shared__ __attribute__((aligned(16))) double ddi_sm[];
    ddn* X = reinterpret_cast<ddn*>(ddi_sm);                  // [128][8]: rows of the block, this workgroup's 8 columns
    double* W = reinterpret_cast<double*>(X + 128 * 8);       // [8][16][16]: W[g][i + 16 k] = (L_gg^-1)(i, k)
    double* Lt = W + 8 * 256;                                 // the block's 16 x 16 tiles (g, h), h <= g: tile at 256 (g (g + 1) / 2 + h), entry (i, k) at [16 k + i]
    const int tid = threadIdx.x, J = blockIdx.y, r = blockIdx.x;  // (dispatch order: the column groups with the most rows below them first, for every block)
    const int row0 = r * 128, nrows = min(128, n - row0), j0 = 8 * J;
    const int g0 = J >> 1;  // the rows above the first column of this column group are zero
    // tiles with g0 <= h <= g; identity beyond the matrix, unit diagonal if UNIT.  (Read from global memory where they are used, every one of the up to 28 (g, h)
    // steps of a thread waited for a memory round trip.)
    {
        const int i = tid & 15, k = tid >> 4;
        double v[36];  // (every load requested before the first store)
#pragma unroll
        for (int g = 0; g < 8; ++g)
#pragma unroll
            for (int h = 0; h <= g; ++h) {
                const int gi = 16 * g + i, gk = 16 * h + k;
                double e = gi == gk ? 1.0 : 0.0;
                if (h >= g0 && gi < nrows && gk < nrows && !(UNIT && gi == gk)) e = gi >= gk ? L[(size_t)(row0 + gi) + (size_t)(row0 + gk) * ld] : 0.0;
                v[g * (g + 1) / 2 + h] = e;
            }
#pragma unroll
        for (int t = 0; t < 36; ++t) Lt[256 * t + 16 * k + i] = v[t];
    }
    for (int idx = tid; idx < 128 * 8; idx += 256) X[idx] = {0.0, 0.0};
    __syncthreads();
    auto lval = [&](int gi, int gk) -> double { return Lt[256 * ((gi >> 4) * ((gi >> 4) + 1) / 2 + (gk >> 4)) + 16 * (gk & 15) + (gi & 15)]; };  // gi >= gk >= 16 g0
    // ---- W_g: one thread per column of every 16 x 16 diagonal piece this workgroup needs
    if (tid < 128 && (tid >> 4) >= g0) {
        const int g = tid >> 4, k = tid & 15;
        double w[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double sacc = i == k ? 1.0 : 0.0;
#pragma unroll
            for (int m2 = 0; m2 < 16; ++m2)
                if (m2 < i) sacc -= (m2 >= k ? w[m2] : 0.0) * lval(16 * g + i, 16 * g + m2);
            w[i] = i < k ? 0.0 : sacc / lval(16 * g + i, 16 * g + i);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) W[g * 256 + i + 16 * k] = w[i];
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, jq = 2 * wave + ((lane >> 4) & 1), ks = lane >> 5;
    for (int g = g0; g < 8; ++g) {
        ddn ta = {0.0, 0.0}, tb = {0.0, 0.0};
        for (int h = g0; h < g; ++h) {
            double lrow[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) lrow[k] = lval(16 * g + i, 16 * h + 8 * ks + k);
#pragma unroll
            for (int k = 0; k < 8; k += 2) {  // (two sums: the chain through one double-double sum is what the thread waits for)
                ta = dd_add(ta, dd_mul_d(X[(16 * h + 8 * ks + k) * 8 + jq], lrow[k]));
                tb = dd_add(tb, dd_mul_d(X[(16 * h + 8 * ks + k + 1) * 8 + jq], lrow[k + 1]));
            }
        }
        const ddn tsum = dd_lane_sum32(dd_add(ta, tb));
        const ddn t0 = dd_add({(16 * g + i == j0 + jq) ? 1.0 : 0.0, 0.0}, dd_neg(tsum));
        // X_g = W_g T: T through LDS (the rows of X_g are free until now)
        if (ks == 0) X[(16 * g + i) * 8 + jq] = t0;
        __syncthreads();
        ddn xa = {0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int kk = 8 * ks + k;
            if (kk <= i) xa = dd_add(xa, dd_mul_d(X[(16 * g + kk) * 8 + jq], W[g * 256 + i + 16 * kk]));  // W_g is lower triangular
        }
        xa = dd_lane_sum32(xa);
        __syncthreads();
        if (ks == 0) X[(16 * g + i) * 8 + jq] = xa;
        __syncthreads();
    }
    // ---- out: the inverse rounded to double, a plain 128 x 128 column-major block (zeros above the diagonal)
    double* vs = Vsq + (size_t)r * 128 * 128;
    for (int idx = tid; idx < 128 * 8; idx += 256) {
        const int row = idx & 127, cl = idx >> 7, col = j0 + cl;
        const ddn v = X[row * 8 + cl];
        vs[(size_t)col * 128 + row] = row < col ? 0.0 : v.h + v.l;
    }
}
void launch_block_inverse_dd(bool unit, const double* L, int ld, int n, double* Vsq, hipStream_t s)
{
    if (n <= 0) return;
    static PerDeviceOnce attr_set;
    attr_set([&] {
        PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_block_inverse_dd<false>), hipFuncAttributeMaxDynamicSharedMemorySize, DDI_LDS_BYTES));
        PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_block_inverse_dd<true>), hipFuncAttributeMaxDynamicSharedMemorySize, DDI_LDS_BYTES));
    });
    const int nblk = div_up(n, 128);
    if (unit) hipLaunchKernelGGL(k_block_inverse_dd<true>, dim3(nblk, 16), dim3(256), DDI_LDS_BYTES, s, L, ld, n, Vsq);
    else hipLaunchKernelGGL(k_block_inverse_dd<false>, dim3(nblk, 16), dim3(256), DDI_LDS_BYTES, s, L, ld, n, Vsq);
    PQ_HIP(hipGetLastError());
}
size_t block_inverse_dd_doubles(int n) { return (size_t)div_up(n, 128) * 128 * 128; }

// (Round 1 had served a block row with several workgroups and measured it slower -- 203 / 222 us per sweep with one workgroup per row, 222 / 234 with two,
// 257 / 267 with four: with a 5 us diagonal step the extra hand-off cost more than the shared streaming saved.  With the diagonal step down to one product the
// streaming IS what a block row waits for, and round 5's helpers pay: see INV below.  Round 2 had tried the inverted diagonal block formed in double: 108 / 151 us
// per sweep, but the residuals on the rho = delta = 1e-10 states doubled and two iteration-parity tests moved.)
// (Round 3 tried folding W_g into the blocks of the diagonal step once per factorisation -- S_gj = W_g L_gj, so that x_g = (W_g b_g) - sum_j S_gj x_j needs one
// dependent 16 x 16 product per group instead of two: 700 instead of 850 cycles per group, 0.315 instead of 0.330 ms per solve, every accuracy gate held, but
// the dense mm_QAFIRO solve left the oracle's iteration count -- an LP at rho = delta = 1e-10 whose count the oracle keeps even when compiled with FMA
// contraction.  Removed: 1 % of the step is not worth a parity exception.)
// The diagonal step runs in eight groups of 16 columns: x_g = W_g b_g with the explicitly inverted 16 x 16 diagonal piece W_g of the factorisation
// (the conditioning level of the panel solve, which holds the accuracy gates of tests/dense_replay.py -- unlike a 128 x 128 inverse), one wave;
// then every thread takes its 8 columns of the group off the rows still to be solved.  Two LDS barriers per group instead of a 128-step
// dependent chain on one wave (4.2 us per block in round 1).  W16 == nullptr keeps that substitution (reciprocal pivots in rdiag).
// INV (round 5; Vinv = the rounded inverses of the 128 x 128 diagonal blocks, launch_block_inverse_dd): the diagonal step is ONE product x_r = V_r b_r; 512 threads,
// four per row; H helper workgroups per block row multiply the operand blocks of all producers but the owner's last two
template <bool FWD, bool INV>
__global__ __launch_bounds__(INV ? 512 : 256) void k_trsv_persistent(const double* __restrict__ L, int ld, int n, double* __restrict__ x, const double* __restrict__ rdiag,
                                                         int nblk, double* __restrict__ ysrc, double* __restrict__ yother, const double* __restrict__ dscale, int* __restrict__ err,
                                                         const double* __restrict__ W16, long long* __restrict__ ts, const double* __restrict__ Vinv, int* __restrict__ tkt, int tbase, int H, double* __restrict__ psrc, double* __restrict__ pother, double* __restrict__ ylsrc, double* __restrict__ ylother)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* Ls = sm;                    // own diagonal block, Ls[c * (TB+1) + r]
    double* xs = sm + TB * (TB + 1);    // x_j of the block being consumed
    double* bs = xs + TB;
    double* rd = bs + TB;
    double* Wg = rd + TB;               // eight inverted 16 x 16 diagonal pieces, column-major
    double* up = Wg + 8 * 256;          // scratch of the diagonal step
    __shared__ int ok_s;
    __shared__ int sync_w[4];           // diagonal step: [0] groups solved by the chain wave, [1], [2] groups applied by the helper waves
    __shared__ int ridx_s;
    const bool xq = tkt != nullptr;     // one-XCD mode: block rows by ticket among the workgroups on XCD 0
    if (xq && xcc_id_of_wave() != 0) return;
    if (threadIdx.x < 4) sync_w[threadIdx.x] = 0;
    if (threadIdx.x == 0) { ok_s = 1; ridx_s = xq ? atomicAdd(tkt, 1) - tbase : (int)blockIdx.x; }
    __syncthreads();
    // INV, H > 0: every block row has H helper workgroups in front of its owner (workgroup = ridx (1 + H) + role, the owner's role is H).  Helper h multiplies the
    // operand blocks of the producers t = h, h + H, ... < nsteps - 2 and hands its partial sums over (psrc, polled like the solution values); the owner keeps the last
    // two producers, the diagonal step and the hand-over.  One workgroup per block row streams its whole row of L through one compute unit -- 131 KB per step, ~2.5 us
    // -- which no chain shorter than that can hide; with the helpers a row's operand blocks arrive through up to eight compute units.
    const int role = (INV && H > 0) ? ridx_s % (1 + H) : 0;
    const bool helper = INV && H > 0 && role < H;
    if (ridx_s < 0 || ridx_s / ((INV && H > 0) ? 1 + H : 1) >= nblk) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // threads per row in the product phase, CW columns of the operand block each.  INV: four (512 threads, two waves per SIMD: one wave issues an instruction every
    // ~5 cycles, and a step of 64 multiply-adds, their LDS reads and the moves out of the AGPRs was 1.2 us of every block row); 256 registers each are enough for two
    // operand blocks of 32 columns in flight and the 32 entries of the inverse
    constexpr int NQ = INV ? 4 : 2;
    constexpr int NTH = 128 * NQ;
    constexpr int CW = TB / NQ;
    const bool act = true;
    const int ridx = (INV && H > 0) ? ridx_s / (1 + H) : ridx_s;  // position of the block in sweep order
    const int r = FWD ? ridx : nblk - 1 - ridx;
    const int row0 = r * TB, nrows = min(TB, n - row0);
    // stage the diagonal block (transposed for the backward sweep) and reciprocal pivots -- or, Vh != nullptr, its double-double inverse by diagonals (hi | lo: the
    // same 16 512 doubles), see k_block_inverse_dd
    if constexpr (INV) {
        // the rounded inverse of the block (128 x 128 column-major, zeros included): Ls[c * (TB + 1) + row] = V(row, c) for the forward sweep, V(c, row) for the
        // backward one (x = V^T b)
        const double* vs = Vinv + (size_t)r * TB * TB;
        for (int base = 0; base < (helper ? 0 : TB * TB); base += 8 * NTH) {
            double a[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = vs[base + q * NTH + tid];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int idx = base + q * NTH + tid, lo = idx & (TB - 1), hi = idx >> 7;  // entry (lo, hi) of V
                if (FWD) Ls[hi * (TB + 1) + lo] = a[q]; else Ls[lo * (TB + 1) + hi] = a[q];
            }
        }
        if (!helper) __syncthreads();  // (the owner of a row with one producer moves V into registers before any other barrier)
    } else stage_lower_block<TB, TB + 1, 256, !FWD>(L + row0 + (size_t)row0 * ld, ld, nrows, Ls, tid);
    if (tid < TB) rd[tid] = (rdiag && tid < nrows) ? rdiag[row0 + tid] : 1.0;
    if (W16 && !INV) {
#pragma unroll
        for (int u = 0; u < 8; ++u) Wg[u * 256 + tid] = W16[(size_t)r * 8 * 256 + u * 256 + tid];
    }
    const int row = tid & 127, half = (tid >> 7) & 1, quarter = tid >> 7;  // (256 threads: quarter == half)
    (void)half;
    // the right-hand side: x for the forward sweep; the forward sweep's solution (times 1 / D for LDLt) for the backward one
    double mine = 0.0;
    if (!helper && quarter == 0 && row < nrows) {
        if (FWD) mine = x[row0 + row];
        else { mine = yother[row0 + row]; if (dscale) mine *= dscale[row0 + row]; }
    }
    // INV: this thread's 64 entries of the inverse, V(row, 64 half + c) resp. V(64 half + c, row), move from LDS into registers when the block turns to its LAST
    // producer -- into the operand buffer that has no next block to receive (earlier they would sit next to two operand blocks in flight and spill)
    constexpr int LOADV = -2;
    auto load_v = [&](double (&vr)[CW]) {
#pragma unroll
        for (int c = 0; c < CW; ++c) vr[c] = Ls[(quarter * CW + c) * (TB + 1) + row];
        // (pinned here: left to itself the compiler sinks these reads to their use, behind the arrival of the last producer's values -- 1 000 cycles of LDS
        // traffic on the chain of every block row)
#pragma unroll
        for (int c = 0; c < CW; ++c) asm volatile("" : "+v"(vr[c]));
    };
    double acc = 0.0;
    const int nsteps = ridx;  // producers of this block row, in sweep order t = 0 .. nsteps - 1
    // operand block of step t (producer block j_t) into registers; two steps are kept in flight: the block for the next step is requested
    // before the wait for x_{j_t}, so its latency never sits between the arrival of x and the hand-off to the next block
    auto load_block = [&](int t, double (&lv)[CW]) {
        if (!act) return;
        const int j = FWD ? t : nblk - 1 - t;
        const int c0 = j * TB;
        const int nc = min(TB, n - c0);
        if (FWD) {  // rows of block r, columns of block j: L[row0+row, c0 + half*64 + c]
            const double* Lp = L + (row0 + row) + (size_t)(c0 + quarter * CW) * ld;
#pragma unroll
            for (int c = 0; c < CW; ++c) lv[c] = (row < nrows) ? Lp[(size_t)c * ld] : 0.0;
        } else {    // transposed: L[c0 + half*64 + c, row0+row] (column row0+row of L, contiguous in c)
            const double* Lp = L + (c0 + quarter * CW) + (size_t)(row0 + row) * ld;
            if (((ld & 1) == 0) && ((reinterpret_cast<uintptr_t>(L) & 15) == 0) && nc == TB && row < nrows) {
#pragma unroll
                for (int c = 0; c < CW; c += 2) {
                    const d2 t2 = *reinterpret_cast<const d2*>(Lp + c);
                    lv[c] = t2.x; lv[c + 1] = t2.y;
                }
            } else {
#pragma unroll
                for (int c = 0; c < CW; ++c) lv[c] = (row < nrows && quarter * CW + c < nc) ? Lp[c] : 0.0;
            }
        }
    };
    // workgroup barrier that waits for LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would wait for the operand block
    // requested for the NEXT step.  Everything exchanged between the waves inside the loop goes through LDS.
    auto lds_barrier = [] {
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
    };
    // consumes step t with the operand block lv; the request for the next step (into nxt) is issued right AFTER x_j has arrived, so that the
    // in-order return of x_j does not wait behind it
    auto consume = [&](int t, const double (&lv)[CW], double (&nxt)[CW], int t_next) -> bool {
        const int j = FWD ? t : nblk - 1 - t;
        const int c0 = j * TB;
        const int nc = min(TB, n - c0);
        if (INV && t_next == LOADV) load_v(nxt);
        double xv = 0.0;
        if (tid < TB && tid < nc) {
            unsigned spins = 0;
            // (the owners of a launch with seven helpers per block row all sit on one XCD: they hand over among themselves through that XCD's L2 -- ylsrc, stored
            // plainly -- while the helpers on the other XCDs read the copy that went to the memory side)
            const double* yp = (INV && ylsrc && !helper) ? ylsrc : ysrc;
            while (true) {
                xv = ld_agent(yp + c0 + tid);
                if ((u64)__double_as_longlong(xv) != TRSV_SENT) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 20000000u) { ok_s = 0; __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
        if (ts && tid == 0 && t + 1 == nsteps) ts[4 * r + 0] = wall_clock64();  // debugging aid: the last producer's values seen (by the first lane)
        if (t_next >= 0) load_block(t_next, nxt);
        if (tid < TB) xs[tid] = xv;
        lds_barrier();
        if (!ok_s) return false;
        if (act) {
#pragma unroll
            for (int c = 0; c < CW; ++c) acc += lv[c] * xs[quarter * CW + c];
        }
        lds_barrier();
        return true;
    };
    // a producer never arrived (bounded wait; cannot happen with a healthy device): this block's part of the solution becomes NaN, which the caller's
    // allFinite check (kkt_system.hpp:266,305) turns into a failed solve -- a wrong x is never returned as a success
    auto poison = [&] { if (tid < nrows) { st_agent(ysrc + row0 + tid, __longlong_as_double(0x7ff8000000000000LL)); if (!FWD) x[row0 + tid] = __longlong_as_double(0x7ff8000000000000LL); st_agent_bits(yother + row0 + tid, TRSV_SENT); } };
    // the solution of this block row: to the consumers (and, backward sweep, to the caller), and the sentinels back into the other direction's buffer
    auto publish = [&](int i, double v) {
        if (INV && ylsrc) { st_wg(ylsrc + row0 + i, v); st_wg(ylother + row0 + i, __longlong_as_double((long long)TRSV_SENT)); }
        if (xq) st_wg(ysrc + row0 + i, v); else st_agent(ysrc + row0 + i, v);
        if (!FWD) x[row0 + i] = v;
        st_agent_bits(yother + row0 + i, TRSV_SENT);
    };
    {
        double lvA[CW], lvB[CW];
        if (!INV && nsteps > 0) load_block(0, lvA);
        int q = 0;
        if constexpr (INV) {
            // x = V b (forward) / V^T b (backward): two threads per row, 64 terms each from registers, the right-hand side read from LDS at one address per read for
            // the whole wave, every read requested before the first multiplication
            auto inv_tail = [&](const double (&vr)[CW]) {
                __syncthreads();
                if (ts && tid == 0) ts[4 * r + 1] = wall_clock64();  // products done
                if (quarter > 0) Wg[(quarter - 1) * TB + row] = acc;
                __syncthreads();
                if (quarter == 0) bs[row] = mine - (((acc + Wg[row]) + Wg[TB + row]) + Wg[2 * TB + row]);
                __syncthreads();
                double vv[CW];
#pragma unroll
                for (int c = 0; c < CW; ++c) vv[c] = bs[quarter * CW + c];
                double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
                for (int c = 0; c < CW; c += 4) {
                    s0 = __builtin_fma(vr[c], vv[c], s0); s1 = __builtin_fma(vr[c + 1], vv[c + 1], s1);
                    s2 = __builtin_fma(vr[c + 2], vv[c + 2], s2); s3 = __builtin_fma(vr[c + 3], vv[c + 3], s3);
                }
                const double pq = (s0 + s1) + (s2 + s3);
                if (quarter > 0) Wg[(quarter - 1) * TB + row] = pq;  // (the partial sums of the right-hand side were read before the last barrier)
                __syncthreads();
                if (quarter == 0 && row < nrows) publish(row, ((pq + Wg[row]) + Wg[TB + row]) + Wg[2 * TB + row]);
                if (ts && tid == 0) { ts[4 * r + 2] = wall_clock64(); ts[4 * r + 3] = wall_clock64(); }
            };
            // ONE loop for both roles (every further inlined copy of a step costs registers at the joins): producers t0, t0 + stride, ... < tend
            const int first = H > 0 ? max(0, nsteps - 2) : 0;  // the owner's producers: first .. nsteps - 1
            if (helper && role >= first) return;              // (no producer of its own: the owner does not look at its partial sums either)
            const int t0 = helper ? role : first, stride = helper ? H : 1, tend = helper ? first : nsteps;
            const bool pre = !helper && H > 0 && t0 + 1 < tend;  // the owner of a row with helpers requests both its operand blocks at once
            if (t0 < tend) load_block(t0, lvA);
            if (pre) load_block(t0 + 1, lvB);
            // (the other direction's partial sums of this block row -- whose helpers are not the ones of this direction: all of them -- ready for the next sweep)
            if (!helper && H > 0 && quarter == 0)
                for (int h = 0; h < H; ++h) st_agent_bits(pother + ((size_t)r * H + h) * TB + row, TRSV_SENT);
            // the helpers' partial sums: asked for AFTER the owner's first step.  The last of them comes from the producer three block rows back, through a helper on
            // another XCD -- a good 2.5 us from that producer's hand-over to here, more than two block rows of the chain take: waited for up front, every third block
            // row stalled on them (the timeline's gaps went 1.5, 0.15, 0.15, 1.5, ... us)
            auto poll_partials = [&] {
                if (quarter != 0) return;
                constexpr int HMAX = 7;
                const int hl = min(H, first);  // helpers with producers of their own
                double pv[HMAX];
#pragma unroll
                for (int h = 0; h < HMAX; ++h) pv[h] = h < hl ? ld_agent(psrc + ((size_t)r * H + h) * TB + row) : 0.0;
                double ps = 0.0;
#pragma unroll
                for (int h = 0; h < HMAX; ++h) {
                    if (h < hl) {
                        unsigned spins = 0;
                        while ((u64)__double_as_longlong(pv[h]) == TRSV_SENT) {
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > 20000000u) { ok_s = 0; __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                            pv[h] = ld_agent(psrc + ((size_t)r * H + h) * TB + row);
                        }
                        ps += pv[h];
                    }
                }
                acc += ps;
            };
            auto next_of = [&](int tt) { const int tn = tt + stride; return tn >= tend ? (helper ? -1 : LOADV) : ((pre && tt == t0) ? -1 : tn); };
            bool fine = true, v_in_b = false;
            int t = t0;
            if (t >= tend && !helper) { __syncthreads(); load_v(lvA); }
            while (t < tend) {
                fine = consume(t, lvA, lvB, next_of(t));
                v_in_b = true;
                if (pre && t == t0) poll_partials();  // (pre: the only owners whose helpers have producers)
                t += stride;
                if (!fine || t >= tend) break;
                fine = consume(t, lvB, lvA, next_of(t));
                v_in_b = false;
                t += stride;
                if (!fine) break;
            }
            if (helper) {
                // (a failed wait leaves NaN partial sums: the owner's solution, and with it everything after, is NaN)
                __syncthreads();
                if (quarter > 0) Wg[(quarter - 1) * TB + row] = acc;
                __syncthreads();
                if (quarter == 0) st_agent(psrc + ((size_t)r * H + role) * TB + row, fine ? ((acc + Wg[row]) + Wg[TB + row]) + Wg[2 * TB + row] : __longlong_as_double(0x7ff8000000000000LL));
                return;
            }
            if (!fine) { poison(); return; }
            if (v_in_b) inv_tail(lvB); else inv_tail(lvA);
            return;
        } else {
            for (; q + 1 < nsteps; q += 2) {
                if (!consume(q, lvA, lvB, q + 1)) { poison(); return; }
                if (!consume(q + 1, lvB, lvA, q + 2 < nsteps ? q + 2 : -1)) { poison(); return; }
            }
            if (q < nsteps) { if (!consume(q, lvA, lvB, -1)) { poison(); return; } }
        }
    }
    __syncthreads();
    if (ts && tid == 0) ts[4 * r + 1] = wall_clock64();  // products done
    if (half == 1) bs[row] = acc;
    __syncthreads();
    if (half == 0) bs[row] = mine - (acc + bs[row]);
    if (W16) {
        // The diagonal step as a dataflow inside the workgroup, no barriers (round 2; two barriers per group of 16 cost 8 500 cycles per block,
        // more than the hand-off between the blocks).  Groups of 16 columns, gi = position in sweep order; every wave has one job:
        //   wave 0 (the chain)  x_g = W_g (b_g - L_{g,g-1} x_{g-1} - far_g - near_g): only the product with the group solved LAST and the
        //                       product with the inverted diagonal piece -- 16 x 16, lane = (row i, quarter q), the four partial sums of a row
        //                       added in q order over the quad by DPP.  One wave issues in order, so every instruction here is on the clock;
        //   wave 3 (near)       near_g = L_{g,g-3} x_{g-3} + L_{g,g-2} x_{g-2} for the group the chain reaches next (one chain step of slack);
        //   waves 1, 2 (far)    one row each: far_row += L[row, group] x_group for every group solved at least four groups before the row's own
        //                       (three chain steps of slack); running value in LDS after every group.
        // Monotonic LDS words: cprog = groups solved by the chain, hp[w] = groups applied by far wave w, np = groups whose near term is there
        // (one wave's LDS instructions execute in order, so a word written after the data is seen after the data).
        lds_vint* cprog = (lds_vint*)&sync_w[0];
        lds_vint* hp = (lds_vint*)&sync_w[1];
        lds_vint* np = (lds_vint*)&sync_w[3];
        double* far = rd;        // reciprocal pivots are not used on this path
        double* rt = up;         // the chain's 16-vector on its way from "one value per quad" to "four values per lane"
        double* nearv = up + 64; // [TB]
        auto quad_sum = [&](double part) {  // ((p0 + p1) + p2) + p3 in every lane of the quad (quad_perm broadcasts)
            const int plo = __double2loint(part), phi = __double2hiint(part);
#define PQ_QUAD_BC(K) __hiloint2double(__builtin_amdgcn_update_dpp(0, phi, (K) * 0x55, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0, plo, (K) * 0x55, 0xf, 0xf, false))
            const double rs = ((PQ_QUAD_BC(0) + PQ_QUAD_BC(1)) + PQ_QUAD_BC(2)) + PQ_QUAD_BC(3);
#undef PQ_QUAD_BC
            return rs;
        };
        if (wave == 0) {
            const int i = lane >> 2, q = lane & 3;
            double wv[4], l1[4] = {0.0, 0.0, 0.0, 0.0};
            double tv;
            {
                const int g = FWD ? 0 : 7;
                const double* Wp = Wg + g * 256;
#pragma unroll
                for (int t = 0; t < 4; ++t) { const int c = q + 4 * t; wv[t] = FWD ? Wp[c * 16 + i] : Wp[i * 16 + c]; }
                tv = bs[16 * g + i];
            }
#pragma unroll
            for (int gi = 0; gi < 8; ++gi) {
                const int g = FWD ? gi : 7 - gi;
                if (gi >= 2) {
                    // what the other waves took off these rows (they have had a whole chain step).  The words and the values come back from ONE
                    // round of LDS reads; reads execute in order, so values read after a word that says "done" are the finished ones.  The terms
                    // leave b in ascending column order (far, near, last group), like a plain substitution.
                    const int hw = (16 * g) >> 6;  // far wave that owns these rows
                    double fv, nv;
                    while (true) {
                        const int n_done = *np, h_done = hp[hw];
                        fv = far[16 * g + i]; nv = nearv[16 * g + i];
                        if (n_done >= gi && (gi < 4 || h_done >= gi - 3)) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (gi >= 4) tv -= fv;
                    tv -= nv;
                }
                if (gi >= 1) {
                    const int g1 = FWD ? g - 1 : g + 1;
                    double part = 0.0;
#pragma unroll
                    for (int t = 0; t < 4; ++t) part += l1[t] * xs[16 * g1 + q + 4 * t];
                    tv -= quad_sum(part);
                }
                if (q == 0) rt[i] = tv;
                wave_lds_sync();
                // ---- group gi + 1: W row, b, the L entries of its last product -- fetched while x_gi is in flight ----
                double wn[4] = {0.0, 0.0, 0.0, 0.0}, ln[4] = {0.0, 0.0, 0.0, 0.0}, tn = 0.0;
                if (gi < 7) {
                    const int gn = FWD ? g + 1 : g - 1;
                    const double* Wn = Wg + gn * 256;
#pragma unroll
                    for (int t = 0; t < 4; ++t) { const int c = q + 4 * t; wn[t] = FWD ? Wn[c * 16 + i] : Wn[i * 16 + c]; }
#pragma unroll
                    for (int t = 0; t < 4; ++t) ln[t] = Ls[(16 * g + q + 4 * t) * (TB + 1) + 16 * gn + i];
                    tn = bs[16 * gn + i];
                }
                // ---- x_gi = W r ----
                double part = 0.0;
#pragma unroll
                for (int t = 0; t < 4; ++t) part += wv[t] * rt[q + 4 * t];
                const double xi = quad_sum(part);
                if (q == 0) { xs[16 * g + i] = xi; bs[16 * g + i] = xi; }
                wave_lds_sync();
                if (lane == 0) *cprog = gi + 1;
                asm volatile("" ::: "memory");
                if (ts && lane == 0 && r == 1) ts[4 * nblk + gi] = wall_clock64();  // debugging aid: the groups of block 1
                tv = tn;
#pragma unroll
                for (int t = 0; t < 4; ++t) { wv[t] = wn[t]; l1[t] = ln[t]; }
            }
        } else if (wave == 3) {
            const int i = lane >> 2, q = lane & 3;
#pragma unroll 1
            for (int G = 2; G < 8; ++G) {
                const int g = FWD ? G : 7 - G;
                double n3 = 0.0;
                if (G >= 3) {
                    const int g3 = FWD ? g - 3 : g + 3;
                    while (*cprog < G - 2) __builtin_amdgcn_s_sleep(1);
                    double p = 0.0;
#pragma unroll
                    for (int t = 0; t < 4; ++t) { const int c = 16 * g3 + q + 4 * t; p += Ls[c * (TB + 1) + 16 * g + i] * xs[c]; }
                    n3 = quad_sum(p);
                }
                const int g2 = FWD ? g - 2 : g + 2;
                double l2[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) l2[t] = Ls[(16 * g2 + q + 4 * t) * (TB + 1) + 16 * g + i];
                while (*cprog < G - 1) __builtin_amdgcn_s_sleep(1);
                double p = 0.0;
#pragma unroll
                for (int t = 0; t < 4; ++t) p += l2[t] * xs[16 * g2 + q + 4 * t];
                const double nv = n3 + quad_sum(p);
                if (q == 0) nearv[16 * g + i] = nv;
                wave_lds_sync();
                if (lane == 0) *np = G;
                asm volatile("" ::: "memory");
            }
        } else {
            const int hrow = tid - 64;
            const int G = FWD ? (hrow >> 4) : 7 - (hrow >> 4);  // the row's own group, in sweep order
            double f = 0.0;
#pragma unroll 1
            for (int gi = 0; gi < 4; ++gi) {
                const int g = FWD ? gi : 7 - gi;
                while (*cprog < gi + 1) __builtin_amdgcn_s_sleep(1);
                if (gi <= G - 4) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) f += Ls[(16 * g + c) * (TB + 1) + hrow] * xs[16 * g + c];
                    far[hrow] = f;
                }
                wave_lds_sync();
                if (lane == 0) hp[wave - 1] = gi + 1;
                asm volatile("" ::: "memory");
            }
        }
        __syncthreads();
        if (ts && tid == 0) ts[4 * r + 2] = wall_clock64();  // diagonal block solved
        if (tid < nrows) publish(tid, bs[tid]);
        if (ts && tid == 0) ts[4 * r + 3] = wall_clock64();  // published
        return;
    }
    __syncthreads();
    if (wave == 0) {
        double b0 = bs[lane], b1 = bs[lane + 64];
        if (FWD) diag_solve_fwd(Ls, rd, lane, b0, b1);
        else diag_solve_bwd(Ls, rd, lane, b0, b1);
        if (lane < nrows) publish(lane, b0);
        if (lane + 64 < nrows) publish(lane + 64, b1);
    }
}

__global__ void k_mul_vec(int n, const double* __restrict__ d, double* __restrict__ x)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= d[i];
}

size_t trsv_poll_doubles(int n) { return 18 * (size_t)div_up(n, TB) * TB; }  // solution values of the two sweeps, partial sums of up to seven helpers per block row and sweep
__global__ void k_fill_bits(size_t n, u64 v, double* __restrict__ p)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) reinterpret_cast<u64*>(p)[i] = v;
}
void launch_trsv_poll_init(double* ypoll, int n, hipStream_t s)
{
    const size_t cnt = trsv_poll_doubles(n);
    if (cnt == 0) return;
    hipLaunchKernelGGL(k_fill_bits, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, cnt, TRSV_SENT, ypoll);
    PQ_HIP(hipGetLastError());
}

// LLT: L y = b, L^T x = y with rdiag = 1/diag(L).  LDLt: unit L, then x *= rdiag (= 1/D), then unit L^T.
// `ypoll` = trsv_poll_doubles(n) doubles of scratch, prepared ONCE by launch_trsv_poll_init (every sweep leaves the other direction's half prepared again), `err` one int;
// ypoll == nullptr, or more blocks than can be resident at once, falls back to one launch per block step.
void launch_trsv(const double* L, int ld, int n, double* x, const double* rdiag, bool ldlt, double* ypoll, int* ctl, const double* w16, hipStream_t s, long long* ts, const double* Vinv,
                 int xcd_seq)
{
    if (n <= 0) return;
    static PerDeviceOnce attr_set;
    attr_set([&] {
        PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_trsv_fwd_step), hipFuncAttributeMaxDynamicSharedMemorySize, TRSV_LDS_BYTES));
        PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_trsv_bwd_step), hipFuncAttributeMaxDynamicSharedMemorySize, TRSV_LDS_BYTES));
        PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_trsv_persistent<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, TRSV_P_LDS_BYTES));
        PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_trsv_persistent<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, TRSV_P_LDS_BYTES));
        PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_trsv_persistent<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, TRSV_P_LDS_BYTES));
        PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_trsv_persistent<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, TRSV_P_LDS_BYTES));
    });
    const int nblk = div_up(n, TB);
    const double* rd = ldlt ? nullptr : rdiag;
    const bool persistent = ypoll != nullptr && nblk <= 224;  // every block resident (one per CU) with room to spare
    if (persistent) {
        double* yf = ypoll;
        double* yb = ypoll + (size_t)nblk * TB;
        const double* dsc = ldlt ? rdiag : nullptr;  // the 1 / D between the sweeps of L D L^T: applied where the backward sweep reads its right-hand side
        // ctl: [0] error word, [1] / [2] tickets of the forward / backward sweeps in one-XCD mode (xcd_seq >= 0: the number of such solves on this ctl before)
        int* err = ctl;
        // helper workgroups (with the inverses only): as many per block row as leave the whole launch resident at once
        // (all waits target workgroups with lower indices, so the launch need not be resident as a whole: 8 x 32 workgroups at n = 4096)
        const int H = (Vinv && nblk >= 8) ? std::min(7, 256 / nblk - 1) : 0;
        double* pf = ypoll + 2 * (size_t)nblk * TB;
        double* pb = pf + 7 * (size_t)nblk * TB;
        // seven helpers: the owners are the workgroups 7, 15, 23, ... -- one XCD when the launch is dealt out round robin (xcd_seq >= 0: the probe said so)
        static const bool local_off = debug_token("sweep_local") && std::atoi(debug_token("sweep_local")) == 0;
        const bool local = H == 7 && xcd_seq >= 0 && !local_off;
        double* ylf = local ? pb + 7 * (size_t)nblk * TB : nullptr;
        double* ylb = local ? ylf + (size_t)nblk * TB : nullptr;
        const bool one_xcd = xcd_seq >= 0 && nblk <= 32 && H == 0;
        int* tf = one_xcd ? ctl + 1 : nullptr;
        int* tb = one_xcd ? ctl + 2 : nullptr;
        // (the two ticket words start every pair of sweeps from zero: a launch that is dealt more or fewer workgroups on XCD 0 than the probe at handle creation saw
        // -- another process on the device, another partition mode -- then only affects itself: surplus workgroups leave, missing ones end in the bounded waits'
        // error word; with counters running on across launches every later solve on the handle would have been out of step)
        if (one_xcd) PQ_HIP(hipMemsetAsync(ctl + 1, 0, 2 * sizeof(int), s));
        const int tbase = 0;
        const dim3 grid(one_xcd ? 8 * nblk : nblk * (1 + H));
        if (Vinv) hipLaunchKernelGGL((k_trsv_persistent<true, true>), grid, dim3(512), TRSV_P_LDS_BYTES, s, L, ld, n, x, rd, nblk, yf, yb, (const double*)nullptr, err, w16, ts, Vinv, tf, tbase, H, pf, pb, ylf, ylb);
        else hipLaunchKernelGGL((k_trsv_persistent<true, false>), grid, dim3(256), TRSV_P_LDS_BYTES, s, L, ld, n, x, rd, nblk, yf, yb, (const double*)nullptr, err, w16, ts, Vinv, tf, tbase, H, pf, pb, ylf, ylb);
        if (Vinv) hipLaunchKernelGGL((k_trsv_persistent<false, true>), grid, dim3(512), TRSV_P_LDS_BYTES, s, L, ld, n, x, rd, nblk, yb, yf, dsc, err, w16, (long long*)nullptr, Vinv, tb, tbase, H, pb, pf, ylb, ylf);
        else hipLaunchKernelGGL((k_trsv_persistent<false, false>), grid, dim3(256), TRSV_P_LDS_BYTES, s, L, ld, n, x, rd, nblk, yb, yf, dsc, err, w16, (long long*)nullptr, Vinv, tb, tbase, H, pb, pf, ylb, ylf);
    } else {
        for (int j = 0; j < nblk; ++j)
            hipLaunchKernelGGL(k_trsv_fwd_step, dim3(nblk - j), dim3(256), TRSV_LDS_BYTES, s, L, ld, n, x, rd, j);
        if (ldlt) hipLaunchKernelGGL(k_mul_vec, dim3(div_up(n, 256)), dim3(256), 0, s, n, rdiag, x);
        for (int j = nblk - 1; j >= 0; --j)
            hipLaunchKernelGGL(k_trsv_bwd_step, dim3(j + 1), dim3(256), TRSV_LDS_BYTES, s, L, ld, n, x, rd, j, nblk);
    }
    PQ_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// GEMV, "N" form:  part[ks][i] = alpha * sum_{c in slice ks} M[i,c] * (scale ? scale[c] : 1) * v[c]
// Thread per row pair, slices of the column range across blockIdx.y; partial sums are combined by
// k_reduce_partials in a fixed order (bitwise reproducible -- the reference's clone test needs it).
constexpr int GEMV_N_ROWS = 512;  // rows per block (256 threads x 2)
__global__ __launch_bounds__(256) void k_gemv_n_partial(int rows, int cols, const double* __restrict__ M, int ld, const double* __restrict__ v,
                                                        const double* __restrict__ scale, double alpha, int cols_per_slice, double* __restrict__ part)
{
    __shared__ double vs[256];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * GEMV_N_ROWS + 2 * tid;
    const int c_begin = blockIdx.y * cols_per_slice;
    const int c_end = min(cols, c_begin + cols_per_slice);
    double s0 = 0.0, s1 = 0.0;
    const bool vec_ok = ((ld & 1) == 0) && ((reinterpret_cast<uintptr_t>(M) & 15) == 0);
    for (int cb = c_begin; cb < c_end; cb += 256) {
        const int cn = min(256, c_end - cb);
        __syncthreads();
        if (tid < cn) vs[tid] = alpha * v[cb + tid] * (scale ? scale[cb + tid] : 1.0);
        __syncthreads();
        if (i + 1 < rows && vec_ok) {
            const double* Mp = M + i + (size_t)cb * ld;
#pragma unroll 4
            for (int c = 0; c < cn; ++c) {
                const d2 mv = *reinterpret_cast<const d2*>(Mp + (size_t)c * ld);
                s0 += mv.x * vs[c];
                s1 += mv.y * vs[c];
            }
        } else if (i < rows) {
            const double* Mp = M + i + (size_t)cb * ld;
            for (int c = 0; c < cn; ++c) {
                s0 += Mp[(size_t)c * ld] * vs[c];
                if (i + 1 < rows) s1 += Mp[1 + (size_t)c * ld] * vs[c];
            }
        }
    }
    if (i < rows) part[(size_t)blockIdx.y * rows + i] = s0;
    if (i + 1 < rows) part[(size_t)blockIdx.y * rows + i + 1] = s1;
}

// y[i] = (base ? base[i] : 0) + sum_s part[s][i].  32 rows x 8 slice groups per workgroup: group g adds the slices g, g + 8, ... in order, the
// eight group sums are added in group order -- a fixed summation tree (bitwise reproducible), 128 workgroups at n = 4096 instead of the 16 that
// made this trivial 4 MB reduction take 30 us (one thread walking up to 128 slices with dependent adds).
__global__ __launch_bounds__(256) void k_reduce_partials(int rows, int nslices, const double* __restrict__ part, const double* __restrict__ base, double* __restrict__ y)
{
    __shared__ double sh[8][33];
    const int ri = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + ri;
    double s = 0.0;
    if (i < rows)
        for (int k = g; k < nslices; k += 8) s += part[(size_t)k * rows + i];
    sh[g][ri] = s;
    __syncthreads();
    if (g == 0 && i < rows) {
        double t = base ? base[i] : 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += sh[q][ri];
        y[i] = t;
    }
}

int gemv_n_slices(int rows, int cols)
{
    // enough blocks to cover the chip (256 CUs x ~4) without making slices thinner than 32 columns
    const int row_blocks = div_up(rows, GEMV_N_ROWS);
    int want = div_up(1024, row_blocks > 0 ? row_blocks : 1);
    int maxs = div_up(cols, 32);
    int sl = want < maxs ? want : maxs;
    if (sl < 1) sl = 1;
    return sl;
}

int launch_gemv_n_partial(int rows, int cols, const double* M, int ld, const double* v, const double* scale, double alpha, double* part, hipStream_t s)
{
    if (rows <= 0 || cols <= 0) return 0;
    const int sl = gemv_n_slices(rows, cols);
    const int cps = div_up(cols, sl);
    const int sl_eff = div_up(cols, cps);
    hipLaunchKernelGGL(k_gemv_n_partial, dim3(div_up(rows, GEMV_N_ROWS), sl_eff), dim3(256), 0, s, rows, cols, M, ld, v, scale, alpha, cps, part);
    PQ_HIP(hipGetLastError());
    return sl_eff;
}

void launch_reduce_partials(int rows, int nslices, const double* part, const double* base, double* y, hipStream_t s)
{
    if (rows <= 0) return;
    hipLaunchKernelGGL(k_reduce_partials, dim3(div_up(rows, 32)), dim3(256), 0, s, rows, nslices, part, base, y);
    PQ_HIP(hipGetLastError());
}

// GEMV, "T" form: one wave per column.  out[j] = (alpha * dot(M[:,j], v) + beta * c[j]) * (sc ? sc[j] : 1)
__global__ __launch_bounds__(256) void k_gemv_t(int rows, int cols, const double* __restrict__ M, int ld, const double* __restrict__ v, double alpha,
                                                double beta, const double* __restrict__ c, const double* __restrict__ sc, double* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= cols) return;
    const double* Mp = M + (size_t)j * ld;
    double s0 = 0.0, s1 = 0.0;
    const bool vec_ok = ((ld & 1) == 0) && ((reinterpret_cast<uintptr_t>(M) & 15) == 0) && ((reinterpret_cast<uintptr_t>(v) & 15) == 0);
    if (vec_ok) {
        int i = 2 * lane;
        for (; i + 1 < rows; i += 128) {
            const d2 mv = *reinterpret_cast<const d2*>(Mp + i);
            const d2 xv = *reinterpret_cast<const d2*>(v + i);
            s0 += mv.x * xv.x;
            s1 += mv.y * xv.y;
        }
        if (i < rows) s0 += Mp[i] * v[i];
    } else {
        for (int i = lane; i < rows; i += 64) s0 += Mp[i] * v[i];
    }
    double sacc = s0 + s1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sacc += __shfl_down(sacc, off, 64);
    if (lane == 0) {
        double r = alpha * sacc;
        if (c) r += beta * c[j];
        if (sc) r *= sc[j];
        out[j] = r;
    }
}

void launch_gemv_t(int rows, int cols, const double* M, int ld, const double* v, double alpha, double beta, const double* c, const double* sc, double* out, hipStream_t s)
{
    if (cols <= 0) return;
    hipLaunchKernelGGL(k_gemv_t, dim3(div_up(cols, 4)), dim3(256), 0, s, rows, cols, M, ld, v, alpha, beta, c, sc, out);
    PQ_HIP(hipGetLastError());
}

void launch_symmetrize_upper(const double* Pu, int n, double* Pf, double* pdiag, hipStream_t s) { launch_symmetrize(Pu, n, false, n, Pf, pdiag, s); }
void launch_symmetrize(const double* A, int lda, bool from_lower, int n, double* Pf, double* pdiag, hipStream_t s)
{
    if (n <= 0) return;
    dim3 grid((n + 31) / 32, (n + 31) / 32);
    if (from_lower) hipLaunchKernelGGL(k_symmetrize_upper<true>, grid, dim3(256), 0, s, A, lda, n, Pf, pdiag);
    else hipLaunchKernelGGL(k_symmetrize_upper<false>, grid, dim3(256), 0, s, A, lda, n, Pf, pdiag);
    PQ_HIP(hipGetLastError());
}

// z_reg_inv = 1 / z_reg   (dense/kkt.hpp:78)
// dense_ldlt_no_pivot on the condensed KKT matrix (this library's extension: dense/kkt.hpp itself only calls Eigen::LLT).  The matrix is positive definite by construction,
// so a pivot that is not positive is a numerical breakdown of the factorisation -- LDLTNoPivot's own test (== 0, ldlt_no_pivot.hpp:307) is met by the reference's unblocked loop
// on degenerate problems and practically never by a blocked summation order, which then carries a NEGATIVE pivot on and hands the interior-point loop a wrong
// step (round 6: QBEACONF, QGROW15, QGROW22 ended MAX_ITER that way).  The first such column is reported like a failed LLT pivot: the solver regularises and factors again.
__global__ __launch_bounds__(256) void k_flag_nonpositive(int n, const double* __restrict__ rdiag, int* __restrict__ info)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !(rdiag[i] > 0.0)) atomicMin(reinterpret_cast<unsigned*>(info), (unsigned)i);  // (info = -1 = UINT_MAX while nothing has failed)
}
void launch_flag_nonpositive(int n, const double* rdiag, int* info, hipStream_t s)
{
    hipLaunchKernelGGL(k_flag_nonpositive, dim3(div_up(n, 256)), dim3(256), 0, s, n, rdiag, info);
}

__global__ void k_reciprocal(int n, const double* __restrict__ a, double* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = 1.0 / a[i];
}
void launch_reciprocal(int n, const double* a, double* out, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_reciprocal, dim3(div_up(n, 256)), dim3(256), 0, s, n, a, out);
    PQ_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// micro-benchmarks used by the measurement harness (peak fp64 MFMA issue rate, HBM copy rate)
// The peak is the best of several launch shapes (waves per SIMD x independent accumulator chains per wave): a register-resident MFMA stream is
// sensitive to both, and a "peak" that a product kernel beats is not one (the fixed 8 x 8 shape sustained 46 TFLOP/s where the
// assembly kernel reaches 53).  Operands differ per lane and per chain so that no two products are the same instruction on the same data.
template <int ACC>
__global__ __launch_bounds__(256) void k_mfma_f64_peak(int iters, double* out)
{
    d4 acc[ACC];
    double a[ACC], b[ACC];
#pragma unroll
    for (int q = 0; q < ACC; ++q) {
        acc[q] = (d4){0.0, 0.0, 0.0, 0.0};
        a[q] = 1.0 + (threadIdx.x + 64 * q) * 1e-9;
        b[q] = 1.0 - (threadIdx.x * 3 + q) * 1e-9;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < ACC; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc[q], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < ACC; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    if (s == 12345.678) out[0] = s;
}
// U independent 16-byte loads in flight per thread before the stores (grid-stride over blocks of U * gridDim * 256 elements)
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy_d2(size_t n2, const d2* __restrict__ in, d2* __restrict__ out)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n2; i += U * stride) {
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) v[u] = __builtin_nontemporal_load(&in[i + u * stride]);
            else v[u] = in[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) __builtin_nontemporal_store(v[u], &out[i + u * stride]);
            else out[i + u * stride] = v[u];
        }
    }
    for (; i < n2; i += stride) out[i] = in[i];
}

static int microbench_cus()
{
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
}

template <int ACC>
static double mfma_variant(int blocks, int iters, double* out, hipStream_t s, hipEvent_t e0, hipEvent_t e1)
{
    hipLaunchKernelGGL(k_mfma_f64_peak<ACC>, dim3(blocks), dim3(256), 0, s, 10, out);
    stream_wait(s);
    PQ_HIP(hipEventRecord(e0, s));
    hipLaunchKernelGGL(k_mfma_f64_peak<ACC>, dim3(blocks), dim3(256), 0, s, iters, out);
    PQ_HIP(hipEventRecord(e1, s));
    PQ_HIP(hipEventSynchronize(e1));
    float ms = 0;
    PQ_HIP(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)blocks * 4.0 * (double)iters * ACC * 2.0 * 16 * 16 * 4;
    return flops / (ms * 1e-3) * 1e-12;
}

double microbench_mfma_f64(int iters, hipStream_t s)
{
    DBuf<double> out(8);
    hipEvent_t e0, e1;
    PQ_HIP(hipEventCreate(&e0));
    PQ_HIP(hipEventCreate(&e1));
    const int cus = microbench_cus();
    const bool verbose = debug_token("microbench") != nullptr;
    double best = 0.0;
    for (int per_cu : {2, 4, 8}) {
        const int blocks = cus * per_cu;
        const int it = std::max(50, iters * 8 / per_cu / 4);
        const double v[3] = {mfma_variant<2>(blocks, it * 4, out.p, s, e0, e1), mfma_variant<4>(blocks, it * 2, out.p, s, e0, e1),
                             mfma_variant<8>(blocks, it, out.p, s, e0, e1)};
        for (int q = 0; q < 3; ++q) {
            if (verbose) fprintf(stderr, "[piqp_amd] mfma_f64 microbench: %d waves/SIMD x %2d chains: %.1f TFLOP/s\n", per_cu, 2 << q, v[q]);
            best = std::max(best, v[q]);
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return best;
}

__global__ void k_fill_spd_block(double* A, int n)
{
    const int r = threadIdx.x, c = blockIdx.x;
    if (r < n && c < n) A[r + (size_t)c * n] = (r == c) ? (double)n + 1.0 : 1.0 / (1.0 + (double)((r * 7 + c * 13) % 17));
}
double microbench_potrf_block(bool ldlt, int reps, long long* stamps64, hipStream_t s)
{
    const int n = 128;
    DBuf<double> A0((size_t)n * n), A((size_t)n * n), rdiag(n), dvec(n), pack(FACTOR_PACK_DOUBLES);
    DBuf<int> info(1);
    DBuf<long long> ts(64);
    hipLaunchKernelGGL(k_fill_spd_block, dim3(n), dim3(n), 0, s, A0.p, n);
    PQ_HIP(hipMemsetAsync(ts.p, 0, 64 * sizeof(long long), s));
    PQ_HIP(hipMemsetAsync(info.p, 0xFF, sizeof(int), s));
    hipEvent_t e0, e1;
    PQ_HIP(hipEventCreate(&e0));
    PQ_HIP(hipEventCreate(&e1));
    float total = 0.f;
    for (int r = 0; r < reps + 1; ++r) {
        PQ_HIP(hipMemcpyAsync(A.p, A0.p, A.bytes(), hipMemcpyDeviceToDevice, s));
        PQ_HIP(hipEventRecord(e0, s));
        launch_potrf_diag(ldlt, A.p, n, n, 0, info.p, rdiag.p, dvec.p, pack.p, nullptr, s, r == reps ? ts.p : nullptr);
        PQ_HIP(hipEventRecord(e1, s));
        PQ_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        PQ_HIP(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && r < reps) total += ms;
    }
    if (stamps64) PQ_HIP(hipMemcpy(stamps64, ts.p, 64 * sizeof(long long), hipMemcpyDeviceToHost));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return reps > 1 ? total / (reps - 1) * 1e3 : 0.0;
}

// testing aid: `reps` factorisations of the caller's block of order nb <= 128 (host, column-major, leading dimension 128) by the diagonal-block kernel;
// the results of the last one to the host, and the number of repetitions whose factor / pivots / pack differ from the first one's in any bit
int debug_potrf_block(bool ldlt, int nb, int reps, const double* A_host, double* L_host, double* rdiag_host, double* dvec_host, double* pack_host, int* info_host, hipStream_t s)
{
    const int n = 128;
    DBuf<double> A0((size_t)n * n), A((size_t)n * n), rdiag(n), dvec(n), pack(FACTOR_PACK_DOUBLES);
    DBuf<int> info(1);
    PQ_HIP(hipMemcpy(A0.p, A_host, A0.bytes(), hipMemcpyHostToDevice));
    std::vector<double> first, cur((size_t)n * n + 2 * n + FACTOR_PACK_DOUBLES);
    int differ = 0;
    for (int r = 0; r < reps; ++r) {
        PQ_HIP(hipMemcpyAsync(A.p, A0.p, A.bytes(), hipMemcpyDeviceToDevice, s));
        PQ_HIP(hipMemsetAsync(info.p, 0xFF, sizeof(int), s));
        PQ_HIP(hipMemsetAsync(rdiag.p, 0, rdiag.bytes(), s));
        PQ_HIP(hipMemsetAsync(dvec.p, 0, dvec.bytes(), s));
        PQ_HIP(hipMemsetAsync(pack.p, 0, pack.bytes(), s));
        launch_potrf_diag(ldlt, A.p, n, nb, 0, info.p, rdiag.p, dvec.p, pack.p, nullptr, s, nullptr);
        stream_wait(s);
        PQ_HIP(hipMemcpy(cur.data(), A.p, A.bytes(), hipMemcpyDeviceToHost));
        PQ_HIP(hipMemcpy(cur.data() + (size_t)n * n, rdiag.p, rdiag.bytes(), hipMemcpyDeviceToHost));
        PQ_HIP(hipMemcpy(cur.data() + (size_t)n * n + n, dvec.p, dvec.bytes(), hipMemcpyDeviceToHost));
        PQ_HIP(hipMemcpy(cur.data() + (size_t)n * n + 2 * n, pack.p, pack.bytes(), hipMemcpyDeviceToHost));
        if (r == 0) first = cur;
        else if (memcmp(first.data(), cur.data(), cur.size() * sizeof(double)) != 0) ++differ;
    }
    memcpy(L_host, cur.data(), (size_t)n * n * sizeof(double));
    memcpy(rdiag_host, cur.data() + (size_t)n * n, n * sizeof(double));
    memcpy(dvec_host, cur.data() + (size_t)n * n + n, n * sizeof(double));
    memcpy(pack_host, cur.data() + (size_t)n * n + 2 * n, FACTOR_PACK_DOUBLES * sizeof(double));
    PQ_HIP(hipMemcpy(info_host, info.p, sizeof(int), hipMemcpyDeviceToHost));
    return differ;
}

template <int U, bool NT>
static double copy_variant(int blocks, size_t n2, const d2* a, d2* b, int iters, hipStream_t s, hipEvent_t e0, hipEvent_t e1)
{
    hipLaunchKernelGGL((k_copy_d2<U, NT>), dim3(blocks), dim3(256), 0, s, n2, a, b);
    stream_wait(s);
    PQ_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_copy_d2<U, NT>), dim3(blocks), dim3(256), 0, s, n2, a, b);
    PQ_HIP(hipEventRecord(e1, s));
    PQ_HIP(hipEventSynchronize(e1));
    float ms = 0;
    PQ_HIP(hipEventElapsedTime(&ms, e0, e1));
    return 2.0 * (double)n2 * 16.0 * iters / (ms * 1e-3) * 1e-9;
}

// best of several shapes (workgroups per CU x loads in flight per thread x plain / non-temporal): read + write bytes per second
double microbench_hbm_copy(size_t bytes, int iters, hipStream_t s)
{
    const size_t n2 = bytes / 16;
    DBuf<d2> a(n2), b(n2);
    PQ_HIP(hipMemsetAsync(a.p, 1, n2 * 16, s));
    hipEvent_t e0, e1;
    PQ_HIP(hipEventCreate(&e0));
    PQ_HIP(hipEventCreate(&e1));
    const int cus = microbench_cus();
    const bool verbose = debug_token("microbench") != nullptr;
    double best = 0.0;
    for (int per_cu : {1, 2, 3, 4, 6, 8, 16}) {  // (fewer, longer-running workgroups won on MI355X: 5.5 TB/s at 4 per CU against 4.6 at 16-32)
        const int blocks = cus * per_cu;
        const double v[4] = {copy_variant<1, false>(blocks, n2, a.p, b.p, iters, s, e0, e1), copy_variant<2, false>(blocks, n2, a.p, b.p, iters, s, e0, e1),
                             copy_variant<1, true>(blocks, n2, a.p, b.p, iters, s, e0, e1), copy_variant<2, true>(blocks, n2, a.p, b.p, iters, s, e0, e1)};
        static const char* nm[4] = {"1 load", "2 loads", "1 nt load", "2 nt loads"};
        for (int q = 0; q < 4; ++q) {
            if (verbose) fprintf(stderr, "[piqp_amd] hbm copy microbench: %2d workgroups/CU, %-10s in flight: %.0f GB/s\n", per_cu, nm[q], v[q]);
            best = std::max(best, v[q]);
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return best;
}

}  // namespace dense
}  // namespace pq
