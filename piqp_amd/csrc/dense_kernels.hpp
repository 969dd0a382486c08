// piqp_amd/csrc/dense_kernels.hpp -- launchers of the dense-path gfx950 kernels (see dense_kernels.hip)
#pragma once

#include "common.hpp"

namespace pq {
namespace dense {

enum { EPI_ASSEMBLE = 0, EPI_SUBTRACT = 1, EPI_STORE = 2, EPI_SUBTRACT_POTRF = 3 };

struct SyrkArgs {
    int n = 0;     // C is n x n, lower triangle written
    int kdim = 0;  // inner dimension
    const double* A = nullptr; int lda = 0;  // row operand   (n x kdim, column-major)
    const double* B = nullptr; int ldb = 0;  // column operand (n x kdim, column-major)
    const double* w = nullptr;               // optional per-k scale applied to B
    double* C = nullptr; int ldc = 0;
    // EPI_ASSEMBLE extras (dense/kkt.hpp:144-151)
    const double* Pfull = nullptr; int ldp = 0;
    const double* x_reg = nullptr;
    const double* ATA = nullptr; int ldata = 0;
    double dinv = 0.0;
    int unaligned = 0;  // set by the launcher
    int ncol = 1 << 30; // only the columns [0, ncol) of C are written (the trailing updates of the sparse fronts with several panels)
    // tail balancing (set by the launcher): blocks [0, ...) of a split launch handle tile `tile_begin + b / k_split`,
    // K-slice `b % k_split`, and write raw 128x128 partial tiles to `part` instead of running the epilogue
    int tile_begin = 0;
    int k_split = 1;
    double* part = nullptr;
    int first_col_only = 0;  // 1: only the tiles (ti, 0) of the first 128-column strip (panel look-ahead)
    // XCD-aware tile order (set by the launcher for long inner dimensions): tile of linear block b = tile_order[b] = ti << 16 | tj.
    // Workgroups go to the eight XCDs round-robin, so the order hands every XCD a compact 8 x 8 patch of tiles: the 64 tiles it works on at
    // the same time share 16 operand panels in its own L2 instead of nearly all of them
    const int* tile_order = nullptr;
    // EPI_SUBTRACT_POTRF: tile (0,0) of the trailing matrix is the NEXT diagonal block; the workgroup that updates it keeps it in
    // LDS and factors it right away (the serial k_potrf_diag of the next panel disappears behind the rest of this launch)
    int fuse_nb = 0;            // order of the next diagonal block (<= 128)
    int fuse_kglobal = 0;       // global index of its first column
    int fuse_ldlt = 0;
    int* fuse_info = nullptr;
    double* fuse_rdiag = nullptr;
    double* fuse_dvec = nullptr;  // LDLT: D of the next panel (the per-k scale of the next trailing update), nullable
    double* fuse_pack = nullptr;
    // the next diagonal block's update is shared by nine workgroups: owner + helpers on other CUs (dense_kernels.hip fused_next_diag)
    int fuse_token = 0;             // launch-unique value the helpers publish in fuse_flags[role]
    int* fuse_flags = nullptr;      // >= 9 ints (one per helper role)
    double* fuse_scratch = nullptr; // 36 x 256 doubles: the helpers' tile blocks on their way to the owner's LDS
    int* fuse_cnt = nullptr;        // 8 monotonic counters: what potrf_block has published of its step k (see there); non-null = the workgroups of the first
                                    // tile column solve the next panel in this launch, following the factorisation of the diagonal block
    double* fuse_w16 = nullptr;   // inverted 16 x 16 diagonal pieces of the next diagonal block (8 x 256 doubles), kept for the sweeps; nullable
    int* fuse_xpub = nullptr;     // persistent launch only: eight counters, one per 16-column slice, that every strip wave of the first panel row bumps once
    const int* fuse_xcnt = nullptr; // ... and the counter the crew of the next diagonal block follows (its operand arrives slice by slice), with its base value
    int fuse_xwant = 0;
    double* fuse_side = nullptr;    // persistent launch only: the side copy of the panel being solved (same leading dimension and offsets as C)
    long long* fuse_tr2 = nullptr;  // debugging aid (PIQP_AMD_DEBUG=chol_trace): wall-clock stamps of the hand-over between the first panel row and the next crew
    long long* fuse_tr2n = nullptr;
    int* fuse_abort = nullptr;    // persistent launch only: a word any workgroup sets when a bounded wait gave up; the waits inside poll it
    // persistent launch, panel tiles: the tile itself (C) is ready before the two operand rows are -- it is requested first, and the workgroup waits for the
    // rows (counters late_p[q] reaching late_w[q]) while those loads are in flight
    const int* late_p[2] = {nullptr, nullptr};
    int late_w[2] = {0, 0};
    long long* fuse_ts = nullptr; // debugging aid (PIQP_AMD_DEBUG=fused_ts=<panel>): 96 clock stamps -- the workgroup that owns the next diagonal block, the first panel
                                  // workgroup, an ordinary tile
};

void launch_syrk(int epi, const SyrkArgs& args, hipStream_t s, double* split_ws = nullptr, size_t split_ws_doubles = 0);
size_t syrk_split_workspace_doubles(int n, int kdim);
void syrk_prepare(int n);  // allocates what launch_syrk(EPI_ASSEMBLE, n) would otherwise allocate on first use
void launch_assemble_no_g(int n, const double* Pf, const double* x_reg, const double* ATA, double dinv, double* C, hipStream_t s);
// diagonal block of order nb <= 128 at A: factor in place, reciprocal pivots to rdiag[kglobal..], D to dvec[0..nb) (LDLT, nullable), and the
// operand pack of the panel solve (FACTOR_PACK_DOUBLES doubles, nullable: inverted 16 x 16 diagonal pieces + negated off-diagonal blocks)
void launch_potrf_diag(bool ldlt, double* A, int lda, int nb, int kglobal, int* info, double* rdiag, double* dvec, double* pack, double* w16, hipStream_t s,
                       long long* ts = nullptr);  // w16 (nullable): the eight inverted 16 x 16 diagonal pieces of this block, 8 x 256 doubles (for launch_trsv)
// debugging aid: `reps` factorisations of a synthetic 128 x 128 block; average microseconds and the 64 shader-clock stamps of potrf_block
double microbench_potrf_block(bool ldlt, int reps, long long* stamps64, hipStream_t s);
// testing aid: the caller's block of order nb <= 128 (host, leading dimension 128) through the same kernel, `reps` times; outputs of the last repetition to the
// host (factor 128 x 128, reciprocal pivots 128, D 128, pack FACTOR_PACK_DOUBLES, info); returns the number of repetitions that differ from the first in any bit
int debug_potrf_block(bool ldlt, int nb, int reps, const double* A_host, double* L_host, double* rdiag_host, double* dvec_host, double* pack_host, int* info_host, hipStream_t s);
// rows k0 + nb .. n of the panel at column k0:  A21 <- A21 L11^-T (D^-1), with the pack written by the factorisation of L11
void launch_trsm_panel(bool ldlt, double* A, int lda, int k0, int nb, int n, const double* pack, const double* rdiag, hipStream_t s);
// the persistent factorisation (k_chol_persistent, dense_kernels.hip): every round after the first diagonal block and panel in one launch
bool chol_persistent_supported(int n);
bool chol_prepare(int n);      // builds the device task list of this size (allocates: call at create time); false = use the launch-per-panel path
size_t chol_flag_ints(int n);  // ints of flag storage, zeroed once at allocation
// fused assembly (round 4): the tiles of block columns >= 1 are assembled by tasks of the persistent launch (dense/kkt.hpp:140-160); block column 0 by
// launch_syrk_first_col before it.  m % 128 == 0, m >= 512 (chol_fused_supported).
void launch_pack_row_panels(const double* G, int n, int m, double* Gp, hipStream_t s);  // n % 128 == 0: row panel i of G (ld = n) -> Gp + i 128 m, ld = 128
struct CholAssembly {
    const double* GT = nullptr; int ldg = 0, m = 0; const double* zinv = nullptr;  // GT: packed as row panels (launch_pack_row_panels), ldg unused
    const double* Pfull = nullptr; int ldp = 0; const double* x_reg = nullptr; const double* ATA = nullptr; int ldata = 0; double dinv = 0.0;
    double* part = nullptr;  // chol_prepare_fused(n, m) doubles
};
int chol_debug_plan(int T, int mchunks, int* out6, int capacity_tasks);  // host-only: the task list (mchunks > 0: with the fused assembly), 6 ints per task; returns the task count
bool chol_fused_supported(int n, int m);
size_t chol_prepare_fused(int n, int m);  // builds the fused task list (allocates: create time); doubles of partial-sum workspace, 0 = not available
// the assembly of block column 0 alone (tiles (i, 0)), K split over `ks` workgroups per tile with the fixed-order reduce of the tail tiles; ws: T * ks * 128 * 128 doubles
void launch_syrk_first_col(const SyrkArgs& args, int ks, double* ws, hipStream_t s);
bool launch_chol_persistent(bool ldlt, double* A, double* side, int lda, int n, int* info, double* rdiag, double* dvec, double* pack2, double* w16, double* scratch, int* fuse_flags, int* fuse_cnt,
                            int token_base, int* flags, int gen, int fcount, hipStream_t s, const CholAssembly* fused = nullptr);
size_t trsv_poll_doubles(int n);  // scratch of the persistent sweeps (launch_trsv), to be prepared once with launch_trsv_poll_init
void launch_trsv_poll_init(double* ypoll, int n, hipStream_t s);
// w16: the inverted 16 x 16 diagonal pieces written by the factorisation (8 x 256 doubles per 128-column panel), nullptr: substitution only
// Vinv (nullable; persistent sweeps only): the inverses of the 128-row diagonal blocks written by launch_block_inverse_dd -- the diagonal step of a block row
// becomes one product with them instead of eight dependent 16-column groups
// ctl: 3 ints, zeroed by the owner: error word, tickets of the two sweeps.  xcd_seq >= 0 (only when probe_one_xcd_sweeps() said yes; sweeps of at most 32 block rows):
// the sweep runs on XCD 0 alone (launch_trsv zeroes the two ticket words on the stream before every pair of sweeps; the value of xcd_seq beyond its sign is no longer used)
void launch_trsv(const double* L, int ld, int n, double* x, const double* rdiag, bool ldlt, double* ypoll, int* ctl, const double* w16, hipStream_t s, long long* ts = nullptr,  // ts: debugging aid, 4 stamps per block of the forward sweep
                 const double* Vinv = nullptr, int xcd_seq = -1);
bool probe_one_xcd_sweeps(hipStream_t s);
void launch_block_inverse_dd(bool unit, const double* L, int ld, int n, double* Vsq, hipStream_t s);  // unit: unit lower triangle (LDLt), the stored diagonal is D.  Vsq: 128 x 128 doubles per block
size_t block_inverse_dd_doubles(int n);  // doubles of Vh (and of Vl)
int gemv_n_slices(int rows, int cols);
int launch_gemv_n_partial(int rows, int cols, const double* M, int ld, const double* v, const double* scale, double alpha, double* part, hipStream_t s);
void launch_reduce_partials(int rows, int nslices, const double* part, const double* base, double* y, hipStream_t s);
void launch_gemv_t(int rows, int cols, const double* M, int ld, const double* v, double alpha, double beta, const double* c, const double* sc, double* out, hipStream_t s);
void launch_symmetrize_upper(const double* Pu, int n, double* Pf, double* pdiag, hipStream_t s);
// ... of a matrix with leading dimension lda of which only ONE triangle is valid: the upper one, or (from_lower) the lower one
void launch_symmetrize(const double* A, int lda, bool from_lower, int n, double* Pf, double* pdiag, hipStream_t s);
void launch_reciprocal(int n, const double* a, double* out, hipStream_t s);
// dense_ldlt_no_pivot: *info <- the first column whose reciprocal pivot is not positive (unchanged if there is none or an earlier column has failed already)
void launch_flag_nonpositive(int n, const double* rdiag, int* info, hipStream_t s);
double microbench_mfma_f64(int iters, hipStream_t s);
double microbench_hbm_copy(size_t bytes, int iters, hipStream_t s);

constexpr int FACTOR_NB = 128;  // panel width of the blocked factorisation
constexpr int FACTOR_PACK_DOUBLES = 36 * 256;

// One front of a sparse assembly tree on the dense panel kernels: partial LDLt of the first w columns of the f x f column-major F, the
// Schur complement left in the trailing block.  The big fronts of one tree level go through launch_front_panels together, panel by panel.
struct FrontJob {
    double* F = nullptr;
    int f = 0, w = 0, first = 0;  // order, pivot columns, global index of the first pivot (rdiag / info)
    double* pack = nullptr;       // FACTOR_PACK_DOUBLES of scratch owned by this front for the duration of the level
    double* dvec = nullptr;       // w doubles (rounded up to FACTOR_NB), likewise: D of the front, panel p at dvec + p * FACTOR_NB
    int kind = 0;                 // 0: diagonal block + panel by launch_front_diag_panels; 1: the caller factors the (single, w <= FACTOR_NB) panel
                                  //    itself and leaves D in dvec -- only the trailing update runs here
    int multi = 0;                // 1: several panels and an update matrix -- the per-panel trailing updates write the remaining pivot columns only, the region
                                  //    [w, f)^2 receives every panel in one pass after the last one (launch_front_updates_multi)
    int* cnt = nullptr;           // FRONT_CNT_INTS x FRONT_CNT_PANELS counters of this front's panels (zeroed by the caller before every factorisation): per panel
                                  //    [0, 8) steps of the diagonal block, [8] a word only a wait that gave up sets, [32, 160) per 128-row strip below it (<= 16) and
                                  //    step: waves whose 16 solved columns have landed -- the panel rows follow the diagonal block inside one launch
                                  //    (k_potrf_trsm_fronts), the trailing tiles the rows, one K stage behind (k_front_panel_step)
};
constexpr int FRONT_CNT_PANELS = 16, FRONT_CNT_INTS = 160;
// follow: the jobs carry zeroed step counters (FrontJob::cnt)
void launch_front_diag_panels(const FrontJob* jobs_device, int njobs, int panel, int max_rows_below, int* info, double* rdiag, hipStream_t s, bool follow = false);
void launch_front_updates(const FrontJob* jobs_device, int njobs, int panel, int max_rows_below, hipStream_t s, int kind = -1);  // kind: only the fronts of that kind
// diagonal blocks, panel rows and trailing updates of panel `panel` in ONE launch, where the level is a handful of big fronts (kind 0 only, zeroed counters); false:
// not applicable -- the caller runs launch_front_diag_panels + launch_front_updates
void launch_front_updates_multi(const FrontJob* jobs_device, int njobs, int max_update_rows, hipStream_t s);  // the fronts with FrontJob::multi, after their last panel
bool launch_front_panel_step(const FrontJob* jobs_device, int njobs, int panel, int max_rows_below, int* info, double* rdiag, hipStream_t s);  // T -= L D L^T of panel `panel`, all fronts

}  // namespace dense
}  // namespace pq
