/*
 * piqp_amd.h -- C-ABI of the MI355X-native KKT backend for PIQP (libpiqp_amd.so).
 *
 * Drop-in boundary: these entry points are what a PIQP build would bind to replace the bodies of its
 * KKT plugin interface and of the KKTSystem shell that drives it.  Every function cites the reference
 * interface it replaces (paths relative to PIQP v0.6.2 include/piqp/).  INTEGRATION.md shows the
 * reference-side adapter class (a KKTSolverBase<T,I,PIQP_DENSE> subclass calling pq_kkt_*).
 *
 * Conventions
 *   - fp64 everywhere, column-major dense matrices, int32 index lists (common.hpp:38-39, typedefs.hpp:53-68).
 *   - Opaque handles own all device memory; nothing is allocated after *_create (the reference's tests
 *     enforce alloc-free factor/solve: fwd.hpp:44-52).
 *   - Return: >= 0 success (factor calls: 1 = factorised, 0 = numerical failure exactly where the
 *     reference returns false), < 0 = pq_status error.  Nothing throws across this boundary.
 *   - Vector arguments live where the handle's pointer mode says: PQ_MEM_HOST (default; the library
 *     stages them through its own pinned buffers -- this is the reference's calling convention) or
 *     PQ_MEM_DEVICE (already resident in HBM; no copies, calls are asynchronous on the handle's stream
 *     except where a scalar must come back to the host).
 *   - A handle is not thread-safe; distinct handles may be used concurrently (one HIP stream each),
 *     like independent solver copies in the reference (kkt_system.hpp:70-95).
 *   - There is no CPU fallback: if no HIP device is usable, *_create fails with PQ_ERR_HIP.
 */
#ifndef PIQP_AMD_H
#define PIQP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    PQ_OK = 0,
    PQ_ERR_INVALID = -1,     /* bad argument / size mismatch */
    PQ_ERR_HIP = -2,         /* HIP runtime error (see pq_last_error_string) */
    PQ_ERR_UNSUPPORTED = -3, /* kkt solver not supported (kkt_system.hpp:464-465,493-494) */
    PQ_ERR_NOMEM = -4
} pq_status;

typedef enum { PQ_MEM_HOST = 0, PQ_MEM_DEVICE = 1 } pq_mem;

/* settings.hpp:18-26 KKTSolver (+ the in-tree pivot-free LDLt of dense/ldlt_no_pivot.hpp as a second dense kernel) */
typedef enum {
    PQ_DENSE_CHOLESKY = 0,
    PQ_SPARSE_LDLT = 1,
    PQ_SPARSE_LDLT_EQ_COND = 2,
    PQ_SPARSE_LDLT_INEQ_COND = 3,
    PQ_SPARSE_LDLT_COND = 4,
    PQ_SPARSE_MULTISTAGE = 5,
    PQ_DENSE_LDLT_NO_PIVOT = 16,
    /* the two engines behind PQ_SPARSE_LDLT, selectable directly (not in the reference's enum): the reference's own up-looking elimination order on the
     * device (sparse/ldlt.hpp:101-218 bit for bit; PQ_SPARSE_LDLT picks it up to 8192 KKT rows and 4e7 flops per factorisation -- 3e9 in the condensed modes) and the
     * supernodal multifrontal LDLt (everything else) */
    PQ_SPARSE_LDLT_EXACT = 17,
    PQ_SPARSE_LDLT_MULTIFRONTAL = 18
} pq_kkt_solver;

/* kkt_fwd.hpp:23-29 KKTUpdateOptions */
enum { PQ_KKT_UPDATE_NONE = 0, PQ_KKT_UPDATE_P = 1, PQ_KKT_UPDATE_A = 2, PQ_KKT_UPDATE_G = 4 };

/* results.hpp:18-27 Status */
enum {
    PQ_SOLVED = 1,
    PQ_MAX_ITER_REACHED = -1,
    PQ_PRIMAL_INFEASIBLE = -2,
    PQ_DUAL_INFEASIBLE = -3,
    PQ_NUMERICS = -8,
    PQ_UNSOLVED = -9,
    PQ_INVALID_SETTINGS = -10
};

/* dense/data.hpp:22-51 -- the (Ruiz-scaled) problem data a backend reads.  Matrices are the TRANSPOSED
 * constraint matrices exactly as dense::Data stores them.  `mem` says where these arrays live. */
typedef struct {
    int n, p, m;
    const double *P_utri; /* n x n, upper triangle read */
    const double *AT;     /* n x p */
    const double *GT;     /* n x m */
    /* the remaining fields are only read by pq_kktsys_* (KKTSystem), never by pq_kkt_* (backend) */
    int n_h_l, n_h_u, n_x_l, n_x_u;
    const int *h_l_idx, *h_u_idx, *x_l_idx, *x_u_idx; /* finite-bound index lists, ascending */
    const double *x_b_scaling;                         /* n, NULL = all ones */
    int mem;                                           /* pq_mem */
} pq_dense_data;

/* sparse/data.hpp:26-54 -- CSC int32/fp64 (P_utri upper triangle; AT = A^T n x p; GT = G^T n x m) */
typedef struct {
    int n, p, m;
    const int *P_colptr, *P_rowind; const double *P_val;
    const int *AT_colptr, *AT_rowind; const double *AT_val;
    const int *GT_colptr, *GT_rowind; const double *GT_val;
    int n_h_l, n_h_u, n_x_l, n_x_u;
    const int *h_l_idx, *h_u_idx, *x_l_idx, *x_u_idx;
    const double *x_b_scaling;
    int mem; /* host only in this release */
} pq_sparse_data;

/* variables.hpp:19-105 Variables<T>: ten vectors (box vectors length n, compressed to the first
 * n_x_l / n_x_u entries exactly as the reference keeps them) */
typedef struct {
    double *x, *y, *z_l, *z_u, *z_bl, *z_bu, *s_l, *s_u, *s_bl, *s_bu;
} pq_vars;

/* settings.hpp:43-82 Settings<T> (same fields, same defaults via pq_settings_default) */
typedef struct {
    double rho_init, delta_init;
    double eps_abs, eps_rel;
    int check_duality_gap;
    double eps_duality_gap_abs, eps_duality_gap_rel;
    double infeasibility_threshold;
    double reg_lower_limit, reg_finetune_lower_limit;
    int reg_finetune_primal_update_threshold, reg_finetune_dual_update_threshold;
    int max_iter, max_factor_retires;
    int preconditioner_scale_cost, preconditioner_reuse_on_update, preconditioner_iter;
    double tau;
    int kkt_solver; /* pq_kkt_solver */
    int iterative_refinement_always_enabled;
    double iterative_refinement_eps_abs, iterative_refinement_eps_rel;
    int iterative_refinement_max_iter;
    double iterative_refinement_min_improvement_rate;
    double iterative_refinement_static_regularization_eps;
    double iterative_refinement_static_regularization_rel;
    int verbose, compute_timings;
} pq_settings;

/* results.hpp:45-89 Info<T> */
typedef struct {
    int status;
    int iter;
    double rho, delta, mu, sigma, primal_step, dual_step;
    double primal_res, primal_res_rel, dual_res, dual_res_rel;
    double primal_res_reg, primal_res_reg_rel, dual_res_reg, dual_res_reg_rel;
    double primal_prox_inf, dual_prox_inf;
    double prev_primal_res, prev_dual_res;
    double primal_obj, dual_obj, duality_gap, duality_gap_rel;
    int factor_retires;
    double reg_limit;
    int no_primal_update, no_dual_update;
    double setup_time, update_time, solve_time, kkt_factor_time, kkt_solve_time, run_time;
    int n_factor, n_solve, n_backend_solve; /* call counters (not in the reference) */
} pq_info;

void pq_settings_default(pq_settings *s);       /* settings.hpp:45-82 */
const char *pq_last_error_string(void);          /* thread-local text of the last < 0 return */
int pq_device_count(void);                       /* number of visible HIP devices (0 = library unusable) */
const char *pq_version(void);

/* ===================== KKT backend: replaces KKTSolverBase<T,I,MatrixType> ===================== */
typedef struct pq_kkt pq_kkt;

/* dense::KKT ctor, dense/kkt.hpp:39-55 (uploads P_utri/AT/GT, builds AT_A = AT*AT^T).
 * kkt_solver: PQ_DENSE_CHOLESKY (Eigen::LLT semantics, dense/kkt.hpp:82-83) or PQ_DENSE_LDLT_NO_PIVOT. */
int pq_kkt_create_dense(pq_kkt **out, const pq_dense_data *data, int kkt_solver, int device);
/* sparse::KKT ctor, sparse/kkt.hpp:51-70 (assemble KKT by mode, AMD, permute, symbolic, upload) for
 * kkt_solver = PQ_SPARSE_LDLT; MultistageKKT ctor, sparse/multistage_kkt.hpp:76-135 (arrow-structure
 * detection, block containers, AtA) for kkt_solver = PQ_SPARSE_MULTISTAGE */
int pq_kkt_create_sparse(pq_kkt **out, const pq_sparse_data *data, int kkt_solver, int device);
int pq_kkt_clone(const pq_kkt *k, pq_kkt **out); /* kkt_solver_base.hpp:28 clone() */
void pq_kkt_destroy(pq_kkt *k);                  /* kkt_solver_base.hpp:26 */
int pq_kkt_set_pointer_mode(pq_kkt *k, int mem); /* where vector arguments of the calls below live */
/* kkt_solver_base.hpp:30 update_data(data, options): re-reads the flagged matrices from `data`
 * (dense re-uploads all three on every call, see SURVEY.md section 7 last bullet) */
int pq_kkt_update_data_dense(pq_kkt *k, const pq_dense_data *data, int options);
int pq_kkt_update_data_sparse(pq_kkt *k, const pq_sparse_data *data, int options);
/* kkt_solver_base.hpp:32: returns 1 ok / 0 failed.  dense: llt.info()==Success (dense/kkt.hpp:83);
 * sparse: n == cols (sparse/kkt.hpp:104).  x_reg[n], z_reg[m] are consumed during the call. */
int pq_kkt_update_scalings_and_factor(pq_kkt *k, double delta, const double *x_reg, const double *z_reg);
/* kkt_solver_base.hpp:34 solve(): outputs caller-allocated (n, p, m) */
int pq_kkt_solve(pq_kkt *k, const double *rhs_x, const double *rhs_y, const double *rhs_z, double *lhs_x,
                 double *lhs_y, double *lhs_z);
/* kkt_solver_base.hpp:37 z = alpha * P * x */
int pq_kkt_eval_P_x(pq_kkt *k, double alpha, const double *x, double *z);
/* kkt_solver_base.hpp:39 zn = alpha_n * A * xn, zt = alpha_t * A^T * xt */
int pq_kkt_eval_A_xn_and_AT_xt(pq_kkt *k, double alpha_n, double alpha_t, const double *xn, const double *xt,
                               double *zn, double *zt);
/* kkt_solver_base.hpp:41 zn = alpha_n * G * xn, zt = alpha_t * G^T * xt */
int pq_kkt_eval_G_xn_and_GT_xt(pq_kkt *k, double alpha_n, double alpha_t, const double *xn, const double *xt,
                               double *zn, double *zt);
int pq_kkt_print_info(pq_kkt *k);                                   /* kkt_solver_base.hpp:43 */
int pq_kkt_synchronize(pq_kkt *k);                                  /* wait for the handle's stream (everything a later call of the handle depends on is ordered behind it on the device) */
void *pq_kkt_stream(pq_kkt *k);                                     /* hipStream_t of the handle */
/* test hooks: dense/kkt.hpp:134 internal_kkt_mat() and the factor; copy n*n doubles to HOST memory */
int pq_kkt_internal_kkt_mat(pq_kkt *k, double *out_host);
int pq_kkt_internal_factor(pq_kkt *k, double *out_host);
int pq_kkt_dims(const pq_kkt *k, int *n, int *p, int *m);
/* sparse_multistage only: what MultistageKKT::print_info reports (multistage_kkt.hpp:385-393).  Writes up to
 * `capacity` rows of (start, diag_size, off_diag_size) -- the last row is the arrow corner block -- and
 * returns the number of blocks (call with out_host = NULL to size the buffer). */
int pq_kkt_multistage_block_info(pq_kkt *k, int *out_host, int capacity);
/* sparse backends: figures of the symbolic analysis for roofline arithmetic.  out = { N, nnz(PKPt), nnz(L) below
 * the diagonal, supernodes, tree levels, workgroup subtrees, max front order, factorisation flops sum_j (c_j^2 + 3 c_j) } */
int pq_kkt_sparse_stats(pq_kkt *k, double out[8]);
/* ---- stage-partitioned execution of ONE KKT system over several processes, one GPU each (BASELINE configs[4]; the
 * reference has no counterpart, its multistage backend is a serial recurrence, multistage_kkt.hpp:1289-1347,1726-1814).
 * Every process creates the same backend on the same data and calls pq_kkt_partition with its rank.  Vectors stay
 * replicated; the assembly-tree work of factor / solve is split: each rank eliminates the subtrees it owns (for a
 * multistage chain: a contiguous range of stages), the update matrices of the subtree roots are summed across ranks,
 * and the few supernodes above them are eliminated by every rank on identical data.  The library never talks to the
 * network itself: when data has to cross ranks it synchronises its stream and calls `exchange(user, which)`, and the
 * caller performs the collective on the buffer it registered (torch.distributed over RCCL in piqp_amd/dist.py):
 *   which = 0  all-reduce(sum) of buf_factor [sizes[0] doubles]   once per factorisation
 *   which = 1  all-reduce(sum) of buf_forward [sizes[1] doubles]  once per backend solve
 *   which = 2  all-gather in place of buf_gather [world x sizes[2] doubles, this rank's chunk at rank * sizes[2]]
 * The callback returns 0 on success; it must return only when the result is visible to the handle's stream.
 * Supported by the sparse_ldlt* backends and by sparse_multistage when it runs on the tree engine. */
typedef int (*pq_exchange_fn)(void *user, int which);
int pq_kkt_partition(pq_kkt *k, int rank, int world, long long sizes_out[3]);
int pq_kkt_set_exchange(pq_kkt *k, pq_exchange_fn exchange, void *user, double *buf_factor, double *buf_forward,
                        double *buf_gather);
/* Native transport (the north star's "RCCL over xGMI"): instead of a callback the library creates its own communicator and enqueues the three
 * collectives -- ncclAllReduce(sum, fp64) for which = 0, 1 and ncclAllGather for which = 2 -- on the handle's stream, right behind the kernels that
 * pack its own exchange buffers: no stream drain, no host round trip.  Rank 0 obtains the 128-byte id (pq_rccl_unique_id = ncclGetUniqueId) and
 * ships it to the other ranks by any means (MPI_Bcast, a file, torch.distributed.broadcast_object_list in piqp_amd/dist.py); then EVERY rank calls
 * pq_kkt_set_comm_rccl after pq_kkt_partition with the rank / world it partitioned with (collective: ncclCommInitRank).  One process per GPU.
 * librccl is loaded with dlopen at the first of these calls; single-GPU use never needs it. */
int pq_rccl_unique_id(unsigned char id_out[128]);
int pq_kkt_set_comm_rccl(pq_kkt *k, const unsigned char id[128], int rank, int world);
/* SURVEY 8(e) row 2 (reference: the stage-parallel loops of sparse/multistage_kkt.hpp:856-993, 1036-1218, 1529-1643): with a stage partition of a sparse_ldlt
 * (KKT_FULL) backend, (a) the per-factorisation value assembly refreshes only the diagonal entries of the fronts this rank factors, and (b) the refinement
 * residual err = rhs - K lhs (kkt_system.hpp:507-536) is evaluated only on the rows this rank's part of the next solve reads; ||err||_inf crosses the ranks in
 * ONE collective per refinement step:
 *   which = 3  all-reduce(MAX) of buf_norm [1 double]
 * through the same callback (register buf_norm after pq_kkt_set_exchange) or, with the native transport, ncclAllReduce(max) on the library's own buffer.  No halo
 * exchange is needed: the solution vectors are replicated (which = 2 gathers them), so every rank reads what its rows touch.  Without a registered buffer, for the
 * condensed KKT modes, or with PIQP_AMD_DEBUG=replicated_residual the residual is evaluated on every row by every rank as before.  The results are bitwise those of
 * the replicated evaluation (tests/test_partition.py).  pq_kkt_sharded_calls: out[0] = sharded residual evaluations so far, out[1] = rows in this rank's share.
 * Condensed KKT modes and sparse_multistage (tree engine): a partitioned handle re-evaluates, per factorisation, only the values of the fronts it factors (P entries,
 * delta^-1 A^T A entries, G^T W G product terms, diagonal shifts selected by destination front); there pq_kkt_sharded_calls reports out[0] = such assemblies so far,
 * out[1] = source entries this rank evaluates (a partition of one rank selects all of them).
 * Round 5, the solve side of the condensed modes and of sparse_multistage (tree engine; reference: the right-hand-side fold and the recovery of the eliminated
 * multipliers, sparse/kkt.hpp:113-175, multistage_kkt.hpp:234-287): the refinement residual is sharded there as well -- x rows and rows of the block that stays in
 * the system by the rank that eliminates them, rows of an eliminated block by every rank that eliminates an x column they touch.  A backend solve on that residual
 * folds it into this rank's x rows only and recovers the eliminated multipliers on those constraint rows only; at the end of a pq_kkt_system_solve that took at
 * least one refinement step the eliminated multipliers cross the ranks once, each row from its owner rank (which = 2 again; sizes_out[2] covers both uses).
 * pq_kkt_sharded_solve_calls: out[0] sharded residual evaluations, [1] rows of the residual in this rank's share (of n + p + m), [2] backend solves that folded /
 * recovered on this rank's rows only, [3] gathers of the multipliers, [4] x rows folded per such solve, [5] constraint rows recovered per such solve. */
int pq_kkt_set_exchange_norm(pq_kkt *k, double *buf_norm);
int pq_kkt_sharded_calls(pq_kkt *k, int out[2]);
int pq_kkt_sharded_solve_calls(pq_kkt *k, int out[6]);
/* test hook (reference-order engine, PQ_SPARSE_LDLT_EXACT): the factor as sparse/ldlt.hpp:24-37 holds it.  what = 0 nnz(L) | 1 L_cols[N + 1] | 2 L_ind | 3 L_vals |
 * 4 D | 5 D_inv | 6 values of P K P' (CSC order of pq_sparse_kkt_symbolic's PKp / PKi_rows) | 7 perm | 8 .. 17 timelines and schedule of the last factorisation / solve
 * (PIQP_AMD_DEBUG=exact_trace, tools/exact_trace.py); copies the item into out_host (NULL: size only) and returns its length, < 0 on error (another engine) */
long long pq_kkt_exact_factor(pq_kkt *k, int what, void *out_host);
/* test hook (sparse backends): smallest |pivot| of the last factorisation */
int pq_kkt_min_abs_pivot(pq_kkt *k, double *out);
/* collectives the native transport has enqueued so far: out[which] for which = 0, 1, 2 (test / bench bookkeeping) */
int pq_kkt_native_exchange_calls(pq_kkt *k, int out[3]);
/* which transport a partitioned handle uses and what its communicator says about itself: out[0] = 0 none / 1 callback / 2 native RCCL; for the native
 * transport out[1..3] = ncclCommCount, ncclCommUserRank, ncclCommCuDevice of the library's communicator (-1 otherwise) -- the figures a multi-GPU
 * bench line carries so that "RCCL saw N ranks" can be checked from the outside */
int pq_kkt_comm_info(pq_kkt *k, int out[4]);
/* what the partition looks like: out[0] = supernodes owned by this rank, out[1] = shared supernodes, out[2] = boundary
 * subtree roots, out[3..4] = this rank's column span, out[5] = work share of this rank in permille, out[6] = shared
 * (replicated) work in permille */
int pq_kkt_partition_info(pq_kkt *k, int out[8]);
/* host-only planning hook for tests (no GPU needed): analyses the KKT pattern of `mode` (0 full .. 3 all eliminated)
 * and writes the owning rank of every permuted column (-1 = shared top) into owner_out[N]; returns N */
int pq_sparse_partition_plan(const pq_sparse_data *data, int mode, int world, int *owner_out, int capacity,
                             double *work_out /* world + 1: per rank, then shared */);

/* ---- the integer work of sparse::KKT's constructor, host-only (no GPU needed), exported so that the index work of the product can be pinned bit for
 * bit (tests/test_symbolic_parity.py: the reference's exact 4 x 4 case, tests/src/sparse/utils_test.cpp:55-92, and the oracle on the frozen fixtures).
 * pq_sparse_amd_order: Eigen-style approximate minimum degree on the pattern of the upper-triangular CSC matrix (sparse/ordering.hpp:67-84: the
 *   `P_eigen.indices()` of AMDOrdering::init, perm[new] = old).
 * pq_sparse_permute_sym_upper: permute_sparse_symmetric_matrix (sparse/utils.hpp:32-128): C = upper(P A P') for perm_inv[old] = new, with the map
 *   Ai_to_Ci of value positions (its return value); Cp[n + 1], Ci[nnz], Ai_to_Ci[nnz].
 * pq_sparse_kkt_symbolic: create_kkt_matrix of `mode` (kkt_full.hpp:39-170 and the three eliminated variants; 0 full, 1 eq, 2 ineq, 3 all) followed by
 *   the two steps above exactly as sparse/kkt.hpp:51-70 runs them.  Call with all outputs NULL to get sizes: returns N and writes nnz(K) to *nnz_out.
 *   Kp[N + 1], Ki[nnz], perm[N] (the AMD ordering), PKp[N + 1], PKi_rows[nnz] (pattern of PKPt), PKi[nnz] (K value index -> PKPt value index).
 * pq_kkt_sparse_ordering (sparse backends, after create): what the handle actually eliminates in.  fill_perm[N] = the fill-reducing ordering chosen
 *   (AMD as above, or nested dissection; perm[new] = old), elim_perm[N] = the same composed with the assembly-tree postorder / leaf amalgamation the
 *   device schedule uses (any postorder of the elimination tree has the same fill), returns 0 = amd, 1 = nested dissection, < 0 on error. */
int pq_sparse_amd_order(int n, const int *Ap, const int *Ai, int *perm);
int pq_sparse_permute_sym_upper(int n, const int *Ap, const int *Ai, const int *perm_inv, int *Cp, int *Ci, int *Ai_to_Ci);
int pq_sparse_kkt_symbolic(const pq_sparse_data *data, int mode, int *nnz_out, int *Kp, int *Ki, int *perm, int *PKp, int *PKi_rows,
                           int *PKi);
int pq_kkt_sparse_ordering(pq_kkt *k, int *fill_perm, int *elim_perm);
/* host-only planning hook of the reference-order engine (PQ_SPARSE_LDLT_EXACT; no GPU needed): everything sparse/ldlt.hpp:42-169 decides from the pattern of the
 * KKT_FULL matrix alone, and the schedule the device kernels replay it with (csrc/sparse_symbolic.hpp, struct UpLooking).  Items (int32 arrays unless noted):
 *   0 stats { nnz(L), tasks, tree height, dependent steps on the longest root path, nnz(K), tickets }   1 perm[N] (AMD, perm[new] = old)
 *   2 Cp[N + 1]  3 Ci[nnz(K)] (pattern of P K P')   4 diag_pos[N]   5 etree[N]   6 Lp[N + 1]   7 Li[nnz(L)] (L in CSC, rows ascending)
 *   8 Rp[N + 1]  9 Rcol  10 Rpos (row k of L in the reference's visiting order: column, CSC position)
 *   11 task_ptr  12 task_rows (tasks = paths of the elimination tree of at most 64 rows)   13 row_task  14 row_lane  15 row_prev
 *   16 dep_ptr  17 dep (children a row pass waits for)   18 tk_kind  19 tk_id (tickets: 0 row pass of a row, 1 path pass of a task)
 *   20 Rcnt  21 Rtab (per entry: leading column entries the row pass scatters, -1 = own path; table row)   22 tab_ptr  23 mask_ptr  24 task_nU
 *   25 Tmask (uint64: presence bits of every table row)
 *   100 + mode (mode = KKTMode bits: 1 equalities eliminated, 2 inequalities eliminated): { N, nnz(L), kiloflops of the factorisation } of that mode's system -- what the
 *       engine choice of pq_kkt_create_sparse looks at (PIQP_AMD_EXACT_MAX_N, PIQP_AMD_EXACT_MAX_FLOPS)
 * len[q] receives the length of item what[q]; out[q] (may be NULL, as may `out`) receives a copy.  Returns N. */
int pq_sparse_uplooking_plan(const pq_sparse_data *data, int nitems, const int *what, void **out, long long *len);

/* measurement hooks: when enabled, the backend brackets its stages with hipEvents on its own stream.
 * stage 0 = KKT assembly kernel (dense: k_syrk_lower<ASSEMBLE>), 1 = factorisation (all panels),
 * 2 = backend solve.  enable = 2 (dense backend, measurement passes only) also brackets individual launches: 3 = fused trailing update +
 * next diagonal block, 4 = panel solve, 5 = both triangular sweeps.  pq_kkt_get_profile returns the accumulated milliseconds / call count
 * and resets them. */
int pq_kkt_set_profiling(pq_kkt *k, int enable);
int pq_kkt_get_profile(pq_kkt *k, int stage, double *total_ms, int *count);

/* ===================== KKTSystem: replaces piqp::KKTSystem<T,I,MatrixType> ===================== */
typedef struct pq_kktsys pq_kktsys;

/* KKTSystem::init, kkt_system.hpp:97-132 (+ backend factory :455-497 keyed by settings->kkt_solver) */
int pq_kktsys_create_dense(pq_kktsys **out, const pq_dense_data *data, const pq_settings *settings, int device);
int pq_kktsys_create_sparse(pq_kktsys **out, const pq_sparse_data *data, const pq_settings *settings, int device);
int pq_kktsys_clone(const pq_kktsys *k, pq_kktsys **out); /* kkt_system.hpp:70-95 */
void pq_kktsys_destroy(pq_kktsys *k);
int pq_kktsys_set_pointer_mode(pq_kktsys *k, int mem);
pq_kkt *pq_kktsys_backend(pq_kktsys *k); /* borrowed */
/* kkt_system.hpp:134-141 */
int pq_kktsys_update_data_dense(pq_kktsys *k, const pq_dense_data *data, int options);
int pq_kktsys_update_data_sparse(pq_kktsys *k, const pq_sparse_data *data, int options);
/* kkt_system.hpp:143-211; returns 1 ok / 0 factor failed */
int pq_kktsys_update_scalings_and_factor(pq_kktsys *k, int iterative_refinement, double rho, double delta,
                                         const pq_vars *vars);
/* kkt_system.hpp:213-369 incl. the iterative-refinement loop (:256-301) run against device-resident
 * vectors; returns 1 ok / 0 (non-finite).  lhs buffers are written in place (never swapped). */
int pq_kktsys_solve(pq_kktsys *k, const pq_vars *rhs, pq_vars *lhs);
/* kkt_system.hpp:392-425 rhs = K_full * lhs */
int pq_kktsys_mul(pq_kktsys *k, const pq_vars *lhs, pq_vars *rhs);
/* diagnostics of the last solve: refinement steps taken, backend solves, final refine error, rhs norm */
int pq_kktsys_last_solve_stats(const pq_kktsys *k, int *refine_steps, int *backend_solves, double *refine_error,
                               double *rhs_norm);
/* ||rhs_bar - K_cond * lhs||_inf and ||rhs_bar||_inf of the last solve, recomputed on device via
 * mul_condensed_kkt (kkt_system.hpp:507-536); the harness's parity metric (SURVEY.md 8b last row) */
int pq_kktsys_condensed_residual(pq_kktsys *k, double *res_inf, double *rhs_inf);
int pq_kktsys_synchronize(pq_kktsys *k);

/* ===================== Solver: piqp::DenseSolver / SparseSolver front-end (caller of the hot path) ===== */
typedef struct pq_solver pq_solver;

int pq_solver_create(pq_solver **out, int device);
void pq_solver_destroy(pq_solver *s);
int pq_solver_clone(const pq_solver *s, pq_solver **out);
pq_settings *pq_solver_settings(pq_solver *s); /* solver.hpp:65 settings() */
/* DenseSolver::setup, solver.hpp:1266-1277: P n x n, A p x n, G m x n column-major HOST arrays;
 * NULL = nullopt.  Returns 1 when setup_done. */
int pq_solver_setup_dense(pq_solver *s, int n, int p, int m, const double *P, const double *c, const double *A,
                          const double *b, const double *G, const double *h_l, const double *h_u, const double *x_l,
                          const double *x_u);
/* SparseSolver::setup, solver.hpp:1297-1308: CSC HOST arrays (P full or upper) */
int pq_solver_setup_sparse(pq_solver *s, int n, int p, int m, const int *Pp, const int *Pi, const double *Px,
                           const double *c, const int *Ap, const int *Ai, const double *Ax, const double *b,
                           const int *Gp, const int *Gi, const double *Gx, const double *h_l, const double *h_u,
                           const double *x_l, const double *x_u);
/* DenseSolver::update, solver.hpp:1279-1290 */
int pq_solver_update_dense(pq_solver *s, const double *P, const double *c, const double *A, const double *b,
                           const double *G, const double *h_l, const double *h_u, const double *x_l,
                           const double *x_u);
int pq_solver_update_sparse(pq_solver *s, const int *Pp, const int *Pi, const double *Px, const double *c,
                            const int *Ap, const int *Ai, const double *Ax, const double *b, const int *Gp,
                            const int *Gi, const double *Gx, const double *h_l, const double *h_u,
                            const double *x_l, const double *x_u);
int pq_solver_solve(pq_solver *s);               /* solver.hpp:69-148; returns Status */
const pq_info *pq_solver_info(const pq_solver *s); /* result().info */
/* result(): copies the ten solution vectors (sizes n,p,m,m,n,n,m,m,n,n) into host buffers (NULL skipped) */
int pq_solver_get_result(const pq_solver *s, pq_vars *out_host);
int pq_solver_dims(const pq_solver *s, int *n, int *p, int *m);
/* optional per-iteration trace (rows of 11 doubles = the verbose table, solver.hpp:590-602) */
int pq_solver_set_trace(pq_solver *s, double *buf_host, int max_rows);
int pq_solver_trace_rows(const pq_solver *s);
/* pq_kkt_partition / pq_kkt_set_exchange on the solver's KKT backend (after setup): the interior-point loop then runs
 * replicated on every rank, bit for bit the same, with the factor / solve work split across the ranks */
int pq_solver_partition(pq_solver *s, int rank, int world, long long sizes_out[3]);
int pq_solver_set_exchange(pq_solver *s, pq_exchange_fn exchange, void *user, double *buf_factor, double *buf_forward,
                           double *buf_gather);
int pq_solver_set_comm_rccl(pq_solver *s, const unsigned char id[128], int rank, int world);
int pq_solver_native_exchange_calls(pq_solver *s, int out[3]);
int pq_solver_set_exchange_norm(pq_solver *s, double *buf_norm); /* as pq_kkt_set_exchange_norm */
int pq_solver_sharded_calls(pq_solver *s, int out[2]);
int pq_solver_sharded_solve_calls(pq_solver *s, int out[6]);
int pq_solver_comm_info(pq_solver *s, int out[4]); /* as pq_kkt_comm_info */

/* ===================== Batched solver: many structurally identical sparse QPs in one launch ===================== */
/* The reference has no batch API (one SolverBase per QP, solver.hpp:42); this is the device-side equivalent of
 *     for (i < batch) { SparseSolver s; s.settings() = settings; s.setup(P_i, c_i, A_i, ...); s.solve(); }
 * with kkt_solver = sparse_multistage: one workgroup runs the whole interior-point method (solver.hpp:379-1259) of one
 * QP, the batch is the grid.  All instances share the sparsity patterns and the set of finite bounds. */
typedef struct pq_batch pq_batch;

int pq_batch_create(pq_batch **out, int device);
void pq_batch_destroy(pq_batch *s);
pq_settings *pq_batch_settings(pq_batch *s); /* solver.hpp:65 settings(), shared by all instances */
/* SparseSolver::setup (solver.hpp:1297-1308) per instance.  Patterns (CSC, HOST): P n x n (upper triangle taken),
 * A p x n, G m x n; NULL pattern = absent.  Values / vectors are [batch][nnz] resp. [batch][len] row-major HOST
 * arrays; NULL h_l/h_u/x_l/x_u = nullopt.  Returns 1 when set up. */
int pq_batch_setup_sparse(pq_batch *s, int batch, int n, int p, int m, const int *Pp, const int *Pi, const double *Px,
                          const double *c, const int *Ap, const int *Ai, const double *Ax, const double *b,
                          const int *Gp, const int *Gi, const double *Gx, const double *h_l, const double *h_u,
                          const double *x_l, const double *x_u);
/* update() of every instance with new VECTORS only (solver.hpp:218-308 with every optional matrix empty; the Ruiz scaling of the
 * setup is reused, :283-285).  [batch][len] HOST arrays, NULL = unchanged.  The set of finite bounds is part of the shared
 * structure and must not change (error otherwise); a doubly-infinite row of G stays as it was at setup. */
int pq_batch_update(pq_batch *s, const double *c, const double *b, const double *h_l, const double *h_u,
                    const double *x_l, const double *x_u);
/* update() of every instance with new MATRIX VALUES and / or vectors (solver.hpp:218-308 with update_P / update_A / update_G of
 * :317-358).  Px / Ax / Gx: [batch][nnz] HOST arrays in the CSC order of the patterns given at setup (identical sparsity is the
 * reference's precondition too), NULL = unchanged.  Per instance, on the device: unscale_data, assign, scale_data with a fresh Ruiz
 * equilibration (sparse/preconditioner.hpp:65-222) unless settings.preconditioner_reuse_on_update, then the front arenas are rebuilt.
 * With all three matrices NULL this is pq_batch_update.  Returns 1. */
int pq_batch_update_data(pq_batch *s, const double *Px, const double *Ax, const double *Gx, const double *c, const double *b,
                         const double *h_l, const double *h_u, const double *x_l, const double *x_u);
/* solve() of every instance (solver.hpp:69-148); returns the number of instances that ended PQ_SOLVED (>= 0) */
int pq_batch_solve(pq_batch *s);
const pq_info *pq_batch_info(const pq_batch *s, int instance); /* result().info of one instance */
/* result(): field k of Variables (0..9 = x, y, z_l, z_u, z_bl, z_bu, s_l, s_u, s_bl, s_bu) of ALL instances,
 * copied to a HOST array [batch][len] (len = n, p or m) */
int pq_batch_get_result(pq_batch *s, int field, double *out_host);
int pq_batch_dims(const pq_batch *s, int *batch, int *n, int *p, int *m);
int pq_batch_block_info(const pq_batch *s, int *out_host, int capacity); /* as pq_kkt_multistage_block_info */
/* in-kernel device-clock seconds of one instance's last solve: out8 = {KKT assembly, chain factorisation, chain
 * substitution, KKTSystem::solve total, residual updates, whole solve, 0, 0} */
int pq_batch_get_profile(pq_batch *s, int instance, double *out8);
/* hipEvent time of the last solve's kernel and the workgroup size used per QP */
int pq_batch_last_kernel_ms(const pq_batch *s, double *ms, int *threads_per_qp);
/* start order of the instances inside the launch: 1 (default) = the instances that needed most iterations in the PREVIOUS solve of this handle start first
 * (receding-horizon batches repeat their counts; a batch larger than the device holds at once then ends earlier), 0 = always in index order.  Results do not
 * depend on it. */
int pq_batch_set_start_order(pq_batch *s, int longest_first);

/* ===================== the dense factorisation classes as objects of their own ===================== */
/* piqp::dense::LDLTNoPivot<Mat, UpLo> (dense/ldlt_no_pivot.hpp:87-262; kind = PQ_DENSE_LDLT_NO_PIVOT) and the Eigen::LLT<Mat, UpLo> that dense/kkt.hpp:82 uses
 * (kind = PQ_DENSE_CHOLESKY), for either triangle: what tests/src/dense/ldlt_test.cpp and benchmarks/src/dense_cholesky_factorization_benchmark.cpp:16-109 exercise.
 * The kernels are the dense KKT backend's; the Upper variants work on the transposed view as ldlt_no_pivot.hpp:357-371 does, so U = L^T bit for bit. */
enum { PQ_LOWER = 1, PQ_UPPER = 2 }; /* Eigen::Lower, Eigen::Upper */
typedef struct pq_dense_factor pq_dense_factor;
int pq_dense_factor_create(pq_dense_factor **out, int device, int n, int kind, int uplo); /* ldlt_no_pivot.hpp:126-131 (preallocating constructor) */
void pq_dense_factor_destroy(pq_dense_factor *f);
/* compute(), ldlt_no_pivot.hpp:393-423: reads the `uplo` triangle of A (column-major, leading dimension lda >= n, in host or device memory per `mem`).
 * Returns info(): 0 = Eigen::Success, 1 = Eigen::NumericalIssue (LDLTNoPivot: an exact zero pivot, :307; LLT: a pivot <= 0), negative = PQ_ERR_*. */
int pq_dense_factor_compute(pq_dense_factor *f, const double *A, int lda, int mem);
int pq_dense_factor_info(const pq_dense_factor *f); /* ldlt_no_pivot.hpp:231 */
/* solveInPlace(), ldlt_no_pivot.hpp:432-450 / 464-470: x[n] <- A^-1 x */
int pq_dense_factor_solve_in_place(pq_dense_factor *f, double *x, int mem);
/* matrixLDLT(), ldlt_no_pivot.hpp:217 (LLT: matrixLLT()): writes the `uplo` triangle of out_host (column-major, leading dimension ldo): the strictly triangular
 * part of unit L (PQ_UPPER: of U = L^T) with D on the diagonal; LLT: L resp. U.  The other triangle of out_host is not touched. */
int pq_dense_factor_matrix(pq_dense_factor *f, double *out_host, int ldo);
/* out2[0]: device time of the factorisation launches of the last compute() (hipEvents), out2[1]: wall time of the whole call (copy + symmetric completion +
 * factorisation + status read-back), both in ms */
int pq_dense_factor_last_ms(const pq_dense_factor *f, double out2[2]);

/* ===================== small utilities used by the measurement harness ===================== */
/* fp64 MFMA / HBM micro-benchmarks on `device` (used once by bench.py to report measured peaks) */
/* number of device / pinned-host allocations the library has made in this process.  Contract (the reference's tests assert allocation-free
 * factor / solve, fwd.hpp:44-52): the counter moves only inside *_create / *_clone / *_setup / *_update_data / *_partition calls, never in
 * update_scalings_and_factor, solve, eval_*, mul or condensed_residual, in either pointer mode */
long long pq_debug_alloc_count(void);
/* host-only (no device needed): the ticket-ordered task list of the persistent dense factorisation for a T x T grid of 128 x 128 tiles (csrc/dense_kernels.hip
 * k_chol_persistent; mchunks > 0: with the assembly of dense/kkt.hpp:140-160 fused in, m = 128 mchunks), six ints per task (kind, round, a, b, gate, aux).
 * Returns the number of tasks (-1: T outside [3, 1024]).  tests/test_chol_plan.py replays the list on the CPU: every task only waits for EARLIER tickets. */
int pq_debug_chol_plan(int T, int mchunks, int *out6, int capacity_tasks);
int pq_microbench_mfma_f64(int device, int iters, double *tflops_out);
int pq_microbench_hbm_copy(int device, size_t bytes, int iters, double *gbps_out);
/* debugging aid: average microseconds of the 128 x 128 diagonal-block factorisation kernel and 64 in-kernel shader-clock stamps
 * (step k at stamps64[8 k + q], see potrf_block in csrc/dense_kernels.hip); stamps64 may be NULL */
int pq_microbench_potrf_block(int device, int ldlt, int reps, double *us_out, long long *stamps64);
/* testing aid: the diagonal-block factorisation kernel (potrf_block: the unblocked part of Eigen::LLT, dense/kkt.hpp:82, resp. dense/ldlt_no_pivot.hpp:278-311) on the
 * caller's symmetric block A of order nb <= 128 (host memory, column-major, leading dimension 128, lower triangle read), `reps` times.  Outputs of the last repetition
 * (host memory): the factor L (128 x 128, lower triangle; LDLT: unit L with D on the diagonal), the reciprocal pivots rdiag[128] (LLT: 1 / l_cc, LDLT: 1 / d_c), D dvec[128]
 * (LDLT), the operand pack of the panel solve pack[36 * 256] (blocks j (j - 1) / 2 + k: -L(j, k), blocks 28 + k: the inverse of L_kk; column-major 16 x 16 each), info
 * (-1 or the first failing column), and the number of repetitions whose outputs differ from the first one's in any bit (0: the kernel is deterministic) */
int pq_debug_potrf_block(int device, int ldlt, int nb, int reps, const double *A, double *L, double *rdiag, double *dvec, double *pack, int *info, int *differing_reps);

#ifdef __cplusplus
}
#endif
#endif /* PIQP_AMD_H */
