/*
 * oracle/orc.h -- CPU restatement ("oracle") of PIQP v0.6.2's KKT hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under piqp_amd/ (the product) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, as the checker / the timed CPU baseline.
 *
 * Every function cites the reference file:line it follows (paths relative
 * to /root/reference/include/piqp/).  The reference itself cannot be built
 * in this image (Eigen, blasfeo, matio are absent), so this is a restatement
 * in plain C99; it is pinned by the reference's own test properties and
 * known answers (tests/test_oracle_*.py, see DESIGN.md "Oracle pinning").
 * Eigen::LLT's bit-level output and Eigen::AMDOrdering on large inputs are
 * third-party arithmetic with no golden vectors in the reference tree:
 * for those two pieces parity is "unpinned" at the bit level and pinned at
 * the property level (residual, exact 4x4 ordering case).
 *
 * All matrices column-major, fp64; index lists int32.
 */
#ifndef PIQP_ORACLE_ORC_H
#define PIQP_ORACLE_ORC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_INF 1e30 /* fwd.hpp:54 PIQP_INF */

/* kkt_fwd.hpp:23-29 */
enum { ORC_KKT_UPDATE_NONE = 0, ORC_KKT_UPDATE_P = 1, ORC_KKT_UPDATE_A = 2, ORC_KKT_UPDATE_G = 4 };

/* settings.hpp:18-26 (+ one extra id for the in-tree pivot-free LDLt) */
enum {
    ORC_DENSE_CHOLESKY = 0,
    ORC_SPARSE_LDLT = 1,
    ORC_SPARSE_LDLT_EQ_COND = 2,
    ORC_SPARSE_LDLT_INEQ_COND = 3,
    ORC_SPARSE_LDLT_COND = 4,
    ORC_SPARSE_MULTISTAGE = 5,
    ORC_DENSE_LDLT_NO_PIVOT = 16
};

/* results.hpp:18-27 */
enum {
    ORC_SOLVED = 1,
    ORC_MAX_ITER_REACHED = -1,
    ORC_PRIMAL_INFEASIBLE = -2,
    ORC_DUAL_INFEASIBLE = -3,
    ORC_NUMERICS = -8,
    ORC_UNSOLVED = -9,
    ORC_INVALID_SETTINGS = -10
};

/* compressed-sparse-column matrix (typedefs.hpp:53-54), int32 indices */
typedef struct {
    int rows, cols;
    int *colptr; /* cols+1 */
    int *rowind; /* nnz */
    double *val; /* nnz */
} orc_csc;

/* dense/data.hpp:22-51 and sparse/data.hpp:26-54 in one struct */
typedef struct {
    int is_sparse;
    int n, p, m;
    /* dense storage */
    double *P_utri; /* n x n, upper triangle used */
    double *AT;     /* n x p */
    double *GT;     /* n x m */
    /* sparse storage */
    orc_csc sP_utri, sAT, sGT;
    double *c, *b, *h_l, *h_u, *x_l, *x_u;
    int n_h_l, n_h_u, n_x_l, n_x_u;
    int *h_l_idx, *h_u_idx, *x_l_idx, *x_u_idx;
    double *x_b_scaling;
} orc_data;

/* variables.hpp:19-105 */
typedef struct {
    double *x, *y, *z_l, *z_u, *z_bl, *z_bu, *s_l, *s_u, *s_bl, *s_bu;
} orc_vars;

/* settings.hpp:43-82 */
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
    int kkt_solver;
    int iterative_refinement_always_enabled;
    double iterative_refinement_eps_abs, iterative_refinement_eps_rel;
    int iterative_refinement_max_iter;
    double iterative_refinement_min_improvement_rate;
    double iterative_refinement_static_regularization_eps;
    double iterative_refinement_static_regularization_rel;
    int verbose, compute_timings;
} orc_settings;

/* results.hpp:45-89 */
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
    /* extra bookkeeping (not in the reference): counts for the bench harness */
    int n_factor, n_solve, n_backend_solve;
} orc_info;

void orc_settings_default(orc_settings *s);

/* ---- data (dense) ------------------------------------------------------ */
/* P: n x n col-major (upper triangle taken); A: p x n col-major or NULL; G: m x n col-major or NULL;
 * h_l/h_u/x_l/x_u may be NULL (= absent optional). dense/data.hpp:53-208, solver.hpp:169-192 */
orc_data *orc_data_create_dense(int n, int p, int m, const double *P, const double *c, const double *A,
                                const double *b, const double *G, const double *h_l, const double *h_u,
                                const double *x_l, const double *x_u);
/* sparse: CSC inputs (P full or upper: upper triangle taken). sparse/data.hpp, solver.hpp:182-184 */
orc_data *orc_data_create_sparse(int n, int p, int m, const int *Pp, const int *Pi, const double *Px,
                                 const double *c, const int *Ap, const int *Ai, const double *Ax,
                                 const double *b, const int *Gp, const int *Gi, const double *Gx,
                                 const double *h_l, const double *h_u, const double *x_l, const double *x_u);
orc_data *orc_data_clone(const orc_data *d);
void orc_data_free(orc_data *d);
void orc_data_set_h_l(orc_data *d, const double *h_l);
void orc_data_set_h_u(orc_data *d, const double *h_u);
void orc_data_disable_inf_constraints(orc_data *d);
void orc_data_set_x_l(orc_data *d, const double *x_l);
void orc_data_set_x_u(orc_data *d, const double *x_u);

/* ---- KKT backend vtable (kkt_solver_base.hpp:20-44) -------------------- */
typedef struct orc_kkt orc_kkt;
struct orc_kkt {
    orc_kkt *(*clone)(const orc_kkt *self);
    void (*update_data)(orc_kkt *self, const orc_data *d, int options);
    int (*update_scalings_and_factor)(orc_kkt *self, const orc_data *d, double delta, const double *x_reg,
                                      const double *z_reg); /* 1 = ok, 0 = failed */
    void (*solve)(orc_kkt *self, const orc_data *d, const double *rhs_x, const double *rhs_y,
                  const double *rhs_z, double *lhs_x, double *lhs_y, double *lhs_z);
    void (*eval_P_x)(orc_kkt *self, const orc_data *d, double alpha, const double *x, double *z);
    void (*eval_A_xn_and_AT_xt)(orc_kkt *self, const orc_data *d, double alpha_n, double alpha_t,
                                const double *xn, const double *xt, double *zn, double *zt);
    void (*eval_G_xn_and_GT_xt)(orc_kkt *self, const orc_data *d, double alpha_n, double alpha_t,
                                const double *xn, const double *xt, double *zn, double *zt);
    void (*print_info)(orc_kkt *self);
    void (*destroy)(orc_kkt *self);
};

/* dense/kkt.hpp:39-55; use_ldlt=1 swaps Eigen::LLT for dense/ldlt_no_pivot.hpp */
orc_kkt *orc_dense_kkt_create(const orc_data *d, int use_ldlt);
/* test hook: dense/kkt.hpp:134 internal_kkt_mat() */
const double *orc_dense_kkt_internal_kkt_mat(const orc_kkt *k);
const double *orc_dense_kkt_internal_factor(const orc_kkt *k);
void orc_set_num_threads(int t); /* threads used by the dense SYRK/GEMM helpers (baseline timing only) */

/* stand-alone factorisations (dense/ldlt_no_pivot.hpp:278-354,393-450; Eigen LLT semantics SURVEY A.4) */
int orc_llt_compute(double *a, int n, int lda);   /* returns -1 ok, k = first non-positive pivot */
void orc_llt_solve_inplace(const double *l, int n, int lda, double *x);
int orc_ldlt_no_pivot_compute(double *a, int n, int lda, double *work_n); /* -1 ok, k on zero pivot */
void orc_ldlt_no_pivot_solve_inplace(const double *ld, int n, int lda, double *x);

/* ---- sparse pieces ----------------------------------------------------- */
/* sparse/kkt.hpp:51-70 ; mode = KKTMode bits (kkt_fwd.hpp:15-21) */
orc_kkt *orc_sparse_kkt_create(const orc_data *d, int mode);
/* sparse/ldlt.hpp stand-alone (for tests/src/sparse/ldlt_test.cpp analogue) */
typedef struct orc_sparse_ldlt orc_sparse_ldlt;
orc_sparse_ldlt *orc_sparse_ldlt_create(void);
void orc_sparse_ldlt_free(orc_sparse_ldlt *f);
void orc_sparse_ldlt_symbolic(orc_sparse_ldlt *f, int n, const int *Ap, const int *Ai);
int orc_sparse_ldlt_numeric(orc_sparse_ldlt *f, int n, const int *Ap, const int *Ai, const double *Ax);
void orc_sparse_ldlt_solve_inplace(const orc_sparse_ldlt *f, double *x);
int orc_sparse_ldlt_nnz(const orc_sparse_ldlt *f);
orc_sparse_ldlt *orc_sparse_ldlt_clone(const orc_sparse_ldlt *f);
/* condensed modes: KKT_EQ_ELIMINATED = 1, KKT_INEQ_ELIMINATED = 2, KKT_ALL_ELIMINATED = 3 (kkt_fwd.hpp:15-21) */
orc_kkt *orc_sparse_cond_kkt_create(const orc_data *d, int mode);
/* AMD ordering (sparse/ordering.hpp:67-84 -> Eigen::AMDOrdering, third-party) on the pattern of an
 * upper-triangular CSC matrix; writes perm[n] (new -> old). */
void orc_amd_order(int n, const int *Ap, const int *Ai, int *perm);
/* sparse/utils.hpp:32-128: C = P A P^T (upper), returns Ai_to_Ci map; perm_inv old->new */
void orc_permute_sym_upper(int n, const int *Ap, const int *Ai, const double *Ax, const int *perm_inv,
                           int *Cp, int *Ci, double *Cx, int *Ai_to_Ci);
/* test hooks into the sparse backend */
int orc_sparse_kkt_dim(const orc_kkt *k);
const int *orc_sparse_kkt_PKPt_colptr(const orc_kkt *k);
const int *orc_sparse_kkt_PKPt_rowind(const orc_kkt *k);
const double *orc_sparse_kkt_PKPt_val(const orc_kkt *k);
const int *orc_sparse_kkt_perm(const orc_kkt *k);
int orc_sparse_kkt_L_nnz(const orc_kkt *k);
const int *orc_sparse_kkt_L_cols(const orc_kkt *k);
const int *orc_sparse_kkt_L_ind(const orc_kkt *k);
const double *orc_sparse_kkt_L_vals(const orc_kkt *k);
const double *orc_sparse_kkt_D(const orc_kkt *k);
const double *orc_sparse_kkt_D_inv(const orc_kkt *k);
const int *orc_sparse_kkt_etree(const orc_kkt *k);
const int *orc_sparse_kkt_PKi(const orc_kkt *k);
int orc_sparse_kkt_nnz(const orc_kkt *k);
int orc_sparse_cond_kkt_dim(const orc_kkt *k);
int orc_sparse_cond_kkt_nnz(const orc_kkt *k);
const int *orc_sparse_cond_kkt_perm(const orc_kkt *k);
const int *orc_sparse_cond_kkt_PKPt_colptr(const orc_kkt *k);
const int *orc_sparse_cond_kkt_PKPt_rowind(const orc_kkt *k);
const int *orc_sparse_cond_kkt_PKi(const orc_kkt *k);

/* ---- KKTSystem (kkt_system.hpp) ---------------------------------------- */
typedef struct orc_kkt_system orc_kkt_system;
orc_kkt_system *orc_kkt_system_create(const orc_data *d, const orc_settings *s); /* init :97-132 */
orc_kkt_system *orc_kkt_system_clone(const orc_kkt_system *k);
void orc_kkt_system_free(orc_kkt_system *k);
orc_kkt *orc_kkt_system_backend(orc_kkt_system *k);
void orc_kkt_system_update_data(orc_kkt_system *k, const orc_data *d, int options); /* :134-141 */
int orc_kkt_system_update_scalings_and_factor(orc_kkt_system *k, const orc_data *d, const orc_settings *s,
                                              int iterative_refinement, double rho, double delta,
                                              const orc_vars *vars); /* :143-211 */
int orc_kkt_system_solve(orc_kkt_system *k, const orc_data *d, const orc_settings *s, const orc_vars *rhs,
                         orc_vars *lhs); /* :213-369 ; lhs pointers may be swapped */
void orc_kkt_system_mul(orc_kkt_system *k, const orc_data *d, const orc_vars *lhs, orc_vars *rhs); /* :392-425 */
int orc_kkt_system_last_refine_steps(const orc_kkt_system *k);
int orc_kkt_system_solve_copy(orc_kkt_system *k, const orc_data *d, const orc_settings *s, const orc_vars *rhs, orc_vars *out);
const double *orc_kkt_system_x_reg(const orc_kkt_system *k);
const double *orc_kkt_system_z_reg(const orc_kkt_system *k);
const double *orc_kkt_system_rhs_x_bar(const orc_kkt_system *k);
const double *orc_kkt_system_rhs_z_bar(const orc_kkt_system *k);
/* flat dispatch through the backend vtable (FFI convenience) */
orc_kkt *orc_kkt_clone(const orc_kkt *k);
void orc_kkt_destroy(orc_kkt *k);
void orc_kkt_update_data(orc_kkt *k, const orc_data *d, int options);
int orc_kkt_update_scalings_and_factor(orc_kkt *k, const orc_data *d, double delta, const double *x_reg, const double *z_reg);
void orc_kkt_solve(orc_kkt *k, const orc_data *d, const double *rx, const double *ry, const double *rz, double *lx, double *ly, double *lz);
void orc_kkt_eval_P_x(orc_kkt *k, const orc_data *d, double alpha, const double *x, double *z);
void orc_kkt_eval_A_xn_and_AT_xt(orc_kkt *k, const orc_data *d, double an, double at, const double *xn, const double *xt, double *zn, double *zt);
void orc_kkt_eval_G_xn_and_GT_xt(orc_kkt *k, const orc_data *d, double an, double at, const double *xn, const double *xt, double *zn, double *zt);
int orc_kkt_system_backend_solves(const orc_kkt_system *k);

/* ---- Solver (solver.hpp) ------------------------------------------------ */
typedef struct orc_solver orc_solver;
orc_solver *orc_solver_create(void);
orc_solver *orc_solver_clone(const orc_solver *s);
void orc_solver_free(orc_solver *s);
orc_settings *orc_solver_settings(orc_solver *s);
/* takes ownership of data */
int orc_solver_setup(orc_solver *s, orc_data *data); /* setup_impl :151-216 ; 1 = setup_done */
/* update_impl :218-308; NULL = nullopt.  dense: P n x n, A p x n, G m x n col-major.
 * sparse: pass value arrays with identical sparsity via the *_x arguments */
int orc_solver_update_dense(orc_solver *s, const double *P, const double *c, const double *A, const double *b,
                            const double *G, const double *h_l, const double *h_u, const double *x_l,
                            const double *x_u);
int orc_solver_update_sparse(orc_solver *s, const int *Pp, const int *Pi, const double *Px, const double *c,
                             const int *Ap, const int *Ai, const double *Ax, const double *b, const int *Gp,
                             const int *Gi, const double *Gx, const double *h_l, const double *h_u,
                             const double *x_l, const double *x_u);
int orc_solver_solve(orc_solver *s); /* :69-148 */
const orc_info *orc_solver_info(const orc_solver *s);
const orc_vars *orc_solver_result(const orc_solver *s);
const orc_data *orc_solver_data(const orc_solver *s);
/* optional per-iteration trace: rows of 11 doubles as printed by the verbose table (solver.hpp:590-602) */
void orc_solver_set_trace(orc_solver *s, double *buf, int max_rows);
int orc_solver_trace_rows(const orc_solver *s);
/* optional factor-state recorder for the bench replay: called with the KKTSystem inputs of every
 * update_scalings_and_factor / solve (used to capture realistic (rho,delta,s,z) states) */
typedef void (*orc_state_cb)(void *user, int kind /*0 factor, 1 solve*/, int refine, double rho, double delta,
                             const orc_vars *v);
void orc_solver_set_state_callback(orc_solver *s, orc_state_cb cb, void *user);

#ifdef __cplusplus
}
#endif
#endif
