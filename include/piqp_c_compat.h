/* include/piqp_c_compat.h -- the reference's C interface (interfaces/c/include/piqp.h:21-43 and
 * piqp_typedef.h:27-190) served by the MI355X solver: same function names, same argument meaning, same struct
 * layouts, so an existing C client recompiles against this header (or against its own piqp.h: the ABI is the same)
 * and links libpiqp_amd.so instead of the reference's libpiqpc.  SURVEY.md 8f rank 4.
 *
 * Differences a caller can observe: the solver runs on HIP device $PIQP_AMD_DEVICE (default 0); if no device is
 * present piqp_setup_* leaves *workspace = NULL (the reference cannot fail there); kkt_solver additionally accepts
 * PIQP_AMD_DENSE_LDLT_NO_PIVOT. */
#ifndef PIQP_C_COMPAT_H
#define PIQP_C_COMPAT_H

#ifdef __cplusplus
extern "C" {
#endif

#ifndef PIQP_INF
#define PIQP_INF 1e30 /* fwd.hpp:54 */
#endif

typedef double piqp_float; /* fp64 only: PIQP_SINGLE_PRECISION builds are not served */
typedef int piqp_int;      /* int32 only: PIQP_LONG_STORAGE_INDEX builds are not served */

/* compressed sparse column matrix: m rows, n columns, p[n+1] column starts, i[nnz] row indices, x[nnz] values */
typedef struct { piqp_int m, n, nnz; piqp_int *p, *i; piqp_float *x; } piqp_csc;

/* dense problem, matrices ROW-major (interfaces/c/src/piqp.cpp:13): P n x n (upper triangle used), A p x n, G m x n;
 * A, b, G, h_l, h_u, x_l, x_u may be NULL */
typedef struct {
    piqp_int n, p, m;
    piqp_float *P, *c, *A, *b, *G, *h_l, *h_u, *x_l, *x_u;
} piqp_data_dense;

typedef struct {
    piqp_int n, p, m;
    piqp_csc *P; piqp_float *c;
    piqp_csc *A; piqp_float *b;
    piqp_csc *G; piqp_float *h_l, *h_u, *x_l, *x_u;
} piqp_data_sparse;

typedef enum {
    PIQP_DENSE_CHOLESKY, PIQP_SPARSE_LDLT, PIQP_SPARSE_LDLT_EQ_COND, PIQP_SPARSE_LDLT_INEQ_COND, PIQP_SPARSE_LDLT_COND,
    PIQP_SPARSE_MULTISTAGE,
    PIQP_AMD_DENSE_LDLT_NO_PIVOT = 16 /* dense/ldlt_no_pivot.hpp (not selectable in the reference's C interface) */
} piqp_kkt_solver;

typedef struct { /* settings.hpp:43-82, field for field */
    piqp_float rho_init, delta_init, eps_abs, eps_rel;
    piqp_int check_duality_gap;
    piqp_float eps_duality_gap_abs, eps_duality_gap_rel, infeasibility_threshold, reg_lower_limit, reg_finetune_lower_limit;
    piqp_int reg_finetune_primal_update_threshold, reg_finetune_dual_update_threshold, max_iter, max_factor_retires;
    piqp_int preconditioner_scale_cost, preconditioner_reuse_on_update, preconditioner_iter;
    piqp_float tau;
    piqp_kkt_solver kkt_solver;
    piqp_int iterative_refinement_always_enabled;
    piqp_float iterative_refinement_eps_abs, iterative_refinement_eps_rel;
    piqp_int iterative_refinement_max_iter;
    piqp_float iterative_refinement_min_improvement_rate, iterative_refinement_static_regularization_eps,
        iterative_refinement_static_regularization_rel;
    piqp_int verbose, compute_timings;
} piqp_settings;

typedef enum {
    PIQP_SOLVED = 1, PIQP_MAX_ITER_REACHED = -1, PIQP_PRIMAL_INFEASIBLE = -2, PIQP_DUAL_INFEASIBLE = -3, PIQP_NUMERICS = -8,
    PIQP_UNSOLVED = -9, PIQP_INVALID_SETTINGS = -10
} piqp_status;

typedef struct { /* results.hpp Info<T> */
    piqp_status status;
    piqp_int iter;
    piqp_float rho, delta, mu, sigma, primal_step, dual_step;
    piqp_float primal_res, primal_res_rel, dual_res, dual_res_rel;
    piqp_float primal_res_reg, primal_res_reg_rel, dual_res_reg, dual_res_reg_rel;
    piqp_float primal_prox_inf, dual_prox_inf, prev_primal_res, prev_dual_res;
    piqp_float primal_obj, dual_obj, duality_gap, duality_gap_rel;
    piqp_int factor_retires;
    piqp_float reg_limit;
    piqp_int no_primal_update, no_dual_update;
    piqp_float setup_time, update_time, solve_time, kkt_factor_time, kkt_solve_time, run_time;
} piqp_info;

/* solution vectors live inside the workspace (sizes n, p, m, m, n, n, m, m, n, n) and stay valid until piqp_cleanup */
typedef struct {
    const piqp_float *x, *y, *z_l, *z_u, *z_bl, *z_bu, *s_l, *s_u, *s_bl, *s_bu;
    piqp_info info;
} piqp_result;

typedef struct piqp_solver_handle piqp_solver_handle; /* opaque */
typedef struct { piqp_int is_dense, n, p, m; } piqp_solver_info;
typedef struct {
    piqp_solver_handle *solver_handle;
    piqp_solver_info solver_info;
    piqp_result *result;
} piqp_workspace;

/* malloc'ed header around caller-owned arrays (the caller frees the header, as with the reference) */
piqp_csc *piqp_csc_matrix(piqp_int m, piqp_int n, piqp_int nnz, piqp_int *p, piqp_int *i, piqp_float *x);

void piqp_set_default_settings_dense(piqp_settings *settings);  /* kkt_solver = PIQP_DENSE_CHOLESKY */
void piqp_set_default_settings_sparse(piqp_settings *settings); /* kkt_solver = PIQP_SPARSE_LDLT */

void piqp_setup_dense(piqp_workspace **workspace, const piqp_data_dense *data, const piqp_settings *settings);
void piqp_setup_sparse(piqp_workspace **workspace, const piqp_data_sparse *data, const piqp_settings *settings);

void piqp_update_settings(piqp_workspace *workspace, const piqp_settings *settings);
/* NULL = unchanged; sparse updates need the sparsity pattern given at setup (solver.hpp:325,341,356) */
void piqp_update_dense(piqp_workspace *workspace, piqp_float *P, piqp_float *c, piqp_float *A, piqp_float *b,
                       piqp_float *G, piqp_float *h_l, piqp_float *h_u, piqp_float *x_l, piqp_float *x_u);
void piqp_update_sparse(piqp_workspace *workspace, piqp_csc *P, piqp_float *c, piqp_csc *A, piqp_float *b, piqp_csc *G,
                        piqp_float *h_l, piqp_float *h_u, piqp_float *x_l, piqp_float *x_u);

piqp_status piqp_solve(piqp_workspace *workspace);

void piqp_cleanup(piqp_workspace *workspace);

#ifdef __cplusplus
}
#endif

#endif /* PIQP_C_COMPAT_H */
