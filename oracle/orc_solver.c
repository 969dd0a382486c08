/*
 * oracle/orc_solver.c -- CPU restatement of piqp::SolverBase / RuizEquilibration (TEST INFRASTRUCTURE ONLY).
 * Follows /root/reference/include/piqp/:
 *   solver.hpp:69-148 solve, :151-216 setup_impl, :218-308 update_impl, :361-377 init_workspace,
 *              :379-882 solve_impl, :884-958 calculate_mu/calculate_step, :960-1105 update_residuals_nr,
 *              :1107-1128 update_residuals_r, :1130-1203 residual norms, :1205-1259 unscale_results/restore_dual
 *   dense/preconditioner.hpp:42-258 and sparse/preconditioner.hpp:45-286 (RuizEquilibration)
 *   settings.hpp:45-82 defaults
 * The IPM loop is not itself accelerated; it is restated so that iteration counts of the HIP-backed
 * host solver can be compared against a CPU run of the same algorithm.
 */
#include "orc.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double *dz(size_t n)
{
    void *p = NULL;
    size_t bytes = (n ? n : 1) * sizeof(double);
    if (posix_memalign(&p, 64, (bytes + 63) & ~(size_t)63)) abort();
    memset(p, 0, bytes);
    return (double *)p;
}
static double *dd(const double *s, size_t n) { double *p = dz(n); if (n) memcpy(p, s, n * sizeof(double)); return p; }
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static double dmax(double a, double b) { return a > b ? a : b; }
static double dmin(double a, double b) { return a < b ? a : b; }

/* settings.hpp:45-82 */
void orc_settings_default(orc_settings *s)
{
    s->rho_init = 1e-6; s->delta_init = 1e-4;
    s->eps_abs = 1e-8; s->eps_rel = 1e-9;
    s->check_duality_gap = 1; s->eps_duality_gap_abs = 1e-8; s->eps_duality_gap_rel = 1e-9;
    s->infeasibility_threshold = 0.9;
    s->reg_lower_limit = 1e-10; s->reg_finetune_lower_limit = 1e-13;
    s->reg_finetune_primal_update_threshold = 7; s->reg_finetune_dual_update_threshold = 7;
    s->max_iter = 250; s->max_factor_retires = 10;
    s->preconditioner_scale_cost = 0; s->preconditioner_reuse_on_update = 0; s->preconditioner_iter = 10;
    s->tau = 0.99;
    s->kkt_solver = ORC_DENSE_CHOLESKY;
    s->iterative_refinement_always_enabled = 0;
    s->iterative_refinement_eps_abs = 1e-12; s->iterative_refinement_eps_rel = 1e-12;
    s->iterative_refinement_max_iter = 10;
    s->iterative_refinement_min_improvement_rate = 5.0;
    s->iterative_refinement_static_regularization_eps = 1e-8;
    s->iterative_refinement_static_regularization_rel = DBL_EPSILON * DBL_EPSILON;
    s->verbose = 0; s->compute_timings = 0;
}

/* settings.hpp:84-106 */
static int verify_settings(const orc_settings *s)
{
    return s->rho_init > 0 && s->delta_init > 0 && s->eps_abs > 0 && s->eps_rel >= 0 &&
           s->eps_duality_gap_abs > 0 && s->eps_duality_gap_rel >= 0 && s->infeasibility_threshold >= 0 &&
           s->reg_lower_limit > 0 && s->reg_finetune_primal_update_threshold >= 0 &&
           s->reg_finetune_dual_update_threshold >= 0 && s->max_iter > 0 && s->max_factor_retires > 0 &&
           s->preconditioner_iter >= 0 && s->tau > 0 && s->tau <= 1 && s->iterative_refinement_eps_abs > 0 &&
           s->iterative_refinement_eps_rel >= 0 && s->iterative_refinement_max_iter >= 0 &&
           s->iterative_refinement_min_improvement_rate >= 1.0 &&
           s->iterative_refinement_static_regularization_eps > 0 &&
           s->iterative_refinement_static_regularization_rel >= 0;
}

/* ------------------------------------------------------ RuizEquilibration */
typedef struct {
    int n, p, m;
    double c, c_inv;
    double *delta, *delta_b, *delta_inv, *delta_b_inv;
} ruiz;

static const double MIN_SCALING = 1e-4, MAX_SCALING = 1e4;
/* dense/preconditioner.hpp:513-523 */
static double limit_scaling(double d) { if (d < MIN_SCALING) return 1.0; if (d > MAX_SCALING) return MAX_SCALING; return d; }

/* dense/preconditioner.hpp:42-60 */
static void ruiz_init(ruiz *r, const orc_data *d)
{
    r->n = d->n; r->p = d->p; r->m = d->m;
    int N = r->n + r->p + r->m;
    free(r->delta); free(r->delta_b); free(r->delta_inv); free(r->delta_b_inv);
    r->delta = dz(N); r->delta_b = dz(r->n); r->delta_inv = dz(N); r->delta_b_inv = dz(r->n);
    r->c = 1.0; r->c_inv = 1.0;
    for (int i = 0; i < N; i++) { r->delta[i] = 1.0; r->delta_inv[i] = 1.0; }
    for (int i = 0; i < r->n; i++) { r->delta_b[i] = 1.0; r->delta_b_inv[i] = 1.0; }
}
static void ruiz_free(ruiz *r) { free(r->delta); free(r->delta_b); free(r->delta_inv); free(r->delta_b_inv); memset(r, 0, sizeof(*r)); }
static void ruiz_copy(ruiz *dst, const ruiz *src)
{
    *dst = *src;
    int N = src->n + src->p + src->m;
    dst->delta = dd(src->delta, N); dst->delta_inv = dd(src->delta_inv, N);
    dst->delta_b = dd(src->delta_b, src->n); dst->delta_b_inv = dd(src->delta_b_inv, src->n);
}

/* symmetric scaling of the stored upper triangle: P_ij *= s_i * s_j (diag twice) and general D1 * M * D2 */
static void scale_P(orc_data *d, const double *s)
{
    int n = d->n;
    if (!d->is_sparse) {
        /* dense/preconditioner.hpp:119-124: col(k).head(k+1) *= s(k); row(k).tail(n-k) *= s(k) */
        for (int k = 0; k < n; k++) { double *col = d->P_utri + (size_t)k * n; for (int i = 0; i <= k; i++) col[i] *= s[k]; }
        for (int k = 0; k < n; k++) for (int j = k; j < n; j++) d->P_utri[k + (size_t)j * n] *= s[k];
    } else {
        /* sparse/utils.hpp:172-199 pre_mult_diagonal then post_mult_diagonal */
        for (int j = 0; j < n; j++) for (int q = d->sP_utri.colptr[j]; q < d->sP_utri.colptr[j + 1]; q++) d->sP_utri.val[q] *= s[d->sP_utri.rowind[q]];
        for (int j = 0; j < n; j++) for (int q = d->sP_utri.colptr[j]; q < d->sP_utri.colptr[j + 1]; q++) d->sP_utri.val[q] *= s[j];
    }
}
static void scale_P_scalar(orc_data *d, double g)
{
    int n = d->n;
    if (!d->is_sparse) { for (size_t i = 0; i < (size_t)n * n; i++) d->P_utri[i] *= g; }
    else { int nnz = d->sP_utri.colptr[n]; for (int q = 0; q < nnz; q++) d->sP_utri.val[q] *= g; }
}
static void scale_T(orc_data *d, int which /*0 AT, 1 GT*/, const double *srow, const double *scol)
{
    int n = d->n, cols = which ? d->m : d->p;
    if (!d->is_sparse) {
        double *M = which ? d->GT : d->AT;
        /* Eigen evaluates D1 * M * D2 as (D1*M)*D2 */
        for (int j = 0; j < cols; j++) { double *col = M + (size_t)j * n; for (int i = 0; i < n; i++) col[i] = (srow[i] * col[i]) * scol[j]; }
    } else {
        orc_csc *M = which ? &d->sGT : &d->sAT;
        for (int j = 0; j < cols; j++) for (int q = M->colptr[j]; q < M->colptr[j + 1]; q++) M->val[q] *= srow[M->rowind[q]];
        for (int j = 0; j < cols; j++) for (int q = M->colptr[j]; q < M->colptr[j + 1]; q++) M->val[q] *= scol[j];
    }
}

/* dense/preconditioner.hpp:62-222 and sparse/preconditioner.hpp:65-250 */
static void ruiz_scale_data(ruiz *r, orc_data *d, int reuse_prev_scaling, int scale_cost, int max_iter, double epsilon)
{
    int n = r->n, p = r->p, m = r->m, N = n + p + m;
    if (!reuse_prev_scaling) {
        r->c = 1.0;
        for (int i = 0; i < N; i++) r->delta[i] = 1.0;
        for (int i = 0; i < n; i++) r->delta_b[i] = 1.0;
        double *delta_iter = r->delta_inv, *delta_iter_b = r->delta_b_inv;
        for (int i = 0; i < N; i++) delta_iter[i] = 0.0;
        for (int i = 0; i < n; i++) delta_iter_b[i] = 0.0;
        for (int it = 0; it < max_iter; it++) {
            double dev = 0.0;
            for (int i = 0; i < N; i++) dev = dmax(dev, fabs(1.0 - delta_iter[i]));
            for (int i = 0; i < n; i++) dev = dmax(dev, fabs(1.0 - delta_iter_b[i]));
            if (!(dev > epsilon)) break;

            if (!d->is_sparse) {
                for (int k = 0; k < n; k++) {
                    double v = 0.0;
                    for (int i = 0; i < k; i++) v = dmax(v, fabs(d->P_utri[i + (size_t)k * n]));
                    for (int j = k; j < n; j++) v = dmax(v, fabs(d->P_utri[k + (size_t)j * n]));
                    if (p > 0) for (int j = 0; j < p; j++) v = dmax(v, fabs(d->AT[k + (size_t)j * n]));
                    if (m > 0) for (int j = 0; j < m; j++) v = dmax(v, fabs(d->GT[k + (size_t)j * n]));
                    v = dmax(v, d->x_b_scaling[k]);
                    delta_iter[k] = v;
                }
                for (int k = 0; k < p; k++) { double v = 0.0; for (int i = 0; i < n; i++) v = dmax(v, fabs(d->AT[i + (size_t)k * n])); delta_iter[n + k] = v; }
                for (int k = 0; k < m; k++) { double v = 0.0; for (int i = 0; i < n; i++) v = dmax(v, fabs(d->GT[i + (size_t)k * n])); delta_iter[n + p + k] = v; }
            } else {
                for (int i = 0; i < N; i++) delta_iter[i] = 0.0;
                for (int j = 0; j < n; j++) {
                    for (int q = d->sP_utri.colptr[j]; q < d->sP_utri.colptr[j + 1]; q++) {
                        int i_row = d->sP_utri.rowind[q]; double a = fabs(d->sP_utri.val[q]);
                        delta_iter[j] = dmax(delta_iter[j], a);
                        if (i_row != j) delta_iter[i_row] = dmax(delta_iter[i_row], a);
                    }
                    delta_iter[j] = dmax(delta_iter[j], d->x_b_scaling[j]);
                }
                for (int j = 0; j < p; j++) for (int q = d->sAT.colptr[j]; q < d->sAT.colptr[j + 1]; q++) {
                    int i_row = d->sAT.rowind[q]; double a = fabs(d->sAT.val[q]);
                    delta_iter[i_row] = dmax(delta_iter[i_row], a); delta_iter[n + j] = dmax(delta_iter[n + j], a);
                }
                for (int j = 0; j < m; j++) for (int q = d->sGT.colptr[j]; q < d->sGT.colptr[j + 1]; q++) {
                    int i_row = d->sGT.rowind[q]; double a = fabs(d->sGT.val[q]);
                    delta_iter[i_row] = dmax(delta_iter[i_row], a); delta_iter[n + p + j] = dmax(delta_iter[n + p + j], a);
                }
            }
            for (int i = 0; i < n; i++) delta_iter_b[i] = d->x_b_scaling[i];

            for (int i = 0; i < N; i++) delta_iter[i] = 1.0 / sqrt(limit_scaling(delta_iter[i]));
            for (int i = 0; i < n; i++) delta_iter_b[i] = 1.0 / sqrt(limit_scaling(delta_iter_b[i]));

            scale_P(d, delta_iter);
            for (int i = 0; i < n; i++) d->c[i] *= delta_iter[i];
            scale_T(d, 0, delta_iter, delta_iter + n);
            scale_T(d, 1, delta_iter, delta_iter + n + p);
            for (int i = 0; i < n; i++) d->x_b_scaling[i] *= delta_iter_b[i] * delta_iter[i];
            for (int i = 0; i < N; i++) r->delta[i] *= delta_iter[i];
            for (int i = 0; i < n; i++) r->delta_b[i] *= delta_iter_b[i];

            if (scale_cost) {
                double gamma = 0.0;
                if (!d->is_sparse) {
                    for (int k = 0; k < n; k++) {
                        double a = 0.0, b = 0.0;
                        for (int i = 0; i < k; i++) a = dmax(a, fabs(d->P_utri[i + (size_t)k * n]));
                        for (int j = k; j < n; j++) b = dmax(b, fabs(d->P_utri[k + (size_t)j * n]));
                        gamma += dmax(a, b);
                    }
                } else {
                    double *tmp = dz(n);
                    for (int j = 0; j < n; j++) for (int q = d->sP_utri.colptr[j]; q < d->sP_utri.colptr[j + 1]; q++) {
                        int i_row = d->sP_utri.rowind[q]; double a = fabs(d->sP_utri.val[q]);
                        tmp[j] = dmax(tmp[j], a); if (i_row != j) tmp[i_row] = dmax(tmp[i_row], a);
                    }
                    for (int j = 0; j < n; j++) gamma += tmp[j];
                    free(tmp);
                }
                gamma /= (double)n;
                gamma = limit_scaling(gamma);
                double cinf = 0.0; for (int i = 0; i < n; i++) cinf = dmax(cinf, fabs(d->c[i]));
                gamma = dmax(gamma, cinf);
                gamma = limit_scaling(gamma);
                gamma = 1.0 / gamma;
                scale_P_scalar(d, gamma);
                for (int i = 0; i < n; i++) d->c[i] *= gamma;
                r->c *= gamma;
            }
        }
        r->c_inv = 1.0 / r->c;
        for (int i = 0; i < N; i++) r->delta_inv[i] = 1.0 / r->delta[i];
        for (int i = 0; i < n; i++) r->delta_b_inv[i] = 1.0 / r->delta_b[i];
    } else {
        scale_P_scalar(d, r->c);
        scale_P(d, r->delta);
        for (int i = 0; i < n; i++) d->c[i] *= r->c * r->delta[i];
        scale_T(d, 0, r->delta, r->delta + n);
        scale_T(d, 1, r->delta, r->delta + n + p);
        for (int i = 0; i < n; i++) d->x_b_scaling[i] *= r->delta_b[i] * r->delta[i];
    }
    for (int i = 0; i < p; i++) d->b[i] *= r->delta[n + i];
    for (int i = 0; i < m; i++) { d->h_l[i] *= r->delta[n + p + i]; d->h_u[i] *= r->delta[n + p + i]; }
    for (int i = 0; i < d->n_x_l; i++) d->x_l[i] *= r->delta_b[d->x_l_idx[i]];
    for (int i = 0; i < d->n_x_u; i++) d->x_u[i] *= r->delta_b[d->x_u_idx[i]];
}

/* dense/preconditioner.hpp:224-258 */
static void ruiz_unscale_data(ruiz *r, orc_data *d)
{
    int n = r->n, p = r->p, m = r->m;
    scale_P_scalar(d, r->c_inv);
    scale_P(d, r->delta_inv);
    for (int i = 0; i < n; i++) d->c[i] *= r->c_inv * r->delta_inv[i];
    scale_T(d, 0, r->delta_inv, r->delta_inv + n);
    scale_T(d, 1, r->delta_inv, r->delta_inv + n + p);
    for (int i = 0; i < n; i++) d->x_b_scaling[i] *= r->delta_b_inv[i] * r->delta_inv[i];
    for (int i = 0; i < p; i++) d->b[i] *= r->delta_inv[n + i];
    for (int i = 0; i < m; i++) { d->h_l[i] *= r->delta_inv[n + p + i]; d->h_u[i] *= r->delta_inv[n + p + i]; }
    for (int i = 0; i < d->n_x_l; i++) d->x_l[i] *= r->delta_b_inv[d->x_l_idx[i]];
    for (int i = 0; i < d->n_x_u; i++) d->x_u[i] *= r->delta_b_inv[d->x_u_idx[i]];
}

/* ---------------------------------------------------------------- solver */
struct orc_solver {
    orc_vars result;      /* m_result (Variables part) */
    orc_info info;
    orc_settings settings;
    orc_data *data;
    ruiz precond;
    orc_kkt_system *kkt;
    int first_run, setup_done, enable_iterative_refinement;
    /* res_nr (BasicVariables), res, step, prox_vars (BasicVariables) */
    orc_vars res_nr, res, step, prox;
    double *trace; int trace_max, trace_rows;
    orc_state_cb cb; void *cb_user;
};

static void vars_alloc(orc_vars *v, int n, int p, int m)
{
    v->x = dz(n); v->y = dz(p); v->z_l = dz(m); v->z_u = dz(m); v->z_bl = dz(n); v->z_bu = dz(n);
    v->s_l = dz(m); v->s_u = dz(m); v->s_bl = dz(n); v->s_bu = dz(n);
}
static void vars_free(orc_vars *v)
{
    free(v->x); free(v->y); free(v->z_l); free(v->z_u); free(v->z_bl); free(v->z_bu);
    free(v->s_l); free(v->s_u); free(v->s_bl); free(v->s_bu);
    memset(v, 0, sizeof(*v));
}
static void vars_copy(orc_vars *dst, const orc_vars *src, int n, int p, int m)
{
    dst->x = dd(src->x, n); dst->y = dd(src->y, p); dst->z_l = dd(src->z_l, m); dst->z_u = dd(src->z_u, m);
    dst->z_bl = dd(src->z_bl, n); dst->z_bu = dd(src->z_bu, n); dst->s_l = dd(src->s_l, m); dst->s_u = dd(src->s_u, m);
    dst->s_bl = dd(src->s_bl, n); dst->s_bu = dd(src->s_bu, n);
}

orc_solver *orc_solver_create(void)
{
    orc_solver *s = (orc_solver *)calloc(1, sizeof(*s));
    orc_settings_default(&s->settings);
    s->first_run = 1;
    return s;
}

static void solver_release(orc_solver *s)
{
    if (s->data) {
        vars_free(&s->result); vars_free(&s->res_nr); vars_free(&s->res); vars_free(&s->step); vars_free(&s->prox);
    }
    orc_kkt_system_free(s->kkt); s->kkt = NULL;
    ruiz_free(&s->precond);
    orc_data_free(s->data); s->data = NULL;
}
void orc_solver_free(orc_solver *s) { if (!s) return; solver_release(s); free(s); }

/* copy-construction of a solver (tests/src/dense/solver_test.cpp:379-401 CopyConstructor) */
orc_solver *orc_solver_clone(const orc_solver *o)
{
    orc_solver *s = (orc_solver *)calloc(1, sizeof(*s));
    *s = *o;
    s->trace = NULL; s->trace_max = s->trace_rows = 0; s->cb = NULL; s->cb_user = NULL;
    if (o->data) {
        int n = o->data->n, p = o->data->p, m = o->data->m;
        s->data = orc_data_clone(o->data);
        vars_copy(&s->result, &o->result, n, p, m); vars_copy(&s->res_nr, &o->res_nr, n, p, m);
        vars_copy(&s->res, &o->res, n, p, m); vars_copy(&s->step, &o->step, n, p, m); vars_copy(&s->prox, &o->prox, n, p, m);
        ruiz_copy(&s->precond, &o->precond);
        s->kkt = o->kkt ? orc_kkt_system_clone(o->kkt) : NULL;
    }
    return s;
}

orc_settings *orc_solver_settings(orc_solver *s) { return &s->settings; }
const orc_info *orc_solver_info(const orc_solver *s) { return &s->info; }
const orc_vars *orc_solver_result(const orc_solver *s) { return &s->result; }
const orc_data *orc_solver_data(const orc_solver *s) { return s->data; }
void orc_solver_set_trace(orc_solver *s, double *buf, int max_rows) { s->trace = buf; s->trace_max = max_rows; s->trace_rows = 0; }
int orc_solver_trace_rows(const orc_solver *s) { return s->trace_rows; }
void orc_solver_set_state_callback(orc_solver *s, orc_state_cb cb, void *user) { s->cb = cb; s->cb_user = user; }

/* solver.hpp:151-216 (data already holds P_utri/AT/GT/bounds, i.e. lines 169-192 are orc_data_create_*) */
int orc_solver_setup(orc_solver *s, orc_data *data)
{
    double t0 = now_s();
    solver_release(s);
    s->data = data;
    int n = data->n, p = data->p, m = data->m;
    /* init_workspace :361-377 */
    vars_alloc(&s->result, n, p, m); vars_alloc(&s->res_nr, n, p, m); vars_alloc(&s->res, n, p, m);
    vars_alloc(&s->step, n, p, m); vars_alloc(&s->prox, n, p, m);
    memset(&s->info, 0, sizeof(s->info));
    s->info.rho = s->settings.rho_init; s->info.delta = s->settings.delta_init;
    ruiz_init(&s->precond, data);
    ruiz_scale_data(&s->precond, data, 0, s->settings.preconditioner_scale_cost, s->settings.preconditioner_iter, 1e-3);
    s->kkt = orc_kkt_system_create(data, &s->settings);
    if (!s->kkt) { s->setup_done = 0; return 0; }
    s->first_run = 1; s->setup_done = 1;
    s->info.setup_time = now_s() - t0;
    return 1;
}

/* solver.hpp:218-308, dense flavour of update_P/A/G (:311-351) */
int orc_solver_update_dense(orc_solver *s, const double *P, const double *c, const double *A, const double *b,
                            const double *G, const double *h_l, const double *h_u, const double *x_l, const double *x_u)
{
    if (!s->setup_done) { fprintf(stderr, "Solver not setup yet\n"); return 0; }
    double t0 = now_s();
    orc_data *d = s->data;
    int n = d->n, p = d->p, m = d->m;
    ruiz_unscale_data(&s->precond, d);
    int opt = ORC_KKT_UPDATE_NONE;
    if (P) { memset(d->P_utri, 0, sizeof(double) * (size_t)n * n);
             for (int j = 0; j < n; j++) for (int i = 0; i <= j; i++) d->P_utri[i + (size_t)j * n] = P[i + (size_t)j * n];
             opt |= ORC_KKT_UPDATE_P; }
    if (A) { for (int k = 0; k < p; k++) for (int i = 0; i < n; i++) d->AT[i + (size_t)k * n] = A[k + (size_t)i * p]; opt |= ORC_KKT_UPDATE_A; }
    if (G) { for (int k = 0; k < m; k++) for (int i = 0; i < n; i++) d->GT[i + (size_t)k * n] = G[k + (size_t)i * m]; opt |= ORC_KKT_UPDATE_G; }
    if (c) memcpy(d->c, c, sizeof(double) * (size_t)n);
    if (b) memcpy(d->b, b, sizeof(double) * (size_t)p);
    if (h_l) orc_data_set_h_l(d, h_l);
    if (h_u) orc_data_set_h_u(d, h_u);
    if (h_l || h_u) orc_data_disable_inf_constraints(d);
    if (x_l) orc_data_set_x_l(d, x_l);
    if (x_u) orc_data_set_x_u(d, x_u);
    int reuse = s->settings.preconditioner_reuse_on_update;
    if (opt == ORC_KKT_UPDATE_NONE) reuse = 1;
    ruiz_scale_data(&s->precond, d, reuse, s->settings.preconditioner_scale_cost, s->settings.preconditioner_iter, 1e-3);
    orc_kkt_system_update_data(s->kkt, d, opt);
    s->info.update_time = now_s() - t0;
    return 1;
}

/* solver.hpp:218-308, sparse flavour (:317-358): same sparsity required; A/G given as CSC of A/G (not transposed) */
int orc_solver_update_sparse(orc_solver *s, const int *Pp, const int *Pi, const double *Px, const double *c,
                             const int *Ap, const int *Ai, const double *Ax, const double *b, const int *Gp,
                             const int *Gi, const double *Gx, const double *h_l, const double *h_u,
                             const double *x_l, const double *x_u)
{
    if (!s->setup_done) { fprintf(stderr, "Solver not setup yet\n"); return 0; }
    double t0 = now_s();
    orc_data *d = s->data;
    int n = d->n, p = d->p, m = d->m;
    ruiz_unscale_data(&s->precond, d);
    int opt = ORC_KKT_UPDATE_NONE;
    if (Px) {
        /* :318-329: first P_utri_col_nnz entries of each column of P are its upper-triangular part */
        for (int j = 0; j < n; j++) {
            int P_col_nnz = Pp[j + 1] - Pp[j];
            int U_col_nnz = d->sP_utri.colptr[j + 1] - d->sP_utri.colptr[j];
            if (P_col_nnz < U_col_nnz) { fprintf(stderr, "P nonzeros missmatch\n"); return 0; }
            memcpy(d->sP_utri.val + d->sP_utri.colptr[j], Px + Pp[j], sizeof(double) * (size_t)U_col_nnz);
        }
        (void)Pi;
        opt |= ORC_KKT_UPDATE_P;
    }
    if (Ax) {
        if (Ap[n] != d->sAT.colptr[p]) { fprintf(stderr, "A nonzeros missmatch\n"); return 0; }
        /* sparse/utils.hpp:138-163 transpose_no_allocation */
        int *next = (int *)malloc(sizeof(int) * (size_t)(p + 1)); memcpy(next, d->sAT.colptr, sizeof(int) * (size_t)(p + 1));
        for (int j = 0; j < n; j++) for (int q = Ap[j]; q < Ap[j + 1]; q++) { int t = next[Ai[q]]++; d->sAT.rowind[t] = j; d->sAT.val[t] = Ax[q]; }
        free(next);
        opt |= ORC_KKT_UPDATE_A;
    }
    if (Gx) {
        if (Gp[n] != d->sGT.colptr[m]) { fprintf(stderr, "G nonzeros missmatch\n"); return 0; }
        int *next = (int *)malloc(sizeof(int) * (size_t)(m + 1)); memcpy(next, d->sGT.colptr, sizeof(int) * (size_t)(m + 1));
        for (int j = 0; j < n; j++) for (int q = Gp[j]; q < Gp[j + 1]; q++) { int t = next[Gi[q]]++; d->sGT.rowind[t] = j; d->sGT.val[t] = Gx[q]; }
        free(next);
        opt |= ORC_KKT_UPDATE_G;
    }
    if (c) memcpy(d->c, c, sizeof(double) * (size_t)n);
    if (b) memcpy(d->b, b, sizeof(double) * (size_t)p);
    if (h_l) orc_data_set_h_l(d, h_l);
    if (h_u) orc_data_set_h_u(d, h_u);
    if (h_l || h_u) orc_data_disable_inf_constraints(d);
    if (x_l) orc_data_set_x_l(d, x_l);
    if (x_u) orc_data_set_x_u(d, x_u);
    int reuse = s->settings.preconditioner_reuse_on_update;
    if (opt == ORC_KKT_UPDATE_NONE) reuse = 1;
    ruiz_scale_data(&s->precond, d, reuse, s->settings.preconditioner_scale_cost, s->settings.preconditioner_iter, 1e-3);
    orc_kkt_system_update_data(s->kkt, d, opt);
    s->info.update_time = now_s() - t0;
    return 1;
}

static double dot(const double *a, const double *b, int n) { double s = 0.0; for (int i = 0; i < n; i++) s += a[i] * b[i]; return s; }
static double min_coeff(const double *a, int n) { double r = a[0]; for (int i = 1; i < n; i++) if (a[i] < r) r = a[i]; return r; }

/* solver.hpp:884-891 */
static double calculate_mu(orc_solver *s)
{
    const orc_data *d = s->data; const orc_vars *r = &s->result;
    return (dot(r->s_l, r->z_l, d->m) + dot(r->s_u, r->z_u, d->m) + dot(r->s_bl, r->z_bl, d->n_x_l) + dot(r->s_bu, r->z_bu, d->n_x_u))
           / (double)(d->n_h_l + d->n_h_u + d->n_x_l + d->n_x_u);
}

/* solver.hpp:893-958 */
static void calculate_step(orc_solver *s, double *alpha_s, double *alpha_z)
{
    const orc_data *d = s->data; const orc_vars *r = &s->result; const orc_vars *st = &s->step;
    double as = 1.0, az = 1.0;
    for (int i = 0; i < d->m; i++) {
        if (st->s_l[i] < 0) as = dmin(as, -r->s_l[i] / st->s_l[i]);
        if (st->s_u[i] < 0) as = dmin(as, -r->s_u[i] / st->s_u[i]);
        if (st->z_l[i] < 0) az = dmin(az, -r->z_l[i] / st->z_l[i]);
        if (st->z_u[i] < 0) az = dmin(az, -r->z_u[i] / st->z_u[i]);
    }
    for (int i = 0; i < d->n_x_l; i++) {
        if (st->s_bl[i] < 0) as = dmin(as, -r->s_bl[i] / st->s_bl[i]);
        if (st->z_bl[i] < 0) az = dmin(az, -r->z_bl[i] / st->z_bl[i]);
    }
    for (int i = 0; i < d->n_x_u; i++) {
        if (st->s_bu[i] < 0) as = dmin(as, -r->s_bu[i] / st->s_bu[i]);
        if (st->z_bu[i] < 0) az = dmin(az, -r->z_bu[i] / st->z_bu[i]);
    }
    *alpha_s = as; *alpha_z = az;
}

/* preconditioner unscale helpers (dense/preconditioner.hpp:260-470) */
#define P_DELTA(s) ((s)->precond.delta)
#define P_DINV(s) ((s)->precond.delta_inv)

static double inf_scaled(const double *v, const double *sc, double c, int n)
{
    double r = 0.0;
    for (int i = 0; i < n; i++) { double a = fabs(v[i] * c * sc[i]); if (a > r || a != a) r = a; }
    return r;
}

/* solver.hpp:1130-1203 */
static double primal_res_of(orc_solver *s, const orc_vars *v)
{
    const orc_data *d = s->data; int n = d->n, p = d->p, m = d->m;
    double inf = inf_scaled(v->y, P_DINV(s) + n, 1.0, p);
    inf = dmax(inf, inf_scaled(v->z_l, P_DINV(s) + n + p, 1.0, m));
    inf = dmax(inf, inf_scaled(v->z_u, P_DINV(s) + n + p, 1.0, m));
    for (int i = 0; i < d->n_x_l; i++) inf = dmax(inf, v->z_bl[i] * s->precond.delta_b_inv[d->x_l_idx[i]]);
    for (int i = 0; i < d->n_x_u; i++) inf = dmax(inf, v->z_bu[i] * s->precond.delta_b_inv[d->x_u_idx[i]]);
    return inf;
}
static double dual_res_of(orc_solver *s, const double *x) { return inf_scaled(x, P_DINV(s), s->precond.c_inv, s->data->n); }

static double primal_prox_inf(orc_solver *s)
{
    const orc_data *d = s->data; int n = d->n, p = d->p, m = d->m;
    const orc_vars *r = &s->result, *px = &s->prox;
    double ci = s->precond.c_inv, inf = 0.0;
    for (int i = 0; i < p; i++) inf = dmax(inf, fabs((px->y[i] - r->y[i]) * ci * P_DELTA(s)[n + i]));
    for (int i = 0; i < m; i++) inf = dmax(inf, fabs((px->z_l[i] - r->z_l[i]) * ci * P_DELTA(s)[n + p + i]));
    for (int i = 0; i < m; i++) inf = dmax(inf, fabs((px->z_u[i] - r->z_u[i]) * ci * P_DELTA(s)[n + p + i]));
    for (int i = 0; i < d->n_x_l; i++) inf = dmax(inf, (px->z_bl[i] - r->z_bl[i]) * ci * s->precond.delta_b[d->x_l_idx[i]]);
    for (int i = 0; i < d->n_x_u; i++) inf = dmax(inf, (px->z_bu[i] - r->z_bu[i]) * ci * s->precond.delta_b[d->x_u_idx[i]]);
    return inf;
}
static double dual_prox_inf(orc_solver *s)
{
    int n = s->data->n; double inf = 0.0;
    for (int i = 0; i < n; i++) inf = dmax(inf, fabs((s->result.x[i] - s->prox.x[i]) * P_DELTA(s)[i]));
    return inf;
}

/* solver.hpp:960-1105 */
static void update_residuals_nr(orc_solver *s)
{
    const orc_data *d = s->data; int n = d->n, p = d->p, m = d->m;
    orc_vars *r = &s->result, *nr = &s->res_nr;
    orc_kkt *b = orc_kkt_system_backend(s->kkt);
    double *work_x = s->step.x, *work_z = s->step.z_l;
    double ci = s->precond.c_inv;
    const double *dinv = P_DINV(s);

    b->eval_A_xn_and_AT_xt(b, d, -1.0, 1.0, r->x, r->y, nr->y, work_x);
    for (int i = 0; i < m; i++) work_z[i] = r->z_u[i] - r->z_l[i];
    double *work_x_2 = nr->x;
    b->eval_G_xn_and_GT_xt(b, d, 1.0, 1.0, r->x, work_z, nr->z_l, work_x_2);
    for (int i = 0; i < m; i++) nr->z_u[i] = -nr->z_l[i];
    for (int i = 0; i < n; i++) work_x[i] += work_x_2[i];

    b->eval_P_x(b, d, -1.0, r->x, nr->x);
    double dual_rel_norm = inf_scaled(nr->x, dinv, ci, n);

    double tmp = -dot(r->x, nr->x, n);
    s->info.primal_obj = 0.5 * tmp;
    s->info.dual_obj = -0.5 * tmp;
    double dg_rel = ci * fabs(tmp);
    tmp = dot(d->c, r->x, n);
    s->info.primal_obj += tmp;
    dg_rel = dmax(dg_rel, ci * fabs(tmp));
    tmp = dot(d->b, r->y, p);
    s->info.dual_obj -= tmp;
    dg_rel = dmax(dg_rel, ci * fabs(tmp));
    tmp = -dot(d->h_l, r->z_l, m);
    s->info.dual_obj -= tmp;
    dg_rel = dmax(dg_rel, ci * fabs(tmp));
    tmp = dot(d->h_u, r->z_u, m);
    s->info.dual_obj -= tmp;
    dg_rel = dmax(dg_rel, ci * fabs(tmp));
    tmp = -dot(d->x_l, r->z_bl, d->n_x_l);
    s->info.dual_obj -= tmp;
    dg_rel = dmax(dg_rel, ci * fabs(tmp));
    tmp = dot(d->x_u, r->z_bu, d->n_x_u);
    s->info.dual_obj -= tmp;
    dg_rel = dmax(dg_rel, ci * fabs(tmp));

    s->info.duality_gap = fabs(s->info.primal_obj - s->info.dual_obj);
    s->info.primal_obj = ci * s->info.primal_obj;
    s->info.dual_obj = ci * s->info.dual_obj;
    s->info.duality_gap = ci * s->info.duality_gap;
    s->info.duality_gap_rel = s->info.duality_gap / dmax(1.0, dg_rel);

    for (int i = 0; i < n; i++) nr->x[i] -= d->c[i];
    dual_rel_norm = dmax(dual_rel_norm, inf_scaled(d->c, dinv, ci, n));
    for (int i = 0; i < d->n_x_l; i++) { int idx = d->x_l_idx[i]; work_x[idx] -= d->x_b_scaling[idx] * r->z_bl[i]; }
    for (int i = 0; i < d->n_x_u; i++) { int idx = d->x_u_idx[i]; work_x[idx] += d->x_b_scaling[idx] * r->z_bu[i]; }
    dual_rel_norm = dmax(dual_rel_norm, inf_scaled(work_x, dinv, ci, n));
    for (int i = 0; i < n; i++) nr->x[i] -= work_x[i];

    double primal_rel_norm = inf_scaled(nr->y, dinv + n, 1.0, p);
    for (int i = 0; i < p; i++) nr->y[i] += d->b[i];
    primal_rel_norm = dmax(primal_rel_norm, inf_scaled(d->b, dinv + n, 1.0, p));

    const double *dz_inv = dinv + n + p;
    int i = 0;
    for (int ii = 0; ii < d->n_h_l; ii++) {
        int idx = d->h_l_idx[ii];
        while (i < idx) nr->z_l[i++] = 0.0;
        /* NB: the reference takes max with the SIGNED scaled value here (no abs), :1047-1050 */
        primal_rel_norm = dmax(primal_rel_norm, nr->z_l[i] * dz_inv[i]);
        nr->z_l[i] += -d->h_l[i] - r->s_l[i];
        primal_rel_norm = dmax(primal_rel_norm, d->h_l[i] * dz_inv[i]);
        primal_rel_norm = dmax(primal_rel_norm, r->s_l[i] * dz_inv[i]);
        i++;
    }
    while (i < m) nr->z_l[i++] = 0.0;
    i = 0;
    for (int ii = 0; ii < d->n_h_u; ii++) {
        int idx = d->h_u_idx[ii];
        while (i < idx) nr->z_u[i++] = 0.0;
        primal_rel_norm = dmax(primal_rel_norm, nr->z_u[i] * dz_inv[i]);
        nr->z_u[i] += d->h_u[i] - r->s_u[i];
        primal_rel_norm = dmax(primal_rel_norm, d->h_u[i] * dz_inv[i]);
        primal_rel_norm = dmax(primal_rel_norm, r->s_u[i] * dz_inv[i]);
        i++;
    }
    while (i < m) nr->z_u[i++] = 0.0;

    const double *dbi = s->precond.delta_b_inv;
    for (i = 0; i < d->n_x_l; i++) {
        int idx = d->x_l_idx[i];
        nr->z_bl[i] = d->x_b_scaling[idx] * r->x[idx];
        primal_rel_norm = dmax(primal_rel_norm, nr->z_bl[i] * dbi[idx]);
        primal_rel_norm = dmax(primal_rel_norm, d->x_l[i] * dbi[idx]);
        primal_rel_norm = dmax(primal_rel_norm, r->s_bl[i] * dbi[idx]);
    }
    for (i = 0; i < d->n_x_l; i++) nr->z_bl[i] += -d->x_l[i] - r->s_bl[i];
    for (i = 0; i < d->n_x_u; i++) {
        int idx = d->x_u_idx[i];
        nr->z_bu[i] = -d->x_b_scaling[idx] * r->x[idx];
        primal_rel_norm = dmax(primal_rel_norm, nr->z_bu[i] * dbi[idx]);
        primal_rel_norm = dmax(primal_rel_norm, d->x_u[i] * dbi[idx]);
        primal_rel_norm = dmax(primal_rel_norm, r->s_bu[i] * dbi[idx]);
    }
    for (i = 0; i < d->n_x_u; i++) nr->z_bu[i] += d->x_u[i] - r->s_bu[i];

    s->info.prev_primal_res = s->info.primal_res;
    s->info.prev_dual_res = s->info.dual_res;
    s->info.primal_res = primal_res_of(s, nr);
    s->info.primal_res_rel = s->info.primal_res / dmax(1.0, primal_rel_norm);
    s->info.dual_res = dual_res_of(s, nr->x);
    s->info.dual_res_rel = s->info.dual_res / dmax(1.0, dual_rel_norm);
}

/* solver.hpp:1107-1128 */
static void update_residuals_r(orc_solver *s)
{
    const orc_data *d = s->data; int n = d->n, p = d->p, m = d->m;
    orc_vars *r = &s->result, *nr = &s->res_nr, *res = &s->res, *px = &s->prox;
    double rho = s->info.rho, delta = s->info.delta;
    for (int i = 0; i < n; i++) res->x[i] = nr->x[i] - rho * (r->x[i] - px->x[i]);
    for (int i = 0; i < p; i++) res->y[i] = nr->y[i] - delta * (px->y[i] - r->y[i]);
    for (int i = 0; i < m; i++) res->z_l[i] = nr->z_l[i] - delta * (px->z_l[i] - r->z_l[i]);
    for (int i = 0; i < m; i++) res->z_u[i] = nr->z_u[i] - delta * (px->z_u[i] - r->z_u[i]);
    for (int i = 0; i < d->n_x_l; i++) res->z_bl[i] = nr->z_bl[i] - delta * (px->z_bl[i] - r->z_bl[i]);
    for (int i = 0; i < d->n_x_u; i++) res->z_bu[i] = nr->z_bu[i] - delta * (px->z_bu[i] - r->z_bu[i]);

    double primal_rel_scaling = s->info.primal_res_rel > 0 ? s->info.primal_res / s->info.primal_res_rel : 1.0;
    double dual_rel_scaling = s->info.dual_res_rel > 0 ? s->info.dual_res / s->info.dual_res_rel : 1.0;
    s->info.primal_res_reg = primal_res_of(s, res);
    s->info.primal_res_reg_rel = s->info.primal_res_reg / primal_rel_scaling;
    s->info.dual_res_reg = dual_res_of(s, res->x);
    s->info.dual_res_reg_rel = s->info.dual_res_reg / dual_rel_scaling;
    s->info.primal_prox_inf = primal_prox_inf(s) * s->info.delta;
    s->info.dual_prox_inf = dual_prox_inf(s) * s->info.rho;
}

static int kkt_factor(orc_solver *s)
{
    s->info.n_factor++;
    if (s->cb) s->cb(s->cb_user, 0, s->enable_iterative_refinement, s->info.rho, s->info.delta, &s->result);
    return orc_kkt_system_update_scalings_and_factor(s->kkt, s->data, &s->settings, s->enable_iterative_refinement,
                                                     s->info.rho, s->info.delta, &s->result);
}
static void kkt_solve(orc_solver *s, const orc_vars *rhs, orc_vars *lhs)
{
    s->info.n_solve++;
    if (s->cb) s->cb(s->cb_user, 1, s->enable_iterative_refinement, s->info.rho, s->info.delta, rhs);
    orc_kkt_system_solve(s->kkt, s->data, &s->settings, rhs, lhs);
}

/* solver.hpp:379-882 */
static int solve_impl(orc_solver *s)
{
    orc_info *info = &s->info;
    const orc_settings *set = &s->settings;
    if (!s->setup_done) { fprintf(stderr, "Solver not setup yet\n"); info->status = ORC_UNSOLVED; return info->status; }
    if (!verify_settings(set)) { info->status = ORC_INVALID_SETTINGS; return info->status; }
    const orc_data *d = s->data;
    int n = d->n, p = d->p, m = d->m;
    orc_vars *r = &s->result, *res = &s->res, *step = &s->step, *px = &s->prox;
    double t0;

    info->kkt_factor_time = 0; info->kkt_solve_time = 0;
    info->n_factor = info->n_solve = 0;
    info->status = ORC_UNSOLVED;
    info->iter = 0;
    info->reg_limit = set->reg_lower_limit;
    info->factor_retires = 0; info->no_primal_update = 0; info->no_dual_update = 0;
    info->mu = 0; info->primal_step = 0; info->dual_step = 0;
    info->rho = set->rho_init; info->delta = set->delta_init;

    for (int i = 0; i < m; i++) { r->s_l[i] = r->s_u[i] = r->z_l[i] = r->z_u[i] = 0.0; }
    for (int i = 0; i < d->n_h_l; i++) { int idx = d->h_l_idx[i]; r->s_l[idx] = 1.0; r->z_l[idx] = 1.0; }
    for (int i = 0; i < d->n_h_u; i++) { int idx = d->h_u_idx[i]; r->s_u[idx] = 1.0; r->z_u[idx] = 1.0; }
    for (int i = 0; i < d->n_x_l; i++) { r->s_bl[i] = 1.0; r->z_bl[i] = 1.0; }
    for (int i = 0; i < d->n_x_u; i++) { r->s_bu[i] = 1.0; r->z_bu[i] = 1.0; }

    s->enable_iterative_refinement = set->iterative_refinement_always_enabled;

    t0 = now_s();
    while (!kkt_factor(s)) {
        if (!s->enable_iterative_refinement) s->enable_iterative_refinement = 1;
        else if (info->factor_retires < set->max_factor_retires) {
            info->delta *= 100; info->rho *= 100; info->factor_retires++;
            info->reg_limit = dmin(10 * info->reg_limit, set->eps_abs);
        } else { info->status = ORC_NUMERICS; return info->status; }
    }
    info->factor_retires = 0;
    info->kkt_factor_time += now_s() - t0;

    for (int i = 0; i < n; i++) res->x[i] = -d->c[i];
    for (int i = 0; i < p; i++) res->y[i] = d->b[i];
    for (int i = 0; i < m; i++) { res->z_l[i] = -d->h_l[i]; res->z_u[i] = d->h_u[i]; }
    /* res.z_bl = -x_l, res.z_bu = x_u : full-length copies (entries past n_x_l are stale bounds, unused) */
    for (int i = 0; i < n; i++) { res->z_bl[i] = -d->x_l[i]; res->z_bu[i] = d->x_u[i]; }
    for (int i = 0; i < m; i++) { res->s_l[i] = 0.0; res->s_u[i] = 0.0; }
    for (int i = 0; i < n; i++) { res->s_bl[i] = 0.0; res->s_bu[i] = 0.0; }

    t0 = now_s();
    kkt_solve(s, res, r);
    info->kkt_solve_time += now_s() - t0;

    if (m + d->n_x_l + d->n_x_u > 0) {
        double delta_s = 0.0, delta_z = 0.0;
        if (m > 0) { delta_s = dmax(delta_s, -min_coeff(r->s_l, m)); delta_s = dmax(delta_s, -min_coeff(r->s_u, m)); }
        if (d->n_x_l > 0) delta_s = dmax(delta_s, -min_coeff(r->s_bl, d->n_x_l));
        if (d->n_x_u > 0) delta_s = dmax(delta_s, -min_coeff(r->s_bu, d->n_x_u));
        if (m > 0) { delta_z = dmax(delta_z, -min_coeff(r->z_l, m)); delta_z = dmax(delta_z, -min_coeff(r->z_u, m)); }
        if (d->n_x_l > 0) delta_z = dmax(delta_z, -min_coeff(r->z_bl, d->n_x_l));
        if (d->n_x_u > 0) delta_z = dmax(delta_z, -min_coeff(r->z_bu, d->n_x_u));

        for (int i = 0; i < d->n_h_l; i++) { int idx = d->h_l_idx[i]; r->s_l[idx] += delta_s; r->z_l[idx] += delta_z; }
        for (int i = 0; i < d->n_h_u; i++) { int idx = d->h_u_idx[i]; r->s_u[idx] += delta_s; r->z_u[idx] += delta_z; }
        for (int i = 0; i < d->n_x_l; i++) { r->s_bl[i] += delta_s; r->z_bl[i] += delta_z; }
        for (int i = 0; i < d->n_x_u; i++) { r->s_bu[i] += delta_s; r->z_bu[i] += delta_z; }

        info->mu = dmax(calculate_mu(s), 1e-10);

        for (int i = 0; i < d->n_h_l; i++) {
            int idx = d->h_l_idx[i];
            double c = r->z_l[idx] - delta_z;
            r->z_l[idx] = (c + sqrt(c * c + 4 * info->mu)) / 2;
            r->s_l[idx] = r->z_l[idx] - c;
        }
        for (int i = 0; i < d->n_h_u; i++) {
            int idx = d->h_u_idx[i];
            double c = r->z_u[idx] - delta_z;
            r->z_u[idx] = (c + sqrt(c * c + 4 * info->mu)) / 2;
            r->s_u[idx] = r->z_u[idx] - c;
        }
        for (int i = 0; i < d->n_x_l; i++) {
            double c = r->z_bl[i] - delta_z;
            r->z_bl[i] = (c + sqrt(c * c + 4 * info->mu)) / 2;
            r->s_bl[i] = r->z_bl[i] - c;
        }
        for (int i = 0; i < d->n_x_u; i++) {
            double c = r->z_bu[i] - delta_z;
            r->z_bu[i] = (c + sqrt(c * c + 4 * info->mu)) / 2;
            r->s_bu[i] = r->z_bu[i] - c;
        }
        info->mu = calculate_mu(s);
    }

    memcpy(px->x, r->x, sizeof(double) * (size_t)n);
    memcpy(px->y, r->y, sizeof(double) * (size_t)p);
    memcpy(px->z_l, r->z_l, sizeof(double) * (size_t)m);
    memcpy(px->z_u, r->z_u, sizeof(double) * (size_t)m);
    memcpy(px->z_bl, r->z_bl, sizeof(double) * (size_t)d->n_x_l);
    memcpy(px->z_bu, r->z_bu, sizeof(double) * (size_t)d->n_x_u);

    while (info->iter < set->max_iter) {
        if (info->iter == 0) {
            update_residuals_nr(s);
            info->prev_primal_res = info->primal_res;
            info->prev_dual_res = info->dual_res;
        }
        if (s->trace && s->trace_rows < s->trace_max) {
            double *row = s->trace + (size_t)s->trace_rows * 11;
            row[0] = info->iter; row[1] = info->primal_obj; row[2] = info->dual_obj; row[3] = info->duality_gap;
            row[4] = info->primal_res; row[5] = info->dual_res; row[6] = info->rho; row[7] = info->delta;
            row[8] = info->mu; row[9] = info->primal_step; row[10] = info->dual_step;
            s->trace_rows++;
        }
        if (set->verbose) {
            printf("%3d   % .5e   % .5e   %.5e   %.5e   %.5e   %.3e   %.3e   %.3e   %.4f   %.4f\n", info->iter,
                   info->primal_obj, info->dual_obj, info->duality_gap, info->primal_res, info->dual_res, info->rho,
                   info->delta, info->mu, info->primal_step, info->dual_step);
            fflush(stdout);
        }

        if ((info->primal_res < set->eps_abs || info->primal_res_rel < set->eps_rel) &&
            (info->dual_res < set->eps_abs || info->dual_res_rel < set->eps_rel) &&
            (!set->check_duality_gap || info->duality_gap < set->eps_duality_gap_abs || info->duality_gap_rel < set->eps_duality_gap_rel)) {
            info->status = ORC_SOLVED;
            return info->status;
        }

        update_residuals_r(s);

        int thr_d = set->reg_finetune_dual_update_threshold < 5 ? set->reg_finetune_dual_update_threshold : 5;
        int thr_p = set->reg_finetune_primal_update_threshold < 5 ? set->reg_finetune_primal_update_threshold : 5;
        if (info->no_dual_update > thr_d && info->primal_prox_inf > set->infeasibility_threshold &&
            (info->primal_res_reg < set->eps_abs || info->primal_res_reg_rel < set->eps_rel)) {
            info->status = ORC_PRIMAL_INFEASIBLE;
            return info->status;
        }
        if (info->no_primal_update > thr_p && info->dual_prox_inf > set->infeasibility_threshold &&
            (info->dual_res_reg < set->eps_abs || info->dual_res_reg_rel < set->eps_rel)) {
            info->status = ORC_DUAL_INFEASIBLE;
            return info->status;
        }

        info->iter++;

        int boundary_shifted = 0;
        double epsilon = DBL_EPSILON;
        for (int i = 0; i < d->n_h_l; i++) { int idx = d->h_l_idx[i]; if (r->z_l[idx] < epsilon) { r->z_l[idx] += epsilon; boundary_shifted = 1; } }
        for (int i = 0; i < d->n_h_u; i++) { int idx = d->h_u_idx[i]; if (r->z_u[idx] < epsilon) { r->z_u[idx] += epsilon; boundary_shifted = 1; } }
        if (d->n_x_l > 0 && min_coeff(r->z_bl, d->n_x_l) < epsilon) { for (int i = 0; i < d->n_x_l; i++) r->z_bl[i] += epsilon; boundary_shifted = 1; }
        if (d->n_x_u > 0 && min_coeff(r->z_bu, d->n_x_u) < epsilon) { for (int i = 0; i < d->n_x_u; i++) r->z_bu[i] += epsilon; boundary_shifted = 1; }
        if (boundary_shifted) info->mu = calculate_mu(s);

        if ((info->no_primal_update > set->reg_finetune_primal_update_threshold && info->rho == info->reg_limit &&
             info->reg_limit != set->reg_finetune_lower_limit) ||
            (info->no_dual_update > set->reg_finetune_dual_update_threshold && info->delta == info->reg_limit &&
             info->reg_limit != set->reg_finetune_lower_limit)) {
            if (info->dual_prox_inf < set->infeasibility_threshold && info->primal_prox_inf < set->infeasibility_threshold) {
                info->reg_limit = set->reg_finetune_lower_limit;
                info->no_primal_update = 0;
                info->no_dual_update = 0;
            }
        }

        t0 = now_s();
        int regularization_changed = 0;
        while (!kkt_factor(s)) {
            if (!s->enable_iterative_refinement) { s->enable_iterative_refinement = 1; continue; }
            if (info->factor_retires < set->max_factor_retires) {
                info->delta *= 100; info->rho *= 100; info->factor_retires++;
                info->reg_limit = dmin(10 * info->reg_limit, set->eps_abs);
                regularization_changed = 1;
                continue;
            }
            info->status = ORC_NUMERICS;
            return info->status;
        }
        info->factor_retires = 0;
        info->kkt_factor_time += now_s() - t0;

        if (regularization_changed) update_residuals_r(s);

        if (m + d->n_x_l + d->n_x_u > 0) {
            /* predictor */
            for (int i = 0; i < m; i++) { res->s_l[i] = -r->s_l[i] * r->z_l[i]; res->s_u[i] = -r->s_u[i] * r->z_u[i]; }
            for (int i = 0; i < d->n_x_l; i++) res->s_bl[i] = -r->s_bl[i] * r->z_bl[i];
            for (int i = 0; i < d->n_x_u; i++) res->s_bu[i] = -r->s_bu[i] * r->z_bu[i];

            t0 = now_s();
            kkt_solve(s, res, step);
            info->kkt_solve_time += now_s() - t0;

            double alpha_s, alpha_z;
            calculate_step(s, &alpha_s, &alpha_z);
            alpha_s *= set->tau; alpha_z *= set->tau;

            double sigma = 0.0, acc = 0.0;
            for (int i = 0; i < m; i++) acc += (r->s_l[i] + alpha_s * step->s_l[i]) * (r->z_l[i] + alpha_z * step->z_l[i]);
            sigma = acc; acc = 0.0;
            for (int i = 0; i < m; i++) acc += (r->s_u[i] + alpha_s * step->s_u[i]) * (r->z_u[i] + alpha_z * step->z_u[i]);
            sigma += acc; acc = 0.0;
            for (int i = 0; i < d->n_x_l; i++) acc += (r->s_bl[i] + alpha_s * step->s_bl[i]) * (r->z_bl[i] + alpha_z * step->z_bl[i]);
            sigma += acc; acc = 0.0;
            for (int i = 0; i < d->n_x_u; i++) acc += (r->s_bu[i] + alpha_s * step->s_bu[i]) * (r->z_bu[i] + alpha_z * step->z_bu[i]);
            sigma += acc;
            sigma /= (info->mu * (double)(d->n_h_l + d->n_h_u + d->n_x_l + d->n_x_u));
            sigma = dmax(0.0, dmin(1.0, sigma));
            sigma = sigma * sigma * sigma;
            info->sigma = sigma;

            /* corrector */
            double sm = info->sigma * info->mu;
            for (int i = 0; i < m; i++) { res->s_l[i] += -step->s_l[i] * step->z_l[i] + sm; res->s_u[i] += -step->s_u[i] * step->z_u[i] + sm; }
            for (int i = 0; i < d->n_x_l; i++) res->s_bl[i] += -step->s_bl[i] * step->z_bl[i] + sm;
            for (int i = 0; i < d->n_x_u; i++) res->s_bu[i] += -step->s_bu[i] * step->z_bu[i] + sm;

            t0 = now_s();
            kkt_solve(s, res, step);
            info->kkt_solve_time += now_s() - t0;

            calculate_step(s, &alpha_s, &alpha_z);
            info->primal_step = alpha_s * set->tau;
            info->dual_step = alpha_z * set->tau;

            for (int i = 0; i < n; i++) r->x[i] += info->primal_step * step->x[i];
            for (int i = 0; i < p; i++) r->y[i] += info->dual_step * step->y[i];
            for (int i = 0; i < m; i++) { r->z_l[i] += info->dual_step * step->z_l[i]; r->z_u[i] += info->dual_step * step->z_u[i]; }
            for (int i = 0; i < d->n_x_l; i++) r->z_bl[i] += info->dual_step * step->z_bl[i];
            for (int i = 0; i < d->n_x_u; i++) r->z_bu[i] += info->dual_step * step->z_bu[i];
            for (int i = 0; i < m; i++) { r->s_l[i] += info->primal_step * step->s_l[i]; r->s_u[i] += info->primal_step * step->s_u[i]; }
            for (int i = 0; i < d->n_x_l; i++) r->s_bl[i] += info->primal_step * step->s_bl[i];
            for (int i = 0; i < d->n_x_u; i++) r->s_bu[i] += info->primal_step * step->s_bu[i];

            double mu_prev = info->mu;
            info->mu = calculate_mu(s);
            double mu_rate = dmax(0.0, (mu_prev - info->mu) / mu_prev);

            update_residuals_nr(s);

            if (info->dual_res < 0.95 * info->prev_dual_res ||
                (info->dual_res < set->eps_abs || info->dual_res_rel < set->eps_rel) ||
                (info->rho == set->reg_finetune_lower_limit && info->dual_prox_inf < set->infeasibility_threshold)) {
                memcpy(px->x, r->x, sizeof(double) * (size_t)n);
                info->rho = dmax(info->reg_limit, (1.0 - mu_rate) * info->rho);
            } else {
                info->no_primal_update++;
                if (info->iter < 5 || info->dual_prox_inf < set->infeasibility_threshold)
                    info->rho = dmax(info->reg_limit, (1.0 - 0.666 * mu_rate) * info->rho);
            }
            if (info->primal_res < 0.95 * info->prev_primal_res ||
                (info->primal_res < set->eps_abs || info->primal_res_rel < set->eps_rel) ||
                (info->delta == set->reg_finetune_lower_limit && info->primal_prox_inf < set->infeasibility_threshold)) {
                memcpy(px->y, r->y, sizeof(double) * (size_t)p);
                memcpy(px->z_l, r->z_l, sizeof(double) * (size_t)m);
                memcpy(px->z_u, r->z_u, sizeof(double) * (size_t)m);
                memcpy(px->z_bl, r->z_bl, sizeof(double) * (size_t)d->n_x_l);
                memcpy(px->z_bu, r->z_bu, sizeof(double) * (size_t)d->n_x_u);
                info->delta = dmax(info->reg_limit, (1.0 - mu_rate) * info->delta);
            } else {
                info->no_dual_update++;
                if (info->iter < 5 || info->primal_prox_inf < set->infeasibility_threshold)
                    info->delta = dmax(info->reg_limit, (1.0 - 0.666 * mu_rate) * info->delta);
            }
        } else {
            t0 = now_s();
            kkt_solve(s, res, step);
            info->kkt_solve_time += now_s() - t0;
            info->primal_step = 1.0; info->dual_step = 1.0;
            for (int i = 0; i < n; i++) r->x[i] += info->primal_step * step->x[i];
            for (int i = 0; i < p; i++) r->y[i] += info->dual_step * step->y[i];

            update_residuals_nr(s);

            if (info->dual_res < 0.95 * info->prev_dual_res || (info->dual_res < set->eps_abs || info->dual_res_rel < set->eps_rel)) {
                memcpy(px->x, r->x, sizeof(double) * (size_t)n);
                info->rho = dmax(info->reg_limit, 0.1 * info->rho);
            } else {
                info->no_primal_update++;
                if (info->iter < 5 || info->dual_prox_inf < set->infeasibility_threshold) info->rho = dmax(info->reg_limit, 0.5 * info->rho);
            }
            if (info->primal_res < 0.95 * info->prev_primal_res || (info->primal_res < set->eps_abs || info->primal_res_rel < set->eps_rel)) {
                memcpy(px->y, r->y, sizeof(double) * (size_t)p);
                info->delta = dmax(info->reg_limit, 0.1 * info->delta);
            } else {
                info->no_dual_update++;
                if (info->iter < 5 || info->primal_prox_inf < set->infeasibility_threshold) info->delta = dmax(info->reg_limit, 0.5 * info->delta);
            }
        }
    }
    info->status = ORC_MAX_ITER_REACHED;
    return info->status;
}

/* solver.hpp:1205-1227 */
static void unscale_results(orc_solver *s)
{
    const orc_data *d = s->data; int n = d->n, p = d->p, m = d->m;
    orc_vars *r = &s->result; const ruiz *pc = &s->precond;
    for (int i = 0; i < n; i++) r->x[i] = r->x[i] * pc->delta[i];
    for (int i = 0; i < p; i++) r->y[i] = r->y[i] * pc->c_inv * pc->delta[n + i];
    for (int i = 0; i < m; i++) { r->z_l[i] = r->z_l[i] * pc->c_inv * pc->delta[n + p + i]; r->z_u[i] = r->z_u[i] * pc->c_inv * pc->delta[n + p + i]; }
    for (int i = 0; i < m; i++) { r->s_l[i] = r->s_l[i] * pc->delta_inv[n + p + i]; r->s_u[i] = r->s_u[i] * pc->delta_inv[n + p + i]; }
    for (int i = 0; i < d->n_x_l; i++) { int idx = d->x_l_idx[i]; r->z_bl[i] = r->z_bl[i] * pc->c_inv * pc->delta_b[idx]; r->s_bl[i] = r->s_bl[i] * pc->delta_b_inv[idx]; }
    for (int i = 0; i < d->n_x_u; i++) { int idx = d->x_u_idx[i]; r->z_bu[i] = r->z_bu[i] * pc->c_inv * pc->delta_b[idx]; r->s_bu[i] = r->s_bu[i] * pc->delta_b_inv[idx]; }
}

/* solver.hpp:1229-1259 */
static void restore_dual(orc_solver *s)
{
    const orc_data *d = s->data; int n = d->n, m = d->m;
    orc_vars *r = &s->result;
    for (int i = 0; i < m; i++) { if (r->z_l[i] == 0) r->s_l[i] = ORC_INF; if (r->z_u[i] == 0) r->s_u[i] = ORC_INF; }
    for (int i = d->n_x_l; i < n; i++) { r->z_bl[i] = 0.0; r->s_bl[i] = ORC_INF; }
    for (int i = d->n_x_u; i < n; i++) { r->z_bu[i] = 0.0; r->s_bu[i] = ORC_INF; }
    for (int i = d->n_x_l - 1; i >= 0; i--) {
        int idx = d->x_l_idx[i]; double t;
        t = r->z_bl[i]; r->z_bl[i] = r->z_bl[idx]; r->z_bl[idx] = t;
        t = r->s_bl[i]; r->s_bl[i] = r->s_bl[idx]; r->s_bl[idx] = t;
    }
    for (int i = d->n_x_u - 1; i >= 0; i--) {
        int idx = d->x_u_idx[i]; double t;
        t = r->z_bu[i]; r->z_bu[i] = r->z_bu[idx]; r->z_bu[idx] = t;
        t = r->s_bu[i]; r->s_bu[i] = r->s_bu[idx]; r->s_bu[idx] = t;
    }
}

/* solver.hpp:69-148 */
int orc_solver_solve(orc_solver *s)
{
    double t0 = now_s();
    int status = solve_impl(s);
    if (s->setup_done && status != ORC_INVALID_SETTINGS) { unscale_results(s); restore_dual(s); }
    s->info.solve_time = now_s() - t0;
    s->info.run_time = (s->first_run ? s->info.setup_time : s->info.update_time) + s->info.solve_time;
    if (s->settings.verbose) {
        printf("\nstatus:               %d\nnumber of iterations: %d\nobjective:            %.5e\n", status, s->info.iter, s->info.primal_obj);
    }
    s->first_run = 0;
    return status;
}
