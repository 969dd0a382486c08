/*
 * oracle/orc_kkt_system.c -- CPU restatement of piqp::KKTSystem (TEST INFRASTRUCTURE ONLY).
 * Follows /root/reference/include/piqp/kkt_system.hpp line by line:
 *   init :97-132, update_data :134-141, update_scalings_and_factor :143-211, solve :213-369,
 *   mul :392-425, extract_P_diag :430-453, init_kkt_solver :455-497, inf_norm :499-505,
 *   mul_condensed_kkt :507-519, get_refine_error :522-536.
 */
#include "orc.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct orc_kkt_system {
    int n, p, m;
    double m_rho, m_delta;
    double *P_diag;
    double *m_s_l, *m_s_u, *m_s_bl, *m_s_bu;
    double *m_z_l_inv, *m_z_u_inv, *m_z_bl_inv, *m_z_bu_inv;
    double *m_x_reg, *m_z_reg;
    double *rhs_x_bar, *rhs_z_bar;
    double *work_x, *work_z;
    double *ref_err_x, *ref_err_y, *ref_err_z;
    double *ref_lhs_x, *ref_lhs_y, *ref_lhs_z;
    int use_iterative_refinement;
    orc_kkt *kkt_solver;
    int last_refine_steps;
    int backend_solves;
};

static double *dz(size_t n)
{
    void *p = NULL;
    size_t bytes = (n ? n : 1) * sizeof(double);
    if (posix_memalign(&p, 64, (bytes + 63) & ~(size_t)63)) abort();
    memset(p, 0, bytes);
    return (double *)p;
}
static double *dd(const double *s, size_t n) { double *p = dz(n); if (n) memcpy(p, s, n * sizeof(double)); return p; }

/* kkt_system.hpp:430-453 */
static void extract_P_diag(orc_kkt_system *k, const orc_data *d)
{
    if (!d->is_sparse) {
        for (int j = 0; j < d->n; j++) k->P_diag[j] = d->P_utri[j + (size_t)j * d->n];
    } else {
        for (int j = 0; j < d->n; j++)
            for (int q = d->sP_utri.colptr[j]; q < d->sP_utri.colptr[j + 1]; q++)
                if (d->sP_utri.rowind[q] == j) k->P_diag[j] = d->sP_utri.val[q];
    }
}

orc_kkt *orc_multistage_kkt_create(const orc_data *d) __attribute__((weak));
orc_kkt *orc_sparse_kkt_create(const orc_data *d, int mode) __attribute__((weak));

/* kkt_system.hpp:97-132 + :455-497 */
orc_kkt_system *orc_kkt_system_create(const orc_data *d, const orc_settings *s)
{
    orc_kkt_system *k = (orc_kkt_system *)calloc(1, sizeof(*k));
    int n = d->n, p = d->p, m = d->m;
    k->n = n; k->p = p; k->m = m;
    k->P_diag = dz(n);
    k->m_s_l = dz(m); k->m_s_u = dz(m); k->m_s_bl = dz(n); k->m_s_bu = dz(n);
    k->m_z_l_inv = dz(m); k->m_z_u_inv = dz(m); k->m_z_bl_inv = dz(n); k->m_z_bu_inv = dz(n);
    k->m_x_reg = dz(n); k->m_z_reg = dz(m);
    k->rhs_x_bar = dz(n); k->rhs_z_bar = dz(m);
    k->work_x = dz(n); k->work_z = dz(m);
    k->ref_err_x = dz(n); k->ref_err_y = dz(p); k->ref_err_z = dz(m);
    k->ref_lhs_x = dz(n); k->ref_lhs_y = dz(p); k->ref_lhs_z = dz(m);
    extract_P_diag(k, d);
    if (!d->is_sparse) {
        switch (s->kkt_solver) {
        case ORC_DENSE_CHOLESKY: k->kkt_solver = orc_dense_kkt_create(d, 0); break;
        case ORC_DENSE_LDLT_NO_PIVOT: k->kkt_solver = orc_dense_kkt_create(d, 1); break;
        default: fprintf(stderr, "kkt solver not supported\n"); break;
        }
    } else {
        if (!orc_sparse_kkt_create) { fprintf(stderr, "kkt solver not supported\n"); orc_kkt_system_free(k); return NULL; }
        switch (s->kkt_solver) {
        case ORC_SPARSE_LDLT: k->kkt_solver = orc_sparse_kkt_create(d, 0); break;
        case ORC_SPARSE_LDLT_EQ_COND: k->kkt_solver = orc_sparse_kkt_create(d, 1); break;
        case ORC_SPARSE_LDLT_INEQ_COND: k->kkt_solver = orc_sparse_kkt_create(d, 2); break;
        case ORC_SPARSE_LDLT_COND: k->kkt_solver = orc_sparse_kkt_create(d, 3); break;
        case ORC_SPARSE_MULTISTAGE:
            if (orc_multistage_kkt_create) k->kkt_solver = orc_multistage_kkt_create(d);
            else fprintf(stderr, "kkt solver not supported\n");
            break;
        default: fprintf(stderr, "kkt solver not supported\n"); break;
        }
    }
    if (!k->kkt_solver) { orc_kkt_system_free(k); return NULL; }
    return k;
}

/* kkt_system.hpp:70-95 (copy ctor: work/ref buffers are resized, not copied) */
orc_kkt_system *orc_kkt_system_clone(const orc_kkt_system *o)
{
    orc_kkt_system *k = (orc_kkt_system *)calloc(1, sizeof(*k));
    int n = o->n, p = o->p, m = o->m;
    *k = *o;
    k->P_diag = dd(o->P_diag, n);
    k->m_s_l = dd(o->m_s_l, m); k->m_s_u = dd(o->m_s_u, m); k->m_s_bl = dd(o->m_s_bl, n); k->m_s_bu = dd(o->m_s_bu, n);
    k->m_z_l_inv = dd(o->m_z_l_inv, m); k->m_z_u_inv = dd(o->m_z_u_inv, m);
    k->m_z_bl_inv = dd(o->m_z_bl_inv, n); k->m_z_bu_inv = dd(o->m_z_bu_inv, n);
    k->m_x_reg = dd(o->m_x_reg, n); k->m_z_reg = dd(o->m_z_reg, m);
    k->rhs_x_bar = dd(o->rhs_x_bar, n); k->rhs_z_bar = dd(o->rhs_z_bar, m);
    k->work_x = dz(n); k->work_z = dz(m);
    k->ref_err_x = dz(n); k->ref_err_y = dz(p); k->ref_err_z = dz(m);
    k->ref_lhs_x = dz(n); k->ref_lhs_y = dz(p); k->ref_lhs_z = dz(m);
    k->kkt_solver = o->kkt_solver ? o->kkt_solver->clone(o->kkt_solver) : NULL;
    return k;
}

void orc_kkt_system_free(orc_kkt_system *k)
{
    if (!k) return;
    free(k->P_diag);
    free(k->m_s_l); free(k->m_s_u); free(k->m_s_bl); free(k->m_s_bu);
    free(k->m_z_l_inv); free(k->m_z_u_inv); free(k->m_z_bl_inv); free(k->m_z_bu_inv);
    free(k->m_x_reg); free(k->m_z_reg); free(k->rhs_x_bar); free(k->rhs_z_bar);
    free(k->work_x); free(k->work_z);
    free(k->ref_err_x); free(k->ref_err_y); free(k->ref_err_z);
    free(k->ref_lhs_x); free(k->ref_lhs_y); free(k->ref_lhs_z);
    if (k->kkt_solver) k->kkt_solver->destroy(k->kkt_solver);
    free(k);
}

orc_kkt *orc_kkt_system_backend(orc_kkt_system *k) { return k->kkt_solver; }
int orc_kkt_system_last_refine_steps(const orc_kkt_system *k) { return k->last_refine_steps; }
int orc_kkt_system_backend_solves(const orc_kkt_system *k) { return k->backend_solves; }

/* kkt_system.hpp:134-141 */
void orc_kkt_system_update_data(orc_kkt_system *k, const orc_data *d, int options)
{
    if (options & ORC_KKT_UPDATE_P) extract_P_diag(k, d);
    k->kkt_solver->update_data(k->kkt_solver, d, options);
}

static double amax(const double *x, int n)
{
    double r = 0.0;
    for (int i = 0; i < n; i++) { double a = fabs(x[i]); if (a > r || a != a) r = a; }
    return r;
}

/* kkt_system.hpp:143-211 */
int orc_kkt_system_update_scalings_and_factor(orc_kkt_system *k, const orc_data *d, const orc_settings *s,
                                              int iterative_refinement, double rho, double delta,
                                              const orc_vars *vars)
{
    int n = d->n, m = d->m;
    double *m_z_reg_iter_ref = k->work_z;
    k->m_rho = rho;
    k->m_delta = delta;
    memcpy(k->m_s_l, vars->s_l, sizeof(double) * (size_t)m);
    memcpy(k->m_s_u, vars->s_u, sizeof(double) * (size_t)m);
    memcpy(k->m_s_bl, vars->s_bl, sizeof(double) * (size_t)d->n_x_l);
    memcpy(k->m_s_bu, vars->s_bu, sizeof(double) * (size_t)d->n_x_u);
    for (int i = 0; i < m; i++) { k->m_z_l_inv[i] = 1.0 / vars->z_l[i]; k->m_z_u_inv[i] = 1.0 / vars->z_u[i]; }
    for (int i = 0; i < d->n_x_l; i++) k->m_z_bl_inv[i] = 1.0 / vars->z_bl[i];
    for (int i = 0; i < d->n_x_u; i++) k->m_z_bu_inv[i] = 1.0 / vars->z_bu[i];

    for (int i = 0; i < n; i++) k->m_x_reg[i] = rho;
    for (int i = 0; i < d->n_x_l; i++) {
        int idx = d->x_l_idx[i];
        k->m_x_reg[idx] += d->x_b_scaling[idx] * d->x_b_scaling[idx] / (k->m_z_bl_inv[i] * k->m_s_bl[i] + k->m_delta);
    }
    for (int i = 0; i < d->n_x_u; i++) {
        int idx = d->x_u_idx[i];
        k->m_x_reg[idx] += d->x_b_scaling[idx] * d->x_b_scaling[idx] / (k->m_z_bu_inv[i] * k->m_s_bu[i] + k->m_delta);
    }

    for (int i = 0; i < m; i++) k->m_z_reg[i] = 0.0;
    for (int i = 0; i < d->n_h_l; i++) {
        int idx = d->h_l_idx[i];
        k->m_z_reg[idx] += 1.0 / (k->m_z_l_inv[idx] * k->m_s_l[idx] + delta);
    }
    for (int i = 0; i < d->n_h_u; i++) {
        int idx = d->h_u_idx[i];
        k->m_z_reg[idx] += 1.0 / (k->m_z_u_inv[idx] * k->m_s_u[idx] + delta);
    }
    for (int i = 0; i < m; i++) { k->m_z_reg[i] = 1.0 / k->m_z_reg[i]; m_z_reg_iter_ref[i] = k->m_z_reg[i]; }

    double delta_reg = delta;
    if (iterative_refinement) {
        double max_diag = 0.0;
        for (int i = 0; i < n; i++) { double a = fabs(k->P_diag[i] + k->m_x_reg[i]); if (a > max_diag) max_diag = a; }
        double zmax = amax(m_z_reg_iter_ref, m);
        if (zmax > max_diag) max_diag = zmax;
        double reg = s->iterative_refinement_static_regularization_eps
                   + s->iterative_refinement_static_regularization_rel * max_diag;
        delta_reg += reg;
        for (int i = 0; i < n; i++) k->m_x_reg[i] += reg;
        for (int i = 0; i < m; i++) m_z_reg_iter_ref[i] += reg;
    }
    k->use_iterative_refinement = iterative_refinement;
    return k->kkt_solver->update_scalings_and_factor(k->kkt_solver, d, delta_reg, k->m_x_reg, m_z_reg_iter_ref);
}

/* kkt_system.hpp:499-505 */
static double inf_norm3(const double *x, int n, const double *y, int p, const double *z, int m)
{
    double r = amax(x, n);
    double a = amax(y, p); if (a > r || a != a) r = a;
    a = amax(z, m); if (a > r || a != a) r = a;
    return r;
}

/* kkt_system.hpp:507-519 */
static void mul_condensed_kkt(orc_kkt_system *k, const orc_data *d, const double *lhs_x, const double *lhs_y,
                              const double *lhs_z, double *rhs_x, double *rhs_y, double *rhs_z)
{
    orc_kkt *b = k->kkt_solver;
    int n = k->n, p = k->p, m = k->m;
    b->eval_P_x(b, d, 1.0, lhs_x, rhs_x);
    for (int i = 0; i < n; i++) rhs_x[i] += k->m_x_reg[i] * lhs_x[i];
    b->eval_A_xn_and_AT_xt(b, d, 1.0, 1.0, lhs_x, lhs_y, rhs_y, k->work_x);
    for (int i = 0; i < n; i++) rhs_x[i] += k->work_x[i];
    for (int i = 0; i < p; i++) rhs_y[i] -= k->m_delta * lhs_y[i];
    b->eval_G_xn_and_GT_xt(b, d, 1.0, 1.0, lhs_x, lhs_z, rhs_z, k->work_x);
    for (int i = 0; i < n; i++) rhs_x[i] += k->work_x[i];
    for (int i = 0; i < m; i++) rhs_z[i] -= k->m_z_reg[i] * lhs_z[i];
}

/* kkt_system.hpp:522-536 */
static double get_refine_error(orc_kkt_system *k, const orc_data *d, const double *lhs_x, const double *lhs_y,
                               const double *lhs_z, const double *rhs_x, const double *rhs_y, const double *rhs_z,
                               double *err_x, double *err_y, double *err_z)
{
    mul_condensed_kkt(k, d, lhs_x, lhs_y, lhs_z, err_x, err_y, err_z);
    for (int i = 0; i < k->n; i++) err_x[i] = rhs_x[i] - err_x[i];
    for (int i = 0; i < k->p; i++) err_y[i] = rhs_y[i] - err_y[i];
    for (int i = 0; i < k->m; i++) err_z[i] = rhs_z[i] - err_z[i];
    return inf_norm3(err_x, k->n, err_y, k->p, err_z, k->m);
}

static int all_finite(const double *x, int n)
{
    for (int i = 0; i < n; i++) if (!isfinite(x[i])) return 0;
    return 1;
}
#define SWAPP(a, b) do { double *t_ = (a); (a) = (b); (b) = t_; } while (0)

/* kkt_system.hpp:213-369 */
int orc_kkt_system_solve(orc_kkt_system *k, const orc_data *d, const orc_settings *s, const orc_vars *rhs, orc_vars *lhs)
{
    int m = d->m, n = d->n;
    orc_kkt *b = k->kkt_solver;
    double **lhs_z = &k->work_z; /* Vec<T>& lhs_z = work_z (may be swapped with ref_lhs_z) */
    k->last_refine_steps = 0;

    /* :219-234 rhs_z_bar */
    for (int i = 0; i < m; i++) k->rhs_z_bar[i] = 0.0;
    for (int i = 0; i < d->n_h_l; i++) {
        int idx = d->h_l_idx[i];
        k->rhs_z_bar[idx] -= 1.0 / (k->m_z_l_inv[idx] * k->m_s_l[idx] + k->m_delta) * (rhs->z_l[idx] - k->m_z_l_inv[idx] * rhs->s_l[idx]);
    }
    for (int i = 0; i < d->n_h_u; i++) {
        int idx = d->h_u_idx[i];
        k->rhs_z_bar[idx] += 1.0 / (k->m_z_u_inv[idx] * k->m_s_u[idx] + k->m_delta) * (rhs->z_u[idx] - k->m_z_u_inv[idx] * rhs->s_u[idx]);
    }
    for (int i = 0; i < m; i++) k->rhs_z_bar[i] *= k->m_z_reg[i];

    /* :236-252 rhs_x_bar */
    memcpy(k->rhs_x_bar, rhs->x, sizeof(double) * (size_t)n);
    for (int i = 0; i < d->n_x_l; i++) {
        int idx = d->x_l_idx[i];
        k->rhs_x_bar[idx] -= d->x_b_scaling[idx] * (rhs->z_bl[i] - k->m_z_bl_inv[i] * rhs->s_bl[i])
                             / (k->m_s_bl[i] * k->m_z_bl_inv[i] + k->m_delta);
    }
    for (int i = 0; i < d->n_x_u; i++) {
        int idx = d->x_u_idx[i];
        k->rhs_x_bar[idx] += d->x_b_scaling[idx] * (rhs->z_bu[i] - k->m_z_bu_inv[i] * rhs->s_bu[i])
                             / (k->m_s_bu[i] * k->m_z_bu_inv[i] + k->m_delta);
    }

    b->solve(b, d, k->rhs_x_bar, rhs->y, k->rhs_z_bar, lhs->x, lhs->y, *lhs_z);
    k->backend_solves++;

    if (k->use_iterative_refinement) {
        double rhs_norm = inf_norm3(k->rhs_x_bar, n, rhs->y, d->p, k->rhs_z_bar, m);
        double refine_error = get_refine_error(k, d, lhs->x, lhs->y, *lhs_z, k->rhs_x_bar, rhs->y, k->rhs_z_bar,
                                               k->ref_err_x, k->ref_err_y, k->ref_err_z);
        if (!isfinite(refine_error)) return 0;
        for (int it = 0; it < s->iterative_refinement_max_iter; it++) {
            if (refine_error <= s->iterative_refinement_eps_abs + s->iterative_refinement_eps_rel * rhs_norm) break;
            double prev_refine_error = refine_error;
            b->solve(b, d, k->ref_err_x, k->ref_err_y, k->ref_err_z, k->ref_lhs_x, k->ref_lhs_y, k->ref_lhs_z);
            k->backend_solves++;
            k->last_refine_steps++;
            for (int i = 0; i < n; i++) k->ref_lhs_x[i] += lhs->x[i];
            for (int i = 0; i < d->p; i++) k->ref_lhs_y[i] += lhs->y[i];
            for (int i = 0; i < m; i++) k->ref_lhs_z[i] += (*lhs_z)[i];
            refine_error = get_refine_error(k, d, k->ref_lhs_x, k->ref_lhs_y, k->ref_lhs_z, k->rhs_x_bar, rhs->y,
                                            k->rhs_z_bar, k->ref_err_x, k->ref_err_y, k->ref_err_z);
            if (!isfinite(refine_error)) return 0;
            double improvement_rate = prev_refine_error / refine_error;
            if (improvement_rate < s->iterative_refinement_min_improvement_rate) {
                if (improvement_rate > 1.0) {
                    SWAPP(lhs->x, k->ref_lhs_x); SWAPP(lhs->y, k->ref_lhs_y); SWAPP(*lhs_z, k->ref_lhs_z);
                }
                break;
            }
            SWAPP(lhs->x, k->ref_lhs_x); SWAPP(lhs->y, k->ref_lhs_y); SWAPP(*lhs_z, k->ref_lhs_z);
        }
    } else {
        if (!all_finite(lhs->x, n) || !all_finite(lhs->y, d->p) || !all_finite(*lhs_z, m)) return 0;
    }

    /* :310-345 dual recovery */
    {
        const double *lz = *lhs_z;
        int i_l = 0, i_u = 0;
        for (int i = 0; i < m; i++) {
            /* :316-319 -- advance both cursors to the first finite-bound index >= i (the reference's
             * post-increment form can read one slot past the list; the intended value there is "none") */
            while (i_l < d->n_h_l && d->h_l_idx[i_l] < i) i_l++;
            while (i_u < d->n_h_u && d->h_u_idx[i_u] < i) i_u++;
            int idx_l = i_l < d->n_h_l ? d->h_l_idx[i_l] : -1;
            int idx_u = i_u < d->n_h_u ? d->h_u_idx[i_u] : -1;
            if (idx_l == i && idx_u == i) {
                double rz_l_bar = rhs->z_l[i] - k->m_z_l_inv[i] * rhs->s_l[i];
                double W_l_inv = 1.0 / (k->m_z_l_inv[i] * k->m_s_l[i] + k->m_delta);
                double rz_u_bar = rhs->z_u[i] - k->m_z_u_inv[i] * rhs->s_u[i];
                double W_u_inv = 1.0 / (k->m_z_u_inv[i] * k->m_s_u[i] + k->m_delta);
                double r_sum = W_l_inv * W_u_inv * (rz_l_bar + rz_u_bar);
                lhs->z_l[i] = -k->m_z_reg[i] * (r_sum + W_l_inv * lz[i]);
                lhs->z_u[i] = -k->m_z_reg[i] * (r_sum - W_u_inv * lz[i]);
                lhs->s_l[i] = k->m_z_l_inv[i] * (rhs->s_l[i] - k->m_s_l[i] * lhs->z_l[i]);
                lhs->s_u[i] = k->m_z_u_inv[i] * (rhs->s_u[i] - k->m_s_u[i] * lhs->z_u[i]);
            } else if (idx_l == i) {
                lhs->z_l[i] = -lz[i];
                lhs->z_u[i] = 0.0;
                lhs->s_l[i] = k->m_z_l_inv[i] * (rhs->s_l[i] - k->m_s_l[i] * lhs->z_l[i]);
                lhs->s_u[i] = 0.0;
            } else if (idx_u == i) {
                lhs->z_l[i] = 0.0;
                lhs->z_u[i] = lz[i];
                lhs->s_l[i] = 0.0;
                lhs->s_u[i] = k->m_z_u_inv[i] * (rhs->s_u[i] - k->m_s_u[i] * lhs->z_u[i]);
            } else {
                /* unreachable in the reference (assert): a row with no finite bound is disabled in data */
                lhs->z_l[i] = lhs->z_u[i] = lhs->s_l[i] = lhs->s_u[i] = 0.0;
            }
        }
    }
    /* :347-366 box dual recovery */
    for (int i = 0; i < d->n_x_l; i++) {
        int idx = d->x_l_idx[i];
        lhs->z_bl[i] = (-d->x_b_scaling[idx] * lhs->x[idx] - rhs->z_bl[i] + k->m_z_bl_inv[i] * rhs->s_bl[i])
                       / (k->m_s_bl[i] * k->m_z_bl_inv[i] + k->m_delta);
    }
    for (int i = 0; i < d->n_x_u; i++) {
        int idx = d->x_u_idx[i];
        lhs->z_bu[i] = (d->x_b_scaling[idx] * lhs->x[idx] - rhs->z_bu[i] + k->m_z_bu_inv[i] * rhs->s_bu[i])
                       / (k->m_s_bu[i] * k->m_z_bu_inv[i] + k->m_delta);
    }
    for (int i = 0; i < d->n_x_l; i++) lhs->s_bl[i] = k->m_z_bl_inv[i] * (rhs->s_bl[i] - k->m_s_bl[i] * lhs->z_bl[i]);
    for (int i = 0; i < d->n_x_u; i++) lhs->s_bu[i] = k->m_z_bu_inv[i] * (rhs->s_bu[i] - k->m_s_bu[i] * lhs->z_bu[i]);
    return 1;
}

/* kkt_system.hpp:392-425 */
void orc_kkt_system_mul(orc_kkt_system *k, const orc_data *d, const orc_vars *lhs, orc_vars *rhs)
{
    orc_kkt *b = k->kkt_solver;
    int n = d->n, p = d->p, m = d->m;
    b->eval_P_x(b, d, 1.0, lhs->x, rhs->x);
    for (int i = 0; i < n; i++) rhs->x[i] += k->m_rho * lhs->x[i];
    b->eval_A_xn_and_AT_xt(b, d, 1.0, 1.0, lhs->x, lhs->y, rhs->y, k->work_x);
    for (int i = 0; i < n; i++) rhs->x[i] += k->work_x[i];
    for (int i = 0; i < p; i++) rhs->y[i] -= k->m_delta * lhs->y[i];
    for (int i = 0; i < m; i++) rhs->s_l[i] = lhs->z_u[i] - lhs->z_l[i];
    b->eval_G_xn_and_GT_xt(b, d, 1.0, 1.0, lhs->x, rhs->s_l, rhs->z_u, k->work_x);
    for (int i = 0; i < m; i++) rhs->z_l[i] = -rhs->z_u[i];
    for (int i = 0; i < n; i++) rhs->x[i] += k->work_x[i];
    for (int i = 0; i < m; i++) rhs->z_l[i] += lhs->s_l[i] - k->m_delta * lhs->z_l[i];
    for (int i = 0; i < m; i++) rhs->z_u[i] += lhs->s_u[i] - k->m_delta * lhs->z_u[i];
    for (int i = 0; i < m; i++) rhs->s_l[i] = k->m_s_l[i] * lhs->z_l[i] + lhs->s_l[i] / k->m_z_l_inv[i];
    for (int i = 0; i < m; i++) rhs->s_u[i] = k->m_s_u[i] * lhs->z_u[i] + lhs->s_u[i] / k->m_z_u_inv[i];
    for (int i = 0; i < d->n_x_l; i++) {
        int idx = d->x_l_idx[i];
        rhs->x[idx] -= d->x_b_scaling[idx] * lhs->z_bl[i];
        rhs->z_bl[i] = -d->x_b_scaling[idx] * lhs->x[idx] - k->m_delta * lhs->z_bl[i] + lhs->s_bl[i];
    }
    for (int i = 0; i < d->n_x_l; i++) rhs->s_bl[i] = k->m_s_bl[i] * lhs->z_bl[i] + lhs->s_bl[i] / k->m_z_bl_inv[i];
    for (int i = 0; i < d->n_x_u; i++) {
        int idx = d->x_u_idx[i];
        rhs->x[idx] += d->x_b_scaling[idx] * lhs->z_bu[i];
        rhs->z_bu[i] = d->x_b_scaling[idx] * lhs->x[idx] - k->m_delta * lhs->z_bu[i] + lhs->s_bu[i];
    }
    for (int i = 0; i < d->n_x_u; i++) rhs->s_bu[i] = k->m_s_bu[i] * lhs->z_bu[i] + lhs->s_bu[i] / k->m_z_bu_inv[i];
}

/* ---- flat dispatch helpers for FFI callers (ctypes cannot call through the vtable conveniently) ---- */
orc_kkt *orc_kkt_clone(const orc_kkt *k) { return k->clone(k); }
void orc_kkt_destroy(orc_kkt *k) { if (k) k->destroy(k); }
void orc_kkt_update_data(orc_kkt *k, const orc_data *d, int options) { k->update_data(k, d, options); }
int orc_kkt_update_scalings_and_factor(orc_kkt *k, const orc_data *d, double delta, const double *x_reg, const double *z_reg)
{ return k->update_scalings_and_factor(k, d, delta, x_reg, z_reg); }
void orc_kkt_solve(orc_kkt *k, const orc_data *d, const double *rx, const double *ry, const double *rz, double *lx, double *ly, double *lz)
{ k->solve(k, d, rx, ry, rz, lx, ly, lz); }
void orc_kkt_eval_P_x(orc_kkt *k, const orc_data *d, double alpha, const double *x, double *z) { k->eval_P_x(k, d, alpha, x, z); }
void orc_kkt_eval_A_xn_and_AT_xt(orc_kkt *k, const orc_data *d, double an, double at, const double *xn, const double *xt, double *zn, double *zt)
{ k->eval_A_xn_and_AT_xt(k, d, an, at, xn, xt, zn, zt); }
void orc_kkt_eval_G_xn_and_GT_xt(orc_kkt *k, const orc_data *d, double an, double at, const double *xn, const double *xt, double *zn, double *zt)
{ k->eval_G_xn_and_GT_xt(k, d, an, at, xn, xt, zn, zt); }

/* KKTSystem::solve for callers that own fixed output buffers: solves into internal vectors (whose
 * pointers the refinement loop may swap, kkt_system.hpp:292-300) and copies the result out. */
int orc_kkt_system_solve_copy(orc_kkt_system *k, const orc_data *d, const orc_settings *s, const orc_vars *rhs, orc_vars *out)
{
    int n = d->n, p = d->p, m = d->m;
    orc_vars tmp;
    tmp.x = dz(n); tmp.y = dz(p); tmp.z_l = dz(m); tmp.z_u = dz(m); tmp.z_bl = dz(n); tmp.z_bu = dz(n);
    tmp.s_l = dz(m); tmp.s_u = dz(m); tmp.s_bl = dz(n); tmp.s_bu = dz(n);
    int ok = orc_kkt_system_solve(k, d, s, rhs, &tmp);
    memcpy(out->x, tmp.x, sizeof(double) * (size_t)n); memcpy(out->y, tmp.y, sizeof(double) * (size_t)p);
    memcpy(out->z_l, tmp.z_l, sizeof(double) * (size_t)m); memcpy(out->z_u, tmp.z_u, sizeof(double) * (size_t)m);
    memcpy(out->z_bl, tmp.z_bl, sizeof(double) * (size_t)n); memcpy(out->z_bu, tmp.z_bu, sizeof(double) * (size_t)n);
    memcpy(out->s_l, tmp.s_l, sizeof(double) * (size_t)m); memcpy(out->s_u, tmp.s_u, sizeof(double) * (size_t)m);
    memcpy(out->s_bl, tmp.s_bl, sizeof(double) * (size_t)n); memcpy(out->s_bu, tmp.s_bu, sizeof(double) * (size_t)n);
    free(tmp.x); free(tmp.y); free(tmp.z_l); free(tmp.z_u); free(tmp.z_bl); free(tmp.z_bu);
    free(tmp.s_l); free(tmp.s_u); free(tmp.s_bl); free(tmp.s_bu);
    return ok;
}
/* condensed-system accessors for tests (x_reg / z_reg actually handed to the backend, rhs_bar) */
const double *orc_kkt_system_x_reg(const orc_kkt_system *k) { return k->m_x_reg; }
const double *orc_kkt_system_z_reg(const orc_kkt_system *k) { return k->m_z_reg; }
const double *orc_kkt_system_rhs_x_bar(const orc_kkt_system *k) { return k->rhs_x_bar; }
const double *orc_kkt_system_rhs_z_bar(const orc_kkt_system *k) { return k->rhs_z_bar; }
