/*
 * oracle/orc_sparse_cond.c -- CPU restatement of the reference's CONDENSED sparse KKT modes
 * (KKTSolver::sparse_ldlt_eq_cond / sparse_ldlt_ineq_cond / sparse_ldlt_cond):
 *   sparse/kkt.hpp:51-176 (ctor, update_scalings_and_factor, solve branches)
 *   sparse/kkt_eq_eliminated.hpp    K = [P + x_reg + d^-1 A'A, G'; G, -z_reg]           (n + m)
 *   sparse/kkt_ineq_eliminated.hpp  K = [P + x_reg + G' Z^-1 G, A'; A, -d I]            (n + p)
 *   sparse/kkt_all_eliminated.hpp   K =  P + x_reg + d^-1 A'A + G' Z^-1 G               (n)
 * with the same ordering (AMD), permutation and up-looking LDLt as the KKT_FULL mode (orc_sparse.c).
 *
 * TEST INFRASTRUCTURE ONLY (see orc.h).  Pinned by the reference's sparse/kkt_test.cpp properties for all four
 * modes (K lhs ~ rhs at 1e-8, update == fresh) and by agreement with the KKT_FULL mode -> tests/test_oracle_sparse.py.
 */
#include "orc.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    orc_kkt base;
    int n, p, m, N, mode, eq, ineq; /* eq / ineq: that block is eliminated */
    double m_delta;
    double *m_z_reg_inv, *work_z;
    orc_csc A, G;          /* p x n and m x n (transposes of data.AT / data.GT) */
    orc_csc AT_A, GT_G;    /* upper triangles, n x n (GT_G = G' (W+delta)^-1 G values) */
    double *tmp_scatter;
    int *P, *P_inv;
    int *PKPt_p, *PKPt_i; double *PKPt_x; int nnzK;
    int *PKi;
    int *P_utri_to_Ki, *AT_A_to_Ki, *GT_G_to_Ki, *AT_to_Ki, *GT_to_Ki;
    int zP, zAA, zGG, zA, zG; /* lengths of the five maps */
    orc_sparse_ldlt *ldlt;
    double *rhs, *rhs_perm;
} cond_kkt;

static void *xc(size_t n, size_t s) { void *q = calloc(n ? n : 1, s); if (!q) { fprintf(stderr, "oom\n"); abort(); } return q; }
static int *idup(const int *s, int n) { int *q = (int *)xc((size_t)n, sizeof(int)); if (n) memcpy(q, s, sizeof(int) * (size_t)n); return q; }
static double *ddup(const double *s, int n) { double *q = (double *)xc((size_t)n, sizeof(double)); if (n) memcpy(q, s, sizeof(double) * (size_t)n); return q; }

static void csc_free(orc_csc *M) { free(M->colptr); free(M->rowind); free(M->val); M->colptr = M->rowind = NULL; M->val = NULL; }
static orc_csc csc_clone(const orc_csc *M)
{
    orc_csc R = *M;
    if (!M->colptr) return R;
    int nnz = M->colptr[M->cols];
    R.colptr = idup(M->colptr, M->cols + 1); R.rowind = idup(M->rowind, nnz); R.val = ddup(M->val, nnz);
    return R;
}
/* T = M^T with sorted columns */
static orc_csc csc_transpose(const orc_csc *M)
{
    orc_csc T;
    int nnz = M->colptr[M->cols];
    T.rows = M->cols; T.cols = M->rows;
    T.colptr = (int *)xc((size_t)T.cols + 1, sizeof(int)); T.rowind = (int *)xc((size_t)nnz, sizeof(int)); T.val = (double *)xc((size_t)nnz, sizeof(double));
    for (int q = 0; q < nnz; q++) T.colptr[M->rowind[q] + 1]++;
    for (int j = 0; j < T.cols; j++) T.colptr[j + 1] += T.colptr[j];
    int *nx = idup(T.colptr, T.cols);
    for (int j = 0; j < M->cols; j++)
        for (int q = M->colptr[j]; q < M->colptr[j + 1]; q++) { int t = nx[M->rowind[q]]++; T.rowind[t] = j; T.val[t] = M->val[q]; }
    free(nx);
    return T;
}
/* transpose_no_allocation (sparse/utils.hpp): refresh the values of T = M^T, same pattern */
static void csc_transpose_values(const orc_csc *M, orc_csc *T)
{
    int *nx = idup(T->colptr, T->cols);
    for (int j = 0; j < M->cols; j++)
        for (int q = M->colptr[j]; q < M->colptr[j + 1]; q++) T->val[nx[M->rowind[q]]++] = M->val[q];
    free(nx);
}
/* pattern of upper(MT * M) for MT n x k (structural product, sorted columns), values 0 */
static orc_csc gram_upper_pattern(const orc_csc *MT, const orc_csc *M /* = MT^T */, int n)
{
    orc_csc R;
    R.rows = R.cols = n;
    R.colptr = (int *)xc((size_t)n + 1, sizeof(int));
    int *mark = (int *)xc((size_t)n, sizeof(int));
    for (int i = 0; i < n; i++) mark[i] = -1;
    int cap = 16, nz = 0;
    int *ri = (int *)xc((size_t)cap, sizeof(int));
    for (int j = 0; j < n; j++) {
        int start = nz;
        for (int q = M->colptr[j]; q < M->colptr[j + 1]; q++) {
            int k = M->rowind[q];
            for (int t = MT->colptr[k]; t < MT->colptr[k + 1]; t++) {
                int i = MT->rowind[t];
                if (i > j || mark[i] == j) continue;
                mark[i] = j;
                if (nz == cap) { cap *= 2; ri = (int *)realloc(ri, sizeof(int) * (size_t)cap); }
                ri[nz++] = i;
            }
        }
        for (int a = start + 1; a < nz; a++) { int v = ri[a], b = a - 1; while (b >= start && ri[b] > v) { ri[b + 1] = ri[b]; b--; } ri[b + 1] = v; }
        R.colptr[j + 1] = nz;
    }
    free(mark);
    R.rowind = ri;
    R.val = (double *)xc((size_t)nz, sizeof(double));
    return R;
}

/* update_AT_A / update_GT_W_delta_inv_G (kkt_all_eliminated.hpp:184-223): w == NULL -> unit weights */
static void gram_upper_values(cond_kkt *k, const orc_csc *MT, const orc_csc *M, const double *w, orc_csc *R)
{
    int n = k->n;
    double *tmp = k->tmp_scatter;
    for (int j = 0; j < n; j++) {
        for (int q = M->colptr[j]; q < M->colptr[j + 1]; q++) {
            int kk = M->rowind[q];
            double v = M->val[q];
            for (int t = MT->colptr[kk]; t < MT->colptr[kk + 1]; t++) {
                int i = MT->rowind[t];
                if (i > j) continue;
                tmp[i] += w ? v * MT->val[t] / w[kk] : v * MT->val[t];
            }
        }
        for (int q = R->colptr[j]; q < R->colptr[j + 1]; q++) { R->val[q] = tmp[R->rowind[q]]; tmp[R->rowind[q]] = 0.0; }
    }
}

static void cond_destroy(orc_kkt *self)
{
    cond_kkt *k = (cond_kkt *)self;
    free(k->m_z_reg_inv); free(k->work_z); free(k->tmp_scatter);
    csc_free(&k->A); csc_free(&k->G); csc_free(&k->AT_A); csc_free(&k->GT_G);
    free(k->P); free(k->P_inv); free(k->PKPt_p); free(k->PKPt_i); free(k->PKPt_x); free(k->PKi);
    free(k->P_utri_to_Ki); free(k->AT_A_to_Ki); free(k->GT_G_to_Ki); free(k->AT_to_Ki); free(k->GT_to_Ki);
    orc_sparse_ldlt_free(k->ldlt);
    free(k->rhs); free(k->rhs_perm);
    free(k);
}

/* update_data_impl of the three headers */
static void cond_update_data(orc_kkt *self, const orc_data *d, int options)
{
    cond_kkt *k = (cond_kkt *)self;
    if ((options & ORC_KKT_UPDATE_A) && k->eq) { csc_transpose_values(&d->sAT, &k->A); gram_upper_values(k, &d->sAT, &k->A, NULL, &k->AT_A); }
    if ((options & ORC_KKT_UPDATE_G) && k->ineq) csc_transpose_values(&d->sGT, &k->G);
}

/* sparse/kkt.hpp:83-105 with update_kkt_{cost_scalings,equality_scalings,inequality_scaling} of the mode */
static int cond_factor(orc_kkt *self, const orc_data *d, double delta, const double *x_reg, const double *z_reg)
{
    cond_kkt *k = (cond_kkt *)self;
    int n = k->n, p = k->p, m = k->m;
    double *X = k->PKPt_x;
    k->m_delta = delta;
    for (int i = 0; i < m; i++) k->m_z_reg_inv[i] = 1.0 / z_reg[i];
    /* cost scalings */
    memset(X, 0, sizeof(double) * (size_t)k->nnzK);
    for (int q = 0; q < d->sP_utri.colptr[n]; q++) X[k->PKi[k->P_utri_to_Ki[q]]] += d->sP_utri.val[q];
    for (int col = 0; col < n; col++) X[k->PKPt_p[k->P_inv[col] + 1] - 1] += x_reg[col];
    /* equality scalings */
    if (k->eq) {
        double delta_inv = 1.0 / delta;
        for (int q = 0; q < k->AT_A.colptr[n]; q++) X[k->PKi[k->AT_A_to_Ki[q]]] += delta_inv * k->AT_A.val[q];
    } else {
        for (int q = 0; q < d->sAT.colptr[p]; q++) X[k->PKi[k->AT_to_Ki[q]]] = d->sAT.val[q];
        for (int col = n; col < n + p; col++) X[k->PKPt_p[k->P_inv[col] + 1] - 1] = -delta;
    }
    /* inequality scaling */
    if (k->ineq) {
        gram_upper_values(k, &d->sGT, &k->G, z_reg, &k->GT_G);
        for (int q = 0; q < k->GT_G.colptr[n]; q++) X[k->PKi[k->GT_G_to_Ki[q]]] += k->GT_G.val[q];
    } else {
        int base = n + (k->eq ? 0 : p);
        for (int q = 0; q < d->sGT.colptr[m]; q++) X[k->PKi[k->GT_to_Ki[q]]] = d->sGT.val[q];
        for (int col = base, i = 0; col < base + m; col++, i++) X[k->PKPt_p[k->P_inv[col] + 1] - 1] = -z_reg[i];
    }
    return orc_sparse_ldlt_numeric(k->ldlt, k->N, k->PKPt_p, k->PKPt_i, X) == k->N;
}

static void add_MT_x(const orc_csc *MT, double alpha, const double *x, double *y) /* y += alpha * MT * x */
{
    for (int j = 0; j < MT->cols; j++) {
        double xj = alpha * x[j];
        for (int q = MT->colptr[j]; q < MT->colptr[j + 1]; q++) y[MT->rowind[q]] += MT->val[q] * xj;
    }
}
static void M_x(const orc_csc *MT, double alpha, const double *x, double *y) /* y = alpha * MT^T * x */
{
    for (int j = 0; j < MT->cols; j++) {
        double s = 0.0;
        for (int q = MT->colptr[j]; q < MT->colptr[j + 1]; q++) s += MT->val[q] * x[MT->rowind[q]];
        y[j] = alpha * s;
    }
}

/* sparse/kkt.hpp:107-176 */
static void cond_solve(orc_kkt *self, const orc_data *d, const double *rhs_x, const double *rhs_y, const double *rhs_z,
                       double *lhs_x, double *lhs_y, double *lhs_z)
{
    cond_kkt *k = (cond_kkt *)self;
    int n = k->n, p = k->p, m = k->m, N = k->N;
    double delta_inv = 1.0 / k->m_delta;
    double *rhs = k->rhs;
    memcpy(rhs, rhs_x, sizeof(double) * (size_t)n);
    if (k->ineq) {
        for (int i = 0; i < m; i++) k->work_z[i] = k->m_z_reg_inv[i] * rhs_z[i];
        add_MT_x(&d->sGT, 1.0, k->work_z, rhs);
    }
    if (k->eq) add_MT_x(&d->sAT, delta_inv, rhs_y, rhs);
    if (k->mode == 1) memcpy(rhs + n, rhs_z, sizeof(double) * (size_t)m);      /* EQ eliminated: tail(m) = rhs_z */
    else if (k->mode == 2) memcpy(rhs + n, rhs_y, sizeof(double) * (size_t)p); /* INEQ eliminated: tail(p) = rhs_y */
    for (int j = 0; j < N; j++) k->rhs_perm[j] = rhs[k->P[j]];
    orc_sparse_ldlt_solve_inplace(k->ldlt, k->rhs_perm);
    for (int j = 0; j < N; j++) rhs[k->P[j]] = k->rhs_perm[j];
    memcpy(lhs_x, rhs, sizeof(double) * (size_t)n);
    if (k->eq) {
        M_x(&d->sAT, delta_inv, lhs_x, lhs_y);
        for (int i = 0; i < p; i++) lhs_y[i] -= delta_inv * rhs_y[i];
    } else {
        memcpy(lhs_y, rhs + n, sizeof(double) * (size_t)p);
    }
    if (k->ineq) {
        M_x(&d->sGT, 1.0, lhs_x, lhs_z);
        for (int i = 0; i < m; i++) lhs_z[i] = (lhs_z[i] - rhs_z[i]) * k->m_z_reg_inv[i];
    } else {
        memcpy(lhs_z, rhs + n, sizeof(double) * (size_t)m);
    }
}

/* eval_* are mode independent (sparse/kkt.hpp:179-203) */
static void cond_eval_P_x(orc_kkt *self, const orc_data *d, double alpha, const double *x, double *z)
{
    int n = d->n;
    const orc_csc *U = &d->sP_utri;
    (void)self;
    memset(z, 0, sizeof(double) * (size_t)n);
    for (int j = 0; j < n; j++) {
        double xj = alpha * x[j];
        for (int q = U->colptr[j]; q < U->colptr[j + 1]; q++) z[U->rowind[q]] += U->val[q] * xj;
    }
    for (int j = 0; j < n; j++) {
        double s = 0.0;
        for (int q = U->colptr[j]; q < U->colptr[j + 1]; q++) if (U->rowind[q] < j) s += U->val[q] * x[U->rowind[q]];
        z[j] += alpha * s;
    }
}
static void cond_eval_A(orc_kkt *self, const orc_data *d, double an, double at, const double *xn, const double *xt, double *zn, double *zt)
{
    (void)self;
    M_x(&d->sAT, an, xn, zn);
    memset(zt, 0, sizeof(double) * (size_t)d->n);
    add_MT_x(&d->sAT, at, xt, zt);
}
static void cond_eval_G(orc_kkt *self, const orc_data *d, double an, double at, const double *xn, const double *xt, double *zn, double *zt)
{
    (void)self;
    M_x(&d->sGT, an, xn, zn);
    memset(zt, 0, sizeof(double) * (size_t)d->n);
    add_MT_x(&d->sGT, at, xt, zt);
}
static void cond_print_info(orc_kkt *self) { (void)self; }
static orc_kkt *cond_clone(const orc_kkt *self);

static void cond_vtable(cond_kkt *k)
{
    k->base.clone = cond_clone;
    k->base.update_data = cond_update_data;
    k->base.update_scalings_and_factor = cond_factor;
    k->base.solve = cond_solve;
    k->base.eval_P_x = cond_eval_P_x;
    k->base.eval_A_xn_and_AT_xt = cond_eval_A;
    k->base.eval_G_xn_and_GT_xt = cond_eval_G;
    k->base.print_info = cond_print_info;
    k->base.destroy = cond_destroy;
}

/* ctor: init_workspace + create_kkt_matrix of the mode, ordering, permutation, symbolic (sparse/kkt.hpp:51-70) */
orc_kkt *orc_sparse_cond_kkt_create(const orc_data *d, int mode)
{
    if (mode < 1 || mode > 3) return NULL;
    cond_kkt *k = (cond_kkt *)xc(1, sizeof(cond_kkt));
    cond_vtable(k);
    int n = d->n, p = d->p, m = d->m;
    k->n = n; k->p = p; k->m = m; k->mode = mode; k->eq = mode & 1; k->ineq = (mode & 2) != 0;
    int N = n + (k->eq ? 0 : p) + (k->ineq ? 0 : m);
    k->N = N;
    k->m_delta = 0.0; /* uninitialised in the reference ctor; every value it touches is rewritten before the first factorisation */
    k->m_z_reg_inv = (double *)xc((size_t)m, sizeof(double)); k->work_z = (double *)xc((size_t)m, sizeof(double));
    k->tmp_scatter = (double *)xc((size_t)n, sizeof(double));
    const orc_csc *U = &d->sP_utri, *AT = &d->sAT, *GT = &d->sGT;
    if (k->eq) { k->A = csc_transpose(AT); k->AT_A = gram_upper_pattern(AT, &k->A, n); gram_upper_values(k, AT, &k->A, NULL, &k->AT_A); }
    if (k->ineq) { k->G = csc_transpose(GT); k->GT_G = gram_upper_pattern(GT, &k->G, n); }
    int nzP = U->colptr[n], nzAA = k->eq ? k->AT_A.colptr[n] : 0, nzGG = k->ineq ? k->GT_G.colptr[n] : 0;
    k->zP = nzP; k->zAA = nzAA; k->zGG = nzGG; k->zA = k->eq ? 0 : AT->colptr[p]; k->zG = k->ineq ? 0 : GT->colptr[m];
    k->P_utri_to_Ki = (int *)xc((size_t)nzP, sizeof(int));
    k->AT_A_to_Ki = (int *)xc((size_t)nzAA, sizeof(int)); k->GT_G_to_Ki = (int *)xc((size_t)nzGG, sizeof(int));
    k->AT_to_Ki = (int *)xc((size_t)(k->eq ? 0 : AT->colptr[p]), sizeof(int)); k->GT_to_Ki = (int *)xc((size_t)(k->ineq ? 0 : GT->colptr[m]), sizeof(int));
    /* top-left block: union of P_utri, I and the eliminated Gram matrices, column by column (sorted merge) */
    int cap = nzP + n + nzAA + nzGG + 16, nz = 0;
    int *Kp = (int *)xc((size_t)N + 1, sizeof(int));
    int *Ki = (int *)xc((size_t)cap + (size_t)(k->eq ? 0 : AT->colptr[p] + p) + (size_t)(k->ineq ? 0 : GT->colptr[m] + m), sizeof(int));
    for (int j = 0; j < n; j++) {
        int a = U->colptr[j], ae = U->colptr[j + 1];
        int b = k->eq ? k->AT_A.colptr[j] : 0, be = k->eq ? k->AT_A.colptr[j + 1] : 0;
        int c = k->ineq ? k->GT_G.colptr[j] : 0, ce = k->ineq ? k->GT_G.colptr[j + 1] : 0;
        int diag_done = 0;
        for (;;) {
            int r = n + 1;
            if (a < ae && U->rowind[a] < r) r = U->rowind[a];
            if (b < be && k->AT_A.rowind[b] < r) r = k->AT_A.rowind[b];
            if (c < ce && k->GT_G.rowind[c] < r) r = k->GT_G.rowind[c];
            if (!diag_done && j < r) r = j;
            if (r > n) break;
            if (a < ae && U->rowind[a] == r) k->P_utri_to_Ki[a++] = nz;
            if (b < be && k->AT_A.rowind[b] == r) k->AT_A_to_Ki[b++] = nz;
            if (c < ce && k->GT_G.rowind[c] == r) k->GT_G_to_Ki[c++] = nz;
            if (r == j) diag_done = 1;
            Ki[nz++] = r;
        }
        Kp[j + 1] = nz;
    }
    int jk = n;
    if (!k->eq)
        for (int j = 0; j < p; j++, jk++) {
            for (int q = AT->colptr[j]; q < AT->colptr[j + 1]; q++) { k->AT_to_Ki[q] = nz; Ki[nz++] = AT->rowind[q]; }
            Ki[nz++] = jk;
            Kp[jk + 1] = nz;
        }
    if (!k->ineq)
        for (int j = 0; j < m; j++, jk++) {
            for (int q = GT->colptr[j]; q < GT->colptr[j + 1]; q++) { k->GT_to_Ki[q] = nz; Ki[nz++] = GT->rowind[q]; }
            Ki[nz++] = jk;
            Kp[jk + 1] = nz;
        }
    k->nnzK = nz;
    k->P = (int *)xc((size_t)N, sizeof(int)); k->P_inv = (int *)xc((size_t)N, sizeof(int));
    orc_amd_order(N, Kp, Ki, k->P);
    for (int i = 0; i < N; i++) k->P_inv[k->P[i]] = i;
    k->PKPt_p = (int *)xc((size_t)N + 1, sizeof(int)); k->PKPt_i = (int *)xc((size_t)nz, sizeof(int)); k->PKPt_x = (double *)xc((size_t)nz, sizeof(double));
    k->PKi = (int *)xc((size_t)nz, sizeof(int));
    orc_permute_sym_upper(N, Kp, Ki, NULL, k->P_inv, k->PKPt_p, k->PKPt_i, k->PKPt_x, k->PKi);
    k->ldlt = orc_sparse_ldlt_create();
    orc_sparse_ldlt_symbolic(k->ldlt, N, k->PKPt_p, k->PKPt_i);
    k->rhs = (double *)xc((size_t)N, sizeof(double)); k->rhs_perm = (double *)xc((size_t)N, sizeof(double));
    free(Kp); free(Ki);
    return &k->base;
}

static orc_kkt *cond_clone(const orc_kkt *self)
{
    const cond_kkt *s = (const cond_kkt *)self;
    cond_kkt *k = (cond_kkt *)xc(1, sizeof(cond_kkt));
    *k = *s;
    int n = s->n, p = s->p, m = s->m, N = s->N, nz = s->nnzK;
    k->m_z_reg_inv = ddup(s->m_z_reg_inv, m); k->work_z = ddup(s->work_z, m); k->tmp_scatter = (double *)xc((size_t)n, sizeof(double));
    k->A = csc_clone(&s->A); k->G = csc_clone(&s->G); k->AT_A = csc_clone(&s->AT_A); k->GT_G = csc_clone(&s->GT_G);
    k->P = idup(s->P, N); k->P_inv = idup(s->P_inv, N);
    k->PKPt_p = idup(s->PKPt_p, N + 1); k->PKPt_i = idup(s->PKPt_i, nz); k->PKPt_x = ddup(s->PKPt_x, nz); k->PKi = idup(s->PKi, nz);
    k->P_utri_to_Ki = idup(s->P_utri_to_Ki, s->zP); k->AT_A_to_Ki = idup(s->AT_A_to_Ki, s->zAA); k->GT_G_to_Ki = idup(s->GT_G_to_Ki, s->zGG);
    k->AT_to_Ki = idup(s->AT_to_Ki, s->zA); k->GT_to_Ki = idup(s->GT_to_Ki, s->zG);
    (void)p;
    k->ldlt = orc_sparse_ldlt_clone(s->ldlt);
    k->rhs = (double *)xc((size_t)N, sizeof(double)); k->rhs_perm = (double *)xc((size_t)N, sizeof(double));
    return &k->base;
}

/* test hooks (the integer work of the constructor, for tests/test_symbolic_parity.py) */
int orc_sparse_cond_kkt_dim(const orc_kkt *k) { return ((const cond_kkt *)k)->N; }
int orc_sparse_cond_kkt_nnz(const orc_kkt *k) { return ((const cond_kkt *)k)->nnzK; }
const int *orc_sparse_cond_kkt_perm(const orc_kkt *k) { return ((const cond_kkt *)k)->P; }
const int *orc_sparse_cond_kkt_PKPt_colptr(const orc_kkt *k) { return ((const cond_kkt *)k)->PKPt_p; }
const int *orc_sparse_cond_kkt_PKPt_rowind(const orc_kkt *k) { return ((const cond_kkt *)k)->PKPt_i; }
const int *orc_sparse_cond_kkt_PKi(const orc_kkt *k) { return ((const cond_kkt *)k)->PKi; }
