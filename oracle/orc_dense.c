/*
 * oracle/orc_dense.c -- CPU restatement of PIQP's dense KKT backend (TEST INFRASTRUCTURE ONLY).
 *
 * Follows (paths under /root/reference/include/piqp/):
 *   dense/data.hpp:53-208          Data ctor, set_h_l/u, disable_inf_constraints, set_x_l/u
 *   dense/kkt.hpp:39-160           KKT ctor, update_data, update_scalings_and_factor, solve, eval_*, update_kkt
 *   dense/ldlt_no_pivot.hpp:278-354,393-450   LDLTNoPivot unblocked/blocked/compute/solve
 *   Eigen 3.4 Cholesky/LLT.h (third-party, absent here; semantics per SURVEY.md A.4):
 *       unblocked: x = a_kk - |A10|^2; fail if x <= 0; a_kk = sqrt(x); A21 = (A21 - A20 A10^T)/a_kk
 *       blocked:   bs = clamp(((n/8)/16)*16, 8, 128); potrf(A11); A21 <- A21 A11^-T; A22_L -= A21 A21^T
 *
 * The register-blocked GEMM micro-kernel below exists only so that the CPU baseline is not a
 * straw man (BASELINE.md "Fairness guard"); it computes the same sums in a different order.
 */
#include "orc.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 1;
void orc_set_num_threads(int t) { g_threads = t < 1 ? 1 : t; }

static void *xmalloc(size_t bytes)
{
    void *p = NULL;
    if (bytes == 0) bytes = 64;
    if (posix_memalign(&p, 64, (bytes + 63) & ~(size_t)63)) { fprintf(stderr, "orc: out of memory\n"); abort(); }
    return p;
}
static double *dalloc(size_t n) { double *p = (double *)xmalloc(n * sizeof(double)); memset(p, 0, n * sizeof(double)); return p; }
static int *ialloc(size_t n) { int *p = (int *)xmalloc(n * sizeof(int)); memset(p, 0, n * sizeof(int)); return p; }
static double *ddup(const double *s, size_t n) { double *p = dalloc(n); if (s && n) memcpy(p, s, n * sizeof(double)); return p; }
static int *idup(const int *s, size_t n) { int *p = ialloc(n); if (s && n) memcpy(p, s, n * sizeof(int)); return p; }

/* ------------------------------------------------------------------ data */

/* dense/data.hpp:98-119 */
void orc_data_set_h_l(orc_data *d, const double *h_l)
{
    d->n_h_l = 0;
    if (h_l) {
        int i_l = 0;
        for (int i = 0; i < d->m; i++) {
            double v = h_l[i];
            if (v > -ORC_INF) { d->n_h_l += 1; d->h_l[i] = v; d->h_l_idx[i_l++] = i; }
            else d->h_l[i] = -ORC_INF;
        }
    } else for (int i = 0; i < d->m; i++) d->h_l[i] = -ORC_INF;
}
/* dense/data.hpp:121-142 */
void orc_data_set_h_u(orc_data *d, const double *h_u)
{
    d->n_h_u = 0;
    if (h_u) {
        int i_u = 0;
        for (int i = 0; i < d->m; i++) {
            double v = h_u[i];
            if (v < ORC_INF) { d->n_h_u += 1; d->h_u[i] = v; d->h_u_idx[i_u++] = i; }
            else d->h_u[i] = ORC_INF;
        }
    } else for (int i = 0; i < d->m; i++) d->h_u[i] = ORC_INF;
}
/* dense/data.hpp:144-169, sparse/data.hpp analogous (row of G zeroed = entries of GT column zeroed) */
void orc_data_disable_inf_constraints(orc_data *d)
{
    int changed = 0;
    for (int i = 0; i < d->m; i++) {
        if (d->h_l[i] <= -ORC_INF && d->h_u[i] >= ORC_INF) {
            if (d->is_sparse) {
                for (int k = d->sGT.colptr[i]; k < d->sGT.colptr[i + 1]; k++) d->sGT.val[k] = 0.0;
            } else {
                memset(d->GT + (size_t)i * d->n, 0, sizeof(double) * (size_t)d->n);
            }
            d->h_l[i] = -1.0;
            d->h_u[i] = 1.0;
            changed = 1;
        }
    }
    if (changed) {
        double *hl = ddup(d->h_l, d->m), *hu = ddup(d->h_u, d->m);
        orc_data_set_h_l(d, hl);
        orc_data_set_h_u(d, hu);
        free(hl); free(hu);
    }
}
/* dense/data.hpp:171-188 */
void orc_data_set_x_l(orc_data *d, const double *x_l)
{
    d->n_x_l = 0;
    if (x_l) {
        int i_l = 0;
        for (int i = 0; i < d->n; i++)
            if (x_l[i] > -ORC_INF) { d->n_x_l += 1; d->x_l[i_l] = x_l[i]; d->x_l_idx[i_l] = i; i_l++; }
    }
}
/* dense/data.hpp:190-207 */
void orc_data_set_x_u(orc_data *d, const double *x_u)
{
    d->n_x_u = 0;
    if (x_u) {
        int i_u = 0;
        for (int i = 0; i < d->n; i++)
            if (x_u[i] < ORC_INF) { d->n_x_u += 1; d->x_u[i_u] = x_u[i]; d->x_u_idx[i_u] = i; i_u++; }
    }
}

static void data_alloc_common(orc_data *d)
{
    int n = d->n, p = d->p, m = d->m;
    d->c = dalloc(n); d->b = dalloc(p); d->h_l = dalloc(m); d->h_u = dalloc(m);
    d->x_l = dalloc(n); d->x_u = dalloc(n);
    d->h_l_idx = ialloc(m); d->h_u_idx = ialloc(m); d->x_l_idx = ialloc(n); d->x_u_idx = ialloc(n);
    d->x_b_scaling = dalloc(n);
    for (int i = 0; i < n; i++) d->x_b_scaling[i] = 1.0;
}

static void data_set_vectors(orc_data *d, const double *c, const double *b, const double *h_l, const double *h_u,
                             const double *x_l, const double *x_u)
{
    memcpy(d->c, c, sizeof(double) * (size_t)d->n);
    if (b && d->p) memcpy(d->b, b, sizeof(double) * (size_t)d->p);
    orc_data_set_h_l(d, h_l);
    orc_data_set_h_u(d, h_u);
    orc_data_disable_inf_constraints(d);
    orc_data_set_x_l(d, x_l);
    orc_data_set_x_u(d, x_u);
}

/* solver.hpp:169-192 (setup_impl data part), dense/data.hpp:53-96 */
orc_data *orc_data_create_dense(int n, int p, int m, const double *P, const double *c, const double *A,
                                const double *b, const double *G, const double *h_l, const double *h_u,
                                const double *x_l, const double *x_u)
{
    orc_data *d = (orc_data *)calloc(1, sizeof(orc_data));
    d->is_sparse = 0; d->n = n; d->p = A ? p : 0; d->m = G ? m : 0;
    p = d->p; m = d->m;
    d->P_utri = dalloc((size_t)n * n);
    d->AT = dalloc((size_t)n * p);
    d->GT = dalloc((size_t)n * m);
    for (int j = 0; j < n; j++)
        for (int i = 0; i <= j; i++) d->P_utri[i + (size_t)j * n] = P[i + (size_t)j * n];
    for (int k = 0; k < p; k++)
        for (int i = 0; i < n; i++) d->AT[i + (size_t)k * n] = A[k + (size_t)i * p];
    for (int k = 0; k < m; k++)
        for (int i = 0; i < n; i++) d->GT[i + (size_t)k * n] = G[k + (size_t)i * m];
    data_alloc_common(d);
    data_set_vectors(d, c, b, h_l, h_u, x_l, x_u);
    return d;
}

static void csc_copy(orc_csc *dst, const orc_csc *src)
{
    dst->rows = src->rows; dst->cols = src->cols;
    int nnz = src->colptr ? src->colptr[src->cols] : 0;
    dst->colptr = idup(src->colptr, (size_t)src->cols + 1);
    dst->rowind = idup(src->rowind, nnz);
    dst->val = ddup(src->val, nnz);
}
static void csc_free(orc_csc *c) { free(c->colptr); free(c->rowind); free(c->val); memset(c, 0, sizeof(*c)); }

/* transpose of a CSC matrix (rows x cols) -> CSC (cols x rows); entries come out sorted by row */
static void csc_transpose(int rows, int cols, const int *Ap, const int *Ai, const double *Ax, orc_csc *T)
{
    int nnz = Ap ? Ap[cols] : 0;
    T->rows = cols; T->cols = rows;
    T->colptr = ialloc((size_t)rows + 1);
    T->rowind = ialloc(nnz);
    T->val = dalloc(nnz);
    for (int k = 0; k < nnz; k++) T->colptr[Ai[k] + 1]++;
    for (int i = 0; i < rows; i++) T->colptr[i + 1] += T->colptr[i];
    int *next = idup(T->colptr, rows + 1);
    for (int j = 0; j < cols; j++)
        for (int k = Ap[j]; k < Ap[j + 1]; k++) {
            int q = next[Ai[k]]++;
            T->rowind[q] = j;
            T->val[q] = Ax[k];
        }
    free(next);
}

/* solver.hpp:182-184 for SparseSolver: P_utri = upper triangle of P, AT = A^T, GT = G^T */
orc_data *orc_data_create_sparse(int n, int p, int m, const int *Pp, const int *Pi, const double *Px,
                                 const double *c, const int *Ap, const int *Ai, const double *Ax,
                                 const double *b, const int *Gp, const int *Gi, const double *Gx,
                                 const double *h_l, const double *h_u, const double *x_l, const double *x_u)
{
    orc_data *d = (orc_data *)calloc(1, sizeof(orc_data));
    d->is_sparse = 1; d->n = n; d->p = Ap ? p : 0; d->m = Gp ? m : 0;
    p = d->p; m = d->m;
    /* upper triangle of P, rows sorted */
    int nnz_u = 0;
    for (int j = 0; j < n; j++) for (int k = Pp[j]; k < Pp[j + 1]; k++) if (Pi[k] <= j) nnz_u++;
    d->sP_utri.rows = n; d->sP_utri.cols = n;
    d->sP_utri.colptr = ialloc((size_t)n + 1);
    d->sP_utri.rowind = ialloc(nnz_u);
    d->sP_utri.val = dalloc(nnz_u);
    int q = 0;
    for (int j = 0; j < n; j++) {
        int start = q;
        for (int k = Pp[j]; k < Pp[j + 1]; k++) if (Pi[k] <= j) { d->sP_utri.rowind[q] = Pi[k]; d->sP_utri.val[q] = Px[k]; q++; }
        /* insertion sort by row (inputs are normally sorted already) */
        for (int a = start + 1; a < q; a++) {
            int ri = d->sP_utri.rowind[a]; double rv = d->sP_utri.val[a]; int bq = a - 1;
            while (bq >= start && d->sP_utri.rowind[bq] > ri) { d->sP_utri.rowind[bq + 1] = d->sP_utri.rowind[bq]; d->sP_utri.val[bq + 1] = d->sP_utri.val[bq]; bq--; }
            d->sP_utri.rowind[bq + 1] = ri; d->sP_utri.val[bq + 1] = rv;
        }
        d->sP_utri.colptr[j + 1] = q;
    }
    if (p > 0) csc_transpose(p, n, Ap, Ai, Ax, &d->sAT);
    else { d->sAT.rows = n; d->sAT.cols = 0; d->sAT.colptr = ialloc(1); d->sAT.rowind = ialloc(0); d->sAT.val = dalloc(0); }
    if (m > 0) csc_transpose(m, n, Gp, Gi, Gx, &d->sGT);
    else { d->sGT.rows = n; d->sGT.cols = 0; d->sGT.colptr = ialloc(1); d->sGT.rowind = ialloc(0); d->sGT.val = dalloc(0); }
    data_alloc_common(d);
    data_set_vectors(d, c, b, h_l, h_u, x_l, x_u);
    return d;
}

orc_data *orc_data_clone(const orc_data *s)
{
    orc_data *d = (orc_data *)calloc(1, sizeof(orc_data));
    *d = *s;
    int n = s->n, p = s->p, m = s->m;
    if (s->is_sparse) {
        csc_copy(&d->sP_utri, &s->sP_utri); csc_copy(&d->sAT, &s->sAT); csc_copy(&d->sGT, &s->sGT);
        d->P_utri = d->AT = d->GT = NULL;
    } else {
        d->P_utri = ddup(s->P_utri, (size_t)n * n); d->AT = ddup(s->AT, (size_t)n * p); d->GT = ddup(s->GT, (size_t)n * m);
    }
    d->c = ddup(s->c, n); d->b = ddup(s->b, p); d->h_l = ddup(s->h_l, m); d->h_u = ddup(s->h_u, m);
    d->x_l = ddup(s->x_l, n); d->x_u = ddup(s->x_u, n);
    d->h_l_idx = idup(s->h_l_idx, m); d->h_u_idx = idup(s->h_u_idx, m);
    d->x_l_idx = idup(s->x_l_idx, n); d->x_u_idx = idup(s->x_u_idx, n);
    d->x_b_scaling = ddup(s->x_b_scaling, n);
    return d;
}

void orc_data_free(orc_data *d)
{
    if (!d) return;
    free(d->P_utri); free(d->AT); free(d->GT);
    if (d->is_sparse) { csc_free(&d->sP_utri); csc_free(&d->sAT); csc_free(&d->sGT); }
    free(d->c); free(d->b); free(d->h_l); free(d->h_u); free(d->x_l); free(d->x_u);
    free(d->h_l_idx); free(d->h_u_idx); free(d->x_l_idx); free(d->x_u_idx); free(d->x_b_scaling);
    free(d);
}

/* -------------------------------------------------- blocked GEMM helper (baseline fairness only) */
/*
 * C_lower(n x n) += alpha * A(n x k) * B(k x n), with A col-major (lda) and B given through its
 * TRANSPOSE Bt (n x k col-major, ldb) plus an optional per-k scale w[k] (B[k][j] = w[k]*Bt[j][k]).
 * This covers  K_L += GT * (diag(w) GT^T)  (dense/kkt.hpp:157-158),  AT_A = AT * AT^T  (:53) and
 * A22_L -= A21 * A21^T  (Eigen LLT rankUpdate).
 * Register-blocked MR x NR micro-kernel on packed panels (GotoBLAS structure).
 */
#if defined(__AVX512F__)
typedef double vd __attribute__((vector_size(64)));
#define VL 8
#define NR 12
#else
typedef double vd __attribute__((vector_size(32)));
#define VL 4
#define NR 6
#endif
#define MR (2 * VL)
#define KC 256
#define MC 192 /* multiple of MR for both vector lengths (192 = 12*16 = 24*8) */

static inline void micro_kernel(int kc, const double *restrict Ap, const double *restrict Bp, double *restrict acc /* MR x NR col-major */)
{
    vd c0[NR], c1[NR];
    for (int j = 0; j < NR; j++) { c0[j] = (vd){0}; c1[j] = (vd){0}; }
    for (int k = 0; k < kc; k++) {
        vd a0, a1;
        memcpy(&a0, Ap + (size_t)k * MR, sizeof(vd));
        memcpy(&a1, Ap + (size_t)k * MR + VL, sizeof(vd));
        const double *b = Bp + (size_t)k * NR;
#pragma GCC unroll 12
        for (int j = 0; j < NR; j++) {
            vd bj = (vd){0} + b[j];
            c0[j] += a0 * bj;
            c1[j] += a1 * bj;
        }
    }
    for (int j = 0; j < NR; j++) {
        memcpy(acc + (size_t)j * MR, &c0[j], sizeof(vd));
        memcpy(acc + (size_t)j * MR + VL, &c1[j], sizeof(vd));
    }
}

static void syrk_like_lower(int n, int k, double alpha, const double *A, int lda, const double *Bt, int ldb,
                            const double *w, double *C, int ldc)
{
    if (n <= 0 || k <= 0) return;
    int nthreads = g_threads;
#ifndef _OPENMP
    nthreads = 1;
#endif
    /* Tasks = lower-triangular pairs (row block of MC rows, column block of NCB columns), largest first, drawn dynamically: with row blocks only
     * (rounds 1-3) n = 4096 gave 22 tasks of very different size to 32 threads and 9 GFLOP/s per thread; a K block is the outermost loop so that
     * every task adds its block's contribution to a C block no other task touches. */
    enum { NCB = 192 }; /* multiple of NR for both vector lengths */
    const int nrow_blocks = (n + MC - 1) / MC, ncol_blocks = (n + NCB - 1) / NCB;
    int ntask = 0;
    int *task = (int *)xmalloc(sizeof(int) * 2 * (size_t)nrow_blocks * ncol_blocks);
    for (int ib = nrow_blocks - 1; ib >= 0; ib--)
        for (int jb = 0; jb < ncol_blocks; jb++)
            if (jb * NCB < ib * MC + MC && jb * NCB < n) { task[2 * ntask] = ib; task[2 * ntask + 1] = jb; ntask++; }
    for (int pc = 0; pc < k; pc += KC) {
        int kc = k - pc < KC ? k - pc : KC;
        int ncol_panels = (n + NR - 1) / NR;
        /* pack all of B for this K block: panels of NR columns */
        double *Bpack = (double *)xmalloc(sizeof(double) * (size_t)ncol_panels * NR * kc);
#pragma omp parallel for num_threads(nthreads) schedule(static)
        for (int jp = 0; jp < ncol_panels; jp++) {
            double *dst = Bpack + (size_t)jp * NR * kc;
            for (int kk = 0; kk < kc; kk++) {
                double wk = w ? w[pc + kk] : 1.0;
                for (int jj = 0; jj < NR; jj++) {
                    int j = jp * NR + jj;
                    dst[(size_t)kk * NR + jj] = j < n ? wk * Bt[j + (size_t)(pc + kk) * ldb] : 0.0;
                }
            }
        }
#pragma omp parallel num_threads(nthreads)
        {
            double *Apack = (double *)xmalloc(sizeof(double) * (size_t)MC * kc);
            double acc[MR * NR];
            int packed_ib = -1;
#pragma omp for schedule(dynamic, 1)
            for (int t = 0; t < ntask; t++) {
                const int ib = task[2 * t], jb = task[2 * t + 1];
                int i0 = ib * MC;
                int mc = n - i0 < MC ? n - i0 : MC;
                int nrp = (mc + MR - 1) / MR;
                if (packed_ib != ib) { /* (a thread often draws neighbouring column blocks of one row block) */
                    for (int ip = 0; ip < nrp; ip++) {
                        double *dst = Apack + (size_t)ip * MR * kc;
                        for (int kk = 0; kk < kc; kk++)
                            for (int ii = 0; ii < MR; ii++) {
                                int i = i0 + ip * MR + ii;
                                dst[(size_t)kk * MR + ii] = i < n ? A[i + (size_t)(pc + kk) * lda] : 0.0;
                            }
                    }
                    packed_ib = ib;
                }
                /* column panels of this column block, up to the last row of the row block */
                int jmax = i0 + mc; /* exclusive */
                int jp_begin = jb * NCB / NR, jp_end = (jb * NCB + NCB) / NR;
                int jp_lim = (jmax + NR - 1) / NR;
                if (jp_end > jp_lim) jp_end = jp_lim;
                if (jp_end > ncol_panels) jp_end = ncol_panels;
                for (int jp = jp_begin; jp < jp_end; jp++) {
                    int j0 = jp * NR;
                    for (int ip = 0; ip < nrp; ip++) {
                        int r0 = i0 + ip * MR;
                        if (r0 + MR <= j0) continue; /* tile entirely above the diagonal */
                        micro_kernel(kc, Apack + (size_t)ip * MR * kc, Bpack + (size_t)jp * NR * kc, acc);
                        for (int jj = 0; jj < NR; jj++) {
                            int j = j0 + jj;
                            if (j >= n) break;
                            for (int ii = 0; ii < MR; ii++) {
                                int i = r0 + ii;
                                if (i >= n) break;
                                if (i >= j) C[i + (size_t)j * ldc] += alpha * acc[ii + jj * MR];
                            }
                        }
                    }
                }
            }
            free(Apack);
        }
        free(Bpack);
    }
    free(task);
}

/* ------------------------------------------------------------ Eigen::LLT */

/* Eigen LLT.h llt_inplace<Lower>::unblocked */
static int llt_unblocked(double *a, int n, int lda)
{
    for (int k = 0; k < n; k++) {
        int rs = n - k - 1;
        double x = a[k + (size_t)k * lda];
        if (k > 0) {
            double s = 0.0;
            for (int j = 0; j < k; j++) { double v = a[k + (size_t)j * lda]; s += v * v; }
            x -= s;
        }
        if (!(x > 0.0)) return k; /* x <= 0 (NaN also fails: sqrt would poison the factor) */
        x = sqrt(x);
        a[k + (size_t)k * lda] = x;
        if (k > 0 && rs > 0) {
            /* A21 -= A20 * A10^T */
            for (int j = 0; j < k; j++) {
                double akj = a[k + (size_t)j * lda];
                const double *col = a + (k + 1) + (size_t)j * lda;
                double *dst = a + (k + 1) + (size_t)k * lda;
                for (int i = 0; i < rs; i++) dst[i] -= col[i] * akj;
            }
        }
        if (rs > 0) {
            double *dst = a + (k + 1) + (size_t)k * lda;
            for (int i = 0; i < rs; i++) dst[i] /= x;
        }
    }
    return -1;
}

static int block_size_rule(int size)
{
    /* dense/ldlt_no_pivot.hpp:321-323 == Eigen LLT.h blocked() */
    int bs = size / 8;
    bs = (bs / 16) * 16;
    if (bs < 8) bs = 8;
    if (bs > 128) bs = 128;
    return bs;
}

/* A21 <- A21 * L11^-T  (Eigen: A11.adjoint().triangularView<Upper>().solveInPlace<OnTheRight>(A21)) */
static void trsm_right_lower_trans(int rs, int bs, const double *l11, int ldl, double *a21, int lda, int unit_diag)
{
    int nthreads = g_threads;
#ifndef _OPENMP
    nthreads = 1;
#endif
    const int RB = 64;
    int nblk = (rs + RB - 1) / RB;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int ib = 0; ib < nblk; ib++) {
        int r0 = ib * RB, rb = rs - r0 < RB ? rs - r0 : RB;
        for (int j = 0; j < bs; j++) {
            double *xj = a21 + r0 + (size_t)j * lda;
            for (int k = 0; k < j; k++) {
                double ljk = l11[j + (size_t)k * ldl];
                const double *xk = a21 + r0 + (size_t)k * lda;
                for (int i = 0; i < rb; i++) xj[i] -= xk[i] * ljk;
            }
            if (!unit_diag) {
                double d = l11[j + (size_t)j * ldl];
                for (int i = 0; i < rb; i++) xj[i] /= d;
            }
        }
    }
}

/* Eigen LLT.h llt_inplace<Lower>::blocked */
int orc_llt_compute(double *a, int n, int lda)
{
    if (n < 32) return llt_unblocked(a, n, lda);
    int blockSize = block_size_rule(n);
    for (int k = 0; k < n; k += blockSize) {
        int bs = n - k < blockSize ? n - k : blockSize;
        int rs = n - k - bs;
        double *A11 = a + k + (size_t)k * lda;
        double *A21 = a + (k + bs) + (size_t)k * lda;
        double *A22 = a + (k + bs) + (size_t)(k + bs) * lda;
        int ret = llt_unblocked(A11, bs, lda);
        if (ret >= 0) return k + ret;
        if (rs > 0) {
            trsm_right_lower_trans(rs, bs, A11, lda, A21, lda, 0);
            syrk_like_lower(rs, bs, -1.0, A21, lda, A21, lda, NULL, A22, lda);
        }
    }
    return -1;
}

/* Eigen LLT::solveInPlace: matrixL().solveInPlace(b); matrixU().solveInPlace(b) */
void orc_llt_solve_inplace(const double *l, int n, int lda, double *x)
{
    for (int j = 0; j < n; j++) { /* column-oriented forward substitution */
        double xj = x[j] / l[j + (size_t)j * lda];
        x[j] = xj;
        const double *col = l + (size_t)j * lda;
        for (int i = j + 1; i < n; i++) x[i] -= col[i] * xj;
    }
    for (int j = n - 1; j >= 0; j--) { /* L^T x = y : row-oriented (dot with column j of L) */
        const double *col = l + (size_t)j * lda;
        double s = x[j];
        for (int i = j + 1; i < n; i++) s -= col[i] * x[i];
        x[j] = s / col[j];
    }
}

/* ----------------------------------------------------------- LDLTNoPivot */

/* dense/ldlt_no_pivot.hpp:278-311 */
static int ldlt_unblocked(double *a, int n, int lda, double *temp)
{
    for (int k = 0; k < n; k++) {
        int rs = n - k - 1;
        if (k > 0) {
            double s = 0.0;
            for (int j = 0; j < k; j++) { temp[j] = a[j + (size_t)j * lda] * a[k + (size_t)j * lda]; }
            for (int j = 0; j < k; j++) s += a[k + (size_t)j * lda] * temp[j];
            a[k + (size_t)k * lda] -= s;
            if (rs > 0) {
                double *dst = a + (k + 1) + (size_t)k * lda;
                for (int j = 0; j < k; j++) {
                    const double *col = a + (k + 1) + (size_t)j * lda;
                    double t = temp[j];
                    for (int i = 0; i < rs; i++) dst[i] -= col[i] * t;
                }
            }
        }
        double x = a[k + (size_t)k * lda];
        if (x == 0.0) return k;
        if (rs > 0) {
            double *dst = a + (k + 1) + (size_t)k * lda;
            for (int i = 0; i < rs; i++) dst[i] /= x;
        }
    }
    return -1;
}

/* dense/ldlt_no_pivot.hpp:313-354 (blocked) -- the A21_tmp scratch of :338 is a private buffer here
 * because callers of this oracle may keep meaningful data in the upper triangle. */
int orc_ldlt_no_pivot_compute(double *a, int n, int lda, double *work_n)
{
    if (n < 32) return ldlt_unblocked(a, n, lda, work_n);
    int blockSize = block_size_rule(n);
    double *tmp = dalloc((size_t)n * blockSize);
    double *dvec = dalloc(blockSize);
    int result = -1;
    for (int k = 0; k < n; k += blockSize) {
        int bs = n - k < blockSize ? n - k : blockSize;
        int rs = n - k - bs;
        double *A11 = a + k + (size_t)k * lda;
        double *A21 = a + (k + bs) + (size_t)k * lda;
        double *A22 = a + (k + bs) + (size_t)(k + bs) * lda;
        int ret = ldlt_unblocked(A11, bs, lda, work_n);
        if (ret >= 0) { result = k + ret; break; }
        if (rs > 0) {
            /* :345-346  A21 = A21 (A11^T unit-upper)^-1 D11^-1 */
            trsm_right_lower_trans(rs, bs, A11, lda, A21, lda, 1);
            for (int j = 0; j < bs; j++) {
                double dinv = 1.0 / A11[j + (size_t)j * lda];
                dvec[j] = A11[j + (size_t)j * lda];
                double *col = A21 + (size_t)j * lda;
                for (int i = 0; i < rs; i++) col[i] *= dinv;
            }
            /* :349-350  A22_L -= (A21 D11) A21^T */
            for (int j = 0; j < bs; j++) {
                const double *col = A21 + (size_t)j * lda;
                double *t = tmp + (size_t)j * rs;
                for (int i = 0; i < rs; i++) t[i] = col[i] * dvec[j];
            }
            syrk_like_lower(rs, bs, -1.0, tmp, rs, A21, lda, NULL, A22, lda);
        }
    }
    free(tmp); free(dvec);
    return result;
}

/* dense/ldlt_no_pivot.hpp:432-450 */
void orc_ldlt_no_pivot_solve_inplace(const double *ld, int n, int lda, double *x)
{
    for (int j = 0; j < n; j++) {
        double xj = x[j];
        const double *col = ld + (size_t)j * lda;
        for (int i = j + 1; i < n; i++) x[i] -= col[i] * xj;
    }
    for (int j = 0; j < n; j++) x[j] /= ld[j + (size_t)j * lda];
    for (int j = n - 1; j >= 0; j--) {
        const double *col = ld + (size_t)j * lda;
        double s = x[j];
        for (int i = j + 1; i < n; i++) s -= col[i] * x[i];
        x[j] = s;
    }
}

/* ------------------------------------------------------- dense::KKT<T> */

typedef struct {
    orc_kkt base;
    int n, p, m;
    int use_ldlt;
    double m_delta;
    double *m_z_reg_inv; /* m */
    double *kkt_mat;     /* n x n (lower used) */
    double *fac;         /* n x n: the factor (Eigen::LLT keeps its own copy, dense/kkt.hpp:82) */
    double *AT_A;        /* n x n lower, only if p > 0 */
    double *work_z;      /* m */
    double *work_n;      /* n */
} dense_kkt;

/* dense/kkt.hpp:51-54, :66-69 : AT_A.lower = AT * AT^T */
static void dense_compute_ATA(dense_kkt *k, const orc_data *d)
{
    if (k->p <= 0) return;
    memset(k->AT_A, 0, sizeof(double) * (size_t)k->n * k->n);
    syrk_like_lower(k->n, k->p, 1.0, d->AT, k->n, d->AT, k->n, NULL, k->AT_A, k->n);
}

static orc_kkt *dense_clone(const orc_kkt *self);
static void dense_destroy(orc_kkt *self)
{
    dense_kkt *k = (dense_kkt *)self;
    free(k->m_z_reg_inv); free(k->kkt_mat); free(k->fac); free(k->AT_A); free(k->work_z); free(k->work_n);
    free(k);
}

/* dense/kkt.hpp:62-71 */
static void dense_update_data(orc_kkt *self, const orc_data *d, int options)
{
    dense_kkt *k = (dense_kkt *)self;
    if (options & ORC_KKT_UPDATE_A) dense_compute_ATA(k, d);
}

/* dense/kkt.hpp:140-160 */
static void dense_update_kkt(dense_kkt *k, const orc_data *d, const double *x_reg)
{
    int n = k->n;
    for (int j = 0; j < n; j++) {
        double *col = k->kkt_mat + (size_t)j * n;
        for (int i = j; i < n; i++) col[i] = d->P_utri[j + (size_t)i * n]; /* lower = P_utri^T */
        col[j] += x_reg[j];
    }
    if (k->p > 0) {
        double dinv = 1.0 / k->m_delta;
        for (int j = 0; j < n; j++) {
            double *col = k->kkt_mat + (size_t)j * n;
            const double *src = k->AT_A + (size_t)j * n;
            for (int i = j; i < n; i++) col[i] += dinv * src[i];
        }
    }
    if (k->m > 0) {
        /* W_delta_inv_G = diag(z_reg_inv) * GT^T ; kkt_mat.lower += GT * W_delta_inv_G */
        syrk_like_lower(n, k->m, 1.0, d->GT, n, d->GT, n, k->m_z_reg_inv, k->kkt_mat, n);
    }
}

/* dense/kkt.hpp:73-84 */
static int dense_factor(orc_kkt *self, const orc_data *d, double delta, const double *x_reg, const double *z_reg)
{
    dense_kkt *k = (dense_kkt *)self;
    k->m_delta = delta;
    for (int i = 0; i < k->m; i++) k->m_z_reg_inv[i] = 1.0 / z_reg[i];
    dense_update_kkt(k, d, x_reg);
    /* llt.compute(kkt_mat): copies the matrix then factors the copy in place */
    int n = k->n;
    for (int j = 0; j < n; j++)
        memcpy(k->fac + j + (size_t)j * n, k->kkt_mat + j + (size_t)j * n, sizeof(double) * (size_t)(n - j));
    int ret = k->use_ldlt ? orc_ldlt_no_pivot_compute(k->fac, n, n, k->work_n) : orc_llt_compute(k->fac, n, n);
    return ret == -1;
}

static void gemv_n(int rows, int cols, double alpha, const double *A, int lda, const double *x, double *y /* += */)
{
    for (int j = 0; j < cols; j++) {
        double xj = alpha * x[j];
        const double *col = A + (size_t)j * lda;
        for (int i = 0; i < rows; i++) y[i] += col[i] * xj;
    }
}
static void gemv_t(int rows, int cols, double alpha, const double *A, int lda, const double *x, double *y /* = */)
{
    for (int j = 0; j < cols; j++) {
        const double *col = A + (size_t)j * lda;
        double s = 0.0;
        for (int i = 0; i < rows; i++) s += col[i] * x[i];
        y[j] = alpha * s;
    }
}

/* dense/kkt.hpp:86-105 */
static void dense_solve(orc_kkt *self, const orc_data *d, const double *rhs_x, const double *rhs_y, const double *rhs_z,
                        double *lhs_x, double *lhs_y, double *lhs_z)
{
    dense_kkt *k = (dense_kkt *)self;
    int n = k->n, p = k->p, m = k->m;
    double delta_inv = 1.0 / k->m_delta;
    memcpy(lhs_x, rhs_x, sizeof(double) * (size_t)n);
    for (int i = 0; i < m; i++) k->work_z[i] = k->m_z_reg_inv[i] * rhs_z[i];
    gemv_n(n, m, 1.0, d->GT, n, k->work_z, lhs_x);
    gemv_n(n, p, delta_inv, d->AT, n, rhs_y, lhs_x);
    if (k->use_ldlt) orc_ldlt_no_pivot_solve_inplace(k->fac, n, n, lhs_x);
    else orc_llt_solve_inplace(k->fac, n, n, lhs_x);
    gemv_t(n, p, delta_inv, d->AT, n, lhs_x, lhs_y);
    for (int i = 0; i < p; i++) lhs_y[i] -= delta_inv * rhs_y[i];
    gemv_t(n, m, 1.0, d->GT, n, lhs_x, lhs_z);
    for (int i = 0; i < m; i++) { lhs_z[i] -= rhs_z[i]; lhs_z[i] *= k->m_z_reg_inv[i]; }
}

/* dense/kkt.hpp:108-114 : z = alpha * sym(P_utri) * x */
static void dense_eval_P_x(orc_kkt *self, const orc_data *d, double alpha, const double *x, double *z)
{
    dense_kkt *k = (dense_kkt *)self;
    int n = k->n;
    memset(z, 0, sizeof(double) * (size_t)n);
    for (int j = 0; j < n; j++) {
        const double *col = d->P_utri + (size_t)j * n;
        double xj = alpha * x[j];
        double s = 0.0;
        for (int i = 0; i < j; i++) { z[i] += col[i] * xj; s += col[i] * x[i]; }
        z[j] += col[j] * xj + alpha * s;
    }
}
/* dense/kkt.hpp:117-123 */
static void dense_eval_A(orc_kkt *self, const orc_data *d, double alpha_n, double alpha_t, const double *xn,
                         const double *xt, double *zn, double *zt)
{
    dense_kkt *k = (dense_kkt *)self;
    gemv_t(k->n, k->p, alpha_n, d->AT, k->n, xn, zn);
    memset(zt, 0, sizeof(double) * (size_t)k->n);
    gemv_n(k->n, k->p, alpha_t, d->AT, k->n, xt, zt);
}
/* dense/kkt.hpp:126-132 */
static void dense_eval_G(orc_kkt *self, const orc_data *d, double alpha_n, double alpha_t, const double *xn,
                         const double *xt, double *zn, double *zt)
{
    dense_kkt *k = (dense_kkt *)self;
    gemv_t(k->n, k->m, alpha_n, d->GT, k->n, xn, zn);
    memset(zt, 0, sizeof(double) * (size_t)k->n);
    gemv_n(k->n, k->m, alpha_t, d->GT, k->n, xt, zt);
}
static void dense_print_info(orc_kkt *self) { (void)self; }

static void dense_fill_vtable(dense_kkt *k)
{
    k->base.clone = dense_clone;
    k->base.update_data = dense_update_data;
    k->base.update_scalings_and_factor = dense_factor;
    k->base.solve = dense_solve;
    k->base.eval_P_x = dense_eval_P_x;
    k->base.eval_A_xn_and_AT_xt = dense_eval_A;
    k->base.eval_G_xn_and_GT_xt = dense_eval_G;
    k->base.print_info = dense_print_info;
    k->base.destroy = dense_destroy;
}

/* dense/kkt.hpp:39-55 */
orc_kkt *orc_dense_kkt_create(const orc_data *d, int use_ldlt)
{
    dense_kkt *k = (dense_kkt *)calloc(1, sizeof(dense_kkt));
    dense_fill_vtable(k);
    k->n = d->n; k->p = d->p; k->m = d->m; k->use_ldlt = use_ldlt;
    k->m_delta = 1.0;
    k->m_z_reg_inv = dalloc(k->m);
    k->work_z = dalloc(k->m);
    k->work_n = dalloc(k->n);
    k->kkt_mat = dalloc((size_t)k->n * k->n);
    k->fac = dalloc((size_t)k->n * k->n);
    if (k->p > 0) { k->AT_A = dalloc((size_t)k->n * k->n); dense_compute_ATA(k, d); }
    return &k->base;
}

/* dense/kkt.hpp:57-60 */
static orc_kkt *dense_clone(const orc_kkt *self)
{
    const dense_kkt *s = (const dense_kkt *)self;
    dense_kkt *k = (dense_kkt *)calloc(1, sizeof(dense_kkt));
    *k = *s;
    size_t nn = (size_t)s->n * s->n;
    k->m_z_reg_inv = ddup(s->m_z_reg_inv, s->m);
    k->work_z = ddup(s->work_z, s->m);
    k->work_n = ddup(s->work_n, s->n);
    k->kkt_mat = ddup(s->kkt_mat, nn);
    k->fac = ddup(s->fac, nn);
    k->AT_A = s->AT_A ? ddup(s->AT_A, nn) : NULL;
    return &k->base;
}

const double *orc_dense_kkt_internal_kkt_mat(const orc_kkt *k) { return ((const dense_kkt *)k)->kkt_mat; }
const double *orc_dense_kkt_internal_factor(const orc_kkt *k) { return ((const dense_kkt *)k)->fac; }
