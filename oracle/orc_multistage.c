/*
 * oracle/orc_multistage.c -- CPU restatement of PIQP's `sparse_multistage` KKT backend.
 *
 * TEST INFRASTRUCTURE ONLY (see orc.h).  Follows sparse/multistage_kkt.hpp and
 * sparse/blocksparse/{block_info,block_kkt,block_mat,block_vec}.hpp of the reference.
 *
 * The reference stores its blocks in blasfeo's panel-major `blasfeo_dmat` and calls blasfeo's
 * BLAS-like kernels (un-vendored, unpinned third party).  Here every block is a plain column-major
 * array and every blasfeo call is a plain loop with the semantics documented in
 * utils/blasfeo_wrapper.hpp:20-119; the summation order inside a kernel is therefore NOT the
 * reference's ("parity unpinned" at the bit level), the block structure, the elimination order and
 * the algebra are.  Pinned by: the recorded block structure of the notebook QP (8,6 8,6 8,6 14,0 x3,
 * arrow 8) and by agreement with the sparse_ldlt backend to 1e-8 on the reference's own multistage
 * fixtures (tests/src/sparse/multistage_kkt_test.cpp:208-211) -> tests/test_oracle_multistage.py.
 *
 * Deviations (documented, both only reachable in degenerate structures):
 *  - derived blocks (AtA, GtG, kkt_fac) are always allocated at full block size and zeroed before
 *    accumulation; the reference allocates some of them smaller (multistage_kkt.hpp:879-883,960-964)
 *    and, with assertions compiled out, `blasfeo_dtrcpsc_l` into a larger kkt_fac.D block would leave
 *    stale entries (:1063-1066).  The arithmetic on the entries that exist is the same.
 *  - a non-positive Cholesky pivot zeroes its column (blasfeo reference dpotrf semantics:
 *    inverse pivot := 0); update_scalings_and_factor always reports success (:218).
 */
#include "orc.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned long long u64;

typedef struct { int start, diag, off; } blk_info; /* blocksparse/block_info.hpp:20-25 */

/* column-major block; a == NULL encodes a structurally absent block (unique_ptr == nullptr) */
typedef struct { int r, c; double *a; } dmat;

/* blocksparse/block_kkt.hpp: lower block-tridiagonal-arrow matrix. D[N], B[N-2], E[N-1] */
typedef struct { int N; dmat *D, *B, *E; } block_kkt;

/* blocksparse/block_mat.hpp:23-45.  Blocks are stored TRANSPOSED (store_transpose = true,
 * multistage_kkt.hpp:106-107): D[i] is diag_i x rows_i, B[i] is off_i x rows_i, E[i] is arrow x rows_i */
typedef struct {
    int N, rows_total, blocked_rows;
    int *perm, *perm_inv, *row_sizes /* N-1 */, *row_acc /* N */;
    dmat *D, *B, *E;
} block_mat;

typedef struct {
    orc_kkt base;
    int n, p, m, N, arrow;
    blk_info *bi;
    double m_delta;
    double *z_reg_inv, *work_z;
    block_kkt P, AtA, GtG, fac;
    block_mat AT, GT, GTs;
    double *G_scaling;             /* permuted constraint order */
    double *bx1, *bx2, *by1, *by2, *bz1, *bz2; /* BlockVec work buffers (contiguous, block offsets = bi[].start / row_acc[]) */
} ms_kkt;

static void *xcalloc(size_t n, size_t s) { void *p = calloc(n ? n : 1, s); if (!p) { fprintf(stderr, "oom\n"); abort(); } return p; }
static int imax(int a, int b) { return a > b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }
#define EL(M, i, j) ((M).a[(size_t)(i) + (size_t)(j) * (size_t)(M).r])

static void dmat_alloc(dmat *M, int r, int c) { if (!M->a) { M->r = r; M->c = c; M->a = (double *)xcalloc((size_t)r * (size_t)c, sizeof(double)); } }
static void dmat_zero(dmat *M) { if (M->a) memset(M->a, 0, sizeof(double) * (size_t)M->r * (size_t)M->c); }
static void dmat_free(dmat *M) { free(M->a); M->a = NULL; }
static dmat dmat_clone(const dmat *M)
{
    dmat R = *M;
    if (M->a) { size_t sz = (size_t)M->r * (size_t)M->c; R.a = (double *)xcalloc(sz, sizeof(double)); memcpy(R.a, M->a, sizeof(double) * sz); }
    return R;
}

static void bk_init(block_kkt *K, int N)
{
    K->N = N;
    K->D = (dmat *)xcalloc((size_t)N, sizeof(dmat));
    K->B = (dmat *)xcalloc((size_t)imax(N - 2, 0), sizeof(dmat));
    K->E = (dmat *)xcalloc((size_t)(N - 1), sizeof(dmat));
}
static void bk_free(block_kkt *K)
{
    if (!K->D) return;
    for (int i = 0; i < K->N; i++) dmat_free(&K->D[i]);
    for (int i = 0; i < K->N - 2; i++) dmat_free(&K->B[i]);
    for (int i = 0; i < K->N - 1; i++) dmat_free(&K->E[i]);
    free(K->D); free(K->B); free(K->E);
}
static block_kkt bk_clone(const block_kkt *K)
{
    block_kkt R; bk_init(&R, K->N);
    for (int i = 0; i < K->N; i++) R.D[i] = dmat_clone(&K->D[i]);
    for (int i = 0; i < K->N - 2; i++) R.B[i] = dmat_clone(&K->B[i]);
    for (int i = 0; i < K->N - 1; i++) R.E[i] = dmat_clone(&K->E[i]);
    return R;
}
static void bm_free(block_mat *A)
{
    if (!A->D) return;
    for (int i = 0; i < A->N - 1; i++) { dmat_free(&A->D[i]); dmat_free(&A->E[i]); }
    for (int i = 0; i < A->N - 2; i++) dmat_free(&A->B[i]);
    free(A->D); free(A->B); free(A->E); free(A->perm); free(A->perm_inv); free(A->row_sizes); free(A->row_acc);
}
static int *iclone(const int *s, int n) { int *p = (int *)xcalloc((size_t)n, sizeof(int)); if (n) memcpy(p, s, sizeof(int) * (size_t)n); return p; }
static block_mat bm_clone(const block_mat *A)
{
    block_mat R = *A;
    int N = A->N;
    R.perm = iclone(A->perm, A->rows_total); R.perm_inv = iclone(A->perm_inv, A->rows_total);
    R.row_sizes = iclone(A->row_sizes, N - 1); R.row_acc = iclone(A->row_acc, N);
    R.D = (dmat *)xcalloc((size_t)(N - 1), sizeof(dmat));
    R.B = (dmat *)xcalloc((size_t)imax(N - 2, 0), sizeof(dmat));
    R.E = (dmat *)xcalloc((size_t)(N - 1), sizeof(dmat));
    for (int i = 0; i < N - 1; i++) { R.D[i] = dmat_clone(&A->D[i]); R.E[i] = dmat_clone(&A->E[i]); }
    for (int i = 0; i < N - 2; i++) R.B[i] = dmat_clone(&A->B[i]);
    return R;
}

/* ---- flop models, multistage_kkt.hpp:396-418 (unsigned 64-bit, integer division as written) ---- */
static u64 flops_gemm(u64 m, u64 n, u64 k) { return 2 * m * n * k; }
static u64 flops_trsm(u64 m, u64 n) { return m * m * n; }
static u64 flops_syrk(u64 n, u64 k) { return n * n * k; }
static u64 flops_potrf(u64 n) { return n * n * n / 3; }

/* structural lower triangle of  P_ltri + I + AT*AT^T + GT*GT^T  (multistage_kkt.hpp:424-431; Eigen's
 * sparse product keeps every structural entry, explicit zeros included), column-compressed with
 * sorted row indices.  Column i of the lower triangle == row i of the upper triangle (:448-452). */
static void condensed_pattern(const orc_data *d, int **Cp_out, int **Ci_out)
{
    int n = d->n;
    const orc_csc *U = &d->sP_utri;
    const orc_csc *T[2] = { &d->sAT, &d->sGT };
    /* row-compressed views of AT and GT (= which constraints touch variable j) */
    int *rp[2], *ri[2];
    for (int t = 0; t < 2; t++) {
        int nc = T[t]->cols, nz = T[t]->colptr[nc];
        rp[t] = (int *)xcalloc((size_t)n + 1, sizeof(int));
        ri[t] = (int *)xcalloc((size_t)nz, sizeof(int));
        for (int q = 0; q < nz; q++) rp[t][T[t]->rowind[q] + 1]++;
        for (int j = 0; j < n; j++) rp[t][j + 1] += rp[t][j];
        int *fill = iclone(rp[t], n);
        for (int c = 0; c < nc; c++)
            for (int q = T[t]->colptr[c]; q < T[t]->colptr[c + 1]; q++) ri[t][fill[T[t]->rowind[q]]++] = c;
        free(fill);
    }
    /* P_ltri column j = row j of P_utri: entries (j, c) with c >= j */
    int *prp = (int *)xcalloc((size_t)n + 1, sizeof(int));
    int nzP = U->colptr[n];
    int *pri = (int *)xcalloc((size_t)nzP, sizeof(int));
    for (int q = 0; q < nzP; q++) prp[U->rowind[q] + 1]++;
    for (int j = 0; j < n; j++) prp[j + 1] += prp[j];
    {
        int *fill = iclone(prp, n);
        for (int c = 0; c < n; c++)
            for (int q = U->colptr[c]; q < U->colptr[c + 1]; q++) pri[fill[U->rowind[q]]++] = c;
        free(fill);
    }
    int *mark = (int *)xcalloc((size_t)n, sizeof(int));
    for (int i = 0; i < n; i++) mark[i] = -1;
    int *Cp = (int *)xcalloc((size_t)n + 1, sizeof(int));
    size_t cap = (size_t)nzP + (size_t)n + 16, nz = 0;
    int *Ci = (int *)xcalloc(cap, sizeof(int));
    int *colbuf = (int *)xcalloc((size_t)n, sizeof(int));
    for (int j = 0; j < n; j++) {
        int cnt = 0;
        mark[j] = j; colbuf[cnt++] = j;                               /* identity */
        for (int q = prp[j]; q < prp[j + 1]; q++) {                    /* P_ltri */
            int r = pri[q];
            if (r >= j && mark[r] != j) { mark[r] = j; colbuf[cnt++] = r; }
        }
        for (int t = 0; t < 2; t++)                                    /* lower(AT AT^T), lower(GT GT^T) */
            for (int q = rp[t][j]; q < rp[t][j + 1]; q++) {
                int c = ri[t][q];
                for (int qq = T[t]->colptr[c]; qq < T[t]->colptr[c + 1]; qq++) {
                    int r = T[t]->rowind[qq];
                    if (r >= j && mark[r] != j) { mark[r] = j; colbuf[cnt++] = r; }
                }
            }
        /* sort ascending (insertion sort: columns are short) */
        for (int a = 1; a < cnt; a++) { int v = colbuf[a], b = a - 1; while (b >= 0 && colbuf[b] > v) { colbuf[b + 1] = colbuf[b]; b--; } colbuf[b + 1] = v; }
        if (nz + (size_t)cnt > cap) { cap = (nz + (size_t)cnt) * 2; Ci = (int *)realloc(Ci, cap * sizeof(int)); }
        memcpy(Ci + nz, colbuf, sizeof(int) * (size_t)cnt);
        nz += (size_t)cnt; Cp[j + 1] = (int)nz;
    }
    free(colbuf); free(mark); free(prp); free(pri);
    for (int t = 0; t < 2; t++) { free(rp[t]); free(ri[t]); }
    *Cp_out = Cp; *Ci_out = Ci;
}

typedef struct { int prev_diag, start, diag, off, arrow; } bsi; /* multistage_kkt.hpp:433-439 */

/* the `get_next_block_structure` lambda, multistage_kkt.hpp:454-521 */
static bsi next_block_structure(const int *Cp, const int *Ci, int n, int row, bsi cur, u64 flops_tridiag,
                                u64 fa_no_syrk, u64 fa_syrk)
{
    bsi nx = cur;
    for (int q = Cp[row]; q < Cp[row + 1]; q++) {
        int col = Ci[q];
        if (col >= nx.start && col + nx.arrow < n) {
            int current_block_size = nx.diag + nx.off;
            int new_block_size = imax(col - nx.start + 1, current_block_size);
            int max_diag_block_size = row - nx.start + 1;
            int new_min_diag_block_size = imax(nx.diag, (new_block_size + 1) / 2);
            int new_diag = imax(new_min_diag_block_size, max_diag_block_size);
            int new_off = new_block_size - new_diag;
            int remaining_width = n - nx.start - nx.diag - nx.off;
            int new_arrow = imin(imax(nx.arrow, n - col), remaining_width);

            u64 flops_tridiag_new = flops_tridiag;
            flops_tridiag_new += flops_syrk((u64)new_diag, (u64)nx.prev_diag);
            flops_tridiag_new += flops_potrf((u64)new_diag);
            flops_tridiag_new += flops_trsm((u64)new_diag, (u64)new_off);

            int aw = ((nx.arrow + 3) / 4) * 4;
            int naw = ((new_arrow + 3) / 4) * 4;
            u64 flops_arrow = (u64)aw * fa_no_syrk + (u64)aw * (u64)aw * fa_syrk + flops_potrf((u64)aw);
            u64 flops_arrow_new = (u64)naw * fa_no_syrk + (u64)naw * (u64)naw * fa_syrk;
            flops_arrow_new += flops_gemm((u64)naw, (u64)nx.prev_diag, (u64)new_diag);
            flops_arrow_new += flops_trsm((u64)new_diag, (u64)naw);
            flops_arrow_new += flops_syrk((u64)naw, (u64)new_diag);
            flops_arrow_new += flops_potrf((u64)naw);

            if (flops_tridiag_new - flops_tridiag <= flops_arrow_new - flops_arrow) {
                nx.diag = new_diag;
                nx.off = new_off;
            } else {
                nx.arrow = new_arrow;
            }
        }
    }
    return nx;
}

/* multistage_kkt.hpp:420-597.  Returns the block list (last entry = arrow corner block). */
static blk_info *extract_arrow_structure(const orc_data *d, int *N_out)
{
    int n = d->n;
    int *Cp, *Ci;
    condensed_pattern(d, &Cp, &Ci);
    int cap = 16, nb = 0;
    blk_info *bi = (blk_info *)xcalloc((size_t)cap, sizeof(blk_info));
#define PUSH(S, D, O) do { if (nb + 2 > cap) { cap *= 2; bi = (blk_info *)realloc(bi, (size_t)cap * sizeof(blk_info)); } \
                           bi[nb].start = (S); bi[nb].diag = (D); bi[nb].off = (O); nb++; } while (0)
    bsi cur = { 0, 0, 0, 0, 0 };
    u64 flops_tridiag = 0, fa_no_syrk = 0, fa_syrk = 0;
    for (int i = 0; i < n; i++) {
        cur = next_block_structure(Cp, Ci, n, i, cur, flops_tridiag, fa_no_syrk, fa_syrk);
        if (i + 1 >= cur.start + cur.diag) {
            int hit_optimal_ratio = cur.diag >= 2 * cur.off;
            int at_end = i + 1 >= n - cur.arrow;
            int grows = 0;
            if (!hit_optimal_ratio && !at_end) { /* next_block_grows(), evaluated only when reached (:531-535) */
                bsi nb_info = next_block_structure(Cp, Ci, n, i + 1, cur, flops_tridiag, fa_no_syrk, fa_syrk);
                grows = nb_info.diag + nb_info.off > cur.diag + cur.off;
            }
            if (hit_optimal_ratio || at_end || grows) {
                PUSH(cur.start, cur.diag, cur.off);
                flops_tridiag += flops_syrk((u64)cur.diag, (u64)(cur.prev_diag + 1));
                flops_tridiag += flops_potrf((u64)cur.diag);
                flops_tridiag += flops_trsm((u64)cur.diag, (u64)cur.off);
                fa_no_syrk += flops_gemm(1, (u64)cur.prev_diag, (u64)cur.diag);
                fa_no_syrk += flops_trsm((u64)cur.diag, 1);
                fa_syrk += flops_syrk(1, (u64)cur.diag);
                cur.start += cur.diag;
                cur.prev_diag = cur.diag;
                cur.diag = cur.off;
                cur.off = 0;
            }
            if (at_end && cur.diag > 0) {
                PUSH(cur.start, cur.diag, cur.off);
                cur.start += cur.diag;
                cur.prev_diag = cur.diag;
                cur.diag = cur.off;
                cur.off = 0;
            }
            if (at_end) break;
        }
    }
    /* merge split blocks (:573-583); the index advances after an erase exactly as in the reference */
    for (int i = 0; i + 1 < nb; i++) {
        if (bi[i].off == bi[i + 1].diag && bi[i + 1].off == 0) {
            bi[i].diag += bi[i].off;
            bi[i].off = 0;
            memmove(&bi[i + 1], &bi[i + 2], sizeof(blk_info) * (size_t)(nb - i - 2));
            nb--;
        }
    }
    PUSH(cur.start, cur.arrow, 0);
#undef PUSH
    free(Cp); free(Ci);
    *N_out = nb;
    return bi;
}

/* multistage_kkt.hpp:599-670: scatter an upper-triangular CSC matrix into lower blocks */
static void utri_to_kkt(const ms_kkt *k, const orc_csc *U, block_kkt *K)
{
    const blk_info *bi = k->bi;
    int n = U->cols, arrow = k->arrow;
    int bidx = 0, bstart = bi[0].start, bdiag = bi[0].diag;
    for (int i = 0; i < n; i++) {
        if (i >= bstart + bdiag) { bidx++; bstart = bi[bidx].start; bdiag = bi[bidx].diag; }
        int ab = 0, abstart = bi[0].start, abw = bi[0].diag;
        for (int q = U->colptr[i]; q < U->colptr[i + 1]; q++) {
            int j = U->rowind[q];
            double v = U->val[q];
            if (j >= bstart) {
                dmat_alloc(&K->D[bidx], bdiag, bdiag);
                EL(K->D[bidx], i - bstart, j - bstart) = v;
            } else if (i >= n - arrow) {
                while (abstart + abw - 1 < j) { ab++; abstart = bi[ab].start; abw = bi[ab].diag; }
                dmat_alloc(&K->E[ab], arrow, abw);
                EL(K->E[ab], i - bstart, j - abstart) = v;
            } else {
                int ls = bi[bidx - 1].start, ld = bi[bidx - 1].diag, lo = bi[bidx - 1].off;
                dmat_alloc(&K->B[bidx - 1], lo, ld);
                EL(K->B[bidx - 1], i - bstart, j - ls) = v;
            }
        }
    }
}

/* multistage_kkt.hpp:672-818 with store_transpose = true.  sAT is n x rows CSC: column i = constraint row i */
static void transpose_to_block_mat(const ms_kkt *k, const orc_csc *sAT, int init, block_mat *A)
{
    const blk_info *bi = k->bi;
    int N = k->N, arrow = k->arrow;
    int rows = sAT->cols, cols = sAT->rows;
    if (init) {
        A->N = N; A->rows_total = rows;
        A->perm = (int *)xcalloc((size_t)rows, sizeof(int));
        A->perm_inv = (int *)xcalloc((size_t)rows, sizeof(int));
        A->row_sizes = (int *)xcalloc((size_t)(N - 1), sizeof(int));
        A->row_acc = (int *)xcalloc((size_t)N, sizeof(int));
        A->D = (dmat *)xcalloc((size_t)(N - 1), sizeof(dmat));
        A->B = (dmat *)xcalloc((size_t)imax(N - 2, 0), sizeof(dmat));
        A->E = (dmat *)xcalloc((size_t)(N - 1), sizeof(dmat));
        for (int i = 0; i < rows; i++) {
            if (sAT->colptr[i] < sAT->colptr[i + 1]) {
                int j = sAT->rowind[sAT->colptr[i]];
                int b = 0;
                while (bi[b].start + bi[b].diag <= j && b + 1 < N - 1) b++;
                A->row_sizes[b]++;
            }
        }
        A->row_acc[0] = 0;
        for (int i = 0; i < N - 1; i++) A->row_acc[i + 1] = A->row_acc[i] + A->row_sizes[i];
        A->blocked_rows = A->row_acc[N - 1];
    }
    int *fill = (int *)xcalloc((size_t)(N - 1), sizeof(int));
    int no_block_counter = 0;
    for (int i = 0; i < rows; i++) {
        int q0 = sAT->colptr[i], q1 = sAT->colptr[i + 1];
        int b = 0, block_i = 0;
        if (q0 < q1) {
            int j = sAT->rowind[q0];
            while (bi[b].start + bi[b].diag <= j && b + 1 < N - 1) b++;
            block_i = fill[b]++;
            if (init) A->perm[i] = A->row_acc[b] + block_i;
        } else if (init) {
            A->perm[i] = A->row_acc[N - 1] + no_block_counter++; /* empty rows go to the back (:738-741) */
        }
        int bstart = bi[b].start, bdiag = bi[b].diag;
        for (int q = q0; q < q1; q++) {
            int j = sAT->rowind[q];
            double v = sAT->val[q];
            if (j + arrow >= cols) {
                if (init) dmat_alloc(&A->E[b], arrow, A->row_sizes[b]);
                EL(A->E[b], j + arrow - cols, block_i) = v;
            } else if (j < bstart + bdiag) {
                if (init) dmat_alloc(&A->D[b], bdiag, A->row_sizes[b]);
                EL(A->D[b], j - bstart, block_i) = v;
            } else {
                int boff = bi[b].off;
                if (j >= bstart + bdiag + boff) { fprintf(stderr, "orc_multistage: index in no valid block\n"); abort(); }
                if (init) dmat_alloc(&A->B[b], boff, A->row_sizes[b]);
                EL(A->B[b], j - bstart - bdiag, block_i) = v;
            }
        }
    }
    free(fill);
    if (init) for (int i = 0; i < rows; i++) A->perm_inv[A->perm[i]] = i;
}

/* ---- BLAS-like helpers on column-major blocks (semantics: utils/blasfeo_wrapper.hpp) ---- */
/* C[0:m,0:m] (lower) += A * B^T, A, B m x k */
static void syrk_ln_acc(const dmat *A, const dmat *B, dmat *C)
{
    int m = A->r, kk = A->c;
    for (int j = 0; j < m; j++)
        for (int i = j; i < m; i++) {
            double s = 0.0;
            for (int l = 0; l < kk; l++) s += EL(*A, i, l) * EL(*B, j, l);
            EL(*C, i, j) += s;
        }
}
/* C[0:m,0:n] = beta*C + alpha * A * B^T, A m x k, B n x k */
static void gemm_nt(double alpha, const dmat *A, const dmat *B, double beta, dmat *C)
{
    int m = A->r, nn = B->r, kk = A->c;
    for (int j = 0; j < nn; j++)
        for (int i = 0; i < m; i++) {
            double s = 0.0;
            for (int l = 0; l < kk; l++) s += EL(*A, i, l) * EL(*B, j, l);
            EL(*C, i, j) = (beta == 0.0 ? 0.0 : beta * EL(*C, i, j)) + alpha * s;
        }
}
/* lower Cholesky in place on the leading m x m block; non-positive pivot -> zero column */
static void potrf_l(dmat *A, int m)
{
    for (int j = 0; j < m; j++) {
        double c = EL(*A, j, j);
        for (int l = 0; l < j; l++) c -= EL(*A, j, l) * EL(*A, j, l);
        double inv = c > 0.0 ? 1.0 / sqrt(c) : 0.0;
        EL(*A, j, j) = c * inv;
        for (int i = j + 1; i < m; i++) {
            double s = EL(*A, i, j);
            for (int l = 0; l < j; l++) s -= EL(*A, i, l) * EL(*A, j, l);
            EL(*A, i, j) = s * inv;
        }
    }
}
/* X[0:m,0:n] = X * L^{-T}, L n x n lower (blasfeo_dtrsm_rltn) */
static void trsm_rltn(const dmat *L, int n, dmat *X, int m)
{
    for (int j = 0; j < n; j++) {
        double ljj = EL(*L, j, j);
        double inv = ljj != 0.0 ? 1.0 / ljj : 0.0;
        for (int i = 0; i < m; i++) {
            double s = EL(*X, i, j);
            for (int l = 0; l < j; l++) s -= EL(*X, i, l) * EL(*L, j, l);
            EL(*X, i, j) = s * inv;
        }
    }
}
static void trsv_lnn(const dmat *L, int m, double *x)
{
    for (int i = 0; i < m; i++) {
        double s = x[i];
        for (int l = 0; l < i; l++) s -= EL(*L, i, l) * x[l];
        double lii = EL(*L, i, i);
        x[i] = lii != 0.0 ? s / lii : 0.0;
    }
}
static void trsv_ltn(const dmat *L, int m, double *x)
{
    for (int i = m - 1; i >= 0; i--) {
        double s = x[i];
        for (int l = i + 1; l < m; l++) s -= EL(*L, l, i) * x[l];
        double lii = EL(*L, i, i);
        x[i] = lii != 0.0 ? s / lii : 0.0;
    }
}
/* z[0:m] += alpha * A * x */
static void gemv_n_acc(double alpha, const dmat *A, const double *x, double *z)
{
    for (int j = 0; j < A->c; j++) {
        double xj = alpha * x[j];
        for (int i = 0; i < A->r; i++) z[i] += EL(*A, i, j) * xj;
    }
}
/* z[0:n] += alpha * A^T * x */
static void gemv_t_acc(double alpha, const dmat *A, const double *x, double *z)
{
    for (int j = 0; j < A->c; j++) {
        double s = 0.0;
        for (int i = 0; i < A->r; i++) s += EL(*A, i, j) * x[i];
        z[j] += alpha * s;
    }
}

/* multistage_kkt.hpp:832-994: sD = lower block structure of sA * sB^T; calc != 0 accumulates values,
 * calc == 0 only allocates (blocks allocated at full size, see header) */
static void block_syrk_ln(const ms_kkt *k, const block_mat *sA, const block_mat *sB, block_kkt *sD, int calc)
{
    int N = k->N, arrow = k->arrow;
    const blk_info *bi = k->bi;
    for (int i = 0; i < N - 1; i++) {
        if (calc) dmat_zero(&sD->D[i]);
        if (sA->D[i].a && sB->D[i].a) {
            if (!calc) dmat_alloc(&sD->D[i], bi[i].diag, bi[i].diag);
            else syrk_ln_acc(&sA->D[i], &sB->D[i], &sD->D[i]);
        }
        if (i > 0 && sA->B[i - 1].a && sB->B[i - 1].a) {
            if (!calc) dmat_alloc(&sD->D[i], bi[i].diag, bi[i].diag);
            else syrk_ln_acc(&sA->B[i - 1], &sB->B[i - 1], &sD->D[i]);
        }
    }
    if (arrow > 0) {
        if (calc) dmat_zero(&sD->D[N - 1]);
        for (int i = 0; i < N - 1; i++)
            if (sA->E[i].a && sB->E[i].a) {
                if (!calc) dmat_alloc(&sD->D[N - 1], arrow, arrow);
                else syrk_ln_acc(&sA->E[i], &sB->E[i], &sD->D[N - 1]);
            }
    }
    for (int i = 0; i < N - 2; i++)
        if (sA->B[i].a && sB->D[i].a) {
            if (!calc) dmat_alloc(&sD->B[i], bi[i].off, bi[i].diag);
            else gemm_nt(1.0, &sA->B[i], &sB->D[i], 0.0, &sD->B[i]);
        }
    if (arrow > 0)
        for (int i = 0; i < N - 1; i++) {
            if (calc) dmat_zero(&sD->E[i]);
            if (sA->E[i].a && sB->D[i].a) {
                if (!calc) dmat_alloc(&sD->E[i], arrow, bi[i].diag);
                else gemm_nt(1.0, &sA->E[i], &sB->D[i], 1.0, &sD->E[i]);
            }
            if (i > 0 && sA->E[i - 1].a && sB->B[i - 1].a) {
                if (!calc) dmat_alloc(&sD->E[i], arrow, bi[i].diag);
                else gemm_nt(1.0, &sA->E[i - 1], &sB->B[i - 1], 1.0, &sD->E[i]);
            }
        }
}

/* B[0:r,0:c] += alpha * A (blasfeo_dgead), lower != 0 restricts to the lower triangle */
static void gead(double alpha, const dmat *A, dmat *B)
{
    for (int j = 0; j < A->c; j++)
        for (int i = 0; i < A->r; i++) EL(*B, i, j) += alpha * EL(*A, i, j);
}

/* multistage_kkt.hpp:1008-1219: kkt_fac = P + delta^-1 AtA + GtG + diag(x_reg) blockwise */
static void construct_kkt_fac(ms_kkt *k, const double *x_reg, int allocate)
{
    int N = k->N, arrow = k->arrow;
    const blk_info *bi = k->bi;
    double delta_inv = 1.0 / k->m_delta;
    block_kkt *F = &k->fac;
    for (int i = 0; i < N; i++) {
        int mm = bi[i].diag;
        if (allocate) { dmat_alloc(&F->D[i], mm, mm); continue; }
        dmat_zero(&F->D[i]);
        if (k->P.D[i].a) gead(1.0, &k->P.D[i], &F->D[i]);
        if (k->AtA.D[i].a) gead(delta_inv, &k->AtA.D[i], &F->D[i]);
        if (k->GtG.D[i].a) gead(1.0, &k->GtG.D[i], &F->D[i]);
        for (int j = 0; j < mm; j++) EL(F->D[i], j, j) += x_reg[bi[i].start + j];
    }
    for (int i = 0; i < N - 2; i++) {
        int any = k->P.B[i].a || k->AtA.B[i].a || k->GtG.B[i].a;
        if (allocate) { if (any) dmat_alloc(&F->B[i], bi[i].off, bi[i].diag); continue; }
        if (!any) continue;
        dmat_zero(&F->B[i]);
        if (k->P.B[i].a) gead(1.0, &k->P.B[i], &F->B[i]);
        if (k->AtA.B[i].a) gead(delta_inv, &k->AtA.B[i], &F->B[i]);
        if (k->GtG.B[i].a) gead(1.0, &k->GtG.B[i], &F->B[i]);
    }
    if (arrow > 0)
        for (int i = 0; i < N - 1; i++) {
            int any = k->P.E[i].a || k->AtA.E[i].a || k->GtG.E[i].a;
            if (allocate) {
                /* fill-in of the arrow through the factorisation (:1192-1198) */
                if (any || (i > 0 && F->E[i - 1].a && F->B[i - 1].a)) dmat_alloc(&F->E[i], arrow, bi[i].diag);
                continue;
            }
            if (!F->E[i].a) continue;
            dmat_zero(&F->E[i]);
            if (k->P.E[i].a) gead(1.0, &k->P.E[i], &F->E[i]);
            if (k->AtA.E[i].a) gead(delta_inv, &k->AtA.E[i], &F->E[i]);
            if (k->GtG.E[i].a) gead(1.0, &k->GtG.E[i], &F->E[i]);
        }
}

/* multistage_kkt.hpp:1253-1352 */
static void factor_kkt(ms_kkt *k)
{
    int N = k->N, arrow = k->arrow;
    block_kkt *F = &k->fac;
    potrf_l(&F->D[0], F->D[0].r);
    if (N > 2 && F->B[0].a) trsm_rltn(&F->D[0], F->B[0].c, &F->B[0], F->B[0].r);
    if (arrow > 0 && F->E[0].a) {
        trsm_rltn(&F->D[0], F->E[0].c, &F->E[0], F->E[0].r);
        for (int j = 0; j < arrow; j++)
            for (int i = j; i < arrow; i++) {
                double s = 0.0;
                for (int l = 0; l < F->E[0].c; l++) s += EL(F->E[0], i, l) * EL(F->E[0], j, l);
                EL(F->D[N - 1], i, j) -= s;
            }
    }
    for (int i = 1; i < N - 1; i++) {
        if (F->B[i - 1].a) {
            const dmat *C = &F->B[i - 1];
            for (int j = 0; j < C->r; j++)
                for (int r = j; r < C->r; r++) {
                    double s = 0.0;
                    for (int l = 0; l < C->c; l++) s += EL(*C, r, l) * EL(*C, j, l);
                    EL(F->D[i], r, j) -= s;
                }
        }
        potrf_l(&F->D[i], F->D[i].r);
        if (i < N - 2 && F->B[i].a) trsm_rltn(&F->D[i], F->B[i].c, &F->B[i], F->B[i].r);
        if (arrow > 0) {
            if (F->E[i].a && F->E[i - 1].a && F->B[i - 1].a) {
                /* F_i = (E_i - F_{i-1} C_{i-1}^T) L_i^{-T} */
                const dmat *Ep = &F->E[i - 1], *Cp = &F->B[i - 1];
                for (int j = 0; j < Cp->r; j++)
                    for (int r = 0; r < arrow; r++) {
                        double s = 0.0;
                        for (int l = 0; l < Ep->c; l++) s += EL(*Ep, r, l) * EL(*Cp, j, l);
                        EL(F->E[i], r, j) -= s;
                    }
                trsm_rltn(&F->D[i], F->D[i].r, &F->E[i], arrow);
            } else if (F->E[i].a) {
                trsm_rltn(&F->D[i], F->E[i].c, &F->E[i], F->E[i].r);
            }
            if (F->E[i].a)
                for (int j = 0; j < arrow; j++)
                    for (int r = j; r < arrow; r++) {
                        double s = 0.0;
                        for (int l = 0; l < F->E[i].c; l++) s += EL(F->E[i], r, l) * EL(F->E[i], j, l);
                        EL(F->D[N - 1], r, j) -= s;
                    }
        }
    }
    potrf_l(&F->D[N - 1], arrow);
}

/* multistage_kkt.hpp:1709-1816; x is a BlockVec over block_info (contiguous, offsets = bi[].start) */
static void solve_llt_in_place(const ms_kkt *k, double *x)
{
    int N = k->N, arrow = k->arrow;
    const blk_info *bi = k->bi;
    const block_kkt *F = &k->fac;
    trsv_lnn(&F->D[0], F->D[0].r, x + bi[0].start);
    for (int i = 1; i < N - 1; i++) {
        if (F->B[i - 1].a) gemv_n_acc(-1.0, &F->B[i - 1], x + bi[i - 1].start, x + bi[i].start);
        trsv_lnn(&F->D[i], F->D[i].r, x + bi[i].start);
    }
    if (arrow > 0) {
        for (int i = 0; i < N - 1; i++)
            if (F->E[i].a) gemv_n_acc(-1.0, &F->E[i], x + bi[i].start, x + bi[N - 1].start);
        trsv_lnn(&F->D[N - 1], arrow, x + bi[N - 1].start);
    }
    if (arrow > 0) {
        trsv_ltn(&F->D[N - 1], arrow, x + bi[N - 1].start);
        if (F->E[N - 2].a) gemv_t_acc(-1.0, &F->E[N - 2], x + bi[N - 1].start, x + bi[N - 2].start);
    }
    trsv_ltn(&F->D[N - 2], F->D[N - 2].r, x + bi[N - 2].start);
    for (int i = N - 3; i >= 0; i--) {
        if (F->B[i].a) gemv_t_acc(-1.0, &F->B[i], x + bi[i + 1].start, x + bi[i].start);
        if (F->E[i].a) gemv_t_acc(-1.0, &F->E[i], x + bi[N - 1].start, x + bi[i].start);
        trsv_ltn(&F->D[i], F->D[i].r, x + bi[i].start);
    }
}

/* z = beta*z + alpha * A_block * x with A_block the TRANSPOSED block matrix (n x rows), :1520-1577.
 * x in permuted-constraint block order, z over block_info. */
static void block_t_gemv_n(const ms_kkt *k, double alpha, const block_mat *A, const double *x, double beta, double *z)
{
    int N = k->N, arrow = k->arrow, n = k->n;
    const blk_info *bi = k->bi;
    for (int j = 0; j < n; j++) z[j] = beta == 0.0 ? 0.0 : beta * z[j];
    for (int i = 0; i < N - 1; i++) {
        if (A->D[i].a) gemv_n_acc(alpha, &A->D[i], x + A->row_acc[i], z + bi[i].start);
        if (i > 0 && A->B[i - 1].a) gemv_n_acc(alpha, &A->B[i - 1], x + A->row_acc[i - 1], z + bi[i].start);
    }
    if (arrow > 0)
        for (int i = 0; i < N - 1; i++)
            if (A->E[i].a) gemv_n_acc(alpha, &A->E[i], x + A->row_acc[i], z + bi[N - 1].start);
}
/* z = alpha * A_block^T * x, :1589-1640; x over block_info, z in permuted-constraint block order */
static void block_t_gemv_t(const ms_kkt *k, double alpha, const block_mat *A, const double *x, double *z)
{
    int N = k->N, arrow = k->arrow;
    const blk_info *bi = k->bi;
    for (int j = 0; j < A->blocked_rows; j++) z[j] = 0.0;
    for (int i = 0; i < N - 1; i++) {
        if (A->D[i].a) gemv_t_acc(alpha, &A->D[i], x + bi[i].start, z + A->row_acc[i]);
        if (i < N - 2 && A->B[i].a) gemv_t_acc(alpha, &A->B[i], x + bi[i + 1].start, z + A->row_acc[i]);
        if (arrow > 0 && A->E[i].a) gemv_t_acc(alpha, &A->E[i], x + bi[N - 1].start, z + A->row_acc[i]);
    }
}
/* BlockVec::assign / load with perm_inv (blocksparse/block_vec.hpp:97-147) */
static void bv_assign_perm(const block_mat *A, const double *src, double *dst)
{
    for (int i = 0; i < A->blocked_rows; i++) dst[i] = src[A->perm_inv[i]];
}
static void bv_load_perm(const block_mat *A, const double *src, double *dst)
{
    for (int i = 0; i < A->blocked_rows; i++) dst[A->perm_inv[i]] = src[i];
    for (int i = A->blocked_rows; i < A->rows_total; i++) dst[A->perm_inv[i]] = 0.0;
}

/* ---- KKTSolverBase implementation ------------------------------------------------------------- */
static void ms_update_data(orc_kkt *self, const orc_data *d, int options) /* :142-178 */
{
    ms_kkt *k = (ms_kkt *)self;
    if (options & ORC_KKT_UPDATE_P) utri_to_kkt(k, &d->sP_utri, &k->P);
    if (options & ORC_KKT_UPDATE_A) {
        transpose_to_block_mat(k, &d->sAT, 0, &k->AT);
        block_syrk_ln(k, &k->AT, &k->AT, &k->AtA, 1);
    }
    if (options & ORC_KKT_UPDATE_G) transpose_to_block_mat(k, &d->sGT, 0, &k->GT);
}

static int ms_factor(orc_kkt *self, const orc_data *d, double delta, const double *x_reg, const double *z_reg) /* :180-219 */
{
    ms_kkt *k = (ms_kkt *)self;
    (void)d;
    k->m_delta = delta;
    for (int i = 0; i < k->m; i++) k->z_reg_inv[i] = 1.0 / z_reg[i];
    for (int i = 0; i < k->GT.blocked_rows; i++) k->G_scaling[i] = sqrt(k->z_reg_inv[k->GT.perm_inv[i]]);
    /* block_gemm_nd (:1222-1251): GT_scaled = GT * diag(G_scaling) */
    for (int i = 0; i < k->N - 1; i++) {
        const double *sc = k->G_scaling + k->GT.row_acc[i];
        dmat *src[3] = { &k->GT.D[i], i < k->N - 2 ? &k->GT.B[i] : NULL, &k->GT.E[i] };
        dmat *dst[3] = { &k->GTs.D[i], i < k->N - 2 ? &k->GTs.B[i] : NULL, &k->GTs.E[i] };
        for (int t = 0; t < 3; t++)
            if (src[t] && src[t]->a)
                for (int j = 0; j < src[t]->c; j++)
                    for (int r = 0; r < src[t]->r; r++) EL(*dst[t], r, j) = EL(*src[t], r, j) * sc[j];
    }
    block_syrk_ln(k, &k->GTs, &k->GTs, &k->GtG, 1);
    construct_kkt_fac(k, x_reg, 0);
    factor_kkt(k);
    return 1;
}

static void ms_solve(orc_kkt *self, const orc_data *d, const double *rhs_x, const double *rhs_y, const double *rhs_z,
                     double *lhs_x, double *lhs_y, double *lhs_z) /* :221-288 */
{
    ms_kkt *k = (ms_kkt *)self;
    (void)d;
    double delta_inv = 1.0 / k->m_delta;
    for (int i = 0; i < k->m; i++) k->work_z[i] = k->z_reg_inv[i] * rhs_z[i];
    memcpy(k->bx1, rhs_x, sizeof(double) * (size_t)k->n);
    bv_assign_perm(&k->AT, rhs_y, k->by1);
    bv_assign_perm(&k->GT, k->work_z, k->bz1);
    block_t_gemv_n(k, 1.0, &k->GT, k->bz1, 1.0, k->bx1);
    block_t_gemv_n(k, delta_inv, &k->AT, k->by1, 1.0, k->bx1);
    solve_llt_in_place(k, k->bx1);
    block_t_gemv_t(k, delta_inv, &k->AT, k->bx1, k->by1);
    block_t_gemv_t(k, 1.0, &k->GT, k->bx1, k->bz1);
    memcpy(lhs_x, k->bx1, sizeof(double) * (size_t)k->n);
    bv_load_perm(&k->AT, k->by1, lhs_y);
    bv_load_perm(&k->GT, k->bz1, lhs_z);
    for (int i = 0; i < k->p; i++) lhs_y[i] -= delta_inv * rhs_y[i];
    for (int i = 0; i < k->m; i++) lhs_z[i] = (lhs_z[i] - rhs_z[i]) * k->z_reg_inv[i];
}

static void ms_eval_P_x(orc_kkt *self, const orc_data *d, double alpha, const double *x, double *z) /* :291-316 + block_symv_l :1355-1404 */
{
    ms_kkt *k = (ms_kkt *)self;
    (void)d;
    int N = k->N, arrow = k->arrow;
    const blk_info *bi = k->bi;
    const block_kkt *P = &k->P;
    for (int i = 0; i < k->n; i++) z[i] = 0.0;
    for (int i = 0; i < N; i++)
        if (P->D[i].a) {
            const dmat *D = &P->D[i];
            const double *xi = x + bi[i].start;
            double *zi = z + bi[i].start;
            for (int c = 0; c < D->c; c++) {
                zi[c] += alpha * EL(*D, c, c) * xi[c];
                for (int r = c + 1; r < D->r; r++) {
                    double v = alpha * EL(*D, r, c);
                    zi[r] += v * xi[c];
                    zi[c] += v * xi[r];
                }
            }
        }
    for (int i = 0; i < N - 2; i++)
        if (P->B[i].a) {
            gemv_n_acc(alpha, &P->B[i], x + bi[i].start, z + bi[i + 1].start);
            gemv_t_acc(alpha, &P->B[i], x + bi[i + 1].start, z + bi[i].start);
        }
    if (arrow > 0)
        for (int i = 0; i < N - 1; i++)
            if (P->E[i].a) {
                gemv_n_acc(alpha, &P->E[i], x + bi[i].start, z + bi[N - 1].start);
                gemv_t_acc(alpha, &P->E[i], x + bi[N - 1].start, z + bi[i].start);
            }
}

static void ms_eval_AG(ms_kkt *k, const block_mat *A, double *bn, double *bt, double an, double at, const double *xn,
                       const double *xt, double *zn, double *zt)
{
    /* zt = alpha_t * AT * xt ; zn = alpha_n * A * xn (:318-383) */
    bv_assign_perm(A, xt, bt);
    block_t_gemv_n(k, at, A, bt, 0.0, k->bx2);
    block_t_gemv_t(k, an, A, xn, bn);
    bv_load_perm(A, bn, zn);
    memcpy(zt, k->bx2, sizeof(double) * (size_t)k->n);
}
static void ms_eval_A(orc_kkt *self, const orc_data *d, double an, double at, const double *xn, const double *xt, double *zn, double *zt)
{
    ms_kkt *k = (ms_kkt *)self; (void)d;
    ms_eval_AG(k, &k->AT, k->by2, k->by1, an, at, xn, xt, zn, zt);
}
static void ms_eval_G(orc_kkt *self, const orc_data *d, double an, double at, const double *xn, const double *xt, double *zn, double *zt)
{
    ms_kkt *k = (ms_kkt *)self; (void)d;
    ms_eval_AG(k, &k->GT, k->bz2, k->bz1, an, at, xn, xt, zn, zt);
}

static void ms_print_info(orc_kkt *self) /* :385-393 */
{
    ms_kkt *k = (ms_kkt *)self;
    printf("block sizes:");
    for (int i = 0; i < k->N - 1; i++) printf(" %d,%d", k->bi[i].diag, k->bi[i].off);
    printf("\narrow width: %d\n", k->bi[k->N - 1].diag);
}

static void ms_destroy(orc_kkt *self)
{
    ms_kkt *k = (ms_kkt *)self;
    bk_free(&k->P); bk_free(&k->AtA); bk_free(&k->GtG); bk_free(&k->fac);
    bm_free(&k->AT); bm_free(&k->GT); bm_free(&k->GTs);
    free(k->bi); free(k->z_reg_inv); free(k->work_z); free(k->G_scaling);
    free(k->bx1); free(k->bx2); free(k->by1); free(k->by2); free(k->bz1); free(k->bz2);
    free(k);
}

static orc_kkt *ms_clone(const orc_kkt *self);
static void ms_fill_vtable(ms_kkt *k)
{
    k->base.clone = ms_clone;
    k->base.update_data = ms_update_data;
    k->base.update_scalings_and_factor = ms_factor;
    k->base.solve = ms_solve;
    k->base.eval_P_x = ms_eval_P_x;
    k->base.eval_A_xn_and_AT_xt = ms_eval_A;
    k->base.eval_G_xn_and_GT_xt = ms_eval_G;
    k->base.print_info = ms_print_info;
    k->base.destroy = ms_destroy;
}
static double *dclone(const double *s, int n) { double *p = (double *)xcalloc((size_t)n, sizeof(double)); if (n) memcpy(p, s, sizeof(double) * (size_t)n); return p; }
static void ms_alloc_work(ms_kkt *k)
{
    k->bx1 = (double *)xcalloc((size_t)k->n, sizeof(double)); k->bx2 = (double *)xcalloc((size_t)k->n, sizeof(double));
    k->by1 = (double *)xcalloc((size_t)k->p, sizeof(double)); k->by2 = (double *)xcalloc((size_t)k->p, sizeof(double));
    k->bz1 = (double *)xcalloc((size_t)k->m, sizeof(double)); k->bz2 = (double *)xcalloc((size_t)k->m, sizeof(double));
}
static orc_kkt *ms_clone(const orc_kkt *self)
{
    const ms_kkt *o = (const ms_kkt *)self;
    ms_kkt *k = (ms_kkt *)xcalloc(1, sizeof(ms_kkt));
    *k = *o;
    k->bi = (blk_info *)xcalloc((size_t)o->N, sizeof(blk_info));
    memcpy(k->bi, o->bi, sizeof(blk_info) * (size_t)o->N);
    k->z_reg_inv = dclone(o->z_reg_inv, o->m); k->work_z = dclone(o->work_z, o->m);
    k->G_scaling = dclone(o->G_scaling, o->m);
    k->P = bk_clone(&o->P); k->AtA = bk_clone(&o->AtA); k->GtG = bk_clone(&o->GtG); k->fac = bk_clone(&o->fac);
    k->AT = bm_clone(&o->AT); k->GT = bm_clone(&o->GT); k->GTs = bm_clone(&o->GTs);
    ms_alloc_work(k);
    return &k->base;
}

/* MultistageKKT ctor, multistage_kkt.hpp:76-135 */
orc_kkt *orc_multistage_kkt_create(const orc_data *d)
{
    if (!d->is_sparse) return NULL;
    ms_kkt *k = (ms_kkt *)xcalloc(1, sizeof(ms_kkt));
    ms_fill_vtable(k);
    k->n = d->n; k->p = d->p; k->m = d->m;
    k->m_delta = 1.0;
    k->z_reg_inv = (double *)xcalloc((size_t)d->m, sizeof(double));
    k->work_z = (double *)xcalloc((size_t)d->m, sizeof(double));
    k->G_scaling = (double *)xcalloc((size_t)d->m, sizeof(double));
    k->bi = extract_arrow_structure(d, &k->N);
    k->arrow = k->bi[k->N - 1].diag;
    int N = k->N;
    {
        int acc = 0;
        for (int i = 0; i < N; i++) { if (k->bi[i].start != acc) { fprintf(stderr, "orc_multistage: non-contiguous blocks\n"); abort(); } acc += k->bi[i].diag; }
        if (acc != d->n) { fprintf(stderr, "orc_multistage: blocks do not cover n\n"); abort(); }
    }
    bk_init(&k->P, N); bk_init(&k->AtA, N); bk_init(&k->GtG, N); bk_init(&k->fac, N);
    utri_to_kkt(k, &d->sP_utri, &k->P);
    transpose_to_block_mat(k, &d->sAT, 1, &k->AT);
    transpose_to_block_mat(k, &d->sGT, 1, &k->GT);
    k->GTs = bm_clone(&k->GT);
    block_syrk_ln(k, &k->AT, &k->AT, &k->AtA, 0);
    block_syrk_ln(k, &k->AT, &k->AT, &k->AtA, 1);
    block_syrk_ln(k, &k->GT, &k->GTs, &k->GtG, 0);
    construct_kkt_fac(k, NULL, 1);
    ms_alloc_work(k);
    return &k->base;
}

/* ---- test hooks -------------------------------------------------------------------------------- */
int orc_multistage_num_blocks(const orc_kkt *self) { return ((const ms_kkt *)self)->N; }
/* out: N rows of (start, diag_size, off_diag_size) */
void orc_multistage_block_info(const orc_kkt *self, int *out)
{
    const ms_kkt *k = (const ms_kkt *)self;
    for (int i = 0; i < k->N; i++) { out[3 * i] = k->bi[i].start; out[3 * i + 1] = k->bi[i].diag; out[3 * i + 2] = k->bi[i].off; }
}
/* which = 0 (AT) / 1 (GT): perm[rows], row_sizes[N-1] */
void orc_multistage_row_perm(const orc_kkt *self, int which, int *perm, int *row_sizes)
{
    const ms_kkt *k = (const ms_kkt *)self;
    const block_mat *A = which ? &k->GT : &k->AT;
    memcpy(perm, A->perm, sizeof(int) * (size_t)A->rows_total);
    memcpy(row_sizes, A->row_sizes, sizeof(int) * (size_t)(k->N - 1));
}
