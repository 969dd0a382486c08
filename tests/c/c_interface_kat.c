/* tests/c/c_interface_kat.c -- a plain C client of the reference's C interface, built against include/piqp_c_compat.h and
 * libpiqp_amd.so.  Known answers: the two-variable QP and its update from the reference's own C-interface test
 * (interfaces/c/tests/src/c_interface_test.cpp: SimpleDenseQPWithUpdate, SimpleSparseQPWithUpdate):
 *   min 3 x1^2 + 2 x2^2 - x1 - 4 x2  s.t. -1 <= x <= 1, x1 = 2 x2      ->  x = (0.4285714, 0.2142857), y = -1.5714286
 *   min 4 x1^2 + 2 x2^2 - x1 - 4 x2  s.t. -1 <= x <= 2, x1 = 3 x2      ->  x = (0.2763157, 0.0921056), y = -1.2105263
 * plus every sparse kkt_solver of the enum on the same problem.  Exit code 0 = all checks passed. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "piqp_c_compat.h"

static int failures = 0;
#define NEAR(a, b) do { if (!(fabs((a) - (b)) <= 1e-6)) { printf("FAIL %s:%d %s = %.9g, expected %.9g\n", __FILE__, __LINE__, #a, (double)(a), (double)(b)); ++failures; } } while (0)
#define TRUE_(c) do { if (!(c)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++failures; } } while (0)

static void check_first(const piqp_result* r)
{
    NEAR(r->x[0], 0.4285714); NEAR(r->x[1], 0.2142857); NEAR(r->y[0], -1.5714286);
    for (int i = 0; i < 3; ++i) { NEAR(r->z_l[i], 0); NEAR(r->z_u[i], 0); }
    for (int i = 0; i < 2; ++i) { NEAR(r->z_bl[i], 0); NEAR(r->z_bu[i], 0); }
}
static void check_second(const piqp_result* r)
{
    NEAR(r->x[0], 0.2763157); NEAR(r->x[1], 0.0921056); NEAR(r->y[0], -1.2105263);
    for (int i = 0; i < 3; ++i) { NEAR(r->z_l[i], 0); NEAR(r->z_u[i], 0); }
    for (int i = 0; i < 2; ++i) { NEAR(r->z_bl[i], 0); NEAR(r->z_bu[i], 0); }
}

static void dense_case(void)
{
    piqp_float P[4] = {6, 0, 0, 4}, c[2] = {-1, -4}, A[2] = {1, -2}, b[1] = {0}, G[6] = {1, 0, 1, 0, 1, 0};
    piqp_float h_l[3] = {-1, -PIQP_INF, -2}, h_u[3] = {PIQP_INF, 1, 2}, x_l[2] = {-PIQP_INF, -1}, x_u[2] = {PIQP_INF, 1};
    piqp_settings settings;
    piqp_data_dense data = {2, 1, 3, P, c, A, b, G, h_l, h_u, x_l, x_u};
    piqp_workspace* work = NULL;
    piqp_set_default_settings_dense(&settings);
    TRUE_(settings.kkt_solver == PIQP_DENSE_CHOLESKY && settings.max_iter == 250 && settings.eps_abs == 1e-8);
    piqp_setup_dense(&work, &data, &settings);
    TRUE_(work != NULL);
    if (!work) return;
    TRUE_(work->solver_info.is_dense == 1 && work->solver_info.n == 2 && work->solver_info.p == 1 && work->solver_info.m == 3);
    TRUE_(piqp_solve(work) == PIQP_SOLVED);
    TRUE_(work->result->info.status == PIQP_SOLVED && work->result->info.iter > 0 && work->result->info.iter < 30);
    check_first(work->result);
    P[0] = 8; A[1] = -3; h_u[0] = 2; x_u[1] = 2;
    piqp_update_dense(work, P, NULL, A, NULL, NULL, NULL, h_u, NULL, x_u);
    TRUE_(piqp_solve(work) == PIQP_SOLVED);
    check_second(work->result);
    /* settings can change between solves */
    settings.max_iter = 1;
    piqp_update_settings(work, &settings);
    TRUE_(piqp_solve(work) == PIQP_MAX_ITER_REACHED);
    piqp_cleanup(work);
}

static void sparse_case(piqp_kkt_solver ks)
{
    piqp_float P_x[2] = {6, 4}, c[2] = {-1, -4}, A_x[2] = {1, -2}, b[1] = {0}, G_x[3] = {1, 1, 1};
    piqp_int P_p[3] = {0, 1, 2}, P_i[2] = {0, 1}, A_p[3] = {0, 1, 2}, A_i[2] = {0, 0}, G_p[3] = {0, 3, 3}, G_i[3] = {0, 1, 2};
    piqp_float h_l[3] = {-1, -PIQP_INF, -2}, h_u[3] = {PIQP_INF, 1, 2}, x_l[2] = {-PIQP_INF, -1}, x_u[2] = {PIQP_INF, 1};
    piqp_settings settings;
    piqp_data_sparse data;
    piqp_workspace* work = NULL;
    piqp_set_default_settings_sparse(&settings);
    TRUE_(settings.kkt_solver == PIQP_SPARSE_LDLT);
    settings.kkt_solver = ks;
    data.n = 2; data.p = 1; data.m = 3;
    data.P = piqp_csc_matrix(2, 2, 2, P_p, P_i, P_x); data.c = c;
    data.A = piqp_csc_matrix(1, 2, 2, A_p, A_i, A_x); data.b = b;
    data.G = piqp_csc_matrix(3, 2, 3, G_p, G_i, G_x); data.h_l = h_l; data.h_u = h_u; data.x_l = x_l; data.x_u = x_u;
    piqp_setup_sparse(&work, &data, &settings);
    TRUE_(work != NULL);
    if (work) {
        TRUE_(work->solver_info.is_dense == 0);
        TRUE_(piqp_solve(work) == PIQP_SOLVED);
        check_first(work->result);
        P_x[0] = 8; A_x[1] = -3; h_u[0] = 2; x_u[1] = 2;
        piqp_update_sparse(work, data.P, NULL, data.A, NULL, NULL, NULL, h_u, NULL, x_u);
        TRUE_(piqp_solve(work) == PIQP_SOLVED);
        check_second(work->result);
        piqp_cleanup(work);
    }
    free(data.P); free(data.A); free(data.G);
}

int main(void)
{
    dense_case();
    sparse_case(PIQP_SPARSE_LDLT);
    sparse_case(PIQP_SPARSE_LDLT_EQ_COND);
    sparse_case(PIQP_SPARSE_LDLT_INEQ_COND);
    sparse_case(PIQP_SPARSE_LDLT_COND);
    sparse_case(PIQP_SPARSE_MULTISTAGE);
    printf(failures ? "%d check(s) failed\n" : "c interface: all checks passed%.0d\n", failures);
    return failures ? 1 : 0;
}
