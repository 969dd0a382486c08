// piqp_amd/csrc/piqp_c_api.cpp -- the reference's C interface (interfaces/c/src/piqp.cpp) on top of pq_solver_*:
// same entry points and struct layouts (include/piqp_c_compat.h), served by the device solver.  The glue only converts
// layouts: the reference's dense C interface is ROW-major (piqp.cpp:13 `CMat = ... RowMajor`) while pq_solver_setup_dense
// takes column-major matrices, settings / info are copied field by field, and the solution vectors are kept in
// workspace-owned buffers so that result->x etc. stay valid between calls like the reference's Result<T> members.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <new>
#include <thread>
#include <vector>

#include "../../include/piqp_amd.h"
#include "../../include/piqp_c_compat.h"

struct piqp_solver_handle {
    pq_solver* solver = nullptr;
    std::vector<double> vars[10];  // x, y, z_l, z_u, z_bl, z_bu, s_l, s_u, s_bl, s_bu
};

namespace {

void to_pq(const piqp_settings& s, pq_settings& o)
{
    o.rho_init = s.rho_init; o.delta_init = s.delta_init; o.eps_abs = s.eps_abs; o.eps_rel = s.eps_rel;
    o.check_duality_gap = s.check_duality_gap; o.eps_duality_gap_abs = s.eps_duality_gap_abs; o.eps_duality_gap_rel = s.eps_duality_gap_rel;
    o.infeasibility_threshold = s.infeasibility_threshold; o.reg_lower_limit = s.reg_lower_limit; o.reg_finetune_lower_limit = s.reg_finetune_lower_limit;
    o.reg_finetune_primal_update_threshold = s.reg_finetune_primal_update_threshold; o.reg_finetune_dual_update_threshold = s.reg_finetune_dual_update_threshold;
    o.max_iter = s.max_iter; o.max_factor_retires = s.max_factor_retires;
    o.preconditioner_scale_cost = s.preconditioner_scale_cost; o.preconditioner_reuse_on_update = s.preconditioner_reuse_on_update; o.preconditioner_iter = s.preconditioner_iter;
    o.tau = s.tau; o.kkt_solver = (int)s.kkt_solver;  // the enumerators carry the same values (include/piqp_amd.h pq_kkt_solver)
    o.iterative_refinement_always_enabled = s.iterative_refinement_always_enabled; o.iterative_refinement_eps_abs = s.iterative_refinement_eps_abs;
    o.iterative_refinement_eps_rel = s.iterative_refinement_eps_rel; o.iterative_refinement_max_iter = s.iterative_refinement_max_iter;
    o.iterative_refinement_min_improvement_rate = s.iterative_refinement_min_improvement_rate;
    o.iterative_refinement_static_regularization_eps = s.iterative_refinement_static_regularization_eps;
    o.iterative_refinement_static_regularization_rel = s.iterative_refinement_static_regularization_rel;
    o.verbose = s.verbose; o.compute_timings = s.compute_timings;
}
void from_pq(const pq_settings& s, piqp_settings& o)
{
    o.rho_init = s.rho_init; o.delta_init = s.delta_init; o.eps_abs = s.eps_abs; o.eps_rel = s.eps_rel;
    o.check_duality_gap = s.check_duality_gap; o.eps_duality_gap_abs = s.eps_duality_gap_abs; o.eps_duality_gap_rel = s.eps_duality_gap_rel;
    o.infeasibility_threshold = s.infeasibility_threshold; o.reg_lower_limit = s.reg_lower_limit; o.reg_finetune_lower_limit = s.reg_finetune_lower_limit;
    o.reg_finetune_primal_update_threshold = s.reg_finetune_primal_update_threshold; o.reg_finetune_dual_update_threshold = s.reg_finetune_dual_update_threshold;
    o.max_iter = s.max_iter; o.max_factor_retires = s.max_factor_retires;
    o.preconditioner_scale_cost = s.preconditioner_scale_cost; o.preconditioner_reuse_on_update = s.preconditioner_reuse_on_update; o.preconditioner_iter = s.preconditioner_iter;
    o.tau = s.tau; o.kkt_solver = (piqp_kkt_solver)s.kkt_solver;
    o.iterative_refinement_always_enabled = s.iterative_refinement_always_enabled; o.iterative_refinement_eps_abs = s.iterative_refinement_eps_abs;
    o.iterative_refinement_eps_rel = s.iterative_refinement_eps_rel; o.iterative_refinement_max_iter = s.iterative_refinement_max_iter;
    o.iterative_refinement_min_improvement_rate = s.iterative_refinement_min_improvement_rate;
    o.iterative_refinement_static_regularization_eps = s.iterative_refinement_static_regularization_eps;
    o.iterative_refinement_static_regularization_rel = s.iterative_refinement_static_regularization_rel;
    o.verbose = s.verbose; o.compute_timings = s.compute_timings;
}

// piqp_update_result (piqp.cpp:69-121): pointers + a copy of info
void refresh_result(piqp_workspace* w)
{
    piqp_solver_handle* h = w->solver_handle;
    piqp_result* r = w->result;
    pq_vars out{h->vars[0].data(), h->vars[1].data(), h->vars[2].data(), h->vars[3].data(), h->vars[4].data(),
                h->vars[5].data(), h->vars[6].data(), h->vars[7].data(), h->vars[8].data(), h->vars[9].data()};
    pq_solver_get_result(h->solver, &out);
    r->x = out.x; r->y = out.y; r->z_l = out.z_l; r->z_u = out.z_u; r->z_bl = out.z_bl; r->z_bu = out.z_bu;
    r->s_l = out.s_l; r->s_u = out.s_u; r->s_bl = out.s_bl; r->s_bu = out.s_bu;
    const pq_info* i = pq_solver_info(h->solver);
    piqp_info& o = r->info;
    o.status = (piqp_status)i->status; o.iter = i->iter;
    o.rho = i->rho; o.delta = i->delta; o.mu = i->mu; o.sigma = i->sigma; o.primal_step = i->primal_step; o.dual_step = i->dual_step;
    o.primal_res = i->primal_res; o.primal_res_rel = i->primal_res_rel; o.dual_res = i->dual_res; o.dual_res_rel = i->dual_res_rel;
    o.primal_res_reg = i->primal_res_reg; o.primal_res_reg_rel = i->primal_res_reg_rel; o.dual_res_reg = i->dual_res_reg; o.dual_res_reg_rel = i->dual_res_reg_rel;
    o.primal_prox_inf = i->primal_prox_inf; o.dual_prox_inf = i->dual_prox_inf; o.prev_primal_res = i->prev_primal_res; o.prev_dual_res = i->prev_dual_res;
    o.primal_obj = i->primal_obj; o.dual_obj = i->dual_obj; o.duality_gap = i->duality_gap; o.duality_gap_rel = i->duality_gap_rel;
    o.factor_retires = i->factor_retires; o.reg_limit = i->reg_limit; o.no_primal_update = i->no_primal_update; o.no_dual_update = i->no_dual_update;
    o.setup_time = i->setup_time; o.update_time = i->update_time; o.solve_time = i->solve_time;
    o.kkt_factor_time = i->kkt_factor_time; o.kkt_solve_time = i->kkt_solve_time; o.run_time = i->run_time;
}

int device_from_env()
{
    const char* e = std::getenv("PIQP_AMD_DEVICE");
    return e ? std::atoi(e) : 0;
}

// rows x cols row-major -> column-major, 32 x 32 tiles (both sides of the transpose stay in cache), a few threads for large matrices
std::vector<double> to_col_major(const double* a, int rows, int cols)
{
    std::vector<double> o((size_t)rows * cols);
    auto work = [&](int jb_lo, int jb_hi) {
        for (int jb = jb_lo; jb < jb_hi; jb += 32)
            for (int ib = 0; ib < rows; ib += 32)
                for (int j = jb; j < std::min(cols, jb + 32); ++j)
                    for (int i = ib; i < std::min(rows, ib + 32); ++i) o[i + (size_t)j * rows] = a[(size_t)i * cols + j];
    };
    const long long total = (long long)rows * cols;
    int T = total < (1LL << 20) ? 1 : (int)std::min<unsigned>(8, std::max(1u, std::thread::hardware_concurrency()));
    if (T <= 1) { work(0, cols); return o; }
    std::vector<std::thread> th;
    const int chunk = ((cols + T - 1) / T + 31) / 32 * 32;
    for (int lo = 0; lo < cols; lo += chunk) th.emplace_back(work, lo, std::min(cols, lo + chunk));
    for (auto& t : th) t.join();
    return o;
}

piqp_workspace* make_workspace(int is_dense, int n, int p, int m, const piqp_settings* settings)
{
    piqp_workspace* w = new (std::nothrow) piqp_workspace;
    if (!w) return nullptr;
    w->solver_handle = new piqp_solver_handle;
    w->result = new piqp_result;
    std::memset(w->result, 0, sizeof(piqp_result));
    w->solver_info.is_dense = is_dense; w->solver_info.n = n; w->solver_info.p = p; w->solver_info.m = m;
    if (pq_solver_create(&w->solver_handle->solver, device_from_env()) != PQ_OK) {
        std::fprintf(stderr, "piqp_setup: %s\n", pq_last_error_string());
        delete w->result; delete w->solver_handle; delete w;
        return nullptr;
    }
    pq_settings* st = pq_solver_settings(w->solver_handle->solver);
    st->kkt_solver = is_dense ? PQ_DENSE_CHOLESKY : PQ_SPARSE_LDLT;  // DenseSolver / SparseSolver constructors (solver.hpp:1265,1299)
    if (settings) to_pq(*settings, *st);
    const int len[10] = {n, p, m, m, n, n, m, m, n, n};
    for (int k = 0; k < 10; ++k) w->solver_handle->vars[k].assign((size_t)(len[k] > 0 ? len[k] : 0) + 1, 0.0);
    return w;
}

}  // namespace

extern "C" {

piqp_csc* piqp_csc_matrix(piqp_int m, piqp_int n, piqp_int nnz, piqp_int* p, piqp_int* i, piqp_float* x)
{
    piqp_csc* M = (piqp_csc*)std::malloc(sizeof(piqp_csc));
    if (!M) return nullptr;
    M->m = m; M->n = n; M->nnz = nnz; M->p = p; M->i = i; M->x = x;
    return M;
}

void piqp_set_default_settings_dense(piqp_settings* settings)
{
    pq_settings d;
    pq_settings_default(&d);
    d.kkt_solver = PQ_DENSE_CHOLESKY;
    from_pq(d, *settings);
}
void piqp_set_default_settings_sparse(piqp_settings* settings)
{
    pq_settings d;
    pq_settings_default(&d);
    d.kkt_solver = PQ_SPARSE_LDLT;
    from_pq(d, *settings);
}

void piqp_setup_dense(piqp_workspace** workspace, const piqp_data_dense* data, const piqp_settings* settings)
{
    *workspace = make_workspace(1, data->n, data->p, data->m, settings);
    piqp_workspace* w = *workspace;
    if (!w) return;
    const int n = data->n, p = data->p, m = data->m;
    // row-major P: its upper triangle is the lower triangle of the same buffer read column-major, so transpose like A and G
    std::vector<double> P = to_col_major(data->P, n, n), A, G;
    if (data->A) A = to_col_major(data->A, p, n);
    if (data->G) G = to_col_major(data->G, m, n);
    if (pq_solver_setup_dense(w->solver_handle->solver, n, p, m, P.data(), data->c, data->A ? A.data() : nullptr, data->b, data->G ? G.data() : nullptr, data->h_l, data->h_u,
                              data->x_l, data->x_u) <= 0)
        std::fprintf(stderr, "piqp_setup_dense: %s\n", pq_last_error_string());
    refresh_result(w);
}

void piqp_setup_sparse(piqp_workspace** workspace, const piqp_data_sparse* data, const piqp_settings* settings)
{
    *workspace = make_workspace(0, data->n, data->p, data->m, settings);
    piqp_workspace* w = *workspace;
    if (!w) return;
    const piqp_csc *P = data->P, *A = data->A, *G = data->G;
    if (pq_solver_setup_sparse(w->solver_handle->solver, data->n, data->p, data->m, P->p, P->i, P->x, data->c, A ? A->p : nullptr, A ? A->i : nullptr, A ? A->x : nullptr, data->b,
                               G ? G->p : nullptr, G ? G->i : nullptr, G ? G->x : nullptr, data->h_l, data->h_u, data->x_l, data->x_u) <= 0)
        std::fprintf(stderr, "piqp_setup_sparse: %s\n", pq_last_error_string());
    refresh_result(w);
}

void piqp_update_settings(piqp_workspace* workspace, const piqp_settings* settings)
{
    if (!workspace || !settings) return;
    to_pq(*settings, *pq_solver_settings(workspace->solver_handle->solver));
}

void piqp_update_dense(piqp_workspace* workspace, piqp_float* P, piqp_float* c, piqp_float* A, piqp_float* b, piqp_float* G, piqp_float* h_l, piqp_float* h_u, piqp_float* x_l,
                       piqp_float* x_u)
{
    if (!workspace) return;
    const int n = workspace->solver_info.n, p = workspace->solver_info.p, m = workspace->solver_info.m;
    std::vector<double> Pc, Ac, Gc;
    if (P) Pc = to_col_major(P, n, n);
    if (A) Ac = to_col_major(A, p, n);
    if (G) Gc = to_col_major(G, m, n);
    if (pq_solver_update_dense(workspace->solver_handle->solver, P ? Pc.data() : nullptr, c, A ? Ac.data() : nullptr, b, G ? Gc.data() : nullptr, h_l, h_u, x_l, x_u) <= 0)
        std::fprintf(stderr, "piqp_update_dense: %s\n", pq_last_error_string());
}

void piqp_update_sparse(piqp_workspace* workspace, piqp_csc* P, piqp_float* c, piqp_csc* A, piqp_float* b, piqp_csc* G, piqp_float* h_l, piqp_float* h_u, piqp_float* x_l,
                        piqp_float* x_u)
{
    if (!workspace) return;
    if (pq_solver_update_sparse(workspace->solver_handle->solver, P ? P->p : nullptr, P ? P->i : nullptr, P ? P->x : nullptr, c, A ? A->p : nullptr, A ? A->i : nullptr,
                                A ? A->x : nullptr, b, G ? G->p : nullptr, G ? G->i : nullptr, G ? G->x : nullptr, h_l, h_u, x_l, x_u) <= 0)
        std::fprintf(stderr, "piqp_update_sparse: %s\n", pq_last_error_string());
}

piqp_status piqp_solve(piqp_workspace* workspace)
{
    if (!workspace) return PIQP_UNSOLVED;
    const int status = pq_solver_solve(workspace->solver_handle->solver);
    refresh_result(workspace);
    return (piqp_status)status;
}

void piqp_cleanup(piqp_workspace* workspace)
{
    if (!workspace) return;
    if (workspace->solver_handle) { pq_solver_destroy(workspace->solver_handle->solver); delete workspace->solver_handle; }
    delete workspace->result;
    delete workspace;
}

}  // extern "C"
