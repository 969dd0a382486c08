#include "common.hpp"
using namespace pq;
extern "C" {
int pq_solver_create(pq_solver** out, int) { if (out) *out = nullptr; return fail(PQ_ERR_UNSUPPORTED, "solver front-end not built"); }
void pq_solver_destroy(pq_solver*) {}
int pq_solver_clone(const pq_solver*, pq_solver** out) { if (out) *out = nullptr; return fail(PQ_ERR_UNSUPPORTED, "solver front-end not built"); }
pq_settings* pq_solver_settings(pq_solver*) { return nullptr; }
int pq_solver_setup_dense(pq_solver*, int, int, int, const double*, const double*, const double*, const double*, const double*, const double*, const double*, const double*, const double*) { return PQ_ERR_UNSUPPORTED; }
int pq_solver_setup_sparse(pq_solver*, int, int, int, const int*, const int*, const double*, const double*, const int*, const int*, const double*, const double*, const int*, const int*, const double*, const double*, const double*, const double*, const double*) { return PQ_ERR_UNSUPPORTED; }
int pq_solver_update_dense(pq_solver*, const double*, const double*, const double*, const double*, const double*, const double*, const double*, const double*, const double*) { return PQ_ERR_UNSUPPORTED; }
int pq_solver_update_sparse(pq_solver*, const int*, const int*, const double*, const double*, const int*, const int*, const double*, const double*, const int*, const int*, const double*, const double*, const double*, const double*, const double*) { return PQ_ERR_UNSUPPORTED; }
int pq_solver_solve(pq_solver*) { return PQ_UNSOLVED; }
const pq_info* pq_solver_info(const pq_solver*) { return nullptr; }
int pq_solver_get_result(const pq_solver*, pq_vars*) { return PQ_ERR_UNSUPPORTED; }
int pq_solver_dims(const pq_solver*, int*, int*, int*) { return PQ_ERR_UNSUPPORTED; }
int pq_solver_set_trace(pq_solver*, double*, int) { return PQ_ERR_UNSUPPORTED; }
int pq_solver_trace_rows(const pq_solver*) { return 0; }
}
