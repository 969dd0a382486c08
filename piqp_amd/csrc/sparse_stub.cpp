// placeholder until the sparse backend lands: the factory reports "kkt solver not supported"
// exactly like KKTSystem::init_kkt_solver's default branch (kkt_system.hpp:493-494).
#include "kkt_solver_base.hpp"
namespace pq {
KKTSolverBase* make_sparse_kkt(const pq_sparse_data*, int, int) { return nullptr; }
}
