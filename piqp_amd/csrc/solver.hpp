// piqp_amd/csrc/solver.hpp -- host-side DenseSolver/SparseSolver front-end (reference solver.hpp) over the
// device-resident KKTSystem.  Declared here, defined in solver.cpp; exported through capi (pq_solver_*).
#pragma once
#include "common.hpp"
