"""piqp_amd -- MI355X-native (gfx950) KKT backend for the PIQP interior-point QP solver.

Only the hot path: KKT factor / solve / iterative refinement behind PIQP's KKTSolverBase + KKTSystem
interface, as hand-written HIP kernels behind the C-ABI of include/piqp_amd.h.  No CPU fallback.
"""
from . import _lib, kkt  # noqa: F401
from .kkt import (DENSE_CHOLESKY, DENSE_LDLT_NO_PIVOT, KKT_UPDATE_A, KKT_UPDATE_G, KKT_UPDATE_NONE, KKT_UPDATE_P,  # noqa: F401
                  Data, DenseKKT, DenseSolver, KKTSystem, SparseData, SparseSolver, Variables, default_settings)
SparseKKT = DenseKKT  # same handle type: pq_kkt_* dispatches on the backend (KKTSolverBase is one interface)
SPARSE_LDLT = 1
from .kkt import SPARSE_LDLT_EXACT, SPARSE_LDLT_MULTIFRONTAL  # noqa: E402,F401
SPARSE_MULTISTAGE = 5
MultistageKKT = DenseKKT
from .batch import BatchSparseSolver  # noqa: E402,F401
from .factor import LLT, LDLTNoPivot, LOWER, UPPER  # noqa: E402,F401
