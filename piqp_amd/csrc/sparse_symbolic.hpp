// piqp_amd/csrc/sparse_symbolic.hpp -- host-side (setup-time) analysis for the sparse KKT backend.
//
// Replaces the integer work of sparse::KKT's constructor (reference sparse/kkt.hpp:51-70):
//   create_kkt_matrix (sparse/kkt_full.hpp:39-170)            -> build_kkt_full
//   AMDOrdering::init (sparse/ordering.hpp:67-84, Eigen AMD)  -> amd_order
//   permute_sparse_symmetric_matrix (sparse/utils.hpp:32-128) -> permute_sym_upper
//   LDLt::factorize_symbolic (sparse/ldlt.hpp:42-99)          -> etree / column structure, extended here to
//                                                                supernodes + multifrontal assembly maps
// The reference's numeric phase is an up-looking row-by-row LDLt (strictly serial).  The device numeric
// phase is a supernodal multifrontal LDLt: the elimination tree is postordered, columns with identical
// structure are merged into supernodes, every supernode owns a dense frontal matrix in HBM, and fronts of
// the same tree level are factored concurrently.  This file computes everything that depends only on the
// sparsity pattern.
#pragma once

#include <vector>

#include "common.hpp"

namespace pq {
namespace sparse {

using IVec = std::vector<int>;
using DVec = std::vector<double>;

// Fronts that go through the dense multi-workgroup matrix-core kernels of sparse_kkt.hip instead of one workgroup's pivot loop (f rows, w pivots): shared by the
// numeric engine, the ordering cost model and the spine merging of sparse_symbolic.cpp (round-4 advice: the model priced `f >= 192 && w >= 32` only)
__host__ __device__ inline bool big_front(int f, int w) { return f >= 192 && (w >= 32 || (long long)f * w > 12288 || f >= 768); }

struct Symbolic {
    int n = 0, p = 0, m = 0, N = 0;  // N = KKT dimension of the mode (n + p + m for KKT_FULL)
    int mode = 0;                    // KKTMode bits: 1 = equalities eliminated, 2 = inequalities eliminated
    // eliminated blocks: pattern of upper(MT MT^T) and, per entry, the value-index pairs / constraint of every product term
    struct Gram {
        IVec colptr, rowind;  // n x n upper CSC pattern
        IVec ptr;             // per entry: range in q1/q2/k
        IVec q1, q2, k;       // value indices of (i, k) and (j, k) in MT, and the constraint k (weight index)
    } gramA, gramG;
    IVec gramA_to_Ki, gramG_to_Ki;
    // K (upper, diagonal last in every column) and the value maps of kkt_full.hpp
    IVec Kp, Ki;
    DVec Kx;
    IVec P_utri_to_Ki, AT_to_Ki, GT_to_Ki;
    // ordering: P[new] = old, P_inv[old] = new (AMD composed with the etree postorder)
    IVec P, P_inv;
    IVec fill_perm;  // the fill-reducing ordering itself (AMD: what Eigen::AMDOrdering returns, sparse/ordering.hpp:72-76; or nested dissection), before the postorder
    // permuted matrix PKPt (upper) and K-index -> PKPt-index map
    IVec Cp, Ci, PKi;
    IVec diag_pos;  // PKPt value index of the diagonal of ORIGINAL column `col` (kkt_full.hpp:181,194,207)
    // elimination tree / supernodes / fronts
    IVec etree;
    int nsuper = 0;
    IVec sn_first;          // nsuper+1: first column of each supernode
    IVec sn_of_col;         // N
    IVec sn_parent;         // assembly tree
    IVec sn_nind;           // leading pivot columns of the supernode that are mutually independent (merged sibling leaves): no updates among them
    IVec front_rows_ptr;    // nsuper+1 into front_rows
    IVec front_rows;        // global (permuted) indices of every front, pivots first, sorted
    std::vector<long long> front_off;  // nsuper+1 offsets (in doubles) into the front workspace
    IVec level_ptr, level_sn;          // supernodes grouped by level (leaves first)
    int nlevels = 0;
    // device schedule: small subtrees are walked by ONE workgroup each (supernodes sub_lo[k]..sub_hi[k], a postorder range)
    // in a single launch; only the supernodes above them are processed level by level (top_level_*)
    IVec sub_lo, sub_hi;
    int nsub = 0, sub_max_front = 0;
    IVec top_level_ptr, top_level_sn;
    int top_nlevels = 0;
    // the substitution sweeps use their own, finer partition of the same tree (shorter single-wave walks, more supernodes in the
    // flag-ordered top): same meaning as the five members above
    IVec solve_sub_lo, solve_sub_hi;
    IVec solve_top_level_ptr, solve_top_level_sn;
    int solve_top_nlevels = 0, solve_sub_max_front = 0;
    // the top supernodes of the substitution grouped into walks: maximal runs lo..hi of consecutive supernodes with parent[t] == t + 1
    // (a chain needs no flag between its links), ordered by the level of their last supernode
    IVec solve_walk_lo, solve_walk_hi;
    // assembly: PKPt value q goes to fronts[a_dst[q]]
    std::vector<long long> a_dst;
    IVec fe_ptr, fe_q, fe_off;  // the same map grouped by owning supernode: entries fe_ptr[s]..fe_ptr[s+1]: value index, offset inside the front
    // extend-add: for child c, rel[rel_ptr[c] + i] = position in the parent's front of c's i-th update row
    IVec rel_ptr, rel;
    IVec child_ptr, child;  // children lists (fixed order = increasing supernode id)
    long long front_doubles = 0;
    long long nnzL = 0;      // entries of L below the diagonal (for the roofline byte counts)
    double flops = 0.0;      // sum_j (c_j^2 + 3 c_j)
    int max_front = 0;
    const char* ordering = "amd";
};

// ---- the reference's own elimination, statement for statement (round 5): sparse::KKT with AMDOrdering + LDLt (sparse/kkt.hpp:51-70, ldlt.hpp:42-169).
// Everything the up-looking numeric phase decides at run time from the pattern alone is fixed here once: the AMD permutation WITHOUT a postorder (the
// reference factors P K P' in AMD's own order), the elimination tree, L's column structure, and for every row k of L its pattern IN THE ORDER the
// reference's depth-first walk produces it (ldlt.hpp:127-143) -- the order in which the terms of every entry are subtracted.  The device kernels of
// sparse_exact.hip replay exactly that order, so that L, D and every solve are bitwise the reference's (the CPU oracle's, which restates it).
struct UpLooking {
    int n = 0, p = 0, m = 0, N = 0, mode = 0;
    IVec perm, perm_inv;        // perm[new] = old (Eigen::AMDOrdering, sparse/ordering.hpp:72-76)
    IVec Cp, Ci, PKi;           // C = upper(P K P'), sorted columns; K value index -> C value index (sparse/utils.hpp:32-128)
    IVec diag_pos;              // C value index of the diagonal of ORIGINAL column col (kkt_full.hpp:181,194,207: the last entry of column perm_inv[col])
    IVec mapP, mapA, mapG;      // caller's P_utri / AT / GT value index -> C value index (kkt_full.hpp:219-249)
    IVec etree;                 // ldlt.hpp:61-83
    IVec Lp, Li, Lcol;          // L strictly lower, CSC with ascending rows (the reference fills every column in row order); Lcol = column of every entry
    IVec Rp, Rcol, Rpos;        // row k of L: entries Rp[k] .. Rp[k+1] in the reference's topological order: column index, position in the CSC arrays
    // schedule of the factorisation (see analyse_uplooking): tasks = paths of the elimination tree of at most 64 rows
    IVec task_ptr, task_rows;   // rows of every task, bottom-up (each the parent of the one before)
    IVec row_task, row_lane, row_prev;  // per row: its task, its position in it, the row before it in the task (-1: first)
    IVec dep_ptr, dep;          // per row: the children outside its task (rows whose completion the row pass waits for)
    IVec tk_kind, tk_id;        // tickets in issue order: 0 = row pass of row id, 1 = path pass of task id
    IVec Rcnt, Rtab;            // per entry of a row: leading entries of its column the row pass scatters (-1: column of the task's own path); row of the task's table
    IVec tab_ptr, mask_ptr, task_nU;  // per task: offset of its (nU + W) x W value table, of its nU + W presence words, columns outside the path that reach it
    std::vector<unsigned long long> Tmask;
    // substitution on the same tasks (see analyse_uplooking): table rows of every task sorted by column, the tasks a task's forward pass waits for, the task of
    // its last row's parent, the tasks in the order of their last rows, and per CSC entry the lane of its row inside its column's task (-1: another task)
    IVec fs_ptr, fs_u, fs_col, tdep_ptr, tdep, tparent, tsort, Lsrc;
    long long nnzL = 0;
    double flops = 0.0;         // sum_j (c_j^2 + 3 c_j)
    int height = 0;             // elimination tree height (rows)
    long long crit_steps = 0;   // longest root path counted in row entries (the dependent steps of the up-looking loop)
};
// K of the mode must already be in S (analyse_kkt_pattern)
void analyse_uplooking(const Symbolic& S, const pq_sparse_data* d, UpLooking& U);

// Stage partition of the assembly tree over `world` processes: disjoint subtrees are owned by one rank each,
// the supernodes above them ("shared top") are processed by every rank on exchanged data.
struct Partition {
    int world = 1;
    IVec owner;                       // per supernode: owning rank, -1 = shared top
    IVec boundary;                    // owned supernodes whose parent is shared, ascending: their update matrix / vector crosses ranks
    std::vector<long long> bmat_off;  // boundary.size()+1 offsets (doubles) of the u x u update blocks in the factor exchange buffer
    IVec bvec_off;                    // boundary.size()+1 offsets of the u-vectors in the forward-substitution exchange buffer
    IVec span_lo, span_hi;            // per rank: permuted-column range holding every column the rank owns (other columns inside are shared)
    int max_span = 0;
    DVec work;                        // per rank: factorisation flops of the owned subtrees
    double shared_work = 0.0, total_work = 0.0;
};
void partition_tree(const Symbolic& S, int world, Partition& P);

void amd_order(int n, const int* Ap, const int* Ai, int* perm);
// permute_sparse_symmetric_matrix (sparse/utils.hpp:32-128): C = upper(P A P') with sorted columns, Ai_to_Ci = map of value positions
void permute_sym_upper(int n, const IVec& Ap, const IVec& Ai, const int* perm_inv, IVec& Cp, IVec& Ci, IVec& Ai_to_Ci);
// nested dissection by BFS level structures, AMD inside parts of at most `leaf` nodes; perm[new] = old
void nd_order(int n, const int* Ap, const int* Ai, int* perm, int leaf);
void analyse_kkt_full(const pq_sparse_data* d, Symbolic& S);
// any KKTMode (kkt_fwd.hpp:15-21): 0 full, 1 eq eliminated, 2 ineq eliminated, 3 all eliminated
void analyse_kkt(const pq_sparse_data* d, int mode, Symbolic& S);
// its first half only: K of the mode (create_kkt_matrix) and the value maps into it, no ordering
void analyse_kkt_pattern(const pq_sparse_data* d, int mode, Symbolic& S);

}  // namespace sparse
}  // namespace pq
