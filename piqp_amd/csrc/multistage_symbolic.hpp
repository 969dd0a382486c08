// piqp_amd/csrc/multistage_symbolic.hpp -- host-side (setup-time) structure analysis of the
// `sparse_multistage` backend (reference include/piqp/sparse/multistage_kkt.hpp).
//
//   reference                                              here
//   extract_arrow_structure (:420-597)                     detect_arrow_structure: the same greedy flop-count heuristic
//                                                          on the pattern of P + I + AᵀA + GᵀG, same BlockInfo list
//   BlockKKT / BlockMat / BlockVec (blocksparse/*.hpp),    one *frontal* layout: stage i owns a dense square front of
//   utri_to_kkt (:599-670), transpose_to_block_mat         order h_i = diag_i + off_i + arrow whose leading diag_i
//   (:672-818)                                             columns are the reference's [D_i; B_i; E_i] column panel
//                                                          and whose trailing (off_i + arrow)² block is where the
//                                                          stage's contribution to D_{i+1}, E_{i+1} and the arrow
//                                                          corner is accumulated (multifrontal "update matrix").
//                                                          Constraint rows are grouped by stage exactly like
//                                                          transpose_to_block_mat does (block of the first column,
//                                                          empty rows last); stage i's rows form a dense h_i x r_i
//                                                          matrix X_i (the stacked [D;B;E] blocks of BlockMat).
// Everything here is integer work that depends only on the sparsity pattern; it yields scatter maps
// (CSC value index -> offset in the front / X arenas) so that update_data is a device scatter.
#pragma once

#include <vector>

#include "common.hpp"

namespace pq {
namespace multistage {

struct BlockInfo {  // blocksparse/block_info.hpp:20-25
    int start, diag_size, off_diag_size;
};

struct Symbolic {
    int n = 0, p = 0, m = 0;
    int N = 0;      // number of blocks including the arrow corner block (block_info.size())
    int arrow = 0;  // block_info.back().diag_size
    std::vector<BlockInfo> block_info;
    // fronts: stage i in [0, N-1): order h[i], pivots w[i] = diag_size, ld = h[i]; front N-1 = arrow corner (h = w = arrow)
    std::vector<int> w, off, h;
    std::vector<long long> front_off;  // N+1 offsets (doubles) into a front arena
    long long front_doubles = 0;
    // factor storage read by the solves: stage i keeps its h_i x w_i column panel (L_ii on top, [C_i; F_i] below)
    // followed by the w_i x w_i inverse of L_ii
    std::vector<long long> pan_off;  // N+1
    long long pan_doubles = 0;
    // compact variant used by the single-wave chain: per stage only L_ii^{-1} (w x w) and Q_i = [C_i; F_i] L_ii^{-1} ((h-w) x w)
    std::vector<long long> qpan_off;  // N+1
    long long qpan_doubles = 0;
    // constraint matrices grouped by stage: X_i is h_i x rows_i, column-major, ld = h_i
    struct Grouped {
        std::vector<int> row_ptr;          // N: first grouped row of each stage (row_ptr[N-1] = number of grouped rows)
        std::vector<int> rows;             // grouped position -> original constraint index (BlockMat::perm_inv)
        std::vector<int> perm;             // original constraint index -> grouped position (empty rows last)
        std::vector<long long> x_off;      // N offsets (doubles) of X_i in the X arena
        long long x_doubles = 0;
        std::vector<long long> dst;        // CSC value index -> offset in the X arena
    } A, G;
    std::vector<long long> P_dst;  // P_utri value index -> offset in the front arena
    int max_h = 0, max_w = 0, max_rows = 0;
    double flops_factor = 0.0;  // sum_i w^3/3 + (h-w) w^2 + (h-w)^2 w   
};

// the reference's extract_arrow_structure on CSC patterns (P_utri n x n upper, AT n x p, GT n x m)
std::vector<BlockInfo> detect_arrow_structure(int n, const int* Pp, const int* Pi, int p, const int* ATp, const int* ATi, int m, const int* GTp, const int* GTi);

void analyse(const pq_sparse_data* d, Symbolic& S);

}  // namespace multistage
}  // namespace pq
