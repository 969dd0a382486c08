// piqp_amd/csrc/multistage_symbolic.cpp -- see multistage_symbolic.hpp.
#include "multistage_symbolic.hpp"

#include <algorithm>
#include <cstdint>
#include <stdexcept>

namespace pq {
namespace multistage {

namespace {

using u64 = std::uint64_t;

// flop models of multistage_kkt.hpp:396-418 (size_t arithmetic, integer division)
struct Flops {
    static u64 gemm(u64 m, u64 n, u64 k) { return 2 * m * n * k; }
    static u64 trsm(u64 m, u64 n) { return m * m * n; }
    static u64 syrk(u64 n, u64 k) { return n * n * k; }
    static u64 potrf(u64 n) { return n * n * n / 3; }
};

// Lower triangle (column-compressed, rows ascending) of the structural pattern of
// P_ltri + I + AT*AT^T + GT*GT^T (multistage_kkt.hpp:424-431).  Column c lists the rows r >= c.
struct LowerPattern {
    std::vector<int> ptr, idx;
};

struct RowView {  // row-compressed view of an n x cols CSC pattern
    std::vector<int> ptr, col;
    RowView(int n, int cols, const int* Mp, const int* Mi)
    {
        const int nnz = cols ? Mp[cols] : 0;
        ptr.assign(n + 1, 0);
        col.assign(nnz, 0);
        for (int q = 0; q < nnz; ++q) ptr[Mi[q] + 1]++;
        for (int i = 0; i < n; ++i) ptr[i + 1] += ptr[i];
        std::vector<int> next(ptr.begin(), ptr.end() - 1);
        for (int c = 0; c < cols; ++c)
            for (int q = Mp[c]; q < Mp[c + 1]; ++q) col[next[Mi[q]]++] = c;
    }
};

LowerPattern condensed_lower_pattern(int n, const int* Pp, const int* Pi, int p, const int* ATp, const int* ATi, int m, const int* GTp, const int* GTi)
{
    RowView Prow(n, n, Pp, Pi), Arow(n, p, ATp, ATi), Grow(n, m, GTp, GTi);
    LowerPattern C;
    C.ptr.assign(n + 1, 0);
    std::vector<int> seen(n, -1), col;
    for (int c = 0; c < n; ++c) {
        col.clear();
        auto touch = [&](int r) {
            if (r >= c && seen[r] != c) { seen[r] = c; col.push_back(r); }
        };
        touch(c);                                                                 // identity
        for (int q = Prow.ptr[c]; q < Prow.ptr[c + 1]; ++q) touch(Prow.col[q]);    // P_utri(c, r) == P_ltri(r, c)
        for (int q = Arow.ptr[c]; q < Arow.ptr[c + 1]; ++q) {                     // every constraint containing variable c couples c to its other variables
            const int k = Arow.col[q];
            for (int t = ATp[k]; t < ATp[k + 1]; ++t) touch(ATi[t]);
        }
        for (int q = Grow.ptr[c]; q < Grow.ptr[c + 1]; ++q) {
            const int k = Grow.col[q];
            for (int t = GTp[k]; t < GTp[k + 1]; ++t) touch(GTi[t]);
        }
        std::sort(col.begin(), col.end());
        C.idx.insert(C.idx.end(), col.begin(), col.end());
        C.ptr[c + 1] = (int)C.idx.size();
    }
    return C;
}

struct Candidate {  // block_structure_info, multistage_kkt.hpp:433-439
    int prev_diag = 0, start = 0, diag = 0, off = 0, arrow = 0;
};

struct Tally {  // running flop counts, multistage_kkt.hpp:442-446
    u64 tridiag = 0, arrow_no_syrk = 0, arrow_syrk = 0;
};

// get_next_block_structure (:454-521): scan row `row` of the upper pattern and, entry by entry, either grow the
// current diagonal/off-diagonal block or widen the arrow, whichever adds fewer flops
Candidate scan_row(const LowerPattern& C, int n, int row, Candidate cand, const Tally& t)
{
    for (int q = C.ptr[row]; q < C.ptr[row + 1]; ++q) {
        const int col = C.idx[q];
        if (col < cand.start || col + cand.arrow >= n) continue;
        const int block_size = std::max(col - cand.start + 1, cand.diag + cand.off);
        const int diag_cap = row - cand.start + 1;
        const int diag_min = std::max(cand.diag, (block_size + 1) / 2);
        const int new_diag = std::max(diag_min, diag_cap);
        const int new_off = block_size - new_diag;
        const int remaining = n - cand.start - cand.diag - cand.off;
        const int new_arrow = std::min(std::max(cand.arrow, n - col), remaining);

        const u64 d_tridiag = Flops::syrk((u64)new_diag, (u64)cand.prev_diag) + Flops::potrf((u64)new_diag) + Flops::trsm((u64)new_diag, (u64)new_off);

        const u64 aw = (u64)(((cand.arrow + 3) / 4) * 4), naw = (u64)(((new_arrow + 3) / 4) * 4);  // dense kernels work on multiples of 4
        const u64 arrow_now = aw * t.arrow_no_syrk + aw * aw * t.arrow_syrk + Flops::potrf(aw);
        u64 arrow_new = naw * t.arrow_no_syrk + naw * naw * t.arrow_syrk;
        arrow_new += Flops::gemm(naw, (u64)cand.prev_diag, (u64)new_diag);
        arrow_new += Flops::trsm((u64)new_diag, naw);
        arrow_new += Flops::syrk(naw, (u64)new_diag);
        arrow_new += Flops::potrf(naw);

        // unsigned differences exactly as in the reference (flops_tridiag_new - flops_tridiag == d_tridiag)
        if (d_tridiag <= arrow_new - arrow_now) {
            cand.diag = new_diag;
            cand.off = new_off;
        } else {
            cand.arrow = new_arrow;
        }
    }
    return cand;
}

}  // namespace

std::vector<BlockInfo> detect_arrow_structure(int n, const int* Pp, const int* Pi, int p, const int* ATp, const int* ATi, int m, const int* GTp, const int* GTi)
{
    const LowerPattern C = condensed_lower_pattern(n, Pp, Pi, p, ATp, ATi, m, GTp, GTi);
    std::vector<BlockInfo> blocks;
    Candidate cur;
    Tally tally;
    auto advance = [&]() {
        cur.start += cur.diag;
        cur.prev_diag = cur.diag;
        cur.diag = cur.off;
        cur.off = 0;
    };
    for (int i = 0; i < n; ++i) {
        cur = scan_row(C, n, i, cur, tally);
        if (i + 1 < cur.start + cur.diag) continue;

        const bool good_ratio = cur.diag >= 2 * cur.off;
        const bool at_end = i + 1 >= n - cur.arrow;
        bool close = good_ratio || at_end;
        if (!close) {  // would the next row make the block grow?  then close it now (:531-535)
            const Candidate nxt = scan_row(C, n, i + 1, cur, tally);
            close = nxt.diag + nxt.off > cur.diag + cur.off;
        }
        if (close) {
            blocks.push_back({cur.start, cur.diag, cur.off});
            tally.tridiag += Flops::syrk((u64)cur.diag, (u64)(cur.prev_diag + 1)) + Flops::potrf((u64)cur.diag) + Flops::trsm((u64)cur.diag, (u64)cur.off);
            tally.arrow_no_syrk += Flops::gemm(1, (u64)cur.prev_diag, (u64)cur.diag) + Flops::trsm((u64)cur.diag, 1);
            tally.arrow_syrk += Flops::syrk(1, (u64)cur.diag);
            advance();
        }
        if (at_end && cur.diag > 0) {  // last block in front of the arrow (:555-564)
            blocks.push_back({cur.start, cur.diag, cur.off});
            advance();
        }
        if (at_end) break;
    }
    // blocks that were split in two are merged again (:573-583); the scan index moves on after a merge
    for (std::size_t i = 0; i + 1 < blocks.size(); ++i) {
        if (blocks[i].off_diag_size == blocks[i + 1].diag_size && blocks[i + 1].off_diag_size == 0) {
            blocks[i].diag_size += blocks[i].off_diag_size;
            blocks[i].off_diag_size = 0;
            blocks.erase(blocks.begin() + (std::ptrdiff_t)i + 1);
        }
    }
    blocks.push_back({cur.start, cur.arrow, 0});  // arrow corner block
    return blocks;
}

namespace {

// stage of a constraint row = block holding its first stored column, capped at the last non-arrow block (:712-714)
int stage_of_first_column(const std::vector<BlockInfo>& bi, int j)
{
    const int last = (int)bi.size() - 2;
    int b = 0;
    while (bi[b].start + bi[b].diag_size <= j && b < last) ++b;
    return b;
}

// local row of variable j inside stage b's front: [diag | off | arrow]
int local_row(const Symbolic& S, int b, int j)
{
    const BlockInfo& B = S.block_info[b];
    if (j >= S.n - S.arrow) return S.w[b] + S.off[b] + (j - (S.n - S.arrow));
    if (j < B.start) throw std::runtime_error("multistage: index in no valid block");
    if (j < B.start + B.diag_size) return j - B.start;
    const int o = j - B.start - B.diag_size;
    if (o >= B.off_diag_size) throw std::runtime_error("multistage: index in no valid block");
    return S.w[b] + o;
}

void group_constraints(const Symbolic& S, int rows, const int* Mp, const int* Mi, Symbolic::Grouped& Gp)
{
    const int stages = S.N - 1;
    std::vector<int> count(stages, 0), stage(rows, -1);
    for (int k = 0; k < rows; ++k)
        if (Mp[k] < Mp[k + 1]) { stage[k] = stage_of_first_column(S.block_info, Mi[Mp[k]]); count[stage[k]]++; }
    Gp.row_ptr.assign(S.N, 0);
    for (int b = 0; b < stages; ++b) Gp.row_ptr[b + 1] = Gp.row_ptr[b] + count[b];
    const int grouped = Gp.row_ptr[stages];
    Gp.perm.assign(rows, 0);
    Gp.rows.assign(rows, 0);
    std::vector<int> fill(Gp.row_ptr.begin(), Gp.row_ptr.end() - 1);
    int tail = grouped;
    for (int k = 0; k < rows; ++k) Gp.perm[k] = stage[k] >= 0 ? fill[stage[k]]++ : tail++;  // empty rows last (:738-741)
    for (int k = 0; k < rows; ++k) Gp.rows[Gp.perm[k]] = k;
    Gp.x_off.assign(S.N, 0);
    for (int b = 0; b < stages; ++b) Gp.x_off[b + 1] = Gp.x_off[b] + (long long)S.h[b] * count[b];
    Gp.x_doubles = Gp.x_off[stages];
    const int nnz = rows ? Mp[rows] : 0;
    Gp.dst.assign(nnz, 0);
    for (int k = 0; k < rows; ++k) {
        if (stage[k] < 0) continue;
        const int b = stage[k];
        const long long col = Gp.perm[k] - Gp.row_ptr[b];
        for (int q = Mp[k]; q < Mp[k + 1]; ++q) Gp.dst[q] = Gp.x_off[b] + local_row(S, b, Mi[q]) + col * S.h[b];
    }
}

}  // namespace

void analyse(const pq_sparse_data* d, Symbolic& S)
{
    S.n = d->n; S.p = d->p; S.m = d->m;
    static const int zero_ptr[1] = {0};
    const int* ATp = S.p ? d->AT_colptr : zero_ptr;
    const int* GTp = S.m ? d->GT_colptr : zero_ptr;
    S.block_info = detect_arrow_structure(S.n, d->P_colptr, d->P_rowind, S.p, ATp, d->AT_rowind, S.m, GTp, d->GT_rowind);
    S.N = (int)S.block_info.size();
    if (S.N < 2) throw std::runtime_error("multistage: structure detection produced no block");
    S.arrow = S.block_info.back().diag_size;
    {
        int acc = 0;
        for (const BlockInfo& b : S.block_info) {
            if (b.start != acc) throw std::runtime_error("multistage: blocks are not contiguous");
            acc += b.diag_size;
        }
        if (acc != S.n) throw std::runtime_error("multistage: blocks do not cover all variables");
    }
    const int N = S.N;
    S.w.assign(N, 0); S.off.assign(N, 0); S.h.assign(N, 0);
    S.front_off.assign(N + 1, 0); S.pan_off.assign(N + 1, 0); S.qpan_off.assign(N + 1, 0);
    for (int b = 0; b < N; ++b) {
        S.w[b] = S.block_info[b].diag_size;
        S.off[b] = b < N - 1 ? S.block_info[b].off_diag_size : 0;
        S.h[b] = b < N - 1 ? S.w[b] + S.off[b] + S.arrow : S.arrow;
        if (b + 1 < N - 1 && S.off[b] > S.block_info[b + 1].diag_size) throw std::runtime_error("multistage: off-diagonal block wider than the next stage");
        S.front_off[b + 1] = S.front_off[b] + (long long)S.h[b] * S.h[b];
        S.pan_off[b + 1] = S.pan_off[b] + (long long)S.h[b] * S.w[b] + (long long)S.w[b] * S.w[b];
        S.qpan_off[b + 1] = S.qpan_off[b] + (long long)S.h[b] * S.w[b];  // w*w + (h-w)*w
        S.max_h = std::max(S.max_h, S.h[b]);
        S.max_w = std::max(S.max_w, S.w[b]);
        const double wv = S.w[b], u = S.h[b] - S.w[b];
        S.flops_factor += wv * wv * wv / 3.0 + u * wv * wv + u * u * wv;
    }
    S.front_doubles = S.front_off[N];
    S.pan_doubles = S.pan_off[N];
    S.qpan_doubles = S.qpan_off[N];
    // P_utri entry (row j <= column i) is the lower entry (i, j): it lives in the front of the stage that owns column j
    const int nzP = d->P_colptr[S.n];
    S.P_dst.assign(nzP, 0);
    {
        int b_of_i = 0;  // block containing variable i (utri_to_kkt :617-623)
        std::vector<int> block_of(S.n, 0);
        for (int i = 0; i < S.n; ++i) {
            while (i >= S.block_info[b_of_i].start + S.block_info[b_of_i].diag_size) ++b_of_i;
            block_of[i] = b_of_i;
        }
        for (int i = 0; i < S.n; ++i)
            for (int q = d->P_colptr[i]; q < d->P_colptr[i + 1]; ++q) {
                const int j = d->P_rowind[q];
                if (j > i) throw std::runtime_error("multistage: P is not upper triangular");
                const int b = block_of[j];
                const int lc = j - S.block_info[b].start;
                const int lr = b == N - 1 ? i - S.block_info[b].start : local_row(S, b, i);
                S.P_dst[q] = S.front_off[b] + lr + (long long)lc * S.h[b];
            }
    }
    group_constraints(S, S.p, ATp, d->AT_rowind, S.A);
    group_constraints(S, S.m, GTp, d->GT_rowind, S.G);
    for (int b = 0; b + 1 < N; ++b) S.max_rows = std::max(S.max_rows, std::max(S.A.row_ptr[b + 1] - S.A.row_ptr[b], S.G.row_ptr[b + 1] - S.G.row_ptr[b]));
}

}  // namespace multistage
}  // namespace pq
