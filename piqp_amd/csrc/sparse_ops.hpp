// piqp_amd/csrc/sparse_ops.hpp -- device-resident CSC copies of the problem matrices and the mat-vec
// products every sparse backend needs (reference sparse/kkt.hpp:179-203; the multistage backend's
// block_symv_l / block_t_gemv_* of sparse/multistage_kkt.hpp:291-383,1355-1706 compute the same products).
//
// P is kept symmetrised, A and G in both orientations, so each product is a conflict-free column dot
// (one thread per output entry, no atomics -> bitwise reproducible).  The original-order device value
// arrays are exposed so the owning backend can remap them into its own storage without another H2D copy.
#pragma once

#include <vector>

#include "common.hpp"

namespace pq {

template <class T>
inline void upload_vec(DBuf<T>& d, const std::vector<T>& h, hipStream_t st)
{
    d.alloc(h.size() ? h.size() : 1);
    if (!h.empty()) {
        PQ_HIP(hipMemcpyAsync(d.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, st));
        stream_wait(st);  // setup-time only; `h` is often a temporary
    }
}

class CscOperators {
public:
    // patterns (host CSC of P_utri n x n, AT n x p, GT n x m) + values; synchronises `st`
    void init(const pq_sparse_data* d, hipStream_t st);
    // values only, identical sparsity (solver.hpp:325,341,356); synchronises `st`
    void upload_values(const pq_sparse_data* d, hipStream_t st);
    void clone_from(const CscOperators& o, hipStream_t st);

    void eval_P_x(double alpha, const double* x, double* z, hipStream_t st) const;
    void eval_A_xn_and_AT_xt(double an, double at, const double* xn, const double* xt, double* zn, double* zt, hipStream_t st) const;
    void eval_G_xn_and_GT_xt(double an, double at, const double* xn, const double* xt, double* zn, double* zt, hipStream_t st) const;
    // z += alpha * AT * y  /  z += alpha * GT * y  (n-vectors; used by the condensed right-hand sides)
    void add_AT_y(double alpha, const double* y, double* z, hipStream_t st) const;
    void add_GT_y(double alpha, const double* y, double* z, hipStream_t st) const;

    // out = rhs_x + GT * (zinv .* rhs_z) + delta_inv * AT * rhs_y   (condensed right-hand side, multistage_kkt.hpp:234-252)
    void fold_rhs(const double* rhs_x, const double* rhs_y, const double* rhs_z, const double* zinv, double delta_inv, double* out, hipStream_t st, bool with_A = true,
                  bool with_G = true) const;
    // lhs_y = delta_inv * A x - delta_inv * rhs_y ;  lhs_z = (G x - rhs_z) .* zinv   (multistage_kkt.hpp:266-287)
    void recover_duals(const double* x, const double* rhs_y, const double* rhs_z, const double* zinv, double delta_inv, double* lhs_y, double* lhs_z, hipStream_t st,
                       bool with_A = true, bool with_G = true) const;

    // The same two on LISTED rows only (stage partition of the condensed backends): out[j] for the listed x rows j, lhs_y / lhs_z for the
    // listed constraint rows -- one thread per listed row, column sums left to right in the statements of the kernels above: bitwise their values in those rows.
    void fold_rhs_rows(const int* rows_x, int nx, const double* rhs_x, const double* rhs_y, const double* rhs_z, const double* zinv, double delta_inv, double* out, hipStream_t st,
                       bool with_A, bool with_G) const;
    void recover_duals_rows(const int* rows_y, int ny, const int* rows_z, int nz, const double* x, const double* rhs_y, const double* rhs_z, const double* zinv, double delta_inv,
                            double* lhs_y, double* lhs_z, hipStream_t st) const;

    // Refinement residual on LISTED rows only (stage partition): err_x[i] = rhs_x[i] - (((P x)_i + x_reg_i x_i) + (A^T y)_i) + (G^T z)_i),
    // err_y[j] = rhs_y[j] - ((A x)_j - delta y_j), err_z[k] = rhs_z[k] - ((G x)_k - z_reg_k z_k) -- one thread per listed row, the column sums left to right with
    // the multiply-add of the eval_* kernels and the statements of k_err_x / k_err_yz: bitwise the values the full evaluation leaves in those rows.
    // |err| of the listed rows is max-combined into absmax_bits[0] (NaN-propagating, like k_absmax).  false: some column is too long for this form.
    bool residual_rows(const int* rows_x, int nx, const int* rows_y, int ny, const int* rows_z, int nz, const double* lhs_x, const double* lhs_y, const double* lhs_z,
                       const double* rhs_x, const double* rhs_y, const double* rhs_z, const double* x_reg, double delta, const double* z_reg, double* err_x, double* err_y,
                       double* err_z, unsigned long long* absmax_bits, hipStream_t st) const;
    // reference-order mode (the backend of sparse_exact.hip sets it): every product is formed term by term in the order of the reference's loops (sparse/kkt.hpp:179-203
    // as restated by the CPU oracle), columns of any length left to right -- with the file built without FMA contraction the mat-vecs are then bitwise the oracle's
    void set_reference_order(bool on) { ref_order_ = on; }
    bool reference_order() const { return ref_order_; }
    bool has_long_columns() const { return nlong_[0] + nlong_[1] + nlong_[2] + nlong_[3] + nlong_[4] > 0; }

    int n() const { return n_; }
    int p() const { return p_; }
    int m() const { return m_; }
    int nzP() const { return nzP_; }
    // host copies of the column patterns of A (p x n) and G (m x n): the constraint rows each x variable appears in (the backend's stage partition reads them once)
    void download_column_patterns(std::vector<int>& Ap, std::vector<int>& Ai, std::vector<int>& Gp, std::vector<int>& Gi, hipStream_t st) const
    {
        auto get = [&](const DBuf<int>& b, size_t cnt, std::vector<int>& out) {
            out.assign(cnt, 0);
            if (cnt) PQ_HIP(hipMemcpyAsync(out.data(), b.p, sizeof(int) * cnt, hipMemcpyDeviceToHost, st));
        };
        Ap.assign((size_t)n_ + 1, 0); Gp.assign((size_t)n_ + 1, 0); Ai.clear(); Gi.clear();
        if (p_ > 0) { get(A_p_, (size_t)n_ + 1, Ap); get(A_i_, (size_t)nzA_, Ai); }
        if (m_ > 0) { get(G_p_, (size_t)n_ + 1, Gp); get(G_i_, (size_t)nzG_, Gi); }
        PQ_HIP(hipStreamSynchronize(st));
    }
    int nzA() const { return nzA_; }
    int nzG() const { return nzG_; }
    const double* P_x() const { return P_x_.p; }    // P_utri values, caller's CSC order
    const double* AT_x() const { return AT_x_.p; }  // AT values, caller's CSC order
    const double* GT_x() const { return GT_x_.p; }
    const double* P_diag() const { return Pdiag_.p; }  // 0 where P has no structural diagonal (kkt_system.hpp:437-453)

private:
    int n_ = 0, p_ = 0, m_ = 0, nzP_ = 0, nzA_ = 0, nzG_ = 0, nzPf_ = 0;
    DBuf<double> P_x_, Pf_x_, AT_x_, A_x_, GT_x_, G_x_, Pdiag_;
    DBuf<int> Pf_p_, Pf_i_, Pf_src_, AT_p_, AT_i_, A_p_, A_i_, A_src_, GT_p_, GT_i_, G_p_, G_i_, G_src_;
    DBuf<int> long_Pf_, long_AT_, long_A_, long_GT_, long_G_;  // columns with more than SPMV_LONG_COL entries, per copy
    int nlong_[5] = {0, 0, 0, 0, 0};
    bool ref_order_ = false;
};

// generic value movers shared by the sparse backends
void launch_remap_values(int nnz, const int* dst_idx, const double* src, double* dst, hipStream_t st);          // dst[dst_idx[q]] = src[q]
void launch_remap_values64(int nnz, const long long* dst_idx, const double* src, double* dst, hipStream_t st);  // 64-bit destinations
void launch_gather_values(int nnz, const int* src_idx, const double* src, double* dst, hipStream_t st);         // dst[q] = src[src_idx[q]]

}  // namespace pq
