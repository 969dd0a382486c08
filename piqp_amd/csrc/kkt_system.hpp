// piqp_amd/csrc/kkt_system.hpp -- mirror of piqp::KKTSystem<T,I,MatrixType> (reference kkt_system.hpp:27-538)
// with the same member names; every vector lives in HBM.  See kkt_system.hip.
#pragma once

#include <cmath>
#include <stdexcept>

#include "kkt_solver_base.hpp"

namespace pq {

class KKTSystem {
public:
    // takes ownership of `backend` (the reference holds it in a unique_ptr, kkt_system.hpp:65)
    KKTSystem(KKTSolverBase* backend, const pq_settings& settings);
    ~KKTSystem();
    KKTSystem(const KKTSystem&) = delete;
    KKTSystem& operator=(const KKTSystem&) = delete;

    KKTSystem* clone() const;  // kkt_system.hpp:70-95

    // finite-bound index lists + x_b_scaling of dense::Data / sparse::Data (host or device arrays per `mem`)
    void set_bounds(int n_h_l, int n_h_u, int n_x_l, int n_x_u, const int* h_l_idx, const int* h_u_idx, const int* x_l_idx, const int* x_u_idx, const double* x_b_scaling, int mem);

    // kkt_system.hpp:143-211 (vars: device pointers)
    bool update_scalings_and_factor(bool iterative_refinement, double rho, double delta, const pq_vars& vars);
    // kkt_system.hpp:213-369 (rhs/lhs: device pointers; lhs written in place)
    bool solve(const pq_vars& rhs, pq_vars& lhs);
    // kkt_system.hpp:392-425
    void mul(const pq_vars& lhs, pq_vars& rhs);
    void condensed_residual(const double* lhs_x, const double* lhs_y, double* res_inf, double* rhs_inf);

    KKTSolverBase* backend() { return kkt_solver; }
    hipStream_t stream() const { return st_; }
    int device() const { return dev_; }
    int n() const { return n_; }
    int p() const { return p_; }
    int m() const { return m_; }
    int n_x_l_count() const { return n_x_l; }
    int n_x_u_count() const { return n_x_u; }
    void set_settings(const pq_settings& s) { settings_ = s; }

    int last_refine_steps = 0, last_backend_solves = 0;
    double last_refine_error = 0.0, last_rhs_norm = 0.0;
    const double* last_rhs_y = nullptr;

private:
    void alloc();
    double read_scalar_max(int slot);
    double get_refine_error(const double* lhs_x, const double* lhs_y, const double* lhs_z, const double* rhs_x, const double* rhs_y, const double* rhs_z, double* err_x,
                            double* err_y, double* err_z);

    KKTSolverBase* kkt_solver;  // kkt_system.hpp:65
    pq_settings settings_;
    int n_ = 0, p_ = 0, m_ = 0, dev_ = 0;
    hipStream_t st_ = nullptr;
    int n_h_l = 0, n_h_u = 0, n_x_l = 0, n_x_u = 0;

    double m_rho = 0.0, m_delta = 0.0;  // kkt_system.hpp:32-33
    // kkt_system.hpp:37-63 (same names)
    DBuf<double> m_s_l, m_s_u, m_s_bl, m_s_bu, m_z_l_inv, m_z_u_inv, m_z_bl_inv, m_z_bu_inv;
    DBuf<double> m_x_reg, m_z_reg, rhs_x_bar, rhs_z_bar;
    DBuf<double> work_x, work_x2, work_x3, work_y, work_z, work_z2, lhs_z_buf;
    DBuf<double> ref_err_x, ref_err_y, ref_err_z, ref_lhs_x, ref_lhs_y, ref_lhs_z, rhs_y_keep;
    bool use_iterative_refinement = false;  // kkt_system.hpp:64
    bool finite_check_pending = false;

    // expanded index lists
    DBuf<int> has_l, has_u, pos_l, pos_u, h_l_idx, h_u_idx, x_l_idx, x_u_idx;
    DBuf<double> x_b_scaling;
    DBuf<unsigned long long> scal_d;
    HBuf<unsigned long long> scal_h;
};

}  // namespace pq
