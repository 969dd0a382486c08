// piqp_amd/csrc/solver.hpp -- host-side front end: piqp::DenseSolver / SparseSolver (reference solver.hpp)
// over the device-resident KKTSystem.  The interior-point loop is the CALLER of the hot path: it is
// restated here on the host so that iteration counts can be compared with the CPU path; every
// KKT factor / solve / mat-vec it issues goes to the GPU through pq::KKTSystem.
#pragma once

#include <memory>
#include <vector>

#include "kkt_system.hpp"
#include "ruiz_device.hpp"

namespace pq {

using Vec = std::vector<double>;
using IVec = std::vector<int>;

// compressed sparse column, int32 / fp64 (typedefs.hpp:53-54)
struct Csc {
    int rows = 0, cols = 0;
    IVec colptr, rowind;
    Vec val;
    int nnz() const { return colptr.empty() ? 0 : colptr[cols]; }
};

// dense::Data<T> (dense/data.hpp:22-208) and sparse::Data<T,I> (sparse/data.hpp:26-231) in one struct
struct HostData {
    bool sparse = false;
    int n = 0, p = 0, m = 0;
    Vec P_utri, AT, GT;  // dense, column-major (n x n, n x p, n x m)
    Csc sP_utri, sAT, sGT;
    Vec c, b, h_l, h_u, x_l, x_u, x_b_scaling;
    int n_h_l = 0, n_h_u = 0, n_x_l = 0, n_x_u = 0;
    IVec h_l_idx, h_u_idx, x_l_idx, x_u_idx;

    void resize_vectors();
    void set_h_l(const double* v);
    void set_h_u(const double* v);
    bool disable_inf_constraints(std::vector<int>* rows = nullptr);  // true if a row was disabled (its G row zeroed; the rows are appended to *rows)
    void set_x_l(const double* v);
    void set_x_u(const double* v);
    pq_dense_data dense_descriptor() const;
    pq_sparse_data sparse_descriptor() const;
};

// dense::RuizEquilibration / sparse::RuizEquilibration (dense/preconditioner.hpp, sparse/preconditioner.hpp)
struct Ruiz {
    int n = 0, p = 0, m = 0;
    double c = 1.0, c_inv = 1.0;
    Vec delta, delta_b, delta_inv, delta_b_inv;
    void init(const HostData& d);
    void scale_data(HostData& d, bool reuse_prev_scaling, bool scale_cost, int max_iter, double epsilon = 1e-3);
    void unscale_data(HostData& d);
    // the vector part of unscale_data / scale_data(reuse): updates that touch no matrix leave the scaled matrices alone
    void unscale_vectors(HostData& d) const;
    void scale_vectors(HostData& d) const;
    // b, h_l, h_u, x_l, x_u alone (the matrices, c and x_b_scaling are DeviceRuiz's)
    void scale_bounds(HostData& d) const;
    void unscale_bounds(HostData& d) const;
};

struct HostVars {
    Vec x, y, z_l, z_u, z_bl, z_bu, s_l, s_u, s_bl, s_bu;
    void resize(int n, int p, int m);
    Vec& field(int k);
    const Vec& field(int k) const { return const_cast<HostVars*>(this)->field(k); }
};

// the interior-point loop with device-resident vectors (device_ipm.hip); the default path of Solver::solve
class DeviceIpm {
public:
    DeviceIpm();
    ~DeviceIpm();
    void init(const HostData& d, const Ruiz& rz, hipStream_t st);
    void refresh_data(const HostData& d, const Ruiz& rz);  // after setup / update: scaled vectors, Ruiz scalings, finite-bound masks
    int solve(KKTSystem& kkt, const pq_settings& set, const Ruiz& rz, pq_info& info, double* trace, int trace_max, int* trace_rows);
    void download_result(HostVars& out);  // unscaled, expanded (unscale_results + restore_dual already applied)

private:
    struct Impl;
    std::unique_ptr<Impl> I;
};

class Solver {
public:
    explicit Solver(int device);
    ~Solver();
    Solver* clone() const;

    pq_settings& settings() { return m_settings; }
    bool setup(std::unique_ptr<HostData> data);  // solver.hpp:151-216
    bool update_dense(const double* P, const double* c, const double* A, const double* b, const double* G, const double* h_l, const double* h_u, const double* x_l,
                      const double* x_u);  // solver.hpp:218-308
    bool update_sparse(const int* Pp, const int* Pi, const double* Px, const double* c, const int* Ap, const int* Ai, const double* Ax, const double* b, const int* Gp,
                       const int* Gi, const double* Gx, const double* h_l, const double* h_u, const double* x_l, const double* x_u);
    int solve();  // solver.hpp:69-148
    const pq_info& info() const { return m_info; }
    const HostVars& result() const { return m_result; }
    const HostData* data() const { return m_data.get(); }
    void set_trace(double* buf, int max_rows) { trace_ = buf; trace_max_ = max_rows; trace_rows_ = 0; }
    int trace_rows() const { return trace_rows_; }
    int device() const { return device_; }
    KKTSolverBase* backend() { return m_kkt_system ? m_kkt_system->backend() : nullptr; }

private:
    int solve_impl();
    bool update_vectors_only(const double* c, const double* b, const double* h_l, const double* h_u, const double* x_l, const double* x_u, double t0);
    bool kkt_factor();
    void kkt_solve(const HostVars& rhs, HostVars& lhs);
    void eval_P_x(double alpha, const Vec& x, Vec& z);
    void eval_A(double an, double at, const Vec& xn, const Vec& xt, Vec& zn, Vec& zt);
    void eval_G(double an, double at, const Vec& xn, const Vec& xt, Vec& zn, Vec& zt);
    double calculate_mu() const;
    void calculate_step(double& alpha_s, double& alpha_z) const;
    void update_residuals_nr();
    void update_residuals_r();
    double primal_res_of(const HostVars& v) const;
    double dual_res_of(const Vec& x) const;
    double primal_prox_inf() const;
    double dual_prox_inf() const;
    void unscale_results();
    void restore_dual();
    void make_kkt();
    void scale_problem(bool reuse);
    void unscale_problem();
    void release_dense_staging();
    void stage_alloc();
    void to_device(const HostVars& h, pq_vars& d);
    void from_device(const pq_vars& d, HostVars& h);

    int device_;
    pq_settings m_settings;
    pq_info m_info{};
    std::unique_ptr<HostData> m_data;
    Ruiz m_preconditioner;
    std::unique_ptr<DeviceRuiz> druiz_;  // device side of m_preconditioner; null only under PIQP_AMD_DEBUG=host_ruiz
    std::unique_ptr<KKTSystem> m_kkt_system;
    bool m_first_run = true, m_setup_done = false, m_enable_iterative_refinement = false;
    HostVars m_result, res_nr, res, step, prox_vars;

    // device staging: two Variables sets + three work vectors per size class
    std::vector<DBuf<double>> dev_in_, dev_out_;
    pq_vars din_{}, dout_{};
    DBuf<double> dxa_, dxb_, dxc_, dya_, dyb_, dza_, dzb_;
    std::unique_ptr<DeviceIpm> dipm_;  // null when PIQP_AMD_HOST_IPM=1
    double* trace_ = nullptr;
    int trace_max_ = 0, trace_rows_ = 0;
};

bool verify_settings(const pq_settings& s);  // settings.hpp:84-106

std::unique_ptr<HostData> make_dense_host_data(int n, int p, int m, const double* P, const double* c, const double* A, const double* b, const double* G, const double* h_l,
                                               const double* h_u, const double* x_l, const double* x_u);
std::unique_ptr<HostData> make_sparse_host_data(int n, int p, int m, const int* Pp, const int* Pi, const double* Px, const double* c, const int* Ap, const int* Ai,
                                                const double* Ax, const double* b, const int* Gp, const int* Gi, const double* Gx, const double* h_l, const double* h_u,
                                                const double* x_l, const double* x_u);

}  // namespace pq
