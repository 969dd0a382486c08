// piqp_amd/csrc/ruiz_device.hpp -- Ruiz equilibration on the device (dense/preconditioner.hpp:62-258, sparse/preconditioner.hpp:65-290).
//
// Two shapes of the same arithmetic:
//   * sparse, batched: ONE workgroup per problem instance walks the shared CSC patterns; used by the batched solver (thousands of small
//     instances in one launch) and by the sparse Solver (one instance, one wide workgroup);
//   * dense: tiled multi-workgroup kernels over the column-major P_utri / AT / GT that stay resident in HBM (the dense Solver hands the
//     scaled device copies to the KKT backend; no scaled host mirror exists).
// Both reproduce the host routine Ruiz::scale_data of solver.cpp bit for bit: column / row inf-norms are max-reductions (exact in any order,
// done with integer atomic max on the bit patterns of the non-negative values), every product is applied in the host's order, the one sum
// (the cost scaling's mean column norm) is accumulated sequentially by one lane.
#pragma once

#include <memory>

#include "common.hpp"

namespace pq {

enum RuizMode { RUIZ_COMPUTE = 0, RUIZ_REUSE = 1, RUIZ_UNSCALE = 2 };

struct RuizSparseArgs {
    int n = 0, p = 0, m = 0;
    const int *Pp = nullptr, *Pi = nullptr, *ATp = nullptr, *ATi = nullptr, *GTp = nullptr, *GTi = nullptr;  // device patterns: P upper (n x n), AT (n x p), GT (n x m)
    long long stride = 0;                                                                                  // doubles between consecutive instances, every pointer below
    double *Px = nullptr, *ATx = nullptr, *GTx = nullptr, *c = nullptr, *xbs = nullptr;                    // scaled in place
    double *delta = nullptr, *delta_inv = nullptr, *delta_b = nullptr, *delta_b_inv = nullptr;             // n + p + m, n + p + m, n, n
    double* tmp = nullptr;                                                                                 // n doubles of scratch
    double *b = nullptr, *h_l = nullptr, *h_u = nullptr, *x_l = nullptr, *x_u = nullptr;                   // optional tail (preconditioner.hpp:208-221); x_l / x_u compressed
    const int *x_l_idx = nullptr, *x_u_idx = nullptr;
    int n_x_l = 0, n_x_u = 0;
    double* c_scale = nullptr;  // [batch], not strided: the cost scaling c of every instance (out for RUIZ_COMPUTE, in otherwise)
    int mode = RUIZ_COMPUTE, scale_cost = 0, max_iter = 10;
    double eps = 1e-3;
    unsigned long long* grid_ws = nullptr;  // optional, ruiz_grid_ws_words() words: with it a single instance (batch == 1) is equilibrated by a grid of workgroups
};
void launch_ruiz_sparse(const RuizSparseArgs& a, int batch, int threads, hipStream_t s);
size_t ruiz_grid_ws_words();

struct HostData;
struct Ruiz;

// Device side of the preconditioner of ONE Solver.  Dense problems: owns the (scaled) device matrices.  Sparse problems: owns device copies
// of the patterns and a value staging area; the scaled values return to the host mirror (the sparse backends take host CSC).
class DeviceRuiz {
public:
    DeviceRuiz(int device, const HostData& d);
    ~DeviceRuiz();
    DeviceRuiz(const DeviceRuiz&) = delete;
    DeviceRuiz& operator=(const DeviceRuiz&) = delete;
    std::unique_ptr<DeviceRuiz> clone() const;

    // dense: (re)load the unscaled matrices named by `options` (PQ_KKT_UPDATE_*) from the host staging copies in d
    void upload_dense(const HostData& d, int options);
    void zero_G_rows(const std::vector<int>& rows);  // dense: rows of G disabled by data.hpp:144-169 after the upload
    // scale_data / unscale_data of the matrices, c and x_b_scaling; the scalings land in rz, c / x_b_scaling in d (sparse: the matrix values too)
    void scale(HostData& d, Ruiz& rz, bool reuse_prev_scaling, bool scale_cost, int max_iter, double eps = 1e-3);
    void unscale(HostData& d, Ruiz& rz);
    pq_dense_data dense_descriptor(const HostData& d) const;  // device-resident matrices + host index lists

private:
    void run(HostData& d, Ruiz& rz, int mode, bool scale_cost, int max_iter, double eps);
    struct Impl;
    std::unique_ptr<Impl> I;
};

}  // namespace pq
