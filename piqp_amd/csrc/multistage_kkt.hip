// piqp_amd/csrc/multistage_kkt.hip -- device-resident replacement of piqp::sparse::MultistageKKT<T,I>
// (reference include/piqp/sparse/multistage_kkt.hpp, KKTSolver::sparse_multistage).
//
//   reference (blasfeo calls, OpenMP over stages)               here
//   block_gemm_nd + block_syrk_ln_calc + populate_kkt_fac        k_ms_assemble: one launch, a workgroup (or several) per stage builds the
//     (:186-216, :832-994, :1008-1219)                             stage's whole frontal matrix  P + diag(x_reg) + delta^-1 AtA + X_G W X_G^T
//                                                                  (the sqrt scaling of GT is folded into the product as W = diag(1/z_reg))
//   block_syrk_ln_calc(AT, AT, AtA) at setup / update (:113,161)  k_ms_gram on the grouped equality rows (once per update_data)
//   factor_kkt (:1253-1352), serial recurrence over stages        k_ms_factor: ONE launch, one workgroup walks the chain; the front of the
//                                                                  current stage sits in LDS (when it fits), the Schur complement of stage i
//                                                                  (C_i C_i^T, F_i C_i^T, F_i F_i^T) is carried to stage i+1 / the arrow corner
//                                                                  inside LDS as the multifrontal update matrix
//   solve_llt_in_place (:1709-1816)                               k_ms_solve: ONE launch for the forward + backward sweep; the diagonal blocks
//                                                                  are applied through their explicit inverses (computed during the factorisation)
//                                                                  so every stage is two dense mat-vecs instead of a column-serial trsv
//   fold / recover in solve (:221-288), eval_* (:291-383)         CSC column dots of CscOperators (same products, caller's row order, so the
//                                                                  BlockVec permutations disappear)
// Like the reference, update_scalings_and_factor always reports success (:218): a non-positive pivot
// zeroes its column (blasfeo dpotrf semantics) and the caller's refinement loop notices.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>

#include "trace.hpp"
#include "kkt_solver_base.hpp"
#include "multistage_device.hpp"
#include "multistage_symbolic.hpp"
#include "sparse_ops.hpp"

namespace pq {

namespace {

constexpr int NT = 256;
constexpr int LDS_LIMIT_BYTES = 159 * 1024;  // gfx950: 160 KiB of LDS per workgroup
constexpr int ASM_CHUNK = 4096;              // front entries per assembly workgroup

using msdev::GroupMeta;
using msdev::MsMeta;

__global__ void k_ms_reciprocal(int m, const double* __restrict__ z, double* __restrict__ zinv)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) zinv[i] = 1.0 / z[i];
}

// grid (stage, chunk): lower triangle of X_b X_b^T for every stage (equality rows; the reference's AtA, :113 / :161)
__global__ __launch_bounds__(NT) void k_ms_gram(MsMeta M, GroupMeta Gm, const double* __restrict__ X, double* __restrict__ out)
{
    const int b = blockIdx.x;
    const int total = M.H(b) * M.H(b);
    const int lo = blockIdx.y * ASM_CHUNK, hi = min(total, lo + ASM_CHUNK);
    msdev::gram_stage<NT>(M, Gm, X, out, b, lo, hi);
}

// grid (block, chunk): frontal matrix of every stage + the arrow corner
__global__ __launch_bounds__(NT) void k_ms_assemble(MsMeta M, GroupMeta Gm, const double* __restrict__ XG, const double* __restrict__ Pf, const double* __restrict__ AtAf,
                                                    const double* __restrict__ zinv, const double* __restrict__ x_reg, double delta_inv, double* __restrict__ F)
{
    const int b = blockIdx.x;
    const int total = M.H(b) * M.H(b);
    const int lo = blockIdx.y * ASM_CHUNK, hi = min(total, lo + ASM_CHUNK);
    if (lo >= hi) return;
    msdev::assemble_stage<NT>(M, Gm, XG, Pf, AtAf, zinv, x_reg, delta_inv, F, b, lo, hi);
}

template <bool LDS>
__global__ __launch_bounds__(NT) void k_ms_factor(MsMeta M, double* __restrict__ fronts, double* __restrict__ pan, int fcap, int lofs, int li_in_lds)
{
    extern __shared__ double sm[];
    msdev::factor_chain<NT, LDS>(M, fronts, pan, sm, fcap, lofs, li_in_lds);
}

template <bool LDS>
__global__ __launch_bounds__(NT) void k_ms_solve(MsMeta M, const double* __restrict__ pan, double* __restrict__ x, int hcap)
{
    extern __shared__ double sm[];
    msdev::solve_chain<NT, LDS>(M, pan, x, sm, hcap);
}

template <class T>
void clone_buf(DBuf<T>& d, const DBuf<T>& s, hipStream_t st)
{
    d.alloc(s.n ? s.n : 1);
    if (s.n) PQ_HIP(hipMemcpyAsync(d.p, s.p, s.bytes(), hipMemcpyDeviceToDevice, st));
}

class MultistageKKT final : public KKTSolverBase {
public:
    MultistageKKT(const pq_sparse_data* d, int device) : dev_(device)
    {
        if (d->mem != PQ_MEM_HOST) throw std::runtime_error("sparse data must be host-resident");
        PQ_HIP(hipSetDevice(dev_));
        PQ_HIP(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
        multistage::analyse(d, S_);
        n_ = S_.n; p_ = S_.p; m_ = S_.m;
        plan_lds();
        // device copies of the symbolic analysis
        std::vector<int> start(S_.N);
        for (int b = 0; b < S_.N; ++b) start[b] = S_.block_info[b].start;
        upload_vec(w_, S_.w, st_); upload_vec(off_, S_.off, st_); upload_vec(h_, S_.h, st_); upload_vec(start_, start, st_);
        upload_vec(front_off_, S_.front_off, st_); upload_vec(pan_off_, S_.pan_off, st_);
        upload_vec(a_row_ptr_, S_.A.row_ptr, st_); upload_vec(a_rows_, S_.A.rows, st_); upload_vec(a_x_off_, S_.A.x_off, st_); upload_vec(a_dst_, S_.A.dst, st_);
        upload_vec(g_row_ptr_, S_.G.row_ptr, st_); upload_vec(g_rows_, S_.G.rows, st_); upload_vec(g_x_off_, S_.G.x_off, st_); upload_vec(g_dst_, S_.G.dst, st_);
        upload_vec(p_dst_, S_.P_dst, st_);
        auto arena = [&](DBuf<double>& b, long long cnt) { b.alloc(cnt > 0 ? (size_t)cnt : 1); b.zero(st_); };
        arena(Pf_, S_.front_doubles); arena(AtAf_, S_.front_doubles); arena(F_, S_.front_doubles); arena(pan_, S_.pan_doubles);
        arena(XA_, S_.A.x_doubles); arena(XG_, S_.G.x_doubles);
        zinv_.alloc(m_ ? m_ : 1);
        ops_.init(d, st_);
        scatter_values(KKT_ALL);
    }
    ~MultistageKKT() override
    {
        (void)hipSetDevice(dev_);
        if (st_) { (void)hipStreamSynchronize(st_); (void)hipStreamDestroy(st_); }
    }

    KKTSolverBase* clone() const override
    {
        PQ_HIP(hipSetDevice(dev_));
        stream_wait(st_);
        MultistageKKT* c = new MultistageKKT(*this, 0);
        if (tree_) c->tree_.reset(tree_->clone());
        return c;
    }

    // Stage-parallel elimination.  The serial recurrence of factor_kkt is the reference's algorithm, not a property of the
    // matrix: the condensed block-tridiagonal-arrow system can equally be eliminated along a nested-dissection tree
    // (segments of stages in parallel, separators last), which is what the multifrontal engine does for
    // KKTMode::KKT_ALL_ELIMINATED on exactly this matrix.  When a tree engine is attached, factor / solve go through it
    // (one stream: the engine runs on its own stream, so calls are bracketed by synchronisation points).
    void attach_tree_engine(KKTSolverBase* engine) { tree_.reset(engine); }
    bool uses_tree_engine() const { return (bool)tree_; }

    // multistage_kkt.hpp:142-178
    void update_data_sparse(const pq_sparse_data* d, int options) override
    {
        PQ_HIP(hipSetDevice(dev_));
        ops_.upload_values(d, st_);
        scatter_values(KKT_ALL);  // Solver::update rewrites all three matrices through unscale -> rescale; refresh everything stored
        if (tree_) tree_->update_data_sparse(d, options);
    }

    // multistage_kkt.hpp:180-219
    bool update_scalings_and_factor(double delta, const double* x_reg, const double* z_reg) override
    {
        PQ_ZONE("piqp_amd::MultistageKKT::update_scalings_and_factor");
        PQ_HIP(hipSetDevice(dev_));
        delta_ = delta;
        if (tree_) {
            stream_wait(st_);  // x_reg / z_reg were produced on this handle's stream
            const int t = prof_.begin(1, st_);
            tree_->update_scalings_and_factor(delta, x_reg, z_reg);
            stream_wait(tree_->stream());
            prof_.end(1, t, st_);
            return true;  // :218
        }
        const int t0 = prof_.begin(0, st_);
        if (m_ > 0) hipLaunchKernelGGL(k_ms_reciprocal, dim3((m_ + 255) / 256), dim3(256), 0, st_, m_, z_reg, zinv_.p);
        hipLaunchKernelGGL(k_ms_assemble, dim3(S_.N, asm_chunks_), dim3(NT), 0, st_, meta(), gmeta(), XG_.p, Pf_.p, AtAf_.p, zinv_.p, x_reg, 1.0 / delta, F_.p);
        prof_.end(0, t0, st_);
        const int t1 = prof_.begin(1, st_);
        if (factor_in_lds_) hipLaunchKernelGGL(k_ms_factor<true>, dim3(1), dim3(NT), factor_lds_bytes_, st_, meta(), F_.p, pan_.p, fcap_, fcap_ + ucap_, 1);
        else hipLaunchKernelGGL(k_ms_factor<false>, dim3(1), dim3(NT), factor_lds_bytes_, st_, meta(), F_.p, pan_.p, 0, 0, li_in_lds_ ? 1 : 0);
        prof_.end(1, t1, st_);
        PQ_HIP(hipGetLastError());
        return true;  // :218
    }

    // multistage_kkt.hpp:221-288
    void solve(const double* rhs_x, const double* rhs_y, const double* rhs_z, double* lhs_x, double* lhs_y, double* lhs_z) override
    {
        PQ_ZONE("piqp_amd::MultistageKKT::solve");
        PQ_HIP(hipSetDevice(dev_));
        if (tree_) {
            stream_wait(st_);
            const int t = prof_.begin(2, st_);
            tree_->solve(rhs_x, rhs_y, rhs_z, lhs_x, lhs_y, lhs_z);
            stream_wait(tree_->stream());
            prof_.end(2, t, st_);
            return;
        }
        const int tk = prof_.begin(2, st_);
        const double delta_inv = 1.0 / delta_;
        ops_.fold_rhs(rhs_x, rhs_y, rhs_z, zinv_.p, delta_inv, lhs_x, st_);
        if (solve_in_lds_) hipLaunchKernelGGL(k_ms_solve<true>, dim3(1), dim3(NT), solve_lds_bytes_, st_, meta(), pan_.p, lhs_x, hcap_);
        else hipLaunchKernelGGL(k_ms_solve<false>, dim3(1), dim3(NT), 2 * hcap_ * (int)sizeof(double), st_, meta(), pan_.p, lhs_x, hcap_);
        ops_.recover_duals(lhs_x, rhs_y, rhs_z, zinv_.p, delta_inv, lhs_y, lhs_z, st_);
        PQ_HIP(hipGetLastError());
        prof_.end(2, tk, st_);
    }

    void eval_P_x(double alpha, const double* x, double* z) override
    {
        PQ_ZONE("piqp_amd::MultistageKKT::eval_P_x");
        PQ_HIP(hipSetDevice(dev_));
        ops_.eval_P_x(alpha, x, z, st_);
    }
    void eval_A_xn_and_AT_xt(double alpha_n, double alpha_t, const double* xn, const double* xt, double* zn, double* zt) override
    {
        PQ_HIP(hipSetDevice(dev_));
        ops_.eval_A_xn_and_AT_xt(alpha_n, alpha_t, xn, xt, zn, zt, st_);
    }
    void eval_G_xn_and_GT_xt(double alpha_n, double alpha_t, const double* xn, const double* xt, double* zn, double* zt) override
    {
        PQ_HIP(hipSetDevice(dev_));
        ops_.eval_G_xn_and_GT_xt(alpha_n, alpha_t, xn, xt, zn, zt, st_);
    }

    int sparse_ordering(int* fill_perm, int* elim_perm) const override
    {
        if (!tree_) throw std::runtime_error("sparse_ordering: chain engine (see multistage_block_info)");
        return tree_->sparse_ordering(fill_perm, elim_perm);
    }
    void sparse_stats(double out[8]) const override
    {
        if (!tree_) throw std::runtime_error("sparse_stats: chain engine (see multistage_block_info)");
        tree_->sparse_stats(out);
    }
    // stage partition over several processes: the stage-parallel (tree) engine carries it; the serial recurrence cannot be split.
    // The engine is chosen by a symbolic cost model (make_multistage_kkt); partitioned runs may force it (PIQP_AMD_MULTISTAGE=tree) to keep
    // every rank on the same code path.
    void partition(int rank, int world, long long sizes[3]) override
    {
        if (!tree_) throw std::runtime_error("partition: sparse_multistage must run on the tree engine (set PIQP_AMD_MULTISTAGE=tree before setup)");
        tree_->partition(rank, world, sizes);
    }
    void set_exchange(pq_exchange_fn fn, void* user, double* buf_factor, double* buf_forward, double* buf_gather) override
    {
        if (!tree_) throw std::runtime_error("set_exchange: not partitioned");
        tree_->set_exchange(fn, user, buf_factor, buf_forward, buf_gather);
    }
    void set_comm_rccl(const unsigned char* id128, int rank, int world) override
    {
        if (!tree_) throw std::runtime_error("set_comm_rccl: not partitioned");
        tree_->set_comm_rccl(id128, rank, world);
    }
    double min_abs_pivot() override { if (!tree_) throw std::runtime_error("min_abs_pivot: chain engine"); return tree_->min_abs_pivot(); }
    void native_exchange_calls(int out[3]) const override { if (tree_) tree_->native_exchange_calls(out); else out[0] = out[1] = out[2] = 0; }
    void comm_info(int out[4]) const override { if (tree_) tree_->comm_info(out); else KKTSolverBase::comm_info(out); }
    void partition_info(int out[8]) const override
    {
        if (!tree_) throw std::runtime_error("partition_info: not partitioned");
        tree_->partition_info(out);
    }
    void sharded_calls(int out[2]) const override { if (tree_) tree_->sharded_calls(out); else out[0] = out[1] = 0; }  // (the tree engine's sharded value assembly)
    // SURVEY 8(e) row 2 (round 5): the tree engine's sharded refinement residual, partial fold / recovery and the gather of the multipliers -- this handle's stream
    // produced the operands, the tree engine's stream the results
    void set_exchange_norm(double* buf_norm) override
    {
        if (!tree_) throw std::runtime_error("set_exchange_norm: not partitioned");
        tree_->set_exchange_norm(buf_norm);
    }
    bool refine_error_sharded(const double* lhs_x, const double* lhs_y, const double* lhs_z, const double* rhs_x, const double* rhs_y, const double* rhs_z, const double* x_reg,
                              double delta, const double* z_reg, double* err_x, double* err_y, double* err_z, double* norm) override
    {
        if (!tree_) return false;
        stream_wait(st_);
        return tree_->refine_error_sharded(lhs_x, lhs_y, lhs_z, rhs_x, rhs_y, rhs_z, x_reg, delta, z_reg, err_x, err_y, err_z, norm);  // (ends with a read-back: complete)
    }
    void finish_sharded_solve(double* lhs_y, double* lhs_z) override
    {
        if (!tree_) return;
        stream_wait(st_);
        tree_->finish_sharded_solve(lhs_y, lhs_z);
        stream_wait(tree_->stream());
    }
    void sharded_solve_calls(int out[6]) const override { if (tree_) tree_->sharded_solve_calls(out); else KKTSolverBase::sharded_solve_calls(out); }

    // multistage_kkt.hpp:385-393 (same text), plus where the chain runs
    void print_info() override
    {
        std::printf("block sizes:");
        for (int b = 0; b + 1 < S_.N; ++b) std::printf(" %d,%d", S_.block_info[b].diag_size, S_.block_info[b].off_diag_size);
        std::printf("\narrow width: %d\n", S_.arrow);
        std::printf("multistage chain: %d stages, max front %d, factor %s, solve %s, front storage %.2f MB\n", S_.N - 1, S_.max_h, factor_in_lds_ ? "in LDS" : "in HBM",
                    solve_in_lds_ ? "in LDS" : "in HBM", S_.front_doubles * 8.0 / 1e6);
        if (tree_) { std::printf("elimination: stage-parallel (nested-dissection tree of the condensed system), engine: "); tree_->print_info(); }
        else std::printf("elimination: serial stage recurrence (factor_kkt order)\n");
    }

    const double* P_diag_device() const override { return ops_.P_diag(); }
    int n() const override { return n_; }
    int p() const override { return p_; }
    int m() const override { return m_; }
    hipStream_t stream() const override { return st_; }
    int device() const override { return dev_; }
    void set_profiling(int level) override { prof_.enabled = level != 0; }
    void get_profile(int stage, double* total_ms, int* count) override
    {
        if (stage < 0 || stage >= StageProfiler::NSTAGE) throw std::runtime_error("bad stage");
        PQ_HIP(hipSetDevice(dev_));
        prof_.collect(stage, st_, total_ms, count);
    }
    void multistage_block_info(std::vector<int>& out) const override
    {
        out.clear();
        for (const auto& b : S_.block_info) { out.push_back(b.start); out.push_back(b.diag_size); out.push_back(b.off_diag_size); }
    }

private:
    enum { KKT_ALL = 7 };

    MultistageKKT(const MultistageKKT& o, int)
        : dev_(o.dev_), n_(o.n_), p_(o.p_), m_(o.m_), delta_(o.delta_), S_(o.S_), factor_in_lds_(o.factor_in_lds_), li_in_lds_(o.li_in_lds_), solve_in_lds_(o.solve_in_lds_),
          factor_lds_bytes_(o.factor_lds_bytes_), solve_lds_bytes_(o.solve_lds_bytes_), fcap_(o.fcap_), ucap_(o.ucap_), pcap_(o.pcap_), hcap_(o.hcap_), asm_chunks_(o.asm_chunks_)
    {
        PQ_HIP(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
        ops_.clone_from(o.ops_, st_);
        clone_buf(w_, o.w_, st_); clone_buf(off_, o.off_, st_); clone_buf(h_, o.h_, st_); clone_buf(start_, o.start_, st_);
        clone_buf(front_off_, o.front_off_, st_); clone_buf(pan_off_, o.pan_off_, st_);
        clone_buf(a_row_ptr_, o.a_row_ptr_, st_); clone_buf(a_rows_, o.a_rows_, st_); clone_buf(a_x_off_, o.a_x_off_, st_); clone_buf(a_dst_, o.a_dst_, st_);
        clone_buf(g_row_ptr_, o.g_row_ptr_, st_); clone_buf(g_rows_, o.g_rows_, st_); clone_buf(g_x_off_, o.g_x_off_, st_); clone_buf(g_dst_, o.g_dst_, st_);
        clone_buf(p_dst_, o.p_dst_, st_);
        clone_buf(Pf_, o.Pf_, st_); clone_buf(AtAf_, o.AtAf_, st_); clone_buf(F_, o.F_, st_); clone_buf(pan_, o.pan_, st_);
        clone_buf(XA_, o.XA_, st_); clone_buf(XG_, o.XG_, st_); clone_buf(zinv_, o.zinv_, st_);
        stream_wait(st_);
    }

    MsMeta meta() const { return MsMeta{S_.N, S_.arrow, n_, w_.p, off_.p, h_.p, start_.p, front_off_.p, pan_off_.p}; }
    GroupMeta gmeta() const { return GroupMeta{g_row_ptr_.p, g_rows_.p, g_x_off_.p}; }
    GroupMeta ameta() const { return GroupMeta{a_row_ptr_.p, a_rows_.p, a_x_off_.p}; }

    // where the chain kernels keep their working set
    void plan_lds()
    {
        int max_u = 0;
        long long max_pan = 0;
        for (int b = 0; b < S_.N; ++b) {
            max_u = std::max(max_u, S_.h[b] - S_.w[b]);
            max_pan = std::max(max_pan, (long long)S_.h[b] * S_.w[b] + (long long)S_.w[b] * S_.w[b]);
        }
        fcap_ = S_.max_h * S_.max_h;
        ucap_ = max_u * max_u;
        const long long lbytes = (long long)S_.max_w * S_.max_w * (long long)sizeof(double);
        const long long fbytes = ((long long)fcap_ + ucap_) * (long long)sizeof(double) + lbytes;
        factor_in_lds_ = fbytes <= LDS_LIMIT_BYTES;
        li_in_lds_ = factor_in_lds_ || lbytes <= LDS_LIMIT_BYTES;
        factor_lds_bytes_ = factor_in_lds_ ? (int)fbytes : (li_in_lds_ ? (int)lbytes : 0);
        hcap_ = std::max(1, S_.max_h);
        pcap_ = (int)std::min<long long>(max_pan, 1 << 30);
        const long long sbytes = (2LL * hcap_ + max_pan) * (long long)sizeof(double);
        solve_in_lds_ = sbytes <= LDS_LIMIT_BYTES;
        solve_lds_bytes_ = solve_in_lds_ ? (int)sbytes : 0;
        if (2LL * hcap_ * (long long)sizeof(double) > LDS_LIMIT_BYTES) throw std::runtime_error("multistage: a stage is too wide for this backend");
        asm_chunks_ = std::max(1, (fcap_ + ASM_CHUNK - 1) / ASM_CHUNK);
        static PerDeviceOnce attr_set;
        attr_set([&] {
            PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ms_factor<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT_BYTES));
            PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ms_factor<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT_BYTES));
            PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ms_solve<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT_BYTES));
            PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ms_solve<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT_BYTES));
        });
    }

    // caller's CSC values -> front / grouped-row arenas (utri_to_kkt :599-670, transpose_to_block_mat :672-818), then AtA (:161)
    void scatter_values(int options)
    {
        if (options & 1) launch_remap_values64(ops_.nzP(), p_dst_.p, ops_.P_x(), Pf_.p, st_);
        if (options & 2) {
            launch_remap_values64(ops_.nzA(), a_dst_.p, ops_.AT_x(), XA_.p, st_);
            hipLaunchKernelGGL(k_ms_gram, dim3(S_.N - 1, asm_chunks_), dim3(NT), 0, st_, meta(), ameta(), XA_.p, AtAf_.p);
        }
        if (options & 4) launch_remap_values64(ops_.nzG(), g_dst_.p, ops_.GT_x(), XG_.p, st_);
        PQ_HIP(hipGetLastError());
        stream_wait(st_);
    }

    int dev_, n_ = 0, p_ = 0, m_ = 0;
    double delta_ = 1.0;
    hipStream_t st_ = nullptr;
    multistage::Symbolic S_;
    bool factor_in_lds_ = true, li_in_lds_ = true, solve_in_lds_ = true;
    int factor_lds_bytes_ = 0, solve_lds_bytes_ = 0, fcap_ = 0, ucap_ = 0, pcap_ = 0, hcap_ = 1, asm_chunks_ = 1;
    CscOperators ops_;
    DBuf<int> w_, off_, h_, start_, a_row_ptr_, a_rows_, g_row_ptr_, g_rows_;
    DBuf<long long> front_off_, pan_off_, a_x_off_, g_x_off_, a_dst_, g_dst_, p_dst_;
    DBuf<double> Pf_, AtAf_, F_, pan_, XA_, XG_, zinv_;
    StageProfiler prof_;
    std::unique_ptr<KKTSolverBase> tree_;
};

// MultistageKKT ctor (multistage_kkt.hpp:76-135).  Chains of 16+ stages get both elimination orders analysed and a SYMBOLIC cost
// model picks one (PIQP_AMD_MULTISTAGE=chain|tree forces it).  The choice depends on the sparsity structure only, so two processes
// (and two boxes) always run the same arithmetic on the same input.  The model is microseconds per IPM iteration's backend work
// (1 factorisation + 2 substitutions), fitted once on an MI355X to tools/calib_multistage.py (27 structures, fixtures and synthetic
// chains, profiles/r02_calib_multistage.jsonl); it ranks the engines correctly on 26 of them (the 27th costs 14 %):
//   chain (one workgroup walks the recurrence):  20 + sum_i (0.9 w_i + 2.6e-4 h_i^2 w_i)  +  2 sum_i (2.2 + 0.005 h_i w_i)
//   tree  (critical path = levels of the assembly tree, c = nnz(L) / N the mean column count):
//         70 + levels (1 + 0.35 c + 0.005 c^2) + flops / 2.5e5          +  2 levels (3.5 + 0.3 c)
// The tree engine is taken when it is at least 10 % cheaper.
double chain_cost_us(const std::vector<int>& bi)
{
    const int stages = (int)bi.size() / 3 - 1;
    const double arrow = (double)bi[3 * stages + 1];
    double fac = 20.0, sol = 0.0;
    for (int i = 0; i < stages; ++i) {
        const double w = bi[3 * i + 1], h = w + bi[3 * i + 2] + arrow;
        fac += 0.9 * w + 2.6e-4 * h * h * w;
        sol += 2.2 + 0.005 * h * w;
    }
    return fac + 2.0 * sol;
}
double tree_cost_us(const double st[8])
{
    const double N = st[0] > 0 ? st[0] : 1.0, c = st[2] / N, levels = st[4], flops = st[7];
    return 70.0 + levels * (1.0 + 0.35 * c + 0.005 * c * c) + flops / 2.5e5 + 2.0 * levels * (3.5 + 0.3 * c);
}

}  // namespace

KKTSolverBase* make_multistage_kkt(const pq_sparse_data* data, int device)
{
    std::unique_ptr<MultistageKKT> ms(new MultistageKKT(data, device));
    const char* env = std::getenv("PIQP_AMD_MULTISTAGE");
    const std::string want = env ? env : "auto";
    std::vector<int> bi;
    ms->multistage_block_info(bi);
    const int stages = (int)bi.size() / 3 - 1;
    if (want == "chain" || (want == "auto" && stages < 16)) return ms.release();
    std::unique_ptr<KKTSolverBase> tree(make_multifrontal_kkt(data, 3, device));  // (KKT_ALL_ELIMINATED on the multifrontal engine: the tree engine of this backend)
    if (!tree) return ms.release();
    if (want != "tree") {
        double st[8];
        tree->sparse_stats(st);
        if (tree_cost_us(st) > 0.9 * chain_cost_us(bi)) return ms.release();
    }
    ms->attach_tree_engine(tree.release());
    return ms.release();
}

}  // namespace pq
