// piqp_amd/csrc/dense_kkt.cpp -- device-resident replacement of piqp::dense::KKT<T>
// (reference include/piqp/dense/kkt.hpp:24-180).  Same members, same call sequence; the bodies
// launch the gfx950 kernels of dense_kernels.hip on the handle's stream.
//
//   reference member           here
//   m_delta                    delta_
//   m_z_reg_inv                z_reg_inv_ (device, m)
//   kkt_mat + llt's copy       fac_ (device, n x n; assembly writes the lower triangle straight into the
//                              buffer that is then factored in place -- Eigen::LLT::compute makes that copy
//                              internally, dense/kkt.hpp:82)
//   AT_A                       ATA_ (device, n x n lower, only if p > 0)
//   W_delta_inv_G              not materialised: diag(z_reg_inv) is applied to the column operand while it is
//                              staged into LDS (saves the 134 MB write + read at n = m = 4096)
//   data.P_utri/AT/GT          Pfull_ (symmetric completion), AT_, GT_ (device copies, refreshed by update_data)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "trace.hpp"
#include "dense_kernels.hpp"
#include "kkt_solver_base.hpp"

namespace pq {

namespace {

class DenseKKT final : public KKTSolverBase {
public:
    DenseKKT(const pq_dense_data* d, int kkt_solver, int device) : dev_(device), n_(d->n), p_(d->p), m_(d->m), ldlt_(kkt_solver == PQ_DENSE_LDLT_NO_PIVOT)
    {
        if (n_ <= 0 || p_ < 0 || m_ < 0) throw std::runtime_error("dense KKT: bad dimensions");
        PQ_HIP(hipSetDevice(dev_));
        PQ_HIP(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
        alloc();
        upload(d);
    }

    ~DenseKKT() override
    {
        (void)hipSetDevice(dev_);
        if (st_inv_) { (void)hipStreamSynchronize(st_inv_); (void)hipStreamDestroy(st_inv_); }
        if (ev_fac_) (void)hipEventDestroy(ev_fac_);
        if (ev_inv_) (void)hipEventDestroy(ev_inv_);
        if (st_) { (void)hipStreamSynchronize(st_); (void)hipStreamDestroy(st_); }
    }

    // dense/kkt.hpp:57-60
    KKTSolverBase* clone() const override
    {
        PQ_HIP(hipSetDevice(dev_));
        const_cast<DenseKKT*>(this)->join_inverses();
        stream_wait(st_);
        DenseKKT* k = new DenseKKT(*this, 0);
        return k;
    }

    // dense/kkt.hpp:62-71.  The reference backend re-reads data.P_utri/AT/GT at every factor and solve, and
    // Solver::update rewrites them (unscale -> rescale) even when no matrix is passed, so every update_data
    // call refreshes all three device copies; AT_A is rebuilt as in :66-69.
    void update_data_dense(const pq_dense_data* d, int options) override
    {
        (void)options;
        if (d->n != n_ || d->p != p_ || d->m != m_) throw std::runtime_error("update_data: dimension mismatch");
        PQ_HIP(hipSetDevice(dev_));
        upload(d);
    }

    // dense/kkt.hpp:73-84
    bool update_scalings_and_factor(double delta, const double* x_reg, const double* z_reg) override
    {
        PQ_ZONE("piqp_amd::DenseKKT::update_scalings_and_factor");
        PQ_HIP(hipSetDevice(dev_));
        delta_ = delta;
        join_inverses();  // (a factorisation nobody solved with: its inverses still read the factor this one overwrites)
        dense::launch_reciprocal(m_, z_reg, z_reg_inv_.p, st_);
        PQ_HIP(hipMemcpyAsync(x_reg_last_.p, x_reg, sizeof(double) * n_, hipMemcpyDeviceToDevice, st_));
        int t0 = prof_.begin(0, st_);
        if (chol_fused_) assemble_first_block_column(x_reg_last_.p);  // (the other block columns are assembled inside the persistent launch)
        else update_kkt(x_reg_last_.p, fac_.p);
        prof_.end(0, t0, st_);
        int t1 = prof_.begin(1, st_);
        launch_factor_panels();
        // the double-double inverses of the 128-row diagonal blocks for the sweeps (dense_kernels.hip, k_block_inverse_dd): 4 x blocks workgroups, ~25 us
        prof_.end(1, t1, st_);
        // the inverses of the 128-row diagonal blocks for the sweeps: on a stream of their own, next to whatever follows the factorisation (the right-hand sides of
        // the first solve), joined where the first sweep starts
        if (inv_sweeps_) {
            PQ_HIP(hipEventRecord(ev_fac_, st_));
            PQ_HIP(hipStreamWaitEvent(st_inv_, ev_fac_, 0));
            dense::launch_block_inverse_dd(ldlt_, fac_.p, n_, n_, vinv_.p, st_inv_);
            PQ_HIP(hipEventRecord(ev_inv_, st_inv_));
            inv_pending_ = true;
        }
        return factor_status();
    }

    // dense/kkt.hpp:86-105
    void solve(const double* rhs_x, const double* rhs_y, const double* rhs_z, double* lhs_x, double* lhs_y, double* lhs_z) override
    {
        PQ_ZONE("piqp_amd::DenseKKT::solve");
        PQ_HIP(hipSetDevice(dev_));
        const int tk = prof_.begin(2, st_);
        const double delta_inv = 1.0 / delta_;
        // lhs_x = rhs_x + GT * (z_reg_inv o rhs_z) + delta_inv * AT * rhs_y
        int nsl = 0;
        if (m_ > 0) nsl += dense::launch_gemv_n_partial(n_, m_, GT_.p, n_, rhs_z, z_reg_inv_.p, 1.0, part_.p + (size_t)nsl * n_, st_);
        if (p_ > 0) nsl += dense::launch_gemv_n_partial(n_, p_, AT_.p, n_, rhs_y, nullptr, delta_inv, part_.p + (size_t)nsl * n_, st_);
        dense::launch_reduce_partials(n_, nsl, part_.p, rhs_x, lhs_x, st_);
        // solve_ldlt_in_place: llt.solveInPlace(lhs_x)
        join_inverses();
        { const int tt = prof_.begin(5, st_); dense::launch_trsv(fac_.p, n_, n_, lhs_x, rdiag_.p, ldlt_, ypoll_.p, flags_.p, w16_.p, st_, trsv_ts_.p, inv_sweeps_ ? vinv_.p : nullptr, next_xcd_seq());
          if (trsv_ts_.p) dump_trsv_ts(); prof_.end(5, tt, st_); }
        // lhs_y = delta_inv * AT^T lhs_x - delta_inv * rhs_y
        if (p_ > 0) dense::launch_gemv_t(n_, p_, AT_.p, n_, lhs_x, delta_inv, -delta_inv, rhs_y, nullptr, lhs_y, st_);
        // lhs_z = (GT^T lhs_x - rhs_z) o z_reg_inv
        if (m_ > 0) dense::launch_gemv_t(n_, m_, GT_.p, n_, lhs_x, 1.0, -1.0, rhs_z, z_reg_inv_.p, lhs_z, st_);
        prof_.end(2, tk, st_);
    }

    // dense/kkt.hpp:108-114
    void eval_P_x(double alpha, const double* x, double* z) override
    {
        PQ_ZONE("piqp_amd::DenseKKT::eval_P_x");
        PQ_HIP(hipSetDevice(dev_));
        dense::launch_gemv_t(n_, n_, Pfull_.p, n_, x, alpha, 0.0, nullptr, nullptr, z, st_);
    }

    // dense/kkt.hpp:117-123
    void eval_A_xn_and_AT_xt(double alpha_n, double alpha_t, const double* xn, const double* xt, double* zn, double* zt) override
    {
        PQ_HIP(hipSetDevice(dev_));
        if (p_ > 0) dense::launch_gemv_t(n_, p_, AT_.p, n_, xn, alpha_n, 0.0, nullptr, nullptr, zn, st_);
        int nsl = 0;
        if (p_ > 0) nsl = dense::launch_gemv_n_partial(n_, p_, AT_.p, n_, xt, nullptr, alpha_t, part_.p, st_);
        dense::launch_reduce_partials(n_, nsl, part_.p, nullptr, zt, st_);
    }

    // dense/kkt.hpp:126-132
    void eval_G_xn_and_GT_xt(double alpha_n, double alpha_t, const double* xn, const double* xt, double* zn, double* zt) override
    {
        PQ_HIP(hipSetDevice(dev_));
        if (m_ > 0) dense::launch_gemv_t(n_, m_, GT_.p, n_, xn, alpha_n, 0.0, nullptr, nullptr, zn, st_);
        int nsl = 0;
        if (m_ > 0) nsl = dense::launch_gemv_n_partial(n_, m_, GT_.p, n_, xt, nullptr, alpha_t, part_.p, st_);
        dense::launch_reduce_partials(n_, nsl, part_.p, nullptr, zt, st_);
    }

    void print_info() override {}

    const double* P_diag_device() const override { return Pdiag_.p; }
    int n() const override { return n_; }
    int p() const override { return p_; }
    int m() const override { return m_; }
    hipStream_t stream() const override { return st_; }
    int device() const override { return dev_; }

    // dense/kkt.hpp:134 internal_kkt_mat(): re-assembled on demand into a scratch buffer (test hook only)
    void internal_kkt_mat(double* out_host) override
    {
        PQ_HIP(hipSetDevice(dev_));
        DBuf<double> tmp((size_t)n_ * n_);
        tmp.zero(st_);
        update_kkt(x_reg_last_.p, tmp.p);
        PQ_HIP(hipMemcpyAsync(out_host, tmp.p, sizeof(double) * (size_t)n_ * n_, hipMemcpyDeviceToHost, st_));
        stream_wait(st_);
    }
    void set_profiling(int level) override { prof_.enabled = level != 0; prof_.level = level; }
    void get_profile(int stage, double* total_ms, int* count) override
    {
        if (stage < 0 || stage >= StageProfiler::NSTAGE) throw std::runtime_error("bad stage");
        PQ_HIP(hipSetDevice(dev_));
        prof_.collect(stage, st_, total_ms, count);
    }
    void set_class_failure_semantics(bool on) override { class_semantics_ = on; }
    bool factor_symmetric(const double* A_dev, int lda, bool from_lower) override
    {
        if (p_ != 0 || m_ != 0) throw std::runtime_error("factor_symmetric: a handle with p = m = 0 only");
        PQ_HIP(hipSetDevice(dev_));
        join_inverses();
        delta_ = 1.0;
        dense::launch_symmetrize(A_dev, lda, from_lower, n_, fac_.p, nullptr, st_);  // (the factorisation reads the lower triangle of fac_)
        const int t1 = prof_.begin(1, st_);
        launch_factor_panels();
        prof_.end(1, t1, st_);
        if (inv_sweeps_) {
            PQ_HIP(hipEventRecord(ev_fac_, st_));
            PQ_HIP(hipStreamWaitEvent(st_inv_, ev_fac_, 0));
            dense::launch_block_inverse_dd(ldlt_, fac_.p, n_, n_, vinv_.p, st_inv_);
            PQ_HIP(hipEventRecord(ev_inv_, st_inv_));
            inv_pending_ = true;
        }
        return factor_status();
    }
    void internal_factor(double* out_host) override
    {
        PQ_HIP(hipSetDevice(dev_));
        PQ_HIP(hipMemcpyAsync(out_host, fac_.p, sizeof(double) * (size_t)n_ * n_, hipMemcpyDeviceToHost, st_));
        stream_wait(st_);
    }

private:
    // copy-construction for clone(): same device, fresh stream, deep copies of all state
    DenseKKT(const DenseKKT& o, int) : dev_(o.dev_), n_(o.n_), p_(o.p_), m_(o.m_), ldlt_(o.ldlt_), class_semantics_(o.class_semantics_), delta_(o.delta_)
    {
        PQ_HIP(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
        alloc();
        auto cp = [&](DBuf<double>& dst, const DBuf<double>& src) { if (src.n) PQ_HIP(hipMemcpyAsync(dst.p, src.p, src.bytes(), hipMemcpyDeviceToDevice, st_)); };
        cp(Pfull_, o.Pfull_); cp(Pdiag_, o.Pdiag_); cp(AT_, o.AT_); cp(GT_, o.GT_); cp(ATA_, o.ATA_); cp(fac_, o.fac_);
        cp(GTp_, o.GTp_); cp(vinv_, o.vinv_);
        cp(z_reg_inv_, o.z_reg_inv_); cp(x_reg_last_, o.x_reg_last_); cp(rdiag_, o.rdiag_); cp(w16_, o.w16_);
        stream_wait(st_);
    }

    void alloc()
    {
        const size_t nn = (size_t)n_ * n_;
        Pfull_.alloc(nn); Pdiag_.alloc(n_);
        AT_.alloc((size_t)n_ * p_); GT_.alloc((size_t)n_ * m_);
        if (p_ > 0) ATA_.alloc(nn);
        fac_.alloc(nn);
        z_reg_inv_.alloc(m_); x_reg_last_.alloc(n_); dvec_.alloc(n_); rdiag_.alloc(n_);
        const int sl = dense::gemv_n_slices(n_, m_ > 0 ? m_ : 1) + dense::gemv_n_slices(n_, p_ > 0 ? p_ : 1) + dense::gemv_n_slices(n_, n_);
        part_.alloc((size_t)sl * n_);
        split_ws_.alloc(dense::syrk_split_workspace_doubles(n_, m_ > 0 ? m_ : 1));
        pack_.alloc(dense::FACTOR_PACK_DOUBLES);
        fuse_scratch_.alloc(dense::FACTOR_PACK_DOUBLES); fuse_flags_.alloc(16); fuse_flags_.zero(st_); fuse_cnt_.alloc(8); fuse_cnt_.zero(st_);
        w16_.alloc((size_t)((n_ + 127) / 128) * 8 * 256);  // inverted 16 x 16 diagonal pieces of the whole factor (potrf_block -> launch_trsv)
        dense::syrk_prepare(n_);
        info_.alloc(1);
        info_h_.alloc(2);
        info_h_.p[0] = -1; info_h_.p[1] = 0;
        // PIQP_AMD_DEBUG=chol_launches: the launch-per-panel factorisation for every size (the bitwise comparison of tests/test_dense_gpu.py)
        chol_persistent_ = !debug_token("chol_launches") && dense::chol_prepare(n_);
        if (chol_persistent_) { side_.alloc((size_t)n_ * n_); pack2_.alloc(2 * (size_t)dense::FACTOR_PACK_DOUBLES); chol_flags_.alloc(dense::chol_flag_ints(n_)); chol_flags_.zero(st_); }
        // round 4, EXPERIMENTAL (PIQP_AMD_DEBUG=chol_fused; default: assembly launch, then the factorisation): the assembly (dense/kkt.hpp:140-160) as tasks of the
        // persistent launch, overlapped with the factorisation.  Correct (tools/chk_chol_fused.py) and slower at n = 4096: see DESIGN.md section 4.
        if (chol_persistent_ && m_ > 0 && debug_token("chol_fused")) {
            const size_t d = dense::chol_prepare_fused(n_, m_);
            if (d > 0) {
                chol_fused_ = true;
                asm_part_.alloc(d);
                GTp_.alloc((size_t)n_ * m_);
                col0_ks_ = 8;
                col0_ws_.alloc((size_t)(n_ / dense::FACTOR_NB) * col0_ks_ * 128 * 128);
            }
        }
        flags_.alloc(3); flags_.zero(st_);  // (error word and tickets of the sweeps)
        one_xcd_ = (n_ + 127) / 128 <= 32 && !debug_token("no_one_xcd") && dense::probe_one_xcd_sweeps(st_);
        ypoll_.alloc(dense::trsv_poll_doubles(n_)); dense::launch_trsv_poll_init(ypoll_.p, n_, st_);
        // the diagonal step of the sweeps as one product with the inverse of the 128-row block (computed in double-double after every factorisation, rounded to
        // double), helper workgroups streaming the block rows: from eight block rows on (below that the sweeps are not what a solve waits for, and the small
        // problems of the parity suite keep the substitution's arithmetic).  PIQP_AMD_DEBUG=inv_sweeps=0 / =1: never / whenever the sweeps are persistent
        inv_sweeps_ = (n_ + 127) / 128 >= 8 && (n_ + 127) / 128 <= 224;
        if (const char* e = debug_token("inv_sweeps")) inv_sweeps_ = std::atoi(e) != 0 && (n_ + 127) / 128 <= 224;
        if (inv_sweeps_) {
            vinv_.alloc(dense::block_inverse_dd_doubles(n_));
            PQ_HIP(hipStreamCreateWithFlags(&st_inv_, hipStreamNonBlocking));
            PQ_HIP(hipEventCreateWithFlags(&ev_fac_, hipEventDisableTiming));
            PQ_HIP(hipEventCreateWithFlags(&ev_inv_, hipEventDisableTiming));
        }
        if (debug_token("trsv_ts")) { trsv_ts_.alloc(4 * ((n_ + 127) / 128) + 8); trsv_ts_.zero(st_); }
        if (const char* e = debug_token("fused_ts")) { dbg_panel_ = std::atoi(e); dbg_ts_.alloc(96); dbg_ts_.zero(st_); }
        x_reg_last_.zero(st_);
        fac_.zero(st_);
    }

    void upload(const pq_dense_data* d)
    {
        const size_t nn = (size_t)n_ * n_;
        // P_utri staged through the factor buffer, completed to the symmetric Pfull_ + diag(P)
        copy_in(fac_.p, d->P_utri, nn * sizeof(double), d->mem, st_);
        dense::launch_symmetrize_upper(fac_.p, n_, Pfull_.p, Pdiag_.p, st_);
        copy_in(AT_.p, d->AT, (size_t)n_ * p_ * sizeof(double), d->mem, st_);
        copy_in(GT_.p, d->GT, (size_t)n_ * m_ * sizeof(double), d->mem, st_);
        if (chol_fused_) dense::launch_pack_row_panels(GT_.p, n_, m_, GTp_.p, st_);
        if (p_ > 0) {
            // dense/kkt.hpp:53 AT_A.lower = AT * AT^T
            dense::SyrkArgs a;
            a.n = n_; a.kdim = p_; a.A = AT_.p; a.lda = n_; a.B = AT_.p; a.ldb = n_; a.C = ATA_.p; a.ldc = n_;
            dense::launch_syrk(dense::EPI_STORE, a, st_);
        }
        stream_wait(st_);  // host source buffers may be released by the caller after return
    }

    // dense/kkt.hpp:140-160 into the lower triangle of `out`
    void update_kkt(const double* x_reg, double* out)
    {
        const double dinv = 1.0 / delta_;
        if (m_ > 0) {
            dense::launch_syrk(dense::EPI_ASSEMBLE, assembly_args(x_reg, out), st_, split_ws_.p, split_ws_.n);
        } else {
            dense::launch_assemble_no_g(n_, Pfull_.p, x_reg, p_ > 0 ? ATA_.p : nullptr, dinv, out, st_);
        }
    }

    dense::SyrkArgs assembly_args(const double* x_reg, double* out) const
    {
        dense::SyrkArgs a;
        a.n = n_; a.kdim = m_; a.A = GT_.p; a.lda = n_; a.B = GT_.p; a.ldb = n_; a.w = z_reg_inv_.p; a.C = out; a.ldc = n_;
        a.Pfull = Pfull_.p; a.ldp = n_; a.x_reg = x_reg; a.ATA = p_ > 0 ? ATA_.p : nullptr; a.ldata = n_; a.dinv = 1.0 / delta_;
        return a;
    }
    void assemble_first_block_column(const double* x_reg) { dense::launch_syrk_first_col(assembly_args(x_reg, fac_.p), col0_ks_, col0_ws_.p, st_); }

    // blocked right-looking factorisation of the lower triangle of fac_ (panel width 128): Eigen::LLT::compute (dense/kkt.hpp:82) or
    // LDLTNoPivot::compute (dense/ldlt_no_pivot.hpp:313-354).  One launch per panel: the fused trailing update of panel k also factors the NEXT
    // diagonal block (workgroups 0-8) and solves the NEXT panel behind it (the workgroups of the first tile column), EPI_SUBTRACT_POTRF; only the
    // first diagonal block and the first panel have launches of their own.  (A look-ahead variant on a second, CU-masked stream was measured
    // slower in round 1 -- the cross-stream event latency exceeds the overlap: 4.1 -> 4.6 ms at n = 4096 -- and removed.)
    void launch_factor_panels()
    {
        PQ_HIP(hipMemsetAsync(info_.p, 0xFF, sizeof(int), st_));  // -1
        const int NB = dense::FACTOR_NB;
        if (chol_persistent_) {
            // first diagonal block and first panel by their own launches, every round after them in ONE persistent launch (k_chol_persistent)
            dense::launch_potrf_diag(ldlt_, fac_.p, n_, NB, 0, info_.p, rdiag_.p, dvec_.p, pack_.p, w16_.p, st_);
            { const int tt = prof_.begin(4, st_); dense::launch_trsm_panel(ldlt_, fac_.p, n_, 0, NB, n_, pack_.p, rdiag_.p, st_); prof_.end(4, tt, st_); }
            const int T = n_ / NB;
            if (chol_gen_ > 0x3fffffff - 2 * (T + 2)) { chol_flags_.zero(st_); chol_gen_ = 0; chol_fcount_ = 0; }  // (flag values are compared as signed differences)
            chol_gen_ += T + 2;
            const int tt = prof_.begin(3, st_);
            // (false = no task list for this size on the current device; chol_prepare() built it at create time, so this is a programming error, not a state
            // to continue from with only the first panel factored)
            dense::CholAssembly fa;
            if (chol_fused_) {
                fa.GT = GTp_.p; fa.ldg = 128; fa.m = m_; fa.zinv = z_reg_inv_.p; fa.Pfull = Pfull_.p; fa.ldp = n_; fa.x_reg = x_reg_last_.p;
                fa.ATA = p_ > 0 ? ATA_.p : nullptr; fa.ldata = n_; fa.dinv = 1.0 / delta_; fa.part = asm_part_.p;
            }
            if (!dense::launch_chol_persistent(ldlt_, fac_.p, side_.p, n_, n_, info_.p, rdiag_.p, dvec_.p, pack2_.p, w16_.p, fuse_scratch_.p, fuse_flags_.p, fuse_cnt_.p, fuse_token_, chol_flags_.p,
                                               chol_gen_, chol_fcount_, st_, chol_fused_ ? &fa : nullptr))
                throw std::runtime_error("dense factorisation: the persistent launch has no plan for this size / device");
            prof_.end(3, tt, st_);
            fuse_token_ += T - 1;
            ++chol_fcount_;
            return;
        }
        int p = 0;
        for (int k = 0; k < n_; k += NB, ++p) {
            const int nb = (n_ - k < NB) ? n_ - k : NB;
            const int rs = n_ - k - nb;
            double* A11 = fac_.p + k + (size_t)k * n_;
            if (k == 0) dense::launch_potrf_diag(ldlt_, A11, n_, nb, k, info_.p, rdiag_.p, dvec_.p + k, rs > 0 ? pack_.p : nullptr, w16_.p, st_);
            if (rs <= 0) break;
            // (the panels after the first are solved inside the previous fused launch, behind the factorisation of their diagonal block)
            if (k == 0) { const int tt = prof_.begin(4, st_); dense::launch_trsm_panel(ldlt_, fac_.p, n_, k, nb, n_, pack_.p, rdiag_.p, st_); prof_.end(4, tt, st_); }
            dense::SyrkArgs a;
            a.n = rs; a.kdim = nb;
            a.A = fac_.p + (k + nb) + (size_t)k * n_; a.lda = n_;
            a.B = a.A; a.ldb = n_;
            if (ldlt_) a.w = dvec_.p + k;  // D of this panel, written by its diagonal-block factorisation
            a.C = fac_.p + (k + nb) + (size_t)(k + nb) * n_; a.ldc = n_;
            a.fuse_nb = rs < NB ? rs : NB; a.fuse_kglobal = k + nb; a.fuse_ldlt = ldlt_ ? 1 : 0; a.fuse_info = info_.p; a.fuse_rdiag = rdiag_.p; a.fuse_dvec = dvec_.p + k + nb;
            a.fuse_pack = (rs - a.fuse_nb > 0) ? pack_.p : nullptr;
            a.fuse_w16 = w16_.p + (size_t)((k + nb) / 16) * 256;
            a.fuse_token = ++fuse_token_; a.fuse_flags = fuse_flags_.p; a.fuse_scratch = fuse_scratch_.p; a.fuse_cnt = fuse_cnt_.p;
            a.fuse_ts = (dbg_panel_ == p) ? dbg_ts_.p : nullptr;
            { const int tt = prof_.begin(3, st_); dense::launch_syrk(dense::EPI_SUBTRACT_POTRF, a, st_); prof_.end(3, tt, st_); }
            if (a.fuse_ts) dump_fused_ts(p);
        }
    }
    // PIQP_AMD_DEBUG=fused_ts=<panel>: in-kernel timeline of the workgroup that updates and factors the next diagonal block, to stderr
    void dump_fused_ts(int panel)
    {
        long long h[96];
        PQ_HIP(hipMemcpyAsync(h, dbg_ts_.p, sizeof(h), hipMemcpyDeviceToHost, st_));
        stream_wait(st_);
        std::fprintf(stderr, "[piqp_amd] fused panel %d (cycles): C + panel staged %lld, MFMA loop %lld, tiles->LDS %lld, potrf_block %lld; potrf steps (factor/subst/update):", panel, h[4] - h[0], h[1] - h[4], h[2] - h[1], h[3] - h[2]);
        for (int k = 0; k < 8; ++k) std::fprintf(stderr, " %lld/%lld/%lld", h[8 + 8 * k + 1] - h[8 + 8 * k], h[8 + 8 * k + 3] - h[8 + 8 * k + 2], h[8 + 8 * k + 5] - h[8 + 8 * k + 4]);
        std::fprintf(stderr, "\n[piqp_amd]   potrf step start / factored (since launch):");
        for (int k = 0; k < 8; ++k) std::fprintf(stderr, " %lld/%lld", h[8 + 8 * k] - h[0], h[8 + 8 * k + 1] - h[0]);
        std::fprintf(stderr, "; potrf_block end %lld\n[piqp_amd]   first panel workgroup: tile in registers %lld, step k operands seen:", h[3] - h[0], h[80] - h[0]);
        for (int k = 0; k < 8; ++k) std::fprintf(stderr, " %lld", h[72 + k] - h[0]);
        std::fprintf(stderr, ", panel stored %lld\n", h[81] - h[0]);
        std::fprintf(stderr, "[piqp_amd]   an ordinary tile (2, 1): first operand stage in LDS %lld, K loop (incl. the wait for C) %lld, stores drained %lld cycles\n", h[85] - h[84], h[86] - h[85],
                     h[87] - h[86]);
    }
    // llt.info() == Success (dense/kkt.hpp:83): one 4-byte read-back per factor call
    bool factor_status()
    {
        if (ldlt_ && !class_semantics_) dense::launch_flag_nonpositive(n_, rdiag_.p, info_.p, st_);  // (a negative pivot of this positive definite matrix is a breakdown: see k_flag_nonpositive)
        PQ_HIP(hipMemcpyAsync(info_h_.p, info_.p, sizeof(int), hipMemcpyDeviceToHost, st_));
        if (chol_persistent_) PQ_HIP(hipMemcpyAsync(info_h_.p + 1, chol_flags_.p + 1, sizeof(int), hipMemcpyDeviceToHost, st_));  // the launch's abort word
        stream_wait(st_);
        if (chol_persistent_ && info_h_.p[1] != 0) {
            // a bounded wait inside the persistent launch gave up (never seen on a healthy device): the cumulative counters of this handle are no longer
            // consistent -- start them over, and report the factorisation as failed (the caller regularises and factors again)
            chol_flags_.zero(st_); fuse_cnt_.zero(st_); fuse_flags_.zero(st_);
            chol_gen_ = 0; chol_fcount_ = 0; fuse_token_ = 0;
            stream_wait(st_);
            return false;
        }
        return info_h_.p[0] == -1;
    }

    int dev_, n_, p_, m_;
    bool ldlt_;
    bool class_semantics_ = false;  // see KKTSolverBase::set_class_failure_semantics
    double delta_ = 1.0;
    hipStream_t st_ = nullptr;
    DBuf<double> Pfull_, Pdiag_, AT_, GT_, ATA_, fac_, z_reg_inv_, x_reg_last_, dvec_, part_, rdiag_, split_ws_, pack_, w16_, fuse_scratch_;
    DBuf<int> info_, flags_, fuse_flags_, fuse_cnt_, chol_flags_;
    DBuf<double> pack2_, side_;  // side_: the solved panels once more, at addresses the persistent launch has never read before (dense_kernels.hip)
    bool inv_sweeps_ = false, inv_pending_ = false;
    hipStream_t st_inv_ = nullptr;
    hipEvent_t ev_fac_ = nullptr, ev_inv_ = nullptr;
    void join_inverses()
    {
        if (!inv_pending_) return;
        PQ_HIP(hipStreamWaitEvent(st_, ev_inv_, 0));
        inv_pending_ = false;
    }
    DBuf<double> vinv_;  // inverses of the 128-row diagonal blocks of the factor (dense_kernels.hip, launch_block_inverse_dd)
    bool chol_persistent_ = false, chol_fused_ = false;
    DBuf<double> GTp_;  // fused assembly: GT once more, as row panels of 128 rows (consecutive operand stages)
    DBuf<double> asm_part_, col0_ws_;  // fused assembly: partial sums of the K-sliced tiles; of block column 0's launch
    int col0_ks_ = 1;
    int chol_gen_ = 0, chol_fcount_ = 0;
    int fuse_token_ = 0;
    DBuf<double> ypoll_;  // the sweeps' hand-over buffers (dense_kernels.hip, launch_trsv)
    bool one_xcd_ = false;
    int xcd_seq_ = 0;
    int next_xcd_seq()
    {
        if (!one_xcd_) return -1;
        if (xcd_seq_ >= 50000000) { flags_.zero(st_); xcd_seq_ = 0; }
        return xcd_seq_++;
    }
    HBuf<int> info_h_;
    StageProfiler prof_;
    int dbg_panel_ = -1;
    DBuf<long long> dbg_ts_, trsv_ts_;
    // PIQP_AMD_DEBUG=trsv_ts: per block of the forward sweep, ticks of the device-wide 100 MHz clock (the compute units' own cycle counters are not
    // comparable with each other) from seeing the last producer's values to products done / diagonal block solved / handed over, the gaps between one block's
    // hand-over and the next block seeing it, and the hand-over times since block 1's
    void dump_trsv_ts()
    {
        std::vector<long long> h(trsv_ts_.n);
        PQ_HIP(hipMemcpyAsync(h.data(), trsv_ts_.p, trsv_ts_.bytes(), hipMemcpyDeviceToHost, st_));
        stream_wait(st_);
        std::fprintf(stderr, "[piqp_amd] forward sweep, ticks of the device-wide 100 MHz clock; per block (last producer's values seen -> products / -> solved / -> handed over):");
        const size_t nb = (h.size() - 8) / 4;
        for (size_t r = 1; r < nb; ++r) std::fprintf(stderr, " %lld/%lld/%lld", h[4 * r + 1] - h[4 * r], h[4 * r + 2] - h[4 * r], h[4 * r + 3] - h[4 * r]);
        std::fprintf(stderr, "\n[piqp_amd]   from one block's published to the next one's seen:");
        for (size_t r = 2; r < nb; ++r) std::fprintf(stderr, " %lld", h[4 * r] - h[4 * (r - 1) + 3]);
        std::fprintf(stderr, "\n[piqp_amd]   published at (since block 1's):");
        for (size_t r = 1; r < nb; ++r) std::fprintf(stderr, " %lld", h[4 * r + 3] - h[4 * 1 + 3]);
        if (h[4 * nb] != 0) {  // (the substitution form only)
            std::fprintf(stderr, "\n[piqp_amd]   block 1, groups of the diagonal step done at (since products):");
            for (int g = 0; g < 8; ++g) std::fprintf(stderr, " %lld", h[4 * nb + g] - h[4 * 1 + 1]);
        }
        std::fprintf(stderr, "\n");
    }
};

}  // namespace

KKTSolverBase* make_dense_kkt(const pq_dense_data* data, int kkt_solver, int device)
{
    if (kkt_solver != PQ_DENSE_CHOLESKY && kkt_solver != PQ_DENSE_LDLT_NO_PIVOT) throw std::runtime_error("kkt solver not supported");
    return new DenseKKT(data, kkt_solver, device);
}

}  // namespace pq
