// piqp_amd/csrc/capi.cpp -- the extern "C" boundary declared in include/piqp_amd.h.
// Thin: argument checks, host<->device staging for PQ_MEM_HOST callers, exception fencing.
#include <algorithm>
#include <chrono>
#include <vector>
#include <memory>
#include <stdexcept>

#include "dense_kernels.hpp"
#include "kkt_solver_base.hpp"
#include "rccl_transport.hpp"
#include "kkt_system.hpp"
#include "sparse_symbolic.hpp"
#include "solver.hpp"

using namespace pq;

namespace {

struct VarSizes {
    int n, p, m;
    int size(int k) const
    {
        switch (k) { case 0: return n; case 1: return p; case 2: case 3: return m; case 4: case 5: return n; case 6: case 7: return m; default: return n; }
    }
};
inline double** var_field(pq_vars& v, int k)
{
    switch (k) { case 0: return &v.x; case 1: return &v.y; case 2: return &v.z_l; case 3: return &v.z_u; case 4: return &v.z_bl; case 5: return &v.z_bu;
                 case 6: return &v.s_l; case 7: return &v.s_u; case 8: return &v.s_bl; default: return &v.s_bu; }
}
inline double* const* var_field(const pq_vars& v, int k) { return var_field(const_cast<pq_vars&>(v), k); }

// device staging for one Variables set
struct VarStage {
    DBuf<double> buf[10];
    pq_vars v{};
    void alloc(VarSizes s)
    {
        for (int k = 0; k < 10; ++k) { buf[k].alloc(s.size(k) > 0 ? s.size(k) : 1); *var_field(v, k) = buf[k].p; }
    }
    void zero(hipStream_t st) { for (int k = 0; k < 10; ++k) buf[k].zero(st); }
};

}  // namespace

struct pq_kkt {
    KKTSolverBase* impl = nullptr;
    bool owned = true;
    int ptr_mode = PQ_MEM_HOST;
    // staging (host pointer mode)
    DBuf<double> sx, sy, sz, lx, ly, lz, xr, zr;
    bool staged = false;
    void ensure_staging()
    {
        if (staged) return;
        const int n = impl->n(), p = impl->p(), m = impl->m();
        sx.alloc(n); sy.alloc(p > 0 ? p : 1); sz.alloc(m > 0 ? m : 1);
        lx.alloc(n); ly.alloc(p > 0 ? p : 1); lz.alloc(m > 0 ? m : 1);
        xr.alloc(n); zr.alloc(m > 0 ? m : 1);
        staged = true;
    }
    ~pq_kkt() { if (owned) delete impl; }
};

struct pq_kktsys {
    KKTSystem* impl = nullptr;
    pq_kkt backend_view;
    int ptr_mode = PQ_MEM_HOST;
    VarStage in, out, mul_in, mul_out;  // mul_*: scratch Variables of pq_kktsys_mul (the last solve's staged vectors stay intact)
    bool staged = false;
    const double* last_lhs_x = nullptr;
    const double* last_lhs_y = nullptr;
    void ensure_staging()
    {
        if (staged) return;
        VarSizes s{impl->n(), impl->p(), impl->m()};
        in.alloc(s); out.alloc(s); mul_in.alloc(s); mul_out.alloc(s);
        in.zero(impl->stream()); out.zero(impl->stream());
        staged = true;
    }
    ~pq_kktsys() { delete impl; }
};

static void h2d(double* dst, const double* src, int n, hipStream_t st)
{
    if (n > 0) PQ_HIP(hipMemcpyAsync(dst, src, sizeof(double) * n, hipMemcpyHostToDevice, st));
}
static void d2h(double* dst, const double* src, int n, hipStream_t st)
{
    if (n > 0) PQ_HIP(hipMemcpyAsync(dst, src, sizeof(double) * n, hipMemcpyDeviceToHost, st));
}

extern "C" {

void pq_settings_default(pq_settings* s)
{
    // settings.hpp:45-82
    s->rho_init = 1e-6; s->delta_init = 1e-4;
    s->eps_abs = 1e-8; s->eps_rel = 1e-9;
    s->check_duality_gap = 1; s->eps_duality_gap_abs = 1e-8; s->eps_duality_gap_rel = 1e-9;
    s->infeasibility_threshold = 0.9;
    s->reg_lower_limit = 1e-10; s->reg_finetune_lower_limit = 1e-13;
    s->reg_finetune_primal_update_threshold = 7; s->reg_finetune_dual_update_threshold = 7;
    s->max_iter = 250; s->max_factor_retires = 10;
    s->preconditioner_scale_cost = 0; s->preconditioner_reuse_on_update = 0; s->preconditioner_iter = 10;
    s->tau = 0.99;
    s->kkt_solver = PQ_DENSE_CHOLESKY;
    s->iterative_refinement_always_enabled = 0;
    s->iterative_refinement_eps_abs = 1e-12; s->iterative_refinement_eps_rel = 1e-12;
    s->iterative_refinement_max_iter = 10;
    s->iterative_refinement_min_improvement_rate = 5.0;
    s->iterative_refinement_static_regularization_eps = 1e-8;
    s->iterative_refinement_static_regularization_rel = 2.220446049250313e-16 * 2.220446049250313e-16;
    s->verbose = 0; s->compute_timings = 0;
}

const char* pq_last_error_string(void) { return last_error().c_str(); }
const char* pq_version(void) { return "piqp_amd 0.1 (gfx950; KKT path of PIQP v0.6.2)"; }

int pq_device_count(void)
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

static int check_device(int device)
{
    int c = pq_device_count();
    if (c <= 0) return fail(PQ_ERR_HIP, "no HIP device visible: this library has no CPU fallback");
    if (device < 0 || device >= c) return fail(PQ_ERR_INVALID, "device %d out of range (%d visible)", device, c);
    return PQ_OK;
}

// ------------------------------------------------------------------------------------ backend
int pq_kkt_create_dense(pq_kkt** out, const pq_dense_data* data, int kkt_solver, int device)
{
    if (!out || !data) return fail(PQ_ERR_INVALID, "null argument");
    *out = nullptr;
    if (kkt_solver != PQ_DENSE_CHOLESKY && kkt_solver != PQ_DENSE_LDLT_NO_PIVOT) return fail(PQ_ERR_UNSUPPORTED, "kkt solver not supported");
    int rc = check_device(device);
    if (rc < 0) return rc;
    return guarded([&] {
        std::unique_ptr<pq_kkt> h(new pq_kkt);
        h->impl = make_dense_kkt(data, kkt_solver, device);
        h->ensure_staging();  // host-pointer-mode staging: allocated here, never later (pq_debug_alloc_count)
        *out = h.release();
        return (int)PQ_OK;
    });
}

int pq_kkt_create_sparse(pq_kkt** out, const pq_sparse_data* data, int kkt_solver, int device)
{
    if (!out || !data) return fail(PQ_ERR_INVALID, "null argument");
    *out = nullptr;
    int rc = check_device(device);
    if (rc < 0) return rc;
    return guarded([&] {
        std::unique_ptr<pq_kkt> h(new pq_kkt);
        h->impl = make_sparse_kkt(data, kkt_solver, device);
        if (!h->impl) return fail(PQ_ERR_UNSUPPORTED, "kkt solver not supported");
        h->ensure_staging();
        *out = h.release();
        return (int)PQ_OK;
    });
}

int pq_kkt_clone(const pq_kkt* k, pq_kkt** out)
{
    if (!k || !out) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] {
        std::unique_ptr<pq_kkt> h(new pq_kkt);
        h->impl = k->impl->clone();
        h->ptr_mode = k->ptr_mode;
        h->ensure_staging();
        *out = h.release();
        return (int)PQ_OK;
    });
}

void pq_kkt_destroy(pq_kkt* k)
{
    if (k && k->owned) delete k;
}

int pq_kkt_set_pointer_mode(pq_kkt* k, int mem)
{
    if (!k || (mem != PQ_MEM_HOST && mem != PQ_MEM_DEVICE)) return fail(PQ_ERR_INVALID, "bad pointer mode");
    k->ptr_mode = mem;
    return PQ_OK;
}

int pq_kkt_update_data_dense(pq_kkt* k, const pq_dense_data* data, int options)
{
    if (!k || !data) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { k->impl->update_data_dense(data, options); return (int)PQ_OK; });
}
int pq_kkt_update_data_sparse(pq_kkt* k, const pq_sparse_data* data, int options)
{
    if (!k || !data) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { k->impl->update_data_sparse(data, options); return (int)PQ_OK; });
}

int pq_kkt_update_scalings_and_factor(pq_kkt* k, double delta, const double* x_reg, const double* z_reg)
{
    if (!k || !x_reg || (k->impl->m() > 0 && !z_reg)) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] {
        PQ_HIP(hipSetDevice(k->impl->device()));
        hipStream_t st = k->impl->stream();
        if (k->ptr_mode == PQ_MEM_HOST) {
            k->ensure_staging();
            h2d(k->xr.p, x_reg, k->impl->n(), st);
            h2d(k->zr.p, z_reg, k->impl->m(), st);
            return k->impl->update_scalings_and_factor(delta, k->xr.p, k->zr.p) ? 1 : 0;
        }
        return k->impl->update_scalings_and_factor(delta, x_reg, z_reg) ? 1 : 0;
    });
}

int pq_kkt_solve(pq_kkt* k, const double* rhs_x, const double* rhs_y, const double* rhs_z, double* lhs_x, double* lhs_y, double* lhs_z)
{
    if (!k || !rhs_x || !lhs_x) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] {
        PQ_HIP(hipSetDevice(k->impl->device()));
        hipStream_t st = k->impl->stream();
        const int n = k->impl->n(), p = k->impl->p(), m = k->impl->m();
        if (k->ptr_mode == PQ_MEM_HOST) {
            k->ensure_staging();
            h2d(k->sx.p, rhs_x, n, st); h2d(k->sy.p, rhs_y, p, st); h2d(k->sz.p, rhs_z, m, st);
            k->impl->solve(k->sx.p, k->sy.p, k->sz.p, k->lx.p, k->ly.p, k->lz.p);
            d2h(lhs_x, k->lx.p, n, st); d2h(lhs_y, k->ly.p, p, st); d2h(lhs_z, k->lz.p, m, st);
            stream_wait(st);
        } else {
            k->impl->solve(rhs_x, rhs_y, rhs_z, lhs_x, lhs_y, lhs_z);
        }
        return (int)PQ_OK;
    });
}

int pq_kkt_eval_P_x(pq_kkt* k, double alpha, const double* x, double* z)
{
    if (!k || !x || !z) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] {
        PQ_HIP(hipSetDevice(k->impl->device()));
        hipStream_t st = k->impl->stream();
        const int n = k->impl->n();
        if (k->ptr_mode == PQ_MEM_HOST) {
            k->ensure_staging();
            h2d(k->sx.p, x, n, st);
            k->impl->eval_P_x(alpha, k->sx.p, k->lx.p);
            d2h(z, k->lx.p, n, st);
            stream_wait(st);
        } else {
            k->impl->eval_P_x(alpha, x, z);
        }
        return (int)PQ_OK;
    });
}

static int eval_pair(pq_kkt* k, bool isG, double an, double at, const double* xn, const double* xt, double* zn, double* zt)
{
    if (!k) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] {
        PQ_HIP(hipSetDevice(k->impl->device()));
        hipStream_t st = k->impl->stream();
        const int n = k->impl->n(), q = isG ? k->impl->m() : k->impl->p();
        if (k->ptr_mode == PQ_MEM_HOST) {
            k->ensure_staging();
            double* sq = isG ? k->sz.p : k->sy.p;
            double* lq = isG ? k->lz.p : k->ly.p;
            h2d(k->sx.p, xn, n, st); h2d(sq, xt, q, st);
            if (isG) k->impl->eval_G_xn_and_GT_xt(an, at, k->sx.p, sq, lq, k->lx.p);
            else k->impl->eval_A_xn_and_AT_xt(an, at, k->sx.p, sq, lq, k->lx.p);
            d2h(zn, lq, q, st); d2h(zt, k->lx.p, n, st);
            stream_wait(st);
        } else {
            if (isG) k->impl->eval_G_xn_and_GT_xt(an, at, xn, xt, zn, zt);
            else k->impl->eval_A_xn_and_AT_xt(an, at, xn, xt, zn, zt);
        }
        return (int)PQ_OK;
    });
}

int pq_kkt_eval_A_xn_and_AT_xt(pq_kkt* k, double alpha_n, double alpha_t, const double* xn, const double* xt, double* zn, double* zt)
{
    return eval_pair(k, false, alpha_n, alpha_t, xn, xt, zn, zt);
}
int pq_kkt_eval_G_xn_and_GT_xt(pq_kkt* k, double alpha_n, double alpha_t, const double* xn, const double* xt, double* zn, double* zt)
{
    return eval_pair(k, true, alpha_n, alpha_t, xn, xt, zn, zt);
}

int pq_kkt_print_info(pq_kkt* k)
{
    if (!k) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { k->impl->print_info(); return (int)PQ_OK; });
}
int pq_kkt_synchronize(pq_kkt* k)
{
    if (!k) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { PQ_HIP(hipSetDevice(k->impl->device())); stream_wait(k->impl->stream()); return (int)PQ_OK; });
}
void* pq_kkt_stream(pq_kkt* k) { return k ? (void*)k->impl->stream() : nullptr; }
int pq_kkt_internal_kkt_mat(pq_kkt* k, double* out_host)
{
    if (!k || !out_host) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { k->impl->internal_kkt_mat(out_host); return (int)PQ_OK; });
}
int pq_kkt_internal_factor(pq_kkt* k, double* out_host)
{
    if (!k || !out_host) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { k->impl->internal_factor(out_host); return (int)PQ_OK; });
}
// ---------------------------------------------------------------------------------------------------------------------------------------------------
// The reference's dense factorisation CLASSES as objects of their own: piqp::dense::LDLTNoPivot<Mat, UpLo> (dense/ldlt_no_pivot.hpp:87-262) and the Eigen::LLT
// dense/kkt.hpp:82 uses, for either triangle -- what benchmarks/src/dense_cholesky_factorization_benchmark.cpp and tests/src/dense/ldlt_test.cpp exercise.  The
// arithmetic is the dense backend's (a backend with p = m = 0, zero regularisation: its assembly is P + 0); the Upper variants work on the transposed view exactly as
// ldlt_no_pivot.hpp:357-371 does (Transpose<MatrixType>), so U = L^T bit for bit.
struct pq_dense_factor {
    std::unique_ptr<KKTSolverBase> impl;
    int device = 0, n = 0, kind = PQ_DENSE_LDLT_NO_PIVOT, uplo = PQ_LOWER, info = 1;
    DBuf<double> stage, full, zero, xb;
    std::vector<double> host;
    double last_ms[2] = {0.0, 0.0};
};

int pq_dense_factor_create(pq_dense_factor** out, int device, int n, int kind, int uplo)
{
    if (!out) return fail(PQ_ERR_INVALID, "null argument");
    if (n <= 0 || (kind != PQ_DENSE_LDLT_NO_PIVOT && kind != PQ_DENSE_CHOLESKY) || (uplo != PQ_LOWER && uplo != PQ_UPPER)) return fail(PQ_ERR_INVALID, "dense factor: bad n / kind / uplo");
    return guarded([&] {
        PQ_HIP(hipSetDevice(device));
        std::unique_ptr<pq_dense_factor> f(new pq_dense_factor);
        f->device = device; f->n = n; f->kind = kind; f->uplo = uplo;
        const size_t nn = (size_t)n * n;
        f->stage.alloc(nn); f->full.alloc(nn); f->zero.alloc(n); f->xb.alloc(n);
        f->zero.zero(nullptr);
        PQ_HIP(hipDeviceSynchronize());
        *out = f.release();
        return (int)PQ_OK;
    });
}
void pq_dense_factor_destroy(pq_dense_factor* f) { delete f; }

// LDLTNoPivot::compute (ldlt_no_pivot.hpp:393-423) / Eigen::LLT::compute: reads the `uplo` triangle of A (column-major, leading dimension lda >= n; mem says where A
// lives).  Returns info(): 0 = Eigen::Success, 1 = Eigen::NumericalIssue (:231, :418-420).
int pq_dense_factor_compute(pq_dense_factor* f, const double* A, int lda, int mem)
{
    if (!f || !A) return fail(PQ_ERR_INVALID, "null argument");
    if (lda < f->n) return fail(PQ_ERR_INVALID, "dense factor: lda < n");
    int info = 1;
    const int rc = guarded([&] {
        PQ_HIP(hipSetDevice(f->device));
        const int n = f->n;
        const auto w0 = std::chrono::steady_clock::now();
        if (!f->impl) {
            // the backend object behind this one: a dense KKT backend with p = m = 0 (its P is never used: factor_symmetric writes the factor buffer directly)
            f->full.zero(nullptr);
            PQ_HIP(hipStreamSynchronize(nullptr));
            pq_dense_data d{};
            d.n = n; d.p = 0; d.m = 0; d.P_utri = f->full.p; d.mem = PQ_MEM_DEVICE;
            f->impl.reset(make_dense_kkt(&d, f->kind, f->device));
            f->impl->set_class_failure_semantics(true);
            f->impl->set_profiling(1);
        }
        hipStream_t st = f->impl->stream();
        const double* src = A;
        int ld = lda;
        if (mem != PQ_MEM_DEVICE) {  // a host matrix: one copy to the device (the leading dimension kept out of it)
            PQ_HIP(hipMemcpy2DAsync(f->stage.p, sizeof(double) * n, A, sizeof(double) * lda, sizeof(double) * n, n, hipMemcpyHostToDevice, st));
            src = f->stage.p; ld = n;
        }
        double before = 0.0; int cnt = 0;
        f->impl->get_profile(1, &before, &cnt);
        const bool ok = f->impl->factor_symmetric(src, ld, f->uplo == PQ_LOWER);
        double after = 0.0;
        f->impl->get_profile(1, &after, &cnt);
        f->last_ms[0] = after - before;
        f->last_ms[1] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
        info = ok ? 0 : 1;
        f->info = info;
        return (int)PQ_OK;
    });
    return rc == PQ_OK ? info : rc;
}
int pq_dense_factor_info(const pq_dense_factor* f) { return f ? f->info : fail(PQ_ERR_INVALID, "null argument"); }

// solveInPlace (ldlt_no_pivot.hpp:432-450: L, D, U sweeps / Eigen::LLT::solveInPlace), one right-hand side of n entries
int pq_dense_factor_solve_in_place(pq_dense_factor* f, double* x, int mem)
{
    if (!f || !x) return fail(PQ_ERR_INVALID, "null argument");
    if (!f->impl) return fail(PQ_ERR_INVALID, "dense factor: solve before compute");
    return guarded([&] {
        PQ_HIP(hipSetDevice(f->device));
        hipStream_t st = f->impl->stream();
        double* xd = x;
        if (mem != PQ_MEM_DEVICE) { copy_in(f->xb.p, x, sizeof(double) * f->n, PQ_MEM_HOST, st); xd = f->xb.p; }
        f->impl->solve(xd, nullptr, nullptr, xd, nullptr, nullptr);
        if (mem != PQ_MEM_DEVICE) PQ_HIP(hipMemcpyAsync(x, xd, sizeof(double) * f->n, hipMemcpyDeviceToHost, st));
        stream_wait(st);
        return (int)PQ_OK;
    });
}

// matrixLDLT() (ldlt_no_pivot.hpp:217) / Eigen::LLT::matrixLLT(): the `uplo` triangle of out (host, column-major, leading dimension ldo) receives the factor -- the
// strictly lower part of unit L with D on the diagonal, resp. L; for PQ_UPPER the transposes (U, D) -- the other triangle is left alone.
int pq_dense_factor_matrix(pq_dense_factor* f, double* out_host, int ldo)
{
    if (!f || !out_host) return fail(PQ_ERR_INVALID, "null argument");
    if (!f->impl) return fail(PQ_ERR_INVALID, "dense factor: no factorisation yet");
    if (ldo < f->n) return fail(PQ_ERR_INVALID, "dense factor: ldo < n");
    return guarded([&] {
        const int n = f->n;
        f->host.resize((size_t)n * n);
        f->impl->internal_factor(f->host.data());
        for (int j = 0; j < n; ++j)
            for (int i = j; i < n; ++i) {
                const double v = f->host[i + (size_t)j * n];
                if (f->uplo == PQ_LOWER) out_host[i + (size_t)j * ldo] = v;
                else out_host[j + (size_t)i * ldo] = v;
            }
        return (int)PQ_OK;
    });
}

// device time of the factorisation launches of the last compute() (hipEvents) and the wall time of the whole call, ms
int pq_dense_factor_last_ms(const pq_dense_factor* f, double out2[2])
{
    if (!f || !out2) return fail(PQ_ERR_INVALID, "null argument");
    out2[0] = f->last_ms[0]; out2[1] = f->last_ms[1];
    return PQ_OK;
}

int pq_kkt_set_profiling(pq_kkt* k, int enable)
{
    if (!k) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { k->impl->set_profiling(enable); return (int)PQ_OK; });
}
int pq_kkt_get_profile(pq_kkt* k, int stage, double* total_ms, int* count)
{
    if (!k || !total_ms || !count) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { k->impl->get_profile(stage, total_ms, count); return (int)PQ_OK; });
}
int pq_kkt_multistage_block_info(pq_kkt* k, int* out_host, int capacity)
{
    if (!k) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] {
        std::vector<int> bi;
        k->impl->multistage_block_info(bi);
        const int N = (int)bi.size() / 3;
        if (out_host) for (int i = 0; i < 3 * std::min(N, capacity); ++i) out_host[i] = bi[i];
        return N;
    });
}
int pq_kkt_sparse_stats(pq_kkt* k, double out[8])
{
    if (!k || !out) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { k->impl->sparse_stats(out); return (int)PQ_OK; });
}
int pq_kkt_partition(pq_kkt* k, int rank, int world, long long sizes_out[3])
{
    if (!k || !sizes_out) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { k->impl->partition(rank, world, sizes_out); return (int)PQ_OK; });
}
int pq_kkt_set_exchange(pq_kkt* k, pq_exchange_fn exchange, void* user, double* buf_factor, double* buf_forward, double* buf_gather)
{
    if (!k) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { k->impl->set_exchange(exchange, user, buf_factor, buf_forward, buf_gather); return (int)PQ_OK; });
}
int pq_rccl_unique_id(unsigned char out[128])
{
    if (!out) return fail(PQ_ERR_INVALID, "null output");
    return guarded([&] { rccl::unique_id(out); return (int)PQ_OK; });
}
int pq_kkt_set_comm_rccl(pq_kkt* k, const unsigned char id[128], int rank, int world)
{
    if (!k || !id) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { k->impl->set_comm_rccl(id, rank, world); return (int)PQ_OK; });
}
int pq_kkt_min_abs_pivot(pq_kkt* k, double* out)
{
    if (!k || !out) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { *out = k->impl->min_abs_pivot(); return (int)PQ_OK; });
}
long long pq_kkt_exact_factor(pq_kkt* k, int what, void* out_host)
{
    if (!k) return fail(PQ_ERR_INVALID, "null handle");
    long long r = -1;
    const int rc = guarded([&] { r = k->impl->exact_factor(what, out_host); return (int)PQ_OK; });
    return rc < 0 ? rc : r;
}
int pq_kkt_comm_info(pq_kkt* k, int out[4])
{
    if (!k || !out) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { k->impl->comm_info(out); return (int)PQ_OK; });
}
int pq_kkt_set_exchange_norm(pq_kkt* k, double* buf_norm)
{
    if (!k) return fail(PQ_ERR_INVALID, "null handle");
    return guarded([&] { k->impl->set_exchange_norm(buf_norm); return (int)PQ_OK; });
}
int pq_kkt_sharded_calls(pq_kkt* k, int out[2])
{
    if (!k || !out) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { k->impl->sharded_calls(out); return (int)PQ_OK; });
}
int pq_kkt_sharded_solve_calls(pq_kkt* k, int out[6])
{
    if (!k || !out) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { k->impl->sharded_solve_calls(out); return (int)PQ_OK; });
}
int pq_kkt_native_exchange_calls(pq_kkt* k, int out[3])
{
    if (!k || !out) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { k->impl->native_exchange_calls(out); return (int)PQ_OK; });
}
int pq_kkt_partition_info(pq_kkt* k, int out[8])
{
    if (!k || !out) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { k->impl->partition_info(out); return (int)PQ_OK; });
}
int pq_sparse_partition_plan(const pq_sparse_data* data, int mode, int world, int* owner_out, int capacity, double* work_out)
{
    if (!data || world < 1) return fail(PQ_ERR_INVALID, "bad argument");
    return guarded([&] {
        sparse::Symbolic S;
        sparse::analyse_kkt(data, mode, S);
        sparse::Partition P;
        sparse::partition_tree(S, world, P);
        if (owner_out)
            for (int s = 0; s < S.nsuper; ++s)
                for (int j = S.sn_first[s]; j < S.sn_first[s + 1] && j < capacity; ++j) owner_out[j] = P.owner[s];
        if (work_out) { for (int r = 0; r < world; ++r) work_out[r] = P.work[r]; work_out[world] = P.shared_work; }
        return S.N;
    });
}
int pq_sparse_amd_order(int n, const int* Ap, const int* Ai, int* perm)
{
    if (n < 0 || !Ap || (n && !perm)) return fail(PQ_ERR_INVALID, "bad argument");
    return guarded([&] { sparse::amd_order(n, Ap, Ai, perm); return (int)PQ_OK; });
}
int pq_sparse_permute_sym_upper(int n, const int* Ap, const int* Ai, const int* perm_inv, int* Cp, int* Ci, int* Ai_to_Ci)
{
    if (n < 0 || !Ap || !perm_inv || !Cp) return fail(PQ_ERR_INVALID, "bad argument");
    return guarded([&] {
        sparse::IVec ap(Ap, Ap + n + 1), ai(Ai, Ai + Ap[n]), cp, ci, map;
        sparse::permute_sym_upper(n, ap, ai, perm_inv, cp, ci, map);
        std::copy(cp.begin(), cp.end(), Cp);
        if (Ci) std::copy(ci.begin(), ci.end(), Ci);
        if (Ai_to_Ci) std::copy(map.begin(), map.end(), Ai_to_Ci);
        return (int)PQ_OK;
    });
}
int pq_sparse_kkt_symbolic(const pq_sparse_data* data, int mode, int* nnz_out, int* Kp, int* Ki, int* perm, int* PKp, int* PKi_rows, int* PKi)
{
    if (!data || mode < 0 || mode > 3) return fail(PQ_ERR_INVALID, "bad argument");
    return guarded([&] {
        // analyse_kkt builds K exactly like create_kkt_matrix of the mode; the ordering / permutation steps are repeated here in the reference's own
        // sequence (sparse/kkt.hpp:57-63: ordering.init(KKT); PKi = permute_sparse_symmetric_matrix(KKT, PKPt, ordering)) without the device schedule's
        // postorder, which is what pq_kkt_sparse_ordering reports
        sparse::Symbolic S;
        sparse::analyse_kkt_pattern(data, mode, S);
        const int N = S.N, nnz = S.Kp[N];
        if (nnz_out) *nnz_out = nnz;
        if (Kp) std::copy(S.Kp.begin(), S.Kp.end(), Kp);
        if (Ki) std::copy(S.Ki.begin(), S.Ki.end(), Ki);
        if (perm || PKp || PKi_rows || PKi) {
            sparse::IVec pm(N), pinv(N), cp, ci, map;
            sparse::amd_order(N, S.Kp.data(), S.Ki.data(), pm.data());
            for (int i = 0; i < N; ++i) pinv[pm[i]] = i;
            sparse::permute_sym_upper(N, S.Kp, S.Ki, pinv.data(), cp, ci, map);
            if (perm) std::copy(pm.begin(), pm.end(), perm);
            if (PKp) std::copy(cp.begin(), cp.end(), PKp);
            if (PKi_rows) std::copy(ci.begin(), ci.end(), PKi_rows);
            if (PKi) std::copy(map.begin(), map.end(), PKi);
        }
        return N;
    });
}
int pq_sparse_uplooking_plan(const pq_sparse_data* data, int nitems, const int* what, void** out, long long* len)
{
    if (!data || nitems < 0 || (nitems && (!what || !len))) return fail(PQ_ERR_INVALID, "bad argument");
    return guarded([&] {
        sparse::Symbolic S;
        sparse::analyse_kkt_pattern(data, 0, S);
        sparse::UpLooking U;
        sparse::analyse_uplooking(S, data, U);
        const sparse::IVec stats = {(int)U.nnzL, (int)(U.task_ptr.size() - 1), U.height, (int)std::min<long long>(U.crit_steps, 2147483647LL), U.Cp[U.N], (int)U.tk_kind.size()};
        sparse::IVec mode_stats;
        for (int q = 0; q < nitems; ++q) {
            const sparse::IVec* v = nullptr;
            if (what[q] >= 100 && what[q] <= 103) {
                // the plan of another KKT mode (what - 100: bit 0 equalities eliminated, bit 1 inequalities eliminated): {N, nnz(L), kiloflops of the factorisation}
                sparse::Symbolic Sm;
                sparse::analyse_kkt_pattern(data, what[q] - 100, Sm);
                sparse::UpLooking Um;
                sparse::analyse_uplooking(Sm, data, Um);
                mode_stats = {Um.N, (int)std::min<long long>(Um.nnzL, 2147483647LL), (int)std::min(Um.flops / 1e3, 2147483647.0)};
                len[q] = 3;
                if (out && out[q]) std::copy(mode_stats.begin(), mode_stats.end(), (int*)out[q]);
                continue;
            }
            switch (what[q]) {
            case 0: v = &stats; break;
            case 1: v = &U.perm; break;
            case 2: v = &U.Cp; break;
            case 3: v = &U.Ci; break;
            case 4: v = &U.diag_pos; break;
            case 5: v = &U.etree; break;
            case 6: v = &U.Lp; break;
            case 7: v = &U.Li; break;
            case 8: v = &U.Rp; break;
            case 9: v = &U.Rcol; break;
            case 10: v = &U.Rpos; break;
            case 11: v = &U.task_ptr; break;
            case 12: v = &U.task_rows; break;
            case 13: v = &U.row_task; break;
            case 14: v = &U.row_lane; break;
            case 15: v = &U.row_prev; break;
            case 16: v = &U.dep_ptr; break;
            case 17: v = &U.dep; break;
            case 18: v = &U.tk_kind; break;
            case 19: v = &U.tk_id; break;
            case 20: v = &U.Rcnt; break;
            case 21: v = &U.Rtab; break;
            case 22: v = &U.tab_ptr; break;
            case 23: v = &U.mask_ptr; break;
            case 24: v = &U.task_nU; break;
            case 25:
                len[q] = (long long)U.Tmask.size();
                if (out && out[q]) std::copy(U.Tmask.begin(), U.Tmask.end(), (unsigned long long*)out[q]);
                continue;
            default: throw std::runtime_error("uplooking plan: unknown item");
            }
            len[q] = (long long)v->size();
            if (out && out[q]) std::copy(v->begin(), v->end(), (int*)out[q]);
        }
        return U.N;
    });
}
int pq_kkt_sparse_ordering(pq_kkt* k, int* fill_perm, int* elim_perm)
{
    if (!k) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { return k->impl->sparse_ordering(fill_perm, elim_perm); });
}
int pq_kkt_dims(const pq_kkt* k, int* n, int* p, int* m)
{
    if (!k) return fail(PQ_ERR_INVALID, "null argument");
    if (n) *n = k->impl->n();
    if (p) *p = k->impl->p();
    if (m) *m = k->impl->m();
    return PQ_OK;
}

// ------------------------------------------------------------------------------------ KKTSystem
static pq_kktsys* wrap_kktsys(KKTSystem* sys)
{
    pq_kktsys* h = new pq_kktsys;
    h->impl = sys;
    h->backend_view.impl = sys->backend();
    h->backend_view.owned = false;
    h->ensure_staging();
    h->backend_view.ensure_staging();
    return h;
}

int pq_kktsys_create_dense(pq_kktsys** out, const pq_dense_data* data, const pq_settings* settings, int device)
{
    if (!out || !data || !settings) return fail(PQ_ERR_INVALID, "null argument");
    *out = nullptr;
    // KKTSystem::init_kkt_solver<PIQP_DENSE>, kkt_system.hpp:455-468
    if (settings->kkt_solver != PQ_DENSE_CHOLESKY && settings->kkt_solver != PQ_DENSE_LDLT_NO_PIVOT) return fail(PQ_ERR_UNSUPPORTED, "kkt solver not supported");
    int rc = check_device(device);
    if (rc < 0) return rc;
    return guarded([&] {
        std::unique_ptr<KKTSolverBase> b(make_dense_kkt(data, settings->kkt_solver, device));
        std::unique_ptr<KKTSystem> sys(new KKTSystem(b.get(), *settings));
        b.release();
        sys->set_bounds(data->n_h_l, data->n_h_u, data->n_x_l, data->n_x_u, data->h_l_idx, data->h_u_idx, data->x_l_idx, data->x_u_idx, data->x_b_scaling, data->mem);
        *out = wrap_kktsys(sys.release());
        return (int)PQ_OK;
    });
}

int pq_kktsys_create_sparse(pq_kktsys** out, const pq_sparse_data* data, const pq_settings* settings, int device)
{
    if (!out || !data || !settings) return fail(PQ_ERR_INVALID, "null argument");
    *out = nullptr;
    int rc = check_device(device);
    if (rc < 0) return rc;
    return guarded([&] {
        std::unique_ptr<KKTSolverBase> b(make_sparse_kkt(data, settings->kkt_solver, device));
        if (!b) return fail(PQ_ERR_UNSUPPORTED, "kkt solver not supported");
        std::unique_ptr<KKTSystem> sys(new KKTSystem(b.get(), *settings));
        b.release();
        sys->set_bounds(data->n_h_l, data->n_h_u, data->n_x_l, data->n_x_u, data->h_l_idx, data->h_u_idx, data->x_l_idx, data->x_u_idx, data->x_b_scaling, data->mem);
        *out = wrap_kktsys(sys.release());
        return (int)PQ_OK;
    });
}

int pq_kktsys_clone(const pq_kktsys* k, pq_kktsys** out)
{
    if (!k || !out) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] {
        pq_kktsys* h = wrap_kktsys(k->impl->clone());
        h->ptr_mode = k->ptr_mode;
        *out = h;
        return (int)PQ_OK;
    });
}

void pq_kktsys_destroy(pq_kktsys* k) { delete k; }

int pq_kktsys_set_pointer_mode(pq_kktsys* k, int mem)
{
    if (!k || (mem != PQ_MEM_HOST && mem != PQ_MEM_DEVICE)) return fail(PQ_ERR_INVALID, "bad pointer mode");
    k->ptr_mode = mem;
    k->backend_view.ptr_mode = mem;
    return PQ_OK;
}

pq_kkt* pq_kktsys_backend(pq_kktsys* k) { return k ? &k->backend_view : nullptr; }

int pq_kktsys_update_data_dense(pq_kktsys* k, const pq_dense_data* data, int options)
{
    if (!k || !data) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] {
        // kkt_system.hpp:134-141 (P_diag is refreshed by the backend's upload)
        k->impl->backend()->update_data_dense(data, options);
        k->impl->set_bounds(data->n_h_l, data->n_h_u, data->n_x_l, data->n_x_u, data->h_l_idx, data->h_u_idx, data->x_l_idx, data->x_u_idx, data->x_b_scaling, data->mem);
        return (int)PQ_OK;
    });
}
int pq_kktsys_update_data_sparse(pq_kktsys* k, const pq_sparse_data* data, int options)
{
    if (!k || !data) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] {
        k->impl->backend()->update_data_sparse(data, options);
        k->impl->set_bounds(data->n_h_l, data->n_h_u, data->n_x_l, data->n_x_u, data->h_l_idx, data->h_u_idx, data->x_l_idx, data->x_u_idx, data->x_b_scaling, data->mem);
        return (int)PQ_OK;
    });
}

// copy the used part of a host Variables set into the staging set
static void stage_in(pq_kktsys* k, const pq_vars* src, VarStage& dst)
{
    hipStream_t st = k->impl->stream();
    VarSizes s{k->impl->n(), k->impl->p(), k->impl->m()};
    for (int f = 0; f < 10; ++f) {
        const double* hp = *var_field(*src, f);
        if (hp) h2d(dst.buf[f].p, hp, s.size(f), st);
    }
}
static void stage_out(pq_kktsys* k, VarStage& src, pq_vars* dst)
{
    hipStream_t st = k->impl->stream();
    VarSizes s{k->impl->n(), k->impl->p(), k->impl->m()};
    for (int f = 0; f < 10; ++f) {
        double* hp = *var_field(*dst, f);
        if (hp) d2h(hp, src.buf[f].p, s.size(f), st);
    }
    stream_wait(st);
}

int pq_kktsys_update_scalings_and_factor(pq_kktsys* k, int iterative_refinement, double rho, double delta, const pq_vars* vars)
{
    if (!k || !vars) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] {
        PQ_HIP(hipSetDevice(k->impl->device()));
        if (k->ptr_mode == PQ_MEM_HOST) {
            k->ensure_staging();
            stage_in(k, vars, k->in);
            return k->impl->update_scalings_and_factor(iterative_refinement != 0, rho, delta, k->in.v) ? 1 : 0;
        }
        return k->impl->update_scalings_and_factor(iterative_refinement != 0, rho, delta, *vars) ? 1 : 0;
    });
}

int pq_kktsys_solve(pq_kktsys* k, const pq_vars* rhs, pq_vars* lhs)
{
    if (!k || !rhs || !lhs) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] {
        PQ_HIP(hipSetDevice(k->impl->device()));
        bool ok;
        if (k->ptr_mode == PQ_MEM_HOST) {
            k->ensure_staging();
            stage_in(k, rhs, k->in);
            ok = k->impl->solve(k->in.v, k->out.v);
            k->last_lhs_x = k->out.v.x; k->last_lhs_y = k->out.v.y;
            stage_out(k, k->out, lhs);
        } else {
            ok = k->impl->solve(*rhs, *lhs);
            k->last_lhs_x = lhs->x; k->last_lhs_y = lhs->y;
        }
        return ok ? 1 : 0;
    });
}

int pq_kktsys_mul(pq_kktsys* k, const pq_vars* lhs, pq_vars* rhs)
{
    if (!k || !rhs || !lhs) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] {
        PQ_HIP(hipSetDevice(k->impl->device()));
        if (k->ptr_mode == PQ_MEM_HOST) {
            k->ensure_staging();
            // keep the last solve's staged vectors intact: multiply through the scratch Variables sets allocated with the handle
            k->mul_in.zero(k->impl->stream());
            k->mul_out.zero(k->impl->stream());
            stage_in(k, lhs, k->mul_in);
            k->impl->mul(k->mul_in.v, k->mul_out.v);
            stage_out(k, k->mul_out, rhs);
        } else {
            k->impl->mul(*lhs, *rhs);
        }
        return (int)PQ_OK;
    });
}

int pq_kktsys_last_solve_stats(const pq_kktsys* k, int* refine_steps, int* backend_solves, double* refine_error, double* rhs_norm)
{
    if (!k) return fail(PQ_ERR_INVALID, "null argument");
    if (refine_steps) *refine_steps = k->impl->last_refine_steps;
    if (backend_solves) *backend_solves = k->impl->last_backend_solves;
    if (refine_error) *refine_error = k->impl->last_refine_error;
    if (rhs_norm) *rhs_norm = k->impl->last_rhs_norm;
    return PQ_OK;
}

int pq_kktsys_condensed_residual(pq_kktsys* k, double* res_inf, double* rhs_inf)
{
    if (!k || !res_inf || !rhs_inf) return fail(PQ_ERR_INVALID, "null argument");
    if (!k->last_lhs_x) return fail(PQ_ERR_INVALID, "no solve yet");
    return guarded([&] { k->impl->condensed_residual(k->last_lhs_x, k->last_lhs_y, res_inf, rhs_inf); return (int)PQ_OK; });
}

int pq_kktsys_synchronize(pq_kktsys* k)
{
    if (!k) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { PQ_HIP(hipSetDevice(k->impl->device())); stream_wait(k->impl->stream()); return (int)PQ_OK; });
}

// ------------------------------------------------------------------------------------ micro-benchmarks
long long pq_debug_alloc_count(void) { return alloc_counter().load(); }
int pq_debug_chol_plan(int T, int mchunks, int* out6, int capacity_tasks) { return T >= 3 && T <= 1024 ? dense::chol_debug_plan(T, mchunks, out6, capacity_tasks) : -1; }

int pq_microbench_mfma_f64(int device, int iters, double* tflops_out)
{
    int rc = check_device(device);
    if (rc < 0) return rc;
    return guarded([&] { PQ_HIP(hipSetDevice(device)); *tflops_out = dense::microbench_mfma_f64(iters, nullptr); return (int)PQ_OK; });
}
int pq_microbench_potrf_block(int device, int ldlt, int reps, double* us_out, long long* stamps64)
{
    if (!us_out) return fail(PQ_ERR_INVALID, "null output");
    int rc = check_device(device);
    if (rc < 0) return rc;
    return guarded([&] { PQ_HIP(hipSetDevice(device)); *us_out = dense::microbench_potrf_block(ldlt != 0, reps, stamps64, nullptr); return (int)PQ_OK; });
}
int pq_debug_potrf_block(int device, int ldlt, int nb, int reps, const double* A, double* L, double* rdiag, double* dvec, double* pack, int* info, int* differing_reps)
{
    if (!A || !L || !rdiag || !dvec || !pack || !info || !differing_reps || nb < 1 || nb > 128 || reps < 1) return fail(PQ_ERR_INVALID, "bad argument");
    int rc = check_device(device);
    if (rc < 0) return rc;
    return guarded([&] { PQ_HIP(hipSetDevice(device)); *differing_reps = dense::debug_potrf_block(ldlt != 0, nb, reps, A, L, rdiag, dvec, pack, info, nullptr); return (int)PQ_OK; });
}
int pq_microbench_hbm_copy(int device, size_t bytes, int iters, double* gbps_out)
{
    int rc = check_device(device);
    if (rc < 0) return rc;
    return guarded([&] { PQ_HIP(hipSetDevice(device)); *gbps_out = dense::microbench_hbm_copy(bytes, iters, nullptr); return (int)PQ_OK; });
}

}  // extern "C"
