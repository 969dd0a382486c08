// piqp_amd/csrc/ruiz_kernels.hip -- device Ruiz equilibration (see ruiz_device.hpp).  HBM-bound byte pushing: no matrix cores here.
#include "ruiz_device.hpp"

#include <memory>
#include <stdexcept>

#include "solver.hpp"

namespace pq {
namespace {

__device__ __forceinline__ double limit_scaling(double d) { return d < 1e-4 ? 1.0 : (d > 1e4 ? 1e4 : d); }  // dense/preconditioner.hpp:513-523
__device__ __forceinline__ double inv_sqrt_limited(double d) { return 1.0 / sqrt(limit_scaling(d)); }
// max of non-negative doubles as an integer max of their bit patterns: exact and order independent
__device__ __forceinline__ void amax(double* a, double v)
{
    if (v > 0.0) atomicMax(reinterpret_cast<unsigned long long*>(a), (unsigned long long)__double_as_longlong(v));
}
__device__ __forceinline__ void amax_lds(double* a, double v)
{
    if (v > 0.0) atomicMax(reinterpret_cast<unsigned long long*>(a), (unsigned long long)__double_as_longlong(v));
}
__device__ __forceinline__ double wave_max(double v)
{
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
// the value of `v` maximised over the workgroup, returned to every thread (red: 16 doubles of LDS)
__device__ __forceinline__ double block_max(double v, double* red)
{
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
    for (int w = 0; w < (int)(blockDim.x + 63) / 64; ++w) r = fmax(r, red[w]);
    return r;
}
// workgroup barrier that also orders plain accesses against the L2 atomics of other waves (vector L1 is not coherent with them)
__device__ __forceinline__ void sync_mem()
{
    __threadfence();
    __syncthreads();
    __threadfence();
}

// ------------------------------------------------------------------------------------------------ sparse, one workgroup per instance
// inf-norms of the columns of the symmetric matrix stored as the upper triangle (Px): out[k] (zeroed by the caller) via atomic max
__device__ __forceinline__ void sym_col_norms(int n, const int* __restrict__ Pp, const int* __restrict__ Pi, const double* Px, double* out, int t0, int nt)
{
    for (int j = t0; j < n; j += nt) {
        double cm = 0.0;
        for (int q = Pp[j]; q < Pp[j + 1]; ++q) {
            const int r = Pi[q];
            const double a = fabs(Px[q]);
            cm = fmax(cm, a);
            if (r != j) amax(out + r, a);
        }
        amax(out + j, cm);
    }
}
// inf-norms of the rows (-> rown, atomic) and the columns (-> coln, plain store) of a CSC matrix
__device__ __forceinline__ void rows_and_cols(int cols, const int* __restrict__ Tp, const int* __restrict__ Ti, const double* Tx, double* rown, double* coln, int t0, int nt)
{
    for (int j = t0; j < cols; j += nt) {
        double cm = 0.0;
        for (int q = Tp[j]; q < Tp[j + 1]; ++q) {
            const double a = fabs(Tx[q]);
            cm = fmax(cm, a);
            amax(rown + Ti[q], a);
        }
        coln[j] = cm;
    }
}
// sparse/utils.hpp:172-199 order: rows first, then columns; `pre` multiplies first when USE_PRE (scale_P_scalar before scale_P)
template <bool USE_PRE>
__device__ __forceinline__ void scale_csc(int cols, const int* __restrict__ Tp, const int* __restrict__ Ti, double* Tx, const double* srow, const double* scol, double pre, int t0, int nt)
{
    for (int j = t0; j < cols; j += nt) {
        const double sc = scol[j];
        for (int q = Tp[j]; q < Tp[j + 1]; ++q) {
            double v = Tx[q];
            if (USE_PRE) v *= pre;
            Tx[q] = (v * srow[Ti[q]]) * sc;
        }
    }
}

// GRID = false: one workgroup per instance (the batched solver; a single small QP).  GRID = true (round 4): the gridDim.x workgroups of the launch share ONE
// instance -- every loop strides over the whole grid, the workgroup barriers become grid barriers (a monotonic counter in a.grid_ws; all workgroups are
// resident: the launch has at most one per CU), the two maxima are reduced through one slot per iteration.  The statements, and the order of the one sum that
// has an order (the mean of the column norms, thread 0 of workgroup 0), are the same: bitwise the same scaling.  One workgroup needed 13 ms for the C3 pattern.
template <bool GRID>
__device__ __forceinline__ void ruiz_sync(unsigned long long* ws, unsigned& epoch)
{
    if constexpr (!GRID) { sync_mem(); return; }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        epoch += gridDim.x;
        unsigned* cnt = reinterpret_cast<unsigned*>(ws);
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < epoch) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    __threadfence();
}
// maximum over all threads that share the instance, returned to every thread; slot: a zero-initialised word of a.grid_ws used once
template <bool GRID>
__device__ __forceinline__ double ruiz_max(double v, double* red, unsigned long long* ws, int slot, unsigned& epoch)
{
    v = block_max(v, red);
    if constexpr (!GRID) return v;
    if (threadIdx.x == 0) amax(reinterpret_cast<double*>(ws + slot), v);
    ruiz_sync<GRID>(ws, epoch);
    return __longlong_as_double((long long)__hip_atomic_load(ws + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
constexpr int RUIZ_WS_DEV = 2, RUIZ_WS_CINF = 66, RUIZ_WS_GSUM = 130, RUIZ_WS_WORDS = 132;  // counter | per-iteration maxima (<= 64 iterations) | the mean's sum

template <bool GRID>
__global__ void k_ruiz_sparse(RuizSparseArgs a)
{
    __shared__ double red[16];
    __shared__ double gsum;
    const long long o = GRID ? 0 : (long long)blockIdx.x * a.stride;
    const int inst = GRID ? 0 : (int)blockIdx.x;
    const int n = a.n, p = a.p, m = a.m, N = n + p + m;
    const int t = GRID ? (int)(blockIdx.x * blockDim.x + threadIdx.x) : (int)threadIdx.x, NT = GRID ? (int)(gridDim.x * blockDim.x) : (int)blockDim.x;
    unsigned long long* ws = a.grid_ws;
    unsigned epoch = 0;
    double *Px = a.Px + o, *ATx = a.ATx + o, *GTx = a.GTx + o, *c = a.c + o, *xbs = a.xbs + o;
    double *delta = a.delta + o, *delta_b = a.delta_b + o, *di = a.delta_inv + o, *dib = a.delta_b_inv + o, *tmp = a.tmp + o;
    const double *s = delta, *sb = delta_b;  // what the tail multiplies with
    if (a.mode == RUIZ_COMPUTE) {
        // sparse/preconditioner.hpp:65-206.  di / dib (the inverse slots, rewritten at the end) hold the scaling of the current pass.
        double cs = 1.0;
        for (int i = t; i < N; i += NT) { delta[i] = 1.0; di[i] = 0.0; }
        for (int i = t; i < n; i += NT) { delta_b[i] = 1.0; dib[i] = 0.0; }
        ruiz_sync<GRID>(ws, epoch);
        for (int it = 0; it < a.max_iter; ++it) {
            double dev = 0.0;
            for (int i = t; i < N; i += NT) dev = fmax(dev, fabs(1.0 - di[i]));
            for (int i = t; i < n; i += NT) dev = fmax(dev, fabs(1.0 - dib[i]));
            dev = ruiz_max<GRID>(dev, red, ws, RUIZ_WS_DEV + it, epoch);
            if (!(dev > a.eps)) break;
            for (int i = t; i < N; i += NT) di[i] = 0.0;
            ruiz_sync<GRID>(ws, epoch);
            sym_col_norms(n, a.Pp, a.Pi, Px, di, t, NT);
            if (p > 0) rows_and_cols(p, a.ATp, a.ATi, ATx, di, di + n, t, NT);
            if (m > 0) rows_and_cols(m, a.GTp, a.GTi, GTx, di, di + n + p, t, NT);
            ruiz_sync<GRID>(ws, epoch);
            for (int i = t; i < N; i += NT) {
                if (i < n) {
                    const double xb = xbs[i];
                    di[i] = inv_sqrt_limited(fmax(di[i], xb));
                    dib[i] = inv_sqrt_limited(xb);
                } else {
                    di[i] = inv_sqrt_limited(di[i]);
                }
            }
            ruiz_sync<GRID>(ws, epoch);
            scale_csc<false>(n, a.Pp, a.Pi, Px, di, di, 1.0, t, NT);
            for (int i = t; i < n; i += NT) c[i] *= di[i];
            if (p > 0) scale_csc<false>(p, a.ATp, a.ATi, ATx, di, di + n, 1.0, t, NT);
            if (m > 0) scale_csc<false>(m, a.GTp, a.GTi, GTx, di, di + n + p, 1.0, t, NT);
            for (int i = t; i < n; i += NT) { xbs[i] *= dib[i] * di[i]; delta_b[i] *= dib[i]; }
            for (int i = t; i < N; i += NT) delta[i] *= di[i];
            if (a.scale_cost) {
                for (int i = t; i < n; i += NT) tmp[i] = 0.0;
                ruiz_sync<GRID>(ws, epoch);
                sym_col_norms(n, a.Pp, a.Pi, Px, tmp, t, NT);
                ruiz_sync<GRID>(ws, epoch);
                if (t == 0) {
                    double g = 0.0;
                    for (int k = 0; k < n; ++k) g += tmp[k];
                    if constexpr (GRID) __hip_atomic_store(reinterpret_cast<double*>(ws + RUIZ_WS_GSUM), g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else gsum = g;
                }
                double cinf = 0.0;
                for (int i = t; i < n; i += NT) cinf = fmax(cinf, fabs(c[i]));
                cinf = ruiz_max<GRID>(cinf, red, ws, RUIZ_WS_CINF + it, epoch);  // (its barriers also publish gsum)
                double gamma = (GRID ? __hip_atomic_load(reinterpret_cast<double*>(ws + RUIZ_WS_GSUM), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : gsum) / (double)n;
                gamma = limit_scaling(gamma);
                gamma = limit_scaling(fmax(gamma, cinf));
                gamma = 1.0 / gamma;
                for (int j = t; j < n; j += NT) for (int q = a.Pp[j]; q < a.Pp[j + 1]; ++q) Px[q] *= gamma;
                for (int i = t; i < n; i += NT) c[i] *= gamma;
                cs *= gamma;
            }
            ruiz_sync<GRID>(ws, epoch);
        }
        ruiz_sync<GRID>(ws, epoch);
        for (int i = t; i < N; i += NT) di[i] = 1.0 / delta[i];
        for (int i = t; i < n; i += NT) dib[i] = 1.0 / delta_b[i];
        if (t == 0) a.c_scale[inst] = cs;
        ruiz_sync<GRID>(ws, epoch);
    } else {
        // scale_data with reuse_prev_scaling (:207-217) / unscale_data (:224-258): the same products with (c, delta) or their inverses
        double cs = a.c_scale[inst];
        if (a.mode == RUIZ_UNSCALE) { cs = 1.0 / cs; s = di; sb = dib; }
        scale_csc<true>(n, a.Pp, a.Pi, Px, s, s, cs, t, NT);
        for (int i = t; i < n; i += NT) c[i] *= cs * s[i];
        if (p > 0) scale_csc<false>(p, a.ATp, a.ATi, ATx, s, s + n, 1.0, t, NT);
        if (m > 0) scale_csc<false>(m, a.GTp, a.GTi, GTx, s, s + n + p, 1.0, t, NT);
        for (int i = t; i < n; i += NT) xbs[i] *= sb[i] * s[i];
    }
    if (a.b) for (int i = t; i < p; i += NT) a.b[o + i] *= s[n + i];
    if (a.h_l) for (int i = t; i < m; i += NT) a.h_l[o + i] *= s[n + p + i];
    if (a.h_u) for (int i = t; i < m; i += NT) a.h_u[o + i] *= s[n + p + i];
    if (a.x_l) for (int k = t; k < a.n_x_l; k += NT) a.x_l[o + k] *= sb[a.x_l_idx[k]];
    if (a.x_u) for (int k = t; k < a.n_x_u; k += NT) a.x_u[o + k] *= sb[a.x_u_idx[k]];
}

// ------------------------------------------------------------------------------------------------ dense, tiled
constexpr int TR = 256, TC = 32;  // tile = 256 rows (one per thread, coalesced along the column) x 32 columns

struct DenseState {  // device-resident control block of one equilibration
    double c, c_inv, gamma;
    int done;
};

__global__ void k_rzd_begin(int N, int n, double* delta, double* delta_b, double* di, double* dib, DenseState* st)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) { delta[i] = 1.0; di[i] = 0.0; }
    if (i < n) { delta_b[i] = 1.0; dib[i] = 0.0; }
    if (i == 0) { st->c = 1.0; st->gamma = 1.0; st->done = 0; }
}
// head of one pass: converged? (dense/preconditioner.hpp:82) -- otherwise clear the norm accumulators
__global__ void k_rzd_check(int N, int n, double* di, const double* dib, double* tmp, double eps, DenseState* st)
{
    __shared__ double red[16];
    if (st->done) return;
    double dev = 0.0;
    for (int i = threadIdx.x; i < N; i += blockDim.x) dev = fmax(dev, fabs(1.0 - di[i]));
    for (int i = threadIdx.x; i < n; i += blockDim.x) dev = fmax(dev, fabs(1.0 - dib[i]));
    dev = block_max(dev, red);
    if (!(dev > eps)) { if (threadIdx.x == 0) st->done = 1; return; }
    for (int i = threadIdx.x; i < N; i += blockDim.x) di[i] = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) tmp[i] = 0.0;
}
// One pass over a column-major rows x cols matrix, tile by tile: optional scalar factor (USE_PRE: M *= *pre_ptr, scale_P_scalar), optional
// row / column scaling (VEC), optional inf-norms of the resulting rows and columns accumulated into rown[i] / coln[j] by atomic max.
//   SYM: only the upper triangle i <= j is stored / touched and both norms describe the symmetric matrix (rown == coln array); the products
//   are (M * scol) * srow (dense scale_P, preconditioner.hpp:119-124), otherwise (srow * M) * scol (A, G: :127-131).
template <bool SYM, bool VEC, bool NORMS, bool USE_PRE>
__global__ __launch_bounds__(TR) void k_rzd_pass(int rows, int cols, double* __restrict__ M, const double* __restrict__ srow, const double* __restrict__ scol, const double* pre_ptr,
                                                 double* rown, double* coln, const DenseState* st, int ignore_done)
{
    __shared__ double colmax[TC];
    if (!ignore_done && st->done) return;
    const int rb = blockIdx.x, cb = blockIdx.y;
    const int j0 = cb * TC, i = rb * TR + threadIdx.x;
    if (SYM && rb * TR > j0 + TC - 1) return;  // tile strictly below the diagonal
    const int jn = min(TC, cols - j0);
    if (NORMS) { if (threadIdx.x < TC) colmax[threadIdx.x] = 0.0; __syncthreads(); }
    const double pre = USE_PRE ? *pre_ptr : 1.0;
    const double si = (VEC && i < rows) ? srow[i] : 1.0;
    double rmax = 0.0;
    for (int jj = 0; jj < jn; ++jj) {
        const int j = j0 + jj;
        double a = 0.0;
        if (i < rows && (!SYM || i <= j)) {
            double* e = M + (size_t)j * rows + i;
            double v = *e;
            if (USE_PRE) v *= pre;
            if (VEC) v = SYM ? (v * scol[j]) * si : (si * v) * scol[j];
            if (VEC || USE_PRE) *e = v;
            a = fabs(v);
        }
        if (NORMS) {
            rmax = fmax(rmax, a);
            const double cm = wave_max(a);
            if ((threadIdx.x & 63) == 0) amax_lds(&colmax[jj], cm);
        }
    }
    if (NORMS) {
        if (i < rows) amax(rown + i, rmax);
        __syncthreads();
        if (threadIdx.x < jn) amax(coln + j0 + threadIdx.x, colmax[threadIdx.x]);
    }
}
// dense/preconditioner.hpp:112-117,133-139: this pass's scaling from the norms; c, x_b_scaling, delta, delta_b follow
__global__ void k_rzd_delta(int N, int n, double* di, double* dib, double* c, double* xbs, double* delta, double* delta_b, const DenseState* st)
{
    if (st->done) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    if (i < n) {
        const double xb = xbs[i];
        const double d = inv_sqrt_limited(fmax(di[i], xb)), db = inv_sqrt_limited(xb);
        di[i] = d; dib[i] = db;
        c[i] *= d;
        xbs[i] = xb * (db * d);
        delta_b[i] *= db;
        delta[i] *= d;
    } else {
        const double d = inv_sqrt_limited(di[i]);
        di[i] = d;
        delta[i] *= d;
    }
}
// :141-158 cost scaling: gamma = 1 / limit(max(limit(mean column norm of P), |c|_inf))
__global__ void k_rzd_gamma(int n, const double* tmp, double* c, DenseState* st)
{
    __shared__ double red[16];
    __shared__ double gsum;
    if (st->done) return;
    if (threadIdx.x == 0) {
        double g = 0.0;
        for (int k = 0; k < n; ++k) g += tmp[k];
        gsum = g;
    }
    double cinf = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) cinf = fmax(cinf, fabs(c[i]));
    cinf = block_max(cinf, red);
    double gamma = gsum / (double)n;
    gamma = limit_scaling(gamma);
    gamma = limit_scaling(fmax(gamma, cinf));
    gamma = 1.0 / gamma;
    for (int i = threadIdx.x; i < n; i += blockDim.x) c[i] *= gamma;
    if (threadIdx.x == 0) { st->gamma = gamma; st->c *= gamma; }
}
__global__ void k_rzd_finish(int N, int n, const double* delta, const double* delta_b, double* di, double* dib, DenseState* st)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) di[i] = 1.0 / delta[i];
    if (i < n) dib[i] = 1.0 / delta_b[i];
    if (i == 0) st->c_inv = 1.0 / st->c;
}
// the vector part of the reuse / unscale branches: c *= cs * s, x_b_scaling *= sb * s
__global__ void k_rzd_vectors(int n, double* c, double* xbs, const double* s, const double* sb, const double* cs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { c[i] *= *cs * s[i]; xbs[i] *= sb[i] * s[i]; }
}

}  // namespace

void launch_ruiz_sparse(const RuizSparseArgs& a, int batch, int threads, hipStream_t s)
{
    if (batch <= 0) return;
    static const bool one_wg = debug_token("ruiz_one_wg") != nullptr;  // debugging aid: PIQP_AMD_DEBUG=ruiz_one_wg
    if (batch == 1 && a.grid_ws && a.max_iter <= 64 && !one_wg) {
        // one instance on the whole chip: workgroups of 256 threads, at most one per CU (every workgroup must be resident: grid barriers)
        int cus = 0, dev = 0;
        PQ_HIP(hipGetDevice(&dev));
        PQ_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        const int N = a.n + a.p + a.m;
        // 512 rows per workgroup, at most 48 workgroups: a barrier costs more the more workgroups poll its counter (C3 pattern, N = 100 000: 5.3 ms on 49
        // workgroups, 12.4 ms on 196), and below 2048 rows four workgroups still halve the one-workgroup time (chain-mass N = 1822: update 1.9 -> 1.1 ms)
        const int g = std::max(1, std::min(std::min(cus > 0 ? cus : 64, 48), (N + 511) / 512));
        if (g > 1) {
            PQ_HIP(hipMemsetAsync(a.grid_ws, 0, sizeof(unsigned long long) * RUIZ_WS_WORDS, s));
            hipLaunchKernelGGL(k_ruiz_sparse<true>, dim3(g), dim3(256), 0, s, a);
            PQ_HIP(hipGetLastError());
            return;
        }
    }
    hipLaunchKernelGGL(k_ruiz_sparse<false>, dim3(batch), dim3(threads), 0, s, a);
    PQ_HIP(hipGetLastError());
}
size_t ruiz_grid_ws_words() { return RUIZ_WS_WORDS; }

// ------------------------------------------------------------------------------------------------ DeviceRuiz
struct DeviceRuiz::Impl {
    int device = 0, n = 0, p = 0, m = 0;
    bool sparse = false;
    hipStream_t st = nullptr;
    // dense: the matrices (scaled once scale() ran); sparse: value staging
    DBuf<double> P, AT, GT;
    DBuf<int> Pp, Pi, ATp, ATi, GTp, GTi;
    DBuf<double> vec;  // c | xbs | delta | delta_inv | delta_b | delta_b_inv | tmp
    DBuf<DenseState> state;
    DBuf<double> cscale;
    DBuf<unsigned long long> gridws;  // sparse: barrier counter and reduction slots of the grid-wide equilibration
    HBuf<double> hvec;
    double *c = nullptr, *xbs = nullptr, *delta = nullptr, *delta_inv = nullptr, *delta_b = nullptr, *delta_b_inv = nullptr, *tmp = nullptr;
    size_t vec_len = 0;
    ~Impl()
    {
        (void)hipSetDevice(device);
        if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    }
};

DeviceRuiz::DeviceRuiz(int device, const HostData& d) : I(new Impl)
{
    Impl& s = *I;
    s.device = device; s.n = d.n; s.p = d.p; s.m = d.m; s.sparse = d.sparse;
    PQ_HIP(hipSetDevice(device));
    PQ_HIP(hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking));
    const size_t n = d.n, N = (size_t)d.n + d.p + d.m;
    s.vec_len = 5 * n + 2 * N;
    s.vec.alloc(s.vec_len);
    s.hvec.alloc(s.vec_len);
    double* v = s.vec.p;
    s.c = v; v += n; s.xbs = v; v += n; s.delta = v; v += N; s.delta_inv = v; v += N; s.delta_b = v; v += n; s.delta_b_inv = v; v += n; s.tmp = v;
    s.state.alloc(1);
    s.cscale.alloc(1);
    if (!d.sparse) {
        s.P.alloc(n * n);
        s.AT.alloc(n * (size_t)d.p);
        s.GT.alloc(n * (size_t)d.m);
    } else {
        auto up = [&](DBuf<int>& dst, const IVec& h) { dst.alloc(std::max<size_t>(h.size(), 1)); if (!h.empty()) PQ_HIP(hipMemcpyAsync(dst.p, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice, s.st)); };
        up(s.Pp, d.sP_utri.colptr); up(s.Pi, d.sP_utri.rowind);
        up(s.ATp, d.sAT.colptr); up(s.ATi, d.sAT.rowind);
        up(s.GTp, d.sGT.colptr); up(s.GTi, d.sGT.rowind);
        s.P.alloc(std::max<size_t>(d.sP_utri.val.size(), 1));
        s.AT.alloc(std::max<size_t>(d.sAT.val.size(), 1));
        s.GT.alloc(std::max<size_t>(d.sGT.val.size(), 1));
        stream_wait(s.st);
    }
}
DeviceRuiz::~DeviceRuiz() = default;

std::unique_ptr<DeviceRuiz> DeviceRuiz::clone() const
{
    const Impl& s = *I;
    HostData shape;
    shape.sparse = false; shape.n = s.n; shape.p = s.p; shape.m = s.m;
    if (s.sparse) throw std::runtime_error("DeviceRuiz::clone: sparse instances are rebuilt from the host data");
    std::unique_ptr<DeviceRuiz> r(new DeviceRuiz(s.device, shape));
    Impl& t = *r->I;
    PQ_HIP(hipSetDevice(s.device));
    stream_wait(s.st);
    auto cp = [&](DBuf<double>& dst, const DBuf<double>& src) { if (src.n) PQ_HIP(hipMemcpyAsync(dst.p, src.p, src.bytes(), hipMemcpyDeviceToDevice, t.st)); };
    cp(t.P, s.P); cp(t.AT, s.AT); cp(t.GT, s.GT); cp(t.vec, s.vec);
    PQ_HIP(hipMemcpyAsync(t.state.p, s.state.p, sizeof(DenseState), hipMemcpyDeviceToDevice, t.st));
    stream_wait(t.st);
    return r;
}

void DeviceRuiz::upload_dense(const HostData& d, int options)
{
    Impl& s = *I;
    if (s.sparse) throw std::runtime_error("DeviceRuiz::upload_dense on a sparse problem");
    PQ_HIP(hipSetDevice(s.device));
    const size_t n = s.n;
    if ((options & PQ_KKT_UPDATE_P) && s.P.n) PQ_HIP(hipMemcpyAsync(s.P.p, d.P_utri.data(), s.P.bytes(), hipMemcpyHostToDevice, s.st));
    if ((options & PQ_KKT_UPDATE_A) && s.AT.n) PQ_HIP(hipMemcpyAsync(s.AT.p, d.AT.data(), n * s.p * sizeof(double), hipMemcpyHostToDevice, s.st));
    if ((options & PQ_KKT_UPDATE_G) && s.GT.n) PQ_HIP(hipMemcpyAsync(s.GT.p, d.GT.data(), n * s.m * sizeof(double), hipMemcpyHostToDevice, s.st));
    stream_wait(s.st);
}

void DeviceRuiz::zero_G_rows(const std::vector<int>& rows)
{
    Impl& s = *I;
    if (s.sparse || rows.empty()) return;
    PQ_HIP(hipSetDevice(s.device));
    for (int r : rows) PQ_HIP(hipMemsetAsync(s.GT.p + (size_t)r * s.n, 0, sizeof(double) * s.n, s.st));
    stream_wait(s.st);
}

pq_dense_data DeviceRuiz::dense_descriptor(const HostData& d) const
{
    pq_dense_data desc = d.dense_descriptor();
    desc.P_utri = I->P.p; desc.AT = I->AT.p; desc.GT = I->GT.p;
    desc.mem = PQ_MEM_DEVICE;
    return desc;
}

void DeviceRuiz::scale(HostData& d, Ruiz& rz, bool reuse_prev_scaling, bool scale_cost, int max_iter, double eps)
{
    run(d, rz, reuse_prev_scaling ? RUIZ_REUSE : RUIZ_COMPUTE, scale_cost, max_iter, eps);
}
void DeviceRuiz::unscale(HostData& d, Ruiz& rz) { run(d, rz, RUIZ_UNSCALE, false, 0, 0.0); }

void DeviceRuiz::run(HostData& d, Ruiz& rz, int mode, bool scale_cost, int max_iter, double eps)
{
    Impl& s = *I;
    PQ_HIP(hipSetDevice(s.device));
    const int n = s.n, p = s.p, m = s.m, N = n + p + m;
    hipStream_t st = s.st;
    // ---- host -> device: c, x_b_scaling (always), the scalings (reuse / unscale), the sparse values
    double* h = s.hvec.p;
    const size_t oc = 0, ox = n, od = 2 * (size_t)n, odi = od + N, odb = odi + N, odbi = odb + n;
    std::copy(d.c.begin(), d.c.begin() + n, h + oc);
    std::copy(d.x_b_scaling.begin(), d.x_b_scaling.begin() + n, h + ox);
    if (mode != RUIZ_COMPUTE) {
        std::copy(rz.delta.begin(), rz.delta.end(), h + od); std::copy(rz.delta_inv.begin(), rz.delta_inv.end(), h + odi);
        std::copy(rz.delta_b.begin(), rz.delta_b.end(), h + odb); std::copy(rz.delta_b_inv.begin(), rz.delta_b_inv.end(), h + odbi);
    }
    PQ_HIP(hipMemcpyAsync(s.vec.p, h, (odbi + n) * sizeof(double), hipMemcpyHostToDevice, st));
    if (s.sparse) {
        auto upv = [&](DBuf<double>& dst, const Vec& v) { if (!v.empty()) PQ_HIP(hipMemcpyAsync(dst.p, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice, st)); };
        upv(s.P, d.sP_utri.val); upv(s.AT, d.sAT.val); upv(s.GT, d.sGT.val);
        PQ_HIP(hipMemcpyAsync(s.cscale.p, &rz.c, sizeof(double), hipMemcpyHostToDevice, st));
        RuizSparseArgs a;
        a.n = n; a.p = p; a.m = m;
        a.Pp = s.Pp.p; a.Pi = s.Pi.p; a.ATp = s.ATp.p; a.ATi = s.ATi.p; a.GTp = s.GTp.p; a.GTi = s.GTi.p;
        a.Px = s.P.p; a.ATx = s.AT.p; a.GTx = s.GT.p; a.c = s.c; a.xbs = s.xbs;
        a.delta = s.delta; a.delta_inv = s.delta_inv; a.delta_b = s.delta_b; a.delta_b_inv = s.delta_b_inv; a.tmp = s.tmp;
        a.c_scale = s.cscale.p;
        a.mode = mode; a.scale_cost = scale_cost; a.max_iter = max_iter; a.eps = eps;
        if (!s.gridws.p) s.gridws.alloc(ruiz_grid_ws_words());
        a.grid_ws = s.gridws.p;
        launch_ruiz_sparse(a, 1, 1024, st);
        auto dn = [&](Vec& v, const DBuf<double>& src) { if (!v.empty()) PQ_HIP(hipMemcpyAsync(v.data(), src.p, v.size() * sizeof(double), hipMemcpyDeviceToHost, st)); };
        dn(d.sP_utri.val, s.P); dn(d.sAT.val, s.AT); dn(d.sGT.val, s.GT);
        if (mode == RUIZ_COMPUTE) PQ_HIP(hipMemcpyAsync(&rz.c, s.cscale.p, sizeof(double), hipMemcpyDeviceToHost, st));
    } else {
        DenseState* ds = s.state.p;
        const dim3 gP(div_up(n, TR), div_up(n, TC)), gA(div_up(n, TR), std::max(1, div_up(p, TC))), gG(div_up(n, TR), std::max(1, div_up(m, TC)));
        const int eb = 256, eg = div_up(N, eb);
        if (mode == RUIZ_COMPUTE) {
            hipLaunchKernelGGL(k_rzd_begin, dim3(eg), dim3(eb), 0, st, N, n, s.delta, s.delta_b, s.delta_inv, s.delta_b_inv, ds);
            for (int it = 0; it < max_iter; ++it) {
                hipLaunchKernelGGL(k_rzd_check, dim3(1), dim3(1024), 0, st, N, n, s.delta_inv, s.delta_b_inv, s.tmp, eps, ds);
                hipLaunchKernelGGL((k_rzd_pass<true, false, true, false>), gP, dim3(TR), 0, st, n, n, s.P.p, nullptr, nullptr, nullptr, s.delta_inv, s.delta_inv, ds, 0);
                if (p > 0) hipLaunchKernelGGL((k_rzd_pass<false, false, true, false>), gA, dim3(TR), 0, st, n, p, s.AT.p, nullptr, nullptr, nullptr, s.delta_inv, s.delta_inv + n, ds, 0);
                if (m > 0) hipLaunchKernelGGL((k_rzd_pass<false, false, true, false>), gG, dim3(TR), 0, st, n, m, s.GT.p, nullptr, nullptr, nullptr, s.delta_inv, s.delta_inv + n + p, ds, 0);
                hipLaunchKernelGGL(k_rzd_delta, dim3(eg), dim3(eb), 0, st, N, n, s.delta_inv, s.delta_b_inv, s.c, s.xbs, s.delta, s.delta_b, ds);
                if (scale_cost) hipLaunchKernelGGL((k_rzd_pass<true, true, true, false>), gP, dim3(TR), 0, st, n, n, s.P.p, s.delta_inv, s.delta_inv, nullptr, s.tmp, s.tmp, ds, 0);
                else hipLaunchKernelGGL((k_rzd_pass<true, true, false, false>), gP, dim3(TR), 0, st, n, n, s.P.p, s.delta_inv, s.delta_inv, nullptr, nullptr, nullptr, ds, 0);
                if (p > 0) hipLaunchKernelGGL((k_rzd_pass<false, true, false, false>), gA, dim3(TR), 0, st, n, p, s.AT.p, s.delta_inv, s.delta_inv + n, nullptr, nullptr, nullptr, ds, 0);
                if (m > 0) hipLaunchKernelGGL((k_rzd_pass<false, true, false, false>), gG, dim3(TR), 0, st, n, m, s.GT.p, s.delta_inv, s.delta_inv + n + p, nullptr, nullptr, nullptr, ds, 0);
                if (scale_cost) {
                    hipLaunchKernelGGL(k_rzd_gamma, dim3(1), dim3(1024), 0, st, n, s.tmp, s.c, ds);
                    hipLaunchKernelGGL((k_rzd_pass<true, false, false, true>), gP, dim3(TR), 0, st, n, n, s.P.p, nullptr, nullptr, &ds->gamma, nullptr, nullptr, ds, 0);
                }
            }
            hipLaunchKernelGGL(k_rzd_finish, dim3(eg), dim3(eb), 0, st, N, n, s.delta, s.delta_b, s.delta_inv, s.delta_b_inv, ds);
        } else {
            DenseState hs{rz.c, rz.c_inv, 1.0, 0};
            PQ_HIP(hipMemcpyAsync(ds, &hs, sizeof(DenseState), hipMemcpyHostToDevice, st));
            stream_wait(st);  // hs is a stack object
            const double *sv = mode == RUIZ_REUSE ? s.delta : s.delta_inv, *sb = mode == RUIZ_REUSE ? s.delta_b : s.delta_b_inv;
            const double* cs = mode == RUIZ_REUSE ? &ds->c : &ds->c_inv;
            hipLaunchKernelGGL((k_rzd_pass<true, true, false, true>), gP, dim3(TR), 0, st, n, n, s.P.p, sv, sv, cs, nullptr, nullptr, ds, 1);
            if (p > 0) hipLaunchKernelGGL((k_rzd_pass<false, true, false, false>), gA, dim3(TR), 0, st, n, p, s.AT.p, sv, sv + n, nullptr, nullptr, nullptr, ds, 1);
            if (m > 0) hipLaunchKernelGGL((k_rzd_pass<false, true, false, false>), gG, dim3(TR), 0, st, n, m, s.GT.p, sv, sv + n + p, nullptr, nullptr, nullptr, ds, 1);
            hipLaunchKernelGGL(k_rzd_vectors, dim3(div_up(n, eb)), dim3(eb), 0, st, n, s.c, s.xbs, sv, sb, cs);
        }
        PQ_HIP(hipGetLastError());
    }
    // ---- device -> host: c, x_b_scaling and (after a fresh equilibration) the scalings
    PQ_HIP(hipMemcpyAsync(h, s.vec.p, (odbi + n) * sizeof(double), hipMemcpyDeviceToHost, st));
    DenseState hs{};
    if (!s.sparse && mode == RUIZ_COMPUTE) PQ_HIP(hipMemcpyAsync(&hs, s.state.p, sizeof(DenseState), hipMemcpyDeviceToHost, st));
    stream_wait(st);
    std::copy(h + oc, h + oc + n, d.c.begin());
    std::copy(h + ox, h + ox + n, d.x_b_scaling.begin());
    if (mode == RUIZ_COMPUTE) {
        std::copy(h + od, h + od + N, rz.delta.begin()); std::copy(h + odi, h + odi + N, rz.delta_inv.begin());
        std::copy(h + odb, h + odb + n, rz.delta_b.begin()); std::copy(h + odbi, h + odbi + n, rz.delta_b_inv.begin());
        if (!s.sparse) rz.c = hs.c;
        rz.c_inv = 1.0 / rz.c;
    }
}

}  // namespace pq
