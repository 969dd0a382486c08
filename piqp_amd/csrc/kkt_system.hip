// piqp_amd/csrc/kkt_system.hip -- device-resident replacement of piqp::KKTSystem<T,I,MatrixType>
// (reference include/piqp/kkt_system.hpp).  The scalings, the rhs reduction, the iterative-refinement
// loop and the dual recovery all run on vectors that stay in HBM; only the scalar refinement errors
// (two doubles per refinement step) and the factor status travel to the host, because the loop's
// branches (kkt_system.hpp:266-300) are taken there.
//
// The reference walks the finite-bound index lists (data.h_l_idx ...) sequentially; here every list is
// expanded once into a per-row mask / position table so each formula becomes one elementwise kernel
// with the same operand order as the reference loop bodies.
#include "kkt_system.hpp"
#include "trace.hpp"

namespace pq {

namespace {

// ---- index tables -------------------------------------------------------------------------------
__global__ void k_fill_int(int n, int v, int* out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v;
}
__global__ void k_scatter_pos(int cnt, const int* __restrict__ idx, int* __restrict__ pos)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) pos[idx[i]] = i;
}

// ---- kkt_system.hpp:150-159 copies + reciprocals -------------------------------------------------
__global__ void k_copy_and_invert(int n, const double* __restrict__ s, const double* __restrict__ z, double* __restrict__ s_out, double* __restrict__ zinv_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { s_out[i] = s[i]; zinv_out[i] = 1.0 / z[i]; }
}

// ---- kkt_system.hpp:161-175 x_reg ----------------------------------------------------------------
__global__ void k_x_reg(int n, double rho, double delta, const int* __restrict__ pos_l, const int* __restrict__ pos_u, const double* __restrict__ xbs,
                        const double* __restrict__ s_bl, const double* __restrict__ zinv_bl, const double* __restrict__ s_bu, const double* __restrict__ zinv_bu,
                        double* __restrict__ x_reg)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    double v = rho;
    const double sc = xbs[idx];
    const int il = pos_l[idx], iu = pos_u[idx];
    if (il >= 0) v += sc * sc / (zinv_bl[il] * s_bl[il] + delta);
    if (iu >= 0) v += sc * sc / (zinv_bu[iu] * s_bu[iu] + delta);
    x_reg[idx] = v;
}

// ---- kkt_system.hpp:177-193 z_reg ----------------------------------------------------------------
__global__ void k_z_reg(int m, double delta, const int* __restrict__ has_l, const int* __restrict__ has_u, const double* __restrict__ s_l, const double* __restrict__ zinv_l,
                        const double* __restrict__ s_u, const double* __restrict__ zinv_u, double* __restrict__ z_reg, double* __restrict__ z_reg_iter_ref)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    double v = 0.0;
    if (has_l[i] >= 0) v += 1.0 / (zinv_l[i] * s_l[i] + delta);
    if (has_u[i] >= 0) v += 1.0 / (zinv_u[i] * s_u[i] + delta);
    v = 1.0 / v;
    z_reg[i] = v;
    z_reg_iter_ref[i] = v;
}

// ---- inf-norm reductions (NaN-propagating), result atomically max-combined as ordered bits ---------
__device__ __forceinline__ double wave_max_nan(double a)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(a, off, 64);
        a = (a != a || o != o) ? __builtin_nan("") : (a > o ? a : o);
    }
    return a;
}
// out[slot] = max(out[slot], max_i |a[i] + (b ? b[i] : 0)|); out must be zeroed before the first call of a group.
// |x| >= 0 so the IEEE bit pattern is monotone as unsigned; NaN (0x7ff8...) compares above every finite value.
__global__ __launch_bounds__(256) void k_absmax(int n, const double* __restrict__ a, const double* __restrict__ b, unsigned long long* __restrict__ out)
{
    __shared__ double red[4];
    double v = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const double t = fabs(a[i] + (b ? b[i] : 0.0));
        v = (t != t || v != v) ? __builtin_nan("") : (t > v ? t : v);
    }
    v = wave_max_nan(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = red[0];
        for (int k = 1; k < 4; ++k) r = (r != r || red[k] != red[k]) ? __builtin_nan("") : (red[k] > r ? red[k] : r);
        unsigned long long bits = (unsigned long long)__double_as_longlong(r != r ? __builtin_nan("") : r) & 0x7fffffffffffffffull;
        atomicMax(out, bits);
    }
}

// ---- kkt_system.hpp:204-206 static regularisation ---------------------------------------------------
__global__ void k_add_scalar(int n, double v, double* __restrict__ a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] += v;
}

// ---- kkt_system.hpp:219-234 rhs_z_bar ---------------------------------------------------------------
__global__ void k_rhs_z_bar(int m, double delta, const int* __restrict__ has_l, const int* __restrict__ has_u, const double* __restrict__ s_l, const double* __restrict__ zinv_l,
                            const double* __restrict__ s_u, const double* __restrict__ zinv_u, const double* __restrict__ z_reg, const double* __restrict__ r_z_l,
                            const double* __restrict__ r_s_l, const double* __restrict__ r_z_u, const double* __restrict__ r_s_u, double* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    double v = 0.0;
    if (has_l[i] >= 0) v -= 1.0 / (zinv_l[i] * s_l[i] + delta) * (r_z_l[i] - zinv_l[i] * r_s_l[i]);
    if (has_u[i] >= 0) v += 1.0 / (zinv_u[i] * s_u[i] + delta) * (r_z_u[i] - zinv_u[i] * r_s_u[i]);
    out[i] = v * z_reg[i];
}

// ---- kkt_system.hpp:236-252 rhs_x_bar ---------------------------------------------------------------
__global__ void k_rhs_x_bar(int n, double delta, const int* __restrict__ pos_l, const int* __restrict__ pos_u, const double* __restrict__ xbs, const double* __restrict__ s_bl,
                            const double* __restrict__ zinv_bl, const double* __restrict__ s_bu, const double* __restrict__ zinv_bu, const double* __restrict__ r_x,
                            const double* __restrict__ r_z_bl, const double* __restrict__ r_s_bl, const double* __restrict__ r_z_bu, const double* __restrict__ r_s_bu,
                            double* __restrict__ out)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    double v = r_x[idx];
    const int il = pos_l[idx], iu = pos_u[idx];
    if (il >= 0) v -= xbs[idx] * (r_z_bl[il] - zinv_bl[il] * r_s_bl[il]) / (s_bl[il] * zinv_bl[il] + delta);
    if (iu >= 0) v += xbs[idx] * (r_z_bu[iu] - zinv_bu[iu] * r_s_bu[iu]) / (s_bu[iu] * zinv_bu[iu] + delta);
    out[idx] = v;
}


// ---- fused launches of the per-iteration elementwise work (one launch instead of six / two / six; every formula is the one of the
// single-purpose kernel above it replaces, same operand order, so the values are bitwise the same) -------------------------------
// kkt_system.hpp:150-193: copies + reciprocals + x_reg + z_reg.  x_reg needs the box slacks of OTHER indices, so it recomputes their
// reciprocals from the caller's z instead of reading what a neighbouring thread is writing.
__global__ void k_scalings_fused(int n, int m, int nxl, int nxu, double rho, double delta, const double* __restrict__ v_s_l, const double* __restrict__ v_z_l,
                                 const double* __restrict__ v_s_u, const double* __restrict__ v_z_u, const double* __restrict__ v_s_bl, const double* __restrict__ v_z_bl,
                                 const double* __restrict__ v_s_bu, const double* __restrict__ v_z_bu, const int* __restrict__ has_l, const int* __restrict__ has_u,
                                 const int* __restrict__ pos_l, const int* __restrict__ pos_u, const double* __restrict__ xbs, double* __restrict__ s_l, double* __restrict__ zinv_l,
                                 double* __restrict__ s_u, double* __restrict__ zinv_u, double* __restrict__ s_bl, double* __restrict__ zinv_bl, double* __restrict__ s_bu,
                                 double* __restrict__ zinv_bu, double* __restrict__ x_reg, double* __restrict__ z_reg, double* __restrict__ z_reg_iter_ref)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) {
        const double sl = v_s_l[i], zil = 1.0 / v_z_l[i], su = v_s_u[i], ziu = 1.0 / v_z_u[i];
        s_l[i] = sl; zinv_l[i] = zil; s_u[i] = su; zinv_u[i] = ziu;
        double v = 0.0;
        if (has_l[i] >= 0) v += 1.0 / (zil * sl + delta);
        if (has_u[i] >= 0) v += 1.0 / (ziu * su + delta);
        v = 1.0 / v;
        z_reg[i] = v;
        z_reg_iter_ref[i] = v;
    }
    if (i < nxl) { s_bl[i] = v_s_bl[i]; zinv_bl[i] = 1.0 / v_z_bl[i]; }
    if (i < nxu) { s_bu[i] = v_s_bu[i]; zinv_bu[i] = 1.0 / v_z_bu[i]; }
    if (i < n) {
        double v = rho;
        const double sc = xbs[i];
        const int il = pos_l[i], iu = pos_u[i];
        if (il >= 0) v += sc * sc / ((1.0 / v_z_bl[il]) * v_s_bl[il] + delta);
        if (iu >= 0) v += sc * sc / ((1.0 / v_z_bu[iu]) * v_s_bu[iu] + delta);
        x_reg[i] = v;
    }
}
// kkt_system.hpp:219-252: rhs_z_bar + rhs_x_bar; thread 0 also zeroes the scalar slots of the finite check that follows the backend solve
__global__ void k_rhs_bars_fused(int n, int m, double delta, const int* __restrict__ has_l, const int* __restrict__ has_u, const double* __restrict__ s_l,
                                 const double* __restrict__ zinv_l, const double* __restrict__ s_u, const double* __restrict__ zinv_u, const double* __restrict__ z_reg,
                                 const double* __restrict__ r_z_l, const double* __restrict__ r_s_l, const double* __restrict__ r_z_u, const double* __restrict__ r_s_u,
                                 const int* __restrict__ pos_l, const int* __restrict__ pos_u, const double* __restrict__ xbs, const double* __restrict__ s_bl,
                                 const double* __restrict__ zinv_bl, const double* __restrict__ s_bu, const double* __restrict__ zinv_bu, const double* __restrict__ r_x,
                                 const double* __restrict__ r_z_bl, const double* __restrict__ r_s_bl, const double* __restrict__ r_z_bu, const double* __restrict__ r_s_bu,
                                 double* __restrict__ out_z, double* __restrict__ out_x, unsigned long long* __restrict__ scal, int p, const double* __restrict__ r_y,
                                 double* __restrict__ r_y_keep)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { scal[0] = 0ull; scal[1] = 0ull; scal[2] = 0ull; scal[3] = 0ull; }
    if (i < p && r_y_keep != r_y) r_y_keep[i] = r_y[i];  // kept for condensed_residual(): the caller may reuse its buffer
    if (i < m) {
        double v = 0.0;
        if (has_l[i] >= 0) v -= 1.0 / (zinv_l[i] * s_l[i] + delta) * (r_z_l[i] - zinv_l[i] * r_s_l[i]);
        if (has_u[i] >= 0) v += 1.0 / (zinv_u[i] * s_u[i] + delta) * (r_z_u[i] - zinv_u[i] * r_s_u[i]);
        out_z[i] = v * z_reg[i];
    }
    if (i < n) {
        double v = r_x[i];
        const int il = pos_l[i], iu = pos_u[i];
        if (il >= 0) v -= xbs[i] * (r_z_bl[il] - zinv_bl[il] * r_s_bl[il]) / (s_bl[il] * zinv_bl[il] + delta);
        if (iu >= 0) v += xbs[i] * (r_z_bu[iu] - zinv_bu[iu] * r_s_bu[iu]) / (s_bu[iu] * zinv_bu[iu] + delta);
        out_x[i] = v;
    }
}
// ---- kkt_system.hpp:507-536 condensed residual pieces -----------------------------------------------
// err_x = rhs_x - (((Px + x_reg o lhs_x) + ATy) + GTz)
__global__ void k_err_x(int n, const double* __restrict__ rhs_x, const double* __restrict__ Px, const double* __restrict__ x_reg, const double* __restrict__ lhs_x,
                        const double* __restrict__ ATy, const double* __restrict__ GTz, double* __restrict__ err)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = Px[i];
    v += x_reg[i] * lhs_x[i];
    v += ATy[i];
    v += GTz[i];
    err[i] = rhs_x[i] - v;
}
// err = rhs - (Mx - d o lhs)   with d a scalar (delta) or a vector (z_reg)
__global__ void k_err_yz(int n, const double* __restrict__ rhs, const double* __restrict__ Mx, double dscalar, const double* __restrict__ dvec, const double* __restrict__ lhs,
                         double* __restrict__ err)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = dvec ? dvec[i] : dscalar;
    double v = Mx[i];
    v -= d * lhs[i];
    err[i] = rhs[i] - v;
}
__global__ void k_add_inplace(int n, const double* __restrict__ b, double* __restrict__ a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] += b[i];
}

// ---- kkt_system.hpp:310-345 dual recovery -----------------------------------------------------------
__global__ void k_dual_recovery(int m, double delta, const int* __restrict__ has_l, const int* __restrict__ has_u, const double* __restrict__ s_l, const double* __restrict__ zinv_l,
                                const double* __restrict__ s_u, const double* __restrict__ zinv_u, const double* __restrict__ z_reg, const double* __restrict__ lhs_z,
                                const double* __restrict__ r_z_l, const double* __restrict__ r_s_l, const double* __restrict__ r_z_u, const double* __restrict__ r_s_u,
                                double* __restrict__ o_z_l, double* __restrict__ o_z_u, double* __restrict__ o_s_l, double* __restrict__ o_s_u)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const bool l = has_l[i] >= 0, u = has_u[i] >= 0;
    double zl = 0.0, zu = 0.0, sl = 0.0, su = 0.0;
    if (l && u) {
        const double rz_l_bar = r_z_l[i] - zinv_l[i] * r_s_l[i];
        const double W_l_inv = 1.0 / (zinv_l[i] * s_l[i] + delta);
        const double rz_u_bar = r_z_u[i] - zinv_u[i] * r_s_u[i];
        const double W_u_inv = 1.0 / (zinv_u[i] * s_u[i] + delta);
        const double r_sum = W_l_inv * W_u_inv * (rz_l_bar + rz_u_bar);
        zl = -z_reg[i] * (r_sum + W_l_inv * lhs_z[i]);
        zu = -z_reg[i] * (r_sum - W_u_inv * lhs_z[i]);
        sl = zinv_l[i] * (r_s_l[i] - s_l[i] * zl);
        su = zinv_u[i] * (r_s_u[i] - s_u[i] * zu);
    } else if (l) {
        zl = -lhs_z[i];
        sl = zinv_l[i] * (r_s_l[i] - s_l[i] * zl);
    } else if (u) {
        zu = lhs_z[i];
        su = zinv_u[i] * (r_s_u[i] - s_u[i] * zu);
    }
    o_z_l[i] = zl; o_z_u[i] = zu; o_s_l[i] = sl; o_s_u[i] = su;
}

// ---- kkt_system.hpp:347-366 box dual recovery (sign = -1 lower, +1 upper) ---------------------------
__global__ void k_box_recovery(int cnt, double sign, double delta, const int* __restrict__ idxs, const double* __restrict__ xbs, const double* __restrict__ lhs_x,
                               const double* __restrict__ s_b, const double* __restrict__ zinv_b, const double* __restrict__ r_z_b, const double* __restrict__ r_s_b,
                               double* __restrict__ o_z_b, double* __restrict__ o_s_b)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    const int idx = idxs[i];
    const double zb = (sign * xbs[idx] * lhs_x[idx] - r_z_b[i] + zinv_b[i] * r_s_b[i]) / (s_b[i] * zinv_b[i] + delta);
    o_z_b[i] = zb;
    o_s_b[i] = zinv_b[i] * (r_s_b[i] - s_b[i] * zb);
}

// ---- kkt_system.hpp:392-425 mul: elementwise pieces -------------------------------------------------
__global__ void k_mul_x(int n, double rho, const double* __restrict__ Px, const double* __restrict__ lhs_x, const double* __restrict__ ATy, const double* __restrict__ GTz,
                        const int* __restrict__ pos_l, const int* __restrict__ pos_u, const double* __restrict__ xbs, const double* __restrict__ z_bl, const double* __restrict__ z_bu,
                        double* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = Px[i];
    v += rho * lhs_x[i];
    v += ATy[i];
    v += GTz[i];
    if (pos_l[i] >= 0) v -= xbs[i] * z_bl[pos_l[i]];
    if (pos_u[i] >= 0) v += xbs[i] * z_bu[pos_u[i]];
    out[i] = v;
}
__global__ void k_mul_y(int p, double delta, const double* __restrict__ Ax, const double* __restrict__ y, double* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < p) out[i] = Ax[i] - delta * y[i];
}
__global__ void k_sub(int n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] - b[i];
}
__global__ void k_mul_z(int m, double delta, const double* __restrict__ Gx, const double* __restrict__ z_l, const double* __restrict__ z_u, const double* __restrict__ s_l,
                        const double* __restrict__ s_u, const double* __restrict__ ms_l, const double* __restrict__ ms_u, const double* __restrict__ zinv_l,
                        const double* __restrict__ zinv_u, double* __restrict__ o_z_l, double* __restrict__ o_z_u, double* __restrict__ o_s_l, double* __restrict__ o_s_u)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    o_z_l[i] = -Gx[i] + (s_l[i] - delta * z_l[i]);
    o_z_u[i] = Gx[i] + (s_u[i] - delta * z_u[i]);
    o_s_l[i] = ms_l[i] * z_l[i] + s_l[i] / zinv_l[i];
    o_s_u[i] = ms_u[i] * z_u[i] + s_u[i] / zinv_u[i];
}
__global__ void k_mul_box(int cnt, double sign, double delta, const int* __restrict__ idxs, const double* __restrict__ xbs, const double* __restrict__ lhs_x,
                          const double* __restrict__ z_b, const double* __restrict__ s_b, const double* __restrict__ ms_b, const double* __restrict__ zinv_b,
                          double* __restrict__ o_z_b, double* __restrict__ o_s_b)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    const int idx = idxs[i];
    o_z_b[i] = sign * xbs[idx] * lhs_x[idx] - delta * z_b[i] + s_b[i];
    o_s_b[i] = ms_b[i] * z_b[i] + s_b[i] / zinv_b[i];
}
__global__ void k_fill(int n, double v, double* __restrict__ a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = v;
}

inline dim3 g1(int n) { return dim3(n > 0 ? (n + 255) / 256 : 1); }
// kkt_system.hpp:305-369 without refinement: the dual / box recoveries and the allFinite check of (lhs.x, lhs.y, lhs_z) in one launch.
// The check is the NaN-propagating |.|_inf of k_absmax, max-combined into scal[0] (zeroed by k_rhs_bars_fused).
__global__ __launch_bounds__(256) void k_recover_fused(int n, int p, int m, int nxl, int nxu, double delta, const int* __restrict__ has_l, const int* __restrict__ has_u,
                                                       const double* __restrict__ s_l, const double* __restrict__ zinv_l, const double* __restrict__ s_u,
                                                       const double* __restrict__ zinv_u, const double* __restrict__ z_reg, const double* __restrict__ lhs_z,
                                                       const double* __restrict__ r_z_l, const double* __restrict__ r_s_l, const double* __restrict__ r_z_u,
                                                       const double* __restrict__ r_s_u, double* __restrict__ o_z_l, double* __restrict__ o_z_u, double* __restrict__ o_s_l,
                                                       double* __restrict__ o_s_u, const int* __restrict__ xl_idx, const int* __restrict__ xu_idx, const double* __restrict__ xbs,
                                                       const double* __restrict__ lhs_x, const double* __restrict__ lhs_y, const double* __restrict__ s_bl,
                                                       const double* __restrict__ zinv_bl, const double* __restrict__ r_z_bl, const double* __restrict__ r_s_bl,
                                                       double* __restrict__ o_z_bl, double* __restrict__ o_s_bl, const double* __restrict__ s_bu, const double* __restrict__ zinv_bu,
                                                       const double* __restrict__ r_z_bu, const double* __restrict__ r_s_bu, double* __restrict__ o_z_bu,
                                                       double* __restrict__ o_s_bu, unsigned long long* __restrict__ scal)
{
    __shared__ double red[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < m) {
        const bool l = has_l[i] >= 0, u = has_u[i] >= 0;
        double zl = 0.0, zu = 0.0, sl = 0.0, su = 0.0;
        if (l && u) {
            const double rz_l_bar = r_z_l[i] - zinv_l[i] * r_s_l[i];
            const double W_l_inv = 1.0 / (zinv_l[i] * s_l[i] + delta);
            const double rz_u_bar = r_z_u[i] - zinv_u[i] * r_s_u[i];
            const double W_u_inv = 1.0 / (zinv_u[i] * s_u[i] + delta);
            const double r_sum = W_l_inv * W_u_inv * (rz_l_bar + rz_u_bar);
            zl = -z_reg[i] * (r_sum + W_l_inv * lhs_z[i]);
            zu = -z_reg[i] * (r_sum - W_u_inv * lhs_z[i]);
            sl = zinv_l[i] * (r_s_l[i] - s_l[i] * zl);
            su = zinv_u[i] * (r_s_u[i] - s_u[i] * zu);
        } else if (l) {
            zl = -lhs_z[i];
            sl = zinv_l[i] * (r_s_l[i] - s_l[i] * zl);
        } else if (u) {
            zu = lhs_z[i];
            su = zinv_u[i] * (r_s_u[i] - s_u[i] * zu);
        }
        o_z_l[i] = zl; o_z_u[i] = zu; o_s_l[i] = sl; o_s_u[i] = su;
    }
    if (i < nxl) {
        const int idx = xl_idx[i];
        const double zb = (-1.0 * xbs[idx] * lhs_x[idx] - r_z_bl[i] + zinv_bl[i] * r_s_bl[i]) / (s_bl[i] * zinv_bl[i] + delta);
        o_z_bl[i] = zb;
        o_s_bl[i] = zinv_bl[i] * (r_s_bl[i] - s_bl[i] * zb);
    }
    if (i < nxu) {
        const int idx = xu_idx[i];
        const double zb = (1.0 * xbs[idx] * lhs_x[idx] - r_z_bu[i] + zinv_bu[i] * r_s_bu[i]) / (s_bu[i] * zinv_bu[i] + delta);
        o_z_bu[i] = zb;
        o_s_bu[i] = zinv_bu[i] * (r_s_bu[i] - s_bu[i] * zb);
    }
    double v = 0.0;
    if (i < n) { const double t = fabs(lhs_x[i]); v = (t != t || v != v) ? __builtin_nan("") : (t > v ? t : v); }
    if (i < p) { const double t = fabs(lhs_y[i]); v = (t != t || v != v) ? __builtin_nan("") : (t > v ? t : v); }
    if (i < m) { const double t = fabs(lhs_z[i]); v = (t != t || v != v) ? __builtin_nan("") : (t > v ? t : v); }
    v = wave_max_nan(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = red[0];
        for (int k = 1; k < 4; ++k) r = (r != r || red[k] != red[k]) ? __builtin_nan("") : (red[k] > r ? red[k] : r);
        const unsigned long long bits = (unsigned long long)__double_as_longlong(r != r ? __builtin_nan("") : r) & 0x7fffffffffffffffull;
        if (bits != 0ull) atomicMax(scal, bits);
    }
}

#define LAUNCH1(kern, n, st, ...)                                         \
    do {                                                                   \
        if ((n) > 0) {                                                     \
            hipLaunchKernelGGL(kern, g1(n), dim3(256), 0, st, __VA_ARGS__); \
            PQ_HIP(hipGetLastError());                                     \
        }                                                                  \
    } while (0)

}  // namespace

// =====================================================================================================

KKTSystem::KKTSystem(KKTSolverBase* backend, const pq_settings& settings) : kkt_solver(backend), settings_(settings)
{
    n_ = backend->n(); p_ = backend->p(); m_ = backend->m();
    dev_ = backend->device();
    st_ = backend->stream();
    PQ_HIP(hipSetDevice(dev_));
    alloc();
}

KKTSystem::~KKTSystem() { delete kkt_solver; }

void KKTSystem::alloc()
{
    const int n = n_, p = p_, m = m_;
    m_s_l.alloc(m); m_s_u.alloc(m); m_s_bl.alloc(n); m_s_bu.alloc(n);
    m_z_l_inv.alloc(m); m_z_u_inv.alloc(m); m_z_bl_inv.alloc(n); m_z_bu_inv.alloc(n);
    m_x_reg.alloc(n); m_z_reg.alloc(m);
    rhs_x_bar.alloc(n); rhs_z_bar.alloc(m);
    work_x.alloc(n); work_x2.alloc(n); work_x3.alloc(n); work_y.alloc(p); work_z.alloc(m); work_z2.alloc(m); lhs_z_buf.alloc(m);
    ref_err_x.alloc(n); ref_err_y.alloc(p); ref_err_z.alloc(m);
    ref_lhs_x.alloc(n); ref_lhs_y.alloc(p); ref_lhs_z.alloc(m); rhs_y_keep.alloc(p);
    has_l.alloc(m); has_u.alloc(m); pos_l.alloc(n); pos_u.alloc(n);
    h_l_idx.alloc(m); h_u_idx.alloc(m); x_l_idx.alloc(n); x_u_idx.alloc(n);
    x_b_scaling.alloc(n);
    scal_d.alloc(4);
    scal_h.alloc(4);
    for (DBuf<double>* b : {&m_s_l, &m_s_u, &m_s_bl, &m_s_bu, &m_z_l_inv, &m_z_u_inv, &m_z_bl_inv, &m_z_bu_inv, &m_x_reg, &m_z_reg, &rhs_x_bar, &rhs_z_bar, &lhs_z_buf})
        b->zero(st_);
}

// the index-list part of dense::Data / sparse::Data (dense/data.hpp:41-51)
void KKTSystem::set_bounds(int n_h_l_, int n_h_u_, int n_x_l_, int n_x_u_, const int* hl, const int* hu, const int* xl, const int* xu, const double* xbs, int mem)
{
    PQ_HIP(hipSetDevice(dev_));
    n_h_l = n_h_l_; n_h_u = n_h_u_; n_x_l = n_x_l_; n_x_u = n_x_u_;
    if (n_h_l < 0 || n_h_l > m_ || n_h_u < 0 || n_h_u > m_ || n_x_l < 0 || n_x_l > n_ || n_x_u < 0 || n_x_u > n_) throw std::runtime_error("bad bound counts");
    copy_in(h_l_idx.p, hl, sizeof(int) * n_h_l, mem, st_);
    copy_in(h_u_idx.p, hu, sizeof(int) * n_h_u, mem, st_);
    copy_in(x_l_idx.p, xl, sizeof(int) * n_x_l, mem, st_);
    copy_in(x_u_idx.p, xu, sizeof(int) * n_x_u, mem, st_);
    if (xbs) copy_in(x_b_scaling.p, xbs, sizeof(double) * n_, mem, st_);
    else LAUNCH1(k_fill, n_, st_, n_, 1.0, x_b_scaling.p);
    LAUNCH1(k_fill_int, m_, st_, m_, -1, has_l.p);
    LAUNCH1(k_fill_int, m_, st_, m_, -1, has_u.p);
    LAUNCH1(k_fill_int, n_, st_, n_, -1, pos_l.p);
    LAUNCH1(k_fill_int, n_, st_, n_, -1, pos_u.p);
    LAUNCH1(k_scatter_pos, n_h_l, st_, n_h_l, h_l_idx.p, has_l.p);
    LAUNCH1(k_scatter_pos, n_h_u, st_, n_h_u, h_u_idx.p, has_u.p);
    LAUNCH1(k_scatter_pos, n_x_l, st_, n_x_l, x_l_idx.p, pos_l.p);
    LAUNCH1(k_scatter_pos, n_x_u, st_, n_x_u, x_u_idx.p, pos_u.p);
    stream_wait(st_);
}

// kkt_system.hpp:70-95 (copy ctor): state vectors copied, work/refinement buffers fresh
KKTSystem* KKTSystem::clone() const
{
    PQ_HIP(hipSetDevice(dev_));
    stream_wait(st_);
    KKTSolverBase* b = kkt_solver->clone();
    KKTSystem* k = new KKTSystem(b, settings_);
    k->m_rho = m_rho; k->m_delta = m_delta; k->use_iterative_refinement = use_iterative_refinement;
    k->n_h_l = n_h_l; k->n_h_u = n_h_u; k->n_x_l = n_x_l; k->n_x_u = n_x_u;
    auto cpd = [&](DBuf<double>& d, const DBuf<double>& s) { if (s.n) PQ_HIP(hipMemcpyAsync(d.p, s.p, s.bytes(), hipMemcpyDeviceToDevice, k->st_)); };
    auto cpi = [&](DBuf<int>& d, const DBuf<int>& s) { if (s.n) PQ_HIP(hipMemcpyAsync(d.p, s.p, s.bytes(), hipMemcpyDeviceToDevice, k->st_)); };
    cpd(k->m_s_l, m_s_l); cpd(k->m_s_u, m_s_u); cpd(k->m_s_bl, m_s_bl); cpd(k->m_s_bu, m_s_bu);
    cpd(k->m_z_l_inv, m_z_l_inv); cpd(k->m_z_u_inv, m_z_u_inv); cpd(k->m_z_bl_inv, m_z_bl_inv); cpd(k->m_z_bu_inv, m_z_bu_inv);
    cpd(k->m_x_reg, m_x_reg); cpd(k->m_z_reg, m_z_reg); cpd(k->rhs_x_bar, rhs_x_bar); cpd(k->rhs_z_bar, rhs_z_bar);
    cpd(k->x_b_scaling, x_b_scaling);
    cpi(k->has_l, has_l); cpi(k->has_u, has_u); cpi(k->pos_l, pos_l); cpi(k->pos_u, pos_u);
    cpi(k->h_l_idx, h_l_idx); cpi(k->h_u_idx, h_u_idx); cpi(k->x_l_idx, x_l_idx); cpi(k->x_u_idx, x_u_idx);
    stream_wait(k->st_);
    return k;
}

double KKTSystem::read_scalar_max(int slot)
{
    PQ_HIP(hipMemcpyAsync(scal_h.p, scal_d.p, sizeof(unsigned long long) * 4, hipMemcpyDeviceToHost, st_));
    stream_wait(st_);
    double v;
    std::memcpy(&v, &scal_h.p[slot], sizeof(double));
    return v;
}

// kkt_system.hpp:143-211
bool KKTSystem::update_scalings_and_factor(bool iterative_refinement, double rho, double delta, const pq_vars& vars)
{
    PQ_ZONE("piqp_amd::KKTSystem::update_scalings_and_factor");
    PQ_HIP(hipSetDevice(dev_));
    const int n = n_, m = m_;
    double* m_z_reg_iter_ref = work_z.p;
    m_rho = rho;
    m_delta = delta;
    {
        const int cnt = std::max(std::max(n, m), std::max(n_x_l, n_x_u));
        LAUNCH1(k_scalings_fused, cnt, st_, n, m, n_x_l, n_x_u, rho, delta, vars.s_l, vars.z_l, vars.s_u, vars.z_u, vars.s_bl, vars.z_bl, vars.s_bu, vars.z_bu, has_l.p, has_u.p, pos_l.p,
                pos_u.p, x_b_scaling.p, m_s_l.p, m_z_l_inv.p, m_s_u.p, m_z_u_inv.p, m_s_bl.p, m_z_bl_inv.p, m_s_bu.p, m_z_bu_inv.p, m_x_reg.p, m_z_reg.p, m_z_reg_iter_ref);
    }

    double delta_reg = delta;
    if (iterative_refinement) {
        // :197-206  max_diag = max(|P_diag + x_reg|_inf, |z_reg|_inf)
        PQ_HIP(hipMemsetAsync(scal_d.p, 0, sizeof(unsigned long long) * 4, st_));
        LAUNCH1(k_absmax, n, st_, n, kkt_solver->P_diag_device(), m_x_reg.p, scal_d.p);
        LAUNCH1(k_absmax, m, st_, m, m_z_reg_iter_ref, (const double*)nullptr, scal_d.p);
        const double max_diag = read_scalar_max(0);
        const double reg = settings_.iterative_refinement_static_regularization_eps + settings_.iterative_refinement_static_regularization_rel * max_diag;
        delta_reg += reg;
        LAUNCH1(k_add_scalar, n, st_, n, reg, m_x_reg.p);
        LAUNCH1(k_add_scalar, m, st_, m, reg, m_z_reg_iter_ref);
    }
    use_iterative_refinement = iterative_refinement;
    return kkt_solver->update_scalings_and_factor(delta_reg, m_x_reg.p, m_z_reg_iter_ref);
}

// kkt_system.hpp:507-536; returns ||err||_inf (and leaves err in err_x/err_y/err_z)
double KKTSystem::get_refine_error(const double* lhs_x, const double* lhs_y, const double* lhs_z, const double* rhs_x, const double* rhs_y, const double* rhs_z,
                                   double* err_x, double* err_y, double* err_z)
{
    PQ_ZONE("piqp_amd::KKTSystem::get_refine_error");
    const int n = n_, p = p_, m = m_;
    {   // a stage-partitioned backend evaluates the residual on its own rows only and all-reduces the norm (kkt_solver_base.hpp)
        double nrm = 0.0;
        if (kkt_solver->refine_error_sharded(lhs_x, lhs_y, lhs_z, rhs_x, rhs_y, rhs_z, m_x_reg.p, m_delta, m_z_reg.p, err_x, err_y, err_z, &nrm)) return nrm;
    }
    // mul_condensed_kkt: Px -> work_x ; A x -> work_y, AT y -> work_x2 ; G x -> work_z2, GT z -> work_x3
    kkt_solver->eval_P_x(1.0, lhs_x, work_x.p);
    kkt_solver->eval_A_xn_and_AT_xt(1.0, 1.0, lhs_x, lhs_y, work_y.p, work_x2.p);
    kkt_solver->eval_G_xn_and_GT_xt(1.0, 1.0, lhs_x, lhs_z, work_z2.p, work_x3.p);
    LAUNCH1(k_err_x, n, st_, n, rhs_x, work_x.p, m_x_reg.p, lhs_x, work_x2.p, work_x3.p, err_x);
    LAUNCH1(k_err_yz, p, st_, p, rhs_y, work_y.p, m_delta, (const double*)nullptr, lhs_y, err_y);
    LAUNCH1(k_err_yz, m, st_, m, rhs_z, work_z2.p, 0.0, m_z_reg.p, lhs_z, err_z);
    PQ_HIP(hipMemsetAsync(scal_d.p, 0, sizeof(unsigned long long) * 4, st_));
    LAUNCH1(k_absmax, n, st_, n, err_x, (const double*)nullptr, scal_d.p);
    LAUNCH1(k_absmax, p, st_, p, err_y, (const double*)nullptr, scal_d.p);
    LAUNCH1(k_absmax, m, st_, m, err_z, (const double*)nullptr, scal_d.p);
    return read_scalar_max(0);
}

static inline void d2d(double* dst, const double* src, int n, hipStream_t s)
{
    if (n > 0 && dst != src) PQ_HIP(hipMemcpyAsync(dst, src, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
}

// kkt_system.hpp:213-369
bool KKTSystem::solve(const pq_vars& rhs, pq_vars& lhs)
{
    PQ_ZONE("piqp_amd::KKTSystem::solve");
    PQ_HIP(hipSetDevice(dev_));
    const int n = n_, p = p_, m = m_;
    double* lhs_z = lhs_z_buf.p;  // the reference aliases work_z; a dedicated buffer avoids the z_reg_iter_ref alias hazard
    last_refine_steps = 0; last_backend_solves = 0; last_refine_error = 0.0; last_rhs_norm = 0.0;

    LAUNCH1(k_rhs_bars_fused, std::max(std::max(n, m), std::max(p, 1)), st_, n, m, m_delta, has_l.p, has_u.p, m_s_l.p, m_z_l_inv.p, m_s_u.p, m_z_u_inv.p, m_z_reg.p, rhs.z_l, rhs.s_l, rhs.z_u,
            rhs.s_u, pos_l.p, pos_u.p, x_b_scaling.p, m_s_bl.p, m_z_bl_inv.p, m_s_bu.p, m_z_bu_inv.p, rhs.x, rhs.z_bl, rhs.s_bl, rhs.z_bu, rhs.s_bu, rhs_z_bar.p, rhs_x_bar.p,
            scal_d.p, p, rhs.y, rhs_y_keep.p);

    kkt_solver->solve(rhs_x_bar.p, rhs.y, rhs_z_bar.p, lhs.x, lhs.y, lhs_z);
    last_backend_solves++;
    last_rhs_y = rhs_y_keep.p;

    if (use_iterative_refinement) {
        // :259  rhs_norm
        PQ_HIP(hipMemsetAsync(scal_d.p, 0, sizeof(unsigned long long) * 4, st_));
        LAUNCH1(k_absmax, n, st_, n, rhs_x_bar.p, (const double*)nullptr, scal_d.p);
        LAUNCH1(k_absmax, p, st_, p, rhs.y, (const double*)nullptr, scal_d.p);
        LAUNCH1(k_absmax, m, st_, m, rhs_z_bar.p, (const double*)nullptr, scal_d.p);
        const double rhs_norm = read_scalar_max(0);
        last_rhs_norm = rhs_norm;

        double refine_error = get_refine_error(lhs.x, lhs.y, lhs_z, rhs_x_bar.p, rhs.y, rhs_z_bar.p, ref_err_x.p, ref_err_y.p, ref_err_z.p);
        last_refine_error = refine_error;
        if (!std::isfinite(refine_error)) return false;

        PQ_ZONE("piqp_amd::KKTSystem::solve::iterative_refinement");
        for (int i = 0; i < settings_.iterative_refinement_max_iter; i++) {
            if (refine_error <= settings_.iterative_refinement_eps_abs + settings_.iterative_refinement_eps_rel * rhs_norm) break;
            const double prev_refine_error = refine_error;

            kkt_solver->solve(ref_err_x.p, ref_err_y.p, ref_err_z.p, ref_lhs_x.p, ref_lhs_y.p, ref_lhs_z.p);
            last_backend_solves++;
            last_refine_steps++;
            LAUNCH1(k_add_inplace, n, st_, n, lhs.x, ref_lhs_x.p);
            LAUNCH1(k_add_inplace, p, st_, p, lhs.y, ref_lhs_y.p);
            LAUNCH1(k_add_inplace, m, st_, m, lhs_z, ref_lhs_z.p);

            refine_error = get_refine_error(ref_lhs_x.p, ref_lhs_y.p, ref_lhs_z.p, rhs_x_bar.p, rhs.y, rhs_z_bar.p, ref_err_x.p, ref_err_y.p, ref_err_z.p);
            if (!std::isfinite(refine_error)) return false;

            const double improvement_rate = prev_refine_error / refine_error;
            // the reference swaps lhs <-> ref_lhs (:292-300); callers of this ABI own fixed buffers, so the
            // accepted iterate is copied into them instead (ref_lhs is overwritten by the next solve anyway)
            if (improvement_rate < settings_.iterative_refinement_min_improvement_rate) {
                if (improvement_rate > 1.0) {
                    d2d(lhs.x, ref_lhs_x.p, n, st_); d2d(lhs.y, ref_lhs_y.p, p, st_); d2d(lhs_z, ref_lhs_z.p, m, st_);
                    last_refine_error = refine_error;
                }
                break;
            }
            d2d(lhs.x, ref_lhs_x.p, n, st_); d2d(lhs.y, ref_lhs_y.p, p, st_); d2d(lhs_z, ref_lhs_z.p, m, st_);
            last_refine_error = refine_error;
        }
        // (a stage-partitioned condensed backend refined the eliminated multipliers on their owner ranks only: one all-gather per solve, kkt_solver_base.hpp)
        if (last_refine_steps > 0) kkt_solver->finish_sharded_solve(lhs.y, lhs_z);
        LAUNCH1(k_dual_recovery, m, st_, m, m_delta, has_l.p, has_u.p, m_s_l.p, m_z_l_inv.p, m_s_u.p, m_z_u_inv.p, m_z_reg.p, lhs_z, rhs.z_l, rhs.s_l, rhs.z_u, rhs.s_u, lhs.z_l,
                lhs.z_u, lhs.s_l, lhs.s_u);
        LAUNCH1(k_box_recovery, n_x_l, st_, n_x_l, -1.0, m_delta, x_l_idx.p, x_b_scaling.p, lhs.x, m_s_bl.p, m_z_bl_inv.p, rhs.z_bl, rhs.s_bl, lhs.z_bl, lhs.s_bl);
        LAUNCH1(k_box_recovery, n_x_u, st_, n_x_u, 1.0, m_delta, x_u_idx.p, x_b_scaling.p, lhs.x, m_s_bu.p, m_z_bu_inv.p, rhs.z_bu, rhs.s_bu, lhs.z_bu, lhs.s_bu);
        // the results must be complete when the call returns (the other branch ends with the read-back of its finiteness check): found in round 4 by ranks sharing
        // one GPU -- a caller that read lhs.z_bu through another stream right away saw it before the last recovery kernel had run, one run in three
        stream_wait(st_);
    } else {
        // :305 allFinite check (NaN-propagating |.|_inf is finite iff every entry is) + the recoveries, one launch; scal_d was zeroed by
        // k_rhs_bars_fused
        const int cnt = std::max(std::max(std::max(n, p), m), std::max(n_x_l, n_x_u));
        LAUNCH1(k_recover_fused, cnt, st_, n, p, m, n_x_l, n_x_u, m_delta, has_l.p, has_u.p, m_s_l.p, m_z_l_inv.p, m_s_u.p, m_z_u_inv.p, m_z_reg.p, lhs_z, rhs.z_l, rhs.s_l, rhs.z_u,
                rhs.s_u, lhs.z_l, lhs.z_u, lhs.s_l, lhs.s_u, x_l_idx.p, x_u_idx.p, x_b_scaling.p, lhs.x, lhs.y, m_s_bl.p, m_z_bl_inv.p, rhs.z_bl, rhs.s_bl, lhs.z_bl, lhs.s_bl, m_s_bu.p,
                m_z_bu_inv.p, rhs.z_bu, rhs.s_bu, lhs.z_bu, lhs.s_bu, scal_d.p);
        finite_check_pending = true;
    }

    if (finite_check_pending) {
        finite_check_pending = false;
        // one read-back per solve (the reference returns this bool; its caller ignores it, solver.hpp:487,732,765)
        const double mx = read_scalar_max(0);
        if (!std::isfinite(mx)) return false;
    }
    return true;
}

// ||rhs_bar - K_cond lhs||_inf of the last solve (parity metric of the harness)
void KKTSystem::condensed_residual(const double* lhs_x, const double* lhs_y, double* res_inf, double* rhs_inf)
{
    PQ_HIP(hipSetDevice(dev_));
    PQ_HIP(hipMemsetAsync(scal_d.p, 0, sizeof(unsigned long long) * 4, st_));
    LAUNCH1(k_absmax, n_, st_, n_, rhs_x_bar.p, (const double*)nullptr, scal_d.p);
    LAUNCH1(k_absmax, p_, st_, p_, last_rhs_y, (const double*)nullptr, scal_d.p);
    LAUNCH1(k_absmax, m_, st_, m_, rhs_z_bar.p, (const double*)nullptr, scal_d.p);
    *rhs_inf = read_scalar_max(0);
    *res_inf = get_refine_error(lhs_x, lhs_y, lhs_z_buf.p, rhs_x_bar.p, last_rhs_y, rhs_z_bar.p, ref_err_x.p, ref_err_y.p, ref_err_z.p);
}

// kkt_system.hpp:392-425
void KKTSystem::mul(const pq_vars& lhs, pq_vars& rhs)
{
    PQ_HIP(hipSetDevice(dev_));
    const int n = n_, p = p_, m = m_;
    kkt_solver->eval_P_x(1.0, lhs.x, work_x.p);
    kkt_solver->eval_A_xn_and_AT_xt(1.0, 1.0, lhs.x, lhs.y, work_y.p, work_x2.p);
    LAUNCH1(k_sub, m, st_, m, lhs.z_u, lhs.z_l, work_z.p);  // z_u - z_l
    kkt_solver->eval_G_xn_and_GT_xt(1.0, 1.0, lhs.x, work_z.p, work_z2.p, work_x3.p);
    LAUNCH1(k_mul_x, n, st_, n, m_rho, work_x.p, lhs.x, work_x2.p, work_x3.p, pos_l.p, pos_u.p, x_b_scaling.p, lhs.z_bl, lhs.z_bu, rhs.x);
    LAUNCH1(k_mul_y, p, st_, p, m_delta, work_y.p, lhs.y, rhs.y);
    LAUNCH1(k_mul_z, m, st_, m, m_delta, work_z2.p, lhs.z_l, lhs.z_u, lhs.s_l, lhs.s_u, m_s_l.p, m_s_u.p, m_z_l_inv.p, m_z_u_inv.p, rhs.z_l, rhs.z_u, rhs.s_l, rhs.s_u);
    LAUNCH1(k_mul_box, n_x_l, st_, n_x_l, -1.0, m_delta, x_l_idx.p, x_b_scaling.p, lhs.x, lhs.z_bl, lhs.s_bl, m_s_bl.p, m_z_bl_inv.p, rhs.z_bl, rhs.s_bl);
    LAUNCH1(k_mul_box, n_x_u, st_, n_x_u, 1.0, m_delta, x_u_idx.p, x_b_scaling.p, lhs.x, lhs.z_bu, lhs.s_bu, m_s_bu.p, m_z_bu_inv.p, rhs.z_bu, rhs.s_bu);
}

}  // namespace pq
