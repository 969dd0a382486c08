// piqp_amd/csrc/batch_solver.hip -- batched interior-point solves of structurally identical sparse QPs with the
// sparse_multistage KKT backend: ONE workgroup runs the WHOLE proximal interior-point method of ONE QP
// (reference solver.hpp:379-1259 solve_impl + kkt_system.hpp + sparse/multistage_kkt.hpp) inside a single kernel
// launch; the batch dimension is the grid.  Independent QPs share nothing, so there is no lock-step, no host
// round trip per iteration and no collective: a QP that converges early simply retires its workgroup.
//
// The reference has no batch API (one SolverBase per QP, solver.hpp:42); `pq_batch_*` is the device-side
// equivalent of looping `SparseSolver::setup(...); solve();` over the instances.  Per instance the arithmetic
// follows the single-QP path operation by operation:
//   setup (host): sparse::Data + RuizEquilibration::scale_data exactly as Solver::setup (solver.hpp:151-216)
//   device:       solve_impl: initial factor/solve, shift to the interior, predictor-corrector loop with the
//                 proximal updates of rho/delta, KKTSystem scalings / condensed right-hand sides / iterative refinement /
//                 dual recovery, multistage chain Cholesky (msdev::factor_chain) and block substitution
//                 (msdev::solve_chain), residuals and termination tests; unscale + restore_dual at the end.
// All instances must share the sparsity patterns AND the set of finite bounds (checked at setup).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <thread>
#include <type_traits>

#include "multistage_device.hpp"
#include "multistage_symbolic.hpp"
#include "solver.hpp"
#include "sparse_ops.hpp"

namespace pq {

namespace {

using msdev::GroupMeta;
using msdev::MsMeta;
using msdev::PackedMeta;

constexpr int LDS_LIMIT_BYTES = 152 * 1024;  // dynamic LDS budget (the kernel also keeps ~3 KB of static LDS)
constexpr int RESIDENT_LIMIT_BYTES = 40 * 1024;  // keeps >= 4 workgroups per CU in resident mode

// fields of a Variables set (variables.hpp:19-105)
enum { FX = 0, FY, FZL, FZU, FZBL, FZBU, FSL, FSU, FSBL, FSBU, NF };
// per-instance arena slots
enum {
    D_PX = 0, D_ATX, D_GTX, D_C, D_B, D_HL, D_HU, D_XL, D_XU, D_XBS, D_DL, D_DLI, D_DB, D_DBI,
    V_R, V_NR = V_R + NF, V_RS = V_NR + NF, V_ST = V_RS + NF, V_PX = V_ST + NF,
    K_SL = V_PX + NF, K_SU, K_SBL, K_SBU, K_ZLI, K_ZUI, K_ZBLI, K_ZBUI, K_XREG, K_ZREG, K_ZREGR, K_RXB, K_RZB, K_WX, K_LZ, K_EX, K_EY, K_EZ, K_RLX, K_RLY, K_RLZ,
    B_ZINV, B_PF, B_ATAF, B_F, B_PAN, B_XA, B_XG,
    NSLOT
};

struct BatchShared {
    int n, p, m, n_h_l, n_h_u, n_x_l, n_x_u;
    const int *h_l_idx, *h_u_idx, *x_l_idx, *x_u_idx, *has_l, *has_u, *pos_l, *pos_u;
    const int *Pf_p, *Pf_i, *Pf_src, *AT_p, *AT_i, *A_p, *A_i, *A_src, *GT_p, *GT_i, *G_p, *G_i, *G_src;
    int nzP, nzA, nzG;
    MsMeta M;
    GroupMeta GA, GG;
    const long long *P_dst, *A_dst, *G_dst;
    const int *ent_b, *ent_rc;  // lower-triangular entries of all fronts (flat assembly)
    const int *ent_at, *ent_xr;  // ... as offsets into the front arena / index of the x_reg entry added there (-1: none)
    int ent_flat;
    int n_ent;
    int fcap, lofs, hcap, chain_lds_doubles;
    int chain_rows;  // the uniform run of the chain is FACTORED with one lane per row of a front (msdev::factor_chain_rows) instead of one lane per entry
    int chain_reg_w, chain_reg_k, chain_reg_nst;  // chain_reg_w > 0: stages 0 .. chain_reg_k - 1 of the chain_reg_nst stages form a uniform gap-free chain without arrow -- register-carried substitution (msdev::solve_chain_wave_reg)
    int meta_ofs;  // LDS offset (doubles) of the per-stage structure tables copied in at kernel start
    int res_f, res_pan, res_x, res_chain;  // MODE_RESIDENT: LDS offsets (doubles) of the fronts, the factor panels, the solve vector, chain scratch
    long long off[NSLOT];
    long long stride;
    pq_settings set;
};

// Everything in the per-instance arena and every shared index array is device memory, and the interior-point code says so in its pointer types: through
// generic pointers (what a pointer read from a struct or passed to an out-of-line member is to the compiler) every access was a FLAT instruction, which
// counts on the LDS counter as well -- each read of the descriptor or the solver state (LDS) then also waited for all vector loads in flight.
typedef __attribute__((address_space(1))) double gdbl;
template <class T>
__device__ __forceinline__ const __attribute__((address_space(1))) T* g(const T* p) { return (const __attribute__((address_space(1))) T*)p; }
__device__ __forceinline__ double* gen(gdbl* p) { return (double*)p; }              // for the routines shared with the LDS-resident modes
__device__ __forceinline__ const double* gen(const gdbl* p) { return (const double*)p; }

// ---- workgroup collectives ------------------------------------------------------------------------------------
struct OpSum { __device__ double operator()(double a, double b) const { return a + b; } };
struct OpMax { __device__ double operator()(double a, double b) const { return a < b ? b : a; } };            // std::max
struct OpMin { __device__ double operator()(double a, double b) const { return b < a ? b : a; } };            // std::min
struct OpAbsMaxNan { __device__ double operator()(double r, double a) const { return (a > r || a != a) ? a : r; } };  // Eigen lpNorm<Infinity>: NaN propagates

// value of `v` in the lane the DPP control selects (quad_perm / row_half_mirror / row_mirror: all lanes have a source)
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// A sum depends on its order: the xor butterfly the parity tests were pinned with, `for (o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o)`.  The same tree -- at every
// level every lane adds the value of the same partner lane, and an addition commutes, so the sum is the butterfly's bit for bit -- without the six round trips
// through the LDS crossbar: the two halves / the odd and even rows of 16 lanes through the gfx950 lane swaps, the rest inside a row through DPP.
__device__ __forceinline__ double wave_sum_butterfly(double v)
{
    {   // lane ^ 32: v_permlane32_swap leaves one half of the wave's values in every lane of its first operand and the other half in its second
        const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
        v = __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
    }
    {   // lane ^ 16: odd rows of the first operand against even rows of the second
        const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
        v = __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
    }
    v += dpp_move<0x128>(v);  // lane ^ 8: row_ror:8
    {   // lane ^ 4: the lanes 0-3 and 8-11 of a row read four lanes up (row_shl:4, banks 0 and 2), the others four lanes down (row_shr:4, banks 1 and 3)
        const int l = __double2loint(v), h = __double2hiint(v);
        int pl = __builtin_amdgcn_update_dpp(l, l, 0x104, 0xf, 0x5, false), ph = __builtin_amdgcn_update_dpp(h, h, 0x104, 0xf, 0x5, false);
        pl = __builtin_amdgcn_update_dpp(pl, l, 0x114, 0xf, 0xa, false); ph = __builtin_amdgcn_update_dpp(ph, h, 0x114, 0xf, 0xa, false);
        v += __hiloint2double(ph, pl);
    }
    v += dpp_move<0x4e>(v);  // lane ^ 2: quad_perm [2, 3, 0, 1]
    v += dpp_move<0xb1>(v);  // lane ^ 1: quad_perm [1, 0, 3, 2]
    return v;
}
template <int NT, class Op>
__device__ __forceinline__ double wg_reduce(double v, Op op, double* red)
{
    if constexpr (std::is_same<Op, OpSum>::value) {
        v = wave_sum_butterfly(v);
    } else {
        // maxima and minima do not: four DPP steps inside each row of 16 lanes, then the four rows through v_readlane -- ~150 cycles instead of ~800
        v = op(v, dpp_move<0xb1>(v));   // quad_perm [1, 0, 3, 2]
        v = op(v, dpp_move<0x4e>(v));   // quad_perm [2, 3, 0, 1]
        v = op(v, dpp_move<0x141>(v));  // row_half_mirror
        v = op(v, dpp_move<0x140>(v));  // row_mirror
        auto row = [&](int l) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l)); };
        v = op(op(op(row(0), row(16)), row(32)), row(48));
    }
    if constexpr (NT > 64) {
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        v = red[0];
        for (int i = 1; i < NT / 64; ++i) v = op(v, red[i]);
    }
    return v;
}

// in-kernel stage clock (wall_clock64, 100 MHz): [0] assemble, [1] chain factor, [2] chain solve, [3] KKTSystem::solve total,
// [4] residuals, [5] whole solve
enum { T_ASM = 0, T_FAC, T_CHAIN, T_KS, T_RES, T_ALL, NPROF = 8 };

struct IpmState {
    long long prof[NPROF];
    pq_info info;
    double rz_c, rz_c_inv, ks_rho, ks_delta, be_delta;
    int refine_enabled, ks_use_refine;
};

// MODE: where the multistage chain keeps its working set
//   MODE_HBM       fronts and factor panels in HBM/L2, only the diagonal-block inverse staged in LDS (wide stages)
//   MODE_STAGED    the current stage's front / panel is copied into LDS, the arenas stay in HBM
//   MODE_WAVE      64-thread workgroups, every front has <= 64 entries: factor panels + solve vector resident in LDS, the
//                  assembled fronts stay in HBM and are prefetched one stage ahead into a register; single-wave chain
//                  routines without barriers (msdev::factor_chain_wave / solve_chain_wave)
//   MODE_RESIDENT  ALL fronts, ALL factor panels and the solve vector live in LDS for the whole solve (small QPs:
//                  the chain never waits on HBM)
enum { MODE_HBM = 0, MODE_STAGED = 1, MODE_RESIDENT = 2, MODE_WAVE = 3 };

// WPE (the kernel's waves-per-SIMD setting) is part of the type although nothing in the class reads it: the out-of-line members are then compiled once per
// kernel variant, with that variant's register budget.  Shared between the variants they got the budget of the MOST restrictive one (eight waves per SIMD:
// 64 VGPRs and 184 bytes of spills in be_factor, whatever the launched kernel allowed).
template <int NT, int MODE, int WPE>
struct Ipm {
    static constexpr bool LDS = MODE == MODE_STAGED;
    static constexpr bool RES = MODE == MODE_RESIDENT;
    static constexpr bool WAVE = MODE == MODE_WAVE;
    const BatchShared& S;
    gdbl* base;
    double* sm;   // chain workspace (dynamic LDS)
    double* red;  // reduction scratch
    // Scalar solver state.  Every thread computes the same scalars; they are kept ONCE PER WAVE in LDS instead of in
    // per-lane private memory (a 350-byte Info per lane would turn into scratch and throttle occupancy).  A wave only
    // ever touches its own copy, so there are no cross-wave hazards.
    IpmState& st;
    pq_info& info;
    double& rz_c;
    double& rz_c_inv;
    double& ks_rho;
    double& ks_delta;
    double& be_delta;
    int& refine_enabled;
    int& ks_use_refine;

    __device__ Ipm(const BatchShared& s, gdbl* b, double* sm_, double* red_, IpmState& state)
        : S(s), base(b), sm(sm_), red(red_), st(state), info(state.info), rz_c(state.rz_c), rz_c_inv(state.rz_c_inv), ks_rho(state.ks_rho), ks_delta(state.ks_delta),
          be_delta(state.be_delta), refine_enabled(state.refine_enabled), ks_use_refine(state.ks_use_refine)
    {
    }

    // The out-of-line member functions see everything through generic pointers (FLAT loads / stores, which also tie the LDS and the memory wait
    // counters together).  What lives in LDS -- the shared descriptor, the solver state -- is stated as an assumption; address-space inference
    // then turns those accesses into ds_read / ds_write (8192 QPs: 9.2 -> 8.8 ms for the descriptor alone).
    __device__ __forceinline__ gdbl* arena_ptr(long long off) const
    {
        return base + off;  // (stating "neither LDS nor private" as an assumption does not make these global_load with this compiler; they stay FLAT)
    }
    __device__ __forceinline__ const BatchShared& shared() const
    {
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_assume(__builtin_amdgcn_is_shared((const void*)&S));
#endif
        return S;
    }
    // the solver state, the reduction scratch and the chain workspace live in LDS as well; called first thing in every out-of-line member
    __device__ __forceinline__ void assume_lds() const
    {
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_assume(__builtin_amdgcn_is_shared((const void*)&S));
        __builtin_assume(__builtin_amdgcn_is_shared((const void*)&st));
        __builtin_assume(__builtin_amdgcn_is_shared((const void*)&info));
        __builtin_assume(__builtin_amdgcn_is_shared((const void*)red));
        __builtin_assume(__builtin_amdgcn_is_shared((const void*)sm));
        __builtin_assume(__builtin_amdgcn_is_shared((const void*)&rz_c));
        __builtin_assume(__builtin_amdgcn_is_shared((const void*)&rz_c_inv));
        __builtin_assume(__builtin_amdgcn_is_shared((const void*)&ks_rho));
        __builtin_assume(__builtin_amdgcn_is_shared((const void*)&ks_delta));
        __builtin_assume(__builtin_amdgcn_is_shared((const void*)&be_delta));
        __builtin_assume(__builtin_amdgcn_is_shared((const void*)&refine_enabled));
        __builtin_assume(__builtin_amdgcn_is_shared((const void*)&ks_use_refine));
#endif
    }
    __device__ __forceinline__ gdbl* at(int slot) const { return arena_ptr(shared().off[slot]); }
    __device__ __forceinline__ gdbl* v(int set, int f) const { return arena_ptr(shared().off[set + f]); }
    __device__ __forceinline__ int tid() const { return threadIdx.x; }

    template <class Op>
    __device__ __forceinline__ double reduce(double x, Op op) const { return wg_reduce<NT>(x, op, red); }

    // ---- element-wise passes ---------------------------------------------------------------------------------------------------
    // f(i, x) for every i < cnt with x[q] = in[q][i]: the lane's two next trips (i, i + NT) are loaded together -- the second from a clamped address when it
    // does not exist -- before f stores anything.  A plain lane-strided loop waits for memory once per trip (the compiler cannot move the second trip's loads
    // over the first trip's stores: any arena vector may alias any other), and a wave of this kernel spends three quarters of its life in such waits
    // (profiles/r03_pmc_batch_c4.txt).  The arithmetic per element is whatever f does: unchanged.  f must not read, through other pointers, what an earlier
    // call of f in the same pass has written (element-wise passes do not).
    template <int NIN, class F>
    __device__ __forceinline__ void map2(int cnt, const gdbl* const (&in)[NIN], F f) const
    {
        for (int i0 = tid(); i0 < cnt; i0 += 2 * NT) {
            const int i1 = i0 + NT;
            const bool ok1 = i1 < cnt;
            const int j1 = ok1 ? i1 : i0;
            double x0[NIN], x1[NIN];
#pragma unroll
            for (int q = 0; q < NIN; ++q) { x0[q] = in[q][i0]; x1[q] = in[q][j1]; }
            f(i0, x0);
            if (ok1) f(i1, x1);
        }
    }

    // ---- mat-vecs (sparse/kkt.hpp:179-203 / multistage_kkt.hpp:291-383), ending with a barrier --------------------
    __device__ void eval_P_x(double alpha, const gdbl* x, gdbl* z) const
    {
        const gdbl* Px = at(D_PX);
        for (int j = tid(); j < S.n; j += NT) {
            double s = 0.0;
            for (int q = g(S.Pf_p)[j]; q < g(S.Pf_p)[j + 1]; ++q) s += Px[g(S.Pf_src)[q]] * x[g(S.Pf_i)[q]];
            z[j] = alpha * s;
        }
        __syncthreads();
    }
    __device__ void eval_A(double an, double at_, const gdbl* xn, const gdbl* xt, gdbl* zn, gdbl* zt) const
    {
        const gdbl* Ax = at(D_ATX);
        for (int k = tid(); k < S.p; k += NT) {
            double s = 0.0;
            for (int q = g(S.AT_p)[k]; q < g(S.AT_p)[k + 1]; ++q) s += Ax[q] * xn[g(S.AT_i)[q]];
            zn[k] = an * s;
        }
        for (int j = tid(); j < S.n; j += NT) {
            double s = 0.0;
            for (int q = g(S.A_p)[j]; q < g(S.A_p)[j + 1]; ++q) s += Ax[g(S.A_src)[q]] * xt[g(S.A_i)[q]];
            zt[j] = at_ * s;
        }
        __syncthreads();
    }
    __device__ void eval_G(double an, double at_, const gdbl* xn, const gdbl* xt, gdbl* zn, gdbl* zt) const
    {
        const gdbl* Gx = at(D_GTX);
        for (int k = tid(); k < S.m; k += NT) {
            double s = 0.0;
            for (int q = g(S.GT_p)[k]; q < g(S.GT_p)[k + 1]; ++q) s += Gx[q] * xn[g(S.GT_i)[q]];
            zn[k] = an * s;
        }
        for (int j = tid(); j < S.n; j += NT) {
            double s = 0.0;
            for (int q = g(S.G_p)[j]; q < g(S.G_p)[j + 1]; ++q) s += Gx[g(S.G_src)[q]] * xt[g(S.G_i)[q]];
            zt[j] = at_ * s;
        }
        __syncthreads();
    }

    // ---- multistage backend (multistage_kkt.hpp:180-288) -----------------------------------------------------
    // LDS-derived views: built from the `extern __shared__` symbol (not from the generic pointers stored in this object) so
    // that address-space inference turns the chain's accesses into ds_read / ds_write
    static __device__ __forceinline__ PackedMeta packed_meta(const BatchShared& s, double* dyn)
    {
        const int* mi = reinterpret_cast<const int*>(dyn + s.meta_ofs);
        return PackedMeta{s.M.N, s.M.arrow, s.M.n, mi, reinterpret_cast<const long long*>(mi + 4 * s.M.N)};
    }

    // (be_factor / be_solve / refine_error, the scalings and the two residual updates are inlined into their callers -- ks_solve, with three call sites, is not:
    // inlined too, the chain substitution spills, 7.8 ms --: an out-of-line member that uses the kernel's 96 VGPRs saves and restores ~30
    // callee-saved registers per call through scratch -- 64 KB of scratch per wave times 4 600 waves in flight is far beyond the L2, so that was HBM traffic:
    // 18.7 -> 16.6 GB per launch and 7.8 -> 7.2 ms with these three inlined.  Inlining EVERYTHING into one function was measured too: 9.1 ms, the chain
    // substitution spills inside a 240 KB function.)
    __device__ __forceinline__ void be_factor(double delta, const gdbl* x_reg, const gdbl* z_reg)
    {
        assume_lds();
        extern __shared__ double dyn[];
        const PackedMeta PM = packed_meta(S, dyn);
        gdbl* zinv = at(B_ZINV);
        for (int i = tid(); i < S.m; i += NT) zinv[i] = 1.0 / z_reg[i];
        __syncthreads();
        be_delta = delta;
        const double delta_inv = 1.0 / delta;
        const long long t0 = wall_clock64();
        long long t1;
        if constexpr (WAVE) {  // fronts in the arena (device memory), factor panels in LDS
            gdbl* F = at(B_F);
            if (S.m == 0 && S.ent_flat) {
                // no inequality rows: an entry is P + delta^-1 A'A (+ x_reg on the pivots) -- offsets and x_reg indices come precomputed, two trips at once
                const gdbl*Pf = at(B_PF), *AtAf = at(B_ATAF);
                const auto eat = g(S.ent_at), exr = g(S.ent_xr);
                for (int e0 = tid(); e0 < S.n_ent; e0 += 2 * NT) {
                    const int e1 = e0 + NT;
                    const bool ok1 = e1 < S.n_ent;
                    const int ee[2] = {e0, ok1 ? e1 : e0};
                    int a[2], xr[2];
                    double pv[2], av[2], rv[2];
#pragma unroll
                    for (int r = 0; r < 2; ++r) { a[r] = eat[ee[r]]; xr[r] = exr[ee[r]]; }
#pragma unroll
                    for (int r = 0; r < 2; ++r) { pv[r] = Pf[a[r]]; av[r] = AtAf[a[r]]; rv[r] = x_reg[xr[r] > 0 ? xr[r] : 0]; }
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        if (r == 1 && !ok1) break;
                        double v = pv[r] + delta_inv * av[r] + 0.0;  // (+ s with an empty row sum s = 0.0, as assemble_flat adds it)
                        if (xr[r] >= 0) v += rv[r];
                        F[a[r]] = v;
                    }
                }
            } else
            msdev::assemble_flat<NT>(PM, S.GG, at(B_XG), at(B_PF), at(B_ATAF), zinv, x_reg, delta_inv, F, g(S.ent_b), g(S.ent_rc), S.n_ent);
            __syncthreads();
            t1 = wall_clock64();
            msdev::factor_chain_wave<(WPE <= 3)>(PM, (msdev::global_cdouble*)F, dyn + S.res_pan, S.chain_reg_w > 0 ? S.chain_reg_k : 0, S.chain_rows != 0 ? dyn + S.res_x : nullptr);  // (the solve vector's LDS is free during a factorisation)
        } else {
            double* F = RES ? dyn + S.res_f : gen(at(B_F));
            double* PAN = RES ? dyn + S.res_pan : gen(at(B_PAN));
            double* CH = RES ? dyn + S.res_chain : dyn;
            msdev::assemble_flat<NT>(PM, S.GG, at(B_XG), at(B_PF), at(B_ATAF), zinv, x_reg, delta_inv, F, g(S.ent_b), g(S.ent_rc), S.n_ent);
            __syncthreads();
            t1 = wall_clock64();
            msdev::factor_chain<NT, LDS>(PM, F, PAN, CH, S.fcap, S.lofs, 1);
        }
        __syncthreads();
        const long long t2 = wall_clock64();
        st.prof[T_ASM] += t1 - t0; st.prof[T_FAC] += t2 - t1;
        info.n_factor++;
    }
    // XW: the vector the chain works on -- an LDS copy of x in the resident / single-wave modes, lhs_x itself (arena) otherwise
    template <class XW>
    __device__ __forceinline__ void be_solve_with(XW xw, const gdbl* rhs_x, const gdbl* rhs_y, const gdbl* rhs_z, gdbl* lhs_x, gdbl* lhs_y, gdbl* lhs_z)
    {
        extern __shared__ double dyn[];
        const PackedMeta PM = packed_meta(S, dyn);
        const gdbl* zinv = at(B_ZINV);
        const gdbl* Ax = at(D_ATX);
        const gdbl* Gx = at(D_GTX);
        const double delta_inv = 1.0 / be_delta;
        for (int j = tid(); j < S.n; j += NT) {
            double sg = 0.0, sa = 0.0;
            for (int q = g(S.G_p)[j]; q < g(S.G_p)[j + 1]; ++q) { const int i = g(S.G_i)[q]; sg += Gx[g(S.G_src)[q]] * (zinv[i] * rhs_z[i]); }
            for (int q = g(S.A_p)[j]; q < g(S.A_p)[j + 1]; ++q) sa += Ax[g(S.A_src)[q]] * rhs_y[g(S.A_i)[q]];
            xw[j] = (rhs_x[j] + sg) + delta_inv * sa;
        }
        __syncthreads();
        const long long t0 = wall_clock64();
        if constexpr (WAVE) { msdev::solve_chain_wave(PM, dyn + S.res_pan, xw, S.chain_reg_w, S.chain_reg_k, S.chain_reg_nst); __syncthreads(); }
        else if constexpr (RES) msdev::solve_chain<NT, LDS>(PM, dyn + S.res_pan, xw, dyn + S.res_chain, S.hcap);
        else msdev::solve_chain<NT, LDS>(PM, gen(at(B_PAN)), gen(xw), dyn, S.hcap);
        st.prof[T_CHAIN] += wall_clock64() - t0;
        if constexpr (RES || WAVE) for (int j = tid(); j < S.n; j += NT) lhs_x[j] = xw[j];
        for (int k = tid(); k < S.p; k += NT) {
            double s = 0.0;
            for (int q = g(S.AT_p)[k]; q < g(S.AT_p)[k + 1]; ++q) s += Ax[q] * xw[g(S.AT_i)[q]];
            lhs_y[k] = delta_inv * s - delta_inv * rhs_y[k];
        }
        for (int i = tid(); i < S.m; i += NT) {
            double s = 0.0;
            for (int q = g(S.GT_p)[i]; q < g(S.GT_p)[i + 1]; ++q) s += Gx[q] * xw[g(S.GT_i)[q]];
            lhs_z[i] = (s - rhs_z[i]) * zinv[i];
        }
        __syncthreads();
        info.n_backend_solve++;
    }
    __device__ __forceinline__ void be_solve(const gdbl* rhs_x, const gdbl* rhs_y, const gdbl* rhs_z, gdbl* lhs_x, gdbl* lhs_y, gdbl* lhs_z)
    {
        assume_lds();
        extern __shared__ double dyn[];
        if constexpr (RES || WAVE) be_solve_with<double*>(dyn + S.res_x, rhs_x, rhs_y, rhs_z, lhs_x, lhs_y, lhs_z);
        else be_solve_with<gdbl*>(lhs_x, rhs_x, rhs_y, rhs_z, lhs_x, lhs_y, lhs_z);
    }

    // ---- KKTSystem (kkt_system.hpp) -----------------------------------------------------------------------------
    // :143-211
    __device__ __forceinline__ bool ks_update_scalings_and_factor(bool iterative_refinement, double rho, double delta)
    {
        assume_lds();
        const int n = S.n, m = S.m;
        ks_rho = rho; ks_delta = delta;
        const gdbl* xbs = at(D_XBS);
        gdbl*s_l = at(K_SL), *s_u = at(K_SU), *s_bl = at(K_SBL), *s_bu = at(K_SBU);
        gdbl*zli = at(K_ZLI), *zui = at(K_ZUI), *zbli = at(K_ZBLI), *zbui = at(K_ZBUI);
        gdbl*x_reg = at(K_XREG), *z_reg = at(K_ZREG), *z_reg_ref = at(K_ZREGR);
        for (int i = tid(); i < m; i += NT) {
            s_l[i] = v(V_R, FSL)[i]; s_u[i] = v(V_R, FSU)[i];
            zli[i] = 1.0 / v(V_R, FZL)[i]; zui[i] = 1.0 / v(V_R, FZU)[i];
        }
        map2<2>(S.n_x_l, {v(V_R, FSBL), v(V_R, FZBL)}, [&](int i, const double (&x)[2]) { s_bl[i] = x[0]; zbli[i] = 1.0 / x[1]; });
        map2<2>(S.n_x_u, {v(V_R, FSBU), v(V_R, FZBU)}, [&](int i, const double (&x)[2]) { s_bu[i] = x[0]; zbui[i] = 1.0 / x[1]; });
        __syncthreads();
        {
            // (two trips at once, the gathered operands through clamped indices: see map2)
            const auto pos_l = g(S.pos_l), pos_u = g(S.pos_u);
            for (int j0 = tid(); j0 < n; j0 += 2 * NT) {
                const int j1 = j0 + NT;
                const bool ok1 = j1 < n;
                const int jj[2] = {j0, ok1 ? j1 : j0};
                int il[2], iu[2];
                double xb[2], zl[2], sl[2], zu[2], su[2];
#pragma unroll
                for (int r = 0; r < 2; ++r) { il[r] = pos_l[jj[r]]; iu[r] = pos_u[jj[r]]; xb[r] = xbs[jj[r]]; }
#pragma unroll
                for (int r = 0; r < 2; ++r) { const int a = il[r] > 0 ? il[r] : 0, b = iu[r] > 0 ? iu[r] : 0; zl[r] = zbli[a]; sl[r] = s_bl[a]; zu[r] = zbui[b]; su[r] = s_bu[b]; }
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    if (r == 1 && !ok1) break;
                    double xr = rho;
                    if (il[r] >= 0) xr += xb[r] * xb[r] / (zl[r] * sl[r] + delta);
                    if (iu[r] >= 0) xr += xb[r] * xb[r] / (zu[r] * su[r] + delta);
                    x_reg[jj[r]] = xr;
                }
            }
        }
        for (int i = tid(); i < m; i += NT) {
            double zr = 0.0;
            if (g(S.has_l)[i]) zr += 1.0 / (zli[i] * s_l[i] + delta);
            if (g(S.has_u)[i]) zr += 1.0 / (zui[i] * s_u[i] + delta);
            zr = 1.0 / zr;
            z_reg[i] = zr; z_reg_ref[i] = zr;
        }
        __syncthreads();
        double delta_reg = delta;
        if (iterative_refinement) {
            const gdbl* Pd = at(D_PX);  // diag(P) gathered below through the symmetrised pattern
            double mx = 0.0;
            for (int j = tid(); j < n; j += NT) {
                double pd = 0.0;
                for (int q = g(S.Pf_p)[j]; q < g(S.Pf_p)[j + 1]; ++q) if (g(S.Pf_i)[q] == j) pd = Pd[g(S.Pf_src)[q]];
                const double a = fabs(pd + x_reg[j]);
                if (a > mx) mx = a;
            }
            double zm = 0.0;
            for (int i = tid(); i < m; i += NT) { const double a = fabs(z_reg_ref[i]); if (a > zm || a != a) zm = a; }
            double max_diag = reduce(mx, OpMax());
            const double zmax = reduce(zm, OpAbsMaxNan());
            if (zmax > max_diag) max_diag = zmax;
            const double reg = S.set.iterative_refinement_static_regularization_eps + S.set.iterative_refinement_static_regularization_rel * max_diag;
            delta_reg += reg;
            __syncthreads();
            for (int j = tid(); j < n; j += NT) x_reg[j] += reg;
            for (int i = tid(); i < m; i += NT) z_reg_ref[i] += reg;
            __syncthreads();
        }
        ks_use_refine = iterative_refinement ? 1 : 0;
        be_factor(delta_reg, x_reg, z_reg_ref);
        return true;  // multistage_kkt.hpp:218
    }

    // :507-536 err = rhs - K_cond * lhs, returns |err|_inf (NaN-propagating)
    __device__ __forceinline__ double refine_error(const gdbl* lx, const gdbl* ly, const gdbl* lz, const gdbl* rx, const gdbl* ry, const gdbl* rz, gdbl* ex, gdbl* ey,
                                   gdbl* ez)
    {
        assume_lds();
        const int n = S.n, p = S.p, m = S.m;
        const gdbl* Px = at(D_PX);
        const gdbl* Ax = at(D_ATX);
        const gdbl* Gx = at(D_GTX);
        const gdbl* x_reg = at(K_XREG);
        const gdbl* z_reg = at(K_ZREG);
        double mx = 0.0;
        for (int j = tid(); j < n; j += NT) {
            double sp = 0.0, sa = 0.0, sg = 0.0;
            for (int q = g(S.Pf_p)[j]; q < g(S.Pf_p)[j + 1]; ++q) sp += Px[g(S.Pf_src)[q]] * lx[g(S.Pf_i)[q]];
            for (int q = g(S.A_p)[j]; q < g(S.A_p)[j + 1]; ++q) sa += Ax[g(S.A_src)[q]] * ly[g(S.A_i)[q]];
            for (int q = g(S.G_p)[j]; q < g(S.G_p)[j + 1]; ++q) sg += Gx[g(S.G_src)[q]] * lz[g(S.G_i)[q]];
            const double e = rx[j] - (((sp + x_reg[j] * lx[j]) + sa) + sg);
            ex[j] = e;
            const double a = fabs(e);
            if (a > mx || a != a) mx = a;
        }
        for (int k = tid(); k < p; k += NT) {
            double s = 0.0;
            for (int q = g(S.AT_p)[k]; q < g(S.AT_p)[k + 1]; ++q) s += Ax[q] * lx[g(S.AT_i)[q]];
            const double e = ry[k] - (s - ks_delta * ly[k]);
            ey[k] = e;
            const double a = fabs(e);
            if (a > mx || a != a) mx = a;
        }
        for (int i = tid(); i < m; i += NT) {
            double s = 0.0;
            for (int q = g(S.GT_p)[i]; q < g(S.GT_p)[i + 1]; ++q) s += Gx[q] * lx[g(S.GT_i)[q]];
            const double e = rz[i] - (s - z_reg[i] * lz[i]);
            ez[i] = e;
            const double a = fabs(e);
            if (a > mx || a != a) mx = a;
        }
        const double r = reduce(mx, OpAbsMaxNan());
        __syncthreads();
        return r;
    }

    __device__ double inf_norm3(const gdbl* a, int na, const gdbl* b, int nb, const gdbl* c, int nc)
    {
        double mx = 0.0;
        for (int i = tid(); i < na; i += NT) { const double t = fabs(a[i]); if (t > mx || t != t) mx = t; }
        for (int i = tid(); i < nb; i += NT) { const double t = fabs(b[i]); if (t > mx || t != t) mx = t; }
        for (int i = tid(); i < nc; i += NT) { const double t = fabs(c[i]); if (t > mx || t != t) mx = t; }
        return reduce(mx, OpAbsMaxNan());
    }

    // :213-369  (rhs, lhs = Variables sets)
    __device__ __noinline__ bool ks_solve(int rhs, int lhs)
    {
        assume_lds();
        const long long t_begin = wall_clock64();
        const bool ok = ks_solve_impl(rhs, lhs);
        st.prof[T_KS] += wall_clock64() - t_begin;
        return ok;
    }
    __device__ __forceinline__ bool ks_solve_impl(int rhs, int lhs)
    {
        const int n = S.n, p = S.p, m = S.m;
        const gdbl* xbs = at(D_XBS);
        const gdbl*s_l = at(K_SL), *s_u = at(K_SU), *s_bl = at(K_SBL), *s_bu = at(K_SBU);
        const gdbl*zli = at(K_ZLI), *zui = at(K_ZUI), *zbli = at(K_ZBLI), *zbui = at(K_ZBUI);
        const gdbl* z_reg = at(K_ZREG);
        gdbl*rxb = at(K_RXB), *rzb = at(K_RZB), *lz = at(K_LZ);
        const double delta = ks_delta;
        for (int i = tid(); i < m; i += NT) {
            double r = 0.0;
            if (g(S.has_l)[i]) r -= 1.0 / (zli[i] * s_l[i] + delta) * (v(rhs, FZL)[i] - zli[i] * v(rhs, FSL)[i]);
            if (g(S.has_u)[i]) r += 1.0 / (zui[i] * s_u[i] + delta) * (v(rhs, FZU)[i] - zui[i] * v(rhs, FSU)[i]);
            rzb[i] = r * z_reg[i];
        }
        {
            const auto pos_l = g(S.pos_l), pos_u = g(S.pos_u);
            const gdbl*rx_ = v(rhs, FX), *rzbl = v(rhs, FZBL), *rsbl = v(rhs, FSBL), *rzbu = v(rhs, FZBU), *rsbu = v(rhs, FSBU);
            for (int j0 = tid(); j0 < n; j0 += 2 * NT) {
                const int j1 = j0 + NT;
                const bool ok1 = j1 < n;
                const int jj[2] = {j0, ok1 ? j1 : j0};
                int il[2], iu[2];
                double r0[2], xb[2], a_zl[2], a_sl[2], a_zi[2], a_s[2], b_zu[2], b_su[2], b_zi[2], b_s[2];
#pragma unroll
                for (int r = 0; r < 2; ++r) { il[r] = pos_l[jj[r]]; iu[r] = pos_u[jj[r]]; r0[r] = rx_[jj[r]]; xb[r] = xbs[jj[r]]; }
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int a = il[r] > 0 ? il[r] : 0, b = iu[r] > 0 ? iu[r] : 0;
                    a_zl[r] = rzbl[a]; a_sl[r] = rsbl[a]; a_zi[r] = zbli[a]; a_s[r] = s_bl[a];
                    b_zu[r] = rzbu[b]; b_su[r] = rsbu[b]; b_zi[r] = zbui[b]; b_s[r] = s_bu[b];
                }
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    if (r == 1 && !ok1) break;
                    double rr = r0[r];
                    if (il[r] >= 0) rr -= xb[r] * (a_zl[r] - a_zi[r] * a_sl[r]) / (a_s[r] * a_zi[r] + delta);
                    if (iu[r] >= 0) rr += xb[r] * (b_zu[r] - b_zi[r] * b_su[r]) / (b_s[r] * b_zi[r] + delta);
                    rxb[jj[r]] = rr;
                }
            }
        }
        __syncthreads();
        gdbl*lx = v(lhs, FX), *ly = v(lhs, FY);
        const gdbl* ry = v(rhs, FY);
        be_solve(rxb, ry, rzb, lx, ly, lz);

        if (ks_use_refine) {
            gdbl*ex = at(K_EX), *ey = at(K_EY), *ez = at(K_EZ), *rlx = at(K_RLX), *rly = at(K_RLY), *rlz = at(K_RLZ);
            const double rhs_norm = inf_norm3(rxb, n, ry, p, rzb, m);
            double err = refine_error(lx, ly, lz, rxb, ry, rzb, ex, ey, ez);
            if (!isfinite(err)) return false;
            for (int it = 0; it < S.set.iterative_refinement_max_iter; ++it) {
                if (err <= S.set.iterative_refinement_eps_abs + S.set.iterative_refinement_eps_rel * rhs_norm) break;
                const double prev = err;
                be_solve(ex, ey, ez, rlx, rly, rlz);
                map2<2>(n, {rlx, lx}, [&](int j, const double (&x)[2]) { rlx[j] = x[0] + x[1]; });
                map2<2>(p, {rly, ly}, [&](int k, const double (&x)[2]) { rly[k] = x[0] + x[1]; });
                map2<2>(m, {rlz, lz}, [&](int i, const double (&x)[2]) { rlz[i] = x[0] + x[1]; });
                __syncthreads();
                err = refine_error(rlx, rly, rlz, rxb, ry, rzb, ex, ey, ez);
                if (!isfinite(err)) return false;
                const double rate = prev / err;
                const bool stop = rate < S.set.iterative_refinement_min_improvement_rate;
                if (!stop || rate > 1.0) {  // the reference swaps the buffers (:292-300); the refined iterate becomes lhs
                    map2<1>(n, {rlx}, [&](int j, const double (&x)[1]) { lx[j] = x[0]; });
                    map2<1>(p, {rly}, [&](int k, const double (&x)[1]) { ly[k] = x[0]; });
                    map2<1>(m, {rlz}, [&](int i, const double (&x)[1]) { lz[i] = x[0]; });
                    __syncthreads();
                }
                if (stop) break;
            }
        } else {
            double bad = 0.0;
            map2<1>(n, {lx}, [&](int, const double (&x)[1]) { if (!isfinite(x[0])) bad = 1.0; });
            map2<1>(p, {ly}, [&](int, const double (&x)[1]) { if (!isfinite(x[0])) bad = 1.0; });
            map2<1>(m, {lz}, [&](int, const double (&x)[1]) { if (!isfinite(x[0])) bad = 1.0; });
            if (reduce(bad, OpMax()) > 0.0) return false;
        }

        // :310-345 dual recovery
        for (int i = tid(); i < m; i += NT) {
            const bool hl = g(S.has_l)[i] != 0, hu = g(S.has_u)[i] != 0;
            double zl = 0.0, zu = 0.0, sl = 0.0, su = 0.0;
            if (hl && hu) {
                const double rz_l_bar = v(rhs, FZL)[i] - zli[i] * v(rhs, FSL)[i];
                const double W_l_inv = 1.0 / (zli[i] * s_l[i] + delta);
                const double rz_u_bar = v(rhs, FZU)[i] - zui[i] * v(rhs, FSU)[i];
                const double W_u_inv = 1.0 / (zui[i] * s_u[i] + delta);
                const double r_sum = W_l_inv * W_u_inv * (rz_l_bar + rz_u_bar);
                zl = -z_reg[i] * (r_sum + W_l_inv * lz[i]);
                zu = -z_reg[i] * (r_sum - W_u_inv * lz[i]);
                sl = zli[i] * (v(rhs, FSL)[i] - s_l[i] * zl);
                su = zui[i] * (v(rhs, FSU)[i] - s_u[i] * zu);
            } else if (hl) {
                zl = -lz[i];
                sl = zli[i] * (v(rhs, FSL)[i] - s_l[i] * zl);
            } else if (hu) {
                zu = lz[i];
                su = zui[i] * (v(rhs, FSU)[i] - s_u[i] * zu);
            }
            v(lhs, FZL)[i] = zl; v(lhs, FZU)[i] = zu; v(lhs, FSL)[i] = sl; v(lhs, FSU)[i] = su;
        }
        // :347-366 box dual recovery
        {
            // box dual recovery, two trips at once: the index and the element-wise operands first, then the two gathered ones, then the stores
            auto box = [&](int cnt, const __attribute__((address_space(1))) int* idxs, const gdbl* rz, const gdbl* rs, const gdbl* zi, const gdbl* sb, gdbl* oz, gdbl* os, const bool upper) {
                for (int i0 = tid(); i0 < cnt; i0 += 2 * NT) {
                    const int i1 = i0 + NT;
                    const bool ok1 = i1 < cnt;
                    const int ii[2] = {i0, ok1 ? i1 : i0};
                    int idx[2];
                    double a_rz[2], a_rs[2], a_zi[2], a_sb[2], xb[2], lxv[2];
#pragma unroll
                    for (int r = 0; r < 2; ++r) { idx[r] = idxs[ii[r]]; a_rz[r] = rz[ii[r]]; a_rs[r] = rs[ii[r]]; a_zi[r] = zi[ii[r]]; a_sb[r] = sb[ii[r]]; }
#pragma unroll
                    for (int r = 0; r < 2; ++r) { xb[r] = xbs[idx[r]]; lxv[r] = lx[idx[r]]; }
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        if (r == 1 && !ok1) break;
                        const double z = upper ? (xb[r] * lxv[r] - a_rz[r] + a_zi[r] * a_rs[r]) / (a_sb[r] * a_zi[r] + delta)
                                               : (-xb[r] * lxv[r] - a_rz[r] + a_zi[r] * a_rs[r]) / (a_sb[r] * a_zi[r] + delta);
                        oz[ii[r]] = z;
                        os[ii[r]] = a_zi[r] * (a_rs[r] - a_sb[r] * z);
                    }
                }
            };
            box(S.n_x_l, g(S.x_l_idx), v(rhs, FZBL), v(rhs, FSBL), zbli, s_bl, v(lhs, FZBL), v(lhs, FSBL), false);
            box(S.n_x_u, g(S.x_u_idx), v(rhs, FZBU), v(rhs, FSBU), zbui, s_bu, v(lhs, FZBU), v(lhs, FSBU), true);
        }
        __syncthreads();
        info.n_solve++;
        return true;
    }

    // ---- solver.hpp helpers --------------------------------------------------------------------------------------
    __device__ __noinline__ double dot2(const gdbl* a, const gdbl* b, int cnt)
    {
        assume_lds();
        if (cnt <= 0) return 0.0;  // (an empty block: the sum of the lanes' zeros)
        double s = 0.0;
        // two trips of the lane-strided loop at once: the second element comes from a clamped (valid) address and is left out by its mask, so that all four
        // loads are in flight together -- a wave is waiting for memory three quarters of its life (profiles/r03_pmc_batch_c4.txt), and with a run-time trip
        // count that differs between the lanes the compiler's own unrolling runs a remainder trip first: two waits again.  Same sum, same order.
        for (int i0 = tid(); i0 < cnt; i0 += 2 * NT) {
            const int i1 = i0 + NT;
            const bool ok1 = i1 < cnt;
            const int j1 = ok1 ? i1 : i0;
            const double a0 = a[i0], b0 = b[i0], a1 = a[j1], b1 = b[j1];
            s += a0 * b0;
            if (ok1) s += a1 * b1;
        }
        return reduce(s, OpSum());
    }
    // :884-891
    __device__ __noinline__ double calculate_mu()
    {
        assume_lds();
        const double s = dot2(v(V_R, FSL), v(V_R, FZL), S.m) + dot2(v(V_R, FSU), v(V_R, FZU), S.m) + dot2(v(V_R, FSBL), v(V_R, FZBL), S.n_x_l) +
                         dot2(v(V_R, FSBU), v(V_R, FZBU), S.n_x_u);
        return s / (double)(S.n_h_l + S.n_h_u + S.n_x_l + S.n_x_u);
    }
    // :893-958
    __device__ __noinline__ void calculate_step(double& alpha_s, double& alpha_z)
    {
        assume_lds();
        double as = 1.0, az = 1.0;
        auto upd = [](double& a, double r, double st) { if (st < 0) { const double c = -r / st; if (c < a) a = c; } };
        for (int i = tid(); i < S.m; i += NT) {
            upd(as, v(V_R, FSL)[i], v(V_ST, FSL)[i]); upd(as, v(V_R, FSU)[i], v(V_ST, FSU)[i]);
            upd(az, v(V_R, FZL)[i], v(V_ST, FZL)[i]); upd(az, v(V_R, FZU)[i], v(V_ST, FZU)[i]);
        }
        for (int i = tid(); i < S.n_x_l; i += NT) { upd(as, v(V_R, FSBL)[i], v(V_ST, FSBL)[i]); upd(az, v(V_R, FZBL)[i], v(V_ST, FZBL)[i]); }
        for (int i = tid(); i < S.n_x_u; i += NT) { upd(as, v(V_R, FSBU)[i], v(V_ST, FSBU)[i]); upd(az, v(V_R, FZBU)[i], v(V_ST, FZBU)[i]); }
        alpha_s = reduce(as, OpMin());
        alpha_z = reduce(az, OpMin());
    }
    __device__ __noinline__ double min_coeff(const gdbl* a, int cnt)
    {
        assume_lds();
        if (cnt <= 0) return DBL_MAX;
        double mn = DBL_MAX;
        for (int i0 = tid(); i0 < cnt; i0 += 2 * NT) {
            const int i1 = i0 + NT;
            const double a0 = a[i0], a1 = a[i1 < cnt ? i1 : i0];  // (the clamped duplicate cannot change a minimum)
            if (a0 < mn) mn = a0;
            if (a1 < mn) mn = a1;
        }
        return reduce(mn, OpMin());
    }
    __device__ __noinline__ double inf_scaled(const gdbl* a, const gdbl* sc, double c, int cnt)
    {
        assume_lds();
        if (cnt <= 0) return 0.0;
        double mx = 0.0;
        for (int i0 = tid(); i0 < cnt; i0 += 2 * NT) {
            const int i1 = i0 + NT;
            const int j1 = i1 < cnt ? i1 : i0;  // (a clamped duplicate cannot change a maximum)
            const double a0 = a[i0], s0 = sc[i0], a1 = a[j1], s1 = sc[j1];
            const double t0 = fabs(a0 * c * s0), t1 = fabs(a1 * c * s1);
            if (t0 > mx || t0 != t0) mx = t0;
            if (t1 > mx || t1 != t1) mx = t1;
        }
        return reduce(mx, OpAbsMaxNan());
    }
    // :1130-1164
    __device__ __noinline__ double primal_res_of(int set)
    {
        assume_lds();
        const int n = S.n, p = S.p, m = S.m;
        const gdbl* dinv = at(D_DLI);
        const gdbl* dbi = at(D_DBI);
        double inf = inf_scaled(v(set, FY), dinv + n, 1.0, p);
        inf = fmax_std(inf, inf_scaled(v(set, FZL), dinv + n + p, 1.0, m));
        inf = fmax_std(inf, inf_scaled(v(set, FZU), dinv + n + p, 1.0, m));
        double mx = inf;
        auto boxmax = [&](int cnt, const __attribute__((address_space(1))) int* idxs, const gdbl* z) {
            for (int i0 = tid(); i0 < cnt; i0 += 2 * NT) {
                const int i1 = i0 + NT;
                const int j1 = i1 < cnt ? i1 : i0;  // (a clamped duplicate cannot change a maximum)
                const int x0 = idxs[i0], x1 = idxs[j1];
                const double z0 = z[i0], z1 = z[j1];
                const double t0 = z0 * dbi[x0], t1 = z1 * dbi[x1];
                if (mx < t0) mx = t0;
                if (mx < t1) mx = t1;
            }
        };
        boxmax(S.n_x_l, g(S.x_l_idx), v(set, FZBL));
        boxmax(S.n_x_u, g(S.x_u_idx), v(set, FZBU));
        return reduce(mx, OpMax());
    }
    static __device__ __forceinline__ double fmax_std(double a, double b) { return a < b ? b : a; }
    // :1184-1196
    __device__ double dual_res_of(const gdbl* x) { return inf_scaled(x, at(D_DLI), rz_c_inv, S.n); }
    // :1166-1182
    __device__ double primal_prox_inf()
    {
        const int n = S.n, p = S.p, m = S.m;
        const gdbl* dl = at(D_DL);
        const gdbl* db = at(D_DB);
        const double ci = rz_c_inv;
        double mx = 0.0;
        map2<3>(p, {v(V_PX, FY), v(V_R, FY), dl + n}, [&](int, const double (&x)[3]) { const double t = fabs((x[0] - x[1]) * ci * x[2]); if (mx < t) mx = t; });
        for (int i = tid(); i < m; i += NT) {
            double t = fabs((v(V_PX, FZL)[i] - v(V_R, FZL)[i]) * ci * dl[n + p + i]); if (mx < t) mx = t;
            t = fabs((v(V_PX, FZU)[i] - v(V_R, FZU)[i]) * ci * dl[n + p + i]); if (mx < t) mx = t;
        }
        auto boxprox = [&](int cnt, const __attribute__((address_space(1))) int* idxs, const gdbl* a, const gdbl* b) {
            for (int i0 = tid(); i0 < cnt; i0 += 2 * NT) {
                const int i1 = i0 + NT;
                const int j1 = i1 < cnt ? i1 : i0;  // (a clamped duplicate cannot change a maximum)
                const int x0 = idxs[i0], x1 = idxs[j1];
                const double a0 = a[i0], b0 = b[i0], a1 = a[j1], b1 = b[j1];
                const double t0 = (a0 - b0) * ci * db[x0], t1 = (a1 - b1) * ci * db[x1];
                if (mx < t0) mx = t0;
                if (mx < t1) mx = t1;
            }
        };
        boxprox(S.n_x_l, g(S.x_l_idx), v(V_PX, FZBL), v(V_R, FZBL));
        boxprox(S.n_x_u, g(S.x_u_idx), v(V_PX, FZBU), v(V_R, FZBU));
        return reduce(mx, OpMax());
    }
    // :1198-1203
    __device__ double dual_prox_inf()
    {
        const gdbl* dl = at(D_DL);
        double mx = 0.0;
        map2<3>(S.n, {v(V_R, FX), v(V_PX, FX), dl}, [&](int, const double (&x)[3]) { const double t = fabs((x[0] - x[1]) * x[2]); if (mx < t) mx = t; });
        return reduce(mx, OpMax());
    }

    // :960-1105
    __device__ __forceinline__ void update_residuals_nr()
    {
        assume_lds();
        const long long t_begin = wall_clock64();
        update_residuals_nr_impl();
        st.prof[T_RES] += wall_clock64() - t_begin;
    }
    __device__ __forceinline__ void update_residuals_nr_impl()
    {
        const int n = S.n, p = S.p, m = S.m;
        const double ci = rz_c_inv;
        const gdbl* dinv = at(D_DLI);
        const gdbl* dbi = at(D_DBI);
        const gdbl* xbs = at(D_XBS);
        gdbl* work_x = v(V_ST, FX);
        gdbl* work_z = v(V_ST, FZL);
        gdbl* nrx = v(V_NR, FX);
        const gdbl*rx = v(V_R, FX), *c = at(D_C), *bb = at(D_B), *hl = at(D_HL), *hu = at(D_HU), *xl = at(D_XL), *xu = at(D_XU);

        eval_A(-1.0, 1.0, rx, v(V_R, FY), v(V_NR, FY), work_x);
        for (int i = tid(); i < m; i += NT) work_z[i] = v(V_R, FZU)[i] - v(V_R, FZL)[i];
        __syncthreads();
        gdbl* work_x_2 = nrx;
        eval_G(1.0, 1.0, rx, work_z, v(V_NR, FZL), work_x_2);
        for (int i = tid(); i < m; i += NT) v(V_NR, FZU)[i] = -v(V_NR, FZL)[i];
        map2<2>(n, {work_x, work_x_2}, [&](int j, const double (&x)[2]) { work_x[j] = x[0] + x[1]; });
        __syncthreads();

        eval_P_x(-1.0, rx, nrx);
        double dual_rel_norm = inf_scaled(nrx, dinv, ci, n);

        double tmp = -dot2(rx, nrx, n);
        info.primal_obj = 0.5 * tmp;
        info.dual_obj = -0.5 * tmp;
        double dg_rel = ci * fabs(tmp);
        tmp = dot2(c, rx, n); info.primal_obj += tmp; dg_rel = fmax_std(dg_rel, ci * fabs(tmp));
        tmp = dot2(bb, v(V_R, FY), p); info.dual_obj -= tmp; dg_rel = fmax_std(dg_rel, ci * fabs(tmp));
        tmp = -dot2(hl, v(V_R, FZL), m); info.dual_obj -= tmp; dg_rel = fmax_std(dg_rel, ci * fabs(tmp));
        tmp = dot2(hu, v(V_R, FZU), m); info.dual_obj -= tmp; dg_rel = fmax_std(dg_rel, ci * fabs(tmp));
        tmp = -dot2(xl, v(V_R, FZBL), S.n_x_l); info.dual_obj -= tmp; dg_rel = fmax_std(dg_rel, ci * fabs(tmp));
        tmp = dot2(xu, v(V_R, FZBU), S.n_x_u); info.dual_obj -= tmp; dg_rel = fmax_std(dg_rel, ci * fabs(tmp));

        info.duality_gap = fabs(info.primal_obj - info.dual_obj);
        info.primal_obj *= ci; info.dual_obj *= ci; info.duality_gap *= ci;
        info.duality_gap_rel = info.duality_gap / fmax_std(1.0, dg_rel);

        __syncthreads();
        {
            const auto pos_l = g(S.pos_l), pos_u = g(S.pos_u);
            const gdbl*zbl = v(V_R, FZBL), *zbu = v(V_R, FZBU);
            for (int j0 = tid(); j0 < n; j0 += 2 * NT) {
                const int j1 = j0 + NT;
                const bool ok1 = j1 < n;
                const int jj[2] = {j0, ok1 ? j1 : j0};
                int il[2], iu[2];
                double a_n[2], a_c[2], a_w[2], xb[2], gl[2], gu[2];
#pragma unroll
                for (int r = 0; r < 2; ++r) { il[r] = pos_l[jj[r]]; iu[r] = pos_u[jj[r]]; a_n[r] = nrx[jj[r]]; a_c[r] = c[jj[r]]; a_w[r] = work_x[jj[r]]; xb[r] = xbs[jj[r]]; }
#pragma unroll
                for (int r = 0; r < 2; ++r) { gl[r] = zbl[il[r] > 0 ? il[r] : 0]; gu[r] = zbu[iu[r] > 0 ? iu[r] : 0]; }
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    if (r == 1 && !ok1) break;
                    nrx[jj[r]] = a_n[r] - a_c[r];
                    double wx = a_w[r];
                    if (il[r] >= 0) wx -= xb[r] * gl[r];
                    if (iu[r] >= 0) wx += xb[r] * gu[r];
                    work_x[jj[r]] = wx;
                }
            }
        }
        __syncthreads();
        dual_rel_norm = fmax_std(dual_rel_norm, inf_scaled(c, dinv, ci, n));
        dual_rel_norm = fmax_std(dual_rel_norm, inf_scaled(work_x, dinv, ci, n));
        map2<2>(n, {nrx, work_x}, [&](int j, const double (&x)[2]) { nrx[j] = x[0] - x[1]; });

        double primal_rel_norm = inf_scaled(v(V_NR, FY), dinv + n, 1.0, p);
        __syncthreads();
        { gdbl* o = v(V_NR, FY); map2<2>(p, {v(V_NR, FY), bb}, [&](int i, const double (&x)[2]) { o[i] = x[0] + x[1]; }); }
        primal_rel_norm = fmax_std(primal_rel_norm, inf_scaled(bb, dinv + n, 1.0, p));

        const gdbl* dz = dinv + n + p;
        double mx = primal_rel_norm;
        for (int i = tid(); i < m; i += NT) {
            if (g(S.has_l)[i]) {
                gdbl* z = v(V_NR, FZL);
                mx = fmax_std(mx, z[i] * dz[i]);  // signed, like the reference (:1047)
                z[i] += -hl[i] - v(V_R, FSL)[i];
                mx = fmax_std(mx, hl[i] * dz[i]);
                mx = fmax_std(mx, v(V_R, FSL)[i] * dz[i]);
            } else {
                v(V_NR, FZL)[i] = 0.0;
            }
            if (g(S.has_u)[i]) {
                gdbl* z = v(V_NR, FZU);
                mx = fmax_std(mx, z[i] * dz[i]);
                z[i] += hu[i] - v(V_R, FSU)[i];
                mx = fmax_std(mx, hu[i] * dz[i]);
                mx = fmax_std(mx, v(V_R, FSU)[i] * dz[i]);
            } else {
                v(V_NR, FZU)[i] = 0.0;
            }
        }
        {
            auto boxres = [&](int cnt, const __attribute__((address_space(1))) int* idxs, const gdbl* xb_, const gdbl* sb, gdbl* o, const bool upper) {
                for (int i0 = tid(); i0 < cnt; i0 += 2 * NT) {
                    const int i1 = i0 + NT;
                    const bool ok1 = i1 < cnt;
                    const int ii[2] = {i0, ok1 ? i1 : i0};
                    int idx[2];
                    double a_b[2], a_s[2], gx[2], gr[2], gd[2];
#pragma unroll
                    for (int r = 0; r < 2; ++r) { idx[r] = idxs[ii[r]]; a_b[r] = xb_[ii[r]]; a_s[r] = sb[ii[r]]; }
#pragma unroll
                    for (int r = 0; r < 2; ++r) { gx[r] = xbs[idx[r]]; gr[r] = rx[idx[r]]; gd[r] = dbi[idx[r]]; }
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        if (r == 1 && !ok1) break;
                        const double t = upper ? -gx[r] * gr[r] : gx[r] * gr[r];
                        mx = fmax_std(mx, t * gd[r]);
                        mx = fmax_std(mx, a_b[r] * gd[r]);
                        mx = fmax_std(mx, a_s[r] * gd[r]);
                        o[ii[r]] = upper ? t + (a_b[r] - a_s[r]) : t + (-a_b[r] - a_s[r]);
                    }
                }
            };
            boxres(S.n_x_l, g(S.x_l_idx), xl, v(V_R, FSBL), v(V_NR, FZBL), false);
            boxres(S.n_x_u, g(S.x_u_idx), xu, v(V_R, FSBU), v(V_NR, FZBU), true);
        }
        primal_rel_norm = reduce(mx, OpMax());
        __syncthreads();

        info.prev_primal_res = info.primal_res;
        info.prev_dual_res = info.dual_res;
        info.primal_res = primal_res_of(V_NR);
        info.primal_res_rel = info.primal_res / fmax_std(1.0, primal_rel_norm);
        info.dual_res = dual_res_of(nrx);
        info.dual_res_rel = info.dual_res / fmax_std(1.0, dual_rel_norm);
    }

    // :1107-1128
    __device__ __forceinline__ void update_residuals_r()
    {
        assume_lds();
        const int n = S.n, p = S.p, m = S.m;
        const double rho = info.rho, delta = info.delta;
        { gdbl* o = v(V_RS, FX); map2<3>(n, {v(V_NR, FX), v(V_R, FX), v(V_PX, FX)}, [&](int j, const double (&x)[3]) { o[j] = x[0] - rho * (x[1] - x[2]); }); }
        auto prox_dual = [&](int f, int cnt) { gdbl* o = v(V_RS, f); map2<3>(cnt, {v(V_NR, f), v(V_PX, f), v(V_R, f)}, [&](int i, const double (&x)[3]) { o[i] = x[0] - delta * (x[1] - x[2]); }); };
        prox_dual(FY, p);
        prox_dual(FZL, m);
        prox_dual(FZU, m);
        prox_dual(FZBL, S.n_x_l);
        prox_dual(FZBU, S.n_x_u);
        __syncthreads();
        const double primal_rel_scaling = info.primal_res_rel > 0 ? info.primal_res / info.primal_res_rel : 1.0;
        const double dual_rel_scaling = info.dual_res_rel > 0 ? info.dual_res / info.dual_res_rel : 1.0;
        info.primal_res_reg = primal_res_of(V_RS);
        info.primal_res_reg_rel = info.primal_res_reg / primal_rel_scaling;
        info.dual_res_reg = dual_res_of(v(V_RS, FX));
        info.dual_res_reg_rel = info.dual_res_reg / dual_rel_scaling;
        info.primal_prox_inf = primal_prox_inf() * info.delta;
        info.dual_prox_inf = dual_prox_inf() * info.rho;
    }

    // the factor-with-retries loops of :446-465 / :688-708; returns false on PIQP_NUMERICS
    __device__ bool factor_with_retries(bool in_loop, bool& regularization_changed)
    {
        while (!ks_update_scalings_and_factor(refine_enabled != 0, info.rho, info.delta)) {
            if (!refine_enabled) { refine_enabled = 1; continue; }
            if (info.factor_retires < S.set.max_factor_retires) {
                info.delta *= 100; info.rho *= 100; info.factor_retires++;
                info.reg_limit = fmin(10 * info.reg_limit, S.set.eps_abs);
                if (in_loop) regularization_changed = true;
                continue;
            }
            return false;
        }
        info.factor_retires = 0;
        return true;
    }

    __device__ void copy_prox_duals()
    {
        auto cp = [&](int f, int cnt) { gdbl* o = v(V_PX, f); map2<1>(cnt, {v(V_R, f)}, [&](int i, const double (&x)[1]) { o[i] = x[0]; }); };
        cp(FY, S.p); cp(FZL, S.m); cp(FZU, S.m); cp(FZBL, S.n_x_l); cp(FZBU, S.n_x_u);
    }

    // solver.hpp:379-882
    __device__ int solve_impl()
    {
        const pq_settings& set = S.set;
        const int n = S.n, p = S.p, m = S.m;
        info.kkt_factor_time = 0; info.kkt_solve_time = 0;
        info.n_factor = info.n_solve = info.n_backend_solve = 0;
        info.status = PQ_UNSOLVED;
        info.iter = 0;
        info.reg_limit = set.reg_lower_limit;
        info.factor_retires = 0; info.no_primal_update = 0; info.no_dual_update = 0;
        info.mu = 0; info.primal_step = 0; info.dual_step = 0; info.sigma = 0;
        info.rho = set.rho_init; info.delta = set.delta_init;
        info.primal_res = info.dual_res = info.primal_res_rel = info.dual_res_rel = 0;

        // :416-437
        for (int i = tid(); i < m; i += NT) {
            const double l = g(S.has_l)[i] ? 1.0 : 0.0, u = g(S.has_u)[i] ? 1.0 : 0.0;
            v(V_R, FSL)[i] = l; v(V_R, FZL)[i] = l; v(V_R, FSU)[i] = u; v(V_R, FZU)[i] = u;
        }
        for (int i = tid(); i < n; i += NT) {
            const double l = i < S.n_x_l ? 1.0 : 0.0, u = i < S.n_x_u ? 1.0 : 0.0;
            v(V_R, FSBL)[i] = l; v(V_R, FZBL)[i] = l; v(V_R, FSBU)[i] = u; v(V_R, FZBU)[i] = u;
        }
        __syncthreads();
        refine_enabled = set.iterative_refinement_always_enabled != 0 ? 1 : 0;
        bool dummy = false;
        if (!factor_with_retries(false, dummy)) { info.status = PQ_NUMERICS; return info.status; }

        for (int i = tid(); i < n; i += NT) { v(V_RS, FX)[i] = -at(D_C)[i]; v(V_RS, FSBL)[i] = 0.0; v(V_RS, FSBU)[i] = 0.0; }
        for (int i = tid(); i < p; i += NT) v(V_RS, FY)[i] = at(D_B)[i];
        for (int i = tid(); i < m; i += NT) { v(V_RS, FZL)[i] = -at(D_HL)[i]; v(V_RS, FZU)[i] = at(D_HU)[i]; v(V_RS, FSL)[i] = 0.0; v(V_RS, FSU)[i] = 0.0; }
        // x_l / x_u are stored compressed to the first n_x_l / n_x_u entries (sparse/data.hpp)
        for (int i = tid(); i < n; i += NT) { v(V_RS, FZBL)[i] = -at(D_XL)[i]; v(V_RS, FZBU)[i] = at(D_XU)[i]; }
        __syncthreads();
        ks_solve(V_RS, V_R);

        if (m + S.n_x_l + S.n_x_u > 0) {
            // :504-570
            double delta_s = 0.0, delta_z = 0.0;
            if (m > 0) { delta_s = fmax_std(delta_s, -min_coeff(v(V_R, FSL), m)); delta_s = fmax_std(delta_s, -min_coeff(v(V_R, FSU), m)); }
            if (S.n_x_l > 0) delta_s = fmax_std(delta_s, -min_coeff(v(V_R, FSBL), S.n_x_l));
            if (S.n_x_u > 0) delta_s = fmax_std(delta_s, -min_coeff(v(V_R, FSBU), S.n_x_u));
            if (m > 0) { delta_z = fmax_std(delta_z, -min_coeff(v(V_R, FZL), m)); delta_z = fmax_std(delta_z, -min_coeff(v(V_R, FZU), m)); }
            if (S.n_x_l > 0) delta_z = fmax_std(delta_z, -min_coeff(v(V_R, FZBL), S.n_x_l));
            if (S.n_x_u > 0) delta_z = fmax_std(delta_z, -min_coeff(v(V_R, FZBU), S.n_x_u));
            __syncthreads();
            for (int i = tid(); i < m; i += NT) {
                if (g(S.has_l)[i]) { v(V_R, FSL)[i] += delta_s; v(V_R, FZL)[i] += delta_z; }
                if (g(S.has_u)[i]) { v(V_R, FSU)[i] += delta_s; v(V_R, FZU)[i] += delta_z; }
            }
            for (int i = tid(); i < S.n_x_l; i += NT) { v(V_R, FSBL)[i] += delta_s; v(V_R, FZBL)[i] += delta_z; }
            for (int i = tid(); i < S.n_x_u; i += NT) { v(V_R, FSBU)[i] += delta_s; v(V_R, FZBU)[i] += delta_z; }
            __syncthreads();
            info.mu = fmax_std(calculate_mu(), 1e-10);
            const double mu = info.mu;
            auto centre = [&](gdbl& z, gdbl& s) { const double cc = z - delta_z; z = (cc + sqrt(cc * cc + 4 * mu)) / 2; s = z - cc; };
            __syncthreads();
            for (int i = tid(); i < m; i += NT) {
                if (g(S.has_l)[i]) centre(v(V_R, FZL)[i], v(V_R, FSL)[i]);
                if (g(S.has_u)[i]) centre(v(V_R, FZU)[i], v(V_R, FSU)[i]);
            }
            for (int i = tid(); i < S.n_x_l; i += NT) centre(v(V_R, FZBL)[i], v(V_R, FSBL)[i]);
            for (int i = tid(); i < S.n_x_u; i += NT) centre(v(V_R, FZBU)[i], v(V_R, FSBU)[i]);
            __syncthreads();
            info.mu = calculate_mu();
        }

        { gdbl* o = v(V_PX, FX); map2<1>(n, {v(V_R, FX)}, [&](int i, const double (&x)[1]) { o[i] = x[0]; }); }
        copy_prox_duals();
        __syncthreads();

        while (info.iter < set.max_iter) {
            if (info.iter == 0) {
                update_residuals_nr();
                info.prev_primal_res = info.primal_res;
                info.prev_dual_res = info.dual_res;
            }
            if ((info.primal_res < set.eps_abs || info.primal_res_rel < set.eps_rel) && (info.dual_res < set.eps_abs || info.dual_res_rel < set.eps_rel) &&
                (!set.check_duality_gap || info.duality_gap < set.eps_duality_gap_abs || info.duality_gap_rel < set.eps_duality_gap_rel)) {
                info.status = PQ_SOLVED;
                return info.status;
            }
            update_residuals_r();
            if (info.no_dual_update > min(5, set.reg_finetune_dual_update_threshold) && info.primal_prox_inf > set.infeasibility_threshold &&
                (info.primal_res_reg < set.eps_abs || info.primal_res_reg_rel < set.eps_rel)) {
                info.status = PQ_PRIMAL_INFEASIBLE;
                return info.status;
            }
            if (info.no_primal_update > min(5, set.reg_finetune_primal_update_threshold) && info.dual_prox_inf > set.infeasibility_threshold &&
                (info.dual_res_reg < set.eps_abs || info.dual_res_reg_rel < set.eps_rel)) {
                info.status = PQ_DUAL_INFEASIBLE;
                return info.status;
            }
            info.iter++;

            // :634-666 keep z off the boundary
            const double epsilon = DBL_EPSILON;
            double shifted = 0.0;
            for (int i = tid(); i < m; i += NT) {
                if (g(S.has_l)[i] && v(V_R, FZL)[i] < epsilon) { v(V_R, FZL)[i] += epsilon; shifted = 1.0; }
                if (g(S.has_u)[i] && v(V_R, FZU)[i] < epsilon) { v(V_R, FZU)[i] += epsilon; shifted = 1.0; }
            }
            bool boundary_shifted = reduce(shifted, OpMax()) > 0.0;
            if (S.n_x_l > 0 && min_coeff(v(V_R, FZBL), S.n_x_l) < epsilon) {
                __syncthreads();
                for (int i = tid(); i < S.n_x_l; i += NT) v(V_R, FZBL)[i] += epsilon;
                boundary_shifted = true;
            }
            if (S.n_x_u > 0 && min_coeff(v(V_R, FZBU), S.n_x_u) < epsilon) {
                __syncthreads();
                for (int i = tid(); i < S.n_x_u; i += NT) v(V_R, FZBU)[i] += epsilon;
                boundary_shifted = true;
            }
            __syncthreads();
            if (boundary_shifted) info.mu = calculate_mu();

            // :668-681
            if ((info.no_primal_update > set.reg_finetune_primal_update_threshold && info.rho == info.reg_limit && info.reg_limit != set.reg_finetune_lower_limit) ||
                (info.no_dual_update > set.reg_finetune_dual_update_threshold && info.delta == info.reg_limit && info.reg_limit != set.reg_finetune_lower_limit)) {
                if (info.dual_prox_inf < set.infeasibility_threshold && info.primal_prox_inf < set.infeasibility_threshold) {
                    info.reg_limit = set.reg_finetune_lower_limit;
                    info.no_primal_update = 0;
                    info.no_dual_update = 0;
                }
            }

            bool regularization_changed = false;
            if (!factor_with_retries(true, regularization_changed)) { info.status = PQ_NUMERICS; return info.status; }
            if (regularization_changed) update_residuals_r();

            if (m + S.n_x_l + S.n_x_u > 0) {
                // predictor
                auto pred = [&](int fs, int fz, int cnt) { gdbl* o = v(V_RS, fs); map2<2>(cnt, {v(V_R, fs), v(V_R, fz)}, [&](int i, const double (&x)[2]) { o[i] = -x[0] * x[1]; }); };
                pred(FSL, FZL, m); pred(FSU, FZU, m); pred(FSBL, FZBL, S.n_x_l); pred(FSBU, FZBU, S.n_x_u);
                __syncthreads();
                ks_solve(V_RS, V_ST);

                double alpha_s, alpha_z;
                calculate_step(alpha_s, alpha_z);
                alpha_s *= set.tau; alpha_z *= set.tau;

                auto sdot = [&](int fs, int fz, int cnt) {
                    if (cnt <= 0) return 0.0;
                    double acc = 0.0;
                    map2<4>(cnt, {v(V_R, fs), v(V_ST, fs), v(V_R, fz), v(V_ST, fz)}, [&](int, const double (&x)[4]) { acc += (x[0] + alpha_s * x[1]) * (x[2] + alpha_z * x[3]); });
                    return reduce(acc, OpSum());
                };
                double sigma = sdot(FSL, FZL, m);
                sigma += sdot(FSU, FZU, m);
                sigma += sdot(FSBL, FZBL, S.n_x_l);
                sigma += sdot(FSBU, FZBU, S.n_x_u);
                sigma /= (info.mu * (double)(S.n_h_l + S.n_h_u + S.n_x_l + S.n_x_u));
                sigma = fmax_std(0.0, sigma < 1.0 ? sigma : 1.0);
                info.sigma = sigma * sigma * sigma;

                // corrector
                const double smu = info.sigma * info.mu;
                __syncthreads();
                auto corr = [&](int fs, int fz, int cnt) { gdbl* o = v(V_RS, fs); map2<3>(cnt, {v(V_RS, fs), v(V_ST, fs), v(V_ST, fz)}, [&](int i, const double (&x)[3]) { o[i] = x[0] + (-x[1] * x[2] + smu); }); };
                corr(FSL, FZL, m); corr(FSU, FZU, m); corr(FSBL, FZBL, S.n_x_l); corr(FSBU, FZBU, S.n_x_u);
                __syncthreads();
                ks_solve(V_RS, V_ST);

                calculate_step(alpha_s, alpha_z);
                info.primal_step = alpha_s * set.tau;
                info.dual_step = alpha_z * set.tau;
                const double ps = info.primal_step, ds = info.dual_step;
                __syncthreads();
                auto step = [&](int f, double a, int cnt) { gdbl* o = v(V_R, f); map2<2>(cnt, {v(V_R, f), v(V_ST, f)}, [&](int i, const double (&x)[2]) { o[i] = x[0] + a * x[1]; }); };
                step(FX, ps, n); step(FY, ds, p);
                step(FZL, ds, m); step(FZU, ds, m); step(FSL, ps, m); step(FSU, ps, m);
                step(FZBL, ds, S.n_x_l); step(FSBL, ps, S.n_x_l); step(FZBU, ds, S.n_x_u); step(FSBU, ps, S.n_x_u);
                __syncthreads();

                const double mu_prev = info.mu;
                info.mu = calculate_mu();
                const double mu_rate = fmax_std(0.0, (mu_prev - info.mu) / mu_prev);

                update_residuals_nr();

                if (info.dual_res < 0.95 * info.prev_dual_res || (info.dual_res < set.eps_abs || info.dual_res_rel < set.eps_rel) ||
                    (info.rho == set.reg_finetune_lower_limit && info.dual_prox_inf < set.infeasibility_threshold)) {
                    { gdbl* o = v(V_PX, FX); map2<1>(n, {v(V_R, FX)}, [&](int i, const double (&x)[1]) { o[i] = x[0]; }); }
                    info.rho = fmax_std(info.reg_limit, (1.0 - mu_rate) * info.rho);
                } else {
                    info.no_primal_update++;
                    if (info.iter < 5 || info.dual_prox_inf < set.infeasibility_threshold) info.rho = fmax_std(info.reg_limit, (1.0 - 0.666 * mu_rate) * info.rho);
                }
                if (info.primal_res < 0.95 * info.prev_primal_res || (info.primal_res < set.eps_abs || info.primal_res_rel < set.eps_rel) ||
                    (info.delta == set.reg_finetune_lower_limit && info.primal_prox_inf < set.infeasibility_threshold)) {
                    copy_prox_duals();
                    info.delta = fmax_std(info.reg_limit, (1.0 - mu_rate) * info.delta);
                } else {
                    info.no_dual_update++;
                    if (info.iter < 5 || info.primal_prox_inf < set.infeasibility_threshold) info.delta = fmax_std(info.reg_limit, (1.0 - 0.666 * mu_rate) * info.delta);
                }
                __syncthreads();
            } else {
                // :831-877 no inequalities: one solve, full step
                ks_solve(V_RS, V_ST);
                info.primal_step = 1.0; info.dual_step = 1.0;
                for (int i = tid(); i < n; i += NT) v(V_R, FX)[i] += v(V_ST, FX)[i];
                for (int i = tid(); i < p; i += NT) v(V_R, FY)[i] += v(V_ST, FY)[i];
                __syncthreads();
                update_residuals_nr();
                if (info.dual_res < 0.95 * info.prev_dual_res || (info.dual_res < set.eps_abs || info.dual_res_rel < set.eps_rel)) {
                    { gdbl* o = v(V_PX, FX); map2<1>(n, {v(V_R, FX)}, [&](int i, const double (&x)[1]) { o[i] = x[0]; }); }
                    info.rho = fmax_std(info.reg_limit, 0.1 * info.rho);
                } else {
                    info.no_primal_update++;
                    if (info.iter < 5 || info.dual_prox_inf < set.infeasibility_threshold) info.rho = fmax_std(info.reg_limit, 0.5 * info.rho);
                }
                if (info.primal_res < 0.95 * info.prev_primal_res || (info.primal_res < set.eps_abs || info.primal_res_rel < set.eps_rel)) {
                    for (int i = tid(); i < p; i += NT) v(V_PX, FY)[i] = v(V_R, FY)[i];
                    info.delta = fmax_std(info.reg_limit, 0.1 * info.delta);
                } else {
                    info.no_dual_update++;
                    if (info.iter < 5 || info.primal_prox_inf < set.infeasibility_threshold) info.delta = fmax_std(info.reg_limit, 0.5 * info.delta);
                }
                __syncthreads();
            }
        }
        info.status = PQ_MAX_ITER_REACHED;
        return info.status;
    }

    // solver.hpp:1205-1259 unscale_results + restore_dual; the expanded box duals go to the RS set (scratch) and back
    __device__ void finish()
    {
        const int n = S.n, p = S.p, m = S.m;
        const gdbl* dl = at(D_DL);
        const gdbl* dli = at(D_DLI);
        const gdbl* db = at(D_DB);
        const gdbl* dbi = at(D_DBI);
        const double ci = rz_c_inv;
        for (int i = tid(); i < n; i += NT) v(V_R, FX)[i] *= dl[i];
        for (int i = tid(); i < p; i += NT) v(V_R, FY)[i] = v(V_R, FY)[i] * ci * dl[n + i];
        for (int i = tid(); i < m; i += NT) {
            double zl = v(V_R, FZL)[i] * ci * dl[n + p + i], zu = v(V_R, FZU)[i] * ci * dl[n + p + i];
            double sl = v(V_R, FSL)[i] * dli[n + p + i], su = v(V_R, FSU)[i] * dli[n + p + i];
            if (zl == 0) sl = 1e30;
            if (zu == 0) su = 1e30;
            v(V_R, FZL)[i] = zl; v(V_R, FZU)[i] = zu; v(V_R, FSL)[i] = sl; v(V_R, FSU)[i] = su;
        }
        for (int j = tid(); j < n; j += NT) {
            const int il = g(S.pos_l)[j], iu = g(S.pos_u)[j];
            v(V_RS, FZBL)[j] = il >= 0 ? v(V_R, FZBL)[il] * ci * db[j] : 0.0;
            v(V_RS, FSBL)[j] = il >= 0 ? v(V_R, FSBL)[il] * dbi[j] : 1e30;
            v(V_RS, FZBU)[j] = iu >= 0 ? v(V_R, FZBU)[iu] * ci * db[j] : 0.0;
            v(V_RS, FSBU)[j] = iu >= 0 ? v(V_R, FSBU)[iu] * dbi[j] : 1e30;
        }
        __syncthreads();
        for (int j = tid(); j < n; j += NT) {
            v(V_R, FZBL)[j] = v(V_RS, FZBL)[j]; v(V_R, FSBL)[j] = v(V_RS, FSBL)[j];
            v(V_R, FZBU)[j] = v(V_RS, FZBU)[j]; v(V_R, FSBU)[j] = v(V_RS, FSBU)[j];
        }
        __syncthreads();
    }
};

// update() of every instance with new vectors only (solver.hpp:218-308 with every optional matrix empty): the new values are scaled on
// the device with the Ruiz factors the instance got at setup (reuse_prev_scaling is forced for such updates, :283-285) and written over
// the scaled copies in the arena.  NULL = unchanged.  Non-finite bounds are stored as -/+1e30 times the scaling, like set_h_l / set_h_u.
__global__ void k_batch_update_vectors(const BatchShared* __restrict__ Sp, double* __restrict__ arena, const double* __restrict__ ruiz_c, const int* __restrict__ disabled,
                                       const double* __restrict__ c, const double* __restrict__ b, const double* __restrict__ h_l, const double* __restrict__ h_u,
                                       const double* __restrict__ x_l, const double* __restrict__ x_u)
{
    const BatchShared& S = *Sp;
    const int q = blockIdx.x, n = S.n, p = S.p, m = S.m;
    double* base = arena + (long long)q * S.stride;
    const double* dl = base + S.off[D_DL];
    const double* db = base + S.off[D_DB];
    const double rc = ruiz_c[q];
    if (c) for (int i = threadIdx.x; i < n; i += blockDim.x) base[S.off[D_C] + i] = c[(size_t)q * n + i] * (rc * dl[i]);
    if (b) for (int i = threadIdx.x; i < p; i += blockDim.x) base[S.off[D_B] + i] = b[(size_t)q * p + i] * dl[n + i];
    // (rows of G without any finite bound stay disabled: zero row, h = (-1, 1), data.hpp:144-169)
    if (h_l) for (int i = threadIdx.x; i < m; i += blockDim.x) { const double v = h_l[(size_t)q * m + i]; base[S.off[D_HL] + i] = (disabled[i] ? -1.0 : (v > -1e30 ? v : -1e30)) * dl[n + p + i]; }
    if (h_u) for (int i = threadIdx.x; i < m; i += blockDim.x) { const double v = h_u[(size_t)q * m + i]; base[S.off[D_HU] + i] = (disabled[i] ? 1.0 : (v < 1e30 ? v : 1e30)) * dl[n + p + i]; }
    if (x_l) for (int k = threadIdx.x; k < S.n_x_l; k += blockDim.x) { const int idx = S.x_l_idx[k]; base[S.off[D_XL] + k] = x_l[(size_t)q * n + idx] * db[idx]; }
    if (x_u) for (int k = threadIdx.x; k < S.n_x_u; k += blockDim.x) { const int idx = S.x_u_idx[k]; base[S.off[D_XU] + k] = x_u[(size_t)q * n + idx] * db[idx]; }
}

// update() with matrices (solver.hpp:218-308, update_P / update_A / update_G of :317-358): the caller's new values, in the CSC order they had at
// setup, overwrite the UNSCALED copies in the arena (k_ruiz_sparse ran in RUIZ_UNSCALE mode before this and scales everything again afterwards).
// map*[k] = position of the k-th input value in the stored upper-triangular P / transposed A, G, or -1 (strictly-lower entries of P; entries of a
// disabled row of G, which stay zero).  NULL = unchanged.
struct BatchAssignArgs {
    int nzP_in, nzA_in, nzG_in;
    const int *mapP, *mapA, *mapG, *disabled;
    const double *Px, *Ax, *Gx, *c, *b, *h_l, *h_u, *x_l, *x_u;
};
__global__ void k_batch_assign(const BatchShared* __restrict__ Sp, double* __restrict__ arena, BatchAssignArgs a)
{
    const BatchShared& S = *Sp;
    const int q = blockIdx.x, n = S.n, p = S.p, m = S.m, t = threadIdx.x, NT = blockDim.x;
    double* base = arena + (long long)q * S.stride;
    if (a.Px) for (int k = t; k < a.nzP_in; k += NT) { const int d = a.mapP[k]; if (d >= 0) base[S.off[D_PX] + d] = a.Px[(size_t)q * a.nzP_in + k]; }
    if (a.Ax) for (int k = t; k < a.nzA_in; k += NT) { const int d = a.mapA[k]; if (d >= 0) base[S.off[D_ATX] + d] = a.Ax[(size_t)q * a.nzA_in + k]; }
    if (a.Gx) for (int k = t; k < a.nzG_in; k += NT) { const int d = a.mapG[k]; if (d >= 0) base[S.off[D_GTX] + d] = a.Gx[(size_t)q * a.nzG_in + k]; }
    if (a.c) for (int i = t; i < n; i += NT) base[S.off[D_C] + i] = a.c[(size_t)q * n + i];
    if (a.b) for (int i = t; i < p; i += NT) base[S.off[D_B] + i] = a.b[(size_t)q * p + i];
    if (a.h_l) for (int i = t; i < m; i += NT) { const double v = a.h_l[(size_t)q * m + i]; base[S.off[D_HL] + i] = a.disabled[i] ? -1.0 : (v > -1e30 ? v : -1e30); }
    if (a.h_u) for (int i = t; i < m; i += NT) { const double v = a.h_u[(size_t)q * m + i]; base[S.off[D_HU] + i] = a.disabled[i] ? 1.0 : (v < 1e30 ? v : 1e30); }
    if (a.x_l) for (int k = t; k < S.n_x_l; k += NT) base[S.off[D_XL] + k] = a.x_l[(size_t)q * n + S.x_l_idx[k]];
    if (a.x_u) for (int k = t; k < S.n_x_u; k += NT) base[S.off[D_XU] + k] = a.x_u[(size_t)q * n + S.x_u_idx[k]];
}

// per instance: caller's (Ruiz-scaled) values -> front / grouped-row arenas and the AtA fronts (multistage ctor, :76-135)
template <int NT>
__global__ __launch_bounds__(NT) void k_batch_prepare(const BatchShared* __restrict__ Sp, double* __restrict__ arena)
{
    const BatchShared& S = *Sp;
    double* base = arena + (long long)blockIdx.x * S.stride;
    const int tid = threadIdx.x;
    double *Pf = base + S.off[B_PF], *XA = base + S.off[B_XA], *XG = base + S.off[B_XG];
    const double *Px = base + S.off[D_PX], *Ax = base + S.off[D_ATX], *Gx = base + S.off[D_GTX];
    for (int q = tid; q < S.nzP; q += NT) Pf[S.P_dst[q]] = Px[q];
    for (int q = tid; q < S.nzA; q += NT) XA[S.A_dst[q]] = Ax[q];
    for (int q = tid; q < S.nzG; q += NT) XG[S.G_dst[q]] = Gx[q];
    __syncthreads();
    for (int b = 0; b + 1 < S.M.N; ++b) {
        const int h = S.M.h[b];
        msdev::gram_stage<NT>(S.M, S.GA, XA, base + S.off[B_ATAF], b, 0, h * h);
    }
}

// WPE = waves per SIMD the register allocator must leave room for (occupancy vs spills; measured)
template <int NT, int MODE, int WPE>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_batch_ipm(const BatchShared* __restrict__ Sp, double* __restrict__ arena, const double* __restrict__ ruiz_c, pq_info* __restrict__ infos,
                                                                                                   double* __restrict__ prof_out, const int* __restrict__ order)
{
    extern __shared__ double sm[];
    __shared__ double red[NT / 64 > 0 ? NT / 64 : 1];
    // The shared descriptor and the per-stage structure tables are read on every stage of every chain sweep: keep them in
    // LDS (a dependent chain of global loads per stage was the largest single latency of the first version).
    __shared__ BatchShared Ssh;
    {
        const int words = (int)(sizeof(BatchShared) / sizeof(int));
        const int* src = reinterpret_cast<const int*>(Sp);
        int* dst = reinterpret_cast<int*>(&Ssh);
        for (int i = threadIdx.x; i < words; i += NT) dst[i] = src[i];
        __syncthreads();
        const int N = Ssh.M.N;
        int* mi = reinterpret_cast<int*>(sm + Ssh.meta_ofs);
        long long* ml = reinterpret_cast<long long*>(mi + 4 * N);
        for (int i = threadIdx.x; i < N; i += NT) {
            mi[i] = Ssh.M.w[i]; mi[N + i] = Ssh.M.off[i]; mi[2 * N + i] = Ssh.M.h[i]; mi[3 * N + i] = Ssh.M.start[i];
            ml[i] = Ssh.M.front_off[i]; ml[N + i] = Ssh.M.pan_off[i];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            Ssh.M.w = mi; Ssh.M.off = mi + N; Ssh.M.h = mi + 2 * N; Ssh.M.start = mi + 3 * N;
            Ssh.M.front_off = ml; Ssh.M.pan_off = ml + N;
        }
        __syncthreads();
    }
    const BatchShared& S = Ssh;
    const int q = order[blockIdx.x];  // workgroups start in block order: the instances expected to take longest first (pq_batch::solve)
    __shared__ IpmState state[NT / 64 > 0 ? NT / 64 : 1];
    IpmState& my = state[threadIdx.x >> 6];
    Ipm<NT, MODE, WPE> ipm(S, (gdbl*)(arena + (long long)q * S.stride), sm, red, my);
    my.info = pq_info{};
    for (int i = 0; i < NPROF; ++i) my.prof[i] = 0;
    const long long t_start = wall_clock64();
    my.rz_c = ruiz_c[q];
    my.rz_c_inv = 1.0 / ruiz_c[q];
    my.ks_rho = 0.0; my.ks_delta = 0.0; my.be_delta = 1.0;
    my.refine_enabled = 0; my.ks_use_refine = 0;
    ipm.solve_impl();
    __syncthreads();
    ipm.finish();
    if (threadIdx.x == 0) {
        my.prof[T_ALL] = wall_clock64() - t_start;
        my.prof[6] = t_start;  // absolute start tick (concurrency analysis)
        // Info's timing fields (results.hpp:76-82) from the device clock: seconds spent in factorisations / KKT solves / the whole solve
        my.info.kkt_factor_time = (double)(my.prof[T_ASM] + my.prof[T_FAC]) * 1e-8;
        my.info.kkt_solve_time = (double)my.prof[T_KS] * 1e-8;
        my.info.solve_time = (double)my.prof[T_ALL] * 1e-8;
        infos[q] = my.info;
        for (int i = 0; i < NPROF; ++i) prof_out[(long long)q * NPROF + i] = (double)my.prof[i] * 1e-8;
    }
}

struct Layout {
    long long off[NSLOT];
    long long stride = 0;
};

}  // namespace

// ------------------------------------------------------------------------------------------------ host class
class BatchSolver {
public:
    explicit BatchSolver(int device) : dev_(device)
    {
        pq_settings_default(&settings_);
        settings_.kkt_solver = PQ_SPARSE_MULTISTAGE;
        PQ_HIP(hipSetDevice(dev_));
        PQ_HIP(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
    }
    ~BatchSolver()
    {
        (void)hipSetDevice(dev_);
        if (st_) { (void)hipStreamSynchronize(st_); (void)hipStreamDestroy(st_); }
    }
    pq_settings& settings() { return settings_; }
    int batch() const { return batch_; }
    int n() const { return n_; }
    int p() const { return p_; }
    int m() const { return m_; }

    // SparseSolver::setup (solver.hpp:1297-1308) for every instance; patterns shared, values [batch][nnz] / [batch][len]
    bool setup(int batch, int n, int p, int m, const int* Pp, const int* Pi, const double* Px, const double* c, const int* Ap, const int* Ai, const double* Ax, const double* b,
               const int* Gp, const int* Gi, const double* Gx, const double* h_l, const double* h_u, const double* x_l, const double* x_u)
    {
        if (batch <= 0 || n <= 0) throw std::runtime_error("batch setup: bad dimensions");
        if (settings_.kkt_solver != PQ_SPARSE_MULTISTAGE) throw std::runtime_error("batch mode: only kkt_solver = sparse_multistage");
        PQ_HIP(hipSetDevice(dev_));
        batch_ = batch;
        const int nzP = Pp[n], nzA = Ap ? Ap[n] : 0, nzG = Gp ? Gp[n] : 0;
        // ---- which bounds are finite is part of the shared structure: the caller's pattern must be the same in every instance ----
        auto finite_pattern = [&](const double* v, int len, bool lower, std::vector<char>& out) {
            out.assign(len, 0);
            if (!v) return;
            for (int i = 0; i < len; ++i) out[i] = lower ? v[i] > -1e30 : v[i] < 1e30;
            for (int q = 1; q < batch; ++q)
                for (int i = 0; i < len; ++i)
                    if ((lower ? v[(size_t)q * len + i] > -1e30 : v[(size_t)q * len + i] < 1e30) != (out[i] != 0)) throw std::runtime_error("batch setup: instances differ in which bounds are finite");
        };
        finite_pattern(Gp ? h_l : nullptr, Gp ? m : 0, true, fin_hl_); finite_pattern(Gp ? h_u : nullptr, Gp ? m : 0, false, fin_hu_);
        finite_pattern(x_l, n, true, fin_xl_); finite_pattern(x_u, n, false, fin_xu_);
        // ---- per-instance Data on the host (transposes, bound lists; the instances are independent: one host thread per slice) ----
        std::vector<std::unique_ptr<HostData>> data(batch);
        auto build = [&](int lo, int hi) {
            for (int i = lo; i < hi; ++i) {
                data[i] = make_sparse_host_data(n, p, m, Pp, Pi, Px + (size_t)i * nzP, c + (size_t)i * n, Ap, Ai, Ax ? Ax + (size_t)i * nzA : nullptr, b ? b + (size_t)i * p : nullptr,
                                                Gp, Gi, Gx ? Gx + (size_t)i * nzG : nullptr, h_l ? h_l + (size_t)i * m : nullptr, h_u ? h_u + (size_t)i * m : nullptr,
                                                x_l ? x_l + (size_t)i * n : nullptr, x_u ? x_u + (size_t)i * n : nullptr);
            }
        };
        {
            const int nthreads = std::max(1, std::min<int>(batch, (int)std::thread::hardware_concurrency()));
            std::vector<std::thread> pool;
            std::vector<std::exception_ptr> errs(nthreads);
            for (int t = 0; t < nthreads; ++t) {
                const int lo = (int)((long long)batch * t / nthreads), hi = (int)((long long)batch * (t + 1) / nthreads);
                pool.emplace_back([&, t, lo, hi] { try { build(lo, hi); } catch (...) { errs[t] = std::current_exception(); } });
            }
            for (auto& th : pool) th.join();
            for (auto& e : errs) if (e) std::rethrow_exception(e);
        }
        const HostData& d0 = *data[0];
        n_ = d0.n; p_ = d0.p; m_ = d0.m;
        for (int i = 1; i < batch; ++i) {
            const HostData& d = *data[i];
            const bool same = d.n_h_l == d0.n_h_l && d.n_h_u == d0.n_h_u && d.n_x_l == d0.n_x_l && d.n_x_u == d0.n_x_u &&
                              std::equal(d.h_l_idx.begin(), d.h_l_idx.begin() + d.n_h_l, d0.h_l_idx.begin()) &&
                              std::equal(d.h_u_idx.begin(), d.h_u_idx.begin() + d.n_h_u, d0.h_u_idx.begin()) &&
                              std::equal(d.x_l_idx.begin(), d.x_l_idx.begin() + d.n_x_l, d0.x_l_idx.begin()) &&
                              std::equal(d.x_u_idx.begin(), d.x_u_idx.begin() + d.n_x_u, d0.x_u_idx.begin());
            if (!same) throw std::runtime_error("batch setup: instances differ in which bounds are finite");
        }
        fin_hl_.resize(m_, 0); fin_hu_.resize(m_, 0);
        // ---- shared structure ----
        pq_sparse_data desc = d0.sparse_descriptor();
        multistage::analyse(&desc, sym_);
        build_shared(d0);
        build_update_maps(n, p, m, Pp, Pi, c, Ap, Ai, b, Gp, Gi, h_l, h_u, x_l, x_u, d0);
        // ---- per-instance arena ----
        arena_.alloc((size_t)layout_.stride * batch);
        arena_.zero(st_);
        ruiz_c_.alloc(batch);
        infos_.alloc(batch);
        order_.alloc(batch);
        order_h_.resize((size_t)batch);
        for (int i = 0; i < batch; ++i) order_h_[(size_t)i] = i;
        PQ_HIP(hipMemcpyAsync(order_.p, order_h_.data(), sizeof(int) * (size_t)batch, hipMemcpyHostToDevice, st_));
        prof_.alloc((size_t)batch * NPROF);
        infos_h_.resize(batch);
        std::vector<double> stage((size_t)layout_.stride * std::min(batch, STAGE_INST));
        for (int i0 = 0; i0 < batch; i0 += STAGE_INST) {
            const int cnt = std::min(STAGE_INST, batch - i0);
            std::fill(stage.begin(), stage.begin() + (size_t)layout_.stride * cnt, 0.0);
            for (int k = 0; k < cnt; ++k) pack_instance(*data[i0 + k], stage.data() + (size_t)layout_.stride * k);
            PQ_HIP(hipMemcpyAsync(arena_.p + (size_t)layout_.stride * i0, stage.data(), sizeof(double) * (size_t)layout_.stride * cnt, hipMemcpyHostToDevice, st_));
            stream_wait(st_);
        }
        // RuizEquilibration::scale_data of every instance (sparse/preconditioner.hpp:65-222) in one launch, then the front arenas
        launch_ruiz(RUIZ_COMPUTE);
        launch_prepare();
        stream_wait(st_);
        setup_done_ = true;
        return true;
    }

    // solve() of every instance; returns the number of instances that ended PIQP_SOLVED
    int solve()
    {
        if (!setup_done_) throw std::runtime_error("batch solver not set up");
        if (!verify_settings(settings_)) {
            for (auto& i : infos_h_) { i = pq_info{}; i.status = PQ_INVALID_SETTINGS; }
            return 0;
        }
        PQ_HIP(hipSetDevice(dev_));
        shared_h_.set = settings_;
        PQ_HIP(hipMemcpyAsync(shared_.p, &shared_h_, sizeof(BatchShared), hipMemcpyHostToDevice, st_));
        hipEvent_t e0, e1;
        PQ_HIP(hipEventCreate(&e0)); PQ_HIP(hipEventCreate(&e1));
        PQ_HIP(hipEventRecord(e0, st_));
        launch_ipm();
        PQ_HIP(hipEventRecord(e1, st_));
        PQ_HIP(hipGetLastError());
        PQ_HIP(hipMemcpyAsync(infos_h_.data(), infos_.p, sizeof(pq_info) * batch_, hipMemcpyDeviceToHost, st_));
        stream_wait(st_);
        float ms = 0.f;
        PQ_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        last_kernel_ms_ = ms;
        int solved = 0;
        for (const auto& i : infos_h_) solved += i.status == PQ_SOLVED;
        // Start order of the NEXT solve: a batch of more instances than the device holds at once runs as a list of workgroups whose durations follow their
        // iteration counts (7 .. 11 at C4), and the last ones to start decide when the launch ends; receding-horizon batches repeat their counts from one
        // solve to the next almost exactly, so the instances that took longest start first next time (a stable counting sort: deterministic; the results of
        // an instance do not depend on where it starts).  8192 QPs: the tail behind the second round of workgroups 0.87 -> ~0.4 ms.
        if (lpt_ && !debug_token("batch_no_lpt")) {
            const int cap = std::max(settings_.max_iter, 1) + 1;
            std::vector<int> cnt((size_t)cap + 1, 0);
            for (const auto& i : infos_h_) cnt[(size_t)std::min(std::max(i.iter, 0), cap - 1)]++;
            std::vector<int> first((size_t)cap, 0);
            int run = 0;
            for (int it = cap - 1; it >= 0; --it) { first[(size_t)it] = run; run += cnt[(size_t)it]; }
            for (int i = 0; i < batch_; ++i) order_h_[(size_t)first[(size_t)std::min(std::max(infos_h_[(size_t)i].iter, 0), cap - 1)]++] = i;
            PQ_HIP(hipMemcpyAsync(order_.p, order_h_.data(), sizeof(int) * (size_t)batch_, hipMemcpyHostToDevice, st_));
        }
        return solved;
    }
    // update() of every instance, vectors only; the set of finite bounds must be the one given at setup (it is part of the shared structure)
    bool update_vectors(const double* c, const double* b, const double* h_l, const double* h_u, const double* x_l, const double* x_u)
    {
        if (!setup_done_) throw std::runtime_error("batch solver not set up");
        PQ_HIP(hipSetDevice(dev_));
        check_patterns(h_l, h_u, x_l, x_u);
        DBuf<double> dc, dbv, dhl, dhu, dxl, dxu;
        auto up = [&](DBuf<double>& d, const double* v, int len) -> const double* {
            if (!v || len == 0) return nullptr;
            d.alloc((size_t)batch_ * len);
            PQ_HIP(hipMemcpyAsync(d.p, v, sizeof(double) * (size_t)batch_ * len, hipMemcpyHostToDevice, st_));
            return d.p;
        };
        const double *pc = up(dc, c, n_), *pb = up(dbv, b, p_), *phl = up(dhl, h_l, m_), *phu = up(dhu, h_u, m_), *pxl = up(dxl, x_l, n_), *pxu = up(dxu, x_u, n_);
        hipLaunchKernelGGL(k_batch_update_vectors, dim3(batch_), dim3(128), 0, st_, shared_.p, arena_.p, ruiz_c_.p, disabled_d_, pc, pb, phl, phu, pxl, pxu);
        PQ_HIP(hipGetLastError());
        stream_wait(st_);
        return true;
    }
    // update() of every instance with new matrix values (patterns and the set of finite bounds as given at setup) and / or vectors: per instance
    // unscale_data -> assign -> scale_data (fresh equilibration unless settings.preconditioner_reuse_on_update), all on the device
    bool update_data(const double* Px, const double* Ax, const double* Gx, const double* c, const double* b, const double* h_l, const double* h_u, const double* x_l,
                     const double* x_u)
    {
        if (!Px && !Ax && !Gx) return update_vectors(c, b, h_l, h_u, x_l, x_u);
        if (!setup_done_) throw std::runtime_error("batch solver not set up");
        PQ_HIP(hipSetDevice(dev_));
        check_patterns(h_l, h_u, x_l, x_u);
        DBuf<double> dP, dA, dG, dc, dbv, dhl, dhu, dxl, dxu;
        auto up = [&](DBuf<double>& d, const double* v, int len) -> const double* {
            if (!v || len == 0) return nullptr;
            d.alloc((size_t)batch_ * len);
            PQ_HIP(hipMemcpyAsync(d.p, v, sizeof(double) * (size_t)batch_ * len, hipMemcpyHostToDevice, st_));
            return d.p;
        };
        BatchAssignArgs a{};
        a.nzP_in = nzP_in_; a.nzA_in = nzA_in_; a.nzG_in = nzG_in_;
        a.mapP = mapP_; a.mapA = mapA_; a.mapG = mapG_; a.disabled = disabled_d_;
        a.Px = up(dP, Px, nzP_in_); a.Ax = up(dA, Ax, nzA_in_); a.Gx = up(dG, Gx, nzG_in_);
        a.c = up(dc, c, n_); a.b = up(dbv, b, p_); a.h_l = up(dhl, h_l, m_); a.h_u = up(dhu, h_u, m_); a.x_l = up(dxl, x_l, n_); a.x_u = up(dxu, x_u, n_);
        launch_ruiz(RUIZ_UNSCALE);
        hipLaunchKernelGGL(k_batch_assign, dim3(batch_), dim3(128), 0, st_, shared_.p, arena_.p, a);
        PQ_HIP(hipGetLastError());
        launch_ruiz(settings_.preconditioner_reuse_on_update ? RUIZ_REUSE : RUIZ_COMPUTE);
        launch_prepare();
        stream_wait(st_);
        return true;
    }
    double last_kernel_ms() const { return last_kernel_ms_; }
    void set_start_order(bool longest_first)
    {
        lpt_ = longest_first;
        if (!lpt_ && setup_done_) {
            PQ_HIP(hipSetDevice(dev_));
            for (int i = 0; i < batch_; ++i) order_h_[(size_t)i] = i;
            PQ_HIP(hipMemcpyAsync(order_.p, order_h_.data(), sizeof(int) * (size_t)batch_, hipMemcpyHostToDevice, st_));
            stream_wait(st_);
        }
    }
    const pq_info& info(int i) const { return infos_h_.at(i); }

    // field k of Variables (x, y, z_l, z_u, z_bl, z_bu, s_l, s_u, s_bl, s_bu) of all instances -> host [batch][len]
    void get_result(int field, double* out_host)
    {
        const int len = field_len(field);
        if (len == 0) return;
        PQ_HIP(hipSetDevice(dev_));
        PQ_HIP(hipMemcpy2DAsync(out_host, sizeof(double) * len, arena_.p + layout_.off[V_R + field], sizeof(double) * layout_.stride, sizeof(double) * len, batch_,
                                hipMemcpyDeviceToHost, st_));
        stream_wait(st_);
    }
    int field_len(int field) const
    {
        switch (field) { case FX: case FZBL: case FZBU: case FSBL: case FSBU: return n_; case FY: return p_; default: return m_; }
    }
    void block_info(std::vector<int>& out) const
    {
        out.clear();
        for (const auto& b : sym_.block_info) { out.push_back(b.start); out.push_back(b.diag_size); out.push_back(b.off_diag_size); }
    }
    // seconds per stage of one instance, measured by the device clock inside the kernel (see T_* above)
    void get_profile(int instance, double* out8)
    {
        PQ_HIP(hipSetDevice(dev_));
        PQ_HIP(hipMemcpy(out8, prof_.p + (size_t)instance * NPROF, sizeof(double) * NPROF, hipMemcpyDeviceToHost));
    }
    int threads_per_qp() const { return nt_; }
    int mode() const { return mode_; }

private:
    static constexpr int STAGE_INST = 256;

    void check_patterns(const double* h_l, const double* h_u, const double* x_l, const double* x_u) const
    {
        auto pattern_ok = [&](const double* v, int len, const std::vector<char>& finite, bool lower) {
            if (!v) return true;
            for (int q = 0; q < batch_; ++q)
                for (int i = 0; i < len; ++i) {
                    const double x = v[(size_t)q * len + i];
                    if ((lower ? x > -1e30 : x < 1e30) != (finite[i] != 0)) return false;
                }
            return true;
        };
        if (!pattern_ok(h_l, m_, fin_hl_, true) || !pattern_ok(h_u, m_, fin_hu_, false) || !pattern_ok(x_l, n_, fin_xl_, true) || !pattern_ok(x_u, n_, fin_xu_, false))
            throw std::runtime_error("batch update: the set of finite bounds differs from the one given at setup");
    }

    // Where the caller's k-th value of P / A / G lives in the arena: found by pushing the values 1, 2, 3, ... through the same host path the real
    // values took at setup (make_sparse_host_data: upper-triangle extraction, transposes, disabled rows of G zeroed).
    void build_update_maps(int n, int p, int m, const int* Pp, const int* Pi, const double* c, const int* Ap, const int* Ai, const double* b, const int* Gp, const int* Gi,
                           const double* h_l, const double* h_u, const double* x_l, const double* x_u, const HostData& d0)
    {
        nzP_in_ = Pp[n]; nzA_in_ = Ap ? Ap[n] : 0; nzG_in_ = Gp ? Gp[n] : 0;
        Vec iP(std::max(nzP_in_, 1)), iA(std::max(nzA_in_, 1)), iG(std::max(nzG_in_, 1));
        for (int k = 0; k < nzP_in_; ++k) iP[k] = k + 1;
        for (int k = 0; k < nzA_in_; ++k) iA[k] = k + 1;
        for (int k = 0; k < nzG_in_; ++k) iG[k] = k + 1;
        auto probe = make_sparse_host_data(n, p, m, Pp, Pi, iP.data(), c, Ap, Ai, Ap ? iA.data() : nullptr, b, Gp, Gi, Gp ? iG.data() : nullptr, h_l, h_u, x_l, x_u);
        auto invert = [&](const Vec& stored, int nz_in) {
            std::vector<int> map(std::max(nz_in, 1), -1);
            for (size_t t = 0; t < stored.size(); ++t) if (stored[t] > 0.0) map[(int)stored[t] - 1] = (int)t;
            return map;
        };
        mapP_ = up(ibufs_, invert(probe->sP_utri.val, nzP_in_));
        mapA_ = up(ibufs_, invert(probe->sAT.val, nzA_in_));
        mapG_ = up(ibufs_, invert(probe->sGT.val, nzG_in_));
        std::vector<int> dis(std::max(m_, 1), 0);
        for (int i = 0; i < m_; ++i) dis[i] = !fin_hl_[i] && !fin_hu_[i];
        disabled_d_ = up(ibufs_, dis);
        rzPp_ = up(ibufs_, d0.sP_utri.colptr);
        rzPi_ = up(ibufs_, d0.sP_utri.rowind.empty() ? std::vector<int>(1, 0) : d0.sP_utri.rowind);
        stream_wait(st_);
    }

    // sparse/preconditioner.hpp on the arena: one workgroup per instance (ruiz_kernels.hip)
    void launch_ruiz(int mode)
    {
        const BatchShared& S = shared_h_;
        RuizSparseArgs a;
        a.n = n_; a.p = p_; a.m = m_;
        a.Pp = rzPp_; a.Pi = rzPi_; a.ATp = S.AT_p; a.ATi = S.AT_i; a.GTp = S.GT_p; a.GTi = S.GT_i;
        a.stride = layout_.stride;
        auto at = [&](int slot) { return arena_.p + layout_.off[slot]; };
        a.Px = at(D_PX); a.ATx = at(D_ATX); a.GTx = at(D_GTX); a.c = at(D_C); a.xbs = at(D_XBS);
        a.delta = at(D_DL); a.delta_inv = at(D_DLI); a.delta_b = at(D_DB); a.delta_b_inv = at(D_DBI); a.tmp = at(K_WX);
        a.b = at(D_B); a.h_l = at(D_HL); a.h_u = at(D_HU); a.x_l = at(D_XL); a.x_u = at(D_XU);
        a.x_l_idx = S.x_l_idx; a.x_u_idx = S.x_u_idx; a.n_x_l = S.n_x_l; a.n_x_u = S.n_x_u;
        a.c_scale = ruiz_c_.p;
        a.mode = mode; a.scale_cost = settings_.preconditioner_scale_cost != 0; a.max_iter = settings_.preconditioner_iter;
        launch_ruiz_sparse(a, batch_, n_ + p_ + m_ <= 512 ? 64 : 256, st_);
    }

    template <class T>
    const T* up(std::vector<DBuf<T>>& pool, const std::vector<T>& h)
    {
        pool.emplace_back();
        upload_vec(pool.back(), h, st_);
        return pool.back().p;
    }

    void build_shared(const HostData& d)
    {
        const int n = n_, p = p_, m = m_;
        BatchShared& S = shared_h_;
        S = BatchShared{};
        S.n = n; S.p = p; S.m = m; S.n_h_l = d.n_h_l; S.n_h_u = d.n_h_u; S.n_x_l = d.n_x_l; S.n_x_u = d.n_x_u;
        ibufs_.clear(); lbufs_.clear();
        ibufs_.reserve(64); lbufs_.reserve(16);
        std::vector<int> has_l(std::max(m, 1), 0), has_u(std::max(m, 1), 0), pos_l(n, -1), pos_u(n, -1);
        for (int i = 0; i < d.n_h_l; ++i) has_l[d.h_l_idx[i]] = 1;
        for (int i = 0; i < d.n_h_u; ++i) has_u[d.h_u_idx[i]] = 1;
        for (int i = 0; i < d.n_x_l; ++i) pos_l[d.x_l_idx[i]] = i;
        for (int i = 0; i < d.n_x_u; ++i) pos_u[d.x_u_idx[i]] = i;
        S.h_l_idx = up(ibufs_, d.h_l_idx); S.h_u_idx = up(ibufs_, d.h_u_idx); S.x_l_idx = up(ibufs_, d.x_l_idx); S.x_u_idx = up(ibufs_, d.x_u_idx);
        S.has_l = up(ibufs_, has_l); S.has_u = up(ibufs_, has_u); S.pos_l = up(ibufs_, pos_l); S.pos_u = up(ibufs_, pos_u);
        // symmetrised P and row-oriented A, G with value-source maps
        const Csc& U = d.sP_utri;
        S.nzP = U.nnz(); S.nzA = d.sAT.nnz(); S.nzG = d.sGT.nnz();
        {
            std::vector<int> fp(n + 1, 0);
            for (int j = 0; j < n; ++j) for (int q = U.colptr[j]; q < U.colptr[j + 1]; ++q) { fp[j + 1]++; if (U.rowind[q] != j) fp[U.rowind[q] + 1]++; }
            for (int j = 0; j < n; ++j) fp[j + 1] += fp[j];
            std::vector<int> fi(fp[n]), src(fp[n]), nx(fp.begin(), fp.end() - 1);
            for (int j = 0; j < n; ++j) for (int q = U.colptr[j]; q < U.colptr[j + 1]; ++q) { const int t = nx[j]++; fi[t] = U.rowind[q]; src[t] = q; }
            for (int j = 0; j < n; ++j) for (int q = U.colptr[j]; q < U.colptr[j + 1]; ++q) { const int i = U.rowind[q]; if (i != j) { const int t = nx[i]++; fi[t] = j; src[t] = q; } }
            S.Pf_p = up(ibufs_, fp); S.Pf_i = up(ibufs_, fi); S.Pf_src = up(ibufs_, src);
        }
        auto rows_of = [&](const Csc& T, const int*& Tp, const int*& Ti, const int*& Rp, const int*& Ri, const int*& Rs) {
            std::vector<int> tp(T.colptr), ti(T.rowind);
            if (tp.empty()) tp.assign(1, 0);
            Tp = up(ibufs_, tp); Ti = up(ibufs_, ti);
            const int cols = T.cols, nnz = T.nnz();
            std::vector<int> rp(n + 1, 0), ri(nnz), rs(nnz);
            for (int q = 0; q < nnz; ++q) rp[T.rowind[q] + 1]++;
            for (int j = 0; j < n; ++j) rp[j + 1] += rp[j];
            std::vector<int> nx(rp.begin(), rp.end() - 1);
            for (int k = 0; k < cols; ++k) for (int q = T.colptr[k]; q < T.colptr[k + 1]; ++q) { const int t = nx[T.rowind[q]]++; ri[t] = k; rs[t] = q; }
            Rp = up(ibufs_, rp); Ri = up(ibufs_, ri); Rs = up(ibufs_, rs);
        };
        rows_of(d.sAT, S.AT_p, S.AT_i, S.A_p, S.A_i, S.A_src);
        rows_of(d.sGT, S.GT_p, S.GT_i, S.G_p, S.G_i, S.G_src);
        // multistage structure
        std::vector<int> start(sym_.N);
        for (int b = 0; b < sym_.N; ++b) start[b] = sym_.block_info[b].start;
        S.M = MsMeta{sym_.N, sym_.arrow, n, up(ibufs_, sym_.w), up(ibufs_, sym_.off), up(ibufs_, sym_.h), up(ibufs_, start), up(lbufs_, sym_.front_off), up(lbufs_, sym_.pan_off)};
        S.GA = GroupMeta{up(ibufs_, sym_.A.row_ptr), up(ibufs_, sym_.A.rows), up(lbufs_, sym_.A.x_off)};
        S.GG = GroupMeta{up(ibufs_, sym_.G.row_ptr), up(ibufs_, sym_.G.rows), up(lbufs_, sym_.G.x_off)};
        S.P_dst = up(lbufs_, sym_.P_dst); S.A_dst = up(lbufs_, sym_.A.dst); S.G_dst = up(lbufs_, sym_.G.dst);
        {
            std::vector<int> eb, erc;
            for (int b = 0; b < sym_.N; ++b) {
                const int h = sym_.h[b];
                if (h >= 65536) throw std::runtime_error("batch setup: a stage is too wide for this backend");
                for (int c = 0; c < h; ++c) for (int r = c; r < h; ++r) { eb.push_back(b); erc.push_back(r | (c << 16)); }
            }
            S.n_ent = (int)eb.size();
            S.ent_b = up(ibufs_, eb); S.ent_rc = up(ibufs_, erc);
            // the same entries as absolute front-arena offsets + the x_reg entry that lands on them (-1: none): what a QP without inequality rows needs of
            // the assembly (no per-stage row loop), two loads instead of a chain through the stage tables
            std::vector<int> eat(eb.size()), exr(eb.size());
            bool fits = sym_.front_doubles < (1LL << 31);
            for (size_t e = 0; e < eb.size(); ++e) {
                const int b = eb[e], r = erc[e] & 0xffff, c = erc[e] >> 16;
                eat[e] = (int)(sym_.front_off[b] + r + (long long)c * sym_.h[b]);
                exr[e] = (r == c && c < sym_.w[b]) ? sym_.block_info[b].start + c : -1;
            }
            S.ent_at = up(ibufs_, eat); S.ent_xr = up(ibufs_, exr);
            S.ent_flat = fits ? 1 : 0;
        }
        // chain workspace: front + carried update + inverse (factor) / 2 vectors + panel (solve)
        int max_u = 0;
        long long max_pan = 0;
        for (int b = 0; b < sym_.N; ++b) {
            max_u = std::max(max_u, sym_.h[b] - sym_.w[b]);
            max_pan = std::max(max_pan, (long long)sym_.h[b] * sym_.w[b] + (long long)sym_.w[b] * sym_.w[b]);
        }
        S.fcap = sym_.max_h * sym_.max_h;
        S.lofs = S.fcap + max_u * max_u;
        S.hcap = std::max(1, sym_.max_h);
        const long long fdoubles = (long long)S.lofs + (long long)sym_.max_w * sym_.max_w;
        const long long sdoubles = 2LL * S.hcap + max_pan;
        const long long small_chain = std::max<long long>((long long)sym_.max_w * sym_.max_w, 2LL * S.hcap);  // inverse (factor) / two stage vectors (solve)
        const long long res_doubles = sym_.front_doubles + sym_.pan_doubles + n + small_chain;
        bool wave_pan = false;
        const long long wave_doubles = sym_.qpan_doubles + n;
        nt_ = (n <= 512 && sym_.max_h <= 24) ? 64 : 256;
        if (const char* e = debug_token("batch_wpe")) wpe_ = std::atoi(e);  // waves per SIMD the kernel is compiled for (2 .. 6)
        if (const char* e = debug_token("batch_mode")) forced_mode_ = std::atoi(e);  // forces the chain working-set mode (tests of the fallback modes)
        if (nt_ == 64 && sym_.max_h * sym_.max_h <= 64 && wave_doubles * (long long)sizeof(double) <= RESIDENT_LIMIT_BYTES && sym_.max_w <= msdev::WAVE_WMAX && (forced_mode_ < 0 || forced_mode_ == MODE_WAVE)) {
            mode_ = MODE_WAVE;
            // Register budget against waves in flight (round 3, after the out-of-line members got the launched variant's budget): the interior-point code
            // spills at 102 registers, and under load the kernel is bound by what its waves move through memory -- scratch included --, not by how many
            // of them wait: 8192 QPs 6.77 ms compiled for four waves per SIMD (128 registers, 16 workgroups per CU), 7.25 for five (18 per CU, the LDS
            // bound), 7.54 for three, 8.97 for six.  A batch that fits in one round of workgroups anyway takes the largest budget that still holds it:
            // 1024 QPs 2.41 ms for two waves per SIMD, 2.43 three, 2.52 four, 2.67 five.
            if (!debug_token("batch_wpe")) {
                int dev = 0, cus = 256;
                if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
                const long long per_cu = ((long long)batch_ + cus - 1) / std::max(cus, 1);
                wpe_ = per_cu <= 8 ? 2 : (per_cu <= 12 ? 3 : 4);
            }
            S.res_f = 0; S.res_pan = 0; S.res_x = (int)sym_.qpan_doubles; S.res_chain = S.res_x + n;
            S.fcap = 0; S.lofs = 0;
            S.chain_lds_doubles = (int)wave_doubles;
            wave_pan = true;
            {   // the register-carried chain substitution: no arrow, no gaps, and a leading run of stages that all eliminate W columns (W <= 6: the row shift
                // and h = W + u <= 16 stay inside one row of lanes); what follows the run -- usually one terminal stage -- takes the LDS steps
                int nst = sym_.N;
                while (nst > 0 && sym_.h[nst - 1] == 0) --nst;
                const int W0 = nst > 0 ? sym_.w[0] : 0;
                bool chain = sym_.arrow == 0 && nst >= 1 && nst <= 64 && W0 >= 1 && W0 <= 6 && !debug_token("batch_no_chain_reg");
                for (int b = 0; chain && b < nst; ++b) {
                    chain = sym_.h[b] == sym_.w[b] + sym_.off[b] && sym_.h[b] <= 16;
                    if (chain && b + 1 < nst) chain = sym_.block_info[b + 1].start == sym_.block_info[b].start + sym_.w[b] && sym_.off[b] <= sym_.w[b + 1];
                    if (chain && b + 1 == nst) chain = sym_.off[b] == 0;
                }
                int K = 0;
                while (chain && K < nst && sym_.w[K] == W0 && sym_.off[K] == sym_.off[0] && sym_.off[K] <= W0 && sym_.off[K] >= 1 &&
                       sym_.qpan_off[K + 1] - sym_.qpan_off[K] == (long long)W0 * W0 + (long long)sym_.off[0] * W0 &&
                       sym_.front_off[K + 1] - sym_.front_off[K] == (long long)sym_.h[0] * sym_.h[0] && sym_.h[K] == sym_.h[0]) ++K;
                const bool ok = chain && K >= 4;
                S.chain_rows = (ok && 4 * W0 * W0 <= n && !debug_token("batch_no_chain_rows")) ? 1 : 0;  // ((2 W)^2 doubles of staging in the solve vector's LDS)
                if (debug_token("batch_chain_info"))
                    std::fprintf(stderr, "[piqp_amd] batch chain: %d stages, arrow %d, register-carried substitution for the first %d (w = %d, u = %d); one lane per row in their factorisation: %s "
                                 "(kernel variant for %d waves per SIMD)\n", nst, sym_.arrow, ok ? K : 0, W0, nst > 0 ? sym_.off[0] : 0, S.chain_rows ? (wpe_ <= 3 ? "yes" : "no (variant for full compute units)") : "no", wpe_);
                S.chain_reg_w = ok ? W0 : 0;
                S.chain_reg_k = ok ? K : 0;
                S.chain_reg_nst = ok ? nst : 0;
            }
        } else if (res_doubles * (long long)sizeof(double) <= RESIDENT_LIMIT_BYTES && (forced_mode_ < 0 || forced_mode_ == MODE_RESIDENT)) {
            mode_ = MODE_RESIDENT;
            S.res_f = 0; S.res_pan = (int)sym_.front_doubles; S.res_x = S.res_pan + (int)sym_.pan_doubles; S.res_chain = S.res_x + n;
            S.fcap = 0; S.lofs = 0;
            S.chain_lds_doubles = (int)res_doubles;
        } else if (std::max(fdoubles, sdoubles) * (long long)sizeof(double) <= LDS_LIMIT_BYTES) {
            mode_ = MODE_STAGED;
            S.chain_lds_doubles = (int)std::max(fdoubles, sdoubles);
        } else {
            mode_ = MODE_HBM;
            S.fcap = 0;
            S.lofs = 0;
            if (small_chain * (long long)sizeof(double) > LDS_LIMIT_BYTES) throw std::runtime_error("batch setup: a stage is too wide for this backend");
            S.chain_lds_doubles = (int)small_chain;
        }
        if (wave_pan) S.M.pan_off = up(lbufs_, sym_.qpan_off);  // compact Linv | Q panels
        S.meta_ofs = S.chain_lds_doubles;
        S.chain_lds_doubles += 4 * sym_.N + 2;  // 4 int + 2 int64 tables of N entries
        // arena layout
        long long o = 0;
        auto put = [&](int slot, long long cnt) { layout_.off[slot] = o; o += (cnt + 1) & ~1LL; };
        put(D_PX, S.nzP); put(D_ATX, S.nzA); put(D_GTX, S.nzG); put(D_C, n); put(D_B, p); put(D_HL, m); put(D_HU, m); put(D_XL, n); put(D_XU, n); put(D_XBS, n);
        put(D_DL, n + p + m); put(D_DLI, n + p + m); put(D_DB, n); put(D_DBI, n);
        const long long flen[NF] = {n, p, m, m, n, n, m, m, n, n};
        for (int set : {(int)V_R, (int)V_NR, (int)V_RS, (int)V_ST, (int)V_PX}) for (int f = 0; f < NF; ++f) put(set + f, flen[f]);
        put(K_SL, m); put(K_SU, m); put(K_SBL, n); put(K_SBU, n); put(K_ZLI, m); put(K_ZUI, m); put(K_ZBLI, n); put(K_ZBUI, n); put(K_XREG, n); put(K_ZREG, m); put(K_ZREGR, m);
        put(K_RXB, n); put(K_RZB, m); put(K_WX, n); put(K_LZ, m); put(K_EX, n); put(K_EY, p); put(K_EZ, m); put(K_RLX, n); put(K_RLY, p); put(K_RLZ, m);
        put(B_ZINV, m); put(B_PF, sym_.front_doubles); put(B_ATAF, sym_.front_doubles); put(B_F, sym_.front_doubles); put(B_PAN, sym_.pan_doubles);
        put(B_XA, sym_.A.x_doubles); put(B_XG, sym_.G.x_doubles);
        layout_.stride = o;
        for (int s = 0; s < NSLOT; ++s) S.off[s] = layout_.off[s];
        S.stride = layout_.stride;
        S.set = settings_;
        shared_.alloc(1);
        PQ_HIP(hipMemcpyAsync(shared_.p, &S, sizeof(BatchShared), hipMemcpyHostToDevice, st_));
        stream_wait(st_);
    }

    void pack_instance(const HostData& d, double* dst) const
    {
        auto cp = [&](int slot, const Vec& v, size_t cnt) { if (cnt) std::copy(v.begin(), v.begin() + cnt, dst + layout_.off[slot]); };
        cp(D_PX, d.sP_utri.val, d.sP_utri.val.size()); cp(D_ATX, d.sAT.val, d.sAT.val.size()); cp(D_GTX, d.sGT.val, d.sGT.val.size());
        cp(D_C, d.c, n_); cp(D_B, d.b, p_); cp(D_HL, d.h_l, m_); cp(D_HU, d.h_u, m_); cp(D_XL, d.x_l, n_); cp(D_XU, d.x_u, n_); cp(D_XBS, d.x_b_scaling, n_);
    }

    void launch_prepare()
    {
        if (nt_ == 64) hipLaunchKernelGGL(k_batch_prepare<64>, dim3(batch_), dim3(64), 0, st_, shared_.p, arena_.p);
        else hipLaunchKernelGGL(k_batch_prepare<256>, dim3(batch_), dim3(256), 0, st_, shared_.p, arena_.p);
        PQ_HIP(hipGetLastError());
    }
    template <int NTv, int MODEv, int WPEv>
    void launch_ipm_with()
    {
        int bytes = shared_h_.chain_lds_doubles * (int)sizeof(double);
        if (const char* e = debug_token("batch_per_cu")) {
            // measurement aid: at most <n> workgroups per compute unit (the launch asks for as much LDS as 1 / n of a unit's 160 KB: fewer resident instances, a smaller
            // footprint in the caches)
            const int n = std::max(1, std::atoi(e));
            bytes = std::max(bytes, std::min(LDS_LIMIT_BYTES, (160 * 1024) / n - 2560 - 512));
        }
        static PerDeviceOnce attr;
        attr([&] {
            PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_batch_ipm<NTv, MODEv, WPEv>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT_BYTES));
        });
        hipLaunchKernelGGL((k_batch_ipm<NTv, MODEv, WPEv>), dim3(batch_), dim3(NTv), bytes, st_, shared_.p, arena_.p, ruiz_c_.p, infos_.p, prof_.p, order_.p);
    }
    template <int NTv, int MODEv>
    void launch_ipm_as()
    {
        if constexpr (NTv == 64) {
            if (wpe_ == 2) launch_ipm_with<NTv, MODEv, 2>();
            else if (wpe_ == 3) launch_ipm_with<NTv, MODEv, 3>();
            else if (wpe_ == 5 && MODEv == MODE_WAVE) launch_ipm_with<NTv, MODE_WAVE, 5>();
            else if (wpe_ == 6 && MODEv == MODE_WAVE) launch_ipm_with<NTv, MODE_WAVE, 6>();
            else launch_ipm_with<NTv, MODEv, 4>();
        } else {
            launch_ipm_with<NTv, MODEv, 2>();
        }
    }
    void launch_ipm()
    {
        if (nt_ == 64) {
            if (mode_ == MODE_WAVE) launch_ipm_as<64, MODE_WAVE>();
            else if (mode_ == MODE_RESIDENT) launch_ipm_as<64, MODE_RESIDENT>();
            else if (mode_ == MODE_STAGED) launch_ipm_as<64, MODE_STAGED>();
            else launch_ipm_as<64, MODE_HBM>();
        } else {
            if (mode_ == MODE_RESIDENT) launch_ipm_as<256, MODE_RESIDENT>();
            else if (mode_ == MODE_STAGED) launch_ipm_as<256, MODE_STAGED>();
            else launch_ipm_as<256, MODE_HBM>();
        }
    }

    int dev_, batch_ = 0, n_ = 0, p_ = 0, m_ = 0, nt_ = 64;
    int mode_ = MODE_STAGED, wpe_ = 4, forced_mode_ = -1;
    bool setup_done_ = false;
    std::vector<char> fin_hl_, fin_hu_, fin_xl_, fin_xu_;  // which of the caller's bounds are finite (shared by all instances)
    int nzP_in_ = 0, nzA_in_ = 0, nzG_in_ = 0;             // the caller's nonzero counts (P may carry its lower triangle)
    const int *mapP_ = nullptr, *mapA_ = nullptr, *mapG_ = nullptr, *disabled_d_ = nullptr, *rzPp_ = nullptr, *rzPi_ = nullptr;  // device, owned by ibufs_
    double last_kernel_ms_ = 0.0;
    bool lpt_ = true;
    hipStream_t st_ = nullptr;
    pq_settings settings_;
    multistage::Symbolic sym_;
    Layout layout_;
    BatchShared shared_h_{};
    DBuf<BatchShared> shared_;
    std::vector<DBuf<int>> ibufs_;
    std::vector<DBuf<long long>> lbufs_;
    DBuf<double> arena_, ruiz_c_, prof_;
    DBuf<pq_info> infos_;
    DBuf<int> order_;            // block -> instance of the next launch
    std::vector<int> order_h_;
    std::vector<pq_info> infos_h_;
};

}  // namespace pq

// ------------------------------------------------------------------ C-ABI (include/piqp_amd.h, "Batched solver" section)
using namespace pq;

struct pq_batch {
    std::unique_ptr<BatchSolver> impl;
};

extern "C" {

int pq_batch_create(pq_batch** out, int device)
{
    if (!out) return fail(PQ_ERR_INVALID, "null argument");
    *out = nullptr;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return fail(PQ_ERR_HIP, "no HIP device visible: this library has no CPU fallback");
    if (device < 0 || device >= cnt) return fail(PQ_ERR_INVALID, "device %d out of range", device);
    return guarded([&] { auto* h = new pq_batch; h->impl.reset(new BatchSolver(device)); *out = h; return (int)PQ_OK; });
}
void pq_batch_destroy(pq_batch* s) { delete s; }
pq_settings* pq_batch_settings(pq_batch* s) { return s ? &s->impl->settings() : nullptr; }
int pq_batch_setup_sparse(pq_batch* s, int batch, int n, int p, int m, const int* Pp, const int* Pi, const double* Px, const double* c, const int* Ap, const int* Ai,
                          const double* Ax, const double* b, const int* Gp, const int* Gi, const double* Gx, const double* h_l, const double* h_u, const double* x_l,
                          const double* x_u)
{
    if (!s || !Pp || !Pi || !Px || !c) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { return s->impl->setup(batch, n, p, m, Pp, Pi, Px, c, Ap, Ai, Ax, b, Gp, Gi, Gx, h_l, h_u, x_l, x_u) ? 1 : 0; });
}
int pq_batch_update(pq_batch* s, const double* c, const double* b, const double* h_l, const double* h_u, const double* x_l, const double* x_u)
{
    if (!s) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { return s->impl->update_vectors(c, b, h_l, h_u, x_l, x_u) ? 1 : 0; });
}
int pq_batch_update_data(pq_batch* s, const double* Px, const double* Ax, const double* Gx, const double* c, const double* b, const double* h_l, const double* h_u,
                         const double* x_l, const double* x_u)
{
    if (!s) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { return s->impl->update_data(Px, Ax, Gx, c, b, h_l, h_u, x_l, x_u) ? 1 : 0; });
}
int pq_batch_solve(pq_batch* s)
{
    if (!s) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { return s->impl->solve(); });
}
const pq_info* pq_batch_info(const pq_batch* s, int instance)
{
    if (!s || instance < 0 || instance >= s->impl->batch()) return nullptr;
    return &s->impl->info(instance);
}
int pq_batch_get_result(pq_batch* s, int field, double* out_host)
{
    if (!s || !out_host || field < 0 || field >= 10) return fail(PQ_ERR_INVALID, "bad argument");
    return guarded([&] { s->impl->get_result(field, out_host); return (int)PQ_OK; });
}
int pq_batch_dims(const pq_batch* s, int* batch, int* n, int* p, int* m)
{
    if (!s) return fail(PQ_ERR_INVALID, "null argument");
    if (batch) *batch = s->impl->batch();
    if (n) *n = s->impl->n();
    if (p) *p = s->impl->p();
    if (m) *m = s->impl->m();
    return PQ_OK;
}
int pq_batch_block_info(const pq_batch* s, int* out_host, int capacity)
{
    if (!s) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] {
        std::vector<int> bi;
        s->impl->block_info(bi);
        const int N = (int)bi.size() / 3;
        if (out_host) for (int i = 0; i < 3 * std::min(N, capacity); ++i) out_host[i] = bi[i];
        return N;
    });
}
int pq_batch_get_profile(pq_batch* s, int instance, double* out8)
{
    if (!s || !out8 || instance < 0 || instance >= s->impl->batch()) return fail(PQ_ERR_INVALID, "bad argument");
    return guarded([&] { s->impl->get_profile(instance, out8); return (int)PQ_OK; });
}
int pq_batch_set_start_order(pq_batch* s, int longest_first)
{
    if (!s) return fail(PQ_ERR_INVALID, "null argument");
    return guarded([&] { s->impl->set_start_order(longest_first != 0); return (int)PQ_OK; });
}
int pq_batch_last_kernel_ms(const pq_batch* s, double* ms, int* threads_per_qp)
{
    if (!s) return fail(PQ_ERR_INVALID, "null argument");
    if (ms) *ms = s->impl->last_kernel_ms();
    if (threads_per_qp) *threads_per_qp = s->impl->threads_per_qp();
    return PQ_OK;
}

}  // extern "C"
