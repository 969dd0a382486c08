// Round 6 micro-benchmark: the 16 x 16 diagonal piece factored in L D L^T arithmetic (short pivot chain: v_rcp_f64 + one cubic correction,
// no reciprocal square root on the chain) and the look-ahead follower of the piece below it (Y = T U^-T by rank-1 matrix-core updates whose next
// column comes from a vector FMA, own diagonal tile updated behind every column).  Also: what a v_mfma_f64_16x16x4 costs the vector ALU.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=on tools/ub/ldl16.hip -o tools/ub/ub_ldl16 && tools/ub/ub_ldl16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) volatile double lds_vdouble;
typedef __attribute__((address_space(3))) volatile int lds_vint;
typedef __attribute__((address_space(3))) const double lds_cdouble;
__host__ __device__ constexpr int pi16(int x) { return (x >> 2) + 4 * (x & 3); }
__device__ __forceinline__ double readlane_d(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double writelane_d(double v, int l, double old)  // v uniform (it comes from a lane read): lane l of old <- v
{
    int lo = __double2loint(old), hi = __double2hiint(old);
    const int slo = __builtin_amdgcn_readfirstlane(__double2loint(v)), shi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    asm("v_writelane_b32 %0, %1, %2" : "+v"(lo) : "s"(slo), "i"(l));
    asm("v_writelane_b32 %0, %1, %2" : "+v"(hi) : "s"(shi), "i"(l));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ long long tick()
{
    long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}
// 1 / d: hardware reciprocal + one cubic correction  y0 (1 + e + e^2), e = 1 - d y0
__device__ __forceinline__ double rcp_cubic(double d)
{
    const double y0 = __builtin_amdgcn_rcp(d);
    const double e = __builtin_fma(-d, y0, 1.0);
    const double p = __builtin_fma(e, e, e);
    return __builtin_fma(y0, p, y0);
}

// ---------------- primitives ----------------
template <int MODE>
__global__ void k_prim(double* out, long long* ts, int n)
{
    const int lane = threadIdx.x;
    d4 a0 = {1.0 + lane, 2.0, 3.0, 4.0}, a1 = a0 * 2.0, a2 = a0 * 3.0, a3 = a0 * 4.0;
    double x = 1.0 + 1e-3 * lane, y = 0.5, f0 = 1.0, f1 = 1.1, f2 = 1.2, f3 = 1.3, f4 = 1.4, f5 = 1.5, f6 = 1.6, f7 = 1.7;
    float g0 = 1.0f, g1 = 1.1f, g2 = 1.2f, g3 = 1.3f, g4 = 1.4f, g5 = 1.5f, g6 = 1.6f, g7 = 1.7f;
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(x), "+v"(y));
    const long long t0 = tick();
    for (int it = 0; it < n; ++it) {
        if (MODE == 0) {  // four independent accumulators
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
        } else if (MODE == 1) {  // one accumulator (acc -> acc)
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        } else if (MODE == 2) {  // acc -> B operand
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, a0[0], a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, a0[1], a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, a0[2], a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, a0[3], a0, 0, 0, 0);
        } else if (MODE == 3 || MODE == 4 || MODE == 5) {  // 4 x (one matrix-core op + 8 independent f64 FMAs (3) / 8 f32 FMAs (4) / nothing but the FMAs (5))
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (MODE != 5) a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
                if (MODE == 3 || MODE == 5) {
                    f0 = __builtin_fma(f0, x, y); f1 = __builtin_fma(f1, x, y); f2 = __builtin_fma(f2, x, y); f3 = __builtin_fma(f3, x, y);
                    f4 = __builtin_fma(f4, x, y); f5 = __builtin_fma(f5, x, y); f6 = __builtin_fma(f6, x, y); f7 = __builtin_fma(f7, x, y);
                    asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));
                } else {
                    g0 = __builtin_fmaf(g0, 1.5f, 0.5f); g1 = __builtin_fmaf(g1, 1.5f, 0.5f); g2 = __builtin_fmaf(g2, 1.5f, 0.5f); g3 = __builtin_fmaf(g3, 1.5f, 0.5f);
                    g4 = __builtin_fmaf(g4, 1.5f, 0.5f); g5 = __builtin_fmaf(g5, 1.5f, 0.5f); g6 = __builtin_fmaf(g6, 1.5f, 0.5f); g7 = __builtin_fmaf(g7, 1.5f, 0.5f);
                    asm volatile("" : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3), "+v"(g4), "+v"(g5), "+v"(g6), "+v"(g7));
                }
            }
        } else if (MODE == 6) {  // dependent f64 FMA chain
            f0 = __builtin_fma(f0, x, y); f0 = __builtin_fma(f0, x, y); f0 = __builtin_fma(f0, x, y); f0 = __builtin_fma(f0, x, y);
        } else if (MODE == 7) {  // dependent v_rcp_f64
            f0 = __builtin_amdgcn_rcp(f0); f0 = __builtin_amdgcn_rcp(f0); f0 = __builtin_amdgcn_rcp(f0); f0 = __builtin_amdgcn_rcp(f0);
        } else if (MODE == 8) {  // dependent v_rsq_f64
            f0 = __builtin_amdgcn_rsq(f0); f0 = __builtin_amdgcn_rsq(f0); f0 = __builtin_amdgcn_rsq(f0); f0 = __builtin_amdgcn_rsq(f0);
        } else if (MODE == 9) {  // lane read -> vector op -> lane read
#pragma unroll
            for (int u = 0; u < 4; ++u) { const double s = readlane_d(f0, 5 + u); f0 = __builtin_fma(f0, s, y); }
        } else if (MODE == 10) {  // matrix-core op -> vector read of its result -> operand of the next (the crossing of factor16)
#pragma unroll
            for (int u = 0; u < 4; ++u) { a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0); x = a0[0] * 0.5; }
        } else if (MODE == 11) {  // 4 x (matrix-core op + a DEPENDENT chain of 6 f64 FMAs): do they overlap?
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
                f0 = __builtin_fma(f0, x, y); f0 = __builtin_fma(f0, x, y); f0 = __builtin_fma(f0, x, y); f0 = __builtin_fma(f0, x, y); f0 = __builtin_fma(f0, x, y); f0 = __builtin_fma(f0, x, y);
                asm volatile("" : "+v"(f0));
            }
        } else if (MODE == 12) {  // the same chain alone
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                f0 = __builtin_fma(f0, x, y); f0 = __builtin_fma(f0, x, y); f0 = __builtin_fma(f0, x, y); f0 = __builtin_fma(f0, x, y); f0 = __builtin_fma(f0, x, y); f0 = __builtin_fma(f0, x, y);
                asm volatile("" : "+v"(f0));
            }
        } else if (MODE == 13) {  // v_mfma_f64_4x4x4 (four blocks), independent
            f0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, f0, 0, 0, 0); f1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, f1, 0, 0, 0);
            f2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, f2, 0, 0, 0); f3 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, f3, 0, 0, 0);
        }
        asm volatile("" : "+v"(a0), "+v"(f0));
    }
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(f0));
    const long long t1 = tick();
    out[lane] = a0[0] + a1[1] + a2[2] + a3[3] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + g0 + g1 + g2 + g3 + g4 + g5 + g6 + g7 + x;
    if (lane == 0) ts[0] = t1 - t0;
}

// ---------------- the factorisation of the piece ----------------
// Layout PP of a 16 x 16 tile in a wave: lane (ip = lane & 15, g = lane >> 4), register r  <->  logical element [pi(ip)][4 g + r].  Raw LDS image: [r * 64 + lane].
// On entry t = the tile (symmetric, lower part used).  On return t = -U (U unit lower triangular, strictly lower part; zeros elsewhere), A = U D U^T.
// Published after every fourth column: raw image of -U into Upub, reciprocal pivots rpub[c], pivots rpub[64 + c], then *prog = base + c + 1.
template <int TRICK>
__device__ __forceinline__ void factor16_ldl(d4& t, int lane, lds_vdouble* Upub, lds_vdouble* rpub, lds_vint* prog, int base)
{
    const int ip = lane & 15, g = lane >> 4, il = pi16(ip);
    d4 s = {-t[0], -t[1], -t[2], -t[3]};
    d4 Un = {0.0, 0.0, 0.0, 0.0};
    double col = s[0];
    double dv = 1.0;  // lane c <- -d_c
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int gc = c >> 2, rc = c & 3;
        const int c1 = (c + 1) & 15, rn = c1 & 3;
        const int lc = 16 * gc + pi16(c), lc1 = 16 * gc + pi16(c1);
        const double dneg = readlane_d(col, lc);
        const double y0 = __builtin_amdgcn_rcp(-dneg);
        const double e = __builtin_fma(dneg, y0, 1.0);
        const double p = __builtin_fma(e, e, e);
        const double rinv = __builtin_fma(y0, p, y0);
        dv = writelane_d(dneg, c, dv);
        const bool below = (g == gc) && (il > c);
        const double un = below ? col * rinv : 0.0;
        Un[rc] += un;
        if (c < 15) {
            const double a1n = readlane_d(col, lc1);
            double u1n;
            if (TRICK) { const double m = a1n * y0; u1n = __builtin_fma(m, p, m); }
            else u1n = a1n * rinv;
            const double cold = col;
            if (rn != 0) {
                col = __builtin_fma(cold, u1n, s[rn]);
                s = __builtin_amdgcn_mfma_f64_16x16x4f64(un, cold, s, 0, 0, 0);
            } else {
                s = __builtin_amdgcn_mfma_f64_16x16x4f64(un, cold, s, 0, 0, 0);
                col = s[0];
            }
        }
        if (rc == 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) Upub[q * 64 + lane] = Un[q];
            rpub[lane] = rcp_cubic(-dv);
            rpub[64 + lane] = -dv;
            asm volatile("" ::: "memory");
            if (lane == 0) *prog = base + c + 1;
            asm volatile("" ::: "memory");
        }
    }
    t = Un;
}
// The piece below: acc = T (layout PP) becomes Y = T U^-T four columns behind the factorisation; dn = MINUS the follower's own diagonal tile (layout PP) receives
// y_c (y_c / d_c)^T behind every column (HOT).  Zt: 256 zero doubles, Zr: 4 zero doubles (the lanes outside the current column group read their operands there).
template <bool HOT>
__device__ __forceinline__ void follow16(d4& acc, d4& dn, int lane, const lds_vdouble* Upub, const lds_vdouble* rpub, const lds_vint* prog, int base, const lds_vdouble* Zt, const lds_vdouble* Zr)
{
    const int g = lane >> 4;
#pragma unroll
    for (int G = 0; G < 4; ++G) {
        while (*prog < base + 4 * G + 4) __builtin_amdgcn_s_sleep(1);
        const lds_vdouble* up = (g == G) ? Upub + lane : Zt + lane;
        const d4 un = {up[0], up[64], up[128], up[192]};
        d4 rm = {0.0, 0.0, 0.0, 0.0};
        if (HOT) { const lds_vdouble* rp = (g == G) ? rpub + 4 * G : Zr; rm = (d4){rp[0], rp[1], rp[2], rp[3]}; }
        double ycol = acc[0];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = 4 * G + q;
            double ynext = 0.0;
            if (q < 3) {
                const double u1n = readlane_d(un[q], 16 * G + pi16(c + 1));
                ynext = __builtin_fma(ycol, u1n, acc[q + 1]);
            }
            if (c < 15) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(un[q], ycol, acc, 0, 0, 0);
            if (HOT) dn = __builtin_amdgcn_mfma_f64_16x16x4f64(ycol * rm[q], ycol, dn, 0, 0, 0);
            ycol = ynext;
        }
    }
}

// the form the in-kernel attempt used (tools/ub/potrf_block_ldl_attempt.hip.txt): reciprocals kept in registers and written by lane 0 every fourth column, pivots
// read off the accumulator's diagonal at the end, no per-pivot bookkeeping.  VARIANT bit 0: s_setprio 3 around it; bit 1: column masks precomputed (ballots before the loop)
template <int VARIANT>
__device__ __forceinline__ void factor16_k(d4& sn, double& ddiag, int lane, lds_vdouble* Upub, lds_vdouble* rpub, lds_vint* prog, int base)
{
    const int g = lane >> 4, il = pi16(lane & 15);
    d4 s = sn;
    d4 Un = {0.0, 0.0, 0.0, 0.0};
    double col = s[0];
    double rq[4];
    if (VARIANT & 1) __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int gc = c >> 2, rc = c & 3;
        const int c1 = (c + 1) & 15, rn = c1 & 3;
        const double dneg = readlane_d(col, 16 * gc + pi16(c));
        const double y0 = __builtin_amdgcn_rcp(-dneg);
        const double e = __builtin_fma(dneg, y0, 1.0);
        const double p = __builtin_fma(e, e, e);
        const double rinv = __builtin_fma(y0, p, y0);
        rq[rc] = rinv;
        const double un = (g == gc && il > c) ? col * rinv : 0.0;
        Un[rc] += un;
        if (c < 15) {
            const double u1n = readlane_d(col, 16 * gc + pi16(c1)) * rinv;
            const double cold = col;
            if (rn != 0) {
                col = __builtin_fma(cold, u1n, s[rn]);
                s = __builtin_amdgcn_mfma_f64_16x16x4f64(un, cold, s, 0, 0, 0);
            } else {
                s = __builtin_amdgcn_mfma_f64_16x16x4f64(un, cold, s, 0, 0, 0);
                col = s[0];
                asm volatile("" : "+v"(col));
            }
        }
        if (rc == 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) Upub[q * 64 + lane] = Un[q];
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) rpub[4 * gc + q] = rq[q];
            }
            asm volatile("" ::: "memory");
            if (lane == 0) *prog = base + c + 1;
            asm volatile("" ::: "memory");
        }
    }
    if (VARIANT & 1) __builtin_amdgcn_s_setprio(0);
    {
        const int r = il & 3;
        ddiag = r == 0 ? s[0] : (r == 1 ? s[1] : (r == 2 ? s[2] : s[3]));
    }
    sn = Un;
}
// the in-kernel conditions one by one: wave 0 factors; waves 1 .. are OTHERS bit 0: the hot follower (wave 1), bit 1: plain followers (waves 2, 3), bit 2: waves that
// only poll the progress word with s_sleep (waves 4 ..), like the waves of a block that wait for a later step
template <int VARIANT, int OTHERS>
__global__ void k_panel_k(const double* __restrict__ A, int lda, double* out, long long* ts)
{
    __shared__ double Upub[256], rpub[128], Zt[256], Zr[4];
    __shared__ int prog;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, il = pi16(lane & 15);
    for (int e = tid; e < 256; e += blockDim.x) Zt[e] = 0.0;
    if (tid < 4) Zr[tid] = 0.0;
    if (tid == 0) prog = 0;
    __syncthreads();
    if (wave == 0) {
        d4 t;
        for (int r = 0; r < 4; ++r) t[r] = -A[il + (size_t)(4 * g + r) * lda];
        double dd;
        asm volatile("" : "+v"(t));
        const long long t0 = tick();
        factor16_k<VARIANT>(t, dd, lane, (lds_vdouble*)Upub, (lds_vdouble*)rpub, (lds_vint*)&prog, 0);
        asm volatile("" : "+v"(t), "+v"(dd));
        const long long t1 = tick();
        if (lane == 0) { ts[0] = t0; ts[1] = t1; }
        for (int r = 0; r < 4; ++r) out[r * 64 + lane] = t[r] + dd;
    } else if ((wave == 1 && (OTHERS & 1)) || ((wave == 2 || wave == 3) && (OTHERS & 2))) {
        d4 acc, dn;
        for (int r = 0; r < 4; ++r) { acc[r] = A[16 + il + (size_t)(4 * g + r) * lda]; dn[r] = -A[16 + il + (size_t)(16 + 4 * g + r) * lda]; }
        asm volatile("" : "+v"(acc), "+v"(dn));
        if (wave == 1) follow16<true>(acc, dn, lane, (lds_vdouble*)Upub, (lds_vdouble*)rpub, (lds_vint*)&prog, 0, (lds_vdouble*)Zt, (lds_vdouble*)Zr);
        else follow16<false>(acc, dn, lane, (lds_vdouble*)Upub, (lds_vdouble*)rpub, (lds_vint*)&prog, 0, (lds_vdouble*)Zt, (lds_vdouble*)Zr);
        asm volatile("" : "+v"(acc), "+v"(dn));
        const long long t2 = tick();
        if (wave == 1 && lane == 0) ts[2] = t2;
        out[1024 + wave * 64 + lane] = acc[0] + acc[1] + acc[2] + acc[3] + dn[0];
    } else if (wave >= 4 && (OTHERS & 4)) {
        lds_vint* pr = (lds_vint*)&prog;
        while (*pr < 16) __builtin_amdgcn_s_sleep(8);
        out[2048 + tid] = 1.0;
    }
}

// one workgroup: wave 0 factors A00, wave 1 follows with A10 and its diagonal tile A11 (HOT), waves 2.. follow with copies of A10 (not hot: the load of the other SIMDs)
template <int TRICK>
__global__ void k_panel(const double* __restrict__ A, int lda, double* out, long long* ts, int nrep)
{
    __shared__ double Upub[256], rpub[128], Zt[256], Zr[4];
    __shared__ int prog;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ip = lane & 15, g = lane >> 4, il = pi16(ip);
    for (int e = tid; e < 256; e += blockDim.x) Zt[e] = 0.0;
    if (tid < 4) Zr[tid] = 0.0;
    if (tid == 0) prog = 0;
    __syncthreads();
    for (int rep = 0; rep < nrep; ++rep) {
        const int base = 16 * rep;
        if (wave == 0) {
            d4 t;
            for (int r = 0; r < 4; ++r) t[r] = A[il + (size_t)(4 * g + r) * lda];
            asm volatile("" : "+v"(t));
            const long long t0 = tick();
            factor16_ldl<TRICK>(t, lane, (lds_vdouble*)Upub, (lds_vdouble*)rpub, (lds_vint*)&prog, base);
            asm volatile("" : "+v"(t));
            const long long t1 = tick();
            if (lane == 0) { ts[0] = t0; ts[1] = t1; }
            for (int r = 0; r < 4; ++r) out[r * 64 + lane] = t[r];  // -U raw
        } else {
            d4 acc, dn;
            for (int r = 0; r < 4; ++r) { acc[r] = A[16 + il + (size_t)(4 * g + r) * lda]; dn[r] = -A[16 + il + (size_t)(16 + 4 * g + r) * lda]; }
            asm volatile("" : "+v"(acc), "+v"(dn));
            if (wave == 1) follow16<true>(acc, dn, lane, (lds_vdouble*)Upub, (lds_vdouble*)rpub, (lds_vint*)&prog, base, (lds_vdouble*)Zt, (lds_vdouble*)Zr);
            else follow16<false>(acc, dn, lane, (lds_vdouble*)Upub, (lds_vdouble*)rpub, (lds_vint*)&prog, base, (lds_vdouble*)Zt, (lds_vdouble*)Zr);
            asm volatile("" : "+v"(acc), "+v"(dn));
            const long long t2 = tick();
            if (wave == 1) {
                if (lane == 0) ts[2] = t2;
                for (int r = 0; r < 4; ++r) { out[256 + r * 64 + lane] = acc[r]; out[512 + r * 64 + lane] = dn[r]; }
            } else out[1024 + wave * 64 + lane] = acc[0] + acc[1] + acc[2] + acc[3];
        }
        __syncthreads();
        if (wave == 0 && lane < 16) { out[768 + lane] = rpub[lane]; out[784 + lane] = rpub[64 + lane]; }
        __syncthreads();
    }
}

int main()
{
    double* out; long long* t;
    (void)hipMalloc(&out, 8192 * 8); (void)hipMalloc(&t, 64);
    long long h[4];
    auto rep = [&](const char* name, int n, int per) {
        (void)hipDeviceSynchronize(); (void)hipMemcpy(h, t, 8, hipMemcpyDeviceToHost);
        printf("%-72s %8lld ticks / %d = %7.1f\n", name, h[0], n * per, (double)h[0] / (n * per));
    };
    const int N = 64;
    for (int r = 0; r < 2; ++r) {
        hipLaunchKernelGGL(k_prim<0>, dim3(1), dim3(64), 0, 0, out, t, N); rep("v_mfma_f64_16x16x4, four independent accumulators", N, 4);
        hipLaunchKernelGGL(k_prim<1>, dim3(1), dim3(64), 0, 0, out, t, N); rep("v_mfma_f64_16x16x4, one accumulator", N, 4);
        hipLaunchKernelGGL(k_prim<2>, dim3(1), dim3(64), 0, 0, out, t, N); rep("v_mfma_f64_16x16x4, result -> B operand of the next", N, 4);
        hipLaunchKernelGGL(k_prim<3>, dim3(1), dim3(64), 0, 0, out, t, N); rep("one matrix-core op + 8 independent f64 FMAs", N, 4);
        hipLaunchKernelGGL(k_prim<5>, dim3(1), dim3(64), 0, 0, out, t, N); rep("8 independent f64 FMAs alone", N, 4);
        hipLaunchKernelGGL(k_prim<4>, dim3(1), dim3(64), 0, 0, out, t, N); rep("one matrix-core op + 8 independent f32 FMAs", N, 4);
        hipLaunchKernelGGL(k_prim<11>, dim3(1), dim3(64), 0, 0, out, t, N); rep("one matrix-core op + a dependent chain of 6 f64 FMAs", N, 4);
        hipLaunchKernelGGL(k_prim<12>, dim3(1), dim3(64), 0, 0, out, t, N); rep("a dependent chain of 6 f64 FMAs alone", N, 4);
        hipLaunchKernelGGL(k_prim<6>, dim3(1), dim3(64), 0, 0, out, t, N); rep("dependent v_fma_f64", N, 4);
        hipLaunchKernelGGL(k_prim<7>, dim3(1), dim3(64), 0, 0, out, t, N); rep("dependent v_rcp_f64", N, 4);
        hipLaunchKernelGGL(k_prim<8>, dim3(1), dim3(64), 0, 0, out, t, N); rep("dependent v_rsq_f64", N, 4);
        hipLaunchKernelGGL(k_prim<9>, dim3(1), dim3(64), 0, 0, out, t, N); rep("lane read (2 x v_readlane) -> v_fma_f64 -> ...", N, 4);
        hipLaunchKernelGGL(k_prim<10>, dim3(1), dim3(64), 0, 0, out, t, N); rep("matrix-core op -> vector multiply of its result -> its operand", N, 4);
        hipLaunchKernelGGL(k_prim<13>, dim3(1), dim3(64), 0, 0, out, t, N); rep("v_mfma_f64_4x4x4 (four blocks), independent", N, 4);
    }
    // ---- the panel: a 32 x 32 symmetric positive definite matrix, its first 16 columns factored ----
    const int n = 32;
    std::vector<double> A(n * n);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0 - 0.5; };
    std::vector<double> B(n * n);
    for (auto& v : B) v = rnd();
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double acc = 0.0;
            for (int k = 0; k < n; ++k) acc += B[i + k * n] * B[j + k * n];
            A[i + j * n] = acc + (i == j ? 0.05 : 0.0);
        }
    // host reference in long double
    std::vector<long double> U(16 * 16, 0.0L), d(16), Y(16 * 16), D2(16 * 16);
    {
        std::vector<long double> W(n * n);
        for (int e = 0; e < n * n; ++e) W[e] = A[e];
        for (int c = 0; c < 16; ++c) {
            d[c] = W[c + c * n];
            for (int i = c + 1; i < n; ++i) {
                const long double u = W[i + c * n] / d[c];
                for (int j = c + 1; j <= i; ++j) W[i + j * n] -= u * W[j + c * n];
                if (i < 16) U[i + c * 16] = u;
            }
        }
        // Y = A10 U^-T = (U D)-part below: W[i + c n] (i >= 16) holds y_c after the updates of columns < c
        for (int c = 0; c < 16; ++c)
            for (int i = 0; i < 16; ++i) Y[i + c * 16] = W[16 + i + c * n];
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) D2[i + j * 16] = W[16 + (i >= j ? i : j) + (16 + (i >= j ? j : i)) * n];
    }
    double* dA;
    (void)hipMalloc(&dA, n * n * 8);
    (void)hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
    std::vector<double> o(8192);
    {
        auto run = [&](const char* name, auto kern, int nw) {
            long long best = 1LL << 60, tail = 0;
            for (int it = 0; it < 5; ++it) {
                hipLaunchKernelGGL(kern, dim3(1), dim3(64 * nw), 0, 0, dA, n, out, t);
                (void)hipDeviceSynchronize();
                (void)hipMemcpy(h, t, 24, hipMemcpyDeviceToHost);
                if (h[1] - h[0] < best) { best = h[1] - h[0]; tail = h[2] - h[1]; }
            }
            printf("%-96s factor16 %5lld ticks (%5.1f per pivot), hot follower's tail %5lld\n", name, best, best / 16.0, tail);
        };
        run("in-kernel form, alone (1 wave)", k_panel_k<0, 0>, 1);
        run("in-kernel form + s_setprio 3, alone", k_panel_k<1, 0>, 1);
        run("in-kernel form, + hot follower on the next SIMD", k_panel_k<0, 1>, 2);
        run("in-kernel form, + hot follower + two followers (4 waves, one per SIMD)", k_panel_k<0, 3>, 4);
        run("in-kernel form, + hot + two followers + four pollers (8 waves; wave 4 shares the factoring SIMD)", k_panel_k<0, 7>, 8);
        run("in-kernel form + s_setprio 3, the same 8 waves", k_panel_k<1, 7>, 8);
        run("in-kernel form, 8 waves, only pollers beside it", k_panel_k<0, 4>, 8);
    }
    for (int trick = 0; trick < 2; ++trick)
        for (int nw : {2, 4, 8}) {
            long long best[3] = {1LL << 60, 1LL << 60, 1LL << 60};
            for (int it = 0; it < 5; ++it) {
                if (trick) hipLaunchKernelGGL(k_panel<1>, dim3(1), dim3(64 * nw), 0, 0, dA, n, out, t, 1);
                else hipLaunchKernelGGL(k_panel<0>, dim3(1), dim3(64 * nw), 0, 0, dA, n, out, t, 1);
                (void)hipDeviceSynchronize();
                (void)hipMemcpy(h, t, 24, hipMemcpyDeviceToHost);
                const long long f = h[1] - h[0], tail = h[2] - h[1], tot = h[2] - h[0];
                if (tot < best[2]) { best[0] = f; best[1] = tail; best[2] = tot; }
            }
            (void)hipMemcpy(o.data(), out, 8192 * 8, hipMemcpyDeviceToHost);
            double eu = 0, ey = 0, ed = 0, er = 0, nu = 0, ny = 0, nd = 0;
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r) {
                    const int i = pi16(lane & 15), j = 4 * (lane >> 4) + r;
                    const double un = o[r * 64 + lane], yv = o[256 + r * 64 + lane], dnv = o[512 + r * 64 + lane];
                    const double uref = i > j ? (double)U[i + j * 16] : 0.0;
                    eu = fmax(eu, fabs(-un - uref)); nu = fmax(nu, fabs(uref));
                    ey = fmax(ey, fabs(yv - (double)Y[i + j * 16])); ny = fmax(ny, fabs((double)Y[i + j * 16]));
                    if (i >= j) { ed = fmax(ed, fabs(-dnv - (double)D2[i + j * 16])); nd = fmax(nd, fabs((double)D2[i + j * 16])); }
                }
            for (int c = 0; c < 16; ++c) er = fmax(er, fabs(o[784 + c] - (double)d[c]) / fabs((double)d[c]) + fabs(o[768 + c] * (double)d[c] - 1.0));
            printf("panel, trick %d, %d waves: factor16 %5lld ticks (%5.1f per pivot), follower tail %5lld, total %5lld | rel err U %.1e Y %.1e D' %.1e pivots %.1e\n", trick, nw, best[0], best[0] / 16.0,
                   best[1], best[2], eu / nu, ey / ny, ed / nd, er);
        }
    return 0;
}
