// Isolated timing of the 16 x 16 in-register factorisation step of potrf_block (csrc/dense_kernels.hip: factor16_tile), one wave:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=on tools/ub/factor16.hip -o /tmp/ub_f16 && /tmp/ub_f16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) volatile double lds_vdouble;
typedef __attribute__((address_space(3))) volatile int lds_vint;
__device__ __forceinline__ double readlane_d(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rsqrt_newton(double d)
{
    double y = __builtin_amdgcn_rsq(d);
    y = y * (1.5 - 0.5 * d * y * y);
    y = y * (1.5 - 0.5 * d * y * y);
    return y;
}
__device__ __forceinline__ long long tick4(d4& x)
{
    long long t;
    asm volatile("" : "+v"(x));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    asm volatile("" : "+v"(x));
    return t;
}
// MODE bit 0: matrix-core rank-1 update; bit 1: publication every 4 columns; bit 2: rd / dd bookkeeping
template <int MODE>
__device__ __forceinline__ void factor16(d4& t, int lane, double& rd, double& dd, lds_vdouble* Dpub, lds_vdouble* rpub, lds_vint* prog, int base)
{
    const int i = lane & 15, g = lane >> 4;
    double dk = readlane_d(t[0], 0);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int g0 = c & 3, r0 = c >> 2;
        const bool incol = (g == g0), below = incol && (i > c);
        const int c1 = c + 1, g1 = c1 & 3, r1 = (c1 >> 2) & 3;
        double dk_next = 0.0;
        const double r = rsqrt_newton(dk);
        const double lcol = below ? t[r0] * r : 0.0;
        t[r0] = incol ? ((i == c) ? dk * r : lcol) : t[r0];
        if (c < 15) {
            const double l1 = readlane_d(lcol, 16 * g0 + c1), t11 = readlane_d(t[r1], 16 * g1 + c1);
            dk_next = __builtin_fma(l1, -l1, t11);
            if (MODE & 1) t = __builtin_amdgcn_mfma_f64_16x16x4f64(lcol, -lcol, t, 0, 0, 0);
        }
        if (MODE & 4) { if (lane == c) { rd = r; dd = dk; } }
        if ((MODE & 2) && (c & 3) == 3) {
            Dpub[(g + 4 * r0) * 16 + i] = t[r0];
            if (lane < 16 && (lane >> 2) == r0) rpub[lane] = rd;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) *prog = base + c + 1;
        }
        dk = dk_next;
    }
}
// the form potrf_block uses now: negated tile, factor collected in a second tile, pivots kept in uniform registers
__device__ __forceinline__ void factor16_new(d4& t, int lane, lds_vdouble* Dpub, lds_vdouble* rpub, lds_vint* prog, int base)
{
    const int i = lane & 15, g = lane >> 4;
    d4 s = {-t[0], -t[1], -t[2], -t[3]};
    d4 Lo = {0.0, 0.0, 0.0, 0.0};
    double dk = readlane_d(t[0], 0);
    double rq[4];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int g0 = c & 3, r0 = c >> 2;
        const bool incol = (g == g0);
        const int c1 = c + 1, g1 = c1 & 3, r1 = (c1 >> 2) & 3;
        double dk_next = 0.0;
        const double r = rsqrt_newton(dk);
        const double lneg = (incol && i >= c) ? s[r0] * r : 0.0;
        Lo[r0] -= lneg;
        if (c < 15) {
            const double l1 = readlane_d(lneg, 16 * g0 + c1), s11 = readlane_d(s[r1], 16 * g1 + c1);
            dk_next = __builtin_fma(-l1, l1, -s11);
            s = __builtin_amdgcn_mfma_f64_16x16x4f64(lneg, lneg, s, 0, 0, 0);
        }
        rq[c & 3] = r;
        if ((c & 3) == 3) {
            Dpub[(g + 4 * r0) * 16 + i] = Lo[r0];
            const int q = lane & 3;
            const double rv = q == 0 ? rq[0] : (q == 1 ? rq[1] : (q == 2 ? rq[2] : rq[3]));
            if (lane < 4) rpub[4 * r0 + lane] = rv;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) *prog = base + c + 1;
        }
        dk = dk_next;
    }
    t = Lo;
}
__device__ __forceinline__ constexpr int pi16(int x) { return (x >> 2) + 4 * (x & 3); }
// permuted tile (rows and columns relabelled by the 4 x 4 index transpose): the next pivot column is updated on the vector ALU
template <bool PUBLISH>
__device__ __forceinline__ void factor16_perm(d4& t, int lane, lds_vdouble* Dpub, lds_vdouble* rpub, lds_vint* prog, int base)
{
    const int ip = lane & 15, g = lane >> 4, il = pi16(ip);
    d4 s = {-t[0], -t[1], -t[2], -t[3]};
    d4 Lo = {0.0, 0.0, 0.0, 0.0};
    double col = s[0];
    double dk = -readlane_d(col, 0);
    double rq[4];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int gc = c >> 2, rc = c & 3;
        const bool incol = (g == gc);
        const int c1 = (c + 1) & 15, gn = c1 >> 2, rn = c1 & 3;
        const int src1 = 16 * gc + pi16(c1);
        const double r = rsqrt_newton(dk);
        const double lneg = (incol && il >= c) ? col * r : 0.0;
        Lo[rc] -= lneg;
        rq[rc] = r;
        if (c < 15) {
            const double l1 = readlane_d(lneg, src1);
            if (rn != 0) {
                col = __builtin_fma(lneg, l1, s[rn]);
                s = __builtin_amdgcn_mfma_f64_16x16x4f64(lneg, lneg, s, 0, 0, 0);
            } else {
                s = __builtin_amdgcn_mfma_f64_16x16x4f64(lneg, lneg, s, 0, 0, 0);
                col = s[0];
            }
            dk = -readlane_d(col, 16 * gn + pi16(c1));
        }
        if (PUBLISH && rc == 3) {
            if (incol) {
#pragma unroll
                for (int q = 0; q < 4; ++q) Dpub[(4 * gc + q) * 16 + il] = Lo[q];
            }
            const int q = lane & 3;
            const double rv = q == 0 ? rq[0] : (q == 1 ? rq[1] : (q == 2 ? rq[2] : rq[3]));
            if (lane < 4) rpub[4 * gc + lane] = rv;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) *prog = base + c + 1;
        }
    }
    t = Lo;
}
template <bool PUBLISH>
__global__ void k_f16_perm(double* out, long long* ts)
{
    __shared__ double D[256], R[32];
    __shared__ int prog;
    const int lane = threadIdx.x & 63;
    d4 t;
    for (int r = 0; r < 4; ++r) { const int i = pi16(lane & 15), j = 4 * (lane >> 4) + r; t[r] = i == j ? 20.0 + i : 1.0 / (1 + i + j); }
    long long t0 = tick4(t);
    factor16_perm<PUBLISH>(t, lane, (lds_vdouble*)D, (lds_vdouble*)R, (lds_vint*)&prog, 0);
    long long t1 = tick4(t);
    out[lane] = t[0] + t[1] + t[2] + t[3] + D[lane] + R[lane & 15] + prog;
    if (lane == 0) ts[0] = t1 - t0;
}
__global__ void k_f16_new(double* out, long long* ts, int nfollow)
{
    __shared__ double D[256], R[32];
    __shared__ int prog;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) prog = 0;
    __syncthreads();
    if (threadIdx.x >= 64) {
        // followers like tile_trsm_rt_follow: poll the progress word, then four dependent matrix-core updates per published register
        d4 x = {1.0, 2.0, 3.0, 4.0};
        lds_vint* pr = (lds_vint*)&prog;
        for (int r0 = 0; r0 < 4; ++r0) {
            while (*pr < 4 * r0 + 4) __builtin_amdgcn_s_sleep(2);
            const double l = ((lds_vdouble*)D)[((lane >> 4) + 4 * r0) * 16 + (lane & 15)];
            for (int g0 = 0; g0 < 4; ++g0) x = __builtin_amdgcn_mfma_f64_16x16x4f64(((lane >> 4) == g0) ? l : 0.0, x[r0], x, 0, 0, 0);
        }
        out[threadIdx.x] = x[0] + x[1] + x[2] + x[3];
        return;
    }
    d4 t;
    for (int r = 0; r < 4; ++r) { const int i = lane & 15, j = (lane >> 4) + 4 * r; t[r] = i == j ? 20.0 + i : 1.0 / (1 + i + j); }
    long long t0 = tick4(t);
    factor16_new(t, lane, (lds_vdouble*)D, (lds_vdouble*)R, (lds_vint*)&prog, 0);
    long long t1 = tick4(t);
    out[lane] = t[0] + t[1] + t[2] + t[3] + D[lane] + R[lane & 15] + prog;
    if (lane == 0) ts[0] = t1 - t0;
}
template <int MODE>
__global__ void k_f16(double* out, long long* ts)
{
    __shared__ double D[256], R[16];
    __shared__ int prog;
    const int lane = threadIdx.x;
    // a diagonally dominant tile in tile form: lane (i, g) reg r <-> [i][g + 4 r]
    d4 t;
    for (int r = 0; r < 4; ++r) { const int i = lane & 15, j = (lane >> 4) + 4 * r; t[r] = i == j ? 20.0 + i : 1.0 / (1 + i + j); }
    double rd = 0, dd = 0;
    long long t0 = tick4(t);
    factor16<MODE>(t, lane, rd, dd, (lds_vdouble*)D, (lds_vdouble*)R, (lds_vint*)&prog, 0);
    long long t1 = tick4(t);
    out[lane] = t[0] + t[1] + t[2] + t[3] + rd + dd + D[lane] + R[lane & 15] + prog;
    if (lane == 0) ts[0] = t1 - t0;
}

// ---- round 3: the product's loop (CUR) against the same arithmetic with the pivot recurrence kept UNIFORM in vector registers (UNI): the entry of row c + 1 and
// the diagonal entry of column c + 1 are read off the chain (before 1 / l is known), l_{c+1,c} = entry * (1 / l) and d_{c+1} = -fma(l, l, diag) are computed in
// every lane, so no lane read sits between two reciprocal square roots.  Bitwise the same values (checked below).
template <bool UNI>
__device__ __forceinline__ void factor16_r3(d4& t, int lane, lds_vdouble* Dpub, lds_vdouble* rpub, lds_vint* prog, int base)
{
    const int ip = lane & 15, g = lane >> 4, il = pi16(ip);
    d4 s = {-t[0], -t[1], -t[2], -t[3]};
    d4 Lo = {0.0, 0.0, 0.0, 0.0};
    double col = s[0];
    double dk = -readlane_d(col, 0);
    double rq[4];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int gc = c >> 2, rc = c & 3;
        const bool incol = (g == gc);
        const int c1 = (c + 1) & 15, gn = c1 >> 2, rn = c1 & 3;
        const int src1 = 16 * gc + pi16(c1);
        if (!UNI) {
            const double r = rsqrt_newton(dk);
            const double lneg = (incol && il >= c) ? col * r : 0.0;
            Lo[rc] -= lneg;
            rq[rc] = r;
            if (c < 15) {
                const double l1 = readlane_d(lneg, src1);
                if (rn != 0) {
                    col = __builtin_fma(lneg, l1, s[rn]);
                    s = __builtin_amdgcn_mfma_f64_16x16x4f64(lneg, lneg, s, 0, 0, 0);
                } else {
                    s = __builtin_amdgcn_mfma_f64_16x16x4f64(lneg, lneg, s, 0, 0, 0);
                    col = s[0];
                }
                dk = -readlane_d(col, 16 * gn + pi16(c1));
            }
        } else {
            const double colm = (incol && il >= c) ? col : 0.0;
            double sc1 = 0.0, sd = 0.0;
            if (c < 15) { sc1 = readlane_d(colm, src1); if (rn != 0) sd = readlane_d(s[rn], src1); }
            const double r = rsqrt_newton(dk);
            const double lneg = colm * r;
            Lo[rc] -= lneg;
            rq[rc] = r;
            if (c < 15) {
                const double l1 = sc1 * r;
                if (rn != 0) {
                    dk = -__builtin_fma(l1, l1, sd);
                    col = __builtin_fma(lneg, l1, s[rn]);
                    s = __builtin_amdgcn_mfma_f64_16x16x4f64(lneg, lneg, s, 0, 0, 0);
                } else {
                    s = __builtin_amdgcn_mfma_f64_16x16x4f64(lneg, lneg, s, 0, 0, 0);
                    col = s[0];
                    dk = -readlane_d(col, 16 * gn + pi16(c1));
                }
            }
        }
        if (rc == 3) {
            if (incol) {
#pragma unroll
                for (int q = 0; q < 4; ++q) Dpub[(4 * gc + q) * 16 + il] = Lo[q];
            }
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) rpub[4 * gc + q] = rq[q];
            }
            asm volatile("" ::: "memory");
            if (lane == 0) *prog = base + c + 1;
            asm volatile("" ::: "memory");
        }
    }
    t = Lo;
}
template <bool UNI>
__global__ void k_f16_r3(double* out, long long* ts)
{
    __shared__ double D[256], R[32];
    __shared__ int prog;
    const int lane = threadIdx.x & 63;
    for (int e = lane; e < 256; e += 64) D[e] = 0.0;
    __syncthreads();
    d4 t;
    for (int r = 0; r < 4; ++r) { const int i = pi16(lane & 15), j = 4 * (lane >> 4) + r; t[r] = i == j ? 2.0 + 0.37 * i : 0.9 / (1.3 + i + j) + 1e-3 * ((i * 7 + j * 3) % 11); }
    long long t0 = tick4(t);
    factor16_r3<UNI>(t, lane, (lds_vdouble*)D, (lds_vdouble*)R, (lds_vint*)&prog, 0);
    long long t1 = tick4(t);
    __syncthreads();
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = t[r];
    for (int e = lane; e < 256; e += 64) out[256 + e] = D[e];
    if (lane < 16) out[512 + lane] = R[lane];
    if (lane == 0) ts[0] = t1 - t0;
}
// does the scalar-operand FMA reproduce the matrix core's value of the next pivot element bit for bit?
__global__ void k_check(double* out, int* mism)
{
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    d4 t;
    for (int r = 0; r < 4; ++r) { const int j = g + 4 * r; t[r] = i == j ? 2.0 + 0.37 * i : 0.9 / (1.3 + i + j) + 1e-3 * ((i * 7 + j * 3) % 11); }
    int bad = 0;
    double dk = readlane_d(t[0], 0);
#pragma unroll
    for (int c = 0; c < 15; ++c) {
        const int g0 = c & 3, r0 = c >> 2, c1 = c + 1, g1 = c1 & 3, r1 = c1 >> 2;
        const bool below = (g == g0) && (i > c);
        const double r = rsqrt_newton(dk);
        const double lcol = below ? t[r0] * r : 0.0;
        const double l1 = readlane_d(lcol, 16 * g0 + c1), t11 = readlane_d(t[r1], 16 * g1 + c1);
        const double dk_next = __builtin_fma(l1, -l1, t11);
        t = __builtin_amdgcn_mfma_f64_16x16x4f64(lcol, -lcol, t, 0, 0, 0);
        const double viamfma = readlane_d(t[r1], 16 * g1 + c1);
        if (__double_as_longlong(viamfma) != __double_as_longlong(dk_next)) bad |= 1 << c;
        dk = dk_next;
    }
    out[lane] = t[0];
    if (lane == 0) *mism = bad;
}
int main()
{
    {
        double* o; int* m; int hm = -1;
        (void)hipMalloc(&o, 64 * 8); (void)hipMalloc(&m, 4);
        hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, o, m);
        (void)hipMemcpy(&hm, m, 4, hipMemcpyDeviceToHost);
        printf("scalar FMA vs matrix core on the next pivot element: mismatch mask = 0x%x (0 = bitwise equal at all 15 pivots)\n", hm);
    }
    double* out; long long* t;
    (void)hipMalloc(&out, 4096 * 8); (void)hipMalloc(&t, 64);
    long long h[2];
    auto rep = [&](const char* name) { (void)hipDeviceSynchronize(); (void)hipMemcpy(h, t, 16, hipMemcpyDeviceToHost); printf("%-48s %7lld ticks / 16 pivots = %6.1f per pivot\n", name, h[0], h[0] / 16.0); };
    for (int r = 0; r < 2; ++r) {
        hipLaunchKernelGGL(k_f16<0>, dim3(1), dim3(64), 0, 0, out, t); rep("recurrence + column selects");
        hipLaunchKernelGGL(k_f16<1>, dim3(1), dim3(64), 0, 0, out, t); rep("+ matrix-core rank-1 update");
        hipLaunchKernelGGL(k_f16<5>, dim3(1), dim3(64), 0, 0, out, t); rep("+ rd / dd bookkeeping");
        hipLaunchKernelGGL(k_f16<7>, dim3(1), dim3(64), 0, 0, out, t); rep("+ publication every 4 columns (full routine)");
        hipLaunchKernelGGL(k_f16<3>, dim3(1), dim3(64), 0, 0, out, t); rep("update + publication, no bookkeeping");
        hipLaunchKernelGGL(k_f16_perm<false>, dim3(1), dim3(64), 0, 0, out, t); rep("PERMUTED form, no publication");
        hipLaunchKernelGGL(k_f16_perm<true>, dim3(1), dim3(64), 0, 0, out, t); rep("PERMUTED form, full");
        hipLaunchKernelGGL(k_f16_new, dim3(1), dim3(64), 0, 0, out, t, 0); rep("NEW form, alone");
        hipLaunchKernelGGL(k_f16_new, dim3(1), dim3(256), 0, 0, out, t, 3); rep("NEW form, 3 followers on the other SIMDs");
        hipLaunchKernelGGL(k_f16_new, dim3(1), dim3(512), 0, 0, out, t, 7); rep("NEW form, 7 followers (one shares the SIMD)");
    }
    {
        static double a[528], b[528];
        hipLaunchKernelGGL(k_f16_r3<false>, dim3(1), dim3(64), 0, 0, out, t); rep("round 3: product loop (CUR)");
        (void)hipMemcpy(a, out, sizeof(a), hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k_f16_r3<true>, dim3(1), dim3(64), 0, 0, out, t); rep("round 3: uniform pivot recurrence (UNI)");
        (void)hipMemcpy(b, out, sizeof(b), hipMemcpyDeviceToHost);
        int diff = 0;
        for (int e = 0; e < 528; ++e) diff += memcmp(&a[e], &b[e], 8) != 0;
        printf("CUR vs UNI: %d of 528 output doubles differ (0 = bitwise equal); L[5][3] = %.17g\n", diff, a[256 + 3 * 16 + 5]);
        hipLaunchKernelGGL(k_f16_r3<false>, dim3(1), dim3(64), 0, 0, out, t); rep("round 3: product loop (CUR)");
        hipLaunchKernelGGL(k_f16_r3<true>, dim3(1), dim3(64), 0, 0, out, t); rep("round 3: uniform pivot recurrence (UNI)");
    }
    return 0;
}
