// Microbenchmarks of dependent-instruction latencies on one wave (gfx950): build and run on the GPU box
//   hipcc --offload-arch=gfx950 -O3 tools/ub/chain.hip -o /tmp/ub_chain && /tmp/ub_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double readlane_d(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
#define N 256
// volatile asm statements keep their order; passing the live value through one before and after each clock read pins the loop between the reads
__device__ __forceinline__ long long tick(double& x)
{
    long long t;
    asm volatile("" : "+v"(x));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    asm volatile("" : "+v"(x));
    return t;
}
__device__ __forceinline__ long long tick4(d4& x)
{
    long long t;
    asm volatile("" : "+v"(x));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    asm volatile("" : "+v"(x));
    return t;
}
__global__ void k_fma(double* out, long long* t, double a, double b)
{
    double x = out[threadIdx.x];
    long long w0 = wall_clock64(); long long t0 = tick(x);
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_fma(x, a, b);
    long long t1 = tick(x); long long w1 = wall_clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) { t[0] = t1 - t0; t[1] = w1 - w0; }
}
__global__ void k_mul(double* out, long long* t, double a)
{
    double x = out[threadIdx.x];
    long long t0 = tick(x);
#pragma unroll
    for (int i = 0; i < N; ++i) x = x * a;
    long long t1 = tick(x);
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_rsq(double* out, long long* t)
{
    double x = out[threadIdx.x];
    long long t0 = tick(x);
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_amdgcn_rsq(x);
    long long t1 = tick(x);
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_readlane(double* out, long long* t)
{
    double x = out[threadIdx.x];
    long long t0 = tick(x);
#pragma unroll
    for (int i = 0; i < N; ++i) { const double s = readlane_d(x, (i * 7) & 63); x = __builtin_fma(s, -s, x); }
    long long t1 = tick(x);
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_cndmask(double* out, long long* t)
{
    double x = out[threadIdx.x];
    const bool m = threadIdx.x & 1;
    long long t0 = tick(x);
#pragma unroll
    for (int i = 0; i < N; ++i) { x = m ? x * 1.0000001 : 0.0; }
    long long t1 = tick(x);
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_mfma(double* out, long long* t)
{
    d4 acc = {out[threadIdx.x], 0, 0, 0};
    double a = out[threadIdx.x + 64], b = out[threadIdx.x + 128];
    long long t0 = tick4(acc);
#pragma unroll
    for (int i = 0; i < N; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    long long t1 = tick4(acc);
    out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0) t[0] = t1 - t0;
}
// dependent MFMA whose operand comes from the previous result through VALU (the factorisation's shape)
__global__ void k_mfma_valu(double* out, long long* t)
{
    d4 acc = {out[threadIdx.x], 1, 2, 3};
    long long t0 = tick4(acc);
#pragma unroll
    for (int i = 0; i < N; ++i) { const double a = acc[i & 3] * 0.5; acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, -a, acc, 0, 0, 0); }
    long long t1 = tick4(acc);
    out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0) t[0] = t1 - t0;
}
// the pivot recurrence of factor16_tile without the matrix core: fma -> rsq -> 2 Newton steps -> scale -> readlane
__global__ void k_pivot(double* out, long long* t)
{
    double col = out[threadIdx.x] + 2.0, dk = 3.0;
    long long t0 = tick(col);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double y = __builtin_amdgcn_rsq(dk);
        y = y * (1.5 - 0.5 * dk * y * y);
        y = y * (1.5 - 0.5 * dk * y * y);
        const double l = col * y;
        const double l1 = readlane_d(l, (i + 1) & 63), t11 = readlane_d(col, (i + 17) & 63);
        dk = __builtin_fma(l1, -l1, t11) + 4.0;
        col = l + 1.0;
    }
    long long t1 = tick(col);
    out[threadIdx.x] = col + dk;
    if (threadIdx.x == 0) t[0] = t1 - t0;
}
// same, with a second wave of the workgroup polling LDS with s_sleep like the follower waves do
__global__ void k_pivot_shared(double* out, long long* t, int sleep_arg)
{
    __shared__ volatile int prog;
    if (threadIdx.x == 0) prog = 0;
    __syncthreads();
    if (threadIdx.x >= 64) {
        while (prog < 1) { if (sleep_arg) __builtin_amdgcn_s_sleep(2); }
        return;
    }
    double col = out[threadIdx.x] + 2.0, dk = 3.0;
    long long t0 = tick(col);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double y = __builtin_amdgcn_rsq(dk);
        y = y * (1.5 - 0.5 * dk * y * y);
        y = y * (1.5 - 0.5 * dk * y * y);
        const double l = col * y;
        const double l1 = readlane_d(l, (i + 1) & 63), t11 = readlane_d(col, (i + 17) & 63);
        dk = __builtin_fma(l1, -l1, t11) + 4.0;
        col = l + 1.0;
    }
    long long t1 = tick(col);
    out[threadIdx.x] = col + dk;
    if (threadIdx.x == 0) { t[0] = t1 - t0; prog = 1; }
}
// ---- round 5: an ordered chain of 64 subtractions s -= p[l] (the substitution's backward pass) with the terms in LDS (every lane reads the same words: 16-byte or
// 8-byte reads) or in the lanes of a register (two lane reads per term)
typedef double d2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k_chain64(double* out, long long* t)
{
    __shared__ double p[64 * 4];
    const int lane = threadIdx.x;
    for (int r = 0; r < 4; ++r) p[lane + 64 * r] = 1e-9 * (lane + 1 + r);
    __syncthreads();
    double s = out[lane];
    const double pr = 1e-9 * (lane + 1);
    long long t0 = tick(s);
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
        const double* q = p + 64 * rep;
        if (MODE == 0) {
            d2 tt[4], nn[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) tt[k] = *reinterpret_cast<const d2*>(q + 2 * k);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                if (g < 7) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) nn[k] = *reinterpret_cast<const d2*>(q + 8 * (g + 1) + 2 * k);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) { s = s - tt[k].x; s = s - tt[k].y; }
#pragma unroll
                for (int k = 0; k < 4; ++k) tt[k] = nn[k];
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int l = 0; l < 64; ++l) s = s - q[l];
        } else {
#pragma unroll
            for (int l = 0; l < 64; ++l) s = s - readlane_d(pr, l);
        }
    }
    long long t1 = tick(s);
    out[lane] = s;
    if (lane == 0) t[0] = t1 - t0;
}
int main()
{
    double* out; long long* t;
    hipMalloc(&out, 4096 * 8); hipMalloc(&t, 64);
    hipMemset(out, 0, 4096 * 8);
    long long h[2];
    auto rep = [&](const char* name, int per) { hipDeviceSynchronize(); hipMemcpy(h, t, 16, hipMemcpyDeviceToHost); printf("%-28s %8.1f cycles/iter (%d dependent ops)\n", name, (double)h[0] / N, per); };
    for (int r = 0; r < 2; ++r) {
        hipLaunchKernelGGL(k_fma, dim3(1), dim3(64), 0, 0, out, t, 1.0000001, 1e-9); rep("v_fma_f64", 1);
        printf("   shader clock / 100 MHz clock = %.2f -> %.0f MHz\n", (double)h[0] / h[1], 100.0 * h[0] / h[1]);
        hipLaunchKernelGGL(k_mul, dim3(1), dim3(64), 0, 0, out, t, 1.0000001); rep("v_mul_f64", 1);
        hipLaunchKernelGGL(k_rsq, dim3(1), dim3(64), 0, 0, out, t); rep("v_rsq_f64", 1);
        hipLaunchKernelGGL(k_readlane, dim3(1), dim3(64), 0, 0, out, t); rep("readlane x2 + fma", 3);
        hipLaunchKernelGGL(k_cndmask, dim3(1), dim3(64), 0, 0, out, t); rep("mul + cndmask x2", 3);
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, out, t); rep("mfma f64 16x16x4 (acc chain)", 1);
        hipLaunchKernelGGL(k_mfma_valu, dim3(1), dim3(64), 0, 0, out, t); rep("mul -> mfma -> mul ...", 2);
        hipLaunchKernelGGL(k_pivot, dim3(1), dim3(64), 0, 0, out, t); rep("pivot recurrence (no mfma)", 12);
        hipLaunchKernelGGL(k_pivot_shared, dim3(1), dim3(128), 0, 0, out, t, 1); rep("pivot + sleeping poller", 12);
        hipLaunchKernelGGL(k_pivot_shared, dim3(1), dim3(128), 0, 0, out, t, 0); rep("pivot + busy poller", 12);
        hipLaunchKernelGGL(k_pivot_shared, dim3(1), dim3(320), 0, 0, out, t, 1); rep("pivot + 4 sleeping pollers", 12);
        hipLaunchKernelGGL(k_chain64<0>, dim3(1), dim3(64), 0, 0, out, t); rep("chain: 16-byte LDS reads ahead", 1);
        hipLaunchKernelGGL(k_chain64<1>, dim3(1), dim3(64), 0, 0, out, t); rep("chain: 8-byte LDS reads", 1);
        hipLaunchKernelGGL(k_chain64<2>, dim3(1), dim3(64), 0, 0, out, t); rep("chain: two lane reads per term", 1);
    }
    return 0;
}
