// Accuracy of v_rsq_f64 / v_rcp_f64 and of one / two Newton steps on top (the pivot recurrence of factor16_tile, csrc/dense_kernels.hip):
//   hipcc --offload-arch=gfx950 -O3 tools/ub/rsq_accuracy.hip -o /tmp/ub_rsq && /tmp/ub_rsq
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
__global__ void k(int n, const double* d, double* out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = d[i];
    double y = __builtin_amdgcn_rsq(x);
    out[i] = y;
    y = y * (1.5 - 0.5 * x * y * y);
    out[n + i] = y;
    y = y * (1.5 - 0.5 * x * y * y);
    out[2 * n + i] = y;
    double r = __builtin_amdgcn_rcp(x);
    out[3 * n + i] = r;
    r = r * (2.0 - x * r);
    out[4 * n + i] = r;
    r = r * (2.0 - x * r);
    out[5 * n + i] = r;
}
int main()
{
    const int n = 1 << 20;
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> e(-40.0, 40.0), m(1.0, 2.0);
    std::vector<double> h(n), o(6 * (size_t)n);
    for (auto& v : h) v = std::ldexp(m(g), (int)e(g));
    double *d, *out;
    hipMalloc(&d, n * 8); hipMalloc(&out, 6 * (size_t)n * 8);
    hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, n, d, out);
    hipMemcpy(o.data(), out, 6 * (size_t)n * 8, hipMemcpyDeviceToHost);
    const char* nm[6] = {"v_rsq_f64", "rsq + 1 Newton", "rsq + 2 Newton", "v_rcp_f64", "rcp + 1 Newton", "rcp + 2 Newton"};
    for (int q = 0; q < 6; ++q) {
        long double worst = 0;
        for (int i = 0; i < n; ++i) {
            const long double ex = q < 3 ? 1.0L / sqrtl((long double)h[i]) : 1.0L / (long double)h[i];
            const long double rel = fabsl(((long double)o[(size_t)q * n + i] - ex) / ex);
            if (rel > worst) worst = rel;
        }
        std::printf("%-16s max relative error %.3Le = %.2Lf ulp (2^-53)\n", nm[q], worst, worst / 1.1102230246251565e-16L);
    }
    return 0;
}
