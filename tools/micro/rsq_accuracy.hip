// Accuracy of v_rsq_f64 and of one / two Newton steps behind it (the pivot chain of k_potrf64): max relative error over 2^20 samples.
// hipcc --offload-arch=gfx950 -O2 tools/micro/rsq_accuracy.hip -o /tmp/rsq && /tmp/rsq
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *p, double *e0, double *e1, double *e2, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = p[i];
    double y = __builtin_amdgcn_rsq(x);
    e0[i] = y;
    y = y * (1.5 - 0.5 * x * y * y);
    e1[i] = y;
    y = y * (1.5 - 0.5 * x * y * y);
    e2[i] = y;
}
int main() {
    const int n = 1 << 20;
    std::vector<double> h(n), a(n), b(n), c(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = std::ldexp(1.0 + (double)(s >> 11) / 9007199254740992.0, (int)(s % 41) - 20); }
    double *dp, *d0, *d1, *d2;
    hipMalloc(&dp, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dp, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dp, d0, d1, d2, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0;
    for (int i = 0; i < n; i++) {
        const long double t = 1.0L / sqrtl((long double)h[i]);
        m0 = fmax(m0, (double)fabsl(((long double)a[i] - t) / t)); m1 = fmax(m1, (double)fabsl(((long double)b[i] - t) / t)); m2 = fmax(m2, (double)fabsl(((long double)c[i] - t) / t));
    }
    std::printf("max relative error: v_rsq_f64 %.3e   + 1 Newton step %.3e   + 2 steps %.3e   (eps = 2.22e-16)\n", m0, m1, m2);
    return 0;
}
