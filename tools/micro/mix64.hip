// micro-benchmark: do the FP64 MFMA pipe and the FP64 VALU FMA pipe run CONCURRENTLY on MI355X?
// A workgroup has 8 waves (2 per SIMD): `nm` of them issue v_mfma_f64_16x16x4_f64 back to back, the rest
// v_fma_f64 on 32 independent accumulators. Alone: MFMA 36 TF/s, VALU 64 TF/s; if the pipes overlap the mixed
// configuration approaches the sum.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k_mix(double *out, int iters, int nm, double a0, double b0) {
    const int wave = threadIdx.x >> 6;
    double s = 0;
    if (wave < nm) {
        d4 acc[8];
        for (int i = 0; i < 8; i++) acc[i] = (d4){0, 0, 0, 0};
        double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double acc[32];
        for (int i = 0; i < 32; i++) acc[i] = 1e-3 * i;
        double a = a0 + threadIdx.x * 1e-6, b = b0 * 1e-3;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 32; i++) acc[i] = fma(acc[i], a, b);
        }
        for (int i = 0; i < 32; i++) s += acc[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    double *out; hipMalloc(&out, (size_t)1 << 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int wgs = 1024, iters = 4000;
    for (int nm : {0, 8, 4, 2, 6}) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0); hipLaunchKernelGGL(k_mix, dim3(wgs), dim3(512), 0, 0, out, iters, nm, 1.0001, 0.9999); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fm = (double)wgs * nm * iters * 8 * 2048.0, fv = (double)wgs * (8 - nm) * iters * 32 * 64 * 2.0;
            if (rep) printf("mfma waves %d / valu waves %d per WG: %.3f ms  MFMA %.1f TF/s + VALU %.1f TF/s = %.1f TF/s\n", nm, 8 - nm, ms, fm / ms / 1e9,
                            fv / ms / 1e9, (fm + fv) / ms / 1e9);
        }
    }
    return 0;
}
