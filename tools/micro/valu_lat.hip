// micro-benchmark: latency / issue cost of the FP64 VALU instructions the diagonal-block kernel chains
// (one workgroup of 256 threads = one wave per SIMD of one CU, like k_potrf64), from wall time over long loops.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 valu_lat.hip -o valu_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#define HC(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
constexpr int N = 1 << 16;
__global__ __launch_bounds__(256) void k_dep_fma(double *out, double a, double b) {
    double x = out[threadIdx.x];
#pragma unroll 16
    for (int i = 0; i < N; i++) x = __builtin_fma(x, a, b);
    out[threadIdx.x] = x;
}
__global__ __launch_bounds__(256) void k_ind_fma(double *out, double a, double b) {
    double x[8];
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = out[threadIdx.x] + k;
#pragma unroll 2
    for (int i = 0; i < N / 8; i++)
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = __builtin_fma(x[k], a, b);
    double s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s += x[k];
    out[threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_dep_rsq(double *out) {
    double x = out[threadIdx.x];
#pragma unroll 16
    for (int i = 0; i < N; i++) x = __builtin_amdgcn_rsq(x) + 1.0;
    out[threadIdx.x] = x;
}
__global__ __launch_bounds__(256) void k_lds_rt(double *out) {
    __shared__ double sh[256];
    double x = out[threadIdx.x];
    const int t = threadIdx.x;
    for (int i = 0; i < N / 16; i++) {
        sh[t] = x;
        __syncthreads();
        x = sh[(t + 64) & 255] + 1.0;
        __syncthreads();
    }
    out[t] = x;
}
int main() {
    double *d; HC(hipMalloc(&d, 256 * 8)); HC(hipMemset(d, 0, 256 * 8));
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 2; rep++) {
        HC(hipEventRecord(e0)); hipLaunchKernelGGL(k_dep_fma, dim3(1), dim3(256), 0, 0, d, 0.999, 0.001); HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
        HC(hipEventElapsedTime(&ms, e0, e1)); printf("dependent v_fma_f64: %.2f ns each\n", ms * 1e6 / N);
        HC(hipEventRecord(e0)); hipLaunchKernelGGL(k_ind_fma, dim3(1), dim3(256), 0, 0, d, 0.999, 0.001); HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
        HC(hipEventElapsedTime(&ms, e0, e1)); printf("independent v_fma_f64 (8 chains): %.2f ns each\n", ms * 1e6 / N);
        HC(hipEventRecord(e0)); hipLaunchKernelGGL(k_dep_rsq, dim3(1), dim3(256), 0, 0, d); HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
        HC(hipEventElapsedTime(&ms, e0, e1)); printf("dependent v_rsq_f64 + v_add_f64: %.2f ns per pair\n", ms * 1e6 / N);
        HC(hipEventRecord(e0)); hipLaunchKernelGGL(k_lds_rt, dim3(1), dim3(256), 0, 0, d); HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
        HC(hipEventElapsedTime(&ms, e0, e1)); printf("LDS write + barrier + read + barrier: %.2f ns per round\n", ms * 1e6 / (N / 16));
    }
    return 0;
}
