// micro-benchmark: what does a workgroup that exits at once cost? (grids sized by the largest front of a level
// launch many of them). N workgroups of 256 threads, LDS bytes of static LDS, every workgroup reads one int and returns.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 empty_wg.hip -o empty_wg
#include <hip/hip_runtime.h>
#include <cstdio>
#define HC(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
template <int LDS>
__global__ __launch_bounds__(256) void k_empty(const int *__restrict__ lim, double *out) {
    __shared__ double sh[LDS / 8 > 0 ? LDS / 8 : 1];
    if ((int)blockIdx.x >= lim[0]) return;
    sh[threadIdx.x] = threadIdx.x;
    __syncthreads();
    out[threadIdx.x] = sh[255 - threadIdx.x];
}
int main() {
    int *lim; double *out; HC(hipMalloc(&lim, 4)); HC(hipMalloc(&out, 256 * 8));
    int zero = 0; HC(hipMemcpy(lim, &zero, 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    float ms;
    for (int n : {1, 1024, 8192, 32768, 131072}) {
        for (int rep = 0; rep < 2; rep++) {
            HC(hipEventRecord(e0));
            for (int r = 0; r < 20; r++) hipLaunchKernelGGL(k_empty<33280>, dim3(n), dim3(256), 0, 0, lim, out);
            HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1)); HC(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("33 KB LDS: %6d empty workgroups: %7.2f us per launch (%.1f ns per workgroup)\n", n, ms * 1e3 / 20, ms * 1e6 / 20 / n);
            HC(hipEventRecord(e0));
            for (int r = 0; r < 20; r++) hipLaunchKernelGGL(k_empty<2048>, dim3(n), dim3(256), 0, 0, lim, out);
            HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1)); HC(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf(" 2 KB LDS: %6d empty workgroups: %7.2f us per launch (%.1f ns per workgroup)\n", n, ms * 1e3 / 20, ms * 1e6 / 20 / n);
        }
    }
    return 0;
}
