// micro-benchmark: what one DEPENDENT launch costs on the GPU side, empty kernels and tiny real ones, enqueued in a stream (host far
// ahead) and replayed from a captured hipGraph. The top of the elimination tree is a chain of ~350 such launches per factorisation.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 launch_floor.hip -o launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define HC(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
__global__ void k_empty() {}
__global__ void k_touch(double *p, int n) {      // one dependent global round trip + store, like the smallest chain kernels
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0000001 + 1e-9;
}
__global__ void k_spin(double *p, int n, long long cycles) {     // the same with ~cycles of work: the host gets ahead of the GPU
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const long long t0 = clock64();
    double v = i < n ? p[i] : 0.0;
    while (clock64() - t0 < cycles) v = v * 1.0000001 + 1e-9;
    if (i < n) p[i] = v;
}
int main() {
    hipStream_t st; HC(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    double *d; HC(hipMalloc(&d, 1 << 20)); HC(hipMemset(d, 0, 1 << 20));
    const int N = 2000;
    auto wall = [&](auto enqueue) {
        enqueue(); hipStreamSynchronize(st);
        auto t0 = std::chrono::steady_clock::now();
        enqueue(); hipStreamSynchronize(st);
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
    };
    for (int grid : {1, 64, 256}) {
        const double e = wall([&] { for (int k = 0; k < N; k++) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, st); });
        const double t = wall([&] { for (int k = 0; k < N; k++) hipLaunchKernelGGL(k_touch, dim3(grid), dim3(256), 0, st, d, 65536); });
        printf("stream, %3d workgroups: empty kernel %.2f us per dependent launch, one-round-trip kernel %.2f us\n", grid, e, t);
    }
    for (long long cyc : {12000LL, 24000LL}) {           // ~5 / ~10 us kernels (clock64 ticks at the shader clock)
        const double a = wall([&] { for (int k = 0; k < N; k++) hipLaunchKernelGGL(k_spin, dim3(1), dim3(256), 0, st, d, 65536, cyc); });
        hipGraph_t g; hipGraphExec_t ge;
        HC(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int k = 0; k < N; k++) hipLaunchKernelGGL(k_spin, dim3(1), dim3(256), 0, st, d, 65536, cyc);
        HC(hipStreamEndCapture(st, &g));
        HC(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        HC(hipGraphLaunch(ge, st)); HC(hipStreamSynchronize(st));
        auto t0 = std::chrono::steady_clock::now();
        HC(hipGraphLaunch(ge, st)); HC(hipStreamSynchronize(st));
        const double b = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        printf("kernels that spin %lld cycles: %.2f us per dependent launch in a stream, %.2f us per node of a hipGraph\n", cyc, a, b);
        HC(hipGraphExecDestroy(ge)); HC(hipGraphDestroy(g));
    }
    for (int grid : {1, 64}) {
        hipGraph_t g; hipGraphExec_t ge;
        HC(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int k = 0; k < N; k++) hipLaunchKernelGGL(k_touch, dim3(grid), dim3(256), 0, st, d, 65536);
        HC(hipStreamEndCapture(st, &g));
        HC(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        HC(hipGraphLaunch(ge, st)); HC(hipStreamSynchronize(st));
        auto t0 = std::chrono::steady_clock::now();
        HC(hipGraphLaunch(ge, st)); HC(hipStreamSynchronize(st));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        printf("hipGraph, %3d workgroups: one-round-trip kernel %.2f us per dependent node\n", grid, us);
        HC(hipGraphExecDestroy(ge)); HC(hipGraphDestroy(g));
    }
    return 0;
}
