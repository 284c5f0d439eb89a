// micro-benchmark: the k-loop of k_syrk_cb in isolation (C = A A' on 64 x 64 workgroup tiles, 4 waves x 32 x 32, operands
// straight from global memory) and variants of it, on an ideal shape. Answers: what is the ceiling of that structure, and
// which change raises it (deeper load batches, explicit double buffering, wider wave tiles, LDS staging)?
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/syrk_direct.hip -o tools/micro/syrk_direct
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#define HC(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef double d4 __attribute__((ext_vector_type(4)));

// V0 / V1: direct loads, KU k-steps per batch (KU = 4: the product kernel)
template <int KU>
__global__ __launch_bounds__(256) void k_direct(const double *__restrict__ A, double *__restrict__ C, int M, int K, int ld) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lm = lane & 15, lk = lane >> 4;
    const int i0 = blockIdx.x * 64 + (wave & 1) * 32, j0 = blockIdx.y * 64 + (wave >> 1) * 32;
    d4 acc[2][2] = {};
    const double *pa[2] = {A + j0 + lm, A + j0 + 16 + lm}, *pb[2] = {A + i0 + lm, A + i0 + 16 + lm};
    for (int q0 = 0; q0 < K; q0 += 4 * KU) {
        double av[KU][2], bv[KU][2];
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const long long q = q0 + 4 * u + lk;
#pragma unroll
            for (int a = 0; a < 2; a++) av[u][a] = pa[a][q * ld];
#pragma unroll
            for (int b = 0; b < 2; b++) bv[u][b] = pb[b][q * ld];
        }
#pragma unroll
        for (int u = 0; u < KU; u++)
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][a], bv[u][b], acc[a][b], 0, 0, 0);
    }
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) C[i0 + b * 16 + lm + (long long)(j0 + a * 16 + lk + 4 * rr) * M] = acc[a][b][rr];
}

// V2: the same with the next batch requested before the MFMAs of the current one
__global__ __launch_bounds__(256) void k_direct_db(const double *__restrict__ A, double *__restrict__ C, int M, int K, int ld) {
    constexpr int KU = 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lm = lane & 15, lk = lane >> 4;
    const int i0 = blockIdx.x * 64 + (wave & 1) * 32, j0 = blockIdx.y * 64 + (wave >> 1) * 32;
    d4 acc[2][2] = {};
    const double *pa[2] = {A + j0 + lm, A + j0 + 16 + lm}, *pb[2] = {A + i0 + lm, A + i0 + 16 + lm};
    double av[2][KU][2], bv[2][KU][2];
    auto req = [&](int q0, double (&x)[KU][2], double (&y)[KU][2]) {
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const long long q = min(q0 + 4 * u + lk, K - 1);
#pragma unroll
            for (int a = 0; a < 2; a++) x[u][a] = pa[a][q * ld];
#pragma unroll
            for (int b = 0; b < 2; b++) y[u][b] = pb[b][q * ld];
        }
    };
    auto mm = [&](double (&x)[KU][2], double (&y)[KU][2]) {
#pragma unroll
        for (int u = 0; u < KU; u++)
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u][a], y[u][b], acc[a][b], 0, 0, 0);
    };
    req(0, av[0], bv[0]);
    for (int q0 = 0; q0 < K; q0 += 8 * KU) {
        req(q0 + 4 * KU, av[1], bv[1]);
        mm(av[0], bv[0]);
        req(q0 + 8 * KU, av[0], bv[0]);
        mm(av[1], bv[1]);
    }
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) C[i0 + b * 16 + lm + (long long)(j0 + a * 16 + lk + 4 * rr) * M] = acc[a][b][rr];
}

// V3: wave tile 32 (j) x 64 (i): 6 operand loads per 8 MFMAs; workgroup tile 128 (i) x 64 (j)
__global__ __launch_bounds__(256) void k_direct_wide(const double *__restrict__ A, double *__restrict__ C, int M, int K, int ld) {
    constexpr int KU = 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lm = lane & 15, lk = lane >> 4;
    const int i0 = blockIdx.x * 128 + (wave & 1) * 64, j0 = blockIdx.y * 64 + (wave >> 1) * 32;
    d4 acc[2][4] = {};
    const double *pa[2] = {A + j0 + lm, A + j0 + 16 + lm};
    const double *pb[4] = {A + i0 + lm, A + i0 + 16 + lm, A + i0 + 32 + lm, A + i0 + 48 + lm};
    for (int q0 = 0; q0 < K; q0 += 4 * KU) {
        double av[KU][2], bv[KU][4];
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const long long q = q0 + 4 * u + lk;
#pragma unroll
            for (int a = 0; a < 2; a++) av[u][a] = pa[a][q * ld];
#pragma unroll
            for (int b = 0; b < 4; b++) bv[u][b] = pb[b][q * ld];
        }
#pragma unroll
        for (int u = 0; u < KU; u++)
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][a], bv[u][b], acc[a][b], 0, 0, 0);
    }
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) C[i0 + b * 16 + lm + (long long)(j0 + a * 16 + lk + 4 * rr) * M] = acc[a][b][rr];
}


// V5: 64 x 64 workgroup tile on TWO waves of 32 (j) x 64 (i) each (the product's LDS tile stays 64 x 64)
template <int KU>
__global__ __launch_bounds__(128) void k_direct_2w(const double *__restrict__ A, double *__restrict__ C, int M, int K, int ld) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lm = lane & 15, lk = lane >> 4;
    const int i0 = blockIdx.x * 64, j0 = blockIdx.y * 64 + wave * 32;
    d4 acc[2][4] = {};
    const double *pa[2] = {A + j0 + lm, A + j0 + 16 + lm};
    const double *pb[4] = {A + i0 + lm, A + i0 + 16 + lm, A + i0 + 32 + lm, A + i0 + 48 + lm};
    for (int q0 = 0; q0 < K; q0 += 4 * KU) {
        double av[KU][2], bv[KU][4];
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const long long q = q0 + 4 * u + lk;
#pragma unroll
            for (int a = 0; a < 2; a++) av[u][a] = pa[a][q * ld];
#pragma unroll
            for (int b = 0; b < 4; b++) bv[u][b] = pb[b][q * ld];
        }
#pragma unroll
        for (int u = 0; u < KU; u++)
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][a], bv[u][b], acc[a][b], 0, 0, 0);
    }
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) C[i0 + b * 16 + lm + (long long)(j0 + a * 16 + lk + 4 * rr) * M] = acc[a][b][rr];
}

// V4: 64 x 64 workgroup tile, operands staged through LDS in 16-column chunks (double-buffered), 4 waves x 32 x 32
__global__ __launch_bounds__(256) void k_lds64(const double *__restrict__ A, double *__restrict__ C, int M, int K, int ld) {
    constexpr int KB = 16;
    __shared__ double As[2][KB][64 + 8], Bs[2][KB][64 + 8];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lm = lane & 15, lk = lane >> 4;
    const int ti0 = blockIdx.x * 64, tj0 = blockIdx.y * 64;
    const int wi = (wave & 1) * 32, wj = (wave >> 1) * 32;
    d4 acc[2][2] = {};
    const int lr = tid & 63, l4 = (tid >> 6) * 4;           // 256 threads: 64 rows x 4 k-groups of 4
    const double *pa = A + tj0 + lr + (long long)l4 * ld, *pb = A + ti0 + lr + (long long)l4 * ld;
    double ra[4], rb[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { ra[q] = pa[(long long)q * ld]; rb[q] = pb[(long long)q * ld]; }
#pragma unroll
    for (int q = 0; q < 4; q++) { As[0][l4 + q][lr] = ra[q]; Bs[0][l4 + q][lr] = rb[q]; }
    __syncthreads();
    const int nk = K / KB;
    for (int kb = 0; kb < nk; kb++) {
        const int cur = kb & 1;
        if (kb + 1 < nk) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                ra[q] = pa[(long long)((kb + 1) * KB + q) * ld];
                rb[q] = pb[(long long)((kb + 1) * KB + q) * ld];
            }
        }
#pragma unroll
        for (int s = 0; s < KB / 4; s++) {
            double av[2], bv[2];
#pragma unroll
            for (int a = 0; a < 2; a++) av[a] = As[cur][4 * s + lk][wj + 16 * a + lm];
#pragma unroll
            for (int b = 0; b < 2; b++) bv[b] = Bs[cur][4 * s + lk][wi + 16 * b + lm];
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        if (kb + 1 < nk) {
#pragma unroll
            for (int q = 0; q < 4; q++) { As[cur ^ 1][l4 + q][lr] = ra[q]; Bs[cur ^ 1][l4 + q][lr] = rb[q]; }
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++)
                C[ti0 + wi + b * 16 + lm + (long long)(tj0 + wj + a * 16 + lk + 4 * rr) * M] = acc[a][b][rr];
}

int main() {
    const int M = 4096, K = 1024;
    std::vector<double> hA((size_t)M * K);
    for (size_t i = 0; i < hA.size(); i++) hA[i] = ((i * 2654435761u) % 1000) / 1000.0 - 0.5;
    double *A, *C;
    HC(hipMalloc(&A, hA.size() * 8)); HC(hipMalloc(&C, (size_t)M * M * 8));
    HC(hipMemcpy(A, hA.data(), hA.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    std::vector<double> hC((size_t)M * M);
    auto run = [&](const char *name, auto launch) {
        launch();
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(hC.data(), C, hC.size() * 8, hipMemcpyDeviceToHost);
        double err = 0;
        for (int t = 0; t < 100; t++) {
            int i = (t * 7919) % M, j = (t * 104729) % M;
            double s = 0; for (int k = 0; k < K; k++) s += hA[i + (size_t)k * M] * hA[j + (size_t)k * M];
            err = fmax(err, fabs(hC[i + (size_t)j * M] - s));
        }
        const int reps = 20; float ms;
        (void)hipEventRecord(e0);
        for (int r = 0; r < reps; r++) launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %.3f ms, %5.1f TFLOP/s, err %.1e\n", name, ms / reps, 2.0 * M * M * K / (ms / reps * 1e-3) / 1e12, err);
    };
    // dynamic LDS bytes only cap the workgroups per CU (the product kernel holds a 33 KB tile: 4 per CU)
    for (int lds : {0, 33280}) {
        printf("-- %d B of LDS per workgroup\n", lds);
        run("direct KU=4 (product)", [&] { hipLaunchKernelGGL(k_direct<4>, dim3(M / 64, M / 64), dim3(256), lds, 0, A, C, M, K, M); });
        run("direct KU=8", [&] { hipLaunchKernelGGL(k_direct<8>, dim3(M / 64, M / 64), dim3(256), lds, 0, A, C, M, K, M); });
        run("direct KU=2", [&] { hipLaunchKernelGGL(k_direct<2>, dim3(M / 64, M / 64), dim3(256), lds, 0, A, C, M, K, M); });
        run("direct KU=4 double-buffered", [&] { hipLaunchKernelGGL(k_direct_db, dim3(M / 64, M / 64), dim3(256), lds, 0, A, C, M, K, M); });
        run("2 waves x 32x64, KU=4", [&] { hipLaunchKernelGGL(k_direct_2w<4>, dim3(M / 64, M / 64), dim3(128), lds, 0, A, C, M, K, M); });
        run("2 waves x 32x64, KU=2", [&] { hipLaunchKernelGGL(k_direct_2w<2>, dim3(M / 64, M / 64), dim3(128), lds, 0, A, C, M, K, M); });
        run("direct wave tile 32x64", [&] { hipLaunchKernelGGL(k_direct_wide, dim3(M / 128, M / 64), dim3(256), lds, 0, A, C, M, K, M); });
    }
    run("LDS-staged 64x64", [&] { hipLaunchKernelGGL(k_lds64, dim3(M / 64, M / 64), dim3(256), 0, 0, A, C, M, K, M); });
    return 0;
}
