// micro-benchmark: can a VALU (v_fma_f64) DGEMM tile kernel beat the FP64 MFMA ceiling (36 TF/s) on MI355X?
// Measured: 39.7 TFLOP/s (this version: 16-byte LDS reads, k loop not unrolled; 8-byte reads 37.7; a fully
// unrolled / register-double-buffered k loop spills and drops to 5).
// C (M x N, column-major) -= A (M x K) * B (N x K)'   -- the shape of the contribution-block SYRK / panel GEMMs.
// 128 x 128 workgroup tile, 256 threads, 8 x 8 register patch per thread, operands staged through LDS.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 dgemm_valu.hip -o dgemm_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#define HC(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

constexpr int TM = 128, TN = 128, KB = 8;

__global__ __launch_bounds__(256, 2) void k_dgemm_valu(const double *__restrict__ A, const double *__restrict__ B, double *__restrict__ C,
                                                       int M, int N, int K, int lda, int ldb, int ldc) {
    __shared__ __attribute__((aligned(16))) double As[2][KB][TM], Bs[2][KB][TN];
    const int tid = threadIdx.x;
    const int ty = tid >> 4, tx = tid & 15;
    const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;
    double acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = 0.0;
    // global -> register staging: thread loads row (tid % 128), k = (tid / 128) * 4 .. +3 of A and of B
    const int lr = tid & 127, lk = (tid >> 7) * 4;
    const double *pa = A + m0 + lr + (long long)lk * lda;
    const double *pb = B + n0 + lr + (long long)lk * ldb;
    double ra[4], rb[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { ra[q] = pa[(long long)q * lda]; rb[q] = pb[(long long)q * ldb]; }
#pragma unroll
    for (int q = 0; q < 4; q++) { As[0][lk + q][lr] = ra[q]; Bs[0][lk + q][lr] = rb[q]; }
    __syncthreads();
    const int nk = K / KB;
    for (int kb = 0; kb < nk; kb++) {
        const int cur = kb & 1;
        if (kb + 1 < nk) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                ra[q] = pa[(long long)((kb + 1) * KB + q) * lda];
                rb[q] = pb[(long long)((kb + 1) * KB + q) * ldb];
            }
        }
#pragma unroll 1
        for (int k = 0; k < KB; k++) {
            double a[8], b[8];      // rows 2 ty + 32 i' + {0,1}, columns 2 tx + 32 j' + {0,1}: 16-byte LDS reads
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const double2 v = *reinterpret_cast<const double2 *>(&As[cur][k][2 * ty + 32 * i]);
                a[2 * i] = v.x; a[2 * i + 1] = v.y;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const double2 v = *reinterpret_cast<const double2 *>(&Bs[cur][k][2 * tx + 32 * j]);
                b[2 * j] = v.x; b[2 * j + 1] = v.y;
            }
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int j = 0; j < 8; j++) acc[i][j] = fma(a[i], b[j], acc[i][j]);
        }
        if (kb + 1 < nk) {
#pragma unroll
            for (int q = 0; q < 4; q++) { As[cur ^ 1][lk + q][lr] = ra[q]; Bs[cur ^ 1][lk + q][lr] = rb[q]; }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            double *c = C + (m0 + 2 * ty + 32 * (i >> 1) + (i & 1)) + (long long)(n0 + 2 * tx + 32 * (j >> 1) + (j & 1)) * ldc;
            *c -= acc[i][j];
        }
}

int main() {
    const int M = 4096, N = 4096, K = 1024;
    std::vector<double> hA((size_t)M * K), hB((size_t)N * K), hC((size_t)M * N, 0.0);
    for (size_t i = 0; i < hA.size(); i++) hA[i] = ((i * 2654435761u) % 1000) / 1000.0 - 0.5;
    for (size_t i = 0; i < hB.size(); i++) hB[i] = ((i * 40503u + 7) % 1000) / 1000.0 - 0.5;
    double *A, *B, *C;
    HC(hipMalloc(&A, hA.size() * 8)); HC(hipMalloc(&B, hB.size() * 8)); HC(hipMalloc(&C, hC.size() * 8));
    HC(hipMemcpy(A, hA.data(), hA.size() * 8, hipMemcpyHostToDevice));
    HC(hipMemcpy(B, hB.data(), hB.size() * 8, hipMemcpyHostToDevice));
    HC(hipMemset(C, 0, hC.size() * 8));
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    dim3 grid(M / TM, N / TN);
    hipLaunchKernelGGL(k_dgemm_valu, grid, dim3(256), 0, 0, A, B, C, M, N, K, M, N, M);
    HC(hipDeviceSynchronize());
    HC(hipMemcpy(hC.data(), C, hC.size() * 8, hipMemcpyDeviceToHost));
    double err = 0;
    for (int t = 0; t < 200; t++) {
        int i = (t * 7919) % M, j = (t * 104729) % N;
        double s = 0; for (int k = 0; k < K; k++) s += hA[i + (size_t)k * M] * hB[j + (size_t)k * N];
        err = fmax(err, fabs(hC[i + (size_t)j * M] + s));
    }
    const int reps = 20; float ms;
    HC(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_dgemm_valu, grid, dim3(256), 0, 0, A, B, C, M, N, K, M, N, M);
    HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1)); HC(hipEventElapsedTime(&ms, e0, e1));
    printf("VALU dgemm 128x128 tiles: %d x %d x %d: %.3f ms per launch, %.1f TFLOP/s; max err on samples %.2e\n", M, N, K, ms / reps,
           2.0 * M * N * K / (ms / reps * 1e-3) / 1e12, err);
    return 0;
}
