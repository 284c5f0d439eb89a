// micro-benchmark: FP64 MFMA DGEMM with LDS-staged 128 x 128 workgroup tiles and 8 waves per workgroup
// (two MFMA-issuing waves per SIMD: a single wave only issues one v_mfma_f64_16x16x4_f64 per ~138 cycles,
// two waves on a SIMD interleave to the full 64-cycle rate -- tools/micro/mix64.hip).
// C (M x N, column-major) -= A (M x K) * B (N x K)'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#define HC(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int TM = 128, TN = 128, KB = 16;

__global__ __launch_bounds__(512) void k_dgemm_mfma(const double *__restrict__ A, const double *__restrict__ B, double *__restrict__ C,
                                                    int M, int N, int K, int lda, int ldb, int ldc) {
    __shared__ double As[2][KB][TM + 8], Bs[2][KB][TN + 8];     // +8: rows k, k+1, .. land in different banks
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lm = lane & 15, lk = lane >> 4;
    const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;
    const int wi = (wave & 3) * 32, wj = (wave >> 2) * 64;      // wave sub-tile: 32 rows x 64 columns
    d4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = (d4){0, 0, 0, 0};
    // staging: 512 threads, A tile 128 x 16 = 2048 doubles -> 4 per thread (row tid % 128, k = (tid / 128) * 4 ..)
    const int lr = tid & 127, l4 = (tid >> 7) * 4;
    const double *pa = A + m0 + lr + (long long)l4 * lda;
    const double *pb = B + n0 + lr + (long long)l4 * ldb;
    double ra[4], rb[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { ra[q] = pa[(long long)q * lda]; rb[q] = pb[(long long)q * ldb]; }
#pragma unroll
    for (int q = 0; q < 4; q++) { As[0][l4 + q][lr] = ra[q]; Bs[0][l4 + q][lr] = rb[q]; }
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
#pragma unroll
        for (int s = 0; s < KB / 4; s++) {
            double av[2], bv[4];
#pragma unroll
            for (int a = 0; a < 2; a++) av[a] = As[cur][4 * s + lk][wi + 16 * a + lm];
#pragma unroll
            for (int b = 0; b < 4; b++) bv[b] = Bs[cur][4 * s + lk][wj + 16 * b + lm];
            // D[m = column j][n = row i] : first operand = rows of B, second = rows of A (lanes walk i)
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[b], av[a], acc[a][b], 0, 0, 0);
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
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int i = m0 + wi + 16 * a + lm, j = n0 + wj + 16 * b + lk + 4 * rr;
                C[i + (long long)j * ldc] -= acc[a][b][rr];
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
    hipLaunchKernelGGL(k_dgemm_mfma, grid, dim3(512), 0, 0, A, B, C, M, N, K, M, N, M);
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
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_dgemm_mfma, grid, dim3(512), 0, 0, A, B, C, M, N, K, M, N, M);
    HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1)); HC(hipEventElapsedTime(&ms, e0, e1));
    printf("MFMA dgemm 128x128 tiles, 8 waves: %d x %d x %d: %.3f ms per launch, %.1f TFLOP/s; max err on samples %.2e\n", M, N, K, ms / reps,
           2.0 * M * N * K / (ms / reps * 1e-3) / 1e12, err);
    return 0;
}
