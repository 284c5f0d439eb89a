// gemm128.hip -- C -= A A' on 128 x 128 workgroup tiles staged through LDS, 8 waves per workgroup (two
// MFMA-issuing waves per SIMD): the kernel of tools/micro/dgemm_mfma.hip (54 TFLOP/s on large uniform problems
// against ~40-46 for the direct-from-L2 64 x 64 tiles of kernels.hip) for the launches that are big enough to fill
// the chip several times over with such tiles -- the huge fronts of 3-D problems. 2-D problems never get here:
// their top fronts have a few hundred 128-tiles at most and lose more to tile quantisation than they gain.
//
// Two uses, both "lower trapezoid of C -= (rows of A) x (rows of A)'":
//   * panel update of the blocked factorisation (what k_gemm_nt does): C = panel columns [c0, min(c1, c)) rows c0..,
//     A = the finished panel columns k0 .. k0 + K - 1 of the same rows;
//   * contribution block: CB -= L21 L21' after k_syrk_cb<1> has written the gathered children into CB.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace gmrfx {

typedef double d4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int TM = 128, KB = 16;
}

// mode 0: panel update (k0, K, c0, c1 as in k_gemm_nt); mode 1: contribution block of the front (CB, K = c)
__global__ __launch_bounds__(512) void k_nt128(DevSym S, const int *__restrict__ list, int mode, int k0, int Kin, int c0, int c1,
                                               double *__restrict__ L, double *__restrict__ CB, FrontArg fa) {
    __shared__ double As[2][KB][TM + 8], Bs[2][KB][TM + 8];     // +8: rows k, k+1, .. land in different banks
    const FrontView fv = front_view(S, list, blockIdx.z, fa);
    const int c = fv.c, r = fv.r, ld = fv.ld;
    double *P = L + fv.pp;
    int M, N, K, ldc;
    const double *A;
    double *C;
    if (mode == 0) {
        if (c0 >= c) return;
        M = r - c0; N = min(c1, c) - c0; K = Kin; ldc = ld;
        A = P + c0 + (long long)k0 * ld;
        C = P + c0 + (long long)c0 * ld;
    } else {
        M = N = r - c; K = c; ldc = M;
        A = P + c;
        C = CB + S.cbptr[fv.s];
    }
    const int bi = blockIdx.x, bj = blockIdx.y;
    if (bj > bi || bi * TM >= M || bj * TM >= N) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lm = lane & 15, lk = lane >> 4;
    const int m0 = bi * TM, n0 = bj * TM;
    const int wi = (wave & 3) * 32, wj = (wave >> 2) * 64;      // wave sub-tile: 32 rows x 64 columns
    d4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
    // staging: 512 threads, one 128 x 16 slab of each operand = 4 values per thread (row tid % 128, k = (tid / 128) * 4 ..)
    const int lr = tid & 127, l4 = (tid >> 7) * 4;
    const double *pa = A + min(m0 + lr, M - 1);          // rows of C's rows
    const double *pb = A + min(n0 + lr, M - 1);          // rows of C's columns (N <= M: same matrix)
    double ra[4], rb[4];
    auto fetch = [&](int kb) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int k = kb * KB + l4 + q;
            const long long off = (long long)min(k, K - 1) * ld;
            const double mk = k < K ? 1.0 : 0.0;
            ra[q] = pa[off] * mk;
            rb[q] = pb[off] * mk;
        }
    };
    const int nk = (K + KB - 1) / KB;
    fetch(0);
#pragma unroll
    for (int q = 0; q < 4; q++) { As[0][l4 + q][lr] = ra[q]; Bs[0][l4 + q][lr] = rb[q]; }
    __syncthreads();
    for (int kb = 0; kb < nk; kb++) {
        const int cur = kb & 1;
        if (kb + 1 < nk) fetch(kb + 1);
#pragma unroll
        for (int s4 = 0; s4 < KB / 4; s4++) {
            double av[2], bv[4];
#pragma unroll
            for (int a = 0; a < 2; a++) av[a] = As[cur][4 * s4 + lk][wi + 16 * a + lm];
#pragma unroll
            for (int b = 0; b < 4; b++) bv[b] = Bs[cur][4 * s4 + lk][wj + 16 * b + lm];
            // D[m = column j][n = row i]: first operand = rows of B, second = rows of A (lanes walk i)
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
    // C -= acc on the lower trapezoid; loads of a batch of 16 first, then its stores
#pragma unroll
    for (int a = 0; a < 2; a++) {
        double cv[4][4];
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int i = min(m0 + wi + 16 * a + lm, M - 1), j = min(n0 + wj + 16 * b + lk + 4 * rr, N - 1);
                cv[b][rr] = C[i + (long long)j * ldc];
            }
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int i = m0 + wi + 16 * a + lm, j = n0 + wj + 16 * b + lk + 4 * rr;
                if (i < M && j < N && i >= j) C[i + (long long)j * ldc] = cv[b][rr] - acc[a][b][rr];
            }
    }
}

static inline int cdiv128(int a) { return (a + TM - 1) / TM; }

void launch_nt128_panel(hipStream_t st, const DevSym &S, const int *list, int nactive, int k0, int K, int c0, int c1,
                        int maxM, int maxN, double *L, const FrontArg &fa) {
    hipLaunchKernelGGL(k_nt128, dim3(cdiv128(maxM) | 1, cdiv128(maxN) | 1, nactive), dim3(512), 0, st, S, list, 0, k0, K, c0, c1, L,
                       (double *)nullptr, fa);
}
void launch_nt128_cb(hipStream_t st, const DevSym &S, const int *list, int nfronts, int max_trail, double *L, double *CB) {
    hipLaunchKernelGGL(k_nt128, dim3(cdiv128(max_trail) | 1, cdiv128(max_trail) | 1, nfronts), dim3(512), 0, st, S, list, 1, 0, 0, 0, 0, L,
                       CB, FrontArg{0, 0, 0, 0, 0, 0, 0});
}

}  // namespace gmrfx
