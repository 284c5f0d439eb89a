// dense.hip -- the dense-operator leg of the separable (Kronecker) path, SURVEY 8 f3: `Q = kron(Q_1, Q_2)`
// (SeparableModel, /root/reference/src/latent_models/separable.jl:122-172) is solved / sampled as a sweep over the large factor
// followed by the SMALL factor applied as a dense n1 x n1 operator (Q_1^-1, or A_1 = P_1' L_1^-T for samples) to the row-major
// n1 x n2 result:  R = D T.  This is a genuine dense contraction (n1 = 512, n2 = 250 000 at BASELINE cfg 5: 1.3e11 flops), so
// it runs on the FP64 MFMA with the library's own 32 x 32 wave products (kernels.h) -- no vendor GEMM.
//   k_dense_apply   R[i1, i2] = sum_k D[i1, k] T[k, i2]      (all row-major; T and R are n1 x n2, D is n1 x n1)
//   k_transpose     dst[j, i] = src[i, j]                    (row-major rows x cols -> cols x rows; two large factors)
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace gmrfx {

typedef gmrfx_d4 d4;
typedef gmrfx_d2u d2u;

// One workgroup = one 64 (i1) x 64 (i2) tile of R, four waves of 32 x 32. In the wave products' terms the "m" index is i1
// (first operand D: one row pointer per tile row, contiguous k: the _kr form) and the "n" index is i2 (second operand T:
// rows i2 contiguous along the lanes, stride n2 per k), so a lane's two column tiles are i2 = n0 + 2 lm, + 1: 16-byte loads of
// T and 16-byte stores of R, 256 contiguous bytes per 16 lanes. Tiles are dealt so that the n1 / 64 workgroups that share one
// 64-column slab of T (they differ in i1 only) run on ONE XCD (workgroup id mod 8) one after the other: the slab is read from
// HBM once, by one L2.
__global__ __launch_bounds__(256) void k_dense_apply(const double *__restrict__ D, const double *__restrict__ T, double *__restrict__ R,
                                                     int n1, long long n2, int mt) {
    const int xcd = blockIdx.x & 7;
    const long long slot = blockIdx.x >> 3;
    const int bm = (int)(slot % mt);
    const long long bn = (slot / mt) * 8 + xcd;
    if (bn * 64 >= n2) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lm = lane & 15, lk = lane >> 4;
    const int m0 = bm * 64 + (wave & 1) * 32;
    const long long n0 = bn * 64 + (wave >> 1) * 32;
    if (m0 >= n1 || n0 >= n2) return;
    d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
    // clamped operand rows (lanes past the edge re-read the last valid pair; never stored). n2 odd: the pair that starts at the
    // last element of a row reads one double past it -- the next row's first element. For the LAST row of T that would be past
    // the array: the pair path stops one row short (qpair) and the masked element path below takes the rest, so an exactly
    // sized T is never over-read (no slack contract on the public entry point gmrfx_dense_apply_dev)
    const int i1a = min(m0 + 2 * lm, n1 - 1), i1b = min(m0 + 2 * lm + 1, n1 - 1);
    const long long i2 = min(n0 + 2 * lm, (n2 - 1) & ~1ll);
    const double *pa0 = D + (long long)i1a * n1, *pa1 = D + (long long)i1b * n1;
    const double *pb2 = T + i2;
    const int qpair = (n2 & 1) ? n1 - 1 : n1;
    int qd = wave_gemm_32x32_kr(acc, pa0, pa1, pb2, n2, 0, qpair, lk);
    if (qd < n1) {
        auto fa = [&](int i, int q) { return D[(long long)min(i, n1 - 1) * n1 + min(q, n1 - 1)]; };
        auto fb = [&](int q, int j) { return T[(long long)min(q, n1 - 1) * n2 + min((long long)j + n0, n2 - 1)]; };
        wave_gemm_32x32_pm(acc, m0, 0, qd, n1, fa, fb, lm, lk);
    }
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int i1 = m0 + 2 * (lk + 4 * rr) + a;
            const long long j = n0 + 2 * lm;
            if (i1 < n1 && j < n2) {
                double *dst = R + (long long)i1 * n2 + j;
                if (j + 1 < n2) *(d2u *)dst = (d2u){acc[a][0][rr], acc[a][1][rr]};
                else dst[0] = acc[a][0][rr];
            }
        }
}

// dst (cols x rows, row-major) = src' (src rows x cols, row-major): 64 x 64 tiles through LDS, both sides in 512-byte runs
__global__ __launch_bounds__(256) void k_transpose(const double *__restrict__ src, double *__restrict__ dst, long long rows, long long cols) {
    __shared__ double Tl[64 * 65];
    const long long nbc = (cols + 63) >> 6;
    const long long bi = blockIdx.x / nbc, bj = blockIdx.x % nbc;
    const int a = threadIdx.x & 63, b = threadIdx.x >> 6;
    const long long i0 = bi * 64, j0 = bj * 64;
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) {
        const long long i = min(i0 + b + 4 * u, rows - 1), j = min(j0 + a, cols - 1);
        v[u] = src[i * cols + j];
    }
#pragma unroll
    for (int u = 0; u < 16; u++) Tl[(b + 4 * u) * 65 + a] = v[u];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 16; u++) {
        const long long j = j0 + b + 4 * u, i = i0 + a;
        if (j < cols && i < rows) dst[j * rows + i] = Tl[a * 65 + b + 4 * u];
    }
}

void launch_dense_apply(hipStream_t st, const double *D, const double *T, double *R, int n1, long long n2) {
    if (n1 <= 0 || n2 <= 0) return;
    const int mt = (n1 + 63) / 64;
    const long long nt = (n2 + 63) / 64;
    const long long groups = (nt + 7) / 8;            // eight column tiles (one per XCD) per group
    hipLaunchKernelGGL(k_dense_apply, dim3((unsigned)(groups * mt * 8)), dim3(256), 0, st, D, T, R, n1, n2, mt);
}
void launch_transpose(hipStream_t st, const double *src, double *dst, long long rows, long long cols) {
    if (rows <= 0 || cols <= 0) return;
    const long long nb = ((rows + 63) >> 6) * ((cols + 63) >> 6);
    hipLaunchKernelGGL(k_transpose, dim3((unsigned)nb), dim3(256), 0, st, src, dst, rows, cols);
}

}  // namespace gmrfx
