// quadform.hip -- q_v = (x_v - mu)' Q (x_v - mu) on the caller's CSC values, for a batch of vectors.
//
// This is the `dot(r, d.precision * r)` of logpdf(::WorkspaceGMRF, z)
// (/root/reference/src/workspace/workspace_gmrf.jl:288-292) and of sqmahal
// (/root/reference/src/gmrf.jl:94-97). Together with gmrfx_refactorize + gmrfx_logdet it makes the
// hyper-parameter loop of docs/src/literate-tutorials/workspace_factorization_reuse.jl:94-102 run
// without Q's values or z ever leaving HBM.
//
// HBM-bound: one pass over the values (8 B) and row indices (4 B) of the stored pattern per vector;
// x is gathered through L2. Only the triangle that defines Q (Symmetric(Q) semantics, Symbolic::in_use)
// contributes: diagonal entries once, off-diagonal entries twice, the other triangle is skipped.
// Sums are formed in a fixed order (16-lane groups -> block tree -> one block per vector), so the
// result is reproducible from run to run.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace gmrfx {

namespace {
constexpr int QF_COLS = 128;   // columns per workgroup: 16 groups of 16 lanes, 8 columns each
}

__global__ __launch_bounds__(256) void k_quadform(int n, const long long *__restrict__ colptr, const int *__restrict__ row,
                                                  const double *__restrict__ val, int use_lower,
                                                  const double *__restrict__ X, long long ldx,
                                                  const double *__restrict__ mu, double *__restrict__ part) {
    __shared__ double sh[256];
    const int tid = threadIdx.x;
    const int g = tid >> 4, l = tid & 15;
    const double *x = X + (long long)blockIdx.y * ldx;
    double acc = 0.0;
#pragma unroll 2
    for (int t = 0; t < QF_COLS / 16; t++) {
        const int j = blockIdx.x * QF_COLS + t * 16 + g;   // neighbouring groups walk neighbouring columns
        if (j < n) {
            const long long p0 = colptr[j], p1 = colptr[j + 1];
            const double dj = x[j] - (mu ? mu[j] : 0.0);
            double a = 0.0;
            for (long long p = p0 + l; p < p1; p += 16) {
                const int i = row[p];
                const bool in_tri = use_lower ? (i > j) : (i < j);
                const double wgt = (i == j) ? 1.0 : (in_tri ? 2.0 : 0.0);
                a += wgt * val[p] * (x[i] - (mu ? mu[i] : 0.0));
            }
            acc += a * dj;
        }
    }
    sh[tid] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) sh[tid] += sh[tid + st];
        __syncthreads();
    }
    if (tid == 0) part[(long long)blockIdx.y * gridDim.x + blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(256) void k_quadform_final(const double *__restrict__ part, int nblk, double *__restrict__ out) {
    __shared__ double sh[256];
    const int tid = threadIdx.x;
    const double *p = part + (long long)blockIdx.x * nblk;
    double acc = 0.0;
    for (int i = tid; i < nblk; i += 256) acc += p[i];
    sh[tid] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) sh[tid] += sh[tid + st];
        __syncthreads();
    }
    if (tid == 0) out[blockIdx.x] = sh[0];
}

int quadform_blocks(int n) { return (n + QF_COLS - 1) / QF_COLS; }

void launch_quadform(hipStream_t st, int n, const long long *colptr, const int *row, const double *val, int use_lower,
                     const double *X, long long ldx, int nvec, const double *mu, double *part, double *out) {
    if (nvec <= 0 || n <= 0) return;
    const int nblk = quadform_blocks(n);
    hipLaunchKernelGGL(k_quadform, dim3(nblk, nvec), dim3(256), 0, st, n, colptr, row, val, use_lower, X, ldx, mu, part);
    hipLaunchKernelGGL(k_quadform_final, dim3(nvec), dim3(256), 0, st, part, nblk, out);
}

}  // namespace gmrfx
