// kernels.hip -- HIP kernels (gfx950 / CDNA4) for the numeric phases of the multifrontal
// supernodal Cholesky: assembly, dense partial factorisation of fronts, multi-RHS triangular
// sweeps, log-determinant and permutation/transposition of right-hand sides.
//
// Layout in HBM
//   L      supernode panels, column-major r_s x c_s, leading dimension ld_s, 128-B aligned
//   CB     contribution blocks (r-c)x(r-c), column-major, lower triangle meaningful
//   X      right-hand sides in elimination order, ROW-major n x nrhs (a row = one DoF, so a
//          gather/scatter of a front's rows moves whole 8*nrhs-byte segments)
//   W      per-supernode update vectors (r-c) x nrhs, row-major (forward sweep hand-off)
// Dense contractions run on the FP64 matrix cores: v_mfma_f64_16x16x4_f64, whose C/D map is
// col = lane&15, row = (lane>>4) + 4*reg and A/B maps are A[lane&15][lane>>4],
// B[lane>>4][lane&15] (cdna_hip_programming.md section 3).
#include <hip/hip_runtime.h>

#include <climits>

#include "kernels.h"

namespace gmrfx {

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int lower_bound_i32(const int *a, int n, int v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ------------------------------------------------------------------------------------------
// Factorisation
// ------------------------------------------------------------------------------------------

// Zero the front, scatter Q's values, extend-add the children's contribution blocks.
// Block (bx, f) owns front-local columns [bx*CW, bx*CW+CW) of front f: every target entry has
// exactly one owner block, children are applied one after the other, so the sum order is
// fixed and the result is bit-reproducible (no atomics).
__global__ __launch_bounds__(256) void k_assemble(DevSym S, const int *__restrict__ list,
                                                  const double *__restrict__ nzval, double *__restrict__ L,
                                                  double *__restrict__ CB) {
    const int s = list[blockIdx.y];
    const int c = S.sfirst[s + 1] - S.sfirst[s];
    const int r = (int)(S.rowptr[s + 1] - S.rowptr[s]);
    const int col0 = blockIdx.x * ASM_CW;
    if (col0 >= r) return;
    const int col1 = min(col0 + ASM_CW, r);
    const int ld = S.ld[s];
    const int m = r - c;
    double *P = L + S.panelptr[s];
    double *U = CB + S.cbptr[s];
    const int tid = threadIdx.x;
    for (int col = col0; col < col1; col++) {
        if (col < c) { for (int i = tid; i < ld; i += 256) P[i + (long long)col * ld] = 0.0; }
        else { for (int i = tid; i < m; i += 256) U[i + (long long)(col - c) * m] = 0.0; }
    }
    __syncthreads();
    if (col0 < c) {
        const long long q0 = S.qptr[s];
        const int nq = (int)(S.qptr[s + 1] - q0);
        const int *qd = S.qdst + q0;
        const int *qs = S.qsrc + q0;
        const int lo = lower_bound_i32(qd, nq, col0 * ld);
        const int hi = lower_bound_i32(qd, nq, min(col1, c) * ld);
        for (int q = lo + tid; q < hi; q += 256) P[qd[q]] = nzval[qs[q]];
    }
    __syncthreads();
    for (long long ch = S.childptr[s]; ch < S.childptr[s + 1]; ch++) {
        const int d = S.children[ch];
        const int cd = S.sfirst[d + 1] - S.sfirst[d];
        const int md = (int)(S.rowptr[d + 1] - S.rowptr[d]) - cd;
        const int *reld = S.rel + S.rowptr[d] + cd;
        const double *Ud = CB + S.cbptr[d];
        const int j0 = lower_bound_i32(reld, md, col0);
        const int j1 = lower_bound_i32(reld, md, col1);
        for (int j = j0; j < j1; j++) {
            const int tc = reld[j];
            for (int i = j + tid; i < md; i += 256) {
                const int ti = reld[i];
                const double v = Ud[i + (long long)j * md];
                if (tc < c) P[ti + (long long)tc * ld] += v;
                else U[(ti - c) + (long long)(tc - c) * m] += v;
            }
        }
        __syncthreads();
    }
}

// Cholesky of the NB x NB diagonal block at block-column kb of every active front.
// One workgroup per front; LDL'-style right-looking updates need one barrier per column, the
// square roots are applied when the block is written back.
__global__ __launch_bounds__(256) void k_potrf(DevSym S, const int *__restrict__ list, int kb,
                                               double *__restrict__ L, int *__restrict__ info) {
    __shared__ double D[NB * (NB + 1)];
    const int s = list[blockIdx.x];
    const int c = S.sfirst[s + 1] - S.sfirst[s];
    if (kb >= c) return;
    const int w = min(NB, c - kb);
    const int ld = S.ld[s];
    double *P = L + S.panelptr[s] + kb + (long long)kb * ld;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < w * w; idx += 256) {
        const int i = idx % w, j = idx / w;
        D[i + j * (NB + 1)] = (i >= j) ? P[i + (long long)j * ld] : 0.0;
    }
    const int tx = tid & 15, ty = tid >> 4;
    for (int j = 0; j < w; j++) {
        __syncthreads();
        const double dj = D[j + j * (NB + 1)];
        const double inv = 1.0 / dj;
        for (int k = j + 1 + ty; k < w; k += 16) {
            const double wk = D[k + j * (NB + 1)] * inv;
            for (int i = k + tx; i < w; i += 16) D[i + k * (NB + 1)] -= D[i + j * (NB + 1)] * wk;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < w * w; idx += 256) {
        const int i = idx % w, j = idx / w;
        if (i < j) continue;
        const double dj = D[j + j * (NB + 1)];
        const double sq = sqrt(dj);
        if (i == j) {
            if (!(dj > 0.0)) atomicMin(info, S.sfirst[s] + kb + j);
            P[i + (long long)j * ld] = sq;
        } else {
            P[i + (long long)j * ld] = D[i + j * (NB + 1)] / sq;
        }
    }
}

// Rows below the diagonal block: X * D' = A. A workgroup owns 64 rows; the 64 x w row block
// sits in LDS (transposed: R[k][i], conflict-free), thread (i, g) updates the columns k = g mod 4
// of row i in a column sweep (one barrier per column). The diagonal of D holds reciprocals.
__global__ __launch_bounds__(256) void k_trsm(DevSym S, const int *__restrict__ list, int kb,
                                              double *__restrict__ L) {
    __shared__ double D[NB * NB];
    __shared__ double R[NB * TRSM_ROWS];
    const int s = list[blockIdx.y];
    const int c = S.sfirst[s + 1] - S.sfirst[s];
    const int r = (int)(S.rowptr[s + 1] - S.rowptr[s]);
    if (kb >= c) return;
    const int w = min(NB, c - kb);
    const int row0 = kb + w + blockIdx.x * TRSM_ROWS;
    if (row0 >= r) return;
    const int nrow = min(TRSM_ROWS, r - row0);
    const int ld = S.ld[s];
    double *Pp = L + S.panelptr[s];
    const double *Dg = Pp + kb + (long long)kb * ld;
    double *A = Pp + row0 + (long long)kb * ld;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < w * w; idx += 256) {
        const int i = idx % w, j = idx / w;
        double v = 0.0;
        if (i >= j) v = Dg[i + (long long)j * ld];
        if (i == j) v = 1.0 / v;
        D[i + j * NB] = v;
    }
    for (int idx = tid; idx < w * TRSM_ROWS; idx += 256) {
        const int i = idx % TRSM_ROWS, k = idx / TRSM_ROWS;
        R[k * TRSM_ROWS + i] = (i < nrow) ? A[i + (long long)k * ld] : 0.0;
    }
    const int i = tid % TRSM_ROWS, g = tid / TRSM_ROWS;
    constexpr int G = 256 / TRSM_ROWS;
    for (int q = 0; q < w; q++) {
        __syncthreads();
        const double xq = R[q * TRSM_ROWS + i] * D[q + q * NB];
        for (int k = q + 1 + g; k < w; k += G) R[k * TRSM_ROWS + i] -= xq * D[k + q * NB];
    }
    __syncthreads();
    for (int idx = tid; idx < w * TRSM_ROWS; idx += 256) {
        const int ii = idx % TRSM_ROWS, k = idx / TRSM_ROWS;
        if (ii < nrow) A[ii + (long long)k * ld] = R[k * TRSM_ROWS + ii] * D[k + k * NB];
    }
}

// C[i,j] -= sum_k A[i,k] * B[j,k]  on 64x64 tiles (4 waves x 32x32), FP64 MFMA, operands read
// straight from HBM/L2. The MFMA is issued "transposed" (first operand = rows of B) so that
// the 16 lanes sharing a register index walk down a COLUMN of the column-major C.
// mode 0: trailing update inside the panel after block-column kb; mode 1: contribution block
// CB -= L21 L21' (K = all c columns).
__global__ __launch_bounds__(256) void k_gemm_nt(DevSym S, const int *__restrict__ list, int kb, int mode,
                                                 double *__restrict__ L, double *__restrict__ CB) {
    const int s = list[blockIdx.z];
    const int c = S.sfirst[s + 1] - S.sfirst[s];
    const int r = (int)(S.rowptr[s + 1] - S.rowptr[s]);
    const int ld = S.ld[s];
    double *P = L + S.panelptr[s];
    int M, N, K, ldc;
    const double *A;
    double *C;
    if (mode == 0) {
        if (kb + NB >= c) return;
        const int o = kb + NB;
        M = r - o; N = c - o; K = NB;
        A = P + o + (long long)kb * ld;
        C = P + o + (long long)o * ld;
        ldc = ld;
    } else {
        M = N = r - c; K = c;
        A = P + c;
        C = CB + S.cbptr[s];
        ldc = M;
    }
    const int bi = blockIdx.x, bj = blockIdx.y;
    if (bj > bi || bi * 64 >= M || bj * 64 >= N) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i0 = bi * 64 + (wave & 1) * 32, j0 = bj * 64 + (wave >> 1) * 32;
    if (i0 >= M || j0 >= N || j0 > i0 + 31) return;
    const int lm = lane & 15, lk = lane >> 4;
    d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int kk = k0 + lk;
        double av[2], bv[2];
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const int i = i0 + a * 16 + lm;
            av[a] = (i < M && kk < K) ? A[i + (long long)kk * ld] : 0.0;
        }
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int j = j0 + b * 16 + lm;
            bv[b] = (j < N && kk < K) ? A[j + (long long)kk * ld] : 0.0;
        }
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
                acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[b], av[a], acc[a][b], 0, 0, 0);
    }
    // D[m][n]: m (rows of the first operand = C's column) = lk + 4*reg, n = lm = C's row
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int i = i0 + a * 16 + lm;
                const int j = j0 + b * 16 + lk + 4 * rr;
                if (i < M && j < N && i >= j) C[i + (long long)j * ldc] -= acc[a][b][rr];
            }
}

// ------------------------------------------------------------------------------------------
// Triangular sweeps, X row-major (ldx doubles per row), nr <= 64 right-hand sides per pass
// ------------------------------------------------------------------------------------------

// Forward: add the children's update vectors into this front's own rows of X and into W_s.
__global__ __launch_bounds__(256) void k_fwd_assemble(DevSym S, const int *__restrict__ list, double *__restrict__ X,
                                                      double *__restrict__ W, int nr, int ldx) {
    const int s = list[blockIdx.y];
    const int c = S.sfirst[s + 1] - S.sfirst[s];
    const int r = (int)(S.rowptr[s + 1] - S.rowptr[s]);
    const int i0 = blockIdx.x * FWD_RB;
    if (i0 >= r) return;
    const int i1 = min(i0 + FWD_RB, r);
    const int first = S.sfirst[s];
    double *Ws = W + S.wptr[s] * ldx;
    const int tid = threadIdx.x;
    {
        const int a = max(i0, c);
        const int cnt = (i1 - a) * nr;
        for (int idx = tid; idx < cnt; idx += 256) {
            const int i = a + idx / nr, j = idx % nr;
            Ws[(long long)(i - c) * ldx + j] = 0.0;
        }
    }
    __syncthreads();
    for (long long ch = S.childptr[s]; ch < S.childptr[s + 1]; ch++) {
        const int d = S.children[ch];
        const int cd = S.sfirst[d + 1] - S.sfirst[d];
        const int md = (int)(S.rowptr[d + 1] - S.rowptr[d]) - cd;
        const int *reld = S.rel + S.rowptr[d] + cd;
        const double *Wd = W + S.wptr[d] * ldx;
        const int a0 = lower_bound_i32(reld, md, i0);
        const int a1 = lower_bound_i32(reld, md, i1);
        const int cnt = (a1 - a0) * nr;
        for (int idx = tid; idx < cnt; idx += 256) {
            const int a = a0 + idx / nr, j = idx % nr;
            const int ti = reld[a];
            const double v = Wd[(long long)a * ldx + j];
            if (ti < c) X[(long long)(first + ti) * ldx + j] += v;
            else Ws[(long long)(ti - c) * ldx + j] += v;
        }
        __syncthreads();
    }
}

// Solve with the diagonal block of block-column kb. trans = 0: D y = b (forward);
// trans = 1: D' x = y (backward). One workgroup per front.
__global__ __launch_bounds__(256) void k_solve_diag(DevSym S, const int *__restrict__ list, int kb, int trans,
                                                    const double *__restrict__ L, double *__restrict__ X, int nr,
                                                    int ldx) {
    __shared__ double D[NB * NB];
    __shared__ double Y[NB * 64];
    const int s = list[blockIdx.x];
    const int c = S.sfirst[s + 1] - S.sfirst[s];
    if (kb >= c) return;
    const int w = min(NB, c - kb);
    const int ld = S.ld[s];
    const double *Dg = L + S.panelptr[s] + kb + (long long)kb * ld;
    double *Xb = X + (long long)(S.sfirst[s] + kb) * ldx;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < w * w; idx += 256) {
        const int i = idx % w, j = idx / w;
        D[i + j * NB] = (i >= j) ? Dg[i + (long long)j * ld] : 0.0;
    }
    for (int idx = tid; idx < w * nr; idx += 256) {
        const int k = idx / nr, j = idx % nr;
        Y[k * 64 + j] = Xb[(long long)k * ldx + j];
    }
    int npad = 1;
    while (npad < nr) npad <<= 1;
    const int j = tid & (npad - 1), g = tid / npad, G = 256 / npad;
    if (!trans) {
        for (int k = 0; k < w; k++) {
            __syncthreads();
            if (j < nr) {
                const double yk = Y[k * 64 + j] / D[k + k * NB];
                for (int i = k + 1 + g; i < w; i += G) Y[i * 64 + j] -= D[i + k * NB] * yk;
            }
        }
    } else {
        for (int k = w - 1; k >= 0; k--) {
            __syncthreads();
            if (j < nr) {
                const double xk = Y[k * 64 + j] / D[k + k * NB];
                for (int i = g; i < k; i += G) Y[i * 64 + j] -= D[k + i * NB] * xk;
            }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < w * nr; idx += 256) {
        const int k = idx / nr, jj = idx % nr;
        Xb[(long long)k * ldx + jj] = Y[k * 64 + jj] / D[k + k * NB];
    }
}

// Forward update after block-column kb: T[i,:] -= L[i, kb:kb+w] * y  for the front rows below
// the block. T is X for the front's own rows and W_s for its trailing rows. One wave = 16 rows
// x up to 64 right-hand sides (4 MFMA tiles); the result tile has the right-hand-side index on
// the lanes, i.e. contiguous in the row-major X/W.
__global__ __launch_bounds__(256) void k_fwd_update(DevSym S, const int *__restrict__ list, int kb,
                                                    const double *__restrict__ L, double *__restrict__ X,
                                                    double *__restrict__ W, int nr, int ldx) {
    const int s = list[blockIdx.y];
    const int c = S.sfirst[s + 1] - S.sfirst[s];
    const int r = (int)(S.rowptr[s + 1] - S.rowptr[s]);
    if (kb >= c) return;
    const int w = min(NB, c - kb);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i0 = kb + w + (blockIdx.x * 4 + wave) * 16;
    if (i0 >= r) return;
    const int ld = S.ld[s];
    const int first = S.sfirst[s];
    const double *P = L + S.panelptr[s];
    const double *Yb = X + (long long)(first + kb) * ldx;
    double *Ws = W + S.wptr[s] * ldx;
    const int lm = lane & 15, lk = lane >> 4;
    const int nt = (nr + 15) >> 4;
    d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < w; k0 += 4) {
        const int kk = k0 + lk;
        const int i = i0 + lm;
        const double a = (i < r && kk < w) ? P[i + (long long)(kb + kk) * ld] : 0.0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (t < nt) {
                const int j = t * 16 + lm;
                const double b = (kk < w && j < nr) ? Yb[(long long)kk * ldx + j] : 0.0;
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
        if (t < nt) {
            const int j = t * 16 + lm;
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int i = i0 + lk + 4 * rr;
                if (i < r && j < nr) {
                    double *dst = (i < c) ? X + (long long)(first + i) * ldx + j : Ws + (long long)(i - c) * ldx + j;
                    *dst -= acc[t][rr];
                }
            }
        }
    }
}

// Backward update: X[own col i,:] -= sum_{q in [q0,q1)} L[q,i] * X[rows[q],:] for own columns
// i < ncols_out. mode 0: q over the trailing rows [c,r) (gathered through rows[]), outputs
// all c columns; mode 1: q over block-column kb's rows, outputs the columns left of it.
__global__ __launch_bounds__(256) void k_bwd_gemm(DevSym S, const int *__restrict__ list, int kb, int mode,
                                                  const double *__restrict__ L, double *__restrict__ X, int nr,
                                                  int ldx) {
    const int s = list[blockIdx.y];
    const int c = S.sfirst[s + 1] - S.sfirst[s];
    const int r = (int)(S.rowptr[s + 1] - S.rowptr[s]);
    int q0, q1, nout;
    if (mode == 0) { q0 = c; q1 = r; nout = c; }
    else { if (kb >= c) return; q0 = kb; q1 = min(kb + NB, c); nout = kb; }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i0 = (blockIdx.x * 4 + wave) * 16;
    if (i0 >= nout || q0 >= q1) return;
    const int ld = S.ld[s];
    const int first = S.sfirst[s];
    const double *P = L + S.panelptr[s];
    const int *rows = S.rows + S.rowptr[s];
    const int lm = lane & 15, lk = lane >> 4;
    const int nt = (nr + 15) >> 4;
    d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    for (int k0 = q0; k0 < q1; k0 += 4) {
        const int q = k0 + lk;
        const int col = i0 + lm;
        const double a = (q < q1 && col < nout) ? P[q + (long long)col * ld] : 0.0;
        const long long xr = (q < q1) ? rows[q] : 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (t < nt) {
                const int j = t * 16 + lm;
                const double b = (q < q1 && j < nr) ? X[xr * ldx + j] : 0.0;
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
        if (t < nt) {
            const int j = t * 16 + lm;
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int col = i0 + lk + 4 * rr;
                if (col < nout && j < nr) X[(long long)(first + col) * ldx + j] -= acc[t][rr];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Right-hand-side permutation + transposition (column-major caller layout <-> row-major X)
// ------------------------------------------------------------------------------------------
// dir 0: X[k, j] = B[perm[k] + j*ldb]   (perm == nullptr: identity)
// dir 1: B[perm[k] + j*ldb] = X[k, j]
__global__ __launch_bounds__(256) void k_permute(const int *__restrict__ perm, int n, double *__restrict__ Bc,
                                                 long long ldb, double *__restrict__ X, int nr, int ldx, int dir) {
    __shared__ double T[64 * 65];
    const int k0 = blockIdx.x * 64;
    const int tid = threadIdx.x;
    const int a = tid & 63, b = tid >> 6;
    if (dir == 0) {
        const int k = k0 + a;
        const long long src = (k < n) ? (perm ? perm[k] : k) : 0;
        for (int j = b; j < nr; j += 4) T[a * 65 + j] = (k < n) ? Bc[src + (long long)j * ldb] : 0.0;
        __syncthreads();
        for (int kk = b; kk < 64; kk += 4)
            if (k0 + kk < n && a < nr) X[(long long)(k0 + kk) * ldx + a] = T[kk * 65 + a];
    } else {
        for (int kk = b; kk < 64; kk += 4)
            if (k0 + kk < n && a < nr) T[kk * 65 + a] = X[(long long)(k0 + kk) * ldx + a];
        __syncthreads();
        const int k = k0 + a;
        if (k < n) {
            const long long dst = perm ? perm[k] : k;
            for (int j = b; j < nr; j += 4) Bc[dst + (long long)j * ldb] = T[a * 65 + j];
        }
    }
}

// ------------------------------------------------------------------------------------------
// log det Q = 2 sum_k log L_kk, fixed-order two-stage reduction (bit-reproducible)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_logdet_partial(const double *__restrict__ L, const long long *__restrict__ diagoff,
                                                        int n, double *__restrict__ part) {
    __shared__ double sh[256];
    const int tid = threadIdx.x;
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int k0 = blockIdx.x * per, k1 = min(n, k0 + per);
    double acc = 0.0;
    for (int k = k0 + tid; k < k1; k += 256) acc += log(L[diagoff[k]]);
    sh[tid] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) sh[tid] += sh[tid + st];
        __syncthreads();
    }
    if (tid == 0) part[blockIdx.x] = sh[0];
}
__global__ void k_logdet_final(const double *__restrict__ part, int nparts, double *__restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double acc = 0.0;
        for (int i = 0; i < nparts; i++) acc += part[i];
        out[0] = 2.0 * acc;
    }
}

// Gather values at precomputed offsets (-1 -> 0.0): selected-inverse extraction.
__global__ __launch_bounds__(256) void k_gather(const double *__restrict__ src, const long long *__restrict__ off,
                                                long long cnt, double *__restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < cnt) { const long long o = off[i]; out[i] = (o >= 0) ? src[o] : 0.0; }
}
__global__ __launch_bounds__(256) void k_gather_diag(const double *__restrict__ src, const long long *__restrict__ diagoff,
                                                     const int *__restrict__ perm, int n, double *__restrict__ out) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < n) out[perm[k]] = src[diagoff[k]];
}

// ------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

void launch_assemble(hipStream_t st, const DevSym &S, const int *list, int nfronts, int max_rows,
                     const double *nzval, double *L, double *CB) {
    if (nfronts <= 0) return;
    hipLaunchKernelGGL(k_assemble, dim3(cdiv(max_rows, ASM_CW), nfronts), dim3(256), 0, st, S, list, nzval, L, CB);
}
void launch_potrf(hipStream_t st, const DevSym &S, const int *list, int nactive, int kb, double *L, int *info) {
    if (nactive <= 0) return;
    hipLaunchKernelGGL(k_potrf, dim3(nactive), dim3(256), 0, st, S, list, kb, L, info);
}
void launch_trsm(hipStream_t st, const DevSym &S, const int *list, int nactive, int kb, int max_rows_below, double *L) {
    if (nactive <= 0 || max_rows_below <= 0) return;
    hipLaunchKernelGGL(k_trsm, dim3(cdiv(max_rows_below, TRSM_ROWS), nactive), dim3(256), 0, st, S, list, kb, L);
}
void launch_gemm_nt(hipStream_t st, const DevSym &S, const int *list, int nactive, int kb, int mode, int maxM,
                    int maxN, double *L, double *CB) {
    if (nactive <= 0 || maxM <= 0 || maxN <= 0) return;
    hipLaunchKernelGGL(k_gemm_nt, dim3(cdiv(maxM, 64), cdiv(maxN, 64), nactive), dim3(256), 0, st, S, list, kb, mode, L, CB);
}
void launch_fwd_assemble(hipStream_t st, const DevSym &S, const int *list, int nfronts, int max_rows, double *X,
                         double *W, int nr, int ldx) {
    if (nfronts <= 0) return;
    hipLaunchKernelGGL(k_fwd_assemble, dim3(cdiv(max_rows, FWD_RB), nfronts), dim3(256), 0, st, S, list, X, W, nr, ldx);
}
void launch_solve_diag(hipStream_t st, const DevSym &S, const int *list, int nactive, int kb, int trans,
                       const double *L, double *X, int nr, int ldx) {
    if (nactive <= 0) return;
    hipLaunchKernelGGL(k_solve_diag, dim3(nactive), dim3(256), 0, st, S, list, kb, trans, L, X, nr, ldx);
}
void launch_fwd_update(hipStream_t st, const DevSym &S, const int *list, int nactive, int kb, int max_rows_below,
                       const double *L, double *X, double *W, int nr, int ldx) {
    if (nactive <= 0 || max_rows_below <= 0) return;
    hipLaunchKernelGGL(k_fwd_update, dim3(cdiv(max_rows_below, 64), nactive), dim3(256), 0, st, S, list, kb, L, X, W, nr, ldx);
}
void launch_bwd_gemm(hipStream_t st, const DevSym &S, const int *list, int nactive, int kb, int mode, int max_out,
                     const double *L, double *X, int nr, int ldx) {
    if (nactive <= 0 || max_out <= 0) return;
    hipLaunchKernelGGL(k_bwd_gemm, dim3(cdiv(max_out, 64), nactive), dim3(256), 0, st, S, list, kb, mode, L, X, nr, ldx);
}
void launch_permute(hipStream_t st, const int *perm, int n, double *Bc, long long ldb, double *X, int nr, int ldx, int dir) {
    hipLaunchKernelGGL(k_permute, dim3(cdiv(n, 64)), dim3(256), 0, st, perm, n, Bc, ldb, X, nr, ldx, dir);
}
void launch_logdet(hipStream_t st, const double *L, const long long *diagoff, int n, double *part, int nparts, double *out) {
    hipLaunchKernelGGL(k_logdet_partial, dim3(nparts), dim3(256), 0, st, L, diagoff, n, part);
    hipLaunchKernelGGL(k_logdet_final, dim3(1), dim3(64), 0, st, part, nparts, out);
}
void launch_gather(hipStream_t st, const double *src, const long long *off, long long cnt, double *out) {
    if (cnt <= 0) return;
    hipLaunchKernelGGL(k_gather, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, src, off, cnt, out);
}
void launch_gather_diag(hipStream_t st, const double *src, const long long *diagoff, const int *perm, int n, double *out) {
    hipLaunchKernelGGL(k_gather_diag, dim3(cdiv(n, 256)), dim3(256), 0, st, src, diagoff, perm, n, out);
}

}  // namespace gmrfx
