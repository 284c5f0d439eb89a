// selinv.hip -- Takahashi selected inversion on the supernodal structure, top-down.
// Replaces SelectedInversion.selinv(F) (src/workspace/backend.jl:226-257 in the reference).
//
// For a front s with factor panel [L11; L21] the "inverse front" is the symmetric r x r matrix
// Zf = Sigma[rows_s, rows_s]. Its trailing (r-c)x(r-c) part is a sub-matrix of the PARENT's
// inverse front (gathered through rel[]), its first c columns are computed block-column by
// block-column from the right:  with D the diagonal block at kb, "below" = rows after it,
//     Yh          = L[below, blk] D^-1
//     Z[below,blk] = - Zf[below, below] Yh
//     Z[blk,blk]   = D^-T D^-1 - Yh' Z[below, blk]
// i.e. the Takahashi recursion applied to NB-wide virtual supernodes (Rue & Held 2005, 2.4).
// Z panels share the layout of L; the trailing parts live in the contribution-block arena,
// which is free once the factorisation is done.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace gmrfx {

typedef double d4 __attribute__((ext_vector_type(4)));

// ZB_s[i,j] = Zf_parent(rel[i], rel[j]) for i >= j.
__global__ __launch_bounds__(256) void k_sel_gather(DevSym S, const int *__restrict__ list,
                                                    const double *__restrict__ Z, double *__restrict__ ZB) {
    const int s = list[blockIdx.y];
    const int p = S.sparent[s];
    if (p < 0) return;
    const int c = S.sfirst[s + 1] - S.sfirst[s];
    const int r = (int)(S.rowptr[s + 1] - S.rowptr[s]);
    const int m = r - c;
    const int j0 = blockIdx.x * 16;
    if (j0 >= m) return;
    const int j1 = min(j0 + 16, m);
    const int *rel = S.rel + S.rowptr[s] + c;
    const int cp = S.sfirst[p + 1] - S.sfirst[p];
    const int rp = (int)(S.rowptr[p + 1] - S.rowptr[p]);
    const int mp = rp - cp;
    const int ldp = S.ld[p];
    const double *Zp = Z + S.panelptr[p];
    const double *ZBp = ZB + S.cbptr[p];
    double *out = ZB + S.cbptr[s];
    for (int j = j0; j < j1; j++) {
        const int b = rel[j];
        for (int i = j + threadIdx.x; i < m; i += 256) {
            const int a = rel[i];
            const double v = (b < cp) ? Zp[a + (long long)b * ldp] : ZBp[(a - cp) + (long long)(b - cp) * mp];
            out[i + (long long)j * m] = v;
        }
    }
}

__device__ __forceinline__ double zf_sym(const double *Zp, const double *ZBs, int ld, int c, int m, int i, int q) {
    const int a = max(i, q), b = min(i, q);
    return (b < c) ? Zp[a + (long long)b * ld] : ZBs[(a - c) + (long long)(b - c) * m];
}

// Z[i, kb+k] = - sum_{q below} Zf(i,q) Yh[q,k]; one wave = 16 rows x up to 64 columns.
__global__ __launch_bounds__(256) void k_sel_symm(DevSym S, const int *__restrict__ list, int kb,
                                                  double *__restrict__ Z, const double *__restrict__ ZB,
                                                  const double *__restrict__ Yh, const long long *__restrict__ yoff) {
    const int s = list[blockIdx.y];
    const int c = S.sfirst[s + 1] - S.sfirst[s];
    const int r = (int)(S.rowptr[s + 1] - S.rowptr[s]);
    if (kb >= c) return;
    const int w = min(NB, c - kb);
    const int o = kb + w;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i0 = o + (blockIdx.x * 4 + wave) * 16;
    if (i0 >= r) return;
    const int ld = S.ld[s];
    const int m = r - c;
    double *Zp = Z + S.panelptr[s];
    const double *ZBs = ZB + S.cbptr[s];
    const double *Y = Yh + yoff[s];
    const int lm = lane & 15, lk = lane >> 4;
    const int nt = (w + 15) >> 4;
    d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    for (int q0 = o; q0 < r; q0 += 4) {
        const int q = q0 + lk;
        const int i = i0 + lm;
        // second operand B[k=q][n=i] = Zf(i, q)
        const double zb = (q < r && i < r) ? zf_sym(Zp, ZBs, ld, c, m, i, q) : 0.0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (t < nt) {
                const int k = t * 16 + lm;
                // first operand A[m=k][kk=q] = Yh[q, k]
                const double ya = (q < r && k < w) ? Y[q + (long long)k * r] : 0.0;
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ya, zb, acc[t], 0, 0, 0);
            }
        }
    }
    // D[m][n]: m = lk + 4*reg -> column k, n = lm -> row i
#pragma unroll
    for (int t = 0; t < 4; t++) {
        if (t < nt) {
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int k = t * 16 + lk + 4 * rr;
                const int i = i0 + lm;
                if (k < w && i < r) Zp[i + (long long)(kb + k) * ld] = -acc[t][rr];
            }
        }
    }
}

// Z[blk,blk] = D^-T D^-1 - Yh' Z[below,blk]; one workgroup per front.
__global__ __launch_bounds__(256) void k_sel_diag(DevSym S, const int *__restrict__ list, int kb,
                                                  const double *__restrict__ L, double *__restrict__ Z,
                                                  const double *__restrict__ Yh, const long long *__restrict__ yoff) {
    __shared__ double D[NB * NB];   // G = Yh' Znew
    __shared__ double T[NB * NB];   // T[j + i*NB] = (D^-1)[i][j]  (transposed inverse)
    const int s = list[blockIdx.x];
    const int c = S.sfirst[s + 1] - S.sfirst[s];
    const int r = (int)(S.rowptr[s + 1] - S.rowptr[s]);
    if (kb >= c) return;
    const int w = min(NB, c - kb);
    const int o = kb + w;
    const int ld = S.ld[s];
    const double *Dg = L + S.panelptr[s] + kb + (long long)kb * ld;
    double *Zp = Z + S.panelptr[s];
    const double *Y = Yh + yoff[s];
    const int tid = threadIdx.x;
    // T[j + i*NB] = Linv[i][j]: strict lower part is stored in the strict upper triangle of the
    // factor's diagonal block by k_potrf, the diagonal is the reciprocal of L's.
    for (int idx = tid; idx < NB * NB; idx += 256) {
        const int j = idx % NB, i = idx / NB;
        double v = 0.0;
        if (i < w && j < w) {
            if (j < i) v = Dg[j + (long long)i * ld];
            else if (j == i) v = 1.0 / Dg[i + (long long)i * ld];
        }
        T[idx] = v;
    }
    // G = Yh' * Znew (K = rows below), kept in registers; wave t owns tile-row t (16 x 64)
    const int wave = tid >> 6, lane = tid & 63;
    const int lm = lane & 15, lk = lane >> 4;
    const int a0 = wave * 16;
    d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    if (a0 < w) {
        for (int q0 = o; q0 < r; q0 += 4) {
            const int q = q0 + lk;
            const int ka = a0 + lm;
            const double ya = (q < r && ka < w) ? Y[q + (long long)ka * r] : 0.0;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int kbcol = t * 16 + lm;
                const double zb = (q < r && kbcol < w) ? Zp[q + (long long)(kb + kbcol) * ld] : 0.0;
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ya, zb, acc[t], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    // store G in LDS. D[m][n] of the MFMA -> G[a0 + lk + 4*rr][t*16 + lm]
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int rr = 0; rr < 4; rr++) D[(a0 + lk + 4 * rr) + (t * 16 + lm) * NB] = acc[t][rr];
    __syncthreads();
    // Z[a,b] = sum_{k >= a} Dinv[k,a] Dinv[k,b] - G[a,b], a >= b
    for (int idx = tid; idx < w * w; idx += 256) {
        const int a = idx % w, b = idx / w;
        if (a < b) continue;
        double v = 0.0;
        for (int k = a; k < w; k++) v += T[a + k * NB] * T[b + k * NB];
        Zp[(kb + a) + (long long)(kb + b) * ld] = v - D[a + b * NB];
    }
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

void launch_sel_gather(hipStream_t st, const DevSym &S, const int *list, int nfronts, int max_trail,
                       const double *Z, double *ZB) {
    if (nfronts <= 0 || max_trail <= 0) return;
    hipLaunchKernelGGL(k_sel_gather, dim3(cdiv(max_trail, 16), nfronts), dim3(256), 0, st, S, list, Z, ZB);
}
void launch_sel_symm(hipStream_t st, const DevSym &S, const int *list, int nactive, int kb, int max_rows_below,
                     double *Z, const double *ZB, const double *Yh, const long long *yoff) {
    if (nactive <= 0 || max_rows_below <= 0) return;
    hipLaunchKernelGGL(k_sel_symm, dim3(cdiv(max_rows_below, 64), nactive), dim3(256), 0, st, S, list, kb, Z, ZB, Yh, yoff);
}
void launch_sel_diag(hipStream_t st, const DevSym &S, const int *list, int nactive, int kb, const double *L,
                     double *Z, const double *Yh, const long long *yoff) {
    if (nactive <= 0) return;
    hipLaunchKernelGGL(k_sel_diag, dim3(nactive), dim3(256), 0, st, S, list, kb, L, Z, Yh, yoff);
}

}  // namespace gmrfx
