// inverse.hip -- dense inverse X = L11^-1 of the diagonal block of every big front, and the
// sweep kernels that use it.
//
// Why: with 64-column blocks the triangular sweeps of the top separator fronts are a chain of
// ~c/64 dependent (diagonal solve, update) launch pairs per level and direction -- ~120 latency-
// bound steps of 30-40 us each on the 1M-node benchmark. L11 of a separator front is well
// conditioned (cond ~ 1e2 on SPDE precisions), so its explicit inverse turns the whole diagonal
// solve of a level into ONE triangular matrix product that parallelises over row tiles.
//
// Storage: X[k][q] (k > q) lives at P[q + k*ld], i.e. transposed in the strict upper triangle of
// the c x c diagonal block of the panel, which the factorisation never touches (the 64 x 64
// diagonal sub-blocks are already filled by k_potrf64); diag(X) = 1/diag(L) stays implicit.
//
// Construction by recursive doubling, B = 64, 128, 256, ...: for every aligned pair of blocks
//   [ A  0 ]^-1   [ A^-1           0   ]
//   [ Bm C ]    = [ -C^-1 Bm A^-1  C^-1 ]        T' = (Bm A^-1)'  (phase 1),  X10 = -C^-1 T (phase 2)
// all pairs of all fronts of a stage run in the same two launches (FP64 MFMA, 64x64 tiles).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels.h"

namespace gmrfx {

typedef gmrfx_d4 d4;

// One 64x64 output tile per workgroup (4 waves x 32x32). phase 1: T'[j][i] = sum_q Bm[i][q] Ainv[q][j];
// phase 2: X10[i][j] = - sum_q Cinv[i][q] T[q][j], written transposed into the upper triangle.
// blockIdx = (tile, pair, front).
__global__ __launch_bounds__(256) void k_inv_stage(DevSym S, const int *__restrict__ list, int B, int phase,
                                                   double *__restrict__ L, double *__restrict__ T,
                                                   const long long *__restrict__ toff) {
    const int s = list[blockIdx.z];
    const int c = S.sfirst[s + 1] - S.sfirst[s];
    const int o = 2 * B * blockIdx.y;          // first column of block A
    if (o + B >= c) return;                     // no C block
    const int nC = min(B, c - o - B);           // rows of C (and of Bm)
    const int ld = S.ld[s];
    double *P = L + S.panelptr[s];
    double *Tp = T + toff[blockIdx.z] + (long long)blockIdx.y * B * B;   // T' stored [j + i*B]
    const int ntj = B >> 6;                      // tiles along j (columns of A)
    const int ti = blockIdx.x / ntj, tj = blockIdx.x % ntj;
    if (ti * 64 >= nC) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lm = lane & 15, lk = lane >> 4;
    const int i0 = ti * 64 + (wave & 1) * 32, j0 = tj * 64 + (wave >> 1) * 32;
    if (i0 >= nC) return;
    d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
    // operand rows in pairs (kernels.h, wave_gemm_32x32_pm / _rr / _kr): tile a of the rows is i0 + 2 lm + a on the operand
    // side and i0 + 2 (lk + 4 rr) + a in the accumulators, tile b of the columns is j0 + 2 lm + b
    const int nlast = (nC - 1) & ~1;
    if (phase == 1) {
        // D[m][n]: m = i (rows of Bm), n = j.  A_mfma[m=i][q] = Bm[i][q] = P[(o+B+i) + (o+q)*ld]
        //                                      B_mfma[q][n=j] = Ainv[q][j] (q >= j) = X[o+q][o+j]
        const double *Ablk = P + o + (long long)o * ld;   // origin of block A inside the panel
        auto fa = [&](int i, int q) { return P[(o + B + min(i, nC - 1)) + (long long)(o + min(max(q, 0), B - 1)) * ld]; };
        auto fb = [&](int q, int j) { return xinv_elem(Ablk, ld, B, q, j); };
        // q >= j0 + 32 is below the diagonal of Ainv for every column j of this wave tile: plain elements
        // Ablk[j + q ld], pointer form; only the first k-steps need the masked accessor
        const int qs = min(j0 + 32, B);
        if (B > qs) wave_gemm_32x32_rr(acc, P + (o + B + min(i0 + 2 * lm, nlast)) + (long long)o * ld, ld, Ablk + j0 + 2 * lm, ld, qs, B, lk);
        wave_gemm_32x32_pm(acc, i0, j0, j0 & ~15, qs, fa, fb, lm, lk);
        // D[m = i][n = j] -> T'[j + i*B], j on the lanes (contiguous): 16 bytes per lane
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int i = i0 + 2 * (lk + 4 * rr) + a, j = j0 + 2 * lm;
                if (i < nC) *(gmrfx_d2u *)(Tp + j + (long long)i * B) = (gmrfx_d2u){acc[a][0][rr], acc[a][1][rr]};
            }
    } else {
        // D[m][n]: m = i (rows of C), n = j.  A_mfma[m=i][q] = Cinv[i][q] (q <= i) = X[o+B+i][o+B+q]
        //                                      B_mfma[q][n=j] = T[q][j] = T'[j + q*B]
        const double *Cblk = P + (o + B) + (long long)(o + B) * ld;
        auto fa = [&](int i, int q) { return xinv_elem(Cblk, ld, nC, i, q); };
        auto fb = [&](int q, int j) { return Tp[j + (long long)min(max(q, 0), nC - 1) * B]; };
        const int qhi = min(nC, i0 + 32);
        // q < i0 is left of the diagonal of Cinv for every row i of this wave tile: plain elements Cblk[q + i ld]
        // (contiguous along q: k pairs); T' rows contiguous along j (row pairs)
        const int qb = min(i0, nC) & ~7;
        int qd = 0;
        if (qb > 0)
            qd = wave_gemm_32x32_kr(acc, Cblk + (long long)min(i0 + 2 * lm, nC - 1) * ld, Cblk + (long long)min(i0 + 2 * lm + 1, nC - 1) * ld,
                                    Tp + j0 + 2 * lm, B, 0, qb, lk);
        wave_gemm_32x32_pm(acc, i0, j0, qd, qhi, fa, fb, lm, lk);
        // X10[i][j] = -acc, stored at upper (row o+j, col o+B+i): j on the lanes (contiguous)
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int i = i0 + 2 * (lk + 4 * rr) + a, j = j0 + 2 * lm;
                if (i < nC) *(gmrfx_d2u *)(P + (o + j) + (long long)(o + B + i) * ld) = (gmrfx_d2u){-acc[a][0][rr], -acc[a][1][rr]};
            }
    }
}

// Diagonal solve of a whole big front as a triangular product with X = L11^-1:
//   trans = 0: Y[k] = sum_{q <= k} X[k][q] b[q]        (forward)
//   trans = 1: Y[k] = sum_{q >= k} X[q][k] b[q]        (backward)
// b = rows first..first+c of Xin (row-major, ldx), result goes to the same rows of Xout (it cannot
// be written in place: other workgroups still need b). One wave = 16 rows x up to 64 RHS.
template <int NW>   // waves per workgroup splitting K; 8 for launches with about one workgroup per CU (kernels.h)
__global__ __launch_bounds__(64 * NW) void k_xmul(DevSym S, const int *__restrict__ list, int trans,
                                              const double *__restrict__ L, const double *__restrict__ Xin,
                                              double *__restrict__ Xout, int nr, int ldx, int blk, int cap) {
    // Fronts wider than `cap` columns only have the inverses of their cap x cap diagonal blocks (the recursive
    // doubling stops there: a full inverse costs O(c^3)); the sweeps then substitute block by block and this
    // kernel is called once per block `blk`: the block is treated as a front of its own.
    const int s = list[blockIdx.y];
    const int cfull = S.sfirst[s + 1] - S.sfirst[s];
    const int col0 = blk * cap;
    const int c = min(cap, cfull - col0);
    __shared__ double red[NW == 4 ? 3 * 64 * 16 : NW * 16 * 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k0 = blockIdx.x * 16;      // one 16-row tile per workgroup, the 4 waves split the K range
    if (k0 >= c) return;
    const int ld = S.ld[s];
    const int first = S.sfirst[s] + col0;
    const double *P = L + S.panelptr[s] + col0 + (long long)col0 * ld;
    const double *Bb = Xin + (long long)first * ldx;
    const int lm = lane & 15, lk = lane >> 4;
    d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    const int qlo = trans ? k0 : 0, qhi = trans ? c : min(c, k0 + 16);
    constexpr int KU = 4;
    // (Round 5, measured and dropped: plain entries of the upper triangle instead of the accessor away from the tile's own diagonal
    //  block, and the operands of batch n + 1 requested before the MFMAs of batch n -- 164 instead of ~110 VGPRs, forward / backward
    //  sweep 2.04 / 1.60 ms against 1.94 / 1.55 on the same box. And for passes of at most 16 right-hand sides: one wave per 16-row
    //  tile over the whole K range instead of the split-K workgroup, as the update kernels of such passes do -- 1-RHS solve 2.37 ->
    //  2.51 ms: here the K range is a chain of up to eight dependent batches, which four or eight waves shorten.)
#pragma unroll 1
    for (int q0 = qlo + wave * 4 * KU; q0 < qhi; q0 += NW * 4 * KU) {
        double av[KU], bv[KU][4];
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const int q = q0 + 4 * u + lk;
            av[u] = trans ? xinv_elem(P, ld, c, q, k0 + lm) : xinv_elem(P, ld, c, k0 + lm, q);
            // right-hand sides in pairs: column tile t is right-hand side 32 (t >> 1) + 2 lm + (t & 1), one 16-byte load
            // feeds two tiles (the pair behind the last right-hand side reads into the next row; never stored)
#pragma unroll
            for (int t2 = 0; t2 < 2; t2++) {
                const gmrfx_d2u y = *(const gmrfx_d2u *)(Bb + (long long)min(q, c - 1) * ldx + min(32 * t2 + 2 * lm, nr - 1));
                bv[u][2 * t2] = y.x; bv[u][2 * t2 + 1] = y.y;
            }
        }
#pragma unroll
        for (int u = 0; u < KU; u++)
#pragma unroll
            for (int t = 0; t < 4; t++)
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u][t], acc[t], 0, 0, 0);
    }
    // distributed split-K reduction: wave w ends up with the complete 16 x 16 tile of RHS block w
    if (NW == 4) {
        d4 (&acc1)[1][4] = reinterpret_cast<d4 (&)[1][4]>(acc);
        splitk_reduce4<1>(acc1, red, wave, lane);
    } else splitk_reduce_nw<NW>(acc, red, wave, lane);
    double *Yb = Xout + (long long)first * ldx;
#pragma unroll
    for (int t = 0; t < 4; t++)
        if (t == wave) {
            const int j = 32 * (t >> 1) + 2 * lm + (t & 1);
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int k = k0 + lk + 4 * rr;
                if (k < c && j < nr) Yb[(long long)k * ldx + j] = acc[t][rr];
            }
        }
}

// The same product for passes of at most 16 right-hand sides: ONE right-hand-side tile instead of four (a quarter of the MFMAs and of
// the partial tiles, 2 KB of LDS per wave instead of 8: twice the resident workgroups), the K range still split over the waves of
// the workgroup -- a one-wave form was measured slower (note above): partial sums in wave order, like the 64-column kernel.
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_xmul_narrow(DevSym S, const int *__restrict__ list, int trans, const double *__restrict__ L,
                                                         const double *__restrict__ Xin, double *__restrict__ Xout, int nr, int ldx, int blk, int cap) {
    { const int jt = 16 * blockIdx.z; Xin += jt; Xout += jt; nr = min(nr - jt, 16); }       // (round 6) blockIdx.z = 16-column tile of the right-hand sides
    const int s = list[blockIdx.y];
    const int cfull = S.sfirst[s + 1] - S.sfirst[s];
    const int col0 = blk * cap;
    const int c = min(cap, cfull - col0);
    __shared__ double red[NW * 4 * 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k0 = blockIdx.x * 16;
    if (k0 >= c) return;
    const int ld = S.ld[s];
    const int first = S.sfirst[s] + col0;
    const double *P = L + S.panelptr[s] + col0 + (long long)col0 * ld;
    const int lm = lane & 15, lk = lane >> 4;
    const double *Bb = Xin + (long long)first * ldx + min(lm, nr - 1);
    d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
    const int qlo = trans ? k0 : 0, qhi = trans ? c : min(c, k0 + 16);
    constexpr int KU = 4;
#pragma unroll 1
    for (int q0 = qlo + wave * 4 * KU; q0 < qhi; q0 += NW * 4 * KU) {
        double av[KU], bv[KU];
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const int q = q0 + 4 * u + lk;
            av[u] = trans ? xinv_elem(P, ld, c, q, k0 + lm) : xinv_elem(P, ld, c, k0 + lm, q);
            bv[u] = Bb[(long long)min(q, c - 1) * ldx];
        }
#pragma unroll
        for (int u = 0; u < KU; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int rr = 0; rr < 4; rr++) red[(wave * 4 + rr) * 64 + lane] = acc[rr];
    __syncthreads();
    if (wave == 0 && lm < nr) {
        double *Yb = Xout + (long long)first * ldx + lm;
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            double sum = red[rr * 64 + lane];
#pragma unroll
            for (int w = 1; w < NW; w++) sum += red[(w * 4 + rr) * 64 + lane];
            const int k = k0 + lk + 4 * rr;
            if (k < c) Yb[(long long)k * ldx] = sum;
        }
    }
}

// Xdst[own rows of the listed fronts] = Xsrc[same rows]
__global__ __launch_bounds__(256) void k_copy_own(DevSym S, const int *__restrict__ list,
                                                  const double *__restrict__ Xsrc, double *__restrict__ Xdst, int nr,
                                                  int ldx, int blk, int cap) {
    // own rows of block blk (the whole front when cap covers it)
    const int s = list[blockIdx.y];
    const int cfull = S.sfirst[s + 1] - S.sfirst[s];
    const int col0 = blk * cap;
    const int c = min(cap, cfull - col0);
    if (c <= 0) return;
    const long long base = (long long)(S.sfirst[s] + col0) * ldx;
    const long long cnt = (long long)c * ldx;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < cnt; i += (long long)gridDim.x * 256) {
        if ((int)(i % ldx) < nr) Xdst[base + i] = Xsrc[base + i];
    }
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

void launch_inv_stage(hipStream_t st, const DevSym &S, const int *list, int nactive, int B, int max_c, int phase,
                      double *L, double *T, const long long *toff) {
    if (nactive <= 0) return;
    const int npair = cdiv(max_c, 2 * B);
    const int ntile = (B / 64) * (B / 64);
    // odd y extent: most fronts only have pair 0, and with an even extent every (pair 0, front z)
    // workgroup would land on the same XCD (linear workgroup id mod 8)
    hipLaunchKernelGGL(k_inv_stage, dim3(ntile, npair | 1, nactive), dim3(256), 0, st, S, list, B, phase, L, T, toff);
}
void launch_xmul(hipStream_t st, const DevSym &S, const int *list, int nfronts, int max_c, int trans, const double *L,
                 const double *Xin, double *Xout, int nr, int ldx, int blk, int cap) {
    if (nfronts <= 0 || max_c <= 0) return;
    max_c = std::min(max_c - blk * cap, cap);      // width of block `blk` of the widest front
    if (max_c <= 0) return;
    if (nr <= (trans ? narrow_pass_max_bwd() : narrow_pass_max())) {          // one right-hand-side tile per workgroup, ceil(nr / 16) tiles in grid z
        const unsigned jt = (unsigned)cdiv(nr, 16);
        if ((long long)cdiv(max_c, 16) * nfronts <= 256)
            hipLaunchKernelGGL(k_xmul_narrow<8>, dim3((unsigned)(cdiv(max_c, 16) | 1), nfronts, jt), dim3(512), 0, st, S, list, trans, L, Xin, Xout, nr, ldx, blk, cap);
        else
            hipLaunchKernelGGL(k_xmul_narrow<4>, dim3((unsigned)(cdiv(max_c, 16) | 1), nfronts, jt), dim3(256), 0, st, S, list, trans, L, Xin, Xout, nr, ldx, blk, cap);
        return;
    }
    if ((long long)cdiv(max_c, 16) * nfronts <= 256)
        hipLaunchKernelGGL(k_xmul<8>, dim3((unsigned)(cdiv(max_c, 16) | 1), nfronts), dim3(512), 0, st, S, list, trans, L, Xin, Xout, nr, ldx, blk, cap);
    else
        hipLaunchKernelGGL(k_xmul<4>, dim3((unsigned)(cdiv(max_c, 16) | 1), nfronts), dim3(256), 0, st, S, list, trans, L, Xin, Xout, nr, ldx, blk, cap);
}
void launch_copy_own(hipStream_t st, const DevSym &S, const int *list, int nfronts, int max_c, const double *Xsrc,
                     double *Xdst, int nr, int ldx, int blk, int cap) {
    max_c = std::min(max_c - blk * cap, cap);
    if (nfronts <= 0 || max_c <= 0) return;
    const int nb = std::max(1, std::min(64, cdiv(max_c * ldx, 256)));
    hipLaunchKernelGGL(k_copy_own, dim3(nb, nfronts), dim3(256), 0, st, S, list, Xsrc, Xdst, nr, ldx, blk, cap);
}

}  // namespace gmrfx
