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

#include <cstdlib>

#include "kernels.h"

namespace gmrfx {

typedef gmrfx_d4 d4;

// ZB_s[i,j] = Zf_parent(rel[i], rel[j]) for i >= j.
__global__ __launch_bounds__(256) void k_sel_gather(const SelRec *__restrict__ recs, DevSym S, const int *__restrict__ list,
                                                    const double *__restrict__ Z, double *__restrict__ ZB) {
    // front -> ONE 64-byte record (its geometry and its parent's) instead of front -> parent -> two sets of index arrays
    const SelRec R = recs[list[blockIdx.y]];
    if (R.p < 0 || R.foreign) return;        // root; sharded: the block was gathered by the parent's owner and sent here
    const int m = R.m;
    const int j0 = blockIdx.x * 16;
    if (j0 >= m) return;
    const int j1 = min(j0 + 16, m);
    const int *rel = S.rel + R.rel;
    const int cp = R.cp, mp = R.mp, ldp = R.ldp;
    const double *Zp = Z + R.zp;
    const double *ZBp = ZB + R.zbp;
    double *out = ZB + R.out;
    // the 16 source columns of this tile (uniform per workgroup), then every thread walks the rows with the 16
    // loads of its row in flight at once (rows above the diagonal of the tile are clamped onto it and not stored)
    const double *src[16];
    int bb[16];
#pragma unroll
    for (int jj = 0; jj < 16; jj++) {
        const int b = rel[min(j0 + jj, m - 1)];
        bb[jj] = b;
        src[jj] = (b < cp) ? Zp + (long long)b * ldp : ZBp + (long long)(b - cp) * mp - cp;
    }
    for (int i = j0 + threadIdx.x; i < m; i += blockDim.x) {
        const int a = rel[i];
        double v[16];
#pragma unroll
        for (int jj = 0; jj < 16; jj++) v[jj] = src[jj][max(a, bb[jj])];
#pragma unroll
        for (int jj = 0; jj < 16; jj++)
            if (j0 + jj < j1 && i >= j0 + jj) out[i + (long long)(j0 + jj) * m] = v[jj];
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

// Z[blk,blk] = D^-T D^-1 - Yh' Z[below,blk]; one workgroup per front, wave t owns the 16 rows a = 16 t ..
// Both products run on the MFMA with the SAME accumulator layout (m = b, n = a: a on the lanes, so the final
// stores walk down a column of Z), G = Yh' Znew straight from HBM with batched clamped loads, D^-T D^-1 from
// the transposed inverse staged in LDS; no second LDS tile, no scalar dot products.
__global__ __launch_bounds__(256) void k_sel_diag(DevSym S, const int *__restrict__ list, int kb,
                                                  const double *__restrict__ L, double *__restrict__ Z,
                                                  const double *__restrict__ Yh, const long long *__restrict__ yoff) {
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
            else if (j == i) v = fast_rcp(Dg[i + (long long)i * ld]);
        }
        T[idx] = v;
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    const int lm = lane & 15, lk = lane >> 4;
    const int a0 = wave * 16;
    if (a0 >= w) return;
    d4 g[4], x[4];
#pragma unroll
    for (int t = 0; t < 4; t++) { g[t] = (d4){0.0, 0.0, 0.0, 0.0}; x[t] = (d4){0.0, 0.0, 0.0, 0.0}; }
    // G[a][b] = sum_{q below} Yh[q][a] Znew[q][b]; rows q clamped, the Yh operand carries the mask
    const int ka = min(a0 + lm, w - 1);
    const double am = (a0 + lm < w) ? 1.0 : 0.0;
    const double *py = Y + (long long)ka * r;
    const double *pz[4];
#pragma unroll
    for (int t = 0; t < 4; t++) pz[t] = Zp + (long long)(kb + min(t * 16 + lm, w - 1)) * ld;
    for (int q0 = o; q0 < r; q0 += 16) {
        double ya[4], zb[4][4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int q = q0 + 4 * u + lk;
            const int qc = min(q, r - 1);
            ya[u] = py[qc] * (q < r ? am : 0.0);
#pragma unroll
            for (int t = 0; t < 4; t++) zb[u][t] = pz[t][qc];
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int t = 0; t < 4; t++) g[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(zb[u][t], ya[u], g[t], 0, 0, 0);
    }
    // X[a][b] = sum_k Dinv[k][a] Dinv[k][b] (zero for k < a): k-steps from a0 on
    for (int k0 = a0; k0 < NB; k0 += 4) {
        const double ta = T[(a0 + lm) + (k0 + lk) * NB];
#pragma unroll
        for (int t = 0; t < 4; t++) x[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(T[(t * 16 + lm) + (k0 + lk) * NB], ta, x[t], 0, 0, 0);
    }
    // D[m = b][n = a]: b = 16 t + lk + 4 rr, a = a0 + lm
    const int a = a0 + lm;
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int bcol = t * 16 + lk + 4 * rr;
            if (a < w && bcol < w && a >= bcol) Zp[(kb + a) + (long long)(kb + bcol) * ld] = x[t][rr] - g[t][rr];
        }
}

// ------------------------------------------------------------------------------------------
// Big fronts: one Takahashi step for the WHOLE front through the dense inverse X = L11^-1
// (inverse.hip) -- three batched GEMM launches per level instead of 3 per 64-column block:
//   phase 0   Yt[k][i]  = sum_{q>=k} L21[i][q] X[q][k]            (Y = L21 X, stored transposed)
//   phase 1   Z21[i][k] = - sum_q Z22[i][q] Y[q][k]               (also kept transposed in Z21t)
//   phase 2   Z11[a][b] = sum_{k>=a} X[k][a] X[k][b] - sum_q Y[q][a] Z21[q][b],  a >= b
// 64x64 output tiles, 4 waves x 32x32, FP64 MFMA; operand orientations chosen so that all but
// one operand stream (the upper half of the symmetric Z22) are contiguous along the lanes.
// ------------------------------------------------------------------------------------------
// Workgroup tile 64 x 64 = 2 x 2 waves of 32 x 32. (Measured and removed in round 4, DESIGN.md section 3: 128 x 128 tiles of
// sixteen waves -- 14.9 vs 13.0 ms --, whole fronts dealt to one XCD each -- 14.0-14.3 ms. Round 5, the same deal only on levels with
// at least 8 / 16 / 32 / 64 / 256 / 1024 fronts, where it is balanced: 14.8 / 14.6 / 14.2 / 14.5 / 14.4 / 14.1 against 13.2 ms on the same box --
// a front's operand slabs are served faster by eight L2s than by one, at every level of the tree. Also round 5: one kernel per phase
// (phase 1 then needs 76 + 32 registers and runs four waves per SIMD instead of three): 13.3 -> 14.9-15.0 ms; two / one workgroups per
// CU (unused dynamic LDS): 14.8 / 20.4 ms. Three waves per SIMD of this tile shape is the optimum.)
// WT = 1 (one wave, a 32 x 32 tile) on levels whose fronts have at most 64 columns: with the 64 x 64 tile half of the four waves of such a
// front's workgroups find no column (with the narrower k_sel_gather workgroups of the same levels: selected inversion of cfg 3 13.3-13.5 ->
// 13.1 ms on the same box).
template <int WT>
__global__ __launch_bounds__(64 * WT * WT) void k_sel_dense(DevSym S, const int *__restrict__ list, int phase,
                                                   const double *__restrict__ L, double *__restrict__ Z,
                                                   const double *__restrict__ ZB, double *__restrict__ Yt,
                                                   double *__restrict__ Z21t, const long long *__restrict__ woff) {
    const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    const int s = list[bz];
    const int c = S.sfirst[s + 1] - S.sfirst[s];
    const int r = (int)(S.rowptr[s + 1] - S.rowptr[s]);
    const int m = r - c;
    const int ld = S.ld[s];
    const double *P = L + S.panelptr[s];
    double *Zp = Z + S.panelptr[s];
    const double *ZBs = ZB + S.cbptr[s];
    double *Y = Yt + woff[s];        // Yt[k + i*c]
    double *Zt = Z21t + woff[s] + (long long)m * c;   // Z21t[k + i*c], right behind Yt in the same slab
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lm = lane & 15, lk = lane >> 4;
    const int M = phase == 0 ? m : c;        // extent of the MFMA "m" index
    const int N = phase == 1 ? m : c;        // extent of the MFMA "n" index (on the lanes)
    if (m == 0 && phase < 2) return;
    const int bm = bx, bn = by;
    if (bm * (32 * WT) >= M || bn * (32 * WT) >= N) return;
    if (phase == 2 && bn < bm) return;       // a-tile >= b-tile only
    const int m0 = bm * (32 * WT) + (wave % WT) * 32, n0 = bn * (32 * WT) + (wave / WT) * 32;
    if (m0 >= M || n0 >= N) return;
    d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
    // Operand rows in PAIRS throughout (kernels.h, wave_gemm_32x32_pm / _rr / _rk): tile a of the "m" index is
    // m0 + 2 lm + a on the operand side and m0 + 2 (lk + 4 rr) + a in the accumulators, tile b of the "n" index is
    // n0 + 2 lm + b -- 16-byte loads and stores, half the vector memory instructions. Lanes past the last row re-read
    // the last pair; their results are never stored.
    const int mlast = max(m - 1, 0) & ~1, clast = (c - 1) & ~1;
    if (phase == 0) {
        // m = i (row of L21), n = k
        auto fa = [&](int i, int q) { return P[(c + min(i, m - 1)) + (long long)min(max(q, 0), c - 1) * ld]; };
        auto fb = [&](int q, int k) { return xinv_elem(P, ld, c, q, k); };
        // q < n0 + 32 touches the diagonal of X for this tile's columns: masked accessors; beyond it every
        // X[q][k] (k < q) is a plain element P[k + q ld]: pointer form (rows / columns clamped at the edges only)
        const int qs = min((n0 + 32 + 3) & ~3, c), qe = c & ~3;
        wave_gemm_32x32_pm(acc, m0, n0, n0, qs, fa, fb, lm, lk);
        if (qe > qs) wave_gemm_32x32_rr(acc, P + c + min(m0 + 2 * lm, mlast), ld, P + min(n0 + 2 * lm, clast), ld, qs, qe, lk);
        if (c > max(qe, qs)) wave_gemm_32x32_pm(acc, m0, n0, max(qe, qs), c, fa, fb, lm, lk);
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int i = m0 + 2 * (lk + 4 * rr) + a, k = n0 + 2 * lm;
                if (i < m && k < c) {
                    double *dst = Y + k + (long long)i * c;
                    if (k + 1 < c) *(gmrfx_d2u *)dst = (gmrfx_d2u){acc[a][0][rr], acc[a][1][rr]};
                    else dst[0] = acc[a][0][rr];
                }
            }
    } else if (phase == 1) {
        // m = k, n = i
        auto fa = [&](int k, int q) { return Y[min(k, c - 1) + (long long)min(max(q, 0), m - 1) * c]; };
        auto fb = [&](int q, int i) {
            const int qq = min(max(q, 0), m - 1), ii = min(i, m - 1);
            return ZBs[max(ii, qq) + (long long)min(ii, qq) * m];
        };
        // Z22 is symmetric with only its lower triangle stored: q below the tile's rows i reads ZB[i + q m]
        // (contiguous along the lanes: row pairs), q above them ZB[q + i m] (contiguous along q: k pairs); the 32 q's
        // on the tile's own diagonal block go through the accessor
        const int ql = min(n0 & ~3, m), qh = min((n0 + 32 + 3) & ~3, m);
        const double *pa2 = Y + min(m0 + 2 * lm, clast);
        if (ql > 0) wave_gemm_32x32_rr(acc, pa2, c, ZBs + min(n0 + 2 * lm, mlast), m, 0, ql, lk);
        wave_gemm_32x32_pm(acc, m0, n0, ql, qh, fa, fb, lm, lk);
        int qd = qh;
        if (m - qh >= 8)
            qd = wave_gemm_32x32_rk(acc, pa2, c, ZBs + (long long)min(n0 + 2 * lm, m - 1) * m,
                                    ZBs + (long long)min(n0 + 2 * lm + 1, m - 1) * m, qh, m, lk);
        if (m > qd) wave_gemm_32x32_pm(acc, m0, n0, qd, m, fa, fb, lm, lk);
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int k = m0 + 2 * (lk + 4 * rr) + a, i = n0 + 2 * lm;
                if (k < c && i < m) {
                    double *dst = Zp + (c + i) + (long long)k * ld;
                    Zt[k + (long long)i * c] = -acc[a][0][rr];
                    if (i + 1 < m) {
                        *(gmrfx_d2u *)dst = (gmrfx_d2u){-acc[a][0][rr], -acc[a][1][rr]};
                        Zt[k + (long long)(i + 1) * c] = -acc[a][1][rr];
                    } else dst[0] = -acc[a][0][rr];
                }
            }
    } else {
        // m = b, n = a (a >= b)
        auto fa1 = [&](int b, int k) { return xinv_elem(P, ld, c, k, b); };
        auto fb1 = [&](int k, int a) { return xinv_elem(P, ld, c, k, a); };
        // a >= b tiles only (n0 >= m0): k < n0 + 32 touches the diagonal of X for the tile's a's; beyond it both
        // X[k][b] and X[k][a] are plain elements of the upper triangle
        const int ks = min((n0 + 32 + 3) & ~3, c), ke = c & ~3;
        wave_gemm_32x32_pm(acc, m0, n0, n0, ks, fa1, fb1, lm, lk);
        if (ke > ks) wave_gemm_32x32_rr(acc, P + min(m0 + 2 * lm, clast), ld, P + min(n0 + 2 * lm, clast), ld, ks, ke, lk);
        if (c > max(ke, ks)) wave_gemm_32x32_pm(acc, m0, n0, max(ke, ks), c, fa1, fb1, lm, lk);
        if (m > 0) {
            auto fa2 = [&](int b, int q) { return -Zt[min(b, c - 1) + (long long)min(max(q, 0), m - 1) * c]; };
            auto fb2 = [&](int q, int a) { return Y[min(a, c - 1) + (long long)min(max(q, 0), m - 1) * c]; };
            // no masks at all in this product: pointer form over the whole q range (its sign is applied at the
            // end: the partial sums of the two products are kept apart in two accumulators)
            d4 acc2[2][2];
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++) acc2[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
            const int me = m & ~3;
            wave_gemm_32x32_rr(acc2, Zt + min(m0 + 2 * lm, clast), c, Y + min(n0 + 2 * lm, clast), c, 0, me, lk);
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++) acc[a][b] -= acc2[a][b];
            if (m > me) wave_gemm_32x32_pm(acc, m0, n0, me, m, fa2, fb2, lm, lk);
        }
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int bb = m0 + 2 * (lk + 4 * rr) + a, aa = n0 + 2 * lm;
                if (bb < c) {
                    double *dst = Zp + aa + (long long)bb * ld;
                    const bool v0 = aa < c && aa >= bb, v1 = aa + 1 < c && aa + 1 >= bb;
                    if (v0 && v1) *(gmrfx_d2u *)dst = (gmrfx_d2u){acc[a][0][rr], acc[a][1][rr]};
                    else {
                        if (v0) dst[0] = acc[a][0][rr];
                        if (v1) dst[1] = acc[a][1][rr];
                    }
                }
            }
    }
}

// Phase 1 of the big fronts on 128 x 128 tiles staged through LDS (round 5). The 64 x 64 direct-operand tiles of k_sel_dense
// re-read Y and Z22 once per tile: on levels 9-16 of cfg 3 the L2 missed 20-38 x the minimal bytes (profiles/r04_cfg3_selinv_levels.txt)
// and the phase ran at 34-41 TFLOP/s where a front's operands (15 MB at level 13) no longer fit the XCD's L2. Here eight waves
// share a 128 (k) x 128 (i) tile, the operands come in 16-deep slabs through a double-buffered LDS stage -- a quarter of the
// operand bytes per flop -- and the symmetric Z22 is read from whichever stored side is contiguous: rows q above the tile's
// i range as ZB[i + q m] (along the lanes), below it as ZB[q + i m] (four consecutive q per thread), across it element-wise.
__global__ __launch_bounds__(512) void k_sel_z21_big(DevSym S, const int *__restrict__ list, double *__restrict__ Z, const double *__restrict__ ZB,
                                                     const double *__restrict__ Yt, double *__restrict__ Z21t, const long long *__restrict__ woff) {
    constexpr int TM = 128, KB = 16;
    __shared__ double As[2][KB][TM + 8], Bs[2][KB][TM + 8];
    const int s = list[blockIdx.z];
    const int c = S.sfirst[s + 1] - S.sfirst[s];
    const int r = (int)(S.rowptr[s + 1] - S.rowptr[s]);
    const int m = r - c;
    const int k0t = blockIdx.x * TM, i0t = blockIdx.y * TM;
    if (m <= 0 || k0t >= c || i0t >= m) return;
    const int ld = S.ld[s];
    double *Zp = Z + S.panelptr[s];
    const double *ZBs = ZB + S.cbptr[s];
    const double *Y = Yt + woff[s];
    double *Zt = Z21t + woff[s] + (long long)m * c;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lm = lane & 15, lk = lane >> 4;
    const int wi = (wave & 3) * 32, wj = (wave >> 2) * 64;       // wave sub-tile: 32 k's x 64 i's
    d4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
    const int lr = tid & 127, l4 = (tid >> 7) * 4;
    const int ka = min(k0t + lr, c - 1);            // this thread's k (operand A: Y[k + q c]) ...
    const int ib = min(i0t + lr, m - 1);            // ... and i (operand B: Z22[q][i]) of the staged slabs
    double ra[4], rb[4];
    auto fetch = [&](int qb) {
        const int q0 = qb * KB + l4;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int q = q0 + u, qc = min(q, m - 1);
            ra[u] = Y[ka + (long long)qc * c] * (q < m ? 1.0 : 0.0);
        }
        if (q0 + 3 < i0t) {                      // above the tile's rows: ZB[i + q m], contiguous along the threads
#pragma unroll
            for (int u = 0; u < 4; u++) rb[u] = ZBs[ib + (long long)(q0 + u) * m];
        } else if (q0 >= i0t + TM && q0 + 3 < m) {         // below: ZB[q + i m], four consecutive q
#pragma unroll
            for (int u = 0; u < 4; u++) rb[u] = ZBs[(q0 + u) + (long long)ib * m];
        } else {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int qc = min(q0 + u, m - 1);
                rb[u] = ZBs[max(ib, qc) + (long long)min(ib, qc) * m];
            }
        }
    };
    fetch(0);
#pragma unroll
    for (int u = 0; u < 4; u++) { As[0][l4 + u][lr] = ra[u]; Bs[0][l4 + u][lr] = rb[u]; }
    __syncthreads();
    const int nq = (m + KB - 1) / KB;
    for (int qb = 0; qb < nq; qb++) {
        const int cur = qb & 1;
        if (qb + 1 < nq) fetch(qb + 1);
#pragma unroll
        for (int sidx = 0; sidx < KB / 4; sidx++) {
            double av[2], bv[4];
#pragma unroll
            for (int a = 0; a < 2; a++) av[a] = As[cur][4 * sidx + lk][wi + 16 * a + lm];
#pragma unroll
            for (int b = 0; b < 4; b++) bv[b] = Bs[cur][4 * sidx + lk][wj + 16 * b + lm];
            // D[i][k]: first operand = the i's (register index walks them), second = the k's (the lanes walk them)
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[b], av[a], acc[a][b], 0, 0, 0);
        }
        if (qb + 1 < nq) {
#pragma unroll
            for (int u = 0; u < 4; u++) { As[cur ^ 1][l4 + u][lr] = ra[u]; Bs[cur ^ 1][l4 + u][lr] = rb[u]; }
        }
        __syncthreads();
    }
    // Z21[i][k] = -acc: into the panel of Z (rows c + i) and, transposed, behind Yt for phase 2
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int k = k0t + wi + 16 * a + lm;
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int i = i0t + wj + 16 * b + lk + 4 * rr;
                if (k < c && i < m) {
                    const double v = -acc[a][b][rr];
                    Zp[(c + i) + (long long)k * ld] = v;
                    Zt[k + (long long)i * c] = v;
                }
            }
        }
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

void launch_sel_gather(hipStream_t st, const SelRec *recs, const DevSym &S, const int *list, int nfronts, int max_trail,
                       const double *Z, double *ZB) {
    if (nfronts <= 0 || max_trail <= 0) return;
    // (a thread per row of a 16-column strip: levels whose fronts have at most 64 / 128 trailing rows get one / two waves per strip
    //  instead of four, three of which would find no row)
    const unsigned nthr = max_trail <= 64 ? 64 : max_trail <= 128 ? 128 : 256;
    hipLaunchKernelGGL(k_sel_gather, dim3((unsigned)(cdiv(max_trail, 16) | 1), nfronts), dim3(nthr), 0, st, recs, S, list, Z, ZB);
}
void launch_sel_symm(hipStream_t st, const DevSym &S, const int *list, int nactive, int kb, int max_rows_below,
                     double *Z, const double *ZB, const double *Yh, const long long *yoff) {
    if (nactive <= 0 || max_rows_below <= 0) return;
    hipLaunchKernelGGL(k_sel_symm, dim3((unsigned)(cdiv(max_rows_below, 64) | 1), nactive), dim3(256), 0, st, S, list, kb, Z, ZB, Yh, yoff);
}
void launch_sel_diag(hipStream_t st, const DevSym &S, const int *list, int nactive, int kb, const double *L,
                     double *Z, const double *Yh, const long long *yoff) {
    if (nactive <= 0) return;
    hipLaunchKernelGGL(k_sel_diag, dim3(nactive), dim3(256), 0, st, S, list, kb, L, Z, Yh, yoff);
}

// (Round 5, measured and removed: phase 1 of the big fronts -- Z21' = -Y Z22, two thirds to four fifths of the flops on the top
//  levels -- with a 64 x 64 WAVE tile, 4 x 4 MFMA tiles per wave, every operand value re-used four times from registers (one 16-byte
//  load per four MFMAs instead of two; rows 4 lm + a / columns 4 lm + b so that a lane's operands and results are 32 contiguous
//  bytes), 218 VGPRs = two waves per SIMD. Correct on the first run (36 selected-inversion tests incl. cfg 3 / cfg 4 at full size);
//  L2-miss traffic of the selected inversion 61 -> 52 GB, the phase itself 41-50 TFLOP/s -- the same as the 32 x 32 tile it replaced
//  (levels 11-16: 4.07 ms against 3.9), selected inversion 13.0 -> 13.3 ms. The dense phases are not bound by their re-reads: they sit
//  at the 49-55 TFLOP/s this chip gives a direct-operand FP64 MFMA product at two to three waves per SIMD.)
void launch_sel_dense(hipStream_t st, const DevSym &S, const int *list, int nfronts, int phase, int max_c, int max_trail,
                      const double *L, double *Z, const double *ZB, double *Yt, double *Z21t, const long long *woff) {
    if (nfronts <= 0) return;
    const int M = phase == 0 ? max_trail : max_c, N = phase == 1 ? max_trail : max_c;
    if (M <= 0 || N <= 0) return;
    // Fronts of 3-D problems only (>= 1024 columns over >= 1024 rows): a 72^3-node 3-D SPDE went from 273 to 250 ms with it; on the 2-D
    // cfg 3, whose widest levels are 2-4 fronts of 1000 columns, the 128 x 128 tiles are too few to fill 256 CUs (13.9 -> 14.5 ms with
    // a 128-column threshold) -- tools/ab_selinv.py.
    if (phase == 1 && max_c >= 1024 && max_trail >= 1024) {
        hipLaunchKernelGGL(k_sel_z21_big, dim3((unsigned)(cdiv(max_c, 128) | 1), (unsigned)(cdiv(max_trail, 128) | 1), nfronts), dim3(512), 0, st,
                           S, list, Z, ZB, Yt, Z21t, woff);
        return;
    }
    if (max_c <= 64) {
        const int gx = cdiv(M, 32), gy = cdiv(N, 32);
        hipLaunchKernelGGL(k_sel_dense<1>, dim3((unsigned)(gx | 1), (unsigned)(gy | 1), nfronts), dim3(64), 0, st, S, list, phase, L, Z, ZB, Yt, Z21t, woff);
        return;
    }
    const int gx = cdiv(M, 64), gy = cdiv(N, 64);
    hipLaunchKernelGGL(k_sel_dense<2>, dim3((unsigned)(gx | 1), (unsigned)(gy | 1), nfronts), dim3(256), 0, st, S, list, phase, L, Z, ZB, Yt, Z21t, woff);
}

}  // namespace gmrfx
