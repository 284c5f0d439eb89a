// small.hip -- fused kernels for SMALL fronts (r <= RMAX rows, c <= 64 columns): one workgroup
// owns a whole front, which lives in LDS from assembly to write-out, so a level of tens of
// thousands of tiny supernodes costs one launch instead of ~6 and its panel / contribution block
// are written to HBM exactly once. ~95 % of the supernodes of a 2-D SPDE precision are small.
//
// Factorisation of one front F (r x r, lower) in LDS, 4 columns (one MFMA k-step) at a time:
//   panel   thread i solves its row against the 4x4 diagonal block (every thread refactors that
//           4x4 block redundantly from LDS: 10 values, ~30 flops)
//   inverse threads 128.. carry the block forward substitution  L X = I  along: rows of the
//           current panel of X = L11^-1 become final (W), rows below get M -= L[i,p] W.  M / X
//           live in the strict UPPER triangle of F's c x c block, transposed -- exactly where the
//           HBM panel stores (L11^-1)' for the solve kernels (see k_potrf in kernels.hip)
//   update  F[i,k] -= sum_q P[q][i] P[q][k] on 16x16 tiles with v_mfma_f64_16x16x4_f64, tiles
//           read-modify-written in LDS (leading dimension = 2 mod 32 doubles: conflict-free)
#include <hip/hip_runtime.h>

#include <climits>

#include "kernels.h"

namespace gmrfx {

typedef double d4 __attribute__((ext_vector_type(4)));


constexpr int lds_ldf(int R) { return R + 2; }                                   // = 2 (mod 4), odd half: conflict-free tile RMW
constexpr int lds_ldp(int R) { return (R + 16) % 32 == 16 ? R + 16 : R + 32; }   // = 16 (mod 32): conflict-free operand reads
constexpr int LDW = 80;                                                          // 64 columns, = 16 (mod 32)

// Partial Cholesky (first c columns) of the r x r front held in LDS (F, column-major, leading
// dimension lds_ldf(RMAX)), 4 columns per step, together with X = L11^-1 (stored transposed in
// the strict upper triangle of F's c x c block). Must be called by all 256 threads.
template <int RMAX>
__device__ __forceinline__ void lds_partial_cholesky(double *F, double *Pn, double *Wn, const int c, const int r,
                                                      const int first, int *__restrict__ info,
                                                      double *__restrict__ Pg, const int ldg) {
    // Pg/ldg: the front's panel in HBM. L values and the rows of X = L11^-1 are written there AS
    // SOON AS THEY ARE FINAL (coalesced), not kept in F: that removes one of the three barriers of
    // a panel step (nothing is written to F between the reads of the step and its tile updates)
    constexpr int LDF = lds_ldf(RMAX);
    constexpr int LDP = lds_ldp(RMAX);
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int lm = lane & 15, lk = lane >> 4;
    for (int j0 = 0; j0 < c; j0 += 4) {
        const int nbk = min(4, c - j0);
        // 4x4 diagonal block (identity-padded) -> its Cholesky factor, redundantly per thread
        double dd[4][4];
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b <= a; b++)
                dd[a][b] = (a < nbk) ? F[(j0 + b) * LDF + j0 + a] : ((a == b) ? 1.0 : 0.0);
        // reciprocal square roots of the pivots by v_rsq_f64 + two Newton steps (full double
        // precision, ~10 dependent FMAs) instead of sqrt + division (~2 x 15-instruction
        // sequences): this 4x4 chain sits on the critical path of every panel step
        auto rsqrt_nr = [](double p) {
            double y = __builtin_amdgcn_rsq(p);
            y = y * (1.5 - 0.5 * p * y * y);
            y = y * (1.5 - 0.5 * p * y * y);
            return y;
        };
        const double p0 = dd[0][0];
        const double i00 = rsqrt_nr(p0);
        const double l10 = dd[1][0] * i00, l20 = dd[2][0] * i00, l30 = dd[3][0] * i00;
        const double p1 = dd[1][1] - l10 * l10;
        const double i11 = rsqrt_nr(p1);
        const double l21 = (dd[2][1] - l20 * l10) * i11, l31 = (dd[3][1] - l30 * l10) * i11;
        const double p2 = dd[2][2] - l20 * l20 - l21 * l21;
        const double i22 = rsqrt_nr(p2);
        const double l32 = (dd[3][2] - l30 * l20 - l31 * l21) * i22;
        const double p3 = dd[3][3] - l30 * l30 - l31 * l31 - l32 * l32;
        const double i33 = rsqrt_nr(p3);
        if (tid == 0) {
            int bad = -1;
            if (!(p3 > 0.0) && nbk > 3) bad = 3;
            if (!(p2 > 0.0) && nbk > 2) bad = 2;
            if (!(p1 > 0.0) && nbk > 1) bad = 1;
            if (!(p0 > 0.0)) bad = 0;
            if (bad >= 0) atomicMin(info, first + j0 + bad);
        }
        // own work item of this phase, read before anything is overwritten
        double x[4] = {0.0, 0.0, 0.0, 0.0};
        bool row_active = false;
        double wv[4] = {0.0, 0.0, 0.0, 0.0};
        bool w_active = false;
        if (tid < RMAX) {
            const int i = tid;
            if (i >= j0 && i < r) {
                row_active = true;
                double p[4];
#pragma unroll
                for (int k = 0; k < 4; k++) p[k] = (k < nbk) ? F[(j0 + k) * LDF + i] : 0.0;
                const int a = i - j0;   // position inside the diagonal block if < 4
                x[0] = p[0] * i00;
                x[1] = (a >= 1) ? (p[1] - x[0] * l10) * i11 : 0.0;
                x[2] = (a >= 2) ? (p[2] - x[0] * l20 - x[1] * l21) * i22 : 0.0;
                x[3] = (a >= 3) ? (p[3] - x[0] * l30 - x[1] * l31 - x[2] * l32) * i33 : 0.0;
#pragma unroll
                for (int k = 0; k < 4; k++) if (k >= nbk) x[k] = 0.0;
            }
        } else if (tid >= 128 && tid < 128 + 64) {
            // column b of the current 4 rows of X = L11^-1:  W = Lpp^-1 * M_p
            const int b = tid - 128;
            if (b < j0 + nbk) {
                w_active = true;
                double mv[4];
#pragma unroll
                for (int a = 0; a < 4; a++) {
                    const int row = j0 + a;   // row of X / M
                    double v = 0.0;
                    if (a < nbk) {
                        if (b < j0) v = F[row * LDF + b];        // M[row][b] lives at F(b, row)
                        else if (b == row) v = 1.0;
                    }
                    mv[a] = v;
                }
                wv[0] = mv[0] * i00;
                wv[1] = (mv[1] - l10 * wv[0]) * i11;
                wv[2] = (mv[2] - l20 * wv[0] - l21 * wv[1]) * i22;
                wv[3] = (mv[3] - l30 * wv[0] - l31 * wv[1] - l32 * wv[2]) * i33;
#pragma unroll
                for (int a = 0; a < 4; a++) if (a >= nbk || b > j0 + a) wv[a] = 0.0;
            }
        }
        // Pn / Wn were last read by the previous step's tile updates (a barrier ago) and F is not
        // written in this phase: no barrier needed before publishing the panel
        if (tid < RMAX) {
            const int i = tid;
#pragma unroll
            for (int k = 0; k < 4; k++) Pn[k * LDP + i] = x[k];
            if (row_active) {
#pragma unroll
                for (int k = 0; k < 4; k++) if (k < nbk && i >= j0 + k) Pg[i + (long long)(j0 + k) * ldg] = x[k];
            }
        } else if (tid >= 128 && tid < 128 + 64) {
            const int b = tid - 128;
#pragma unroll
            for (int a = 0; a < 4; a++) Wn[a * LDW + b] = wv[a];
            if (w_active) {
#pragma unroll
                for (int a = 0; a < 4; a++) if (a < nbk && b < j0 + a) Pg[b + (long long)(j0 + a) * ldg] = wv[a];   // X[j0+a][b]
            }
        }
        __syncthreads();
        // ---- rank-4 updates on 16x16 tiles ------------------------------------------------------
        const int jn = j0 + nbk;                 // first non-final column
        if (jn < r) {
            const int tlo = jn >> 4, thi = (r - 1) >> 4;
            const int nt = thi - tlo + 1;
            const int ntile = nt * (nt + 1) / 2;
            // (a) trailing part of F: tiles (ti >= tj), element (i,k): i >= k >= jn
            for (int t = wave; t < ntile; t += 4) {
                int ti = 0, acc0 = 0;
                while (acc0 + ti + 1 <= t) { acc0 += ti + 1; ti++; }
                const int tj = t - acc0;
                const int ib = (tlo + ti) << 4, kbb = (tlo + tj) << 4;
                // D[m][n]: m = row (lk + 4 reg), n = column (lm)
                const int col = kbb + lm;
                d4 cc;
#pragma unroll
                for (int rr = 0; rr < 4; rr++) cc[rr] = F[col * LDF + ib + lk + 4 * rr];
                const double av = -Pn[lk * LDP + ib + lm];                       // A[m = row][k]
                const double bv = (col >= jn) ? Pn[lk * LDP + col] : 0.0;      // B[k][n = col], final columns masked
                cc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, cc, 0, 0, 0);
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const int row = ib + lk + 4 * rr;
                    if (row >= col) F[col * LDF + row] = cc[rr];                // lower part only
                }
            }
            // (b) M[i][b] -= sum_q L[i][j0+q] W[q][b] for jn <= i < c, b < jn; stored at F(b, i)
            if (jn < c) {
                const int tihi = (c - 1) >> 4;                 // column tiles (index i) tlo..tihi
                const int tbhi = (jn - 1) >> 4;                // row tiles (index b) 0..tbhi
                const int nti = tihi - tlo + 1, ntb = tbhi + 1;
                for (int t = wave; t < nti * ntb; t += 4) {
                    const int ti = tlo + t / ntb, tb = t % ntb;
                    if (tb > ti) continue;
                    const int icol = (ti << 4) + lm;           // F column = X row index i
                    d4 cc;
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) cc[rr] = F[icol * LDF + (tb << 4) + lk + 4 * rr];
                    const double av = -Wn[lk * LDW + (tb << 4) + lm];                       // A[m = b][k]
                    const double bv = (icol >= jn && icol < c) ? Pn[lk * LDP + icol] : 0.0; // B[k][n = i]
                    cc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, cc, 0, 0, 0);
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) {
                        const int brow = (tb << 4) + lk + 4 * rr;
                        if (brow < icol) F[icol * LDF + brow] = cc[rr];             // strict upper part only
                    }
                }
            }
        }
        __syncthreads();
    }

}

template <int RMAX>
__device__ __forceinline__ void factor_small_front(const DevSym &S, const int s, const double *__restrict__ nzval,
                                                   double *__restrict__ L, double *__restrict__ CB,
                                                   int *__restrict__ info, double *F, double *Pn, double *Wn) {
    constexpr int LDF = lds_ldf(RMAX);
    const int first = S.sfirst[s];
    const int c = S.sfirst[s + 1] - first;
    const int r = (int)(S.rowptr[s + 1] - S.rowptr[s]);
    const int ld = S.ld[s];
    const int m = r - c;
    const int tid = threadIdx.x;
    double *P = L + S.panelptr[s];

    // ---- assembly in LDS -----------------------------------------------------------------
    for (int idx = tid; idx < LDF * RMAX; idx += 256) F[idx] = 0.0;
    __syncthreads();
    {
        const long long q0 = S.qptr[s];
        const int nq = (int)(S.qptr[s + 1] - q0);
        for (int q = tid; q < nq; q += 256) {
            const int col = S.qcol[q0 + q], row = S.qdst[q0 + q];
            F[col * LDF + row] = nzval[S.qsrc[q0 + q]];
        }
    }
    __syncthreads();
    for (long long ch = S.childptr[s]; ch < S.childptr[s + 1]; ch++) {
        const EdgeRec er = S.edge[ch];
        const int md = er.md;
        const int *reld = S.rel + er.reloff;
        const double *Ud = CB + er.cboff;
        // relative indices of the child once into LDS (Pn is free during assembly), then the
        // child's lower triangle in batches of 8 columns per thread: 8 independent loads in flight
        int *relL = reinterpret_cast<int *>(Pn);
        for (int k = tid; k < md; k += 256) relL[k] = reld[k];
        __syncthreads();
        const int lane = tid & 63, jj = tid >> 6;
        for (int ib = 0; ib < md; ib += 64) {
            const int i = ib + lane;
            const int ic = min(i, md - 1);
            const int ti = relL[ic];
            for (int j0 = 0; j0 < md; j0 += 32) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int j = min(j0 + jj + 4 * u, md - 1);
                    v[u] = Ud[max(ic, j) + (long long)min(ic, j) * md];     // always a valid lower entry
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int j = j0 + jj + 4 * u;
                    if (i < md && j < md && i >= j) F[relL[j] * LDF + ti] += v[u];
                }
            }
        }
        __syncthreads();
    }

    lds_partial_cholesky<RMAX>(F, Pn, Wn, c, r, first, info, P, ld);
    const int wave = tid >> 6, lane = tid & 63;

    // ---- write-out: the panel went to HBM column block by column block; only CB (lower) is left ---
    double *U = CB + S.cbptr[s];
    for (int j = wave; j < m; j += 4) {
        double *dst = U + (long long)j * m;
        for (int i = j + lane; i < m; i += 64) dst[i] = F[(c + j) * LDF + c + i];
    }
}

template <int RMAX>
__global__ __launch_bounds__(256, (RMAX <= 64 ? 4 : (RMAX <= 96 ? 2 : 1))) void k_factor_small(DevSym S, const int *__restrict__ list,
                                                      const double *__restrict__ nzval, double *__restrict__ L,
                                                      double *__restrict__ CB, int *__restrict__ info) {
    __shared__ double F[lds_ldf(RMAX) * RMAX];
    __shared__ double Pn[4 * lds_ldp(RMAX)];   // current panel, Pn[q][i] = L[i][j0+q] (0 outside the panel rows)
    __shared__ double Wn[4 * LDW];             // current rows of L11^-1, Wn[q][b] = X[j0+q][b] (0 for b > j0+q)
    factor_small_front<RMAX>(S, list[blockIdx.x], nzval, L, CB, info, F, Pn, Wn);
}

// Subtree task: ONE workgroup factors a whole small subtree (supernode ids first..last are its
// fronts in postorder), so the bottom levels of the tree cost one launch and a child's contribution
// block is consumed by the same CU that produced it (L2-hot) instead of one launch per level.
template <int RMAX>
__global__ __launch_bounds__(256, (RMAX <= 64 ? 4 : 2)) void k_factor_subtree(DevSym S, const int *__restrict__ sub_first,
                                                        const int *__restrict__ sub_last,
                                                        const double *__restrict__ nzval, double *__restrict__ L,
                                                        double *__restrict__ CB, int *__restrict__ info) {
    __shared__ double F[lds_ldf(RMAX) * RMAX];
    __shared__ double Pn[4 * lds_ldp(RMAX)];
    __shared__ double Wn[4 * LDW];
    const int s0 = sub_first[blockIdx.x], s1 = sub_last[blockIdx.x];
    for (int s = s0; s <= s1; s++) {
        factor_small_front<RMAX>(S, s, nzval, L, CB, info, F, Pn, Wn);
        __threadfence_block();
        __syncthreads();   // this front's CB (global) is visible to the wave that reads it for the parent
    }
}

// ------------------------------------------------------------------------------------------
// Fused sweeps for small fronts. The front's slice of the right-hand sides (r rows x up to 64
// columns) sits in LDS; the panel streams through the MFMA A operand straight from HBM (each
// element read once), the diagonal block is applied through its stored inverse.
// ------------------------------------------------------------------------------------------
constexpr int LDV = 80;   // LDS row stride of the right-hand-side slice, = 16 (mod 32)

// element (k,q) of L11^-1 from the panel: strict lower part is stored transposed in the strict
// upper triangle, the diagonal is the reciprocal of L's. Arithmetic masking keeps the load
// unconditional.
__device__ __forceinline__ double linv_elem(const double *__restrict__ P, int ld, int c, int k, int q, bool lower) {
    const int kk = min(k, c - 1), qq = min(q, c - 1);
    const double v = P[min(kk, qq) + (long long)max(kk, qq) * ld];
    const bool on = (k < c && q < c) && (lower ? (q < k) : (q > k));
    double x = v * (on ? 1.0 : 0.0);
    if (k == q && k < c) x = fast_rcp(v);
    return x;
}

template <int RMAX>
__device__ __forceinline__ void fwd_small_front(const DevSym &S, const int s, const double *__restrict__ L,
                                                double *__restrict__ X, double *__restrict__ W, const int nr,
                                                const int ldx, double *fv) {
    const int first = S.sfirst[s];
    const int c = S.sfirst[s + 1] - first;
    const int r = (int)(S.rowptr[s + 1] - S.rowptr[s]);
    const int ld = S.ld[s];
    const double *P = L + S.panelptr[s];
    double *Ws = W + S.wptr[s] * ldx;
    const int tid = threadIdx.x;
    const int j = tid & 63, g = tid >> 6;
    const int jc = min(j, nr - 1);
    const double jm = j < nr ? 1.0 : 0.0;
    // own rows <- b ; all other rows (trailing and r..RMAX-1) <- 0: masked MFMA k-steps still multiply
    // 0 by whatever is there. Unconditional clamped loads, 8 in flight per thread.
    {   // c <= 64 own rows: ONE batch of 16 clamped loads per thread (rows g, g+4, ..); everything else <- 0
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = X[(long long)(first + min(g + 4 * u, c - 1)) * ldx + jc];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int i = g + 4 * u;
            if (i < RMAX) fv[i * LDV + j] = (i < c) ? v[u] * jm : 0.0;
        }
        for (int i = 64 + g; i < RMAX; i += 4) fv[i * LDV + j] = 0.0;
    }
    __syncthreads();
    for (long long ch = S.childptr[s]; ch < S.childptr[s + 1]; ch++) {
        const EdgeRec er = S.edge[ch];
        const int md = er.md;
        const int *reld = S.rel + er.reloff;
        const double *Wd = W + er.woff * ldx;
        for (int a0 = g; a0 < md; a0 += 64) {      // 16 child rows per thread and pass: one pass for md <= 64
            double v[16]; int tr[16];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int a = min(a0 + 4 * u, md - 1);
                tr[u] = reld[a];
                v[u] = Wd[(long long)a * ldx + jc];
            }
#pragma unroll
            for (int u = 0; u < 16; u++) if (a0 + 4 * u < md) fv[tr[u] * LDV + j] += v[u] * jm;
        }
        __syncthreads();
    }
    const int wave = tid >> 6, lane = tid & 63;
    const int lm = lane & 15, lk = lane >> 4;
    const int nt = (nr + 15) >> 4;
    // ---- y = L11^-1 b ---------------------------------------------------------------------
    {
        const int k0 = wave * 16;
        d4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
        if (k0 < c) {
            const int qhi = min(c, k0 + 16);
#pragma unroll 1
            for (int q0 = 0; q0 < qhi; q0 += 16) {
                double av[4];
#pragma unroll
                for (int u = 0; u < 4; u++) av[u] = linv_elem(P, ld, c, k0 + lm, q0 + 4 * u + lk, true);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int q = q0 + 4 * u + lk;   // < 64 <= RMAX: always inside fv
#pragma unroll
                    for (int t = 0; t < 4; t++)
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], fv[q * LDV + t * 16 + lm], acc[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();   // everyone has read b
        if (k0 < c) {
#pragma unroll
            for (int t = 0; t < 4; t++) {
                if (t < nt) {
                    const int jj = t * 16 + lm;
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) {
                        const int k = k0 + lk + 4 * rr;
                        if (k < c) {
                            fv[k * LDV + jj] = acc[t][rr];
                            if (jj < nr) X[(long long)(first + k) * ldx + jj] = acc[t][rr];
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    // ---- W_s = (children's contributions) - L21 y --------------------------------------------
    const int ntile = (r - c + 15) >> 4;
#pragma unroll 1
    for (int it = wave; it < ntile; it += 4) {
        const int i0 = c + it * 16;
        const double *pa = P + min(i0 + lm, r - 1);
        d4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
        for (int q0 = 0; q0 < c; q0 += 16) {
            double av[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int q = q0 + 4 * u + lk;
                av[u] = pa[(long long)min(q, c - 1) * ld] * (q < c ? 1.0 : 0.0);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                // unconditional: av[u] is zero beyond column c and fv is finite everywhere (no runtime-conditional
                // MFMA inside the unrolled loop: see the rules in DESIGN.md)
                const int q = min(q0 + 4 * u + lk, RMAX - 1);
#pragma unroll
                for (int t = 0; t < 4; t++)
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], fv[q * LDV + t * 16 + lm], acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (t < nt) {
                const int jj = t * 16 + lm;
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const int i = i0 + lk + 4 * rr;
                    if (i < r && jj < nr) Ws[(long long)(i - c) * ldx + jj] = fv[i * LDV + jj] - acc[t][rr];
                }
            }
        }
    }
}

template <int RMAX>
__global__ __launch_bounds__(256, (RMAX <= 48 ? 4 : (RMAX <= 64 ? 3 : (RMAX <= 96 ? 2 : 1)))) void k_fwd_small(DevSym S, const int *__restrict__ list,
                                                   const double *__restrict__ L, double *__restrict__ X,
                                                   double *__restrict__ W, int nr, int ldx) {
    __shared__ double fv[RMAX * LDV];
    fwd_small_front<RMAX>(S, list[blockIdx.x], L, X, W, nr, ldx, fv);
}
// forward sweep of a whole small subtree by one workgroup (fronts in postorder)
template <int RMAX>
__global__ __launch_bounds__(256, (RMAX <= 48 ? 4 : (RMAX <= 64 ? 3 : 2))) void k_fwd_subtree(DevSym S, const int *__restrict__ sub_first,
                                                     const int *__restrict__ sub_last, const double *__restrict__ L,
                                                     double *__restrict__ X, double *__restrict__ W, int nr, int ldx) {
    __shared__ double fv[RMAX * LDV];
    const int s0 = sub_first[blockIdx.x], s1 = sub_last[blockIdx.x];
    for (int s = s0; s <= s1; s++) {
        fwd_small_front<RMAX>(S, s, L, X, W, nr, ldx, fv);
        __threadfence_block();
        __syncthreads();
    }
}

template <int RMAX>
__device__ __forceinline__ void bwd_small_front(const DevSym &S, const int s, const double *__restrict__ L,
                                                double *__restrict__ X, const int nr, const int ldx, double *fv,
                                                int *rowsL) {
    const int first = S.sfirst[s];
    const int c = S.sfirst[s + 1] - first;
    const int r = (int)(S.rowptr[s + 1] - S.rowptr[s]);
    const int ld = S.ld[s];
    const double *P = L + S.panelptr[s];
    const int *rows = S.rows + S.rowptr[s];
    const int tid = threadIdx.x;
    {
        const int j = tid & 63, g = tid >> 6;
        const int jc = min(j, nr - 1);
        const double jm = j < nr ? 1.0 : 0.0;
        for (int k = tid; k < RMAX; k += 256) rowsL[k] = rows[min(k, r - 1)];
        __syncthreads();
        for (int i0 = g; i0 < RMAX; i0 += 32) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = X[(long long)rowsL[min(i0 + 4 * u, RMAX - 1)] * ldx + jc];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = i0 + 4 * u;
                if (i < RMAX) fv[i * LDV + j] = (i < r) ? v[u] * jm : 0.0;
            }
        }
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    const int lm = lane & 15, lk = lane >> 4;
    const int nt = (nr + 15) >> 4;
    const int k0 = wave * 16;
    // ---- t = y - L21' x_R (own rows k0..k0+15 of this wave) -------------------------------------
    if (k0 < c) {
        const double *pa = P + (long long)min(k0 + lm, c - 1) * ld;
        d4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
        // all of this column block's L21 values at once (r - c < RMAX: RMAX / 4 clamped loads in
        // flight instead of a chain of (r - c) / 16 round trips), masks at use
        constexpr int NQ = RMAX / 16;
        double av[NQ][4];
#pragma unroll
        for (int qb = 0; qb < NQ; qb++)
#pragma unroll
            for (int u = 0; u < 4; u++) av[qb][u] = pa[min(c + qb * 16 + 4 * u + lk, r - 1)];
#pragma unroll
        for (int qb = 0; qb < NQ; qb++) {
            if (c + qb * 16 < r) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int qq = c + qb * 16 + 4 * u + lk;
                    const int q = min(qq, RMAX - 1);
                    const double a_ = av[qb][u] * (qq < r ? 1.0 : 0.0);
#pragma unroll
                    for (int t = 0; t < 4; t++)
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_, fv[q * LDV + t * 16 + lm], acc[t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (t < nt) {
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const int k = k0 + lk + 4 * rr;
                    if (k < c) fv[k * LDV + t * 16 + lm] -= acc[t][rr];
                }
            }
        }
    }
    __syncthreads();
    // ---- x = L11^-T t ------------------------------------------------------------------------
    if (k0 < c) {
        d4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
        for (int q0 = k0; q0 < c; q0 += 16) {
            double av[4];
#pragma unroll
            for (int u = 0; u < 4; u++) av[u] = linv_elem(P, ld, c, k0 + lm, q0 + 4 * u + lk, false);   // Linv[q][k], q > k
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int q = q0 + 4 * u + lk;   // < 64 + 16 <= RMAX
#pragma unroll
                for (int t = 0; t < 4; t++)
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], fv[min(q, RMAX - 1) * LDV + t * 16 + lm], acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (t < nt) {
                const int jj = t * 16 + lm;
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const int k = k0 + lk + 4 * rr;
                    if (k < c && jj < nr) X[(long long)(first + k) * ldx + jj] = acc[t][rr];
                }
            }
        }
    }
}

template <int RMAX>
__global__ __launch_bounds__(256, (RMAX <= 48 ? 4 : (RMAX <= 64 ? 3 : (RMAX <= 96 ? 2 : 1)))) void k_bwd_small(DevSym S, const int *__restrict__ list,
                                                   const double *__restrict__ L, double *__restrict__ X, int nr,
                                                   int ldx) {
    __shared__ double fv[RMAX * LDV];
    __shared__ int rowsL[RMAX];
    bwd_small_front<RMAX>(S, list[blockIdx.x], L, X, nr, ldx, fv, rowsL);
}
// backward sweep of a whole small subtree by one workgroup (fronts in reverse postorder)
template <int RMAX>
__global__ __launch_bounds__(256, (RMAX <= 48 ? 4 : (RMAX <= 64 ? 3 : 2))) void k_bwd_subtree(DevSym S, const int *__restrict__ sub_first,
                                                     const int *__restrict__ sub_last, const double *__restrict__ L,
                                                     double *__restrict__ X, int nr, int ldx) {
    __shared__ double fv[RMAX * LDV];
    __shared__ int rowsL[RMAX];
    const int s0 = sub_first[blockIdx.x], s1 = sub_last[blockIdx.x];
    for (int s = s1; s >= s0; s--) {
        bwd_small_front<RMAX>(S, s, L, X, nr, ldx, fv, rowsL);
        __threadfence_block();
        __syncthreads();
    }
}


void launch_factor_small(hipStream_t st, const DevSym &S, const int *list, int nfronts, int rmax,
                         const double *nzval, double *L, double *CB, int *info) {
    if (nfronts <= 0) return;
    if (rmax <= 48) hipLaunchKernelGGL(k_factor_small<48>, dim3(nfronts), dim3(256), 0, st, S, list, nzval, L, CB, info);
    else if (rmax <= 64) hipLaunchKernelGGL(k_factor_small<64>, dim3(nfronts), dim3(256), 0, st, S, list, nzval, L, CB, info);
    else if (rmax <= 96) hipLaunchKernelGGL(k_factor_small<96>, dim3(nfronts), dim3(256), 0, st, S, list, nzval, L, CB, info);
    else hipLaunchKernelGGL(k_factor_small<128>, dim3(nfronts), dim3(256), 0, st, S, list, nzval, L, CB, info);
}

void launch_subtree(hipStream_t st, const DevSym &S, int phase, const int *sub_first, const int *sub_last, int ntasks,
                    int rmax, const double *nzval, double *L, double *CB, int *info, double *X, double *W, int nr, int ldx) {
    if (ntasks <= 0) return;
    const dim3 g(ntasks), b(256);
#define GMRFX_SUB(R)                                                                                                   \
    do {                                                                                                               \
        if (phase == 0) hipLaunchKernelGGL(k_factor_subtree<R>, g, b, 0, st, S, sub_first, sub_last, nzval, L, CB, info); \
        else if (phase == 1) hipLaunchKernelGGL(k_fwd_subtree<R>, g, b, 0, st, S, sub_first, sub_last, L, X, W, nr, ldx);  \
        else hipLaunchKernelGGL(k_bwd_subtree<R>, g, b, 0, st, S, sub_first, sub_last, L, X, nr, ldx);                    \
    } while (0)
    if (rmax <= 48) GMRFX_SUB(48);
    else if (rmax <= 64) GMRFX_SUB(64);
    else GMRFX_SUB(96);
#undef GMRFX_SUB
}
void launch_fwd_small(hipStream_t st, const DevSym &S, const int *list, int nfronts, int rmax, const double *L,
                      double *X, double *W, int nr, int ldx) {
    if (nfronts <= 0) return;
    if (rmax <= 48) hipLaunchKernelGGL(k_fwd_small<48>, dim3(nfronts), dim3(256), 0, st, S, list, L, X, W, nr, ldx);
    else if (rmax <= 64) hipLaunchKernelGGL(k_fwd_small<64>, dim3(nfronts), dim3(256), 0, st, S, list, L, X, W, nr, ldx);
    else if (rmax <= 96) hipLaunchKernelGGL(k_fwd_small<96>, dim3(nfronts), dim3(256), 0, st, S, list, L, X, W, nr, ldx);
    else hipLaunchKernelGGL(k_fwd_small<128>, dim3(nfronts), dim3(256), 0, st, S, list, L, X, W, nr, ldx);
}
void launch_bwd_small(hipStream_t st, const DevSym &S, const int *list, int nfronts, int rmax, const double *L,
                      double *X, int nr, int ldx) {
    if (nfronts <= 0) return;
    if (rmax <= 48) hipLaunchKernelGGL(k_bwd_small<48>, dim3(nfronts), dim3(256), 0, st, S, list, L, X, nr, ldx);
    else if (rmax <= 64) hipLaunchKernelGGL(k_bwd_small<64>, dim3(nfronts), dim3(256), 0, st, S, list, L, X, nr, ldx);
    else if (rmax <= 96) hipLaunchKernelGGL(k_bwd_small<96>, dim3(nfronts), dim3(256), 0, st, S, list, L, X, nr, ldx);
    else hipLaunchKernelGGL(k_bwd_small<128>, dim3(nfronts), dim3(256), 0, st, S, list, L, X, nr, ldx);
}

}  // namespace gmrfx
