// potrf64.hip -- Cholesky factor AND inverse of one 64 x 64 diagonal block per workgroup.
//
// This kernel sits on the critical path of every big front (one call per 64 columns; the root
// front of a 1000 x 1000 grid alone needs 47 of them, one after the other), so it is built for
// LATENCY, not throughput:
//   * the block lives in REGISTERS: thread (ty, tx) of a 16 x 16 thread grid owns the 4 x 4 patch
//     rows 4ty.., columns 4tx..; patches on and below the diagonal hold A (later L), patches
//     above it hold M = the running forward substitution L X = I, transposed -- exactly where the
//     HBM panel keeps (L11^-1)' for the solve / selected-inversion kernels
//   * ONE barrier per 4-column step: at the end of step p the thread column tx = p+1 publishes
//     its (fully updated) patches as a 64 x 4 strip in LDS; after the barrier EVERY thread reads
//     the strip's 4 x 4 diagonal block and refactors it redundantly (4 rsqrt chains), applies it
//     to the two strip blocks it needs (rows of ty and of tx) and updates its patch with 64 VALU
//     FMAs. FP64 VALU FMA sustains 64 TFLOP/s on this chip against 36 for the FP64 MFMA, and a
//     register-resident patch needs no LDS read-modify-write
//   * A and M patches share one update formula: with Y_b = S_b Lpp^-T (S = strip block of
//     thread-row b) the A update is A -= Y_ty Y_tx' and the M update is M' -= W' L' = Y_ty Y_tx'
//     because the rows of X that become final in step p are W' = S_ty Lpp^-T as well
// Reference semantics: cholesky!(F, Q) of CHOLMOD as used by
// /root/reference/src/workspace/backend.jl:91-96 (update_factorization!) -- here for one dense
// diagonal block of a supernode.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace gmrfx {

typedef gmrfx_d4 d4;

namespace {

// phase-cycle instrumentation for tools/micro/potrf_prof.hip (compiled out of the library)
#ifdef GMRFX_CYC
__device__ long long g_cyc64[4][16];
#define C64_DECL long long cyc_t = clock64(); const int cyc_w = (threadIdx.x & 63) == 0 ? (int)(threadIdx.x >> 6) : -1
#define C64_MARK(k) do { __builtin_amdgcn_sched_barrier(0); long long t_ = clock64(); if (cyc_w >= 0) g_cyc64[cyc_w][k] += t_ - cyc_t; cyc_t = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define C64_DECL
#define C64_MARK(k)
#endif

__device__ __forceinline__ double rsqrt_nr(double p) {
    // v_rsq_f64 + two Newton steps: full double precision without sqrt + division
    double y = __builtin_amdgcn_rsq(p);
    y = y * (1.5 - 0.5 * p * y * y);
    y = y * (1.5 - 0.5 * p * y * y);
    return y;
}

}  // namespace

// The 256 threads (tid 0..255) of a workgroup factor the w x w block whose current values are src[i + j * sld] (the panel
// itself, or the block a look-ahead prologue left in LDS) and write L / (L^-1)' to the panel block P (leading dimension ld).
// Sb: 2 x 4 x 64 doubles of LDS.
__device__ __forceinline__ void potrf64_body(const double *src, const int sld, double *__restrict__ P, const int ld, const int w,
                                             double (*Sb)[4 * 64], int *__restrict__ info, const int first_col, const int tid) {
    const int ty = tid & 15, tx = tid >> 4;     // lanes walk rows: coalesced panel loads / stores
    const int i0 = 4 * ty, j0 = 4 * tx;

    // patch a[r][cc] = element (i0 + r, j0 + cc); identity padding beyond w; M part starts at 0
    double a[4][4];
#pragma unroll
    for (int cc = 0; cc < 4; cc++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = i0 + r, j = j0 + cc;
            const double v = src[min(i, w - 1) + min(j, w - 1) * sld];
            const double mk = (i < w && j < w && i >= j) ? 1.0 : 0.0;
            a[r][cc] = v * mk + ((i == j && i >= w) ? 1.0 : 0.0);
        }
    if (tx == 0) {
#pragma unroll
        for (int cc = 0; cc < 4; cc++)
#pragma unroll
            for (int r = 0; r < 4; r++) Sb[0][cc * 64 + i0 + r] = a[r][cc];
    }
    __syncthreads();

    const int np = (w + 3) >> 2;
    int badcol = 0x7fffffff;     // first non-positive pivot seen by this thread (diagonal patches only)
    C64_DECL;
    for (int p = 0; p < np; p++) {
        C64_MARK(0);
        const double *Sp = Sb[p & 1];
        if (tx >= p) {
            // diagonal 4 x 4 block of the strip -> Lpp (l..) and the reciprocals of its diagonal
            double d[4][4];
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int r = 0; r < 4; r++) d[r][q] = Sp[q * 64 + 4 * p + r];
            double sy[4][4], sx[4][4];
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    sy[r][q] = Sp[q * 64 + i0 + r];
                    sx[r][q] = Sp[q * 64 + j0 + r];
                }
            if (ty == p) {
                // rows of X that become final now start from the identity (L X = I)
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int r = 0; r < 4; r++) sy[r][q] = (r == q) ? 1.0 : 0.0;
            }
#ifdef GMRFX_CYC
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            C64_MARK(1);
            const double p0 = d[0][0];
            const double i00 = rsqrt_nr(p0);
            const double l10 = d[1][0] * i00, l20 = d[2][0] * i00, l30 = d[3][0] * i00;
            const double p1 = d[1][1] - l10 * l10;
            const double i11 = rsqrt_nr(p1);
            const double l21 = (d[2][1] - l20 * l10) * i11, l31 = (d[3][1] - l30 * l10) * i11;
            const double p2 = d[2][2] - l20 * l20 - l21 * l21;
            const double i22 = rsqrt_nr(p2);
            const double l32 = (d[3][2] - l30 * l20 - l31 * l21) * i22;
            const double p3 = d[3][3] - l30 * l30 - l31 * l31 - l32 * l32;
            const double i33 = rsqrt_nr(p3);
#ifdef GMRFX_CYC
            asm volatile("" ::"v"(i33));
#endif
            C64_MARK(2);
            // Y = S Lpp^-T, row by row
            double yy[4][4], yx[4][4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                yy[r][0] = sy[r][0] * i00;
                yy[r][1] = (sy[r][1] - yy[r][0] * l10) * i11;
                yy[r][2] = (sy[r][2] - yy[r][0] * l20 - yy[r][1] * l21) * i22;
                yy[r][3] = (sy[r][3] - yy[r][0] * l30 - yy[r][1] * l31 - yy[r][2] * l32) * i33;
                yx[r][0] = sx[r][0] * i00;
                yx[r][1] = (sx[r][1] - yx[r][0] * l10) * i11;
                yx[r][2] = (sx[r][2] - yx[r][0] * l20 - yx[r][1] * l21) * i22;
                yx[r][3] = (sx[r][3] - yx[r][0] * l30 - yx[r][1] * l31 - yx[r][2] * l32) * i33;
            }
#ifdef GMRFX_CYC
            asm volatile("" ::"v"(yy[3][3]), "v"(yx[3][3]));
#endif
            C64_MARK(3);
            if (tx == p) {
                // this thread column is final: L below the diagonal block, X' above it. The values stay in the
                // patch registers and go to the panel in ONE store pass after the loop (16 predicated stores per
                // step from a quarter of a wave cost 0.15 us of the 1.1 us step), the pivot check is collected in
                // a register and reported by one atomic at the end.
                if (ty == p) {
                    // the diagonal block itself: Lpp in the lower part (diag = pivot * rsqrt),
                    // Lpp^-1 transposed (= yy of the identity) in the strict upper part
                    const double pv[4] = {p0, p1, p2, p3};
#pragma unroll
                    for (int q = 3; q >= 0; q--)
                        badcol = (!(pv[q] > 0.0) && 4 * p + q < w) ? min(badcol, 4 * p + q) : badcol;
                    yy[0][0] = p0 * i00; yy[1][1] = p1 * i11; yy[2][2] = p2 * i22; yy[3][3] = p3 * i33;
                    yy[1][0] = l10; yy[2][0] = l20; yy[3][0] = l30;
                    yy[2][1] = l21; yy[3][1] = l31;
                    yy[3][2] = l32;
                }
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int r = 0; r < 4; r++) a[r][q] = yy[r][q];
            } else if (ty >= tx || ty <= p) {
                // trailing A patch, or M patch whose rows of X are already final (b <= 4p+3)
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int cc = 0; cc < 4; cc++) {
                        double acc = a[r][cc];
#pragma unroll
                        for (int q = 0; q < 4; q++) acc -= yy[r][q] * yx[cc][q];
                        a[r][cc] = acc;
                    }
                if (tx == p + 1) {
                    double *Sn = Sb[(p + 1) & 1];
#pragma unroll
                    for (int cc = 0; cc < 4; cc++)
#pragma unroll
                        for (int r = 0; r < 4; r++) Sn[cc * 64 + i0 + r] = a[r][cc];
                }
            }
        }
        C64_MARK(4);
        __syncthreads();
        C64_MARK(5);
    }
    if (badcol != 0x7fffffff) atomicMin(info, first_col + badcol);
    double *Pt = P + i0 + (long long)j0 * ld;
    if (w == NB) {
        // full block (all but the last block column of a front): no per-element predicates
#pragma unroll
        for (int cc = 0; cc < 4; cc++)
#pragma unroll
            for (int r = 0; r < 4; r++) Pt[r + (long long)cc * ld] = a[r][cc];
    } else {
#pragma unroll
        for (int cc = 0; cc < 4; cc++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (i0 + r < w && j0 + cc < w) Pt[r + (long long)cc * ld] = a[r][cc];
    }
}

__global__ __launch_bounds__(256) void k_potrf64(DevSym S, const FrontView *__restrict__ frec, int kb,
                                                 double *__restrict__ L, int *__restrict__ info, FrontArg fa) {
    __shared__ __attribute__((aligned(16))) double Sb[2][4 * 64];   // strip [parity][q * 64 + row]
    const FrontView fv = front_view(frec, blockIdx.x, fa);
    const int c = fv.c;
    if (kb >= c) return;
    const int w = min(NB, c - kb);
    const int ld = fv.ld;
    double *P = L + fv.pp + kb + (long long)kb * ld;
    potrf64_body(P, ld, P, ld, w, Sb, info, fv.first + kb, threadIdx.x);
}

// ---- LOOK-AHEAD form (round 3) ------------------------------------------------------------------------------------------
// The panel chain of a big front was potrf64 -> trsm -> gemm per 64-column block: three dependent launches, two of which
// only exist on the critical path because the NEXT diagonal block needs their results in two 64 x 64 tiles. Here the
// diagonal chain serves itself: before it factors block b (b > first block of its 256-column outer block), the workgroup
// brings the BAND -- the sub-diagonal tile (b, b-1) and the diagonal tile (b, b) -- up to date left-looking,
//     S    = A[b, b-1] - sum_{j < b-1} L[b, j] L[b-1, j]'        (j runs over the earlier blocks of the outer block)
//     Lsub = S Linv[b-1]'                                        -> written to L[b, b-1]
//     D    = A[b, b]   - sum_{j < b-1} L[b, j] L[b, j]' - Lsub Lsub'
// on the MFMA with sixteen waves (<= 2 us), and factors D from LDS. The bulk kernels (trsm of the rows below the band,
// trailing update of everything but the band tiles) run one step behind on a second stream and never touch these tiles
// (k_trsm `la`, k_gemm_nt `band`): the critical path per block is this one kernel.
__global__ __launch_bounds__(512) void k_potrf64_la(DevSym S, const FrontView *__restrict__ frec, int kb, int kb0,
                                                    double *__restrict__ L, int *__restrict__ info, FrontArg fa) {
    // 8 waves (the factorisation below needs ~150 registers per thread: 16 waves would spill it): wave v owns the 16 x 16
    // tiles (ti, tj) and (ti, tj + 1), ti = v & 3, tj = 2 (v >> 2) -- they share the A operand
    __shared__ __attribute__((aligned(16))) double Sb[2][4 * 64];
    __shared__ double Sm[NB * NB];      // S (column-major), later D
    __shared__ double Ls[NB * NB];      // Lsub (column-major)
    __shared__ double Ti[NB * NB];      // Linv of block b-1: Ti[k * NB + q] = Linv[k][q]
    const FrontView fv = front_view(frec, blockIdx.x, fa);
    const int c = fv.c;
    if (kb >= c) return;
    const int w = min(NB, c - kb);
    const int ld = fv.ld;
    double *Pf = L + fv.pp;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, lm = lane & 15, lk = lane >> 4;
    const int ti = wave & 3, tj0 = 2 * (wave >> 2);
    const int kp = kb - NB;                             // first column of block b-1 (a full block: one follows it)
    // ---- phase A: the sums over the earlier blocks of the outer block (columns kb0 .. kp-1), S and D tiles at once -----
    const int ia = min(16 * ti + lm, w - 1);            // row of block b (A operand), clamped
    const double *Ab = Pf + kb + (long long)kb0 * ld;   // row block b, from column kb0
    const double *Ap = Pf + kp + (long long)kb0 * ld;   // row block b-1
    d4 accS[2], accD[2];
#pragma unroll
    for (int e = 0; e < 2; e++) { accS[e] = (d4){0.0, 0.0, 0.0, 0.0}; accD[e] = (d4){0.0, 0.0, 0.0, 0.0}; }
    const int Kprev = kp - kb0;                         // multiple of 64
    for (int k0 = 0; k0 < Kprev; k0 += 16) {
        double av[4], bp[2][4], bb[2][4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const long long off = (long long)(k0 + 4 * u + lk) * ld;
            av[u] = Ab[ia + off];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                bp[e][u] = Ap[16 * (tj0 + e) + lm + off];
                bb[e][u] = Ab[min(16 * (tj0 + e) + lm, w - 1) + off];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int e = 0; e < 2; e++) {
                accS[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bp[e][u], accS[e], 0, 0, 0);
                accD[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bb[e][u], accD[e], 0, 0, 0);
            }
    }
    // tile element D[i = 4 rr + lk][j = lm]; the panel's own values of the tiles
    {
        double so[2][4], d0[2][4];
#pragma unroll
        for (int e = 0; e < 2; e++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int i = min(16 * ti + 4 * rr + lk, w - 1), j = 16 * (tj0 + e) + lm;
                so[e][rr] = Pf[kb + i + (long long)(kp + j) * ld];
                d0[e][rr] = Pf[kb + i + (long long)(kb + min(j, w - 1)) * ld];
            }
#pragma unroll
        for (int e = 0; e < 2; e++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int i = 16 * ti + 4 * rr + lk, j = 16 * (tj0 + e) + lm;
                Sm[j * NB + i] = so[e][rr] - accS[e][rr];
                accD[e][rr] = d0[e][rr] - accD[e][rr];          // kept in registers until phase C
            }
    }
    stage_linv(Pf + kp + (long long)kp * ld, ld, NB, Ti, tid & 255);      // (two copies of the same values: harmless)
    __syncthreads();
    // ---- phase B: Lsub = S Linv[b-1]' (Linv lower: k <= q) ------------------------------------------------------------
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const int tj = tj0 + e;
        d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
        const int ku = 4 * (tj + 1);        // wave-uniform
#pragma unroll 4
        for (int u = 0; u < ku; u++) {
            const int k = 4 * u + lk;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Sm[k * NB + 16 * ti + lm], Ti[(16 * tj + lm) * NB + k], acc, 0, 0, 0);
        }
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int i = 16 * ti + 4 * rr + lk, q = 16 * tj + lm;
            Ls[q * NB + i] = acc[rr];
            if (i < w) Pf[kb + i + (long long)(kp + q) * ld] = acc[rr];
        }
    }
    __syncthreads();
    // ---- phase C: D -= Lsub Lsub' (lower tiles only), D -> LDS (every wave has read S before the barrier above: its
    //      buffer becomes D) -------------------------------------------------------------------------------------------
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const int tj = tj0 + e;
        if (ti >= tj) {
            d4 accC = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int q = 4 * u + lk;
                accC = __builtin_amdgcn_mfma_f64_16x16x4f64(Ls[q * NB + 16 * ti + lm], Ls[q * NB + 16 * tj + lm], accC, 0, 0, 0);
            }
#pragma unroll
            for (int rr = 0; rr < 4; rr++) Sm[(16 * tj + lm) * NB + 16 * ti + 4 * rr + lk] = accD[e][rr] - accC[rr];
        }
    }
    __syncthreads();
    if (tid >= 256) return;             // (a finished wave no longer counts at the barriers of the factorisation below)
    potrf64_body(Sm, NB, Pf + kb + (long long)kb * ld, ld, w, Sb, info, fv.first + kb, tid);
}

void launch_potrf64(hipStream_t st, const DevSym &S, const FrontView *frec, int nactive, int kb, double *L, int *info,
                    const FrontArg &fa) {
    if (nactive <= 0) return;
    hipLaunchKernelGGL(k_potrf64, dim3(nactive), dim3(256), 0, st, S, frec, kb, L, info, fa);
}
void launch_potrf64_la(hipStream_t st, const DevSym &S, const FrontView *frec, int nactive, int kb, int kb0, double *L, int *info,
                       const FrontArg &fa) {
    if (nactive <= 0) return;
    hipLaunchKernelGGL(k_potrf64_la, dim3(nactive), dim3(512), 0, st, S, frec, kb, kb0, L, info, fa);
}

}  // namespace gmrfx
