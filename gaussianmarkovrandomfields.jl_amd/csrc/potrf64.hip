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
#include "potrf64_body.h"
#include "potrf64_blocked.h"

namespace gmrfx {

typedef gmrfx_d4 d4;

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

// ---- 16-COLUMN-STEP form (round 4; potrf64_blocked.h): one wave eliminates 16 x 16 diagonal blocks in registers, helper waves
// keep the rest of the block and its inverse up to date on the MFMA beside it. The product path; GMRFX_POTRF=1 selects the
// register-patch kernel above.
template <int NW, int ND>
__global__ __launch_bounds__(64 * NW) void k_potrf64_b(DevSym S, const FrontView *__restrict__ frec, int kb,
                                                       double *__restrict__ L, int *__restrict__ info, FrontArg fa) {
    __shared__ __attribute__((aligned(16))) pb::Smem<ND> sm;
    const FrontView fv = front_view(frec, blockIdx.x, fa);
    const int c = fv.c;
    if (kb >= c) return;
    const int w = min(NB, c - kb);
    const int ld = fv.ld;
    double *P = L + fv.pp + kb + (long long)kb * ld;
    pb::potrf64_blocked<NW, ND>(P, ld, P, ld, w, sm, info, fv.first + kb, threadIdx.x);
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
                    const FrontArg &fa, int form, int wmax) {
    if (nactive <= 0) return;
    if (form == 3) {
        // the workgroup shape follows the widest block of the launch (potrf64_blocked.h)
        if (wmax <= 16) hipLaunchKernelGGL((k_potrf64_b<1, 1>), dim3(nactive), dim3(64), 0, st, S, frec, kb, L, info, fa);
        else if (wmax <= 32) hipLaunchKernelGGL((k_potrf64_b<2, 2>), dim3(nactive), dim3(128), 0, st, S, frec, kb, L, info, fa);
        else if (wmax <= 48) hipLaunchKernelGGL((k_potrf64_b<4, 3>), dim3(nactive), dim3(256), 0, st, S, frec, kb, L, info, fa);
        else hipLaunchKernelGGL((k_potrf64_b<8, 4>), dim3(nactive), dim3(512), 0, st, S, frec, kb, L, info, fa);
    } else hipLaunchKernelGGL(k_potrf64, dim3(nactive), dim3(256), 0, st, S, frec, kb, L, info, fa);
}
void launch_potrf64_la(hipStream_t st, const DevSym &S, const FrontView *frec, int nactive, int kb, int kb0, double *L, int *info,
                       const FrontArg &fa) {
    if (nactive <= 0) return;
    hipLaunchKernelGGL(k_potrf64_la, dim3(nactive), dim3(512), 0, st, S, frec, kb, kb0, L, info, fa);
}

}  // namespace gmrfx
