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

#include <algorithm>

#include <climits>

#include <cstdlib>

#include "kernels.h"

namespace gmrfx {

typedef gmrfx_d4 d4;
typedef gmrfx_d2u d2u;
typedef int i2u __attribute__((ext_vector_type(2), aligned(4)));   // two ints, 4-byte aligned: one 8-byte load

// Two lower bounds in the sorted array a[0..n) at once, by all 64 lanes of a wave together
// (64-ary search: 2 rounds of one load each for n <= 4096 instead of 12 dependent loads each).
// Must be called in wave-uniform control flow; the results are wave-uniform.
__device__ __forceinline__ void wave_lower_bound2(const int *__restrict__ a, const int n, const int k0, const int k1,
                                                  const int lane, int &r0, int &r1) {
    int lo0 = 0, hi0 = n, lo1 = 0, hi1 = n;
    while (lo0 < hi0 || lo1 < hi1) {
        const int st0 = max((hi0 - lo0 + 63) >> 6, 1), st1 = max((hi1 - lo1 + 63) >> 6, 1);
        const int x0 = lo0 + lane * st0, x1 = lo1 + lane * st1;
        const int v0 = a[min(x0, n - 1)], v1 = a[min(x1, n - 1)];
        const int c0 = __popcll(__ballot(x0 < hi0 && v0 < k0));
        const int c1 = __popcll(__ballot(x1 < hi1 && v1 < k1));
        if (lo0 < hi0) {
            if (c0 == 0) hi0 = lo0;
            else { const int nl = lo0 + (c0 - 1) * st0 + 1; hi0 = min(lo0 + c0 * st0, hi0); lo0 = nl; }
        }
        if (lo1 < hi1) {
            if (c1 == 0) hi1 = lo1;
            else { const int nl = lo1 + (c1 - 1) * st1 + 1; hi1 = min(lo1 + c1 * st1, hi1); lo1 = nl; }
        }
    }
    r0 = lo0;
    r1 = lo1;
}

// ------------------------------------------------------------------------------------------
// Factorisation
// ------------------------------------------------------------------------------------------

// Zero the panel, scatter Q's values, extend-add the children's contribution blocks.
// ONE WAVE owns one front-local column (workgroup (bx, f) = columns 4 bx .. 4 bx + 3 of front f):
// every target entry has exactly one owner, which applies the children one after the other, so
// the sum order is fixed (bit-reproducible, no atomics) and no barrier is needed at all; the
// kernel is a chain of dependent HBM round trips, kept short by the wave-wide searches.
template <int WIDE>   // 0: one WAVE per column; 1: one WORKGROUP per column (levels with a few tall fronts)
__global__ __launch_bounds__(256) void k_assemble(DevSym S, const int *__restrict__ list,
                                                  const double *__restrict__ nzval, double *__restrict__ L,
                                                  double *__restrict__ CB, int cyc_w, int cyc_r, int cyc_compact) {
    // PANEL part of the front only (front-local columns < c). The contribution-block part is
    // assembled inside k_syrk_cb (children gathered into an LDS tile, CB written exactly once).
    // WIDE: a column of a top-of-tree front has thousands of rows and the level only has a handful
    // of fronts -- the whole workgroup shares one column (a quarter of the dependent round trips
    // per wave); the phases are then separated by barriers (different waves touch the same rows).
    constexpr int NL = WIDE ? 256 : 64;          // lanes cooperating on one column
    const int s = list[blockIdx.y];
    const int c = S.sfirst[s + 1] - S.sfirst[s];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tl = WIDE ? (int)threadIdx.x : lane;
    const int tc = WIDE ? (int)blockIdx.x : blockIdx.x * ASM_CW + __builtin_amdgcn_readfirstlane(wave);
    if (tc >= c) return;
    // distributed root (cyc_w > 0): this rank assembles the 256-column blocks it owns, block b on rank b mod cyc_w
    if (cyc_w > 0 && (tc >> 8) % cyc_w != cyc_r) return;
    const int ld = S.ld[s];
    // block-cyclic STORAGE (cyc_compact; round 6): this rank keeps only its own 256-column blocks of the front, one behind the
    // other -- block b at local position b / cyc_w: column tc sits 256 (b - b / cyc_w) columns further down than in the full panel
    const int tcs = cyc_compact ? tc - 256 * ((tc >> 8) - (tc >> 8) / cyc_w) : tc;
    double *Pc = L + S.panelptr[s] + (long long)tcs * ld;
    for (int i = 2 * tl; i < ld; i += 2 * NL) *(d2u *)(Pc + i) = (d2u){0.0, 0.0};      // ld is even
    if (WIDE) __syncthreads();
    {   // Q's entries of this column: [qcolptr[k], qcolptr[k + 1]) for column k of L (no search)
        const int gk = S.sfirst[s] + tc;
        const int lo = S.qcolptr[gk], hi = S.qcolptr[gk + 1];
        for (int q = lo + tl; q < hi; q += NL) Pc[S.qdst[q]] = nzval[S.qsrc[q]];
    }
    if (WIDE) __syncthreads();
    for (long long ch = S.childptr[s]; ch < S.childptr[s + 1]; ch++) {
        const EdgeRec er = S.edge[ch];
        const int md = er.md;
        const int *reld = S.rel + er.reloff;
        const int j = S.erow[er.eoff + tc];       // the child's row that maps to column tc (table, no search)
        if (j < 0) continue;                      // none (workgroup-uniform)
        const double *Uc = CB + er.cboff + (long long)j * md;
        // four independent row chunks in flight per lane (rel -> P read-modify-write chain; row pairs per lane as in
        // k_assemble_lds were measured slower here: the scattered read-modify-write of P is the long pole, not the loads)
        for (int i0 = j + tl; i0 < md; i0 += 4 * NL) {
            int ri[4];
            double u[4], pv[4];
#pragma unroll
            for (int q = 0; q < 4; q++) ri[q] = reld[min(i0 + NL * q, md - 1)];
#pragma unroll
            for (int q = 0; q < 4; q++) u[q] = Uc[min(i0 + NL * q, md - 1)];
#pragma unroll
            for (int q = 0; q < 4; q++) pv[q] = Pc[ri[q]];
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (i0 + NL * q < md) Pc[ri[q]] = pv[q] + u[q];
        }
        if (WIDE) __syncthreads();
    }
}

// Panel rows below the diagonal block as a GEMM with the inverted block (FP64 MFMA):
//   mode 0 (factorisation):      A[i, blk] <- A[i, blk] * Linv'      (in place, = L21 rows)
//   mode 1 (selected inversion): Yh[i, :]  <- L[i, blk] * Linv
// One wave owns 16 rows (reads all of them before it writes), a workgroup 64 rows.
template <int MODE, int SPLIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void k_trsm(DevSym S, const FrontView *__restrict__ frec, int kb,
                                              double *__restrict__ L, double *__restrict__ Yh,
                                              const long long *__restrict__ yoff, FrontArg fa) {
    // SPLIT = 0: a workgroup owns 128 rows, each wave 32 of them (all four 16-column tiles) as 16 row PAIRS: MFMA
    // row lm of tile 0 / 1 is row 2 lm / 2 lm + 1 of the wave's 32, so one 16-byte load per lane and k-step feeds both
    // tiles and the results leave 16 bytes at a time (half the vector memory instructions, half the LDS reads and half
    // the stagings of the inverse block per row; the kernel streams the block column once in, once out);
    // SPLIT = 1 (latency variant for levels with a handful of fronts): a workgroup owns 16 rows
    // and each wave ONE column tile of them -- many more workgroups, a quarter of the MFMA
    // chain per wave (a single CU sustains only ~0.14 TFLOP/s of FP64 MFMA).
    __shared__ double Ti[NB * NB];
    const FrontView fv = front_view(frec, blockIdx.y, fa);
    const int s = fv.s, c = fv.c, r = fv.r;
    if (kb >= c) return;
    const int w = min(NB, c - kb);
    const int row0 = kb + w + blockIdx.x * (SPLIT ? 16 : 128);
    if (row0 >= r) return;
    const int ld = fv.ld;
    double *Pp = L + fv.pp;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lm = lane & 15, lk = lane >> 4;
    const double *A = Pp + (long long)kb * ld;
    double *out = MODE == 0 ? Pp + (long long)kb * ld : Yh + yoff[s];
    const int ldo = MODE == 0 ? ld : r;
    // Linv is lower triangular: MODE 0 (A Linv') needs q <= k, MODE 1 (L Linv) needs q >= k
    if (SPLIT) {
        const int i0 = row0;
        const int i = i0 + lm;
        const double *pa = A + min(i, r - 1);
        // this wave's rows of the block column are requested BEFORE the inverse block is staged: the two global
        // round trips of this latency-bound kernel overlap instead of following each other
        double bv[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int q = 4 * u + lk;
            bv[u] = pa[(long long)min(q, w - 1) * ld];   // B[kk=q][n=i]; Ti is zero for q >= w, rows >= r never stored
        }
        stage_linv(Pp + kb + (long long)kb * ld, ld, w, Ti, threadIdx.x);
        __syncthreads();
        const int t = wave;
        d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
        const int k = t * 16 + lm;
        const int ulo = MODE == 0 ? 0 : 4 * t, uhi = MODE == 0 ? 4 * t + 4 : 16;
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int q = 4 * u + lk;                                             // A[m=k][kk=q]
            const double av = MODE == 0 ? Ti[k * NB + q] : Ti[q * NB + k];
            if (u >= ulo && u < uhi && 4 * u < w) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[u], acc, 0, 0, 0);
        }
        __syncthreads();   // in place: the other waves read the columns this wave overwrites
        if (i >= r) return;
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int kk = t * 16 + lk + 4 * rr;
            if (kk < w) out[i + (long long)kk * ldo] = acc[rr];
        }
    } else {
        const int i0 = row0 + wave * 32;
        const int i = i0 + 2 * lm;                     // this lane's row pair: i, i + 1
        // (lanes past the last row re-read the last row's pair; its second half is padding, the next column's first
        //  entry or the slack behind the array -- never stored)
        const double *pa = A + min(i, r - 1);
        d2u bv[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int q = 4 * u + lk;
            bv[u] = *(const d2u *)(pa + (long long)min(q, w - 1) * ld);
        }
        stage_linv(Pp + kb + (long long)kb * ld, ld, w, Ti, threadIdx.x);
        __syncthreads();
        if (i0 >= r) return;
        d4 acc[2][4];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int t = 0; t < 4; t++) acc[a][t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int q = 4 * u + lk;
            if (4 * u < w) {
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    if (MODE == 0 ? (u <= 4 * t + 3) : (u >= 4 * t)) {
                        const int k = t * 16 + lm;                                    // A[m=k][kk=q]
                        const double av = MODE == 0 ? Ti[k * NB + q] : Ti[q * NB + k];
                        acc[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[u].x, acc[0][t], 0, 0, 0);
                        acc[1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[u].y, acc[1][t], 0, 0, 0);
                    }
                }
            }
        }
        if (i >= r) return;
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int k = t * 16 + lk + 4 * rr;
                if (k < w) {
                    double *dst = out + i + (long long)k * ldo;
                    if (i + 1 < r) *(d2u *)dst = (d2u){acc[0][t][rr], acc[1][t][rr]};
                    else dst[0] = acc[0][t][rr];
                }
            }
    }
}

// The same product for blocks at most 32 columns wide (the panels of the wide levels of the tree: thousands of narrow fronts per
// launch). The general kernel holds 16 k-steps of operands and 8 accumulator tiles per wave and lives on three waves per SIMD;
// half of that is never used here. 8 k-steps, 4 tiles, an 8 KB inverse block: five to six waves per SIMD. The k-steps that exist
// are issued in the same order with the same operands: the same bits.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void k_trsm_narrow(const FrontView *__restrict__ frec, int kb, double *__restrict__ L) {
    constexpr int W = 32;
    __shared__ double Ti[W * W];
    const FrontView fv = front_view(frec, blockIdx.y, FrontArg{0, 0, 0, 0, 0, 0, 0});
    const int c = fv.c, r = fv.r;
    if (kb >= c) return;
    const int w = min(NB, c - kb);          // <= 32 by the launch's contract
    const int row0 = kb + w + blockIdx.x * 128;
    if (row0 >= r) return;
    const int ld = fv.ld;
    double *Pp = L + fv.pp;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lm = lane & 15, lk = lane >> 4;
    double *A = Pp + (long long)kb * ld;
    const int i0 = row0 + wave * 32;
    const int i = i0 + 2 * lm;                     // this lane's row pair: i, i + 1
    const double *pa = A + min(i, r - 1);
    d2u bv[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const int q = 4 * u + lk;
        bv[u] = *(const d2u *)(pa + (long long)min(q, w - 1) * ld);
    }
    {   // the inverse block: Ti[k * W + q] = Linv[k][q] (stored transposed in the strict upper triangle; diag = 1 / L[k][k])
        const double *Dg = Pp + kb + (long long)kb * ld;
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int idx = threadIdx.x + 256 * u;
            const int q = idx % W, k = idx / W;
            const int qq = min(q, w - 1), kk = min(k, w - 1);
            v[u] = Dg[min(qq, kk) + (long long)max(qq, kk) * ld];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int idx = threadIdx.x + 256 * u;
            const int q = idx % W, k = idx / W;
            const double mk = (k < w && q < k) ? 1.0 : 0.0;
            double x = v[u] * mk;
            if (q == k && k < w) x = fast_rcp(v[u]);
            Ti[k * W + q] = x;
        }
    }
    __syncthreads();
    if (i0 >= r) return;
    d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int t = 0; t < 2; t++) acc[a][t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const int q = 4 * u + lk;
        if (4 * u < w) {
#pragma unroll
            for (int t = 0; t < 2; t++) {
                if (u <= 4 * t + 3) {
                    const int k = t * 16 + lm;                                    // A[m=k][kk=q]
                    const double av = Ti[k * W + q];
                    acc[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[u].x, acc[0][t], 0, 0, 0);
                    acc[1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[u].y, acc[1][t], 0, 0, 0);
                }
            }
        }
    }
    if (i >= r) return;
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int k = t * 16 + lk + 4 * rr;
            if (k < w) {
                double *dst = A + i + (long long)k * ld;
                if (i + 1 < r) *(d2u *)dst = (d2u){acc[0][t][rr], acc[1][t][rr]};
                else dst[0] = acc[0][t][rr];
            }
        }
}

// C[i,j] -= sum_k A[i,k] * B[j,k]  on 64x64 tiles (4 waves x 32x32), FP64 MFMA, operands read
// straight from HBM/L2. The MFMA is issued "transposed" (first operand = rows of B) so that
// the 16 lanes sharing a register index walk down a COLUMN of the column-major C.
// Trailing update inside the panel (see the comment in the kernel); the contribution block is k_syrk_cb.
// CB -= L21 L21' (K = all c columns).
template <int TW>   // MFMA tiles per wave and dimension: wave tile 16*TW squared, workgroup tile twice that
__global__ __launch_bounds__(256) void k_gemm_nt(DevSym S, const FrontView *__restrict__ frec, int k0, int K, int c0, int c1,
                                                 double *__restrict__ L, FrontArg fa) {
    // panel columns [c0, min(c1, c)) of the front, rows c0 .. r-1:  C -= A A'  with A = the K
    // (finished) panel columns k0 .. k0+K-1 of those rows. Two-level blocking: K = 64 updates stay
    // inside the current 256-column block, the rest of the panel is updated once per 256 columns
    // with K = 256 (a quarter of the read-modify-write traffic of a flat right-looking sweep).
    const FrontView fv = front_view(frec, blockIdx.z, fa);
    const int c = fv.c;
    if (c0 >= c) return;
    const int r = fv.r;
    const int ld = fv.ld;
    double *P = L + fv.pp;
    const int M = r - c0, N = min(c1, c) - c0, ldc = ld;
    const double *A = (fa.on && fa.ppa != kNoPpa ? L + fa.ppa : P) + c0 + (long long)k0 * ld;
    double *C = P + c0 + (long long)c0 * ld;
    const int bi = blockIdx.x, bj = blockIdx.y;
    constexpr int WT = 16 * TW, GT = 2 * WT;
    if (bj > bi || bi * GT >= M || bj * GT >= N) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i0 = bi * GT + (wave & 1) * WT, j0 = bj * GT + (wave >> 1) * WT;
    if (i0 >= M || j0 >= N || j0 > i0 + WT - 1) return;
    const int lm = lane & 15, lk = lane >> 4;
    d4 acc[TW][TW];
#pragma unroll
    for (int a = 0; a < TW; a++)
#pragma unroll
        for (int b = 0; b < TW; b++) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
    // Operand rows are clamped (always-valid addresses, values masked afterwards) so that the
    // loads of a whole batch of KU k-steps issue back to back; the next batch is fetched into
    // a second register set before the current batch's MFMAs (software double buffering).
    constexpr int KU = TW == 2 ? 4 : 16;      // TW = 1 (latency variant): a K = 64 update is ONE batch -- one round trip for all operands
    // TW = 2: operand rows in PAIRS -- MFMA row lm of tile 0 / 1 is row 2 lm / 2 lm + 1 of the wave's 32 -- so one 16-byte
    // load per lane feeds both tiles, and C is read and written 16 bytes at a time as well: half the vector memory
    // instructions (the CU's address unit, not the MFMA pipe, is the busiest unit of this kernel). Lanes past the last
    // row re-read the last pair; the odd row after an odd count is padding or the next column's first entry.
    const int Mlast = (M - 1) & ~1, Nlast = (N - 1) & ~1;
    const double *pa[TW], *pb[TW];
#pragma unroll
    for (int a = 0; a < TW; a++) pa[a] = A + (TW == 2 ? min(i0 + 2 * lm, Mlast) : min(i0 + a * 16 + lm, M - 1));
#pragma unroll
    for (int b = 0; b < TW; b++) pb[b] = A + (TW == 2 ? min(j0 + 2 * lm, Nlast) : min(j0 + b * 16 + lm, N - 1));
    double ca[KU][TW], cb[KU][TW];
    auto load_step = [&](long long off, double (&xa)[TW], double (&xb)[TW]) {
        if constexpr (TW == 2) {
            const d2u va = *(const d2u *)(pa[0] + off), vb = *(const d2u *)(pb[0] + off);
            xa[0] = va.x; xa[1] = va.y; xb[0] = vb.x; xb[1] = vb.y;
        } else {
#pragma unroll
            for (int a = 0; a < TW; a++) xa[a] = pa[a][off];
#pragma unroll
            for (int b = 0; b < TW; b++) xb[b] = pb[b][off];
        }
    };
    // full batches: no masking at all, so the prefetch of batch k+1 really overlaps the MFMAs of
    // batch k (nothing consumes the loaded registers before the MFMAs that need them)
    auto fetch = [&](int k0, double (&xa)[KU][TW], double (&xb)[KU][TW]) {
#pragma unroll
        for (int u = 0; u < KU; u++) load_step((long long)(k0 + 4 * u + lk) * ld, xa[u], xb[u]);
    };
    auto mma = [&](double (&xa)[KU][TW], double (&xb)[KU][TW]) {
#pragma unroll
        for (int u = 0; u < KU; u++)
#pragma unroll
            for (int a = 0; a < TW; a++)
#pragma unroll
                for (int b = 0; b < TW; b++)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[u][b], xa[u][a], acc[a][b], 0, 0, 0);
    };
    // TW = 1: the tile's own values are requested BEFORE the operands (they only meet in the epilogue): one round trip less
    // on the latency-bound levels this variant serves
    double cv[TW][TW][4];
    if constexpr (TW != 2) {
#pragma unroll
        for (int a = 0; a < TW; a++)
#pragma unroll
            for (int b = 0; b < TW; b++)
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const int i = min(i0 + a * 16 + lm, M - 1);
                    const int j = min(j0 + b * 16 + lk + 4 * rr, N - 1);
                    cv[a][b][rr] = C[i + (long long)j * ldc];
                }
    }
    const int kfull = K / (4 * KU) * (4 * KU);
    // single-buffered batches: latency is hidden by the other resident waves (4-5 per SIMD at
    // this register budget); hipcc turns a register double-buffer into vmcnt(0) at the loop head
    // anyway, which defeats the overlap. (Round 6: the three-stage loop of k_syrk_cb_rec<true> -- unconditional refills, scheduling
    // barriers -- does overlap; built here for K >= 48, bit-identical, and dropped: factor 8.44 / 8.44 -> 8.34 / 8.46 ms on one box, inside
    // the noise -- these launches sit on the panel chain and are bounded by their own latency, not by the product loop.)
    for (int k0 = 0; k0 < kfull; k0 += 4 * KU) {
        fetch(k0, ca, cb);
        mma(ca, cb);
    }
    if (kfull < K) {   // masked tail (k beyond K contributes 0 via an arithmetic mask on one operand;
                       // a select would let the compiler sink the load under a branch)
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const int kk = kfull + 4 * u + lk;
            const double mk = kk < K ? 1.0 : 0.0;
            load_step((long long)min(kk, K - 1) * ld, ca[u], cb[u]);
#pragma unroll
            for (int a = 0; a < TW; a++) ca[u][a] *= mk;
        }
        mma(ca, cb);
    }
    // D[m][n]: m (rows of the first operand = C's column) = lk + 4*reg, n = lm = C's row
    // read-modify-write of C in two passes (all loads, then all stores): one round trip instead of
    // a chain of 4 TW^2 (the compiler cannot reorder a load of C past the previous store to C)
    if constexpr (TW == 2) {
        // tile a of the rows = row i0 + 2 lm + a, tile b of the columns = column j0 + 2 (lk + 4 rr) + b
        d2u cv[2][4];
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int j = min(j0 + 2 * (lk + 4 * rr) + b, N - 1);
                cv[b][rr] = *(const d2u *)(C + min(i0 + 2 * lm, Mlast) + (long long)j * ldc);
            }
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int i = i0 + 2 * lm, j = j0 + 2 * (lk + 4 * rr) + b;
                const bool v0 = i < M && j < N && i >= j, v1 = i + 1 < M && j < N && i + 1 >= j;
                double *dst = C + i + (long long)j * ldc;
                const double x0 = cv[b][rr].x - acc[0][b][rr], x1 = cv[b][rr].y - acc[1][b][rr];
                if (v0 && v1) *(d2u *)dst = (d2u){x0, x1};
                else {
                    if (v0) dst[0] = x0;
                    if (v1) dst[1] = x1;
                }
            }
    } else {
#pragma unroll
        for (int a = 0; a < TW; a++)
#pragma unroll
            for (int b = 0; b < TW; b++)
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const int i = i0 + a * 16 + lm;
                    const int j = j0 + b * 16 + lk + 4 * rr;
                    if (i < M && j < N && i >= j) C[i + (long long)j * ldc] = cv[a][b][rr] - acc[a][b][rr];
                }
    }
}

// Contribution block of a big front, written ONCE:  CB = (extend-add of the children's CBs) - L21 L21'.
// One workgroup per 64x64 lower tile: the children's entries that fall into the tile are gathered
// into an LDS tile (fixed child order, no atomics), the product runs on the FP64 MFMA, the
// epilogue stores LDS tile minus accumulators. No zero-fill, no read-modify-write of CB in HBM.
// (Reference form on a plain 3-D grid, front x tile row x tile column: GMRFX_SYRK_XCD=0. The product path is
// k_syrk_cb_rec below -- same arithmetic, tiles handed out per XCD from self-contained records.)
__global__ __launch_bounds__(256) void k_syrk_cb(DevSym S, const int *__restrict__ list, const double *__restrict__ L,
                                                 double *__restrict__ CB, int cyc_w, int cyc_r, int cyc_b0) {
    __shared__ double Tl[64 * 65];
    const int s = list[blockIdx.z], bi = blockIdx.x, bj = blockIdx.y;
    const int c = S.sfirst[s + 1] - S.sfirst[s];
    const int r = (int)(S.rowptr[s + 1] - S.rowptr[s]);
    const int m = r - c;
    if (bj > bi || bi * 64 >= m) return;
    // distributed front (cyc_w > 0): this rank computes the 256-column blocks of the contribution block it owns -- block q on
    // position (cyc_b0 + q) mod cyc_w of the group, cyc_b0 = the panel's blocks (the dealing continues behind the panel)
    if (cyc_w > 0 && (cyc_b0 + (bj >> 2)) % cyc_w != cyc_r) return;
    const int ld = S.ld[s];
    const double *A = L + S.panelptr[s] + c;
    double *C = CB + S.cbptr[s];
    const int tid = threadIdx.x;
    const int ti0 = bi * 64, tj0 = bj * 64;      // tile origin inside CB
    for (int idx = tid; idx < 64 * 65; idx += 256) Tl[idx] = 0.0;
    __syncthreads();
    {
        // Children two at a time: edge records and tile ranges of both first (two round trips for
        // the pair), then 16 entries per thread and child with all loads in flight at once. The
        // children are still ADDED one after the other (fixed order, bit-reproducible).
        const long long ch0 = S.childptr[s], ch1 = S.childptr[s + 1];
        const int nT = (m + 31) >> 5;
        const int la = tid & 63, lb = tid >> 6;
        for (long long cb = ch0; cb < ch1; cb += 2) {
            EdgeRec er[2];
            int a0[2], a1[2], b0[2], b1[2];
#pragma unroll
            for (int q = 0; q < 2; q++) er[q] = S.edge[min(cb + q, ch1 - 1)];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int *et = S.etile + er[q].tptr;
                a0[q] = et[2 * bi]; a1[q] = et[min(2 * bi + 2, nT)];
                b0[q] = et[2 * bj]; b1[q] = et[min(2 * bj + 2, nT)];
            }
#pragma unroll
            for (int q = 0; q < 2; q++) {
                if (cb + q < ch1) {
                    const int md = er[q].md;
                    const int *reld = S.rel + er[q].reloff;
                    const double *Ud = CB + er[q].cboff;
                    const int a = a0[q] + la, ac = min(a, md - 1);
                    const int ti = reld[ac] - c - ti0;
                    int tb[16];
                    double uv[16];
#pragma unroll
                    for (int u = 0; u < 16; u++) {
                        const int b = min(b0[q] + lb + 4 * u, md - 1);
                        tb[u] = reld[b];
                        uv[u] = Ud[ac + (long long)b * md];
                    }
#pragma unroll
                    for (int u = 0; u < 16; u++) {
                        const int b = b0[q] + lb + 4 * u;
                        if (a < a1[q] && b < b1[q] && a >= b) Tl[ti + (tb[u] - c - tj0) * 65] += uv[u];
                    }
                    __syncthreads();
                }
            }
        }
    }
    const int wave = tid >> 6, lane = tid & 63;
    const int i0 = ti0 + (wave & 1) * 32, j0 = tj0 + (wave >> 1) * 32;
    if (i0 >= m || j0 >= m || j0 > i0 + 31) return;
    const int lm = lane & 15, lk = lane >> 4;
    d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
    auto fa = [&](int j, int q) { return A[min(j, m - 1) + (long long)min(max(q, 0), c - 1) * ld]; };
    auto fb = [&](int q, int i) { return A[min(i, m - 1) + (long long)min(max(q, 0), c - 1) * ld]; };
    // D[m_ = j][n = i] = sum_q L21[j][q] L21[i][q]: rows i on the lanes (contiguous in column-major CB)
    wave_gemm_32x32(acc, j0, i0, 0, c, fa, fb, lm, lk);
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int j = j0 + a * 16 + lk + 4 * rr, i = i0 + b * 16 + lm;
                if (i < m && j < m && i >= j) C[i + (long long)j * m] = Tl[(i - ti0) + (j - tj0) * 65] - acc[a][b][rr];
            }
}

// The product form of the same tile, driven by one 128-byte record per tile (SyrkTile, device.h) and handed out per
// XCD: workgroups go to the 8 XCDs round-robin by linear id, so id & 7 is the XCD and id >> 3 the position in that
// XCD's run of the level's tile list. A run holds whole fronts or compact 8 x 8-tile squares of one front, so the L21
// row blocks its tiles share are fetched into ONE L2 instead of all eight (L2-miss traffic of the launches of one
// factorisation: 17.2 GB on the 3-D grid, 6.95 GB here, 5.77 GB algorithmic -- tools/syrk_levels.py traffic).
// A tile of a narrow front is a chain of round trips, not arithmetic: the record arrives in one scalar load (instead of
// tile -> front geometry -> edge records -> tile ranges), the first k-batch of the product and the first child's entries
// are requested right behind it, and only then does anything wait. Children are still added one after the other, the k
// order is unchanged: bit-identical to k_syrk_cb.
struct SyrkOps { d2u a[2], b[2]; };       // the operands of two k-steps (rows in pairs): one stage of the pipelined product loop
template <bool PIPED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void k_syrk_cb_rec(DevSym S, const SyrkTile *__restrict__ recs, const SyrkSplit split,
                                                     const double *__restrict__ L, double *__restrict__ CB, int noprod) {
    // noprod: the children's extend-add only -- the product follows as its own launch on 128 x 128 staged tiles (k_syrk_big: the
    // huge fronts of 3-D problems)
    __shared__ double Tl[64 * 65];
    const int x = blockIdx.x & 7;
    const int t = split.start[x] + (int)(blockIdx.x >> 3);
    if (t >= split.start[x + 1]) return;
    const SyrkTile T = recs[t];
    const int c = T.c, m = T.m, ld = T.ld, bi = T.bi, bj = T.bj;
    const double *A = L + T.pa;
    double *C = CB + T.cb;
    const int tid = threadIdx.x;
    const int ti0 = bi * 64, tj0 = bj * 64;
    const int wave = tid >> 6, lane = tid & 63;
    const int lm = lane & 15, lk = lane >> 4;
    const int i0 = ti0 + (wave & 1) * 32, j0 = tj0 + (wave >> 1) * 32;
    const bool live = !(i0 >= m || j0 >= m || j0 > i0 + 31);       // this wave's 32 x 32 part reaches the lower triangle
    constexpr int KU = 4;
    // Operand rows in PAIRS: MFMA row lm of tile 0 / tile 1 is row 2 lm / 2 lm + 1 of the wave's 32 (not lm / 16 + lm),
    // so one 16-byte load per lane feeds both tiles -- half the vector memory instructions of the k-loop, which is
    // what these kernels are bound by (see above). Lanes past the last row re-read the last pair (never stored); the
    // odd row after an odd m is padding or the next column's first entry (never stored either).
    const int mlast = (m - 1) & ~1;
    const double *pa2 = A + min(j0 + 2 * lm, mlast);
    const double *pb2 = A + min(i0 + 2 * lm, mlast);
    double av[KU][2], bv[KU][2];
    auto request = [&](int q0) {
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const long long ko = (long long)min(q0 + 4 * u + lk, c - 1) * ld;
            const d2u xa = *(const d2u *)(pa2 + ko), xb = *(const d2u *)(pb2 + ko);
            av[u][0] = xa.x; av[u][1] = xa.y;
            bv[u][0] = xb.x; bv[u][1] = xb.y;
        }
    };
    // Tiles of WIDE fronts (c >= pipe_min): the product loop in three stages of two k-steps, each stage requested two stages ahead
    // (see the loop below)
    constexpr bool piped = PIPED;
    SyrkOps oA, oB, oC;
    // Addresses without vector arithmetic: a SCALAR base per k-step (the record is the same for every lane) + a 32-bit lane offset
    // (the lane's row pair + its column lk of the k-step). Requests behind the last k-step are clamped to the last four columns of
    // the PADDED panel (symbolic.cpp, panel_span: zero columns up to a multiple of 4): in vain, or masked in the tail.
    const unsigned voa = (unsigned)(min(j0 + 2 * lm, mlast) + lk * ld) * 8u, vob = (unsigned)(min(i0 + 2 * lm, mlast) + lk * ld) * 8u;
    const int cp4 = ((c + 3) & ~3) - 4;
    auto req2 = [&](SyrkOps &x, int q0) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const char *sb = (const char *)(A + (long long)min(q0 + 4 * u, cp4) * ld);
            x.a[u] = *(const d2u *)(sb + voa); x.b[u] = *(const d2u *)(sb + vob);
        }
    };
    if (live && !noprod) {
        if (piped) { req2(oA, 0); req2(oB, 8); req2(oC, 16); }
        else request(0);
    }
    // the first two children's entries: (row la, column lb + 4 u) of the child's rows / columns inside this tile.
    // Round trips: record -> [k-batch 0 + child 0] -> child 1 -> k-batch 1 ...
    // Every vector memory instruction costs the CU's address unit ~16 cycles whatever its lanes do, and these levels
    // are bound by exactly that (TA busy 87 %): so the child's column indices come in ONE load (lane l holds the
    // index of column b0 + l; each use reads its lane), columns beyond the tile's range issue nothing at all, and
    // neither do lanes beyond its row range.
    const int la = lane;
    const int lb = __builtin_amdgcn_readfirstlane(wave);
    // BOTH of the first two children are requested up front (two register sets): the second child's round trip used to start
    // only after the first child had been added -- on the narrow fronts of the mid levels a tile is little else than these
    // round trips
    int ti[2], rb[2];
    double uv[2][16];
    auto fetch = [&](int q) {
        const int md = T.md[q];
        const int *reld = S.rel + T.reloff[q];
        const double *Ud = CB + T.cboff[q];
        const int a = T.a0[q] + la;
        const int ac = min(a, md - 1);
        ti[q] = reld[ac];
        rb[q] = reld[min(T.b0[q] + la, md - 1)];
        const int nb = T.b1[q] - T.b0[q] - lb;          // this wave's columns: b0 + lb + 4 u < b1  <=>  4 u < nb
        if (a < T.a1[q]) {
#pragma unroll
            for (int u = 0; u < 16; u++)
                if (4 * u < nb) uv[q][u] = Ud[ac + (long long)(T.b0[q] + lb + 4 * u) * md];
        }
    };
    auto add = [&](int q) {
        const int a = T.a0[q] + la;
        const int nb = T.b1[q] - T.b0[q] - lb;
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (4 * u < nb) {
                const int tc = __builtin_amdgcn_readlane(rb[q], lb + 4 * u) - c - tj0;
                if (a < T.a1[q] && a >= T.b0[q] + lb + 4 * u) Tl[(ti[q] - c - ti0) + tc * 65] += uv[q][u];
            }
        }
    };
    d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
    auto mfma_batch = [&](int q0) {
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const double mk = (q0 + 4 * u + lk) < c ? 1.0 : 0.0;
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][a] * mk, bv[u][b], acc[a][b], 0, 0, 0);
        }
    };
    if (T.nch > 0 && !piped) fetch(0);        // (piped: behind the product -- its three operand stages take the registers of a child's entries)
    for (int idx = tid; idx < 64 * 65; idx += 256) Tl[idx] = 0.0;
    // The product needs nothing from the children: it runs HERE, between the children's requests and their use, so that its
    // MFMAs cover the children's round trips (the accumulators meet the gathered tile only in the epilogue). Measured by
    // compiling parts out (profiles/r04_syrk_parts.txt): gather, product and store used to follow each other, a third of a
    // mid-level tile's time each.
    // D[m_ = j][n = i] = sum_q L21[j][q] L21[i][q]: rows i on the lanes (contiguous in column-major CB)
    auto mma2 = [&](const SyrkOps &x, int q0, bool masked) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const double mk = (!masked || (q0 + 4 * u + lk) < c) ? 1.0 : 0.0;
            const double a0 = masked ? x.a[u].x * mk : x.a[u].x, a1 = masked ? x.a[u].y * mk : x.a[u].y;
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, x.b[u].x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, x.b[u].y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, x.b[u].x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, x.b[u].y, acc[1][1], 0, 0, 0);
        }
    };
    if (live && !noprod) {
        if (piped) {
            // same k-steps in the same order as the plain loop (bit-identical sums); a stage's registers are refilled as soon as
            // its MFMAs have read them, two stages (16 MFMAs) before they are used again. The refills are UNCONDITIONAL (clamped
            // rows: the last ones of a tile are requested in vain): a conditional request makes the compiler count zero loads behind
            // every stage, i.e. wait for everything in flight.
            int q0 = 0;
            for (; q0 + 24 <= c; q0 += 24) {
                // (the scheduler would sink all refills to the end of the iteration: one exposed round trip per iteration again)
                mma2(oA, q0, false); __builtin_amdgcn_sched_barrier(0); req2(oA, q0 + 24); __builtin_amdgcn_sched_barrier(0);
                mma2(oB, q0 + 8, false); __builtin_amdgcn_sched_barrier(0); req2(oB, q0 + 32); __builtin_amdgcn_sched_barrier(0);
                mma2(oC, q0 + 16, false); __builtin_amdgcn_sched_barrier(0); req2(oC, q0 + 40); __builtin_amdgcn_sched_barrier(0);
            }
            if (q0 < c) {
                mma2(oA, q0, true);
                if (q0 + 8 < c) {
                    mma2(oB, q0 + 8, true);
                    if (q0 + 16 < c) mma2(oC, q0 + 16, true);
                }
            }
        } else
            for (int q0 = 0; q0 < c; q0 += 4 * KU) {
                if (q0 > 0) request(q0);
                mfma_batch(q0);
            }
    }
    if (T.nch > 0 && piped) fetch(0);
    if (T.nch > 1) fetch(1);        // (the second child's registers would not fit beside the product's: behind it, before the first is added)
    __syncthreads();
    if (T.nch > 0) {
        add(0);
        __syncthreads();
    }
    if (T.nch > 1) {
        add(1);
        __syncthreads();
    }
    if (T.nch > 2) {       // further children: edge record -> tile ranges -> entries, one child at a time
        const int nT = (m + 31) >> 5;
        for (long long cb = T.ch0 + 2; cb < T.ch0 + T.nch; cb++) {
            const EdgeRec er = S.edge[cb];
            const int *et = S.etile + er.tptr;
            const int a0 = et[2 * bi], a1 = et[min(2 * bi + 2, nT)], b0 = et[2 * bj], b1 = et[min(2 * bj + 2, nT)];
            const int md = er.md;
            const int *reld = S.rel + er.reloff;
            const double *Ud = CB + er.cboff;
            const int a = a0 + la, ac = min(a, md - 1);
            const int tr = reld[ac] - c - ti0;
            int tc[16];
            double w[16];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int b = min(b0 + lb + 4 * u, md - 1);
                tc[u] = reld[b];
                w[u] = Ud[ac + (long long)b * md];
            }
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int b = b0 + lb + 4 * u;
                if (a < a1 && b < b1 && a >= b) Tl[tr + (tc[u] - c - tj0) * 65] += w[u];
            }
            __syncthreads();
        }
    }
    if (!live) return;
    // rows i, i + 1 of column j leave together (16 bytes) wherever both lie inside the lower triangle
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int j = j0 + 2 * (lk + 4 * rr) + a, i = i0 + 2 * lm;
            if (j < m) {
                const double *tl = Tl + (i - ti0) + (j - tj0) * 65;
                double *dst = C + i + (long long)j * m;
                const bool v0 = i < m && i >= j, v1 = i + 1 < m && i + 1 >= j;
                if (v0 && v1) *(d2u *)dst = (d2u){tl[0] - acc[a][0][rr], tl[1] - acc[a][1][rr]};
                else {
                    if (v0) dst[0] = tl[0] - acc[a][0][rr];
                    if (v1) dst[1] = tl[1] - acc[a][1][rr];
                }
            }
        }
}

// ------------------------------------------------------------------------------------------
// Triangular sweeps, X row-major (ldx doubles per row), nr <= 64 right-hand sides per pass
// ------------------------------------------------------------------------------------------

// Forward: add the children's update vectors into this front's OWN rows of X (the trailing rows, W_s, are assembled
// inside k_fwd_update_longk and written once). A workgroup owns FWD_RB own rows x 64 right-hand sides; which row of a
// child lands in own row tc comes from the per-edge table DevSym::erow (no search), the rows of X are read once,
// receive the children one after the other (fixed order) in registers and are written once.
__global__ __launch_bounds__(256) void k_fwd_assemble(DevSym S, const int *__restrict__ list, double *__restrict__ X,
                                                      const double *__restrict__ W, int nr, int ldx) {
    const int s = list[blockIdx.y];
    const int c = S.sfirst[s + 1] - S.sfirst[s];
    const int i0 = blockIdx.x * FWD_RB;
    if (i0 >= c) return;
    const long long ch0 = S.childptr[s], ch1 = S.childptr[s + 1];
    if (ch0 == ch1) return;
    const int first = S.sfirst[s];
    const int j = threadIdx.x & 63, g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (j >= nr) return;
    constexpr int NU = FWD_RB / 4;                       // rows per thread: i0 + g + 4 u
    double x[NU];
#pragma unroll
    for (int u = 0; u < NU; u++) x[u] = X[(long long)(first + min(i0 + g + 4 * u, c - 1)) * ldx + j];
    for (long long ch = ch0; ch < ch1; ch++) {
        const EdgeRec er = S.edge[ch];
        const int *er_row = S.erow + er.eoff;
        const double *Wd = W + er.woff * ldx;
        int jr[NU];
#pragma unroll
        for (int u = 0; u < NU; u++) jr[u] = er_row[min(i0 + g + 4 * u, c - 1)];      // wave-uniform
        double v[NU];
#pragma unroll
        for (int u = 0; u < NU; u++) v[u] = jr[u] >= 0 ? Wd[(long long)jr[u] * ldx + j] : 0.0;
#pragma unroll
        for (int u = 0; u < NU; u++) x[u] += v[u];
    }
#pragma unroll
    for (int u = 0; u < NU; u++)
        if (i0 + g + 4 * u < c) X[(long long)(first + i0 + g + 4 * u) * ldx + j] = x[u];
}

// Forward update of a big front after y = L11^-1 b: W_s = (children) - L21 y with K = all c columns.
// A workgroup owns 32
// trailing rows x 64 right-hand sides, every wave sweeps a quarter of the K range for the WHOLE
// tile (16 MFMA tiles per k-step from 4 + 4 operand loads; 512-B contiguous panel segments per
// column), the partial tiles are summed through LDS and each wave writes one row tile.
template <int NA>   // 16-row tiles per workgroup: 2 normally, 1 for levels with a handful of fronts (twice the
                    // workgroups, half the MFMA chain of each: one CU only sustains ~0.14 TFLOP/s of FP64 MFMA)
__global__ __launch_bounds__(256) void k_fwd_update_longk(DevSym S, const int *__restrict__ list,
                                                         const double *__restrict__ L, double *__restrict__ X,
                                                         double *__restrict__ W, int nr, int ldx, int cmin) {
    __shared__ double red[3 * 16 * 64];
    constexpr int RT = 16 * NA;
    __shared__ double Tl[RT * 64];   // children's contributions to this tile of W_s
    const int s = list[blockIdx.y];
    const int c = S.sfirst[s + 1] - S.sfirst[s];
    if (c <= cmin) return;           // (narrow passes: k_fwd_update_wave has these fronts)
    const int r = (int)(S.rowptr[s + 1] - S.rowptr[s]);
    const int i0 = c + blockIdx.x * RT;
    if (i0 >= r) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lm = lane & 15, lk = lane >> 4;
    const int ld = S.ld[s];
    const double *P = L + S.panelptr[s];
    const double *Yb = X + (long long)S.sfirst[s] * ldx;
    double *Ws = W + S.wptr[s] * ldx;
    // ---- gather the children's update vectors for these rows into LDS (fixed child order, no
    //      atomics): W_s is then written exactly once, with no zero-fill / read-modify-write passes
    {
        const int j = threadIdx.x & 63, g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int jcl = min(j, nr - 1);
        const double jm = j < nr ? 1.0 : 0.0;
        for (int i = g; i < RT; i += 4) Tl[i * 64 + j] = 0.0;
        __syncthreads();
        // children two at a time (see k_syrk_cb): records + tile ranges first, then at most 32
        // child rows per tile and child, all loads in flight at once; added in child order. The target rows of a
        // child come in ONE load (lane l: row a0 + l; each use reads its lane) and rows past the tile's range issue
        // nothing: a vector memory instruction costs the address unit ~16 cycles whatever its lanes do.
        const long long ch0 = S.childptr[s], ch1 = S.childptr[s + 1];
        const int T = (blockIdx.x * RT) >> 5;   // 32-row granularity of the tile table
        for (long long cb = ch0; cb < ch1; cb += 2) {
            EdgeRec er[2];
            int a0[2], a1[2];
#pragma unroll
            for (int q = 0; q < 2; q++) er[q] = S.edge[min(cb + q, ch1 - 1)];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                a0[q] = S.etile[er[q].tptr + T];
                a1[q] = S.etile[er[q].tptr + T + 1];
            }
#pragma unroll
            for (int q = 0; q < 2; q++) {
                if (cb + q < ch1) {
                    const int *reld = S.rel + er[q].reloff;
                    const double *Wd = W + er[q].woff * ldx;
                    const int rt = reld[min(a0[q] + j, er[q].md - 1)];
                    const int na = a1[q] - a0[q] - g;            // this wave's rows: a0 + g + 4 u < a1  <=>  4 u < na
                    double wv[8];
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        if (4 * u < na) wv[u] = Wd[(long long)(a0[q] + g + 4 * u) * ldx + jcl];
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        if (4 * u < na) {
                            const int tr = __builtin_amdgcn_readlane(rt, g + 4 * u);
                            if (tr >= i0 && tr < i0 + RT) Tl[(tr - i0) * 64 + j] += wv[u] * jm;
                        }
                    __syncthreads();
                }
            }
        }
    }
    d4 acc[NA][4];
#pragma unroll
    for (int a = 0; a < NA; a++)
#pragma unroll
        for (int t = 0; t < 4; t++) acc[a][t] = (d4){0.0, 0.0, 0.0, 0.0};
    // Operands in PAIRS (16-byte loads): NA = 2: MFMA row lm of row tile 0 / 1 is row 2 lm / 2 lm + 1 of the 32; column
    // tile t is right-hand side 32 (t >> 1) + 2 lm + (t & 1): one load feeds two tiles, three loads per k-step instead
    // of six. Lanes past the last row / right-hand side re-read the last one's pair (results never stored).
    const double *pa = P + (NA == 2 ? min(i0 + 2 * lm, r - 1) : min(i0 + lm, r - 1));
    const int jb[2] = {min(2 * lm, nr - 1), min(32 + 2 * lm, nr - 1)};
    constexpr int KU = 4;
    for (int k0 = wave * 4 * KU; k0 < c; k0 += 16 * KU) {
        double av[KU][NA], bv[KU][4];
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const int kk = k0 + 4 * u + lk;
            const int kc = min(kk, c - 1);
            const double mk = kk < c ? 1.0 : 0.0;
            if constexpr (NA == 2) {
                const d2u x = *(const d2u *)(pa + (long long)kc * ld);
                av[u][0] = x.x * mk; av[u][1] = x.y * mk;
            } else {
                av[u][0] = pa[(long long)kc * ld] * mk;
            }
#pragma unroll
            for (int t2 = 0; t2 < 2; t2++) {
                const d2u y = *(const d2u *)(Yb + (long long)kc * ldx + jb[t2]);
                bv[u][2 * t2] = y.x; bv[u][2 * t2 + 1] = y.y;
            }
        }
#pragma unroll
        for (int u = 0; u < KU; u++)
#pragma unroll
            for (int a = 0; a < NA; a++)
#pragma unroll
                for (int t = 0; t < 4; t++)
                    acc[a][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][a], bv[u][t], acc[a][t], 0, 0, 0);
    }
    if constexpr (NA == 2) {
        // wave w owns row tile w >> 1 (rows i0 + 2 (lk + 4 rr) + (w >> 1)) x right-hand sides 32 (w & 1) + 2 lm, + 1:
        // W leaves 16 bytes per lane
        splitk_reduce4_pairs(acc, red, wave, lane);
        const int a = wave >> 1, hc = wave & 1;
        const int j = 32 * hc + 2 * lm;
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int i = i0 + 2 * (lk + 4 * rr) + a;
            if (i < r && j < nr) {
                double *dst = Ws + (long long)(i - c) * ldx + j;
                const double *tl = Tl + (i - i0) * 64 + j;
                double x0 = 0.0, x1 = 0.0;
#pragma unroll
                for (int aa = 0; aa < 2; aa++)          // (a is wave-uniform; the accumulator index must be a constant)
#pragma unroll
                    for (int h = 0; h < 2; h++)
                        if (aa == a && h == hc) { x0 = tl[0] - acc[aa][2 * h][rr]; x1 = tl[1] - acc[aa][2 * h + 1][rr]; }
                if (j + 1 < nr) *(d2u *)dst = (d2u){x0, x1};
                else dst[0] = x0;
            }
        }
    } else {
        splitk_reduce4<NA>(acc, red, wave, lane);
        // wave w owns column tile w (right-hand sides 32 (w >> 1) + 2 lm + (w & 1))
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (t == wave) {
                const int j = 32 * (t >> 1) + 2 * lm + (t & 1);
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const int i = i0 + lk + 4 * rr;
                    if (i < r && j < nr) Ws[(long long)(i - c) * ldx + j] = Tl[(i - i0) * 64 + j] - acc[0][t][rr];
                }
            }
        }
    }
}

// The same update driven by one 128-byte record per 32-row tile (FwdTile, device.h), handed out in one contiguous run per
// XCD like the contribution-block tiles: a tile of a mid-level front is a chain of round trips (front -> geometry ->
// edge records -> tile ranges -> entries -> k-batches), not arithmetic. Here the record arrives in one scalar load, the
// first k-batch and the first child's entries are requested right behind it. Same sums in the same order as
// k_fwd_update_longk<2>.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void k_fwd_update_rec(DevSym S, const FwdTile *__restrict__ recs, const SyrkSplit split,
                                                        const double *__restrict__ L, const double *__restrict__ X,
                                                        double *__restrict__ W, int nr, int ldx, int cmin) {
    __shared__ double red[3 * 16 * 64];
    __shared__ double Tl[32 * 64];   // children's contributions to this tile of W_s
    const int xcd = blockIdx.x & 7;
    const int tix = split.start[xcd] + (int)(blockIdx.x >> 3);
    if (tix >= split.start[xcd + 1]) return;
    const FwdTile T = recs[tix];
    if (T.c <= cmin) return;
    const int c = T.c, r = T.r, ld = T.ld, i0 = T.i0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lm = lane & 15, lk = lane >> 4;
    const double *P = L + T.pp;
    const double *Yb = X + T.xoff * ldx;
    double *Ws = W + T.woff * ldx;
    // ---- first k-batch of this wave (rows / right-hand sides in pairs, see k_fwd_update_longk)
    const double *pa = P + min(i0 + 2 * lm, r - 1);
    const int jb[2] = {min(2 * lm, nr - 1), min(32 + 2 * lm, nr - 1)};
    constexpr int KU = 4;
    double av[KU][2], bv[KU][4];
    auto request = [&](int k0) {
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const int kc = min(k0 + 4 * u + lk, c - 1);
            const d2u x = *(const d2u *)(pa + (long long)kc * ld);
            av[u][0] = x.x; av[u][1] = x.y;
#pragma unroll
            for (int t2 = 0; t2 < 2; t2++) {
                const d2u y = *(const d2u *)(Yb + (long long)kc * ldx + jb[t2]);
                bv[u][2 * t2] = y.x; bv[u][2 * t2 + 1] = y.y;
            }
        }
    };
    const int kfirst = wave * 4 * KU;
    if (kfirst < c) request(kfirst);
    // ---- the children's update vectors for these rows, gathered into LDS in child order
    const int j = lane, g = __builtin_amdgcn_readfirstlane(wave);
    const int jcl = min(j, nr - 1);
    const double jm = j < nr ? 1.0 : 0.0;
    int rt;
    double wv[8];
    auto fetch = [&](const int *reld, const double *Wd, int md, int a0, int a1) {
        rt = reld[min(a0 + j, md - 1)];
        const int na = a1 - a0 - g;                 // this wave's rows: a0 + g + 4 u < a1  <=>  4 u < na
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (4 * u < na) wv[u] = Wd[(long long)(a0 + g + 4 * u) * ldx + jcl];
    };
    auto add = [&](int a0, int a1) {
        const int na = a1 - a0 - g;
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (4 * u < na) {
                const int tr = __builtin_amdgcn_readlane(rt, g + 4 * u);
                if (tr >= i0 && tr < i0 + 32) Tl[(tr - i0) * 64 + j] += wv[u] * jm;
            }
    };
    if (T.nch > 0) fetch(S.rel + T.reloff[0], W + T.cwoff[0] * ldx, T.md[0], T.a0[0], T.a1[0]);
    for (int i = g; i < 32; i += 4) Tl[i * 64 + j] = 0.0;
    __syncthreads();
    if (T.nch > 0) {
        add(T.a0[0], T.a1[0]);
        __syncthreads();
    }
    if (T.nch > 1) {
        fetch(S.rel + T.reloff[1], W + T.cwoff[1] * ldx, T.md[1], T.a0[1], T.a1[1]);
        add(T.a0[1], T.a1[1]);
        __syncthreads();
    }
    for (long long cb = T.ch0 + 2; cb < T.ch0 + T.nch; cb++) {      // further children: the long way
        const EdgeRec er = S.edge[cb];
        const int a0 = S.etile[er.tptr + T.tile], a1 = S.etile[er.tptr + T.tile + 1];
        fetch(S.rel + er.reloff, W + er.woff * ldx, er.md, a0, a1);
        add(a0, a1);
        __syncthreads();
    }
    d4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int t = 0; t < 4; t++) acc[a][t] = (d4){0.0, 0.0, 0.0, 0.0};
    for (int k0 = kfirst; k0 < c; k0 += 16 * KU) {
        if (k0 > kfirst) request(k0);
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const double mk = (k0 + 4 * u + lk) < c ? 1.0 : 0.0;
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int t = 0; t < 4; t++)
                    acc[a][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][a] * mk, bv[u][t], acc[a][t], 0, 0, 0);
        }
    }
    // wave w owns row tile w >> 1 (rows i0 + 2 (lk + 4 rr) + (w >> 1)) x right-hand sides 32 (w & 1) + 2 lm, + 1
    splitk_reduce4_pairs(acc, red, wave, lane);
    const int a = wave >> 1, hc = wave & 1;
    const int jj = 32 * hc + 2 * lm;
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
        const int i = i0 + 2 * (lk + 4 * rr) + a;
        if (i < r && jj < nr) {
            double *dst = Ws + (long long)(i - c) * ldx + jj;
            const double *tl = Tl + (i - i0) * 64 + jj;
            double x0 = 0.0, x1 = 0.0;
#pragma unroll
            for (int aa = 0; aa < 2; aa++)
#pragma unroll
                for (int h = 0; h < 2; h++)
                    if (aa == a && h == hc) { x0 = tl[0] - acc[aa][2 * h][rr]; x1 = tl[1] - acc[aa][2 * h + 1][rr]; }
            if (jj + 1 < nr) *(d2u *)dst = (d2u){x0, x1};
            else dst[0] = x0;
        }
    }
}

// (Measured and dropped, round 5: the same update with ONE WAVE per record, no LDS and no barrier -- the children's rows entering
//  through the matrix pipe as k-steps against an indicator operand (-1 at the target row), sixteen independent chains per CU
//  instead of three. Forward sweep of cfg 2: 1.96 ms with this kernel, 1.89-1.91 with the wave form on the levels of >= 2 500-6 000
//  tiles: the mid levels already move their bytes -- panel rows, W written once and read once -- at ~4.5 TB/s; what is left is
//  the hand-off of W itself. And its sums round differently from k_fwd_update_longk's, which the sharded rehearsal compares bit for bit.)

// Passes of at most 16 right-hand sides (the single solve, the Newton step): the update of fronts up to `cmax` columns wide with
// ONE WAVE per record, no LDS and no barrier. The 64-column kernels above spend a 1-column pass on the same chain of round trips
// with a barrier between any two of them, three workgroups per CU: 0.71 of the 2.84 ms of a single-RHS solve of cfg 2 went there.
// Here a wave owns the 32 rows x 16 right-hand sides of a record for the whole K range, and the children's update vectors enter
// THROUGH THE MATRIX PIPE: a child row that lands on tile row i is one more k-step whose first operand is the indicator (-1 at
// row i, 0 elsewhere) and whose second operand is the child's row -- the accumulator ends as L21 y - (children), every child row
// added exactly once (a product with 0 adds an exact 0), children in edge order, rows in order: reproducible, and the same for a
// front whatever the level list it comes in (sharded or not). Eight independent chains per SIMD.
// Wider fronts (the top of the tree: K in the hundreds to thousands over a handful of tiles) keep the split-K kernels (cmin).
// NW = 4 (levels with fronts wider than kWaveSplitCols columns): the K range of such a front's tile is split over four waves -- a
// chain of up to 16 dependent batches otherwise --, their partial tiles summed in wave order through 8 KB of LDS; narrower fronts of
// the same launch are still done by wave 0 alone (the rule depends on the front's width only: its sums do not depend on the launch).
constexpr int kWaveSplitCols = 128;
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_fwd_update_wave(DevSym S, const FwdTile *__restrict__ recs, const SyrkSplit split,
                                                        const double *__restrict__ L, const double *__restrict__ X,
                                                        double *__restrict__ W, int nr, int ldx, int cmax) {
    __shared__ double red[NW > 1 ? NW * 8 * 64 : 1];
    // (round 6) blockIdx.y = 16-column tile of the right-hand sides: a pass of 17 .. 32 columns runs these kernels on two tiles
    { const int jt = 16 * blockIdx.y; X += jt; W += jt; nr = min(nr - jt, 16); }
    const int xcd = blockIdx.x & 7;
    const int tix = split.start[xcd] + (int)(blockIdx.x >> 3);
    if (tix >= split.start[xcd + 1]) return;
    const FwdTile T = recs[tix];
    const int c = T.c, r = T.r, ld = T.ld, i0 = T.i0;
    if (c > cmax) return;
    const int wave = NW > 1 ? (int)(threadIdx.x >> 6) : 0;
    const int kw = (NW > 1 && c > kWaveSplitCols) ? NW : 1;       // waves that share this tile's K range
    if (wave >= kw) return;
    const int lane = threadIdx.x & 63;
    const int lm = lane & 15, lk = lane >> 4;
    const double *P = L + T.pp;
    const double *Yb = X + T.xoff * ldx;
    double *Ws = W + T.woff * ldx;
    const double *pa = P + min(i0 + 2 * lm, r - 1);       // rows in pairs: MFMA row lm of row tile 0 / 1 = tile row 2 lm / 2 lm + 1
    const int jl = min(lm, nr - 1);
    constexpr int KU = 8;
    d4 acc[2] = {(d4){0.0, 0.0, 0.0, 0.0}, (d4){0.0, 0.0, 0.0, 0.0}};
    for (int k0 = wave * 4 * KU; k0 < c; k0 += kw * 4 * KU) {
        double av[KU][2], bv[KU];
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const int kk = k0 + 4 * u + lk;
            const int kc = min(kk, c - 1);
            const double mk = kk < c ? 1.0 : 0.0;
            const d2u x = *(const d2u *)(pa + (long long)kc * ld);
            av[u][0] = x.x * mk; av[u][1] = x.y * mk;
            bv[u] = Yb[(long long)kc * ldx + jl];
        }
#pragma unroll
        for (int u = 0; u < KU; u++)
            if (k0 + 4 * u < c) {
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][0], bv[u], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][1], bv[u], acc[1], 0, 0, 0);
            }
    }
    const int myrow = i0 + 2 * lm;
    auto child = [&](const int *__restrict__ reld, const double *__restrict__ Wd, int a0, int a1) {
        for (int b0 = a0; b0 < a1; b0 += 4 * KU) {
            double sv[KU][2], wv[KU];
#pragma unroll
            for (int u = 0; u < KU; u++) {
                const int row = b0 + 4 * u + lk;
                const int rc = min(row, a1 - 1);
                const int d = reld[rc] - myrow;
                const bool ok = row < a1;
                sv[u][0] = (ok && d == 0) ? -1.0 : 0.0;
                sv[u][1] = (ok && d == 1) ? -1.0 : 0.0;
                wv[u] = Wd[(long long)rc * ldx + jl];
            }
#pragma unroll
            for (int u = 0; u < KU; u++)
                if (b0 + 4 * u < a1) {
                    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(sv[u][0], wv[u], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(sv[u][1], wv[u], acc[1], 0, 0, 0);
                }
        }
    };
    if (wave == 0) {
        if (T.nch > 0) child(S.rel + T.reloff[0], W + T.cwoff[0] * ldx, T.a0[0], T.a1[0]);
        if (T.nch > 1) child(S.rel + T.reloff[1], W + T.cwoff[1] * ldx, T.a0[1], T.a1[1]);
        for (long long cb = T.ch0 + 2; cb < T.ch0 + T.nch; cb++) {      // further children: the long way
            const EdgeRec er = S.edge[cb];
            const int a0 = S.etile[er.tptr + T.tile], a1 = S.etile[er.tptr + T.tile + 1];
            child(S.rel + er.reloff, W + er.woff * ldx, a0, a1);
        }
    }
    if constexpr (NW > 1) {
        if (kw > 1) {       // (all NW waves of the workgroup are here: none has left)
            if (wave > 0) {
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) red[(wave * 8 + a * 4 + rr) * 64 + lane] = acc[a][rr];
            }
            __syncthreads();
            if (wave > 0) return;
#pragma unroll
            for (int w = 1; w < NW; w++)
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) acc[a][rr] += red[(w * 8 + a * 4 + rr) * 64 + lane];
        }
    }
    // W = -(acc): lane (lm, lk), register rr of row tile a = row i0 + 2 (lk + 4 rr) + a, right-hand side lm
    if (lm < nr) {
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int i = i0 + 2 * (lk + 4 * rr) + a;
                if (i < r) Ws[(long long)(i - c) * ldx + lm] = -acc[a][rr];
            }
    }
}

// Blocked forward substitution inside a front wider than `cap` columns: after y_blk = X_blk b_blk, the own rows
// below the block get  b[i] -= sum_{q in block} L[i][q] y[q].  A workgroup owns 32 rows x 64 right-hand sides,
// its four waves split the K range (the block's columns); the partial tiles are summed through LDS.
__global__ __launch_bounds__(256) void k_fwd_own_update(DevSym S, const int *__restrict__ list,
                                                        const double *__restrict__ L, const double *__restrict__ Y,
                                                        double *__restrict__ X, int nr, int ldx, int blk, int cap) {
    __shared__ double red[3 * 16 * 64];
    const int s = list[blockIdx.y];
    const int c = S.sfirst[s + 1] - S.sfirst[s];
    const int q0 = blk * cap, q1 = min(c, (blk + 1) * cap);     // K range: the block's columns
    const int i0 = q1 + blockIdx.x * 32;                        // own rows below the block
    if (i0 >= c) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lm = lane & 15, lk = lane >> 4;
    const int ld = S.ld[s];
    const int first = S.sfirst[s];
    const double *P = L + S.panelptr[s];
    const double *Yb = Y + (long long)first * ldx;
    double *Xb = X + (long long)first * ldx;
    const int nt = (nr + 15) >> 4;
    d4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int t = 0; t < 4; t++) acc[a][t] = (d4){0.0, 0.0, 0.0, 0.0};
    const double *pa[2] = {P + min(i0 + lm, c - 1), P + min(i0 + 16 + lm, c - 1)};
    const int jc[4] = {min(lm, nr - 1), min(16 + lm, nr - 1), min(32 + lm, nr - 1), min(48 + lm, nr - 1)};
    constexpr int KU = 4;
    for (int k0 = q0 + wave * 4 * KU; k0 < q1; k0 += 16 * KU) {
        double av[KU][2], bv[KU][4];
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const int kk = k0 + 4 * u + lk;
            const int kc = min(kk, q1 - 1);
            const double mk = kk < q1 ? 1.0 : 0.0;
#pragma unroll
            for (int a = 0; a < 2; a++) av[u][a] = pa[a][(long long)kc * ld] * mk;
#pragma unroll
            for (int t = 0; t < 4; t++) bv[u][t] = Yb[(long long)kc * ldx + jc[t]];
        }
#pragma unroll
        for (int u = 0; u < KU; u++)
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int t = 0; t < 4; t++)
                    acc[a][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][a], bv[u][t], acc[a][t], 0, 0, 0);
    }
    splitk_reduce4<2>(acc, red, wave, lane);
    // wave w owns the 16 right-hand sides 16 w .. of both row tiles: X -= acc (loads first, then stores)
#pragma unroll
    for (int t = 0; t < 4; t++) {
        if (t == wave && t < nt) {
            const int j = t * 16 + lm, jcl = min(j, nr - 1);
            double xv[2][4];
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int rr = 0; rr < 4; rr++) xv[a][rr] = Xb[(long long)min(i0 + a * 16 + lk + 4 * rr, c - 1) * ldx + jcl];
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const int i = i0 + a * 16 + lk + 4 * rr;
                    if (i < c && j < nr) Xb[(long long)i * ldx + j] = xv[a][rr] - acc[a][t][rr];
                }
        }
    }
}

// Passes of at most 16 right-hand sides: t = y - L21' x[trailing rows] of a front with at most `mmax` trailing rows, ONE WAVE per
// 16 own columns for the whole K range -- no LDS, no barrier, a quarter of the registers of the 64-column kernels: eight chains
// per SIMD (k_fwd_update_wave is the forward twin). Operands in pairs along K as in k_bwd_gemm_longk: a lane loads rows q, q + 1
// of its column and of the row list, q = batch + 8 h + 2 lk, and feeds k-steps 2 h and 2 h + 1 with them. The row indices of
// batch k + 1 are requested with the operands of batch k. Fronts with more trailing rows keep the split-K kernels (mmin).
// NW = 4 (levels with fronts of more than kWaveSplitRows trailing rows): four waves share such a front's K range, partial tiles
// summed in wave order through LDS; fronts with fewer rows are still done by wave 0 alone (the rule depends on the front only).
constexpr int kWaveSplitRows = 256;
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_bwd_wave(DevSym S, const int *__restrict__ list, const double *__restrict__ L, const double *X,
                                                 double *Xown, int nr, int ldx, int mmax) {
    __shared__ double red[NW > 1 ? NW * 4 * 64 : 1];
    { const int jt = 16 * blockIdx.z; X += jt; Xown += jt; nr = min(nr - jt, 16); }       // (round 6) blockIdx.z = 16-column tile of the right-hand sides
    const int s = list[blockIdx.y];
    const int c = S.sfirst[s + 1] - S.sfirst[s];
    const int r = (int)(S.rowptr[s + 1] - S.rowptr[s]);
    const int i0 = blockIdx.x * 16;
    if (i0 >= c || r <= c || r - c > mmax) return;
    const int wave = NW > 1 ? (int)(threadIdx.x >> 6) : 0;
    const int kw = (NW > 1 && r - c > kWaveSplitRows) ? NW : 1;
    if (wave >= kw) return;
    const int lane = threadIdx.x & 63;
    const int lm = lane & 15, lk = lane >> 4;
    const int ld = S.ld[s];
    const int first = S.sfirst[s];
    const double *pa = L + S.panelptr[s] + (long long)min(i0 + lm, c - 1) * ld;
    const int *rows = S.rows + S.rowptr[s];
    const int jl = min(lm, nr - 1);
    d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
    constexpr int NH = 4;                   // pairs per lane and batch: 8 k-steps = 32 rows
    long long xr[2 * NH], xn[2 * NH];
    auto request_rows = [&](int kb) {
#pragma unroll
        for (int h = 0; h < NH; h++) {
            const int q = kb + 8 * h + 2 * lk;
            const i2u v = *(const i2u *)(rows + min(q, r - 1));
            xn[2 * h] = v.x;
            xn[2 * h + 1] = q + 1 < r ? v.y : v.x;          // past the list: any valid row (its product is masked)
        }
    };
    request_rows(c + wave * 8 * NH);
    for (int k0 = c + wave * 8 * NH; k0 < r; k0 += kw * 8 * NH) {
        double av[2 * NH], bv[2 * NH];
#pragma unroll
        for (int u = 0; u < 2 * NH; u++) xr[u] = xn[u];
        request_rows(k0 + kw * 8 * NH);
#pragma unroll
        for (int h = 0; h < NH; h++) {
            const int q = k0 + 8 * h + 2 * lk;
            const d2u v = *(const d2u *)(pa + min(q, r - 1));
            av[2 * h] = v.x * (q < r ? 1.0 : 0.0); av[2 * h + 1] = v.y * (q + 1 < r ? 1.0 : 0.0);
        }
#pragma unroll
        for (int u = 0; u < 2 * NH; u++) bv[u] = X[xr[u] * ldx + jl];
#pragma unroll
        for (int u = 0; u < 2 * NH; u++)
            if (k0 + 8 * (u >> 1) < r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
    }
    if constexpr (NW > 1) {
        if (kw > 1) {
            if (wave > 0) {
#pragma unroll
                for (int rr = 0; rr < 4; rr++) red[(wave * 4 + rr) * 64 + lane] = acc[rr];
            }
            __syncthreads();
            if (wave > 0) return;
#pragma unroll
            for (int w = 1; w < NW; w++)
#pragma unroll
                for (int rr = 0; rr < 4; rr++) acc[rr] += red[(w * 4 + rr) * 64 + lane];
        }
    }
    // t[col][rhs]: register rr of lane (lm, lk) = own column i0 + lk + 4 rr, right-hand side lm (all loads, then all stores)
    double xv[4];
#pragma unroll
    for (int rr = 0; rr < 4; rr++) xv[rr] = Xown[(long long)(first + min(i0 + lk + 4 * rr, c - 1)) * ldx + jl];
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
        const int col = i0 + lk + 4 * rr;
        if (col < c && lm < nr) Xown[(long long)(first + col) * ldx + lm] = xv[rr] - acc[rr];
    }
}

// Backward update of a big front: own columns -= L21' * x_R over ALL trailing rows: a
// workgroup owns 64 own columns x 64 right-hand sides, its waves split the trailing rows.
template <int NA, int NW>   // NA: 16-column tiles of own columns per workgroup; NW: waves per workgroup splitting K
__global__ __launch_bounds__(64 * NW) void k_bwd_gemm_longk(DevSym S, const int *__restrict__ list,
                                                          const double *__restrict__ L, const double *X, double *Xown, int nr,
                                                          int ldx, int blk, int cap, int mmin) {
    // blk < 0: all own columns, K = the trailing rows [c, r). blk >= 0 (blocked substitution inside a front wider
    // than `cap` columns): own columns of block blk only, K = the OWN rows below the block, [(blk + 1) cap, c)
    // -- the same product with other bounds (rows[] lists the own columns first, so x of own rows is found the
    // same way as x of trailing rows).
    __shared__ double red[NW == 4 ? 3 * 16 * 64 : NW * 16 * 64];
    const int s = list[blockIdx.y];
    const int cfull = S.sfirst[s + 1] - S.sfirst[s];
    const int rfull = (int)(S.rowptr[s + 1] - S.rowptr[s]);
    const int col0 = blk < 0 ? 0 : blk * cap;
    const int c = blk < 0 ? cfull : min(cfull, (blk + 1) * cap);      // own columns [col0, c); K starts at row c
    const int r = blk < 0 ? rfull : cfull;                             // K ends at row r
    const int i0 = col0 + blockIdx.x * 16 * NA;
    if (i0 >= c || r <= c) return;
    if (blk < 0 && r - c <= mmin) return;       // (narrow passes: k_bwd_wave has the fronts with at most mmin trailing rows)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lm = lane & 15, lk = lane >> 4;
    const int ld = S.ld[s];
    const int first = S.sfirst[s];
    const double *P = L + S.panelptr[s];
    const int *rows = S.rows + S.rowptr[s];
    d4 acc[NA][4];
#pragma unroll
    for (int a = 0; a < NA; a++)
#pragma unroll
        for (int t = 0; t < 4; t++) acc[a][t] = (d4){0.0, 0.0, 0.0, 0.0};
    const double *pa[NA];
#pragma unroll
    for (int a = 0; a < NA; a++) pa[a] = P + (long long)min(i0 + a * 16 + lm, c - 1) * ld;
    // Everything in PAIRS (16-byte / 8-byte loads; the CU's address unit is what this kernel keeps busiest):
    //  * the panel along k: a lane loads rows q, q + 1 of its column, q = batch + 8 h + 2 lk, and feeds k-steps 2 h and
    //    2 h + 1 with them (k-step 2 h + e covers rows batch + 8 h + 2 lk + e, lk = 0..3);
    //  * the row indices of those rows the same way;
    //  * x along the right-hand sides: column tile t is right-hand side 32 (t >> 1) + 2 lm + (t & 1).
    // Per batch of 16 rows: 2 NA + 8 + 2 loads instead of 4 NA + 16 + 4.
    // (Round 6, measured and dropped: this loop as a three-stage software pipeline of half-batches -- operands requested two stages
    //  ahead, their row indices three, unconditional requests and scheduling barriers as in k_syrk_cb_rec<true>; 126 / 166 VGPRs for
    //  the <1, 8> / <2, 8> forms, same occupancy, bit-identical. Backward sweep of cfg 2: 1.434-1.441 ms without, 1.434-1.440 with it on
    //  either or both forms. The top-level launches are not a chain of exposed round trips: a level of 126 workgroups puts 2000 MFMAs
    //  on each of 126 compute units -- 13.8 us at the pipe's peak -- while the other half of the chip idles; only spreading a front's
    //  work over more compute units would shorten them.)
    const int jb[2] = {min(2 * lm, nr - 1), min(32 + 2 * lm, nr - 1)};
    constexpr int KU = 4;
    // The row indices of batch k+1 are requested together with the operands of batch k: one round
    // trip per batch instead of two (index -> X row). Long trailing parts (K = r - c up to 2000 at
    // the top of the tree) make this loop a pure latency chain.
    long long xr[KU], xn[KU];
    auto request_rows = [&](int kb) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int q = kb + 8 * h + 2 * lk;
            const i2u v = *(const i2u *)(rows + min(q, r - 1));
            xn[2 * h] = v.x;
            xn[2 * h + 1] = q + 1 < r ? v.y : v.x;          // past the list: any valid row (its product is masked)
        }
    };
    request_rows(c + wave * 4 * KU);
    for (int k0 = c + wave * 4 * KU; k0 < r; k0 += NW * 4 * KU) {
        double av[KU][NA], bv[KU][4];
#pragma unroll
        for (int u = 0; u < KU; u++) xr[u] = xn[u];
        request_rows(k0 + NW * 4 * KU);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int q = k0 + 8 * h + 2 * lk;
            const double m0 = q < r ? 1.0 : 0.0, m1 = q + 1 < r ? 1.0 : 0.0;
#pragma unroll
            for (int a = 0; a < NA; a++) {
                const d2u v = *(const d2u *)(pa[a] + min(q, r - 1));
                av[2 * h][a] = v.x * m0; av[2 * h + 1][a] = v.y * m1;
            }
        }
#pragma unroll
        for (int u = 0; u < KU; u++)
#pragma unroll
            for (int t2 = 0; t2 < 2; t2++) {
                const d2u y = *(const d2u *)(X + xr[u] * ldx + jb[t2]);
                bv[u][2 * t2] = y.x; bv[u][2 * t2 + 1] = y.y;
            }
#pragma unroll
        for (int u = 0; u < KU; u++)
#pragma unroll
            for (int a = 0; a < NA; a++)
#pragma unroll
                for (int t = 0; t < 4; t++)
                    acc[a][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][a], bv[u][t], acc[a][t], 0, 0, 0);
    }
    if constexpr (NA == 2 && NW == 4) {
        // wave w owns row tile w >> 1 (own columns i0 + 16 (w >> 1) + lk + 4 rr) x right-hand sides 32 (w & 1) + 2 lm, + 1:
        // x is read and written 16 bytes per lane (all loads, then all stores)
        splitk_reduce4_pairs(acc, red, wave, lane);
        const int a = wave >> 1, hc = wave & 1;
        const int j = 32 * hc + 2 * lm;
        d2u xv[4];
#pragma unroll
        for (int rr = 0; rr < 4; rr++)
            xv[rr] = *(const d2u *)(Xown + (long long)(first + min(i0 + a * 16 + lk + 4 * rr, c - 1)) * ldx + min(j, nr - 1));
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int col = i0 + a * 16 + lk + 4 * rr;
            if (col < c && j < nr) {
                double x0 = 0.0, x1 = 0.0;
#pragma unroll
                for (int aa = 0; aa < 2; aa++)
#pragma unroll
                    for (int h = 0; h < 2; h++)
                        if (aa == a && h == hc) { x0 = xv[rr].x - acc[aa][2 * h][rr]; x1 = xv[rr].y - acc[aa][2 * h + 1][rr]; }
                double *dst = Xown + (long long)(first + col) * ldx + j;
                if (j + 1 < nr) *(d2u *)dst = (d2u){x0, x1};
                else dst[0] = x0;
            }
        }
    } else {
        if (NW == 4) splitk_reduce4<NA>(acc, red, wave, lane);
        else {
#pragma unroll
            for (int a = 0; a < NA; a++) {
                if (a > 0) __syncthreads();         // (the buffer of the partial tiles is reused)
                splitk_reduce_nw<NW>(acc[a], red, wave, lane);
            }
        }
        // X -= acc in two passes (all loads, then all stores: one round trip instead of a chain of 8)
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (t == wave) {
                const int j = 32 * (t >> 1) + 2 * lm + (t & 1);
                const int jcl = min(j, nr - 1);
                double xv[NA][4];
#pragma unroll
                for (int a = 0; a < NA; a++)
#pragma unroll
                    for (int rr = 0; rr < 4; rr++)
                        xv[a][rr] = Xown[(long long)(first + min(i0 + a * 16 + lk + 4 * rr, c - 1)) * ldx + jcl];
#pragma unroll
                for (int a = 0; a < NA; a++)
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) {
                        const int col = i0 + a * 16 + lk + 4 * rr;
                        if (col < c && j < nr) Xown[(long long)(first + col) * ldx + j] = xv[a][rr] - acc[a][t][rr];
                    }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Right-hand-side permutation + transposition (column-major caller layout <-> row-major X)
// ------------------------------------------------------------------------------------------
// dir 0: X[k, j] = B[perm[k] + j*ldb]   (perm == nullptr: identity)
// dir 1: B[perm[k] + j*ldb] = X[k, j]
__global__ __launch_bounds__(256) void k_permute(const int *__restrict__ iperm, int n, double *__restrict__ Bc,
                                                 long long ldb, double *__restrict__ X, int nr, int ldx, int dir) {
    // Caller side: column-major n x nr (a DoF is strided by ldb); solver side: row-major, one DoF =
    // one contiguous 8 nr-byte row, in elimination order. A workgroup owns 64 consecutive ORIGINAL
    // rows i: the caller side is then read / written in 512-byte runs per column and the solver side
    // one whole row (iperm[i]) at a time -- both sides coalesced, the 64 x 64 transpose goes through
    // LDS. (Walking the elimination order instead and gathering caller rows perm[k] costs 2.7x the
    // bytes in partially used 64-byte sectors.) iperm == nullptr: identity.
    __shared__ double T[64 * 65];
    __shared__ int rowL[64];
    const int i0 = blockIdx.x * 64;
    const int tid = threadIdx.x;
    const int a = tid & 63, b = tid >> 6;
    if (tid < 64) rowL[tid] = (i0 + tid < n) ? (iperm ? iperm[i0 + tid] : i0 + tid) : 0;
    // all 16 loads of a thread are issued before the first use (clamped addresses, predicated stores): the kernel only
    // moves bytes, and a load -> LDS store -> load chain leaves 15 of every 16 round trips idle
    double v[16];
    if (dir == 0) {
        const int i = i0 + a, ic = min(i, n - 1);
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = Bc[ic + (long long)min(b + 4 * u, nr - 1) * ldb];
#pragma unroll
        for (int u = 0; u < 16; u++) T[a * 65 + b + 4 * u] = v[u];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int kk = b + 4 * u;
            if (i0 + kk < n && a < nr) X[(long long)rowL[kk] * ldx + a] = T[kk * 65 + a];
        }
    } else {
        __syncthreads();
        const int ac = min(a, nr - 1);
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = X[(long long)rowL[b + 4 * u] * ldx + ac];
#pragma unroll
        for (int u = 0; u < 16; u++) T[(b + 4 * u) * 65 + a] = v[u];
        __syncthreads();
        const int i = i0 + a;
        if (i < n) {
#pragma unroll
            for (int u = 0; u < 16; u++)
                if (b + 4 * u < nr) Bc[i + (long long)(b + 4 * u) * ldb] = T[a * 65 + b + 4 * u];
        }
    }
}
// The same for passes of at most 8 right-hand sides (the single solve): a thread per ORIGINAL row -- the caller side coalesced, the
// solver side 8 nr-byte pieces. The 64 x 64 transpose above spends 60 + 41 us on a 1-column pass of 10^6 rows (15 625 workgroups
// of which one column in 64 carries data); this one 8 + 8.
__global__ __launch_bounds__(256) void k_permute_narrow(const int *__restrict__ iperm, int n, double *__restrict__ Bc,
                                                        long long ldb, double *__restrict__ X, int nr, int ldx, int dir) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long row = iperm ? iperm[i] : i;
    double v[8];
    if (dir == 0) {
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = Bc[i + (long long)min(j, nr - 1) * ldb];
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (j < nr) X[row * ldx + j] = v[j];
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = X[row * ldx + min(j, nr - 1)];
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (j < nr) Bc[i + (long long)j * ldb] = v[j];
    }
}

// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// log det Q = 2 sum_k log L_kk, fixed-order two-stage reduction (bit-reproducible)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_logdet_partial(const double *__restrict__ L, const long long *__restrict__ diagoff,
                                                        const unsigned char *__restrict__ own, int n, double *__restrict__ part) {
    __shared__ double sh[256];
    const int tid = threadIdx.x;
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int k0 = blockIdx.x * per, k1 = min(n, k0 + per);
    double acc = 0.0;
    // own (sharded handles): only the columns of the fronts this rank factored; nullptr = all
    for (int k = k0 + tid; k < k1; k += 256)
        if (!own || own[k]) acc += log(L[diagoff[k]]);
    sh[tid] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) sh[tid] += sh[tid + st];
        __syncthreads();
    }
    if (tid == 0) part[blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(256) void k_logdet_final(const double *__restrict__ part, int nparts, double *__restrict__ out) {
    // fixed tree over the (at most 1024) block sums: reproducible, and 4 us instead of 50 for one serial thread
    __shared__ double sh[256];
    const int tid = threadIdx.x;
    double acc = 0.0;
    for (int i = tid; i < nparts; i += 256) acc += part[i];
    sh[tid] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) sh[tid] += sh[tid + st];
        __syncthreads();
    }
    if (tid == 0) out[0] = 2.0 * sh[0];
}

// Gather values at precomputed offsets (-1 -> 0.0): selected-inverse extraction.
__global__ __launch_bounds__(256) void k_gather(const double *__restrict__ src, const long long *__restrict__ off,
                                                long long cnt, double *__restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < cnt) { const long long o = off[i]; out[i] = (o >= 0) ? src[o] : 0.0; }
}
// out[g] = sum over t in [segptr[g], segptr[g+1]) of w[t] * src[off[t]] (off < 0 -> 0): one wave per segment, lanes
// stride the segment, butterfly sum in a fixed order (reproducible). Consumers of the selected inverse that only
// need contractions (diag(A Sigma A'), tr(Sigma B)) never move Sigma's values to the host.
__global__ __launch_bounds__(64) void k_seg_wsum(const double *__restrict__ src, const long long *__restrict__ segptr,
                                                 const long long *__restrict__ off, const double *__restrict__ w,
                                                 double *__restrict__ out) {
    const long long t0 = segptr[blockIdx.x], t1 = segptr[blockIdx.x + 1];
    double acc = 0.0;
    for (long long t = t0 + threadIdx.x; t < t1; t += 64) {
        const long long o = off[t];
        acc += w[t] * src[o >= 0 ? o : 0] * (o >= 0 ? 1.0 : 0.0);
    }
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) acc += __shfl_xor(acc, sh, 64);
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}
// The same with the weights formed on the fly from the values of a sparse design matrix: entry t is the pair
// (p, q) of entries of one row, weight = A_p A_q (twice for p != q: Sigma is symmetric and only q <= p is listed).
__global__ __launch_bounds__(64) void k_seg_wsum_pairs(const double *__restrict__ src, const long long *__restrict__ segptr,
                                                       const long long *__restrict__ off, const int *__restrict__ pi,
                                                       const int *__restrict__ qi, const double *__restrict__ vals,
                                                       double *__restrict__ out) {
    const long long t0 = segptr[blockIdx.x], t1 = segptr[blockIdx.x + 1];
    double acc = 0.0;
    for (long long t = t0 + threadIdx.x; t < t1; t += 64) {
        const long long o = off[t];
        const int p = pi[t], q = qi[t];
        acc += (p == q ? 1.0 : 2.0) * vals[p] * vals[q] * src[o >= 0 ? o : 0] * (o >= 0 ? 1.0 : 0.0);
    }
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) acc += __shfl_xor(acc, sh, 64);
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
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

// Write-once assembly for fronts whose columns fit in LDS (all but the top levels): one wave builds
// its column in LDS -- zero, Q's values, the children's contributions in fixed order -- and stores it
// to HBM ONCE. The HBM version above zero-fills the panel and then read-modify-writes it per child
// (measured: k_assemble moved 6.5 GB per step and ran at ~4.6 TB/s, i.e. HBM bound on bytes it need
// not move). Same summation order, bit-identical panels. Dynamic LDS: 4 * ldmax doubles.
// nzp[q] = nzval[qsrc[q]]: Q's values in the order the assembly reads them (once per factorisation, 56 MB at cfg 2)
__global__ __launch_bounds__(256) void k_gather_values(const double *__restrict__ nzval, const int *__restrict__ qsrc, double *__restrict__ out, long long cnt) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < cnt; i += (long long)gridDim.x * 256) out[i] = nzval[qsrc[i]];
}
void launch_gather_values(hipStream_t st, const double *nzval, const int *qsrc, double *out, long long cnt) {
    if (cnt <= 0) return;
    hipLaunchKernelGGL(k_gather_values, dim3((unsigned)std::min<long long>(8192, (cnt + 255) / 256)), dim3(256), 0, st, nzval, qsrc, out, cnt);
}

template <int WIDE>   // 0: one WAVE per column (four columns per workgroup); 1: one WORKGROUP per column (tall columns: top of the tree)
__global__ __launch_bounds__(256) void k_assemble_lds(DevSym S, const AsmRec *__restrict__ arec,
                                                      const double *__restrict__ nzp, double *__restrict__ L,
                                                      const double *__restrict__ CB, int ldmax) {
    extern __shared__ double col_lds[];
    constexpr int NL = WIDE ? 256 : 64, PW = 2 * NL;       // lanes on one column; rows one pair-load of all of them covers
    const AsmRec R = arec[blockIdx.y];                     // the front and its first two children: one scalar load
    const int c = R.c;
    const int wave = threadIdx.x >> 6;
    const int lane = WIDE ? (int)threadIdx.x : (int)(threadIdx.x & 63);         // position among the column's lanes
    const int tc = WIDE ? (int)blockIdx.x : blockIdx.x * ASM_CW + __builtin_amdgcn_readfirstlane(wave);
    if (tc >= c) return;
    const int ld = R.ld;
    double *Cw = WIDE ? col_lds : col_lds + wave * ldmax;
    double *Pc = L + R.pp + (long long)tc * ld;
    // A column is a chain of dependent round trips (front -> Q's range / child records -> the child's row -> entries): everything
    // the FIRST TWO children and Q's first 64 entries need is requested before any of it is used -- records and rows of both
    // children side by side, then all entry loads -- and only then does the column build up in LDS, in the old order (zero, Q,
    // child by child): same bits, three round trips (record | Q's range, the children's rows | entries and Q's values) instead of nine.
    const int nch = R.nch;
    const long long ch0 = R.ch0, ch1 = ch0 + nch;
    const int gk = R.first + tc;
    const int qlo = S.qcolptr[gk], qhi = S.qcolptr[gk + 1];
    struct { int md; long long reloff, cboff; } er[2] = {{R.md[0], R.reloff[0], R.cboff[0]}, {R.md[1], R.reloff[1], R.cboff[1]}};
    int jj[2] = {-1, -1};
#pragma unroll
    for (int q = 0; q < 2; q++)
        if (q < nch) jj[q] = S.erow[R.eoff[q] + tc];        // the child's row that maps to column tc (table, no search); < 0: none
    int qd0 = 0;
    double qv0 = 0.0;
    if (qlo + lane < qhi) { qd0 = S.qdst[qlo + lane]; qv0 = nzp[qlo + lane]; }        // (values in assembly order: no index in between)
    // (WIDE: the waves of the workgroup touch the same rows: a barrier between the phases; rows are distinct within a phase)
    // Rows in PAIRS per lane (one 16-byte value load + one 8-byte index load cover 128 rows of the column), four
    // chunks in flight, and chunks past the end of the child's column issue nothing: the kernel is bound by the
    // CU's address unit (a vector memory instruction costs it ~16 cycles whatever its lanes do), not by HBM.
    // The pair that starts at the last row reads one element past the column: the next column, or the 16 bytes of
    // slack every device array ends in (Device::dalloc); never used.
    i2u ri[2][4];
    d2u u[2][4];
#pragma unroll
    for (int q = 0; q < 2; q++)
        if (jj[q] >= 0) {
            const int md = er[q].md;
            const int *reld = S.rel + er[q].reloff;
            const double *Uc = CB + er[q].cboff + (long long)jj[q] * md;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (jj[q] + PW * k < md) {
                    const int ic = min(jj[q] + PW * k + 2 * lane, md - 1);
                    ri[q][k] = *(const i2u *)(reld + ic);
                    u[q][k] = *(const d2u *)(Uc + ic);
                }
        }
    for (int i = lane; i < ld; i += NL) Cw[i] = 0.0;
    if (WIDE) __syncthreads();
    // Q's entries of this column: [qcolptr[k], qcolptr[k + 1]) for column k of L (no search)
    if (qlo + lane < qhi) Cw[qd0] = qv0;
    for (int q = qlo + NL + lane; q < qhi; q += NL) Cw[S.qdst[q]] = nzp[q];
    if (WIDE) __syncthreads();
    auto chunk = [&](const int *reld, const double *Uc, int md, int base) {
        i2u r2[4];
        d2u u2[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (base + PW * k < md) {
                const int ic = min(base + PW * k + 2 * lane, md - 1);
                r2[k] = *(const i2u *)(reld + ic);
                u2[k] = *(const d2u *)(Uc + ic);
            }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (base + PW * k < md) {
                const int i = base + PW * k + 2 * lane;
                if (i < md) Cw[r2[k].x] += u2[k].x;          // distinct rows within a child: no conflicts
                if (i + 1 < md) Cw[r2[k].y] += u2[k].y;
            }
    };
#pragma unroll
    for (int q = 0; q < 2; q++) {
        if (jj[q] >= 0) {
            const int md = er[q].md;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (jj[q] + PW * k < md) {
                    const int i = jj[q] + PW * k + 2 * lane;
                    if (i < md) Cw[ri[q][k].x] += u[q][k].x;
                    if (i + 1 < md) Cw[ri[q][k].y] += u[q][k].y;
                }
            const int *reld = S.rel + er[q].reloff;
            const double *Uc = CB + er[q].cboff + (long long)jj[q] * md;
            for (int base = jj[q] + 4 * PW; base < md; base += 4 * PW) chunk(reld, Uc, md, base);
        }
        if (WIDE && q < nch) __syncthreads();
    }
    for (long long ch = ch0 + 2; ch < ch1; ch++) {          // further children: one at a time
        const EdgeRec e3 = S.edge[ch];
        const int md = e3.md;
        const int *reld = S.rel + e3.reloff;
        const int j = S.erow[e3.eoff + tc];
        if (j >= 0) {
            const double *Uc = CB + e3.cboff + (long long)j * md;
            for (int base = j; base < md; base += 4 * PW) chunk(reld, Uc, md, base);
        }
        if (WIDE) __syncthreads();
    }
    for (int i = 2 * lane; i < ld; i += PW) *(d2u *)(Pc + i) = (d2u){Cw[i], Cw[i + 1]};      // ld is even
}

// Workgroups are handed to the 8 XCDs round-robin by linear id (x fastest). Rectangular grids whose
// x extent (or x*y extent) is a multiple of 8 put tile (bi, bj) of EVERY front on the same XCD --
// with triangular / ragged tile sets that leaves some XCDs idle and others with twice the work
// (measured: the 4x4-tile level of k_syrk_cb ran 1.9x longer than the 3x3 and 6x6 levels around
// it). Odd extents make consecutive fronts rotate through all XCDs; the extra workgroups exit at
// once through the kernels' own range checks.
static inline unsigned odd(int v) { return (unsigned)(v | 1); }

void launch_assemble(hipStream_t st, const DevSym &S, const int *list, const AsmRec *arec, const double *nzp, int nfronts, int max_cols, int max_rows,
                     const double *nzval, double *L, double *CB) {
    if (nfronts <= 0) return;
    const int ldmax = (max_rows + 1) & ~1;       // Symbolic rounds ld up to even
    // Columns through LDS, written once: a wave per column while four columns of a workgroup fit in 40 KB, a workgroup per column
    // for the tall columns of the top of the tree (up to 128 KB). Measured at cfg 2 (factorisation): HBM assembly above 1280 rows
    // 9.33 ms; wave-per-column up to 2048 rows 9.10; + workgroup-per-column above: 9.03; wave-per-column up to 1280, workgroup-per-column
    // above: 8.96.
    constexpr int lds_cols_max = 1280, lds_wide_max = 16384;
    if (ldmax <= lds_cols_max && ldmax <= 2048) {
        hipLaunchKernelGGL(k_assemble_lds<0>, dim3(odd(cdiv(max_cols, ASM_CW)), nfronts), dim3(256), (size_t)4 * ldmax * sizeof(double), st,
                           S, arec, nzp, L, CB, ldmax);
        return;
    }
    if (ldmax <= lds_wide_max && ldmax <= 16384) {
        const size_t lds = (size_t)ldmax * sizeof(double);
        if (lds > 65536) {
            static const bool once = [] { return hipFuncSetAttribute((const void *)k_assemble_lds<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) == hipSuccess; }();
            if (!once) goto hbm;
        }
        hipLaunchKernelGGL(k_assemble_lds<1>, dim3(odd(max_cols), nfronts), dim3(256), lds, st, S, arec, nzp, L, CB, ldmax);
        return;
    }
hbm:
    if ((long long)cdiv(max_cols, ASM_CW) * nfronts <= 2200)
        hipLaunchKernelGGL(k_assemble<1>, dim3(odd(max_cols), nfronts), dim3(256), 0, st, S, list, nzval, L, CB, 0, 0, 0);
    else
        hipLaunchKernelGGL(k_assemble<0>, dim3(odd(cdiv(max_cols, ASM_CW)), nfronts), dim3(256), 0, st, S, list, nzval, L, CB, 0, 0, 0);
}
// the distributed root: one front, only the 256-column blocks b with b mod cyc_w == cyc_r (one workgroup per column)
void launch_assemble_cyclic(hipStream_t st, const DevSym &S, const int *list, int ncols, const double *nzval, double *L, double *CB,
                            int cyc_w, int cyc_r, bool compact) {
    hipLaunchKernelGGL(k_assemble<1>, dim3(odd(ncols), 1), dim3(256), 0, st, S, list, nzval, L, CB, cyc_w, cyc_r, compact ? 1 : 0);
}
void launch_syrk_cb(hipStream_t st, const DevSym &S, const int *list, int nfronts, int max_trail, const double *L, double *CB) {
    if (nfronts <= 0 || max_trail <= 0) return;
    hipLaunchKernelGGL(k_syrk_cb, dim3(odd(cdiv(max_trail, 64)), odd(cdiv(max_trail, 64)), nfronts), dim3(256), 0, st, S, list, L, CB, 0, 0, 0);
}
void launch_syrk_cb_cyclic(hipStream_t st, const DevSym &S, const int *list, int trail, const double *L, double *CB, int cyc_w, int cyc_r, int cyc_b0) {
    if (trail <= 0) return;
    hipLaunchKernelGGL(k_syrk_cb, dim3(odd(cdiv(trail, 64)), odd(cdiv(trail, 64)), 1), dim3(256), 0, st, S, list, L, CB, cyc_w, cyc_r, cyc_b0);
}
void launch_syrk_cb_recs(hipStream_t st, const DevSym &S, const SyrkTile *recs, const SyrkSplit &split, int per_xcd, const double *L, double *CB,
                         int noprod, bool piped) {
    if (per_xcd <= 0) return;
    if (piped) hipLaunchKernelGGL(k_syrk_cb_rec<true>, dim3(8 * (unsigned)per_xcd), dim3(256), 0, st, S, recs, split, L, CB, noprod);
    else hipLaunchKernelGGL(k_syrk_cb_rec<false>, dim3(8 * (unsigned)per_xcd), dim3(256), 0, st, S, recs, split, L, CB, noprod);
}
__global__ void k_syrk_big(DevSym S, const int *__restrict__ list, const double *__restrict__ L, double *__restrict__ CB);   // with k_gemm_nt_big, below
void launch_syrk_big(hipStream_t st, const DevSym &S, const int *list, int nfronts, int max_trail, const double *L, double *CB) {
    if (nfronts <= 0 || max_trail <= 0) return;
    const int nt = cdiv(max_trail, 128);
    hipLaunchKernelGGL(k_syrk_big, dim3(odd(nt), odd(nt), nfronts), dim3(512), 0, st, S, list, L, CB);
}
void launch_trsm(hipStream_t st, const DevSym &S, const FrontView *frec, int nactive, int kb, int mode, int max_rows_below,
                 double *L, double *Yh, const long long *yoff, const FrontArg &fa) {
    if (nactive <= 0 || max_rows_below <= 0) return;
    const bool split = (long long)cdiv(max_rows_below, 64) * nactive <= 128;
    const dim3 grid(odd(cdiv(max_rows_below, split ? 16 : 128)), nactive);
    if (mode == 0) {
        if (split) hipLaunchKernelGGL((k_trsm<0, 1>), grid, dim3(256), 0, st, S, frec, kb, L, Yh, yoff, fa);
        else hipLaunchKernelGGL((k_trsm<0, 0>), grid, dim3(256), 0, st, S, frec, kb, L, Yh, yoff, fa);
    } else {
        if (split) hipLaunchKernelGGL((k_trsm<1, 1>), grid, dim3(256), 0, st, S, frec, kb, L, Yh, yoff, fa);
        else hipLaunchKernelGGL((k_trsm<1, 0>), grid, dim3(256), 0, st, S, frec, kb, L, Yh, yoff, fa);
    }
}
void launch_trsm_narrow(hipStream_t st, const FrontView *frec, int nactive, int kb, int max_rows_below, double *L) {
    if (nactive <= 0 || max_rows_below <= 0) return;
    hipLaunchKernelGGL(k_trsm_narrow, dim3(odd(cdiv(max_rows_below, 128)), nactive), dim3(256), 0, st, frec, kb, L);
}
// The same update for the HUGE fronts of 3-D problems (round 5): 128 x 128 workgroup tiles, operands staged through LDS -- 16 k
// at a time, double-buffered through registers -- and shared by eight waves of 32 x 64 (tools/micro/dgemm_mfma.hip: 54 TFLOP/s on
// an ideal shape against 47-49 for the direct-operand 64 x 64 tile above, which is what the three top levels of the 126^3 mesh
// ran at). The transplant lost twice at cfg 2 (a quarter of the tiles, one wave per SIMD on half the chip);
// it is only used where a launch has thousands of such tiles: K a multiple of 16, at least 4096 rows below the block.
// Same arithmetic per entry (the k order inside an entry's sum is the same); the strict upper triangle of the diagonal tiles
// is computed and not stored.
// C (M x N, leading dimension ldc; lower part, i >= j) -= A[0 .. M) A[0 .. N)' over K columns of A (leading dimension lda); K any
// (a k beyond K is staged as zero)
__device__ __forceinline__ void gemm_nt_big_tile(const double *__restrict__ A, int lda, double *__restrict__ C, long long ldc, int M, int N, int K) {
    constexpr int TM = 128, KB = 16;
    __shared__ double As[2][KB][TM + 8], Bs[2][KB][TM + 8];      // +8: consecutive k rows start in different banks
    const int bi = blockIdx.x, bj = blockIdx.y;
    if (bj > bi || bi * TM >= M || bj * TM >= N) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lm = lane & 15, lk = lane >> 4;
    const int m0 = bi * TM, n0 = bj * TM;
    const int wi = (wave & 3) * 32, wj = (wave >> 2) * 64;       // wave sub-tile: 32 rows x 64 columns
    d4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
    // staging: 512 threads, a 128 x 16 slab = 4 doubles per thread (row tid % 128 -- clamped: rows past the edge are never
    // stored --, k = 4 (tid / 128) ..)
    const int lr = tid & 127, l4 = (tid >> 7) * 4;
    const double *pa = A + min(m0 + lr, M - 1);
    const double *pb = A + min(n0 + lr, M - 1);
    double ra[4], rb[4];
    auto fetch = [&](int kb) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int k = kb * KB + l4 + q;
            const long long ko = (long long)min(k, K - 1) * lda;
            const double mk = k < K ? 1.0 : 0.0;
            ra[q] = pa[ko] * mk; rb[q] = pb[ko];
        }
    };
    fetch(0);
#pragma unroll
    for (int q = 0; q < 4; q++) { As[0][l4 + q][lr] = ra[q]; Bs[0][l4 + q][lr] = rb[q]; }
    __syncthreads();
    const int nk = (K + KB - 1) / KB;
    for (int kb = 0; kb < nk; kb++) {
        const int cur = kb & 1;
        if (kb + 1 < nk) fetch(kb + 1);
#pragma unroll
        for (int sidx = 0; sidx < KB / 4; sidx++) {
            double av[2], bv[4];
#pragma unroll
            for (int a = 0; a < 2; a++) av[a] = As[cur][4 * sidx + lk][wi + 16 * a + lm];
#pragma unroll
            for (int b = 0; b < 4; b++) bv[b] = Bs[cur][4 * sidx + lk][wj + 16 * b + lm];
            // D[m = column j][n = row i]: first operand = rows of B (columns of C), second = rows of A (the lanes walk i)
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
    // C -= acc, lower part only (i >= j), all loads before all stores
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int i = m0 + wi + 16 * a + lm;
            double cv[4];
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int j = n0 + wj + 16 * b + lk + 4 * rr;
                cv[rr] = C[min(i, M - 1) + (long long)min(j, N - 1) * ldc];
            }
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int j = n0 + wj + 16 * b + lk + 4 * rr;
                if (i < M && j < N && i >= j) C[i + (long long)j * ldc] = cv[rr] - acc[a][b][rr];
            }
        }
}
__global__ __launch_bounds__(512) void k_gemm_nt_big(const FrontView *__restrict__ frec, int k0, int K, int c0, int c1,
                                                     double *__restrict__ L, FrontArg fa) {
    const FrontView fv = front_view(frec, blockIdx.z, fa);
    const int c = fv.c;
    if (c0 >= c) return;
    double *P = L + fv.pp;
    const double *PA = fa.on && fa.ppa != kNoPpa ? L + fa.ppa : P;
    gemm_nt_big_tile(PA + c0 + (long long)k0 * fv.ld, fv.ld, P + c0 + (long long)c0 * fv.ld, fv.ld, fv.r - c0, min(c1, c) - c0, K);
}
// The contribution block's product on the same tiles: CB -= L21 L21' behind a gather-only pass of k_syrk_cb_rec (noprod). At
// cfg 4 the one-pass kernel's 64 x 64 tiles stream K = 8 000-16 000 columns of both operands per tile (8 flop per byte: it ran at
// ~34 TFLOP/s, 3.5 of the 5.1 s); a 128 x 128 tile halves the operand bytes per flop. The extra read-modify-write of the block
// (m^2 doubles twice) is milliseconds against seconds there.
__global__ __launch_bounds__(512) void k_syrk_big(DevSym S, const int *__restrict__ list, const double *__restrict__ L, double *__restrict__ CB) {
    const int s = list[blockIdx.z];
    const int c = S.sfirst[s + 1] - S.sfirst[s];
    const int r = (int)(S.rowptr[s + 1] - S.rowptr[s]);
    const int m = r - c;
    if (m <= 0) return;
    gemm_nt_big_tile(L + S.panelptr[s] + c, S.ld[s], CB + S.cbptr[s], m, m, m, c);
}

void launch_gemm_nt(hipStream_t st, const DevSym &S, const FrontView *frec, int nactive, int k0, int K, int c0, int c1,
                    int maxM, int maxN, double *L, const FrontArg &fa) {
    if (nactive <= 0 || maxM <= 0 || maxN <= 0) return;
    if (K % 16 == 0 && K >= 256 && maxM >= 4096 && maxN >= 512) {
        hipLaunchKernelGGL(k_gemm_nt_big, dim3(odd(cdiv(maxM, 128)), odd(cdiv(maxN, 128)), nactive), dim3(512), 0, st, frec, k0, K, c0, c1, L, fa);
        return;
    }
    // 64x64 workgroup tiles, operands straight from L2 at 3-4 waves per SIMD. Measured on MI355X: the
    // sustained v_mfma_f64_16x16x4_f64 rate is 36.3 TFLOP/s (tools/micro/mfma64.hip), this kernel reaches
    // ~27 TFLOP/s on the top-of-tree SYRKs; 128x128 tiles (register- or LDS-staged) were tried and lost
    // to it because they drop to one wave per SIMD.
    // Levels with a handful of fronts are latency bound: 32x32 workgroup tiles there (four times
    // the workgroups, a quarter of the MFMA chain per wave).
    if ((long long)cdiv(maxM, 64) * cdiv(maxN, 64) * nactive <= 256)
        hipLaunchKernelGGL(k_gemm_nt<1>, dim3(odd(cdiv(maxM, 32)), odd(cdiv(maxN, 32)), nactive), dim3(256), 0, st, S, frec, k0, K, c0, c1, L, fa);
    else
        hipLaunchKernelGGL(k_gemm_nt<2>, dim3(odd(cdiv(maxM, 64)), odd(cdiv(maxN, 64)), nactive), dim3(256), 0, st, S, frec, k0, K, c0, c1, L, fa);
}
void launch_fwd_assemble(hipStream_t st, const DevSym &S, const int *list, int nfronts, int max_cols, double *X,
                         const double *W, int nr, int ldx) {
    if (nfronts <= 0 || max_cols <= 0) return;
    hipLaunchKernelGGL(k_fwd_assemble, dim3(odd(cdiv(max_cols, FWD_RB)), nfronts), dim3(256), 0, st, S, list, X, W, nr, ldx);
}
void launch_fwd_update(hipStream_t st, const DevSym &S, const int *list, int nfronts, int max_trail, const double *L,
                       double *X, double *W, int nr, int ldx, int cmin) {
    if (nfronts <= 0 || max_trail <= 0) return;
    if ((long long)cdiv(max_trail, 32) * nfronts <= 128)
        hipLaunchKernelGGL(k_fwd_update_longk<1>, dim3(odd(cdiv(max_trail, 16)), nfronts), dim3(256), 0, st, S, list, L, X, W, nr, ldx, cmin);
    else
        hipLaunchKernelGGL(k_fwd_update_longk<2>, dim3(odd(cdiv(max_trail, 32)), nfronts), dim3(256), 0, st, S, list, L, X, W, nr, ldx, cmin);
}
void launch_fwd_update_recs(hipStream_t st, const DevSym &S, const FwdTile *recs, const SyrkSplit &split, int per_xcd, const double *L,
                            double *X, double *W, int nr, int ldx, int cmin) {
    if (per_xcd <= 0) return;
    hipLaunchKernelGGL(k_fwd_update_rec, dim3(8 * (unsigned)per_xcd), dim3(256), 0, st, S, recs, split, L, X, W, nr, ldx, cmin);
}
// Passes of up to 32 right-hand sides take the narrow FORWARD kernels (k_fwd_update_wave, k_xmul_narrow) on two right-hand-side tiles
// (round 6; measured at cfg 2, tools/nrhs_sweep.py: 17 / 24 / 32 columns 3.22 / 3.29 / 3.35 -> 2.98 / 3.05 / 3.09 ms; the same on three
// or four tiles loses: 48 columns 3.54 -> 3.85 ms). The BACKWARD kernels of such passes stay the 64-column ones -- k_bwd_front /
// k_bwd_gemm_longk beat k_bwd_wave on two tiles (backward 1.39 vs 1.42-1.47 ms) --, and so do the bottom tasks (chunk form on two column
// slices: 2.98 ms at 17 columns against 3.37 with one wave per task and tile).
int narrow_pass_max() { return 32; }
int narrow_pass_max_bwd() { return 16; }
int launch_wave_split_cols() { return kWaveSplitCols; }
int launch_wave_split_rows() { return kWaveSplitRows; }
void launch_fwd_update_wave(hipStream_t st, const DevSym &S, const FwdTile *recs, const SyrkSplit &split, int per_xcd, const double *L,
                            double *X, double *W, int nr, int ldx, int cmax, bool split_k) {
    if (per_xcd <= 0) return;
    const unsigned jt = (unsigned)cdiv(nr, 16);
    if (split_k) hipLaunchKernelGGL(k_fwd_update_wave<4>, dim3(8 * (unsigned)per_xcd, jt), dim3(256), 0, st, S, recs, split, L, X, W, nr, ldx, cmax);
    else hipLaunchKernelGGL(k_fwd_update_wave<1>, dim3(8 * (unsigned)per_xcd, jt), dim3(64), 0, st, S, recs, split, L, X, W, nr, ldx, cmax);
}
void launch_bwd_gemm(hipStream_t st, const DevSym &S, const int *list, int nfronts, int max_cols, const double *L,
                     const double *X, double *Xown, int nr, int ldx, int blk, int cap, int mmin) {
    if (nfronts <= 0 || max_cols <= 0) return;
    if (blk >= 0) max_cols = std::min(max_cols - blk * cap, cap);
    if (max_cols <= 0) return;
    // the 8-wave variants up to ~3 workgroups per CU (measured: 512-1024 beats 128 and 2048). Every workgroup of a front gathers ALL
    // of the front's trailing rows of x: 32 own columns per workgroup instead of 16 halves those re-reads (levels 10-13 of cfg 2
    // moved 2.5-3 x their algorithmic bytes) on levels that still fill the chip with them (measured at cfg 2, 32 / 16 columns per
    // workgroup: level 10 (768 workgroups of 32) 72 / 83 us, 11 (576) 79 / 88; 12 (352) 79 / 73, 13: 85 / 75, 14: 66 / 55, 15: 62 / 47)
    const long long wg32 = (long long)cdiv(max_cols, 32) * nfronts;
    if (wg32 <= 768 && wg32 >= 384)
        hipLaunchKernelGGL((k_bwd_gemm_longk<2, 8>), dim3(odd(cdiv(max_cols, 32)), nfronts), dim3(512), 0, st, S, list, L, X, Xown, nr, ldx, blk, cap, mmin);
    else if (wg32 <= 768)
        hipLaunchKernelGGL((k_bwd_gemm_longk<1, 8>), dim3(odd(cdiv(max_cols, 16)), nfronts), dim3(512), 0, st, S, list, L, X, Xown, nr, ldx, blk, cap, mmin);
    else
        hipLaunchKernelGGL((k_bwd_gemm_longk<2, 4>), dim3(odd(cdiv(max_cols, 32)), nfronts), dim3(256), 0, st, S, list, L, X, Xown, nr, ldx, blk, cap, mmin);
}
void launch_bwd_wave(hipStream_t st, const DevSym &S, const int *list, int nfronts, int max_cols, const double *L, const double *X, double *Xown,
                     int nr, int ldx, int mmax, bool split_k) {
    if (nfronts <= 0 || max_cols <= 0) return;
    const unsigned jt = (unsigned)cdiv(nr, 16);
    if (split_k) hipLaunchKernelGGL(k_bwd_wave<4>, dim3(odd(cdiv(max_cols, 16)), nfronts, jt), dim3(256), 0, st, S, list, L, X, Xown, nr, ldx, mmax);
    else hipLaunchKernelGGL(k_bwd_wave<1>, dim3(odd(cdiv(max_cols, 16)), nfronts, jt), dim3(64), 0, st, S, list, L, X, Xown, nr, ldx, mmax);
}
void launch_fwd_own_update(hipStream_t st, const DevSym &S, const int *list, int nfronts, int max_cols, const double *L,
                           const double *Y, double *X, int nr, int ldx, int blk, int cap) {
    const int rows_below = max_cols - (blk + 1) * cap;      // own rows below the block in the widest front
    if (nfronts <= 0 || rows_below <= 0) return;
    hipLaunchKernelGGL(k_fwd_own_update, dim3(odd(cdiv(rows_below, 32)), nfronts), dim3(256), 0, st, S, list, L, Y, X, nr, ldx, blk, cap);
}
// Profiling aid (GMRFX_LEVEL_MARK=1, tools/sweep_levels.py): an empty kernel whose launch geometry names the phase and tree
// level that follows it in the stream, so that a kernel trace / counter pass can be cut into levels without guessing.
__global__ void k_level_mark() {}
void launch_level_mark(hipStream_t st, int phase, int level) {
    hipLaunchKernelGGL(k_level_mark, dim3(level + 2), dim3(64 * phase), 0, st);      // level -1 = the sweep tasks / subtrees
}
void launch_permute(hipStream_t st, const int *perm, int n, double *Bc, long long ldb, double *X, int nr, int ldx, int dir) {
    if (nr <= 8) hipLaunchKernelGGL(k_permute_narrow, dim3(cdiv(n, 256)), dim3(256), 0, st, perm, n, Bc, ldb, X, nr, ldx, dir);
    else hipLaunchKernelGGL(k_permute, dim3(cdiv(n, 64)), dim3(256), 0, st, perm, n, Bc, ldb, X, nr, ldx, dir);
}
// nz[map[k]] = prior[map[k]] - h[k] on top of nz = prior: the Newton-loop update of the reference
// (_update_hessian!, src/workspace/gaussian_approximation.jl:103-129) with Q kept on the device.
__global__ __launch_bounds__(256) void k_copy_values(const double *__restrict__ src, double *__restrict__ dst, long long cnt) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < cnt; i += (long long)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_subtract_at(double *__restrict__ nz, const long long *__restrict__ map,
                                                     const double *__restrict__ h, long long cnt) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k < cnt) nz[map[k]] -= h[k];     // the map is injective (one Q entry per Hessian entry)
}
void launch_newton_update(hipStream_t st, const double *prior, double *nz, long long nnz, const long long *map, const double *h,
                          long long cnt) {
    hipLaunchKernelGGL(k_copy_values, dim3((unsigned)std::min<long long>(4096, (nnz + 255) / 256)), dim3(256), 0, st, prior, nz, nnz);
    if (cnt > 0) hipLaunchKernelGGL(k_subtract_at, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, nz, map, h, cnt);
}
void launch_logdet(hipStream_t st, const double *L, const long long *diagoff, const unsigned char *own, int n, double *part,
                   int nparts, double *out) {
    hipLaunchKernelGGL(k_logdet_partial, dim3(nparts), dim3(256), 0, st, L, diagoff, own, n, part);
    hipLaunchKernelGGL(k_logdet_final, dim3(1), dim3(256), 0, st, part, nparts, out);
}
void launch_gather(hipStream_t st, const double *src, const long long *off, long long cnt, double *out) {
    if (cnt <= 0) return;
    hipLaunchKernelGGL(k_gather, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, src, off, cnt, out);
}
void launch_seg_wsum(hipStream_t st, const double *src, const long long *segptr, long long nseg, const long long *off,
                     const double *w, double *out) {
    if (nseg <= 0) return;
    hipLaunchKernelGGL(k_seg_wsum, dim3((unsigned)nseg), dim3(64), 0, st, src, segptr, off, w, out);
}
void launch_seg_wsum_pairs(hipStream_t st, const double *src, const long long *segptr, long long nseg, const long long *off,
                           const int *pi, const int *qi, const double *vals, double *out) {
    if (nseg <= 0) return;
    hipLaunchKernelGGL(k_seg_wsum_pairs, dim3((unsigned)nseg), dim3(64), 0, st, src, segptr, off, pi, qi, vals, out);
}
void launch_gather_diag(hipStream_t st, const double *src, const long long *diagoff, const int *perm, int n, double *out) {
    hipLaunchKernelGGL(k_gather_diag, dim3(cdiv(n, 256)), dim3(256), 0, st, src, diagoff, perm, n, out);
}

}  // namespace gmrfx
