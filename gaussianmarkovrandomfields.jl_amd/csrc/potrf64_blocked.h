// potrf64_blocked.h -- the 64 x 64 diagonal-block factorisation (Cholesky factor + inverse) in 16-column steps.
//
// The register-patch form (potrf64_body.h) walks the block in sixteen 4-column steps, each a hand-off through LDS and a
// barrier for every wave plus a pivot chain every thread recomputes: 2300 cycles per step, 42 000 per block. Here the
// critical path is ONE wave that never waits for data inside a 16-column step:
//   * the DIAGONAL wave holds the current 16 x 16 diagonal block D and M (starts as I) in registers -- lane (i = lane & 15,
//     q = lane >> 4) owns row i, columns 4 r + q -- and eliminates column by column (fully unrolled, every index static):
//     pivot by v_readlane, column k to the four lanes of each row by ds_bpermute (the LDS crossbar, no LDS memory), row k
//     inside each group of 16 lanes by a DPP row_newbcast that is PART of the update: a[i][.] -= (a_ik / p) a[k][.] is one
//     v_fmac_f64_dpp per live register, and the same on M. 1 / p by v_rcp_f64 + two Newton steps; the columns are stored unscaled
//     and the sixteen square roots taken once per block: L[.][k] = a[.][k] rsqrt(p_k), L^-1 = diag(rsqrt p) M. No barrier, no LDS
//     round trip: 18 instructions per column, 150 cycles (a single wave issues one instruction per 6-8 cycles).
//     (tools/micro/diag16.hip is the first form of this step: v_rsq_f64 per column and v_mov_dpp + v_fma_f64, 225 cycles per column.)
//   * between two diagonal blocks the same wave forms the one strip block and the one update the NEXT diagonal block needs,
//     L[d+1][d] = A[d+1][d] X_dd' and D_{d+1} -= L[d+1][d] L[d+1][d]', as 4 + 4 FP64 MFMAs whose operands are the registers it
//     already holds (the register layout above IS an MFMA operand layout with the contraction index permuted)
//   * everything else -- the other strip blocks, the trailing updates, the off-diagonal blocks of the inverse
//     X_ij = -X_ii sum_k L_ik X_kj -- is 16 x 16 x 16 MFMA products on LDS-resident blocks, done by HELPER waves on the other
//     three SIMDs while the diagonal wave eliminates the next block; two workgroup barriers per 16 columns hand over
//     (X_dd published / L[d+1][d] published), and the helpers are waiting at both when the diagonal wave arrives.
//     The helpers also scale L's columns, and every finished 64 x 16 block column of the panel leaves for memory while the next
//     block is eliminated; the last row of the inverse goes from the helpers' registers straight to the panel.
// Same results as the register-patch form up to rounding (LDL'-style elimination with reciprocal pivots instead of
// square roots inside the updates): parity against the oracle as before, bit-identity only against itself.
// 9.2 us per 64 x 64 block with its inverse (register-patch form: 16.0); cycle budget in DESIGN.md section 3.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace gmrfx {
namespace pb {

typedef gmrfx_d4 d4;

// Workgroup shapes by the widest block of a launch (ND = 16-column steps, NW = waves): narrow fronts come by the thousand per level,
// and what they need is many resident workgroups per CU, not helpers -- <1, 1>: w <= 16, the diagonal wave alone (2.5 KB of LDS);
// <2, 2>: w <= 32, one helper (18 KB); <4, 3>: w <= 48, three helpers (39 KB); <8, 4>: up to 64 columns, the diagonal wave, up to six
// helpers on the other SIMDs and one idle wave (68 KB). Every 16 x 16 product is done by one wave whatever the shape: same bits.
constexpr int ldw_of(int nd) { return 16 * nd + 2; }        // leading dimension (doubles) of the two LDS matrices

// time stamps of the diagonal wave for tools/micro/potrf_prof.hip (compiled out of the library)
#ifdef GMRFX_CYC
__device__ long long g_pb_cyc[64];
__device__ long long g_pb_cycw[8][64];
#define PB_MARK(k) do { if (wave == 0) { __builtin_amdgcn_sched_barrier(0); const long long t_ = clock64(); if (lane == 0) g_pb_cyc[k] = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#define PB_MARKW(k) do { if (wave != 0) { __builtin_amdgcn_sched_barrier(0); const long long t_ = clock64(); if (lane == 0) g_pb_cycw[wave][k] = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define PB_MARK(k)
#define PB_MARKW(k)
#endif

template <int ND>
struct Smem {
    double W[16 * ND * ldw_of(ND)];     // the block: A, then L (lower blocks; strict upper part of diagonal blocks: scratch)
    double V[16 * ND * ldw_of(ND)];     // the inverse: T_ij = sum_k L_ik X_kj, then X_ij; X_dd in the diagonal blocks
    double rs[16];                      // 1 / sqrt(pivot) of the diagonal block being finished
    int simd[8];
};

__device__ __forceinline__ double rsqrt_nr2(double p) {
    double y = __builtin_amdgcn_rsq(p);
    y = y * (1.5 - 0.5 * p * y * y);
    y = y * (1.5 - 0.5 * p * y * y);
    return y;
}
template <int K>
__device__ __forceinline__ double row_bcast(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + K, 0xf, 0xf, true);     // row_newbcast:K
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + K, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bperm(double v, int addr) {
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_d(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

struct Diag16 {
    double a[4], m[4];
    double y0, pv, col;         // of the column about to be eliminated: v_rcp_f64(pivot), the pivot, the column a[i][k] (all rows)
};

#define PB_SB __builtin_amdgcn_sched_barrier(0)

// a -= f * (a of lane K of this lane's row of 16): ONE instruction -- v_fmac_f64 is a VOP2 on this target and takes a DPP
// row_newbcast source. (The compiler does not form it from v_mov_b64_dpp + v_fma_f64, and it cannot see into the asm: the two
// wait states a v_readlane / DPP read of the result needs are given here where the next instruction may be one, NOP = true.)
template <int K, bool NOP>
__device__ __forceinline__ void fmac_bcast(double &a, const double f) {
    if (NOP) asm("v_fmac_f64_dpp %0, %0, -%1 row_newbcast:%2 row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(a) : "v"(f), "n"(K));
    else asm("v_fmac_f64_dpp %0, %0, -%1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(f), "n"(K));
}

// pivot and column of step K into the pipeline registers (the reciprocal starts at once)
template <int K>
__device__ __forceinline__ void diag16_fetch(Diag16 &D, const int i) {
    constexpr int KG = K & 3, KR = K >> 2;
    D.pv = readlane_d(D.a[KR], 16 * KG + K);
    D.col = bperm(D.a[KR], 4 * (16 * KG + i));
    D.y0 = __builtin_amdgcn_rcp(D.pv);
}

// Column K of the 16 x 16 elimination; Wd = the block's position in W (element (i, k) at Wd[k * LDW + i]).
// One wave issues an instruction every 6 - 8 cycles whatever it depends on (tools/micro/valu_lat.hip), so the step is bound
// by its instruction count: v_rcp_f64 + two Newton steps (4) for 1 / p, 2 + 1 for the masked multiplier, one LDS store of the
// UNSCALED column (the square roots wait until the block is done: one v_rsq_f64 chain for all sixteen pivots at once), one
// v_fmac_f64_dpp per live register of [A | M] (5 on average), 2 + 2 + 1 to fetch the next pivot and column.
template <int K, int LDW>
__device__ __forceinline__ void diag16_step(Diag16 &D, double *__restrict__ Wd, const int i, const int q) {
    constexpr int KR = K >> 2, NR = ((K + 1) >> 2) & 3;
    const double y0 = D.y0, pv = D.pv, col = D.col;
    double e = __builtin_fma(-pv, y0, 1.0);
    const double y1 = __builtin_fma(y0, e, y0);
    e = __builtin_fma(-pv, y1, 1.0);
    const double colm = (i > K) ? col : 0.0;
    const double rp = __builtin_fma(y1, e, y1);
    const double f = colm * rp;
    // a[i][K] as it is now for every row = L[i][K] * sqrt(p_K) (rows above the diagonal: scratch nobody reads); the four lanes
    // of a row store the same value
    Wd[K * LDW + i] = col;
    PB_SB;
    if (K < 15) {
        fmac_bcast<K, true>(D.a[NR], f);
        PB_SB;
        diag16_fetch<(K + 1) & 15>(D, i);
        PB_SB;
    }
#pragma unroll
    for (int r = KR; r < 4; r++)
        if (K == 15 || r != NR) fmac_bcast<K, false>(D.a[r], f);
#pragma unroll
    for (int r = 0; r <= KR; r++) fmac_bcast<K, false>(D.m[r], f);
    PB_SB;
}

// D.a holds a new block: M = I, and the first pivot and column on their way (as early as the caller can: right behind the MFMAs that
// produce the block, in front of the barrier that follows them)
__device__ __forceinline__ void diag16_begin(Diag16 &D, const int i, const int q) {
#pragma unroll
    for (int r = 0; r < 4; r++) D.m[r] = (4 * r + q == i) ? 1.0 : 0.0;
    diag16_fetch<0>(D, i);
}

// in: D.a = the block (both triangles), diag16_begin() called. out: the UNSCALED columns a[i][k] in Wd (LDS; L[i][k] = Wd[k][i] Rs[k], lower triangle valid),
// Rs[k] = 1 / sqrt(pivot k) (LDS), x[r] = (L^-1)[i][4 r + q], pivot_ok = this row's pivot was positive
template <int LDW>
__device__ __forceinline__ void diag16_factor(Diag16 &D, double *__restrict__ Wd, double *__restrict__ Rs, double (&x)[4], bool &pivot_ok,
                                              const int i, const int q) {
    diag16_step<0, LDW>(D, Wd, i, q);  diag16_step<1, LDW>(D, Wd, i, q);  diag16_step<2, LDW>(D, Wd, i, q);  diag16_step<3, LDW>(D, Wd, i, q);
    diag16_step<4, LDW>(D, Wd, i, q);  diag16_step<5, LDW>(D, Wd, i, q);  diag16_step<6, LDW>(D, Wd, i, q);  diag16_step<7, LDW>(D, Wd, i, q);
    diag16_step<8, LDW>(D, Wd, i, q);  diag16_step<9, LDW>(D, Wd, i, q);  diag16_step<10, LDW>(D, Wd, i, q); diag16_step<11, LDW>(D, Wd, i, q);
    diag16_step<12, LDW>(D, Wd, i, q); diag16_step<13, LDW>(D, Wd, i, q); diag16_step<14, LDW>(D, Wd, i, q);
    // the sixteen pivots p_i = a[i][i] as their steps left them (rows <= K are never touched again), all square roots at once; the
    // first fifteen come back from this wave's own LDS stores while the last column is eliminated, the sixteenth is in D.pv
    const double pl = Wd[min(i, 14) * LDW + min(i, 14)];
    PB_SB;
    diag16_step<15, LDW>(D, Wd, i, q);
    const double pi = i == 15 ? D.pv : pl;
    pivot_ok = pi > 0.0;
    const double rs = rsqrt_nr2(pi);
    Rs[i] = rs;
    // L^-1 = diag(rs) M; L[i][k] = a[i][k] rs_k is applied to the LDS block by a helper wave (nobody reads L_dd before it leaves)
#pragma unroll
    for (int r = 0; r < 4; r++) x[r] = rs * D.m[r];
}

// One 16 x 16 x 16 product of a helper wave on LDS-resident blocks (block (br, bc) of matrix M at M + 16 bc LDW + 16 br):
//   out[i][j] = (ACC ? acc[i][j] : 0) + (NEG ? -1 : 1) sum_c P[i][c] Q'[j][c],   Q' = QT ? Q' : Q
// returned as reg r of lane (li, lq) = out[li][4 r + lq]; the MFMA computes D[m][n] = out[n][m].
template <bool ACC, bool NEG, bool QT, int LDW>
__device__ __forceinline__ d4 blockop(const double *__restrict__ P, const double *__restrict__ Q, const double *__restrict__ A,
                                      const int li, const int lq) {
    double pv[4], qv[4];
    d4 acc;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        pv[u] = P[(4 * u + lq) * LDW + li];
        qv[u] = QT ? Q[li * LDW + 4 * u + lq] : Q[(4 * u + lq) * LDW + li];
    }
    if (ACC) {
#pragma unroll
        for (int r = 0; r < 4; r++) acc[r] = A[(4 * r + lq) * LDW + li];
    } else acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(NEG ? -qv[u] : qv[u], pv[u], acc, 0, 0, 0);
    return acc;
}
template <int LDW>
__device__ __forceinline__ void block_store(double *__restrict__ O, const d4 v, const int li, const int lq) {
#pragma unroll
    for (int r = 0; r < 4; r++) O[(4 * r + lq) * LDW + li] = v[r];
}

// The THREADS threads of a workgroup factor the w x w block src[i + j * sld] (lower triangle read) and write L (lower
// triangle) and (L^-1)' (strict upper triangle) to P (leading dimension ld). Returns through *info the first column with a
// non-positive pivot (atomicMin of first_col + column).
template <int NW, int ND>
__device__ __forceinline__ void potrf64_blocked(const double *src, const int sld, double *__restrict__ P, const int ld, const int w,
                                                Smem<ND> &S, int *__restrict__ info, const int first_col, const int tid) {
    constexpr int LDW = ldw_of(ND);
    const int wave = tid >> 6, lane = tid & 63, li = lane & 15, lq = lane >> 4;
    const int nd = min((w + 15) >> 4, ND);
    PB_MARK(0);
    Diag16 D;
    int badcol = 0x7fffffff;
    if (lane == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        S.simd[wave] = (int)((hw >> 4) & 3);
    }
    if (wave == 0) {
        // the first diagonal block straight into the registers of the diagonal wave (it starts at once); identity padding beyond w
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int j = 4 * r + lq, hi = max(li, j), lo = min(li, j);
            const double v = src[min(hi, w - 1) + (long long)min(lo, w - 1) * sld];
            D.a[r] = hi < w ? v : (li == j ? 1.0 : 0.0);
        }
        diag16_begin(D, li, lq);
    } else {
        // the other waves: the rest of the lower triangle (rows 16 .. 63) into LDS, the diagonal blocks mirrored (the elimination
        // reads rows as well as columns)
        // (all loads in flight at once: clamped addresses, the predicates only at the LDS writes)
        constexpr int NR = 16 * ND - 16, NC = 16 * ND, NLD = (NR * NC + 64 * (NW - 1) - 1) / (64 * (NW > 1 ? NW - 1 : 1));
        double v[NLD > 0 ? NLD : 1];
#pragma unroll
        for (int p = 0; p < NLD; p++) {
            const int e = tid - 64 + 64 * (NW - 1) * p;
            const int row = 16 + e % (NR > 0 ? NR : 1), c = min(e / (NR > 0 ? NR : 1), NC - 1);
            v[p] = src[min(max(row, c), w - 1) + (long long)min(c, w - 1) * sld];
        }
#pragma unroll
        for (int p = 0; p < NLD; p++) {
            const int e = tid - 64 + 64 * (NW - 1) * p;
            const int row = 16 + e % (NR > 0 ? NR : 1), c = e / (NR > 0 ? NR : 1);
            if (c < NC && row >= c) {
                const double x = (row < w && c < w) ? v[p] : (row == c ? 1.0 : 0.0);
                S.W[c * LDW + row] = x;
                if (row > c && (row >> 4) == (c >> 4)) S.W[row * LDW + c] = x;
            }
        }
    }
    PB_MARK(1);
    int hidx = -1, nh = 0;
    for (int d = 0; d < nd; d++) {
        const bool last = d + 1 == nd;
        double *Wdd = S.W + 16 * d * LDW + 16 * d, *Vdd = S.V + 16 * d * LDW + 16 * d;
        double x[4];
        PB_MARK(2 + 8 * d);
        if (wave == 0) {
            bool pivot_ok;
            diag16_factor<LDW>(D, Wdd, S.rs, x, pivot_ok, li, lq);
#pragma unroll
            for (int r = 0; r < 4; r++) Vdd[(4 * r + lq) * LDW + li] = x[r];
            // first row of this block with a non-positive (or NaN) pivot
            const unsigned long long bm = __ballot(!pivot_ok) & 0xffffull;
            if (bm && badcol == 0x7fffffff) badcol = 16 * d + __builtin_ctzll(bm);
        }
        PB_MARK(3 + 8 * d);
        __syncthreads();                                        // B1: X_dd, L_dd published; the helpers' blocks of step d-1 final
        PB_MARK(4 + 8 * d);
        PB_MARKW(8 * d);
        if (d == 0 && wave != 0) {
            // helpers: the waves that do not share the diagonal wave's SIMD (FP64 MFMA and FP64 VALU share a SIMD's arithmetic)
            const int s0 = S.simd[0];
#pragma unroll
            for (int v = 1; v < NW; v++) {
                const bool h = S.simd[v] != s0;
                if (v == wave && h) hidx = nh;
                nh += h ? 1 : 0;
            }
            if (nh == 0) {          // (every wave on one SIMD: never seen; any wave helps then)
                nh = NW - 1; hidx = wave - 1;
            }
        }
        // ---- window A: the diagonal wave prepares the next diagonal block; helpers: strips with X_dd, finished rows of X
        if (wave == 0) {
            if (!last) {
                const double *Sb = S.W + 16 * d * LDW + 16 * (d + 1);          // A[d+1][d]
                const double *Db = S.W + 16 * (d + 1) * LDW + 16 * (d + 1);    // A[d+1][d+1] (both triangles)
                double s[4];
                d4 acc = (d4){0.0, 0.0, 0.0, 0.0}, dn;
#pragma unroll
                for (int u = 0; u < 4; u++) s[u] = Sb[(4 * u + lq) * LDW + li];
#pragma unroll
                for (int r = 0; r < 4; r++) dn[r] = Db[(4 * r + lq) * LDW + li];
#pragma unroll
                for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u], s[u], acc, 0, 0, 0);
                // acc[r] = L[d+1][d][li][4 r + lq]
                double *Lb = S.W + 16 * d * LDW + 16 * (d + 1);
#pragma unroll
                for (int r = 0; r < 4; r++) Lb[(4 * r + lq) * LDW + li] = acc[r];
#pragma unroll
                for (int u = 0; u < 4; u++) dn = __builtin_amdgcn_mfma_f64_16x16x4f64(-acc[u], acc[u], dn, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) D.a[r] = dn[r];
                diag16_begin(D, li, lq);
            } else {
                // the last diagonal block of the panel: L_dd below / on the diagonal, X_dd' above it
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int c = 4 * r + lq;
                    const double v = li >= c ? Wdd[c * LDW + li] * S.rs[c] : Vdd[li * LDW + c];
                    if (16 * d + li < w && 16 * d + c < w) P[16 * d + li + (long long)(16 * d + c) * ld] = v;
                }
            }
        } else if (hidx >= 0) {
            if (hidx == nh - 1 && !last) {                      // L_dd = (unscaled columns) diag(rs): before B2, where its block column leaves
                double lc[4], rk[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    lc[r] = Wdd[(4 * r + lq) * LDW + li];
                    rk[r] = S.rs[4 * r + lq];
                }
#pragma unroll
                for (int r = 0; r < 4; r++) Wdd[(4 * r + lq) * LDW + li] = lc[r] * rk[r];
            }
            int op = 0, mine = hidx;
            for (int i = d + 2; i < nd; i++, op++) {            // strips L[i][d] = A[i][d] X_dd'
                if (op != mine) continue;
                mine += nh;
                double *B = S.W + 16 * d * LDW + 16 * i;
                const d4 v = blockop<false, false, false, LDW>(B, Vdd, nullptr, li, lq);
                block_store<LDW>(B, v, li, lq);
            }
            for (int j = 0; j < d; j++, op++) {                 // X[d][j] = -X_dd T[d][j]
                if (op != mine) continue;
                mine += nh;
                double *B = S.V + 16 * j * LDW + 16 * d;
                const d4 v = blockop<false, true, true, LDW>(Vdd, B, nullptr, li, lq);
                if (!last) block_store<LDW>(B, v, li, lq);
                else {
                    // nobody reads the last row of X again: straight to the panel (X[16 d + li][16 j + 4 r + lq], transposed)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (16 * d + li < w) P[16 * j + 4 * r + lq + (long long)(16 * d + li) * ld] = v[r];
                }
            }
        }
        if (last) break;
        PB_MARK(5 + 8 * d);
        PB_MARKW(8 * d + 1);
        __syncthreads();                                        // B2: L[d+1][d], the strips and row d of X published
        PB_MARK(6 + 8 * d);
        PB_MARKW(8 * d + 2);
        // ---- window B (beside the elimination of the next diagonal block): trailing updates, products for the inverse, and
        //      the panel's column block d, which is final now, to memory
        if (hidx >= 0) {
            int op = 0, mine = hidx;
            // the two blocks the diagonal wave reads first after the next B1 go first
            for (int j = d + 1; j < nd; j++)
                for (int i = j; i < nd; i++) {
                    if (i == d + 1 && j == d + 1) continue;
                    if (op++ != mine) continue;
                    mine += nh;
                    double *B = S.W + 16 * j * LDW + 16 * i;
                    const d4 v = blockop<true, true, false, LDW>(S.W + 16 * d * LDW + 16 * i, S.W + 16 * d * LDW + 16 * j, B, li, lq);
                    block_store<LDW>(B, v, li, lq);
                }
            for (int i = d + 1; i < nd; i++)
                for (int j = 0; j <= d; j++) {
                    if (op++ != mine) continue;
                    mine += nh;
                    double *B = S.V + 16 * j * LDW + 16 * i;
                    const double *Lid = S.W + 16 * d * LDW + 16 * i, *Xdj = S.V + 16 * j * LDW + 16 * d;
                    const d4 v = j < d ? blockop<true, false, true, LDW>(Lid, Xdj, B, li, lq) : blockop<false, false, true, LDW>(Lid, Xdj, nullptr, li, lq);
                    block_store<LDW>(B, v, li, lq);
                }
        }
        PB_MARKW(8 * d + 3);
        if (wave != 0) {
            constexpr int NS = (16 + NW - 2) / (NW > 1 ? NW - 1 : 1);
            double v[NS];
            const int row = min(lane, 16 * ND - 1);
#pragma unroll
            for (int k = 0; k < NS; k++) {
                const int c = 16 * d + min(wave - 1 + (NW - 1) * k, 15);
                v[k] = row >= c ? S.W[c * LDW + row] : S.V[row * LDW + c];
            }
#pragma unroll
            for (int k = 0; k < NS; k++) {
                const int cc = wave - 1 + (NW - 1) * k;
                if (cc < 16 && lane < w) P[lane + (long long)(16 * d + cc) * ld] = v[k];
            }
        }
        PB_MARKW(8 * d + 4);
    }
    PB_MARK(40);
    PB_MARKW(40);
    if (wave == 0 && lane == 0 && badcol != 0x7fffffff) atomicMin(info, first_col + badcol);
    PB_MARK(41);
}

}  // namespace pb
}  // namespace gmrfx
