// sweep_front.hip -- the backward step of a mid-level front as ONE workgroup and ONE launch (round 5).
// (The same idea for the first half of the FORWARD step -- own rows assembled in LDS, y = L11^-1 b in the same launch, instead of
//  k_fwd_assemble + k_xmul -- was built to parity and measured: 53 / 77 / 82 / 47 / 42 us on levels 5-9 of cfg 2 against 45 / 65 / 68 /
//  37 / 33 for the two launches, whose finer workgroups (32 rows / 16 rows of a front) hide the record -> row table -> child rows
//  chain better than one workgroup per front does. Removed.)
//
// The level schedule ran a big front's backward step as two launches: t = y - L21' x[trailing rows] (k_bwd_gemm_longk: a
// workgroup per 16 or 32 own columns, every one of them gathering ALL trailing rows of x: the mid levels moved 2-3 x their
// algorithmic bytes, profiles/r05_sweep_levels_start_of_round.txt) and x = L11^-T t (k_xmul, t through HBM in between).
// For fronts of at most 128 columns -- every front of levels 5-9 of the 10^6-node 2-D problem, 5 500 of the 5 800 fronts
// outside the sweep tasks -- one workgroup of eight waves now owns the whole front:
//  * the trailing rows of x are gathered ONCE per front: 16 rows x 64 right-hand sides per batch, one 16-byte load per
//    thread, through a double-buffered LDS stage (row stride 640 bytes: the two rows a ds_read_b64 half-wave touches fall
//    into different halves of the banks); their row indices are requested two batches ahead, the rows one batch ahead;
//  * the (16 own columns) x (16 right-hand sides) tiles of the front are dealt to the eight waves so that all four SIMDs work
//    whatever the width: 5-8 column tiles -- a wave per column tile, four right-hand-side tiles each; 3-4 -- two waves per column
//    tile, two right-hand-side tiles each; 1-2 -- four waves per column tile, one tile each (with a wave per column tile only, a
//    32-column front kept two SIMDs busy and two idle: the first form of this kernel was no faster than the two launches). The
//    operand of L21' comes in row pairs (one 16-byte load per lane and 8 rows; the stage holds the rows in the order that pairing implies);
//  * t = y - acc goes to LDS (over the stage), and the same wave forms x = L11^-T t for its 16 rows from the inverse the
//    factorisation left in the panel's upper triangle -- plain loads away from the diagonal tile -- and writes x once.
// Same sums in a fixed order: bit-reproducible. Fronts wider than 128 columns keep the two-launch path.
// (Round 6, measured and dropped: the first half alone -- t = y - L21' x -- for WIDER fronts as one workgroup per 128-column slice, so
//  that a front's trailing rows of x are gathered once per 128 own columns instead of once per 16 or 32 (k_bwd_gemm_longk; levels 10-12
//  of cfg 2 move 2.5-3 x their algorithmic bytes). Built to parity (seven cases against the oracle and the split-K kernel); backward
//  sweep of cfg 2: 1.451 ms without it, 1.446 on levels with >= 192 slices, 1.502 with >= 96 (levels 10-12), 2.008 with >= 32 (levels
//  10-15): a slice walks 60-90 barrier-paced 16-row batches one after the other where the split-K kernel puts eight waves on the K
//  range -- those levels are latency-bound per front, not traffic-bound. Removed.)
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace gmrfx {

typedef gmrfx_d4 d4;
typedef gmrfx_d2u d2u;

constexpr int BF_MAXC = 128;          // columns of a front this kernel takes
constexpr int BF_SS = 80;             // doubles per staged row (64 + 16: see above)
constexpr int BF_TS = 72;             // doubles per row of t (phase 2 reads consecutive rows: a 64-byte skew is enough)

int bwd_front_max_cols() { return BF_MAXC; }

// NTL: right-hand-side tiles per wave (4, 2, 1: see above); wave w owns column tile w / (4 / NTL), tiles t0 .. t0 + NTL - 1
template <int NTL> __device__ __forceinline__ void bwd_front_body(double *sh, const DevSym &S, const int s, const double *__restrict__ L,
                                                                  const double *Xt, const double *Yin, double *Xout, int nr, int ldx) {
    const int first = S.sfirst[s];
    const int c = S.sfirst[s + 1] - first;
    const long long rp = S.rowptr[s];
    const int r = (int)(S.rowptr[s + 1] - rp);
    const int ld = S.ld[s];
    const double *P = L + S.panelptr[s];
    const int *rows = S.rows + rp + c;                 // trailing rows
    const int m = r - c;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, lm = lane & 15, lk = lane >> 4;
    // staging role of this thread: row (tid >> 5) of a batch, right-hand sides 2 seg, 2 seg + 1; the row goes to the stage
    // position the pairing of the operand rows implies: batch row 8 h + 2 q + e -> position 4 (2 h + e) + q
    const int srow = tid >> 5, seg = tid & 31;
    const int spos = 4 * (2 * (srow >> 3) + (srow & 1)) + ((srow >> 1) & 3);
    const int sc0 = min(2 * seg, nr - 1), sc1 = min(2 * seg + 1, nr - 1);
    const double sm0 = 2 * seg < nr ? 1.0 : 0.0, sm1 = 2 * seg + 1 < nr ? 1.0 : 0.0;
    const bool pair_ok = (nr == 64) && ((ldx & 1) == 0);
    auto load_x = [&](int grow) -> d2u {
        const double *px = Xt + (long long)grow * ldx;
        if (pair_ok) return *(const d2u *)(px + 2 * seg);
        return (d2u){px[sc0] * sm0, px[sc1] * sm1};
    };
    const int nb = (m + 15) >> 4;
    d4 acc[NTL];
#pragma unroll
    for (int t = 0; t < NTL; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    constexpr int WPT = 4 / NTL;                        // waves per column tile
    const int ct = w / WPT, t0 = (w % WPT) * NTL;
    const bool active = ct * 16 < c;
    if (nb > 0) {
        // this wave's operand: column min(16 w + lm, c - 1) of L21, rows in pairs (k-step 2 h + e = rows 8 h + 2 lk + e)
        const double *pa = P + (long long)min(ct * 16 + lm, c - 1) * ld + c + 2 * lk;
        int idx1 = rows[min(srow, m - 1)];                      // batch 0
        int idx2 = rows[min(16 + srow, m - 1)];                 // batch 1
        d2u xv = load_x(idx1);
        idx1 = idx2;
        idx2 = rows[min(32 + srow, m - 1)];
        {
            const double z = srow < m ? 1.0 : 0.0;
            *(d2u *)(sh + spos * BF_SS + 2 * seg) = (d2u){xv.x * z, xv.y * z};
        }
        xv = load_x(idx1);                                      // batch 1 (clamped rows: masked when stored)
        // (row offsets clamped so that the FIRST row of a pair stays inside the column; what a pair reads behind the column's
        //  last row -- the next column's first entry, or the zero padding behind the panel -- meets a zero row of the stage)
        const int amax = max(m - 1 - 2 * lk, 0);
        d2u a0 = *(const d2u *)(pa + min(0, amax)), a1 = *(const d2u *)(pa + min(8, amax));
        __syncthreads();
        for (int kb = 0; kb < nb; kb++) {
            const double *st = sh + (kb & 1) * 16 * BF_SS;
            // operands of the next batch (rows clamped into the panel; rows past the end meet zero rows of the stage)
            const int q1 = 16 * (kb + 1);
            const d2u n0 = *(const d2u *)(pa + min(q1, amax)), n1 = *(const d2u *)(pa + min(q1 + 8, amax));
            if (active) {
                // rows past the end of the front in the LAST batch: their stage rows are zero, the operand is whatever the
                // clamped load returned (finite: the panel's own entries)
                const double av[4] = {a0.x, a0.y, a1.x, a1.y};
#pragma unroll
                for (int sidx = 0; sidx < 4; sidx++) {
                    const double *sr = st + (4 * sidx + lk) * BF_SS + 16 * t0 + lm;
#pragma unroll
                    for (int t = 0; t < NTL; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[sidx], sr[16 * t], acc[t], 0, 0, 0);
                }
            }
            // stage batch kb + 1 (its rows arrived during this batch), request batch kb + 2
            if (kb + 1 < nb) {
                const double z = 16 * (kb + 1) + srow < m ? 1.0 : 0.0;
                *(d2u *)(sh + ((kb + 1) & 1) * 16 * BF_SS + spos * BF_SS + 2 * seg) = (d2u){xv.x * z, xv.y * z};
                idx1 = idx2;
                idx2 = rows[min(16 * (kb + 3) + srow, m - 1)];
                xv = load_x(idx1);
            }
            a0 = n0; a1 = n1;
            __syncthreads();
        }
    }
    // ---- t = y - acc into LDS (rows = own columns; zero rows up to the next multiple of 16) ----
    {
        const double *Yb = Yin + (long long)first * ldx;
#pragma unroll
        for (int t = 0; t < NTL; t++) {
            const int j = 16 * (t0 + t) + lm, jc = min(j, nr - 1);
            double yv[4];
#pragma unroll
            for (int rr = 0; rr < 4; rr++) yv[rr] = Yb[(long long)min(ct * 16 + lk + 4 * rr, c - 1) * ldx + jc];
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int i = ct * 16 + lk + 4 * rr;
                if (active) sh[i * BF_TS + j] = (i < c && j < nr) ? yv[rr] - acc[t][rr] : 0.0;
            }
        }
    }
    __syncthreads();
    // ---- x = L11^-T t: rows 16 w .. 16 w + 15; (L11^-T)[i][k] = L11^-1[k][i], k >= i: the panel's upper triangle holds it at
    //      row i, column k (contiguous along i); the diagonal is L's own (inverted here)
    if (!active) return;
    const int i0 = ct * 16;
    d4 x[NTL];
#pragma unroll
    for (int t = 0; t < NTL; t++) x[t] = (d4){0.0, 0.0, 0.0, 0.0};
    const int ic = min(i0 + lm, c - 1);
    {   // diagonal tile: k in [i0, i0 + 16)
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int k = i0 + 4 * u + lk, kc = min(k, c - 1);
            const double v = P[min(ic, kc) + (long long)max(ic, kc) * ld];
            double a = (k < c && i0 + lm < c && k > i0 + lm) ? v : 0.0;
            if (k == i0 + lm && k < c) a = 1.0 / v;
            const double *tr = sh + k * BF_TS + 16 * t0 + lm;
#pragma unroll
            for (int t = 0; t < NTL; t++) x[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, tr[16 * t], x[t], 0, 0, 0);
        }
    }
    const int ctop = (c + 15) & ~15;
#pragma unroll 1
    for (int k0 = i0 + 16; k0 < ctop; k0 += 16) {
        double av[4];
#pragma unroll
        for (int u = 0; u < 4; u++) av[u] = P[ic + (long long)min(k0 + 4 * u + lk, c - 1) * ld];      // (rows k >= c of t are zero)
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const double *tr = sh + (k0 + 4 * u + lk) * BF_TS + 16 * t0 + lm;
#pragma unroll
            for (int t = 0; t < NTL; t++) x[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], tr[16 * t], x[t], 0, 0, 0);
        }
    }
    double *Xo = Xout + (long long)first * ldx;
#pragma unroll
    for (int t = 0; t < NTL; t++)
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int i = i0 + lk + 4 * rr, j = 16 * (t0 + t) + lm;
            if (i < c && j < nr) Xo[(long long)i * ldx + j] = x[t][rr];
        }
}

__global__ __launch_bounds__(512) void k_bwd_front(DevSym S, const int *__restrict__ list, const double *__restrict__ L,
                                                   const double *Xt, const double *Yin, double *Xout, int nr, int ldx) {
    __shared__ double sh[BF_MAXC * BF_TS];            // 72 KB: two stages of 16 x 80 doubles (phase 1), then t (c x 72)
    const int s = list[blockIdx.x];
    const int nct = (S.sfirst[s + 1] - S.sfirst[s] + 15) >> 4;
    if (nct > 4) bwd_front_body<4>(sh, S, s, L, Xt, Yin, Xout, nr, ldx);
    else if (nct > 2) bwd_front_body<2>(sh, S, s, L, Xt, Yin, Xout, nr, ldx);
    else bwd_front_body<1>(sh, S, s, L, Xt, Yin, Xout, nr, ldx);
}

void launch_bwd_front(hipStream_t st, const DevSym &S, const int *list, int nfronts, const double *L, const double *Xt, const double *Yin,
                      double *Xout, int nr, int ldx) {
    if (nfronts <= 0) return;
    hipLaunchKernelGGL(k_bwd_front, dim3(nfronts), dim3(512), 0, st, S, list, L, Xt, Yin, Xout, nr, ldx);
}

// ------------------------------------------------------------------------------------------------------------------------------
// The FORWARD twin (round 6): the whole forward step of a front of at most 128 columns -- own rows assembled from X and the
// children's update vectors, y = L11^-1 b, W = (children) - L21 y -- as ONE workgroup of eight waves and ONE launch instead of three
// (k_fwd_assemble -> k_xmul -> k_fwd_update_rec), for passes wider than the narrow kernels take. Round 5 fused only the first two
// steps and lost (header of this file): the update is two thirds of a mid level's forward time (levels 5-9 of cfg 2: 550 of 790 us)
// and the three launches moved their bytes at 2.7-2.95 TB/s where k_bwd_front moves the same levels at 3.6-4.0.
//  * phase A: b = X[own rows] + the children's rows that land on them (per-edge table DevSym::erow, children in edge order: the
//    same sums in the same order as k_fwd_assemble), 16 rows per thread, into LDS (row stride 72 doubles);
//  * phase B: y = L11^-1 b, a wave per 16 rows (NTL right-hand-side tiles each: the dealing of k_bwd_front), the inverse from the
//    panel's upper triangle -- element (i, k), k < i, sits at row k of column i --, the diagonal tile masked; y goes to Y (the second
//    right-hand-side buffer, where the backward sweep expects it) and, behind a barrier, over b in LDS (rows up to the next
//    multiple of 16 zeroed: phase C's k-steps run over them);
//  * phase C: a wave owns a 32-row tile of the trailing rows for the whole K range and all four right-hand-side tiles, operand rows
//    in pairs (16-byte loads), y from LDS; the children's update vectors enter THROUGH THE MATRIX PIPE like in k_fwd_update_wave
//    (a child row that lands on tile row i is one more k-step against the indicator -1 at row i: exact, fixed order, no LDS tile,
//    no barrier), tile ranges from the per-edge table DevSym::etile; W is written once.
// Same sums in a fixed order per front: bit-reproducible, and the same for a front whatever list it comes in.
constexpr int FF_TS = 72;

template <int NTL> __device__ __forceinline__ void fwd_front_body(double *sh, const DevSym &S, const int s, const double *__restrict__ L,
                                                                  const double *X, double *Y, double *W, int nr, int ldx) {
    const int first = S.sfirst[s];
    const int c = S.sfirst[s + 1] - first;
    const long long rp = S.rowptr[s];
    const int r = (int)(S.rowptr[s + 1] - rp);
    const int ld = S.ld[s];
    const double *P = L + S.panelptr[s];
    const int m = r - c;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, lm = lane & 15, lk = lane >> 4;
    const long long ch0 = S.childptr[s], ch1 = S.childptr[s + 1];
    const int ctop = (c + 15) & ~15;
    // ---- phase A ----
    {
        const int j = lane, jc = min(j, nr - 1);
        const double jm = j < nr ? 1.0 : 0.0;
        constexpr int NU = BF_MAXC / 8;                 // rows w + 8 u of the front
        double x[NU];
#pragma unroll
        for (int u = 0; u < NU; u++) x[u] = (8 * u < c) ? X[(long long)(first + min(w + 8 * u, c - 1)) * ldx + jc] : 0.0;
        for (long long ch = ch0; ch < ch1; ch++) {
            const EdgeRec er = S.edge[ch];
            if (er.nown <= 0) continue;                 // no row of this child lands on an own column
            const int *er_row = S.erow + er.eoff;
            const double *Wd = W + er.woff * ldx;
            int jr[NU];
#pragma unroll
            for (int u = 0; u < NU; u++) jr[u] = (8 * u < c) ? er_row[min(w + 8 * u, c - 1)] : -1;      // wave-uniform
            double v[NU];
#pragma unroll
            for (int u = 0; u < NU; u++) v[u] = jr[u] >= 0 ? Wd[(long long)jr[u] * ldx + jc] : 0.0;
#pragma unroll
            for (int u = 0; u < NU; u++) x[u] += v[u];
        }
#pragma unroll
        for (int u = 0; u < NU; u++) {
            const int i = w + 8 * u;
            if (i < ctop) sh[i * FF_TS + j] = i < c ? x[u] * jm : 0.0;
        }
    }
    __syncthreads();
    // ---- phase B: y = L11^-1 b ----
    constexpr int WPT = 4 / NTL;
    const int ct = w / WPT, t0 = (w % WPT) * NTL;
    const bool active = ct * 16 < c;
    d4 y[NTL];
#pragma unroll
    for (int t = 0; t < NTL; t++) y[t] = (d4){0.0, 0.0, 0.0, 0.0};
    if (active) {
        const int i0 = ct * 16;
        const int ic = min(i0 + lm, c - 1);
        const double *pc = P + (long long)ic * ld;          // column i of the panel: rows k < i hold L11^-1[i][k]
        double an[4];
#pragma unroll
        for (int u = 0; u < 4; u++) an[u] = pc[min(4 * u + lk, c - 1)];
#pragma unroll 1
        for (int k0 = 0; k0 < i0; k0 += 16) {
            double av[4];
#pragma unroll
            for (int u = 0; u < 4; u++) av[u] = an[u];
#pragma unroll
            for (int u = 0; u < 4; u++) an[u] = pc[min(k0 + 16 + 4 * u + lk, c - 1)];       // (the next tile's, requested before this tile's MFMAs)
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const double *br = sh + (k0 + 4 * u + lk) * FF_TS + 16 * t0 + lm;
#pragma unroll
                for (int t = 0; t < NTL; t++) y[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], br[16 * t], y[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {                        // diagonal tile: k <= i
            const int k = i0 + 4 * u + lk, kc = min(k, c - 1);
            const double v = P[min(ic, kc) + (long long)max(ic, kc) * ld];
            double a = (k < c && i0 + lm < c && k < i0 + lm) ? v : 0.0;
            if (k == i0 + lm && k < c) a = 1.0 / v;
            const double *br = sh + k * FF_TS + 16 * t0 + lm;
#pragma unroll
            for (int t = 0; t < NTL; t++) y[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, br[16 * t], y[t], 0, 0, 0);
        }
    }
    __syncthreads();                                         // every wave has read the b it needs
    if (active) {
        double *Yo = Y + (long long)first * ldx;
#pragma unroll
        for (int t = 0; t < NTL; t++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int i = ct * 16 + lk + 4 * rr, j = 16 * (t0 + t) + lm;
                const bool ok = i < c && j < nr;
                sh[i * FF_TS + j] = ok ? y[t][rr] : 0.0;
                if (ok) Yo[(long long)i * ldx + j] = y[t][rr];
            }
    }
    __syncthreads();
    // ---- phase C: W = (children) - L21 y, a wave per 32-row tile ----
    if (m <= 0) return;
    const int nt32 = (m + 31) >> 5;
    double *Ws = W + S.wptr[s] * ldx;
    int jl[4];
#pragma unroll
    for (int t = 0; t < 4; t++) jl[t] = min(16 * t + lm, nr - 1);
    for (int T = w; T < nt32; T += 8) {
        const int i0 = 32 * T;
        d4 acc[2][4];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int t = 0; t < 4; t++) acc[a][t] = (d4){0.0, 0.0, 0.0, 0.0};
        // (a pair that starts at the last trailing row reads one double behind the column: the next column's first entry or the
        //  padding behind the panel; its row is never stored)
        const double *pa = P + c + min(i0 + 2 * lm, m - 1);
        constexpr int KU = 2;
        // (the panel operand of batch n + 1 is requested before the MFMAs of batch n: a batch's 16 MFMAs take about one memory
        //  round trip of a wave's time, and with two waves per SIMD nobody else hides it)
        d2u an[KU];
#pragma unroll
        for (int u = 0; u < KU; u++) an[u] = *(const d2u *)(pa + (long long)min(4 * u + lk, c - 1) * ld);
#pragma unroll 1
        for (int k0 = 0; k0 < c; k0 += 4 * KU) {
            double av[KU][2], bv[KU][4];
#pragma unroll
            for (int u = 0; u < KU; u++) { av[u][0] = an[u].x; av[u][1] = an[u].y; }
#pragma unroll
            for (int u = 0; u < KU; u++) an[u] = *(const d2u *)(pa + (long long)min(k0 + 4 * KU + 4 * u + lk, c - 1) * ld);
#pragma unroll
            for (int u = 0; u < KU; u++) {
                const int kk = k0 + 4 * u + lk;             // (rows c .. ctop - 1 of y are zero in LDS; the panel column is clamped)
                const double *br = sh + kk * FF_TS + lm;
#pragma unroll
                for (int t = 0; t < 4; t++) bv[u][t] = br[16 * t];
            }
#pragma unroll
            for (int u = 0; u < KU; u++)
                if (k0 + 4 * u < c) {
#pragma unroll
                    for (int a = 0; a < 2; a++)
#pragma unroll
                        for (int t = 0; t < 4; t++) acc[a][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][a], bv[u][t], acc[a][t], 0, 0, 0);
                }
        }
        // children: rows [a0, a1) of a child land in this tile; four of them per k-step against the indicator operand
        const int myrow = c + i0 + 2 * lm;
        for (long long ch = ch0; ch < ch1; ch++) {
            const EdgeRec er = S.edge[ch];
            const int a0 = S.etile[er.tptr + T], a1 = S.etile[er.tptr + T + 1];
            const int *reld = S.rel + er.reloff;
            const double *Wd = W + er.woff * ldx;
#pragma unroll 1
            for (int b0 = a0; b0 < a1; b0 += 4 * KU) {
                double sv[KU][2], wv[KU][4];
#pragma unroll
                for (int u = 0; u < KU; u++) {
                    const int row = b0 + 4 * u + lk;
                    const int rc = min(row, a1 - 1);
                    const int d = reld[rc] - myrow;
                    const bool ok = row < a1;
                    sv[u][0] = (ok && d == 0) ? -1.0 : 0.0;
                    sv[u][1] = (ok && d == 1) ? -1.0 : 0.0;
#pragma unroll
                    for (int t = 0; t < 4; t++) wv[u][t] = Wd[(long long)rc * ldx + jl[t]];
                }
#pragma unroll
                for (int u = 0; u < KU; u++)
                    if (b0 + 4 * u < a1) {
#pragma unroll
                        for (int a = 0; a < 2; a++)
#pragma unroll
                            for (int t = 0; t < 4; t++) acc[a][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(sv[u][a], wv[u][t], acc[a][t], 0, 0, 0);
                    }
            }
        }
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int i = i0 + 2 * (lk + 4 * rr) + a;
                if (i < m) {
#pragma unroll
                    for (int t = 0; t < 4; t++)
                        if (16 * t + lm < nr) Ws[(long long)i * ldx + 16 * t + lm] = -acc[a][t][rr];
                }
            }
    }
}

__global__ __launch_bounds__(512) void k_fwd_front(DevSym S, const int *__restrict__ list, const double *__restrict__ L, const double *X, double *Y,
                                                   double *W, int nr, int ldx) {
    __shared__ double sh[BF_MAXC * FF_TS];            // 72 KB: b, then y (c x 72)
    const int s = list[blockIdx.x];
    const int nct = (S.sfirst[s + 1] - S.sfirst[s] + 15) >> 4;
    if (nct > 4) fwd_front_body<4>(sh, S, s, L, X, Y, W, nr, ldx);
    else if (nct > 2) fwd_front_body<2>(sh, S, s, L, X, Y, W, nr, ldx);
    else fwd_front_body<1>(sh, S, s, L, X, Y, W, nr, ldx);
}

void launch_fwd_front(hipStream_t st, const DevSym &S, const int *list, int nfronts, const double *L, const double *X, double *Y, double *W,
                      int nr, int ldx) {
    if (nfronts <= 0) return;
    hipLaunchKernelGGL(k_fwd_front, dim3(nfronts), dim3(512), 0, st, S, list, L, X, Y, W, nr, ldx);
}
}  // namespace gmrfx
