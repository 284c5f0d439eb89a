// sweep_chunk.hip -- the SWEEP TASKS in chunk form (round 5): one workgroup runs the forward (resp. backward) substitution of
// a whole bottom subtree of the supernodal tree on a LOCAL VECTOR kept in LDS (Symbolic::swt_*, swc_*; symbolic.h).
//
// What a task is (unchanged since round 2): a maximal bottom subtree whose fronts have <= 64 columns and whose local vector
// -- the subtree's own columns, contiguous in the elimination order, + the trailing rows of its root -- has <= 288 rows.
// The update vectors between the fronts of a task never exist in HBM.
//
// What changed, and why (profiles/r05_sweep_levels_start_of_round.txt: 0.82 / 0.89 ms for 1.17 GB, 1.3-1.4 TB/s). The
// round-2 kernels (sweep_task.hip, removed) walked the FRONTS of a task; a front cost a compute unit ~7900 cycles:
//  * ~250 vector instructions per wave and front of clamps, masks, triangle selects and 64-bit address products around
//    6-12 MFMAs -- the task kernels were bound by VALU issue, not by memory;
//  * fronts wider than 16 columns (half of the task fronts at cfg 2) took three barriers and loaded their operands inside
//    the k-loop (an exposed L2 round trip per 16 columns); so did every trailing row beyond the first 32 in the backward
//    sweep (four of five fronts);
//  * the backward sweep of a narrow front kept one row-tile slot of four busy: a chain of ~20 dependent MFMAs on two waves.
// Here a task is a list of CHUNKS: every front is cut into column blocks of <= 16 columns, and a chunk is a narrow front of
// its own -- a 16 x 16 diagonal block whose inverse is a diagonal block of L11^-1, K = its columns, targets = the panel
// rows below the block (own rows of the front's later chunks, then the trailing rows), which are CONTIGUOUS in the panel.
//  * No clamps, no masks: the analysis pads every target-row list to a multiple of 32 with a spare row of the local vector
//    (row CH_SPARE: results of padding rows land there in the forward sweep; it stays zero in the backward sweep, where it
//    meets the operand rows read past the panel), panels are followed by zero columns up to a multiple of 4 and by >= 16
//    zero doubles (symbolic.cpp: panel_span), and the inverse diagonal blocks come PACKED in MFMA operand order with zeros
//    outside the block (k_pack_diag, once per factorisation: 2 KB per chunk).
//  * Addresses are one scalar base + one lane offset per chunk: target rows arrive as ready-made LDS byte offsets (one
//    v_xor per row adds the lane's column), operand rows in 16-byte pairs (MFMA row lm of tile 0 / 1 = rows 2 lm / 2 lm + 1
//    of a 32-row pair: the assignment of matrix rows to MFMA rows is free as long as the row list follows it).
//  * Everything a chunk needs from HBM is requested while the chunk before it computes.
//  * Forward: ONE barrier per chunk; every row-tile slot recomputes the 16 x 16 tile y = D^-1 b itself -- in the accumulator
//    layout register u of lane (lk, lm) holds row 4 u + lk, exactly the B operand of k-step u -- and applies it to its pairs
//    of target tiles straight from registers.
//  * Backward: the chunk TREE depth by depth. Chunks of one depth write disjoint own rows and read rows of finished
//    ancestors: slot q runs the chunks at positions q, q + 4, ... of a depth, barriers only between depths (cfg 2: 62 623
//    chunks in 34 820 depth steps). Each slot follows a PROGRAM made by the analysis: its chunks in order, each with the
//    number of barriers to pass first; all slots pass the same number in total.
// Summation order is fixed (one owner per entry, chunks in program order): bit-reproducible like the rest of the solver.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "kernels.h"

namespace gmrfx {

typedef gmrfx_d4 d4;
typedef Symbolic::SwChunk Chunk;
static_assert(sizeof(Chunk) == 32, "chunk records are read as two 16-byte words");

constexpr int CH_ROWS = 305;         // 288 rows of the local vector + 16 (a chunk reads the 16 rows from its first own row) + the spare row
constexpr int CH_SPARE = 304;
constexpr int CH_MAXC = 96;          // chunk records per task (symbolic.cpp enforces)

int sweep_chunk_spare_row() { return CH_SPARE; }

// byte offset of (row, column) of the local vector: row-major, NC columns; the 16-column tiles of odd rows are swapped
// pairwise so that the two k-rows a ds_read_b64 lane group touches fall into different halves of the LDS banks
template <int NC> __device__ __forceinline__ int vbyte(int row, int col) { return (row * NC + (col ^ ((row & 1) << 4))) * 8; }

template <int NC> __device__ __forceinline__ bool chunk_task_of_block(int ntasks, int nr, int &t, int &cbase) {
    const int b = blockIdx.x;
    if (NC == 64) { t = b; cbase = 0; }
    else { t = ((b >> 4) << 3) | (b & 7); cbase = ((b >> 3) & 1) * NC; }     // blocks b and b + 8 (same XCD): the two column halves
    return t < ntasks && cbase < nr;
}

struct UChunk { long long pa; int ld, o, cc, nt, lr, nbar, id; };
// a chunk's record is the same for every lane: into SCALAR registers
__device__ __forceinline__ UChunk uniform_chunk(const Chunk *meta, int f) {
    const int4 a = ((const int4 *)(meta + f))[0], b = ((const int4 *)(meta + f))[1];
    UChunk m;
    m.pa = ((long long)__builtin_amdgcn_readfirstlane(a.y) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(a.x);
    m.ld = __builtin_amdgcn_readfirstlane(a.z);
    const int oc = __builtin_amdgcn_readfirstlane(a.w);
    m.o = oc & 0xffff; m.cc = oc >> 16;
    m.nt = __builtin_amdgcn_readfirstlane(b.x);
    m.lr = __builtin_amdgcn_readfirstlane(b.y);
    m.nbar = __builtin_amdgcn_readfirstlane(b.z);
    m.id = __builtin_amdgcn_readfirstlane(b.w);
    return m;
}

__device__ __forceinline__ double lds_ld(const char *Vb, int off) { return *(const double *)(Vb + off); }
__device__ __forceinline__ void lds_st(char *Vb, int off, double v) { *(double *)(Vb + off) = v; }

// Records into LDS, the task's panels into L2 (contiguous: postorder), the task's slice of X into the local vector, zeros
// behind it. Returns a value that keeps the warm-up loads alive.
template <int NC, int NTHR> __device__ __forceinline__ double chunk_prologue(const SweepTask &T, const Chunk *__restrict__ recs, int nrec, Chunk *meta,
                                                                              const double *__restrict__ L, const double *__restrict__ X,
                                                                              double *V, int nr, int ldx) {
    const int tid = threadIdx.x;
    for (int i = tid; i < 2 * nrec; i += NTHR) ((int4 *)meta)[i] = ((const int4 *)recs)[i];
    double sink = 0.0;
    for (long long q = T.p0 + (long long)tid * 16; q < T.p1; q += NTHR * 16) sink += L[q];
    constexpr int G = NTHR / NC;
    const int j = tid % NC, g = tid / NC;
    const int jc = min(j, nr - 1);
    const double jm = j < nr ? 1.0 : 0.0;
    const int NT = T.nt, col0 = T.col0;
    for (int i0 = g; i0 < NT; i0 += 4 * G) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = X[(long long)(col0 + min(i0 + G * u, NT - 1)) * ldx + jc];
#pragma unroll
        for (int u = 0; u < 4; u++) if (i0 + G * u < NT) V[vbyte<NC>(i0 + G * u, j) >> 3] = v[u] * jm;
    }
    return sink;
}

// ------------------------------------------------------------------------------------------------------------
// forward: V <- [b of the subtree ; 0]; per chunk y = D^-1 b (own rows, written to X), V[targets] -= L[targets, chunk] y
// ------------------------------------------------------------------------------------------------------------
// what a slot needs of a chunk from HBM: the packed inverse diagonal block, its pair of target tiles (pair w) with their rows
// (a chunk has at most 128 target rows = four pairs, one per slot: symbolic.cpp)
struct FBuf { double dt[4]; gmrfx_d2u a[4]; int4 l[2]; };

template <int NC, int TPW> __global__ __launch_bounds__(4 * (NC / 16 / TPW) * 64, (NC == 32 && TPW == 1) ? 4 : 2)
void k_fwd_chunks(const SweepTask *__restrict__ tasks, int ntasks, const Chunk *__restrict__ recs, const int *__restrict__ listf,
                  const double *__restrict__ dtile, const double *__restrict__ L, double *__restrict__ X, double *__restrict__ W,
                  int nr_all, int ldx) {
    constexpr int CT = NC / 16 / TPW, NTHR = 4 * CT * 64;
    __shared__ double V[CH_ROWS * NC];
    __shared__ Chunk meta[CH_MAXC];
    int tsk, cbase;
    if (!chunk_task_of_block<NC>(ntasks, nr_all, tsk, cbase)) return;
    X += cbase; W += cbase;
    const int nr = min(nr_all - cbase, NC);
    const SweepTask T = tasks[tsk];
    const int col0 = T.col0, NT = T.nt, nch = T.nch, mroot = T.mroot;
    const int tid = threadIdx.x;
    const double sink = chunk_prologue<NC, NTHR>(T, recs + T.c0, nch, meta, L, X, V, nr, ldx);
    for (int i = NT * NC + tid; i < CH_ROWS * NC; i += NTHR) V[i] = 0.0;
    __syncthreads();
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wv / CT, th = wv % CT;                // row-tile slot, column group (scalars)
    const int lane = tid & 63, lm = lane & 15, lk = lane >> 4;
    int clb[TPW];
#pragma unroll
    for (int t = 0; t < TPW; t++) clb[t] = ((th * TPW + t) * 16 + lm) * 8;
    char *Vb = (char *)V;
    FBuf A, B;
    auto request = [&](int f, FBuf &x) {
        const UChunk m = uniform_chunk(meta, f);
        const int npair = (m.nt + 31) >> 5;
        if (!(w == 0 || w < npair)) return;
        const double *dp = dtile + (long long)m.id * 256 + lane;
#pragma unroll
        for (int u = 0; u < 4; u++) x.dt[u] = dp[u * 64];
        if (w < npair) {
            const double *base = L + m.pa + (2 * lm + lk * m.ld) + 32 * w;
            const int *lp = listf + m.lr + lk * 8 + 32 * w;
#pragma unroll
            for (int u = 0; u < 4; u++) x.a[u] = *(const gmrfx_d2u *)(base + (long long)(4 * u) * m.ld);
            x.l[0] = *(const int4 *)(lp);
            x.l[1] = *(const int4 *)(lp + 4);
        }
    };
    request(0, A);
    auto chunk = [&](const int f, FBuf &cur, FBuf &nxt) {
        const UChunk m = uniform_chunk(meta, f);
        if (f + 1 < nch) request(f + 1, nxt);
        const int npair = (m.nt + 31) >> 5;
        if (w == 0 || w < npair) {
            const int ku = (m.cc + 3) >> 2;
            const int rowb = m.o + lk;
            const int bb = rowb * NC * 8, sw = (rowb & 1) << 7;
            d4 y[TPW];
#pragma unroll
            for (int t = 0; t < TPW; t++) y[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (u < ku) {
#pragma unroll
                    for (int t = 0; t < TPW; t++)
                        y[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.dt[u], lds_ld(Vb, bb + u * 4 * NC * 8 + (clb[t] ^ sw)), y[t], 0, 0, 0);
                }
            if (w == 0) {
                double *Xo = X + (long long)(col0 + m.o) * ldx;
#pragma unroll
                for (int t = 0; t < TPW; t++)
#pragma unroll
                    for (int rr = 0; rr < 4; rr++)
                        if (lk + 4 * rr < m.cc && (clb[t] >> 3) < nr) Xo[(long long)(lk + 4 * rr) * ldx + (clb[t] >> 3)] = y[t][rr];
            }
            // V[rows of a pair of target tiles] -= (pair's operand rows) y
            auto apply = [&](const gmrfx_d2u (&av)[4], const int4 &l0, const int4 &l1) {
                d4 a0[TPW], a1[TPW];
#pragma unroll
                for (int t = 0; t < TPW; t++) { a0[t] = (d4){0.0, 0.0, 0.0, 0.0}; a1[t] = (d4){0.0, 0.0, 0.0, 0.0}; }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (u < ku) {
#pragma unroll
                        for (int t = 0; t < TPW; t++) {
                            a0[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u].x, y[t][u], a0[t], 0, 0, 0);
                            a1[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u].y, y[t][u], a1[t], 0, 0, 0);
                        }
                    }
                const int r0[4] = {l0.x, l0.y, l0.z, l0.w};
                const int r1[4] = {l1.x, l1.y, l1.z, l1.w};
#pragma unroll
                for (int t = 0; t < TPW; t++) {
                    double v[4];
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) v[rr] = lds_ld(Vb, r0[rr] ^ clb[t]);
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) lds_st(Vb, r0[rr] ^ clb[t], v[rr] - a0[t][rr]);
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) v[rr] = lds_ld(Vb, r1[rr] ^ clb[t]);
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) lds_st(Vb, r1[rr] ^ clb[t], v[rr] - a1[t][rr]);
                }
            };
            if (w < npair) apply(cur.a, cur.l[0], cur.l[1]);
        }
        __syncthreads();
    };
    {
        int f = 0;
        for (; f + 1 < nch; f += 2) { chunk(f, A, B); chunk(f + 1, B, A); }
        if (f < nch) chunk(f, A, B);
    }
    // the root's update vector W (y went to X chunk by chunk)
    {
        constexpr int G = NTHR / NC;
        const int j = tid % NC, g = tid / NC;
        if (j < nr) {
            double *Wr = W + T.woff * ldx;
            for (int i = g; i < mroot; i += G) Wr[(long long)i * ldx + j] = V[vbyte<NC>(NT + i, j) >> 3];
        }
    }
    if (sink == 1.2345678e-300) X[(long long)col0 * ldx] = sink;      // keeps the warm-up loads alive; never true
}

// ------------------------------------------------------------------------------------------------------------
// backward: V <- [y of the subtree ; x of the root's trailing rows]; per chunk t = y - L[targets, chunk]' x[targets],
// x = D^-T t; the slot programs (chunk tree depth by depth)
// ------------------------------------------------------------------------------------------------------------
template <int G> struct BBuf { double dt[4]; gmrfx_d2u a[G][2]; int4 l[G]; };

template <int NC, int TPW, int G> __global__ __launch_bounds__(4 * (NC / 16 / TPW) * 64, (NC == 32 && TPW == 1) ? 4 : 2)
void k_bwd_chunks(DevSym S, const SweepTask *__restrict__ tasks, int ntasks, const Chunk *__restrict__ recs, const int *__restrict__ listb,
                  const double *__restrict__ dtile, const double *__restrict__ L, double *__restrict__ X, int nr_all, int ldx) {
    constexpr int CT = NC / 16 / TPW, NTHR = 4 * CT * 64;
    __shared__ double V[CH_ROWS * NC];
    __shared__ Chunk meta[CH_MAXC];
    int tsk, cbase;
    if (!chunk_task_of_block<NC>(ntasks, nr_all, tsk, cbase)) return;
    X += cbase;
    const int nr = min(nr_all - cbase, NC);
    const SweepTask T = tasks[tsk];
    const int col0 = T.col0, NT = T.nt, mroot = T.mroot;
    const int tid = threadIdx.x;
    const double sink = chunk_prologue<NC, NTHR>(T, recs + T.b0, T.nbw, meta, L, X, V, nr, ldx);
    {   // x of the root's trailing rows (ancestors of the subtree: final), zeros behind them
        constexpr int GR = NTHR / NC;
        const int j = tid % NC, g = tid / NC;
        const int jc = min(j, nr - 1);
        const double jm = j < nr ? 1.0 : 0.0;
        const int *rows = S.rows + T.rroot;
        for (int i0 = g; i0 < mroot; i0 += 4 * GR) {
            int ri[4];
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) ri[u] = rows[min(i0 + GR * u, mroot - 1)];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = X[(long long)ri[u] * ldx + jc];
#pragma unroll
            for (int u = 0; u < 4; u++) if (i0 + GR * u < mroot) V[vbyte<NC>(NT + i0 + GR * u, j) >> 3] = v[u] * jm;
        }
        for (int i = (NT + mroot) * NC + tid; i < CH_ROWS * NC; i += NTHR) V[i] = 0.0;
    }
    __syncthreads();
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wv / CT, th = wv % CT;
    const int lane = tid & 63, lm = lane & 15, lk = lane >> 4;
    int clb[TPW];
#pragma unroll
    for (int t = 0; t < TPW; t++) clb[t] = ((th * TPW + t) * 16 + lm) * 8;
    char *Vb = (char *)V;
    // this slot's program
    const int i0 = (w > 0 ? T.scnt[0] : 0) + (w > 1 ? T.scnt[1] : 0) + (w > 2 ? T.scnt[2] : 0);
    const int cnt = w == 0 ? T.scnt[0] : w == 1 ? T.scnt[1] : w == 2 ? T.scnt[2] : T.scnt[3];
    const int endbar = w == 0 ? T.sbar[0] : w == 1 ? T.sbar[1] : w == 2 ? T.sbar[2] : T.sbar[3];
    BBuf<G> A, B;
    auto request = [&](int f, BBuf<G> &x) {
        const UChunk m = uniform_chunk(meta, f);
        const int nk = (m.nt + 15) >> 4;
        const double *dp = dtile + (long long)m.id * 256 + (16 * lm + lk);
#pragma unroll
        for (int u = 0; u < 4; u++) x.dt[u] = dp[4 * u];
        const double *base = L + m.pa + (2 * lk + (long long)lm * m.ld);
        const int *lp = listb + m.lr + lk * 4;
#pragma unroll
        for (int g = 0; g < G; g++)
            if (g < nk) {
                x.a[g][0] = *(const gmrfx_d2u *)(base + 16 * g);
                x.a[g][1] = *(const gmrfx_d2u *)(base + 16 * g + 8);
                x.l[g] = *(const int4 *)(lp + 16 * g);
            }
    };
    auto tile = [&](d4 (&acc)[TPW], const gmrfx_d2u &a0, const gmrfx_d2u &a1, const int4 &l) {
#pragma unroll
        for (int t = 0; t < TPW; t++) {
            const double b0 = lds_ld(Vb, l.x ^ clb[t]), b1 = lds_ld(Vb, l.y ^ clb[t]), b2 = lds_ld(Vb, l.z ^ clb[t]), b3 = lds_ld(Vb, l.w ^ clb[t]);
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, b0, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, b1, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b2, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, b3, acc[t], 0, 0, 0);
        }
    };
    auto chunk = [&](const int f, const bool last, BBuf<G> &cur, BBuf<G> &nxt) {
        const UChunk m = uniform_chunk(meta, f);
        if (!last) request(f + 1, nxt);
        for (int b = 0; b < m.nbar; b++) __syncthreads();
        const int nk = (m.nt + 15) >> 4;
        d4 acc[TPW];
#pragma unroll
        for (int t = 0; t < TPW; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int g = 0; g < G; g++)
            if (g < nk) tile(acc, cur.a[g][0], cur.a[g][1], cur.l[g]);
        if (nk > G) {
            const double *base = L + m.pa + (2 * lk + (long long)lm * m.ld);
            const int *lp = listb + m.lr + lk * 4;
#pragma unroll 1
            for (int g = G; g < nk; g++) {
                const gmrfx_d2u a0 = *(const gmrfx_d2u *)(base + 16 * g), a1 = *(const gmrfx_d2u *)(base + 16 * g + 8);
                const int4 l = *(const int4 *)(lp + 16 * g);
                tile(acc, a0, a1, l);
            }
        }
        const int ku = (m.cc + 3) >> 2;
        const int rowb = m.o + lk;
        const int bb = rowb * NC * 8, sw = (rowb & 1) << 7;
        d4 x[TPW];
#pragma unroll
        for (int t = 0; t < TPW; t++) x[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (u < ku) {
#pragma unroll
                for (int t = 0; t < TPW; t++) {
                    const double tv = (4 * u + lk < m.cc) ? lds_ld(Vb, bb + u * 4 * NC * 8 + (clb[t] ^ sw)) - acc[t][u] : 0.0;
                    x[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.dt[u], tv, x[t], 0, 0, 0);
                }
            }
#pragma unroll
        for (int t = 0; t < TPW; t++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++)
                if (lk + 4 * rr < m.cc) lds_st(Vb, bb + rr * 4 * NC * 8 + (clb[t] ^ sw), x[t][rr]);
    };
    if (cnt > 0) request(i0, A);
    {
        int k = 0;
        for (; k + 1 < cnt; k += 2) { chunk(i0 + k, false, A, B); chunk(i0 + k + 1, k + 2 >= cnt, B, A); }
        if (k < cnt) chunk(i0 + k, true, A, B);
    }
    for (int b = 0; b < endbar; b++) __syncthreads();
    {
        constexpr int GR = NTHR / NC;
        const int j = tid % NC, g = tid / NC;
        if (j < nr)
            for (int i = g; i < NT; i += GR) X[(long long)(col0 + i) * ldx + j] = V[vbyte<NC>(i, j) >> 3];
    }
    if (sink == 1.2345678e-300) X[(long long)col0 * ldx] = sink;
}

// The inverse of every chunk's diagonal block in MFMA A-operand order: element (lm, 4 u + lk) of D^-1 at [u][lane]
// (lane = 16 lk + lm), zero outside the chunk's cc columns. D^-1 is a diagonal block of L11^-1, which the factorisation
// leaves transposed in the strict upper triangle of the panel's diagonal block (diagonal: L's own, inverted here).
// The backward sweep reads the same tile transposed: element (4 u + lk, lm) = entry 16 lm + 4 u + lk.
__global__ __launch_bounds__(256) void k_pack_diag(const Chunk *__restrict__ recs, int nchunks, const double *__restrict__ L, double *__restrict__ dtile) {
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= nchunks) return;
    const Chunk c = recs[idx];
    const int id = c.id;
    const int lane = threadIdx.x & 63, lm = lane & 15, lk = lane >> 4;
    const double *Pd = L + c.pa - c.cc;
    const int cc = c.cc;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int k = 4 * u + lk;
        double v = 0.0;
        if (lm < cc && k <= lm) {
            const double e = Pd[k + (long long)lm * c.ld];
            v = k == lm ? 1.0 / e : e;
        }
        dtile[(long long)id * 256 + u * 64 + lane] = v;
    }
}

static int chunk_cfg() {      // GMRFX_TASK_CFG: 0 (default) = 32 columns per workgroup, one 16-column tile per wave (8 waves, two workgroups per CU);
                              // 1 = 32 columns, two tiles per wave (4 waves); 2 = 64 columns, two tiles per wave (8 waves, one workgroup per CU)
    static const int v = [] { const char *e = std::getenv("GMRFX_TASK_CFG"); const int x = e ? std::atoi(e) : 0; return x >= 0 && x <= 2 ? x : 0; }();
    return v;
}
int sweep_chunk_nc() { return chunk_cfg() == 2 ? 64 : 32; }

void launch_pack_diag(hipStream_t st, const Symbolic::SwChunk *recs, int nchunks, const double *L, double *dtile) {
    if (nchunks <= 0) return;
    hipLaunchKernelGGL(k_pack_diag, dim3((nchunks + 3) / 4), dim3(256), 0, st, recs, nchunks, L, dtile);
}

void launch_sweep_chunks(hipStream_t st, const DevSym &S, int phase, const SweepTask *tasks, int ntasks, const Symbolic::SwChunk *recs_fwd,
                         const Symbolic::SwChunk *recs_bwd, const int *listf, const int *listb, const double *dtile, const double *L,
                         double *X, double *W, int nr, int ldx, size_t extra_lds) {
    if (ntasks <= 0) return;
    const int cfg = chunk_cfg();
    const int grid32 = ((ntasks + 7) / 8) * 16;      // blocks b and b + 8 (same XCD): the two column halves of one task
    if (phase == 1) {
        if (cfg == 0) hipLaunchKernelGGL((k_fwd_chunks<32, 1>), dim3(grid32), dim3(512), extra_lds, st, tasks, ntasks, recs_fwd, listf, dtile, L, X, W, nr, ldx);
        else if (cfg == 1) hipLaunchKernelGGL((k_fwd_chunks<32, 2>), dim3(grid32), dim3(256), extra_lds, st, tasks, ntasks, recs_fwd, listf, dtile, L, X, W, nr, ldx);
        else hipLaunchKernelGGL((k_fwd_chunks<64, 2>), dim3(ntasks), dim3(512), 0, st, tasks, ntasks, recs_fwd, listf, dtile, L, X, W, nr, ldx);
    } else {
        if (cfg == 0) hipLaunchKernelGGL((k_bwd_chunks<32, 1, 2>), dim3(grid32), dim3(512), 0, st, S, tasks, ntasks, recs_bwd, listb, dtile, L, X, nr, ldx);
        else if (cfg == 1) hipLaunchKernelGGL((k_bwd_chunks<32, 2, 4>), dim3(grid32), dim3(256), 0, st, S, tasks, ntasks, recs_bwd, listb, dtile, L, X, nr, ldx);
        else hipLaunchKernelGGL((k_bwd_chunks<64, 2, 4>), dim3(ntasks), dim3(512), 0, st, S, tasks, ntasks, recs_bwd, listb, dtile, L, X, nr, ldx);
    }
}
}  // namespace gmrfx
