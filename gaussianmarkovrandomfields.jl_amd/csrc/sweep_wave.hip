// sweep_wave.hip -- SWEEP TASKS, one WAVE per (task, 16 right-hand sides): the forward / backward substitution of a whole
// bottom subtree (Symbolic::swt_*, symbolic.h) on a local vector of 16 columns kept in LDS.
//
// Why this shape (round 3; the form for up to 16 right-hand sides -- wider passes take the chunk form, sweep_chunk.hip). Measured on the
// workgroup version (tools/task_dbg.sh): the memory phases of the task kernels (panel warm-up, slice of X in, x out) take
// 0.26 / 0.37 ms of the 0.81 / 0.89 ms; the FRONT LOOP takes 0.55 ms whatever the operands cost (compiled without any
// operand load: unchanged), with one or with two resident workgroups per CU (unchanged): a front costs ~7900 cycles of a CU
// because 16 waves each spend ~200 vector instructions on addresses, clamps and masks around 6 MFMAs, every row-tile slot
// recomputes y to save a barrier, and the slots still meet at one or three barriers per front. The right-hand-side columns
// of a triangular solve never interact, so here a wave OWNS 16 of them (one MFMA N-tile) for the whole task:
//   * no barrier and no cross-wave hand-off anywhere -- a wave's LDS operations execute in order, which is all the
//     synchronisation a substitution on private columns needs;
//   * nothing is recomputed: a front is a list of OPS of four MFMAs each (one 16 x 16 block of L11^-1 or of L21 against four
//     k-rows of the local vector), 1/4 of the address arithmetic of the 16-wave version per front;
//   * the operands of op i + WT_D - 1 are requested while op i computes, ACROSS fronts (an operand never depends on the
//     sweep), so neither the L2 latency nor the front boundary is exposed;
//   * LDS is sized per launch (16 columns x the rows of the task class): 4 to 7 independent waves per CU at different phases
//     of different tasks, so the memory phases of one run under the MFMA chain of another;
//   * 1 to 16 right-hand sides cost a quarter of 64 (one wave per task instead of four).
// The four waves of a task run on the same XCD (block ids b, b + 8, b + 16, b + 24), so its panels leave HBM once.
//
// Jobs of a front with KB = ceil(c / 16) column blocks (every job accumulates ops into one 16 x 16 tile, then finalises):
//   forward : Y(ty), ty = KB-1 .. 0 : y[ty] = sum_{kb <= ty} Linv[ty, kb] b[kb]   (in place: block ty only reads b above it)
//             U(it), it = 0 .. ntile-1 : V[trailing tile it] -= sum_kb L21[it, kb] y[kb]
//   backward: T(ty), ty = 0 .. KB-1 : t[ty] = y[ty] - sum_ch L21[ch, ty]' x[trailing chunk ch]
//             X(tx), tx = 0 .. KB-1 : x[tx] = sum_{kb >= tx} Linv[kb, tx]' t[kb]  (in place: block tx only reads t below it)
// Summation order is fixed: bit-reproducible, and independent of how the columns are split over waves.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "kernels.h"

namespace gmrfx {

typedef gmrfx_d4 d4;

constexpr int WT_D = 4;          // operand sets in flight
constexpr int WT_MAXF = 64;      // fronts per task (host enforces)

struct WMeta { int c, r, ld, o; long long pp, rp; };      // per front: columns, rows, panel ld, first own local row, panel / row-list offsets

// One op = four MFMA k-steps of one 16 x 16 operand block. All fields are wave-uniform.
struct WOp {
    int kind;           // 0 Y, 1 U, 2 T, 3 X; -1 = past the end
    int first, last;    // first / last op of its job
    int c, r, ld, o;    // the front
    int m0, k0;         // block position (meaning per kind, see issue())
    long long pp, rp;
};

struct WBuf {           // one operand set in flight (per lane) + the op it belongs to
    double a[4];
    int l[4];           // local rows: kind 1 = destination rows of the scatter (needed at the job's last op), kind 2 = the B rows
    WOp op;
};

// wave-uniform value -> scalar register
__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long sgpr64(long long v) {
    return ((long long)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}

// ---- op generator: the op sequence of a task, front after front (scalar state) ---------------------------------------
template <bool FWD> struct WGen {
    const WMeta *meta;
    int nf, f, step;            // current front (processing order), and the position inside its op sequence
    int phase, j, k;            // phase 0 / 1, job index, k index inside the job
    WMeta m;
    int KB, nt;                 // column blocks, trailing 16-row tiles (= chunks)
    __device__ __forceinline__ void load_front() {
        const int fi = FWD ? f : nf - 1 - f;
        const WMeta v = meta[fi];
        m.c = sgpr(v.c); m.r = sgpr(v.r); m.ld = sgpr(v.ld); m.o = sgpr(v.o); m.pp = sgpr64(v.pp); m.rp = sgpr64(v.rp);
        KB = (m.c + 15) >> 4;
        nt = (m.r - m.c + 15) >> 4;
        if (FWD) { phase = 0; j = KB - 1; k = 0; }
        else { phase = nt > 0 ? 0 : 1; j = 0; k = 0; }
    }
    __device__ __forceinline__ void init(const WMeta *mt, int nfronts) { meta = mt; nf = nfronts; f = 0; if (nf > 0) load_front(); }
    __device__ __forceinline__ void next_front() { f++; if (f < nf) load_front(); }
    // the current op, then advance
    __device__ __forceinline__ WOp next() {
        WOp op;
        op.kind = -1; op.first = op.last = 0; op.c = op.r = op.ld = op.o = op.m0 = op.k0 = 0; op.pp = op.rp = 0;
        if (f >= nf) return op;
        op.c = m.c; op.r = m.r; op.ld = m.ld; op.o = m.o; op.pp = m.pp; op.rp = m.rp;
        if (FWD) {
            if (phase == 0) {             // Y(j): kb = k = 0 .. j
                op.kind = 0; op.m0 = 16 * j; op.k0 = 16 * k; op.first = k == 0; op.last = k == j;
                if (++k > j) { k = 0; if (--j < 0) { if (nt > 0) { phase = 1; j = 0; } else next_front(); } }
            } else {                      // U(j): kb = k = 0 .. KB-1
                op.kind = 1; op.m0 = m.c + 16 * j; op.k0 = 16 * k; op.first = k == 0; op.last = k == KB - 1;
                if (++k >= KB) { k = 0; if (++j >= nt) next_front(); }
            }
        } else {
            if (phase == 0) {             // T(j): chunk k = 0 .. nt-1
                op.kind = 2; op.m0 = 16 * j; op.k0 = m.c + 16 * k; op.first = k == 0; op.last = k == nt - 1;
                if (++k >= nt) { k = 0; if (++j >= KB) { phase = 1; j = 0; } }
            } else {                      // X(j): kb = j + k, k = 0 .. KB-1-j
                op.kind = 3; op.m0 = 16 * j; op.k0 = 16 * (j + k); op.first = k == 0; op.last = j + k == KB - 1;
                if (j + ++k >= KB) { k = 0; if (++j >= KB) next_front(); }
            }
        }
        return op;
    }
};

// The operands of an op are requested with a FIXED number of loads (four doubles, four ints) whose results are not touched
// before the op is consumed: every mask is folded into the ADDRESS (an element that must read as zero is fetched from a zero
// word, the diagonal of L11^-1 from the precomputed reciprocals), so nothing waits for a load at issue time, and the load
// counter the compiler keeps is exact on every path -- a consume waits for its own operand set only, not for the three
// requested after it. (First version: `value * mask` and conditional row-index loads at issue time -- each op then waited
// for everything in flight, 3600 cycles per op.)
struct WSrc {
    const double *L;        // factor panels
    const double *rdiag;    // 1 / L_jj, elimination order (Device::d_rdiag_)
    const double *zero;     // a zero double
    const int *lrow;        // local rows of the task fronts' trailing rows
};

__device__ __forceinline__ void wissue(WBuf &b, const WOp &op, const WSrc &src, const int first_col, const int lm, const int lk) {
    b.op = op;
    const double *pa[4];
    int li[4];
    const double *P = src.L + op.pp;
    const int c = op.c, r = op.r, ld = op.ld;
#pragma unroll
    for (int u = 0; u < 4; u++) { pa[u] = src.zero; li[u] = 0; }
    if (op.kind == 0 || op.kind == 3) {
        // kind 0: A[m = lm][k = 4u + lk] = Linv[m0 + lm][k0 + 4u + lk] (k0 <= m0);  kind 3: Linv[k0 + 4u + lk][m0 + lm] (k0 >= m0).
        // Linv[i][j], j < i, is stored at (row j, column i) of the panel (transposed into the strict upper triangle).
        const int m = op.m0 + lm;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int q = op.k0 + 4 * u + lk;
            const int i = op.kind == 0 ? m : q, j = op.kind == 0 ? q : m;      // want Linv[i][j], nonzero for j <= i < c
            const double *e = P + (j + i * ld);
            e = i == j ? src.rdiag + (first_col + op.o + i) : e;
            pa[u] = (j <= i && i < c) ? e : src.zero;
        }
    } else if (op.kind == 1) {          // A[m][k] = L21[m0 + lm][k0 + 4u + lk] (m0 = c + 16 it); the tile's destination rows
        const int row = op.m0 + lm;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int q = op.k0 + 4 * u + lk;
            pa[u] = (row < r && q < c) ? P + (row + q * ld) : src.zero;
            li[u] = min(op.m0 + lk + 4 * u, r - 1);
        }
    } else if (op.kind == 2) {          // A[m][k] = L21[k0 + 4u + lk][m0 + lm]' (k0 = c + 16 ch); B rows = local rows of those trailing rows
        const int col = op.m0 + lm;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int q = op.k0 + 4 * u + lk;
            pa[u] = (col < c && q < r) ? P + (q + col * ld) : src.zero;
            li[u] = min(q, r - 1);
        }
    }
    const int *lr = src.lrow + op.rp;
#pragma unroll
    for (int u = 0; u < 4; u++) b.a[u] = *pa[u];
#pragma unroll
    for (int u = 0; u < 4; u++) b.l[u] = lr[li[u]];
}

// compute op `b.op` on the local vector V (16 columns, row-major; row `trash` is scratch) and finalise its job when it is
// the last op. Xt: the task's rows of the right-hand sides, this lane's column (global), row 0 = local row 0.
// CS = columns the local vector STORES (2, 4, 8 or 16: passes of one or two right-hand sides keep 4.6 KB of it instead of 37 KB, so
// 16 waves of a CU are resident instead of 4 -- the task kernels of a 1-RHS solve took 0.31 + 0.08 ms per direction in 4.5 rounds of
// one wave per SIMD; 0.22 + 0.07 now, at the four waves per SIMD their 119-128 VGPRs allow. Forcing five -- 96 VGPRs, 46-63 spilled --
// is slower: 1-RHS solve 2.11 -> 2.66 ms). Lanes of the columns beyond CS read a stored column again and never write.
template <int CS>
__device__ __forceinline__ void wconsume(const WBuf &b, d4 &acc0, d4 &acc1, double *V, double *__restrict__ Xt, const int ldx,
                                         const int trash, const int lm, const int lk, const bool colok) {
    const WOp &op = b.op;
    const int lc = lm & (CS - 1);
    const bool wok = lm < CS;
    if (op.first) { acc0 = (d4){0.0, 0.0, 0.0, 0.0}; acc1 = (d4){0.0, 0.0, 0.0, 0.0}; }
    double bv[4];
    {
        const int base = op.o + op.k0 + lk, top = op.o + op.c - 1;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int row = op.kind == 2 ? max(b.l[u], 0) : min(base + 4 * u, top);
            bv[u] = V[row * CS + lc];
        }
    }
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b.a[0], bv[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b.a[1], bv[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b.a[2], bv[2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b.a[3], bv[3], acc1, 0, 0, 0);
    if (!op.last) return;
    const d4 acc = acc0 + acc1;         // (fixed order: even k-steps, odd k-steps, then their sum)
    // destination rows of the tile (rows that do not exist go to the scratch row): all four read, then all four written
    int dst[4];
    const int lim = op.kind == 1 ? op.r : op.c;
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
        const int k = op.m0 + lk + 4 * rr;
        const int row = op.kind == 1 ? b.l[rr] : op.o + k;
        dst[rr] = (k < lim ? row : trash) * CS + lc;
    }
    if (op.kind == 1 || op.kind == 2) {
        double old[4];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) old[rr] = V[dst[rr]];
        if (wok) {
#pragma unroll
            for (int rr = 0; rr < 4; rr++) V[dst[rr]] = old[rr] - acc[rr];
        }
    } else {
        if (wok) {
#pragma unroll
            for (int rr = 0; rr < 4; rr++) V[dst[rr]] = acc[rr];
        }
        if (op.kind == 0) {             // y goes straight to HBM
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int k = op.m0 + lk + 4 * rr;
                if (k < op.c && colok) Xt[(long long)(op.o + k) * ldx] = acc[rr];
            }
        }
    }
}

// which (task, 16-column tile) this one-wave workgroup runs: the CT tiles of a task are blocks b, b + 8, ... (same XCD)
__device__ __forceinline__ bool wtask_of_block(int ntasks, int CT, int &t, int &tile) {
    const int b = blockIdx.x, per = 8 * CT;
    const int g = b / per, w = b - g * per;
    tile = w >> 3;
    t = g * 8 + (w & 7);
    return t < ntasks;
}

// geometry of the task's fronts -> LDS; one quarter of the task's panels and row lists touched (L2 warm-up: the waves of a
// task start together, the first toucher of a line pays HBM, the others and the operand requests find it in L2)
__device__ __forceinline__ double wprologue(const DevSym &S, const SweepTask &T, WMeta *meta, const double *__restrict__ L, int tile, int CT) {
    const int lane = threadIdx.x;
    const int nf = T.s1 - T.s0 + 1;
    if (lane < nf) {
        const int s = T.s0 + lane;
        WMeta m;
        const int first = S.sfirst[s];
        m.c = S.sfirst[s + 1] - first;
        m.rp = S.rowptr[s];
        m.r = (int)(S.rowptr[s + 1] - m.rp);
        m.ld = S.ld[s];
        m.o = first - T.col0;
        m.pp = S.panelptr[s];
        meta[lane] = m;
    }
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const long long stride = 64LL * 16 * CT;
    long long q = T.p0 + ((long long)tile * 64 + lane) * 16;
    for (; q + 3 * stride < T.p1; q += 4 * stride) { s0 += L[q]; s1 += L[q + stride]; s2 += L[q + 2 * stride]; s3 += L[q + 3 * stride]; }
    for (; q < T.p1; q += stride) s0 += L[q];
    int is = 0;
    for (long long p = T.rp0 + ((long long)tile * 64 + lane) * 32; p < T.rp1; p += 64LL * 32 * CT) is += S.lrow[p];
    return (s0 + s1) + (s2 + s3) + (double)is;
}

template <bool FWD, int CS> __global__ __launch_bounds__(64)
void k_wave_task(DevSym S, const SweepTask *__restrict__ tasks, const int *__restrict__ order, int ntasks, int CT,
                 WSrc src, double *__restrict__ X, double *__restrict__ W, int nr_all, int ldx, int rows_cap) {
    extern __shared__ double wsh[];
    double *V = wsh;                                             // (rows_cap + 1) x CS: the last row is scratch
    WMeta *meta = (WMeta *)(wsh + (size_t)(rows_cap + 1) * CS);  // WT_MAXF
    const double *__restrict__ L = src.L;
    int ti, tile;
    if (!wtask_of_block(ntasks, CT, ti, tile)) return;
    const SweepTask T = tasks[order[ti]];
    const int cbase = tile * 16;
    const int nrl = min(nr_all - cbase, 16);
    const int lane = threadIdx.x, lm = lane & 15, lk = lane >> 4;
    const bool colok = lm < nrl;
    const bool wok = lm < CS;
    const int lc = lm & (CS - 1);
    const int jc = cbase + min(lm, nrl - 1);
    const int col0 = T.col0, NT = T.nt, nf = T.s1 - T.s0 + 1, mroot = T.mroot;
    // ---- the task's slice of the right-hand sides: NT contiguous rows (4 rows per load instruction, 8 in flight) ----
    {
        const double *Xs = X + (long long)col0 * ldx + jc;
        for (int i0 = lk; i0 < NT; i0 += 32) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = Xs[(long long)min(i0 + 4 * u, NT - 1) * ldx];
#pragma unroll
            for (int u = 0; u < 8; u++) if (i0 + 4 * u < NT && wok) V[(i0 + 4 * u) * CS + lc] = colok ? v[u] : 0.0;
        }
        if (FWD) {
            if (wok) for (int i = NT + lk; i < NT + mroot; i += 4) V[i * CS + lc] = 0.0;
        } else {        // x of the root's trailing rows (ancestors of the subtree: final)
            const int *rows = S.rows + T.rroot;
            for (int i0 = lk; i0 < mroot; i0 += 32) {
                int ri[8];
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) ri[u] = rows[min(i0 + 4 * u, mroot - 1)];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = X[(long long)ri[u] * ldx + jc];
#pragma unroll
                for (int u = 0; u < 8; u++) if (i0 + 4 * u < mroot && wok) V[(NT + i0 + 4 * u) * CS + lc] = colok ? v[u] : 0.0;
            }
        }
    }
    const double sink = wprologue(S, T, meta, L, tile, CT);
    __syncthreads();            // (one wave: orders the LDS writes of the metadata before the generator's reads)
    // ---- the op pipeline --------------------------------------------------------------------------------------------
    WGen<FWD> gen;
    gen.init(meta, nf);
    WBuf b0, b1, b2, b3;
    double *Xt = X + (long long)col0 * ldx + cbase + lm;        // row 0 = local row 0, this lane's column
    d4 acc0 = (d4){0.0, 0.0, 0.0, 0.0}, acc1 = (d4){0.0, 0.0, 0.0, 0.0};
    wissue(b0, gen.next(), src, col0, lm, lk);
    wissue(b1, gen.next(), src, col0, lm, lk);
    wissue(b2, gen.next(), src, col0, lm, lk);
    static_assert(WT_D == 4, "the pipeline below rotates four operand sets");
    while (b0.op.kind >= 0) {
        wissue(b3, gen.next(), src, col0, lm, lk);
        wconsume<CS>(b0, acc0, acc1, V, Xt, ldx, rows_cap, lm, lk, colok);
        if (b1.op.kind < 0) break;
        wissue(b0, gen.next(), src, col0, lm, lk);
        wconsume<CS>(b1, acc0, acc1, V, Xt, ldx, rows_cap, lm, lk, colok);
        if (b2.op.kind < 0) break;
        wissue(b1, gen.next(), src, col0, lm, lk);
        wconsume<CS>(b2, acc0, acc1, V, Xt, ldx, rows_cap, lm, lk, colok);
        if (b3.op.kind < 0) break;
        wissue(b2, gen.next(), src, col0, lm, lk);
        wconsume<CS>(b3, acc0, acc1, V, Xt, ldx, rows_cap, lm, lk, colok);
    }
    // ---- write-out ---------------------------------------------------------------------------------------------------
    if (FWD) {          // the root's update vector W (y went to X job by job)
        if (colok) {
            double *Wr = W + T.woff * ldx + cbase + lm;
            for (int i = lk; i < mroot; i += 4) Wr[(long long)i * ldx] = V[(NT + i) * CS + lc];
        }
    } else if (colok) {
        for (int i = lk; i < NT; i += 4) Xt[(long long)i * ldx] = V[i * CS + lc];
    }
    if (sink == 1.2345678e-300) X[(long long)col0 * ldx] = sink;      // keeps the warm-up loads alive; never true
}

// 1 / L_jj for every column (elimination order): the sweep tasks read the diagonal of L11^-1 from here
__global__ __launch_bounds__(256) void k_rdiag(const double *__restrict__ L, const long long *__restrict__ diagoff, int n, double *__restrict__ out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < n) out[j] = 1.0 / L[diagoff[j]];
}
void launch_rdiag(hipStream_t st, const double *L, const long long *diagoff, int n, double *out) {
    hipLaunchKernelGGL(k_rdiag, dim3((n + 255) / 256), dim3(256), 0, st, L, diagoff, n, out);
}

// order[]: task ids of one LDS class, heaviest first; rows_cap: rows of the local vector of that class
void launch_wave_tasks(hipStream_t st, const DevSym &S, int phase, const SweepTask *tasks, const int *order, int ntasks, int rows_cap,
                       const double *L, const double *rdiag, const double *zero, double *X, double *W, int nr, int ldx) {
    if (ntasks <= 0) return;
    const int CT = (nr + 15) / 16;
    const int grid = ((ntasks + 7) / 8) * 8 * CT;
    const WSrc src{L, rdiag, zero, S.lrow};
    // columns the local vector stores: the width of the pass rounded up to 2, 4, 8 or 16
    const int cs = nr <= 2 ? 2 : nr <= 4 ? 4 : nr <= 8 ? 8 : 16;
    const size_t lds = (size_t)(rows_cap + 1) * cs * sizeof(double) + WT_MAXF * sizeof(WMeta);
#define GMRFX_WAVE_TASK(CS_)                                                                                                                       \
    do {                                                                                                                                           \
        if (phase == 1) hipLaunchKernelGGL((k_wave_task<true, CS_>), dim3(grid), dim3(64), lds, st, S, tasks, order, ntasks, CT, src, X, W, nr, ldx, rows_cap); \
        else hipLaunchKernelGGL((k_wave_task<false, CS_>), dim3(grid), dim3(64), lds, st, S, tasks, order, ntasks, CT, src, X, W, nr, ldx, rows_cap);          \
    } while (0)
    if (cs == 2) GMRFX_WAVE_TASK(2);
    else if (cs == 4) GMRFX_WAVE_TASK(4);
    else if (cs == 8) GMRFX_WAVE_TASK(8);
    else GMRFX_WAVE_TASK(16);
#undef GMRFX_WAVE_TASK
}

}  // namespace gmrfx
