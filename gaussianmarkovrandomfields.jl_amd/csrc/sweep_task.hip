// sweep_task.hip -- SWEEP TASKS: one workgroup runs the forward (resp. backward) substitution of a whole bottom
// subtree of the supernodal tree on a LOCAL VECTOR kept in LDS (Symbolic::swt_*, symbolic.h).
//
// Why: in the level-scheduled multifrontal sweeps every front hands its update vector W_s ((r-c) x nrhs doubles) to
// its parent through HBM. At 64 right-hand sides that hand-off is 3.6 GB per forward sweep of the 10^6-node 2-D
// SPDE precision -- more than the factor itself (1.55 GB) -- and the tiny fronts at the bottom of the tree (avg 11
// columns, 38 trailing rows) are a chain of dependent HBM round trips each. Inside a task nothing is handed off:
// the subtree's own rows of X (contiguous in the elimination order) and the root's trailing rows form one local
// vector V (<= TASK_ROWS rows x 64 columns, 144 KB of LDS); front after front (postorder), y_s = L11^-1 b_s and
// V[trailing rows of s] -= L21 y_s is a right-looking update in LDS. The only HBM traffic is the panels (read once,
// prefetched into L2 at the start of the task), the task's slice of X (read once, written once) and the root's
// update vector.
//
// Work split: 16 waves = 4 row-tile slots x 4 column tiles of the right-hand sides. Measured (clock64 stamps inside
// one workgroup, and variants with parts compiled out): a front costs ~4000 cycles whatever the split -- one wave
// issues an FP64 MFMA only every ~138 cycles and spends a few hundred vector instructions per front on clamps, masks
// and addresses -- while barriers and memory latency (operands are requested one front ahead, the panels are warmed
// into L2) are < 15 % of it. One wave per (row tile, 4 column tiles) ran 6000-7000 cycles per front (32 MFMAs from one
// wave), two column tiles per wave the same as one. The task kernels take 0.65 ms (forward) / 0.57 ms (backward) at
// cfg 2 for 86 % of all fronts: the forward sweep drops from 3.15 to 2.70 ms, the backward sweep from 2.28 to 2.20.
//
// Summation order is fixed (fronts in postorder, one owner per entry): bit-reproducible like the rest of the solver.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "kernels.h"

namespace gmrfx {

typedef gmrfx_d4 d4;

constexpr int TASK_ROWS = 288;       // local-vector rows (Symbolic::swt_rows <= this)
constexpr int TASK_MAXF = 64;        // fronts per task (host enforces)
constexpr int TPW = 1;               // 16-column tiles of the right-hand sides per wave (measured: 1 beats 2 and 4)
constexpr int TASK_SLOTS = 4;        // row-tile slots
constexpr int TASK_GROUPS = 16;      // row groups of the copy loops (= threads / NC)
// A workgroup keeps NC columns of the local vector: NC = 64 -> 16 waves (4 row-tile slots x 4 column tiles; one wave issues
// an FP64 MFMA only every ~138 cycles, so the MFMAs of a row tile are spread over the waves of four SIMDs), 144 KB of LDS,
// ONE workgroup per CU. NC = 32 -> 8 waves, 72 KB: the two column halves of a task are two workgroups (ids b and b + 8: same
// XCD, so the second reader of the task's panels finds them in that XCD's L2) and TWO workgroups are resident per CU -- the
// memory phases of one (panel warm-up, the slice of X in, x out) run under the latency-bound front chain of the other.

// V is stored row-major with NC columns; the 16-column tiles of odd rows are swapped pairwise so that the two
// k-rows a ds_read_b64 lane group (lanes 0-31 = two k-rows x 16 columns) touches fall into different halves of the
// 64 LDS banks (row stride = 512 / 256 B = 0 mod 256 would otherwise be a 2-way conflict on every operand read).
template <int NC> __device__ __forceinline__ int vidx(int row, int col) { return row * NC + (col ^ ((row & 1) << 4)); }

// which (task, column half) a workgroup runs; false: nothing to do
template <int NC> __device__ __forceinline__ bool task_of_block(int ntasks, int nr, int &t, int &cbase) {
    const int b = blockIdx.x;
    if (NC == 64) { t = b; cbase = 0; }
    else { t = ((b >> 4) << 3) | (b & 7); cbase = ((b >> 3) & 1) * NC; }
    return t < ntasks && cbase < nr;
}


struct TaskMeta { int c, r, ld, o; long long pp, rp; };   // per front: columns, rows, panel ld, first own local row, panel / row-list offsets

// A front's geometry is the same for every lane: read it into SCALAR registers, so that the address arithmetic of the
// operand requests runs on the scalar unit.
__device__ __forceinline__ TaskMeta uniform_meta(const TaskMeta *meta, int f) {
    const TaskMeta v = meta[f];
    TaskMeta m;
    m.c = __builtin_amdgcn_readfirstlane(v.c);
    m.r = __builtin_amdgcn_readfirstlane(v.r);
    m.ld = __builtin_amdgcn_readfirstlane(v.ld);
    m.o = __builtin_amdgcn_readfirstlane(v.o);
    m.pp = ((long long)__builtin_amdgcn_readfirstlane((int)(v.pp >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v.pp);
    m.rp = ((long long)__builtin_amdgcn_readfirstlane((int)(v.rp >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v.rp);
    return m;
}

// element (k,q) of L11^-1 from the panel (strict lower part stored transposed in the strict upper triangle, diagonal =
// reciprocal of L's): unconditional clamped load + arithmetic mask (see small.hip)
__device__ __forceinline__ double tinv_elem(const double *__restrict__ P, int ld, int c, int k, int q, bool lower) {
    const int kk = min(k, c - 1), qq = min(q, c - 1);
    const double v = P[min(kk, qq) + (long long)max(kk, qq) * ld];
    const bool on = (k < c && q < c) && (lower ? (q < k) : (q > k));
    double x = v * (on ? 1.0 : 0.0);
    if (k == q && k < c) x = fast_rcp(v);
    return x;
}

// Loads the geometry of the task's fronts into LDS (one round trip for the whole task instead of one per front) and
// touches the task's panels and local-row lists, which are contiguous in HBM (postorder), so that the per-front
// operand loads hit L2.
template <int NC> __device__ __forceinline__ double task_prologue(const DevSym &S, const SweepTask &T, TaskMeta *meta, const double *__restrict__ L) {
    constexpr int TASK_THREADS = NC * 16;
    const int tid = threadIdx.x;
    const int nf = T.s1 - T.s0 + 1;
    if (tid < nf) {
        const int s = T.s0 + tid;
        TaskMeta m;
        const int first = S.sfirst[s];
        m.c = S.sfirst[s + 1] - first;
        m.rp = S.rowptr[s];
        m.r = (int)(S.rowptr[s + 1] - m.rp);
        m.ld = S.ld[s];
        m.o = first - T.col0;
        m.pp = S.panelptr[s];
        meta[tid] = m;
    }
    // L2 warm-up: one load per 128-byte line, summed into a value that is never used for arithmetic (returned and
    // stored only under a condition that cannot hold)
    double sink = 0.0;
    for (long long q = T.p0 + (long long)tid * 16; q < T.p1; q += TASK_THREADS * 16) sink += L[q];
    int isink = 0;
    for (long long q = T.rp0 + (long long)tid * 32; q < T.rp1; q += TASK_THREADS * 32) isink += S.lrow[q];
    return sink + (double)isink;
}

// acc[t] += a (this lane's element of a 16 x 4 A operand) x V[row kq][column tile t], t = 0..3
template <int NC> __device__ __forceinline__ void mfma4_lds(d4 (&acc)[TPW], const double a, const double *V, const int kq, const int cl) {
    const double *vr = V + kq * NC;
    const int sw = (kq & 1) << 4;
#pragma unroll
    for (int t = 0; t < TPW; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, vr[(t * 16 + cl) ^ sw], acc[t], 0, 0, 0);
}
// V[rows l2[rr]][all four column tiles] -= acc (rows lk + 4 rr of the tile; distinct rows inside a front)
template <int NC> __device__ __forceinline__ void scatter_sub(double *V, const int (&l2)[4], const d4 (&acc)[TPW], const int nvalid, const int lk, const int cl) {
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
        if (lk + 4 * rr < nvalid) {
            double *vr = V + l2[rr] * NC;
            const int sw = (l2[rr] & 1) << 4;
#pragma unroll
            for (int t = 0; t < TPW; t++) vr[(t * 16 + cl) ^ sw] -= acc[t][rr];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// forward: V <- [b of the subtree ; 0]; per front y = L11^-1 b (own rows, written to X), V[trailing] -= L21 y
// ------------------------------------------------------------------------------------------------------------
template <int NC> __global__ __launch_bounds__(NC * 16) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_fwd_task(DevSym S, const SweepTask *__restrict__ tasks, int ntasks, const double *__restrict__ L, double *__restrict__ X,
                double *__restrict__ W, int nr_all, int ldx) {
    __shared__ double V[TASK_ROWS * NC];
    __shared__ TaskMeta meta[TASK_MAXF];
    int tsk, cbase;
    if (!task_of_block<NC>(ntasks, nr_all, tsk, cbase)) return;
    X += cbase; W += cbase;                             // this workgroup's NC columns of the right-hand sides
    const int nr = min(nr_all - cbase, NC);
    const SweepTask T = tasks[tsk];
    const int col0 = T.col0, NT = T.nt, nf = T.s1 - T.s0 + 1, mroot = T.mroot;
    const int tid = threadIdx.x;
    const int j = tid % NC, g = tid / NC;               // g: TASK_GROUPS row groups
    const int jc = min(j, nr - 1);
    const double jm = j < nr ? 1.0 : 0.0;
    const double sink = task_prologue<NC>(S, T, meta, L);
    // the subtree's slice of X: NT contiguous rows, four row loads in flight per thread
    for (int i0 = g; i0 < NT; i0 += 4 * TASK_GROUPS) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = X[(long long)(col0 + min(i0 + TASK_GROUPS * u, NT - 1)) * ldx + jc];
#pragma unroll
        for (int u = 0; u < 4; u++) if (i0 + TASK_GROUPS * u < NT) V[vidx<NC>(i0 + TASK_GROUPS * u, j)] = v[u] * jm;
    }
    for (int i = NT + g; i < NT + mroot; i += TASK_GROUPS) V[vidx<NC>(i, j)] = 0.0;
    __syncthreads();
    constexpr int CT = NC / 16;                         // column tiles = waves per row-tile slot
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wv / CT, th = wv % CT;                // this wave's row-tile slot and its part of the right-hand sides (scalars)
    const int lane = tid & 63;
    const int lm = lane & 15, lk = lane >> 4;
    const int cl = th * 16 * TPW + lm;                  // column of this lane in the first of its tiles
    // Operands that do not depend on the sweep itself -- this slot's rows of the first 16 columns of L11^-1 / of its
    // trailing row tile, and that tile's local rows -- are requested ONE FRONT AHEAD. All addresses are clamped into
    // the front's own panel / row list, so the loads are valid for any geometry (masks are applied at use).
    // (two operand sets used alternately, the front loop unrolled by two: rotating ONE set through a copy at the end of
    //  an iteration would make the copy wait for the loads just requested -- and with it the whole prefetch)
    double opA_o[4], opA_t[4], opB_o[4], opB_t[4];
    int opA_l[4], opB_l[4];
    auto request = [&](int f, double (&xo)[4], double (&xt)[4], int (&xl)[4]) {
        const TaskMeta m = uniform_meta(meta, f);
        const int ntile = (m.r - m.c + 15) >> 4;
        const bool narrow = m.c <= 16;
        if (!(w == 0 || w < ntile || (!narrow && w * 16 < m.c))) return;      // this slot has no work in front f (scalar branch)
        const double *P = L + m.pp;
        const int *lr = S.lrow + m.rp;
        const int i0 = m.c + w * 16;
        const double *pa = P + min(i0 + lm, m.r - 1);
        const int kk = narrow ? 0 : w * 16;       // narrow fronts: every slot computes the one y tile itself (see below)
#pragma unroll
        for (int u = 0; u < 4; u++) xo[u] = tinv_elem(P, m.ld, m.c, kk + lm, 4 * u + lk, true);
#pragma unroll
        for (int u = 0; u < 4; u++) xt[u] = pa[(long long)min(4 * u + lk, m.c - 1) * m.ld] * ((4 * u + lk) < m.c ? 1.0 : 0.0);
        {   // the tile's 16 local rows in ONE load (lane l: row i0 + (l & 15)), handed to the lanes that use them
            const int v = lr[min(i0 + lm, m.r - 1)];
#pragma unroll
            for (int rr = 0; rr < 4; rr++) xl[rr] = __shfl(v, lk + 4 * rr, 64);
        }
    };
    request(0, opA_o, opA_t, opA_l);
    auto front = [&](const int f, double (&ao)[4], double (&at)[4], int (&li)[4], double (&no)[4], double (&nt_)[4], int (&nli)[4]) {
        const TaskMeta m = uniform_meta(meta, f);
        const int c = m.c, r = m.r, ld = m.ld, o = m.o;
        const double *P = L + m.pp;
        const int *lr = S.lrow + m.rp;
        request(min(f + 1, nf - 1), no, nt_, nli);
        const int ntile = (r - c + 15) >> 4;
        double *Xo = X + (long long)(col0 + o) * ldx;       // the front's own rows of X: y goes straight to HBM
        if (c <= 16) {
            // NARROW FRONT (most of the bottom of the tree): ONE barrier. Every active slot computes the single 16 x 16
            // tile y = L11^-1 b itself (redundantly); in the accumulator layout register u of lane (lk, lm) holds row
            // 4 u + lk -- exactly the B operand of k-step u -- so L21 y runs straight from registers.
            if (w == 0 || w < ntile) {
                d4 y[TPW];
#pragma unroll
                for (int t = 0; t < TPW; t++) y[t] = (d4){0.0, 0.0, 0.0, 0.0};
                const int ku = (c + 3) >> 2;          // k-steps that hold columns of the front (scalar)
#pragma unroll
                for (int u = 0; u < 4; u++) if (u < ku) mfma4_lds<NC>(y, ao[u], V, o + min(4 * u + lk, c - 1), cl);
                if (w == 0) {
#pragma unroll
                    for (int t = 0; t < TPW; t++)
#pragma unroll
                        for (int rr = 0; rr < 4; rr++)
                            if (lk + 4 * rr < c && t * 16 + cl < nr) Xo[(long long)(lk + 4 * rr) * ldx + t * 16 + cl] = y[t][rr];
                }
#pragma unroll 1
                for (int it = w; it < ntile; it += TASK_SLOTS) {
                    const int i0 = c + it * 16;
                    double av[4];
                    int l2[4];
                    if (it == w) {
#pragma unroll
                        for (int u = 0; u < 4; u++) { av[u] = at[u]; l2[u] = li[u]; }
                    } else {
                        const double *pa = P + min(i0 + lm, r - 1);
                        const int v = lr[min(i0 + lm, r - 1)];
#pragma unroll
                        for (int rr = 0; rr < 4; rr++) l2[rr] = __shfl(v, lk + 4 * rr, 64);
#pragma unroll
                        for (int u = 0; u < 4; u++) av[u] = pa[(long long)min(4 * u + lk, c - 1) * ld] * ((4 * u + lk) < c ? 1.0 : 0.0);
                    }
                    d4 acc[TPW];
#pragma unroll
                    for (int t = 0; t < TPW; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (u < ku) {
#pragma unroll
                            for (int t = 0; t < TPW; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], y[t][u], acc[t], 0, 0, 0);
                        }
                    scatter_sub<NC>(V, l2, acc, r - i0, lk, cl);
                }
            }
            __syncthreads();
            return;
        }
        // ---- wide front (17..64 columns): slot w < 4 owns own rows 16 w .. 16 w + 15 -----------------------------
        const int k0 = w * 16;
        d4 acc[TPW];
        if (k0 < c) {
#pragma unroll
            for (int t = 0; t < TPW; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int u = 0; u < 4; u++) mfma4_lds<NC>(acc, ao[u], V, o + min(4 * u + lk, c - 1), cl);     // zero beyond column c / above the diagonal
            const int qhi = min(c, k0 + 16);
#pragma unroll 1
            for (int q0 = 16; q0 < qhi; q0 += 16) {
                double av[4];
#pragma unroll
                for (int u = 0; u < 4; u++) av[u] = tinv_elem(P, ld, c, k0 + lm, q0 + 4 * u + lk, true);
#pragma unroll
                for (int u = 0; u < 4; u++) mfma4_lds<NC>(acc, av[u], V, o + min(q0 + 4 * u + lk, c - 1), cl);
            }
        }
        __syncthreads();            // every slot has read b before any y is written
        if (k0 < c) {
#pragma unroll
            for (int t = 0; t < TPW; t++)
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const int k = k0 + lk + 4 * rr;
                    if (k < c) {
                        V[vidx<NC>(o + k, t * 16 + cl)] = acc[t][rr];
                        if (t * 16 + cl < nr) Xo[(long long)k * ldx + t * 16 + cl] = acc[t][rr];
                    }
                }
        }
        __syncthreads();
        // ---- V[trailing rows] -= L21 y ---------------------------------------------------------------------
#pragma unroll 1
        for (int it = w; it < ntile; it += TASK_SLOTS) {
            const int i0 = c + it * 16;
            const double *pa = P + min(i0 + lm, r - 1);
            int l2[4];
#pragma unroll
            for (int t = 0; t < TPW; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
            if (it == w) {
#pragma unroll
                for (int u = 0; u < 4; u++) l2[u] = li[u];
#pragma unroll
                for (int u = 0; u < 4; u++) mfma4_lds<NC>(acc, at[u], V, o + 4 * u + lk, cl);            // c > 16: rows o .. o+15 exist
            } else {
                const int v = lr[min(i0 + lm, r - 1)];
#pragma unroll
                for (int rr = 0; rr < 4; rr++) l2[rr] = __shfl(v, lk + 4 * rr, 64);
                double av[4];
#pragma unroll
                for (int u = 0; u < 4; u++) av[u] = pa[(long long)(4 * u + lk) * ld];
#pragma unroll
                for (int u = 0; u < 4; u++) mfma4_lds<NC>(acc, av[u], V, o + 4 * u + lk, cl);
            }
#pragma unroll 1
            for (int q0 = 16; q0 < c; q0 += 16) {
                double av[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int q = q0 + 4 * u + lk;
                    av[u] = pa[(long long)min(q, c - 1) * ld] * (q < c ? 1.0 : 0.0);
                }
#pragma unroll
                for (int u = 0; u < 4; u++) mfma4_lds<NC>(acc, av[u], V, o + min(q0 + 4 * u + lk, c - 1), cl);
            }
            scatter_sub<NC>(V, l2, acc, r - i0, lk, cl);     // distinct rows inside a front, one wave per row tile: no conflicts
        }
        __syncthreads();
    };
    {
        int f = 0;
        for (; f + 1 < nf; f += 2) {
            front(f, opA_o, opA_t, opA_l, opB_o, opB_t, opB_l);
            front(f + 1, opB_o, opB_t, opB_l, opA_o, opA_t, opA_l);
        }
        if (f < nf) front(f, opA_o, opA_t, opA_l, opB_o, opB_t, opB_l);
    }
    // ---- write-out: the root's update vector W (y went to X front by front) -----------------------------------
    if (j < nr) {
        double *Wr = W + T.woff * ldx;
        for (int i = g; i < mroot; i += TASK_GROUPS) Wr[(long long)i * ldx + j] = V[vidx<NC>(NT + i, j)];
    }
    if (sink == 1.2345678e-300) X[(long long)col0 * ldx] = sink;      // keeps the warm-up loads alive; never true
}

// ------------------------------------------------------------------------------------------------------------
// backward: V <- [y (or z) of the subtree ; x of the root's trailing rows]; per front, root first:
// t = y - L21' x[trailing], x = L11^-T t
// ------------------------------------------------------------------------------------------------------------
template <int NC> __global__ __launch_bounds__(NC * 16) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_bwd_task(DevSym S, const SweepTask *__restrict__ tasks, int ntasks, const double *__restrict__ L, double *__restrict__ X,
                int nr_all, int ldx) {
    __shared__ double V[TASK_ROWS * NC];
    __shared__ TaskMeta meta[TASK_MAXF];
    int tsk, cbase;
    if (!task_of_block<NC>(ntasks, nr_all, tsk, cbase)) return;
    X += cbase;
    const int nr = min(nr_all - cbase, NC);
    const SweepTask T = tasks[tsk];
    const int col0 = T.col0, NT = T.nt, nf = T.s1 - T.s0 + 1, mroot = T.mroot;
    const int tid = threadIdx.x;
    const int j = tid % NC, g = tid / NC;
    const int jc = min(j, nr - 1);
    const double jm = j < nr ? 1.0 : 0.0;
    const double sink = task_prologue<NC>(S, T, meta, L);
    for (int i0 = g; i0 < NT; i0 += 4 * TASK_GROUPS) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = X[(long long)(col0 + min(i0 + TASK_GROUPS * u, NT - 1)) * ldx + jc];
#pragma unroll
        for (int u = 0; u < 4; u++) if (i0 + TASK_GROUPS * u < NT) V[vidx<NC>(i0 + TASK_GROUPS * u, j)] = v[u] * jm;
    }
    {   // x of the root's trailing rows (ancestors of the subtree: final)
        const int *rows = S.rows + T.rroot;
        for (int i0 = g; i0 < mroot; i0 += 4 * TASK_GROUPS) {
            int ri[4];
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) ri[u] = rows[min(i0 + TASK_GROUPS * u, mroot - 1)];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = X[(long long)ri[u] * ldx + jc];
#pragma unroll
            for (int u = 0; u < 4; u++) if (i0 + TASK_GROUPS * u < mroot) V[vidx<NC>(NT + i0 + TASK_GROUPS * u, j)] = v[u] * jm;
        }
    }
    __syncthreads();
    constexpr int CT = NC / 16;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wv / CT, th = wv % CT;
    const int lane = tid & 63;
    const int lm = lane & 15, lk = lane >> 4;
    const int cl = th * 16 * TPW + lm;
    const int k0 = w * 16;
    // one front ahead (see k_fwd_task): the first 32 trailing rows of this slot's own columns with their local rows,
    // and the diagonal 16 x 16 block of L11^-T. Only slots 0..3 ever own columns.
    double opA_1[8], opA_d[4], opB_1[8], opB_d[4];
    int opA_l[8], opB_l[8];
    auto request = [&](int f, double (&x1)[8], int (&xl)[8], double (&xd)[4]) {
        const TaskMeta m = uniform_meta(meta, f);
        if (k0 >= m.c) return;                                             // this slot has no work in front f (scalar branch)
        const double *P = L + m.pp;
        const int *lr = S.lrow + m.rp;
        const double *pa = P + (long long)min(k0 + lm, m.c - 1) * m.ld;
        // rows in pairs along k: a lane loads rows q, q + 1 (q = c + 8 h + 2 lk) of its column in one 16-byte load and
        // feeds k-steps 2 h and 2 h + 1 with them (k-step 2 h + e covers rows c + 8 h + 2 lk + e); the 32 local rows
        // come in ONE load (lane l: row c + (l & 31)). 5 vector memory instructions instead of 16: the address unit,
        // ~16 cycles per instruction whatever its lanes do, is what bounds a front here.
        {
            const int v = lr[min(m.c + (lane & 31), m.r - 1)];
#pragma unroll
            for (int h = 0; h < 4; h++) {
                const gmrfx_d2u a = *(const gmrfx_d2u *)(pa + min(m.c + 8 * h + 2 * lk, m.r - 1));
                x1[2 * h] = a.x; x1[2 * h + 1] = a.y;
                xl[2 * h] = __shfl(v, 8 * h + 2 * lk, 64);
                xl[2 * h + 1] = __shfl(v, 8 * h + 2 * lk + 1, 64);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) xd[u] = tinv_elem(P, m.ld, m.c, k0 + lm, k0 + 4 * u + lk, false);
    };
    request(nf - 1, opA_1, opA_l, opA_d);
    auto front = [&](const int f, double (&a1)[8], int (&l1)[8], double (&ad)[4], double (&n1)[8], int (&nl1)[8], double (&nd)[4]) {
        const TaskMeta m = uniform_meta(meta, f);
        const int c = m.c, r = m.r, ld = m.ld, o = m.o;
        const double *P = L + m.pp;
        const int *lr = S.lrow + m.rp;
        request(max(f - 1, 0), n1, nl1, nd);
        d4 acc[TPW];
        // ---- t = y - L21' x_R: slot w owns own columns 16 w .. 16 w + 15 (nobody else touches those rows of V here)
        if (k0 < c) {
#pragma unroll
            for (int t = 0; t < TPW; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
            if (r > c) {
                const double *pa = P + (long long)min(k0 + lm, c - 1) * ld;
#pragma unroll
                for (int u = 0; u < 8; u++)         // k-step u = 2 h + e holds rows c + 8 h + 2 lk + e (request())
                    if (8 * (u >> 1) + (u & 1) < r - c)
                        mfma4_lds<NC>(acc, a1[u] * ((c + 8 * (u >> 1) + 2 * lk + (u & 1)) < r ? 1.0 : 0.0), V, max(l1[u], 0), cl);
#pragma unroll 1
                for (int q0 = c + 32; q0 < r; q0 += 16) {      // rare in a task (r - c > 32): one 16-row k-block per pass
                    double av[4];
                    int l2[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int qq = min(q0 + 4 * u + lk, r - 1);
                        av[u] = pa[qq];
                        l2[u] = lr[qq];
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        mfma4_lds<NC>(acc, av[u] * ((q0 + 4 * u + lk) < r ? 1.0 : 0.0), V, l2[u], cl);
                }
            }
        }
        if (c <= 16) {
            // NARROW FRONT: t in the accumulator layout (register u = row 4 u + lk) is the B operand of x = L11^-T t:
            // no round trip through LDS, one barrier per front
            if (w == 0) {
                d4 x[TPW];
#pragma unroll
                for (int t = 0; t < TPW; t++) x[t] = (d4){0.0, 0.0, 0.0, 0.0};
                const int ku = (c + 3) >> 2;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (u < ku) {
                        const int kq = o + min(4 * u + lk, c - 1);
                        const double *vr = V + kq * NC;
                        const int sw = (kq & 1) << 4;
#pragma unroll
                        for (int t = 0; t < TPW; t++)
                            x[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad[u], vr[(t * 16 + cl) ^ sw] - acc[t][u], x[t], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int t = 0; t < TPW; t++)
#pragma unroll
                    for (int rr = 0; rr < 4; rr++)
                        if (lk + 4 * rr < c) V[vidx<NC>(o + lk + 4 * rr, t * 16 + cl)] = x[t][rr];
            }
            __syncthreads();
            return;
        }
        if (k0 < c) {
#pragma unroll
            for (int t = 0; t < TPW; t++)
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const int k = k0 + lk + 4 * rr;
                    if (k < c) V[vidx<NC>(o + k, t * 16 + cl)] -= acc[t][rr];
                }
        }
        __syncthreads();
        // ---- x = L11^-T t ------------------------------------------------------------------------------------
        if (k0 < c) {
#pragma unroll
            for (int t = 0; t < TPW; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int u = 0; u < 4; u++) mfma4_lds<NC>(acc, ad[u], V, o + min(k0 + 4 * u + lk, c - 1), cl);
#pragma unroll 1
            for (int q0 = k0 + 16; q0 < c; q0 += 16) {
                double av[4];
#pragma unroll
                for (int u = 0; u < 4; u++) av[u] = tinv_elem(P, ld, c, k0 + lm, q0 + 4 * u + lk, false);   // Linv[q][k], q >= k
#pragma unroll
                for (int u = 0; u < 4; u++) mfma4_lds<NC>(acc, av[u], V, o + min(q0 + 4 * u + lk, c - 1), cl);
            }
        }
        __syncthreads();            // every slot has read t before any x is written
        if (k0 < c) {
#pragma unroll
            for (int t = 0; t < TPW; t++)
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const int k = k0 + lk + 4 * rr;
                    if (k < c) V[vidx<NC>(o + k, t * 16 + cl)] = acc[t][rr];
                }
        }
        __syncthreads();
    };
    {
        int f = nf - 1;
        for (; f >= 1; f -= 2) {
            front(f, opA_1, opA_l, opA_d, opB_1, opB_l, opB_d);
            front(f - 1, opB_1, opB_l, opB_d, opA_1, opA_l, opA_d);
        }
        if (f == 0) front(0, opA_1, opA_l, opA_d, opB_1, opB_l, opB_d);
    }
    if (j < nr)
        for (int i = g; i < NT; i += TASK_GROUPS) X[(long long)(col0 + i) * ldx + j] = V[vidx<NC>(i, j)];
    if (sink == 1.2345678e-300) X[(long long)col0 * ldx] = sink;
}

static int task_nc() {      // GMRFX_TASK_NC = 64: one 64-column workgroup per task (one per CU); 32 (default): two column halves, two resident per CU
    static const int v = [] { const char *e = std::getenv("GMRFX_TASK_NC"); const int x = e ? std::atoi(e) : 32; return x == 64 ? 64 : 32; }();
    return v;
}
void launch_sweep_tasks(hipStream_t st, const DevSym &S, int phase, const SweepTask *tasks, int ntasks,
                        const double *L, double *X, double *W, int nr, int ldx, size_t extra_lds) {
    if (ntasks <= 0) return;
    // extra_lds: dynamic LDS nobody uses -- it only lowers the number of resident workgroups per compute unit (pipelined factor +
    // solve: two resident 72 KB task workgroups leave 16 KB per CU, and the panel chain's trsm (32 KB) and SYRK (33 KB) workgroups
    // queue behind them)
    if (task_nc() == 64) {
        if (phase == 1) hipLaunchKernelGGL(k_fwd_task<64>, dim3(ntasks), dim3(1024), 0, st, S, tasks, ntasks, L, X, W, nr, ldx);
        else hipLaunchKernelGGL(k_bwd_task<64>, dim3(ntasks), dim3(1024), 0, st, S, tasks, ntasks, L, X, nr, ldx);
    } else {
        const int grid = ((ntasks + 7) / 8) * 16;       // blocks b and b + 8 (same XCD): the two column halves of one task
        if (phase == 1) hipLaunchKernelGGL(k_fwd_task<32>, dim3(grid), dim3(512), extra_lds, st, S, tasks, ntasks, L, X, W, nr, ldx);
        else hipLaunchKernelGGL(k_bwd_task<32>, dim3(grid), dim3(512), 0, st, S, tasks, ntasks, L, X, nr, ldx);
    }
}
int sweep_task_rows_max() { return TASK_ROWS; }
}  // namespace gmrfx
