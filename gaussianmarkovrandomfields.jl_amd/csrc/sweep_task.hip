// sweep_task.hip -- SWEEP TASKS: one workgroup runs the forward (resp. backward) substitution of a whole bottom
// subtree of the supernodal tree on a LOCAL VECTOR kept in LDS (Symbolic::swt_*, symbolic.h).
//
// Why: in the level-scheduled multifrontal sweeps every front hands its update vector W_s ((r-c) x nrhs doubles) to
// its parent through HBM. At 64 right-hand sides that hand-off is 3.6 GB per forward sweep of the 10^6-node 2-D
// SPDE precision -- more than the factor itself (1.55 GB) -- and the tiny fronts at the bottom of the tree (avg 11
// columns, 38 trailing rows) are a chain of dependent HBM round trips each. Inside a task nothing is handed off:
// the subtree's own rows of X (contiguous in the elimination order) and the root's trailing rows form one local
// vector V (<= TASK_ROWS rows x 64 columns, 144 KB of LDS); front after front (postorder), y_s = L11^-1 b_s
// overwrites the front's own rows of V and V[trailing rows of s] -= L21 y_s is a right-looking update in LDS. The
// only HBM traffic is the panels (read once, prefetched into L2 at the start of the task), the task's slice of X
// (read once, written once, contiguous) and the root's update vector.
//
// Summation order is fixed (fronts in postorder, one owner per entry): bit-reproducible like the rest of the solver.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace gmrfx {

typedef gmrfx_d4 d4;

constexpr int TASK_ROWS = 288;   // local-vector rows (Symbolic::swt_rows <= this)
constexpr int TASK_MAXF = 64;    // fronts per task (host enforces)

// V is stored row-major with 64 columns; the 16-column tiles of odd rows are swapped pairwise so that the two
// k-rows a ds_read_b64 lane group (lanes 0-31 = two k-rows x 16 columns) touches fall into different halves of the
// 64 LDS banks (row stride = 512 B = 0 mod 256 would otherwise be a 2-way conflict on every operand read).
__device__ __forceinline__ int vidx(int row, int col) { return row * 64 + (col ^ ((row & 1) << 4)); }

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

struct TaskMeta { int c, r, ld, o; long long pp, rp; };   // per front: columns, rows, panel ld, first own local row, panel / row-list offsets

// Loads the geometry of the task's fronts into LDS (one round trip for the whole task instead of one per front) and
// touches the task's panels, which are contiguous in HBM (postorder), so that the per-front operand loads hit L2.
__device__ __forceinline__ double task_prologue(const DevSym &S, const int s0, const int s1, const int col0, TaskMeta *meta,
                                                const double *__restrict__ L) {
    const int tid = threadIdx.x;
    const int nf = s1 - s0 + 1;
    if (tid < nf) {
        const int s = s0 + tid;
        TaskMeta m;
        const int first = S.sfirst[s];
        m.c = S.sfirst[s + 1] - first;
        m.rp = S.rowptr[s];
        m.r = (int)(S.rowptr[s + 1] - m.rp);
        m.ld = S.ld[s];
        m.o = first - col0;
        m.pp = S.panelptr[s];
        meta[tid] = m;
    }
    // L2 warm-up: one 8-byte load per 128-byte line of the task's panels, summed into a value that is never used for
    // arithmetic (returned and stored only under a condition that cannot hold)
    const long long p0 = S.panelptr[s0], p1 = S.panelptr[s1 + 1];
    double sink = 0.0;
    for (long long q = p0 + (long long)tid * 16; q < p1; q += 256 * 16) sink += L[q];
    return sink;
}

// ------------------------------------------------------------------------------------------------------------
// forward: V <- [b of the subtree ; 0]; per front y = L11^-1 b (own rows), V[trailing] -= L21 y
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void k_fwd_task(DevSym S, const int *__restrict__ tk_first, const int *__restrict__ tk_last,
                                                     const double *__restrict__ L, double *__restrict__ X, double *__restrict__ W,
                                                     int nr, int ldx) {
    __shared__ double V[TASK_ROWS * 64];
    __shared__ TaskMeta meta[TASK_MAXF];
    const int s0 = tk_first[blockIdx.x], s1 = tk_last[blockIdx.x];
    const int col0 = S.sfirst[s0], col1 = S.sfirst[s1 + 1], NT = col1 - col0;
    const int tid = threadIdx.x;
    const int j = tid & 63, g = tid >> 6;
    const int jc = min(j, nr - 1);
    const double jm = j < nr ? 1.0 : 0.0;
    const double sink = task_prologue(S, s0, s1, col0, meta, L);
    // the subtree's slice of X: NT contiguous rows, eight row loads in flight per thread
    for (int i0 = g; i0 < NT; i0 += 32) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = X[(long long)(col0 + min(i0 + 4 * u, NT - 1)) * ldx + jc];
#pragma unroll
        for (int u = 0; u < 8; u++) if (i0 + 4 * u < NT) V[vidx(i0 + 4 * u, j)] = v[u] * jm;
    }
    __syncthreads();
    const int mroot = meta[s1 - s0].r - meta[s1 - s0].c;
    for (int i = NT + g; i < NT + mroot; i += 4) V[vidx(i, j)] = 0.0;
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    const int lm = lane & 15, lk = lane >> 4;
    for (int f = 0; f <= s1 - s0; f++) {
        const TaskMeta m = meta[f];
        const int c = m.c, r = m.r, ld = m.ld, o = m.o;
        const double *P = L + m.pp;
        const int *lr = S.lrow + m.rp;
        // ---- y = L11^-1 b: wave w owns own rows 16 w .. 16 w + 15 -------------------------------------------
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
                for (int u = 0; u < 4; u++) av[u] = tinv_elem(P, ld, c, k0 + lm, q0 + 4 * u + lk, true);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int q = o + min(q0 + 4 * u + lk, c - 1);        // av is zero beyond column c
#pragma unroll
                    for (int t = 0; t < 4; t++)
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], V[vidx(q, t * 16 + lm)], acc[t], 0, 0, 0);
                }
            }
        }
        if (c > 16) __syncthreads();            // every wave has read b before any y is written (one tile: same wave)
        if (k0 < c) {
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const int k = k0 + lk + 4 * rr;
                    if (k < c) V[vidx(o + k, t * 16 + lm)] = acc[t][rr];
                }
        }
        __syncthreads();
        // ---- V[trailing rows] -= L21 y ---------------------------------------------------------------------
        const int ntile = (r - c + 15) >> 4;
#pragma unroll 1
        for (int it = wave; it < ntile; it += 4) {
            const int i0 = c + it * 16;
            const double *pa = P + min(i0 + lm, r - 1);
            int li[4];
#pragma unroll
            for (int rr = 0; rr < 4; rr++) li[rr] = lr[min(i0 + lk + 4 * rr, r - 1)];
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
                    const int q = o + min(q0 + 4 * u + lk, c - 1);
#pragma unroll
                    for (int t = 0; t < 4; t++)
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], V[vidx(q, t * 16 + lm)], acc[t], 0, 0, 0);
                }
            }
            // distinct rows inside a front, one wave per row tile: no conflicts
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                if (i0 + lk + 4 * rr < r) {
#pragma unroll
                    for (int t = 0; t < 4; t++) V[vidx(li[rr], t * 16 + lm)] -= acc[t][rr];
                }
            }
        }
        __syncthreads();
    }
    // ---- write-out: y of the whole subtree (contiguous rows of X) and the root's update vector W ---------------
    if (j < nr) {
        for (int i = g; i < NT; i += 4) X[(long long)(col0 + i) * ldx + j] = V[vidx(i, j)];
        double *Wr = W + S.wptr[s1] * ldx;
        for (int i = g; i < mroot; i += 4) Wr[(long long)i * ldx + j] = V[vidx(NT + i, j)];
    }
    if (sink == 1.2345678e-300) X[(long long)col0 * ldx] = sink;      // keeps the warm-up loads alive; never true
}

// ------------------------------------------------------------------------------------------------------------
// backward: V <- [y (or z) of the subtree ; x of the root's trailing rows]; per front, root first:
// t = y - L21' x[trailing], x = L11^-T t
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void k_bwd_task(DevSym S, const int *__restrict__ tk_first, const int *__restrict__ tk_last,
                                                     const double *__restrict__ L, double *__restrict__ X, int nr, int ldx) {
    __shared__ double V[TASK_ROWS * 64];
    __shared__ TaskMeta meta[TASK_MAXF];
    const int s0 = tk_first[blockIdx.x], s1 = tk_last[blockIdx.x];
    const int col0 = S.sfirst[s0], col1 = S.sfirst[s1 + 1], NT = col1 - col0;
    const int tid = threadIdx.x;
    const int j = tid & 63, g = tid >> 6;
    const int jc = min(j, nr - 1);
    const double jm = j < nr ? 1.0 : 0.0;
    const double sink = task_prologue(S, s0, s1, col0, meta, L);
    for (int i0 = g; i0 < NT; i0 += 32) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = X[(long long)(col0 + min(i0 + 4 * u, NT - 1)) * ldx + jc];
#pragma unroll
        for (int u = 0; u < 8; u++) if (i0 + 4 * u < NT) V[vidx(i0 + 4 * u, j)] = v[u] * jm;
    }
    __syncthreads();
    {   // x of the root's trailing rows (ancestors of the subtree: final)
        const TaskMeta mr = meta[s1 - s0];
        const int mroot = mr.r - mr.c;
        const int *rows = S.rows + mr.rp + mr.c;
        for (int i0 = g; i0 < mroot; i0 += 32) {
            int ri[8];
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) ri[u] = rows[min(i0 + 4 * u, mroot - 1)];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = X[(long long)ri[u] * ldx + jc];
#pragma unroll
            for (int u = 0; u < 8; u++) if (i0 + 4 * u < mroot) V[vidx(NT + i0 + 4 * u, j)] = v[u] * jm;
        }
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    const int lm = lane & 15, lk = lane >> 4;
    for (int f = s1 - s0; f >= 0; f--) {
        const TaskMeta m = meta[f];
        const int c = m.c, r = m.r, ld = m.ld, o = m.o;
        const double *P = L + m.pp;
        const int *lr = S.lrow + m.rp;
        const int k0 = wave * 16;
        d4 acc[4];
        // ---- t = y - L21' x_R: wave w owns own columns 16 w .. 16 w + 15 (nobody else touches those rows of V here)
        if (k0 < c) {
            const double *pa = P + (long long)min(k0 + lm, c - 1) * ld;
#pragma unroll
            for (int t = 0; t < 4; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
            for (int q0 = c; q0 < r; q0 += 32) {      // two 16-row k-blocks per pass: 8 operand + 8 index loads in flight
                double av[8];
                int li[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int qq = min(q0 + 4 * u + lk, r - 1);
                    av[u] = pa[qq];
                    li[u] = lr[qq];
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const double a_ = av[u] * ((q0 + 4 * u + lk) < r ? 1.0 : 0.0);
#pragma unroll
                    for (int t = 0; t < 4; t++)
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_, V[vidx(li[u], t * 16 + lm)], acc[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const int k = k0 + lk + 4 * rr;
                    if (k < c) V[vidx(o + k, t * 16 + lm)] -= acc[t][rr];
                }
        }
        if (c > 16) __syncthreads();
        // ---- x = L11^-T t ------------------------------------------------------------------------------------
        if (k0 < c) {
#pragma unroll
            for (int t = 0; t < 4; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
            for (int q0 = k0; q0 < c; q0 += 16) {
                double av[4];
#pragma unroll
                for (int u = 0; u < 4; u++) av[u] = tinv_elem(P, ld, c, k0 + lm, q0 + 4 * u + lk, false);   // Linv[q][k], q >= k
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int q = o + min(q0 + 4 * u + lk, c - 1);
#pragma unroll
                    for (int t = 0; t < 4; t++)
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], V[vidx(q, t * 16 + lm)], acc[t], 0, 0, 0);
                }
            }
        }
        if (c > 16) __syncthreads();            // every wave has read t before any x is written
        if (k0 < c) {
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const int k = k0 + lk + 4 * rr;
                    if (k < c) V[vidx(o + k, t * 16 + lm)] = acc[t][rr];
                }
        }
        __syncthreads();
    }
    if (j < nr)
        for (int i = g; i < NT; i += 4) X[(long long)(col0 + i) * ldx + j] = V[vidx(i, j)];
    if (sink == 1.2345678e-300) X[(long long)col0 * ldx] = sink;
}

void launch_sweep_tasks(hipStream_t st, const DevSym &S, int phase, const int *tk_first, const int *tk_last, int ntasks,
                        const double *L, double *X, double *W, int nr, int ldx) {
    if (ntasks <= 0) return;
    if (phase == 1) hipLaunchKernelGGL(k_fwd_task, dim3(ntasks), dim3(256), 0, st, S, tk_first, tk_last, L, X, W, nr, ldx);
    else hipLaunchKernelGGL(k_bwd_task, dim3(ntasks), dim3(256), 0, st, S, tk_first, tk_last, L, X, nr, ldx);
}
int sweep_task_rows_max() { return TASK_ROWS; }

}  // namespace gmrfx
