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
//    zero doubles (symbolic.cpp: panel_span -- reads past those reach the next panel's first entries: mapped, finite, and
//    multiplied into discarded rows only, see there), and the inverse diagonal blocks come PACKED in MFMA operand order with
//    zeros outside the block (k_pack_diag, once per factorisation: 2 KB per chunk).
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

constexpr int CH_ROWS = 289;         // 288 rows of the local vector (a chunk reads the 16 rows from its first own row: the analysis keeps
constexpr int CH_SPARE = 288;        // own columns + max(root's trailing rows, 15) <= 288) + the spare row
constexpr int CH_MAXC = 96;          // chunk records per task (symbolic.cpp enforces)

int sweep_chunk_spare_row() { return CH_SPARE; }

// byte offset of (row, column) of the local vector: row-major, NC columns. From 32 columns on the 16-column tiles of odd
// rows are swapped pairwise, so that the two k-rows a ds_read_b64 lane group (lanes 0-31 = two k-rows x 16 columns) touches
// fall into different halves of the LDS banks; with 16 columns a row is 128 bytes and consecutive rows do that by themselves.
template <int NC> __device__ __forceinline__ int vbyte(int row, int col) { return (row * NC + (NC >= 32 ? (col ^ ((row & 1) << 4)) : col)) * 8; }

// which (task, column slice) a workgroup runs: the NC-column slices of one task are blocks b, b + 8, ... (same XCD: the later
// readers of the task's panels find them in that XCD's L2)
template <int NC> __device__ __forceinline__ bool chunk_task_of_block(int ntasks, int nr, int &t, int &cbase) {
    const int b = blockIdx.x;
    if (NC == 64) { t = b; cbase = 0; }
    else if (NC == 32) { t = ((b >> 4) << 3) | (b & 7); cbase = ((b >> 3) & 1) * NC; }
    else { t = ((b >> 5) << 3) | (b & 7); cbase = ((b >> 3) & 3) * NC; }
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

typedef int i4v __attribute__((ext_vector_type(4)));
// An operand load: scalar base (the same for every lane: it comes from a chunk's record) + 32-bit lane offset in bytes +
// immediate -- the compiler keeps the base in scalar registers and spends no vector arithmetic per load.
// (Measured and dropped: the same loads as inline asm with hand-counted s_waitcnt vmcnt(N). The compiler's own waits are
//  conservative -- at the first use of an operand requested a chunk ahead it also waits for the request just issued for the next
//  chunk -- but with four workgroups per compute unit that latency is covered by the other three: 587 / 693 us with the
//  compiler's loads, 650-800 / 643 us with the asm ones, which also pin every buffer register for the whole chunk.)
template <int OFF> __device__ __forceinline__ void gl_f64(double &dst, const void *sbase, unsigned voff) { dst = *(const double *)((const char *)sbase + voff + OFF); }
template <int OFF> __device__ __forceinline__ void gl_d2(gmrfx_d2u &dst, const void *sbase, unsigned voff) { dst = *(const gmrfx_d2u *)((const char *)sbase + voff + OFF); }
template <int OFF> __device__ __forceinline__ void gl_i4(i4v &dst, const void *sbase, unsigned voff) { dst = *(const i4v *)((const char *)sbase + voff + OFF); }

__device__ __forceinline__ double lds_ld(const char *Vb, int off) { return *(const double *)(Vb + off); }
__device__ __forceinline__ void lds_st(char *Vb, int off, double v) { *(double *)(Vb + off) = v; }

// ---- cycle stamps (tools/chunk_cycles.py; compiled out of the library: `make cyc` builds libgmrfx_cyc.so with -DGMRFX_CYC) ----
// Where a chunk's time goes, per row-tile slot (wave role) of the two task kernels: s_memtime stamps around (0) the prologue,
// (1) the ISSUE of the next chunk's operand request, (2) the WAIT until this chunk's operands -- requested one chunk ago -- have
// arrived (an explicit s_waitcnt vmcnt(K), K = the loads this wave has just issued for the next chunk: vector memory returns in
// order), (3) y = D^-1 b (forward) / the k-tile loop t = y - L' x with its in-loop requests (backward), (4) the apply + LDS
// read-modify-write (forward) / x = D^-T t + LDS store (backward), (5) BARRIER wait, (6) the epilogue; [7] = chunk records
// passed, [8] = of which this slot had work. Summed over all waves of a launch with one atomic per wave and category.
#ifdef GMRFX_CYC
__device__ unsigned long long g_ch_cyc[2][4][10];
#define CY_DECL long long cy_t_ = clock64(), cy_n_; unsigned long long cy_a_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define CY_MARK(k) do { __builtin_amdgcn_sched_barrier(0); cy_n_ = clock64(); cy_a_[k] += (unsigned long long)(cy_n_ - cy_t_); cy_t_ = cy_n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define CY_COUNT(k, v) do { cy_a_[k] += (unsigned long long)(v); } while (0)
#define CY_FLUSH(kern, role) do { if ((threadIdx.x & 63) == 0) { for (int q_ = 0; q_ < 10; q_++) atomicAdd(&g_ch_cyc[kern][role][q_], cy_a_[q_]); } } while (0)
// wait until all but the youngest k vector-memory operations of this wave have returned
__device__ __forceinline__ void cy_wait_vm(int k) {
    switch (k) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
#define CY_WAIT_VM(k) cy_wait_vm(k)
#else
#define CY_DECL
#define CY_MARK(k)
#define CY_COUNT(k, v)
#define CY_FLUSH(kern, role)
#define CY_WAIT_VM(k)
#endif
// ---- knock-out variants (`make var` -> libgmrfx_var.so, -DGMRFX_VAR; tools/chunk_variants.py) -------------------------------
// The same kernels with one part switched off by a run-time flag word (WRONG results, timing only): bit 0 = no operand requests
// beyond a task's first chunk (what is left is the LDS / MFMA / barrier chain), bit 1 = no arithmetic (requests and barriers only:
// what this access pattern streams at), bit 2 = no barriers, bit 4 = histogram of the SIMD every row-tile slot's wave runs on
// (HW_ID; one atomic per wave). The flag word is a kernel-uniform scalar; the product build has none of this.
#ifdef GMRFX_VAR
__device__ int g_ch_var;
__device__ unsigned long long g_ch_simd[2][4][4];
#define VAR_DECL const int var_ = __builtin_amdgcn_readfirstlane(g_ch_var)
#define VAR_ON(bit) ((var_ >> (bit)) & 1)
#define VAR_SIMD(kern, role) do { if (VAR_ON(4) && (threadIdx.x & 63) == 0) { unsigned hw_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_)); \
                                  atomicAdd(&g_ch_simd[kern][role][(hw_ >> 4) & 3], 1ull); } } while (0)
#else
#define VAR_DECL
#define VAR_ON(bit) 0
#define VAR_SIMD(kern, role)
#endif

// Records into LDS, the task's slice of X into the local vector. (Round 2's warm-up of the task's panels into L2 is gone: with
// every operand requested a chunk ahead it cost 30-40 us per sweep.)
// Round 6: ALL loads of the prologue are in flight at once. The knock-out timings (profiles/r06_chunk_variants_start.txt) put
// prologue + epilogue at 0.22 of the forward task kernel's 0.62 ms and at 0.36 of the backward one's 0.63: the local vector was
// filled 64 rows per loop iteration, each iteration an exposed memory round trip (up to five for the own rows, and in the backward
// kernel up to five more PAIRS -- row index, then x -- for the root's trailing rows) in a workgroup that lives ~22 us. Now: the
// index loads of the trailing rows first, the own rows' loads behind them, the gathered rows as soon as the indices are back, the
// LDS stores last; iterations beyond the task's rows are skipped by a uniform branch. ROOT = the backward kernel (local rows
// NT .. NT + mroot - 1 = x of the root's trailing rows, rows[] = their positions); the forward kernel zeroes them itself.
template <int NC, int NTHR, bool ROOT> __device__ __forceinline__ void chunk_prologue(const SweepTask &T, const Chunk *__restrict__ recs, int nrec, Chunk *meta,
                                                                                         const int *__restrict__ rows, const double *__restrict__ X,
                                                                                         double *V, int nr, int ldx) {
    const int tid = threadIdx.x;
    constexpr int G = NTHR / NC;                       // row groups: a pass of the loops below covers 4 G rows
    constexpr int NIT = (CH_ROWS - 1 + 4 * G - 1) / (4 * G);
    const int j = tid % NC, g = tid / NC;
    const int jc = min(j, nr - 1);
    const double jm = j < nr ? 1.0 : 0.0;
    const int NT = T.nt, col0 = T.col0, mroot = ROOT ? T.mroot : 0;
    int ri[NIT][4];
    double v[NIT][4], vr[NIT][4];
    if (ROOT) {
#pragma unroll
        for (int it = 0; it < NIT; it++)
            if (it * 4 * G < mroot) {
#pragma unroll
                for (int u = 0; u < 4; u++) ri[it][u] = rows[min(g + (4 * it + u) * G, mroot - 1)];
            }
    }
    int4 mrec[(2 * CH_MAXC + NTHR - 1) / NTHR];
#pragma unroll
    for (int q = 0; q < (2 * CH_MAXC + NTHR - 1) / NTHR; q++) mrec[q] = ((const int4 *)recs)[min(tid + q * NTHR, 2 * nrec - 1)];
#pragma unroll
    for (int it = 0; it < NIT; it++)
        if (it * 4 * G < NT) {
#pragma unroll
            for (int u = 0; u < 4; u++) v[it][u] = X[(long long)(col0 + min(g + (4 * it + u) * G, NT - 1)) * ldx + jc];
        }
    if (ROOT) {
#pragma unroll
        for (int it = 0; it < NIT; it++)
            if (it * 4 * G < mroot) {
#pragma unroll
                for (int u = 0; u < 4; u++) vr[it][u] = X[(long long)ri[it][u] * ldx + jc];
            }
    }
#pragma unroll
    for (int q = 0; q < (2 * CH_MAXC + NTHR - 1) / NTHR; q++) if (tid + q * NTHR < 2 * nrec) ((int4 *)meta)[tid + q * NTHR] = mrec[q];
#pragma unroll
    for (int it = 0; it < NIT; it++)
        if (it * 4 * G < NT) {
#pragma unroll
            for (int u = 0; u < 4; u++) { const int i = g + (4 * it + u) * G; if (i < NT) V[vbyte<NC>(i, j) >> 3] = v[it][u] * jm; }
        }
    if (ROOT) {
#pragma unroll
        for (int it = 0; it < NIT; it++)
            if (it * 4 * G < mroot) {
#pragma unroll
                for (int u = 0; u < 4; u++) { const int i = g + (4 * it + u) * G; if (i < mroot) V[vbyte<NC>(NT + i, j) >> 3] = vr[it][u] * jm; }
            }
    }
}

// ------------------------------------------------------------------------------------------------------------
// forward: V <- [b of the subtree ; 0]; per chunk y = D^-1 b (own rows, written to X), V[targets] -= L[targets, chunk] y
// ------------------------------------------------------------------------------------------------------------
// what a slot needs of a chunk from HBM: the packed inverse diagonal block, its pair of target tiles (pair w) with their rows
// (a chunk has at most 128 target rows = four pairs, one per slot: symbolic.cpp)
struct FBuf { double dt[4]; gmrfx_d2u a[4]; i4v l[2]; };

template <int NC, int TPW> __global__ __launch_bounds__(4 * (NC / 16 / TPW) * 64, (TPW == 1 && NC <= 32) ? 4 : 2)
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
    CY_DECL;
    VAR_DECL;
    chunk_prologue<NC, NTHR, false>(T, recs + T.c0, nch, meta, nullptr, X, V, nr, ldx);
    for (int i = NT * NC + tid; i < CH_ROWS * NC; i += NTHR) V[i] = 0.0;
    __syncthreads();
    CY_MARK(0);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wv / CT, th = wv % CT;                // row-tile slot, column group (scalars)
    const int lane = tid & 63, lm = lane & 15, lk = lane >> 4;
    int clb[TPW];
#pragma unroll
    for (int t = 0; t < TPW; t++) clb[t] = ((th * TPW + t) * 16 + lm) * 8;
    char *Vb = (char *)V;
    FBuf A, B;
    UChunk mA, mB;       // the records travel with the buffers: read from LDS once per chunk
    auto request = [&](int f, FBuf &x, UChunk &m) -> int {        // (returns the vector loads issued: used by the cycle stamps only)
        m = uniform_chunk(meta, f);
        const int npair = (m.nt + 31) >> 5;
        if (!(w == 0 || w < npair)) return 0;
        const double *dp = dtile + (long long)m.id * 256;
        gl_f64<0>(x.dt[0], dp, lane * 8); gl_f64<512>(x.dt[1], dp, lane * 8); gl_f64<1024>(x.dt[2], dp, lane * 8); gl_f64<1536>(x.dt[3], dp, lane * 8);
        if (w >= npair) return 4;
        const double *base = L + m.pa + 32 * w;
        const unsigned vo = (unsigned)(2 * lm + lk * m.ld) * 8u;
        const long long st = 4LL * m.ld;
        gl_d2<0>(x.a[0], base, vo); gl_d2<0>(x.a[1], base + st, vo); gl_d2<0>(x.a[2], base + 2 * st, vo); gl_d2<0>(x.a[3], base + 3 * st, vo);
        const int *lp = listf + m.lr + 32 * w;
        gl_i4<0>(x.l[0], lp, lk * 32);
        gl_i4<16>(x.l[1], lp, lk * 32);
        return 10;
    };
    request(0, A, mA);
#ifdef GMRFX_VAR
    if (VAR_ON(0)) B = A;          // (no later requests: the second buffer keeps valid row lists)
#endif
    auto chunk = [&](const int f, FBuf &cur, const UChunk &m, FBuf &nxt, UChunk &mn) {
        [[maybe_unused]] int issued = 0;
        if (f + 1 < nch) { if (!VAR_ON(0)) issued = request(f + 1, nxt, mn); else mn = uniform_chunk(meta, f + 1); }
        CY_MARK(1);
        CY_WAIT_VM(issued);
        CY_MARK(2);
        CY_COUNT(7, 1);
        const int npair = (m.nt + 31) >> 5;
        if ((w == 0 || w < npair) && !VAR_ON(1)) {
            CY_COUNT(8, 1);
            const int ku = (m.cc + 3) >> 2;
            const int rowb = m.o + lk;
            const int bb = rowb * NC * 8, sw = NC >= 32 ? (rowb & 1) << 7 : 0;
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
            CY_MARK(3);
            // y straight to X (the local vector keeps b: the other slots may still be reading it for their own copy of y,
            // and a chunk with more than 128 target rows comes as several records that each recompute y)
            if (w == 0) {
                double *Xo = X + (long long)(col0 + m.o) * ldx;
#pragma unroll
                for (int t = 0; t < TPW; t++)
#pragma unroll
                    for (int rr = 0; rr < 4; rr++)
                        if (lk + 4 * rr < m.cc && (clb[t] >> 3) < nr) Xo[(long long)(lk + 4 * rr) * ldx + (clb[t] >> 3)] = y[t][rr];
            }
            // V[rows of a pair of target tiles] -= (pair's operand rows) y
            auto apply = [&](const gmrfx_d2u (&av)[4], const i4v &l0, const i4v &l1) {
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
            CY_MARK(4);
        }
        if (!VAR_ON(2)) __syncthreads();
        CY_MARK(5);
    };
    {
        int f = 0;
        for (; f + 1 < nch; f += 2) { chunk(f, A, mA, B, mB); chunk(f + 1, B, mB, A, mA); }
        if (f < nch) chunk(f, A, mA, B, mB);
    }
    // the root's update vector to W (y went to X chunk by chunk)
    {
        constexpr int G = NTHR / NC;
        const int j = tid % NC, g = tid / NC;
        if (j < nr) {
            double *Wr = W + T.woff * ldx;
            for (int i = g; i < mroot; i += G) Wr[(long long)i * ldx + j] = V[vbyte<NC>(NT + i, j) >> 3];
        }
    }
    CY_MARK(6);
    CY_FLUSH(0, w);
    VAR_SIMD(0, w);
}

// ------------------------------------------------------------------------------------------------------------
// backward: V <- [y of the subtree ; x of the root's trailing rows]; per chunk t = y - L[targets, chunk]' x[targets],
// x = D^-T t; the slot programs (chunk tree depth by depth)
// ------------------------------------------------------------------------------------------------------------
template <int G> struct BBuf { double dt[4]; gmrfx_d2u a[G][2]; i4v l[G]; };

template <int NC, int TPW, int G> __global__ __launch_bounds__(4 * (NC / 16 / TPW) * 64, (TPW == 1 && NC <= 32) ? 4 : 2)
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
    CY_DECL;
    VAR_DECL;
    // own rows (y of the subtree) and x of the root's trailing rows (ancestors of the subtree: final), zeros behind them
    chunk_prologue<NC, NTHR, true>(T, recs + T.b0, T.nbw, meta, S.rows + T.rroot, X, V, nr, ldx);
    for (int i = (NT + mroot) * NC + tid; i < CH_ROWS * NC; i += NTHR) V[i] = 0.0;
    __syncthreads();
    CY_MARK(0);
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
    UChunk mA, mB;
    auto request = [&](int f, BBuf<G> &x, UChunk &m) -> int {      // (returns the vector loads issued: used by the cycle stamps only)
        m = uniform_chunk(meta, f);
        const double *dp = dtile + (long long)m.id * 256;
        const unsigned vd = (unsigned)(16 * lm + lk) * 8u;
        gl_f64<0>(x.dt[0], dp, vd); gl_f64<32>(x.dt[1], dp, vd); gl_f64<64>(x.dt[2], dp, vd); gl_f64<96>(x.dt[3], dp, vd);
        const double *base = L + m.pa;
        const unsigned vo = (unsigned)(2 * lk + lm * m.ld) * 8u;
        const int *lp = listb + m.lr;
        const int nk = (m.nt + 15) >> 4;
#pragma unroll
        for (int g = 0; g < G; g++)
            if (g < nk) {
                gl_d2<0>(x.a[g][0], base + 16 * g, vo);
                gl_d2<64>(x.a[g][1], base + 16 * g, vo);
                gl_i4<0>(x.l[g], lp + 16 * g, lk * 16);
            }
        return 4 + 3 * (nk < G ? nk : G);
    };
    auto tile = [&](d4 (&acc)[TPW], const gmrfx_d2u &a0, const gmrfx_d2u &a1, const i4v &l) {
#pragma unroll
        for (int t = 0; t < TPW; t++) {
            const double b0 = lds_ld(Vb, l.x ^ clb[t]), b1 = lds_ld(Vb, l.y ^ clb[t]), b2 = lds_ld(Vb, l.z ^ clb[t]), b3 = lds_ld(Vb, l.w ^ clb[t]);
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, b0, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, b1, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b2, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, b3, acc[t], 0, 0, 0);
        }
    };
    auto chunk = [&](const int f, const bool last, BBuf<G> &cur, const UChunk &m, BBuf<G> &nxt, UChunk &mn) {
        [[maybe_unused]] int issued = 0;
        if (!last) { if (!VAR_ON(0)) issued = request(f + 1, nxt, mn); else mn = uniform_chunk(meta, f + 1); }
        CY_MARK(1);
        if (!VAR_ON(2)) for (int b = 0; b < m.nbar; b++) __syncthreads();
        CY_MARK(5);
        CY_WAIT_VM(issued);
        CY_MARK(2);
        CY_COUNT(7, 1);
        CY_COUNT(8, 1);
        const int nk = VAR_ON(1) ? 0 : (m.nt + 15) >> 4;
        d4 acc[TPW];
#pragma unroll
        for (int t = 0; t < TPW; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
        {   // the first G k-tiles came with the request; k-tile i + G is requested into the registers k-tile i leaves
            const double *base = L + m.pa;
            const unsigned vo = (unsigned)(2 * lk + lm * m.ld) * 8u;
            const int *lp = listb + m.lr;
#pragma unroll 1
            for (int g0 = 0; g0 < nk; g0 += G) {
#pragma unroll
                for (int j = 0; j < G; j++) {
                    const int i = g0 + j;
                    if (i < nk) {
                        tile(acc, cur.a[j][0], cur.a[j][1], cur.l[j]);
                        const int g = i + G;
                        if (g < nk) {
                            gl_d2<0>(cur.a[j][0], base + 16 * g, vo);
                            gl_d2<64>(cur.a[j][1], base + 16 * g, vo);
                            gl_i4<0>(cur.l[j], lp + 16 * g, lk * 16);
                        }
                    }
                }
            }
        }
        CY_MARK(3);
        const int ku = VAR_ON(1) ? 0 : (m.cc + 3) >> 2;
        const int rowb = m.o + lk;
        const int bb = rowb * NC * 8, sw = NC >= 32 ? (rowb & 1) << 7 : 0;
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
                if (lk + 4 * rr < m.cc && !VAR_ON(1)) lds_st(Vb, bb + rr * 4 * NC * 8 + (clb[t] ^ sw), x[t][rr]);
        CY_MARK(4);
    };
    if (cnt > 0) request(i0, A, mA);
#ifdef GMRFX_VAR
    if (VAR_ON(0)) B = A;
#endif
    {
        int k = 0;
        for (; k + 1 < cnt; k += 2) { chunk(i0 + k, false, A, mA, B, mB); chunk(i0 + k + 1, k + 2 >= cnt, B, mB, A, mA); }
        if (k < cnt) chunk(i0 + k, true, A, mA, B, mB);
    }
    if (!VAR_ON(2)) for (int b = 0; b < endbar; b++) __syncthreads();
    CY_MARK(5);
    {
        constexpr int GR = NTHR / NC;
        const int j = tid % NC, g = tid / NC;
        if (j < nr)
            for (int i = g; i < NT; i += GR) X[(long long)(col0 + i) * ldx + j] = V[vbyte<NC>(i, j) >> 3];
    }
    CY_MARK(6);
    CY_FLUSH(1, w);
    VAR_SIMD(1, w);
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

// 16 columns of the right-hand sides per workgroup, one 16-column tile per wave: 4 waves, four workgroups per compute unit.
// (Measured at cfg 2, round 5: 32 columns / 8 waves / two workgroups per CU 0.61 / 0.67 ms forward / backward against 0.56 / 0.64;
//  two tiles per wave, 32 or 64 columns: 0.74-0.75 / 0.86-0.87 ms.)
int sweep_chunk_nc() { return 16; }

#ifdef GMRFX_CYC
// tools/chunk_cycles.py: [kernel 0 = forward, 1 = backward][slot 0..3][category 0..9] (see the CY_ macros above); reset != 0 zeroes the counters
extern "C" int gmrfx_debug_chunk_cycles(unsigned long long *out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ch_cyc), sizeof(g_ch_cyc)) != hipSuccess) return 1;
    if (reset) {
        static const unsigned long long zero[2][4][10] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_ch_cyc), zero, sizeof(zero)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

#ifdef GMRFX_VAR
// tools/chunk_variants.py: set the knock-out flag word; read (and clear) the SIMD histogram [kernel][slot][simd]
extern "C" int gmrfx_debug_chunk_variant(int flags, unsigned long long *simd_out) {
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ch_var), &flags, sizeof(int)) != hipSuccess) return 1;
    if (simd_out) {
        static const unsigned long long zero[2][4][4] = {};
        if (hipMemcpyFromSymbol(simd_out, HIP_SYMBOL(g_ch_simd), sizeof(g_ch_simd)) != hipSuccess) return 1;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_ch_simd), zero, sizeof(zero)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

void launch_pack_diag(hipStream_t st, const Symbolic::SwChunk *recs, int nchunks, const double *L, double *dtile) {
    if (nchunks <= 0) return;
    hipLaunchKernelGGL(k_pack_diag, dim3((nchunks + 3) / 4), dim3(256), 0, st, recs, nchunks, L, dtile);
}

void launch_sweep_chunks(hipStream_t st, const DevSym &S, int phase, const SweepTask *tasks, int ntasks, const Symbolic::SwChunk *recs_fwd,
                         const Symbolic::SwChunk *recs_bwd, const int *listf, const int *listb, const double *dtile, const double *L,
                         double *X, double *W, int nr, int ldx, size_t extra_lds) {
    if (ntasks <= 0) return;
    const int grid = ((ntasks + 7) / 8) * 32;       // blocks b, b + 8, b + 16, b + 24 (same XCD): the four column slices of one task
    if (phase == 1) hipLaunchKernelGGL((k_fwd_chunks<16, 1>), dim3(grid), dim3(256), extra_lds, st, tasks, ntasks, recs_fwd, listf, dtile, L, X, W, nr, ldx);
    else hipLaunchKernelGGL((k_bwd_chunks<16, 1, 3>), dim3(grid), dim3(256), 0, st, S, tasks, ntasks, recs_bwd, listb, dtile, L, X, nr, ldx);
}
}  // namespace gmrfx
