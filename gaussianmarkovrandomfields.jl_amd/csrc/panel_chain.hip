// panel_chain.hip -- the panel factorisation of ONE 256-column outer block of every wide front of a level as ONE persistent
// kernel (round 4).
//
// The top of the tree was a chain of small dependent launches: per 64-column block potrf64 (one workgroup per front) -> trsm ->
// gemm (K = 64, inside the 256-column outer block), 12 launches per outer block, ~350 per factorisation at cfg 2, each at its
// launch-to-launch floor of 6-7 us and -- measured in round 3 and again in round 4 (tools/interference.py, tools/host_io_step.py) --
// slowed by anything that is active in another queue: a chain of dependent dispatches loses the command processor to the other
// queue after every packet. Here the chain needs NO dispatch:
//   * a front's rows below the outer block's first column are cut into 64-row TILES, dealt cyclically to the workgroups of the
//     front's team; a workgroup owns its tiles for the whole launch and is the only writer of their panel entries (no atomics,
//     fixed order: the same bits as the launch chain, tests/test_gpu_parity.py::test_panel_chain_*);
//   * step q (block b = J0 + q): the owner of tile q factors the diagonal block (potrf64_body, write-through stores) and
//     raises the front's P flag; everybody waits for it, stages the inverse block in LDS and, for each of its tiles t > q
//     (ascending): T  L(t, b) = A(t, b) Linv' (kept in LDS too), publishes it when other tiles need it as an operand (the
//     <= 3 tiles of the outer block's own diagonal range), and G  A(t, j) -= L(t, b) L(j, b)' for the column blocks
//     j = b + 1 .. end of the outer block, j <= t;
//   * LOOK-AHEAD for free: the owner of tile q + 1 factors the NEXT diagonal block right after its own T and G on that tile --
//     before its other tiles -- so the chain per block is potrf + one flag hop + one 64 x 64 T + one 64 x 64 G, and the rest of
//     the step runs beside the next potrf on the other workgroups.
// The K = 256 update of the columns beyond the outer block stays the big throughput launch it was (k_gemm_nt<2>): one persistent
// launch + one update launch per outer block instead of 12.
// Hand-offs follow the placement-independent protocol of the CDNA4 guide (cdna_hip_programming.md, Guideline 16): payload
// written through (agent-scope relaxed atomic stores = global_store ... sc1), every storing wave `s_waitcnt vmcnt(0)`, workgroup
// barrier, ONE flag store; consumers poll relaxed with s_sleep, then ONE agent-scope acquire, `s_waitcnt vmcnt(0)`, barrier,
// plain loads. Flags are monotone (a per-launch base, no resets). Every spin is bounded: a timeout raises an error word the
// host turns into an exception -- the grid must be fully resident (<= 256 workgroups of 256 threads, 68 KB of LDS, ~490 registers).
// Replaces, for one dense front panel, cholesky!(F, Q) of the reference's backend (src/workspace/backend.jl:165-189).
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "potrf64_body.h"

namespace gmrfx {

typedef gmrfx_d4 d4;

namespace {

constexpr int CHAIN_THREADS = 256;        // four waves: the inlined diagonal-block body alone needs ~150 VGPRs (eight waves: 256 per lane, spills)
constexpr int CHAIN_SUB = 16 / (CHAIN_THREADS / 64);   // 16 x 16 sub-tiles per wave in one row group of a 64 x 64 tile
constexpr int CHAIN_SPIN_MAX = 1 << 18;        // x ~0.5 us per poll: ~0.1 s, then the error word (which ends every later wait at once)

__device__ __forceinline__ void chain_publish(int *flag, int value) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void chain_wait(const int *flag, int value, int *err) {
    if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - value < 0) {
            __builtin_amdgcn_s_sleep(4);
            ++spins;
            if ((spins & 255) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            if (spins > CHAIN_SPIN_MAX) { atomicExch(err, 1); break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}
__device__ __forceinline__ void store_wt(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

}  // namespace

// LDS of the kernel, at namespace scope: the tile work lives in a function of its own (below), which the compiler must not
// inline -- inlined next to the register-resident diagonal-block factorisation (149 VGPRs) the kernel spilled ~100 doubles per lane
__shared__ __attribute__((aligned(16))) double c_Ti[NB * NB];      // inverse of the current diagonal block (full lower, reciprocal diagonal)
__shared__ __attribute__((aligned(16))) double c_Ls[NB * NB];      // L(t, b) of the tile being worked on: Ls[k * 64 + row]
__shared__ __attribute__((aligned(16))) double c_Sb[2][4 * 64];    // potrf64_body's strips

struct ChainArgs {
    const FrontView *frec;   // the level's big fronts, widest first
    int nfront;              // fronts with more than 64 J0 columns
    int J0;                  // first 64-column block of this outer block
    int stride_cap;          // tiles per workgroup: team size of a front = ceil(tiles / stride_cap)
    int base;                // flag values of this launch are base + 1 .. base + 4
    int *flags;              // 8 ints per front: [0] diagonal block factored, [1 + t] tile t of the diagonal range solved
    int *err;
    double *L;
    int *info;
    long long *trace;        // GMRFX_CHAIN_TRACE (measurement aid): s_memtime stamps of workgroups 0 .. 7, or null
};

// Step q of one workgroup, pass 0 = the tile q + 1 alone (the look-ahead tile), pass 1 = its other tiles above q: T then G per tile.
__device__ __forceinline__ void chain_tiles(double *__restrict__ P, const int ld, const int c, const int r, const int row00,
                                                      const int nt, const int nbi, const int q, const int t0, const int team, const int pass,
                                                      const int kb, const int w, int *fl, const int base, int *err, long long *tr) {
    double *Ti = c_Ti, *Ls = c_Ls;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, lm = lane & 15, lk = lane >> 4;
    const bool has_next = q + 1 < nbi;
    struct { int base; int *err; } a{base, err};
                for (int t = t0; t < nt; t += team) {
                const bool critical = has_next && t == q + 1;
                if (critical != (pass == 0)) continue;
                const int row0 = row00 + NB * t;                  // first row of the tile (front-local)
                // -- T: L(t, b) = A(t, b) Linv'. wave v: row group v & 3 (16 rows), column tiles CHAIN_SUB (v >> 2) .. + CHAIN_SUB - 1
                const int ri = wave & 3, ctp = CHAIN_THREADS == 256 ? 0 : wave >> 2;      // (four waves: ct = e is a compile-time constant below)
                const int i = row0 + 16 * ri + lm;
                const double *pa = P + min(i, r - 1) + (long long)kb * ld;
                double bv[16];
#pragma unroll
                for (int u = 0; u < 16; u++) bv[u] = pa[(long long)min(4 * u + lk, w - 1) * ld];
                d4 acc[CHAIN_SUB];
#pragma unroll
                for (int e = 0; e < CHAIN_SUB; e++) {
                    const int ct = CHAIN_SUB * ctp + e;
                    acc[e] = (d4){0.0, 0.0, 0.0, 0.0};
                    const int k = ct * 16 + lm;
#pragma unroll
                    for (int u = 0; u < 16; u++) {
                        const double av = Ti[k * NB + 4 * u + lk];
                        if (u < 4 * ct + 4 && 4 * u < w) acc[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[u], acc[e], 0, 0, 0);
                    }
                }
                __syncthreads();      // in place: the other waves have read the columns this wave overwrites; Ls is free
                const bool shared_tile = has_next && t < nbi;     // an operand of other tiles' G: written through, then published
#pragma unroll
                for (int e = 0; e < CHAIN_SUB; e++)
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) {
                        const int kk = (CHAIN_SUB * ctp + e) * 16 + lk + 4 * rr;
                        Ls[kk * NB + 16 * ri + lm] = acc[e][rr];
                        if (kk < w && i < r && i >= kb + w) {
                            double *dst = P + i + (long long)(kb + kk) * ld;
                            if (shared_tile) store_wt(dst, acc[e][rr]);
                            else *dst = acc[e][rr];
                        }
                    }
                if (shared_tile) chain_publish(fl + 1 + t, a.base + q + 1);
                else __syncthreads();
                // -- G: A(t, j) -= L(t, b) L(j, b)' for the column blocks j = q + 1 .. min(t, nbi - 1) of this outer block
                if (has_next) {
                    const int jhi = min(t, nbi - 1);
                    for (int jj = q + 1; jj <= jhi; jj++) {
                        const bool diag = jj == t;
                        if (!diag) chain_wait(fl + 1 + jj, a.base + q + 1, a.err);
                        const int cj0 = row00 + NB * jj;          // first column of the block = first row of tile jj
                        const int ncol = min(NB, c - cj0);
                        // 16 sub-tiles (si: rows, sj: columns); wave v: si = v & 3, sj = CHAIN_SUB (v >> 2) + e
                        const int si = wave & 3;
#pragma unroll
                        for (int e = 0; e < CHAIN_SUB; e++) {
                            const int sj = CHAIN_SUB * (CHAIN_THREADS == 256 ? 0 : wave >> 2) + e;
                            if (diag && sj > si) continue;        // strictly upper sub-tiles of a diagonal tile (wave-uniform)
                            const int ii = row0 + 16 * si + lm;   // row of C this lane holds
                            double cv[4];
#pragma unroll
                            for (int rr = 0; rr < 4; rr++) {
                                const int j = min(16 * sj + lk + 4 * rr, ncol - 1);
                                cv[rr] = P[min(ii, r - 1) + (long long)(cj0 + j) * ld];
                            }
                            double xa[16], xb[16];
                            const double *pb = P + min(cj0 + min(16 * sj + lm, ncol - 1), r - 1) + (long long)kb * ld;
#pragma unroll
                            for (int u = 0; u < 16; u++) {
                                xa[u] = Ls[(4 * u + lk) * NB + 16 * si + lm];
                                xb[u] = diag ? Ls[(4 * u + lk) * NB + 16 * sj + lm] : pb[(long long)(4 * u + lk) * ld];
                            }
                            d4 g = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                            for (int u = 0; u < 16; u++) g = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[u], xa[u], g, 0, 0, 0);
#pragma unroll
                            for (int rr = 0; rr < 4; rr++) {
                                const int j = 16 * sj + lk + 4 * rr;
                                if (ii < r && j < ncol && ii >= cj0 + j) P[ii + (long long)(cj0 + j) * ld] = cv[rr] - g[rr];
                            }
                        }
                    }
                }
                __syncthreads();
            }
}

__global__ __launch_bounds__(CHAIN_THREADS) void k_panel_chain(ChainArgs a) {
    double *Ti = c_Ti;
    double (*Sb)[4 * 64] = c_Sb;
    __shared__ int s_front, s_local, s_team;
    const int tid = threadIdx.x;
    if (tid == 0) {
        int acc = 0, f = 0, g = 1;
        for (; f < a.nfront; f++) {
            const int4 *q = reinterpret_cast<const int4 *>(a.frec + f);
            const int4 v = q[0];                                    // s, c, r, ld
            const int rows = v.z - NB * a.J0;
            const int nt = v.y > NB * a.J0 ? (rows + NB - 1) / NB : 0;
            g = (nt + a.stride_cap - 1) / a.stride_cap;
            if ((int)blockIdx.x < acc + g) break;
            acc += g;
        }
        s_front = f; s_local = (int)blockIdx.x - acc; s_team = g;
    }
    __syncthreads();
    if (s_front >= a.nfront) return;
    const FrontArg none{0, 0, 0, 0, 0, 0, 0};
    const FrontView fv = front_view(a.frec, s_front, none);
    const int c = fv.c, r = fv.r, ld = fv.ld;
    double *P = a.L + fv.pp;
    const int local = s_local, team = s_team;
    int *fl = a.flags + 8 * s_front;
    const int row00 = NB * a.J0;
    const int nt = (r - row00 + NB - 1) / NB;                        // tiles of this front (tile t = rows row00 + 64 t ..)
    const int nbi = min(4, (c - row00 + NB - 1) / NB);               // inner blocks of this outer block
    const int wave = tid >> 6, lane = tid & 63, lm = lane & 15, lk = lane >> 4;
#define CHAIN_TR(slot) do { if (a.trace && tid == 0 && blockIdx.x < 8 && q < 6) a.trace[((int)blockIdx.x * 8 + (q + 1)) * 8 + (slot)] = clock64(); } while (0)
    // Pseudo-step q = -1 only factors block 0; step q >= 0: wait for block q, own tiles above it -- the tile q + 1 first, then
    // (ONE call site of the factorisation) the diagonal block q + 1, then the other tiles.
    for (int q = -1; q < nbi; q++) {
        const int kb = row00 + NB * max(q, 0);
        const int w = min(NB, c - kb);
        const bool has_next = q + 1 < nbi;                            // (then block q is a full one: w == 64)
        // first own tile with rows below the diagonal block: above q -- or q itself when the block is a partial one (the last of
        // the front, w < 64): rows kb + w .. kb + 63 then lie in the diagonal block's own tile
        const int tmin = w < NB ? q : q + 1;
        int t0 = local;
        while (t0 < tmin) t0 += team;
        const bool work = q >= 0 && t0 < nt;
        CHAIN_TR(0);
        if (work) {
            chain_wait(fl + 0, a.base + q + 1, a.err);
            CHAIN_TR(1);
            if (tid < 256) stage_linv(P + kb + (long long)kb * ld, ld, w, Ti, tid);
            __syncthreads();
            CHAIN_TR(2);
        }
        // two passes over the own tiles: pass 0 = the tile q + 1 alone (if it is ours), then the next diagonal block; pass 1 = the rest
        for (int pass = 0; pass < 2; pass++) {
            if (work) chain_tiles(P, ld, c, r, row00, nt, nbi, q, t0, team, pass, kb, w, fl, a.base, a.err,
                                  (a.trace && blockIdx.x < 8 && q < 6) ? a.trace + ((int)blockIdx.x * 8 + (q + 1)) * 8 : nullptr);
            CHAIN_TR(3 + 2 * pass);
            if (pass == 0 && has_next && (q + 1) % team == local) {
                // ---- the diagonal block q + 1: its tile is up to date (this workgroup's own updates, just stored; the body reads
                //      past the L1) -- everything else of step q runs beside it on the other workgroups
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                const int kb1 = row00 + NB * (q + 1), w1 = min(NB, c - kb1), np1 = (w1 + 3) >> 2;
                double *Pd = P + kb1 + (long long)kb1 * ld;
                if (CHAIN_THREADS == 256 || tid < 256) potrf64_body<true>(Pd, ld, Pd, ld, w1, Sb, a.info, fv.first + kb1, tid);
                else for (int k = 0; k < np1 + 1; k++) __syncthreads();  // (the body's barriers: one after the first strip, one per step)
                chain_publish(fl + 0, a.base + q + 2);
                CHAIN_TR(4);
            }
        }
    }
}

void launch_panel_chain(hipStream_t st, const FrontView *frec, int nfront, int J0, int stride_cap, int nwg, int base, int *flags, int *err,
                        double *L, int *info, long long *trace) {
    if (nfront <= 0 || nwg <= 0) return;
    ChainArgs a{frec, nfront, J0, stride_cap, base, flags, err, L, info, trace};
    hipLaunchKernelGGL(k_panel_chain, dim3(nwg), dim3(CHAIN_THREADS), 0, st, a);
}

}  // namespace gmrfx
