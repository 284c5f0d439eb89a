// potrf64_body.h -- the 64 x 64 diagonal-block factorisation (Cholesky factor + inverse) as a device function of 256 threads,
// shared by k_potrf64 / k_potrf64_la (potrf64.hip) and the persistent panel-chain kernel (panel_chain.hip). See potrf64.hip
// for the design notes.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace gmrfx {

namespace {

// phase-cycle instrumentation for tools/micro/potrf_prof.hip (compiled out of the library)
#ifdef GMRFX_CYC
__device__ long long g_cyc64[4][16];
#define C64_DECL long long cyc_t = clock64(); const int cyc_w = (threadIdx.x & 63) == 0 ? (int)(threadIdx.x >> 6) : -1
#define C64_MARK(k) do { __builtin_amdgcn_sched_barrier(0); long long t_ = clock64(); if (cyc_w >= 0) g_cyc64[cyc_w][k] += t_ - cyc_t; cyc_t = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define C64_DECL
#define C64_MARK(k)
#endif

__device__ __forceinline__ double rsqrt_nr(double p) {
    // v_rsq_f64 + two Newton steps: full double precision without sqrt + division
    double y = __builtin_amdgcn_rsq(p);
    y = y * (1.5 - 0.5 * p * y * y);
    y = y * (1.5 - 0.5 * p * y * y);
    return y;
}

}  // namespace

// The 256 threads (tid 0..255) of a workgroup factor the w x w block whose current values are src[i + j * sld] (the panel
// itself, or the block a look-ahead prologue left in LDS) and write L / (L^-1)' to the panel block P (leading dimension ld).
// Sb: 2 x 4 x 64 doubles of LDS.
// WT = true: the block is read and the results leave with agent-scope relaxed atomic accesses (global_load / global_store ... sc1:
// past the L1, written through the XCD's L2), so that a workgroup of ANOTHER XCD can read them behind a flag without this
// workgroup paying an L2 write-back (release fence)
template <bool WT = false>
__device__ __forceinline__ void potrf64_body(const double *src, const int sld, double *__restrict__ P, const int ld, const int w,
                                             double (*Sb)[4 * 64], int *__restrict__ info, const int first_col, const int tid) {
    const int ty = tid & 15, tx = tid >> 4;     // lanes walk rows: coalesced panel loads / stores
    const int i0 = 4 * ty, j0 = 4 * tx;

    // patch a[r][cc] = element (i0 + r, j0 + cc); identity padding beyond w; M part starts at 0
    double a[4][4];
#pragma unroll
    for (int cc = 0; cc < 4; cc++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = i0 + r, j = j0 + cc;
            // (WT: read past this CU's L1 as well -- in the persistent chain the same workgroup has just rewritten these lines)
            const double v = WT ? __hip_atomic_load(src + min(i, w - 1) + min(j, w - 1) * sld, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                : src[min(i, w - 1) + min(j, w - 1) * sld];
            const double mk = (i < w && j < w && i >= j) ? 1.0 : 0.0;
            a[r][cc] = v * mk + ((i == j && i >= w) ? 1.0 : 0.0);
        }
    if (tx == 0) {
#pragma unroll
        for (int cc = 0; cc < 4; cc++)
#pragma unroll
            for (int r = 0; r < 4; r++) Sb[0][cc * 64 + i0 + r] = a[r][cc];
    }
    __syncthreads();

    const int np = (w + 3) >> 2;
    int badcol = 0x7fffffff;     // first non-positive pivot seen by this thread (diagonal patches only)
    C64_DECL;
    for (int p = 0; p < np; p++) {
        C64_MARK(0);
        const double *Sp = Sb[p & 1];
        if (tx >= p) {
            // diagonal 4 x 4 block of the strip -> Lpp (l..) and the reciprocals of its diagonal
            double d[4][4];
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int r = 0; r < 4; r++) d[r][q] = Sp[q * 64 + 4 * p + r];
            double sy[4][4], sx[4][4];
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    sy[r][q] = Sp[q * 64 + i0 + r];
                    sx[r][q] = Sp[q * 64 + j0 + r];
                }
            if (ty == p) {
                // rows of X that become final now start from the identity (L X = I)
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int r = 0; r < 4; r++) sy[r][q] = (r == q) ? 1.0 : 0.0;
            }
#ifdef GMRFX_CYC
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            C64_MARK(1);
            const double p0 = d[0][0];
            const double i00 = rsqrt_nr(p0);
            const double l10 = d[1][0] * i00, l20 = d[2][0] * i00, l30 = d[3][0] * i00;
            const double p1 = d[1][1] - l10 * l10;
            const double i11 = rsqrt_nr(p1);
            const double l21 = (d[2][1] - l20 * l10) * i11, l31 = (d[3][1] - l30 * l10) * i11;
            const double p2 = d[2][2] - l20 * l20 - l21 * l21;
            const double i22 = rsqrt_nr(p2);
            const double l32 = (d[3][2] - l30 * l20 - l31 * l21) * i22;
            const double p3 = d[3][3] - l30 * l30 - l31 * l31 - l32 * l32;
            const double i33 = rsqrt_nr(p3);
#ifdef GMRFX_CYC
            asm volatile("" ::"v"(i33));
#endif
            C64_MARK(2);
            // Y = S Lpp^-T, row by row
            double yy[4][4], yx[4][4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                yy[r][0] = sy[r][0] * i00;
                yy[r][1] = (sy[r][1] - yy[r][0] * l10) * i11;
                yy[r][2] = (sy[r][2] - yy[r][0] * l20 - yy[r][1] * l21) * i22;
                yy[r][3] = (sy[r][3] - yy[r][0] * l30 - yy[r][1] * l31 - yy[r][2] * l32) * i33;
                yx[r][0] = sx[r][0] * i00;
                yx[r][1] = (sx[r][1] - yx[r][0] * l10) * i11;
                yx[r][2] = (sx[r][2] - yx[r][0] * l20 - yx[r][1] * l21) * i22;
                yx[r][3] = (sx[r][3] - yx[r][0] * l30 - yx[r][1] * l31 - yx[r][2] * l32) * i33;
            }
#ifdef GMRFX_CYC
            asm volatile("" ::"v"(yy[3][3]), "v"(yx[3][3]));
#endif
            C64_MARK(3);
            if (tx == p) {
                // this thread column is final: L below the diagonal block, X' above it. The values stay in the
                // patch registers and go to the panel in ONE store pass after the loop (16 predicated stores per
                // step from a quarter of a wave cost 0.15 us of the 1.1 us step), the pivot check is collected in
                // a register and reported by one atomic at the end.
                if (ty == p) {
                    // the diagonal block itself: Lpp in the lower part (diag = pivot * rsqrt),
                    // Lpp^-1 transposed (= yy of the identity) in the strict upper part
                    const double pv[4] = {p0, p1, p2, p3};
#pragma unroll
                    for (int q = 3; q >= 0; q--)
                        badcol = (!(pv[q] > 0.0) && 4 * p + q < w) ? min(badcol, 4 * p + q) : badcol;
                    yy[0][0] = p0 * i00; yy[1][1] = p1 * i11; yy[2][2] = p2 * i22; yy[3][3] = p3 * i33;
                    yy[1][0] = l10; yy[2][0] = l20; yy[3][0] = l30;
                    yy[2][1] = l21; yy[3][1] = l31;
                    yy[3][2] = l32;
                }
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int r = 0; r < 4; r++) a[r][q] = yy[r][q];
            } else if (ty >= tx || ty <= p) {
                // trailing A patch, or M patch whose rows of X are already final (b <= 4p+3)
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int cc = 0; cc < 4; cc++) {
                        double acc = a[r][cc];
#pragma unroll
                        for (int q = 0; q < 4; q++) acc -= yy[r][q] * yx[cc][q];
                        a[r][cc] = acc;
                    }
                if (tx == p + 1) {
                    double *Sn = Sb[(p + 1) & 1];
#pragma unroll
                    for (int cc = 0; cc < 4; cc++)
#pragma unroll
                        for (int r = 0; r < 4; r++) Sn[cc * 64 + i0 + r] = a[r][cc];
                }
            }
        }
        C64_MARK(4);
        __syncthreads();
        C64_MARK(5);
    }
    if (badcol != 0x7fffffff) atomicMin(info, first_col + badcol);
    double *Pt = P + i0 + (long long)j0 * ld;
    if (w == NB) {
        // full block (all but the last block column of a front): no per-element predicates
#pragma unroll
        for (int cc = 0; cc < 4; cc++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if constexpr (WT) __hip_atomic_store(Pt + r + (long long)cc * ld, a[r][cc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else Pt[r + (long long)cc * ld] = a[r][cc];
            }
    } else {
#pragma unroll
        for (int cc = 0; cc < 4; cc++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (i0 + r < w && j0 + cc < w) {
                    if constexpr (WT) __hip_atomic_store(Pt + r + (long long)cc * ld, a[r][cc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else Pt[r + (long long)cc * ld] = a[r][cc];
                }
    }
}


}  // namespace gmrfx
