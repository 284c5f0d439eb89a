// kernels.h -- launch wrappers of the HIP kernels (kernels.hip, selinv.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "device.h"

namespace gmrfx {

constexpr int NB = 64;       // block-column width of the dense partial factorisation / sweeps
constexpr int ASM_CW = 4;    // front columns owned by one assembly workgroup
constexpr int FWD_RB = 32;    // front rows owned by one forward-assembly workgroup

// (FrontArg / FrontView -- the geometry of one front for the panel kernels -- live in device.h. The panel chain at the top
// of the tree, potrf -> trsm -> gemm, ~120 dependent launches on a single front, gets it in the kernel arguments; everything
// else reads one record at its position in the level list.)
// two doubles that are only known to be 8-byte aligned (one 16-byte load; the hardware takes unaligned addresses)
typedef double gmrfx_d2u __attribute__((ext_vector_type(2), aligned(8)));

void launch_gather_values(hipStream_t st, const double *nzval, const int *qsrc, double *out, long long cnt);
void launch_assemble(hipStream_t st, const DevSym &S, const int *list, const AsmRec *arec, const double *nzp, int nfronts, int max_cols, int max_rows,
                     const double *nzval, double *L, double *CB);
void launch_syrk_cb(hipStream_t st, const DevSym &S, const int *list, int nfronts, int max_trail, const double *L, double *CB);
// The same tiles from self-contained records, one contiguous run per XCD (workgroup id mod 8 = XCD): see k_syrk_cb_rec.
void launch_syrk_cb_recs(hipStream_t st, const DevSym &S, const SyrkTile *recs, const SyrkSplit &split, int per_xcd, const double *L, double *CB,
                         int noprod = 0, bool piped = false);
// piped: the product loop in three stages of two k-steps, each requested two stages ahead (levels whose widest front has at least
// kSyrkPipedMinCols columns, device.h; measured on cfg 2, round 6: k_syrk_cb_rec 3.60-3.69 -> 3.37-3.45 ms per step with 32 / 64 / 128 /
// 200 / 300 alike, 600: 3.56)
// CB -= L21 L21' on 128 x 128 LDS-staged tiles (behind a gather-only pass: noprod = 1), for levels of huge fronts
void launch_syrk_big(hipStream_t st, const DevSym &S, const int *list, int nfronts, int max_trail, const double *L, double *CB);
// blocks at most 32 columns wide, factorisation only (k_trsm<0, 0>'s arithmetic on half the registers)
void launch_trsm_narrow(hipStream_t st, const FrontView *frec, int nactive, int kb, int max_rows_below, double *L);
void launch_trsm(hipStream_t st, const DevSym &S, const FrontView *frec, int nactive, int kb, int mode, int max_rows_below,
                 double *L, double *Yh, const long long *yoff, const FrontArg &fa);
void launch_gemm_nt(hipStream_t st, const DevSym &S, const FrontView *frec, int nactive, int k0, int K, int c0, int c1,
                    int maxM, int maxN, double *L, const FrontArg &fa);
// cmin: fronts of at most cmin columns are skipped (passes of <= 16 right-hand sides: launch_fwd_update_wave has them, cmax = cmin)
void launch_fwd_update_recs(hipStream_t st, const DevSym &S, const FwdTile *recs, const SyrkSplit &split, int per_xcd, const double *L,
                            double *X, double *W, int nr, int ldx, int cmin = 0);
void launch_fwd_update_wave(hipStream_t st, const DevSym &S, const FwdTile *recs, const SyrkSplit &split, int per_xcd, const double *L,
                            double *X, double *W, int nr, int ldx, int cmax, bool split_k);      // split_k: the level has fronts wider than launch_wave_split_cols()
void launch_fwd_assemble(hipStream_t st, const DevSym &S, const int *list, int nfronts, int max_cols, double *X,
                         const double *W, int nr, int ldx);
void launch_fwd_update(hipStream_t st, const DevSym &S, const int *list, int nfronts, int max_trail, const double *L,
                       double *X, double *W, int nr, int ldx, int cmin = 0);
#ifdef __HIPCC__
typedef double gmrfx_d4 __attribute__((ext_vector_type(4)));
// X[k][q] of the dense inverse X = L11^-1 of a big front (0 above the diagonal): strict lower part
// stored transposed in the strict upper triangle of the panel's diagonal block, diag = 1/L's.
// Unconditional clamped load + arithmetic mask (see inverse.hip).
// 1 / v by v_rcp_f64 + one Newton step (error <= 1 ulp for the normal, positive pivots it is used on): 3
// instructions where the IEEE division sequence takes ~15 -- and the accessors below sit in GEMM inner loops,
// evaluated for EVERY operand element (the diagonal select is computed unconditionally).
__device__ __forceinline__ double fast_rcp(double v) {
    const double y = __builtin_amdgcn_rcp(v);
    return __builtin_fma(__builtin_fma(-v, y, 1.0), y, y);
}
__device__ __forceinline__ double xinv_elem(const double *__restrict__ P, int ld, int c, int k, int q) {
    const int kk = min(max(k, 0), c - 1), qq = min(max(q, 0), c - 1);
    const double v = P[min(kk, qq) + (long long)max(kk, qq) * ld];
    const bool in = k >= 0 && q >= 0 && k < c && q < c;
    double x = v * ((in && q < k) ? 1.0 : 0.0);
    if (in && k == q) x = fast_rcp(v);
    return x;
}
// 32x32 (2x2 MFMA tiles) wave-level product  acc[a][b] += sum_{q in [qlo,qhi)} fa(m0+16a+lm, q) * fb(q, n0+16b+lm)
// with operand accessors that must be safe (clamped) for any index and return 0 outside.
template <class FA, class FB>
__device__ __forceinline__ void wave_gemm_32x32(gmrfx_d4 (&acc)[2][2], int m0, int n0, int qlo, int qhi, FA fa, FB fb,
                                                int lm, int lk) {
    constexpr int KU = 4;
    for (int q0 = qlo & ~3; q0 < qhi; q0 += 4 * KU) {
        double av[KU][2], bv[KU][2];
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const int q = q0 + 4 * u + lk;
            const double mk = (q >= qlo && q < qhi) ? 1.0 : 0.0;
#pragma unroll
            for (int a = 0; a < 2; a++) av[u][a] = fa(m0 + a * 16 + lm, q) * mk;
#pragma unroll
            for (int b = 0; b < 2; b++) bv[u][b] = fb(q, n0 + b * 16 + lm);
        }
#pragma unroll
        for (int u = 0; u < KU; u++)
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][a], bv[u][b], acc[a][b], 0, 0, 0);
    }
}

// ---- the same 32 x 32 wave product with the operand rows in PAIRS: MFMA row lm of tile 0 / 1 is row 2 lm / 2 lm + 1 of the
// wave's 32 (both dimensions), so a 16-byte load feeds two tiles. Output: acc[a][b][rr] = D[m0 + 2 (lk + 4 rr) + a][n0 + 2 lm + b].
// _pm: generic accessors (masked heads / tails); _rr: both operands with contiguous rows (base + q * stride); _rk: first
// operand with contiguous rows, second with contiguous k (one row pointer per tile): k in pairs, k-step 2 h + e of a batch
// holds k = batch + 8 h + 2 lk + e. The three share the row mapping, so pieces of one K range can use different forms.
template <class FA, class FB>
__device__ __forceinline__ void wave_gemm_32x32_pm(gmrfx_d4 (&acc)[2][2], int m0, int n0, int qlo, int qhi, FA fa, FB fb,
                                                   int lm, int lk) {
    constexpr int KU = 4;
    for (int q0 = qlo & ~3; q0 < qhi; q0 += 4 * KU) {
        double av[KU][2], bv[KU][2];
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const int q = q0 + 4 * u + lk;
            const double mk = (q >= qlo && q < qhi) ? 1.0 : 0.0;
#pragma unroll
            for (int a = 0; a < 2; a++) av[u][a] = fa(m0 + 2 * lm + a, q) * mk;
#pragma unroll
            for (int b = 0; b < 2; b++) bv[u][b] = fb(q, n0 + 2 * lm + b);
        }
#pragma unroll
        for (int u = 0; u < KU; u++)
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][a], bv[u][b], acc[a][b], 0, 0, 0);
    }
}
// qlo, qhi multiples of 4
// (Round 6, measured and dropped: the three-stage software pipeline of k_syrk_cb_rec<true> for ranges of >= 48 -- k_sel_dense 107 -> 123
//  VGPRs, still three waves per SIMD, bit-identical; selected inversion of cfg 3 12.59 / 12.65 -> 13.10 / 13.11 ms on the same box: that
//  kernel runs at the fabric's bandwidth limit already, requests further ahead only deepen the queues.)
__device__ __forceinline__ void wave_gemm_32x32_rr(gmrfx_d4 (&acc)[2][2], const double *pa2, long long sa, const double *pb2,
                                                   long long sb, int qlo, int qhi, int lk) {
    constexpr int KU = 4;
    int q0 = qlo;
    for (; q0 + 4 * KU <= qhi; q0 += 4 * KU) {
        gmrfx_d2u av[KU], bv[KU];
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const long long q = q0 + 4 * u + lk;
            av[u] = *(const gmrfx_d2u *)(pa2 + q * sa);
            bv[u] = *(const gmrfx_d2u *)(pb2 + q * sb);
        }
#pragma unroll
        for (int u = 0; u < KU; u++) {
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u].x, bv[u].x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u].x, bv[u].y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u].y, bv[u].x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u].y, bv[u].y, acc[1][1], 0, 0, 0);
        }
    }
    for (; q0 < qhi; q0 += 4) {
        const long long q = q0 + lk;
        const gmrfx_d2u av = *(const gmrfx_d2u *)(pa2 + q * sa), bv = *(const gmrfx_d2u *)(pb2 + q * sb);
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, bv.x, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, bv.y, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, bv.x, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, bv.y, acc[1][1], 0, 0, 0);
    }
}
// batches of 8 k's from qlo while they fit below qhi; returns the first k it did NOT do (the caller finishes with _pm)
__device__ __forceinline__ int wave_gemm_32x32_rk(gmrfx_d4 (&acc)[2][2], const double *pa2, long long sa, const double *pb_t0,
                                                  const double *pb_t1, int qlo, int qhi, int lk) {
    int q0 = qlo;
    for (; q0 + 16 <= qhi; q0 += 16) {
        gmrfx_d2u av[4], b0[2], b1[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const long long q = q0 + 8 * h + 2 * lk;
            b0[h] = *(const gmrfx_d2u *)(pb_t0 + q);
            b1[h] = *(const gmrfx_d2u *)(pb_t1 + q);
            av[2 * h] = *(const gmrfx_d2u *)(pa2 + q * sa);
            av[2 * h + 1] = *(const gmrfx_d2u *)(pa2 + (q + 1) * sa);
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2 * h].x, b0[h].x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2 * h].x, b1[h].x, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2 * h].y, b0[h].x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2 * h].y, b1[h].x, acc[1][1], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2 * h + 1].x, b0[h].y, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2 * h + 1].x, b1[h].y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2 * h + 1].y, b0[h].y, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2 * h + 1].y, b1[h].y, acc[1][1], 0, 0, 0);
        }
    }
    for (; q0 + 8 <= qhi; q0 += 8) {
        const long long q = q0 + 2 * lk;
        const gmrfx_d2u b0 = *(const gmrfx_d2u *)(pb_t0 + q), b1 = *(const gmrfx_d2u *)(pb_t1 + q);
        const gmrfx_d2u a0 = *(const gmrfx_d2u *)(pa2 + q * sa), a1 = *(const gmrfx_d2u *)(pa2 + (q + 1) * sa);
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, b0.x, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, b1.x, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, b0.x, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, b1.x, acc[1][1], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b0.y, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b1.y, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, b0.y, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, b1.y, acc[1][1], 0, 0, 0);
    }
    return q0;
}

// the mirror image of _rk: FIRST operand with contiguous k (one row pointer per tile), second with contiguous rows
__device__ __forceinline__ int wave_gemm_32x32_kr(gmrfx_d4 (&acc)[2][2], const double *pa_t0, const double *pa_t1, const double *pb2,
                                                  long long sb, int qlo, int qhi, int lk) {
    int q0 = qlo;
    for (; q0 + 16 <= qhi; q0 += 16) {
        gmrfx_d2u bv[4], a0[2], a1[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const long long q = q0 + 8 * h + 2 * lk;
            a0[h] = *(const gmrfx_d2u *)(pa_t0 + q);
            a1[h] = *(const gmrfx_d2u *)(pa_t1 + q);
            bv[2 * h] = *(const gmrfx_d2u *)(pb2 + q * sb);
            bv[2 * h + 1] = *(const gmrfx_d2u *)(pb2 + (q + 1) * sb);
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[h].x, bv[2 * h].x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[h].x, bv[2 * h].y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[h].x, bv[2 * h].x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[h].x, bv[2 * h].y, acc[1][1], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[h].y, bv[2 * h + 1].x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[h].y, bv[2 * h + 1].y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[h].y, bv[2 * h + 1].x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[h].y, bv[2 * h + 1].y, acc[1][1], 0, 0, 0);
        }
    }
    for (; q0 + 8 <= qhi; q0 += 8) {
        const long long q = q0 + 2 * lk;
        const gmrfx_d2u a0 = *(const gmrfx_d2u *)(pa_t0 + q), a1 = *(const gmrfx_d2u *)(pa_t1 + q);
        const gmrfx_d2u b0 = *(const gmrfx_d2u *)(pb2 + q * sb), b1 = *(const gmrfx_d2u *)(pb2 + (q + 1) * sb);
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, b0.x, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, b0.y, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b0.x, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b0.y, acc[1][1], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, b1.x, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, b1.y, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, b1.x, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, b1.y, acc[1][1], 0, 0, 0);
    }
    return q0;
}

// Stage the inverse of the diagonal block into LDS as a full w x w lower-triangular matrix
// Ti[k*NB + q] = Linv[k][q] (zero above the diagonal, reciprocal on it).
__device__ __forceinline__ void stage_linv(const double *__restrict__ Dg, int ld, int w, double *Ti, int tid) {
    // 16 independent clamped loads per thread. Every use of the loaded value is unconditional
    // arithmetic (mask multiply / reciprocal), so the compiler cannot sink a load under a
    // branch and the 16 loads issue back to back.
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) {
        const int idx = tid + 256 * u;
        const int q = idx % NB, k = idx / NB;   // element Linv[k][q], stored at (q, k) for q < k
        const int qq = min(q, w - 1), kk = min(k, w - 1);
        v[u] = Dg[min(qq, kk) + (long long)max(qq, kk) * ld];
    }
#pragma unroll
    for (int u = 0; u < 16; u++) {
        const int idx = tid + 256 * u;
        const int q = idx % NB, k = idx / NB;
        const double mk = (k < w && q < k) ? 1.0 : 0.0;
        double x = v[u] * mk;
        if (q == k && k < w) x = fast_rcp(v[u]);
        Ti[k * NB + q] = x;
    }
}

// Geometry of the front a workgroup works on: from the kernel arguments (one active front: the top-of-tree chains) or from
// ONE 32-byte record at the workgroup's position in the level list (Device::d_frec_*) -- not list -> five index arrays,
// which is a dependent round trip more on every launch of the panel chains.
__device__ __forceinline__ FrontView front_view(const FrontView *__restrict__ frec, const int z, const FrontArg &fa) {
    FrontView v;
    if (fa.on) {
        v.s = fa.s; v.c = fa.c; v.r = fa.r; v.ld = fa.ld; v.first = fa.first; v.pad = 0; v.pp = fa.pp;
    } else {
        // two 16-byte loads through a differently typed pointer: written as `v = frec[z]` the compiler merges the two
        // sources into ONE pointer (kernel arguments or record) and reads the fields with flat vector loads, one
        // dependent round trip for `c` and another for the rest
        const int4 *q = reinterpret_cast<const int4 *>(frec + z);
        const int4 a = q[0], b = q[1];
        v.s = a.x; v.c = a.y; v.r = a.z; v.ld = a.w; v.first = b.x; v.pad = 0;
        v.pp = ((long long)b.w << 32) | (unsigned)b.z;
    }
    return v;
}

// Split-K reduction for NW-wave workgroups with ONE 16-row tile (4 RHS tiles of 16 columns): every wave
// writes its four partial tiles, wave t < 4 then adds tile t over the waves in order 0..NW-1 (fixed,
// reproducible) and keeps the result in acc[t]. red: NW * 4 * 4 * 64 doubles.
// NW = 8 is for launches with about one workgroup per CU: a single wave per SIMD can only issue one FP64
// MFMA per ~138 cycles, two per SIMD reach the full 64-cycle rate (tools/micro/mix64.hip) -- and the K
// chain per wave halves as well.
template <int NW>
__device__ __forceinline__ void splitk_reduce_nw(gmrfx_d4 (&acc)[4], double *red, int wave, int lane) {
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int rr = 0; rr < 4; rr++) red[((wave * 4 + t) * 4 + rr) * 64 + lane] = acc[t][rr];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; t++) {
        if (t == wave) {
            gmrfx_d4 sum;
#pragma unroll
            for (int rr = 0; rr < 4; rr++) sum[rr] = red[((0 * 4 + t) * 4 + rr) * 64 + lane];
#pragma unroll
            for (int w = 1; w < NW; w++)
#pragma unroll
                for (int rr = 0; rr < 4; rr++) sum[rr] += red[((w * 4 + t) * 4 + rr) * 64 + lane];
            acc[t] = sum;
        }
    }
}

// Split-K reduction across the 4 waves of a workgroup, DISTRIBUTED: every wave adds up ONE of the
// four 16-column tiles of each row tile (12 LDS reads in flight instead of 48 by a single wave,
// which cost ~100 VGPRs and one wave of occupancy). After the call wave w holds the complete tile
// acc[a][w] for every a; the partial sums are added in wave order 0..3 (fixed, reproducible).
// red: 4 * 3 * 4 * 64 doubles (24 KB).
template <int NA>
__device__ __forceinline__ void splitk_reduce4(gmrfx_d4 (&acc)[NA][4], double *red, int wave, int lane) {
#pragma unroll
    for (int a = 0; a < NA; a++) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (t != wave) {
                const int slot = t < wave ? t : t - 1;
#pragma unroll
                for (int rr = 0; rr < 4; rr++) red[((wave * 3 + slot) * 4 + rr) * 64 + lane] = acc[a][t][rr];
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (t == wave) {
                gmrfx_d4 part[4];
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    if (w == wave) part[w] = acc[a][t];
                    else {
                        const int slot = t < w ? t : t - 1;
#pragma unroll
                        for (int rr = 0; rr < 4; rr++) part[w][rr] = red[((w * 3 + slot) * 4 + rr) * 64 + lane];
                    }
                }
#pragma unroll
                for (int rr = 0; rr < 4; rr++) acc[a][t][rr] = ((part[0][rr] + part[1][rr]) + part[2][rr]) + part[3][rr];
            }
        }
    }
}
// The same for two row tiles with PAIR ownership: after the call wave w holds the complete tiles acc[w >> 1][2 (w & 1)]
// and acc[w >> 1][2 (w & 1) + 1] -- two column tiles of ONE row tile, which the kernels that load right-hand sides in
// pairs (column tile t = right-hand sides 32 (t >> 1) + 2 lm + (t & 1)) then store 16 bytes per lane. Two passes (one
// per column-tile pair), partial sums added in wave order 0..3. red: 12 tiles = 24 KB, as above.
__device__ __forceinline__ void splitk_reduce4_pairs(gmrfx_d4 (&acc)[2][4], double *red, int wave, int lane) {
#pragma unroll
    for (int hc = 0; hc < 2; hc++) {
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const int owner = 2 * a + hc;
            if (wave != owner) {
                const int rank = wave < owner ? wave : wave - 1;
#pragma unroll
                for (int e = 0; e < 2; e++)
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) red[(((a * 3 + rank) * 2 + e) * 4 + rr) * 64 + lane] = acc[a][2 * hc + e][rr];
            }
        }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const int owner = 2 * a + hc;
            if (wave == owner) {
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    gmrfx_d4 part[4];
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        if (w == owner) part[w] = acc[a][2 * hc + e];
                        else {
                            const int rank = w < owner ? w : w - 1;
#pragma unroll
                            for (int rr = 0; rr < 4; rr++) part[w][rr] = red[(((a * 3 + rank) * 2 + e) * 4 + rr) * 64 + lane];
                        }
                    }
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) acc[a][2 * hc + e][rr] = ((part[0][rr] + part[1][rr]) + part[2][rr]) + part[3][rr];
                }
            }
        }
    }
}
#endif

// inverse.hip -- dense L11^-1 of big fronts (recursive doubling) and the sweeps that use it
void launch_inv_stage(hipStream_t st, const DevSym &S, const int *list, int nactive, int B, int max_c, int phase,
                      double *L, double *T, const long long *toff);
void launch_xmul(hipStream_t st, const DevSym &S, const int *list, int nfronts, int max_c, int trans, const double *L,
                 const double *Xin, double *Xout, int nr, int ldx, int blk = 0, int cap = 1 << 30);
void launch_copy_own(hipStream_t st, const DevSym &S, const int *list, int nfronts, int max_c, const double *Xsrc,
                     double *Xdst, int nr, int ldx, int blk = 0, int cap = 1 << 30);
// X: rows of the ancestors (already final x, read only); Xown: the fronts' own rows, updated in place
void launch_bwd_gemm(hipStream_t st, const DevSym &S, const int *list, int nfronts, int max_cols, const double *L,
                     const double *X, double *Xown, int nr, int ldx, int blk = -1, int cap = 1 << 30, int mmin = 0);
// passes of at most 16 right-hand sides: the fronts with at most mmax trailing rows, one wave per 16 own columns (launch_bwd_gemm with
// mmin = mmax has the others)
void launch_bwd_wave(hipStream_t st, const DevSym &S, const int *list, int nfronts, int max_cols, const double *L, const double *X, double *Xown,
                     int nr, int ldx, int mmax, bool split_k);       // split_k: the level has fronts with more than launch_wave_split_rows() trailing rows
int launch_wave_split_cols();
// widest pass (right-hand sides) that takes the narrow level kernels -- one wave / one right-hand-side tile per workgroup, grid z (y
// for k_fwd_update_wave) = ceil(nr / 16) tiles: k_fwd_update_wave, k_bwd_wave, k_xmul_narrow
int narrow_pass_max();
int narrow_pass_max_bwd();
int launch_wave_split_rows();
// blocked substitution inside fronts wider than `cap` columns (forward): own rows below block blk -= L[.., block] y_blk
void launch_fwd_own_update(hipStream_t st, const DevSym &S, const int *list, int nfronts, int max_cols, const double *L,
                           const double *Y, double *X, int nr, int ldx, int blk, int cap);
// iperm: position of original row i in the elimination order (nullptr: identity)
void launch_assemble_cyclic(hipStream_t st, const DevSym &S, const int *list, int ncols, const double *nzval, double *L, double *CB,
                            int cyc_w, int cyc_r, bool compact = false);     // compact: block-cyclic STORAGE (own blocks one behind the other)
void launch_syrk_cb_cyclic(hipStream_t st, const DevSym &S, const int *list, int trail, const double *L, double *CB, int cyc_w, int cyc_r, int cyc_b0);
void launch_level_mark(hipStream_t st, int phase, int level);   // phase 1 = forward sweep, 2 = backward sweep, 3 = factorisation, 4 = selected inversion
void launch_permute(hipStream_t st, const int *iperm, int n, double *Bc, long long ldb, double *X, int nr, int ldx, int dir);
void launch_newton_update(hipStream_t st, const double *prior, double *nz, long long nnz, const long long *map, const double *h,
                          long long cnt);
int quadform_blocks(int n);
void launch_quadform(hipStream_t st, int n, const long long *colptr, const int *row, const double *val, int use_lower,
                     const double *X, long long ldx, int nvec, const double *mu, double *part, double *out);
void launch_seg_wsum(hipStream_t st, const double *src, const long long *segptr, long long nseg, const long long *off,
                     const double *w, double *out);
void launch_seg_wsum_pairs(hipStream_t st, const double *src, const long long *segptr, long long nseg, const long long *off,
                           const int *pi, const int *qi, const double *vals, double *out);
void launch_logdet(hipStream_t st, const double *L, const long long *diagoff, const unsigned char *own, int n, double *part,
                   int nparts, double *out);
void launch_gather(hipStream_t st, const double *src, const long long *off, long long cnt, double *out);
void launch_gather_diag(hipStream_t st, const double *src, const long long *diagoff, const int *perm, int n, double *out);


// dense.hip -- the dense-operator leg of the Kronecker path: R = D T (row-major, D n1 x n1, T / R n1 x n2) and a transpose
void launch_dense_apply(hipStream_t st, const double *D, const double *T, double *R, int n1, long long n2);
void launch_transpose(hipStream_t st, const double *src, double *dst, long long rows, long long cols);

// small.hip -- fused kernels for fronts with r <= 96 / 128 rows and c <= 64 columns
void launch_factor_small(hipStream_t st, const DevSym &S, const int *list, int nfronts, int rmax,
                         const double *nzval, double *L, double *CB, int *info);
// phase 0: factor, 1: forward sweep, 2: backward sweep of whole small subtrees (one workgroup per subtree)
void launch_subtree(hipStream_t st, const DevSym &S, int phase, const int *sub_first, const int *sub_last, int ntasks,
                    int rmax, const double *nzval, double *L, double *CB, int *info, double *X, double *W, int nr, int ldx);
// wmax: the widest block of the launch (<= 64): picks the workgroup shape (potrf64.hip)
void launch_potrf64(hipStream_t st, const DevSym &S, const FrontView *frec, int nactive, int kb, double *L, int *info, const FrontArg &fa,
                    int wmax = 64);
void launch_fwd_small(hipStream_t st, const DevSym &S, const int *list, int nfronts, int rmax, const double *L,
                      double *X, double *W, int nr, int ldx);
void launch_bwd_small(hipStream_t st, const DevSym &S, const int *list, int nfronts, int rmax, const double *L,
                      double *X, int nr, int ldx);

// backward step of a front of at most bwd_front_max_cols() columns as one workgroup (sweep_front.hip): x[own] = L11^-T (Yin[own] - L21' Xt[trailing])
void launch_bwd_front(hipStream_t st, const DevSym &S, const int *list, int nfronts, const double *L, const double *Xt, const double *Yin,
                      double *Xout, int nr, int ldx);
int bwd_front_max_cols();
// the forward twin: the whole forward step of such a front (own rows assembled, Y[own] = L11^-1 b, W = children - L21 y) as one workgroup
void launch_fwd_front(hipStream_t st, const DevSym &S, const int *list, int nfronts, const double *L, const double *X, double *Y, double *W,
                      int nr, int ldx);
// chunk form of the sweep tasks (sweep_chunk.hip)
void launch_pack_diag(hipStream_t st, const Symbolic::SwChunk *recs, int nchunks, const double *L, double *dtile);
void launch_sweep_chunks(hipStream_t st, const DevSym &S, int phase, const SweepTask *tasks, int ntasks, const Symbolic::SwChunk *recs_fwd,
                         const Symbolic::SwChunk *recs_bwd, const int *listf, const int *listb, const double *dtile, const double *L,
                         double *X, double *W, int nr, int ldx, size_t extra_lds);
int sweep_chunk_nc();
int sweep_chunk_spare_row();

// sweep_wave.hip -- the same tasks, one wave per (task, 16 right-hand sides); order = task ids of one LDS class
void launch_wave_tasks(hipStream_t st, const DevSym &S, int phase, const SweepTask *tasks, const int *order, int ntasks, int rows_cap,
                       const double *L, const double *rdiag, const double *zero, double *X, double *W, int nr, int ldx);
void launch_rdiag(hipStream_t st, const double *L, const long long *diagoff, int n, double *out);

// selinv.hip -- Takahashi recursion, top-down over the supernodal tree
void launch_sel_gather(hipStream_t st, const SelRec *recs, const DevSym &S, const int *list, int nfronts, int max_trail,
                       const double *Z, double *ZB);
void launch_sel_symm(hipStream_t st, const DevSym &S, const int *list, int nactive, int kb, int max_rows_below,
                     double *Z, const double *ZB, const double *Yh, const long long *yoff);
void launch_sel_diag(hipStream_t st, const DevSym &S, const int *list, int nactive, int kb, const double *L,
                     double *Z, const double *Yh, const long long *yoff);
// whole-front Takahashi step for big fronts through the dense inverse X = L11^-1:
// phase 0: Yt = (L21 X)', phase 1: Z21 = -Z22 Y, phase 2: Z11 = X'X - Y' Z21
void launch_sel_dense(hipStream_t st, const DevSym &S, const int *list, int nfronts, int phase, int max_c, int max_trail,
                      const double *L, double *Z, const double *ZB, double *Yt, double *Z21t, const long long *woff);

}  // namespace gmrfx
