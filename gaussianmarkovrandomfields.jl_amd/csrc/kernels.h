// kernels.h -- launch wrappers of the HIP kernels (kernels.hip, selinv.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "device.h"

namespace gmrfx {

constexpr int NB = 64;       // block-column width of the dense partial factorisation / sweeps
constexpr int ASM_CW = 16;   // front columns owned by one assembly workgroup
constexpr int FWD_RB = 32;    // front rows owned by one forward-assembly workgroup

void launch_assemble(hipStream_t st, const DevSym &S, const int *list, int nfronts, int max_rows,
                     const double *nzval, double *L, double *CB);
void launch_potrf(hipStream_t st, const DevSym &S, const int *list, int nactive, int kb, double *L, int *info);
void launch_trsm(hipStream_t st, const DevSym &S, const int *list, int nactive, int kb, int mode, int max_rows_below,
                 double *L, double *Yh, const long long *yoff);
void launch_gemm_nt(hipStream_t st, const DevSym &S, const int *list, int nactive, int kb, int mode, int maxM,
                    int maxN, double *L, double *CB);
void launch_fwd_assemble(hipStream_t st, const DevSym &S, const int *list, int nfronts, int max_rows, double *X,
                         double *W, int nr, int ldx);
void launch_solve_diag(hipStream_t st, const DevSym &S, const int *list, int nactive, int kb, int trans,
                       const double *L, double *X, int nr, int ldx);
void launch_fwd_update(hipStream_t st, const DevSym &S, const int *list, int nactive, int kb, int max_rows_below,
                       const double *L, double *X, double *W, int nr, int ldx);
void launch_bwd_gemm(hipStream_t st, const DevSym &S, const int *list, int nactive, int kb, int mode, int max_out,
                     const double *L, double *X, int nr, int ldx);
void launch_permute(hipStream_t st, const int *perm, int n, double *Bc, long long ldb, double *X, int nr, int ldx, int dir);
void launch_logdet(hipStream_t st, const double *L, const long long *diagoff, int n, double *part, int nparts, double *out);
void launch_gather(hipStream_t st, const double *src, const long long *off, long long cnt, double *out);
void launch_gather_diag(hipStream_t st, const double *src, const long long *diagoff, const int *perm, int n, double *out);

// small.hip -- fused kernels for fronts with r <= 96 / 128 rows and c <= 64 columns
void launch_factor_small(hipStream_t st, const DevSym &S, const int *list, int nfronts, int rmax,
                         const double *nzval, double *L, double *CB, int *info);
void launch_potrf_lds(hipStream_t st, const DevSym &S, const int *list, int nactive, int kb, double *L, int *info);
void launch_fwd_small(hipStream_t st, const DevSym &S, const int *list, int nfronts, int rmax, const double *L,
                      double *X, double *W, int nr, int ldx);
void launch_bwd_small(hipStream_t st, const DevSym &S, const int *list, int nfronts, int rmax, const double *L,
                      double *X, int nr, int ldx);

// selinv.hip -- Takahashi recursion, top-down over the supernodal tree
void launch_sel_gather(hipStream_t st, const DevSym &S, const int *list, int nfronts, int max_trail,
                       const double *Z, double *ZB);
void launch_sel_symm(hipStream_t st, const DevSym &S, const int *list, int nactive, int kb, int max_rows_below,
                     double *Z, const double *ZB, const double *Yh, const long long *yoff);
void launch_sel_diag(hipStream_t st, const DevSym &S, const int *list, int nactive, int kb, const double *L,
                     double *Z, const double *Yh, const long long *yoff);

}  // namespace gmrfx
