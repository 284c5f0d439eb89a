// potrf64.hip -- Cholesky factor AND inverse of one 64 x 64 diagonal block per workgroup: the kernel on the critical path of
// every big front (one call per 64 columns; the root front of a 1000 x 1000 grid alone needs 32 of them, one after the other),
// built for LATENCY. The arithmetic lives in potrf64_blocked.h (16-column steps: one wave eliminates 16 x 16 diagonal blocks in
// registers, helper waves keep the rest of the block and its inverse up to date on the MFMA beside it; 9.2 us per block).
// Earlier forms -- register patches with a 4-column step (16-18 us), a role-split block, a look-ahead chain that keeps its own
// band up to date -- were measured and removed; their numbers are in DESIGN.md section 3.
// Reference semantics: cholesky!(F, Q) of CHOLMOD as used by /root/reference/src/workspace/backend.jl:178-189 (refactorize!) --
// here for one dense diagonal block of a supernode.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "potrf64_blocked.h"

namespace gmrfx {

typedef gmrfx_d4 d4;

// The workgroup shape follows the widest block of the launch: <1, 1> for blocks up to 16 columns (the diagonal wave alone),
// <2, 2> up to 32, <4, 3> up to 48, <8, 4> up to 64 -- the wide levels of the tree bring thousands of narrow fronts per launch.
template <int NW, int ND>
__global__ __launch_bounds__(64 * NW) void k_potrf64_b(DevSym S, const FrontView *__restrict__ frec, int kb,
                                                       double *__restrict__ L, int *__restrict__ info, FrontArg fa) {
    __shared__ __attribute__((aligned(16))) pb::Smem<ND> sm;
    const FrontView fv = front_view(frec, blockIdx.x, fa);
    const int c = fv.c;
    if (kb >= c) return;
    const int w = min(NB, c - kb);
    const int ld = fv.ld;
    double *P = L + fv.pp + kb + (long long)kb * ld;
    pb::potrf64_blocked<NW, ND>(P, ld, P, ld, w, sm, info, fv.first + kb, threadIdx.x);
}

void launch_potrf64(hipStream_t st, const DevSym &S, const FrontView *frec, int nactive, int kb, double *L, int *info,
                    const FrontArg &fa, int wmax) {
    if (nactive <= 0) return;
    if (wmax <= 16) hipLaunchKernelGGL((k_potrf64_b<1, 1>), dim3(nactive), dim3(64), 0, st, S, frec, kb, L, info, fa);
    else if (wmax <= 32) hipLaunchKernelGGL((k_potrf64_b<2, 2>), dim3(nactive), dim3(128), 0, st, S, frec, kb, L, info, fa);
    else if (wmax <= 48) hipLaunchKernelGGL((k_potrf64_b<4, 3>), dim3(nactive), dim3(256), 0, st, S, frec, kb, L, info, fa);
    else hipLaunchKernelGGL((k_potrf64_b<8, 4>), dim3(nactive), dim3(512), 0, st, S, frec, kb, L, info, fa);
}

}  // namespace gmrfx
