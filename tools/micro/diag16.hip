// micro-benchmark: Cholesky factor AND inverse of a 16 x 16 block by ONE wave, all in registers -- no LDS hand-off, no barrier.
// Lane (i = lane & 15, q = lane >> 4) holds row i, columns 4 r + q (r = 0..3) of the symmetric block A and of M (starts as I).
// Column step k (fully unrolled, every index static):
//   pivot      a[k][k]            -> v_readlane (scalar)               -> rs = rsqrt, rp = rs^2 (uniform VALU)
//   column k   a[i][k], i = 0..15 -> ds_bpermute (LDS crossbar, no LDS memory) to the four lanes of row i
//   row k      a[k][.], m[k][.]   -> DPP row_newbcast:k inside each group of 16 lanes
//   rows i > k: a[i][.] -= (a_ik rp) a[k][.]; m[i][.] -= (a_ik rp) m[k][.]        (Gaussian elimination on [A | I])
// L[i][k] = a_ik rs; L^-1 = diag(rs) M. This is the core of a 16-column-step 64 x 64 diagonal block (csrc/potrf64.hip).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 diag16.hip -o diag16
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#define HC(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

__device__ __forceinline__ double rsqrt_nr(double p) {
    double y = __builtin_amdgcn_rsq(p);
    y = y * (1.5 - 0.5 * p * y * y);
    y = y * (1.5 - 0.5 * p * y * y);
    return y;
}
template <int K>
__device__ __forceinline__ double row_bcast(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + K, 0xf, 0xf, true);     // row_newbcast:K (bound_ctrl: no old value to keep)
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + K, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bperm(double v, int addr) {
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_d(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

struct Diag16 {
    double a[4], m[4], l[4], rs_own;
    int bad;
};

template <int K>
__device__ __forceinline__ void diag16_step(Diag16 &D, const int i, const int q) {
    constexpr int KG = K & 3, KR = K >> 2;
    const double pv = readlane_d(D.a[KR], 16 * KG + K);
    const double col = bperm(D.a[KR], 4 * (16 * KG + i));
    const double rs = rsqrt_nr(pv);
    const double rp = rs * rs;
    if (!(pv > 0.0)) D.bad = min(D.bad, K);
    const double lcol = col * rs;
    if (q == KG) D.l[KR] = (i >= K) ? lcol : 0.0;
    D.rs_own = (i == K) ? rs : D.rs_own;
    const double f = (i > K) ? col * rp : 0.0;
#pragma unroll
    for (int r = KR; r < 4; r++) D.a[r] -= f * row_bcast<K>(D.a[r]);
#pragma unroll
    for (int r = 0; r <= KR; r++) D.m[r] -= f * row_bcast<K>(D.m[r]);
}

__device__ __forceinline__ void diag16_factor(Diag16 &D, const int i, const int q) {
    diag16_step<0>(D, i, q);  diag16_step<1>(D, i, q);  diag16_step<2>(D, i, q);  diag16_step<3>(D, i, q);
    diag16_step<4>(D, i, q);  diag16_step<5>(D, i, q);  diag16_step<6>(D, i, q);  diag16_step<7>(D, i, q);
    diag16_step<8>(D, i, q);  diag16_step<9>(D, i, q);  diag16_step<10>(D, i, q); diag16_step<11>(D, i, q);
    diag16_step<12>(D, i, q); diag16_step<13>(D, i, q); diag16_step<14>(D, i, q); diag16_step<15>(D, i, q);
}

// A: 16 x 16 column-major SPD. Out: Lo (lower, column-major), Xo = L^-1 (lower, column-major). reps: repeat for timing
__global__ __launch_bounds__(64) void k_diag16(const double *A, double *Lo, double *Xo, long long *cyc, int reps) {
    const int lane = threadIdx.x, i = lane & 15, q = lane >> 4;
    Diag16 D;
    long long t = 0;
    for (int rep = 0; rep < reps; rep++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            D.a[r] = A[i + 16 * (4 * r + q)];
            D.m[r] = (4 * r + q == i) ? 1.0 : 0.0;
            D.l[r] = 0.0;
        }
        D.rs_own = 0.0; D.bad = 99;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const long long t0 = clock64();
        __builtin_amdgcn_sched_barrier(0);
        diag16_factor(D, i, q);
        asm volatile("" ::"v"(D.a[3]), "v"(D.m[3]), "v"(D.l[3]), "v"(D.m[0]), "v"(D.l[0]));
        __builtin_amdgcn_sched_barrier(0);
        t += clock64() - t0;
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        Lo[i + 16 * (4 * r + q)] = D.l[r];
        Xo[i + 16 * (4 * r + q)] = D.rs_own * D.m[r];
    }
    if (lane == 0) { cyc[0] = t / reps; cyc[1] = D.bad; }
}

int main() {
    const int n = 16;
    std::vector<double> A(n * n), L(n * n), X(n * n);
    // SPD: diagonally dominant with a smooth off-diagonal part
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) A[i + n * j] = (i == j) ? 6.0 + 0.1 * i : -1.0 / (1.0 + std::abs(i - j)) + 0.01 * std::cos(i * 3 + j * 3);
    for (int j = 0; j < n; j++) for (int i = 0; i < j; i++) A[i + n * j] = A[j + n * i];
    double *dA, *dL, *dX; long long *dC;
    HC(hipMalloc(&dA, n * n * 8)); HC(hipMalloc(&dL, n * n * 8)); HC(hipMalloc(&dX, n * n * 8)); HC(hipMalloc(&dC, 16));
    HC(hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_diag16, dim3(1), dim3(64), 0, 0, dA, dL, dX, dC, 1);
    HC(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_diag16, dim3(1), dim3(64), 0, 0, dA, dL, dX, dC, 64);
    HC(hipDeviceSynchronize());
    long long c[2];
    HC(hipMemcpy(c, dC, 16, hipMemcpyDeviceToHost));
    HC(hipMemcpy(L.data(), dL, n * n * 8, hipMemcpyDeviceToHost));
    HC(hipMemcpy(X.data(), dX, n * n * 8, hipMemcpyDeviceToHost));
    double e1 = 0, e2 = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++) {
            double s = 0, t = 0;
            for (int k = 0; k <= j; k++) s += L[i + n * k] * L[j + n * k];
            for (int k = j; k <= i; k++) t += L[i + n * k] * X[k + n * j];
            const double d1 = std::abs(s - A[i + n * j]), d2 = std::abs(t - (i == j ? 1.0 : 0.0));
            e1 = (d1 > e1 || d1 != d1) ? d1 : e1;
            e2 = (d2 > e2 || d2 != d2) ? d2 : e2;
        }
    double up = 0;
    for (int j = 0; j < n; j++) for (int i = 0; i < j; i++) up = std::fmax(up, std::fmax(std::abs(L[i + n * j]), std::abs(X[i + n * j])));
    printf("diag16 (one wave, registers + DPP + bpermute): %lld cycles per 16 x 16 block (factor + inverse), %.0f per column; bad %lld\n", c[0], c[0] / 16.0, c[1]);
    printf("  max |LL' - A| = %.3e, max |L X - I| = %.3e, strict upper parts max %.1e\n", e1, e2, up);
    return 0;
}
