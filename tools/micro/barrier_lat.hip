// micro-benchmark: cost of s_barrier and of LDS hand-offs inside ONE workgroup, by number of waves (4 .. 16), on one CU.
// Used to budget the role-split diagonal-block kernel (csrc/potrf64.hip): a 4-column step there is two barriers and two LDS hand-offs.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 barrier_lat.hip -o barrier_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#define HC(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
constexpr int ITERS = 8192;
// MODE 0: barrier only; 1: every wave writes 32 B/lane, barrier, reads the next wave's, barrier (two barriers per iteration);
// 2: ping-pong between wave 0 and the others: others write, barrier, wave 0 reads+writes, barrier, others read (the kernel's shape);
// 3: as 2 with a dependent chain of 40 FP64 FMAs on wave 0 between its read and its write; 4: as 3 plus one FP64 MFMA on every other wave
//    after the second barrier (does the MFMA of the waves on wave 0's SIMD delay its chain?)
template <int MODE>
__global__ void k_bar(double *out, int iters) {
    __shared__ __attribute__((aligned(16))) double buf[2][16 * 64 * 4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 v = (d4){1.0 + lane, 2.0, 3.0, 4.0 + wave};
    d4 acc = (d4){0, 0, 0, 0};
    double chain = 1.0 + 1e-9 * lane;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
            asm volatile("s_barrier" ::: "memory");
        } else if (MODE == 1) {
            *(d4 *)(buf[0] + (wave * 64 + lane) * 4) = v;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            d4 u = *(const d4 *)(buf[0] + (((wave + 1) % nw) * 64 + lane) * 4);
            v += u * 1e-9;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
            if (wave != 0) *(d4 *)(buf[0] + (wave * 64 + lane) * 4) = v;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (wave == 0) {
                d4 u = *(const d4 *)(buf[0] + (1 * 64 + lane) * 4);
                if (MODE >= 3) {
                    double c = u.x * 1e-9 + chain;
#pragma unroll
                    for (int k = 0; k < 40; k++) c = __builtin_fma(c, 0.999999, 1e-7);
                    chain = c; u.y += c;
                }
                *(d4 *)(buf[1] + lane * 4) = u;
            } else if (MODE == 4) {
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v.x, v.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v.z, v.w, acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (wave != 0) { d4 u = *(const d4 *)(buf[1] + lane * 4); v += u * 1e-9; }
        }
    }
    out[threadIdx.x] = v.x + v.w + acc[0] + chain;
}
int main() {
    double *d; HC(hipMalloc(&d, 1024 * 8));
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    float ms;
    for (int nw : {4, 8, 9, 12, 16}) {
        auto run = [&](const char *nm, auto kern, int per) {
            hipLaunchKernelGGL(kern, dim3(1), dim3(64 * nw), 0, 0, d, 64);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kern, dim3(1), dim3(64 * nw), 0, 0, d, ITERS);
            hipEventRecord(e1, 0); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1);
            printf("%2d waves  %-58s %7.1f ns per iteration (%d barriers)\n", nw, nm, ms * 1e6 / ITERS, per);
        };
        run("barrier only", k_bar<0>, 1);
        run("all write, barrier, all read, barrier", k_bar<1>, 2);
        run("others write, B, wave 0 reads + writes, B, others read", k_bar<2>, 2);
        run("  + 40 dependent FP64 FMAs on wave 0", k_bar<3>, 2);
        run("  + 2 FP64 MFMAs on every other wave beside the chain", k_bar<4>, 2);
    }
    return 0;
}
