// klchol.hip -- KL-optimal (Vecchia-type) sparse approximate Cholesky factor of a precision, L L' ~ Theta^-1:
// a batch of independent small dense problems, one workgroup each.
//
// Reference semantics (/root/reference/src/kl_cholesky/kl_cholesky.jl):
//   * :32-55  sparse_approximate_cholesky!(Theta, L): for every column k of the lower-triangular pattern L,
//     S = its row indices in DESCENDING order, M = Theta[S, S] + 1e-6 I = U'U, solve U x = e_last,
//     L[S, k] = x;
//   * :74-113 sparse_approximate_cholesky(Theta, sc::SupernodeClustering): one M = Theta[R, R] + 1e-8 I per
//     supernode (R = its rows, descending), one right-hand side e_{N_k} per member column k (N_k = nnz of column
//     k = a prefix of R), L.nzval[column k] = x[N_k:-1:1].
// Both are the same task: rows R (in the local order the caller chose), a list of columns, and for each column
// the unit vector at position N_k. With M = C C' (C lower = U'), U x = e_{N_k} is the back substitution
// C' x = e_{N_k}, which only touches the leading N_k x N_k block of C.
//
// One workgroup per task: M is gathered from the dense Theta (device resident) into LDS (global scratch for
// local systems of more than 128 rows), factored in place (right-looking, two barriers per column), and every
// wave back-substitutes one right-hand side at a time with the vector held in registers (x_j is broadcast with
// a lane shuffle; no barriers). Work per task: N^2 gathered values, N^3/3 + N_k^2 per column flops -- the
// gather from Theta (8 N^2 bytes, strided) bounds the small systems, the barrier chain the large ones.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>

#include "device.h"
#include "kernels.h"

namespace gmrfx {

struct KlTask {
    long long rows_off;   // into rows[]
    long long cols_off;   // into cols[]
    int nrows, ncols;
};

namespace {

// M: lower triangle of the n x n local matrix (leading dimension ldm), in LDS or in global scratch.
template <int SLOTS>
__device__ __forceinline__ void kl_task(double *M, const int ldm, const int n, const int *__restrict__ R,
                                        const double *__restrict__ theta, const long long ldt, const double reg,
                                        const int *__restrict__ cols, const int ncols,
                                        const long long *__restrict__ Lcolptr, double *__restrict__ nzval,
                                        int *__restrict__ info, const int task_id) {
    const int tid = threadIdx.x;
    for (int idx = tid; idx < n * n; idx += 256) {
        const int i = idx % n, j = idx / n;
        if (i >= j) M[i + j * ldm] = theta[R[i] + (long long)R[j] * ldt] + (i == j ? reg : 0.0);
    }
    __syncthreads();
    // right-looking, two barriers per column; the diagonal keeps the pivots d_j until the loop is done (nobody
    // writes M(j, j) while others may still read it), then one pass turns them into C(j, j) = sqrt(d_j)
    for (int j = 0; j < n; j++) {
        const double d = M[j + j * ldm];
        if (!(d > 0.0) && tid == 0) atomicMin(info, task_id);    // cholesky! would throw PosDefException
        const double inv = 1.0 / sqrt(d);
        for (int i = j + 1 + tid; i < n; i += 256) M[i + j * ldm] *= inv;
        __syncthreads();
        const int m = n - j - 1;
        for (int idx = tid; idx < m * m; idx += 256) {
            const int i = j + 1 + idx % m, k = j + 1 + idx / m;
            if (i >= k) M[i + k * ldm] -= M[i + j * ldm] * M[k + j * ldm];
        }
        __syncthreads();
    }
    for (int j = tid; j < n; j += 256) M[j + j * ldm] = sqrt(M[j + j * ldm]);
    __syncthreads();
    // back substitution C' x = e_{nk}, one wave per right-hand side; lane l holds x_i for i = l + 64 s
    const int wave = tid >> 6, lane = tid & 63;
    for (int q = wave; q < ncols; q += 4) {
        const int col = cols[q];
        const long long p0 = Lcolptr[col];
        const int nk = (int)(Lcolptr[col + 1] - p0);
        double b[SLOTS];
#pragma unroll
        for (int s = 0; s < SLOTS; s++) b[s] = (lane + 64 * s == nk - 1) ? 1.0 : 0.0;
        for (int j = nk - 1; j >= 0; j--) {
            double bj = 0.0;
#pragma unroll
            for (int s = 0; s < SLOTS; s++)
                if ((j >> 6) == s) bj = __shfl(b[s], j & 63, 64);
            const double xj = bj / M[j + j * ldm];
#pragma unroll
            for (int s = 0; s < SLOTS; s++) {
                const int i = lane + 64 * s;
                if (i < j) b[s] -= M[j + i * ldm] * xj;      // C(j, i), row j of the factor
                else if (i == j) b[s] = xj;
            }
        }
#pragma unroll
        for (int s = 0; s < SLOTS; s++) {
            const int i = lane + 64 * s;
            if (i < nk) nzval[p0 + (nk - 1 - i)] = b[s];
        }
    }
}

template <int NMAX>
__global__ __launch_bounds__(256) void k_kl_chol(const KlTask *__restrict__ tasks, const int *__restrict__ order,
                                                 const int *__restrict__ rows, const int *__restrict__ cols,
                                                 const double *__restrict__ theta, long long ldt, double reg,
                                                 const long long *__restrict__ Lcolptr, double *__restrict__ nzval,
                                                 int *__restrict__ info) {
    extern __shared__ double smem[];
    double *M = smem;                                   // NMAX x (NMAX + 1)
    int *Rl = (int *)(smem + NMAX * (NMAX + 1));        // NMAX
    const int t = order[blockIdx.x];
    const KlTask tk = tasks[t];
    for (int i = threadIdx.x; i < tk.nrows; i += 256) Rl[i] = rows[tk.rows_off + i];
    __syncthreads();
    kl_task<(NMAX + 63) / 64>(M, NMAX + 1, tk.nrows, Rl, theta, ldt, reg, cols + tk.cols_off, tk.ncols, Lcolptr, nzval, info, t);
}

constexpr int KL_BIG = 512;   // largest local system (global-scratch variant)

__global__ __launch_bounds__(256) void k_kl_chol_big(const KlTask *__restrict__ tasks, const int *__restrict__ order,
                                                     const int *__restrict__ rows, const int *__restrict__ cols,
                                                     const double *__restrict__ theta, long long ldt, double reg,
                                                     const long long *__restrict__ Lcolptr, double *__restrict__ nzval,
                                                     int *__restrict__ info, double *__restrict__ scratch) {
    const int t = order[blockIdx.x];
    const KlTask tk = tasks[t];
    double *M = scratch + (long long)blockIdx.x * KL_BIG * KL_BIG;
    kl_task<KL_BIG / 64>(M, tk.nrows, tk.nrows, rows + tk.rows_off, theta, ldt, reg, cols + tk.cols_off, tk.ncols, Lcolptr,
                         nzval, info, t);
}

struct DBuf {
    void *p = nullptr;
    ~DBuf() { if (p) (void)hipFree(p); }
    template <class T> T *alloc(size_t cnt) { hip_check(hipMalloc(&p, std::max<size_t>(cnt, 1) * sizeof(T)), "hipMalloc"); return (T *)p; }
};

template <int NMAX> void launch_class(hipStream_t st, int cnt, const KlTask *d_tasks, const int *d_order, const int *d_rows,
                                      const int *d_cols, const double *d_theta, long long ldt, double reg,
                                      const long long *d_colptr, double *d_nz, int *d_info) {
    if (cnt <= 0) return;
    const size_t lds = (size_t)NMAX * (NMAX + 1) * sizeof(double) + (size_t)NMAX * sizeof(int);
    hip_check(hipFuncSetAttribute((const void *)k_kl_chol<NMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute");
    hipLaunchKernelGGL(k_kl_chol<NMAX>, dim3(cnt), dim3(256), lds, st, d_tasks, d_order, d_rows, d_cols, d_theta, ldt, reg,
                       d_colptr, d_nz, d_info);
}

}  // namespace

// Host driver. All index arrays are 0-based here; theta is n x n column-major (host or device).
// Returns -1, or the index of the first task whose local matrix is not positive definite.
long long kl_cholesky_run(int device, long long n, const double *theta, long long ldt, bool theta_on_device,
                          const std::vector<KlTask> &tasks, const std::vector<int> &rows, const std::vector<int> &cols,
                          const long long *Lcolptr, long long nnzL, double reg, double *nzval_out) {
    if (device >= 0) hip_check(hipSetDevice(device), "hipSetDevice");
    const int ntasks = (int)tasks.size();
    if (ntasks == 0) return -1;
    // size classes
    std::vector<int> order[4];
    for (int t = 0; t < ntasks; t++) {
        const int N = tasks[t].nrows;
        if (N > KL_BIG) throw std::invalid_argument("kl_cholesky: a local system has more than 512 rows (pattern too dense)");
        order[N <= 32 ? 0 : (N <= 64 ? 1 : (N <= 128 ? 2 : 3))].push_back(t);
    }
    hipStream_t st;
    hip_check(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate");
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } sg{st};
    DBuf btheta, btasks, border, brows, bcols, bcolptr, bnz, binfo, bscr;
    const double *d_theta = theta;
    if (!theta_on_device) {
        double *p = btheta.alloc<double>((size_t)n * n);
        hip_check(hipMemcpy2DAsync(p, (size_t)n * sizeof(double), theta, (size_t)ldt * sizeof(double), (size_t)n * sizeof(double),
                                   (size_t)n, hipMemcpyHostToDevice, st), "hipMemcpy2D");
        d_theta = p;
        ldt = n;
    }
    KlTask *d_tasks = btasks.alloc<KlTask>(ntasks);
    int *d_rows = brows.alloc<int>(rows.size());
    int *d_cols = bcols.alloc<int>(cols.size());
    long long *d_colptr = bcolptr.alloc<long long>((size_t)n + 1);
    double *d_nz = bnz.alloc<double>((size_t)nnzL);
    int *d_info = binfo.alloc<int>(1);
    std::vector<int> all_order;
    int off[5] = {0, 0, 0, 0, 0};
    for (int k = 0; k < 4; k++) { off[k + 1] = off[k] + (int)order[k].size(); all_order.insert(all_order.end(), order[k].begin(), order[k].end()); }
    int *d_order = border.alloc<int>(all_order.size());
    const int big = 0x7fffffff;
    hip_check(hipMemcpyAsync(d_tasks, tasks.data(), (size_t)ntasks * sizeof(KlTask), hipMemcpyHostToDevice, st), "copy tasks");
    hip_check(hipMemcpyAsync(d_rows, rows.data(), rows.size() * sizeof(int), hipMemcpyHostToDevice, st), "copy rows");
    hip_check(hipMemcpyAsync(d_cols, cols.data(), cols.size() * sizeof(int), hipMemcpyHostToDevice, st), "copy cols");
    hip_check(hipMemcpyAsync(d_colptr, Lcolptr, ((size_t)n + 1) * sizeof(long long), hipMemcpyHostToDevice, st), "copy colptr");
    hip_check(hipMemcpyAsync(d_order, all_order.data(), all_order.size() * sizeof(int), hipMemcpyHostToDevice, st), "copy order");
    hip_check(hipMemcpyAsync(d_info, &big, sizeof(int), hipMemcpyHostToDevice, st), "copy info");
    hip_check(hipMemsetAsync(d_nz, 0, (size_t)nnzL * sizeof(double), st), "memset");
    launch_class<32>(st, (int)order[0].size(), d_tasks, d_order + off[0], d_rows, d_cols, d_theta, ldt, reg, d_colptr, d_nz, d_info);
    launch_class<64>(st, (int)order[1].size(), d_tasks, d_order + off[1], d_rows, d_cols, d_theta, ldt, reg, d_colptr, d_nz, d_info);
    launch_class<128>(st, (int)order[2].size(), d_tasks, d_order + off[2], d_rows, d_cols, d_theta, ldt, reg, d_colptr, d_nz, d_info);
    if (!order[3].empty()) {
        const int chunk = 256;   // 256 x 2 MB of scratch
        double *d_scr = bscr.alloc<double>((size_t)chunk * KL_BIG * KL_BIG);
        for (int b = 0; b < (int)order[3].size(); b += chunk) {
            const int cnt = std::min(chunk, (int)order[3].size() - b);
            hipLaunchKernelGGL(k_kl_chol_big, dim3(cnt), dim3(256), 0, st, d_tasks, d_order + off[3] + b, d_rows, d_cols, d_theta,
                               ldt, reg, d_colptr, d_nz, d_info, d_scr);
        }
    }
    hip_check(hipGetLastError(), "kl_cholesky launch");
    int info = big;
    hip_check(hipMemcpyAsync(nzval_out, d_nz, (size_t)nnzL * sizeof(double), hipMemcpyDeviceToHost, st), "copy out");
    hip_check(hipMemcpyAsync(&info, d_info, sizeof(int), hipMemcpyDeviceToHost, st), "copy info");
    hip_check(hipStreamSynchronize(st), "sync");
    return info == big ? -1 : info;
}

}  // namespace gmrfx
