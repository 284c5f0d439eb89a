// rccl_driver_test.cpp -- the native RCCL driver (libgmrfx_rccl.so, include/gmrfx_rccl.h) on a ONE-rank communicator, no Python in the
// process: a sharded handle of one rank with a forced top (gmrfx_opts.shard_min_top) goes through gmrfx_rccl_refactorize / _solve /
// _backward_solve / _logdet / _selinv_diag and must reproduce an unsharded handle of the same matrix bit for bit (solves, samples)
// resp. to 1e-12 (log-determinant, selected-inverse diagonal). Built by `make -C gaussianmarkovrandomfields.jl_amd rccl`, run on a GPU
// box by tests/test_rccl_world1.py. Exit code 0 + "rccl_driver_test: ok" = pass.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/gmrfx_rccl.h"

static int fails = 0;
#define EXPECT(c) do { if (!(c)) { std::fprintf(stderr, "FAILED: %s (line %d)\n", #c, __LINE__); fails++; } } while (0)
#define HIPOK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { std::fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

int main() {
    // 19-point pattern of (5-point Laplacian)^2 on an nx x ny grid, both triangles; strictly diagonally dominant values
    const int nx = 150, ny = 140;
    const int64_t n = (int64_t)nx * ny;
    std::vector<int64_t> cp(n + 1, 0), ri;
    std::vector<double> nz, xy(2 * n);
    for (int j = 0; j < ny; j++)
        for (int i = 0; i < nx; i++) {
            const int64_t v = (int64_t)j * nx + i;
            xy[2 * v] = i; xy[2 * v + 1] = j;
            for (int dj = -2; dj <= 2; dj++)
                for (int di = -2; di <= 2; di++) {
                    if (std::abs(di) + std::abs(dj) > 2) continue;
                    const int a = i + di, b = j + dj;
                    if (a < 0 || b < 0 || a >= nx || b >= ny) continue;
                    ri.push_back((int64_t)b * nx + a);
                    nz.push_back(di == 0 && dj == 0 ? 13.0 + 0.001 * (double)((i * 7 + j * 3) % 11) : -1.0 / (1.0 + std::abs(di) + std::abs(dj)));
                }
            cp[v + 1] = (int64_t)ri.size();
        }
    gmrfx_opts o;
    std::memset(&o, 0, sizeof(o));
    o.struct_size = (int32_t)sizeof(o);
    o.device = 0; o.coord_dim = 2; o.coords = xy.data();
    gmrfx_handle *ref = nullptr, *sh = nullptr;
    EXPECT(gmrfx_create(n, cp.data(), ri.data(), 0, nullptr, &o, &ref) == GMRFX_OK);
    o.shard_rank = 0; o.shard_world = 1; o.shard_min_top = 3;
    EXPECT(gmrfx_create(n, cp.data(), ri.data(), 0, nullptr, &o, &sh) == GMRFX_OK);
    if (!ref || !sh) { std::fprintf(stderr, "create failed: %s\n", gmrfx_last_create_error()); return 1; }
    int64_t ne = 0, ntop = 0, sl = 0;
    EXPECT(gmrfx_shard_info(sh, &ne, &ntop, &sl) == GMRFX_OK && ntop >= 3 && ne == 0);

    const int64_t nrhs = 70;                        // two passes: 64 + 6
    double *d_nz = nullptr, *d_B = nullptr, *d_X = nullptr, *d_Xr = nullptr;
    HIPOK(hipMalloc((void **)&d_nz, nz.size() * sizeof(double)));
    HIPOK(hipMalloc((void **)&d_B, (size_t)(n * nrhs) * sizeof(double)));
    HIPOK(hipMalloc((void **)&d_X, (size_t)(n * nrhs) * sizeof(double)));
    HIPOK(hipMalloc((void **)&d_Xr, (size_t)(n * nrhs) * sizeof(double)));
    std::vector<double> B((size_t)(n * nrhs)), X(B.size()), Xr(B.size());
    uint64_t s = 0x9e3779b97f4a7c15ull;
    for (auto &v : B) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (double)(int64_t)(s >> 11) / 9007199254740992.0 - 0.5; }
    HIPOK(hipMemcpy(d_nz, nz.data(), nz.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPOK(hipMemcpy(d_B, B.data(), B.size() * sizeof(double), hipMemcpyHostToDevice));

    // reference: the unsharded handle
    int64_t info = -1;
    EXPECT(gmrfx_refactorize_dev(ref, d_nz, &info) == GMRFX_OK && info == 0);
    EXPECT(gmrfx_solve_dev(ref, d_B, n, nrhs, d_Xr, n) == GMRFX_OK);
    double ld_ref = 0;
    EXPECT(gmrfx_logdet(ref, &ld_ref) == GMRFX_OK);
    HIPOK(hipDeviceSynchronize());
    HIPOK(hipMemcpy(Xr.data(), d_Xr, Xr.size() * sizeof(double), hipMemcpyDeviceToHost));

    // the native driver on a one-rank communicator
    unsigned char id[128];
    EXPECT(gmrfx_rccl_unique_id(id) == GMRFX_OK);
    gmrfx_rccl *drv = nullptr;
    EXPECT(gmrfx_rccl_create(sh, 1, 0, id, nullptr, &drv) == GMRFX_OK);
    if (!drv) { std::fprintf(stderr, "gmrfx_rccl_create failed\n"); return 1; }
    std::vector<uint8_t> mask((size_t)n, 0);
    EXPECT(gmrfx_rccl_needed_rows(drv, mask.data()) == GMRFX_OK);
    { int64_t c = 0; for (auto m : mask) c += m; EXPECT(c == n); }
    for (int rep = 0; rep < 2; rep++) {            // twice: the second run reuses every buffer
        if (gmrfx_rccl_refactorize(drv, d_nz) != GMRFX_OK) { std::fprintf(stderr, "refactorize: %s\n", gmrfx_rccl_last_error(drv)); return 1; }
        HIPOK(hipMemset(d_X, 0xff, X.size() * sizeof(double)));
        HIPOK(hipDeviceSynchronize());
        if (gmrfx_rccl_solve(drv, d_B, n, nrhs, d_X, n, rep) != GMRFX_OK) { std::fprintf(stderr, "solve: %s\n", gmrfx_rccl_last_error(drv)); return 1; }
        double ld = 0; info = -1;
        if (gmrfx_rccl_logdet(drv, &ld, &info) != GMRFX_OK) { std::fprintf(stderr, "logdet: %s\n", gmrfx_rccl_last_error(drv)); return 1; }
        EXPECT(info == 0 && std::fabs(ld - ld_ref) <= 1e-12 * std::fabs(ld_ref));
        HIPOK(hipDeviceSynchronize());
        HIPOK(hipMemcpy(X.data(), d_X, X.size() * sizeof(double), hipMemcpyDeviceToHost));
        EXPECT(std::memcmp(X.data(), Xr.data(), X.size() * sizeof(double)) == 0);          // bit for bit
    }
    // backward-only solve (the sampling path)
    EXPECT(gmrfx_backward_solve_dev(ref, d_B, n, nrhs, d_Xr, n) == GMRFX_OK);
    HIPOK(hipDeviceSynchronize());
    if (gmrfx_rccl_backward_solve(drv, d_B, n, nrhs, d_X, n, 1) != GMRFX_OK) { std::fprintf(stderr, "backward solve: %s\n", gmrfx_rccl_last_error(drv)); return 1; }
    HIPOK(hipDeviceSynchronize());
    HIPOK(hipMemcpy(X.data(), d_X, X.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPOK(hipMemcpy(Xr.data(), d_Xr, Xr.size() * sizeof(double), hipMemcpyDeviceToHost));
    EXPECT(std::memcmp(X.data(), Xr.data(), X.size() * sizeof(double)) == 0);
    // selected-inverse diagonal
    std::vector<double> sd((size_t)n), sdr((size_t)n);
    EXPECT(gmrfx_selinv_diag(ref, sdr.data()) == GMRFX_OK);
    if (gmrfx_rccl_selinv_diag(drv, sd.data()) != GMRFX_OK) { std::fprintf(stderr, "selinv: %s\n", gmrfx_rccl_last_error(drv)); return 1; }
    double worst = 0;
    for (int64_t k = 0; k < n; k++) worst = std::fmax(worst, std::fabs(sd[(size_t)k] - sdr[(size_t)k]) / std::fabs(sdr[(size_t)k]));
    EXPECT(worst <= 1e-12);
    // a residual, so that "equal" also means "right": || Q x - b || / || b || of the first column
    {
        double num = 0, den = 0;
        HIPOK(hipMemcpy(Xr.data(), d_Xr, Xr.size() * sizeof(double), hipMemcpyDeviceToHost));
        EXPECT(gmrfx_solve_dev(ref, d_B, n, 1, d_Xr, n) == GMRFX_OK);
        HIPOK(hipDeviceSynchronize());
        HIPOK(hipMemcpy(Xr.data(), d_Xr, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
        for (int64_t j = 0; j < n; j++) {
            double acc = 0;
            for (int64_t p = cp[j]; p < cp[j + 1]; p++) acc += nz[(size_t)p] * Xr[(size_t)ri[(size_t)p]];       // (symmetric: row j = column j)
            num += (acc - B[(size_t)j]) * (acc - B[(size_t)j]); den += B[(size_t)j] * B[(size_t)j];
        }
        EXPECT(std::sqrt(num / den) < 1e-12);
    }
    gmrfx_rccl_destroy(drv);
    gmrfx_destroy(sh);
    gmrfx_destroy(ref);
    (void)hipFree(d_nz); (void)hipFree(d_B); (void)hipFree(d_X); (void)hipFree(d_Xr);
    std::printf("rccl_driver_test: %s (n = %lld, %lld right-hand sides, %lld top fronts, selinv diag max rel diff %.1e)\n", fails ? "FAILED" : "ok",
                (long long)n, (long long)nrhs, (long long)ntop, worst);
    return fails ? 1 : 0;
}
