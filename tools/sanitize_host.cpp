// sanitize_host.cpp -- CPU-only sanitizer job for the host C++ of libgmrfx: the symbolic phase
// (ordering.cpp, symbolic.cpp) spawns threads, gmrfx_api.cpp parses caller arrays. Built twice by
// `make -C gaussianmarkovrandomfields.jl_amd sanitize` (AddressSanitizer + UBSan, ThreadSanitizer) from the SAME
// sources as the product, driven through the C ABI with symbolic_only handles (no GPU is touched; the HIP kernel
// launch wrappers stay unresolved and are never called). Exit code 0 = clean. Run by tests/test_sanitizers.py.
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/gmrfx.h"

// 19-point pattern of (5-point Laplacian)^2 on an nx x ny grid, both triangles, 0-based
static void grid_pattern(int nx, int ny, std::vector<int64_t> &cp, std::vector<int64_t> &ri, std::vector<double> &xy) {
    const int64_t n = (int64_t)nx * ny;
    cp.assign(n + 1, 0);
    ri.clear();
    xy.resize(2 * n);
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
                }
            cp[v + 1] = (int64_t)ri.size();
        }
}

static int fails = 0;
#define EXPECT(c) do { if (!(c)) { std::fprintf(stderr, "FAILED: %s (line %d)\n", #c, __LINE__); fails++; } } while (0)

static void one_handle(const std::vector<int64_t> &cp, const std::vector<int64_t> &ri, const double *coords, int64_t n, int ordering) {
    gmrfx_opts o;
    std::memset(&o, 0, sizeof(o));
    o.struct_size = (int32_t)sizeof(o);
    o.device = -1; o.symbolic_only = 1; o.ordering = ordering;
    if (coords) { o.coord_dim = 2; o.coords = coords; }
    gmrfx_handle *h = nullptr;
    EXPECT(gmrfx_create(n, cp.data(), ri.data(), 0, nullptr, &o, &h) == GMRFX_OK);
    if (!h) return;
    std::vector<int64_t> perm(n);
    EXPECT(gmrfx_get_perm(h, 0, perm.data()) == GMRFX_OK);
    std::vector<char> seen(n, 0);
    for (int64_t k = 0; k < n; k++) { EXPECT(perm[k] >= 0 && perm[k] < n && !seen[perm[k]]); if (perm[k] >= 0 && perm[k] < n) seen[perm[k]] = 1; }
    gmrfx_stats st;
    EXPECT(gmrfx_get_stats(h, &st, (int32_t)sizeof(st)) == GMRFX_OK);
    EXPECT(st.nnz_l >= st.nnz_q_tri && st.nsuper > 0);
    double x = 0;
    EXPECT(gmrfx_logdet(h, &x) == GMRFX_ERR_NO_DEVICE);          // numeric entry points fail loudly without a device
    // a second handle with the first one's permutation (user-permutation path), cloned and destroyed
    gmrfx_handle *h2 = nullptr, *h3 = nullptr;
    EXPECT(gmrfx_create(n, cp.data(), ri.data(), 0, perm.data(), &o, &h2) == GMRFX_OK);
    if (h2) { EXPECT(gmrfx_clone(h2, &h3) == GMRFX_OK); gmrfx_destroy(h3); gmrfx_destroy(h2); }
    gmrfx_destroy(h);
}

// a sharded plan (symbolic only): per-rank layouts, groups of the distributed top fronts, the column-range transfer list.
// Every range must lie inside its child's block, start at whole columns, and join two different ranks.
static void sharded_handles(const std::vector<int64_t> &cp, const std::vector<int64_t> &ri, const double *coords, int64_t n, int world) {
    setenv("GMRFX_DIST_MIN", "128", 1);       // small enough for the test grid's top separators
    for (int rank = 0; rank < world; rank++) {
        gmrfx_opts o;
        std::memset(&o, 0, sizeof(o));
        o.struct_size = (int32_t)sizeof(o);
        o.device = -1; o.symbolic_only = 1; o.ordering = 0; o.shard_rank = rank; o.shard_world = world;
        if (coords) { o.coord_dim = 2; o.coords = coords; }
        gmrfx_handle *h = nullptr;
        EXPECT(gmrfx_create(n, cp.data(), ri.data(), 0, nullptr, &o, &h) == GMRFX_OK);
        if (!h) continue;
        int64_t cnt[4] = {0, 0, 0, 0};
        EXPECT(gmrfx_shard_dist_fronts(h, cnt, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == GMRFX_OK);
        EXPECT(cnt[3] == world && cnt[0] >= 1);
        std::vector<int64_t> front(cnt[0]), cols(cnt[0]), rows(cnt[0]), po(cnt[0]), pl(cnt[0]), lv(cnt[0]), gp(cnt[0] + 1), gr(cnt[1]);
        EXPECT(gmrfx_shard_dist_fronts(h, cnt, front.data(), cols.data(), rows.data(), po.data(), pl.data(), lv.data(), gp.data(), gr.data()) == GMRFX_OK);
        for (int64_t k = 0; k < cnt[0]; k++) {
            EXPECT(gp[k + 1] - gp[k] >= 2 && cols[k] >= 128 && rows[k] >= cols[k] && pl[k] >= rows[k]);
            for (int64_t q = gp[k]; q < gp[k + 1]; q++) EXPECT(gr[q] >= 0 && gr[q] < world && (q == gp[k] || gr[q] > gr[q - 1]));
        }
        std::vector<int64_t> ch(cnt[2]), src(cnt[2]), dst(cnt[2]), lev(cnt[2]), off(cnt[2]), num(cnt[2]), c0(cnt[2]);
        EXPECT(gmrfx_shard_transfers(h, ch.data(), src.data(), dst.data(), lev.data(), off.data(), num.data(), c0.data()) == GMRFX_OK);
        gmrfx_stats st;
        EXPECT(gmrfx_get_stats(h, &st, (int32_t)sizeof(st)) == GMRFX_OK);
        for (int64_t k = 0; k < cnt[2]; k++) {
            EXPECT(src[k] != dst[k] && src[k] >= 0 && src[k] < world && dst[k] >= 0 && dst[k] < world);
            // offsets are per rank: inside this rank's arena when it is an end of the transfer, -1 otherwise
            if (src[k] == rank || dst[k] == rank) EXPECT(off[k] >= 0 && (off[k] + num[k]) * 8.0 <= st.bytes_cb_arena + 1.0);
            else EXPECT(off[k] == -1);
            EXPECT(num[k] > 0 && c0[k] >= 0);
            EXPECT(k == 0 || lev[k] >= lev[k - 1]);
        }
        gmrfx_destroy(h);
    }
    unsetenv("GMRFX_DIST_MIN");
}

int main() {
    std::vector<int64_t> cp, ri;
    std::vector<double> xy;
    // large enough for the threaded nested dissection (> 20 000 vertices per half) and the threaded scatter map (> 2e6 entries)
    grid_pattern(420, 400, cp, ri, xy);
    const int64_t n = 420 * 400;
    one_handle(cp, ri, xy.data(), n, 0);      // geometric nested dissection
    one_handle(cp, ri, nullptr, n, 0);        // graph nested dissection
    one_handle(cp, ri, nullptr, n, 1);        // natural ordering
    sharded_handles(cp, ri, xy.data(), n, 2);
    sharded_handles(cp, ri, xy.data(), n, 4);
    sharded_handles(cp, ri, xy.data(), n, 8);      // the width the driver scales to
    // distinct handles are used concurrently from different host threads (WorkspacePool contract)
    {
        std::vector<int64_t> cp2, ri2; std::vector<double> xy2;
        grid_pattern(90, 70, cp2, ri2, xy2);
        std::vector<std::thread> th;
        th.reserve(4);
        for (int t = 0; t < 4; t++) th.emplace_back([&, t] { one_handle(cp2, ri2, t % 2 ? xy2.data() : nullptr, 90 * 70, 0); });
        for (auto &x : th) x.join();
    }
    // malformed input is rejected, not read out of bounds
    {
        gmrfx_opts o; std::memset(&o, 0, sizeof(o)); o.struct_size = (int32_t)sizeof(o); o.symbolic_only = 1; o.device = -1;
        gmrfx_handle *h = nullptr;
        std::vector<int64_t> bad = cp; bad[5] = bad[4] - 3;
        EXPECT(gmrfx_create(n, bad.data(), ri.data(), 0, nullptr, &o, &h) != GMRFX_OK && h == nullptr);
        std::vector<int64_t> badr = ri; badr[10] = n + 7;
        EXPECT(gmrfx_create(n, cp.data(), badr.data(), 0, nullptr, &o, &h) != GMRFX_OK && h == nullptr);
        std::vector<int64_t> p(n, 0);
        EXPECT(gmrfx_create(n, cp.data(), ri.data(), 0, p.data(), &o, &h) != GMRFX_OK && h == nullptr);
    }
    std::printf("sanitize_host: %s\n", fails ? "FAILED" : "ok");
    return fails ? 1 : 0;
}
