// rccl_driver.cpp -- libgmrfx_rccl.so: the native driver of the sharded protocol over RCCL (include/gmrfx_rccl.h).
//
// Uses ONLY the public C ABI of libgmrfx.so (include/gmrfx.h), the HIP runtime and rccl.h: it is what a host language would write
// itself, kept as a library so that the Julia plug-in -- or anything else with a `ccall` -- reaches the
// multi-GPU split without Python. The sequence is gmrfx/shard.py's, exchange by exchange; the two are compared bit for bit against the
// unsharded handle by the same kind of test (tools/rccl_driver_test.cpp on a one-rank communicator; tests/test_rccl_world1.py).
//
// One stream orders everything: a phase, the ncclSend / ncclRecv group behind it and the next phase are enqueued back to back; the
// host blocks only where a value comes back. (gmrfx/shard.py gets its communication stream from torch and overlaps the look-ahead
// broadcast of a distributed front with the K = 256 updates beside it; here that broadcast is in stream order too -- the simple form.)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/gmrfx_rccl.h"

namespace {

struct Item { int src, dst; int64_t off, cnt; };            // `cnt` doubles at `off` doubles of a library buffer go src -> dst
struct Block { int owner; int64_t row0, nrows; };           // rows of X (elimination order)
struct DistFront { int s; int64_t cols, rows, poff, ld; int level; std::vector<int> group; };

}  // namespace

struct gmrfx_rccl {
    gmrfx_handle *h = nullptr;
    int world = 1, rank = 0;
    ncclComm_t comm = nullptr;
    hipStream_t st = nullptr;
    bool own_stream = false;
    int64_t n = 0, L0 = 0, K = 0, nl = 0;
    std::vector<std::vector<Item>> cb_items, w_items;       // per top level
    std::vector<std::vector<Block>> top_blocks;             // per top level
    std::vector<Block> sub_blocks;
    std::vector<std::vector<int>> dist_of_level;
    std::vector<DistFront> dist;
    // selected inversion: edges (parent's owner -> child's owner) by the child's level
    struct ZEdge { int from, to; int64_t off, cnt; int child_level; };
    std::vector<ZEdge> zedges;
    double *d_scal = nullptr;                                // 2 doubles + 2 int64 of scratch for the all-reduces
    std::string err;
    bool info_pending = false;
};

namespace {

struct Fail : std::runtime_error { using std::runtime_error::runtime_error; };

void ck(int32_t code, gmrfx_handle *h, const char *what) {
    if (code != GMRFX_OK) throw Fail(std::string(what) + ": " + (h ? gmrfx_last_error(h) : "gmrfx error"));
}
void ck(hipError_t e, const char *what) {
    if (e != hipSuccess) throw Fail(std::string(what) + ": " + hipGetErrorString(e));
}
void ck(ncclResult_t e, const char *what) {
    if (e != ncclSuccess) throw Fail(std::string(what) + ": " + ncclGetErrorString(e));
}

double *buf(gmrfx_rccl *d, int which) {
    void *p = gmrfx_device_ptr(d->h, which);
    if (!p) throw Fail("gmrfx_device_ptr returned null");
    return (double *)p;
}

// every rank passes the same list; the transfers this rank takes part in go out as ONE group
void p2p(gmrfx_rccl *d, int which, const std::vector<Item> &items, int64_t scale = 1) {
    bool any = false;
    for (const Item &it : items) any = any || (it.src != it.dst && it.cnt > 0 && (it.src == d->rank || it.dst == d->rank));
    if (!any) return;
    double *base = buf(d, which);
    ck(ncclGroupStart(), "ncclGroupStart");
    for (const Item &it : items) {
        if (it.src == it.dst || it.cnt <= 0) continue;
        if (it.src == d->rank) ck(ncclSend(base + it.off * scale, (size_t)(it.cnt * scale), ncclDouble, it.dst, d->comm, d->st), "ncclSend");
        else if (it.dst == d->rank) ck(ncclRecv(base + it.off * scale, (size_t)(it.cnt * scale), ncclDouble, it.src, d->comm, d->st), "ncclRecv");
    }
    ck(ncclGroupEnd(), "ncclGroupEnd");
}

// `cnt` doubles at `ptr` from `root` to every member of `group` (sorted ranks), in place
void bcast_group(gmrfx_rccl *d, double *ptr, int64_t cnt, int root, const std::vector<int> &group) {
    if ((int)group.size() == d->world) { ck(ncclBroadcast(ptr, ptr, (size_t)cnt, ncclDouble, root, d->comm, d->st), "ncclBroadcast"); return; }
    if (std::find(group.begin(), group.end(), d->rank) == group.end()) return;
    ck(ncclGroupStart(), "ncclGroupStart");
    for (int m : group) {
        if (m == root) continue;
        if (d->rank == root) ck(ncclSend(ptr, (size_t)cnt, ncclDouble, m, d->comm, d->st), "ncclSend");
        else if (d->rank == m) ck(ncclRecv(ptr, (size_t)cnt, ncclDouble, root, d->comm, d->st), "ncclRecv");
    }
    ck(ncclGroupEnd(), "ncclGroupEnd");
}

void factor_distributed_front(gmrfx_rccl *d, const double *d_nz, const DistFront &f) {
    if (std::find(f.group.begin(), f.group.end(), d->rank) == f.group.end()) return;
    gmrfx_handle *h = d->h;
    const int g = (int)f.group.size();
    const int64_t nb = (f.cols + 255) / 256;
    double *panels = buf(d, 1);
    // where THIS rank keeps block b: the whole panel, or its own blocks + a window of two received ones (gmrfx_dist_front_block)
    auto view = [&](int64_t b, double *&p, int64_t &cnt) {
        int64_t off = -1;
        ck(gmrfx_dist_front_block(h, f.s, (int32_t)b, &off, &cnt), h, "gmrfx_dist_front_block");
        if (off < 0) throw Fail("gmrfx_dist_front_block: not a member of the front's group");
        p = panels + off;
    };
    ck(gmrfx_dist_front_phase(h, d_nz, f.s, 0, 0), h, "dist front: assemble");
    ck(gmrfx_dist_front_phase(h, d_nz, f.s, 1, 0), h, "dist front: factor block 0");
    double *p; int64_t cnt;
    view(0, p, cnt);
    bcast_group(d, p, cnt, f.group[0], f.group);
    for (int64_t b = 0; b < nb; b++) {
        if (b + 1 < nb) {
            ck(gmrfx_dist_front_phase(h, d_nz, f.s, 4, (int32_t)b), h, "dist front: apply to the next block");
            ck(gmrfx_dist_front_phase(h, d_nz, f.s, 1, (int32_t)(b + 1)), h, "dist front: factor block");
            view(b + 1, p, cnt);
            bcast_group(d, p, cnt, f.group[(size_t)((b + 1) % g)], f.group);
            ck(gmrfx_dist_front_phase(h, d_nz, f.s, 5, (int32_t)b), h, "dist front: apply to the rest");
        }
    }
    ck(gmrfx_dist_front_phase(h, d_nz, f.s, 3, 0), h, "dist front: contribution block");
}

void solve_pass(gmrfx_rccl *d, const double *d_B, int64_t ldb, int64_t nr, double *d_X, int64_t ldx, bool backward_only, bool gather) {
    gmrfx_handle *h = d->h;
    auto sp = [&](int phase) { ck(gmrfx_solve_phase(h, d_B, ldb, nr, d_X, ldx, phase), h, "gmrfx_solve_phase"); };
    if (backward_only) sp(10);
    else {
        sp(0);
        for (int64_t k = 0; k < d->K; k++) {
            p2p(d, 3, d->w_items[(size_t)k], nr);
            sp(100 + (int)k);
        }
    }
    for (int64_t k = d->K - 1; k >= 0; k--) {
        sp((backward_only ? 300 : 200) + (int)k);
        const auto &blocks = d->top_blocks[(size_t)k];
        if (!blocks.empty() && d->world > 1) {
            double *X = buf(d, 2);
            ck(ncclGroupStart(), "ncclGroupStart");
            for (const Block &b : blocks)
                ck(ncclBroadcast(X + b.row0 * nr, X + b.row0 * nr, (size_t)(b.nrows * nr), ncclDouble, b.owner, d->comm, d->st), "ncclBroadcast");
            ck(ncclGroupEnd(), "ncclGroupEnd");
        } else if (!blocks.empty()) {       // one rank: the collective still runs (in place, root = self)
            double *X = buf(d, 2);
            for (const Block &b : blocks)
                ck(ncclBroadcast(X + b.row0 * nr, X + b.row0 * nr, (size_t)(b.nrows * nr), ncclDouble, b.owner, d->comm, d->st), "ncclBroadcast");
        }
    }
    sp(backward_only ? 12 : 2);
    if (!gather) { sp(3); return; }
    std::vector<Item> home;
    for (const Block &b : d->sub_blocks) home.push_back({b.owner, 0, b.row0, b.nrows});
    p2p(d, 2, home, nr);
    if (d->rank == 0) sp(3);
}

template <class F> int32_t guarded(gmrfx_rccl *d, F &&f) {
    if (!d) return GMRFX_ERR_INVALID_ARG;
    try {
        f();
        return GMRFX_OK;
    } catch (const Fail &e) {
        d->err = e.what();
        return GMRFX_ERR_HIP;
    } catch (const std::exception &e) {
        d->err = e.what();
        return GMRFX_ERR_INVALID_ARG;
    }
}

}  // namespace

extern "C" int32_t gmrfx_rccl_unique_id(void *id128) {
    if (!id128) return GMRFX_ERR_INVALID_ARG;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return GMRFX_ERR_HIP;
    std::memcpy(id128, &id, sizeof(id));
    return GMRFX_OK;
}

extern "C" const char *gmrfx_rccl_last_error(const gmrfx_rccl *d) { return d ? d->err.c_str() : "null driver"; }

extern "C" int32_t gmrfx_rccl_create(gmrfx_handle *h, int32_t world, int32_t rank, const void *id128, void *hip_stream, gmrfx_rccl **out) {
    if (!h || !out || !id128 || world < 1 || rank < 0 || rank >= world) return GMRFX_ERR_INVALID_ARG;
    *out = nullptr;
    gmrfx_rccl *d = new gmrfx_rccl();
    d->h = h; d->world = world; d->rank = rank;
    const int32_t rc = guarded(d, [&] {
        gmrfx_stats st;
        ck(gmrfx_get_stats(h, &st, (int32_t)sizeof(st)), h, "gmrfx_get_stats");
        d->n = st.n; d->nl = st.nlevels;
        int64_t ne = 0, ntop = 0, sl = 0;
        ck(gmrfx_shard_info(h, &ne, &ntop, &sl), h, "gmrfx_shard_info");
        d->L0 = sl; d->K = d->nl - sl;
        if (d->K < 0) throw Fail("not a sharded handle");
        // the stream: the caller's, or one of the driver's own; the handle runs on it with asynchronous phases
        if (hip_stream) d->st = (hipStream_t)hip_stream;
        else { ck(hipStreamCreateWithFlags(&d->st, hipStreamNonBlocking), "hipStreamCreate"); d->own_stream = true; }
        ck(gmrfx_set_stream(h, (void *)d->st, 1, 1), h, "gmrfx_set_stream");
        ncclUniqueId id;
        std::memcpy(&id, id128, sizeof(id));
        ck(ncclCommInitRank(&d->comm, world, id, rank), "ncclCommInitRank");
        ck(hipMalloc((void **)&d->d_scal, 64), "hipMalloc");
        // cross-rank edges: update vectors of the forward sweep per top level; trailing inverse blocks of the selected inversion
        d->cb_items.assign((size_t)d->K, {}); d->w_items.assign((size_t)d->K, {}); d->top_blocks.assign((size_t)d->K, {});
        d->dist_of_level.assign((size_t)d->K, {});
        if (ne > 0) {
            std::vector<int64_t> child(ne), src(ne), dst(ne), lev(ne), cbo(ne), cbc(ne), w0(ne), wn(ne), zbo(ne), cl(ne);
            ck(gmrfx_shard_edges(h, child.data(), src.data(), dst.data(), lev.data(), cbo.data(), cbc.data(), w0.data(), wn.data(), zbo.data(), cl.data()),
               h, "gmrfx_shard_edges");
            for (int64_t k = 0; k < ne; k++) {
                const int64_t t = lev[k] - d->L0;
                if (t >= 0 && t < d->K) d->w_items[(size_t)t].push_back({(int)src[k], (int)dst[k], w0[k], wn[k]});
                d->zedges.push_back({(int)dst[k], (int)src[k], zbo[k], cbc[k], (int)cl[k]});
            }
        }
        // contribution-block transfers (column ranges) + the distributed fronts
        int64_t cnt[4] = {0, 0, 0, 0};
        ck(gmrfx_shard_dist_fronts(h, cnt, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr), h, "gmrfx_shard_dist_fronts");
        if (cnt[0] > 0) {
            const int64_t nf = cnt[0];
            std::vector<int64_t> fr(nf), co(nf), ro(nf), po(nf), pl(nf), lv(nf), gp(nf + 1), gr((size_t)std::max<int64_t>(cnt[1], 1));
            ck(gmrfx_shard_dist_fronts(h, cnt, fr.data(), co.data(), ro.data(), po.data(), pl.data(), lv.data(), gp.data(), gr.data()), h, "gmrfx_shard_dist_fronts");
            for (int64_t i = 0; i < nf; i++) {
                DistFront f{(int)fr[i], co[i], ro[i], po[i], pl[i], (int)lv[i], {}};
                for (int64_t q = gp[i]; q < gp[i + 1]; q++) f.group.push_back((int)gr[q]);
                const int64_t t = lv[i] - d->L0;
                if (t >= 0 && t < d->K) d->dist_of_level[(size_t)t].push_back((int)d->dist.size());
                d->dist.push_back(std::move(f));
            }
        }
        if (cnt[2] > 0) {
            const int64_t nx = cnt[2];
            std::vector<int64_t> ch(nx), src(nx), dst(nx), lev(nx), off(nx), num(nx), c0(nx);
            ck(gmrfx_shard_transfers(h, ch.data(), src.data(), dst.data(), lev.data(), off.data(), num.data(), c0.data()), h, "gmrfx_shard_transfers");
            for (int64_t k = 0; k < nx; k++) {
                const int64_t t = lev[k] - d->L0;
                if (t >= 0 && t < d->K) d->cb_items[(size_t)t].push_back({(int)src[k], (int)dst[k], off[k], num[k]});
            }
        }
        for (int kind = 2; kind <= 3; kind++) {
            int64_t nb = 0;
            ck(gmrfx_shard_rows(h, kind, &nb, nullptr, nullptr, nullptr, nullptr), h, "gmrfx_shard_rows");
            if (nb <= 0) continue;
            std::vector<int64_t> ow(nb), r0(nb), nr(nb), lv(nb);
            ck(gmrfx_shard_rows(h, kind, &nb, ow.data(), r0.data(), nr.data(), lv.data()), h, "gmrfx_shard_rows");
            for (int64_t k = 0; k < nb; k++) {
                if (kind == 3) d->sub_blocks.push_back({(int)ow[k], r0[k], nr[k]});
                else {
                    const int64_t t = lv[k] - d->L0;
                    if (t >= 0 && t < d->K) d->top_blocks[(size_t)t].push_back({(int)ow[k], r0[k], nr[k]});
                }
            }
        }
    });
    if (rc != GMRFX_OK) {       // keep the message reachable: hand the half-built driver back only on success
        static thread_local std::string last;
        last = d->err;
        gmrfx_rccl_destroy(d);
        return rc;
    }
    *out = d;
    return GMRFX_OK;
}

extern "C" void gmrfx_rccl_destroy(gmrfx_rccl *d) {
    if (!d) return;
    if (d->st) (void)hipStreamSynchronize(d->st);
    if (d->h && d->st) (void)gmrfx_set_stream(d->h, nullptr, 0, 0);          // the handle goes back to its own stream
    if (d->comm) (void)ncclCommDestroy(d->comm);
    if (d->d_scal) (void)hipFree(d->d_scal);
    if (d->own_stream && d->st) (void)hipStreamDestroy(d->st);
    delete d;
}

extern "C" int32_t gmrfx_rccl_refactorize(gmrfx_rccl *d, const double *d_nzval) {
    return guarded(d, [&] {
        if (!d_nzval) throw std::invalid_argument("d_nzval is null");
        gmrfx_handle *h = d->h;
        ck(gmrfx_refactorize_phase(h, d_nzval, 0), h, "gmrfx_refactorize_phase(0)");
        for (int64_t k = 0; k < d->K; k++) {
            p2p(d, 0, d->cb_items[(size_t)k]);
            for (int i : d->dist_of_level[(size_t)k]) factor_distributed_front(d, d_nzval, d->dist[(size_t)i]);
            ck(gmrfx_refactorize_phase(h, d_nzval, 1 + (int32_t)k), h, "gmrfx_refactorize_phase");
        }
        d->info_pending = true;
    });
}

extern "C" int32_t gmrfx_rccl_solve(gmrfx_rccl *d, const double *d_B, int64_t ldb, int64_t nrhs, double *d_X, int64_t ldx, int32_t gather) {
    return guarded(d, [&] {
        if (!d_B || !d_X || nrhs < 0) throw std::invalid_argument("bad arguments");
        for (int64_t j0 = 0; j0 < nrhs; j0 += 64)
            solve_pass(d, d_B + j0 * ldb, ldb, std::min<int64_t>(64, nrhs - j0), d_X + j0 * ldx, ldx, false, gather != 0);
    });
}

extern "C" int32_t gmrfx_rccl_backward_solve(gmrfx_rccl *d, const double *d_Z, int64_t ldz, int64_t nrhs, double *d_X, int64_t ldx, int32_t gather) {
    return guarded(d, [&] {
        if (!d_Z || !d_X || nrhs < 0) throw std::invalid_argument("bad arguments");
        for (int64_t j0 = 0; j0 < nrhs; j0 += 64)
            solve_pass(d, d_Z + j0 * ldz, ldz, std::min<int64_t>(64, nrhs - j0), d_X + j0 * ldx, ldx, true, gather != 0);
    });
}

extern "C" int32_t gmrfx_rccl_logdet(gmrfx_rccl *d, double *logdet, int64_t *info) {
    return guarded(d, [&] {
        if (!logdet) throw std::invalid_argument("logdet is null");
        double part = 0.0;
        ck(gmrfx_logdet_partial(d->h, &part), d->h, "gmrfx_logdet_partial");          // (this rank's own pivots; synchronises)
        gmrfx_stats st;
        ck(gmrfx_get_stats(d->h, &st, (int32_t)sizeof(st)), d->h, "gmrfx_get_stats");
        long long fc = st.fail_col >= 0 ? (long long)st.fail_col : (1ll << 62);
        ck(hipMemcpyAsync(d->d_scal, &part, sizeof(double), hipMemcpyHostToDevice, d->st), "hipMemcpy");
        ck(hipMemcpyAsync(d->d_scal + 2, &fc, sizeof(long long), hipMemcpyHostToDevice, d->st), "hipMemcpy");
        ck(ncclAllReduce(d->d_scal, d->d_scal, 1, ncclDouble, ncclSum, d->comm, d->st), "ncclAllReduce(sum)");
        ck(ncclAllReduce(d->d_scal + 2, d->d_scal + 2, 1, ncclInt64, ncclMin, d->comm, d->st), "ncclAllReduce(min)");
        ck(hipMemcpyAsync(&part, d->d_scal, sizeof(double), hipMemcpyDeviceToHost, d->st), "hipMemcpy");
        ck(hipMemcpyAsync(&fc, d->d_scal + 2, sizeof(long long), hipMemcpyDeviceToHost, d->st), "hipMemcpy");
        ck(hipStreamSynchronize(d->st), "hipStreamSynchronize");
        *logdet = part;
        if (info) *info = fc >= (1ll << 62) ? 0 : (int64_t)fc + 1;
        d->info_pending = false;
    });
}

extern "C" int32_t gmrfx_rccl_selinv_diag(gmrfx_rccl *d, double *out_host) {
    return guarded(d, [&] {
        if (!out_host) throw std::invalid_argument("out is null");
        gmrfx_handle *h = d->h;
        auto sel = [&](int what, int hi, int lo) { ck(gmrfx_selinv_phase(h, what, hi, lo), h, "gmrfx_selinv_phase"); };
        sel(0, 0, 0);
        std::vector<int> cut;
        for (const auto &e : d->zedges) cut.push_back(e.child_level);
        std::sort(cut.begin(), cut.end(), std::greater<int>());
        cut.erase(std::unique(cut.begin(), cut.end()), cut.end());
        int hi = (int)d->nl;
        for (int l : cut) {
            if (hi > l + 1) sel(2, hi, l + 1);              // my fronts of the levels above l
            sel(1, l, 0);                                    // gather for the other ranks' fronts of level l
            std::vector<Item> items;
            for (const auto &e : d->zedges) if (e.child_level == l) items.push_back({e.from, e.to, e.off, e.cnt});
            p2p(d, 0, items);
            sel(2, l + 1, l);
            hi = l;
        }
        if (hi > 0) sel(2, hi, 0);
        sel(3, 0, 0);
        ck(hipStreamSynchronize(d->st), "hipStreamSynchronize");
        ck(gmrfx_selinv_diag(h, out_host), h, "gmrfx_selinv_diag");                   // this rank's part (zeros elsewhere)
        double *tmp = nullptr;
        ck(hipMalloc((void **)&tmp, (size_t)d->n * sizeof(double)), "hipMalloc");
        try {
            ck(hipMemcpyAsync(tmp, out_host, (size_t)d->n * sizeof(double), hipMemcpyHostToDevice, d->st), "hipMemcpy");
            ck(ncclAllReduce(tmp, tmp, (size_t)d->n, ncclDouble, ncclSum, d->comm, d->st), "ncclAllReduce");
            ck(hipMemcpyAsync(out_host, tmp, (size_t)d->n * sizeof(double), hipMemcpyDeviceToHost, d->st), "hipMemcpy");
            ck(hipStreamSynchronize(d->st), "hipStreamSynchronize");
        } catch (...) { (void)hipFree(tmp); throw; }
        (void)hipFree(tmp);
    });
}

extern "C" int32_t gmrfx_rccl_needed_rows(const gmrfx_rccl *dc, uint8_t *mask) {
    gmrfx_rccl *d = const_cast<gmrfx_rccl *>(dc);
    return guarded(d, [&] {
        if (!mask) throw std::invalid_argument("mask is null");
        std::vector<uint8_t> elim((size_t)d->n, 0);
        for (const Block &b : d->sub_blocks) if (b.owner == d->rank) std::fill(elim.begin() + b.row0, elim.begin() + b.row0 + b.nrows, 1);
        for (const auto &lv : d->top_blocks)
            for (const Block &b : lv) if (b.owner == d->rank) std::fill(elim.begin() + b.row0, elim.begin() + b.row0 + b.nrows, 1);
        std::vector<int64_t> perm((size_t)d->n);
        ck(gmrfx_get_perm(d->h, 0, perm.data()), d->h, "gmrfx_get_perm");
        std::memset(mask, 0, (size_t)d->n);
        for (int64_t k = 0; k < d->n; k++) mask[perm[(size_t)k]] = elim[(size_t)k];
    });
}
