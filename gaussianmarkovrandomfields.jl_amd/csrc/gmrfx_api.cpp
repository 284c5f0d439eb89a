// gmrfx_api.cpp -- the extern "C" boundary declared in include/gmrfx.h.
#include "../../include/gmrfx.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "device.h"
#include "symbolic.h"

using namespace gmrfx;

struct gmrfx_handle {
    Symbolic S;
    std::unique_ptr<Device> D;   // null for symbolic_only handles
    gmrfx_opts opts{};
    std::string err;
    // lazily built pattern of the de-permuted selected inverse
    bool zpat_built = false;
    std::vector<i64> zcolptr, zrow, zoff;
};

static thread_local std::string g_create_err;

static gmrfx_opts default_opts() {
    gmrfx_opts o;
    std::memset(&o, 0, sizeof(o));
    o.struct_size = (int32_t)sizeof(gmrfx_opts);
    o.device = -1;
    return o;
}

extern "C" const char *gmrfx_last_create_error(void) { return g_create_err.c_str(); }
extern "C" const char *gmrfx_last_error(const gmrfx_handle *h) { return h ? h->err.c_str() : "null handle"; }

template <class F> static int32_t guarded(gmrfx_handle *h, F &&f) {
    if (!h) return GMRFX_ERR_INVALID_ARG;
    try {
        return f();
    } catch (const std::invalid_argument &e) {
        h->err = e.what();
        return GMRFX_ERR_INVALID_ARG;
    } catch (const std::bad_alloc &) {
        h->err = "out of host memory";
        return GMRFX_ERR_ALLOC;
    } catch (const std::exception &e) {
        h->err = e.what();
        return GMRFX_ERR_HIP;
    }
}

static int32_t need_device(gmrfx_handle *h, bool need_factor) {
    if (!h->D) { h->err = "handle has no device state (symbolic_only or no HIP device): numeric entry points are GPU-only"; return GMRFX_ERR_NO_DEVICE; }
    if (need_factor && !h->D->factorized) { h->err = "gmrfx_refactorize has not been called"; return GMRFX_ERR_NOT_FACTORIZED; }
    return GMRFX_OK;
}

extern "C" int32_t gmrfx_create(int64_t n, const int64_t *colptr, const int64_t *rowval, int32_t index_base,
                                const int64_t *perm, const gmrfx_opts *opts, gmrfx_handle **out) {
    if (!out) { g_create_err = "out is null"; return GMRFX_ERR_INVALID_ARG; }
    *out = nullptr;
    if (!colptr || !rowval) { g_create_err = "colptr/rowval is null"; return GMRFX_ERR_INVALID_ARG; }
    std::unique_ptr<gmrfx_handle> h(new gmrfx_handle());
    h->opts = default_opts();
    if (opts) {
        size_t sz = std::min<size_t>((size_t)opts->struct_size, sizeof(gmrfx_opts));
        if (opts->struct_size <= 0) { g_create_err = "opts.struct_size not set"; return GMRFX_ERR_INVALID_ARG; }
        std::memcpy(&h->opts, opts, sz);
    }
    try {
        SymOptions so;
        so.uplo = h->opts.uplo;
        so.ordering = h->opts.ordering;
        so.nd_leaf = h->opts.nd_leaf;
        so.relax_cols = h->opts.relax_cols;
        so.relax_zeros = h->opts.relax_zeros;
        so.coord_dim = h->opts.coords ? h->opts.coord_dim : 0;
        so.coords = h->opts.coords;
        if (so.coords && so.coord_dim != 2 && so.coord_dim != 3) throw std::invalid_argument("coord_dim must be 2 or 3");
        if (const char *e = std::getenv("GMRFX_SMALL_ROWS")) so.small_front_rows = std::atoi(e);   // tuning/testing knob
        if (const char *e = std::getenv("GMRFX_SUBTREE_MAX")) so.subtree_max = std::atoi(e);       // 0 disables subtree tasks
        if (const char *e = std::getenv("GMRFX_SWEEP_TASK_ROWS")) so.sweep_task_rows = std::atoi(e);   // 0 disables sweep tasks
        if (const char *e = std::getenv("GMRFX_MERGE_WIDE")) so.merge_wide = std::atoi(e);   // widest child with siblings that may still be merged into its parent
        if (const char *e = std::getenv("GMRFX_TOP_BY_DEPTH")) so.top_by_depth = std::atoi(e);   // top levels levelled by depth below the root (0: none)
        if (h->opts.shard_world > 1 || (h->opts.shard_world == 1 && h->opts.shard_min_top > 0)) {
            so.shard_min_top = std::max(0, h->opts.shard_min_top);
            if (h->opts.shard_rank < 0 || h->opts.shard_rank >= h->opts.shard_world) throw std::invalid_argument("shard_rank out of range");
            so.shard_rank = h->opts.shard_rank;
            so.shard_world = h->opts.shard_world;
            if (const char *e = std::getenv("GMRFX_DIST_MIN")) so.dist_min_cols = std::atoi(e);   // columns from which a top front is factored by its whole group (0: never)
            so.subtree_max = 0;     // subtree tasks are not shard-aware
        }
        analyze(n, colptr, rowval, index_base, perm, so, h->S);
        h->opts.coords = nullptr;  // caller-owned, not kept
    } catch (const std::invalid_argument &e) {
        g_create_err = e.what();
        return GMRFX_ERR_INVALID_ARG;
    } catch (const std::bad_alloc &) {
        g_create_err = "out of host memory";
        return GMRFX_ERR_ALLOC;
    } catch (const std::exception &e) {
        g_create_err = e.what();
        return GMRFX_ERR_INVALID_ARG;
    }
    if (!h->opts.symbolic_only) {
        try {
            h->D.reset(new Device());
            h->D->init(h->S, h->opts.device);
        } catch (const std::exception &e) {
            g_create_err = e.what();
            return std::string(e.what()).find("no HIP device") != std::string::npos ? GMRFX_ERR_NO_DEVICE : GMRFX_ERR_HIP;
        }
    }
    *out = h.release();
    return GMRFX_OK;
}

extern "C" void gmrfx_destroy(gmrfx_handle *h) { delete h; }

extern "C" int32_t gmrfx_clone(const gmrfx_handle *h, gmrfx_handle **out) {
    if (!h || !out) return GMRFX_ERR_INVALID_ARG;
    *out = nullptr;
    std::unique_ptr<gmrfx_handle> c(new gmrfx_handle());
    try {
        c->S = h->S;
        c->opts = h->opts;
        if (h->D) {
            c->D.reset(new Device());
            c->D->clone_from(*h->D, c->S);
        }
    } catch (const std::exception &e) {
        g_create_err = e.what();
        return GMRFX_ERR_HIP;
    }
    *out = c.release();
    return GMRFX_OK;
}

static int32_t refactorize_impl(gmrfx_handle *h, const double *nz, int64_t *info, bool dev) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, false)) return e;
        if (!nz) throw std::invalid_argument("nzval is null");
        h->D->refactorize(nz, dev);
        long long fc = h->D->fail_col();
        if (info) *info = fc < 0 ? 0 : fc + 1;
        if (fc >= 0 && h->opts.check_posdef) {
            h->err = "matrix is not positive definite (non-positive pivot at elimination step " + std::to_string(fc + 1) + ")";
            return GMRFX_ERR_NOT_POSDEF;
        }
        return GMRFX_OK;
    });
}
extern "C" int32_t gmrfx_refactorize(gmrfx_handle *h, const double *nzval, int64_t *info) { return refactorize_impl(h, nzval, info, false); }

// workspace_solve with a stale factorisation (src/workspace/gmrf_workspace.jl:170-178, 207-215: ensure_numeric! -> refactorize!,
// then backend_solve) as one pipelined call: Device::refactorize_solve. X is only meaningful when *info == 0.
static int32_t refactorize_solve_impl(gmrfx_handle *h, const double *nz, const double *B, int64_t ldb, int64_t nrhs, double *X, int64_t ldx,
                                      int64_t *info, bool dev) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, false)) return e;
        if (!nz) throw std::invalid_argument("nzval is null");
        if (nrhs < 0) throw std::invalid_argument("nrhs < 0");
        if (nrhs > 0 && (!B || !X)) throw std::invalid_argument("B/X is null");
        if (nrhs > 0 && (ldb < h->S.n || ldx < h->S.n)) throw std::invalid_argument("leading dimension smaller than n");
        h->D->refactorize_solve(nz, dev, B, ldb, nrhs, X, ldx, dev);
        long long fc = h->D->fail_col();
        if (info) *info = fc < 0 ? 0 : fc + 1;
        if (fc >= 0 && h->opts.check_posdef) {
            h->err = "matrix is not positive definite (non-positive pivot at elimination step " + std::to_string(fc + 1) + ")";
            return GMRFX_ERR_NOT_POSDEF;
        }
        return GMRFX_OK;
    });
}
extern "C" int32_t gmrfx_refactorize_solve(gmrfx_handle *h, const double *nzval, const double *B, int64_t ldb, int64_t nrhs, double *X,
                                           int64_t ldx, int64_t *info) { return refactorize_solve_impl(h, nzval, B, ldb, nrhs, X, ldx, info, false); }
extern "C" int32_t gmrfx_refactorize_solve_dev(gmrfx_handle *h, const double *d_nzval, const double *d_B, int64_t ldb, int64_t nrhs, double *d_X,
                                               int64_t ldx, int64_t *info) { return refactorize_solve_impl(h, d_nzval, d_B, ldb, nrhs, d_X, ldx, info, true); }

// ---- Newton loop on the device -----------------------------------------------------------
extern "C" int32_t gmrfx_set_prior(gmrfx_handle *h, const double *prior_nzval, const int64_t *map, int64_t cnt, int32_t index_base) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, false)) return e;
        if (!prior_nzval || (cnt > 0 && !map) || cnt < 0) throw std::invalid_argument("prior_nzval / map is null");
        std::vector<long long> m((size_t)cnt);
        for (int64_t k = 0; k < cnt; k++) m[k] = map[k] - index_base;
        h->D->set_prior(prior_nzval, m.data(), cnt);
        return GMRFX_OK;
    });
}
static int32_t refactorize_update_impl(gmrfx_handle *h, const double *hv, int64_t *info, bool dev) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, false)) return e;
        h->D->refactorize_update(hv, dev);
        long long fc = h->D->fail_col();
        if (info) *info = fc < 0 ? 0 : fc + 1;
        if (fc >= 0 && h->opts.check_posdef) {
            h->err = "matrix is not positive definite (non-positive pivot at elimination step " + std::to_string(fc + 1) + ")";
            return GMRFX_ERR_NOT_POSDEF;
        }
        return GMRFX_OK;
    });
}
// One logpdf evaluation of the hyper-parameter loop in one call (Device::refactorize_logpdf): device pointers.
extern "C" int32_t gmrfx_refactorize_logpdf_dev(gmrfx_handle *h, const double *d_nzval, const double *d_X, int64_t ldx, int64_t nvec,
                                                const double *d_mu, double *quad, double *logdet, int64_t *info) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, false)) return e;
        if (!d_nzval) throw std::invalid_argument("nzval is null");
        if (nvec > 0 && (!d_X || !quad)) throw std::invalid_argument("X / quad is null");
        h->D->refactorize_logpdf(d_nzval, d_X, ldx, nvec, d_mu, quad, logdet);
        long long fc = h->D->fail_col();
        if (info) *info = fc < 0 ? 0 : fc + 1;
        if (fc >= 0 && h->opts.check_posdef) {
            h->err = "matrix is not positive definite (non-positive pivot at elimination step " + std::to_string(fc + 1) + ")";
            return GMRFX_ERR_NOT_POSDEF;
        }
        return GMRFX_OK;
    });
}
// One Newton iterate in one pipelined call (Device::refactorize_update_solve): Hessian values in, new mean's solve out.
static int32_t refactorize_update_solve_impl(gmrfx_handle *h, const double *hv, const double *B, int64_t ldb, int64_t nrhs, double *X, int64_t ldx,
                                             int64_t *info, bool dev) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, false)) return e;
        if (nrhs < 0) throw std::invalid_argument("nrhs < 0");
        if (nrhs > 0 && (!B || !X)) throw std::invalid_argument("B/X is null");
        if (nrhs > 0 && (ldb < h->S.n || ldx < h->S.n)) throw std::invalid_argument("leading dimension smaller than n");
        h->D->refactorize_update_solve(hv, dev, B, ldb, nrhs, X, ldx, dev);
        long long fc = h->D->fail_col();
        if (info) *info = fc < 0 ? 0 : fc + 1;
        if (fc >= 0 && h->opts.check_posdef) {
            h->err = "matrix is not positive definite (non-positive pivot at elimination step " + std::to_string(fc + 1) + ")";
            return GMRFX_ERR_NOT_POSDEF;
        }
        return GMRFX_OK;
    });
}
extern "C" int32_t gmrfx_refactorize_update_solve(gmrfx_handle *h, const double *hvals, const double *B, int64_t ldb, int64_t nrhs, double *X,
                                                  int64_t ldx, int64_t *info) { return refactorize_update_solve_impl(h, hvals, B, ldb, nrhs, X, ldx, info, false); }
extern "C" int32_t gmrfx_refactorize_update_solve_dev(gmrfx_handle *h, const double *d_hvals, const double *d_B, int64_t ldb, int64_t nrhs,
                                                      double *d_X, int64_t ldx, int64_t *info) { return refactorize_update_solve_impl(h, d_hvals, d_B, ldb, nrhs, d_X, ldx, info, true); }
extern "C" int32_t gmrfx_refactorize_update(gmrfx_handle *h, const double *hvals, int64_t *info) { return refactorize_update_impl(h, hvals, info, false); }
extern "C" int32_t gmrfx_refactorize_update_dev(gmrfx_handle *h, const double *d_hvals, int64_t *info) { return refactorize_update_impl(h, d_hvals, info, true); }

// ---- sharded factorisation (include/gmrfx.h) ---------------------------------------------------------
extern "C" int32_t gmrfx_refactorize_phase(gmrfx_handle *h, const double *d_nzval, int32_t phase) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, false)) return e;
        if (phase == 0 && !d_nzval) throw std::invalid_argument("d_nzval is null");
        if (phase < 0 || phase > h->S.nlevels - h->S.shard_level) throw std::invalid_argument("phase must be 0 (own subtrees) or 1 + k (top level k)");
        h->D->refactorize_phase(d_nzval, phase);
        return GMRFX_OK;
    });
}
// Profiling aid (handles created with GMRFX_LEVEL_MARK=1): HIP-event time of every tree level of the most recent
// factorisation (which = 0), forward (1) or backward (2) sweep. ms[0] = the sweep tasks, ms[1 + l] = level l.
extern "C" int32_t gmrfx_level_times(gmrfx_handle *h, int32_t which, double *ms, int64_t cap, int64_t *count) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, false)) return e;
        if (!ms || !count) throw std::invalid_argument("null output");
        *count = h->D->level_times(which, ms, (int)cap);
        return GMRFX_OK;
    });
}
// The slicing of host_upload (download = 0) / host_download (1) for an n x nrhs array: plan = {columns per slice, row pieces per
// column, rows per piece, doubles per ring slot, slices, doubles reserved}. No handle, no device: arithmetic only.
extern "C" int32_t gmrfx_host_io_plan(int64_t n, int64_t nrhs, int32_t download, int64_t *plan) {
    if (!plan || n < 0 || nrhs < 0) return GMRFX_ERR_INVALID_ARG;
    const gmrfx::HostIoPlan p = gmrfx::host_io_plan_dir(n, nrhs, download);
    plan[0] = p.cols_per; plan[1] = p.ppc; plan[2] = p.rows_per; plan[3] = p.slot_doubles; plan[4] = p.nsl; plan[5] = p.reserve;
    return GMRFX_OK;
}
extern "C" int32_t gmrfx_set_stream(gmrfx_handle *h, void *hip_stream, int32_t use_external, int32_t async_phases) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, false)) return e;
        h->D->set_external_stream((hipStream_t)hip_stream, use_external != 0, async_phases != 0);
        return GMRFX_OK;
    });
}
extern "C" int32_t gmrfx_shard_info(const gmrfx_handle *h, int64_t *n_edges, int64_t *n_top_fronts, int64_t *shard_level) {
    if (!h) return GMRFX_ERR_INVALID_ARG;
    const Symbolic &S = h->S;
    int64_t ntop = 0;
    for (i32 s = 0; s < S.nsuper; s++) ntop += S.is_top[s];
    if (n_edges) *n_edges = (int64_t)S.shard_edges.size();
    if (n_top_fronts) *n_top_fronts = ntop;
    if (shard_level) *shard_level = S.shard_level;
    return GMRFX_OK;
}
// cross-rank tree edges child -> parent, ordered by the level of the parent: what moves between two phases
extern "C" int32_t gmrfx_shard_edges(const gmrfx_handle *h, int64_t *child, int64_t *src, int64_t *dst, int64_t *level,
                                     int64_t *cb_offset, int64_t *cb_count, int64_t *w_row0, int64_t *w_nrows, int64_t *zb_offset,
                                     int64_t *child_level) {
    if (!h || !child || !src || !dst || !level || !cb_offset || !cb_count || !w_row0 || !w_nrows) return GMRFX_ERR_INVALID_ARG;
    const Symbolic &S = h->S;
    const std::vector<i64> &wptr = S.wptr;      // cross-edge children come first, identically on every rank
    for (size_t k = 0; k < S.shard_edges.size(); k++) {
        const i32 d = S.shard_edges[k], p = S.sparent[d];
        const int64_t m = S.nrows(d) - S.ncols(d);
        child[k] = d; src[k] = S.owner[d]; dst[k] = S.owner[p]; level[k] = S.level[p];
        // offsets in THIS rank's arena (round 6: per-rank layouts); -1 when this rank is neither end of the edge
        const bool end = S.owner[d] == S.shard_rank || S.owner[p] == S.shard_rank;
        cb_offset[k] = end ? S.cbptr[d] : -1; cb_count[k] = m * m;
        w_row0[k] = wptr[d]; w_nrows[k] = m;
        if (zb_offset) zb_offset[k] = end ? S.zbptr[d] : -1;
        if (child_level) child_level[k] = S.level[d];
    }
    return GMRFX_OK;
}
// Distributed top fronts (symbolic.h: Symbolic::dist_fronts). counts[0] = number of distributed fronts, [1] = entries of all
// groups, [2] = contribution-block transfers of the factorisation, [3] = world. Per front (nullable arrays of counts[0]):
// supernode, columns, rows, offset of its panel in gmrfx_device_ptr(h, 1), leading dimension of the panel, tree level; gptr
// (counts[0] + 1) / grank (counts[1]): the ranks of its group. Panel block b (256 columns) is factored by grank[gptr[k] + b mod g].
extern "C" int32_t gmrfx_shard_dist_fronts(const gmrfx_handle *h, int64_t *counts, int64_t *front, int64_t *cols, int64_t *rows,
                                           int64_t *panel_offset, int64_t *panel_ld, int64_t *level, int64_t *gptr, int64_t *grank) {
    if (!h || !counts) return GMRFX_ERR_INVALID_ARG;
    const Symbolic &S = h->S;
    const size_t nf = S.dist_fronts.size();
    counts[0] = (int64_t)nf; counts[1] = (int64_t)S.dist_grank.size(); counts[2] = (int64_t)S.xf_child.size(); counts[3] = S.shard_world;
    for (size_t k = 0; k < nf; k++) {
        const i32 s = S.dist_fronts[k];
        if (front) front[k] = s;
        if (cols) cols[k] = S.ncols(s);
        if (rows) rows[k] = S.nrows(s);
        if (panel_offset) panel_offset[k] = S.panelptr[s];
        if (panel_ld) panel_ld[k] = S.ld[s];
        if (level) level[k] = S.level[s];
    }
    if (gptr) for (size_t k = 0; k < S.dist_gptr.size(); k++) gptr[k] = S.dist_gptr[k];
    if (grank) for (size_t k = 0; k < S.dist_grank.size(); k++) grank[k] = S.dist_grank[k];
    return GMRFX_OK;
}
// Every contribution-block transfer of the sharded factorisation (symbolic.h: xf_*), ordered by the level of the parent:
// Where THIS rank keeps panel block `block` (256 columns) of the distributed front `front`, as an offset into gmrfx_device_ptr(h, 1)
// and a count of doubles (whole columns): the buffer it passes to the broadcast of that block inside the front's group. The owner of
// the front (and every member of a front that has a contribution block) stores the whole panel: offset = panel + 256 block ld. A
// member with block-cyclic storage (Symbolic::compact_here) keeps its OWN blocks one behind the other and receives the others into
// a window of two blocks (block & 1). offset = -1, count = 0 on ranks outside the group.
extern "C" int32_t gmrfx_dist_front_block(const gmrfx_handle *h, int32_t front, int32_t block, int64_t *offset, int64_t *count) {
    if (!h || !offset || !count) return GMRFX_ERR_INVALID_ARG;
    const Symbolic &S = h->S;
    if (front < 0 || front >= S.nsuper || !S.is_dist(front) || block < 0 || block >= S.panel_blocks(front)) return GMRFX_ERR_INVALID_ARG;
    const i32 g = S.group_size(front), me = S.group_pos(front, S.shard_rank);
    *offset = -1; *count = 0;
    if (me < 0) return GMRFX_OK;
    const i64 ld = S.ld[front];
    *count = (i64)std::min<i64>(256, S.ncols(front) - 256 * (i64)block) * ld;
    if (!S.compact_here(front)) *offset = S.panelptr[front] + 256 * (i64)block * ld;
    else if (block % g == me) *offset = S.panelptr[front] + (i64)(block / g) * 256 * ld;
    else *offset = S.panelptr[front] + S.compact_window(front, block & 1);
    return GMRFX_OK;
}
// `count` doubles at `offset` of the arena (gmrfx_device_ptr(h, 0)) -- whole columns of `child`'s block -- go src -> dst before
// the fronts of `level` are assembled; col0 = the first of these columns. (Edges between fronts of one owner, neither
// distributed, have no entry.)
extern "C" int32_t gmrfx_shard_transfers(const gmrfx_handle *h, int64_t *child, int64_t *src, int64_t *dst, int64_t *level,
                                         int64_t *offset, int64_t *count, int64_t *col0) {
    if (!h) return GMRFX_ERR_INVALID_ARG;
    const Symbolic &S = h->S;
    for (size_t k = 0; k < S.xf_child.size(); k++) {
        if (child) child[k] = S.xf_child[k];
        if (src) src[k] = S.xf_src[k];
        if (dst) dst[k] = S.xf_dst[k];
        if (level) level[k] = S.xf_level[k];
        if (offset) offset[k] = S.xf_off[k];
        if (count) count[k] = S.xf_cnt[k];
        if (col0) col0[k] = S.xf_col0[k];
    }
    return GMRFX_OK;
}
// Block phases of distributed front `front` (a supernode of gmrfx_shard_dist_fronts; a no-op on ranks outside its group).
// what = 0: assemble this rank's panel blocks (Q's entries + the children's columns received); 1: factor panel block `block`
// (its owner only); 2: apply panel block `block` (complete on every member after its broadcast) to this rank's later panel
// blocks; 3: this rank's column blocks of the contribution block (children's columns received - L21 L21'). Asynchronous on the
// handle's main stream when async phases are on.
extern "C" int32_t gmrfx_dist_front_phase(gmrfx_handle *h, const double *d_nzval, int32_t front, int32_t what, int32_t block) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, false)) return e;
        h->D->dist_front_phase(d_nzval, front, what, block);
        return GMRFX_OK;
    });
}
extern "C" int32_t gmrfx_shard_owner(const gmrfx_handle *h, int64_t *owner, int64_t *is_top) {
    if (!h || !owner) return GMRFX_ERR_INVALID_ARG;
    for (i32 s = 0; s < h->S.nsuper; s++) { owner[s] = h->S.owner[s]; if (is_top) is_top[s] = h->S.is_top[s]; }
    return GMRFX_OK;
}
extern "C" void *gmrfx_device_ptr(gmrfx_handle *h, int32_t which) {
    if (!h || !h->D) return nullptr;
    try {
        if (which == 2 || which == 3) h->D->ensure_rhs(64);
    } catch (...) { return nullptr; }
    switch (which) {
        case 0: return h->D->cb_arena();
        case 1: return h->D->factor_panels();
        case 2: return h->D->rhs_x();
        case 3: return h->D->rhs_w();
        default: return nullptr;
    }
}
extern "C" int32_t gmrfx_solve_phase(gmrfx_handle *h, const double *d_B, int64_t ldb, int64_t nrhs, double *d_X, int64_t ldx, int32_t phase) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, true)) return e;
        if (phase == 0 && !d_B) throw std::invalid_argument("d_B is null");
        if (phase == 0 && ldb < h->S.n) throw std::invalid_argument("ldb < n");
        if (phase == 3 && (!d_X || ldx < h->S.n)) throw std::invalid_argument("d_X is null or ldx < n");
        h->D->solve_phase(d_B, ldb, nrhs, d_X, ldx, phase);
        return GMRFX_OK;
    });
}
// Row blocks of the right-hand-side buffer X (gmrfx_device_ptr(h, 2); row-major, leading dimension = nrhs) that the
// sharded solve moves: kind 2 = the own columns of every TOP front (owner broadcasts x after its backward step; level[]
// tells in which phase), kind 3 = the columns of every assigned subtree (gathered on rank 0 at the end).
extern "C" int32_t gmrfx_shard_rows(const gmrfx_handle *h, int32_t kind, int64_t *nblocks, int64_t *owner, int64_t *row0, int64_t *nrows,
                                    int64_t *level) {
    if (!h || !nblocks) return GMRFX_ERR_INVALID_ARG;
    const Symbolic &S = h->S;
    std::vector<int64_t> o, a, c, l;
    if (kind == 2) {
        for (i32 s = 0; s < S.nsuper; s++) if (S.is_top[s]) { o.push_back(S.owner[s]); a.push_back(S.sfirst[s]); c.push_back(S.ncols(s)); l.push_back(S.level[s]); }
    } else if (kind == 3) {
        for (size_t k = 0; k < S.shard_sub_root.size(); k++) {
            const i32 t = S.shard_sub_root[k];
            o.push_back(S.owner[t]); a.push_back(S.shard_sub_col0[k]); c.push_back(S.sfirst[t + 1] - S.shard_sub_col0[k]); l.push_back(S.level[t]);
        }
    } else return GMRFX_ERR_INVALID_ARG;
    *nblocks = (int64_t)o.size();
    if (owner && row0 && nrows)
        for (size_t k = 0; k < o.size(); k++) { owner[k] = o[k]; row0[k] = a[k]; nrows[k] = c[k]; if (level) level[k] = l[k]; }
    return GMRFX_OK;
}
extern "C" int32_t gmrfx_logdet_partial(gmrfx_handle *h, double *out) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, true)) return e;
        if (!out) throw std::invalid_argument("out is null");
        *out = h->D->logdet();      // sharded handles sum over their own columns only
        return GMRFX_OK;
    });
}
extern "C" int32_t gmrfx_refactorize_dev(gmrfx_handle *h, const double *d_nzval, int64_t *info) { return refactorize_impl(h, d_nzval, info, true); }

static int32_t solve_impl(gmrfx_handle *h, const double *B, int64_t ldb, int64_t nrhs, double *X, int64_t ldx, bool dev, int mode) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, true)) return e;
        if (nrhs < 0) throw std::invalid_argument("nrhs < 0");
        if (nrhs == 0) return GMRFX_OK;
        if (!B || !X) throw std::invalid_argument("B/X is null");
        if (ldb < h->S.n || ldx < h->S.n) throw std::invalid_argument("leading dimension smaller than n");
        h->D->solve(B, ldb, nrhs, X, ldx, dev, mode);
        return GMRFX_OK;
    });
}
extern "C" int32_t gmrfx_solve(gmrfx_handle *h, const double *B, int64_t ldb, int64_t nrhs, double *X, int64_t ldx) { return solve_impl(h, B, ldb, nrhs, X, ldx, false, 0); }
extern "C" int32_t gmrfx_solve_dev(gmrfx_handle *h, const double *B, int64_t ldb, int64_t nrhs, double *X, int64_t ldx) { return solve_impl(h, B, ldb, nrhs, X, ldx, true, 0); }
extern "C" int32_t gmrfx_backward_solve(gmrfx_handle *h, const double *Z, int64_t ldz, int64_t nrhs, double *X, int64_t ldx) { return solve_impl(h, Z, ldz, nrhs, X, ldx, false, 1); }
extern "C" int32_t gmrfx_backward_solve_dev(gmrfx_handle *h, const double *Z, int64_t ldz, int64_t nrhs, double *X, int64_t ldx) { return solve_impl(h, Z, ldz, nrhs, X, ldx, true, 1); }

extern "C" int32_t gmrfx_logdet(gmrfx_handle *h, double *out) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, true)) return e;
        if (!out) throw std::invalid_argument("out is null");
        *out = h->D->logdet();
        return GMRFX_OK;
    });
}

extern "C" int32_t gmrfx_selinv_compute(gmrfx_handle *h) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, true)) return e;
        h->D->selinv_compute();
        return GMRFX_OK;
    });
}

// sharded selected inversion (include/gmrfx.h): what = 0 begin, 1 gather the trailing inverse blocks of other ranks'
// fronts at level hi (parents owned here), 2 this rank's fronts of levels hi-1 .. lo, 3 end
extern "C" int32_t gmrfx_selinv_phase(gmrfx_handle *h, int32_t what, int32_t hi, int32_t lo) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, true)) return e;
        h->D->selinv_phase(what, hi, lo);
        return GMRFX_OK;
    });
}

extern "C" int32_t gmrfx_selinv_diag(gmrfx_handle *h, double *out) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, true)) return e;
        if (!out) throw std::invalid_argument("out is null");
        h->D->selinv_compute();
        h->D->selinv_diag(out);
        return GMRFX_OK;
    });
}

// Pattern of the de-permuted selected inverse (both triangles, rows sorted) + panel offsets.
static void build_zpattern(gmrfx_handle *h) {
    if (h->zpat_built) return;
    const Symbolic &S = h->S;
    const i64 n = S.n;
    std::vector<i64> cnt(n + 1, 0);
    for (i32 s = 0; s < S.nsuper; s++) {
        const i32 c = S.ncols(s), r = S.nrows(s);
        const i32 *rows = S.rows.data() + S.rowptr[s];
        for (i32 j = 0; j < c; j++) {
            const i32 b = S.perm[S.sfirst[s] + j];
            cnt[b + 1] += r - j;                       // column b gets rows i >= j
            for (i32 i = j + 1; i < r; i++) cnt[S.perm[rows[i]] + 1]++;  // mirrored entry
        }
    }
    h->zcolptr.assign(n + 1, 0);
    for (i64 j = 0; j < n; j++) h->zcolptr[j + 1] = h->zcolptr[j] + cnt[j + 1];
    const i64 nz = h->zcolptr[n];
    std::vector<std::pair<i64, i64>> ent((size_t)nz);  // (row, offset), bucketed by column
    std::vector<i64> w(h->zcolptr.begin(), h->zcolptr.end() - 1);
    for (i32 s = 0; s < S.nsuper; s++) {
        const i32 c = S.ncols(s), r = S.nrows(s);
        const i32 *rows = S.rows.data() + S.rowptr[s];
        for (i32 j = 0; j < c; j++) {
            const i32 b = S.perm[S.sfirst[s] + j];
            for (i32 i = j; i < r; i++) {
                const i32 a = S.perm[rows[i]];
                // a sharded handle holds the panels of its own fronts only: every other entry reads the zeroed slack word
                const i64 off = (!S.shard_plan || S.owner[s] == S.shard_rank) ? S.panelptr[s] + (i64)j * S.ld[s] + i : S.panelptr[S.nsuper];
                ent[w[b]++] = {a, off};
                if (i != j) ent[w[a]++] = {b, off};
            }
        }
    }
    h->zrow.resize(nz);
    h->zoff.resize(nz);
    for (i64 j = 0; j < n; j++) {
        std::sort(ent.begin() + h->zcolptr[j], ent.begin() + h->zcolptr[j + 1]);
        for (i64 p = h->zcolptr[j]; p < h->zcolptr[j + 1]; p++) { h->zrow[p] = ent[p].first; h->zoff[p] = ent[p].second; }
    }
    h->zpat_built = true;
}

extern "C" int32_t gmrfx_selinv_nnz(gmrfx_handle *h, int64_t *nnz) {
    return guarded(h, [&]() -> int32_t {
        if (!nnz) throw std::invalid_argument("nnz is null");
        *nnz = 2 * h->S.nnz_l_stored - h->S.n;
        return GMRFX_OK;
    });
}

extern "C" int32_t gmrfx_selinv_csc(gmrfx_handle *h, int32_t base, int64_t *colptr, int64_t *rowval, double *nzval) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, true)) return e;
        if (!colptr || !rowval || !nzval) throw std::invalid_argument("null output");
        if (base != 0 && base != 1) throw std::invalid_argument("index_base must be 0 or 1");
        h->D->selinv_compute();
        build_zpattern(h);
        const i64 n = h->S.n, nz = h->zcolptr[n];
        for (i64 j = 0; j <= n; j++) colptr[j] = h->zcolptr[j] + base;
        for (i64 p = 0; p < nz; p++) rowval[p] = h->zrow[p] + base;
        h->D->gather_z((const long long *)h->zoff.data(), nz, nzval);
        return GMRFX_OK;
    });
}

// Host-side planning loops (offset lookups into the supernodal structure) over [0, n): split over a few threads
// when long; fn(lo, hi) must only write its own range. Exceptions inside fn are collected and rethrown.
template <class F> static void parallel_ranges(i64 n, F &&fn) {
    const unsigned hw = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
    if (n < 200000 || hw == 1) { fn((i64)0, n); return; }
    std::vector<std::thread> th;
    th.reserve(hw);      // no reallocation (and so no bad_alloc with joinable threads alive) inside the loop
    std::vector<std::exception_ptr> err(hw);
    for (unsigned t = 0; t < hw; t++) {
        auto job = [&, t] { try { fn(n * t / hw, n * (t + 1) / hw); } catch (...) { err[t] = std::current_exception(); } };
        try { th.emplace_back(job); } catch (const std::system_error &) { job(); }     // no thread to be had: inline
    }
    for (auto &x : th) x.join();
    for (auto &e : err) if (e) std::rethrow_exception(e);
}

// A caller-supplied compressed pattern (ptr has ncol + 1 entries): ptr[0] == base, monotone. Everything that
// sizes a buffer from ptr[ncol] and then walks ptr[j] .. ptr[j + 1] checks this first (INVALID_ARG, not a heap overrun).
static void check_compressed_ptr(const int64_t *ptr, int64_t ncol, int32_t base, const char *what) {
    if (ptr[0] != base) throw std::invalid_argument(std::string(what) + "[0] != index_base");
    for (int64_t j = 0; j < ncol; j++)
        if (ptr[j + 1] < ptr[j]) throw std::invalid_argument(std::string(what) + " not monotone");
}

// offset of Sigma(i, j) (original indices) in the selected-inverse panels, -1 outside the factor pattern
static inline long long z_offset(const Symbolic &S, i64 i, i64 j) {
    i32 a = S.iperm[i], b = S.iperm[j];
    if (a < b) std::swap(a, b);
    const i32 s = S.col2super[b];
    const i32 *rows = S.rows.data() + S.rowptr[s];
    const i32 r = S.nrows(s);
    const i32 *it = std::lower_bound(rows, rows + r, a);
    if (it == rows + r || *it != a) return -1;
    if (S.shard_plan && S.owner[s] != S.shard_rank) return (long long)S.panelptr[S.nsuper];      // another rank's panel: the zeroed slack word
    return (long long)(S.panelptr[s] + (i64)(b - S.sfirst[s]) * S.ld[s] + (it - rows));
}

extern "C" int32_t gmrfx_selinv_extract(gmrfx_handle *h, int64_t ncol, const int64_t *colptr, const int64_t *rowval,
                                        int32_t base, double *out) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, true)) return e;
        const Symbolic &S = h->S;
        if (ncol != S.n) throw std::invalid_argument("pattern must have n columns");
        if (!colptr || !rowval || !out) throw std::invalid_argument("null argument");
        if (base != 0 && base != 1) throw std::invalid_argument("index_base must be 0 or 1");
        check_compressed_ptr(colptr, ncol, base, "colptr");
        h->D->selinv_compute();
        const i64 nz = colptr[ncol] - base;
        std::vector<long long> off((size_t)nz);
        parallel_ranges(ncol, [&](i64 lo, i64 hi) {
            for (i64 j = lo; j < hi; j++)
                for (i64 p = colptr[j] - base; p < colptr[j + 1] - base; p++) {
                    i64 i = rowval[p] - base;
                    if (i < 0 || i >= S.n) throw std::invalid_argument("rowval out of range");
                    off[p] = z_offset(S, i, j);
                }
        });
        h->D->gather_z(off.data(), nz, out);
        return GMRFX_OK;
    });
}

extern "C" int32_t gmrfx_selinv_dot(gmrfx_handle *h, int64_t ncol, const int64_t *colptr, const int64_t *rowval,
                                    const double *nzval, int32_t base, double *out) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, true)) return e;
        const Symbolic &S = h->S;
        if (ncol != S.n) throw std::invalid_argument("B must have n columns");
        if (!colptr || !rowval || !nzval || !out) throw std::invalid_argument("null argument");
        if (base != 0 && base != 1) throw std::invalid_argument("index_base must be 0 or 1");
        check_compressed_ptr(colptr, ncol, base, "colptr");
        h->D->selinv_compute();
        const i64 nz = colptr[ncol] - base;
        std::vector<long long> off((size_t)nz);
        parallel_ranges(ncol, [&](i64 lo, i64 hi) {
            for (i64 j = lo; j < hi; j++)
                for (i64 p = colptr[j] - base; p < colptr[j + 1] - base; p++) {
                    i64 i = rowval[p] - base;
                    if (i < 0 || i >= S.n) throw std::invalid_argument("rowval out of range");
                    off[p] = z_offset(S, i, j);
                }
        });
        // fixed chunks of 4096 entries are summed on the device, the chunk sums on the host in order
        const i64 CH = 4096, nseg = (nz + CH - 1) / CH;
        std::vector<long long> seg((size_t)nseg + 1);
        for (i64 g = 0; g <= nseg; g++) seg[g] = std::min(g * CH, nz);
        std::vector<double> part((size_t)nseg);
        h->D->weighted_z_sums(seg.data(), nseg, off.data(), nzval + 0, part.data());
        double acc = 0.0;
        for (double v : part) acc += v;
        *out = acc;
        return GMRFX_OK;
    });
}

// pairs (p, q <= p) of the entries of every row of a sparse design matrix + the offsets of Sigma[j_p, j_q]
static void plan_row_pairs(const Symbolic &S, int64_t m, const int64_t *rowptr, const int64_t *colind, int32_t base,
                           std::vector<long long> &seg, std::vector<long long> &off, std::vector<int> &pi, std::vector<int> &qi) {
    check_compressed_ptr(rowptr, m, base, "rowptr");
    seg.assign((size_t)m + 1, 0);
    for (i64 i = 0; i < m; i++) {
        const i64 k = rowptr[i + 1] - rowptr[i];
        if (k < 0) throw std::invalid_argument("rowptr not monotone");
        seg[i + 1] = seg[i] + k * (k + 1) / 2;
    }
    const i64 nz = rowptr[m] - base;
    if (nz > 0x7fffffffLL) throw std::invalid_argument("design matrix too large");
    off.resize((size_t)seg[m]); pi.resize((size_t)seg[m]); qi.resize((size_t)seg[m]);
    parallel_ranges(m, [&](i64 lo, i64 hi) {
        for (i64 i = lo; i < hi; i++) {
            long long t = seg[i];
            for (i64 p = rowptr[i] - base; p < rowptr[i + 1] - base; p++) {
                const i64 jp = colind[p] - base;
                if (jp < 0 || jp >= S.n) throw std::invalid_argument("colind out of range");
                for (i64 q = rowptr[i] - base; q <= p; q++) {
                    const i64 jq = colind[q] - base;
                    if (jq < 0 || jq >= S.n) throw std::invalid_argument("colind out of range");
                    off[t] = z_offset(S, jp, jq);
                    pi[t] = (int)p; qi[t] = (int)q;
                    t++;
                }
            }
        }
    });
}

extern "C" int32_t gmrfx_selinv_row_diag(gmrfx_handle *h, int64_t m, const int64_t *rowptr, const int64_t *colind,
                                         const double *values, int32_t base, double *out) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, true)) return e;
        if (m < 0 || !rowptr || (m > 0 && !out)) throw std::invalid_argument("null argument");
        if (base != 0 && base != 1) throw std::invalid_argument("index_base must be 0 or 1");
        if (m == 0) return GMRFX_OK;
        if (rowptr[m] - base > 0 && (!colind || !values)) throw std::invalid_argument("null argument");
        h->D->selinv_compute();
        std::vector<long long> seg, off;
        std::vector<int> pi, qi;
        plan_row_pairs(h->S, m, rowptr, colind, base, seg, off, pi, qi);
        std::vector<double> w(off.size());
        for (size_t t = 0; t < w.size(); t++) w[t] = (pi[t] == qi[t] ? 1.0 : 2.0) * values[pi[t]] * values[qi[t]];
        h->D->weighted_z_sums(seg.data(), m, off.data(), w.data(), out);
        return GMRFX_OK;
    });
}

extern "C" int32_t gmrfx_selinv_row_diag_plan(gmrfx_handle *h, int64_t m, const int64_t *rowptr, const int64_t *colind,
                                              int32_t base, int64_t *plan) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, false)) return e;
        if (m < 0 || !rowptr || !plan) throw std::invalid_argument("null argument");
        if (base != 0 && base != 1) throw std::invalid_argument("index_base must be 0 or 1");
        if (m > 0 && rowptr[m] - base > 0 && !colind) throw std::invalid_argument("null argument");
        std::vector<long long> seg, off;
        std::vector<int> pi, qi;
        plan_row_pairs(h->S, m, rowptr, colind, base, seg, off, pi, qi);
        *plan = h->D->rowdiag_plan_create(seg.data(), m, off.data(), pi.data(), qi.data(), m > 0 ? rowptr[m] - base : 0);
        return GMRFX_OK;
    });
}

extern "C" int32_t gmrfx_selinv_row_diag_apply(gmrfx_handle *h, int64_t plan, const double *values, double *out) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, true)) return e;
        if (!values || !out) throw std::invalid_argument("null argument");
        h->D->selinv_compute();
        h->D->rowdiag_plan_apply(plan, values, out);
        return GMRFX_OK;
    });
}

extern "C" int32_t gmrfx_selinv_row_diag_free(gmrfx_handle *h, int64_t plan) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, false)) return e;
        h->D->rowdiag_plan_free(plan);
        return GMRFX_OK;
    });
}

extern "C" int32_t gmrfx_get_perm(const gmrfx_handle *h, int32_t base, int64_t *perm) {
    if (!h || !perm) return GMRFX_ERR_INVALID_ARG;
    for (i64 k = 0; k < h->S.n; k++) perm[k] = (i64)h->S.perm[k] + base;
    return GMRFX_OK;
}

static int32_t quadform_impl(gmrfx_handle *h, const double *nz, const double *X, int64_t ldx, int64_t nvec,
                             const double *mu, double *out, bool dev) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, false)) return e;
        if (nvec < 0 || (nvec > 0 && (!X || !out))) { h->err = "quadform: null X/out or negative nvec"; return GMRFX_ERR_INVALID_ARG; }
        if (nvec == 0) return GMRFX_OK;
        if (ldx < h->S.n) { h->err = "quadform: ldx < n"; return GMRFX_ERR_INVALID_ARG; }
        if (dev) { h->D->quadform(nz, X, ldx, nvec, mu, out); return GMRFX_OK; }
        // host operands: stage them through plain device buffers (freed on return)
        const i64 n = h->S.n;
        struct Buf { void *p = nullptr; ~Buf() { if (p) (void)hipFree(p); } } bx, bm, bn;
        hip_check(hipSetDevice(h->D->device), "hipSetDevice");
        hip_check(hipMalloc(&bx.p, (size_t)std::max<i64>(n * nvec, 1) * sizeof(double)), "hipMalloc");
        hip_check(hipMemcpy2D(bx.p, (size_t)n * sizeof(double), X, (size_t)ldx * sizeof(double), (size_t)n * sizeof(double),
                              (size_t)nvec, hipMemcpyHostToDevice), "hipMemcpy2D");
        if (mu) {
            hip_check(hipMalloc(&bm.p, (size_t)std::max<i64>(n, 1) * sizeof(double)), "hipMalloc");
            hip_check(hipMemcpy(bm.p, mu, (size_t)n * sizeof(double), hipMemcpyHostToDevice), "hipMemcpy");
        }
        if (nz) {
            hip_check(hipMalloc(&bn.p, (size_t)std::max<i64>(h->S.nnz_in, 1) * sizeof(double)), "hipMalloc");
            hip_check(hipMemcpy(bn.p, nz, (size_t)h->S.nnz_in * sizeof(double), hipMemcpyHostToDevice), "hipMemcpy");
        }
        h->D->quadform((const double *)bn.p, (const double *)bx.p, n, nvec, (const double *)bm.p, out);
        return GMRFX_OK;
    });
}
extern "C" int32_t gmrfx_quadform(gmrfx_handle *h, const double *nzval, const double *X, int64_t ldx, int64_t nvec,
                                  const double *mu, double *out) {
    return quadform_impl(h, nzval, X, ldx, nvec, mu, out, false);
}
extern "C" int32_t gmrfx_quadform_dev(gmrfx_handle *h, const double *d_nzval, const double *d_X, int64_t ldx, int64_t nvec,
                                      const double *d_mu, double *out) {
    return quadform_impl(h, d_nzval, d_X, ldx, nvec, d_mu, out, true);
}

// ---- dense-operator leg of the separable (Kronecker) path, SURVEY 8 f3 (separable.jl:122-172) -------------
extern "C" int32_t gmrfx_dense_apply_dev(gmrfx_handle *h, int64_t n1, int64_t n2, const double *d_D, const double *d_T, double *d_R) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, false)) return e;
        if (n1 < 0 || n2 < 0 || n1 > 0x7fffffffLL) throw std::invalid_argument("dense_apply: bad dimensions");
        if (n1 > 0 && n2 > 0 && (!d_D || !d_T || !d_R)) throw std::invalid_argument("dense_apply: null pointer");
        if (d_T == d_R) throw std::invalid_argument("dense_apply: T and R must not alias");
        if (n1 > 0 && n2 > 0 && (n2 + 63) / 64 * ((n1 + 63) / 64) > 0x0fffffffLL) throw std::invalid_argument("dense_apply: too many tiles");
        h->D->dense_apply(d_D, d_T, d_R, n1, n2);
        return GMRFX_OK;
    });
}
extern "C" int32_t gmrfx_transpose_dev(gmrfx_handle *h, int64_t rows, int64_t cols, const double *d_src, double *d_dst) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, false)) return e;
        if (rows < 0 || cols < 0) throw std::invalid_argument("transpose: bad dimensions");
        if (rows > 0 && cols > 0 && (!d_src || !d_dst || d_src == d_dst)) throw std::invalid_argument("transpose: null or aliased pointers");
        if (((rows + 63) >> 6) * ((cols + 63) >> 6) > 0x7fffffffLL) throw std::invalid_argument("transpose: too many tiles");
        h->D->transpose(d_src, d_dst, rows, cols);
        return GMRFX_OK;
    });
}

namespace gmrfx {
struct KlTask { long long rows_off; long long cols_off; int nrows, ncols; };
long long kl_cholesky_run(int device, long long n, const double *theta, long long ldt, bool theta_on_device,
                          const std::vector<KlTask> &tasks, const std::vector<int> &rows, const std::vector<int> &cols,
                          const long long *Lcolptr, long long nnzL, double reg, double *nzval_out);
}

extern "C" int32_t gmrfx_kl_cholesky(int64_t n, const double *theta, int64_t ldt, int32_t theta_on_device,
                                     const int64_t *L_colptr, int64_t ntasks, const int64_t *task_rowptr,
                                     const int64_t *task_rows, const int64_t *task_colptr, const int64_t *task_cols,
                                     int32_t base, double reg, int32_t device, double *nzval, int64_t *info) {
    if (info) *info = 0;
    try {
        if (n <= 0 || !theta || ldt < n || !L_colptr || !nzval) throw std::invalid_argument("kl_cholesky: null argument / ldt < n");
        if (ntasks < 0 || (ntasks > 0 && (!task_rowptr || !task_rows || !task_colptr || !task_cols)))
            throw std::invalid_argument("kl_cholesky: null task arrays");
        if (base != 0 && base != 1) throw std::invalid_argument("index_base must be 0 or 1");
        if (n > 0x7fffffffLL || ntasks > 0x7fffffffLL) throw std::invalid_argument("kl_cholesky: too large");
        std::vector<long long> colptr((size_t)n + 1);
        for (i64 j = 0; j <= n; j++) colptr[j] = L_colptr[j] - base;
        for (i64 j = 0; j < n; j++) if (colptr[j + 1] < colptr[j]) throw std::invalid_argument("L_colptr not monotone");
        const i64 nnzL = colptr[n];
        std::vector<KlTask> tasks((size_t)ntasks);
        const i64 nr = ntasks ? task_rowptr[ntasks] - base : 0, nc = ntasks ? task_colptr[ntasks] - base : 0;
        std::vector<int> rows((size_t)nr), cols((size_t)nc);
        for (i64 k = 0; k < nr; k++) {
            const i64 v = task_rows[k] - base;
            if (v < 0 || v >= n) throw std::invalid_argument("task_rows out of range");
            rows[k] = (int)v;
        }
        for (i64 k = 0; k < nc; k++) {
            const i64 v = task_cols[k] - base;
            if (v < 0 || v >= n) throw std::invalid_argument("task_cols out of range");
            cols[k] = (int)v;
        }
        for (i64 t = 0; t < ntasks; t++) {
            KlTask &tk = tasks[t];
            tk.rows_off = task_rowptr[t] - base; tk.cols_off = task_colptr[t] - base;
            const i64 a = task_rowptr[t + 1] - task_rowptr[t], b = task_colptr[t + 1] - task_colptr[t];
            if (a <= 0 || b < 0) throw std::invalid_argument("kl_cholesky: empty task");
            tk.nrows = (int)a; tk.ncols = (int)b;
            for (i64 q = 0; q < b; q++) {
                const int col = cols[tk.cols_off + q];
                const i64 nk = colptr[col + 1] - colptr[col];
                if (nk < 1 || nk > a) throw std::invalid_argument("kl_cholesky: a column has more entries than its task has rows");
            }
        }
        const long long bad = kl_cholesky_run(device, n, theta, ldt, theta_on_device != 0, tasks, rows, cols, colptr.data(), nnzL, reg, nzval);
        if (bad >= 0) {
            if (info) *info = bad + 1;
            g_create_err = "kl_cholesky: local covariance block of task " + std::to_string(bad) + " is not positive definite";
            return GMRFX_ERR_NOT_POSDEF;
        }
        return GMRFX_OK;
    } catch (const std::invalid_argument &e) {
        g_create_err = e.what();
        return GMRFX_ERR_INVALID_ARG;
    } catch (const std::bad_alloc &) {
        g_create_err = "out of host memory";
        return GMRFX_ERR_ALLOC;
    } catch (const std::exception &e) {
        g_create_err = e.what();
        return std::string(e.what()).find("no HIP device") != std::string::npos ? GMRFX_ERR_NO_DEVICE : GMRFX_ERR_HIP;
    }
}

extern "C" int32_t gmrfx_get_stats(const gmrfx_handle *h, gmrfx_stats *out, int32_t struct_size) {
    if (!h || !out || struct_size <= 0) return GMRFX_ERR_INVALID_ARG;
    gmrfx_stats st;
    std::memset(&st, 0, sizeof(st));
    const Symbolic &S = h->S;
    st.n = S.n; st.nnz_q_tri = S.nnz_q_tri; st.nnz_l = S.nnz_l_true; st.nnz_l_stored = S.nnz_l_stored;
    st.nsuper = S.nsuper; st.nlevels = S.nlevels; st.max_cols = S.max_cols; st.max_rows = S.max_rows;
    st.sum_rows = S.sum_rows; st.n_small_fronts = S.n_small; st.n_big_fronts = S.n_big;
    st.factor_flops = S.flops;
    st.bytes_factor = 8.0 * (double)S.panelptr[S.nsuper];
    st.bytes_cb_arena = 8.0 * (double)S.cb_arena;
    st.ms_symbolic = S.ms_symbolic;
    st.fail_col = -1;
    if (h->D) {
        const Device &D = *h->D;
        st.bytes_device_total = D.bytes_total;
        st.ms_factor = D.ms_factor; st.ms_solve = D.ms_solve; st.ms_solve_fwd = D.ms_fwd; st.ms_solve_bwd = D.ms_bwd;
        st.ms_solve_perm = D.ms_perm; st.ms_backward_solve = D.ms_bsolve; st.ms_logdet = D.ms_logdet; st.ms_selinv = D.ms_selinv;
        st.last_nrhs = D.last_nrhs;
        st.ms_syrk = const_cast<gmrfx::Device &>(D).syrk_ms(); st.syrk_flops = D.syrk_flops; st.syrk_launches = D.syrk_launches;
        st.ms_quadform = D.ms_quadform;
        if (D.factorized) st.fail_col = const_cast<Device &>(D).fail_col();
    }
    std::memcpy(out, &st, std::min<size_t>((size_t)struct_size, sizeof(st)));
    return GMRFX_OK;
}

extern "C" int32_t gmrfx_symbolic_sizes(const gmrfx_handle *h, int64_t *sizes) {
    if (!h || !sizes) return GMRFX_ERR_INVALID_ARG;
    const Symbolic &S = h->S;
    sizes[0] = S.nsuper; sizes[1] = S.sum_rows; sizes[2] = S.panelptr[S.nsuper]; sizes[3] = S.nlevels;
    sizes[4] = S.cb_arena; sizes[5] = (int64_t)S.qsrc.size(); sizes[6] = 0; sizes[7] = 0;
    return GMRFX_OK;
}

extern "C" int32_t gmrfx_symbolic_get(const gmrfx_handle *h, int64_t *super_first, int64_t *super_parent,
                                      int64_t *row_ptr, int64_t *rows, int64_t *rel, int64_t *panel_ptr,
                                      int64_t *panel_ld, int64_t *level, int64_t *q_src, int64_t *q_dst) {
    if (!h) return GMRFX_ERR_INVALID_ARG;
    const Symbolic &S = h->S;
    const i32 ns = S.nsuper;
    if (super_first) for (i32 s = 0; s <= ns; s++) super_first[s] = S.sfirst[s];
    if (super_parent) for (i32 s = 0; s < ns; s++) super_parent[s] = S.sparent[s];
    if (row_ptr) for (i32 s = 0; s <= ns; s++) row_ptr[s] = S.rowptr[s];
    if (rows) for (i64 k = 0; k < S.sum_rows; k++) rows[k] = S.rows[k];
    if (rel) for (i64 k = 0; k < S.sum_rows; k++) rel[k] = S.rel[k];
    if (panel_ptr) for (i32 s = 0; s <= ns; s++) panel_ptr[s] = S.panelptr[s];
    if (panel_ld) for (i32 s = 0; s < ns; s++) panel_ld[s] = S.ld[s];
    if (level) for (i32 s = 0; s < ns; s++) level[s] = S.level[s];
    if (q_src) for (size_t k = 0; k < S.qsrc.size(); k++) q_src[k] = S.qsrc[k];
    if (q_dst) {
        for (size_t k = 0; k < S.qdst.size(); k++) q_dst[k] = S.qdst[k];
        // a sharded handle stores the panels of its own fronts only: the entries of Q that go into another rank's panel
        // have no destination here (-1)
        if (S.shard_plan)
            for (i32 s = 0; s < ns; s++) {
                if (!S.stored_here(s))      // (every member of its group stores the panel of a distributed front ...)
                    for (i64 k = S.qptr[s]; k < S.qptr[s + 1]; k++) q_dst[k] = -1;
                else if (S.compact_here(s)) {   // (... or, block-cyclic storage, its own 256-column blocks one behind the other)
                    const i64 ld = S.ld[s];
                    const i32 g = S.group_size(s), me = S.group_pos(s, S.shard_rank);
                    for (i64 k = S.qptr[s]; k < S.qptr[s + 1]; k++) {
                        const i64 rel = S.qdst[k] - S.panelptr[s], col = rel / ld, row = rel % ld, b = col >> 8;
                        q_dst[k] = b % g != me ? -1 : S.panelptr[s] + (col - 256 * (b - b / g)) * ld + row;
                    }
                }
            }
    }
    return GMRFX_OK;
}

// Sweep tasks (symbolic.h: swt_*): bottom subtrees whose triangular sweeps run on an LDS-resident local vector.
// ntasks / rows_cap always; first / last (ntasks each) and lrow (sum_rows) when non-null.
extern "C" int32_t gmrfx_symbolic_sweep_tasks(const gmrfx_handle *h, int64_t *ntasks, int64_t *rows_cap, int64_t *first,
                                              int64_t *last, int64_t *lrow) {
    if (!h || !ntasks) return GMRFX_ERR_INVALID_ARG;
    const Symbolic &S = h->S;
    *ntasks = (int64_t)S.swt_first.size();
    if (rows_cap) *rows_cap = S.swt_rows;
    if (first) for (size_t k = 0; k < S.swt_first.size(); k++) first[k] = S.swt_first[k];
    if (last) for (size_t k = 0; k < S.swt_last.size(); k++) last[k] = S.swt_last[k];
    if (lrow) for (size_t k = 0; k < S.lrow.size(); k++) lrow[k] = S.lrow[k];
    return GMRFX_OK;
}

extern "C" int32_t gmrfx_symbolic_sweep_chunks(const gmrfx_handle *h, int64_t *nchunks, int64_t *nrows, int64_t *task_ptr,
                                               int64_t *slot, int64_t *fwd, int64_t *bwd, int64_t *rows) {
    if (!h || !nchunks) return GMRFX_ERR_INVALID_ARG;
    const Symbolic &S = h->S;
    nchunks[0] = (int64_t)S.swc_fwd.size();
    nchunks[1] = (int64_t)S.swc_bwd.size();
    if (nrows) *nrows = (int64_t)S.swc_rows.size();
    if (task_ptr) for (size_t k = 0; k < S.swc_ptr.size(); k++) { task_ptr[2 * k] = S.swc_ptr[k]; task_ptr[2 * k + 1] = S.swc_bptr[k]; }
    if (slot) for (size_t k = 0; k < S.swc_slot.size(); k++) slot[k] = S.swc_slot[k];
    auto put = [](const std::vector<Symbolic::SwChunk> &v, int64_t *out) {
        for (size_t k = 0; k < v.size(); k++) {
            const Symbolic::SwChunk &c = v[k];
            int64_t *o = out + 8 * k;
            o[0] = c.pa; o[1] = c.ld; o[2] = c.o; o[3] = c.cc; o[4] = c.nt; o[5] = c.lr; o[6] = c.nbar; o[7] = c.id;
        }
    };
    if (fwd) put(S.swc_fwd, fwd);
    if (bwd) put(S.swc_bwd, bwd);
    if (rows) for (size_t k = 0; k < S.swc_rows.size(); k++) rows[k] = S.swc_rows[k];
    return GMRFX_OK;
}

extern "C" int32_t gmrfx_get_factor_values(gmrfx_handle *h, double *out) {
    return guarded(h, [&]() -> int32_t {
        if (int32_t e = need_device(h, true)) return e;
        if (!out) throw std::invalid_argument("out is null");
        h->D->copy_factor(out);
        return GMRFX_OK;
    });
}
