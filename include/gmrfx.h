/*
 * gmrfx.h -- C ABI of libgmrfx.so, the MI355X-native sparse-Cholesky / triangular-solve /
 * selected-inverse backend for GaussianMarkovRandomFields.jl's GMRF precision-matrix path.
 *
 * Every entry point replaces one call the reference makes into CHOLMOD / SelectedInversion.jl
 * behind its two solver seams (file:line relative to the reference repository):
 *
 *   seam B  WorkspaceBackend protocol            src/workspace/backend.jl:8-30
 *   seam A  LinearSolve algorithm + GMRF hooks   src/solvers/{selinv,backward_solve,logdet}.jl,
 *                                                ext/GaussianMarkovRandomFieldsPardiso.jl:10-80
 *
 * Conventions: plain C types only; int64 indices (Julia `Int`); `index_base` 0 or 1; dense
 * blocks are column-major with explicit leading dimension; every array is caller-owned and
 * only read/written during the call (the library copies what it keeps). A handle owns one HIP
 * stream; a handle is not re-entrant, distinct handles may be used from different threads
 * (the WorkspacePool contract, src/workspace/workspace_pool.jl:15-20). No callbacks, no
 * global mutable state except a thread-local error string for failed gmrfx_create calls.
 *
 * All numeric work runs on the GPU. There is no CPU fallback: without a usable HIP device the
 * numeric entry points return GMRFX_ERR_NO_DEVICE.
 */
#ifndef GMRFX_H
#define GMRFX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gmrfx_handle gmrfx_handle;

enum {
    GMRFX_OK = 0,
    GMRFX_ERR_INVALID_ARG = 1,   /* -> Julia ArgumentError / DimensionMismatch */
    GMRFX_ERR_NO_DEVICE = 2,     /* no HIP device or handle created symbolic_only */
    GMRFX_ERR_HIP = 3,           /* HIP runtime error, text in gmrfx_last_error */
    GMRFX_ERR_NOT_FACTORIZED = 4,/* numeric op before the first gmrfx_refactorize */
    GMRFX_ERR_NOT_POSDEF = 5,    /* only from gmrfx_refactorize when opts.check_posdef != 0 */
    GMRFX_ERR_ALLOC = 6
};

enum { GMRFX_UPLO_UPPER = 0, GMRFX_UPLO_LOWER = 1 };
enum { GMRFX_ORDER_AUTO = 0, GMRFX_ORDER_NATURAL = 1 };

typedef struct gmrfx_opts {
    int32_t struct_size;    /* = sizeof(gmrfx_opts); lets the struct grow compatibly */
    int32_t uplo;           /* which stored triangle defines Q when both are present.
                               GMRFX_UPLO_UPPER mirrors Symmetric(ws.Q) (default :U),
                               src/workspace/gmrf_workspace.jl:176 */
    int32_t ordering;       /* used when perm == NULL: AUTO = nested dissection (geometric if
                               coords != NULL, graph-based otherwise), NATURAL = identity */
    int32_t device;         /* HIP device ordinal; -1 = current device */
    int32_t symbolic_only;  /* 1: host symbolic analysis only (ordering, supernodes, stats,
                               gmrfx_get_perm); numeric entry points fail loudly */
    int32_t check_posdef;   /* 0: seam-B behaviour, never fail on indefiniteness
                               (cholesky!(...; check=false), src/workspace/backend.jl:184);
                               1: seam-A behaviour, return GMRFX_ERR_NOT_POSDEF */
    int32_t nd_leaf;        /* nested-dissection leaf size (0 = default) */
    int32_t relax_cols;     /* supernode amalgamation: merge while width <= relax_cols ... */
    double  relax_zeros;    /* ... or added explicit-zero fraction <= relax_zeros (0 = default) */
    int32_t coord_dim;      /* 0, 2 or 3 */
    int32_t reserved0;
    const double *coords;   /* optional n x coord_dim row-major node coordinates (mesh nodes),
                               enables geometric nested dissection; NULL otherwise */
    /* Sharding ONE factorisation over shard_world processes (one GPU each): every process creates
     * its handle with the same pattern / options and its own shard_rank. The supernodal tree is cut
     * into subtrees dealt to the ranks; every front above them (the "top") is owned by ONE rank of the
     * group whose subtrees it joins (the least loaded owner of its children). See the
     * gmrfx_shard_* entry points. shard_world <= 1: unsharded (default). */
    int32_t shard_rank, shard_world;
    /* Force at least this many fronts into the top of the sharded plan (0 = the cost model decides). With shard_world == 1 a
     * value > 0 makes the handle a SHARDED handle of one rank: the phase entry points, the top levels and the (empty or
     * self-addressed) exchange lists of the protocol below run on a single GPU -- how tests/test_rccl_world1.py executes the
     * RCCL device path of gmrfx/shard.py on the one-GPU boxes of the test pool. */
    int32_t shard_min_top;
    int32_t reserved1;
} gmrfx_opts;

typedef struct gmrfx_stats {
    int64_t n, nnz_q_tri, nnz_l, nnz_l_stored; /* true fill / fill incl. amalgamation zeros */
    int64_t nsuper, nlevels, max_cols, max_rows, sum_rows;
    int64_t n_small_fronts, n_big_fronts;
    double  factor_flops;          /* sum_s c^3/3 + c^2 (r-c) + c (r-c)^2                  */
    double  bytes_factor, bytes_cb_arena, bytes_device_total;
    double  ms_symbolic;           /* host wall time of gmrfx_create's analysis             */
    /* GPU times of the most recent call of each kind, HIP events on the handle's stream    */
    double  ms_factor, ms_solve, ms_solve_fwd, ms_solve_bwd, ms_solve_perm, ms_backward_solve,
            ms_logdet, ms_selinv;
    int64_t last_nrhs;
    int64_t fail_col;              /* -1, or first (permuted, 0-based) non-positive pivot   */
    /* the dominant kernel of the factorisation (contribution-block SYRK, k_syrk_cb): summed HIP-event
     * time of its launches in the most recent refactorisation, their number, and the flops they do
     * (sum over big fronts of c m (m + 1), m = r - c: lower triangle only)                  */
    double  ms_syrk, syrk_flops;
    int64_t syrk_launches;
    double  ms_quadform;           /* most recent gmrfx_quadform(_dev): kernels only             */
} gmrfx_stats;

/* Message for the most recent failed gmrfx_create on this thread. */
const char *gmrfx_last_create_error(void);
/* Message for the most recent failure on this handle. */
const char *gmrfx_last_error(const gmrfx_handle *h);

/* Symbolic analysis (+ device upload). Replaces `cholesky(Q; perm=...)`'s analyse phase:
 * CHOLMODBackend ctor src/workspace/backend.jl:147-153, ordering_permutation :73-133.
 * colptr/rowval: CSC pattern of Q, either one triangle or both (both is what ws.Q holds).
 * perm (nullable): user elimination order, perm[k] = index (index_base-based) of the k-th
 * pivot. The pattern is fixed for the life of the handle; nzval passed later must follow the
 * same CSC order (the workspace guarantees it: gmrf_workspace.jl:131-143). */
int32_t gmrfx_create(int64_t n, const int64_t *colptr, const int64_t *rowval, int32_t index_base,
                     const int64_t *perm, const gmrfx_opts *opts, gmrfx_handle **out);
void    gmrfx_destroy(gmrfx_handle *h);
/* deepcopy(cache) at the start of Newton loops: arithmetic/condition/gaussian_approximation.jl:103-109 */
int32_t gmrfx_clone(const gmrfx_handle *h, gmrfx_handle **out);

/* Numeric refactorisation on the fixed pattern. Replaces
 * `_copy_sparse_values!` + `cholesky!(F, S; check=false)`: src/workspace/backend.jl:165-189.
 * Drops the selected-inverse cache. info (nullable) <- 0, or 1 + first failing pivot column
 * in the elimination order. */
int32_t gmrfx_refactorize(gmrfx_handle *h, const double *nzval, int64_t *info);
int32_t gmrfx_refactorize_dev(gmrfx_handle *h, const double *d_nzval, int64_t *info);
/* Numeric refactorisation AND solve Q X = B in one call. Replaces `workspace_solve(ws, B)` on a workspace whose values have
 * just been updated (src/workspace/gmrf_workspace.jl:170-178 `ensure_numeric!` -> `refactorize!`, :207-215 `backend_solve`):
 * the Newton / hyper-parameter loops always run the two back to back. Same results, bit for bit, as gmrfx_refactorize followed
 * by gmrfx_solve -- but pipelined: the forward sweep follows the factorisation up the elimination tree on a second stream
 * (level l of the sweep starts when level l is factored), so the bottom of the sweep fills the chip while the top of the
 * factorisation is a chain of small dependent launches. X is only meaningful when *info == 0. B / X column-major n x nrhs. */
int32_t gmrfx_refactorize_solve(gmrfx_handle *h, const double *nzval, const double *B, int64_t ldb, int64_t nrhs, double *X, int64_t ldx,
                                int64_t *info);
int32_t gmrfx_refactorize_solve_dev(gmrfx_handle *h, const double *d_nzval, const double *d_B, int64_t ldb, int64_t nrhs, double *d_X,
                                    int64_t ldx, int64_t *info);

/* One evaluation of the hyper-parameter loop (docs/src/literate-tutorials/workspace_factorization_reuse.jl:94-102) in ONE call:
 * new values -> numeric factorisation, quad[k] = (x_k - mu)' Q (x_k - mu) for nvec vectors and *logdet = log det Q, i.e. everything
 * `logpdf(::WorkspaceGMRF, z)` needs (src/workspace/workspace_gmrf.jl:288-292 with `ensure_numeric!` inside,
 * gmrf_workspace.jl:170-178). Device pointers for Q's values, X (column-major n x nvec) and mu (nullable); the quadratic forms run
 * beside the factorisation, one synchronisation, both results arrive through pinned memory. Same bits as gmrfx_refactorize_dev +
 * gmrfx_quadform_dev + gmrfx_logdet. */
int32_t gmrfx_refactorize_logpdf_dev(gmrfx_handle *h, const double *d_nzval, const double *d_X, int64_t ldx, int64_t nvec,
                                     const double *d_mu, double *quad /* host, nvec */, double *logdet /* host */, int64_t *info);

/* ---- Newton loop with Q resident on the device -----------------------------------
 * Replaces `_update_hessian!` + `ensure_numeric!` of the Gaussian-approximation loop
 * (src/workspace/gaussian_approximation.jl:103-129: copyto!(ws.Q.nzval, prior_nzval); nzval[map[k]] -= H.nzval[k]).
 * gmrfx_set_prior uploads the prior's values (same CSC order as the pattern) and the Hessian -> Q index map
 * (`diag_idx` or `_sparse_hessian_map`, index_base-based positions into nzval) ONCE; every iterate then sends
 * only the cnt Hessian values and refactorises: Q_k = Q_prior - H_k is formed on the device. */
int32_t gmrfx_set_prior(gmrfx_handle *h, const double *prior_nzval, const int64_t *map, int64_t cnt, int32_t index_base);
int32_t gmrfx_refactorize_update(gmrfx_handle *h, const double *hvals, int64_t *info);
int32_t gmrfx_refactorize_update_dev(gmrfx_handle *h, const double *d_hvals, int64_t *info);
/* One whole Newton iterate (gaussian_approximation.jl:103-129: `_update_hessian!`, `ensure_numeric!`, the solve for the new mean) as
 * one pipelined call: gmrfx_refactorize_update followed by gmrfx_solve, with the forward sweep running beside the factorisation
 * like gmrfx_refactorize_solve; same bits as the two calls. */
int32_t gmrfx_refactorize_update_solve(gmrfx_handle *h, const double *hvals, const double *B, int64_t ldb, int64_t nrhs, double *X, int64_t ldx,
                                       int64_t *info);
int32_t gmrfx_refactorize_update_solve_dev(gmrfx_handle *h, const double *d_hvals, const double *d_B, int64_t ldb, int64_t nrhs, double *d_X,
                                           int64_t ldx, int64_t *info);

/* ---- sharded factorisation (opts.shard_world > 1); SURVEY section 8(e) -------------------------------
 * ONE factorisation over several GPUs, one process each, driven by the host language over its collective library
 * (RCCL through torch.distributed in the Python binding, gmrfx/shard.py). Every process runs the same analysis: the
 * supernodal tree is cut into subtrees dealt to the ranks, and every front ABOVE them (the "top") is owned by one
 * rank of the group whose subtrees it joins -- independent top fronts run on different GPUs, and data crosses ranks
 * only along the tree edges whose two ends have different owners (gmrfx_shard_edges: child, src, dst, level of the
 * parent, the contribution block's place in the arena gmrfx_device_ptr(h, 0), the update vector's rows in
 * gmrfx_device_ptr(h, 3)). X / W rows are laid out identically on all ranks; the contribution-block ARENA has its own
 * layout on every rank (round 6: a rank only gives slots -- shared by lifetime -- to the blocks it produces or receives),
 * so cb_offset / zb_offset / gmrfx_shard_transfers' offset are THIS rank's offsets (-1 when the rank is neither end):
 * the sender builds its view from its own handle's table, the receiver from its own. K = nlevels - shard_level top levels.
 *   refactorisation  gmrfx_refactorize_phase(h, nzval, 0)      the subtrees this rank owns
 *                    for k = 0 .. K-1:  [contribution blocks of the edges with level = shard_level + k: src -> dst]
 *                                       gmrfx_refactorize_phase(h, nzval, 1 + k)   this rank's fronts of top level k
 *   log det Q        all-reduce (sum) of gmrfx_logdet_partial; pivot failures: all-reduce (min) of gmrfx_stats.fail_col
 *   solve (1..64 right-hand sides, device buffers)
 *                    gmrfx_solve_phase(.., 0)                  transpose in + forward sweep over the own subtrees. Of d_B a rank
 *                                                              READS only the rows of the columns it owns (its subtrees' and its
 *                                                              own top fronts': gmrfx_shard_rows kinds 3 and 2 with owner = rank,
 *                                                              through gmrfx_get_perm): B may be row-sharded over the ranks
 *                    for k = 0 .. K-1:  [update vectors W of the level's edges: src -> dst]   gmrfx_solve_phase(.., 100 + k)
 *                    for k = K-1 .. 0:  gmrfx_solve_phase(.., 200 + k)   [x of that level's top fronts: owner -> all,
 *                                       gmrfx_shard_rows(kind 2)]
 *                    gmrfx_solve_phase(.., 2)                  backward sweep over the own subtrees
 *                    [x of every assigned subtree -> rank 0, gmrfx_shard_rows(kind 3)]   gmrfx_solve_phase(.., 3) on rank 0
 *   selected inversion (top-down)   gmrfx_selinv_phase(h, 0, 0, 0) begin; for every level l from the top: (h, 1, l, 0) the
 *                    owners of the PARENTS gather the trailing inverse blocks of the other ranks' fronts at level l;
 *                    [those blocks (zb_offset / cb_count of the edges with child_level = l, in gmrfx_device_ptr(h, 0)):
 *                    owner of the parent -> owner of the child]; (h, 2, l + 1, l) this rank's fronts of level l
 *                    (levels without cross-rank edges may be run as one range); (h, 3, 0, 0) end. The getters
 *                    (gmrfx_selinv_diag / _extract / _dot ...) then return this rank's PART (zeros for the entries of
 *                    other ranks' fronts): sum over the ranks = the selected inverse. */
int32_t gmrfx_refactorize_phase(gmrfx_handle *h, const double *d_nzval, int32_t phase);
int32_t gmrfx_shard_info(const gmrfx_handle *h, int64_t *n_edges, int64_t *n_top_fronts, int64_t *shard_level);
int32_t gmrfx_shard_edges(const gmrfx_handle *h, int64_t *child, int64_t *src, int64_t *dst, int64_t *level,
                          int64_t *cb_offset, int64_t *cb_count, int64_t *w_row0, int64_t *w_nrows,
                          int64_t *zb_offset /* nullable */, int64_t *child_level /* nullable */);
int32_t gmrfx_selinv_phase(gmrfx_handle *h, int32_t what, int32_t level_hi, int32_t level_lo);
int32_t gmrfx_shard_owner(const gmrfx_handle *h, int64_t *owner /* nsuper, >= 0 */, int64_t *is_top /* nsuper, nullable */);
void   *gmrfx_device_ptr(gmrfx_handle *h, int32_t which /* 0: contribution-block arena, 1: factor panels, 2: X, 3: W */);
int32_t gmrfx_solve_phase(gmrfx_handle *h, const double *d_B, int64_t ldb, int64_t nrhs, double *d_X, int64_t ldx, int32_t phase);
/* Backward-only sharded solve, X = P' L^-T Z (`F.UP \ z`, src/workspace/backend.jl:281-284), through the same entry point:
 * phase 10 = take Z in elimination order (a rank reads the rows of its own columns only, as above); 300 + k = backward over top level k [then broadcast that
 * level's x rows, as after 200 + k]; 12 = backward over the own subtrees; [gather on rank 0]; 3 = transpose out.
 *
 * gmrfx_set_stream: the caller's HIP stream becomes the handle's main stream (use_external = 1; 0 restores the handle's
 * own). A sharded driver passes the stream its communication library orders its operations on (torch's current stream for
 * torch.distributed / RCCL): kernels, copies and transfers are then ordered by the stream, and with async_phases = 1 the
 * phase entry points (gmrfx_refactorize_phase / _solve_phase / _selinv_phase) return after enqueueing -- no host-side
 * synchronisation between a phase and the exchange behind it (their HIP-event timings are not collected then). */
int32_t gmrfx_set_stream(gmrfx_handle *h, void *hip_stream, int32_t use_external, int32_t async_phases);
/* DISTRIBUTED TOP FRONTS (sharded handles; csrc/symbolic.h Symbolic::dist_fronts): a top front with at least 4096 columns
 * (GMRFX_DIST_MIN overrides; 0 = never) whose group -- the ranks owning the subtrees below it -- has more than one rank is
 * factored by the WHOLE GROUP: the dense top fronts of a 3-D problem hold most of the flops (cfg 4: the root alone has 47 628
 * columns) and the single-address-space call this replaces, `cholesky!(F, S; check = false)` (src/workspace/backend.jl:165-189),
 * has no notion of it. Panel columns and contribution-block columns are cut into 256-column blocks dealt cyclically over the
 * group (panel block b -> group[b mod g], contribution-block block q -> group[(panel blocks + q) mod g]); a front WITH a contribution
 * block is stored whole by every member, one without (the root) only by its owner (gmrfx_dist_front_block below); sweeps and
 * selected inversion of the front stay on its owner.
 *   gmrfx_shard_dist_fronts: counts[0] fronts, [1] group entries, [2] transfers, [3] world; per front (nullable): supernode,
 *     columns, rows, panel offset in gmrfx_device_ptr(h, 1), panel leading dimension, tree level; gptr / grank: its group.
 *   gmrfx_shard_transfers: every contribution-block transfer of the factorisation, ordered by the parent's level: `count`
 *     doubles (whole columns of `child`'s block) at `offset` of THIS rank's gmrfx_device_ptr(h, 0) (-1 when this rank is neither
 *     src nor dst; the two ends have different offsets) go src -> dst before level `level` is assembled (col0: the first of
 *     these columns). Replaces the cb_offset / cb_count columns of gmrfx_shard_edges for the factorisation.
 *   gmrfx_dist_front_phase(h, d_nzval, front, what, block): 0 = assemble this rank's panel blocks; 1 = factor panel block
 *     `block` (its owner; a no-op elsewhere) [then broadcast columns 256 block .. of the panel inside the group]; 2 = apply
 *     block `block` to this rank's later panel blocks (4 = to block + 1 only, 5 = to the later blocks except block + 1: the
 *     LOOK-AHEAD split -- the owner of block + 1 applies 4, factors and starts the broadcast of block + 1, and everybody
 *     applies 5 while that broadcast is in flight); 3 = this rank's blocks of the contribution block. A no-op on ranks
 *     outside the group. Same kernels, same sums in the same order as an unsharded handle: bit-identical factor. */
int32_t gmrfx_shard_dist_fronts(const gmrfx_handle *h, int64_t *counts /* 4 */, int64_t *front, int64_t *cols, int64_t *rows,
                                int64_t *panel_offset, int64_t *panel_ld, int64_t *level, int64_t *gptr, int64_t *grank);
int32_t gmrfx_shard_transfers(const gmrfx_handle *h, int64_t *child, int64_t *src, int64_t *dst, int64_t *level,
                              int64_t *offset, int64_t *count, int64_t *col0);
int32_t gmrfx_dist_front_phase(gmrfx_handle *h, const double *d_nzval, int32_t front, int32_t what, int32_t block);
/* Block-cyclic STORAGE of a distributed front without trailing rows (the root; round 6): only the front's owner -- which sweeps and
 * inverts it -- stores the whole panel; every other member keeps its own 256-column blocks and a window of two received blocks
 * (cfg 4 at world 8: 18 GB -> 2.5 GB on seven of eight ranks). gmrfx_dist_front_block tells a rank where IT keeps panel block
 * `block` -- offset into gmrfx_device_ptr(h, 1) and doubles (whole columns) --: the buffer it hands to the broadcast of that block
 * (the ends of a broadcast have different offsets). -1 / 0 outside the group. Fronts with a contribution block stay replicated
 * (every member needs all of L21 for its own column blocks of that block). */
int32_t gmrfx_dist_front_block(const gmrfx_handle *h, int32_t front, int32_t block, int64_t *offset, int64_t *count);
/* Profiling aid (handles created under GMRFX_LEVEL_MARK=1): HIP-event time of every tree level of the most recent
 * factorisation (which = 0), forward (1) or backward (2) sweep: ms[0] = the sweep tasks, ms[1 + l] = level l; *count = entries
 * written (0 when the marks are off). gmrfx/shard.py turns them into the TIME bound of a sharding plan (plan_summary). */
int32_t gmrfx_level_times(gmrfx_handle *h, int32_t which, double *ms, int64_t cap, int64_t *count);
int32_t gmrfx_shard_rows(const gmrfx_handle *h, int32_t kind, int64_t *nblocks, int64_t *owner, int64_t *row0, int64_t *nrows,
                         int64_t *level);
int32_t gmrfx_logdet_partial(gmrfx_handle *h, double *out);

/* How the HOST entry points (gmrfx_solve, gmrfx_refactorize_solve, ... with pageable B / X: what `workspace_solve(ws, B::Matrix)`
 * hands over, src/workspace/gmrf_workspace.jl:207-215) cut a column-major n x nrhs array into slices for the handle's
 * page-locked staging ring of 8 slots (download = 0: B in, 1: X out). plan[0] columns per slice, [1] row pieces per column
 * (> 1 once a column exceeds a slice), [2] rows per piece, [3] doubles per ring slot, [4] slices, [5] doubles reserved
 * (= min(slices, 8) x slot). Slice k covers columns [k plan[0], ...) when plan[1] == 1, else column k / plan[1], rows
 * [(k mod plan[1]) plan[2], ...). Arithmetic only (no handle, no device) -- exported so that the invariant "every slice fits
 * its slot, the slices tile the array" is checked on the CPU for sizes no test box holds. */
int32_t gmrfx_host_io_plan(int64_t n, int64_t nrhs, int32_t download, int64_t *plan /* 6 */);

/* Q X = B. Replaces `F \ b` / `F \ B`: src/workspace/backend.jl:191-209. */
int32_t gmrfx_solve(gmrfx_handle *h, const double *B, int64_t ldb, int64_t nrhs, double *X, int64_t ldx);
int32_t gmrfx_solve_dev(gmrfx_handle *h, const double *d_B, int64_t ldb, int64_t nrhs, double *d_X, int64_t ldx);

/* X = P' L^-T Z (sampling). Replaces `F.UP \ z`: src/workspace/backend.jl:281-284,
 * src/solvers/backward_solve.jl:50-60. Batched over nrhs samples (the reference loops). */
int32_t gmrfx_backward_solve(gmrfx_handle *h, const double *Z, int64_t ldz, int64_t nrhs, double *X, int64_t ldx);
int32_t gmrfx_backward_solve_dev(gmrfx_handle *h, const double *d_Z, int64_t ldz, int64_t nrhs, double *d_X, int64_t ldx);

/* +log det Q = 2 sum log L_jj. Replaces `logdet(F)`: src/workspace/backend.jl:211-213
 * (seam A negates: src/solvers/logdet.jl:27-31). */
int32_t gmrfx_logdet(gmrfx_handle *h, double *out);

/* out[v] = (x_v - mu)' Q (x_v - mu) for the nvec columns of the column-major n x nvec X (mu: n values or NULL for a
 * zero mean), on Q's CSC values `nzval` (same order as the pattern given to gmrfx_create; NULL = the values of the
 * last refactorisation when the handle holds them, i.e. after gmrfx_refactorize or gmrfx_refactorize_update).
 * Replaces `dot(r, d.precision * r)` in logpdf(::WorkspaceGMRF, z), src/workspace/workspace_gmrf.jl:288-292, and
 * sqmahal, src/gmrf.jl:94-97; with gmrfx_refactorize_dev + gmrfx_logdet the hyper-parameter loop of
 * docs/src/literate-tutorials/workspace_factorization_reuse.jl:94-102 keeps Q and z in HBM. Only the stored triangle
 * that defines Q (gmrfx_opts.uplo) is read. Does not need a factorisation. The _dev form takes device pointers for
 * nzval, X and mu; out is a host array of nvec doubles in both. nvec <= 65535. */
int32_t gmrfx_quadform(gmrfx_handle *h, const double *nzval, const double *X, int64_t ldx, int64_t nvec,
                       const double *mu, double *out);
int32_t gmrfx_quadform_dev(gmrfx_handle *h, const double *d_nzval, const double *d_X, int64_t ldx, int64_t nvec,
                           const double *d_mu, double *out);

/* ---- dense-operator leg of the separable (Kronecker) path ---------------------------
 * `Q = kron(Q_1, Q_2)` (SeparableModel, src/latent_models/separable.jl:143-156; logdet rule :122-141): a solve / a sample of all
 * n1 n2 unknowns is ONE sweep over the large factor with n1 right-hand sides (the flat vector x[i1 n2 + i2] is its column-major
 * n2 x n1 panel) followed by the small factor applied as a DENSE n1 x n1 operator D (Q_1^-1, or P_1' L_1^-T for samples) to the
 * row-major n1 x n2 result T:  R = D T.  gmrfx_dense_apply_dev computes that product on the FP64 matrix cores with the library's
 * own kernels (csrc/dense.hip; no vendor GEMM). All three arrays are row-major device arrays, T and R must not overlap, and T
 * must be followed by at least 8 readable bytes when n2 is odd (16-byte operand loads). gmrfx_transpose_dev writes the
 * transpose of a row-major rows x cols device array (dst: cols x rows) -- the layout change between the two sweeps when BOTH
 * factors are large. `h` is any numeric handle of the device the arrays live on; both calls return when the result is complete. */
int32_t gmrfx_dense_apply_dev(gmrfx_handle *h, int64_t n1, int64_t n2, const double *d_D, const double *d_T, double *d_R);
int32_t gmrfx_transpose_dev(gmrfx_handle *h, int64_t rows, int64_t cols, const double *d_src, double *d_dst);

/* Takahashi selected inverse; computed lazily once per refactorisation and cached on the
 * device. Replaces SelectedInversion.selinv / selinv_diag: src/workspace/backend.jl:215-257,
 * src/solvers/selinv.jl:70-125. */
int32_t gmrfx_selinv_compute(gmrfx_handle *h);
int32_t gmrfx_selinv_diag(gmrfx_handle *h, double *out /* n, original ordering */);
/* Full selected inverse, original ordering, both triangles, rows sorted: the pattern of
 * (L+L') de-permuted (a superset of pattern(Q)). Two-call protocol: nnz, then fill. */
int32_t gmrfx_selinv_nnz(gmrfx_handle *h, int64_t *nnz);
int32_t gmrfx_selinv_csc(gmrfx_handle *h, int32_t index_base, int64_t *colptr, int64_t *rowval, double *nzval);
/* Sigma on a caller pattern, 0.0 outside the factor pattern. Replaces
 * SelectedInversion.selinv_extract(Z, B): src/workspace/backend.jl:275-279. */
int32_t gmrfx_selinv_extract(gmrfx_handle *h, int64_t ncol, const int64_t *colptr, const int64_t *rowval,
                             int32_t index_base, double *out_nzval);
/* Contractions with the selected inverse, reduced on the device (only the results cross PCIe).
 * gmrfx_selinv_dot: *out = sum over the stored entries of B (n columns, CSC) of Sigma_ij * B_ij = tr(Q^-1 B) for a
 * symmetric B whose pattern lies inside the factor pattern (entries outside it contribute 0). Replaces
 * selinv_dot(b, B) for Float64 B: src/workspace/backend.jl:258-267 (Dual-valued B: use gmrfx_selinv_extract and
 * contract in Julia).
 * gmrfx_selinv_row_diag: out[i] = sum_{p,q in row i} A_ip A_iq Sigma[j_p, j_q] = (A Sigma A')_ii for the m rows of a
 * sparse design matrix given by rows (CSR arrays = the CSC arrays of A'). Replaces _row_diag_AΣAt(A, Σ) /
 * selinv_extract_at(ws, A' * A): src/linear_predictor_marginals.jl:125-165. */
int32_t gmrfx_selinv_dot(gmrfx_handle *h, int64_t ncol, const int64_t *colptr, const int64_t *rowval,
                         const double *nzval, int32_t index_base, double *out);
int32_t gmrfx_selinv_row_diag(gmrfx_handle *h, int64_t m, const int64_t *rowptr, const int64_t *colind,
                              const double *values, int32_t index_base, double *out);
/* The same in two steps for a design matrix whose PATTERN stays fixed while Q changes (the hyper-parameter loop):
 * _plan looks the pairs up once and keeps them on the device (valid for the lifetime of the handle, across
 * refactorisations), _apply sends A's values (same order as colind) and returns the m variances, _free drops the plan. */
int32_t gmrfx_selinv_row_diag_plan(gmrfx_handle *h, int64_t m, const int64_t *rowptr, const int64_t *colind,
                                   int32_t index_base, int64_t *plan);
int32_t gmrfx_selinv_row_diag_apply(gmrfx_handle *h, int64_t plan, const double *values, double *out);
int32_t gmrfx_selinv_row_diag_free(gmrfx_handle *h, int64_t plan);

/* Elimination order actually used (index_base-based), so CHOLMOD can be run on the identical
 * P Q P' (`CHOLMODBackend(Q; ordering = perm)`, src/workspace/backend.jl:147-149). */
int32_t gmrfx_get_perm(const gmrfx_handle *h, int32_t index_base, int64_t *perm);
int32_t gmrfx_get_stats(const gmrfx_handle *h, gmrfx_stats *out, int32_t struct_size);

/* Symbolic structure, for tests and for tools that want the supernodal layout.
 * sizes[0..7] = {nsuper, sum_rows, nnz_l_stored, n_levels, cb_arena, nnz_q_used, 0, 0}. */
int32_t gmrfx_symbolic_sizes(const gmrfx_handle *h, int64_t *sizes);
/* Any pointer may be NULL. super_first: nsuper+1; super_parent: nsuper; row_ptr: nsuper+1;
 * rows: sum_rows (permuted indices); rel: sum_rows (position in parent's row list for the
 * rows below the diagonal block, -1 elsewhere); panel_ptr: nsuper+1 (offset in doubles);
 * panel_ld: nsuper; level: nsuper; q_src/q_dst: nnz_q_used (index into nzval -> offset in
 * panel storage). */
int32_t gmrfx_symbolic_get(const gmrfx_handle *h, int64_t *super_first, int64_t *super_parent,
                           int64_t *row_ptr, int64_t *rows, int64_t *rel, int64_t *panel_ptr,
                           int64_t *panel_ld, int64_t *level, int64_t *q_src, int64_t *q_dst);
/* Copy the numeric factor panels (nnz_l_stored doubles) to the host: tests compare them
 * with the oracle's L (unique for a given permutation). */
/* Sweep tasks (host analysis; testing / inspection): maximal bottom subtrees of the supernodal tree whose forward /
 * backward substitution one workgroup runs on an LDS-resident local vector (csrc/sweep_task.hip), so the update
 * vectors between their fronts never touch HBM. first / last: supernode range of each task (last = root), lrow: for
 * every entry of the supernode row lists, the local row inside its task (-1 outside tasks / for own rows). */
int32_t gmrfx_symbolic_sweep_tasks(const gmrfx_handle *h, int64_t *ntasks, int64_t *rows_cap, int64_t *first,
                                   int64_t *last, int64_t *lrow);
/* The tasks' CHUNKS (csrc/sweep_chunk.hip; host analysis, testing / inspection): every task front cut into column blocks
 * of at most 16 columns, each a narrow front of its own (targets = the panel rows below its diagonal block). nchunks (two
 * values: forward records -- a chunk with more than 128 target rows is several records -- and backward records = chunks) /
 * nrows always; when non-null: task_ptr (2 per task + 2: first forward / backward record of a task), slot (8 per task: chunks in the backward
 * program of row-tile slot 0..3, then the barriers each slot passes behind its last chunk), fwd / bwd (8 per chunk:
 * offset of (first target row, first column) in the factor storage, panel ld, local row of the first own column,
 * columns, target rows, offset of its padded target-row list in rows, barriers passed before it (backward programs),
 * number of the chunk; fwd = task by task in postorder, bwd = per task the programs of slot 0, 1, 2, 3), rows
 * (nrows: local rows of the targets, every list padded with -1 to a multiple of 32). */
int32_t gmrfx_symbolic_sweep_chunks(const gmrfx_handle *h, int64_t *nchunks, int64_t *nrows, int64_t *task_ptr,
                                    int64_t *slot, int64_t *fwd, int64_t *bwd, int64_t *rows);
int32_t gmrfx_get_factor_values(gmrfx_handle *h, double *out);

/* KL-optimal sparse approximate Cholesky factor, L L' ~ Theta^-1: a batch of small dense problems, one
 * workgroup each. A task = local rows R (task_rows[task_rowptr[t] .. task_rowptr[t+1]), in the caller's local order)
 * + the columns of L it fills (task_cols[task_colptr[t] ..)): M = Theta[R, R] + reg I = U'U, and for every member
 * column k with N_k = nnz(L[:, k]) <= |R|: U x = e_{N_k}, nzval[column k] = x[N_k : -1 : 1].
 *   - sparse_approximate_cholesky!(Theta, L), src/kl_cholesky/kl_cholesky.jl:32-55: one task per column, R = the
 *     column's row indices in DESCENDING order, reg = 1e-6;
 *   - sparse_approximate_cholesky(Theta, sc::SupernodeClustering), :74-113: one task per supernode, R = its rows
 *     (descending), member columns = sc.column_indices[s], reg = 1e-8.
 * Theta: dense n x n column-major (leading dimension ldt), host or device memory (theta_on_device); L_colptr: the n+1
 * column pointers of the pattern of L; nzval: host array of nnz(L) doubles, written in L's storage order. A local block
 * that is not positive definite gives GMRFX_ERR_NOT_POSDEF and *info = 1 + task index (the reference throws
 * PosDefException from cholesky!). No handle: errors are reported through gmrfx_last_create_error(). GPU only. */
int32_t gmrfx_kl_cholesky(int64_t n, const double *theta, int64_t ldt, int32_t theta_on_device,
                          const int64_t *L_colptr, int64_t ntasks, const int64_t *task_rowptr, const int64_t *task_rows,
                          const int64_t *task_colptr, const int64_t *task_cols, int32_t index_base, double reg,
                          int32_t device, double *nzval, int64_t *info);

#ifdef __cplusplus
}
#endif
#endif /* GMRFX_H */
