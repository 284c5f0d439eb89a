/*
 * gmrfx_rccl.h -- C ABI of libgmrfx_rccl.so: the NATIVE driver of the sharded protocol of include/gmrfx.h over RCCL.
 *
 * ONE factorisation / solve / log-determinant / selected-inverse diagonal over the GPUs of a node, one process (or thread) per
 * GPU, driven from any host language that can `ccall` -- no Python, no torch. The library is a thin layer over the PUBLIC entry
 * points of libgmrfx.so (gmrfx_refactorize_phase / gmrfx_dist_front_phase / gmrfx_solve_phase / gmrfx_selinv_phase /
 * gmrfx_shard_*) and RCCL (rccl.h: ncclSend / ncclRecv groups, ncclBroadcast, ncclAllReduce): it is the reference implementation of
 * INTEGRATION.md section 6, line for line the sequence gmrfx/shard.py drives through torch.distributed. What it replaces in the
 * reference: nothing -- CHOLMOD behind src/workspace/backend.jl:165-209 has one address space; the exchange stands in for it.
 *
 * Usage (every rank):   gmrfx_create(... opts.shard_rank = r, opts.shard_world = W ...)   the same pattern on every rank
 *                       rank 0: gmrfx_rccl_unique_id(id); the host language hands the 128 bytes to the other ranks
 *                       gmrfx_rccl_create(h, W, r, id, NULL, &d)
 *                       gmrfx_rccl_refactorize(d, d_nzval); gmrfx_rccl_solve(d, d_B, n, 64, d_X, n, 1); gmrfx_rccl_logdet(d, &ld, &info)
 * Everything is enqueued on ONE HIP stream (the driver's own, or the caller's): phases, transfers and the next phase are ordered
 * by the stream; the host only blocks where a value comes back (log-determinant, pivot report, selected-inverse diagonal).
 * B may be row-sharded: a rank reads the rows gmrfx_rccl_needed_rows names, nothing else.
 */
#ifndef GMRFX_RCCL_H
#define GMRFX_RCCL_H

#include <stdint.h>

#include "gmrfx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gmrfx_rccl gmrfx_rccl;

/* 128 bytes that name a communicator (ncclGetUniqueId): made by one rank, passed to gmrfx_rccl_create by all of them. */
int32_t gmrfx_rccl_unique_id(void *id128);
/* h: a sharded handle (opts.shard_world = world > 1, or world == 1 with opts.shard_min_top > 0: the one-GPU form the tests run).
 * hip_stream: the stream everything is ordered on (NULL: a stream of the driver's own). The handle is switched to that stream
 * with asynchronous phases (gmrfx_set_stream) for the life of the driver. */
int32_t gmrfx_rccl_create(gmrfx_handle *h, int32_t world, int32_t rank, const void *id128, void *hip_stream, gmrfx_rccl **out);
void    gmrfx_rccl_destroy(gmrfx_rccl *d);
const char *gmrfx_rccl_last_error(const gmrfx_rccl *d);

/* Numeric refactorisation: own subtrees, then per top level the contribution-block column ranges src -> dst (gmrfx_shard_transfers),
 * the distributed fronts of the level (block factor -> broadcast inside the group -> K = 256 updates, with the look-ahead split),
 * the rank's own fronts of the level. Returns after enqueueing; the pivot report rides with gmrfx_rccl_logdet. */
int32_t gmrfx_rccl_refactorize(gmrfx_rccl *d, const double *d_nzval);
/* Q X = B, any number of right-hand sides (passes of 64), device pointers, column-major with leading dimensions. gather != 0: all of
 * X on rank 0; gather == 0: every rank's own d_X holds the rows it owns + every top front's (no transfer behind the backward sweep). */
int32_t gmrfx_rccl_solve(gmrfx_rccl *d, const double *d_B, int64_t ldb, int64_t nrhs, double *d_X, int64_t ldx, int32_t gather);
/* X = P' L^-T Z (`F.UP \ z`, src/workspace/backend.jl:281-284), Z in elimination order. */
int32_t gmrfx_rccl_backward_solve(gmrfx_rccl *d, const double *d_Z, int64_t ldz, int64_t nrhs, double *d_X, int64_t ldx, int32_t gather);
/* log det Q (all-reduce of the ranks' partial sums) and the pivot report (0, or 1 + the first failing column over all ranks). Blocks. */
int32_t gmrfx_rccl_logdet(gmrfx_rccl *d, double *logdet, int64_t *info);
/* diag(Q^-1) in the caller's ordering on every rank (host array of n doubles): the sharded Takahashi recursion, then an all-reduce. */
int32_t gmrfx_rccl_selinv_diag(gmrfx_rccl *d, double *out_host);
/* mask[i] = 1 for the rows of B (caller's ordering) this rank reads in gmrfx_rccl_solve: the masks of the ranks partition 0 .. n-1. */
int32_t gmrfx_rccl_needed_rows(const gmrfx_rccl *d, uint8_t *mask);

#ifdef __cplusplus
}
#endif
#endif /* GMRFX_RCCL_H */
