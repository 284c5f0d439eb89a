// symbolic.h -- host-side symbolic analysis for the multifrontal supernodal Cholesky.
// Replaces the analyse phase CHOLMOD runs inside `cholesky(Q; perm)` for the reference
// (src/workspace/backend.jl:147-153): ordering, elimination tree, column counts, supernodes,
// row structures, assembly maps and the level schedule the HIP kernels execute.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace gmrfx {

using i32 = int32_t;
using i64 = int64_t;

struct SymOptions {
    int uplo = 0;            // 0 upper defines Q when both triangles stored, 1 lower
    int ordering = 0;        // 0 auto ND, 1 natural
    int nd_leaf = 0;         // 0 = default
    int relax_cols = 0;      // 0 = default
    double relax_zeros = 0;  // 0 = default
    int merge_wide = -1;     // a child at least this wide that has siblings is never merged into its parent; -1 = default (128), huge = always merge
    int top_by_depth = -1;   // this many levels from the root down are levelled by depth below the root (0: all by height); -1 = default (all)
    int merge_wide_max = -1; // ... and at most this wide (wider fronts are throughput work: merged as usual); -1 = default (4096)
    int coord_dim = 0;
    const double *coords = nullptr;
    int subtree_max = -1;      // max fronts per subtree task; -1 = default (0 = off: bit-identical, measured slower than level batching)
    int small_front_rows = -1; // fronts with r <= this (and <= 64 columns) use the fused LDS kernels; -1 = default (96), 0 = off
    int sweep_task_rows = -1;  // LDS rows of a sweep task's local vector (Symbolic::swt_*); -1 = default, 0 = no sweep tasks
    // multi-GPU sharding of ONE factorisation along the supernodal tree (see Symbolic::owner)
    int shard_rank = 0, shard_world = 1;
    int shard_min_top = 0;         // force at least this many fronts into the TOP of the sharded plan; with shard_world == 1 a value > 0 makes the
                                   // handle a sharded handle of ONE rank (phase entry points, top levels, exchange lists -- all of them empty or
                                   // self-addressed): the protocol's device path on a single GPU (tests/test_rccl_world1.py)
    int dist_min_cols = -1;        // sharded handles: a top front is factored by its whole group when it has at least this many columns
                                   // (Symbolic::dist_fronts); -1 = default (4096), 0 = never
};

// Symmetric adjacency structure without self loops.
struct Graph {
    i64 n = 0;
    std::vector<i64> xadj;
    std::vector<i32> adj;
};

struct Symbolic {
    i64 n = 0;
    i64 nnz_in = 0;               // entries in the caller's CSC
    std::vector<i32> perm, iperm; // perm[k] = original index of pivot k; iperm = inverse
    // supernodes (indices in the permuted ordering)
    i32 nsuper = 0;
    std::vector<i32> sfirst;      // nsuper+1
    std::vector<i32> sparent;     // nsuper, -1 for roots
    std::vector<i32> col2super;   // n
    std::vector<i64> rowptr;      // nsuper+1
    std::vector<i32> rows;        // sum_rows; first c_s entries of a supernode are its own columns
    std::vector<i32> rel;         // sum_rows; for k >= c_s: index of rows[k] in the parent's row list
    std::vector<i64> panelptr;    // nsuper+1, offsets in doubles into the factor storage
    std::vector<i32> ld;          // nsuper, leading dimension of the panel (>= r_s)
    std::vector<i64> cbptr;       // nsuper, offset in doubles of the (r-c)x(r-c) contribution block (arena slots are REUSED:
                                  // a block lives from its front's level to its parent's level)
    std::vector<i64> zbptr;       // nsuper, the same arena as the selected inversion uses it (trailing inverse blocks:
                                  // a block lives from its front's level DOWN to its lowest child's level)
    i64 cb_arena = 0;             // doubles
    std::vector<i64> wptr;        // nsuper+1: first row of front s in the update-vector buffer W ([nsuper] = total rows). Unsharded:
                                  // prefix sum of the trailing rows; sharded: cross-edge children first (same on every rank), then
                                  // this rank's own fronts, nothing for the others
    std::vector<uint8_t> cross_child;   // nsuper: child of an owner-crossing tree edge (sharded handles)
    // supernodal tree
    std::vector<i64> childptr;    // nsuper+1
    std::vector<i32> children;    // children lists (ascending)
    std::vector<i32> level;       // 0 = leaves
    i32 nlevels = 0;
    std::vector<i64> levelptr;    // nlevels+1
    std::vector<i32> levellist;   // supernodes by level; inside a level: small fronts first,
                                  // then big fronts by decreasing column count
    std::vector<i32> level_nsmall;// per level: number of small fronts (prefix of the level's list)
    std::vector<i32> level_ncls;  // per level x 4: small fronts with r <= 48 / 64 / 96 / 128 (in this order in the list)
    std::vector<uint8_t> is_small;// per supernode
    // subtree tasks: maximal all-small subtrees (<= 96 rows per front), each processed by ONE workgroup,
    // fronts in postorder = contiguous supernode id range [sub_first, sub_last]; grouped by row class
    std::vector<i32> sub_first, sub_last;
    i32 nsub_cls[3] = {0, 0, 0};   // tasks whose largest front has <= 48 / 64 / 96 rows (in this order)
    std::vector<uint8_t> in_subtree;
    // level lists over ALL fronts (subtree members included) for the selected inversion: per level
    // small fronts first, then big fronts
    std::vector<i64> sel_levelptr;
    std::vector<i32> sel_levellist, sel_level_nsmall;
    int small_rows = 0;
    // SWEEP TASKS (triangular solves only; the factorisation and the selected inversion keep the level schedule).
    // A task = a maximal subtree whose fronts all have <= 64 columns and whose LOCAL VECTOR -- the subtree's own
    // columns (contiguous in the elimination order: postorder) followed by the trailing rows of its root -- has at
    // most swt_rows rows: ONE workgroup keeps that vector in LDS and runs the whole subtree's forward (resp.
    // backward) substitution on it, so the update vectors between the fronts of the subtree never touch HBM (they
    // are ~60 % of all forward-sweep hand-off traffic of a 2-D SPDE precision). lrow[k] = local row of rows[k] for
    // the trailing rows of task fronts. The sweeps run the tasks first (forward) / last (backward) and the level
    // schedule sw_level* (= the level lists without the task fronts) in between.
    std::vector<i32> swt_first, swt_last;   // per task: first / last (= root) supernode, heaviest task first
    std::vector<uint8_t> in_swt;            // per supernode
    std::vector<i32> lrow;                  // sum_rows (-1 outside tasks / for own rows)
    int swt_rows = 0;
    std::vector<i64> sw_levelptr;
    std::vector<i32> sw_levellist, sw_level_nsmall, sw_level_ncls;
    // CHUNKS of the sweep tasks (round 5, sweep_chunk.hip). Every task front is cut into column chunks of at most 16 columns;
    // a chunk is a narrow front of its own: 16 x 16 diagonal block (its inverse is a diagonal block of L11^-1), targets = the
    // panel rows below it (the own rows of the front's later chunks, then the front's trailing rows), K = its columns. The
    // forward sweep runs a task's chunks one after the other (postorder; ONE barrier per chunk, all row-tile slots of the
    // workgroup on the chunk's target rows). The backward sweep runs the chunk TREE (chain inside a front, the supernodal
    // tree across fronts) depth by depth: chunks of one depth are independent (they write their own rows of the local vector
    // and read rows of finished ancestors), slot q takes the chunks at positions q, q + 4, ... of a depth -- per slot a
    // program of chunks, each with the number of workgroup barriers the slot passes before it (all slots pass the same
    // number in total). Target rows are LOCAL rows of the task, padded to multiples of 32 with -1 (the kernels send those
    // to a spare row), so the kernels need no clamps and no masks.
    struct SwChunk {
        i64 pa;        // offset (doubles) of element (first target row, first column) of the chunk in the factor storage
        i32 ld;        // leading dimension of its panel
        int16_t o;     // local row of its first own column
        int16_t cc;    // columns (1..16)
        i32 nt;        // target rows (panel rows below the chunk's diagonal block)
        i32 lr;        // offset of its target rows in swc_rows (padded to a multiple of 32)
        i32 nbar;      // backward programs: workgroup barriers this slot passes before the chunk
        i32 id;        // number of the chunk (= its packed diagonal tile), 0 .. swc_nchunks - 1
    };
    static constexpr int kSwSlots = 4;
    std::vector<SwChunk> swc_fwd;           // all tasks, task by task, postorder; a chunk with more than 128 target rows as several
                                            // records of <= 128 (same diagonal block: each recomputes the same y)
    std::vector<SwChunk> swc_bwd;           // every chunk once, per task the programs of slot 0, 1, 2, 3 one after the other
    std::vector<i32> swc_ptr, swc_bptr;     // ntasks + 1: first record of a task in swc_fwd / swc_bwd
    i32 swc_nchunks = 0;
    std::vector<i32> swc_slot;              // ntasks x 8: chunks in the program of slot 0..3, then barriers behind the last chunk of slot 0..3
    std::vector<i32> swc_rows;              // padded target-row lists (local rows, -1 = padding)
    // Sharding over `shard_world` ranks (every rank runs the same analysis and gets the same answer). The supernodal
    // tree is cut top-down into subtrees dealt to the ranks (LPT on factorisation flops); the fronts above them -- the
    // TOP, is_top[s] -- are owned ONE BY ONE by a rank of the group whose subtrees they join (the least loaded owner of
    // their children), so independent top fronts run on different GPUs and a contribution block crosses ranks only
    // along a tree edge whose two ends have different owners (shard_edges: the pairwise 4+2+1 pattern of SURVEY 8e).
    // owner[s] >= 0 for every front. Top fronts are pushed to levels >= shard_level, all assigned subtrees live
    // below it, and the level lists of a rank only hold the fronts it executes; the top levels are run one level per
    // phase with the cross-rank edges of that level exchanged in between.
    std::vector<i32> owner;       // nsuper
    std::vector<uint8_t> is_top;  // nsuper
    i32 shard_rank = 0, shard_world = 1, shard_level = 0;   // shard_level = nlevels when there is no plan
    bool shard_plan = false;      // a sharded plan exists: shard_world > 1, or one rank with a forced top (SymOptions::shard_min_top)
    std::vector<i32> shard_edges; // children d with owner[d] != owner[parent(d)], ordered by (level of the parent, d)
    // DISTRIBUTED TOP FRONTS (round 3). The dense fronts at the top of a 3-D problem hold most of the flops (cfg 4: the top
    // three levels are 2.7 of 5.3 s; the root alone 47 628 columns) and each sat on ONE rank. A top front with at least
    // dist_min_cols columns whose GROUP -- the ranks owning the subtrees below it -- has more than one rank is factored by
    // the whole group: its panel columns and the columns of its contribution block are cut into 256-column blocks dealt
    // cyclically over the group (panel block b -> group[b mod g], contribution-block block q -> group[(nbp + q) mod g],
    // nbp = panel blocks). Per panel block: its owner factors the block column (the usual potrf64 / trsm / gemm chain inside
    // the block) and broadcasts it inside the group; every member applies it to its OWN later panel blocks (K = 256 update);
    // when the panel is complete every member computes its OWN blocks of the contribution block. Contribution blocks travel
    // child -> parent by COLUMN RANGES (xf_*): the columns of a child that fall into a block of the parent go from the rank
    // that holds them to the rank that owns the block, into the same arena offset on both ends. Every member stores the whole
    // panel (replicated storage, distributed flops); sweeps and selected inversion of a distributed front stay on owner[s],
    // which holds the complete panel after the last broadcast. Driver: gmrfx/shard.py.
    std::vector<i32> dist_fronts;          // the distributed fronts, ascending (children before parents)
    std::vector<i32> dist_index;           // nsuper: index into dist_fronts or -1
    std::vector<i32> dist_gptr, dist_grank; // ranks of the group of dist_fronts[k]: dist_grank[dist_gptr[k] .. dist_gptr[k+1]) (sorted)
    // every contribution-block transfer of the factorisation, ordered by the level of the parent: columns of child xf_child
    // starting at arena offset xf_off (xf_cnt doubles) go xf_src -> xf_dst
    std::vector<i32> xf_child, xf_src, xf_dst, xf_level, xf_col0;   // (xf_col0: first column of the range in the child's block)
    std::vector<i64> xf_off, xf_cnt;
    bool is_dist(i32 s) const { return !dist_index.empty() && dist_index[s] >= 0; }
    i32 group_size(i32 s) const { const i32 k = dist_index[s]; return dist_gptr[k + 1] - dist_gptr[k]; }
    i32 group_rank(i32 s, i32 idx) const { return dist_grank[dist_gptr[dist_index[s]] + idx]; }
    i32 group_pos(i32 s, i32 rank) const {        // position of `rank` in the group of s, or -1
        const i32 k = dist_index[s];
        for (i32 q = dist_gptr[k]; q < dist_gptr[k + 1]; q++) if (dist_grank[q] == rank) return q - dist_gptr[k];
        return -1;
    }
    i32 panel_blocks(i32 s) const { return (ncols(s) + 255) / 256; }
    // rank holding panel block b / contribution-block column block q of front s
    i32 panel_owner(i32 s, i32 b) const { return is_dist(s) ? group_rank(s, b % group_size(s)) : owner[s]; }
    i32 cb_owner(i32 s, i32 q) const { return is_dist(s) ? group_rank(s, (panel_blocks(s) + q) % group_size(s)) : owner[s]; }
    // BLOCK-CYCLIC STORAGE (round 6): a distributed front WITHOUT trailing rows (the root: the largest panel of a 3-D problem, 18 GB
    // at cfg 4) is stored whole only by its owner (which sweeps / inverts it); every other member of its group keeps its OWN
    // 256-column blocks, one behind the other at full height (block b at local position b / g), and a window of two blocks for
    // the block columns it receives. (A front WITH a contribution block needs all of L21 on every member for its own column
    // blocks of that block: it stays replicated.)
    bool compact_here(i32 s) const { return is_dist(s) && nrows(s) == ncols(s) && owner[s] != shard_rank && group_pos(s, shard_rank) >= 0; }
    i32 own_blocks(i32 s) const { const i32 g = group_size(s), me = group_pos(s, shard_rank), nb = panel_blocks(s); return me < 0 ? 0 : (nb - me + g - 1) / g; }
    i64 compact_span(i32 s) const { return ((i64)(own_blocks(s) + 2) * 256 * ld[s] + 16 + 15) & ~i64(15); }
    i64 compact_window(i32 s, i32 slot) const { return (i64)(own_blocks(s) + slot) * 256 * ld[s]; }      // offset of window `slot` inside the stored span
    bool stored_here(i32 s) const { return !shard_plan || owner[s] == shard_rank || (is_dist(s) && group_pos(s, shard_rank) >= 0); }
    std::vector<i32> shard_sub_root, shard_sub_col0;   // ALL assigned subtrees: root supernode, first column (columns [col0, sfirst[root+1]) are theirs)
    // the caller's pattern (0-based) and which stored triangle defines Q: kept for the quadratic form
    // x'Qx (sqmahal / logpdf), which runs on the caller's CSC values, not on the factor
    std::vector<i64> in_colptr;   // n+1
    std::vector<i32> in_row;      // nnz_in
    int in_use = 0;               // 0: entries with row <= col define Q, 1: row >= col
    // Q scatter map, sorted by destination
    std::vector<i64> qsrc;        // index into caller's nzval
    std::vector<i64> qdst;        // offset in factor storage
    std::vector<i64> qptr;        // nsuper+1: range of (qsrc,qdst) per destination supernode
    // diag offsets (for logdet / selinv diag): offset of L_kk in factor storage
    std::vector<i64> diagoff;     // n
    // statistics
    i64 nnz_l_true = 0, nnz_l_stored = 0, nnz_q_tri = 0, sum_rows = 0;
    i32 max_cols = 0, max_rows = 0;
    i64 n_small = 0, n_big = 0;
    double flops = 0;
    double ms_symbolic = 0;

    i32 ncols(i32 s) const { return sfirst[s + 1] - sfirst[s]; }
    i32 nrows(i32 s) const { return (i32)(rowptr[s + 1] - rowptr[s]); }
};

// Throws std::runtime_error / std::invalid_argument with a message on bad input.
void analyze(i64 n, const i64 *colptr, const i64 *rowval, int index_base, const i64 *user_perm,
             const SymOptions &opt, Symbolic &S);

// ordering.cpp
void build_graph(i64 n, const i64 *colptr, const i64 *rowval, int index_base, Graph &G);
void nested_dissection(const Graph &G, const SymOptions &opt, std::vector<i32> &perm);

}  // namespace gmrfx
