// device.h -- device-resident symbolic structure + numeric drivers (HIP, gfx950).
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <string>
#include <vector>

#include "symbolic.h"

namespace gmrfx { constexpr int kSyrkPipedMinCols = 128; }     // see DeviceFactor::syrk_piped_min_

namespace gmrfx {

// One record per (child -> parent) edge of the assembly tree, in childptr order: everything a
// parent's workgroup needs to gather from that child, in ONE load instead of a chain of dependent
// index loads (children[ch] -> sfirst/rowptr/cbptr/wptr[d] -> ...).
struct EdgeRec {
    int d, md;         // child supernode and its number of trailing rows
    int tptr;          // offset of this edge's tile table in DevSym::etile
    int nown;          // child rows that map into the parent's OWN columns (= etile[tptr])
    long long reloff;  // offset of the child's trailing rows in DevSym::rel
    long long woff;    // DevSym::wptr[d]
    long long cboff;   // DevSym::cbptr[d]
    long long eoff;    // offset of this edge's column table in DevSym::erow
};

// Pointers to the symbolic structure in HBM; passed to kernels by value.
struct DevSym {
    int n, nsuper;
    const int *sfirst;          // nsuper+1
    const long long *rowptr;    // nsuper+1
    const int *rows;            // sum_rows (permuted indices; own columns first)
    const int *rel;             // sum_rows (index in parent's row list for trailing rows)
    const long long *panelptr;  // nsuper+1 (doubles)
    const int *ld;              // nsuper
    const long long *cbptr;     // nsuper (doubles, contribution-block arena)
    const long long *childptr;  // nsuper+1
    const int *children;
    const int *sparent;
    const long long *qptr;      // nsuper+1
    const int *qsrc;            // index into caller nzval
    const int *qdst;            // destination ROW inside the panel column qcol (entries sorted by column, then row)
    const int *qcol;            // front-local destination column (row / column kept apart: a panel may hold more
                                // than 2^31 entries -- the 47 000-column root of a 126^3-node 3-D mesh)
    const int *qcolptr;         // n+1: first entry (index into qsrc / qdst) of every column of L, in elimination order
    const int *erow;            // per child edge, per own column tc of the parent (EdgeRec::eoff + tc): the child's trailing
                                // row that maps to that column, or -1 -- the panel assembly looks up, it does not search
    const long long *wptr;      // nsuper+1: prefix sum of trailing rows (r-c)
    const long long *diagoff;   // n
    const int *perm;            // n
    const EdgeRec *edge;        // one per child edge (same index as `children`)
    // etile[tptr + T], T = 0 .. ceil((r_p - c_p) / 32): first trailing row a of the child whose
    // position in the parent is >= c_p + 32 T (the child's rows falling into the parent's 32-row
    // trailing tile T are [etile[T], etile[T+1]) -- no search at run time)
    const int *etile;
    const int *lrow;            // sum_rows: local row (sweep tasks) of every trailing row of a task front
    const unsigned char *foreign_parent;   // sharded handles: 1 = the parent of this front is owned by another rank
                                           // (its trailing inverse block arrives over the wire: no local gather); else null
};

// One sweep task (Symbolic::swt_*): everything its workgroup needs in ONE load.
struct SweepTask {
    int s0, s1;            // first / last (= root) supernode
    int col0, nt;          // first own column of the subtree, number of own columns (= local rows 0 .. nt-1)
    int mroot, pad;        // trailing rows of the root (= local rows nt .. nt+mroot-1)
    long long p0, p1;      // the subtree's panels in the factor storage (contiguous: postorder)
    long long rp0, rp1;    // its range in the row / local-row lists
    long long rroot;       // offset of the root's trailing rows in DevSym::rows
    long long woff;        // DevSym::wptr[root]
    // chunk form (sweep_chunk.hip; Symbolic::swc_*): first chunk record / number of chunks, the backward programs of the four
    // row-tile slots (chunks per slot, barriers behind a slot's last chunk)
    int c0, nch;           // forward records
    int b0, nbw;           // backward records (every chunk once)
    int scnt[4], sbar[4];
};

// geometry of one front for the panel kernels: in the kernel arguments (FrontArg) or one record per level-list position
// ppa (round 6): where the K operand columns of a panel update live when they are NOT in the panel itself -- the received block of a
// distributed front with block-cyclic storage sits in a window (Device::dist_front_phase); kNoPpa = in the panel, as everywhere else
constexpr long long kNoPpa = (long long)0x8000000000000000ull;
struct FrontArg { int on, s, c, r, ld, first; long long pp; long long ppa = kNoPpa; };
struct FrontView { int s, c, r, ld, first, pad; long long pp; };   // 32 bytes

// Everything a column of the panel assembly (k_assemble_lds) needs to know about its front and the front's first two children,
// in ONE 96-byte record per level-list position (one scalar load) instead of front -> geometry arrays -> edge records.
struct AsmRec {
    long long pp;            // panel offset in the factor storage
    long long ch0;           // first child edge (children beyond the second go the long way)
    int c, ld, first, nch;   // columns, leading dimension, first global column, number of children
    long long reloff[2], cboff[2], eoff[2];     // per child: rel[] offset of its trailing rows, arena offset of its contribution block, erow offset
    int md[2];               // per child: trailing rows
    int pad[2];
};

struct SyrkSplit { int start[9]; };   // tile runs of the 8 XCDs inside a level's tile list

// Everything a workgroup of k_syrk_cb_rec needs for one 64 x 64 contribution-block tile, in ONE 128-byte record (one
// scalar load) instead of four rounds of dependent index loads (tile -> front geometry -> edge records -> tile ranges):
// the levels with narrow fronts spend their time in exactly that chain.
struct SyrkTile {
    long long pa;            // offset of L21 (panel + c rows down) in the factor storage
    long long cb;            // offset of the front's contribution block in the arena
    long long ch0;           // first child edge of the front (children beyond the second go the long way)
    int c, m, ld, nch;       // columns, trailing rows, leading dimension, number of children
    int bi, bj, pad0, pad1;  // tile row / column
    long long reloff[2], cboff[2];   // first two children: relative-row list, contribution block
    int md[2];                       //   trailing rows of the child
    int a0[2], a1[2], b0[2], b1[2];  //   the child's rows that fall into the tile's rows [a0, a1) / columns [b0, b1)
};
static_assert(sizeof(SyrkTile) == 128, "SyrkTile is one 128-byte record");

// The same idea for the forward update of a big front (k_fwd_update_rec): one record per 32-row tile of the trailing rows.
struct FwdTile {
    long long pp;            // offset of the front's panel in the factor storage
    long long xoff;          // first own row of the front in X (= sfirst)
    long long woff;          // first row of the front's update vector in W (= wptr)
    long long ch0;           // first child edge (children beyond the second go the long way)
    int c, r, ld, i0;        // columns, rows, leading dimension, first front row of the tile (>= c)
    int nch, tile;           // number of children, 32-row tile index (for the long way)
    int md[2], a0[2], a1[2]; // first two children: trailing rows, and the rows [a0, a1) that fall into this tile
    int pad[4];
    long long reloff[2], cwoff[2];   // their relative-row lists and update vectors
};
static_assert(sizeof(FwdTile) == 128, "FwdTile is one 128-byte record");

// What k_sel_gather needs about a front and its parent, in one 64-byte record per supernode (selected inversion).
struct SelRec {
    long long rel;           // offset of the front's trailing rows in DevSym::rel
    long long zp;            // parent's panel in Z
    long long zbp;           // parent's trailing inverse block in the arena (selected-inversion layout)
    long long out;           // this front's trailing inverse block
    int m, cp, mp, ldp;      // trailing rows; parent's columns, trailing rows, leading dimension
    int p, foreign, pad[2];  // parent (-1: root), 1 = the block arrives over the wire (sharded)
};
static_assert(sizeof(SelRec) == 64, "SelRec is 64 bytes");

struct LevelInfo {
    int first;        // offset into levellist
    int count;        // fronts in level
    int nsmall;       // prefix handled by the fused small-front kernels
    int ncls[4];      // of which r <= 48 / 64 / 96 / 128 (in this order)
    int max_rows;     // over big fronts
    int max_cols;     // over big fronts (they are sorted by decreasing column count)
    int min_trail = 0; // fewest trailing rows of a big front with any (0: none has)
    std::vector<int> active;  // active[k] = number of big fronts with ncols > k*NB
    int wider[3] = {0, 0, 0}; // big fronts with more than 48 / 32 / 16 columns (first 64-column block: the diagonal-block kernel's shapes)
    // contribution-block SYRK: the level's 64 x 64 tiles in the order they are handed out, cut into one run per XCD
    long long syrk_off = 0;   // offset of the level's tiles in Device::d_syrk_recs_
    SyrkSplit syrk_split{};   // run of XCD x = [start[x], start[x + 1])
    int syrk_per = 0;         // longest run: the grid is 8 * syrk_per workgroups
    // forward update (sweep levels only): 32-row tiles of the trailing rows, same per-XCD hand-out
    long long fwd_off = 0;
    SyrkSplit fwd_split{};
    int fwd_per = 0;
};

class Device {
public:
    Device() = default;
    ~Device();
    Device(const Device &) = delete;
    Device &operator=(const Device &) = delete;

    // Uploads the symbolic structure; allocates factor / arena storage. Throws on HIP errors.
    void init(const Symbolic &S, int device);
    void clone_from(const Device &o, const Symbolic &S);

    void refactorize(const double *nzval, bool on_device);
    // numeric factorisation + solve as ONE pipelined call (gmrfx_refactorize_solve): the forward sweep follows the factorisation
    // up the tree, one level behind, on the side stream
    void refactorize_solve(const double *nzval, bool nz_on_device, const double *B, long long ldb, long long nrhs, double *X, long long ldx,
                           bool b_on_device);
    void refactorize_logpdf(const double *d_nz, const double *d_X, long long ldx, long long nvec, const double *d_mu, double *quad_out,
                            double *logdet_out);
    void refactorize_update_solve(const double *h, bool h_on_device, const double *B, long long ldb, long long nrhs, double *X, long long ldx,
                                  bool b_on_device);
    // Newton loop with Q resident on the device: set_prior uploads the prior's values (and the
    // Hessian -> Q index map) once; refactorize_update forms nz = prior, nz[map[k]] -= h[k] on the device from the
    // cnt Hessian values (host or device) and refactorises -- only h crosses PCIe per iterate.
    void set_prior(const double *prior_nzval, const long long *map, long long cnt);
    void refactorize_update(const double *h, bool on_device);
    // sharded handles: phase 0 = the subtrees this rank owns, phase 1 = the top fronts (rank 0; after the
    // contribution blocks of the other ranks' subtree roots have been written into cb_arena())
    void refactorize_phase(const double *d_nzval, int phase);
    // sharded solve (device buffers, nrhs <= 64): 0 = transpose in + forward over the own subtrees, 1 = the top
    // (forward, then backward; rank 0), 2 = backward over the own subtrees, 3 = transpose out (rank 0, after
    // the owned rows have been gathered). Between the phases the host moves W / X rows (gmrfx/shard.py).
    void solve_phase(const double *d_B, long long ldb, long long nrhs, double *d_X, long long ldx, int phase);
    double *rhs_x() { return d_X_; }
    double *rhs_w() { return d_W_; }
    void ensure_rhs(long long nrhs) { ensure_rhs_capacity(nrhs); }
    // columns per sweep pass for a solve of nrhs right-hand sides: ONE rule for solve() and the pipelined call, so that both give
    // the same bits for the same nrhs. 64 throughout. (Round 6, measured and dropped: 17 .. 32 right-hand sides as TWO 16-column
    // passes side by side on the two lanes -- 17 / 24 / 32 columns 3.01 / 3.35 / 3.71 ms against 3.40 / 3.4 / 3.44 as one pass: two
    // latency-bound passes do not overlap on this runtime (twice the launches through one command processor), the second lane only
    // pays from 65 columns on. What such passes get instead: the narrow level kernels on two right-hand-side tiles, narrow_pass_max().)
    static int pass_width(long long) { return 64; }
    double *cb_arena() { return d_cb_; }
    double *factor_panels() { return d_L_; }
    bool sharded() const { return S_ && S_->shard_plan; }
    // B: column-major n x nrhs (original ordering); mode 0: full solve, 1: backward only (P' L^-T Z)
    void solve(const double *B, long long ldb, long long nrhs, double *X, long long ldx, bool on_device, int mode);
    double logdet();
    // q[v] = (x_v - mu)' Q (x_v - mu), v < nvec, on the device: x_v = d_X + v * ldx, d_mu nullable (zero mean).
    // d_nz = Q's values in the pattern's CSC order (device); nullptr = the values of the last refactorisation
    // when the handle holds them itself (host-pointer refactorize / refactorize_update).
    void quadform(const double *d_nz, const double *d_X, long long ldx, long long nvec, const double *d_mu, double *out_host);
    void selinv_compute();
    // sharded selected inversion, top-down (include/gmrfx.h): 0 = begin, 1 = gather the trailing inverse blocks of the
    // OTHER ranks' fronts at level `hi` whose parents this rank owns (they are then sent to their owners),
    // 2 = this rank's fronts of levels hi-1 .. lo, 3 = end
    void selinv_phase(int what, int hi, int lo);
    void selinv_diag(double *out_host);
    void gather_z(const long long *offsets_host, long long cnt, double *out_host);  // offsets into panel storage, -1 -> 0.0
    // out[g] = sum_{t in segment g} w[t] * Z[off[t]] (off = -1 -> 0), all arrays on the host; Z = selected inverse panels
    void weighted_z_sums(const long long *segptr_host, long long nseg, const long long *off_host, const double *w_host, double *out_host);
    // diag(A Sigma A') with the pair plan of a design matrix kept on the device: only A's values go in and the m
    // results come out per call (the plan depends on the pattern of A and on the symbolic structure only)
    long long rowdiag_plan_create(const long long *segptr_host, long long nseg, const long long *off_host, const int *p_host,
                                  const int *q_host, long long nvals);
    void rowdiag_plan_apply(long long id, const double *values_host, double *out_host);
    void rowdiag_plan_free(long long id);
    void copy_factor(double *out_host);
    // dense-operator leg of the Kronecker path (dense.hip): R = D T, all row-major device arrays; dst = src' (rows x cols)
    void dense_apply(const double *d_D, const double *d_T, double *d_R, long long n1, long long n2);
    void transpose(const double *d_src, double *d_dst, long long rows, long long cols);
    long long fail_col();

    bool factorized = false, selinv_valid = false;
    bool inverse_pending = false;   // dense inverses of the big fronts are computed lazily, on a side stream
    double ms_factor = 0, ms_solve = 0, ms_fwd = 0, ms_bwd = 0, ms_perm = 0, ms_bsolve = 0, ms_logdet = 0, ms_selinv = 0;
    long long last_nrhs = 0;
    double ms_quadform = 0;
    bool syrk_times_pending_ = false;
    double syrk_ms();                 // summed HIP-event time of the SYRK launches of the last factorisation (read lazily)
    double ms_syrk = 0, syrk_flops = 0;   // dominant kernel (k_syrk_cb): live HIP-event time per refactorisation, flops
    long long syrk_launches = 0;
    double bytes_total = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream_ = nullptr;   // the main stream this handle created (`stream` may be the caller's: set_external_stream)
    bool async_phases_ = false;          // sharded phase entry points return after enqueueing (no host synchronisation, no timings)
    hipStream_t stream2 = nullptr;  // side stream: dense-inverse stages overlap the leaf levels of the forward sweep
    hipStream_t stream3 = nullptr;  // second sweep lane (solves with more than 64 right-hand sides)
    hipEvent_t ev_fact_ = nullptr, ev_inv_ = nullptr;

private:
    void upload(const Symbolic &S);
    void ensure_rhs_capacity(long long nrhs);
    void factor_levels(int lo, int hi);
    void forward(int nr, int ldx, int lo, int hi);
    void backward(int nr, int ldx, bool y_in_x2, int hi, int lo);
    template <class T> T *dalloc(size_t count);
    // growable buffer: frees the previous allocation (after the streams have drained) and takes it off the books
    template <class T> T *dregrow(T *old, size_t count);
    std::vector<std::pair<void *, size_t>> allocs_;

    const Symbolic *S_ = nullptr;
    const SelRec *d_selrec_ = nullptr;    // one per supernode
    DevSym ds_{};
    std::vector<LevelInfo> levels_;
    std::vector<LevelInfo> swlevels_;   // the sweeps' level schedule: levels_ without the fronts of the sweep tasks
    int *d_sw_levellist_ = nullptr;
    SweepTask *d_swt_ = nullptr;
    int nswt_ = 0;
    const unsigned char *d_owncol_ = nullptr;   // sharded handles: 1 for the columns of the fronts this rank factors
    const long long *d_zbptr_ = nullptr;   // arena offsets of the trailing inverse blocks (Symbolic::zbptr)
    const int *d_iperm_ = nullptr;   // inverse permutation (original row -> position), used by the RHS transposes
    int *d_levellist_ = nullptr;
    FwdTile *d_fwd_recs_ = nullptr;     // one record per 32-row tile of every big front's update vector (sweep levels)
    double *d_nzp_ = nullptr;           // Q's values in assembly order (nzp[q] = nzval[qsrc[q]]): gathered once per factorisation, one dependent load less per panel column
    long long nq_ = 0;
    hipEvent_t ev_nzp_ = nullptr, ev_nzp0_ = nullptr;
    AsmRec *d_arec_ = nullptr;          // one per position of the level lists (Symbolic::levellist order)
    SyrkTile *d_syrk_recs_ = nullptr;   // one record per contribution-block tile, level by level, in hand-out order
    std::vector<EdgeRec> h_edges_;      // host copies of the edge records / tile tables between upload() and init()
    std::vector<int> h_etile_;
    std::vector<long long> h_wptr_;
    int syrk_piped_min_ = kSyrkPipedMinCols;      // GMRFX_SYRK_PIPED=N: levels whose widest front has >= N columns take the software-pipelined
                                                  // product loop of k_syrk_cb_rec (0: every level; a huge N: none) -- same bits either way
    bool syrk_xcd_ = true;          // GMRFX_SYRK_XCD=0: k_syrk_cb on a plain 3-D grid (front, tile row, tile column) instead
    FrontView *d_frec_ = nullptr, *d_frec2_ = nullptr, *d_sel_frec_ = nullptr;   // geometry records parallel to the level lists
    int *d_levellist2_ = nullptr;   // per level: the big fronts re-ordered [even positions..., odd positions...] (two-stream panel chains)
    // wave tasks (sweep_wave.hip): task ids by LDS class
    static constexpr int kWaveClasses = 2;
    static constexpr int kWaveRows[kWaveClasses] = {160, 288};
    int wave_max_nr_ = 16;        // passes of up to this many right-hand sides use the wave tasks (0: never)
    const int *d_wave_order_ = nullptr;
    int wave_first_[kWaveClasses] = {0, 0}, wave_count_[kWaveClasses] = {0, 0};
    void sweep_tasks(int phase, int nr, int ldx);
public:
    void set_external_stream(hipStream_t s, bool use, bool async);
    int level_times(int phase, double *out, int cap);
    void dist_front_phase(const double *d_nzval, int front, int what, int block);
private:
    void ensure_rdiag();
    bool nzp_pending_ = false;                    // k_gather_values runs on the third stream and the main stream has not waited for it yet
    void ensure_dtile();
    int fwd_front_min_ = 384;                     // the forward twin (k_fwd_front): levels with at least this many such fronts (measured at cfg 2, round 6, one workgroup per
                                                  // front against the three launches, us: level 5 (1472 fronts) 103 / 128, 6 (2271) 167 / 211, 7 (1034) 159 / 207, 8 (513) 103 / 131,
                                                  // 9 (256, 236 of them eligible) 103 + 58 / 113: a level needs about two workgroups per compute unit)
    int bwd_front_min_ = 192;                     // backward step of fronts <= 128 columns wide as one workgroup (sweep_front.hip) on levels with at least this many of them (0: never)
    Symbolic::SwChunk *d_swc_fwd_ = nullptr, *d_swc_bwd_ = nullptr;   // chunk records of the sweep tasks (forward order / backward slot programs)
    int *d_swc_listf_ = nullptr, *d_swc_listb_ = nullptr;               // their target rows as LDS byte offsets, in the lane order of the two kernels
    double *d_dtile_ = nullptr;                   // inverse diagonal blocks of the chunks, packed (k_pack_diag): 256 doubles per chunk
    int nswc_ = 0;
    unsigned long long dtile_for_ = 0;            // d_dtile_ belongs to factorisation number dtile_for_
    double *d_rdiag_ = nullptr;                   // n reciprocals of L's diagonal (+ a zero word): operands of the wave tasks
    unsigned long long factor_serial_ = 1, rdiag_for_ = 0;    // d_rdiag_ belongs to factorisation number rdiag_for_
    std::vector<int> sel_max_cols_, sel_max_trail_;   // per level, over the big fronts of the selected-inversion list
    int *d_dist_list_ = nullptr;  // the distributed fronts (Symbolic::dist_fronts) as a device list for the assembly / SYRK kernels
    bool selinv_begun_ = false;   // sharded selected inversion: phase 0 has run since the last refactorisation (gmrfx_selinv_phase)
    std::vector<hipEvent_t> ev_level_[3];
    int level_slots_[3] = {0, 0, 0};
    void level_event(int phase, int slot);
    bool level_mark_ = false;     // GMRFX_LEVEL_MARK=1: an empty marker kernel in front of every level (profiling aid, tools/sweep_levels.py)
    int *d_sub_first_ = nullptr, *d_sub_last_ = nullptr, *d_sel_levellist_ = nullptr;
    int nsub_cls_[3] = {0, 0, 0};
    double *d_L_ = nullptr, *d_Z_ = nullptr, *d_cb_ = nullptr, *d_nz_ = nullptr;
    const double *nz_src_ = nullptr;
    double *d_prior_ = nullptr, *d_h_ = nullptr;
    long long *d_hmap_ = nullptr;
    long long hmap_cnt_ = 0, hmap_cap_ = 0;   // values of the refactorisation in flight (d_nz_ or the caller's device buffer)
    double *d_X_ = nullptr, *d_X2_ = nullptr, *d_W_ = nullptr, *d_io_ = nullptr, *d_tmp_ = nullptr, *d_part_ = nullptr;
    // dense-inverse stages (inverse.hip)
    int *d_invlist_ = nullptr;
    std::vector<int> inv_nact_;            // per stage: fronts with more than 64<<stage columns
    std::vector<long long *> d_inv_toff_;   // per stage: offsets of the T buffers
    double *d_invT_ = nullptr;
    int inv_maxc_ = 0;
    // Recursive doubling of the diagonal-block inverses stops at inv_cap_ columns (a full inverse of a c-column
    // front costs O(c^3): most of a 3-D solve); wider fronts substitute block by block in the sweeps. The selected
    // inversion needs the full inverses and runs the remaining stages on demand (B from inv_cap_ up).
    int inv_cap_ = 2048;     // measured: cfg 2 flat between 1024 and 4096 (6.18 vs 6.26 ms), 3-D 100^3 solve 44 vs 54 ms
    bool inverse_full_ = false;
    // pipelined factor + solve: per-level "level is factored" events, the dense-inverse stages per level, the highest level a
    // sweep task / small subtree reaches
    bool fused_ = false, fused_fwd_ = false;
    std::vector<hipEvent_t> ev_flevel_;
    int bottom_top_level_ = 0, fused_gate_level_ = 0;
    int *d_inv_lvl_list_ = nullptr;
    std::vector<int> inv_lvl_first_, inv_lvl_maxc_;
    std::vector<std::vector<int>> inv_lvl_nact_;              // [level][stage]
    std::vector<long long *> d_inv_lvl_toff_;                  // per stage: offsets into d_invT_, aligned with d_inv_lvl_list_
    void invert_level(hipStream_t st, int lev);
    bool fact_event_valid_ = false;   // ev_fact_ was recorded at the end of the last factorisation
    int *h_info_ = nullptr;           // pinned: the pivot report of the last factorisation
    double *h_qf_ = nullptr;          // pinned: quadratic forms of refactorize_logpdf
    long long h_qf_cap_ = 0;
    hipEvent_t ev_qf_ = nullptr;
    void prepare_quadform(long long nvec);
    double *h_logdet_ = nullptr;      // pinned: log det of factorisation logdet_for_ (valid once ev_logdet_ has passed)
    unsigned long long logdet_for_ = 0;
    hipEvent_t ev_logdet_ = nullptr;
    void enqueue_logdet(hipStream_t st, bool timed);
    bool info_cached_ = false;
    void invert_diag_blocks(hipStream_t st, int b_from, int b_to);
    void start_inverse_async();
    void wait_inverse();
    int first_multiblock_level_ = 0;
    long long rhs_cap_ = 0, io_cap_ = 0, tmp_cap_ = 0;
    long long *d_yoff_ = nullptr;                 // selected inversion: per-front offsets of the Yh / Yt workspaces
    int *d_fchild_ = nullptr;                     // sharded: other ranks' children of this rank's fronts, by level
    std::vector<int> fc_levelptr_, fc_maxtrail_;  // ranges of d_fchild_ per level, max trailing rows per level
    void selinv_begin();
    void selinv_levels(int hi, int lo);
    // Solves with more than 64 right-hand sides run their 64-column passes on TWO lanes (stream + buffers each):
    // one pass is launch-latency bound (4.6 ms for 1 column, 5.9 for 64), two interleave on the idle CUs.
    double *d_Xb_ = nullptr, *d_X2b_ = nullptr, *d_Wb_ = nullptr;   // lane 1 (lane 0 = d_X_ / d_X2_ / d_W_ on `stream`)
    hipEvent_t ev_lane_[2][5] = {};
    hipEvent_t ev_ready_ = nullptr, ev_ready2_ = nullptr, ev_done1_ = nullptr;
    int *d_info_ = nullptr;
    // host I/O of the pipelined factor + solve call: copy stream, page-locked staging buffer, events
    hipStream_t stream_io_ = nullptr;
    double *h_stage_ = nullptr, *h_nzstage_ = nullptr;
    long long h_stage_cap_ = 0;
    void host_upload_values(const double *nzval);
    hipEvent_t ev_up_ = nullptr, ev_x_ = nullptr;
    std::vector<hipEvent_t> ev_ring_;             // one per slot of the page-locked staging ring (host_upload / host_download)
    void host_io_reserve(long long count);
    void host_upload(const double *B, long long ldb, long long nrhs, double *d_dst);
    void host_download(const double *d_src, long long nrhs, double *X, long long ldx, hipStream_t after);
    struct RowDiagPlan { long long nseg = 0, cnt = 0, nvals = 0; long long *seg = nullptr, *off = nullptr; int *p = nullptr, *q = nullptr; double *vals = nullptr, *out = nullptr; };
    std::vector<RowDiagPlan> rd_plans_;
    // caller's CSC pattern, uploaded on the first quadform call
    long long *d_in_colptr_ = nullptr;
    int *d_in_row_ = nullptr;
    double *d_qf_part_ = nullptr, *d_qf_out_ = nullptr;
    long long qf_cap_ = 0;
    bool nz_held_ = false;    // d_nz_ holds the values of the factorisation
    hipEvent_t ev_[8] = {};
    std::vector<hipEvent_t> ev_syrk_;   // begin/end event of every k_syrk_cb launch
    long long l_size_ = 0, sum_trail_ = 0;
};

void hip_check(hipError_t e, const char *what);

// slicing of a column-major n x nrhs host array for the page-locked staging ring (device.cpp: host_upload / host_download)
struct HostIoPlan { long long cols_per = 1, ppc = 1, rows_per = 0, slot_doubles = 0, nsl = 0, reserve = 0; };
HostIoPlan host_io_plan(long long n, long long nrhs, long long slice_bytes);
HostIoPlan host_io_plan_dir(long long n, long long nrhs, int download);

}  // namespace gmrfx
