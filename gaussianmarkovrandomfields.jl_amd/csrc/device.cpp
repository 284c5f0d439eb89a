// device.cpp -- HBM residency of the symbolic structure and the level-scheduled numeric
// drivers (factorise, sweeps, log-determinant, selected inverse). Compiled with hipcc.
#include "device.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <stdexcept>
#include <thread>

#include "kernels.h"

namespace gmrfx {

void hip_check(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string("HIP error in ") + what + ": " + hipGetErrorString(e));
}
#define HC(x) hip_check((x), #x)

static constexpr size_t kPairSlackBytes = 16;
static constexpr size_t kChunkSlack = 16 * 320;      // doubles behind the factor storage (sweep_chunk.hip reads tiles without clamps)
template <class T> T *Device::dalloc(size_t count) {
    // 16 bytes of slack behind every array: kernels that load rows in pairs (16 bytes per lane) may read one element
    // past the last one; the value is never used, but the address must be mapped
    void *p = nullptr;
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T) + kPairSlackBytes;
    HC(hipMalloc(&p, bytes));
    allocs_.push_back({p, bytes});
    bytes_total += (double)bytes;
    return (T *)p;
}

// Grows a buffer that may be replaced during the life of the handle: the NEW buffer is allocated first, so that a failed
// allocation (exception) leaves the old pointer and its capacity valid; callers raise their *_cap_ only after the return.
template <class T> T *Device::dregrow(T *old, size_t count) {
    T *fresh = dalloc<T>(count);
    if (old) {
        HC(hipDeviceSynchronize());       // nothing in flight may still read the old buffer
        for (size_t k = 0; k < allocs_.size(); k++)
            if (allocs_[k].first == (void *)old) {
                bytes_total -= (double)allocs_[k].second;
                allocs_.erase(allocs_.begin() + (long)k);
                break;
            }
        (void)hipFree(old);
    }
    return fresh;
}

Device::~Device() {
    for (size_t k = 0; k < rd_plans_.size(); k++) rowdiag_plan_free((long long)k);
    if (stream) { (void)hipStreamSynchronize(stream); }
    for (auto &p : allocs_) (void)hipFree(p.first);
    for (auto &e : ev_) if (e) (void)hipEventDestroy(e);
    for (auto &v : ev_level_) for (auto &e : v) (void)hipEventDestroy(e);
    for (auto &e : ev_flevel_) (void)hipEventDestroy(e);
    for (auto &l : ev_lane_) for (auto &e : l) if (e) (void)hipEventDestroy(e);
    if (h_info_) (void)hipHostFree(h_info_);
    if (h_stage_) (void)hipHostFree(h_stage_);
    if (h_nzstage_) (void)hipHostFree(h_nzstage_);
    if (ev_up_) (void)hipEventDestroy(ev_up_);
    if (ev_x_) (void)hipEventDestroy(ev_x_);
    for (auto &e : ev_ring_) (void)hipEventDestroy(e);
    if (stream_io_) (void)hipStreamDestroy(stream_io_);
    if (h_logdet_) (void)hipHostFree(h_logdet_);
    if (h_qf_) (void)hipHostFree(h_qf_);
    if (ev_qf_) (void)hipEventDestroy(ev_qf_);
    if (ev_logdet_) (void)hipEventDestroy(ev_logdet_);
    if (ev_ready_) (void)hipEventDestroy(ev_ready_);
    if (ev_ready2_) (void)hipEventDestroy(ev_ready2_);
    if (ev_done1_) (void)hipEventDestroy(ev_done1_);
    if (stream3) (void)hipStreamDestroy(stream3);
    for (auto &e : ev_syrk_) if (e) (void)hipEventDestroy(e);
    if (ev_fact_) (void)hipEventDestroy(ev_fact_);
    if (ev_nzp_) (void)hipEventDestroy(ev_nzp_);
    if (ev_nzp0_) (void)hipEventDestroy(ev_nzp0_);
    if (ev_inv_) (void)hipEventDestroy(ev_inv_);
    if (stream2) (void)hipStreamDestroy(stream2);
    if (own_stream_) (void)hipStreamDestroy(own_stream_);
}

// The caller's stream becomes the main stream (sharded drivers: torch's current stream, so that the library's kernels,
// torch's copies and the RCCL operations torch enqueues are ordered by the stream itself and no host-side synchronisation
// is needed between a phase and the exchange behind it); async: the phase entry points return after enqueueing.
void Device::set_external_stream(hipStream_t s, bool use, bool async) {
    HC(hipSetDevice(device));
    HC(hipStreamSynchronize(stream));
    stream = use ? s : own_stream_;
    async_phases_ = use && async;
}

template <class T, class U> static std::vector<T> conv(const std::vector<U> &v) { return std::vector<T>(v.begin(), v.end()); }

void Device::init(const Symbolic &S, int dev) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) throw std::runtime_error("no HIP device available (libgmrfx has no CPU fallback)");
    if (dev < 0) HC(hipGetDevice(&dev));
    if (dev >= count) throw std::runtime_error("HIP device ordinal out of range");
    device = dev;
    HC(hipSetDevice(device));
    {
        // the main stream carries the dependent chain of the factorisation / sweeps: highest priority; the side
        // stream (dense inverses) only fills idle capacity: lowest
        int lo = 0, hi = 0;
        HC(hipDeviceGetStreamPriorityRange(&lo, &hi));
        // (Measured and settled in round 4, DESIGN.md section 3: no priorities, idle streams between the handle's streams, a CU mask on
        //  the side stream -- none of them helps; the three streams are created back to back: three distinct hardware queues.)
        HC(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, hi));
        own_stream_ = stream;
        HC(hipStreamCreateWithPriority(&stream2, hipStreamNonBlocking, lo));
        HC(hipStreamCreateWithPriority(&stream3, hipStreamNonBlocking, hi));
        if (const char *c = std::getenv("GMRFX_INV_CAP")) {      // testing knob: power of two >= 64
            int v = std::atoi(c), p2 = NB;
            while (p2 < v) p2 *= 2;
            inv_cap_ = p2;
        }
    }
    HC(hipEventCreateWithFlags(&ev_fact_, hipEventDisableTiming));
    HC(hipEventCreateWithFlags(&ev_inv_, hipEventDisableTiming));
    for (auto &ev : ev_) HC(hipEventCreate(&ev));
    for (auto &l : ev_lane_) for (auto &ev : l) HC(hipEventCreate(&ev));
    HC(hipHostMalloc((void **)&h_info_, 2 * sizeof(int), hipHostMallocDefault));
    h_info_[0] = INT_MAX;
    h_info_[1] = 0;
    HC(hipEventCreateWithFlags(&ev_ready_, hipEventDisableTiming));
    HC(hipEventCreateWithFlags(&ev_ready2_, hipEventDisableTiming));
    HC(hipEventCreateWithFlags(&ev_done1_, hipEventDisableTiming));
    upload(S);
}

void Device::upload(const Symbolic &S) {
    S_ = &S;
    const int ns = S.nsuper;
    if (S.nnz_in >= (i64)INT_MAX) throw std::runtime_error("nnz(Q) >= 2^31 not supported by the device scatter map yet");
    auto up = [&](auto *&dst, const auto &host) {
        using T = typename std::remove_const<typename std::remove_reference<decltype(host[0])>::type>::type;
        T *p = dalloc<T>(host.size());
        if (!host.empty()) HC(hipMemcpyAsync(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice, stream));
        dst = p;
    };
    ds_.n = (int)S.n;
    ds_.nsuper = ns;
    const int *ip; const long long *lp;
    up(ip, S.sfirst); ds_.sfirst = ip;
    std::vector<long long> tmp64;
    tmp64 = conv<long long>(S.rowptr); up(lp, tmp64); ds_.rowptr = lp; HC(hipStreamSynchronize(stream));
    up(ip, S.rows); ds_.rows = ip;
    up(ip, S.rel); ds_.rel = ip;
    tmp64 = conv<long long>(S.panelptr); up(lp, tmp64); ds_.panelptr = lp; HC(hipStreamSynchronize(stream));
    up(ip, S.ld); ds_.ld = ip;
    tmp64 = conv<long long>(S.cbptr); up(lp, tmp64); ds_.cbptr = lp; HC(hipStreamSynchronize(stream));
    tmp64 = conv<long long>(S.zbptr); up(lp, tmp64); d_zbptr_ = lp; HC(hipStreamSynchronize(stream));
    {   // selected inversion: one gather record per supernode
        std::vector<SelRec> sr((size_t)ns);
        for (i32 s = 0; s < ns; s++) {
            SelRec t{};
            const i32 p = S.sparent[s];
            t.p = (int)p;
            t.rel = (long long)S.rowptr[s] + S.ncols(s);
            t.m = S.nrows(s) - S.ncols(s);
            t.out = (long long)S.zbptr[s];
            if (p >= 0) {
                t.zp = (long long)S.panelptr[p]; t.zbp = (long long)S.zbptr[p];
                t.cp = S.ncols(p); t.mp = S.nrows(p) - S.ncols(p); t.ldp = (int)S.ld[p];
            }
            t.foreign = (S.shard_plan && p >= 0 && S.owner[p] != S.shard_rank) ? 1 : 0;   // (= DevSym::foreign_parent)
            sr[(size_t)s] = t;
        }
        const SelRec *sp; up(sp, sr); d_selrec_ = sp;
        HC(hipStreamSynchronize(stream));
    }
    tmp64 = conv<long long>(S.childptr); up(lp, tmp64); ds_.childptr = lp; HC(hipStreamSynchronize(stream));
    up(ip, S.children); ds_.children = ip;
    up(ip, S.sparent); ds_.sparent = ip;
    tmp64 = conv<long long>(S.qptr); up(lp, tmp64); ds_.qptr = lp; HC(hipStreamSynchronize(stream));
    {
        std::vector<int> qs(S.qsrc.size()), qd(S.qdst.size()), qc(S.qdst.size());
        for (i32 s = 0; s < ns; s++)
            for (i64 q = S.qptr[s]; q < S.qptr[s + 1]; q++) {
                qs[q] = (int)S.qsrc[q];
                const i64 rel = S.qdst[q] - S.panelptr[s];      // column-major offset inside the panel (may exceed 2^31)
                qc[q] = (int)(rel / S.ld[s]);
                qd[q] = (int)(rel % S.ld[s]);
            }
        up(ip, qs); ds_.qsrc = ip;
        nq_ = (long long)qs.size();
        d_nzp_ = dalloc<double>((size_t)std::max<long long>(nq_, 1));
        HC(hipEventCreateWithFlags(&ev_nzp_, hipEventDisableTiming));
        HC(hipEventCreateWithFlags(&ev_nzp0_, hipEventDisableTiming));
        up(ip, qd); ds_.qdst = ip;
        up(ip, qc); ds_.qcol = ip;
        // entries are sorted by column inside a front: one pointer per column of L replaces a search per panel column
        std::vector<int> qcp((size_t)S.n + 1);
        for (i32 s = 0; s < ns; s++) {
            i64 q = S.qptr[s];
            for (i32 tc = 0; tc < S.ncols(s); tc++) {
                qcp[(size_t)S.sfirst[s] + tc] = (int)q;
                while (q < S.qptr[s + 1] && qc[q] == tc) q++;
            }
            if (q != S.qptr[s + 1]) throw std::runtime_error("scatter map of a front is not sorted by column");
        }
        qcp[(size_t)S.n] = (int)S.qptr[ns];
        up(ip, qcp); ds_.qcolptr = ip;
        HC(hipStreamSynchronize(stream));
    }
    {
        const std::vector<long long> wptr = conv<long long>(S.wptr);     // (Symbolic::wptr: per-rank layout on sharded handles)
        sum_trail_ = wptr[ns];
        up(lp, wptr); ds_.wptr = lp;
        h_wptr_ = wptr;
        HC(hipStreamSynchronize(stream));
    }
    {
        const std::vector<long long> wptr = conv<long long>(S.wptr);
        std::vector<EdgeRec> edges(S.children.size());
        std::vector<int> etile;
        std::vector<int> erow;
        for (i32 p = 0; p < ns; p++) {
            const int cp = S.ncols(p), mp = S.nrows(p) - cp, nT = (mp + 31) / 32;
            for (i64 ch = S.childptr[p]; ch < S.childptr[p + 1]; ch++) {
                const i32 d = S.children[ch];
                const int cd = S.ncols(d), md = S.nrows(d) - cd;
                const i64 reloff = S.rowptr[d] + cd;
                if (etile.size() + (size_t)nT + 1 >= (size_t)INT_MAX) throw std::runtime_error("edge tile table too large");
                EdgeRec e{d, md, (int)etile.size(), 0, (long long)reloff, wptr[d], (long long)S.cbptr[d], (long long)erow.size()};
                int a = 0;
                for (int T = 0; T <= nT; T++) {
                    const int key = cp + 32 * T;
                    while (a < md && S.rel[reloff + a] < key) a++;
                    etile.push_back(a);
                }
                e.nown = etile[e.tptr];
                erow.resize(erow.size() + (size_t)cp, -1);      // which child row lands in own column tc of the parent
                for (int a2 = 0; a2 < e.nown; a2++) erow[(size_t)e.eoff + (size_t)S.rel[reloff + a2]] = a2;
                edges[ch] = e;
            }
        }
        const EdgeRec *ep; up(ep, edges); ds_.edge = ep;
        if (etile.empty()) etile.push_back(0);
        up(ip, etile); ds_.etile = ip;
        if (erow.empty()) erow.push_back(-1);
        up(ip, erow); ds_.erow = ip;
        HC(hipStreamSynchronize(stream));
        h_edges_.swap(edges);      // read once more by the tile records of the contribution-block SYRK (init)
        h_etile_.swap(etile);
    }
    tmp64 = conv<long long>(S.diagoff); up(lp, tmp64); ds_.diagoff = lp; HC(hipStreamSynchronize(stream));
    up(ip, S.perm); ds_.perm = ip;
    up(ip, S.iperm); d_iperm_ = ip;
    if (S.shard_plan) {
        std::vector<unsigned char> own(S.n, 0);
        for (i32 s = 0; s < ns; s++) {
            const bool mine = S.owner[s] == S.shard_rank;
            if (mine) for (i32 j = S.sfirst[s]; j < S.sfirst[s + 1]; j++) own[j] = 1;
        }
        const unsigned char *op; up(op, own); d_owncol_ = op;
        HC(hipStreamSynchronize(stream));
        // selected inversion across ranks: which of my fronts get their trailing inverse block from another rank, and
        // which fronts of other ranks get theirs from me (gathered here, level by level, then sent)
        std::vector<unsigned char> fp(ns, 0);
        std::vector<std::vector<int>> fc(S.nlevels);
        for (i32 s = 0; s < ns; s++) {
            const i32 pr = S.sparent[s];
            if (pr < 0) continue;
            if (S.owner[pr] != S.shard_rank) fp[s] = 1;
            if (S.owner[pr] == S.shard_rank && S.owner[s] != S.shard_rank) fc[S.level[s]].push_back(s);
        }
        const unsigned char *fpp; up(fpp, fp); ds_.foreign_parent = fpp;
        std::vector<int> flat;
        fc_levelptr_.assign(S.nlevels + 1, 0);
        fc_maxtrail_.assign(S.nlevels, 0);
        for (i32 l = 0; l < S.nlevels; l++) {
            for (int s : fc[l]) { flat.push_back(s); fc_maxtrail_[l] = std::max(fc_maxtrail_[l], S.nrows(s) - S.ncols(s)); }
            fc_levelptr_[l + 1] = (int)flat.size();
        }
        const int *fl; up(fl, flat); d_fchild_ = const_cast<int *>(fl);
        HC(hipStreamSynchronize(stream));
    }
    up(ip, S.lrow); ds_.lrow = ip;
    {
        const int *a; up(a, S.sw_levellist); d_sw_levellist_ = const_cast<int *>(a);
        nswt_ = (int)S.swt_first.size();
        const std::vector<long long> wp = conv<long long>(S.wptr);
        std::vector<SweepTask> tk((size_t)nswt_);
        for (int t = 0; t < nswt_; t++) {
            const i32 f = S.swt_first[t], r = S.swt_last[t];
            SweepTask &T = tk[t];
            T.s0 = f; T.s1 = r; T.col0 = S.sfirst[f]; T.nt = S.sfirst[r + 1] - S.sfirst[f];
            T.mroot = S.nrows(r) - S.ncols(r); T.pad = 0;
            T.p0 = S.panelptr[f]; T.p1 = S.panelptr[r + 1];
            T.rp0 = S.rowptr[f]; T.rp1 = S.rowptr[r + 1];
            T.rroot = S.rowptr[r] + S.ncols(r);
            T.woff = wp[r];
            T.c0 = S.swc_ptr[t]; T.nch = S.swc_ptr[t + 1] - S.swc_ptr[t];
            T.b0 = S.swc_bptr[t]; T.nbw = S.swc_bptr[t + 1] - S.swc_bptr[t];
            for (int q = 0; q < 4; q++) { T.scnt[q] = S.swc_slot[(size_t)8 * t + q]; T.sbar[q] = S.swc_slot[(size_t)8 * t + 4 + q]; }
        }
        // chunk records and their target-row lists. The kernels take a target row as the BYTE offset of its column 0 in the
        // local vector (row-major, NC columns, odd rows with their 16-column tiles swapped: sweep_chunk.hip, vbyte) and add
        // the lane's column with one XOR; padding rows go to the spare row. Forward order per 32-row pair: [lk][tile][rr] =
        // pair row 2 (lk + 4 rr) + tile; backward order per 16-row k-tile: [lk][h][e] = row 8 h + 2 lk + e.
        nswc_ = S.swc_nchunks;
        if (nswc_ > 0) {
            const Symbolic::SwChunk *cp; up(cp, S.swc_fwd); d_swc_fwd_ = const_cast<Symbolic::SwChunk *>(cp);
            up(cp, S.swc_bwd); d_swc_bwd_ = const_cast<Symbolic::SwChunk *>(cp);
            const int nc = sweep_chunk_nc(), spare = sweep_chunk_spare_row();
            auto enc = [&](i32 row) { const int r = row < 0 ? spare : row; return (r * nc + (nc >= 32 ? (r & 1) << 4 : 0)) * 8; };
            // (64 entries of slack: the backward kernel requests a fixed number of k-tiles per chunk, the last chunk's past its list)
            std::vector<int> lf(S.swc_rows.size() + 64, 0), lb(S.swc_rows.size() + 64, 0);
            for (size_t b0 = 0; b0 < S.swc_rows.size(); b0 += 32) {
                const i32 *src = S.swc_rows.data() + b0;
                for (int lk = 0; lk < 4; lk++)
                    for (int tl = 0; tl < 2; tl++)
                        for (int rr = 0; rr < 4; rr++) lf[b0 + lk * 8 + tl * 4 + rr] = enc(src[2 * (lk + 4 * rr) + tl]);
                for (int kt = 0; kt < 2; kt++)
                    for (int lk = 0; lk < 4; lk++)
                        for (int h = 0; h < 2; h++)
                            for (int e = 0; e < 2; e++) lb[b0 + 16 * kt + lk * 4 + h * 2 + e] = enc(src[16 * kt + 8 * h + 2 * lk + e]);
            }
            const int *a; up(a, lf); d_swc_listf_ = const_cast<int *>(a);
            up(a, lb); d_swc_listb_ = const_cast<int *>(a);
            HC(hipStreamSynchronize(stream));          // (the host vectors go out of scope)
            d_dtile_ = dalloc<double>((size_t)nswc_ * 256);
        }
        const SweepTask *tp; up(tp, tk); d_swt_ = const_cast<SweepTask *>(tp);
        // wave tasks (sweep_wave.hip): the tasks by LDS class -- rows of the local vector <= kWaveRows[k] -- heaviest first inside
        // a class (the list is already sorted by work); the big class is launched first
        {
            if (const char *e = std::getenv("GMRFX_TASK_MODE")) {       // A/B knob: "wg" / "wave" force one form for every width
                const std::string m(e);
                wave_max_nr_ = m == "wg" ? 0 : m == "wave" ? 64 : wave_max_nr_;
            }
            std::vector<int> ord;
            for (int k = 0; k < kWaveClasses; k++) {
                wave_first_[k] = (int)ord.size();
                for (int t = 0; t < nswt_; t++) {
                    const int rows = tk[t].nt + tk[t].mroot;
                    if (rows <= kWaveRows[k] && (k == 0 || rows > kWaveRows[k - 1])) ord.push_back(t);
                }
                wave_count_[k] = (int)ord.size() - wave_first_[k];
            }
            if ((int)ord.size() != nswt_) throw std::runtime_error("internal: a sweep task exceeds the largest wave-task class");
            for (int t = 0; t < nswt_; t++) ord.push_back(t);       // ... and all of them, heaviest first (passes of at most 4 columns: one launch)
            const int *op; up(op, ord); d_wave_order_ = op;
        }
        HC(hipStreamSynchronize(stream));
    }
    {
        const int *ll; up(ll, S.levellist); d_levellist_ = const_cast<int *>(ll);
        {   // the same lists with every level's big fronts split into their even / odd positions
            std::vector<int> l2(S.levellist.begin(), S.levellist.end());
            for (i32 l = 0; l < S.nlevels; l++) {
                const i64 f = S.levelptr[l] + S.level_nsmall[l], e = S.levelptr[l + 1];
                i64 w = f;
                for (i64 k = f; k < e; k += 2) l2[w++] = S.levellist[k];
                for (i64 k = f + 1; k < e; k += 2) l2[w++] = S.levellist[k];
            }
            const int *l2p; up(l2p, l2); d_levellist2_ = const_cast<int *>(l2p);
            if (const char *e = std::getenv("GMRFX_LEVEL_MARK")) level_mark_ = std::atoi(e) != 0;
            if (const char *e = std::getenv("GMRFX_BWD_FRONT")) bwd_front_min_ = std::atoi(e);    // fronts a level needs for the one-workgroup backward step (0: never)
            if (const char *e = std::getenv("GMRFX_FWD_FRONT")) fwd_front_min_ = std::atoi(e);    // ... and for the one-workgroup forward step
        }
        const int *a; up(a, S.sub_first); d_sub_first_ = const_cast<int *>(a);
        const int *b; up(b, S.sub_last); d_sub_last_ = const_cast<int *>(b);
        for (int k = 0; k < 3; k++) nsub_cls_[k] = S.nsub_cls[k];
        const int *sl; up(sl, S.sel_levellist); d_sel_levellist_ = const_cast<int *>(sl);
        // one 32-byte geometry record per position of the level lists (kernels.h, front_view)
        auto frecs = [&](const std::vector<i32> &lst, FrontView *&dst) {
            std::vector<FrontView> v(lst.size());
            for (size_t k = 0; k < lst.size(); k++) {
                const i32 s = lst[k];
                v[k] = FrontView{(int)s, S.ncols(s), S.nrows(s), (int)S.ld[s], (int)S.sfirst[s], 0, (long long)S.panelptr[s]};
            }
            const FrontView *p; up(p, v); dst = const_cast<FrontView *>(p);
        };
        frecs(S.levellist, d_frec_);
        frecs(S.sel_levellist, d_sel_frec_);
        {   // ... and of the even / odd re-ordering of the big fronts (two panel chains per level)
            std::vector<i32> l2(S.levellist.begin(), S.levellist.end());
            for (i32 l = 0; l < S.nlevels; l++) {
                const i64 f = S.levelptr[l] + S.level_nsmall[l], e = S.levelptr[l + 1];
                i64 w = f;
                for (i64 k = f; k < e; k += 2) l2[w++] = S.levellist[k];
                for (i64 k = f + 1; k < e; k += 2) l2[w++] = S.levellist[k];
            }
            frecs(l2, d_frec2_);
        }
        HC(hipStreamSynchronize(stream));
    }
    HC(hipStreamSynchronize(stream));

    sel_max_cols_.assign(S.nlevels, 0);
    sel_max_trail_.assign(S.nlevels, 0);
    for (i32 l = 0; l < S.nlevels; l++)
        for (i64 k = S.sel_levelptr[l] + S.sel_level_nsmall[l]; k < S.sel_levelptr[l + 1]; k++) {
            const i32 s = S.sel_levellist[k];
            sel_max_cols_[l] = std::max(sel_max_cols_[l], (int)S.ncols(s));
            sel_max_trail_[l] = std::max(sel_max_trail_[l], (int)(S.nrows(s) - S.ncols(s)));
        }
    auto build_levels = [&](std::vector<LevelInfo> &LV, const std::vector<i64> &lptr, const std::vector<i32> &llist,
                            const std::vector<i32> &lnsmall, const std::vector<i32> &lncls, bool count_flops) {
        LV.clear();
        LV.resize(S.nlevels);
        for (i32 l = 0; l < S.nlevels; l++) {
            LevelInfo &L = LV[l];
            L.first = (int)lptr[l];
            L.count = (int)(lptr[l + 1] - lptr[l]);
            L.nsmall = lnsmall[l];
            for (int k = 0; k < 4; k++) L.ncls[k] = lncls[(size_t)l * 4 + k];
            L.max_rows = L.max_cols = 0;
            int max_trail = 0, min_trail = INT_MAX;
            for (int k = L.nsmall; k < L.count; k++) {
                i32 s = llist[L.first + k];
                L.max_rows = std::max(L.max_rows, S.nrows(s));
                L.max_cols = std::max(L.max_cols, S.ncols(s));
                max_trail = std::max(max_trail, S.nrows(s) - S.ncols(s));
                if (S.nrows(s) > S.ncols(s)) min_trail = std::min(min_trail, S.nrows(s) - S.ncols(s));
                const double cc = S.ncols(s), mm = S.nrows(s) - S.ncols(s);
                if (count_flops) syrk_flops += cc * mm * (mm + 1);   // lower triangle of the contribution block: 2 c flops per entry
            }
            int nblk = (L.max_cols + NB - 1) / NB;
            L.active.assign(nblk + 1, 0);
            for (int b = 0; b <= nblk; b++) {
                int cnt = 0;
                for (int k = L.nsmall; k < L.count; k++) {
                    if (S.ncols(llist[L.first + k]) > b * NB) cnt++; else break;  // sorted by decreasing columns
                }
                L.active[b] = cnt;
            }
            for (int q = 0; q < 3; q++) {
                const int lim = 48 - 16 * q;
                int cnt = 0;
                for (int k = L.nsmall; k < L.count; k++) {
                    if (S.ncols(llist[L.first + k]) > lim) cnt++; else break;
                }
                L.wider[q] = cnt;
            }
            L.min_trail = min_trail == INT_MAX ? 0 : min_trail;
            L.active.push_back(max_trail);  // stash: last element = max trailing rows of the level
        }
    };
    syrk_flops = 0;
    build_levels(levels_, S.levelptr, S.levellist, S.level_nsmall, S.level_ncls, true);
    build_levels(swlevels_, S.sw_levelptr, S.sw_levellist, S.sw_level_nsmall, S.sw_level_ncls, false);

    // fronts with more than one 64-column block, by decreasing width: dense-inverse stages
    {
        std::vector<int> il;
        for (i32 s = 0; s < ns; s++) {
            const bool mine = !S.shard_plan || S.owner[s] == S.shard_rank;
            if (S.ncols(s) > NB && mine) il.push_back(s);     // sharded handles only hold the panels they factored
        }
        std::sort(il.begin(), il.end(), [&](int a, int b) { return S.ncols(a) != S.ncols(b) ? S.ncols(a) > S.ncols(b) : a < b; });
        inv_maxc_ = il.empty() ? 0 : S.ncols(il[0]);
        const int *p; up(p, il); d_invlist_ = const_cast<int *>(p);
        long long tmax = 0;
        for (int B = NB; B < inv_maxc_; B *= 2) {
            int na = 0;
            std::vector<long long> off;
            long long acc = 0;
            for (int s : il) {
                if (S.ncols(s) <= B) break;
                off.push_back(acc);
                acc += (long long)((S.ncols(s) + 2 * B - 1) / (2 * B)) * B * B;
                na++;
            }
            inv_nact_.push_back(na);
            const long long *lq; up(lq, off); d_inv_toff_.push_back(const_cast<long long *>(lq));
            HC(hipStreamSynchronize(stream));
            tmax = std::max(tmax, acc);
        }
        d_invT_ = dalloc<double>((size_t)std::max<long long>(tmax, 1));
        HC(hipStreamSynchronize(stream));
        // the same stages level by level (pipelined factor + solve: the inverses of a level are built right behind its
        // factorisation; the levels follow each other on one stream, so they share the workspace)
        std::vector<int> cat;
        inv_lvl_first_.assign(S.nlevels + 1, 0); inv_lvl_maxc_.assign(S.nlevels, 0); inv_lvl_nact_.assign(S.nlevels, {});
        std::vector<std::vector<int>> per(S.nlevels);
        for (int s : il) per[S.level[s]].push_back(s);      // (il is sorted by decreasing width: so is every level's list)
        for (i32 l = 0; l < S.nlevels; l++) {
            inv_lvl_first_[l] = (int)cat.size();
            inv_lvl_maxc_[l] = per[l].empty() ? 0 : S.ncols(per[l][0]);
            cat.insert(cat.end(), per[l].begin(), per[l].end());
        }
        inv_lvl_first_[S.nlevels] = (int)cat.size();
        if (!cat.empty()) {
            const int *cp; up(cp, cat); d_inv_lvl_list_ = const_cast<int *>(cp);
            for (int B = NB; B < inv_maxc_; B *= 2) {
                std::vector<long long> off(cat.size(), 0);
                for (i32 l = 0; l < S.nlevels; l++) {
                    long long acc = 0;
                    int na = 0;
                    for (int s : per[l]) {
                        if (S.ncols(s) <= B) break;
                        off[(size_t)inv_lvl_first_[l] + na] = acc;
                        acc += (long long)((S.ncols(s) + 2 * B - 1) / (2 * B)) * B * B;
                        na++;
                    }
                    inv_lvl_nact_[l].push_back(na);
                }
                const long long *lq; up(lq, off); d_inv_lvl_toff_.push_back(const_cast<long long *>(lq));
            }
            HC(hipStreamSynchronize(stream));
        }
    }
    {   // the highest level a sweep task or a one-workgroup subtree reaches: the bottom of the forward sweep starts behind it
        bottom_top_level_ = 0;
        for (size_t t = 0; t < S.swt_last.size(); t++) bottom_top_level_ = std::max<int>(bottom_top_level_, S.level[S.swt_last[t]]);
        for (i32 s = 0; s < ns; s++) if (S.in_subtree[s]) bottom_top_level_ = std::max<int>(bottom_top_level_, S.level[s]);
        bottom_top_level_ = std::min<int>(bottom_top_level_, std::max<int>(S.nlevels - 1, 0));
        // The bottom of the forward sweep is throughput work whose workgroups hold a CU's LDS for their whole life; started
        // while the factorisation still is throughput work itself (the wide middle of the tree) it only takes the chip away
        // from it. It is held back until the factorisation reaches its latency-bound top: the first level from which on every
        // level has at most 32 fronts (re-swept in round 4: 4 .. 10 levels below the root are within 0.2 ms of it).
        int gate = S.nlevels - 1;
        while (gate > 0 && levels_[gate - 1].count <= 32) gate--;
        fused_gate_level_ = std::min<int>(std::max(gate, bottom_top_level_), std::max<int>(S.nlevels - 1, 0));
    }

    // Contribution-block tiles of every level in hand-out order: front by front (the level list's order), inside a
    // front by 8 x 8-tile squares of the lower triangle (row-major inside a square), then cut into 8 runs of equal
    // estimated cost -- one per XCD. One self-contained record per tile (kernels.hip, k_syrk_cb_rec).
    {
        if (const char *e = std::getenv("GMRFX_SYRK_XCD")) syrk_xcd_ = std::atoi(e) != 0;
        if (const char *e = std::getenv("GMRFX_SYRK_PIPED")) syrk_piped_min_ = std::atoi(e);
        std::vector<SyrkTile> recs;
        std::vector<double> cost;
        constexpr int SQ = 8;
        for (i32 l = 0; l < S.nlevels; l++) {
            LevelInfo &L = levels_[l];
            cost.clear();
            L.syrk_off = (long long)recs.size();
            for (int k = L.nsmall; syrk_xcd_ && k < L.count; k++) {
                const i32 s = S.levellist[L.first + k];
                const int c = S.ncols(s), m = S.nrows(s) - c;
                const int T = (m + 63) / 64, nT = (m + 31) / 32;
                const i64 ch0 = S.childptr[s];
                const int nch = (int)(S.childptr[s + 1] - ch0);
                for (int I = 0; I < T; I += SQ)
                    for (int J = 0; J <= I; J += SQ)
                        for (int bi = I; bi < std::min(I + SQ, T); bi++)
                            for (int bj = J; bj < std::min(J + SQ, bi + 1); bj++) {
                                // k-loop of 3 (diagonal tile) or 4 waves + the gather / epilogue of a tile, in columns of K
                                cost.push_back((double)(c + 48) * (bi == bj ? 3 : 4));
                                SyrkTile t{};
                                t.pa = (long long)S.panelptr[s] + c;
                                t.cb = (long long)S.cbptr[s];
                                t.ch0 = (long long)ch0;
                                t.c = c; t.m = m; t.ld = (int)S.ld[s]; t.nch = nch;
                                t.bi = bi; t.bj = bj;
                                for (int q = 0; q < std::min(nch, 2); q++) {
                                    const EdgeRec &e = h_edges_[ch0 + q];
                                    const int *et = h_etile_.data() + e.tptr;
                                    t.reloff[q] = e.reloff; t.cboff[q] = e.cboff; t.md[q] = e.md;
                                    t.a0[q] = et[2 * bi]; t.a1[q] = et[std::min(2 * bi + 2, nT)];
                                    t.b0[q] = et[2 * bj]; t.b1[q] = et[std::min(2 * bj + 2, nT)];
                                }
                                recs.push_back(t);
                            }
            }
            const size_t nt = cost.size();
            if (nt >= (size_t)INT_MAX / 8) throw std::runtime_error("too many contribution-block tiles in one level");
            double tot = 0, acc = 0;
            for (size_t t = 0; t < nt; t++) tot += cost[t];
            int x = 0;
            L.syrk_split.start[0] = 0;
            for (size_t t = 0; t < nt; t++) {
                while (x < 7 && acc >= tot * (x + 1) / 8) L.syrk_split.start[++x] = (int)t;
                acc += cost[t];
            }
            while (x < 8) L.syrk_split.start[++x] = (int)nt;
            L.syrk_per = 0;
            for (int q = 0; q < 8; q++) L.syrk_per = std::max(L.syrk_per, L.syrk_split.start[q + 1] - L.syrk_split.start[q]);
        }
        if (recs.empty()) recs.push_back(SyrkTile{});
        const SyrkTile *rp; up(rp, recs); d_syrk_recs_ = const_cast<SyrkTile *>(rp);
        HC(hipStreamSynchronize(stream));
        // forward update: one record per 32-row tile of the trailing rows of every big front of the SWEEP levels, front
        // by front, cut into 8 runs of equal cost (a front's tiles share its y and its children's update vectors)
        std::vector<FwdTile> ft;
        for (i32 l = 0; l < S.nlevels; l++) {
            LevelInfo &L = swlevels_[l];
            cost.clear();
            L.fwd_off = (long long)ft.size();
            for (int k = L.nsmall; syrk_xcd_ && k < L.count; k++) {
                const i32 s = S.sw_levellist[L.first + k];
                const int c = S.ncols(s), r = S.nrows(s), m = r - c;
                const i64 ch0 = S.childptr[s];
                const int nch = (int)(S.childptr[s + 1] - ch0);
                for (int T = 0; T * 32 < m; T++) {
                    FwdTile t{};
                    t.pp = (long long)S.panelptr[s]; t.xoff = S.sfirst[s]; t.woff = h_wptr_[s]; t.ch0 = (long long)ch0;
                    t.c = c; t.r = r; t.ld = (int)S.ld[s]; t.i0 = c + 32 * T; t.nch = nch; t.tile = T;
                    for (int q = 0; q < std::min(nch, 2); q++) {
                        const EdgeRec &e = h_edges_[ch0 + q];
                        t.md[q] = e.md; t.reloff[q] = e.reloff; t.cwoff[q] = e.woff;
                        t.a0[q] = h_etile_[(size_t)e.tptr + T]; t.a1[q] = h_etile_[(size_t)e.tptr + T + 1];
                    }
                    ft.push_back(t);
                    cost.push_back((double)(c + 64));
                }
            }
            const size_t nt = cost.size();
            if (nt >= (size_t)INT_MAX / 8) throw std::runtime_error("too many update-vector tiles in one level");
            double tot = 0, acc = 0;
            for (size_t t = 0; t < nt; t++) tot += cost[t];
            int x = 0;
            L.fwd_split.start[0] = 0;
            for (size_t t = 0; t < nt; t++) {
                while (x < 7 && acc >= tot * (x + 1) / 8) L.fwd_split.start[++x] = (int)t;
                acc += cost[t];
            }
            while (x < 8) L.fwd_split.start[++x] = (int)nt;
            L.fwd_per = 0;
            for (int q = 0; q < 8; q++) L.fwd_per = std::max(L.fwd_per, L.fwd_split.start[q + 1] - L.fwd_split.start[q]);
        }
        if (ft.empty()) ft.push_back(FwdTile{});
        const FwdTile *fp; up(fp, ft); d_fwd_recs_ = const_cast<FwdTile *>(fp);
        HC(hipStreamSynchronize(stream));
        std::vector<long long>().swap(h_wptr_);
        {   // panel-assembly records, one per level-list position
            std::vector<AsmRec> ar(S.levellist.size());
            for (size_t k = 0; k < S.levellist.size(); k++) {
                const i32 s = S.levellist[k];
                AsmRec a{};
                a.pp = (long long)S.panelptr[s];
                a.ch0 = (long long)S.childptr[s];
                a.c = S.ncols(s); a.ld = (int)S.ld[s]; a.first = (int)S.sfirst[s];
                a.nch = (int)(S.childptr[s + 1] - S.childptr[s]);
                for (int q = 0; q < std::min(a.nch, 2); q++) {
                    const EdgeRec &e = h_edges_[a.ch0 + q];
                    a.reloff[q] = e.reloff; a.cboff[q] = e.cboff; a.eoff[q] = e.eoff; a.md[q] = e.md;
                }
                ar[k] = a;
            }
            if (ar.empty()) ar.push_back(AsmRec{});
            const AsmRec *ap; up(ap, ar); d_arec_ = const_cast<AsmRec *>(ap);
            HC(hipStreamSynchronize(stream));
        }
        std::vector<EdgeRec>().swap(h_edges_);
        std::vector<int>().swap(h_etile_);
    }
    ev_syrk_.resize(2 * (size_t)S.nlevels);
    for (auto &e : ev_syrk_) HC(hipEventCreate(&e));
    first_multiblock_level_ = S.nlevels;
    // (over the SWEEP lists: the forward sweep waits there for the dense inverses, and a sharded handle's factor lists leave
    //  out the distributed root, which its owner still sweeps)
    for (i32 l = 0; l < S.nlevels; l++) if (swlevels_[l].max_cols > NB) { first_multiblock_level_ = l; break; }

    // INVARIANT (pair loads): the kernels that read operand rows in 16-byte pairs (sweep_front.hip, k_syrk_cb_rec, selinv.hip)
    // may read ONE double past a column's last row; for the last column of the last panel that is element l_size_ of the
    // buffer. Every buffer that holds panels (d_L_, d_Z_, a clone) is therefore allocated through dalloc (16 bytes of
    // slack) and zeroed INCLUDING the slack, so the extra element is mapped and finite (it only ever meets a 0.0 mask).
    l_size_ = S.panelptr[ns];
    static_assert(kPairSlackBytes >= sizeof(double), "pair loads read one element past the end");
    // (the chunk kernels of the sweep tasks read whole 16-column / 16-row tiles from a chunk's first element without clamps: for
    //  the last task panel of the buffer that is up to 16 columns of <= 304 rows past its end -- mapped, zero, never used)
    d_L_ = dalloc<double>((size_t)l_size_ + kChunkSlack);
    d_cb_ = dalloc<double>((size_t)S.cb_arena);
    d_nz_ = dalloc<double>((size_t)S.nnz_in);
    d_info_ = dalloc<int>(2);
    d_part_ = dalloc<double>(1024 + 8);
    HC(hipMemsetAsync(d_L_, 0, ((size_t)l_size_ + kChunkSlack) * sizeof(double) + kPairSlackBytes, stream));
    HC(hipStreamSynchronize(stream));
}

void Device::clone_from(const Device &o, const Symbolic &S) {
    init(S, o.device);
    if (o.factorized) {
        HC(hipMemcpyAsync(d_L_, o.d_L_, (size_t)l_size_ * sizeof(double), hipMemcpyDeviceToDevice, stream));
        HC(hipMemcpyAsync(d_info_, o.d_info_, sizeof(int), hipMemcpyDeviceToDevice, stream));
        HC(hipStreamSynchronize(stream));
        factorized = true;
        inverse_pending = true;   // the copy may predate the source's lazy inverse: recompute on demand
    }
}

// ---- host I/O of the pipelined factor + solve call (gmrfx_refactorize_solve with HOST B / X: what the reference's
// workspace_solve(ws, B::Matrix) hands over, src/workspace/backend.jl:207-209) ---------------------------------------------
// n x nrhs doubles each way (2 x 512 MB at cfg 2: ~9 ms each at the link's 57 GB/s, tools/pcie_probe.py) around a 14 ms step.
// MEASURED (tools/host_io_step.py, cfg 2): the upload must NOT run beside the factorisation. Next to a 512 MB transfer on
// another queue the factorisation takes 18-24 ms instead of 11.7 when the DMA engines move the bytes, and 25-30 ms when ONE
// long kernel of 16 / 48 / 128 workgroups reads the page-locked buffer over PCIe (k_stream_copy; one launch, no DMA packets) --
// the panel chain's ~500 dependent dispatches lose the command processor to whatever else is active (tools/interference.py
// shows the same for every kind of concurrent work). So the transfers are serial -- B in, then the pipelined step, then X out --
// at the full rate of the DMA engines, and what is overlapped is the host side: pageable memory is staged through a page-locked
// buffer of the handle by several host threads (a single thread moves ~10 GB/s, the link 57), slice k's staging beside the DMA
// of slice k-1, and on the way out slice k's copy to the caller's array beside the DMA of slice k+1. Page-locked caller memory
// (hipHostMalloc / hipHostRegister, torch pin_memory) is handed to the DMA engine as it is.
// true for anything the DMA engines take as it is: page-locked host memory -- and device / managed memory handed to a host entry
// point by mistake or convenience (the copy is then device-to-device; a host thread must never memcpy from it)
static bool host_ptr_is_pinned(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost || a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}
static int host_io_threads() {
    static const int nt = [] {
        const unsigned hc = std::thread::hardware_concurrency();
        return (int)std::min<unsigned>(8, std::max<unsigned>(1, hc / 2));
    }();
    return nt;
}
// Pageable host memory is staged through a RING of page-locked slices that belongs to the handle: kIoRing slices of
// kIoSliceBytes (128 MB per handle, whatever the size of the right-hand sides -- a WorkspacePool keeps one handle per thread; round 4
// held n x nrhs doubles, 512 MB at cfg 2). Slice k uses slot k mod kIoRing once slice k - kIoRing has left it.
static constexpr long long kIoSliceBytes = 16ll << 20;     // staging granularity
static constexpr int kIoRing = 8;

// count doubles copied by up to T host threads (used for values that must arrive before anything can start)
static void parallel_memcpy(double *dst, const double *src, long long count) {
    const int T = (int)std::min<long long>(host_io_threads(), std::max<long long>(1, count >> 18));
    if (T <= 1) { std::memcpy(dst, src, (size_t)count * sizeof(double)); return; }
    std::vector<std::thread> th;
    const long long per = (count + T - 1) / T;
    for (int t = 1; t < T; t++) {
        const long long a = t * per, e = std::min(count, a + per);
        if (a < e) th.emplace_back([=] { std::memcpy(dst + a, src + a, (size_t)(e - a) * sizeof(double)); });
    }
    std::memcpy(dst, src, (size_t)std::min(per, count) * sizeof(double));
    for (auto &x : th) x.join();
}

// How a column-major n x nrhs host array is cut into slices of at most `slice_bytes` for the staging ring: whole columns while a
// column fits a slice (cols_per of them), otherwise ppc row pieces per column of rows_per rows each -- a slot never holds more than
// slice_bytes / 8 doubles whatever n is (round 5 clamped the RING but not the slot: from n > 2^21 on, slots 6 / 7 lay behind the
// page-locked buffer). Pure arithmetic, exported as gmrfx_host_io_plan and walked on the CPU (tests/test_cabi.py, tools/sanitize_host.cpp).
HostIoPlan host_io_plan(long long n, long long nrhs, long long slice_bytes) {
    HostIoPlan p;
    const long long slice_doubles = std::max<long long>(1, slice_bytes / (long long)sizeof(double));
    n = std::max<long long>(n, 1);
    if (n <= slice_doubles) {
        p.cols_per = slice_doubles / n; p.ppc = 1; p.rows_per = n;
        p.slot_doubles = p.cols_per * n;
        p.nsl = (nrhs + p.cols_per - 1) / p.cols_per;
    } else {
        p.cols_per = 1; p.ppc = (n + slice_doubles - 1) / slice_doubles; p.rows_per = (n + p.ppc - 1) / p.ppc;
        p.slot_doubles = p.rows_per;
        p.nsl = nrhs * p.ppc;
    }
    p.reserve = std::min<long long>(p.nsl, kIoRing) * p.slot_doubles;
    return p;
}
HostIoPlan host_io_plan_dir(long long n, long long nrhs, int download) { return host_io_plan(n, nrhs, download ? kIoSliceBytes / 2 : kIoSliceBytes); }
// slice k of the plan: columns [j0, j0 + nc) x rows [r0, r0 + nr)
static inline void host_io_slice(const HostIoPlan &p, long long k, long long n, long long nrhs, long long &j0, long long &nc, long long &r0, long long &nr) {
    if (p.ppc == 1) { j0 = k * p.cols_per; nc = std::min(p.cols_per, nrhs - j0); r0 = 0; nr = n; }
    else { j0 = k / p.ppc; nc = 1; r0 = (k % p.ppc) * p.rows_per; nr = std::max<long long>(0, std::min(p.rows_per, n - r0)); }
}

void Device::host_io_reserve(long long count) {
    if (!stream_io_) {
        HC(hipStreamCreateWithFlags(&stream_io_, hipStreamNonBlocking));
        HC(hipEventCreateWithFlags(&ev_up_, hipEventDisableTiming));
        HC(hipEventCreateWithFlags(&ev_x_, hipEventDisableTiming));
    }
    if (count <= h_stage_cap_) return;
    if (h_stage_) { HC(hipStreamSynchronize(stream_io_)); (void)hipHostFree(h_stage_); h_stage_ = nullptr; h_stage_cap_ = 0; }   // (only the copy stream ever touches it)
    HC(hipHostMalloc((void **)&h_stage_, (size_t)count * sizeof(double), hipHostMallocDefault));
    h_stage_cap_ = count;
}

// Q's values from a host array to d_nz_, ahead of the factorisation (nothing else runs yet: plain DMA)
void Device::host_upload_values(const double *nzval) {
    const long long cnt = S_->nnz_in;
    if (host_ptr_is_pinned(nzval)) {
        HC(hipMemcpyAsync(d_nz_, nzval, (size_t)cnt * sizeof(double), hipMemcpyDefault, stream));
        return;
    }
    if (!h_nzstage_) HC(hipHostMalloc((void **)&h_nzstage_, (size_t)std::max<long long>(cnt, 1) * sizeof(double), hipHostMallocDefault));
    HC(hipStreamSynchronize(stream));                 // (an earlier call's copy out of the staging buffer has finished long ago; cheap)
    parallel_memcpy(h_nzstage_, nzval, cnt);
    HC(hipMemcpyAsync(d_nz_, h_nzstage_, (size_t)cnt * sizeof(double), hipMemcpyHostToDevice, stream));
}

// B (host, column-major n x nrhs, leading dimension ldb) -> d_dst (device, column-major, leading dimension n), on the copy
// stream; returns after every copy has been ENQUEUED and ev_up_ recorded behind the last one.
void Device::host_upload(const double *B, long long ldb, long long nrhs, double *d_dst) {
    const long long n = S_->n;
    host_io_reserve(0);
    if (host_ptr_is_pinned(B)) {
        if (ldb == n) HC(hipMemcpyAsync(d_dst, B, (size_t)(n * nrhs) * sizeof(double), hipMemcpyDefault, stream_io_));
        else HC(hipMemcpy2DAsync(d_dst, n * sizeof(double), B, ldb * sizeof(double), n * sizeof(double), nrhs, hipMemcpyDefault, stream_io_));
        HC(hipEventRecord(ev_up_, stream_io_));
        return;
    }
    // column slices; slice k is staged by host thread k mod T into ring slot k mod kIoRing (once the DMA of slice k - kIoRing has
    // left that slot), then handed to the DMA engine
    const HostIoPlan pl = host_io_plan(n, nrhs, kIoSliceBytes);
    const long long slot_doubles = pl.slot_doubles, nsl = pl.nsl;
    host_io_reserve(pl.reserve);
    const int T = (int)std::min<long long>(host_io_threads(), nsl);
    while ((int)ev_ring_.size() < kIoRing) { hipEvent_t e; HC(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ev_ring_.push_back(e); }
    std::vector<std::atomic<int>> posted((size_t)nsl);
    for (auto &a : posted) a.store(0);
    std::vector<std::thread> th;
    std::exception_ptr err;
    std::atomic<bool> failed{false};
    const int dev = device;
    auto work = [&](int t) {
        try {
            HC(hipSetDevice(dev));
            for (long long k = t; k < nsl && !failed.load(); k += T) {
                long long j0, nc, r0, nr;
                host_io_slice(pl, k, n, nrhs, j0, nc, r0, nr);
                const int slot = (int)(k % kIoRing);
                if ((slot + 1) * slot_doubles > h_stage_cap_ || nc * nr > slot_doubles) throw std::runtime_error("host_upload: slice leaves the staging ring");
                if (k >= kIoRing) {         // the slot's previous slice: its copy has been enqueued (posted), now wait until it has run
                    while (!posted[(size_t)(k - kIoRing)].load(std::memory_order_acquire)) { if (failed.load()) return; std::this_thread::yield(); }
                    HC(hipEventSynchronize(ev_ring_[slot]));
                }
                double *st = h_stage_ + slot * slot_doubles;
                for (long long j = 0; j < nc; j++)
                    std::memcpy(st + j * nr, B + (j0 + j) * ldb + r0, (size_t)nr * sizeof(double));
                if (nc * nr > 0) HC(hipMemcpyAsync(d_dst + j0 * n + r0, st, (size_t)(nc * nr) * sizeof(double), hipMemcpyHostToDevice, stream_io_));
                HC(hipEventRecord(ev_ring_[slot], stream_io_));
                posted[(size_t)k].store(1, std::memory_order_release);
            }
        } catch (...) {
            if (!failed.exchange(true)) err = std::current_exception();
        }
    };
    for (int t = 1; t < T; t++) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    if (err) std::rethrow_exception(err);
    HC(hipEventRecord(ev_up_, stream_io_));
}

// d_src (device, column-major n x nrhs, leading dimension n; final once `after` has passed) -> X (host); returns when X is complete
void Device::host_download(const double *d_src, long long nrhs, double *X, long long ldx, hipStream_t after) {
    const long long n = S_->n;
    host_io_reserve(0);
    // the copies are enqueued only once their source is final: a device-to-host copy that has to wait for an event on another
    // stream took 21 ms instead of 9 here (512 MB; the runtime leaves the DMA path for it)
    HC(hipStreamSynchronize(after));
    if (host_ptr_is_pinned(X)) {
        if (ldx == n) HC(hipMemcpyAsync(X, d_src, (size_t)(n * nrhs) * sizeof(double), hipMemcpyDefault, stream_io_));
        else HC(hipMemcpy2DAsync(X, ldx * sizeof(double), d_src, n * sizeof(double), n * sizeof(double), nrhs, hipMemcpyDefault, stream_io_));
        HC(hipStreamSynchronize(stream_io_));
        return;
    }
    // slice k: device -> ring slot k mod kIoRing (once slice k - kIoRing has been copied out of it) -> the caller's array; host
    // thread k mod T does all three steps, up to T transfers in flight
    const HostIoPlan pl = host_io_plan(n, nrhs, kIoSliceBytes / 2);
    const long long slot_doubles = pl.slot_doubles, nsl = pl.nsl;
    host_io_reserve(pl.reserve);
    while ((int)ev_ring_.size() < kIoRing) { hipEvent_t e; HC(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ev_ring_.push_back(e); }
    std::vector<std::atomic<int>> done((size_t)nsl);
    for (auto &a : done) a.store(0);
    const int T = (int)std::min<long long>(std::min<long long>(host_io_threads(), kIoRing), nsl);
    std::vector<std::thread> th;
    std::exception_ptr err;
    std::atomic<bool> failed{false};
    const int dev = device;
    auto work = [&](int t) {
        try {
            HC(hipSetDevice(dev));
            for (long long k = t; k < nsl && !failed.load(); k += T) {
                long long j0, nc, r0, nr;
                host_io_slice(pl, k, n, nrhs, j0, nc, r0, nr);
                const int slot = (int)(k % kIoRing);
                if ((slot + 1) * slot_doubles > h_stage_cap_ || nc * nr > slot_doubles) throw std::runtime_error("host_download: slice leaves the staging ring");
                if (k >= kIoRing)
                    while (!done[(size_t)(k - kIoRing)].load(std::memory_order_acquire)) { if (failed.load()) return; std::this_thread::yield(); }
                double *st = h_stage_ + slot * slot_doubles;
                if (nc * nr > 0) HC(hipMemcpyAsync(st, d_src + j0 * n + r0, (size_t)(nc * nr) * sizeof(double), hipMemcpyDeviceToHost, stream_io_));
                HC(hipEventRecord(ev_ring_[slot], stream_io_));
                HC(hipEventSynchronize(ev_ring_[slot]));
                for (long long j = 0; j < nc; j++)
                    std::memcpy(X + (j0 + j) * ldx + r0, st + j * nr, (size_t)nr * sizeof(double));
                done[(size_t)k].store(1, std::memory_order_release);
            }
        } catch (...) {
            if (!failed.exchange(true)) err = std::current_exception();
        }
    };
    for (int t = 1; t < T; t++) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    if (err) std::rethrow_exception(err);
}

static const int kClsRows[4] = {48, 64, 96, 128};
static const int OBK = 4;   // 64-column blocks per outer (256-column) block of the panel factorisation
static inline int level_max_trail(const LevelInfo &L) { return L.active.back(); }
static inline int level_nblk(const LevelInfo &L) { return (int)L.active.size() - 2; }

void Device::factor_levels(int lo, int hi) {
    if (lo == 0) {
        const int big = INT_MAX;
        HC(hipMemcpyAsync(d_info_, &big, sizeof(int), hipMemcpyHostToDevice, stream));
        syrk_launches = 0;
        // whole small subtrees first (one workgroup each), then the level schedule of everything above
        for (int k = 0, off = 0; k < 3; off += nsub_cls_[k], k++)
            launch_subtree(stream, ds_, 0, d_sub_first_ + off, d_sub_last_ + off, nsub_cls_[k], kClsRows[k], nz_src_, d_L_, d_cb_,
                           d_info_, nullptr, nullptr, 0, 0);
    }
    // Q's values in assembly order, on the second stream beside the first (small-front) levels; the first big-front assembly waits
    // for them. The gather stays PENDING (nzp_pending_) until some call has made the main stream wait for it: a later phase of a
    // sharded factorisation joins it before its first assembly if the phase that started it never did (no big front below the
    // shard level), and a call that started it and never needed it joins it before returning -- the kernel reads the CALLER's
    // values, which must not be in use once the call is over.
    if (lo == 0) {
        HC(hipEventRecord(ev_nzp0_, stream));
        HC(hipStreamWaitEvent(stream3, ev_nzp0_, 0));
        launch_gather_values(stream3, nz_src_, ds_.qsrc, d_nzp_, nq_);
        HC(hipEventRecord(ev_nzp_, stream3));
        nzp_pending_ = true;
    }
    int nsy = (int)syrk_launches;
    for (int lev = lo; lev < hi; lev++) {
        auto &L = levels_[lev];
        if (level_mark_) { launch_level_mark(stream, 3, lev); level_event(0, lev); }
        const int *list = d_levellist_ + L.first + L.nsmall;
        const int nf = L.count - L.nsmall;
        // The small fronts of a level (fused one-workgroup kernels) and its big fronts (assembly -> panel chain -> SYRK)
        // only depend on the levels below, not on each other: when a level has both, the small ones run on the second
        // stream next to the big-front pipeline, and the level ends when both have.
        // A level with small fronts only still has up to four size classes = four launches with a tail each: the widest
        // non-empty class stays on the main stream, the others go to the second one.
        int ncls_used = 0, widest = -1;
        for (int k = 0; k < 4; k++) if (L.ncls[k] > 0) { ncls_used++; widest = k; }
        const bool split_small = L.nsmall > 0 && (nf > 0 || ncls_used > 1);
        if (split_small) {
            HC(hipEventRecord(ev_ready_, stream));
            HC(hipStreamWaitEvent(stream3, ev_ready_, 0));
        }
        for (int k = 0, off = 0; k < 4; off += L.ncls[k], k++) {
            hipStream_t st_small = !split_small ? stream : (nf > 0 || k != widest) ? stream3 : stream;
            launch_factor_small(st_small, ds_, d_levellist_ + L.first + off, L.ncls[k], kClsRows[k], nz_src_, d_L_, d_cb_, d_info_);
        }
        if (nf > 0 && nzp_pending_) { HC(hipStreamWaitEvent(stream, ev_nzp_, 0)); nzp_pending_ = false; }
        launch_assemble(stream, ds_, list, d_arec_ + L.first + L.nsmall, d_nzp_, nf, L.max_cols, L.max_rows, nz_src_, d_L_, d_cb_);
        const int nblk = level_nblk(L);
        // The panel factorisation of a level is a chain of small dependent launches per 64-column block (potrf64 on ONE
        // workgroup per front -> trsm -> gemm): while the diagonal blocks factor, the chip idles. Levels with several
        // wide fronts run TWO independent chains -- the fronts at even / odd positions of the width-sorted list -- on two
        // streams, so one half's trsm / gemm fills the chip while the other half sits in potrf64. Same arithmetic per
        // front: bit-identical factor.
        // (Measured and dropped -- DESIGN.md section 3: a look-ahead diagonal chain with the bulk one step behind on a second stream,
        //  a persistent kernel per 256-column outer block, a pair chain, a rolling SYRK: the chain is bounded by the 64 x 64
        //  factorisation and by single-CU tile rates, not by its dispatches.)
        const bool two = nf >= 2 && nblk >= 4 && !sharded();
        const int nhalf = two ? 2 : 1;
        if (two) {
            HC(hipEventRecord(ev_ready2_, stream));               // (after the assembly of this level's panels)
            HC(hipStreamWaitEvent(stream3, ev_ready2_, 0));
        }
        // (the two chains are enqueued block by block in turn, not one after the other: the host stays ahead of both)
        for (int it = 0; it < nblk * nhalf; it++) {
                const int b = it / nhalf, hf = it % nhalf;
                hipStream_t st = hf == 0 ? stream : stream3;
                const FrontView *hl = two ? d_frec2_ + L.first + L.nsmall + (hf == 0 ? 0 : (nf + 1) / 2) : d_frec_ + L.first + L.nsmall;
                auto act = [&](int bb) { const int a = L.active[bb]; return two ? (hf == 0 ? (a + 1) / 2 : a / 2) : a; };
                if (act(b) <= 0) continue;
                // geometry of the widest front of the (half-)list (fronts are sorted by decreasing width): when it is
                // the only one still active, the panel kernels get it in their arguments (kernels.h, FrontArg)
                FrontArg f1{0, 0, 0, 0, 0, 0, 0}, f0{0, 0, 0, 0, 0, 0, 0};
                if (nf > hf) {
                    const i32 s1 = S_->levellist[L.first + L.nsmall + hf];
                    f1 = FrontArg{1, (int)s1, S_->ncols(s1), S_->nrows(s1), (int)S_->ld[s1], (int)S_->sfirst[s1], (long long)S_->panelptr[s1]};
                }
                const int kb = b * NB;
                if (b == 0 && act(0) > 1) {
                    // first block of a level with many fronts: one launch per width class (the list is sorted by decreasing width;
                    // the even / odd halves of two chains are sorted as well), each in the workgroup shape that fits it
                    auto half = [&](int a) { return two ? (hf == 0 ? (a + 1) / 2 : a / 2) : a; };
                    const int cut[5] = {0, half(L.wider[0]), half(L.wider[1]), half(L.wider[2]), act(0)};
                    const int wcls[4] = {64, 48, 32, 16};
                    for (int q = 0; q < 4; q++)
                        if (cut[q + 1] > cut[q])
                            launch_potrf64(st, ds_, hl + cut[q], cut[q + 1] - cut[q], kb, d_L_, d_info_, f0, std::min(wcls[q], L.max_cols));
                } else
                    launch_potrf64(st, ds_, hl, act(b), kb, d_L_, d_info_, act(b) == 1 ? f1 : f0, std::min(NB, L.max_cols - kb));
                {
                    // first block of a level with many fronts: the fronts at most 32 columns wide (the tail of the sorted list) go to
                    // the narrow-block kernel (same arithmetic on half the registers: more resident waves)
                    constexpr int narrow_min = 256;
                    auto half = [&](int a) { return two ? (hf == 0 ? (a + 1) / 2 : a / 2) : a; };
                    const int cut32 = b == 0 ? half(L.wider[1]) : act(b);
                    const int nnarrow = act(b) - cut32;
                    if (b == 0 && narrow_min > 0 && nnarrow >= narrow_min) {
                        if (cut32 > 0) launch_trsm(st, ds_, hl, cut32, kb, 0, L.max_rows - kb - 1, d_L_, nullptr, nullptr, f0);
                        launch_trsm_narrow(st, hl + cut32, nnarrow, kb, L.max_rows - kb - 1, d_L_);
                    } else
                        launch_trsm(st, ds_, hl, act(b), kb, 0, L.max_rows - kb - 1, d_L_, nullptr, nullptr, act(b) == 1 ? f1 : f0);
                }
                // two-level blocking: K = 64 updates only inside the current 256-column block, the
                // rest of the panel once per block with K = 256
                const int J1 = (b / OBK + 1) * OBK;   // first 64-block of the next 256-column block
                if (b + 1 < nblk && b + 1 < J1 && act(b + 1) > 0)
                    launch_gemm_nt(st, ds_, hl, act(b + 1), kb, NB, kb + NB, J1 * NB, L.max_rows - kb - NB,
                                   std::min(J1 * NB, L.max_cols) - kb - NB, d_L_, act(b + 1) == 1 ? f1 : f0);
                if (b + 1 == J1 && J1 < nblk && act(J1) > 0)
                    launch_gemm_nt(st, ds_, hl, act(J1), (J1 - OBK) * NB, OBK * NB, J1 * NB, INT_MAX,
                                   L.max_rows - J1 * NB, L.max_cols - J1 * NB, d_L_, act(J1) == 1 ? f1 : f0);
            }
        if (two) {
            HC(hipEventRecord(ev_done1_, stream3));
            HC(hipStreamWaitEvent(stream, ev_done1_, 0));
        }
        if (nf > 0 && level_max_trail(L) > 0) {
            HC(hipEventRecord(ev_syrk_[2 * nsy], stream));
            // levels of HUGE fronts (3-D problems): the children's extend-add alone, then the product on 128 x 128 staged tiles
            const bool huge = syrk_xcd_ && !sharded() && level_max_trail(L) >= 4096 && L.max_cols >= 1024;
            // levels of wide fronts (the product dominates the tile): the software-pipelined product loop (kernels.hip, k_syrk_cb_rec<true>;
            // same sums in the same order: a level's choice does not show in the bits)
            if (syrk_xcd_) launch_syrk_cb_recs(stream, ds_, d_syrk_recs_ + L.syrk_off, L.syrk_split, L.syrk_per, d_L_, d_cb_, huge ? 1 : 0, L.max_cols >= syrk_piped_min_);
            else launch_syrk_cb(stream, ds_, list, nf, level_max_trail(L), d_L_, d_cb_);
            if (huge) launch_syrk_big(stream, ds_, list, nf, level_max_trail(L), d_L_, d_cb_);
            HC(hipEventRecord(ev_syrk_[2 * nsy + 1], stream));
            nsy++;
        }
        if (split_small && !two) {        // (with two chains the join before the SYRK already covered the small fronts)
            HC(hipEventRecord(ev_done1_, stream3));
            HC(hipStreamWaitEvent(stream, ev_done1_, 0));
        }
        if (fused_) HC(hipEventRecord(ev_flevel_[lev], stream));      // the panels of this level are final: its sweep may start
    }
    syrk_launches = nsy;
    if (lo == 0 && nzp_pending_) { HC(hipStreamWaitEvent(stream, ev_nzp_, 0)); nzp_pending_ = false; }     // (see above)
    if (level_mark_) level_event(0, hi);
}

// The dense inverses are only needed by the sweeps and the selected inversion of the big
// fronts, never by logdet: they are computed lazily (first solve / selinv after a
// refactorisation) on a side stream, so that in a refactorise+solve step they overlap the
// small-front levels at the bottom of the forward sweep.
void Device::start_inverse_async() {
    if (!inverse_pending) return;
    // ev_fact_ = "the factor is final": recorded at the end of the factorisation, so that whatever the caller has put on
    // the main stream since then (the transpose of the right-hand sides) does not hold the inverses back
    if (!fact_event_valid_) HC(hipEventRecord(ev_fact_, stream));
    fact_event_valid_ = false;
    HC(hipStreamWaitEvent(stream2, ev_fact_, 0));
    invert_diag_blocks(stream2, NB, inv_cap_);
    HC(hipEventRecord(ev_inv_, stream2));
    inverse_pending = false;
    inverse_full_ = inv_maxc_ <= inv_cap_;
}
void Device::wait_inverse() { HC(hipStreamWaitEvent(stream, ev_inv_, 0)); }

void Device::invert_diag_blocks(hipStream_t stream, int b_from, int b_to) {
    int stage = 0;
    for (int B = NB; B < std::min(inv_maxc_, b_to); B *= 2, stage++) {
        if (B < b_from) continue;
        const int na = inv_nact_[stage];
        if (na <= 0) break;
        launch_inv_stage(stream, ds_, d_invlist_, na, B, inv_maxc_, 1, d_L_, d_invT_, d_inv_toff_[stage]);
        launch_inv_stage(stream, ds_, d_invlist_, na, B, inv_maxc_, 2, d_L_, d_invT_, d_inv_toff_[stage]);
    }
}

void Device::invert_level(hipStream_t st, int lev) {
    if (inv_lvl_first_.empty() || inv_lvl_maxc_[lev] <= NB) return;
    int stage = 0;
    for (int B = NB; B < std::min(inv_lvl_maxc_[lev], inv_cap_); B *= 2, stage++) {
        const int na = inv_lvl_nact_[lev][stage];
        if (na <= 0) break;
        launch_inv_stage(st, ds_, d_inv_lvl_list_ + inv_lvl_first_[lev], na, B, inv_lvl_maxc_[lev], 1, d_L_, d_invT_, d_inv_lvl_toff_[stage] + inv_lvl_first_[lev]);
        launch_inv_stage(st, ds_, d_inv_lvl_list_ + inv_lvl_first_[lev], na, B, inv_lvl_maxc_[lev], 2, d_L_, d_invT_, d_inv_lvl_toff_[stage] + inv_lvl_first_[lev]);
    }
}

// Numeric factorisation + solve in ONE call, pipelined (gmrfx_refactorize_solve; the reference does both inside one call as
// well: workspace_solve = ensure_numeric! -> refactorize!, then backend_solve, src/workspace/gmrf_workspace.jl:170-178, 207-215).
// The top of the factorisation is a chain of small dependent launches that leaves most of the chip idle, the bottom of the
// forward sweep is throughput work that only needs the BOTTOM of the factor: the forward sweep (transpose in, sweep tasks,
// then level by level: dense-inverse stages of the level, assembly, triangular product, update) runs on the low-priority
// side stream and follows the factorisation up the tree -- level l starts when the event "level l is factored" has passed.
// When the root has been factored only the root's own forward step is left; the backward sweep follows on the main stream.
// Same kernels, same operands, same order per front as refactorize() + solve(): bit-identical results.
void Device::refactorize_solve(const double *nzval, bool nz_on_device, const double *B, long long ldb, long long nrhs, double *X, long long ldx_out,
                               bool b_on_device) {
    HC(hipSetDevice(device));
    if (sharded()) throw std::invalid_argument("sharded handle: use the phase entry points (gmrfx/shard.py)");
    if (nrhs <= 0 || stream != own_stream_) {      // nothing to pipeline / the caller's stream: the plain sequence
        refactorize(nzval, nz_on_device);
        if (nrhs > 0) solve(B, ldb, nrhs, X, ldx_out, b_on_device, 0);
        return;
    }
    const long long n = S_->n;
    const int nl = (int)levels_.size();
    const double *src = nzval;
    if (!nz_on_device) {
        host_upload_values(nzval);
        src = d_nz_;
    }
    nz_held_ = (src == d_nz_);
    nz_src_ = src;
    ensure_rhs_capacity(nrhs);
    const double *dB = B;
    double *dXo = X;
    long long ldin = ldb, ldout = ldx_out;
    if (!b_on_device) {
        const long long need = n * nrhs;
        if (need > io_cap_) { const long long cap = std::max(need, 2 * io_cap_); d_io_ = dregrow(d_io_, (size_t)cap); io_cap_ = cap; }
        dB = d_io_; dXo = d_io_; ldin = n; ldout = n;
    }
    while ((int)ev_flevel_.size() < nl + 1) { hipEvent_t e; HC(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ev_flevel_.push_back(e); }
    factor_serial_++;
    // host right-hand sides: in FRONT of the factorisation (see the measurements above host_upload): the transfers are serial --
    // B in, the pipelined step, X out -- never beside the factorisation
    if (!b_on_device) {
        host_upload(B, ldb, nrhs, d_io_);
        HC(hipStreamWaitEvent(stream, ev_up_, 0));
    }
    HC(hipEventRecord(ev_[0], stream));
    HC(hipEventRecord(ev_ready_, stream));                 // the side stream starts behind the uploads / whatever precedes this call
    struct Flag { bool &f; ~Flag() { f = false; } } f1{fused_}, f2{fused_fwd_};
    fused_ = true;
    factor_levels(0, nl);
    fused_ = false;
    HC(hipEventRecord(ev_[1], stream));
    HC(hipEventRecord(ev_fact_, stream));
    fact_event_valid_ = true;
    HC(hipMemcpyAsync(h_info_, d_info_, sizeof(int), hipMemcpyDeviceToHost, stream));
    factorized = true;
    selinv_valid = false;
    // ---- first pass (pass_width(): up to 64 columns, 16 for 17 .. 32 right-hand sides -- the widths solve() uses, so that the two
    // forms of the step give the same bits): forward sweep on the side stream, behind the level events
    const int PW = pass_width(nrhs);
    const int nr = (int)std::min<long long>(PW, nrhs), ldx = nr;
    hipEvent_t *ev = ev_lane_[0];
    {
        const hipStream_t main_stream = stream;
        struct Restore { Device &D; hipStream_t st; ~Restore() { D.stream = st; } } restore{*this, main_stream};
        stream = stream2;
        HC(hipStreamWaitEvent(stream, ev_ready_, 0));
        if (!b_on_device) HC(hipStreamWaitEvent(stream, ev_up_, 0));
        HC(hipEventRecord(ev[0], stream));
        launch_permute(stream, d_iperm_, (int)n, const_cast<double *>(dB), ldin, d_X_, nr, ldx, 0);
        HC(hipEventRecord(ev[1], stream));
        fused_fwd_ = true;
        forward(nr, ldx, 0, nl);
        fused_fwd_ = false;
        HC(hipEventRecord(ev_inv_, stream));             // every level's inverses exist: later solves pass wait_inverse() at once
        HC(hipEventRecord(ev[2], stream));
    }
    inverse_pending = false;
    inverse_full_ = inv_maxc_ <= inv_cap_;
    enqueue_logdet(stream2, false);                     // behind the forward sweep on the side stream: beside the backward sweep
    HC(hipStreamWaitEvent(stream, ev[2], 0));
    backward(nr, ldx, true, nl, 0);
    HC(hipEventRecord(ev[3], stream));
    launch_permute(stream, d_iperm_, (int)n, dXo, ldout, d_X_, nr, ldx, 1);
    HC(hipEventRecord(ev[4], stream));
    // ---- further passes: the factor is complete, plain sweeps on the main stream
    for (long long j0 = PW; j0 < nrhs; j0 += PW) {
        const int nr2 = (int)std::min<long long>(PW, nrhs - j0);
        launch_permute(stream, d_iperm_, (int)n, const_cast<double *>(dB) + j0 * ldin, ldin, d_X_, nr2, nr2, 0);
        forward(nr2, nr2, 0, nl);
        backward(nr2, nr2, true, nl, 0);
        launch_permute(stream, d_iperm_, (int)n, dXo + j0 * ldout, ldout, d_X_, nr2, nr2, 1);
    }
    HC(hipEventRecord(ev_lane_[1][0], stream));
    if (!b_on_device) host_download(d_io_, nrhs, X, ldx_out, stream);
    HC(hipStreamSynchronize(stream));
    info_cached_ = true;
    HC(hipGetLastError());
    float tf = 0, a = 0, b = 0, c = 0, d = 0, tail = 0;
    HC(hipEventElapsedTime(&tf, ev_[0], ev_[1]));
    HC(hipEventElapsedTime(&a, ev[0], ev[1]));
    HC(hipEventElapsedTime(&b, ev_[1], ev[2]));          // what is left of the forward sweep once the factor is complete
    HC(hipEventElapsedTime(&c, ev[2], ev[3]));
    HC(hipEventElapsedTime(&d, ev[3], ev[4]));
    HC(hipEventElapsedTime(&tail, ev_[1], ev_lane_[1][0]));
    ms_factor = tf;
    ms_perm = a + d; ms_fwd = std::max(b, 0.0f); ms_bwd = b >= 0 ? c : c + b;
    ms_solve = tail;                                     // device time behind the factorisation: ms_factor + ms_solve = the step
    syrk_times_pending_ = true;
    last_nrhs = nrhs;
}

void Device::refactorize(const double *nzval, bool on_device) {
    HC(hipSetDevice(device));
    const double *src = nzval;
    if (!on_device) {
        HC(hipMemcpyAsync(d_nz_, nzval, (size_t)S_->nnz_in * sizeof(double), hipMemcpyHostToDevice, stream));
        src = d_nz_;
    }
    nz_held_ = (src == d_nz_);
    // a device-resident nzval is read in place (the Q scatter happens inside the assembly kernels, and
    // this call only returns once they have finished): no private copy
    nz_src_ = src;
    if (sharded()) throw std::invalid_argument("sharded handle: use gmrfx_refactorize_phase (two phases with an exchange in between)");
    factor_serial_++;
    HC(hipEventRecord(ev_[0], stream));
    factor_levels(0, (int)levels_.size());
    HC(hipEventRecord(ev_[1], stream));
    HC(hipEventRecord(ev_fact_, stream));
    fact_event_valid_ = true;
    inverse_pending = true;
    // the pivot report travels with the factorisation (pinned host word): no blocking copy after the synchronisation
    HC(hipMemcpyAsync(h_info_, d_info_, sizeof(int), hipMemcpyDeviceToHost, stream));
    HC(hipStreamSynchronize(stream));
    info_cached_ = true;
    HC(hipGetLastError());
    float ms = 0;
    HC(hipEventElapsedTime(&ms, ev_[0], ev_[1]));
    ms_factor = ms;
    syrk_times_pending_ = true;      // the per-launch event times are only read when somebody asks for the statistics
    factorized = true;
    selinv_valid = false;
}

void Device::refactorize_phase(const double *d_nzval, int phase) {
    HC(hipSetDevice(device));
    if (!sharded()) throw std::invalid_argument("gmrfx_refactorize_phase needs a sharded handle (shard_world > 1, or shard_min_top > 0)");
    const int nl = (int)levels_.size();
    const int split = std::min<int>(S_->shard_level, nl);
    info_cached_ = false;
    fact_event_valid_ = false;
    factor_serial_++;
    HC(hipEventRecord(ev_[0], stream));
    if (phase == 0) {               // the subtrees this rank owns
        nz_src_ = d_nzval;
        nz_held_ = false;       // the values live in the caller's device buffer
        factorized = false;
        factor_levels(0, split);
    } else {                        // top level split + phase - 1: the fronts of that level this rank owns
        const int lev = split + phase - 1;
        if (lev >= nl) throw std::invalid_argument("refactorize phase beyond the last level");
        factor_levels(lev, lev + 1);
    }
    HC(hipEventRecord(ev_[1], stream));
    if (!async_phases_) {
        HC(hipStreamSynchronize(stream));
        HC(hipGetLastError());
        float ms = 0;
        HC(hipEventElapsedTime(&ms, ev_[0], ev_[1]));
        ms_factor = phase == 0 ? ms : ms_factor + ms;
    }
    selinv_begun_ = false;
    if (split + phase >= nl) { factorized = true; selinv_valid = false; inverse_pending = true; }   // last phase done
}

// Distributed top fronts (symbolic.h): block phases of one front's factorisation, driven by gmrfx/shard.py between the
// broadcasts. 256-column blocks dealt cyclically over the front's group; the same kernels and the same sums in the same order
// as the level loop uses for a front of its own (potrf64 -> trsm -> K = 64 update inside the block; K = 256 update of the later
// blocks; children gathered + L21 L21' per 64 x 64 tile of the contribution block), so the factor equals the unsharded one.
void Device::dist_front_phase(const double *d_nzval, int front, int what, int block) {
    HC(hipSetDevice(device));
    if (front < 0 || front >= S_->nsuper || !S_->is_dist(front)) throw std::invalid_argument("not a distributed front of this handle");
    const i32 R = front;
    const int g = S_->group_size(R), me = S_->group_pos(R, S_->shard_rank);
    if (me < 0) return;
    const int c = S_->ncols(R), r = S_->nrows(R);
    const int nob = S_->panel_blocks(R);
    const FrontArg fa{1, (int)R, c, r, (int)S_->ld[R], (int)S_->sfirst[R], (long long)S_->panelptr[R]};
    // Block-cyclic STORAGE (Symbolic::compact_here: a front without trailing rows on a member that is not its owner): this rank holds
    // its own blocks one behind the other + a window of two received blocks. The kernels keep addressing columns globally: block b is
    // handed to them under the pseudo panel base that puts its columns where they are stored, and the K operand of a panel update
    // (the block column just received, or an own one) under a base of its own (FrontArg::ppa).
    const bool compact = S_->compact_here(R);
    const long long ldR = S_->ld[R], ppl = S_->panelptr[R];
    auto blk_pp = [&](int b) -> long long { return compact ? ppl - 256LL * (b - b / g) * ldR : ppl; };
    auto a_pp = [&](int b) -> long long {
        if (!compact) return kNoPpa;
        return b % g == me ? blk_pp(b) : ppl + S_->compact_window(R, b & 1) - 256LL * b * ldR;
    };
    if (!d_dist_list_) {
        d_dist_list_ = dalloc<int>(S_->dist_fronts.size());
        HC(hipMemcpyAsync(d_dist_list_, S_->dist_fronts.data(), S_->dist_fronts.size() * sizeof(int), hipMemcpyHostToDevice, stream));
    }
    const int *list = d_dist_list_ + S_->dist_index[R];
    if (what == 0) {
        if (!d_nzval) throw std::invalid_argument("d_nzval is null");
        launch_assemble_cyclic(stream, ds_, list, c, d_nzval, d_L_, d_cb_, g, me, compact);
    } else if (what == 1) {
        if (block < 0 || block >= nob) throw std::invalid_argument("distributed front: block out of range");
        if (block % g != me) return;
        FrontArg fb = fa;
        fb.pp = blk_pp(block);
        const int b0 = block * OBK, b1 = std::min(b0 + OBK, (c + NB - 1) / NB);
        for (int b = b0; b < b1; b++) {
            const int kb = b * NB;
            launch_potrf64(stream, ds_, nullptr, 1, kb, d_L_, d_info_, fb);
            launch_trsm(stream, ds_, nullptr, 1, kb, 0, r - kb - 1, d_L_, nullptr, nullptr, fb);
            if (b + 1 < b1)
                launch_gemm_nt(stream, ds_, nullptr, 1, kb, NB, kb + NB, b1 * NB, r - kb - NB, std::min(b1 * NB, c) - kb - NB, d_L_, fb);
        }
    } else if (what == 2 || what == 4 || what == 5) {
        // 2: block -> all my later blocks; 4: -> block + 1 only (look-ahead: its owner factors and broadcasts it next, while
        // everybody applies `block` to the rest); 5: -> my later blocks except block + 1. 4 then 5 = 2: same sums, same order per entry.
        if (block < 0 || block >= nob) throw std::invalid_argument("distributed front: block out of range");
        const int k0 = block * 256, K = std::min(256, c - k0);
        const int jlo = what == 5 ? block + 2 : block + 1, jhi = what == 4 ? std::min(block + 2, nob) : nob;
        for (int j = jlo; j < jhi; j++) {
            if (j % g != me) continue;
            const int c0 = j * 256;
            FrontArg fj = fa;
            fj.pp = blk_pp(j);
            fj.ppa = a_pp(block);
            launch_gemm_nt(stream, ds_, nullptr, 1, k0, K, c0, c0 + 256, r - c0, std::min(256, c - c0), d_L_, fj);
        }
    } else if (what == 3) {
        if (r > c) launch_syrk_cb_cyclic(stream, ds_, list, r - c, d_L_, d_cb_, g, me, nob);
    } else throw std::invalid_argument("distributed front phase must be 0 (assemble), 1 (factor block), 2 / 4 / 5 (apply block: all / next / rest) or 3 (contribution block)");
    if (!async_phases_) HC(hipStreamSynchronize(stream));
    HC(hipGetLastError());
}

void Device::set_prior(const double *prior_nzval, const long long *map, long long cnt) {
    HC(hipSetDevice(device));
    const long long nnz = S_->nnz_in;
    for (long long k = 0; k < cnt; k++)
        if (map[k] < 0 || map[k] >= nnz) throw std::invalid_argument("Hessian index map points outside the stored pattern of Q");
    if (!d_prior_) d_prior_ = dalloc<double>((size_t)nnz);
    HC(hipMemcpyAsync(d_prior_, prior_nzval, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice, stream));
    if (cnt > hmap_cap_ || !d_hmap_ || !d_h_) {
        hmap_cap_ = 0;      // (if the second allocation throws, the next call regrows both)
        d_hmap_ = dregrow(d_hmap_, (size_t)std::max<long long>(cnt, 1));
        d_h_ = dregrow(d_h_, (size_t)std::max<long long>(cnt, 1));
        hmap_cap_ = cnt;
    }
    hmap_cnt_ = cnt;
    if (cnt > 0) HC(hipMemcpyAsync(d_hmap_, map, (size_t)cnt * sizeof(long long), hipMemcpyHostToDevice, stream));
    HC(hipStreamSynchronize(stream));
}
void Device::refactorize_update(const double *h, bool on_device) {
    HC(hipSetDevice(device));
    if (!d_prior_) throw std::invalid_argument("gmrfx_set_prior has not been called");
    const double *dh = h;
    if (!on_device && hmap_cnt_ > 0) {
        HC(hipMemcpyAsync(d_h_, h, (size_t)hmap_cnt_ * sizeof(double), hipMemcpyHostToDevice, stream));
        dh = d_h_;
    }
    launch_newton_update(stream, d_prior_, d_nz_, S_->nnz_in, d_hmap_, dh, hmap_cnt_);
    refactorize(d_nz_, true);
}

// One Newton iterate as one pipelined call: Q_k = Q_prior - H_k on the device, numeric factorisation, solve
// (gaussian_approximation.jl:103-129: _update_hessian!, ensure_numeric!, then the solve for the new mean)
void Device::refactorize_update_solve(const double *h, bool h_on_device, const double *B, long long ldb, long long nrhs, double *X, long long ldx,
                                      bool b_on_device) {
    HC(hipSetDevice(device));
    if (!d_prior_) throw std::invalid_argument("gmrfx_set_prior has not been called");
    const double *dh = h;
    if (!h_on_device && hmap_cnt_ > 0) {
        HC(hipMemcpyAsync(d_h_, h, (size_t)hmap_cnt_ * sizeof(double), hipMemcpyHostToDevice, stream));
        dh = d_h_;
    }
    launch_newton_update(stream, d_prior_, d_nz_, S_->nnz_in, d_hmap_, dh, hmap_cnt_);
    refactorize_solve(d_nz_, true, B, ldb, nrhs, X, ldx, b_on_device);
}

double Device::syrk_ms() {
    if (syrk_times_pending_) {
        HC(hipSetDevice(device));
        ms_syrk = 0;
        for (long long k = 0; k < syrk_launches; k++) {
            float t = 0;
            HC(hipEventElapsedTime(&t, ev_syrk_[2 * k], ev_syrk_[2 * k + 1]));
            ms_syrk += t;
        }
        syrk_times_pending_ = false;
    }
    return ms_syrk;
}

long long Device::fail_col() {
    int v = INT_MAX;
    if (info_cached_) v = *h_info_;
    else {
        HC(hipStreamSynchronize(stream));       // (asynchronous phases / an external main stream: the factorisation may still run)
        HC(hipMemcpy(&v, d_info_, sizeof(int), hipMemcpyDeviceToHost));
    }
    return v == INT_MAX ? -1 : v;
}

void Device::ensure_rhs_capacity(long long nrhs) {
    const long long chunk = std::min<long long>(nrhs, 64);
    if (chunk > rhs_cap_) {
        // (old buffers stay in allocs_ until destruction; capacity only ever grows to 64)
        d_X_ = dalloc<double>((size_t)S_->n * 64);
        d_X2_ = dalloc<double>((size_t)S_->n * 64);
        d_W_ = dalloc<double>((size_t)std::max<long long>(sum_trail_, 1) * 64);
        rhs_cap_ = 64;
    }
    if (pass_width(nrhs) < nrhs && !d_Xb_ && !sharded()) {
        d_Xb_ = dalloc<double>((size_t)S_->n * 64);
        d_X2b_ = dalloc<double>((size_t)S_->n * 64);
        d_Wb_ = dalloc<double>((size_t)std::max<long long>(sum_trail_, 1) * 64);
    }
}

// The bottom subtrees. Up to wave_max_nr_ (16) right-hand sides: one wave per (task, 16 columns), sweep_wave.hip, biggest LDS
// class first; wider passes: the chunk form, four waves per (task, 16 columns), sweep_chunk.hip. Measured at cfg 2, round 5
// (tools/nrhs_sweep.py, ms per solve, wave form / chunk form): 1 RHS 2.88 / 3.22, 16: 3.00 / 3.36, 32: 3.71 / 3.47, 64: - / 3.90 at the
// time of the choice; with the narrow level kernels and the local vector as wide as the pass: 1 RHS 1.65, 16: 2.28.
// GMRFX_TASK_MODE = wg / wave forces one form.
void Device::sweep_tasks(int phase, int nr, int ldx) {
    if (nr > wave_max_nr_) {
        // pipelined call: the forward task kernel runs beside the top of the factorisation -- TWO resident workgroups per CU
        // instead of four (16 KB of unused dynamic LDS on top of its 40 KB), so that the panel chain's kernels find LDS
        // (measured at cfg 2, round 5: pad 0 / 8 / 16 / 42 KB -> step 12.59 / 12.61 / 12.38 / 12.91 ms)
        constexpr int pad_kb = 16;
        const size_t extra = (fused_fwd_ && phase == 1) ? (size_t)pad_kb * 1024 : 0;
        ensure_dtile();
        launch_sweep_chunks(stream, ds_, phase, d_swt_, nswt_, d_swc_fwd_, d_swc_bwd_, d_swc_listf_, d_swc_listb_, d_dtile_, d_L_, d_X_,
                            phase == 1 ? d_W_ : nullptr, nr, ldx, extra);
        return;
    }
    ensure_rdiag();
    if (nr <= 4) {      // the local vector of such a pass is 9 KB at most whatever the class: one launch, no tail of the big class before the small one starts
        launch_wave_tasks(stream, ds_, phase, d_swt_, d_wave_order_ + nswt_, nswt_, kWaveRows[kWaveClasses - 1], d_L_, d_rdiag_, d_rdiag_ + S_->n, d_X_,
                          d_W_, nr, ldx);
        return;
    }
    for (int k = kWaveClasses - 1; k >= 0; k--)
        launch_wave_tasks(stream, ds_, phase, d_swt_, d_wave_order_ + wave_first_[k], wave_count_[k], kWaveRows[k], d_L_, d_rdiag_,
                          d_rdiag_ + S_->n, d_X_, d_W_, nr, ldx);
}

// 1 / L_jj (+ the zero word masked operand elements are read from), once per factorisation, on the current `stream`:
// solve() calls this BEFORE its lanes fork, so that a second lane never reads it half-written
void Device::ensure_rdiag() {
    if (wave_max_nr_ <= 0 || nswt_ <= 0) return;
    if (!d_rdiag_) { d_rdiag_ = dalloc<double>((size_t)S_->n + 2); rdiag_for_ = 0; }
    if (rdiag_for_ == factor_serial_) return;
    launch_rdiag(stream, d_L_, ds_.diagoff, (int)S_->n, d_rdiag_);
    HC(hipMemsetAsync(d_rdiag_ + S_->n, 0, 2 * sizeof(double), stream));
    rdiag_for_ = factor_serial_;
}

// the chunks' inverse diagonal blocks in MFMA operand order, once per factorisation, on the current `stream` (solve() calls
// this BEFORE its lanes fork; the pipelined call behind the gate event, when the task fronts are final)
void Device::ensure_dtile() {
    if (nswc_ <= 0 || dtile_for_ == factor_serial_) return;
    launch_pack_diag(stream, d_swc_bwd_, nswc_, d_L_, d_dtile_);      // (the backward records hold every chunk once)
    dtile_for_ = factor_serial_;
}

void Device::forward(int nr, int ldx, int lo, int hi) {
    if (level_mark_ && lo == 0) { launch_level_mark(stream, 1, -1); level_event(1, 0); }
    // pipelined factor + solve (refactorize_solve): the bottom waits for the highest level a task / subtree reaches, every level
    // above for its own "factored" event; the dense inverses are built level by level instead of all at once
    if (fused_fwd_) {
        HC(hipStreamWaitEvent(stream, ev_flevel_[fused_gate_level_], 0));
        if (nr <= wave_max_nr_ && nswt_ > 0) rdiag_for_ = 0;      // 1 / L_jj of the task fronts: their diagonals are final now, the rest is never read
        dtile_for_ = 0;                                            // (the same for the chunks' inverse diagonal blocks)
    }
    if (lo == 0) sweep_tasks(1, nr, ldx);
    if (lo == 0)
        for (int k = 0, off = 0; k < 3; off += nsub_cls_[k], k++)
            launch_subtree(stream, ds_, 1, d_sub_first_ + off, d_sub_last_ + off, nsub_cls_[k], kClsRows[k], nullptr, d_L_, nullptr,
                           nullptr, d_X_, d_W_, nr, ldx);
    for (int lev = lo; lev < hi; lev++) {
        auto &L = swlevels_[lev];
        if (level_mark_) { launch_level_mark(stream, 1, lev); level_event(1, 1 + lev); }
        if (fused_fwd_) {
            if (lev > fused_gate_level_) HC(hipStreamWaitEvent(stream, ev_flevel_[lev], 0));
            invert_level(stream, lev);
        } else if (lev == std::max(lo, first_multiblock_level_)) wait_inverse();
        for (int k = 0, off = 0; k < 4; off += L.ncls[k], k++)
            launch_fwd_small(stream, ds_, d_sw_levellist_ + L.first + off, L.ncls[k], kClsRows[k], d_L_, d_X_, d_W_, nr, ldx);
        const int *list = d_sw_levellist_ + L.first + L.nsmall;
        int nf = L.count - L.nsmall;
        // fronts of at most 128 columns (the tail of the list: sorted by decreasing width): the WHOLE step -- own rows assembled,
        // y = L11^-1 b, W = children - L21 y -- as one workgroup and one launch (k_fwd_front, sweep_front.hip), on levels with enough
        // of them to fill the chip and for passes wider than the narrow kernels take; wider fronts keep the three launches below
        int cmin_front = 0;
        if (fwd_front_min_ > 0 && nr > narrow_pass_max()) {
            const size_t kq = (size_t)std::min(bwd_front_max_cols(), inv_cap_) / NB;       // (a front needs its WHOLE inverse for this)
            const int nwide = kq + 1 < L.active.size() ? L.active[kq] : 0;
            if (nf - nwide >= fwd_front_min_) {
                launch_fwd_front(stream, ds_, list + nwide, nf - nwide, d_L_, d_X_, d_X2_, d_W_, nr, ldx);
                nf = nwide;
                cmin_front = (int)kq * NB;       // the record-driven update below skips the fronts taken here (its records cover the level)
                if (nf == 0) continue;
            }
        }
        launch_fwd_assemble(stream, ds_, list, nf, L.max_cols, d_X_, d_W_, nr, ldx);   // own rows only
        // y = L11^-1 b as one triangular product per front (dense inverse, inverse.hip), then the
        // trailing update W -= L21 y with K = all columns of the front
        // y of the big fronts stays in X2 (no copy back): the update below and the backward sweep read it there
        // fronts wider than inv_cap_: block by block (y_j = X_jj b_j, then the own rows below -= L[.., block j] y_j)
        const int nbk = std::max(1, (L.max_cols + inv_cap_ - 1) / inv_cap_);
        for (int j = 0; j < nbk; j++) {
            // (nf: k_fwd_front may have taken the narrow tail of the list above -- block 0's count is "every big front" otherwise)
            const int na = nbk == 1 ? nf : std::min(nf, L.active[std::min<size_t>((size_t)j * inv_cap_ / NB, L.active.size() - 2)]);
            launch_xmul(stream, ds_, list, na, L.max_cols, 0, d_L_, d_X_, d_X2_, nr, ldx, j, inv_cap_);
            if (j + 1 < nbk) launch_fwd_own_update(stream, ds_, list, std::min(nf, L.active[(size_t)(j + 1) * inv_cap_ / NB]), L.max_cols, d_L_, d_X2_, d_X_, nr, ldx, j, inv_cap_);
        }
        // levels with many tiles: record-driven, per-XCD runs; the handful-of-fronts levels keep the 16-row latency variant
        // Passes of at most 16 right-hand sides: the fronts up to kFwdWaveCols columns wide go one WAVE per 32-row tile (no LDS, no
        // barrier: k_fwd_update_wave), chosen per FRONT so that a front's sums do not depend on the list it comes in
        constexpr int kFwdWaveCols = 1024;      // (measured at cfg 2, 1 RHS, forward ms, with four waves sharing the K range of a front wider than 128
                                                //  columns: 256: 0.933, 512: 0.841, 1024: 0.832, 2048: 0.836)
        const int cwave = syrk_xcd_ && nr <= narrow_pass_max() ? kFwdWaveCols : 0;
        const int cmin = std::max(cwave, cmin_front);          // fronts up to this width are not the split-K kernels' (wave kernel / k_fwd_front)
        const int nwider = (size_t)(kFwdWaveCols / NB + 1) < L.active.size() ? L.active[kFwdWaveCols / NB] : 0;      // fronts wider than that
        if (cwave > 0 && nwider < nf) launch_fwd_update_wave(stream, ds_, d_fwd_recs_ + L.fwd_off, L.fwd_split, L.fwd_per, d_L_, d_X2_, d_W_, nr, ldx, cwave,
                                                              L.max_cols > launch_wave_split_cols());
        if (L.max_cols > cmin) {
            if (syrk_xcd_ && (long long)((level_max_trail(L) + 31) / 32) * nf > 128)
                launch_fwd_update_recs(stream, ds_, d_fwd_recs_ + L.fwd_off, L.fwd_split, L.fwd_per, d_L_, d_X2_, d_W_, nr, ldx, cmin);
            else
                launch_fwd_update(stream, ds_, list, nf, level_max_trail(L), d_L_, d_X2_, d_W_, nr, ldx, cmin);
        }
    }
    if (level_mark_) level_event(1, 1 + hi);
}

// y_in_x2: the forward sweep left y of the big fronts in X2 (full solve). The backward sweep then turns it
// into t = y - L21' x in place there and writes x = L11^-T t straight into X -- no copies. A backward-only
// solve (F.UP \ z) gets z in X: classic path with one copy per level.
void Device::backward(int nr, int ldx, bool y_in_x2, int hi, int lo) {
    wait_inverse();   // (a no-op event wait once the forward sweep has passed it)
    for (int l = hi - 1; l >= lo; l--) {
        auto &L = swlevels_[l];
        if (level_mark_) { launch_level_mark(stream, 2, l); level_event(2, (int)levels_.size() - 1 - l); }
        const int *list = d_sw_levellist_ + L.first + L.nsmall;
        int nf = L.count - L.nsmall;
        for (int k = 0, off = 0; k < 4; off += L.ncls[k], k++)
            launch_bwd_small(stream, ds_, d_sw_levellist_ + L.first + off, L.ncls[k], kClsRows[k], d_L_, d_X_, nr, ldx);
        // fronts of at most 128 columns (the tail of the list: sorted by decreasing width): the whole step as one workgroup and
        // one launch (sweep_front.hip); in place when y sits in X (own rows are read and written by their front alone)
        // Only on levels with enough such fronts to fill the chip: a workgroup walks its front's trailing rows batch after batch,
        // and a level of a hundred fronts with 700 trailing rows each is faster as many small workgroups (the two launches).
        // Passes of at most 16 right-hand sides: t = y - L21' x one WAVE per 16 own columns for the fronts with at most kBwdWaveRows
        // trailing rows (k_bwd_wave), the split-K kernels for the others, x = L11^-T t as everywhere
        constexpr int kBwdWaveRows = 4096;      // (every front of a 2-D problem; measured at cfg 2, 1 RHS, backward ms, with four waves sharing the K range of a front of more than 256
                                                //  rows: 768: 0.848, 1100: 0.824, 1600: 0.814, every front: 0.792; one wave per tile only: 768 was the optimum, 1.177)
        const bool narrow = nr <= narrow_pass_max_bwd();
        if (bwd_front_min_ > 0 && !narrow) {
            // (a front needs its WHOLE inverse for this: at most min(128, inv_cap_) columns)
            const size_t kq = (size_t)std::min(bwd_front_max_cols(), inv_cap_) / NB;
            const int nwide = kq + 1 < L.active.size() ? L.active[kq] : 0;   // fronts with more than kq NB columns (the last entry of `active` is the stash of level_max_trail)
            if (nf - nwide >= bwd_front_min_) {
                launch_bwd_front(stream, ds_, list + nwide, nf - nwide, d_L_, d_X_, y_in_x2 ? d_X2_ : d_X_, d_X_, nr, ldx);
                nf = nwide;
                if (nf == 0) continue;
            }
        }
        const int nbk = std::max(1, (L.max_cols + inv_cap_ - 1) / inv_cap_);
        double *const Xt = y_in_x2 ? d_X2_ : d_X_;
        const int mmin = narrow ? kBwdWaveRows : 0;
        if (narrow && level_max_trail(L) > 0 && L.min_trail <= kBwdWaveRows) launch_bwd_wave(stream, ds_, list, nf, L.max_cols, d_L_, d_X_, Xt, nr, ldx, kBwdWaveRows,
                                                                                                  level_max_trail(L) > launch_wave_split_rows());
        if (y_in_x2) {
            if (level_max_trail(L) > mmin) launch_bwd_gemm(stream, ds_, list, nf, L.max_cols, d_L_, d_X_, d_X2_, nr, ldx, -1, 1 << 30, mmin);
            // fronts wider than inv_cap_: from the last block up, t_j -= L[own rows below, block j]' x, x_j = X_jj' t_j
            for (int j = nbk - 1; j >= 0; j--) {
                // (nf: k_bwd_front may have taken the narrow tail of the list above -- block 0's count is "every big front" otherwise)
                const int na = nbk == 1 ? nf : std::min(nf, L.active[std::min<size_t>((size_t)j * inv_cap_ / NB, L.active.size() - 2)]);
                if (j + 1 < nbk) launch_bwd_gemm(stream, ds_, list, std::min(nf, L.active[(size_t)(j + 1) * inv_cap_ / NB]), L.max_cols, d_L_, d_X_, d_X2_, nr, ldx, j, inv_cap_);
                launch_xmul(stream, ds_, list, na, L.max_cols, 1, d_L_, d_X2_, d_X_, nr, ldx, j, inv_cap_);
            }
        } else {
            if (level_max_trail(L) > mmin) launch_bwd_gemm(stream, ds_, list, nf, L.max_cols, d_L_, d_X_, d_X_, nr, ldx, -1, 1 << 30, mmin);
            for (int j = nbk - 1; j >= 0; j--) {
                const int na = nbk == 1 ? nf : std::min(nf, L.active[std::min<size_t>((size_t)j * inv_cap_ / NB, L.active.size() - 2)]);
                if (j + 1 < nbk) launch_bwd_gemm(stream, ds_, list, std::min(nf, L.active[(size_t)(j + 1) * inv_cap_ / NB]), L.max_cols, d_L_, d_X_, d_X_, nr, ldx, j, inv_cap_);
                launch_xmul(stream, ds_, list, na, L.max_cols, 1, d_L_, d_X_, d_X2_, nr, ldx, j, inv_cap_);
                // x_j has to be back in X before the block above reads it
                launch_copy_own(stream, ds_, list, na, L.max_cols, d_X2_, d_X_, nr, ldx, j, inv_cap_);
            }
        }
    }
    if (level_mark_ && lo == 0) { launch_level_mark(stream, 2, -1); level_event(2, (int)levels_.size()); }
    if (lo == 0)
        for (int k = 0, off = 0; k < 3; off += nsub_cls_[k], k++)
            launch_subtree(stream, ds_, 2, d_sub_first_ + off, d_sub_last_ + off, nsub_cls_[k], kClsRows[k], nullptr, d_L_, nullptr,
                           nullptr, d_X_, nullptr, nr, ldx);
    if (lo == 0) sweep_tasks(2, nr, ldx);
    if (level_mark_ && lo == 0) level_event(2, (int)levels_.size() + 1);
}

// GMRFX_LEVEL_MARK=1: HIP events at the level boundaries of the most recent factorisation / forward / backward sweep
// (slot numbering in level_times()); a profiling aid behind gmrfx_level_times, never on in production runs
void Device::level_event(int phase, int slot) {
    auto &v = ev_level_[phase];
    while ((int)v.size() <= slot) { hipEvent_t e; HC(hipEventCreate(&e)); v.push_back(e); }
    HC(hipEventRecord(v[slot], stream));
    level_slots_[phase] = std::max(level_slots_[phase], slot + 1);
}
// out[0] = the sweep tasks (0 for the factorisation), out[1 + l] = tree level l, milliseconds; returns the count
int Device::level_times(int phase, double *out, int cap) {
    HC(hipSetDevice(device));
    const int nl = (int)levels_.size();
    if (!level_mark_ || phase < 0 || phase > 2 || level_slots_[phase] < (phase == 0 ? nl + 1 : nl + 2)) return 0;
    HC(hipDeviceSynchronize());
    auto &v = ev_level_[phase];
    auto dt = [&](int a, int b) { float ms = 0; HC(hipEventElapsedTime(&ms, v[a], v[b])); return (double)ms; };
    for (int k = 0; k <= nl && k < cap; k++) {
        if (phase == 0) out[k] = k == 0 ? 0.0 : dt(k - 1, k);                       // slots: start of level l = l, end = nl
        else if (phase == 1) out[k] = dt(k, k + 1);                                  // slot 0 = start, 1 + l = start of level l, 1 + nl = end
        else out[k] = k == 0 ? dt(nl, nl + 1) : dt(nl - k, nl - k + 1);              // processing order: level nl-1 .. 0, tasks
    }
    return std::min(nl + 1, cap);
}

void Device::solve_phase(const double *d_B, long long ldb, long long nrhs, double *d_Xout, long long ldx_out, int phase) {
    HC(hipSetDevice(device));
    if (!sharded()) throw std::invalid_argument("gmrfx_solve_phase needs a sharded handle (shard_world > 1, or shard_min_top > 0)");
    if (nrhs <= 0 || nrhs > 64) throw std::invalid_argument("sharded solves take 1..64 right-hand sides per call");
    const int nr = (int)nrhs, ldx = nr, nl = (int)levels_.size();
    const int split = std::min<int>(S_->shard_level, nl);
    const long long n = S_->n;
    HC(hipEventRecord(ev_[0], stream));
    if (phase == 0) {                                     // transpose in + forward over the own subtrees
        ensure_rhs_capacity(nrhs);
        start_inverse_async();
        ensure_rdiag();
        ensure_dtile();
        launch_permute(stream, d_iperm_, (int)n, const_cast<double *>(d_B), ldb, d_X_, nr, ldx, 0);
        forward(nr, ldx, 0, split);
    } else if (phase >= 100 && phase < 100 + (nl - split)) {       // forward, top level split + (phase - 100)
        const int lev = split + phase - 100;
        forward(nr, ldx, lev, lev + 1);
    } else if (phase >= 200 && phase < 200 + (nl - split)) {       // backward, top level split + (phase - 200)
        const int lev = split + phase - 200;
        backward(nr, ldx, true, lev + 1, lev);
    } else if (phase == 2) {                              // backward over the own subtrees
        backward(nr, ldx, true, split, 0);
    } else if (phase == 10) {                             // F.UP \ z: z is taken in elimination order as is, no forward sweep
        ensure_rhs_capacity(nrhs);
        start_inverse_async();
        ensure_rdiag();
        ensure_dtile();
        launch_permute(stream, nullptr, (int)n, const_cast<double *>(d_B), ldb, d_X_, nr, ldx, 0);
    } else if (phase >= 300 && phase < 300 + (nl - split)) {       // backward-only solve, top level split + (phase - 300)
        const int lev = split + phase - 300;
        backward(nr, ldx, false, lev + 1, lev);
    } else if (phase == 12) {                             // backward-only solve over the own subtrees
        backward(nr, ldx, false, split, 0);
    } else if (phase == 3) {                              // transpose out (rank 0, after the gather)
        launch_permute(stream, d_iperm_, (int)n, d_Xout, ldx_out, d_X_, nr, ldx, 1);
    } else throw std::invalid_argument("solve phase must be 0, 2, 3, 10, 12, 100 + k, 200 + k or 300 + k (k = top level)");
    HC(hipEventRecord(ev_[1], stream));
    if (!async_phases_) {
        HC(hipStreamSynchronize(stream));
        HC(hipGetLastError());
        float ms = 0;
        HC(hipEventElapsedTime(&ms, ev_[0], ev_[1]));
        ms_solve = (phase == 0 || phase == 10) ? ms : ms_solve + ms;
    }
    last_nrhs = nrhs;
}

void Device::solve(const double *B, long long ldb, long long nrhs, double *X, long long ldx_out, bool on_device, int mode) {
    HC(hipSetDevice(device));
    if (sharded()) throw std::invalid_argument("sharded handle: use gmrfx_solve_phase (phases with exchanges in between, gmrfx/shard.py)");
    if (nrhs <= 0) return;
    const long long n = S_->n;
    ensure_rhs_capacity(nrhs);
    start_inverse_async();
    ensure_rdiag();
    ensure_dtile();
    const double *dB = B;
    double *dXo = X;
    long long ldin = ldb, ldout = ldx_out;
    if (!on_device) {
        const long long need = n * nrhs;
        if (need > io_cap_) { const long long cap = std::max(need, 2 * io_cap_); d_io_ = dregrow(d_io_, (size_t)cap); io_cap_ = cap; }
        if (ldb == n) HC(hipMemcpyAsync(d_io_, B, (size_t)need * sizeof(double), hipMemcpyHostToDevice, stream));
        else HC(hipMemcpy2DAsync(d_io_, n * sizeof(double), B, ldb * sizeof(double), n * sizeof(double), nrhs, hipMemcpyHostToDevice, stream));
        dB = d_io_; dXo = d_io_; ldin = n; ldout = n;
    }
    double t_perm = 0, t_fwd = 0, t_bwd = 0;
    // 64-column passes, alternating between the two lanes when there is more than one pass. forward() / backward()
    // enqueue on the members `stream`, d_X_, d_X2_, d_W_: a lane is selected by swapping them in for the duration
    // of the (asynchronous) enqueue.
    // passes of pass_width() = 64 columns, alternating between the two lanes when there is more than one pass
    const int PW = pass_width(nrhs);
    const bool two = nrhs > PW && d_Xb_ != nullptr;
    struct LaneState { hipStream_t st; double *X, *X2, *W; };
    LaneState lanes[2] = {{stream, d_X_, d_X2_, d_W_}, {stream3, d_Xb_, d_X2b_, d_Wb_}};
    if (two) {
        // lane 1 starts after everything already enqueued on the main stream (factorisation, upload of B)
        HC(hipEventRecord(ev_ready_, stream));
        HC(hipStreamWaitEvent(stream3, ev_ready_, 0));
    }
    bool busy[2] = {false, false};
    auto collect = [&](int ln) {
        if (!busy[ln]) return;
        HC(hipEventSynchronize(ev_lane_[ln][4]));
        float a, b, c, d;
        HC(hipEventElapsedTime(&a, ev_lane_[ln][0], ev_lane_[ln][1]));
        HC(hipEventElapsedTime(&b, ev_lane_[ln][1], ev_lane_[ln][2]));
        HC(hipEventElapsedTime(&c, ev_lane_[ln][2], ev_lane_[ln][3]));
        HC(hipEventElapsedTime(&d, ev_lane_[ln][3], ev_lane_[ln][4]));
        t_perm += a + d; t_fwd += b; t_bwd += c;
        busy[ln] = false;
    };
    const hipStream_t main_stream = stream;
    double *const X0 = d_X_, *const X20 = d_X2_, *const W0 = d_W_;
    struct Restore {     // the members come back even if a HIP call throws while a lane is swapped in
        Device &D; hipStream_t st; double *X, *X2, *W;
        ~Restore() { D.stream = st; D.d_X_ = X; D.d_X2_ = X2; D.d_W_ = W; }
    } restore{*this, main_stream, X0, X20, W0};
    HC(hipEventRecord(ev_[0], stream));
    int pass = 0;
    for (long long j0 = 0; j0 < nrhs; j0 += PW, pass++) {
        const int nr = (int)std::min<long long>(PW, nrhs - j0);
        const int ldx = nr;
        const int ln = two ? (pass & 1) : 0;
        collect(ln);                         // the lane's previous pass has finished: its buffers are free
        stream = lanes[ln].st; d_X_ = lanes[ln].X; d_X2_ = lanes[ln].X2; d_W_ = lanes[ln].W;
        hipEvent_t *ev = ev_lane_[ln];
        HC(hipEventRecord(ev[0], stream));
        // full solve: X = P b ; backward-only (F.UP \ z): z is taken in elimination order as is
        launch_permute(stream, mode == 0 ? d_iperm_ : nullptr, (int)n, const_cast<double *>(dB) + j0 * ldin, ldin, d_X_, nr, ldx, 0);
        HC(hipEventRecord(ev[1], stream));

        if (mode == 0) forward(nr, ldx, 0, (int)levels_.size());
        HC(hipEventRecord(ev[2], stream));
        backward(nr, ldx, mode == 0, (int)levels_.size(), 0);
        HC(hipEventRecord(ev[3], stream));
        launch_permute(stream, d_iperm_, (int)n, dXo + j0 * ldout, ldout, d_X_, nr, ldx, 1);
        HC(hipEventRecord(ev[4], stream));
        busy[ln] = true;
        stream = main_stream; d_X_ = X0; d_X2_ = X20; d_W_ = W0;
    }
    if (two) {
        HC(hipEventRecord(ev_done1_, stream3));
        HC(hipStreamWaitEvent(stream, ev_done1_, 0));
    }
    HC(hipEventRecord(ev_[1], stream));
    collect(0);
    collect(1);
    HC(hipStreamSynchronize(stream));
    float wall = 0;
    HC(hipEventElapsedTime(&wall, ev_[0], ev_[1]));
    HC(hipGetLastError());
    if (!on_device) {
        const long long need = n * nrhs;
        if (ldx_out == n) HC(hipMemcpyAsync(X, d_io_, (size_t)need * sizeof(double), hipMemcpyDeviceToHost, stream));
        else HC(hipMemcpy2DAsync(X, ldx_out * sizeof(double), d_io_, n * sizeof(double), n * sizeof(double), nrhs, hipMemcpyDeviceToHost, stream));
        HC(hipStreamSynchronize(stream));
    }
    // per-phase times are sums over the passes; with two lanes the passes overlap, so the totals are the
    // elapsed time on the device from the first launch to the last completion
    ms_perm = t_perm; ms_fwd = t_fwd; ms_bwd = t_bwd;
    if (mode == 0) ms_solve = two ? wall : t_perm + t_fwd + t_bwd; else ms_bsolve = two ? wall : t_perm + t_bwd;
    last_nrhs = nrhs;
}

// log det Q = 2 sum log L_jj: two small kernels over the factor's diagonal + one scalar copied to pinned host memory. The result
// is kept per factorisation; the pipelined factor + solve call enqueues it on the side stream beside the backward sweep.
void Device::enqueue_logdet(hipStream_t st, bool timed) {
    if (!h_logdet_) {
        HC(hipHostMalloc((void **)&h_logdet_, sizeof(double), hipHostMallocDefault));
        HC(hipEventCreateWithFlags(&ev_logdet_, hipEventDisableTiming));
    }
    const int nparts = (int)std::min<long long>(1024, std::max<long long>(1, (S_->n + 255) / 256));
    if (timed) HC(hipEventRecord(ev_[0], st));
    launch_logdet(st, d_L_, ds_.diagoff, d_owncol_, (int)S_->n, d_part_, nparts, d_part_ + 1024);
    if (timed) HC(hipEventRecord(ev_[1], st));
    HC(hipMemcpyAsync(h_logdet_, d_part_ + 1024, sizeof(double), hipMemcpyDeviceToHost, st));
    HC(hipEventRecord(ev_logdet_, st));
    logdet_for_ = factor_serial_;
}
double Device::logdet() {
    HC(hipSetDevice(device));
    if (logdet_for_ != factor_serial_ || !h_logdet_) {
        enqueue_logdet(stream, true);
        HC(hipEventSynchronize(ev_logdet_));
        float ms; HC(hipEventElapsedTime(&ms, ev_[0], ev_[1])); ms_logdet = ms;
    } else HC(hipEventSynchronize(ev_logdet_));
    return *h_logdet_;
}

// pattern of Q and the partial-sum buffers of the quadratic-form kernels (grown geometrically)
void Device::prepare_quadform(long long nvec) {
    const Symbolic &S = *S_;
    if (!d_in_colptr_) {
        d_in_colptr_ = dalloc<long long>(S.in_colptr.size());
        d_in_row_ = dalloc<int>(std::max<size_t>(S.in_row.size(), 1));
        HC(hipMemcpyAsync(d_in_colptr_, S.in_colptr.data(), S.in_colptr.size() * sizeof(long long), hipMemcpyHostToDevice, stream));
        HC(hipMemcpyAsync(d_in_row_, S.in_row.data(), S.in_row.size() * sizeof(int), hipMemcpyHostToDevice, stream));
        HC(hipStreamSynchronize(stream));      // (the host vectors may move; first call only)
    }
    const int nblk = quadform_blocks((int)S.n);
    if (nvec > qf_cap_) {
        const long long cap = std::max<long long>(nvec, 2 * qf_cap_);     // geometric growth, the old buffers are freed
        qf_cap_ = 0;        // (a failed second allocation must not leave the pair with different sizes behind one capacity)
        d_qf_part_ = dregrow(d_qf_part_, (size_t)cap * nblk);
        d_qf_out_ = dregrow(d_qf_out_, (size_t)cap);
        qf_cap_ = cap;
    }
}

// One evaluation of the hyper-parameter loop as ONE call (gmrfx_refactorize_logpdf_dev): new values -> numeric factorisation,
// r' Q r for nvec vectors and log det Q -- logpdf(::WorkspaceGMRF, z) = -r'Qr / 2 + logdet(Q) / 2 - n log(2 pi) / 2,
// src/workspace/workspace_gmrf.jl:288-292, with ensure_numeric! inside (gmrf_workspace.jl:170-178). The quadratic forms only
// need Q's values: they run on the side stream beside the factorisation; the log-determinant follows the factorisation on the
// main stream; ONE synchronisation, the scalars arrive in pinned memory. Same kernels: same bits as the three calls.
void Device::refactorize_logpdf(const double *d_nz, const double *d_X, long long ldx, long long nvec, const double *d_mu, double *quad_out,
                                double *logdet_out) {
    HC(hipSetDevice(device));
    const Symbolic &S = *S_;
    if (sharded()) throw std::invalid_argument("sharded handle: use the phase entry points (gmrfx/shard.py)");
    if (nvec < 0 || nvec > 65535) throw std::invalid_argument("refactorize_logpdf: 0..65535 vectors per call");
    if (nvec > 0 && ldx < S.n) throw std::invalid_argument("refactorize_logpdf: ldx < n");
    if (stream != own_stream_) throw std::invalid_argument("refactorize_logpdf: not on a caller's stream");
    if (nvec > 0) {
        prepare_quadform(nvec);
        if (nvec > h_qf_cap_) {
            if (h_qf_) HC(hipHostFree(h_qf_));
            h_qf_ = nullptr; h_qf_cap_ = 0;
            HC(hipHostMalloc((void **)&h_qf_, (size_t)std::max<long long>(nvec, 16) * sizeof(double), hipHostMallocDefault));
            h_qf_cap_ = std::max<long long>(nvec, 16);
        }
        HC(hipEventRecord(ev_ready_, stream));
        HC(hipStreamWaitEvent(stream2, ev_ready_, 0));
        launch_quadform(stream2, (int)S.n, d_in_colptr_, d_in_row_, d_nz, S.in_use, d_X, ldx, (int)nvec, d_mu, d_qf_part_, d_qf_out_);
        HC(hipMemcpyAsync(h_qf_, d_qf_out_, (size_t)nvec * sizeof(double), hipMemcpyDeviceToHost, stream2));
        if (!ev_qf_) HC(hipEventCreateWithFlags(&ev_qf_, hipEventDisableTiming));
        HC(hipEventRecord(ev_qf_, stream2));
    }
    nz_held_ = false;
    nz_src_ = d_nz;
    factor_serial_++;
    HC(hipEventRecord(ev_[0], stream));
    factor_levels(0, (int)levels_.size());
    HC(hipEventRecord(ev_[1], stream));
    HC(hipEventRecord(ev_fact_, stream));
    fact_event_valid_ = true;
    inverse_pending = true;
    HC(hipMemcpyAsync(h_info_, d_info_, sizeof(int), hipMemcpyDeviceToHost, stream));
    factorized = true;
    selinv_valid = false;
    float tf = 0;
    enqueue_logdet(stream, false);
    if (nvec > 0) HC(hipStreamWaitEvent(stream, ev_qf_, 0));
    HC(hipStreamSynchronize(stream));
    info_cached_ = true;
    HC(hipGetLastError());
    HC(hipEventElapsedTime(&tf, ev_[0], ev_[1]));
    ms_factor = tf;
    syrk_times_pending_ = true;
    for (long long k = 0; k < nvec; k++) quad_out[k] = h_qf_[k];
    if (logdet_out) *logdet_out = *h_logdet_;
}

void Device::quadform(const double *d_nz, const double *d_X, long long ldx, long long nvec, const double *d_mu, double *out_host) {
    HC(hipSetDevice(device));
    const Symbolic &S = *S_;
    if (nvec <= 0) return;
    if (nvec > 65535) throw std::invalid_argument("quadform: at most 65535 vectors per call");
    if (ldx < S.n) throw std::invalid_argument("quadform: ldx < n");
    if (!d_nz) {
        if (!nz_held_) throw std::invalid_argument("quadform: the handle does not hold Q's values (last refactorisation read a caller device buffer): pass them");
        d_nz = d_nz_;
    }
    prepare_quadform(nvec);
    HC(hipEventRecord(ev_[0], stream));
    launch_quadform(stream, (int)S.n, d_in_colptr_, d_in_row_, d_nz, S.in_use, d_X, ldx, (int)nvec, d_mu, d_qf_part_, d_qf_out_);
    HC(hipEventRecord(ev_[1], stream));
    HC(hipMemcpyAsync(out_host, d_qf_out_, (size_t)nvec * sizeof(double), hipMemcpyDeviceToHost, stream));
    HC(hipStreamSynchronize(stream));
    float ms; HC(hipEventElapsedTime(&ms, ev_[0], ev_[1])); ms_quadform = ms;
}

void Device::selinv_begin() {
    const Symbolic &S = *S_;
    start_inverse_async();
    wait_inverse();
    if (!inverse_full_) {      // the sweeps only need inv_cap_-column inverses; the Takahashi step needs all of L11^-1
        invert_diag_blocks(stream, inv_cap_, 1 << 30);
        inverse_full_ = true;
    }
    if (!d_Z_) d_Z_ = dalloc<double>((size_t)l_size_);
    // workspaces: small fronts: Yh = r x 64 per front; big fronts: Yt and Z21t = (r-c) x c each
    if (!d_yoff_) {
        std::vector<long long> yoff(S.nsuper, 0);
        long long mx = 0;
        for (i32 l = 0; l < S.nlevels; l++) {
            long long off = 0;
            const i64 lf = S.sel_levelptr[l], cnt = S.sel_levelptr[l + 1] - lf;
            for (i64 k = 0; k < cnt; k++) {
                i32 s = S.sel_levellist[lf + k];
                yoff[s] = off;
                off += k < S.sel_level_nsmall[l] ? (long long)S.nrows(s) * NB : 2LL * (S.nrows(s) - S.ncols(s)) * S.ncols(s);
            }
            mx = std::max(mx, off);
        }
        if (mx > tmp_cap_) { d_tmp_ = dregrow(d_tmp_, (size_t)mx); tmp_cap_ = mx; }
        d_yoff_ = dalloc<long long>(std::max<size_t>(yoff.size(), 1));
        HC(hipMemcpyAsync(d_yoff_, yoff.data(), yoff.size() * sizeof(long long), hipMemcpyHostToDevice, stream));
        HC(hipStreamSynchronize(stream));
    }
    HC(hipMemsetAsync(d_Z_, 0, (size_t)l_size_ * sizeof(double) + kPairSlackBytes, stream));
}

void Device::selinv_levels(int hi, int lo) {
    const Symbolic &S = *S_;
    DevSym dsz = ds_;            // the selected inversion has its own slot layout in the contribution-block arena
    dsz.cbptr = d_zbptr_;
    for (int l = hi - 1; l >= lo; l--) {
        auto &L = levels_[l];
        // all fronts of the level (subtree members included): small first, then big
        const int sfirst = (int)S.sel_levelptr[l], scount = (int)(S.sel_levelptr[l + 1] - S.sel_levelptr[l]);
        const int snsmall = S.sel_level_nsmall[l];
        const int *list = d_sel_levellist_ + sfirst + snsmall;
        const int nf = scount - snsmall;
        // big fronts: whole-front step through the dense inverse (selinv.hip, k_sel_dense).
        // Yt lives at d_tmp_ + yoff[s], Z21t right behind it (offset (r-c)*c): pass both bases.
        // (geometry over the level's fronts of the SELECTED-INVERSION list: the factor's level list of a sharded handle
        //  leaves out the distributed root, which the owner still inverts)
        (void)L;
        if (level_mark_) launch_level_mark(stream, 4, l);      // (profiling aid: tools/cfg3_profile.py cuts the trace into levels)
        launch_sel_gather(stream, d_selrec_, dsz, list, nf, sel_max_trail_[l], d_Z_, d_cb_);
        for (int phase = 0; phase < 3; phase++)
            launch_sel_dense(stream, dsz, list, nf, phase, sel_max_cols_[l], sel_max_trail_[l], d_L_, d_Z_, d_cb_, d_tmp_,
                             d_tmp_, d_yoff_);
        // small fronts of the level (<= 128 rows, <= 64 columns: one block step)
        if (snsmall > 0) {
            const int *sl = d_sel_levellist_ + sfirst;
            launch_sel_gather(stream, d_selrec_, dsz, sl, snsmall, 128, d_Z_, d_cb_);
            launch_trsm(stream, dsz, d_sel_frec_ + sfirst, snsmall, 0, 1, 128, d_L_, d_tmp_, d_yoff_, FrontArg{0, 0, 0, 0, 0, 0, 0});
            launch_sel_symm(stream, dsz, sl, snsmall, 0, 128, d_Z_, d_cb_, d_tmp_, d_yoff_);
            launch_sel_diag(stream, dsz, sl, snsmall, 0, d_L_, d_Z_, d_tmp_, d_yoff_);
        }
    }
}

void Device::selinv_compute() {
    HC(hipSetDevice(device));
    if (selinv_valid) return;
    if (sharded()) throw std::invalid_argument("sharded handle: the selected inversion runs in phases with exchanges in between (gmrfx_selinv_phase, gmrfx/shard.py)");
    selinv_begin();
    HC(hipEventRecord(ev_[0], stream));
    selinv_levels((int)levels_.size(), 0);
    HC(hipEventRecord(ev_[1], stream));
    HC(hipStreamSynchronize(stream));
    HC(hipGetLastError());
    float ms; HC(hipEventElapsedTime(&ms, ev_[0], ev_[1])); ms_selinv = ms;
    selinv_valid = true;
}

void Device::selinv_phase(int what, int hi, int lo) {
    HC(hipSetDevice(device));
    if (!sharded()) throw std::invalid_argument("gmrfx_selinv_phase needs a sharded handle (shard_world > 1, or shard_min_top > 0)");
    const int nl = (int)levels_.size();
    if (what != 0 && !selinv_begun_)       // (phases 1-3 launch kernels on the workspaces phase 0 allocates)
        throw std::invalid_argument("selinv phase 1, 2 or 3 before phase 0 (begin) since the last refactorisation");
    if (what == 0) {
        if (!factorized) throw std::invalid_argument("selinv phase 0 before the factorisation has finished");
        selinv_valid = false;
        selinv_begin();
        selinv_begun_ = true;
        ms_selinv = 0;
    } else if (what == 1) {
        if (hi < 0 || hi >= nl) throw std::invalid_argument("selinv phase: level out of range");
        DevSym dsz = ds_;
        dsz.cbptr = d_zbptr_;
        const int a = fc_levelptr_[hi], b = fc_levelptr_[hi + 1];
        launch_sel_gather(stream, d_selrec_, dsz, d_fchild_ + a, b - a, fc_maxtrail_[hi], d_Z_, d_cb_);
    } else if (what == 2) {
        if (lo < 0 || hi > nl || lo > hi) throw std::invalid_argument("selinv phase: level range out of bounds");
        HC(hipEventRecord(ev_[0], stream));
        selinv_levels(hi, lo);
        HC(hipEventRecord(ev_[1], stream));
        if (!async_phases_) {
            HC(hipStreamSynchronize(stream));
            float ms; HC(hipEventElapsedTime(&ms, ev_[0], ev_[1])); ms_selinv += ms;
        }
    } else if (what == 3) {
        selinv_valid = true;
    } else throw std::invalid_argument("selinv phase must be 0 (begin), 1 (gather for other ranks), 2 (own levels) or 3 (end)");
    if (!async_phases_) HC(hipStreamSynchronize(stream));
    HC(hipGetLastError());
}

void Device::selinv_diag(double *out_host) {
    HC(hipSetDevice(device));
    const long long n = S_->n;
    if (n > io_cap_) { d_io_ = dregrow(d_io_, (size_t)n); io_cap_ = n; }
    launch_gather_diag(stream, d_Z_, ds_.diagoff, ds_.perm, (int)n, d_io_);
    HC(hipMemcpyAsync(out_host, d_io_, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, stream));
    HC(hipStreamSynchronize(stream));
}

void Device::gather_z(const long long *offsets_host, long long cnt, double *out_host) {
    HC(hipSetDevice(device));
    if (cnt <= 0) return;
    long long *d_off = nullptr;
    double *d_out = nullptr;
    HC(hipMalloc((void **)&d_off, (size_t)cnt * sizeof(long long)));
    HC(hipMalloc((void **)&d_out, (size_t)cnt * sizeof(double)));
    HC(hipMemcpyAsync(d_off, offsets_host, (size_t)cnt * sizeof(long long), hipMemcpyHostToDevice, stream));
    launch_gather(stream, d_Z_, d_off, cnt, d_out);
    HC(hipMemcpyAsync(out_host, d_out, (size_t)cnt * sizeof(double), hipMemcpyDeviceToHost, stream));
    HC(hipStreamSynchronize(stream));
    (void)hipFree(d_off);
    (void)hipFree(d_out);
}

void Device::weighted_z_sums(const long long *segptr_host, long long nseg, const long long *off_host, const double *w_host,
                             double *out_host) {
    HC(hipSetDevice(device));
    if (nseg <= 0) return;
    if (nseg > 0x7fffffffLL) throw std::invalid_argument("too many segments");
    const long long cnt = segptr_host[nseg];
    struct Buf { void *p = nullptr; ~Buf() { if (p) (void)hipFree(p); } } bseg, boff, bw, bout;
    HC(hipMalloc(&bseg.p, (size_t)(nseg + 1) * sizeof(long long)));
    HC(hipMalloc(&boff.p, (size_t)std::max<long long>(cnt, 1) * sizeof(long long)));
    HC(hipMalloc(&bw.p, (size_t)std::max<long long>(cnt, 1) * sizeof(double)));
    HC(hipMalloc(&bout.p, (size_t)nseg * sizeof(double)));
    HC(hipMemcpyAsync(bseg.p, segptr_host, (size_t)(nseg + 1) * sizeof(long long), hipMemcpyHostToDevice, stream));
    HC(hipMemcpyAsync(boff.p, off_host, (size_t)cnt * sizeof(long long), hipMemcpyHostToDevice, stream));
    HC(hipMemcpyAsync(bw.p, w_host, (size_t)cnt * sizeof(double), hipMemcpyHostToDevice, stream));
    launch_seg_wsum(stream, d_Z_, (const long long *)bseg.p, nseg, (const long long *)boff.p, (const double *)bw.p, (double *)bout.p);
    HC(hipMemcpyAsync(out_host, bout.p, (size_t)nseg * sizeof(double), hipMemcpyDeviceToHost, stream));
    HC(hipStreamSynchronize(stream));
}

long long Device::rowdiag_plan_create(const long long *segptr_host, long long nseg, const long long *off_host, const int *p_host,
                                      const int *q_host, long long nvals) {
    HC(hipSetDevice(device));
    if (nseg > 0x7fffffffLL) throw std::invalid_argument("too many rows");
    RowDiagPlan P;
    P.nseg = nseg; P.cnt = nseg > 0 ? segptr_host[nseg] : 0; P.nvals = nvals;
    auto grab = [&](void **p, size_t bytes) { HC(hipMalloc(p, std::max<size_t>(bytes, 8))); };
    grab((void **)&P.seg, (size_t)(nseg + 1) * sizeof(long long));
    grab((void **)&P.off, (size_t)P.cnt * sizeof(long long));
    grab((void **)&P.p, (size_t)P.cnt * sizeof(int));
    grab((void **)&P.q, (size_t)P.cnt * sizeof(int));
    grab((void **)&P.vals, (size_t)nvals * sizeof(double));
    grab((void **)&P.out, (size_t)nseg * sizeof(double));
    HC(hipMemcpyAsync(P.seg, segptr_host, (size_t)(nseg + 1) * sizeof(long long), hipMemcpyHostToDevice, stream));
    HC(hipMemcpyAsync(P.off, off_host, (size_t)P.cnt * sizeof(long long), hipMemcpyHostToDevice, stream));
    HC(hipMemcpyAsync(P.p, p_host, (size_t)P.cnt * sizeof(int), hipMemcpyHostToDevice, stream));
    HC(hipMemcpyAsync(P.q, q_host, (size_t)P.cnt * sizeof(int), hipMemcpyHostToDevice, stream));
    HC(hipStreamSynchronize(stream));
    for (size_t k = 0; k < rd_plans_.size(); k++)
        if (!rd_plans_[k].seg) { rd_plans_[k] = P; return (long long)k; }
    rd_plans_.push_back(P);
    return (long long)rd_plans_.size() - 1;
}

void Device::rowdiag_plan_apply(long long id, const double *values_host, double *out_host) {
    HC(hipSetDevice(device));
    if (id < 0 || id >= (long long)rd_plans_.size() || !rd_plans_[id].seg) throw std::invalid_argument("unknown row-diag plan");
    const RowDiagPlan &P = rd_plans_[id];
    if (P.nseg <= 0) return;
    HC(hipMemcpyAsync(P.vals, values_host, (size_t)P.nvals * sizeof(double), hipMemcpyHostToDevice, stream));
    launch_seg_wsum_pairs(stream, d_Z_, P.seg, P.nseg, P.off, P.p, P.q, P.vals, P.out);
    HC(hipMemcpyAsync(out_host, P.out, (size_t)P.nseg * sizeof(double), hipMemcpyDeviceToHost, stream));
    HC(hipStreamSynchronize(stream));
}

void Device::rowdiag_plan_free(long long id) {
    if (id < 0 || id >= (long long)rd_plans_.size() || !rd_plans_[id].seg) return;
    RowDiagPlan &P = rd_plans_[id];
    (void)hipFree(P.seg); (void)hipFree(P.off); (void)hipFree(P.p); (void)hipFree(P.q); (void)hipFree(P.vals); (void)hipFree(P.out);
    P = RowDiagPlan{};
}

void Device::dense_apply(const double *d_D, const double *d_T, double *d_R, long long n1, long long n2) {
    HC(hipSetDevice(device));
    launch_dense_apply(stream, d_D, d_T, d_R, (int)n1, n2);
    HC(hipGetLastError());
    HC(hipStreamSynchronize(stream));
}
void Device::transpose(const double *d_src, double *d_dst, long long rows, long long cols) {
    HC(hipSetDevice(device));
    launch_transpose(stream, d_src, d_dst, rows, cols);
    HC(hipGetLastError());
    HC(hipStreamSynchronize(stream));
}

void Device::copy_factor(double *out_host) {
    HC(hipSetDevice(device));
    HC(hipMemcpy(out_host, d_L_, (size_t)l_size_ * sizeof(double), hipMemcpyDeviceToHost));
}

}  // namespace gmrfx
