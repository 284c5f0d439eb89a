// symbolic.cpp -- elimination tree, column counts, supernodes, row structures, assembly maps
// and level schedule. See symbolic.h. Algorithms restated from the literature:
// elimination tree with path compression (Liu 1990), skeleton/least-common-ancestor column
// counts (Gilbert, Ng & Peyton 1994), relaxed supernode amalgamation (Ashcraft & Grimes 1989).
#include "symbolic.h"

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <map>
#include <numeric>
#include <stdexcept>
#include <system_error>
#include <thread>

namespace gmrfx {
namespace {

// Strict upper pattern of P A P' by column: for new column k the rows a < k, from graph G.
void permuted_upper(const Graph &G, const std::vector<i32> &perm, const std::vector<i32> &iperm,
                    std::vector<i64> &up, std::vector<i32> &ui) {
    i64 n = G.n;
    up.assign(n + 1, 0);
    for (i64 k = 0; k < n; k++) {
        i32 v = perm[k];
        i64 c = 0;
        for (i64 p = G.xadj[v]; p < G.xadj[v + 1]; p++) if (iperm[G.adj[p]] < k) c++;
        up[k + 1] = up[k] + c;
    }
    ui.resize(up[n]);
    for (i64 k = 0; k < n; k++) {
        i32 v = perm[k];
        i64 q = up[k];
        for (i64 p = G.xadj[v]; p < G.xadj[v + 1]; p++) {
            i32 a = iperm[G.adj[p]];
            if (a < k) ui[q++] = a;
        }
        std::sort(ui.begin() + up[k], ui.begin() + up[k + 1]);
    }
}

void etree(i64 n, const std::vector<i64> &up, const std::vector<i32> &ui, std::vector<i32> &parent) {
    parent.assign(n, -1);
    std::vector<i32> anc(n, -1);
    for (i64 k = 0; k < n; k++)
        for (i64 p = up[k]; p < up[k + 1]; p++) {
            i32 i = ui[p];
            while (i != -1 && i < k) {
                i32 nx = anc[i];
                anc[i] = (i32)k;
                if (nx == -1) parent[i] = (i32)k;
                i = nx;
            }
        }
}

// Postorder of a forest given parent[]; children visited in increasing index order.
void postorder(i64 n, const std::vector<i32> &parent, std::vector<i32> &post) {
    std::vector<i32> head(n, -1), next(n, -1);
    for (i64 j = n - 1; j >= 0; j--)
        if (parent[j] != -1) { next[j] = head[parent[j]]; head[parent[j]] = (i32)j; }
    post.resize(n);
    std::vector<i32> stack;
    stack.reserve(64);
    i64 k = 0;
    for (i64 r = 0; r < n; r++) {
        if (parent[r] != -1) continue;
        stack.push_back((i32)r);
        while (!stack.empty()) {
            i32 p = stack.back();
            i32 c = head[p];
            if (c == -1) { stack.pop_back(); post[k++] = p; }
            else { head[p] = next[c]; stack.push_back(c); }
        }
    }
}

// Column counts of L for a postordered tree (node index == postorder rank).
// `up` holds, per column k, rows a<k: by symmetry these are the entries (k, a) of the lower
// triangle, i.e. column a has row k. We need for each column j its lower rows i>j: build the
// transpose on the fly.
void colcounts(i64 n, const std::vector<i64> &up, const std::vector<i32> &ui,
               const std::vector<i32> &parent, std::vector<i32> &cc) {
    // lower pattern by column (rows i > j), rows ascending
    std::vector<i64> lp(n + 1, 0);
    for (i64 k = 0; k < n; k++) for (i64 p = up[k]; p < up[k + 1]; p++) lp[ui[p] + 1]++;
    for (i64 j = 0; j < n; j++) lp[j + 1] += lp[j];
    std::vector<i32> li(lp[n]);
    { std::vector<i64> w(lp.begin(), lp.end() - 1);
      for (i64 k = 0; k < n; k++) for (i64 p = up[k]; p < up[k + 1]; p++) li[w[ui[p]]++] = (i32)k; }
    std::vector<i32> first(n, -1), maxfirst(n, -1), prevleaf(n, -1), ancestor(n);
    std::vector<i64> delta(n, 0);
    for (i64 k = 0; k < n; k++) {
        i64 j = k;
        delta[j] = (first[j] == -1) ? 1 : 0;
        for (; j != -1 && first[j] == -1; j = parent[j]) first[j] = (i32)k;
    }
    for (i64 i = 0; i < n; i++) ancestor[i] = (i32)i;
    for (i64 j = 0; j < n; j++) {
        if (parent[j] != -1) delta[parent[j]]--;
        for (i64 p = lp[j]; p < lp[j + 1]; p++) {
            i32 i = li[p];
            if (first[j] <= maxfirst[i]) continue;  // j is not a leaf of row subtree i
            maxfirst[i] = first[j];
            i32 jprev = prevleaf[i];
            prevleaf[i] = (i32)j;
            if (jprev == -1) { delta[j]++; continue; }  // first leaf
            i32 q = jprev;
            while (q != ancestor[q]) q = ancestor[q];
            for (i32 s = jprev; s != q;) { i32 sp = ancestor[s]; ancestor[s] = q; s = sp; }
            delta[j]++;
            delta[q]--;
        }
        if (parent[j] != -1) ancestor[j] = parent[j];
    }
    std::vector<i64> c(delta);
    for (i64 j = 0; j < n; j++) if (parent[j] != -1) c[parent[j]] += c[j];
    cc.resize(n);
    for (i64 j = 0; j < n; j++) cc[j] = (i32)c[j];
}

}  // namespace

// Doubles a panel of c columns with leading dimension ld occupies in the factor storage: its columns are padded with zero
// columns to a multiple of 4 (one FP64-MFMA k-step) and followed by at least 16 zero doubles, rounded to 128 bytes. The
// padding is never written (the buffer is zeroed once). The chunk kernels of the sweep tasks (sweep_chunk.hip) read whole
// k-steps and whole 32-row pairs without clamps. What they read beyond a panel's entries is the same panel's next column,
// this padding -- or, in a panel's last padded column and behind a ragged last chunk (up to ~31 doubles resp. 15 columns
// further), the FIRST ENTRIES OF THE NEXT PANEL: always mapped memory (the storage ends in 16 x 320 doubles of slack), always
// finite while the neighbouring panels are (they are zero before their first factorisation and hold the previous factor's
// finite values while a pipelined call re-factors them), and only ever multiplied into rows that are discarded: the spare row
// of the local vector (forward) resp. the operand rows that meet its zeros (backward). Not "zeros", as rounds 5's comment said.
static inline i64 panel_span(i64 ld, i64 c) { return (ld * ((c + 3) & ~i64(3)) + 16 + 15) & ~i64(15); }

void analyze(i64 n, const i64 *colptr, const i64 *rowval, int base, const i64 *user_perm,
             const SymOptions &opt, Symbolic &S) {
    auto t0 = std::chrono::steady_clock::now();
    if (n <= 0) throw std::invalid_argument("n must be positive");
    if (n >= (i64)2147483000) throw std::invalid_argument("n too large for 32-bit node indices");
    if (base != 0 && base != 1) throw std::invalid_argument("index_base must be 0 or 1");
    if (colptr[0] != base) throw std::invalid_argument("colptr[0] must equal index_base");
    for (i64 j = 0; j < n; j++) if (colptr[j + 1] < colptr[j]) throw std::invalid_argument("colptr must be non-decreasing");
    S = Symbolic();
    S.n = n;
    S.nnz_in = colptr[n] - base;

    Graph G;
    build_graph(n, colptr, rowval, base, G);

    // ---- ordering ------------------------------------------------------------------------
    std::vector<i32> perm(n);
    if (user_perm) {
        std::vector<uint8_t> seen(n, 0);
        for (i64 k = 0; k < n; k++) {
            i64 v = user_perm[k] - base;
            if (v < 0 || v >= n || seen[v]) throw std::invalid_argument("perm is not a permutation");
            seen[v] = 1;
            perm[k] = (i32)v;
        }
    } else if (opt.ordering == 1) {
        std::iota(perm.begin(), perm.end(), 0);
    } else {
        nested_dissection(G, opt, perm);
    }
    std::vector<i32> iperm(n);
    for (i64 k = 0; k < n; k++) iperm[perm[k]] = (i32)k;

    // ---- etree + postorder (the final order is the caller's/ND order composed with an
    //      etree postorder: an equivalent reordering, same fill) -----------------------------
    std::vector<i64> up; std::vector<i32> ui, parent;
    permuted_upper(G, perm, iperm, up, ui);
    etree(n, up, ui, parent);
    {
        std::vector<i32> post;
        postorder(n, parent, post);
        bool ident = true;
        for (i64 k = 0; k < n; k++) if (post[k] != k) { ident = false; break; }
        if (!ident) {
            std::vector<i32> np(n);
            for (i64 k = 0; k < n; k++) np[k] = perm[post[k]];
            perm.swap(np);
            for (i64 k = 0; k < n; k++) iperm[perm[k]] = (i32)k;
            permuted_upper(G, perm, iperm, up, ui);
            etree(n, up, ui, parent);
        }
    }
    std::vector<i32> cc;
    colcounts(n, up, ui, parent, cc);
    for (i64 j = 0; j < n; j++) S.nnz_l_true += cc[j];

    // ---- supernodes: maximal chains with nested structure, then relaxed amalgamation -------
    std::vector<i32> sfirst;  // start column of each supernode
    std::vector<i32> nchild(n, 0);
    for (i64 j = 0; j < n; j++) if (parent[j] != -1) nchild[parent[j]]++;
    sfirst.push_back(0);
    // A WIDE run of columns that ends below a BRANCHING column j (j has other children besides j - 1) stays a supernode of its
    // own although its structure nests into j's: merged, its panel chain -- one dependent potrf / trsm / gemm step per 64
    // columns -- would run AFTER the sibling subtrees' instead of beside them (cfg 2: the root absorbed the 1000-column
    // separator of one half of the mesh: 47 chain steps behind the 16 of the other half's separator, instead of 31 behind 16
    // for both in parallel). Same fill, same flops.
    // Only where the chain is what a front costs: runs of 128 .. 4096 columns. A wider one (the 15 876-column separators of
    // cfg 4) is throughput work, and merged into its parent it saves a contribution block and an assembly: cfg 4 on one GPU
    // 5.27 s merged, 5.50 s split.
    const i64 merge_wide = opt.merge_wide >= 0 ? opt.merge_wide : 128;
    const i64 merge_wide_max = opt.merge_wide_max >= 0 ? opt.merge_wide_max : 4096;
    for (i64 j = 1; j < n; j++) {
        bool join = parent[j - 1] == j && cc[j - 1] == cc[j] + 1;
        if (join && nchild[j] >= 2 && j - sfirst.back() >= merge_wide && j - sfirst.back() <= merge_wide_max) join = false;
        if (!join) sfirst.push_back((i32)j);
    }
    i32 ns0 = (i32)sfirst.size();
    sfirst.push_back((i32)n);
    // supernodal parent: supernode of parent[last column]
    std::vector<i32> c2s(n);
    for (i32 s = 0; s < ns0; s++) for (i32 j = sfirst[s]; j < sfirst[s + 1]; j++) c2s[j] = s;
    std::vector<i32> sp0(ns0);
    for (i32 s = 0; s < ns0; s++) { i32 pj = parent[sfirst[s + 1] - 1]; sp0[s] = pj == -1 ? -1 : c2s[pj]; }

    // relaxed amalgamation (contiguous child -> parent merges only)
    const int relax_cols = opt.relax_cols > 0 ? opt.relax_cols : 32;
    const double relax_zeros = opt.relax_zeros > 0 ? opt.relax_zeros : 0.15;
    struct SN { i32 first, last1; i64 r; double nnz; i32 parent; bool alive; };
    std::vector<SN> sn(ns0);
    for (i32 s = 0; s < ns0; s++) {
        i64 c = sfirst[s + 1] - sfirst[s], r = cc[sfirst[s]];
        sn[s] = {sfirst[s], sfirst[s + 1], r, (double)(r * c - c * (c - 1) / 2), sp0[s], true};
    }
    std::vector<i32> nkids0(ns0, 0);
    for (i32 s = 0; s < ns0; s++) if (sp0[s] != -1) nkids0[sp0[s]]++;
    // merged_into[s]: representative after merges (child merged into parent => parent's id)
    std::vector<i32> rep(ns0);
    std::iota(rep.begin(), rep.end(), 0);
    auto find = [&](i32 s) { while (rep[s] != s) { rep[s] = rep[rep[s]]; s = rep[s]; } return s; };
    for (i32 p = 0; p < ns0; p++) {
        // p is final up to merges of its trailing children; try to absorb the supernode that
        // ends right before p's (current) first column while it is p's child.
        for (;;) {
            i32 f = sn[p].first;
            if (f == 0) break;
            i32 d = find(c2s[f - 1]);
            if (d == p || !sn[d].alive) break;
            i32 dp = sn[d].parent == -1 ? -1 : find(sn[d].parent);
            if (dp != p) break;
            i64 cd = sn[d].last1 - sn[d].first, cp = sn[p].last1 - sn[p].first;
            i64 c = cd + cp, r = cd + sn[p].r;
            double total = (double)(r * c - c * (c - 1) / 2);
            double nnz = sn[d].nnz + sn[p].nnz;
            double z = (total - nnz) / total;
            bool merge = (c <= 4) || (c <= relax_cols && z <= 0.8 * 0.5) || (z <= relax_zeros && c <= 4 * relax_cols) ||
                         (z <= 0.02);
            // A WIDE child with siblings stays a front of its own even when the merge is free of zeros (its rows cover the
            // parent): merged, its panel chain -- one dependent potrf / trsm / gemm step per 64 columns -- runs AFTER its
            // siblings' instead of beside them. (cfg 2: the root absorbed the 1000-column separator of one half: 47 chain
            // steps behind the 16 of the other half's separator instead of 31 behind 16 for both.)
            if (merge && cd >= merge_wide && cd <= merge_wide_max && nkids0[p] >= 2) merge = false;
            if (!merge) break;
            sn[p].first = sn[d].first;
            sn[p].r = r;
            sn[p].nnz = nnz;
            sn[d].alive = false;
            rep[d] = p;
        }
    }
    // compact
    std::vector<i32> newid(ns0, -1);
    S.sfirst.clear();
    for (i32 s = 0; s < ns0; s++) if (sn[s].alive) { newid[s] = (i32)S.sfirst.size(); S.sfirst.push_back(sn[s].first); }
    S.nsuper = (i32)S.sfirst.size();
    S.sfirst.push_back((i32)n);
    S.sparent.resize(S.nsuper);
    S.col2super.resize(n);
    for (i32 s = 0; s < ns0; s++) if (sn[s].alive) {
        i32 t = newid[s];
        S.sparent[t] = sn[s].parent == -1 ? -1 : newid[find(sn[s].parent)];
        for (i32 j = sn[s].first; j < sn[s].last1; j++) S.col2super[j] = t;
    }
    const i32 ns = S.nsuper;
    std::vector<i64> snr(ns);  // predicted row counts (column counts / amalgamation bookkeeping)
    for (i32 s = 0; s < ns0; s++) if (sn[s].alive) snr[newid[s]] = sn[s].r;

    // children lists
    S.childptr.assign(ns + 1, 0);
    for (i32 s = 0; s < ns; s++) if (S.sparent[s] != -1) S.childptr[S.sparent[s] + 1]++;
    for (i32 s = 0; s < ns; s++) S.childptr[s + 1] += S.childptr[s];
    S.children.resize(S.childptr[ns]);
    { std::vector<i64> w(S.childptr.begin(), S.childptr.end() - 1);
      for (i32 s = 0; s < ns; s++) if (S.sparent[s] != -1) S.children[w[S.sparent[s]]++] = s; }

    // ---- row structures: own columns, A-rows of own columns, children's trailing rows -------
    // lower pattern by column of P A P'
    std::vector<i64> lp(n + 1, 0);
    for (i64 k = 0; k < n; k++) for (i64 p = up[k]; p < up[k + 1]; p++) lp[ui[p] + 1]++;
    for (i64 j = 0; j < n; j++) lp[j + 1] += lp[j];
    std::vector<i32> li(lp[n]);
    { std::vector<i64> w(lp.begin(), lp.end() - 1);
      for (i64 k = 0; k < n; k++) for (i64 p = up[k]; p < up[k + 1]; p++) li[w[ui[p]]++] = (i32)k; }
    S.nnz_q_tri = lp[n] + n;

    S.rowptr.assign(ns + 1, 0);
    S.rows.clear();
    S.rows.reserve((size_t)(S.nnz_l_true / 8 + n));
    std::vector<i32> mark(n, -1);
    for (i32 s = 0; s < ns; s++) {
        i32 f = S.sfirst[s], l1 = S.sfirst[s + 1];
        i64 start = (i64)S.rows.size();
        for (i32 j = f; j < l1; j++) { S.rows.push_back(j); mark[j] = s; }
        i64 tail = (i64)S.rows.size();
        for (i32 j = f; j < l1; j++)
            for (i64 p = lp[j]; p < lp[j + 1]; p++) {
                i32 i = li[p];
                if (i >= l1 && mark[i] != s) { mark[i] = s; S.rows.push_back(i); }
            }
        for (i64 q = S.childptr[s]; q < S.childptr[s + 1]; q++) {
            i32 d = S.children[q];
            i32 cd = S.sfirst[d + 1] - S.sfirst[d];
            for (i64 p = S.rowptr[d] + cd; p < S.rowptr[d + 1]; p++) {
                i32 i = S.rows[p];
                if (i >= l1 && mark[i] != s) { mark[i] = s; S.rows.push_back(i); }
            }
        }
        std::sort(S.rows.begin() + tail, S.rows.end());
        S.rowptr[s + 1] = (i64)S.rows.size();
        if ((i64)S.rows.size() - start != snr[s]) throw std::runtime_error("internal: supernode row count disagrees with column counts");
    }
    S.sum_rows = (i64)S.rows.size();

    // consistency: first trailing row's supernode must be the parent
    for (i32 s = 0; s < ns; s++) {
        i32 c = S.ncols(s), r = S.nrows(s);
        i32 want = r > c ? S.col2super[S.rows[S.rowptr[s] + c]] : -1;
        if (want != S.sparent[s]) throw std::runtime_error("internal: supernodal etree inconsistent");
    }

    // ---- relative indices child -> parent -----------------------------------------------------
    S.rel.assign(S.sum_rows, -1);
    for (i32 s = 0; s < ns; s++) {
        i32 p = S.sparent[s];
        if (p == -1) continue;
        i32 c = S.ncols(s);
        i64 a = S.rowptr[s] + c, ae = S.rowptr[s + 1];
        i64 b = S.rowptr[p], be = S.rowptr[p + 1];
        for (; a < ae; a++) {
            while (b < be && S.rows[b] < S.rows[a]) b++;
            if (b == be || S.rows[b] != S.rows[a]) throw std::runtime_error("internal: child rows not contained in parent");
            S.rel[a] = (i32)(b - S.rowptr[p]);
        }
    }

    // ---- storage layout -------------------------------------------------------------------
    S.panelptr.assign(ns + 1, 0);
    S.ld.resize(ns);
    S.cbptr.assign(ns, 0);
    i64 off = 0, cb = 0;
    S.max_cols = S.max_rows = 0;
    S.flops = 0;
    for (i32 s = 0; s < ns; s++) {
        i64 c = S.ncols(s), r = S.nrows(s);
        i64 ld = (r + 1) & ~i64(1);
        S.ld[s] = (i32)ld;
        S.panelptr[s] = off;
        off += panel_span(ld, c);     // 128-byte aligned panels, zero columns up to a multiple of 4, >= 16 zero doubles behind
        S.cbptr[s] = cb;
        i64 m = r - c;
        cb += m * m;
        cb = (cb + 15) & ~i64(15);
        S.max_cols = std::max<i32>(S.max_cols, (i32)c);
        S.max_rows = std::max<i32>(S.max_rows, (i32)r);
        S.nnz_l_stored += r * c - c * (c - 1) / 2;
        double dc = (double)c, dm = (double)m;
        S.flops += dc * dc * dc / 3.0 + dc * dc * dm + dc * dm * dm;
    }
    S.panelptr[ns] = off;
    S.cb_arena = cb;

    S.diagoff.resize(n);
    for (i32 s = 0; s < ns; s++) {
        i32 c = S.ncols(s);
        for (i32 j = 0; j < c; j++) S.diagoff[S.sfirst[s] + j] = S.panelptr[s] + (i64)j * S.ld[s] + j;
    }

    // ---- levels --------------------------------------------------------------------------
    S.level.assign(ns, 0);
    for (i32 s = 0; s < ns; s++) {
        i32 p = S.sparent[s];
        if (p != -1) S.level[p] = std::max(S.level[p], S.level[s] + 1);
    }
    S.nlevels = 0;
    for (i32 s = 0; s < ns; s++) S.nlevels = std::max(S.nlevels, S.level[s] + 1);
    // Levels by DEPTH below the root instead of height above the leaves. A level costs the longest panel chain among its
    // fronts (one dependent potrf / trsm / gemm step per 64 columns), and sibling subtrees differ in height by a level or two:
    // by height the separators of one dissection generation are spread over two or three levels, each of which then pays that
    // generation's chain. By depth every generation sits on ONE level (cfg 2: 1, 2, 4, ..., 2^k fronts from the root down; sum
    // over the levels of the longest chain 116 -> 107 steps, the tree's critical path is 105) and leaves sit right below their
    // parents. Measured at cfg 2: factorisation 10.63 -> 10.32 ms, pipelined step 14.72 -> 14.39 ms (the top 6 / 9 / 12 levels
    // only: 14.65 / 14.46 / 14.45). top_by_depth (GMRFX_TOP_BY_DEPTH): how many levels from the root down; 0 = all by height.
    {
        const int K = opt.top_by_depth >= 0 ? opt.top_by_depth : (1 << 30);
        const i32 H = S.nlevels - 1;
        std::vector<i32> depth_level(ns, 0);
        for (i32 s = ns - 1; s >= 0; s--) depth_level[s] = S.sparent[s] == -1 ? H : depth_level[S.sparent[s]] - 1;   // parents have larger ids
        for (i32 s = 0; s < ns; s++)
            if (K > 0 && (long long)depth_level[s] >= (long long)H - K + 1) S.level[s] = depth_level[s];   // (an upward-closed set: children stay below parents)
    }
    // ---- sharding over ranks (multi-GPU): split the tree top-down into subtrees, deal them out ----
    S.shard_rank = opt.shard_rank;
    S.shard_world = std::max(1, opt.shard_world);
    S.owner.assign(ns, 0);
    S.is_top.assign(ns, 0);
    S.shard_level = S.nlevels;
    S.shard_plan = S.shard_world > 1 || opt.shard_min_top > 0;
    if (S.shard_plan) {
        const int W = S.shard_world;
        // weight of a front ~ its factorisation flops; of a subtree = sum over its fronts (postorder ids)
        std::vector<double> wsub(ns, 0.0), wown(ns, 0.0);
        for (i32 s = 0; s < ns; s++) {
            const double c = S.ncols(s), m = S.nrows(s) - S.ncols(s);
            wown[s] = c * c * c / 3.0 + c * c * m + c * m * m + 1.0;
            wsub[s] += wown[s];
            if (S.sparent[s] != -1) wsub[S.sparent[s]] += wsub[s];
        }
        std::vector<i32> T;                     // current subtree roots
        std::vector<uint8_t> top(ns, 0);
        for (i32 s = 0; s < ns; s++) if (S.sparent[s] == -1) T.push_back(s);
        // LPT deal of the current subtrees; returns the largest local load. The top fronts run after the local work,
        // each on one rank of its group, independent ones concurrently: a plan costs about
        //     max_r local_r + (heaviest root-to-leaf chain of top fronts),
        // and splitting the heaviest subtree lowers the first term and may raise the second.
        auto makespan = [&](std::vector<i32> &assign_out) {
            std::vector<i32> ord(T);
            std::sort(ord.begin(), ord.end(), [&](i32 a, i32 b) { return wsub[a] != wsub[b] ? wsub[a] > wsub[b] : a < b; });
            std::vector<double> load(W, 0.0);
            assign_out.assign(ns, -2);
            for (i32 s : ord) {
                int r = (int)(std::min_element(load.begin(), load.end()) - load.begin());
                load[r] += wsub[s];
                assign_out[s] = r;
            }
            return *std::max_element(load.begin(), load.end());
        };
        std::vector<i32> asg;
        auto top_chain = [&]() {          // heaviest chain inside the top set (children have smaller ids)
            std::vector<double> cp(ns, 0.0);
            double best = 0.0;
            for (i32 s = 0; s < ns; s++) {
                if (!top[s]) continue;
                double m = 0.0;
                for (i64 q = S.childptr[s]; q < S.childptr[s + 1]; q++) m = std::max(m, cp[S.children[q]]);
                cp[s] = m + wown[s];
                best = std::max(best, cp[s]);
            }
            return best;
        };
        int nsplit = 0;
        for (int it = 0; it < 256; it++) {
            const double cost = makespan(asg) + top_chain();
            i32 best = -1;
            for (i32 s : T) if (S.childptr[s + 1] > S.childptr[s] && (best == -1 || wsub[s] > wsub[best])) best = s;
            if (best == -1) break;
            // tentatively split `best`
            std::vector<i32> T0 = T;
            T.erase(std::find(T.begin(), T.end(), best));
            for (i64 q = S.childptr[best]; q < S.childptr[best + 1]; q++) T.push_back(S.children[q]);
            top[best] = 1;
            std::vector<i32> asg2;
            const double cost2 = makespan(asg2) + top_chain();
            if ((int)T0.size() >= W && cost2 >= 0.98 * cost && nsplit >= opt.shard_min_top) { T = T0; top[best] = 0; break; }     // no longer pays
            nsplit++;
        }
        makespan(asg);
        // owner: subtree root's rank for everything below it (postorder: parents after children)
        S.is_top.assign(ns, 0);
        for (i32 s = ns - 1; s >= 0; s--) {
            if (top[s]) { S.is_top[s] = 1; S.owner[s] = -1; }
            else if (asg[s] >= 0) S.owner[s] = asg[s];
            else S.owner[s] = S.owner[S.sparent[s]];     // inside a subtree: same as the parent (processed first)
        }
        // top fronts, children first: the least loaded (in top flops) of the owners of the children -- one child's
        // contribution block stays where it is, independent top fronts spread over the ranks of their group
        {
            std::vector<double> topload(W, 0.0);
            for (i32 s = 0; s < ns; s++) {
                if (!S.is_top[s]) continue;
                i32 best = -1;
                for (i64 q = S.childptr[s]; q < S.childptr[s + 1]; q++) {
                    const i32 r = S.owner[S.children[q]];
                    if (best == -1 || topload[r] < topload[best] || (topload[r] == topload[best] && r < best)) best = r;
                }
                if (best == -1) best = 0;       // (a top front always has children: it was split)
                S.owner[s] = best;
                topload[best] += wown[s];
            }
        }
        // top fronts go to levels >= shard_level = 1 + highest level of any assigned front
        i32 lmax = -1;
        for (i32 s = 0; s < ns; s++) if (!S.is_top[s]) lmax = std::max(lmax, S.level[s]);
        S.shard_level = lmax + 1;
        for (i32 s = 0; s < ns; s++) {
            if (!S.is_top[s]) continue;
            i32 lv = S.shard_level;
            for (i64 q = S.childptr[s]; q < S.childptr[s + 1]; q++) {
                const i32 d = S.children[q];
                if (S.is_top[d]) lv = std::max(lv, S.level[d] + 1);    // children have smaller ids: already final
            }
            S.level[s] = lv;
        }
        for (i32 d = 0; d < ns; d++) {
            const i32 p = S.sparent[d];
            if (p != -1 && S.owner[p] != S.owner[d]) S.shard_edges.push_back(d);
        }
        std::stable_sort(S.shard_edges.begin(), S.shard_edges.end(), [&](i32 a, i32 b) {
            const i32 la = S.level[S.sparent[a]], lb = S.level[S.sparent[b]];
            return la != lb ? la < lb : a < b;
        });
        {   // every assigned subtree (postorder: a subtree is a contiguous id range ending at its root)
            std::vector<i32> cnt(ns, 1);
            for (i32 s = 0; s < ns; s++) if (S.sparent[s] != -1) cnt[S.sparent[s]] += cnt[s];
            for (i32 s = 0; s < ns; s++) {
                if (S.is_top[s]) continue;
                const i32 p = S.sparent[s];
                if (p != -1 && !S.is_top[p]) continue;          // not a subtree root
                S.shard_sub_root.push_back(s);
                S.shard_sub_col0.push_back(S.sfirst[s - cnt[s] + 1]);
            }
        }
        S.nlevels = 0;
        for (i32 s = 0; s < ns; s++) S.nlevels = std::max(S.nlevels, S.level[s] + 1);
        S.nlevels = std::max(S.nlevels, S.shard_level);
    }
    // distributed top fronts (symbolic.h): wide top fronts whose group has more than one rank
    S.dist_fronts.clear(); S.dist_index.assign(ns, -1); S.dist_gptr.assign(1, 0); S.dist_grank.clear();
    if (S.shard_plan) {
        const int min_cols = opt.dist_min_cols >= 0 ? opt.dist_min_cols : 4096;
        std::vector<std::vector<i32>> grp(ns);          // ranks below (and at) every top front
        for (i32 s = 0; s < ns; s++) {
            if (!S.is_top[s]) continue;
            std::vector<i32> g;
            for (i64 q = S.childptr[s]; q < S.childptr[s + 1]; q++) {
                const i32 d = S.children[q];
                if (S.is_top[d]) g.insert(g.end(), grp[d].begin(), grp[d].end());
                else g.push_back(S.owner[d]);
            }
            std::sort(g.begin(), g.end());
            g.erase(std::unique(g.begin(), g.end()), g.end());
            grp[s] = g;
            if (min_cols > 0 && !(opt.subtree_max > 0) && S.ncols(s) >= min_cols && g.size() >= 2) {
                S.dist_index[s] = (i32)S.dist_fronts.size();
                S.dist_fronts.push_back(s);
                S.dist_grank.insert(S.dist_grank.end(), g.begin(), g.end());
                S.dist_gptr.push_back((i32)S.dist_grank.size());
            }
        }
    }
    // does this rank execute front s?  (a distributed front: owner[s] runs its sweeps / selected inversion)
    auto mine = [&](i32 s) { return !S.shard_plan || S.owner[s] == S.shard_rank; };

    // ---- per-rank storage of a sharded factorisation (round 3) ---------------------------------------------------------
    // A rank only holds what it works on: the PANELS of its own fronts (panels never cross ranks; every other front gets a
    // zero-length panel, and whoever reads an entry of one -- the selected-inverse getters -- is sent to the zeroed slack
    // word behind the buffer, panelptr[nsuper]); update vectors W and contribution blocks of its own fronts plus those of
    // the CROSS-EDGE children (the children of an owner-crossing tree edge, sent child's owner -> parent's owner). The
    // cross-edge children come first in both layouts, in front order, identically on every rank: a transfer still goes
    // into the SAME offset on both ends, and no rank needs to know another rank's private layout.
    S.cross_child.assign(ns, 0);
    for (i32 d : S.shard_edges) S.cross_child[d] = 1;
    for (i32 s : S.dist_fronts) {       // the block of a distributed front is written by several ranks, the blocks of its children
        S.cross_child[s] = 1;           // are read by several: all of them live in the exchange region
        for (i64 q = S.childptr[s]; q < S.childptr[s + 1]; q++) S.cross_child[S.children[q]] = 1;
    }
    if (S.shard_plan) {
        i64 off = 0;
        for (i32 s = 0; s < ns; s++) {
            S.panelptr[s] = off;
            if (S.stored_here(s)) off += S.compact_here(s) ? S.compact_span(s) : panel_span(S.ld[s], S.ncols(s));
        }
        S.panelptr[ns] = off;
        for (i32 s = 0; s < ns; s++)
            for (i32 j = 0; j < S.ncols(s); j++)
                S.diagoff[S.sfirst[s] + j] = mine(s) ? S.panelptr[s] + (i64)j * S.ld[s] + j : off;      // (the root's diagonal counts on owner[root])
    }
    S.wptr.assign(ns + 1, 0);
    {
        i64 w = 0;
        if (S.shard_plan) {
            for (i32 s = 0; s < ns; s++) if (S.cross_child[s]) { S.wptr[s] = w; w += S.nrows(s) - S.ncols(s); }
            for (i32 s = 0; s < ns; s++) if (!S.cross_child[s] && mine(s)) { S.wptr[s] = w; w += S.nrows(s) - S.ncols(s); }
        } else {
            for (i32 s = 0; s < ns; s++) { S.wptr[s] = w; w += S.nrows(s) - S.ncols(s); }
        }
        S.wptr[ns] = w;
    }
    // small fronts (fused LDS kernels, small.hip): r <= 96 or r <= 128 rows and <= 64 columns
    S.small_rows = opt.small_front_rows >= 0 ? opt.small_front_rows : 64;    // measured on cfg 2 with the depth-levelled tree (round 3): 32 -> 14.47, 48 -> 14.20, 64 -> 14.14, 72 -> 14.18, 96 -> 14.32, 128 -> 14.98 ms per step
    S.is_small.resize(ns);
    static const int kClsRows[4] = {48, 64, 96, 128};
    auto cls = [&](i32 s) -> int {
        if (S.ncols(s) > 64) return 4;
        for (int k = 0; k < 4; k++) if (S.nrows(s) <= std::min(kClsRows[k], S.small_rows)) return k;
        return 4;
    };
    for (i32 s = 0; s < ns; s++) S.is_small[s] = cls(s) < 4;
    // ---- subtree tasks ---------------------------------------------------------------------
    // Supernodes are numbered in postorder, so the subtree of s is the id range [s-cnt+1, s].
    S.in_subtree.assign(ns, 0);
    {
        const int submax = opt.subtree_max >= 0 ? opt.subtree_max : 0;   // subtree tasks are off by default: level-batched small fronts measured 3 % faster
        std::vector<i32> cnt(ns, 1), maxr(ns, 0);
        std::vector<uint8_t> ok(ns, 0);
        for (i32 s = 0; s < ns; s++) {
            bool good = submax > 0 && cls(s) <= 2;
            maxr[s] = S.nrows(s);
            for (i64 q = S.childptr[s]; q < S.childptr[s + 1]; q++) {
                const i32 d = S.children[q];
                good = good && ok[d];
                cnt[s] += cnt[d];
                maxr[s] = std::max(maxr[s], maxr[d]);
            }
            ok[s] = good && cnt[s] <= submax;
        }
        std::vector<i32> roots[3];
        for (i32 s = 0; s < ns; s++) {
            if (!ok[s]) continue;
            const i32 p = S.sparent[s];
            if (p != -1 && ok[p]) continue;
            const int k = maxr[s] <= 48 ? 0 : (maxr[s] <= 64 ? 1 : 2);
            roots[k].push_back(s);
            for (i32 t = s - cnt[s] + 1; t <= s; t++) S.in_subtree[t] = 1;
        }
        for (int k = 0; k < 3; k++) {
            S.nsub_cls[k] = (i32)roots[k].size();
            for (i32 s : roots[k]) { S.sub_first.push_back(s - cnt[s] + 1); S.sub_last.push_back(s); }
        }
    }
    S.levelptr.assign(S.nlevels + 1, 0);
    // (distributed fronts are factored by the block phases of Device::dist_phase, not by the level loop)
    for (i32 s = 0; s < ns; s++) if (!S.in_subtree[s] && mine(s) && !S.is_dist(s)) S.levelptr[S.level[s] + 1]++;
    for (i32 l = 0; l < S.nlevels; l++) S.levelptr[l + 1] += S.levelptr[l];
    S.levellist.resize(S.levelptr[S.nlevels]);
    { std::vector<i64> w(S.levelptr.begin(), S.levelptr.end() - 1);
      for (i32 s = 0; s < ns; s++) if (!S.in_subtree[s] && mine(s) && !S.is_dist(s)) S.levellist[w[S.level[s]]++] = s; }
    S.level_nsmall.assign(S.nlevels, 0);
    S.level_ncls.assign((size_t)S.nlevels * 4, 0);
    for (i32 l = 0; l < S.nlevels; l++) {
        auto b = S.levellist.begin() + S.levelptr[l], e = S.levellist.begin() + S.levelptr[l + 1];
        std::stable_sort(b, e, [&](i32 x, i32 y) {
            const int cx = cls(x), cy = cls(y);
            if (cx != cy) return cx < cy;
            if (cx < 4) return x < y;
            return S.ncols(x) != S.ncols(y) ? S.ncols(x) > S.ncols(y) : x < y;
        });
        i32 k = 0;
        for (auto it = b; it != e; ++it) { const int cc = cls(*it); if (cc < 4) { k++; S.level_ncls[(size_t)l * 4 + cc]++; } }
        S.level_nsmall[l] = k;
        S.n_small += k;
        S.n_big += (e - b) - k;
    }
    for (i32 s = 0; s < ns; s++) S.n_small += S.in_subtree[s];
    // all-front level lists for the (top-down) selected inversion
    S.sel_levelptr.assign(S.nlevels + 1, 0);
    for (i32 s = 0; s < ns; s++) if (mine(s)) S.sel_levelptr[S.level[s] + 1]++;
    for (i32 l = 0; l < S.nlevels; l++) S.sel_levelptr[l + 1] += S.sel_levelptr[l];
    S.sel_levellist.resize(S.sel_levelptr[S.nlevels]);
    { std::vector<i64> w(S.sel_levelptr.begin(), S.sel_levelptr.end() - 1);
      for (i32 s = 0; s < ns; s++) if (mine(s)) S.sel_levellist[w[S.level[s]]++] = s; }
    S.sel_level_nsmall.assign(S.nlevels, 0);
    for (i32 l = 0; l < S.nlevels; l++) {
        auto b = S.sel_levellist.begin() + S.sel_levelptr[l], e = S.sel_levellist.begin() + S.sel_levelptr[l + 1];
        std::stable_partition(b, e, [&](i32 x) { return cls(x) < 4; });
        i32 k = 0;
        for (auto it = b; it != e; ++it) if (cls(*it) < 4) k++;
        S.sel_level_nsmall[l] = k;
    }

    // ---- sweep tasks (see symbolic.h) -------------------------------------------------------------
    S.in_swt.assign(ns, 0);
    S.lrow.assign((size_t)S.sum_rows, -1);
    {
        const int rcap = opt.subtree_max > 0 ? 0 : std::min(288,   // (not with the legacy subtree tasks)
                                                                               opt.sweep_task_rows >= 0 ? opt.sweep_task_rows : 288);   // 288 = rows of the local vector in sweep_chunk.hip / sweep_wave.hip
        S.swt_rows = rcap;
        std::vector<i32> cnt(ns, 1), ncol(ns, 0), maxc(ns, 0), nchk(ns, 0);
        std::vector<double> work(ns, 0.0);
        std::vector<uint8_t> ok(ns, 0);
        for (i32 s = 0; s < ns; s++) {
            ncol[s] += S.ncols(s);
            maxc[s] = std::max(maxc[s], S.ncols(s));
            for (i32 j0 = 0; j0 < S.ncols(s); j0 += 16)       // forward chunk records of this front (<= 128 target rows each)
                nchk[s] += std::max(1, (S.nrows(s) - std::min(S.ncols(s), j0 + 16) + 127) / 128);
            work[s] += (double)S.nrows(s) * S.ncols(s);
            bool good = rcap > 0;
            for (i64 q = S.childptr[s]; q < S.childptr[s + 1]; q++) good = good && ok[S.children[q]];
            // (chunk kernels: at most 96 chunk records per task in LDS)
            ok[s] = good && !S.is_top[s] && maxc[s] <= 64 && ncol[s] + std::max(S.nrows(s) - S.ncols(s), 15) <= rcap && cnt[s] <= 64 && nchk[s] <= 96;
            const i32 p = S.sparent[s];
            if (p != -1) { cnt[p] += cnt[s]; ncol[p] += ncol[s]; maxc[p] = std::max(maxc[p], maxc[s]); work[p] += work[s]; nchk[p] += nchk[s]; }
        }
        std::vector<i32> roots;
        for (i32 s = 0; s < ns; s++) {
            if (!ok[s] || cnt[s] < 2) continue;                       // a single front gains nothing over the level kernels
            const i32 p = S.sparent[s];
            if (p != -1 && ok[p]) continue;                           // not maximal
            roots.push_back(s);
        }
        std::stable_sort(roots.begin(), roots.end(), [&](i32 a, i32 b) { return work[a] > work[b]; });
        for (i32 t : roots) {
            const i32 f = t - cnt[t] + 1;
            for (i32 d = f; d <= t; d++) S.in_swt[d] = 1;      // (a task lies inside ONE assigned subtree: one owner)
            if (!mine(t)) continue;                            // sharded handles keep the tasks of their own subtrees
            S.swt_first.push_back(f); S.swt_last.push_back(t);
            const i32 col0 = S.sfirst[f], col1 = S.sfirst[t + 1], nt = col1 - col0;
            const i32 ct = S.ncols(t);
            const i32 *rt = S.rows.data() + S.rowptr[t] + ct;          // the root's trailing rows (sorted)
            const i32 mt = S.nrows(t) - ct;
            for (i32 d = f; d <= t; d++) {
                S.in_swt[d] = 1;
                const i32 cd = S.ncols(d);
                for (i64 k = S.rowptr[d] + cd; k < S.rowptr[d + 1]; k++) {
                    const i32 g = S.rows[k];
                    if (g < col1) {
                        if (g < col0) throw std::runtime_error("internal: sweep task row below its subtree");
                        S.lrow[k] = g - col0;
                    } else {
                        const i32 *it = std::lower_bound(rt, rt + mt, g);
                        if (it == rt + mt || *it != g) throw std::runtime_error("internal: sweep task row outside the root's structure");
                        S.lrow[k] = nt + (i32)(it - rt);
                    }
                }
            }
            // ---- the task's chunks (symbolic.h: SwChunk) ----
            S.swc_ptr.push_back((i32)S.swc_fwd.size());
            S.swc_bptr.push_back((i32)S.swc_bwd.size());
            std::vector<Symbolic::SwChunk> real;                    // the task's chunks, postorder (ids are global: one packed diagonal tile each)
            std::vector<i32> first_chunk(t - f + 2, 0);             // per front of the task: index of its first chunk in `real`
            for (i32 d = f; d <= t; d++) {
                const i32 cd = S.ncols(d), rd = S.nrows(d), od = S.sfirst[d] - col0;
                first_chunk[d - f] = (i32)real.size();
                for (i32 j0 = 0; j0 < cd; j0 += 16) {
                    Symbolic::SwChunk ch;
                    const i32 cc = std::min(16, cd - j0), row0 = j0 + cc;
                    ch.pa = S.panelptr[d] + (i64)j0 * S.ld[d] + row0;
                    ch.ld = S.ld[d];
                    ch.o = (int16_t)(od + j0);
                    ch.cc = (int16_t)cc;
                    ch.nt = rd - row0;
                    ch.lr = (i32)S.swc_rows.size();
                    ch.nbar = 0;
                    ch.id = S.swc_nchunks++;
                    for (i32 k = row0; k < rd; k++) S.swc_rows.push_back(k < cd ? od + k : S.lrow[S.rowptr[d] + k]);
                    while ((S.swc_rows.size() - (size_t)ch.lr) % 32) S.swc_rows.push_back(-1);
                    real.push_back(ch);
                    // forward records: at most 128 target rows each (one pair of 16-row tiles per row-tile slot); a longer
                    // chunk becomes several records with the same diagonal block -- every one recomputes the same y
                    for (i32 k0 = 0; k0 == 0 || k0 < ch.nt; k0 += 128) {
                        Symbolic::SwChunk part = ch;
                        part.pa += k0; part.lr += k0; part.nt = std::min(128, ch.nt - k0);
                        S.swc_fwd.push_back(part);
                    }
                }
            }
            first_chunk[t - f + 1] = (i32)real.size();
            // backward programs: depth of a chunk in the chunk tree (root front's LAST chunk = depth 0; inside a front the
            // chunks run last to first; a child front starts below its parent's first chunk)
            const i32 nch = (i32)real.size();
            std::vector<i32> depth(nch, 0), dstart(t - f + 1, 0);
            i32 ngroups = 0;
            for (i32 d = t; d >= f; d--) {
                const i32 c0 = first_chunk[d - f], c1 = first_chunk[d - f + 1];
                i32 ds = 0;
                if (d != t) { const i32 p = S.sparent[d]; ds = dstart[p - f] + (first_chunk[p - f + 1] - first_chunk[p - f]); }
                dstart[d - f] = ds;
                for (i32 q = c1 - 1; q >= c0; q--) { depth[q] = ds + (c1 - 1 - q); ngroups = std::max(ngroups, depth[q] + 1); }
            }
            std::vector<i32> order(nch);
            std::iota(order.begin(), order.end(), 0);
            std::stable_sort(order.begin(), order.end(), [&](i32 a, i32 b) { return depth[a] != depth[b] ? depth[a] < depth[b] : a > b; });
            std::vector<std::vector<i32>> prog(Symbolic::kSwSlots);
            for (i32 k = 0, g0 = 0; k < nch; k++) {
                if (k > 0 && depth[order[k]] != depth[order[k - 1]]) g0 = k;
                prog[(k - g0) % Symbolic::kSwSlots].push_back(order[k]);
            }
            for (int q = 0; q < Symbolic::kSwSlots; q++) {
                i32 passed = 0;
                for (i32 ci : prog[q]) {
                    Symbolic::SwChunk ch = real[ci];
                    ch.nbar = depth[ci] - passed;
                    passed = depth[ci];
                    S.swc_bwd.push_back(ch);
                }
                S.swc_slot.push_back((i32)prog[q].size());
            }
            for (int q = 0; q < Symbolic::kSwSlots; q++) {
                const i32 passed = prog[q].empty() ? 0 : depth[prog[q].back()];
                S.swc_slot.push_back(ngroups - passed);
            }
        }
        S.swc_ptr.push_back((i32)S.swc_fwd.size());
        S.swc_bptr.push_back((i32)S.swc_bwd.size());
    }
    S.sw_levelptr.assign(S.nlevels + 1, 0);
    for (i32 s = 0; s < ns; s++) if (!S.in_subtree[s] && !S.in_swt[s] && mine(s)) S.sw_levelptr[S.level[s] + 1]++;
    for (i32 l = 0; l < S.nlevels; l++) S.sw_levelptr[l + 1] += S.sw_levelptr[l];
    S.sw_levellist.resize(S.sw_levelptr[S.nlevels]);
    { std::vector<i64> w(S.sw_levelptr.begin(), S.sw_levelptr.end() - 1);
      for (i32 s = 0; s < ns; s++) if (!S.in_subtree[s] && !S.in_swt[s] && mine(s)) S.sw_levellist[w[S.level[s]]++] = s; }
    S.sw_level_nsmall.assign(S.nlevels, 0);
    S.sw_level_ncls.assign((size_t)S.nlevels * 4, 0);
    for (i32 l = 0; l < S.nlevels; l++) {
        auto b = S.sw_levellist.begin() + S.sw_levelptr[l], e = S.sw_levellist.begin() + S.sw_levelptr[l + 1];
        std::stable_sort(b, e, [&](i32 x, i32 y) {
            const int cx = cls(x), cy = cls(y);
            if (cx != cy) return cx < cy;
            if (cx < 4) return x < y;
            return S.ncols(x) != S.ncols(y) ? S.ncols(x) > S.ncols(y) : x < y;
        });
        i32 k = 0;
        for (auto it = b; it != e; ++it) { const int cc = cls(*it); if (cc < 4) { k++; S.sw_level_ncls[(size_t)l * 4 + cc]++; } }
        S.sw_level_nsmall[l] = k;
    }

    // ---- contribution-block arena with lifetime reuse ------------------------------------------
    // The level schedule fixes when a block is written and when it is last read: CB_s lives from level(s) to
    // level(parent(s)) in the factorisation; in the (top-down) selected inversion the trailing inverse block
    // of s lives from level(s) down to the lowest level of its children. Slots are handed out first-fit level
    // by level (the blocks a level reads are only released after the level's own blocks have their slots, so
    // nothing a level writes aliases anything it reads). A sharded handle has its own layout per rank (below).
    // Subtree tasks run fronts out of level order: linear layout.
    S.zbptr = S.cbptr;
    if (!(opt.subtree_max > 0)) {
        struct Arena {      // best fit, coalescing free list (offset-ordered map + size-ordered index)
            std::map<i64, i64> by_off;
            std::multimap<i64, i64> by_size;
            i64 top = 0;
            void drop(std::map<i64, i64>::iterator it) {
                auto r = by_size.equal_range(it->second);
                for (auto k = r.first; k != r.second; ++k) if (k->second == it->first) { by_size.erase(k); break; }
                by_off.erase(it);
            }
            void put(i64 off, i64 sz) { by_off[off] = sz; by_size.emplace(sz, off); }
            i64 alloc(i64 sz) {
                auto k = by_size.lower_bound(sz);
                if (k != by_size.end()) {
                    const i64 off = k->second, rest = k->first - sz;
                    drop(by_off.find(off));
                    if (rest > 0) put(off + sz, rest);
                    return off;
                }
                const i64 off = top;
                top += sz;
                return off;
            }
            void release(i64 off, i64 sz) {
                auto nx = by_off.lower_bound(off);
                if (nx != by_off.end() && off + sz == nx->first) { sz += nx->second; drop(nx); }
                auto pv = by_off.lower_bound(off);
                if (pv != by_off.begin()) {
                    --pv;
                    if (pv->first + pv->second == off) { off = pv->first; sz += pv->second; drop(pv); }
                }
                if (off + sz == top) top = off;      // give the tail back
                else put(off, sz);
            }
        };
        auto bsz = [&](i32 s) { const i64 m = S.nrows(s) - S.ncols(s); return ((m * m) + 15) & ~i64(15); };
        i64 peak = 0;
        if (!S.shard_plan) {
            std::vector<std::vector<i32>> bylevel(S.nlevels);
            for (i32 s = 0; s < ns; s++) bylevel[S.level[s]].push_back(s);
            {   // factorisation: bottom-up
                Arena A;
                for (i32 l = 0; l < S.nlevels; l++) {
                    for (i32 s : bylevel[l]) if (bsz(s) > 0) S.cbptr[s] = A.alloc(bsz(s));
                    peak = std::max(peak, A.top);
                    for (i32 s : bylevel[l])
                        for (i64 q = S.childptr[s]; q < S.childptr[s + 1]; q++) {
                            const i32 d = S.children[q];
                            if (bsz(d) > 0) A.release(S.cbptr[d], bsz(d));
                        }
                }
            }
            {   // selected inversion: top-down
                Arena A;
                std::vector<i32> minchild(ns, -1);
                for (i32 s = 0; s < ns; s++) {
                    i32 mc = S.level[s];                                   // no children: released after its own level
                    for (i64 q = S.childptr[s]; q < S.childptr[s + 1]; q++) mc = std::min(mc, S.level[S.children[q]]);
                    minchild[s] = mc;
                }
                std::vector<std::vector<i32>> dies(S.nlevels);
                for (i32 s = 0; s < ns; s++) dies[minchild[s]].push_back(s);
                for (i32 l = S.nlevels - 1; l >= 0; l--) {
                    for (i32 s : bylevel[l]) if (bsz(s) > 0) S.zbptr[s] = A.alloc(bsz(s));
                    peak = std::max(peak, A.top);
                    for (i32 s : dies[l]) if (bsz(s) > 0) A.release(S.zbptr[s], bsz(s));
                }
            }
        } else {
            // SHARDED (round 6): a PER-RANK layout. Rounds 3-5 put the blocks of every cross-edge child -- and of every distributed
            // front and its children -- into an exchange region laid out identically on every rank, without reuse ("a handful of
            // blocks"): at cfg 4 / world 8 that region is 75 GB of 14 blocks on EVERY rank, 90 of the 120-129 GB a rank needs. A
            // transfer does not need equal offsets on its two ends (each rank builds its views from its own tables:
            // gmrfx_shard_transfers / gmrfx_shard_edges report THIS rank's offsets), so every rank now gives slots only to the
            // blocks it touches, all of them by lifetime:
            //   factorisation  CB_s is PRODUCED here (this rank executes s, or is a member of the group of the distributed front
            //                  s: its own column blocks) -> born at level(s); or only RECEIVED here (whole, or the column ranges
            //                  that fall into this rank's blocks of a distributed parent) -> born when level(parent) starts: the
            //                  receive is posted behind this rank's phases of the level below, stream-ordered; everything dies
            //                  behind level(parent) (a sender's wait() orders its later kernels behind the send).
            //   selected inv.  the trailing inverse block of s lives on owner[s] from level(s) down to its lowest child, and on
            //                  the owner of a parent on another rank from the gather (gmrfx_selinv_phase(1, level(s))) to the
            //                  end of that level (sent away).
            // A block keeps the full (r - c)^2 layout wherever it lives (kernels address columns globally); only the slots move.
            const i32 me = S.shard_rank;
            auto in_group = [&](i32 s) { return S.is_dist(s) && S.group_pos(s, me) >= 0; };
            auto produce = [&](i32 s) { return mine(s) || in_group(s); };
            auto receive = [&](i32 s) { const i32 p = S.sparent[s]; return p >= 0 && (S.is_dist(p) ? in_group(p) : S.owner[p] == me); };
            for (i32 s = 0; s < ns; s++) S.cbptr[s] = S.zbptr[s] = 0;
            std::vector<std::vector<i32>> born(S.nlevels), arrive(S.nlevels), consumed(S.nlevels);
            for (i32 s = 0; s < ns; s++) {
                if (bsz(s) <= 0) continue;
                const i32 p = S.sparent[s];
                const bool pr = produce(s), rc = receive(s);
                if (pr) born[S.level[s]].push_back(s);
                else if (rc) arrive[S.level[p]].push_back(s);
                if ((pr || rc) && p >= 0) consumed[S.level[p]].push_back(s);
            }
            {   // factorisation: bottom-up
                // The handful of blocks that live into the TOP phase (consumed at a level >= shard_level: the contribution blocks of
                // this rank's subtree roots, of the distributed fronts it is a member of, and what it receives) are 4-8 GB each at
                // cfg 4, and the online best-fit allocator left 7 GB of holes between them (rank 1 at world 8: 31.2 GB of arena for
                // 24.3 GB live). They are laid out OFFLINE instead -- largest first, each at the lowest offset free of every
                // already placed block whose lifetime [birth level, consumption level] meets its own -- with the blocks born in
                // the subtree phase stacked at the bottom; the subtree phase's own blocks keep the online allocator, above that
                // stack (they are all dead when the first top level starts, so the other top blocks may lie over them).
                const i32 L0 = S.shard_level;
                struct TB { i32 s; i64 sz; i32 t0, t1; i64 off; };
                std::vector<TB> top;
                std::vector<uint8_t> is_topblk(ns, 0);
                for (i32 l = 0; l < S.nlevels; l++) {
                    for (i32 s : arrive[l]) { const i32 p = S.sparent[s]; if (S.level[p] >= L0) { top.push_back({s, bsz(s), S.level[p], S.level[p], -1}); is_topblk[s] = 1; } }
                    for (i32 s : born[l]) { const i32 p = S.sparent[s]; if (p >= 0 && S.level[p] >= L0) { top.push_back({s, bsz(s), l, S.level[p], -1}); is_topblk[s] = 1; } }
                }
                std::stable_sort(top.begin(), top.end(), [&](const TB &a, const TB &b) {
                    const bool ea = a.t0 < L0, eb = b.t0 < L0;          // born in the subtree phase: first, at the bottom
                    if (ea != eb) return ea;
                    return a.sz != b.sz ? a.sz > b.sz : a.s < b.s;
                });
                i64 bottom = 0, toppeak = 0;
                for (size_t k = 0; k < top.size(); k++) {
                    TB &b = top[k];
                    if (b.t0 < L0) { b.off = bottom; bottom += b.sz; }
                    else {
                        std::vector<std::pair<i64, i64>> busy;
                        for (size_t q = 0; q < k; q++) if (top[q].t0 <= b.t1 && b.t0 <= top[q].t1) busy.emplace_back(top[q].off, top[q].off + top[q].sz);
                        std::sort(busy.begin(), busy.end());
                        i64 at = 0;
                        for (auto &iv : busy) { if (iv.first - at >= b.sz) break; at = std::max(at, iv.second); }
                        b.off = at;
                    }
                    S.cbptr[b.s] = b.off;
                    toppeak = std::max(toppeak, b.off + b.sz);
                }
                Arena A;
                A.top = bottom;
                peak = std::max(peak, toppeak);
                for (i32 l = 0; l < S.nlevels; l++) {
                    for (i32 s : arrive[l]) if (!is_topblk[s]) S.cbptr[s] = A.alloc(bsz(s));
                    for (i32 s : born[l]) if (!is_topblk[s]) S.cbptr[s] = A.alloc(bsz(s));
                    peak = std::max(peak, A.top);
                    for (i32 s : consumed[l]) if (!is_topblk[s]) A.release(S.cbptr[s], bsz(s));
                }
            }
            {   // selected inversion: top-down (a distributed front is inverted by its owner alone)
                Arena A;
                std::vector<std::vector<i32>> zborn(S.nlevels), zsent(S.nlevels), zdies(S.nlevels);
                for (i32 s = 0; s < ns; s++) {
                    if (bsz(s) <= 0) continue;
                    const i32 p = S.sparent[s];
                    if (mine(s)) {
                        i32 mc = S.level[s];
                        for (i64 q = S.childptr[s]; q < S.childptr[s + 1]; q++) mc = std::min(mc, S.level[S.children[q]]);
                        zborn[S.level[s]].push_back(s);
                        zdies[mc].push_back(s);
                    } else if (p >= 0 && S.owner[p] == me) {
                        zborn[S.level[s]].push_back(s);
                        zsent[S.level[s]].push_back(s);
                    }
                }
                for (i32 l = S.nlevels - 1; l >= 0; l--) {
                    for (i32 s : zborn[l]) S.zbptr[s] = A.alloc(bsz(s));
                    peak = std::max(peak, A.top);
                    for (i32 s : zsent[l]) A.release(S.zbptr[s], bsz(s));
                    for (i32 s : zdies[l]) A.release(S.zbptr[s], bsz(s));
                }
            }
        }
        S.cb_arena = std::max<i64>(peak, 16);
    }

    // ---- contribution-block transfers of the factorisation, by column ranges ---------------------------------------------
    // column k of child d (its k-th trailing row) lands in row/column rel[k] of the parent p: a panel column (< c_p) or a column
    // of p's own contribution block. It is HELD by cb_owner(d, k / 256) and NEEDED by the owner of that parent block.
    S.xf_child.clear(); S.xf_src.clear(); S.xf_dst.clear(); S.xf_level.clear(); S.xf_col0.clear(); S.xf_off.clear(); S.xf_cnt.clear();
    if (S.shard_plan) {
        std::vector<std::array<i64, 7>> xf;
        for (i32 d = 0; d < ns; d++) {
            const i32 p = S.sparent[d];
            if (p == -1) continue;
            if (!S.is_dist(d) && !S.is_dist(p) && S.owner[d] == S.owner[p]) continue;
            const i32 cd = S.ncols(d), md = S.nrows(d) - cd, cp = S.ncols(p);
            const i32 *rel = S.rel.data() + S.rowptr[d] + cd;
            auto need = [&](i32 k) { return rel[k] < cp ? S.panel_owner(p, rel[k] / 256) : S.cb_owner(p, (rel[k] - cp) / 256); };
            i32 k = 0;
            while (k < md) {
                const i32 src = S.cb_owner(d, k / 256), dst = need(k);
                i32 k1 = k + 1;
                while (k1 < md && S.cb_owner(d, k1 / 256) == src && need(k1) == dst) k1++;
                if (src != dst)       // (offset: in THIS rank's arena; -1 when this rank is neither end)
                    xf.push_back({(i64)S.level[p], (i64)d, (i64)src, (i64)dst, (src == S.shard_rank || dst == S.shard_rank) ? S.cbptr[d] + (i64)k * md : (i64)-1,
                                  (i64)(k1 - k) * md, (i64)k});
                k = k1;
            }
        }
        std::stable_sort(xf.begin(), xf.end(), [](const std::array<i64, 7> &a, const std::array<i64, 7> &b) { return a[0] < b[0]; });
        for (auto &e : xf) {
            S.xf_level.push_back((i32)e[0]); S.xf_child.push_back((i32)e[1]); S.xf_src.push_back((i32)e[2]); S.xf_dst.push_back((i32)e[3]);
            S.xf_off.push_back(e[4]); S.xf_cnt.push_back(e[5]); S.xf_col0.push_back((i32)e[6]);
        }
    }

    // ---- Q scatter map -------------------------------------------------------------------
    // Which stored triangle defines Q: if both strict triangles are present use opt.uplo,
    // otherwise whatever is stored.
    bool has_up = false, has_lo = false;
    for (i64 j = 0; j < n && !(has_up && has_lo); j++)
        for (i64 p = colptr[j] - base; p < colptr[j + 1] - base; p++) {
            i64 i = rowval[p] - base;
            if (i < j) has_up = true; else if (i > j) has_lo = true;
        }
    int use = (has_up && has_lo) ? (opt.uplo == 0 ? 0 : 1) : (has_up ? 0 : 1);  // 0 upper, 1 lower
    S.in_use = use;
    S.in_colptr.resize(n + 1);
    for (i64 j = 0; j <= n; j++) S.in_colptr[j] = colptr[j] - base;
    S.in_row.resize(S.in_colptr[n]);
    for (i64 p = 0; p < S.in_colptr[n]; p++) S.in_row[p] = (i32)(rowval[p] - base);
    // count per destination supernode
    S.qptr.assign(ns + 1, 0);
    for (i64 j = 0; j < n; j++)
        for (i64 p = colptr[j] - base; p < colptr[j + 1] - base; p++) {
            i64 i = rowval[p] - base;
            if ((use == 0 && i > j) || (use == 1 && i < j)) continue;
            i32 a = iperm[i], b = iperm[j];
            S.qptr[S.col2super[std::min(a, b)] + 1]++;
        }
    for (i32 s = 0; s < ns; s++) S.qptr[s + 1] += S.qptr[s];
    i64 nq = S.qptr[ns];
    S.qsrc.resize(nq);
    S.qdst.resize(nq);
    {
        // position of a row inside its supernode: built per supernode into `pos`
        std::vector<i64> w(S.qptr.begin(), S.qptr.end() - 1);
        // first pass: store (src, packed (a,b)) grouped by supernode
        std::vector<i32> ra(nq), rb(nq);
        for (i64 j = 0; j < n; j++)
            for (i64 p = colptr[j] - base; p < colptr[j + 1] - base; p++) {
                i64 i = rowval[p] - base;
                if ((use == 0 && i > j) || (use == 1 && i < j)) continue;
                i32 a = iperm[i], b = iperm[j];
                if (a < b) std::swap(a, b);
                i64 q = w[S.col2super[b]]++;
                S.qsrc[q] = p;
                ra[q] = a; rb[q] = b;
            }
        // destinations and the per-supernode sort are independent across supernodes: ranges of supernodes with
        // about equal numbers of entries go to a few threads (own scratch each; destinations are unique, so the
        // sorted result does not depend on the split)
        auto work = [&](i32 s0, i32 s1) {
            std::vector<i32> pos(n, -1);
            std::vector<std::pair<i64, i64>> tmp;
            for (i32 s = s0; s < s1; s++) {
                i64 r0 = S.rowptr[s], r1 = S.rowptr[s + 1];
                for (i64 k = r0; k < r1; k++) pos[S.rows[k]] = (i32)(k - r0);
                for (i64 q = S.qptr[s]; q < S.qptr[s + 1]; q++) {
                    i32 pr = pos[ra[q]];
                    if (pr < 0) throw std::runtime_error("internal: Q entry outside supernode structure");
                    S.qdst[q] = S.panelptr[s] + (i64)(rb[q] - S.sfirst[s]) * S.ld[s] + pr;
                }
                for (i64 k = r0; k < r1; k++) pos[S.rows[k]] = -1;
                // sort the supernode's list by destination (column-major inside the panel)
                i64 a = S.qptr[s], b = S.qptr[s + 1];
                tmp.resize(b - a);
                for (i64 q = a; q < b; q++) tmp[q - a] = {S.qdst[q], S.qsrc[q]};
                std::sort(tmp.begin(), tmp.end());
                for (i64 q = a; q < b; q++) { S.qdst[q] = tmp[q - a].first; S.qsrc[q] = tmp[q - a].second; }
            }
        };
        const unsigned hw = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
        if (nq < 2000000 || hw == 1) work(0, ns);
        else {
            std::vector<i32> cut(hw + 1, ns);
            cut[0] = 0;
            for (unsigned t = 1, s = 0; t < hw; t++) {
                while ((i32)s < ns && S.qptr[s] < nq * (i64)t / hw) s++;
                cut[t] = (i32)s;
            }
            std::vector<std::thread> th;
            th.reserve(hw);     // no reallocation inside the loop: nothing can throw with joinable threads alive
            std::vector<std::exception_ptr> err(hw);
            for (unsigned t = 0; t < hw; t++) {
                try {
                    th.emplace_back([&, t] { try { work(cut[t], cut[t + 1]); } catch (...) { err[t] = std::current_exception(); } });
                } catch (const std::system_error &) {      // no thread to be had: this range on the calling thread
                    try { work(cut[t], cut[t + 1]); } catch (...) { err[t] = std::current_exception(); }
                }
            }
            for (auto &x : th) x.join();
            for (auto &e : err) if (e) std::rethrow_exception(e);
        }
    }

    S.perm.swap(perm);
    S.iperm.swap(iperm);
    S.ms_symbolic = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace gmrfx
