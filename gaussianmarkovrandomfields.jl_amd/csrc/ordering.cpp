// ordering.cpp -- fill-reducing nested-dissection ordering (host).
//
// The reference delegates ordering to CHOLMOD's AMD or to a user-supplied permutation /
// CliqueTrees algorithm (src/workspace/backend.jl:73-153). This backend computes its own
// nested dissection (the ordering that exposes the separator-tree parallelism the GPU
// schedule needs) and reports it through gmrfx_get_perm so CHOLMOD can factor the identical
// P Q P'. Two bisectors:
//   * geometric (mesh node coordinates supplied): median cut along the longest axis;
//   * graph-based (no coordinates): BFS level structure from a pseudo-peripheral node.
// In both, the vertex separator is the smaller one-sided boundary of the cut, thinned so that
// every separator vertex touches both sides. Leaves (<= nd_leaf vertices) are ordered by a
// halo-aware minimum-degree on bit masks.
#include <algorithm>
#include <atomic>
#include <bitset>
#include <cstdlib>
#include <exception>
#include <system_error>
#include <thread>
#include <cmath>
#include <numeric>
#include <stdexcept>

#include "symbolic.h"

namespace gmrfx {

void build_graph(i64 n, const i64 *colptr, const i64 *rowval, int base, Graph &G) {
    G.n = n;
    std::vector<i64> deg(n + 1, 0);
    i64 nnz = colptr[n] - base;
    for (i64 j = 0; j < n; j++)
        for (i64 p = colptr[j] - base; p < colptr[j + 1] - base; p++) {
            i64 i = rowval[p] - base;
            if (i < 0 || i >= n) throw std::invalid_argument("rowval out of range");
            if (i != j) { deg[i + 1]++; deg[j + 1]++; }
        }
    (void)nnz;
    std::vector<i64> ptr(n + 1, 0);
    for (i64 i = 0; i < n; i++) ptr[i + 1] = ptr[i] + deg[i + 1];
    std::vector<i32> tmp(ptr[n]);
    std::vector<i64> w(ptr.begin(), ptr.end() - 1);
    for (i64 j = 0; j < n; j++)
        for (i64 p = colptr[j] - base; p < colptr[j + 1] - base; p++) {
            i64 i = rowval[p] - base;
            if (i != j) { tmp[w[i]++] = (i32)j; tmp[w[j]++] = (i32)i; }
        }
    // sort + unique each list (both triangles stored => every edge appears twice)
    G.xadj.assign(n + 1, 0);
    i64 out = 0;
    for (i64 i = 0; i < n; i++) {
        std::sort(tmp.begin() + ptr[i], tmp.begin() + ptr[i + 1]);
        i64 start = out;
        for (i64 p = ptr[i]; p < ptr[i + 1]; p++)
            if (out == start || tmp[out - 1] != tmp[p]) tmp[out++] = tmp[p];
        G.xadj[i + 1] = out;
    }
    tmp.resize(out);
    tmp.shrink_to_fit();
    G.adj.swap(tmp);
}

namespace {

struct ND {
    const Graph &G;
    const double *xy;
    int dim;
    int leaf;
    std::vector<i32> label;   // region id of each vertex; -1 once ordered/separator
    std::vector<i32> &perm;
    std::atomic<i32> next_id{1};   // region ids only have to be distinct: the result does not depend on their values
    std::vector<i32> dist;         // per-vertex BFS scratch (disjoint vertex sets per task)
    // BFS visit order: one per thread (the two halves of a dissection run as separate tasks near the top)
    static std::vector<i32> &tls_queue() { static thread_local std::vector<i32> q; return q; }
#define queue tls_queue()

    ND(const Graph &g, const SymOptions &o, std::vector<i32> &p)
        : G(g), xy(o.coords), dim(o.coords ? o.coord_dim : 0), leaf(o.nd_leaf > 0 ? o.nd_leaf : 64),
          label(g.n, 0), perm(p) {
        if (leaf > 64) leaf = 64;
        dist.assign(g.n, -1);
        lidx.assign(g.n, -1);
    }

    // ---- leaf ordering: minimum degree with a fixed halo --------------------------------
    void order_leaf(i32 *v, i64 cnt, i64 pos) {
        if (cnt <= 2) { for (i64 k = 0; k < cnt; k++) { perm[pos + k] = v[k]; label[v[k]] = -1; } return; }
        constexpr int W = 192;
        using Mask = std::bitset<W>;
        std::vector<Mask> adj(cnt);
        // local numbering: leaf vertices 0..cnt-1, halo cnt..W-1
        std::vector<i32> halo;
        i32 myid = label[v[0]];
        for (i64 k = 0; k < cnt; k++) dist[v[k]] = (i32)k;  // borrow dist[] as local index
        for (i64 k = 0; k < cnt; k++)
            for (i64 p = G.xadj[v[k]]; p < G.xadj[v[k] + 1]; p++) {
                i32 w = G.adj[p];
                if (label[w] == myid) { adj[k].set(dist[w]); continue; }
                // already-ordered separator vertex of an ancestor: part of the halo
                auto it = std::find(halo.begin(), halo.end(), w);
                int h;
                if (it == halo.end()) {
                    if ((int)(cnt + halo.size()) >= W) continue;
                    halo.push_back(w);
                    h = (int)(cnt + halo.size() - 1);
                } else h = (int)(cnt + (it - halo.begin()));
                adj[k].set(h);
            }
        Mask alive;
        for (i64 k = 0; k < cnt; k++) alive.set(k);
        for (i64 step = 0; step < cnt; step++) {
            int best = -1; size_t bestdeg = ~size_t(0);
            for (i64 k = 0; k < cnt; k++)
                if (alive[k]) {
                    size_t d = adj[k].count();
                    if (d < bestdeg) { bestdeg = d; best = (int)k; }
                }
            alive.reset(best);
            Mask nb = adj[best];
            for (i64 k = 0; k < cnt; k++)
                if (alive[k] && nb[k]) { adj[k] |= nb; adj[k].reset(k); adj[k].reset(best); }
            perm[pos + step] = v[best];
        }
        for (i64 k = 0; k < cnt; k++) { dist[v[k]] = -1; label[v[k]] = -1; }
    }

    // Given v[0:cnt) labelled idA / idB: the vertex separator is a MINIMUM VERTEX COVER of the
    // bipartite graph formed by the cut edges (Hopcroft-Karp maximum matching + Koenig's
    // construction) -- the smallest separator this edge cut admits; for the 2-ring stencils of
    // alpha=2 SPDE precisions on jittered meshes it is ~1/3 smaller than a one-sided boundary.
    // Partitions v into [A' | B' | S]. Returns sizes.
    std::vector<i32> lidx;   // scratch: local index of a boundary vertex (size n, -1 = none)
    void split(i32 *v, i64 cnt, i32 idA, i32 idB, i64 &nA, i64 &nB, i64 &nS) {
        std::vector<i32> Lv, Rv;   // boundary vertices of A (left) and B (right)
        for (i64 k = 0; k < cnt; k++) {
            i32 u = v[k]; i32 other = label[u] == idA ? idB : idA;
            for (i64 p = G.xadj[u]; p < G.xadj[u + 1]; p++)
                if (label[G.adj[p]] == other) {
                    if (label[u] == idA) { lidx[u] = (i32)Lv.size(); Lv.push_back(u); }
                    else { lidx[u] = (i32)Rv.size(); Rv.push_back(u); }
                    break;
                }
        }
        const i32 nl = (i32)Lv.size(), nrr = (i32)Rv.size();
        // adjacency L -> R (local indices)
        std::vector<i64> lp(nl + 1, 0);
        for (i32 a = 0; a < nl; a++) {
            i64 c = 0;
            for (i64 p = G.xadj[Lv[a]]; p < G.xadj[Lv[a] + 1]; p++) if (label[G.adj[p]] == idB) c++;
            lp[a + 1] = lp[a] + c;
        }
        std::vector<i32> la(lp[nl]);
        for (i32 a = 0; a < nl; a++) {
            i64 q = lp[a];
            for (i64 p = G.xadj[Lv[a]]; p < G.xadj[Lv[a] + 1]; p++) if (label[G.adj[p]] == idB) la[q++] = lidx[G.adj[p]];
        }
        // Hopcroft-Karp
        std::vector<i32> ml(nl, -1), mr(nrr, -1), dist(nl), stack, it(nl);
        const i32 INF = 0x7fffffff;
        for (;;) {
            // BFS from free left vertices
            std::vector<i32> q;
            for (i32 a = 0; a < nl; a++) { if (ml[a] < 0) { dist[a] = 0; q.push_back(a); } else dist[a] = INF; }
            bool found = false;
            for (size_t h = 0; h < q.size(); h++) {
                i32 a = q[h];
                for (i64 p = lp[a]; p < lp[a + 1]; p++) {
                    i32 b2 = mr[la[p]];
                    if (b2 < 0) found = true;
                    else if (dist[b2] == INF) { dist[b2] = dist[a] + 1; q.push_back(b2); }
                }
            }
            if (!found) break;
            // DFS (iterative) along layered graph
            for (i32 a = 0; a < nl; a++) it[a] = 0;
            for (i32 root = 0; root < nl; root++) {
                if (ml[root] >= 0) continue;
                stack.clear();
                stack.push_back(root);
                while (!stack.empty()) {
                    i32 a = stack.back();
                    if (it[a] >= lp[a + 1] - lp[a]) { dist[a] = INF; stack.pop_back(); continue; }
                    i32 r = la[lp[a] + it[a]++];
                    i32 b2 = mr[r];
                    if (b2 < 0) {
                        // augment along the stack: each stack vertex takes the right vertex it just tried
                        for (i32 t = (i32)stack.size() - 1; t >= 0; t--) {
                            i32 x = stack[t];
                            i32 rr = la[lp[x] + it[x] - 1];
                            mr[rr] = x; ml[x] = rr;
                        }
                        stack.clear();
                    } else if (dist[b2] == dist[a] + 1) {
                        stack.push_back(b2);
                    }
                }
            }
        }
        // Koenig: Z = reachable from free left vertices by alternating paths
        std::vector<uint8_t> zl(nl, 0), zr(nrr, 0);
        {
            std::vector<i32> q;
            for (i32 a = 0; a < nl; a++) if (ml[a] < 0) { zl[a] = 1; q.push_back(a); }
            for (size_t h = 0; h < q.size(); h++) {
                i32 a = q[h];
                for (i64 p = lp[a]; p < lp[a + 1]; p++) {
                    i32 r = la[p];
                    if (zr[r]) continue;
                    zr[r] = 1;
                    i32 b2 = mr[r];
                    if (b2 >= 0 && !zl[b2]) { zl[b2] = 1; q.push_back(b2); }
                }
            }
        }
        const i32 idS = next_id++;
        for (i32 a = 0; a < nl; a++) { if (!zl[a]) label[Lv[a]] = idS; lidx[Lv[a]] = -1; }
        for (i32 r = 0; r < nrr; r++) { if (zr[r]) label[Rv[r]] = idS; lidx[Rv[r]] = -1; }
        // stable 3-way partition
        std::vector<i32> tmp(v, v + cnt);
        nA = nB = nS = 0;
        for (i64 k = 0; k < cnt; k++) if (label[tmp[k]] == idA) nA++; else if (label[tmp[k]] == idB) nB++; else nS++;
        i64 a = 0, b = nA, sidx = nA + nB;
        for (i64 k = 0; k < cnt; k++) {
            i32 u = tmp[k];
            if (label[u] == idA) v[a++] = u; else if (label[u] == idB) v[b++] = u; else v[sidx++] = u;
        }
    }

    bool bisect_geometric(i32 *v, i64 cnt, i32 idA, i32 idB) {
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (i64 k = 0; k < cnt; k++)
            for (int d = 0; d < dim; d++) {
                double c = xy[(i64)v[k] * dim + d];
                lo[d] = std::min(lo[d], c); hi[d] = std::max(hi[d], c);
            }
        int ax = 0;
        for (int d = 1; d < dim; d++) if (hi[d] - lo[d] > hi[ax] - lo[ax]) ax = d;
        if (!(hi[ax] > lo[ax])) return false;
        // Sort along the axis, then slide the cut position inside the middle 20 % of the
        // vertices and keep the position with the FEWEST CUT EDGES: on (jittered) structured
        // meshes that snaps the cut into the gap between two grid lines, on unstructured meshes
        // it picks the locally smoothest interface. The separator is then the minimum vertex
        // cover of exactly that edge cut (split()).
        auto less = [&](i32 a, i32 b) {
            double ca = xy[(i64)a * dim + ax], cb = xy[(i64)b * dim + ax];
            return ca < cb || (ca == cb && a < b);
        };
        {   // only the sliding window needs to be sorted
            const i64 wlo = std::max<i64>(1, (i64)(0.4 * cnt)), whi = std::min<i64>(cnt - 1, (i64)(0.6 * cnt) + 1);
            std::nth_element(v, v + wlo, v + cnt, less);
            std::nth_element(v + wlo, v + whi, v + cnt, less);
            std::sort(v + wlo, v + whi, less);
        }
        for (i64 k = 0; k < cnt; k++) label[v[k]] = idB;
        const i64 klo = std::max<i64>(1, (i64)(0.4 * cnt)), khi = std::min<i64>(cnt - 1, (i64)(0.6 * cnt) + 1);
        i64 cut = 0, best_cut = -1, best_k = cnt / 2;
        for (i64 k = 0; k < khi; k++) {
            const i32 u = v[k];
            for (i64 p = G.xadj[u]; p < G.xadj[u + 1]; p++) {
                const i32 lw = label[G.adj[p]];
                if (lw == idB) cut++; else if (lw == idA) cut--;
            }
            label[u] = idA;
            const i64 na = k + 1;
            if (na >= klo) {
                const i64 dmid = std::llabs(na - cnt / 2), bmid = std::llabs(best_k - cnt / 2);
                if (best_cut < 0 || cut < best_cut || (cut == best_cut && dmid < bmid)) { best_cut = cut; best_k = na; }
            }
        }
        for (i64 k = best_k; k < khi; k++) label[v[k]] = idB;
        return true;
    }

    // BFS inside region `id` from root; fills queue (visit order) and dist. Returns #visited.
    i64 bfs(i32 root, i32 id) {
        queue.clear();
        queue.push_back(root);
        dist[root] = 0;
        for (size_t h = 0; h < queue.size(); h++) {
            i32 u = queue[h];
            for (i64 p = G.xadj[u]; p < G.xadj[u + 1]; p++) {
                i32 w = G.adj[p];
                if (label[w] == id && dist[w] < 0) { dist[w] = dist[u] + 1; queue.push_back(w); }
            }
        }
        return (i64)queue.size();
    }
    void clear_bfs() { for (i32 u : queue) dist[u] = -1; }

    // Graph bisection: returns false if the region is disconnected-and-handled (A = one
    // component, B = rest) -- still a valid labelling with an empty separator.
    void bisect_graph(i32 *v, i64 cnt, i32 id, i32 idA, i32 idB) {
        i32 root = v[0];
        i64 vis = bfs(root, id);
        if (vis < cnt) {  // disconnected: component vs rest
            for (i32 u : queue) label[u] = idA;
            clear_bfs();
            for (i64 k = 0; k < cnt; k++) if (label[v[k]] == id) label[v[k]] = idB;
            return;
        }
        // pseudo-peripheral root: repeat BFS from the farthest, lowest-degree vertex
        for (int it = 0; it < 3; it++) {
            i32 far = queue.back();
            i32 ecc = dist[far];
            clear_bfs();
            bfs(far, id);
            root = far;
            if (dist[queue.back()] <= ecc) break;
        }
        // queue is in BFS order from root: first half (by count, cut at a level boundary) is A
        i64 half = cnt / 2;
        i32 cutlevel = dist[queue[half]];
        if (cutlevel == 0) cutlevel = 1;
        for (i32 u : queue) label[u] = dist[u] < cutlevel ? idA : idB;
        clear_bfs();
    }

    void run(i32 *v, i64 cnt, i64 pos, int depth = 0) {
        if (cnt <= leaf) { order_leaf(v, cnt, pos); return; }
        i32 id = label[v[0]];
        i32 idA = next_id++, idB = next_id++;
        bool ok = false;
        if (dim > 0) ok = bisect_geometric(v, cnt, idA, idB);
        if (!ok) bisect_graph(v, cnt, id, idA, idB);
        i64 nA, nB, nS;
        split(v, cnt, idA, idB, nA, nB, nS);
        if (nA == 0 || nB == 0) {
            // degenerate (e.g. clique): no useful separator -> order as one block
            for (i64 k = 0; k < cnt; k++) { perm[pos + k] = v[k]; label[v[k]] = -1; }
            return;
        }
        for (i64 k = 0; k < nS; k++) { perm[pos + nA + nB + k] = v[nA + nB + k]; label[v[nA + nB + k]] = -1; }
        // The two halves touch disjoint vertices (their only common neighbours are separator vertices, whose
        // labels are final): near the top of the tree they run as two tasks. The ordering does not depend on
        // which thread runs what -- every rank of a sharded factorisation must get the same permutation.
        if (depth < par_depth && std::min(nA, nB) > 20000) {
            std::exception_ptr err;
            std::thread t;
            try {
                t = std::thread([&] { try { run(v, nA, pos, depth + 1); } catch (...) { err = std::current_exception(); } });
            } catch (const std::system_error &) {      // no thread to be had (process limits): same work, serially
                run(v, nA, pos, depth + 1);
            }
            try { run(v + nA, nB, pos + nA, depth + 1); } catch (...) { if (t.joinable()) t.join(); throw; }
            if (t.joinable()) t.join();
            if (err) std::rethrow_exception(err);
        } else {
            run(v, nA, pos, depth + 1);
            run(v + nA, nB, pos + nA, depth + 1);
        }
    }
    int par_depth = 3;   // up to 8 concurrent tasks
#undef queue
};

}  // namespace

void nested_dissection(const Graph &G, const SymOptions &opt, std::vector<i32> &perm) {
    i64 n = G.n;
    perm.assign(n, -1);
    std::vector<i32> verts(n);
    std::iota(verts.begin(), verts.end(), 0);
    ND nd(G, opt, perm);
    if (const char *e = std::getenv("GMRFX_ND_THREADS")) nd.par_depth = std::atoi(e) <= 1 ? 0 : (std::atoi(e) <= 2 ? 1 : (std::atoi(e) <= 4 ? 2 : 3));   // 1 = serial (testing knob)
    else if (std::thread::hardware_concurrency() <= 1) nd.par_depth = 0;
    nd.run(verts.data(), n, 0);
}

}  // namespace gmrfx
