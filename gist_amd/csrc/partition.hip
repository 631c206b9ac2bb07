// Host-side k-way graph partitioner: the library's replacement for dgl.transform.metis_partition
// (reference cluster_gcn/partition_utils.py:11-18; METIS itself is a third-party library that
// is not available offline).  One-time data preparation, cached by ClusterIter in the
// reference's .npy format -- not part of the hot path, plain C++ on the host (METIS is a host
// library too).
//
// Algorithm (round 4): MULTILEVEL, in the family METIS belongs to, with the coarsening that suits graphs
// whose parts are small (Cluster-GCN asks for 1 500 parts of ~100 nodes on Reddit, 15 000 on Amazon2M):
//   coarsen   size-constrained label propagation, twice: every node joins the neighbouring cluster it has the
//             most edge weight to as long as the cluster stays within a THIRD of a part's capacity; the clusters
//             are contracted to weighted vertices (parallel edges merged, weights added); on that graph the same
//             propagation with the bound at a part's full capacity lets the fragments of one dense region find
//             each other (they have far more weight between them than to anything else);
//   initial   on the coarsest graph: weighted linear-deterministic-greedy, heaviest vertices first (a vertex
//             goes to the part it has most weight to, discounted by how full that part is; a vertex nobody
//             claims opens the emptiest part), then refinement as below;
//   uncoarsen level by level: project the parts to the finer graph and refine -- every vertex moves to the
//             part it has the most edge weight to if that is more than it has to its own part and the target
//             has room; first with a sixteenth of slack above capacity (a vertex may enter a full part, the
//             next pass sheds that part's loosest vertex: the effect of a swap), then strictly within capacity;
//   balance   finest level: parts below floor((1 - imbalance) * n / k) pull their best-connected outside
//             nodes from parts that can spare them; no part is empty.
// Round 3's single-level restreaming LDG (the `initial` + refine steps on the input graph alone) reached an
// edge cut of 0.58 on the Reddit-like graph whose planted parts cut 0.42; the multilevel form reaches the
// planted cut on the Reddit-like (153 k nodes / 1 500 parts, 5 s) and Amazon-like (1.71 M / 15 000, 27 s) graphs
// (profiles/r04_partitioner.json).  Label propagation is the coarsening for social / co-purchase graphs; on
// meshes it is about 2x off the ideal cut (tests/test_partitioner.py), where METIS's matching would do better.
// Neighbours are the union of in- and out-edges (multi-edges count with multiplicity); part
// sizes end within [floor((1 - imbalance) n/k), ceil((1 + imbalance) n/k)] whenever the graph
// allows it.  Deterministic for a given seed.
#include <algorithm>
#include <cstdint>
#include <vector>

#include "common.h"

namespace {

struct Rng {      // splitmix64: visiting orders depend on the seed only
    uint64_t s;
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
};

// undirected weighted graph; ew empty = every edge weighs 1 (the input level keeps its multi-edges as
// separate entries, which the accumulators below add up like a weight)
struct WGraph {
    int64_t n = 0;
    std::vector<int64_t> xadj;
    std::vector<int32_t> adj, ew, vw;
    int32_t w(int64_t e) const { return ew.empty() ? 1 : ew[e]; }
};

WGraph input_level(const int32_t *rowptr, const int32_t *col, const int32_t *t_rowptr, const int32_t *t_col, int64_t n) {
    WGraph g;
    g.n = n;
    g.xadj.assign(n + 1, 0);
    g.vw.assign(n, 1);
    for (int64_t v = 0; v < n; ++v) {
        int64_t d = 0;
        for (int32_t e = rowptr[v]; e < rowptr[v + 1]; ++e) d += col[e] != v;
        if (t_rowptr)
            for (int32_t e = t_rowptr[v]; e < t_rowptr[v + 1]; ++e) d += t_col[e] != v;
        g.xadj[v + 1] = g.xadj[v] + d;
    }
    g.adj.resize(g.xadj[n]);
    for (int64_t v = 0; v < n; ++v) {
        int64_t w = g.xadj[v];
        for (int32_t e = rowptr[v]; e < rowptr[v + 1]; ++e)
            if (col[e] != v) g.adj[w++] = col[e];
        if (t_rowptr)
            for (int32_t e = t_rowptr[v]; e < t_rowptr[v + 1]; ++e)
                if (t_col[e] != v) g.adj[w++] = t_col[e];
    }
    return g;
}

std::vector<int32_t> random_order(int64_t n, Rng &rng) {
    std::vector<int32_t> perm(n);
    for (int64_t i = 0; i < n; ++i) perm[i] = (int32_t)i;
    for (int64_t i = n - 1; i > 0; --i) std::swap(perm[i], perm[rng.next() % (uint64_t)(i + 1)]);
    return perm;
}

// BFS order restarted from a seeded permutation (neighbouring vertices are visited close together)
std::vector<int32_t> bfs_order(const WGraph &g, Rng &rng) {
    std::vector<int32_t> order, perm = random_order(g.n, rng);
    order.reserve(g.n);
    std::vector<uint8_t> seen(g.n, 0);
    for (int64_t s = 0; s < g.n; ++s) {
        if (seen[perm[s]]) continue;
        size_t head = order.size();
        order.push_back(perm[s]);
        seen[perm[s]] = 1;
        while (head < order.size()) {
            const int32_t v = order[head++];
            for (int64_t e = g.xadj[v]; e < g.xadj[v + 1]; ++e) {
                const int32_t u = g.adj[e];
                if (!seen[u]) { seen[u] = 1; order.push_back(u); }
            }
        }
    }
    return order;
}

// size-constrained label propagation: label[v] in [0, n_clusters), every cluster's weight <= bound
int64_t cluster_lp(const WGraph &g, int64_t bound, int iters, Rng &rng, std::vector<int32_t> &label) {
    const int64_t n = g.n;
    label.resize(n);
    std::vector<int64_t> cw(n);
    for (int64_t v = 0; v < n; ++v) { label[v] = (int32_t)v; cw[v] = g.vw[v]; }
    std::vector<int64_t> conn(n, 0);
    std::vector<int32_t> touched;
    const std::vector<int32_t> order = random_order(n, rng);
    for (int it = 0; it < iters; ++it) {
        int64_t moved = 0;
        for (int64_t i = 0; i < n; ++i) {
            const int32_t v = order[i];
            if (g.xadj[v] == g.xadj[v + 1]) continue;
            touched.clear();
            for (int64_t e = g.xadj[v]; e < g.xadj[v + 1]; ++e) {
                const int32_t l = label[g.adj[e]];
                if (conn[l] == 0) touched.push_back(l);
                conn[l] += g.w(e);
            }
            const int32_t cur = label[v];
            int32_t best = cur;
            int64_t best_c = conn[cur];           // staying wins ties: the propagation settles
            for (int32_t l : touched) {
                if (l != cur && cw[l] + g.vw[v] <= bound &&
                    (conn[l] > best_c || (conn[l] == best_c && best != cur && cw[l] < cw[best]))) {
                    best = l;
                    best_c = conn[l];
                }
            }
            for (int32_t l : touched) conn[l] = 0;
            if (best != cur) {
                cw[cur] -= g.vw[v];
                cw[best] += g.vw[v];
                label[v] = best;
                ++moved;
            }
        }
        if (moved * 100 < n) break;
    }
    // dense cluster ids in order of first appearance (deterministic)
    std::vector<int32_t> remap(n, -1);
    int64_t nc = 0;
    for (int64_t v = 0; v < n; ++v) {
        if (remap[label[v]] < 0) remap[label[v]] = (int32_t)nc++;
        label[v] = remap[label[v]];
    }
    return nc;
}

WGraph contract(const WGraph &g, const std::vector<int32_t> &label, int64_t nc) {
    WGraph c;
    c.n = nc;
    c.vw.assign(nc, 0);
    std::vector<int64_t> start(nc + 1, 0);
    for (int64_t v = 0; v < g.n; ++v) { c.vw[label[v]] += g.vw[v]; ++start[label[v] + 1]; }
    for (int64_t q = 0; q < nc; ++q) start[q + 1] += start[q];
    std::vector<int32_t> members(g.n);
    {
        std::vector<int64_t> fill(start.begin(), start.end() - 1);
        for (int64_t v = 0; v < g.n; ++v) members[fill[label[v]]++] = (int32_t)v;
    }
    c.xadj.assign(nc + 1, 0);
    std::vector<int64_t> acc(nc, 0);
    std::vector<int32_t> touched;
    for (int64_t q = 0; q < nc; ++q) {
        touched.clear();
        for (int64_t i = start[q]; i < start[q + 1]; ++i) {
            const int32_t v = members[i];
            for (int64_t e = g.xadj[v]; e < g.xadj[v + 1]; ++e) {
                const int32_t l = label[g.adj[e]];
                if (l == q) continue;
                if (acc[l] == 0) touched.push_back(l);
                acc[l] += g.w(e);
            }
        }
        std::sort(touched.begin(), touched.end());
        for (int32_t l : touched) {
            c.adj.push_back(l);
            c.ew.push_back((int32_t)std::min<int64_t>(acc[l], INT32_MAX));
            acc[l] = 0;
        }
        c.xadj[q + 1] = (int64_t)c.adj.size();
    }
    if (c.ew.empty()) c.ew.push_back(0);          // (a coarse level is always "weighted": see WGraph::w)
    return c;
}

struct KWay {
    const WGraph &g;
    int32_t k;
    int64_t cap;
    std::vector<int32_t> &part;
    std::vector<int64_t> size, conn;
    std::vector<int32_t> touched;
    KWay(const WGraph &g_, int32_t k_, int64_t cap_, std::vector<int32_t> &part_)
        : g(g_), k(k_), cap(cap_), part(part_), size(k_, 0), conn(k_, 0) {}

    void gather(int32_t v) {
        touched.clear();
        for (int64_t e = g.xadj[v]; e < g.xadj[v + 1]; ++e) {
            const int32_t p = part[g.adj[e]];
            if (p < 0) continue;
            if (conn[p] == 0) touched.push_back(p);
            conn[p] += g.w(e);
        }
    }
    void release() { for (int32_t p : touched) conn[p] = 0; }

    // initial assignment: most weight to already placed neighbours, discounted by how full the part is
    int32_t greedy_part(int32_t v) {
        gather(v);
        int32_t best = -1;
        double best_score = -1.0;
        for (int32_t p : touched) {
            if (size[p] + g.vw[v] > cap) continue;
            const double sc = (double)conn[p] * (1.0 - (double)size[p] / (double)cap);
            if (sc > best_score || (sc == best_score && (size[p] < size[best] || (size[p] == size[best] && p < best)))) {
                best_score = sc;
                best = p;
            }
        }
        release();
        return best;
    }
    int32_t emptiest(int32_t start, int64_t w) const {
        int32_t best = -1;
        for (int32_t i = 0; i < k; ++i) {
            const int32_t p = (start + i) % k;
            if (size[p] + w <= cap && (best < 0 || size[p] < size[best])) {
                best = p;
                if (size[p] == 0) break;
            }
        }
        if (best < 0) {                            // nothing fits: the lightest part takes it (the finest level repairs)
            best = 0;
            for (int32_t p = 1; p < k; ++p)
                if (size[p] < size[best]) best = p;
        }
        return best;
    }
    void initial(const std::vector<int32_t> &order) {
        for (int64_t v = 0; v < g.n; ++v) part[v] = -1;
        int32_t open_part = 0;
        for (int64_t i = 0; i < g.n; ++i) {
            const int32_t v = order[i];
            int32_t p = greedy_part(v);
            if (p < 0) {
                p = emptiest(open_part, g.vw[v]);
                open_part = (p + 1) % k;
            }
            part[v] = p;
            size[p] += g.vw[v];
        }
    }
    void recount() {
        std::fill(size.begin(), size.end(), 0);
        for (int64_t v = 0; v < g.n; ++v) size[part[v]] += g.vw[v];
    }
    // refinement: strict gain in edge weight, target within capacity; a vertex of an over-full part may also
    // leave at equal weight
    int64_t refine(const std::vector<int32_t> &order, int passes, int64_t limit) {
        int64_t total = 0;
        for (int t = 0; t < passes; ++t) {
            int64_t moved = 0;
            for (int64_t i = 0; i < g.n; ++i) {
                const int32_t v = order[i];
                const int32_t old = part[v];
                gather(v);
                const bool over = size[old] > limit;
                int32_t best = old;
                int64_t best_c = conn[old] - (over ? 1 : 0);
                for (int32_t p : touched) {
                    if (p == old || size[p] + g.vw[v] > limit) continue;
                    if (conn[p] > best_c || (conn[p] == best_c && best != old && size[p] < size[best])) {
                        best = p;
                        best_c = conn[p];
                    }
                }
                release();
                if (best != old) {
                    size[old] -= g.vw[v];
                    size[best] += g.vw[v];
                    part[v] = best;
                    ++moved;
                }
            }
            total += moved;
            if (moved == 0) break;
        }
        return total;
    }
};

}  // namespace

extern "C" int gist_partition_graph(const int32_t *rowptr, const int32_t *col,
                                    const int32_t *t_rowptr, const int32_t *t_col, int64_t n,
                                    int32_t k, uint64_t seed, int32_t n_passes, float imbalance,
                                    int32_t *part_out) {
    GIST_REQUIRE(n > 0 && k > 0 && k <= n, "gist_partition_graph: need 0 < k <= n");
    GIST_REQUIRE(rowptr && col && part_out, "gist_partition_graph: null pointer");
    GIST_REQUIRE(n < (1LL << 31), "gist_partition_graph: n >= 2^31");
    GIST_REQUIRE(imbalance >= 0.f && n_passes >= 0, "gist_partition_graph: bad parameters");
    const int64_t target = (n + k - 1) / k;
    int64_t cap = (int64_t)((1.0 + (double)imbalance) * (double)n / (double)k + 0.999999);
    if (cap < target) cap = target;
    Rng rng{seed};

    // ---- coarsening: cluster weight bounds cap/3, 2 cap/3, cap --------------------------------------
    // (three rounds of merging: fragments of one dense region have far more weight between them than to
    // anything else, so they find each other; the last round's clusters are whole parts or pieces of one)
    std::vector<WGraph> levels;
    std::vector<std::vector<int32_t>> labels;          // labels[l][v] = vertex of level l+1 that v of level l joins
    levels.push_back(input_level(rowptr, col, t_rowptr, t_col, n));
    for (int lvl = 0; lvl < 2 && cap >= 6; ++lvl) {
        const WGraph &g = levels.back();
        if (g.n <= (int64_t)k || g.xadj[g.n] == 0) break;
        const int64_t bound = lvl == 1 ? cap : std::max<int64_t>(2, cap / 3);
        std::vector<int32_t> label;
        const int64_t nc = cluster_lp(g, bound, 8, rng, label);
        if (nc < (int64_t)k || nc == g.n) break;       // too coarse for k parts / nothing merged
        WGraph c = contract(g, label, nc);
        labels.push_back(std::move(label));
        levels.push_back(std::move(c));
    }

    // ---- initial partition of the coarsest level, then refine while uncoarsening --------------------
    std::vector<int32_t> part(levels.back().n, -1);
    for (int lvl = (int)levels.size() - 1; lvl >= 0; --lvl) {
        const WGraph &g = levels[lvl];
        std::vector<int32_t> order = bfs_order(g, rng);
        if (lvl == (int)levels.size() - 1) {
            // heaviest vertices first (stable within equal weight: BFS order): on a coarsened graph the k
            // heaviest clusters seed the parts and the fragments join the part they are tied to
            if (lvl > 0)
                std::stable_sort(order.begin(), order.end(), [&](int32_t a_, int32_t b_) { return g.vw[a_] > g.vw[b_]; });
            KWay kw(g, k, cap, part);
            kw.initial(order);
            kw.refine(order, std::max(n_passes, 1) * 2, cap);
        } else {
            std::vector<int32_t> fine(g.n);
            for (int64_t v = 0; v < g.n; ++v) fine[v] = part[labels[lvl][v]];
            part.swap(fine);
            KWay kw(g, k, cap, part);
            kw.recount();
            // first with a little slack above capacity (a vertex may enter a full part; the pass after sheds that
            // part's loosest vertex: the effect of a swap), then strictly within capacity
            kw.refine(order, std::max(n_passes, 1), cap + std::max<int64_t>(1, cap / 16));
            kw.refine(order, std::max(n_passes, 1), cap);
        }
    }
    for (int64_t v = 0; v < n; ++v) part_out[v] = part[v];
    int32_t *P = part_out;
    std::vector<int64_t> size(k, 0);
    for (int64_t v = 0; v < n; ++v) ++size[P[v]];
    const WGraph &g0 = levels[0];

    // ---- upper bound: an over-full part sheds the nodes with the least weight into it -----------------
    {
        std::vector<int64_t> conn(k, 0);
        std::vector<int32_t> touched;
        for (int64_t v = n - 1; v >= 0; --v) {
            const int32_t old = P[v];
            if (size[old] <= cap) continue;
            touched.clear();
            for (int64_t e = g0.xadj[v]; e < g0.xadj[v + 1]; ++e) {
                const int32_t p = P[g0.adj[e]];
                if (conn[p] == 0) touched.push_back(p);
                ++conn[p];
            }
            int32_t best = -1;
            for (int32_t p : touched)
                if (p != old && size[p] < cap && (best < 0 || conn[p] > conn[best])) best = p;
            for (int32_t p : touched) conn[p] = 0;
            if (best < 0)
                for (int32_t p = 0; p < k; ++p)
                    if (size[p] < cap && (best < 0 || size[p] < size[best])) best = p;
            if (best < 0) break;
            P[v] = best;
            --size[old];
            ++size[best];
        }
    }
    // ---- lower balance bound: a part below lo pulls, one at a time, the outside node with ----
    // ---- most edges into it from a part that can spare one (else any node of the largest) ----
    const int64_t lo = (int64_t)((1.0 - (double)imbalance) * (double)n / (double)k);
    if (lo > 0) {
        std::vector<std::vector<int32_t>> members(k);
        for (int64_t v = 0; v < n; ++v) members[P[v]].push_back((int32_t)v);
        std::vector<int32_t> gain(n, 0), cand;
        for (int32_t p = 0; p < k; ++p) {
            while ((int64_t)members[p].size() < lo) {
                cand.clear();
                for (int32_t v : members[p])
                    for (int64_t e = g0.xadj[v]; e < g0.xadj[v + 1]; ++e) {
                        const int32_t u = g0.adj[e];
                        const int32_t q = P[u];
                        if (q == p || (int64_t)members[q].size() <= lo) continue;
                        if (gain[u]++ == 0) cand.push_back(u);
                    }
                int32_t pick = -1;
                for (int32_t u : cand)
                    if (pick < 0 || gain[u] > gain[pick] || (gain[u] == gain[pick] && u < pick)) pick = u;
                for (int32_t u : cand) gain[u] = 0;
                if (pick < 0) {                      // no outside neighbour to spare: largest part
                    int32_t big = -1;
                    for (int32_t q = 0; q < k; ++q)
                        if (q != p && (big < 0 || members[q].size() > members[big].size())) big = q;
                    if (big < 0 || (int64_t)members[big].size() <= lo) break;
                    pick = members[big].back();
                }
                const int32_t q = P[pick];
                auto &mq = members[q];
                mq.erase(std::find(mq.begin(), mq.end(), pick));
                members[p].push_back(pick);
                P[pick] = p;
            }
        }
        for (int32_t p = 0; p < k; ++p) size[p] = (int64_t)members[p].size();
    }
    // ---- no empty parts -------------------------------------------------------------------
    for (int32_t p = 0; p < k; ++p) {
        if (size[p] > 0) continue;
        int32_t big = 0;
        for (int32_t q = 1; q < k; ++q)
            if (size[q] > size[big]) big = q;
        for (int64_t v = n - 1; v >= 0; --v)
            if (P[v] == big) { P[v] = p; --size[big]; ++size[p]; break; }
    }
    return GIST_OK;
}
