// Host-side k-way graph partitioner: the library's replacement for dgl.transform.metis_partition
// (reference cluster_gcn/partition_utils.py:11-18; METIS itself is a third-party library that
// is not available offline).  One-time data preparation, cached by ClusterIter in the
// reference's .npy format -- not part of the hot path, plain C++ on the host (METIS is a host
// library too).
//
// Algorithm: MULTILEVEL, in the family METIS belongs to, with the coarsening that suits graphs whose parts are small
// (Cluster-GCN asks for 1 500 parts of ~100 nodes on Reddit, 15 000 on Amazon2M):
//   coarsen   size-constrained label propagation, twice: every node joins the neighbouring cluster it has the
//             most edge weight to as long as the cluster stays within a THIRD of a part's capacity; the clusters
//             are contracted to weighted vertices (parallel edges merged, weights added); on that graph the same
//             propagation with the bound at a part's full capacity lets the fragments of one dense region find
//             each other (they have far more weight between them than to anything else).  Round 5: the visiting
//             order is processed in chunks whose edge scans run on a pool of host threads against the labels as
//             they stood at the chunk's start, the moves applied in order -- 3x faster on 8 cores, and a function of
//             (graph, k, seed) alone whatever the thread count;
//   initial   on the coarsest graph: weighted linear-deterministic-greedy, heaviest vertices first (a vertex
//             goes to the part it has most weight to, discounted by how full that part is; a vertex nobody
//             claims opens the emptiest part), then refinement as below;
//   uncoarsen level by level: project the parts to the finer graph and refine -- strict-gain sweeps (every vertex moves
//             to the part it has the most edge weight to if that is more than it has to its own part and the target
//             has room; first with a sixteenth of slack above capacity, then strictly within capacity), then, round 5,
//             localised k-way Fiduccia-Mattheyses searches with rollback from the boundary vertices (KWay::fm: moves of
//             negative gain allowed, the best prefix kept) under a deterministic work budget;
//   balance   finest level: parts below floor((1 - imbalance) * n / k) pull their best-connected outside
//             nodes from parts that can spare them; no part is empty.
// Measured (profiles/r05_partitioner.json, scripts/partition_quality.py): the planted cut on the Reddit-like (153 k nodes /
// 1 500 parts, 2.6 s) and Amazon-like (1.71 M / 15 000, ~10 s on 8 cores; round 4: 27 s) block models; a torus mesh at
// 1.21 x the ideal cut (round 4, without the searches: ~2 x); a power-law community graph (sizes 30-400, mixing 0.3) below
// the cut of a partition built from the ground-truth communities.
// Neighbours are the union of in- and out-edges (multi-edges count with multiplicity); part
// sizes end within [floor((1 - imbalance) n/k), ceil((1 + imbalance) n/k)] whenever the graph
// allows it.  Deterministic for a given seed.
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <mutex>
#include <queue>
#include <thread>
#include <vector>

#include "common.h"

namespace {

// wall seconds of the last call's stages (gist_partition_last_stats): input graph, coarsening, initial partition +
// refinement per level, balance repair; [8] = levels, [9] = coarsest vertices
// (per calling thread: the "last call" is the caller's own, and concurrent calls do not write one array)
thread_local double g_stats[16];
struct StageTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double lap() {
        const auto t1 = std::chrono::steady_clock::now();
        const double s = std::chrono::duration<double>(t1 - t0).count();
        t0 = t1;
        return s;
    }
};

// ---- host threads: a small persistent pool (no OpenMP runtime beside torch's) ------------------------------------
// Every parallel loop below cuts its range into pieces whose RESULTS do not depend on which thread runs them, so the
// partition is a function of (graph, k, seed) alone -- whatever the machine's core count.
int host_threads() {
    const int forced = (int)gist::tune(GIST_TUNE_HOST_THREADS);
    if (forced > 0) return forced > 64 ? 64 : forced;
    int n = (int)std::thread::hardware_concurrency();
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {      // a container's CPU quota, not the host's core count
        long long quota = 0, period = 0;
        if (std::fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) {
            const int q = (int)((quota + period - 1) / period);
            if (q < n) n = q;
        }
        std::fclose(f);
    }
    // (the pool's dense per-thread accumulators are one int64 per vertex each -- 14 MB per thread at 1.7 M vertices --
    // and the 3x over one thread was measured at 8: a machine's core count must not decide the scratch memory)
    return n < 1 ? 1 : (n > 8 ? 8 : n);
}

class Pool {
  public:
    explicit Pool(int n) : n_(n < 1 ? 1 : n) {
        for (int t = 1; t < n_; ++t) workers_.emplace_back([this, t] { loop(t); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; ++gen_; }
        cv_.notify_all();
        for (auto &w : workers_) w.join();
    }
    int size() const { return n_; }
    // fn(thread, piece) for piece in [0, pieces): pieces are handed out in order, any piece to any thread
    void run(int64_t pieces, const std::function<void(int, int64_t)> &fn) {
        if (pieces <= 0) return;
        if (n_ == 1 || pieces == 1) { for (int64_t i = 0; i < pieces; ++i) fn(0, i); return; }
        { std::lock_guard<std::mutex> lk(m_); fn_ = &fn; pieces_ = pieces; next_ = 0; left_ = n_ - 1; ++gen_; }
        cv_.notify_all();
        work(0);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return left_ == 0; });
        fn_ = nullptr;
    }
  private:
    void work(int t) {
        for (;;) {
            int64_t i;
            { std::lock_guard<std::mutex> lk(m_); if (next_ >= pieces_) return; i = next_++; }
            (*fn_)(t, i);
        }
    }
    void loop(int t) {
        uint64_t seen = 0;
        for (;;) {
            { std::unique_lock<std::mutex> lk(m_); cv_.wait(lk, [&] { return gen_ != seen; }); seen = gen_; if (stop_) return; }
            work(t);
            { std::lock_guard<std::mutex> lk(m_); if (--left_ == 0) done_.notify_one(); }
        }
    }
    int n_;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(int, int64_t)> *fn_ = nullptr;
    int64_t pieces_ = 0, next_ = 0;
    int left_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

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

WGraph input_level(const int32_t *rowptr, const int32_t *col, const int32_t *t_rowptr, const int32_t *t_col, int64_t n,
                   Pool &pool) {
    WGraph g;
    g.n = n;
    g.xadj.assign(n + 1, 0);
    g.vw.assign(n, 1);
    const int64_t piece = 16384, pieces = (n + piece - 1) / piece;
    pool.run(pieces, [&](int, int64_t q) {
        for (int64_t v = q * piece; v < std::min(n, (q + 1) * piece); ++v) {
            int64_t d = 0;
            for (int32_t e = rowptr[v]; e < rowptr[v + 1]; ++e) d += col[e] != v;
            if (t_rowptr)
                for (int32_t e = t_rowptr[v]; e < t_rowptr[v + 1]; ++e) d += t_col[e] != v;
            g.xadj[v + 1] = d;
        }
    });
    for (int64_t v = 0; v < n; ++v) g.xadj[v + 1] += g.xadj[v];
    g.adj.resize(g.xadj[n]);
    pool.run(pieces, [&](int, int64_t q) {
        for (int64_t v = q * piece; v < std::min(n, (q + 1) * piece); ++v) {
            int64_t w = g.xadj[v];
            for (int32_t e = rowptr[v]; e < rowptr[v + 1]; ++e)
                if (col[e] != v) g.adj[w++] = col[e];
            if (t_rowptr)
                for (int32_t e = t_rowptr[v]; e < t_rowptr[v + 1]; ++e)
                    if (t_col[e] != v) g.adj[w++] = t_col[e];
        }
    });
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

// size-constrained label propagation: label[v] in [0, n_clusters), every cluster's weight <= bound.
// Round 5: the visiting order is cut into chunks of 8192 vertices; the vertices of a chunk choose their clusters IN
// PARALLEL from the labels as they stood when the chunk began (the edge scans: all of the cost), then the moves are
// applied one by one in visiting order under the weight bound as it then stands.  A function of the seed only.
int64_t cluster_lp(const WGraph &g, int64_t bound, int iters, Rng &rng, std::vector<int32_t> &label, Pool &pool) {
    const int64_t n = g.n;
    label.resize(n);
    std::vector<int64_t> cw(n);
    for (int64_t v = 0; v < n; ++v) { label[v] = (int32_t)v; cw[v] = g.vw[v]; }
    const int T = pool.size();
    std::vector<std::vector<int64_t>> conn(T);
    std::vector<std::vector<int32_t>> touched(T);
    for (int t = 0; t < T; ++t) conn[t].assign(n, 0);
    const std::vector<int32_t> order = random_order(n, rng);
    // (a chunk is at most 1/64 of the graph: two neighbours decided in the same chunk can swap labels instead of merging)
    const int64_t chunk = std::max<int64_t>(64, std::min<int64_t>(8192, n / 64)), sub = 256;
    std::vector<int32_t> want(chunk);
    for (int it = 0; it < iters; ++it) {
        int64_t moved = 0;
        for (int64_t c0 = 0; c0 < n; c0 += chunk) {
            const int64_t c1 = std::min(n, c0 + chunk);
            pool.run((c1 - c0 + sub - 1) / sub, [&](int t, int64_t q) {
                std::vector<int64_t> &cn = conn[t];
                std::vector<int32_t> &tc = touched[t];
                for (int64_t i = c0 + q * sub; i < std::min(c1, c0 + (q + 1) * sub); ++i) {
                    const int32_t v = order[i];
                    const int32_t cur = label[v];
                    want[i - c0] = cur;
                    if (g.xadj[v] == g.xadj[v + 1]) continue;
                    tc.clear();
                    for (int64_t e = g.xadj[v]; e < g.xadj[v + 1]; ++e) {
                        const int32_t l = label[g.adj[e]];
                        if (cn[l] == 0) tc.push_back(l);
                        cn[l] += g.w(e);
                    }
                    int32_t best = cur;
                    int64_t best_c = cn[cur];           // staying wins ties: the propagation settles
                    for (int32_t l : tc) {
                        if (l != cur && cw[l] + g.vw[v] <= bound &&
                            (cn[l] > best_c || (cn[l] == best_c && best != cur && (cw[l] < cw[best] || (cw[l] == cw[best] && l < best))))) {
                            best = l;
                            best_c = cn[l];
                        }
                    }
                    for (int32_t l : tc) cn[l] = 0;
                    want[i - c0] = best;
                }
            });
            for (int64_t i = c0; i < c1; ++i) {          // apply in visiting order; the bound as it stands NOW
                const int32_t v = order[i], best = want[i - c0], cur = label[v];
                if (best == cur || cw[best] + g.vw[v] > bound) continue;
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

WGraph contract(const WGraph &g, const std::vector<int32_t> &label, int64_t nc, Pool &pool) {
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
    // a coarse vertex's neighbour list = its members' edges merged by target cluster, sorted by cluster id: built per
    // piece of 1024 coarse vertices into the piece's own arrays (any thread), concatenated in piece order
    const int64_t piece = 1024, pieces = (nc + piece - 1) / piece;
    const int T = pool.size();
    std::vector<std::vector<int64_t>> acc(T);
    for (int t = 0; t < T; ++t) acc[t].assign(nc, 0);
    std::vector<std::vector<int32_t>> p_adj(pieces), p_ew(pieces), p_deg(pieces);
    pool.run(pieces, [&](int t, int64_t pc) {
        std::vector<int64_t> &ac = acc[t];
        std::vector<int32_t> touched;
        for (int64_t q = pc * piece; q < std::min(nc, (pc + 1) * piece); ++q) {
            touched.clear();
            for (int64_t i = start[q]; i < start[q + 1]; ++i) {
                const int32_t v = members[i];
                for (int64_t e = g.xadj[v]; e < g.xadj[v + 1]; ++e) {
                    const int32_t l = label[g.adj[e]];
                    if (l == q) continue;
                    if (ac[l] == 0) touched.push_back(l);
                    ac[l] += g.w(e);
                }
            }
            std::sort(touched.begin(), touched.end());
            for (int32_t l : touched) {
                p_adj[pc].push_back(l);
                p_ew[pc].push_back((int32_t)std::min<int64_t>(ac[l], INT32_MAX));
                ac[l] = 0;
            }
            p_deg[pc].push_back((int32_t)touched.size());
        }
    });
    c.xadj.assign(nc + 1, 0);
    for (int64_t pc = 0, q = 0; pc < pieces; ++pc)
        for (int32_t d : p_deg[pc]) { c.xadj[q + 1] = c.xadj[q] + d; ++q; }
    c.adj.resize(c.xadj[nc]);
    c.ew.resize(c.xadj[nc]);
    pool.run(pieces, [&](int, int64_t pc) {
        const int64_t at = c.xadj[std::min(nc, pc * piece)];
        std::copy(p_adj[pc].begin(), p_adj[pc].end(), c.adj.begin() + at);
        std::copy(p_ew[pc].begin(), p_ew[pc].end(), c.ew.begin() + at);
    });
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
    // One vertex's best move under the refinement rule: strict gain in edge weight, target within `limit`; a vertex of an
    // over-full part may also leave at equal weight.  cn / tc: the caller's accumulator (conn of this object, or a thread's).
    int32_t best_move(int32_t v, int64_t limit, std::vector<int64_t> &cn, std::vector<int32_t> &tc) const {
        const int32_t old = part[v];
        tc.clear();
        for (int64_t e = g.xadj[v]; e < g.xadj[v + 1]; ++e) {
            const int32_t p = part[g.adj[e]];
            if (p < 0) continue;
            if (cn[p] == 0) tc.push_back(p);
            cn[p] += g.w(e);
        }
        const bool over = size[old] > limit;
        int32_t best = old;
        int64_t best_c = cn[old] - (over ? 1 : 0);
        for (int32_t p : tc) {
            if (p == old || size[p] + g.vw[v] > limit) continue;
            if (cn[p] > best_c || (cn[p] == best_c && best != old && (size[p] < size[best] || (size[p] == size[best] && p < best)))) {
                best = p;
                best_c = cn[p];
            }
        }
        for (int32_t p : tc) cn[p] = 0;
        return best;
    }
    // refinement sweeps.  Round 5: the visiting order in chunks of 8192; the vertices of a chunk are screened IN PARALLEL
    // against the partition as it stood when the chunk began (the edge scans), then those that wanted to move are
    // decided again, one by one in visiting order, against the partition as it stands (exactly the serial rule) and moved.
    int64_t refine(const std::vector<int32_t> &order, int passes, int64_t limit, Pool &pool) {
        int64_t total = 0;
        const int T = pool.size();
        std::vector<std::vector<int64_t>> cns(T);
        std::vector<std::vector<int32_t>> tcs(T);
        for (int t = 0; t < T; ++t) cns[t].assign(k, 0);
        const int64_t chunk = std::max<int64_t>(64, std::min<int64_t>(8192, g.n / 64)), sub = 256;
        std::vector<uint8_t> wants(chunk);
        for (int t = 0; t < passes; ++t) {
            int64_t moved = 0;
            for (int64_t c0 = 0; c0 < g.n; c0 += chunk) {
                const int64_t c1 = std::min(g.n, c0 + chunk);
                pool.run((c1 - c0 + sub - 1) / sub, [&](int th, int64_t q) {
                    for (int64_t i = c0 + q * sub; i < std::min(c1, c0 + (q + 1) * sub); ++i)
                        wants[i - c0] = best_move(order[i], limit, cns[th], tcs[th]) != part[order[i]];
                });
                for (int64_t i = c0; i < c1; ++i) {
                    if (!wants[i - c0]) continue;
                    const int32_t v = order[i], old = part[v];
                    const int32_t best = best_move(v, limit, conn, touched);
                    if (best != old) {
                        size[old] -= g.vw[v];
                        size[best] += g.vw[v];
                        part[v] = best;
                        ++moved;
                    }
                }
            }
            total += moved;
            if (moved == 0) break;
        }
        return total;
    }

    // Localised k-way Fiduccia-Mattheyses with rollback (round 5): from every boundary vertex a short search moves the
    // vertex of highest gain among those touched so far -- gains may be NEGATIVE -- to its best admissible part, remembers the
    // best total seen and undoes everything after it.  What the strict-gain sweeps above cannot do: straighten a jagged
    // boundary, where every single move loses weight and two or three together gain.  `patience` moves without a new
    // best end a search; a vertex moves once per pass.  Serial; cost bounded by max_degree (searches from and through
    // heavier vertices are skipped: on the dense clustered graphs the sweeps already reach the planted cut).
    int64_t fm(const std::vector<int32_t> &order, int passes, int64_t limit, int patience, int64_t max_degree) {
        struct Cand { int64_t gain; int32_t v, to; };
        // edge visits this call may spend: 120 per edge of a sparse level (meshes, road-like graphs: where it pays), 2 per
        // edge of a dense one (clustered graphs, whose cut the sweeps already settle) -- deterministic, like everything here
        const int64_t m = g.xadj[g.n];
        int64_t budget = (m <= 16 * g.n ? 120 : 2) * m + 1024;
        auto eval = [&](int32_t v, Cand &c) {            // best admissible target of v (gain may be <= 0)
            const int32_t old = part[v];
            budget -= g.xadj[v + 1] - g.xadj[v];
            gather(v);
            int32_t best = -1;
            int64_t best_c = 0;
            for (int32_t p : touched) {
                if (p == old || size[p] + g.vw[v] > limit) continue;
                if (best < 0 || conn[p] > best_c || (conn[p] == best_c && (size[p] < size[best] || (size[p] == size[best] && p < best)))) {
                    best = p;
                    best_c = conn[p];
                }
            }
            c.v = v; c.to = best; c.gain = best < 0 ? 0 : best_c - conn[old];
            release();
            return best >= 0;
        };
        auto cmp = [](const Cand &a, const Cand &b) { return a.gain < b.gain || (a.gain == b.gain && a.v > b.v); };
        std::vector<int32_t> stamp(g.n, -1);              // pass in which the vertex moved (and stayed moved)
        std::vector<int32_t> queued(g.n, -1);             // search that last put the vertex into the queue
        std::vector<Cand> log;
        int64_t total_gain = 0;
        int32_t search = 0;
        for (int t = 0; t < passes; ++t) {
            int64_t pass_gain = 0;
            for (int64_t i = 0; i < g.n && budget > 0; ++i) {
                const int32_t s = order[i];
                if (stamp[s] == t || g.xadj[s + 1] - g.xadj[s] > max_degree || g.xadj[s + 1] == g.xadj[s]) continue;
                bool boundary = false;
                for (int64_t e = g.xadj[s]; e < g.xadj[s + 1] && !boundary; ++e) boundary = part[g.adj[e]] != part[s];
                if (!boundary) continue;
                ++search;
                std::priority_queue<Cand, std::vector<Cand>, decltype(cmp)> pq(cmp);
                Cand c;
                if (!eval(s, c)) continue;
                pq.push(c);
                queued[s] = search;
                log.clear();
                int64_t cur = 0, best = 0;
                size_t best_len = 0;
                while (!pq.empty() && (int)(log.size() - best_len) < patience && budget > 0) {
                    Cand top = pq.top();
                    pq.pop();
                    if (stamp[top.v] == t) continue;
                    Cand now;
                    if (!eval(top.v, now)) continue;
                    if (now.gain < top.gain) { pq.push(now); continue; }      // stale: its gain fell, try again later
                    const int32_t v = now.v, from = part[v];
                    size[from] -= g.vw[v];
                    size[now.to] += g.vw[v];
                    part[v] = now.to;
                    stamp[v] = t;
                    cur += now.gain;
                    now.to = from;                         // (the log keeps where it came from)
                    log.push_back(now);
                    if (cur > best) { best = cur; best_len = log.size(); }
                    for (int64_t e = g.xadj[v]; e < g.xadj[v + 1]; ++e) {
                        const int32_t u = g.adj[e];
                        if (stamp[u] == t || g.xadj[u + 1] - g.xadj[u] > max_degree) continue;
                        Cand cu;
                        if (eval(u, cu) && (queued[u] != search || cu.gain > 0)) { pq.push(cu); queued[u] = search; }
                    }
                }
                for (size_t j = log.size(); j > best_len; --j) {      // undo the tail behind the best total
                    const Cand &u = log[j - 1];
                    size[part[u.v]] -= g.vw[u.v];
                    size[u.to] += g.vw[u.v];
                    part[u.v] = u.to;
                    stamp[u.v] = -1;
                }
                pass_gain += best;
            }
            total_gain += pass_gain;
            if (pass_gain == 0 || budget <= 0) break;
        }
        return total_gain;
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
    for (double &x : g_stats) x = 0.0;
    StageTimer tm;

    // ---- coarsening: cluster weight bounds cap/3, 2 cap/3, cap --------------------------------------
    // (three rounds of merging: fragments of one dense region have far more weight between them than to
    // anything else, so they find each other; the last round's clusters are whole parts or pieces of one)
    std::vector<WGraph> levels;
    std::vector<std::vector<int32_t>> labels;          // labels[l][v] = vertex of level l+1 that v of level l joins
    Pool pool(host_threads());
    levels.push_back(input_level(rowptr, col, t_rowptr, t_col, n, pool));
    g_stats[0] = tm.lap();
    for (int lvl = 0; lvl < 2 && cap >= 6; ++lvl) {
        const WGraph &g = levels.back();
        if (g.n <= (int64_t)k || g.xadj[g.n] == 0) break;
        const int64_t bound = lvl == 1 ? cap : std::max<int64_t>(2, cap / 3);
        std::vector<int32_t> label;
        const int64_t nc = cluster_lp(g, bound, 8, rng, label, pool);
        g_stats[1] += tm.lap();
        if (nc < (int64_t)k || nc == g.n) break;       // too coarse for k parts / nothing merged
        WGraph c = contract(g, label, nc, pool);
        g_stats[2] += tm.lap();
        labels.push_back(std::move(label));
        levels.push_back(std::move(c));
    }

    // ---- initial partition of the coarsest level, then refine while uncoarsening --------------------
    std::vector<int32_t> part(levels.back().n, -1);
    g_stats[8] = (double)levels.size();
    g_stats[9] = (double)levels.back().n;
    for (int lvl = (int)levels.size() - 1; lvl >= 0; --lvl) {
        const WGraph &g = levels[lvl];
        std::vector<int32_t> order = bfs_order(g, rng);
        g_stats[3] += tm.lap();
        if (lvl == (int)levels.size() - 1) {
            // heaviest vertices first (stable within equal weight: BFS order): on a coarsened graph the k
            // heaviest clusters seed the parts and the fragments join the part they are tied to
            if (lvl > 0)
                std::stable_sort(order.begin(), order.end(), [&](int32_t a_, int32_t b_) { return g.vw[a_] > g.vw[b_]; });
            KWay kw(g, k, cap, part);
            kw.initial(order);
            kw.refine(order, std::max(n_passes, 1) * 2, cap, pool);
            kw.fm(order, std::max(n_passes, 1), cap, 16, 256);
        } else {
            std::vector<int32_t> fine(g.n);
            for (int64_t v = 0; v < g.n; ++v) fine[v] = part[labels[lvl][v]];
            part.swap(fine);
            KWay kw(g, k, cap, part);
            kw.recount();
            // first with a little slack above capacity (a vertex may enter a full part; the pass after sheds that
            // part's loosest vertex: the effect of a swap), then strictly within capacity
            kw.refine(order, std::max(n_passes, 1), cap + std::max<int64_t>(1, cap / 16), pool);
            kw.refine(order, std::max(n_passes, 1), cap, pool);
            kw.fm(order, std::max(n_passes, 1), cap, 16, lvl == 0 ? 48 : 256);
        }
        g_stats[lvl == 0 ? 5 : 4] += tm.lap();
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
    g_stats[6] = tm.lap();
    return GIST_OK;
}

extern "C" int gist_partition_last_stats(double *out, int32_t n) {
    GIST_REQUIRE(out != nullptr && n >= 0, "gist_partition_last_stats: bad arguments");
    for (int32_t i = 0; i < n && i < 16; ++i) out[i] = g_stats[i];
    return GIST_OK;
}
