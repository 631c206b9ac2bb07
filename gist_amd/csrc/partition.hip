// Host-side k-way graph partitioner: the stand-in for dgl.transform.metis_partition
// (reference cluster_gcn/partition_utils.py:11-18; METIS itself is a third-party library that
// is not available offline).  One-time data preparation, cached by ClusterIter in the
// reference's .npy format -- not part of the hot path, plain C++ on the host.
//
// Algorithm: restreaming linear-deterministic-greedy (LDG) partitioning.
//   pass 0  visits the nodes in BFS order and puts each into the part that holds most of its
//           already placed neighbours, discounted by how full that part is
//           (score = cnt * (1 - size/cap)); a node nobody claims seeds an empty part, so many
//           regions grow at once;
//   pass t  (restreaming) revisits every node in the same order with ALL neighbours placed
//           and moves it to its best part under the same balance-discounted rule;
//   last    parts below floor((1 - imbalance) * n / k) pull their best-connected outside
//           nodes from parts that can spare them.
// Neighbours are the union of in- and out-edges (multi-edges count with multiplicity); part
// sizes end within [floor((1 - imbalance) n/k), ceil((1 + imbalance) n/k)] whenever the graph
// allows it, and no part is empty.
#include <algorithm>
#include <cstdint>
#include <vector>

#include "common.h"

namespace {

struct Rng {      // splitmix64: the order of BFS restarts depends on the seed only
    uint64_t s;
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
};

}  // namespace

extern "C" int gist_partition_graph(const int32_t *rowptr, const int32_t *col,
                                    const int32_t *t_rowptr, const int32_t *t_col, int64_t n,
                                    int32_t k, uint64_t seed, int32_t n_passes, float imbalance,
                                    int32_t *part) {
    GIST_REQUIRE(n > 0 && k > 0 && k <= n, "gist_partition_graph: need 0 < k <= n");
    GIST_REQUIRE(rowptr && col && part, "gist_partition_graph: null pointer");
    GIST_REQUIRE(n < (1LL << 31), "gist_partition_graph: n >= 2^31");
    GIST_REQUIRE(imbalance >= 0.f && n_passes >= 0, "gist_partition_graph: bad parameters");
    const int64_t target = (n + k - 1) / k;
    int64_t cap = (int64_t)((1.0 + (double)imbalance) * (double)n / (double)k + 0.999999);
    if (cap < target) cap = target;

    // ---- visiting order: BFS, restarted from a seeded random permutation -----------------
    std::vector<int32_t> order;
    order.reserve(n);
    {
        std::vector<int32_t> perm(n);
        for (int64_t i = 0; i < n; ++i) perm[i] = (int32_t)i;
        Rng rng{seed};
        for (int64_t i = n - 1; i > 0; --i) std::swap(perm[i], perm[rng.next() % (uint64_t)(i + 1)]);
        std::vector<uint8_t> seen(n, 0);
        for (int64_t s = 0; s < n; ++s) {
            if (seen[perm[s]]) continue;
            size_t head = order.size();
            order.push_back(perm[s]);
            seen[perm[s]] = 1;
            while (head < order.size()) {
                const int32_t v = order[head++];
                for (int pass = 0; pass < 2; ++pass) {
                    const int32_t *rp = pass == 0 ? rowptr : t_rowptr;
                    const int32_t *cl = pass == 0 ? col : t_col;
                    if (!rp) continue;
                    for (int32_t e = rp[v]; e < rp[v + 1]; ++e) {
                        const int32_t u = cl[e];
                        if (!seen[u]) { seen[u] = 1; order.push_back(u); }
                    }
                }
            }
        }
    }

    std::vector<int32_t> size(k, 0), cnt(k, 0), touched;
    touched.reserve(1024);
    for (int64_t v = 0; v < n; ++v) part[v] = -1;
    int32_t open_part = 0;                 // next part a neighbour-less node opens (pass 0)

    auto best_part = [&](int32_t v) -> int32_t {
        touched.clear();
        for (int pass = 0; pass < 2; ++pass) {
            const int32_t *rp = pass == 0 ? rowptr : t_rowptr;
            const int32_t *cl = pass == 0 ? col : t_col;
            if (!rp) continue;
            for (int32_t e = rp[v]; e < rp[v + 1]; ++e) {
                const int32_t u = cl[e];
                if (u == v) continue;
                const int32_t p = part[u];
                if (p < 0) continue;
                if (cnt[p]++ == 0) touched.push_back(p);
            }
        }
        int32_t best = -1;
        double best_score = -1.0;
        for (int32_t p : touched) {
            if (size[p] < cap) {
                const double sc = (double)cnt[p] * (1.0 - (double)size[p] / (double)cap);
                if (sc > best_score || (sc == best_score && (size[p] < size[best] ||
                                                              (size[p] == size[best] && p < best)))) {
                    best_score = sc;
                    best = p;
                }
            }
            cnt[p] = 0;
        }
        return best;
    };
    auto emptiest_from = [&](int32_t start) -> int32_t {   // an empty part if there is one,
        int32_t best = -1;                                   // else the least loaded below target / cap
        for (int32_t i = 0; i < k; ++i) {
            const int32_t p = (start + i) % k;
            if (size[p] < target && (best < 0 || size[p] < size[best])) {
                best = p;
                if (size[p] == 0) break;
            }
        }
        if (best < 0)
            for (int32_t p = 0; p < k; ++p)
                if (size[p] < cap && (best < 0 || size[p] < size[best])) best = p;
        return best;
    };

    // ---- pass 0: a node nobody claims seeds a new part (many regions grow at once) ---------
    for (int64_t i = 0; i < n; ++i) {
        const int32_t v = order[i];
        int32_t p = best_part(v);
        if (p < 0) {
            p = emptiest_from(open_part);
            open_part = (p + 1) % k;
        }
        part[v] = p;
        ++size[p];
    }
    // ---- restreaming ------------------------------------------------------------------------
    for (int t = 0; t < n_passes; ++t) {
        int64_t moved = 0;
        for (int64_t i = 0; i < n; ++i) {
            const int32_t v = order[i];
            const int32_t old = part[v];
            --size[old];
            part[v] = -1;
            int32_t p = best_part(v);
            if (p < 0) p = old;
            part[v] = p;
            ++size[p];
            moved += p != old;
        }
        if (moved == 0) break;
    }
    // ---- lower balance bound: a part below lo pulls, one at a time, the outside node with ----
    // ---- most edges into it from a part that can spare one (else any node of the largest) ----
    const int64_t lo = (int64_t)((1.0 - (double)imbalance) * (double)n / (double)k);
    if (lo > 0) {
        std::vector<std::vector<int32_t>> members(k);
        for (int64_t v = 0; v < n; ++v) members[part[v]].push_back((int32_t)v);
        std::vector<int32_t> gain(n, 0), cand;
        for (int32_t p = 0; p < k; ++p) {
            while ((int64_t)members[p].size() < lo) {
                cand.clear();
                for (int32_t v : members[p])
                    for (int pass = 0; pass < 2; ++pass) {
                        const int32_t *rp = pass == 0 ? rowptr : t_rowptr;
                        const int32_t *cl = pass == 0 ? col : t_col;
                        if (!rp) continue;
                        for (int32_t e = rp[v]; e < rp[v + 1]; ++e) {
                            const int32_t u = cl[e];
                            const int32_t q = part[u];
                            if (q == p || (int64_t)members[q].size() <= lo) continue;
                            if (gain[u]++ == 0) cand.push_back(u);
                        }
                    }
                int32_t pick = -1;
                for (int32_t u : cand) {
                    if (pick < 0 || gain[u] > gain[pick] || (gain[u] == gain[pick] && u < pick)) pick = u;
                }
                for (int32_t u : cand) gain[u] = 0;
                if (pick < 0) {                      // no outside neighbour to spare: largest part
                    int32_t big = -1;
                    for (int32_t q = 0; q < k; ++q)
                        if (q != p && (big < 0 || members[q].size() > members[big].size())) big = q;
                    if (big < 0 || (int64_t)members[big].size() <= lo) break;
                    pick = members[big].back();
                }
                const int32_t q = part[pick];
                auto &mq = members[q];
                mq.erase(std::find(mq.begin(), mq.end(), pick));
                members[p].push_back(pick);
                part[pick] = p;
            }
        }
        for (int32_t p = 0; p < k; ++p) size[p] = (int32_t)members[p].size();
    }
    // ---- no empty parts -------------------------------------------------------------------
    for (int32_t p = 0; p < k; ++p) {
        if (size[p] > 0) continue;
        int32_t big = 0;
        for (int32_t q = 1; q < k; ++q)
            if (size[q] > size[big]) big = q;
        for (int64_t i = n - 1; i >= 0; --i) {
            const int32_t v = order[i];
            if (part[v] == big) { part[v] = p; --size[big]; ++size[p]; break; }
        }
    }
    return GIST_OK;
}
