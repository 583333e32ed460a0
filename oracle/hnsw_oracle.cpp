// TEST / MEASUREMENT INFRASTRUCTURE ONLY -- never imported by the product path.
//
// A restatement of the approximate index the reference actually queries: chromadb 0.4.22 stores its
// vectors in chroma-hnswlib (pinned [EXT] 0.7.3; neither package is in /root/reference nor
// installable here), i.e. a Hierarchical Navigable Small World graph (Malkov & Yashunin, 2016)
// with Chroma's defaults for the collection the reference creates (embedder.py:165-182, SURVEY.md
// Appendix A): space = cosine, M = 16 (32 links on layer 0), ef_construction = 100, search ef =
// max(10, k).  This file restates the PUBLISHED algorithm -- greedy descent through the upper
// layers, best-first search with a bounded candidate set on the target layer, neighbour selection
// by the diversity heuristic (Algorithm 4), level = floor(-ln(U) / ln(M)) -- so that bench.py can
// show what an approximate CPU path of this kind costs and recalls next to the exact scan.
// "restatement, not chromadb": parity unpinned (no hnswlib artefact to compare against).
//
//   hnsw_build(vectors [n,d] f32 unit-norm, n, d, M, ef_construction, seed, n_threads) -> handle
//   hnsw_search(handle, queries [b,d], b, k, ef, out_rows [b,k] i64, out_scores [b,k] f32, n_threads)
//   hnsw_free(handle)
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <queue>
#include <random>
#include <thread>
#include <vector>

namespace {

struct Graph {
    const float *x = nullptr;  // borrowed: the caller keeps the matrix alive
    int64_t n = 0;
    int d = 0, M = 16, M0 = 32, efc = 100;
    double mult = 0.0;
    std::vector<int> level;                           // top layer of each node
    std::vector<std::vector<std::vector<int>>> link;  // link[node][layer] = neighbours
    std::vector<std::mutex> lock;                     // one per node
    std::mutex entry_lock;
    int64_t entry = -1;
    int max_level = -1;

    inline float dist(const float *a, const float *b) const {  // cosine distance on unit vectors
        float s = 0.f;
        for (int i = 0; i < d; ++i) s += a[i] * b[i];
        return 1.0f - s;
    }
    inline const float *vec(int64_t i) const { return x + (size_t)i * d; }
};

typedef std::pair<float, int> DI;  // (distance, node)

// best-first search on one layer; returns up to ef closest as a max-heap (farthest on top)
std::priority_queue<DI> search_layer(Graph &g, const float *q, int ep, float ep_dist, int ef, int layer,
                                     std::vector<uint32_t> &visited, uint32_t stamp, bool locked) {
    std::priority_queue<DI> top;                                            // farthest first
    std::priority_queue<DI, std::vector<DI>, std::greater<DI>> cand;        // closest first
    top.emplace(ep_dist, ep);
    cand.emplace(ep_dist, ep);
    visited[ep] = stamp;
    std::vector<int> nb;
    while (!cand.empty()) {
        const DI c = cand.top();
        if (c.first > top.top().first && (int)top.size() >= ef) break;
        cand.pop();
        if (locked) {
            std::lock_guard<std::mutex> lk(g.lock[c.second]);
            nb = g.link[c.second][layer];
        } else {
            nb = g.link[c.second][layer];
        }
        for (int v : nb) {
            if (visited[v] == stamp) continue;
            visited[v] = stamp;
            const float dv = g.dist(q, g.vec(v));
            if ((int)top.size() < ef || dv < top.top().first) {
                cand.emplace(dv, v);
                top.emplace(dv, v);
                if ((int)top.size() > ef) top.pop();
            }
        }
    }
    return top;
}

// Algorithm 4 (hnswlib getNeighborsByHeuristic2): keep a candidate only if it is closer to the query
// than to every neighbour already kept
std::vector<int> select_neighbours(Graph &g, std::priority_queue<DI> top, int M) {
    std::vector<DI> c;
    while (!top.empty()) {
        c.push_back(top.top());
        top.pop();
    }
    std::reverse(c.begin(), c.end());  // closest first
    std::vector<int> keep;
    for (const DI &e : c) {
        if ((int)keep.size() >= M) break;
        bool good = true;
        for (int r : keep)
            if (g.dist(g.vec(e.second), g.vec(r)) < e.first) {
                good = false;
                break;
            }
        if (good) keep.push_back(e.second);
    }
    return keep;
}

void insert(Graph &g, int node, int lvl, std::vector<uint32_t> &visited, uint32_t &stamp) {
    std::unique_lock<std::mutex> el(g.entry_lock);
    int64_t ep = g.entry;
    const int top_level = g.max_level;
    if (ep < 0) {  // first node
        g.entry = node;
        g.max_level = lvl;
        return;
    }
    if (lvl <= top_level) el.unlock();  // only a new top level keeps the entry point locked
    const float *q = g.vec(node);
    float dcur = g.dist(q, g.vec(ep));
    for (int l = top_level; l > lvl; --l) {  // greedy descent
        bool moved = true;
        while (moved) {
            moved = false;
            std::vector<int> nb;
            {
                std::lock_guard<std::mutex> lk(g.lock[ep]);
                nb = g.link[ep][l];
            }
            for (int v : nb) {
                const float dv = g.dist(q, g.vec(v));
                if (dv < dcur) {
                    dcur = dv;
                    ep = v;
                    moved = true;
                }
            }
        }
    }
    for (int l = std::min(lvl, top_level); l >= 0; --l) {
        auto top = search_layer(g, q, (int)ep, dcur, g.efc, l, visited, ++stamp, true);
        const int Mmax = l == 0 ? g.M0 : g.M;
        std::vector<int> sel = select_neighbours(g, top, g.M);
        {
            std::lock_guard<std::mutex> lk(g.lock[node]);
            g.link[node][l] = sel;
        }
        for (int v : sel) {  // back links, pruned with the same heuristic
            std::lock_guard<std::mutex> lk(g.lock[v]);
            std::vector<int> &lv = g.link[v][l];
            if ((int)lv.size() < Mmax) {
                lv.push_back(node);
            } else {
                std::priority_queue<DI> c;
                c.emplace(g.dist(g.vec(v), q), node);
                for (int u : lv) c.emplace(g.dist(g.vec(v), g.vec(u)), u);
                lv = select_neighbours(g, c, Mmax);
            }
        }
        // next layer starts from the closest found here
        DI best = top.top();
        while (!top.empty()) {
            best = top.top();
            top.pop();
        }
        ep = best.second;
        dcur = best.first;
    }
    if (lvl > top_level) {
        g.entry = node;
        g.max_level = lvl;
    }
}

}  // namespace

extern "C" {

void *hnsw_build(const float *x, int64_t n, int d, int M, int ef_construction, uint64_t seed, int n_threads) {
    Graph *g = new Graph();
    g->x = x, g->n = n, g->d = d, g->M = M, g->M0 = 2 * M, g->efc = ef_construction;
    g->mult = 1.0 / log((double)M);
    g->level.resize(n);
    g->link.resize(n);
    std::vector<std::mutex> locks(n);
    g->lock.swap(locks);
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    for (int64_t i = 0; i < n; ++i) {
        double u = U(rng);
        if (u <= 0.0) u = 1e-300;
        g->level[i] = (int)(-log(u) * g->mult);
        g->link[i].resize(g->level[i] + 1);
    }
    if (n == 0) return g;
    {
        std::vector<uint32_t> visited(n, 0);
        uint32_t stamp = 0;
        insert(*g, 0, g->level[0], visited, stamp);
    }
    if (n_threads < 1) n_threads = 1;
    std::atomic<int64_t> next(1);
    auto work = [&]() {
        std::vector<uint32_t> visited(n, 0);
        uint32_t stamp = 0;
        for (;;) {
            const int64_t i = next.fetch_add(1);
            if (i >= n) break;
            insert(*g, (int)i, g->level[i], visited, stamp);
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) th.emplace_back(work);
    for (auto &t : th) t.join();
    return g;
}

void hnsw_search(void *h, const float *q, int b, int k, int ef, int64_t *out_rows, float *out_scores, int n_threads) {
    Graph &g = *(Graph *)h;
    if (ef < k) ef = k;
    if (n_threads < 1) n_threads = 1;
    std::atomic<int> next(0);
    auto work = [&]() {
        std::vector<uint32_t> visited(g.n, 0);
        uint32_t stamp = 0;
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= b) break;
            const float *qi = q + (size_t)i * g.d;
            for (int j = 0; j < k; ++j) {
                out_rows[(size_t)i * k + j] = -1;
                out_scores[(size_t)i * k + j] = -INFINITY;
            }
            if (g.entry < 0) continue;
            int ep = (int)g.entry;
            float dcur = g.dist(qi, g.vec(ep));
            for (int l = g.max_level; l > 0; --l) {
                bool moved = true;
                while (moved) {
                    moved = false;
                    for (int v : g.link[ep][l]) {
                        const float dv = g.dist(qi, g.vec(v));
                        if (dv < dcur) {
                            dcur = dv;
                            ep = v;
                            moved = true;
                        }
                    }
                }
            }
            auto top = search_layer(g, qi, ep, dcur, ef, 0, visited, ++stamp, false);
            while ((int)top.size() > k) top.pop();
            int j = (int)top.size() - 1;
            while (!top.empty()) {
                out_rows[(size_t)i * k + j] = top.top().second;
                out_scores[(size_t)i * k + j] = 1.0f - top.top().first;
                top.pop();
                --j;
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < std::min(n_threads, b > 0 ? b : 1); ++t) th.emplace_back(work);
    for (auto &t : th) t.join();
}

void hnsw_free(void *h) { delete (Graph *)h; }

}  // extern "C"
