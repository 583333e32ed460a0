// Host half of the sharded search: merge G ranks' local top-k into the global top-k.
// North star: "each GPU returning its local top-k (ids+scores) via RCCL all-gather over xGMI
// for a final host merge".  Pure C++ on host memory; same ordering rule as the device merge
// (score descending, ties -> lower global row).  No reference counterpart (the reference is
// single-process); it is the exchange step of section 8e.
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "mmrag_internal.h"

extern "C" int mmrag_merge_topk_host(const float *scores, const int64_t *rows, int G, int B, int k_in,
                                     int k, float *out_scores, int64_t *out_rows) {
    MMRAG_CHECK_ARG(scores && rows && out_scores && out_rows, "merge_topk_host: null pointer");
    MMRAG_CHECK_ARG(G > 0 && B > 0 && k_in > 0, "merge_topk_host: bad shape G=%d B=%d k_in=%d", G, B, k_in);
    MMRAG_CHECK_ARG(k >= 1 && k <= MMRAG_MAX_K, "merge_topk_host: k=%d outside 1..%d", k, MMRAG_MAX_K);
    struct Cand {
        float s;
        int64_t r;
    };
    std::vector<Cand> c;
    c.reserve((size_t)G * k_in);
    for (int b = 0; b < B; ++b) {
        c.clear();
        for (int g = 0; g < G; ++g) {
            const size_t base = ((size_t)g * B + b) * k_in;
            for (int i = 0; i < k_in; ++i)
                if (rows[base + i] >= 0 && scores[base + i] > -INFINITY) c.push_back({scores[base + i], rows[base + i]});
        }
        const size_t keep = std::min<size_t>(k, c.size());
        std::partial_sort(c.begin(), c.begin() + keep, c.end(),
                          [](const Cand &a, const Cand &b2) { return a.s > b2.s || (a.s == b2.s && a.r < b2.r); });
        for (int i = 0; i < k; ++i) {
            out_scores[(size_t)b * k + i] = (size_t)i < keep ? c[i].s : -INFINITY;
            out_rows[(size_t)b * k + i] = (size_t)i < keep ? c[i].r : -1;
        }
    }
    return MMRAG_OK;
}
