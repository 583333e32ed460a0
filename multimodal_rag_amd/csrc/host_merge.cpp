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

static int merge_host(const float *scores, const int64_t *rows, size_t g_stride_s, size_t g_stride_r, int G, int B,
                      int k_in, int k, float *out_scores, int64_t *out_rows);

extern "C" int mmrag_merge_topk_host(const float *scores, const int64_t *rows, int G, int B, int k_in,
                                     int k, float *out_scores, int64_t *out_rows) {
    return merge_host(scores, rows, (size_t)B * k_in, (size_t)B * k_in, G, B, k_in, k, out_scores, out_rows);
}

// One all-gather instead of two: each rank contributes ONE block
//   [rows B*k_in i64 | scores B*k_in f32 | pad to a multiple of 8 bytes]
extern "C" int mmrag_merge_topk_host_packed(const void *blocks, int G, int B, int k_in, int k, float *out_scores,
                                            int64_t *out_rows) {
    MMRAG_CHECK_ARG(blocks && ((uintptr_t)blocks % 8) == 0, "merge_topk_host_packed: null or misaligned pointer");
    const size_t nb = (size_t)B * k_in;
    const size_t block_bytes = (nb * 12 + 7) / 8 * 8;
    const int64_t *r0 = (const int64_t *)blocks;
    const float *s0 = (const float *)((const char *)blocks + nb * 8);
    return merge_host(s0, r0, block_bytes / 4, block_bytes / 8, G, B, k_in, k, out_scores, out_rows);
}

static int merge_host(const float *scores, const int64_t *rows, size_t g_stride_s, size_t g_stride_r, int G, int B,
                      int k_in, int k, float *out_scores, int64_t *out_rows) {
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
            const float *sg = scores + (size_t)g * g_stride_s + (size_t)b * k_in;
            const int64_t *rg = rows + (size_t)g * g_stride_r + (size_t)b * k_in;
            for (int i = 0; i < k_in; ++i)
                if (rg[i] >= 0 && sg[i] > -INFINITY) c.push_back({sg[i], rg[i]});
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
