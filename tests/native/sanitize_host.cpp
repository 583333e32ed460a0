// Sanitizer driver for the host-side C++ of libmmrag.so (tokenizer.cpp: multi-threaded WordPiece; host_merge.cpp: the
// G*k -> k merge).  Built by tests/test_sanitizers.py with -fsanitize=address,undefined and with -fsanitize=thread and
// run on the CPU box only ("run C++ host code under ASan/TSan in CPU unit tests").  Exercises the
// exported C-ABI exactly as the Python wrappers do and checks the results against straightforward restatements.
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mmrag.h"

// what common.hip provides inside the real library
namespace mmrag {
static thread_local char g_err[512];
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int num_cus() { return 256; }
}  // namespace mmrag

#define CHECK(cond)                                                     \
    do {                                                                \
        if (!(cond)) {                                                  \
            fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            exit(2);                                                    \
        }                                                               \
    } while (0)

static std::vector<uint32_t> cps_of(const std::string &s) {   // ASCII is enough here
    return std::vector<uint32_t>(s.begin(), s.end());
}

static void tokenizer_checks() {
    // vocabulary: specials at BERT's ids would need 30k entries; look-ups are by name, so a small table does
    std::vector<std::string> vocab = {"[PAD]", "[UNK]", "[CLS]", "[SEP]", "hello", "world", "##s", "##ing", "un", "##believ",
                                      "##able", ",", ".", "a", "b", "##b", "the", "quick", "brown", "fox"};
    std::vector<uint32_t> cps;
    std::vector<int64_t> off = {0};
    for (auto &t : vocab) {
        auto c = cps_of(t);
        cps.insert(cps.end(), c.begin(), c.end());
        off.push_back((int64_t)cps.size());
    }
    void *tk = mmrag_wordpiece_create(cps.data(), off.data(), (int)vocab.size(), 1);
    CHECK(tk != nullptr);
    std::mt19937 rng(7);
    std::vector<std::string> words = {"hello", "worlds", "Unbelievable", "the", "quick,", "BROWN", "fox.", "zzz", "a", "abb",
                                      "", "  ", "hello\tworld"};
    const int n = 4000, L = 48;
    std::vector<uint32_t> text;
    std::vector<int64_t> toff = {0};
    for (int i = 0; i < n; ++i) {
        std::string s;
        const int nw = (int)(rng() % 60);       // 0 words = empty text; long texts are truncated to L
        for (int w = 0; w < nw; ++w) s += words[rng() % words.size()] + " ";
        auto c = cps_of(s);
        text.insert(text.end(), c.begin(), c.end());
        toff.push_back((int64_t)text.size());
    }
    std::vector<int32_t> ids1((size_t)n * L, -7), len1(n, -7), ids8((size_t)n * L, -9), len8(n, -9);
    CHECK(mmrag_wordpiece_encode_batch(tk, text.data(), toff.data(), n, L, ids1.data(), len1.data(), 1) == MMRAG_OK);
    CHECK(mmrag_wordpiece_encode_batch(tk, text.data(), toff.data(), n, L, ids8.data(), len8.data(), 8) == MMRAG_OK);
    for (int i = 0; i < n; ++i) {
        CHECK(len1[i] == len8[i] && len1[i] >= 2 && len1[i] <= L);
        CHECK(ids1[(size_t)i * L] == 2 && ids1[(size_t)i * L + len1[i] - 1] == 3);   // [CLS] ... [SEP]
        for (int j = 0; j < len1[i]; ++j) CHECK(ids1[(size_t)i * L + j] == ids8[(size_t)i * L + j]);
    }
    // two batches encoded concurrently through ONE tokenizer object (the service tokenises from worker threads)
    std::vector<int32_t> idsA((size_t)n * L), lenA(n), idsB((size_t)n * L), lenB(n);
    std::thread ta([&] { CHECK(mmrag_wordpiece_encode_batch(tk, text.data(), toff.data(), n, L, idsA.data(), lenA.data(), 4) == MMRAG_OK); });
    std::thread tb([&] { CHECK(mmrag_wordpiece_encode_batch(tk, text.data(), toff.data(), n, L, idsB.data(), lenB.data(), 4) == MMRAG_OK); });
    ta.join();
    tb.join();
    CHECK(lenA == len1 && lenB == len1 && idsA == idsB);
    // bad arguments are refused, not dereferenced
    CHECK(mmrag_wordpiece_encode_batch(tk, text.data(), toff.data(), n, 1, ids1.data(), len1.data(), 1) != MMRAG_OK);
    mmrag_wordpiece_destroy(tk);
    CHECK(mmrag_wordpiece_create(cps.data(), off.data(), 0, 1) == nullptr);
}

static void merge_checks() {
    std::mt19937 rng(11);
    for (int trial = 0; trial < 200; ++trial) {
        const int G = 1 + rng() % 8, B = 1 + rng() % 9, k_in = 1 + rng() % 20, k = 1 + rng() % 20;
        std::vector<float> s((size_t)G * B * k_in);
        std::vector<int64_t> r(s.size());
        int64_t next_row = 0;
        for (int g = 0; g < G; ++g)
            for (int b = 0; b < B; ++b) {
                std::vector<float> v(k_in);
                for (auto &x : v) x = (float)((int)(rng() % 9) - 4);           // few distinct values: many ties
                std::sort(v.begin(), v.end(), std::greater<float>());
                const int valid = (int)(rng() % (k_in + 1));
                for (int i = 0; i < k_in; ++i) {
                    const size_t at = ((size_t)g * B + b) * k_in + i;
                    s[at] = i < valid ? v[i] : -INFINITY;
                    r[at] = i < valid ? next_row++ * 3 + (int64_t)(rng() % 3) : -1;
                }
            }
        std::vector<float> os((size_t)B * k), ps((size_t)B * k);
        std::vector<int64_t> orr((size_t)B * k), pr((size_t)B * k);
        CHECK(mmrag_merge_topk_host(s.data(), r.data(), G, B, k_in, k, os.data(), orr.data()) == MMRAG_OK);
        // the packed form: one block per rank [rows | scores | pad to 8]
        const size_t nb = (size_t)B * k_in, bb = (nb * 12 + 7) / 8 * 8;
        std::vector<uint64_t> blocks((size_t)G * bb / 8, 0);
        for (int g = 0; g < G; ++g) {
            char *blk = (char *)blocks.data() + (size_t)g * bb;
            memcpy(blk, r.data() + (size_t)g * nb, nb * 8);
            memcpy(blk + nb * 8, s.data() + (size_t)g * nb, nb * 4);
        }
        CHECK(mmrag_merge_topk_host_packed(blocks.data(), G, B, k_in, k, ps.data(), pr.data()) == MMRAG_OK);
        CHECK(os == ps && orr == pr);
        for (int b = 0; b < B; ++b) {                     // restatement: sort all valid candidates of the query
            std::vector<std::pair<float, int64_t>> all;
            for (int g = 0; g < G; ++g)
                for (int i = 0; i < k_in; ++i) {
                    const size_t at = ((size_t)g * B + b) * k_in + i;
                    if (r[at] >= 0) all.push_back({s[at], r[at]});
                }
            std::sort(all.begin(), all.end(), [](auto &x, auto &y) { return x.first > y.first || (x.first == y.first && x.second < y.second); });
            for (int i = 0; i < k; ++i) {
                if (i < (int)all.size()) CHECK(os[(size_t)b * k + i] == all[i].first && orr[(size_t)b * k + i] == all[i].second);
                else CHECK(orr[(size_t)b * k + i] == -1);
            }
        }
    }
    float o;
    int64_t orow;
    CHECK(mmrag_merge_topk_host(nullptr, nullptr, 1, 1, 1, 1, &o, &orow) != MMRAG_OK);
}

static void clip_bpe_checks() {
    // byte-level BPE: a handful of merges over ASCII (the printable table maps ASCII letters to themselves)
    std::vector<std::string> vocab = {"a", "b", "c", "a</w>", "b</w>", "c</w>", "ab", "ab</w>", "abc</w>", "'s</w>", "!</w>",
                                      "<|startoftext|>", "<|endoftext|>"};
    std::vector<std::string> merges = {"a b", "a b</w>", "ab c</w>", "' s</w>"};
    auto pack = [](const std::vector<std::string> &v, std::vector<uint32_t> &cps, std::vector<int64_t> &off) {
        off.assign(1, 0);
        for (auto &t : v) {
            auto c = cps_of(t);
            cps.insert(cps.end(), c.begin(), c.end());
            off.push_back((int64_t)cps.size());
        }
    };
    std::vector<uint32_t> vc, mc;
    std::vector<int64_t> vo, mo;
    pack(vocab, vc, vo);
    pack(merges, mc, mo);
    std::vector<int32_t> ids(vocab.size());
    for (size_t i = 0; i < ids.size(); ++i) ids[i] = (int32_t)i;
    void *tk = mmrag_clip_bpe_create(vc.data(), vo.data(), ids.data(), (int)vocab.size(), mc.data(), mo.data(), (int)merges.size());
    CHECK(tk != nullptr);
    std::vector<std::string> words = {"abc", "ab", "a's", "cab!", "", "  ", "<|endoftext|>", "abcabcabc", "zzz", "a b c"};
    std::mt19937 rng(9);
    const int n = 3000, L = 24;
    std::vector<uint32_t> text;
    std::vector<int64_t> toff = {0};
    for (int i = 0; i < n; ++i) {
        std::string s;
        for (int k = (int)(rng() % 8); k > 0; --k) s += words[rng() % words.size()] + (rng() % 3 ? " " : "");
        auto c = cps_of(s);
        text.insert(text.end(), c.begin(), c.end());
        toff.push_back((int64_t)text.size());
    }
    std::vector<int32_t> ids1((size_t)n * L), ids8((size_t)n * L), len1(n), len8(n);
    CHECK(mmrag_clip_bpe_encode_batch(tk, text.data(), toff.data(), n, L, ids1.data(), len1.data(), 1) == MMRAG_OK);
    CHECK(mmrag_clip_bpe_encode_batch(tk, text.data(), toff.data(), n, L, ids8.data(), len8.data(), 8) == MMRAG_OK);
    for (int i = 0; i < n; ++i) {
        CHECK(len1[i] == len8[i] && len1[i] >= 2 && len1[i] <= L);
        CHECK(ids1[(size_t)i * L] == 11 && ids1[(size_t)i * L + len1[i] - 1] == 12);
        for (int j = 0; j < len1[i]; ++j) CHECK(ids1[(size_t)i * L + j] == ids8[(size_t)i * L + j]);
    }
    CHECK(mmrag_clip_bpe_encode_batch(tk, text.data(), toff.data(), n, 1, ids1.data(), len1.data(), 1) != MMRAG_OK);
    mmrag_clip_bpe_destroy(tk);
    CHECK(mmrag_clip_bpe_create(vc.data(), vo.data(), ids.data(), 3, mc.data(), mo.data(), 0) == nullptr);   // no specials
}

int main() {
    tokenizer_checks();
    clip_bpe_checks();
    merge_checks();
    // concurrent merges + tokenisation (no shared mutable state is the claim)
    std::vector<std::thread> th;
    for (int i = 0; i < 6; ++i) th.emplace_back(i % 3 == 0 ? merge_checks : (i % 3 == 1 ? tokenizer_checks : clip_bpe_checks));
    for (auto &t : th) t.join();
    printf("sanitize_host: ok\n");
    return 0;
}
