// Host side of the CLIP text tower (BASELINE config 4): byte-level BPE, the tokenisation step in front of
// mmrag_encoder_forward for the ViT-B/32 text encoder.  The reference names CLIP only in its config
// (config.py:106), so what this must equal is the package's own Python form,
// multimodal_rag_amd/tokenizer.py:ClipBpeTokenizer, which tests pin to transformers.CLIPTokenizer: a Python loop
// does a few hundred captions per second per core, one GPU embeds tens of thousands.
//
// The caller has already applied NFC, collapsed whitespace and lower-cased the text (C-speed str methods in
// Python).  Here: split with CLIP's pattern
//     <|startoftext|> | <|endoftext|> | 's | 't | 're | 've | 'm | 'll | 'd | \p{L}+ | \p{N} | [^\s\p{L}\p{N}]+
// (first alternative that matches at a position, greedy inside it; character classes from the `regex` module the
// Python form matches with: clip_bpe_tables.inc), map the UTF-8 bytes of a piece to GPT-2's printable characters,
// merge adjacent symbols by rank (every occurrence of the best pair, left to right, `</w>` on the last symbol),
// look the symbols up (unknown -> <|endoftext|>), emit [sot] ids[: max_length - 2] [eot].  One thread per slice of
// the batch, a piece -> ids cache per thread.
#include <algorithm>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "mmrag_internal.h"

namespace {

#include "clip_bpe_tables.inc"

typedef std::u32string u32s;

template <size_t N>
bool in_rng(const uint32_t (&t)[N][2], uint32_t c) {
    size_t lo = 0, hi = N;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (c < t[mid][0]) hi = mid;
        else if (c > t[mid][1]) lo = mid + 1;
        else return true;
    }
    return false;
}
inline bool is_l(uint32_t c) { return c < 0x80 ? ((c | 0x20) >= 'a' && (c | 0x20) <= 'z') : in_rng(RANGES_RX_L, c); }
inline bool is_n(uint32_t c) { return c < 0x80 ? (c >= '0' && c <= '9') : in_rng(RANGES_RX_N, c); }
inline bool is_ws(uint32_t c) { return in_rng(RANGES_RX_WS, c); }

struct U32Hash {
    size_t operator()(const u32s &s) const {
        uint64_t h = 1469598103934665603ull;
        for (char32_t c : s) h = (h ^ (uint64_t)c) * 1099511628211ull;
        return (size_t)h;
    }
};

struct Bpe {
    // symbols are interned: id of a symbol string (every vocabulary entry, every merge operand and result)
    std::unordered_map<u32s, int32_t, U32Hash> sym;
    std::vector<int32_t> vocab_id;                      // symbol -> vocabulary id, or -1
    std::unordered_map<uint64_t, std::pair<int32_t, int32_t>> merges;   // (a << 32 | b) -> (rank, merged symbol)
    int32_t byte_sym[256], byte_sym_end[256];           // one-character symbols, plain and with </w>
    int32_t sot, eot;

    int32_t intern(const u32s &s) {
        auto it = sym.find(s);
        if (it != sym.end()) return it->second;
        const int32_t id = (int32_t)vocab_id.size();
        sym.emplace(s, id);
        vocab_id.push_back(-1);
        return id;
    }
};

const char32_t END_W[] = U"</w>";
const char32_t SOT_S[] = U"<|startoftext|>", EOT_S[] = U"<|endoftext|>";

bool starts_with(const uint32_t *p, size_t n, const char32_t *lit, size_t len) {
    if (n < len) return false;
    for (size_t i = 0; i < len; ++i)
        if (p[i] != (uint32_t)lit[i]) return false;
    return true;
}

// ids of one piece (no special tokens): bytes -> symbols -> merges -> vocabulary ids
void bpe_piece(const Bpe &b, const uint32_t *p, size_t n, std::vector<int32_t> &out, std::vector<int32_t> &word) {
    word.clear();
    int last_byte = -1;
    for (size_t i = 0; i < n; ++i) {   // UTF-8 bytes of the piece, each as its printable character's symbol
        const uint32_t c = p[i];
        uint8_t u[4];
        int m;
        if (c < 0x80) u[0] = (uint8_t)c, m = 1;
        else if (c < 0x800) u[0] = (uint8_t)(0xC0 | (c >> 6)), u[1] = (uint8_t)(0x80 | (c & 0x3F)), m = 2;
        else if (c < 0x10000)
            u[0] = (uint8_t)(0xE0 | (c >> 12)), u[1] = (uint8_t)(0x80 | ((c >> 6) & 0x3F)), u[2] = (uint8_t)(0x80 | (c & 0x3F)), m = 3;
        else
            u[0] = (uint8_t)(0xF0 | (c >> 18)), u[1] = (uint8_t)(0x80 | ((c >> 12) & 0x3F)),
            u[2] = (uint8_t)(0x80 | ((c >> 6) & 0x3F)), u[3] = (uint8_t)(0x80 | (c & 0x3F)), m = 4;
        for (int k = 0; k < m; ++k) word.push_back(b.byte_sym[u[k]]);
        last_byte = u[m - 1];
    }
    if (word.empty()) return;
    word.back() = b.byte_sym_end[last_byte];   // the last symbol carries </w>
    while (word.size() > 1) {
        int32_t best_rank = INT32_MAX, best_a = -1, best_b = -1, best_m = -1;
        for (size_t i = 0; i + 1 < word.size(); ++i) {
            auto it = b.merges.find(((uint64_t)(uint32_t)word[i] << 32) | (uint32_t)word[i + 1]);
            if (it != b.merges.end() && it->second.first < best_rank)
                best_rank = it->second.first, best_a = word[i], best_b = word[i + 1], best_m = it->second.second;
        }
        if (best_m < 0) break;
        size_t w = 0;
        for (size_t i = 0; i < word.size();) {   // every occurrence of the best pair, left to right
            if (i + 1 < word.size() && word[i] == best_a && word[i + 1] == best_b) {
                word[w++] = best_m;
                i += 2;
            } else {
                word[w++] = word[i++];
            }
        }
        word.resize(w);
    }
    for (int32_t s : word) out.push_back(b.vocab_id[s] >= 0 ? b.vocab_id[s] : b.eot);
}

void encode_one(const Bpe &b, const uint32_t *t, size_t n, int max_length, int32_t *out, int32_t *out_len,
                std::unordered_map<u32s, std::vector<int32_t>, U32Hash> &cache, std::vector<int32_t> &ids,
                std::vector<int32_t> &word) {
    ids.clear();
    const size_t cap = (size_t)(max_length > 2 ? max_length - 2 : 0);
    size_t i = 0;
    while (i < n && ids.size() < cap) {
        const uint32_t c = t[i];
        if (c == '<' && starts_with(t + i, n - i, SOT_S, 15)) {
            ids.push_back(b.sot);
            i += 15;
            continue;
        }
        if (c == '<' && starts_with(t + i, n - i, EOT_S, 13)) {
            ids.push_back(b.eot);
            i += 13;
            continue;
        }
        size_t len = 0;
        if (c == '\'' && i + 1 < n) {   // 's 't 're 've 'm 'll 'd
            const uint32_t d = t[i + 1], e = i + 2 < n ? t[i + 2] : 0;
            if (d == 's' || d == 't' || d == 'm' || d == 'd') len = 2;
            else if ((d == 'r' && e == 'e') || (d == 'v' && e == 'e') || (d == 'l' && e == 'l')) len = 3;
        }
        if (len == 0) {
            if (is_l(c)) {
                len = 1;
                while (i + len < n && is_l(t[i + len])) ++len;
            } else if (is_n(c)) {
                len = 1;
            } else if (!is_ws(c)) {
                len = 1;
                while (i + len < n && !is_ws(t[i + len]) && !is_l(t[i + len]) && !is_n(t[i + len])) ++len;
            } else {
                ++i;   // whitespace: between pieces
                continue;
            }
        }
        u32s key((const char32_t *)(t + i), len);
        auto it = cache.find(key);
        if (it == cache.end()) {
            std::vector<int32_t> piece;
            bpe_piece(b, t + i, len, piece, word);
            it = cache.emplace(std::move(key), std::move(piece)).first;
        }
        ids.insert(ids.end(), it->second.begin(), it->second.end());
        i += len;
    }
    if (ids.size() > cap) ids.resize(cap);
    int k = 0;
    out[k++] = b.sot;
    for (int32_t v : ids) out[k++] = v;
    out[k++] = b.eot;
    *out_len = k;
}

}  // namespace

using namespace mmrag;

extern "C" {

// vocab: `n_vocab` UTF-32 strings (entry i = vocab_cps[vocab_offsets[i] .. vocab_offsets[i+1]), id vocab_ids[i]);
// merges: `n_merges` UTF-32 strings "first second" in rank order.
void *mmrag_clip_bpe_create(const uint32_t *vocab_cps, const int64_t *vocab_offsets, const int32_t *vocab_ids,
                            int n_vocab, const uint32_t *merge_cps, const int64_t *merge_offsets, int n_merges) {
    if (!vocab_cps || !vocab_offsets || !vocab_ids || n_vocab <= 0 || n_merges < 0 || (n_merges > 0 && (!merge_cps || !merge_offsets))) {
        set_error("clip_bpe_create: empty vocabulary or null pointer");
        return nullptr;
    }
    Bpe *b = new Bpe();
    for (int i = 0; i < n_vocab; ++i) {
        const u32s s((const char32_t *)(vocab_cps + vocab_offsets[i]), (size_t)(vocab_offsets[i + 1] - vocab_offsets[i]));
        const int32_t id = b->intern(s);
        b->vocab_id[id] = vocab_ids[i];
    }
    for (int k = 0; k < 256; ++k) {
        u32s s(1, (char32_t)BYTE_TO_CHAR[k]);
        b->byte_sym[k] = b->intern(s);
        b->byte_sym_end[k] = b->intern(s + END_W);
    }
    for (int r = 0; r < n_merges; ++r) {
        const uint32_t *p = merge_cps + merge_offsets[r];
        const size_t n = (size_t)(merge_offsets[r + 1] - merge_offsets[r]);
        size_t sp = 0;
        while (sp < n && p[sp] != ' ') ++sp;
        if (sp == 0 || sp + 1 >= n) continue;   // not "first second": ignored, as the Python form does
        const u32s a((const char32_t *)p, sp), c((const char32_t *)(p + sp + 1), n - sp - 1);
        const int32_t ia = b->intern(a), ic = b->intern(c), im = b->intern(a + c);
        const uint64_t key = ((uint64_t)(uint32_t)ia << 32) | (uint32_t)ic;
        if (!b->merges.count(key)) b->merges.emplace(key, std::make_pair((int32_t)r, im));
    }
    auto special = [&](const char32_t *s) -> int32_t {
        auto it = b->sym.find(u32s(s));
        return it != b->sym.end() ? b->vocab_id[it->second] : -1;
    };
    b->sot = special(SOT_S);
    b->eot = special(EOT_S);
    if (b->sot < 0 || b->eot < 0) {
        set_error("clip_bpe_create: the vocabulary lacks <|startoftext|> / <|endoftext|>");
        delete b;
        return nullptr;
    }
    return b;
}

void mmrag_clip_bpe_destroy(void *tk) { delete (Bpe *)tk; }

// texts: n UTF-32 strings concatenated, already NFC-normalised, whitespace-collapsed and lower-cased.
// ids [n, max_length] int32 (rows are [sot] ... [eot], the rest untouched), lens [n].
int mmrag_clip_bpe_encode_batch(const void *tk, const uint32_t *cps, const int64_t *offsets, int n, int max_length,
                                int32_t *ids, int32_t *lens, int n_threads) {
    MMRAG_CHECK_ARG(tk && offsets && ids && lens, "clip_bpe_encode_batch: null pointer");
    MMRAG_CHECK_ARG(n >= 0 && max_length >= 2, "clip_bpe_encode_batch: bad shape n=%d max_length=%d", n, max_length);
    const Bpe &b = *(const Bpe *)tk;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n) n_threads = n > 0 ? n : 1;
    // a thread is worth starting for ~16 k code points (a 256-query batch of one-line queries is faster on ONE thread:
    // 0.28 ms against 0.53 ms on eight, all of the difference thread start-up)
    if (n > 0) {
        const int64_t by_work = 1 + (offsets[n] - offsets[0]) / 16384;
        if (n_threads > by_work) n_threads = (int)by_work;
    }
    auto work = [&](int lo, int hi) {
        std::unordered_map<u32s, std::vector<int32_t>, U32Hash> cache;
        std::vector<int32_t> tmp, word;
        for (int i = lo; i < hi; ++i)
            encode_one(b, cps + offsets[i], (size_t)(offsets[i + 1] - offsets[i]), max_length,
                       ids + (size_t)i * max_length, lens + i, cache, tmp, word);
    };
    if (n_threads == 1) {
        work(0, n);
        return MMRAG_OK;
    }
    std::vector<std::thread> th;
    const int per = (n + n_threads - 1) / n_threads;
    for (int t = 0; t < n_threads; ++t) {
        const int lo = t * per, hi = std::min(n, lo + per);
        if (lo < hi) th.emplace_back(work, lo, hi);
    }
    for (auto &t : th) t.join();
    return MMRAG_OK;
}

}  // extern "C"
