// Native WordPiece tokenizer (host, multi-threaded): the step in front of mmrag_encoder_forward.
//
// The reference tokenises inside SentenceTransformer.encode (app/utils/embedder.py:397-403), i.e. with
// Hugging Face `tokenizers` (native code).  The Python restatement in multimodal_rag_amd/tokenizer.py
// (pinned against transformers.BertTokenizer) does ~500 chunks/s per core, 30x below what one MI355X
// embeds, so ingest needs this: BERT BasicTokenizer (clean, CJK spacing, whitespace split, lower-case,
// NFD + strip Mn, punctuation split) + greedy longest-match WordPiece, on UTF-32 input, with a thread
// per slice of the batch.  Unicode data comes from unicode_tables.inc, generated from the same
// interpreter's unicodedata that tokenizer.py uses; tests/test_tokenizer.py fuzzes the two against
// each other.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "mmrag_internal.h"

namespace {

#include "unicode_tables.inc"

template <size_t N>
inline bool in_ranges(const uint32_t (&r)[N][2], uint32_t cp) {
    size_t lo = 0, hi = N;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (cp > r[mid][1])
            lo = mid + 1;
        else if (cp < r[mid][0])
            hi = mid;
        else
            return true;
    }
    return false;
}

template <size_t N, size_t W>
inline const uint32_t *find_map(const uint32_t (&m)[N][W], uint32_t cp) {
    size_t lo = 0, hi = N;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (m[mid][0] < cp)
            lo = mid + 1;
        else
            hi = mid;
    }
    return (lo < N && m[lo][0] == cp) ? m[lo] : nullptr;
}

inline bool is_cjk(uint32_t cp) {
    return (cp >= 0x4E00 && cp <= 0x9FFF) || (cp >= 0x3400 && cp <= 0x4DBF) || (cp >= 0x20000 && cp <= 0x2A6DF) ||
           (cp >= 0x2A700 && cp <= 0x2B73F) || (cp >= 0x2B740 && cp <= 0x2B81F) || (cp >= 0x2B820 && cp <= 0x2CEAF) ||
           (cp >= 0xF900 && cp <= 0xFAFF) || (cp >= 0x2F800 && cp <= 0x2FA1F);
}

inline bool is_punct(uint32_t cp) {
    if ((cp >= 33 && cp <= 47) || (cp >= 58 && cp <= 64) || (cp >= 91 && cp <= 96) || (cp >= 123 && cp <= 126)) return true;
    return cp >= 128 && in_ranges(RANGES_PUNCT, cp);
}

typedef std::vector<uint32_t> u32s;

// str.lower() of one word (context-free table + CPython's final-sigma rule)
void lower_word(const u32s &w, u32s &out) {
    out.clear();
    const size_t n = w.size();
    for (size_t i = 0; i < n; ++i) {
        const uint32_t c = w[i];
        if (c < 128) {
            out.push_back((c >= 'A' && c <= 'Z') ? c + 32 : c);
            continue;
        }
        if (c == 0x3A3) {
            long j = (long)i - 1;
            while (j >= 0 && in_ranges(RANGES_CASE_IGN, w[j])) --j;
            bool fin = j >= 0 && in_ranges(RANGES_CASED, w[j]);
            if (fin) {
                size_t k = i + 1;
                while (k < n && in_ranges(RANGES_CASE_IGN, w[k])) ++k;
                fin = k == n || !in_ranges(RANGES_CASED, w[k]);
            }
            out.push_back(fin ? 0x3C2 : 0x3C3);
            continue;
        }
        if (const uint32_t *m = find_map(MAP_LOWER, c)) {
            for (uint32_t t = 0; t < m[1]; ++t) out.push_back(m[2 + t]);
        } else if (const uint32_t *m4 = find_map(MAP_LOWER4, c)) {
            for (uint32_t t = 0; t < 4; ++t) out.push_back(m4[2 + t]);
        } else {
            out.push_back(c);
        }
    }
}

inline uint32_t ccc_of(uint32_t cp) {
    if (cp < 0x300) return 0;
    const uint32_t *m = find_map(MAP_CCC, cp);
    return m ? m[1] : 0;
}

// unicodedata.normalize("NFD", w) with the Mn characters dropped afterwards
void nfd_strip_mn(const u32s &w, u32s &out, u32s &tmp) {
    tmp.clear();
    for (uint32_t c : w) {
        if (c < 0xC0) {
            tmp.push_back(c);
        } else if (c >= 0xAC00 && c <= 0xD7A3) {  // Hangul syllable: algorithmic decomposition
            const uint32_t s = c - 0xAC00;
            tmp.push_back(0x1100 + s / 588);
            tmp.push_back(0x1161 + (s % 588) / 28);
            if (s % 28) tmp.push_back(0x11A7 + s % 28);
        } else if (const uint32_t *m = find_map(MAP_NFD, c)) {
            for (uint32_t t = 0; t < m[1]; ++t) tmp.push_back(m[2 + t]);
        } else if (const uint32_t *m4 = find_map(MAP_NFD4, c)) {
            for (uint32_t t = 0; t < 4; ++t) tmp.push_back(m4[2 + t]);
        } else {
            tmp.push_back(c);
        }
    }
    // canonical ordering: stable sort of every run of combining marks by combining class
    for (size_t i = 0; i < tmp.size();) {
        if (ccc_of(tmp[i]) == 0) {
            ++i;
            continue;
        }
        size_t j = i;
        while (j < tmp.size() && ccc_of(tmp[j]) != 0) ++j;
        if (j - i > 1)
            std::stable_sort(tmp.begin() + i, tmp.begin() + j, [](uint32_t a, uint32_t b) { return ccc_of(a) < ccc_of(b); });
        i = j;
    }
    out.clear();
    for (uint32_t c : tmp)
        if (!(c >= 0x300 && in_ranges(RANGES_MN, c))) out.push_back(c);
}

struct Vocab {
    // open-addressing table over (codepoint string -> id); `first` = whole entries, `cont` = entries that
    // start with "##", keyed by what follows the prefix
    struct Table {
        std::vector<uint32_t> pool;
        struct Slot {
            uint64_t hash;
            uint32_t off, len;
            int32_t id;
        };
        std::vector<Slot> slots;
        size_t mask = 0;
        static uint64_t hash_of(const uint32_t *p, size_t n) {
            uint64_t h = 1469598103934665603ull;
            for (size_t i = 0; i < n; ++i) {
                h ^= p[i];
                h *= 1099511628211ull;
            }
            return h | 1;  // 0 marks an empty slot
        }
        void build(const std::vector<std::pair<u32s, int32_t>> &items) {
            size_t cap = 16;
            while (cap < items.size() * 2 + 2) cap <<= 1;
            slots.assign(cap, Slot{0, 0, 0, -1});
            mask = cap - 1;
            for (const auto &it : items) {
                const uint64_t h = hash_of(it.first.data(), it.first.size());
                size_t i = h & mask;
                bool dup = false;
                while (slots[i].hash) {
                    if (slots[i].hash == h && slots[i].len == it.first.size() &&
                        !memcmp(&pool[slots[i].off], it.first.data(), it.first.size() * 4)) {
                        slots[i].id = it.second;  // a later duplicate line wins, as a Python dict would
                        dup = true;
                        break;
                    }
                    i = (i + 1) & mask;
                }
                if (dup) continue;
                slots[i] = Slot{h, (uint32_t)pool.size(), (uint32_t)it.first.size(), it.second};
                pool.insert(pool.end(), it.first.begin(), it.first.end());
            }
        }
        int32_t find(const uint32_t *p, size_t n) const {
            if (slots.empty()) return -1;
            const uint64_t h = hash_of(p, n);
            size_t i = h & mask;
            while (slots[i].hash) {
                if (slots[i].hash == h && slots[i].len == n && !memcmp(&pool[slots[i].off], p, n * 4)) return slots[i].id;
                i = (i + 1) & mask;
            }
            return -1;
        }
    };
    Table first, cont;
    int32_t cls = 101, sep = 102, unk = 100;
    int lower = 1;
    int max_chars = 100;
};

void wordpiece(const Vocab &v, const u32s &word, std::vector<int32_t> &ids) {
    const size_t n = word.size();
    if ((int)n > v.max_chars) {
        ids.push_back(v.unk);
        return;
    }
    const size_t mark = ids.size();
    size_t start = 0;
    while (start < n) {
        size_t end = n;
        int32_t cur = -1;
        while (start < end) {
            cur = (start == 0 ? v.first : v.cont).find(&word[start], end - start);
            if (cur >= 0) break;
            --end;
        }
        if (cur < 0) {
            ids.resize(mark);
            ids.push_back(v.unk);
            return;
        }
        ids.push_back(cur);
        start = end;
    }
}

void encode_one(const Vocab &v, const uint32_t *text, size_t n, int max_length, int32_t *out, int32_t *out_len) {
    std::vector<int32_t> ids;
    ids.reserve((size_t)max_length + 8);
    ids.push_back(v.cls);
    u32s word, low, norm, tmp, piece;
    const int stop = max_length - 1;
    auto flush_piece = [&]() {
        if (!piece.empty()) {
            wordpiece(v, piece, ids);
            piece.clear();
        }
    };
    auto flush_word = [&]() -> bool {  // false: enough ids
        if (word.empty()) return true;
        const u32s *w = &word;
        if (v.lower) {
            lower_word(word, low);
            nfd_strip_mn(low, norm, tmp);
            w = &norm;
        }
        // punctuation split.  tokenizer.py checks the id budget once per basic token.
        for (uint32_t c : *w) {
            if (is_punct(c)) {
                flush_piece();
                if ((int)ids.size() >= stop) break;
                piece.push_back(c);
                flush_piece();
                if ((int)ids.size() >= stop) break;
            } else {
                piece.push_back(c);
            }
        }
        if ((int)ids.size() < stop) flush_piece();
        piece.clear();
        word.clear();
        return (int)ids.size() < stop;
    };
    bool more = true;
    for (size_t i = 0; i < n && more; ++i) {
        const uint32_t c = text[i];
        if (c == 0 || c == 0xFFFD) continue;
        const bool ws = c == ' ' || c == '\t' || c == '\n' || c == '\r';
        if (!ws && (c < 0x20 || (c >= 0x7F && in_ranges(RANGES_CTRL, c)))) continue;  // Cc / Cf
        if (ws || (c >= 0xA0 && in_ranges(RANGES_ZS, c)) || c == 0x2028 || c == 0x2029 || c == 0x1C || c == 0x1D ||
            c == 0x1E || c == 0x1F || c == 0x85) {
            more = flush_word();
        } else if (is_cjk(c)) {
            more = flush_word();
            if (more) {
                word.push_back(c);
                more = flush_word();
            }
        } else {
            word.push_back(c);
        }
    }
    if (more) flush_word();
    int len = (int)ids.size();
    if (len > stop) len = stop;
    for (int i = 0; i < len; ++i) out[i] = ids[i];
    out[len] = v.sep;
    *out_len = len + 1;
}

}  // namespace

using namespace mmrag;

extern "C" {

// vocab: `n_tokens` UTF-32 strings concatenated in `cps`, token i = cps[offsets[i] .. offsets[i+1]); id = i.
void *mmrag_wordpiece_create(const uint32_t *cps, const int64_t *offsets, int n_tokens, int lower) {
    if (!cps || !offsets || n_tokens <= 0) {
        set_error("wordpiece_create: empty vocabulary");
        return nullptr;
    }
    Vocab *v = new Vocab();
    v->lower = lower;
    std::vector<std::pair<u32s, int32_t>> first, cont;
    first.reserve(n_tokens);
    for (int i = 0; i < n_tokens; ++i) {
        u32s t(cps + offsets[i], cps + offsets[i + 1]);
        if (t.size() > 2 && t[0] == '#' && t[1] == '#') cont.emplace_back(u32s(t.begin() + 2, t.end()), i);
        first.emplace_back(std::move(t), i);
    }
    v->first.build(first);
    v->cont.build(cont);
    auto special = [&](const char *s, int32_t dflt) {
        u32s k;
        for (const char *p = s; *p; ++p) k.push_back((uint32_t)*p);
        const int32_t id = v->first.find(k.data(), k.size());
        return id >= 0 ? id : dflt;
    };
    v->cls = special("[CLS]", 101);
    v->sep = special("[SEP]", 102);
    v->unk = special("[UNK]", 100);
    return v;
}

void mmrag_wordpiece_destroy(void *tk) { delete (Vocab *)tk; }

// texts: n UTF-32 strings concatenated (text i = cps[offsets[i] .. offsets[i+1])).
// ids [n, max_length] int32 (rows are [CLS] ... [SEP], the rest untouched), lens [n].
int mmrag_wordpiece_encode_batch(const void *tk, const uint32_t *cps, const int64_t *offsets, int n, int max_length,
                                 int32_t *ids, int32_t *lens, int n_threads) {
    MMRAG_CHECK_ARG(tk && offsets && ids && lens, "wordpiece_encode_batch: null pointer");
    MMRAG_CHECK_ARG(n >= 0 && max_length >= 2, "wordpiece_encode_batch: bad shape n=%d max_length=%d", n, max_length);
    const Vocab &v = *(const Vocab *)tk;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n) n_threads = n > 0 ? n : 1;
    // a thread is worth starting for ~16 k code points (a 256-query batch of one-line queries is faster on ONE thread:
    // 0.28 ms against 0.53 ms on eight, all of the difference thread start-up)
    if (n > 0) {
        const int64_t by_work = 1 + (offsets[n] - offsets[0]) / 16384;
        if (n_threads > by_work) n_threads = (int)by_work;
    }
    auto work = [&](int lo, int hi) {
        for (int i = lo; i < hi; ++i)
            encode_one(v, cps + offsets[i], (size_t)(offsets[i + 1] - offsets[i]), max_length,
                       ids + (size_t)i * max_length, lens + i);
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
