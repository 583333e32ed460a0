// Types shared by the two search translation units of libmmrag.so (search.hip: slab-ring kernel for every
// shape; search_qs.hip: query-stationary kernel for big query batches).  Internal, gfx950 only.
#pragma once
#include "mmrag_internal.h"
#include "tile_dma.h"

#include <limits.h>

namespace mmrag_impl {

constexpr float NEG_INF = -__builtin_inff();

__device__ inline bool better(float s, long long r, float s2, long long r2) {
    return s > s2 || (s == s2 && r < r2);
}

template <int K>
struct TopList {
    float v[K];
    int r[K];
    __device__ inline void init() {
#pragma unroll
        for (int i = 0; i < K; ++i) {
            v[i] = NEG_INF;
            r[i] = INT_MAX;
        }
    }
    // full (score desc, row asc) order: for merging lists whose rows interleave
    __device__ inline void insert_ordered(float x, int xr) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const bool b = x > v[j] || (x == v[j] && xr < r[j]);
            const float nv = b ? x : v[j];
            const float nx = b ? v[j] : x;
            const int nr = b ? xr : r[j];
            const int nxr = b ? r[j] : xr;
            v[j] = nv;
            x = nx;
            r[j] = nr;
            xr = nxr;
        }
    }
    // insertion order == row order inside a lane, so "strictly greater" keeps the lower row on ties
    __device__ inline void insert_strict(float x, int xr) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const bool b = x > v[j];
            const float nv = b ? x : v[j];
            const float nx = b ? v[j] : x;
            const int nr = b ? xr : r[j];
            const int nxr = b ? r[j] : xr;
            v[j] = nv;
            x = nx;
            r[j] = nr;
            xr = nxr;
        }
    }
    // the same result with every compare against the OLD list (the list is sorted, so "x > v[j]" is false ... false
    // true ... true): slot j keeps its entry, takes x, or takes its upper neighbour's.  Three dependent steps instead
    // of 2 K: for a wave alone on its SIMD the dependent chain, not the instruction count, is what an insertion costs.
    __device__ inline void insert_strict_flat(float x, int xr) {
        bool c[K];
#pragma unroll
        for (int j = 0; j < K; ++j) c[j] = x > v[j];
#pragma unroll
        for (int j = K - 1; j >= 1; --j) {
            v[j] = c[j] ? (c[j - 1] ? v[j - 1] : x) : v[j];
            r[j] = c[j] ? (c[j - 1] ? r[j - 1] : xr) : r[j];
        }
        v[0] = c[0] ? x : v[0];
        r[0] = c[0] ? xr : r[0];
    }
};

struct KParams {
    const char *q;        // [B, ld]
    const char *corpus;   // [n, ld]
    const uint32_t *alive_bits;
    float *cand_s;        // [Bpad, n_lists, K]
    int *cand_r;
    long long n;
    int B;
    unsigned row_bytes;   // ld * esize, multiple of 128
    int n_tiles;          // corpus tiles of this launch (tile height is the kernel's own)
    int n_lists;          // list slots per query in cand_* (>= walkers)
    int tile0;            // first corpus tile of this launch (sample pre-pass / main pass split)
    int walkers;          // workgroups that walk the tiles, per query group (grid = walkers * query groups)
    int share_l2;         // > 1 query group: corpus slabs are re-read by the sibling groups, keep them in L2
    const float *thr0;    // optional [B]: a known lower bound of each query's final k-th score
    unsigned dbg;         // timing-only switches (DBG_QS_*), reachable through the internal debug entry only
    unsigned long long *stamps;  // DBG_QS_CLOCK: per-workgroup (shader cycles, 100 MHz ticks) of the main loop
    float *sample_best;   // query-stationary kernel, sample pass: [b_pad][walkers] best score per query per workgroup
                          // (no candidate lists are written); null = a normal pass
    // ---- single-launch walk kernel (search_qsw.hip) ----
    float *pub_best;      // [query groups][walkers][256]: every workgroup's best score per query after its first tiles,
                          // -inf until published (the host fills the exchange block with -inf before the launch);
                          // null = no in-kernel threshold seeding
    float *pub_thr;       // [query groups][256]: K-th largest of a query's published bests, -inf until computed
    unsigned *tickets;    // [query groups]: next dynamically handed-out tile, counted from TICKET0
    int p_static;         // tile positions every workgroup walks by the static rule (tile = bx + pos * walkers); the
                          // tiles from p_static * walkers on are handed out by ticket.  Huge = no dynamic hand-out
};
constexpr unsigned TICKET0 = 0xff800000u;   // the exchange block is filled with the bit pattern of -inf

// timing-only ablations of the query-stationary kernel: results are WRONG with any of them set
constexpr unsigned DBG_QS_NO_SELECT = 16u, DBG_QS_NO_DMA = 32u, DBG_QS_NO_BARRIER = 64u, DBG_QS_DMA_L2 = 512u, DBG_QS_NO_WAIT = 1024u;
constexpr unsigned DBG_QS_CLOCK = 256u;  // (valid results) first 16 workgroups stamp their main loop: clock under load

// K-th largest of the (up to) 256 values a wave holds four per lane; every lane gets the result.  Used to turn the
// per-workgroup best scores of one query into a threshold: the values are scores of distinct rows, so K rows reach it.
template <int K>
__device__ inline float wave_kth_largest(float a, float b, float c, float d, int lane) {
    auto ce = [](float &x, float &y) {
        const float hi = fmaxf(x, y), lo = fminf(x, y);
        x = hi;
        y = lo;
    };
    ce(a, b);
    ce(c, d);
    ce(a, c);
    ce(b, d);
    ce(b, c);  // a >= b >= c >= d
    float res = NEG_INF;
    for (int r = 0; r < K; ++r) {
        float m = a;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        res = m;
        const unsigned long long owners = __builtin_amdgcn_ballot_w64(a == m);
        if (owners != 0ull && lane == (int)__builtin_ctzll(owners)) {  // one lane gives up its head
            a = b;
            b = c;
            c = d;
            d = NEG_INF;
        }
    }
    return res;
}

// ---- query-stationary kernel (search_qs.hip) -------------------------------------------------------------
// supported(): storage dtype, row bytes and list depth the kernel is instantiated for
bool qs_supported(int dtype, unsigned row_bytes, int K);
constexpr int QS_TILE_ROWS = 64;   // corpus rows per tile
constexpr int QS_QROWS = 256;      // queries per workgroup
// launches over `grid_x * grid_y` workgroups of 256 threads (walkers x query groups)
int qs_launch(int dtype, int K, const KParams &p, int grid_x, int grid_y, hipStream_t s);
// thr0[q] = K-th largest of best[q][0 .. walkers): the threshold a sample pass yields
int qs_seed_thresholds(int K, const float *best, int walkers, int n_queries, float *thr0, hipStream_t s);

// ---- single-launch walk kernel (search_qsw.hip): same tiles and candidate lists, thresholds seeded inside the launch ---
// mfma: 32 = v_mfma_f32_32x32x16, 16 = v_mfma_f32_16x16x32 (K = 5 only)
constexpr int QSW_DEFAULT_MFMA = 16;   // the shape list depth 5 runs on (A/B in DESIGN.md section 7)
bool qsw_supported(int dtype, unsigned row_bytes, int K, int mfma);
int qsw_launch(int dtype, int K, int mfma, const KParams &p, int grid_x, int grid_y, hipStream_t s);


}  // namespace mmrag_impl
