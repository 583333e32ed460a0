// Fused cosine GEMM + top-k for gfx950 (MI355X): the retrieve half of the hot path.
//
// Replaces chromadb's collection.query (reference call site app/utils/embedder.py:595-601)
// with an exact search:  S = corpus . Q^T  on the matrix cores, top-k selected in the GEMM
// epilogue, so the [B, n] score matrix never exists in HBM.
//
// Shape of the computation (DESIGN.md "search kernel"):
//   * persistent grid: one 512-thread workgroup per CU walks corpus tiles of 256 rows
//     (tile = blockIdx.x + i * gridDim.x); HBM traffic = the corpus, read exactly once;
//   * K is streamed in 128-byte slabs (64 fp16 / 32 fp32 per row) through an NSTAGE-deep
//     LDS ring filled by LDS-DMA (`buffer_load_dwordx4 ... lds`, bounds-checked by the
//     buffer descriptor so ragged tiles read zeros); the query slab rides along (L2-resident);
//   * LDS rows are 128 B; the 16-byte chunk index is XOR-swizzled with (row>>1)&7 on the
//     DMA *source* address and on the ds_read_b128 address (conflict-free fragment reads);
//   * MFMA orientation D[corpus row][query] = A(corpus) x B(Q^T): the query sits on the
//     lane (col = lane & 31), corpus rows sit in the 16 accumulator registers, so top-k
//     selection per query is lane-local register work with no cross-lane traffic;
//   * NW waves = WM x WN (8 waves, or 16 = 4 per SIMD for the 256-query shape: the extra waves
//     hide LDS-read and barrier latency); each wave owns 32 queries x (256/WM) corpus rows;
//   * selection: each lane keeps a sorted K-list (score desc) for its query over the rows it
//     sees; an element enters only if it beats a running threshold (k-th best of the union
//     of the two half-wave lists of that query) -- after the first tiles almost nothing does;
//   * masked / out-of-range rows start their accumulator at -inf, so they cost nothing in
//     the epilogue;
//   * every lane list is written once at kernel end; a second tiny kernel merges the
//     gridDim.x * WM * 2 lists per query into the final [B, k] (ties -> lower row).
#include "search_shared.h"

#include <stdlib.h>

using namespace mmrag;

namespace mmrag_impl {

constexpr int TM = 256;            // corpus rows per tile
constexpr int CORPUS_STAGE = TM * SLAB;  // 32 KiB

template <int DT, int WN, int K, int NSTAGE, int NW, bool SEEDED>
__global__ __launch_bounds__(64 * NW, NW / 4) void cosine_topk_kernel(const KParams p) {
#if defined(__HIP_DEVICE_COMPILE__)  // amdgcn builtins below: the host pass only needs the launch stub
    constexpr int WM = NW / WN;          // waves along corpus rows
    constexpr int RM = 8 / WM;           // 32-row blocks per wave  (== WN)
    constexpr int QROWS = 32 * WN;
    constexpr int STAGE = CORPUS_STAGE + QROWS * SLAB;
    constexpr int CLOADS = 32 / NW;      // 1 KiB DMA instructions per wave for the corpus slab
    constexpr int QLOADS = (4 * WN) / NW; // ... and for the Q slab
    constexpr int LOADS = CLOADS + QLOADS;  // per wave per ring item
    static_assert(CLOADS >= 1 && QLOADS >= 1 && CLOADS * NW == 32 && QLOADS * NW == 4 * WN, "piece split");
    static_assert(WN == 2 || WN == 4 || WN == 8, "WN");
    static_assert(NSTAGE >= 2 && NSTAGE <= 4, "NSTAGE");
    static_assert(NSTAGE * STAGE <= 160 * 1024, "LDS");

    __shared__ __attribute__((aligned(1024))) char smem[NSTAGE * STAGE];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int r32 = lane & 31;
    const int h = lane >> 5;

    const unsigned RB = p.row_bytes;
    const int nk = (int)(RB / SLAB);
    // 1-D grid of walkers * query_groups workgroups, query group major: the query groups of one walker sit
    // walkers (a multiple of 8) ids apart, i.e. on the SAME XCD, and stream the same corpus tiles at about
    // the same time -- one HBM fetch, the other groups hit that XCD's L2 (batches above 256 queries)
    const int walkers = p.walkers;
    const int bx = (int)blockIdx.x % walkers;
    const int by = (int)blockIdx.x / walkers;
    const int q0 = by * QROWS;
    const int my_tiles = (p.n_tiles - bx + walkers - 1) / walkers;
    const int n_items = my_tiles * nk;

    // ---- DMA descriptors ------------------------------------------------------------------
    const long long q_rows_left = (long long)p.B - q0;
    const unsigned q_bytes = (unsigned)((q_rows_left < QROWS ? q_rows_left : QROWS) * (long long)RB);
    const __amdgpu_buffer_rsrc_t rsrc_q =
        __builtin_amdgcn_make_buffer_rsrc((void *)(p.q + (size_t)q0 * RB), 0, q_bytes, 0x00020000);

    // per-lane source offsets (row * RB + swizzled chunk * 16) for this wave's DMA instructions
    const int dma_row = lane >> 3;        // row inside an 8-row (1 KiB) LDS piece
    const int dma_slot = lane & 7;        // 16-byte slot inside the 128-byte LDS row
    unsigned c_off[CLOADS];
#pragma unroll
    for (int i = 0; i < CLOADS; ++i) {
        const int row = (wave * CLOADS + i) * 8 + dma_row;
        c_off[i] = (unsigned)row * RB + (unsigned)((dma_slot ^ ((row >> 1) & 7)) * 16);
    }
    unsigned q_off[QLOADS];
#pragma unroll
    for (int i = 0; i < QLOADS; ++i) {
        const int row = (wave * QLOADS + i) * 8 + dma_row;
        q_off[i] = (unsigned)row * RB + (unsigned)((dma_slot ^ ((row >> 1) & 7)) * 16);
    }

    int is_tile = 0, is_k = 0;  // issue cursor: (index among my tiles, k slab)
    auto issue = [&](int stage_idx) {
        const long long tile = (long long)p.tile0 + bx + (long long)is_tile * walkers;
        const long long row0 = tile * TM;
        const long long rows_left = p.n - row0;
        const unsigned c_bytes = (unsigned)((rows_left < TM ? rows_left : (long long)TM) * (long long)RB);
        const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(p.corpus + (size_t)row0 * RB), 0, c_bytes, 0x00020000);
        char *st = smem + stage_idx * STAGE;
        const unsigned koff = (unsigned)is_k * SLAB;
        // corpus rows are read exactly once, by this CU only: non-temporal (aux = 2) keeps them from
        // displacing the query slab that every workgroup re-reads from L2 (A/B: -2 % at B=256, -5 % at B<=128)
        if (p.share_l2) {
#pragma unroll
            for (int i = 0; i < CLOADS; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_c, (lds_ptr_t)(st + (wave * CLOADS + i) * 1024), 16,
                                                         c_off[i] + koff, 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < CLOADS; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_c, (lds_ptr_t)(st + (wave * CLOADS + i) * 1024), 16,
                                                         c_off[i] + koff, 0, 0, 2);
        }
#pragma unroll
        for (int i = 0; i < QLOADS; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsrc_q, (lds_ptr_t)(st + CORPUS_STAGE + (wave * QLOADS + i) * 1024), 16, q_off[i] + koff, 0,
                0, 0);
        if (++is_k == nk) {
            is_k = 0;
            ++is_tile;
        }
    };

    // ---- accumulators, lists ---------------------------------------------------------------
    f32x16_t acc[RM];
    auto init_acc = [&](int tile_idx) {
        const long long tile = (long long)p.tile0 + bx + (long long)tile_idx * walkers;
        const long long row0 = tile * TM + (long long)wm * (RM * 32);
        const bool ragged = row0 + RM * 32 > p.n;
        if (!ragged && p.alive_bits == nullptr) {
#pragma unroll
            for (int b = 0; b < RM; ++b)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[b][j] = 0.0f;
            return;
        }
#pragma unroll
        for (int b = 0; b < RM; ++b) {
            const long long brow = row0 + b * 32;
            unsigned bits = 0xffffffffu;
            if (p.alive_bits != nullptr && brow < p.n) bits = p.alive_bits[brow >> 5];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int rr = (j & 3) + 8 * (j >> 2) + 4 * h;
                const bool ok = (brow + rr < p.n) && ((bits >> rr) & 1u);
                acc[b][j] = ok ? 0.0f : NEG_INF;
            }
        }
    };

    TopList<K> best;
    best.init();
    float thr = NEG_INF;
    if (q0 + wn * 32 + r32 >= p.B)
        thr = INFINITY;  // padding query slot: its all-zero scores must never open the insertion path
    else if (p.thr0 != nullptr)
        thr = p.thr0[q0 + wn * 32 + r32];

    // ---- fragment addresses (bytes inside a stage) ------------------------------------------
    const int sw = (r32 >> 1) & 7;
    const int a_base = (wm * RM * 32 + r32) * SLAB;
    const int b_base = CORPUS_STAGE + (wn * 32 + r32) * SLAB;

    // ---- prologue ---------------------------------------------------------------------------
    int issued = 0;
    for (; issued < NSTAGE - 1 && issued < n_items; ++issued) issue(issued);
    init_acc(0);

    int tile_idx = 0, ks = 0;
    for (int it = 0; it < n_items; ++it) {
        wait_items<LOADS, NSTAGE - 2>(issued - it - 1);
        __builtin_amdgcn_s_barrier();
        if (issued < n_items) {
            issue(issued % NSTAGE);
            ++issued;
        }
        const char *st = smem + (it % NSTAGE) * STAGE;

        if constexpr (DT == MMRAG_F32 && WN >= 4) {
            // fp32 storage, more than 64 queries (BASELINE config 2 at B = 256): the exact f32 MFMA runs at 1/16 of
            // the bf16 rate and this shape is matrix-bound, so every operand fragment is split on the fly into two
            // bf16 terms (x = hi + lo up to 2^-17 |x|) and the product is taken as hi*hi + hi*lo + lo*hi in fp32
            // accumulators: |score error| <= 3 * 2^-17 * sum|q_i c_i| + 2^-16 <= 4e-5 for unit vectors, inside the
            // 1e-4 parity bound (tests/test_search_gpu.py::test_fp32_split_*), at 3/16 of the f32-MFMA cost.
            // Batches of <= 64 queries are HBM- / launch-bound and keep the exact f32 MFMA below.
            auto split = [&](const char *at0, const char *at1, bf16x8_t &hi, bf16x8_t &lo) {
                const f32x4_t x0 = *(const f32x4_t *)at0, x1 = *(const f32x4_t *)at1;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float x = i < 4 ? x0[i] : x1[i - 4];
                    const __bf16 hb = (__bf16)x;
                    hi[i] = hb;
                    lo[i] = (__bf16)(x - (float)hb);
                }
            };
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {          // two 16-deep bf16 k-steps per 128-byte fp32 slab
                const int o0 = ((4 * s2 + 2 * h) ^ sw) * 16, o1 = ((4 * s2 + 2 * h + 1) ^ sw) * 16;
                bf16x8_t qh, ql;
                split(st + b_base + o0, st + b_base + o1, qh, ql);
#pragma unroll
                for (int b = 0; b < RM; ++b) {
                    bf16x8_t ch, cl;
                    split(st + a_base + b * (32 * SLAB) + o0, st + a_base + b * (32 * SLAB) + o1, ch, cl);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cl, qh, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch, ql, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch, qh, acc[b], 0, 0, 0);
                }
            }
        } else if constexpr (DT == MMRAG_F32) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int off = ((2 * m + h) ^ sw) * 16;
                const f32x4_t bq = *(const f32x4_t *)(st + b_base + off);
#pragma unroll
                for (int b = 0; b < RM; ++b) {
                    const f32x4_t ac = *(const f32x4_t *)(st + a_base + b * (32 * SLAB) + off);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[i], bq[i], acc[b], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int off = ((2 * m + h) ^ sw) * 16;
                if constexpr (DT == MMRAG_F16) {
                    const half8_t bq = *(const half8_t *)(st + b_base + off);
#pragma unroll
                    for (int b = 0; b < RM; ++b) {
                        const half8_t ac = *(const half8_t *)(st + a_base + b * (32 * SLAB) + off);
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ac, bq, acc[b], 0, 0, 0);
                    }
                } else {
                    const bf16x8_t bq = *(const bf16x8_t *)(st + b_base + off);
#pragma unroll
                    for (int b = 0; b < RM; ++b) {
                        const bf16x8_t ac = *(const bf16x8_t *)(st + a_base + b * (32 * SLAB) + off);
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ac, bq, acc[b], 0, 0, 0);
                    }
                }
            }
        }

        if (++ks == nk) {
            // ---- epilogue: lane-local top-K over this wave's 32*RM rows of the tile ----------
            ks = 0;
            const long long tile = (long long)p.tile0 + bx + (long long)tile_idx * walkers;
            const int row_base = (int)(tile * TM) + wm * (RM * 32) + 4 * h;
            if constexpr (K > 5) {
                // deep lists: one (not unrolled) insertion body per 4-row group keeps the code small
                // (a K=20 insertion is ~100 instructions; 128 unrolled copies would not fit the I-cache)
#pragma unroll
                for (int b = 0; b < RM; ++b) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float a0 = acc[b][4 * g], a1 = acc[b][4 * g + 1];
                        const float a2 = acc[b][4 * g + 2], a3 = acc[b][4 * g + 3];
                        if (__builtin_amdgcn_ballot_w64(fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)) >= thr) != 0ull) {
#pragma clang loop unroll(disable)
                            for (int i = 0; i < 4; ++i) {
                                const float s = i == 0 ? a0 : (i == 1 ? a1 : (i == 2 ? a2 : a3));
                                const bool pass = s >= thr;
                                if (__builtin_amdgcn_ballot_w64(pass) != 0ull) {
                                    best.insert_strict(pass ? s : NEG_INF, row_base + b * 32 + i + 8 * g);
                                    thr = fmaxf(thr, best.v[K - 1]);
                                }
                            }
                        }
                    }
                }
            } else if constexpr (SEEDED) {
                // thresholds are warm from the first element (sample pre-pass): test 4 rows at a time
#pragma unroll
                for (int b = 0; b < RM; ++b) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float m4 = fmaxf(fmaxf(acc[b][4 * g], acc[b][4 * g + 1]),
                                               fmaxf(acc[b][4 * g + 2], acc[b][4 * g + 3]));
                        if (__builtin_amdgcn_ballot_w64(m4 >= thr) != 0ull) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float s = acc[b][4 * g + i];
                                const bool pass = s >= thr;
                                if (__builtin_amdgcn_ballot_w64(pass) != 0ull) {
                                    best.insert_strict(pass ? s : NEG_INF, row_base + b * 32 + i + 8 * g);
                                    thr = fmaxf(thr, best.v[K - 1]);
                                }
                            }
                        }
                    }
                }
            } else {
#pragma unroll
                for (int b = 0; b < RM; ++b) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const float s = acc[b][j];
                        const bool pass = s >= thr;
                        if (__builtin_amdgcn_ballot_w64(pass) != 0ull) {
                            best.insert_strict(pass ? s : NEG_INF, row_base + b * 32 + (j & 3) + 8 * (j >> 2));
                            thr = fmaxf(thr, best.v[K - 1]);
                        }
                    }
                }
            }
            // k-th best of the union of the two half-wave lists of this query: a lower bound
            // on the final k-th score, shared by both lanes
            float u = fmaxf(best.v[K - 1], __shfl_xor(best.v[K - 1], 32));
#pragma unroll
            for (int i = 0; i + 1 < K; ++i) u = fmaxf(u, fminf(best.v[i], __shfl_xor(best.v[K - 2 - i], 32)));
            thr = fmaxf(thr, u);
            ++tile_idx;
            if (tile_idx < my_tiles) init_acc(tile_idx);
        }
    }

    // ---- merge the workgroup's WM*2 lists per query through LDS, write ONE list per query -------
    {
        constexpr int NL = WM * 2;  // lists per query inside this workgroup
        static_assert(QROWS * NL * K * 8 <= NSTAGE * STAGE, "list merge scratch must fit in the ring");
        __builtin_amdgcn_s_barrier();  // every wave is done reading the ring
        float *ls = (float *)smem;
        int *lr = (int *)(smem + QROWS * NL * K * 4);
        const int ql = wn * 32 + r32;
        const int slot = (ql * NL + wm * 2 + h) * K;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            ls[slot + i] = best.v[i];
            lr[slot + i] = best.r[i];
        }
        __syncthreads();
        const int t = threadIdx.x;
        if (t < QROWS && q0 + t < p.B) {
            // every list is sorted (score desc, row asc) with its empty slots last: list 0 is taken as
            // it stands, and a list is left at its first entry that does not make the merged top-K
            TopList<K> m;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                m.v[i] = ls[t * NL * K + i];
                m.r[i] = lr[t * NL * K + i];
            }
            for (int l = 1; l < NL; ++l) {
                for (int i = 0; i < K; ++i) {
                    const float x = ls[(t * NL + l) * K + i];
                    const int xr = lr[(t * NL + l) * K + i];
                    if (xr == INT_MAX || !better(x, xr, m.v[K - 1], m.r[K - 1])) break;
                    m.insert_ordered(x, xr);
                }
            }
            const size_t base = ((size_t)(q0 + t) * p.n_lists + bx) * K;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                p.cand_s[base + i] = m.v[i];
                p.cand_r[base + i] = m.r[i];
            }
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

// ---------------------------------------------------------------------------------------------
// merge: per query, reduce `n_cand` (score, row) candidates to the top-k under (score desc,
// row asc).  One 256-thread workgroup per query.  Candidate c of query q lives at
//   (c / inner) * outer_stride + q * q_stride + (c % inner)
// which covers both the kernel's [B, n_lists*K] lists (inner = n_cand) and an all-gathered
// [G, B, k] block (inner = k, outer_stride = B*k, q_stride = k).
// ---------------------------------------------------------------------------------------------
template <int K>
struct FullList {
    float v[K];
    long long r[K];
    __device__ inline void init() {
#pragma unroll
        for (int i = 0; i < K; ++i) {
            v[i] = NEG_INF;
            r[i] = LLONG_MAX;
        }
    }
    __device__ inline void insert(float x, long long xr) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const bool b = better(x, xr, v[j], r[j]);
            const float nv = b ? x : v[j];
            const float nx = b ? v[j] : x;
            const long long nr = b ? xr : r[j];
            const long long nxr = b ? r[j] : xr;
            v[j] = nv;
            x = nx;
            r[j] = nr;
            xr = nxr;
        }
    }
    __device__ inline void pop() {
#pragma unroll
        for (int j = 0; j + 1 < K; ++j) {
            v[j] = v[j + 1];
            r[j] = r[j + 1];
        }
        v[K - 1] = NEG_INF;
        r[K - 1] = LLONG_MAX;
    }
};

// K rounds of wave-wide arg-best over the lanes' list heads; lane 0 receives the results
template <int K>
__device__ inline void wave_extract(FullList<K> &l, float *out_v, long long *out_r) {
#pragma unroll
    for (int round = 0; round < K; ++round) {
        float s = l.v[0];
        long long r = l.r[0];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float os = __shfl_xor(s, off);
            const long long orr = __shfl_xor(r, off);
            if (better(os, orr, s, r)) {
                s = os;
                r = orr;
            }
        }
        out_v[round] = s;
        out_r[round] = r;
        if (l.v[0] == s && l.r[0] == r) l.pop();
    }
}

template <int K, typename RowT>
__global__ __launch_bounds__(256) void merge_topk_kernel(const float *__restrict__ cs,
                                                          const RowT *__restrict__ cr, long long n_cand,
                                                          long long inner, long long outer_stride,
                                                          long long q_stride, int k_out,
                                                          long long row_offset, float *__restrict__ out_s,
                                                          long long *__restrict__ out_r,
                                                          float *__restrict__ thr0, float *__restrict__ seed_s,
                                                          int *__restrict__ seed_r, long long seed_q_stride) {
    __shared__ float sh_v[4 * K];
    __shared__ long long sh_r[4 * K];
    const int q = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    constexpr long long PAD = sizeof(RowT) == 4 ? (long long)INT_MAX : LLONG_MAX;

    FullList<K> l;
    l.init();
    for (long long c = threadIdx.x; c < n_cand; c += 256) {
        const long long a = (c / inner) * outer_stride + (long long)q * q_stride + (c % inner);
        const float s = cs[a];
        long long r = (long long)cr[a];
        if (r == PAD || r < 0) continue;  // padding entry
        if (better(s, r, l.v[K - 1], l.r[K - 1])) l.insert(s, r);
    }
    float ov[K];
    long long orr[K];
    wave_extract<K>(l, ov, orr);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < K; ++i) {
            sh_v[wave * K + i] = ov[i];
            sh_r[wave * K + i] = orr[i];
        }
    }
    __syncthreads();
    if (wave == 0) {
        l.init();
        for (int i = lane; i < 4 * K; i += 64)
            if (sh_r[i] != LLONG_MAX) l.insert(sh_v[i], sh_r[i]);
        wave_extract<K>(l, ov, orr);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < K; ++i) {
                if (i < k_out) {
                    const bool valid = orr[i] != LLONG_MAX && ov[i] > NEG_INF;
                    out_s[(size_t)q * k_out + i] = valid ? ov[i] : NEG_INF;
                    out_r[(size_t)q * k_out + i] = valid ? orr[i] + row_offset : -1;
                }
            }
            if (thr0 != nullptr) {
                // sample pre-pass: thr0[q] = k-th score of the sample's exact top-K (a valid lower bound of the
                // final k-th score; -inf when the sample held fewer than K live rows), and the sample's top-K
                // becomes one more candidate list of the final merge (seed_* points at that list slot)
                const bool full = orr[K - 1] != LLONG_MAX && ov[K - 1] > NEG_INF;
                thr0[q] = full ? ov[K - 1] : NEG_INF;
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    const bool valid = orr[i] != LLONG_MAX && ov[i] > NEG_INF;
                    seed_s[(size_t)q * seed_q_stride + i] = valid ? ov[i] : NEG_INF;
                    seed_r[(size_t)q * seed_q_stride + i] = valid ? (int)orr[i] : INT_MAX;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------------------
struct Plan {
    int K;        // list depth: 5, 10 or 20
    int WN;       // waves along queries
    int NW;       // waves per workgroup
    int grid_x, grid_y;
    int n_tiles;
    int n_lists;
    int b_pad;
    int pre_tiles;  // > 0: a sample pre-pass over the first pre_tiles tiles seeds the selection thresholds
    bool qs_ok;     // shape goes to the query-stationary kernel (search_qs.hip) when dtype / row length allow
    bool qs_room;   // ... could go there (debug switch DBG_FORCE_QS): the workspace keeps room for its sample pass
};

// debug switches of mmrag_internal_cosine_topk_lists_ex (tests and A/B tools only; never set by the product)
constexpr unsigned DBG_NO_PREPASS = 1u, DBG_8_WAVES = 2u, DBG_NO_QS = 4u, DBG_FORCE_QS = 8u;
// query-stationary shapes: the three-launch plan of search_qs.hip instead of the single-launch walk (A/B, tests); the
// walk's MFMA shape forced to 32x32x16 / 16x16x32; no dynamic tile hand-out; no in-kernel threshold seeding
constexpr unsigned DBG_OLD_QS = 0x10000u, DBG_MFMA32 = 0x20000u, DBG_MFMA16 = 0x40000u, DBG_NO_DYN = 0x80000u,
                   DBG_NO_SEED = 0x100000u;

// n_lists, b_pad and pre_tiles depend on (B, n, k) only: mmrag_cosine_topk_select and the workspace query
// have no dtype, so both kernels keep the same list layout
Plan make_plan(int B, long long n, int k, unsigned dbg = 0) {
    Plan pl;
    pl.K = k <= 5 ? 5 : (k <= 10 ? 10 : 20);
    pl.WN = B <= 64 ? 2 : (B <= 128 ? 4 : 8);
    const int qrows = 32 * pl.WN;
    pl.grid_y = (B + qrows - 1) / qrows;
    pl.n_tiles = (int)((n + TM - 1) / TM);
    const int cus = num_cus();
    pl.grid_x = pl.n_tiles < cus ? pl.n_tiles : cus;
    if (pl.grid_y > 1 && pl.n_tiles >= cus) {
        // all query groups of a tile resident together: cus / grid_y walkers, rounded to whole XCD rounds
        int w = cus / pl.grid_y / 8 * 8;
        pl.grid_x = w >= 8 ? w : (cus / pl.grid_y > 0 ? cus / pl.grid_y : 1);
    }
    if (pl.grid_x < 1) pl.grid_x = 1;
    pl.NW = (pl.K == 5 && pl.WN == 8 && !(dbg & DBG_8_WAVES)) ? 16 : 8;  // 4 waves/SIMD hide LDS + barrier latency
    // Sample pre-pass (256-query shape, and every shape with deep lists: there the epilogue, not HBM, is
    // what thresholds relieve).
    // The exact top-K of the first `cus` tiles gives every query a threshold as tight as if each
    // workgroup had already seen 65k rows, so the main pass almost never takes the insertion path.
    pl.pre_tiles = 0;
    const bool want_pre = (pl.WN == 8 || pl.K > 5) && pl.n_tiles >= 3 * cus;
    if (want_pre && !(dbg & DBG_NO_PREPASS)) pl.pre_tiles = cus;
    pl.n_lists = pl.grid_x + (want_pre ? 1 : 0);  // one merged list per workgroup per query (+ the sample's)
    pl.b_pad = pl.grid_y * qrows;
    // query-stationary kernels: more than 128 queries and a long shard.  They need >= 256 rows per walker to run at
    // all (qs_room).  The single-launch walk (list depth 5) pays from 3.75 tiles of 256 rows per CU
    // (profiles/r03_config_sweep.txt, walk vs slab-ring: 1M rows 379 vs 467 us, 500k 224 vs 258, 250k 154 vs 158,
    // 125k 112 vs 99); the three-launch plan (depth 10) from about 6 (profiles/r02_config_sweep.txt: its launches have
    // ~35 us of fixed cost each).
    pl.qs_room = pl.WN == 8 && pl.n_tiles >= cus;
    const bool long_enough = pl.K <= 5 ? pl.n_tiles * 4 >= 15 * cus : pl.n_tiles >= 6 * cus;
    pl.qs_ok = pl.qs_room && !(dbg & DBG_NO_QS) && (long_enough || (dbg & DBG_FORCE_QS));
    return pl;
}

template <int DT, int WN, int K, int NSTAGE, int NW = 8, bool SEEDED = false>
void launch_main(const KParams &p, dim3 grid, hipStream_t s) {
    KParams kp = p;
    kp.walkers = (int)grid.x;
    kp.share_l2 = grid.y > 1;
    cosine_topk_kernel<DT, WN, K, NSTAGE, NW, SEEDED><<<grid.x * grid.y, 64 * NW, 0, s>>>(kp);
}

// deep lists (K = 10, 20): 8 waves (2 per SIMD, 256 registers each) hold the lists next to the accumulators
template <int DT, int K>
void dispatch_deep(const Plan &pl, const KParams &p, dim3 grid, hipStream_t s) {
    if (pl.WN == 2) launch_main<DT, 2, K, 3>(p, grid, s);
    else if (pl.WN == 4) launch_main<DT, 4, K, 3>(p, grid, s);
    else if (p.thr0) launch_main<DT, 8, K, 2, 8, true>(p, grid, s);
    else launch_main<DT, 8, K, 2>(p, grid, s);
}

template <int DT>
int dispatch_main(const Plan &pl, const KParams &p, hipStream_t s) {
    dim3 grid(pl.grid_x, pl.grid_y);
    if (pl.K == 5) {
        if (pl.WN == 2) launch_main<DT, 2, 5, 3>(p, grid, s);
        else if (pl.WN == 4) launch_main<DT, 4, 5, 3>(p, grid, s);
        else if (pl.NW == 16 && p.thr0) launch_main<DT, 8, 5, 2, 16, true>(p, grid, s);
        else if (pl.NW == 16) launch_main<DT, 8, 5, 2, 16>(p, grid, s);
        else if (p.thr0) launch_main<DT, 8, 5, 2, 8, true>(p, grid, s);
        else launch_main<DT, 8, 5, 2>(p, grid, s);
    } else if (pl.K == 10) {
        dispatch_deep<DT, 10>(pl, p, grid, s);
    } else {
        dispatch_deep<DT, 20>(pl, p, grid, s);
    }
    return MMRAG_OK;
}

template <typename RowT>
int launch_merge(int K, const float *cs, const RowT *cr, long long n_cand, long long inner,
                 long long outer_stride, long long q_stride, int B, int k_out, long long row_offset,
                 float *out_s, long long *out_r, hipStream_t s, float *thr0 = nullptr, float *seed_s = nullptr,
                 int *seed_r = nullptr, long long seed_q_stride = 0) {
    if (K == 5)
        merge_topk_kernel<5, RowT><<<B, 256, 0, s>>>(cs, cr, n_cand, inner, outer_stride, q_stride, k_out,
                                                     row_offset, out_s, out_r, thr0, seed_s, seed_r, seed_q_stride);
    else if (K == 10)
        merge_topk_kernel<10, RowT><<<B, 256, 0, s>>>(cs, cr, n_cand, inner, outer_stride, q_stride, k_out,
                                                      row_offset, out_s, out_r, thr0, seed_s, seed_r, seed_q_stride);
    else
        merge_topk_kernel<20, RowT><<<B, 256, 0, s>>>(cs, cr, n_cand, inner, outer_stride, q_stride, k_out,
                                                      row_offset, out_s, out_r, thr0, seed_s, seed_r, seed_q_stride);
    return MMRAG_OK;
}

__global__ void fill_empty_kernel(float *s, long long *r, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        s[i] = NEG_INF;
        r[i] = -1;
    }
}

__global__ void fill_seed_empty_kernel(float *s, int *r, long long q_stride, int K, long long n_queries) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n_queries)
        for (int i = 0; i < K; ++i) {
            s[q * q_stride + i] = NEG_INF;
            r[q * q_stride + i] = INT_MAX;
        }
}

}  // namespace mmrag_impl
using namespace mmrag_impl;

extern "C" {

// workspace: [cand_s entries f32][cand_r entries i32][thr0 b_pad f32][sample top_s b_pad*K f32][pad][sample top_r b_pad*K i64]
struct WsLayout {
    size_t entries, off_r, off_thr, off_ts, off_tr, off_cnt, cnt_bytes, off_pub, off_pthr, off_tk, xch_end, total;
};
static WsLayout ws_layout(const Plan &pl) {
    WsLayout w;
    w.entries = (size_t)pl.b_pad * pl.n_lists * pl.K;
    w.off_r = w.entries * sizeof(float);
    w.off_thr = w.off_r + w.entries * sizeof(int);
    w.off_ts = w.off_thr + (size_t)pl.b_pad * sizeof(float);
    w.off_tr = (w.off_ts + (size_t)pl.b_pad * pl.K * sizeof(float) + 15) / 16 * 16;
    // query-stationary kernel: the per-workgroup bests of its sample pass
    w.off_cnt = (w.off_tr + (size_t)pl.b_pad * pl.K * sizeof(long long) + 63) / 64 * 64;
    w.cnt_bytes = 0;
    w.off_pub = w.off_cnt + w.cnt_bytes;
    // single-launch walk: the exchange block [bests b_pad x walkers | thresholds b_pad | tickets per query group],
    // filled with the bit pattern of -inf before every launch (the three-launch plan uses the first part only)
    w.off_pthr = w.off_pub + (pl.qs_room ? (size_t)pl.b_pad * pl.grid_x * sizeof(float) : 0);
    w.off_tk = w.off_pthr + (pl.qs_room ? (size_t)pl.b_pad * sizeof(float) : 0);
    w.xch_end = w.off_tk + (pl.qs_room ? ((size_t)pl.grid_y * sizeof(unsigned) + 63) / 64 * 64 : 0);
    w.total = w.xch_end;
    return w;
}

size_t mmrag_cosine_topk_workspace_bytes(int B, int64_t n, int k) {
    if (B <= 0 || n < 0 || k < 1 || k > MMRAG_MAX_K) return 0;
    return ws_layout(make_plan(B, n, k)).total + 256;
}

static int check_search_args(const void *q, const void *corpus, int B, int64_t n, int d, int64_t ld, int dtype,
                             int k) {
    MMRAG_CHECK_ARG(dtype >= 0 && dtype <= 2, "cosine_topk: bad dtype %d", dtype);
    MMRAG_CHECK_ARG(B > 0, "cosine_topk: B must be positive (got %d)", B);
    MMRAG_CHECK_ARG(k >= 1 && k <= MMRAG_MAX_K, "cosine_topk: k=%d outside 1..%d", k, MMRAG_MAX_K);
    MMRAG_CHECK_ARG(n >= 0 && n < (int64_t)INT_MAX - TM, "cosine_topk: n=%lld out of range", (long long)n);
    MMRAG_CHECK_ARG(d > 0 && ld >= d, "cosine_topk: need 0 < d <= ld (d=%d ld=%lld)", d, (long long)ld);
    const int64_t row_bytes = ld * esize(dtype);
    MMRAG_CHECK_ARG(row_bytes % SLAB == 0,
                    "cosine_topk: row bytes %lld not a multiple of %d (use mmrag_padded_dim)", (long long)row_bytes, SLAB);
    MMRAG_CHECK_ARG(row_bytes * TM < (int64_t)UINT_MAX, "cosine_topk: rows too long");
    MMRAG_CHECK_ARG(q, "cosine_topk: null q");
    MMRAG_CHECK_ARG(n == 0 || corpus, "cosine_topk: null corpus");
    MMRAG_CHECK_ARG(((uintptr_t)q % 16) == 0 && ((uintptr_t)corpus % 16) == 0,
                    "cosine_topk: q/corpus must be 16-byte aligned");
    return MMRAG_OK;
}

// mmrag_cosine_topk_lists with debug switches (DBG_*): kernel-shape A/B runs and the tests that pin every code
// path against the oracle.  Exported for them, deliberately absent from include/mmrag.h.
int mmrag_internal_cosine_topk_lists_ex(const void *q, const void *corpus, int B, int64_t n, int d, int64_t ld,
                                        int dtype, int k, const uint32_t *alive_bits, void *workspace,
                                        size_t workspace_bytes, void *stream, unsigned dbg) {
    if (int st = check_search_args(q, corpus, B, n, d, ld, dtype, k)) return st;
    if (n == 0) return MMRAG_OK;
    const Plan pl = make_plan(B, n, k, dbg);
    const WsLayout wl = ws_layout(pl);
    if (!workspace || workspace_bytes < wl.total) {
        set_error("cosine_topk: workspace %zu bytes < required %zu", workspace_bytes, wl.total);
        return MMRAG_EWORKSPACE;
    }
    MMRAG_CHECK_ARG(((uintptr_t)workspace % 16) == 0, "cosine_topk: workspace must be 16-byte aligned");

    KParams p;
    p.q = (const char *)q;
    p.corpus = (const char *)corpus;
    p.alive_bits = alive_bits;
    p.cand_s = (float *)workspace;
    p.cand_r = (int *)((char *)workspace + wl.off_r);
    p.n = n;
    p.B = B;
    p.row_bytes = (unsigned)(ld * esize(dtype));
    p.n_tiles = pl.n_tiles;
    p.n_lists = pl.n_lists;
    p.tile0 = 0;
    p.thr0 = nullptr;
    p.dbg = dbg;
    // (per workgroup: 8 words in the three-launch kernel, 32 in the walk kernel -- room for the larger)
    p.stamps = ((dbg & DBG_QS_CLOCK) && workspace_bytes >= wl.total + 256 * (size_t)(pl.grid_x * pl.grid_y)) ? (unsigned long long *)((char *)workspace + wl.total) : nullptr;
    p.walkers = pl.grid_x;
    p.share_l2 = pl.grid_y > 1;
    p.sample_best = nullptr;
    p.pub_best = nullptr;
    p.pub_thr = nullptr;
    p.tickets = nullptr;
    p.p_static = INT_MAX / 2;
    hipStream_t s = (hipStream_t)stream;
    const size_t seed_off = (size_t)(pl.n_lists - 1) * pl.K;  // the sample's list: last slot of every query
    const int walk_mfma = (dbg & DBG_MFMA32) ? 32 : ((dbg & DBG_MFMA16) ? 16 : (pl.K == 5 ? QSW_DEFAULT_MFMA : 32));
    if (pl.qs_ok && !(dbg & DBG_OLD_QS) && qsw_supported(dtype, p.row_bytes, pl.K, walk_mfma)) {
        // single-launch walk (search_qsw.hip): thresholds are exchanged inside the launch through a block the host fills
        // with -inf first; the last tenth of the tiles is handed out by ticket when there is one query group
        const int tiles = (int)((n + QS_TILE_ROWS - 1) / QS_TILE_ROWS);
        const int per = tiles / pl.grid_x;
        p.n_tiles = tiles;
        p.pub_best = nullptr;
        p.pub_thr = nullptr;
        p.tickets = nullptr;
        p.p_static = INT_MAX / 2;
        const bool seed = per >= 12 && !(dbg & DBG_NO_SEED);
        const bool dyn = pl.grid_y == 1 && per >= 24 && !(dbg & DBG_NO_DYN);
        if (seed || dyn) {
            MMRAG_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)((char *)workspace + wl.off_pub), (int)TICKET0,
                                              (wl.xch_end - wl.off_pub) / 4, s));
            if (seed) {
                p.pub_best = (float *)((char *)workspace + wl.off_pub);
                p.pub_thr = (float *)((char *)workspace + wl.off_pthr);
            }
            if (dyn) {
                p.tickets = (unsigned *)((char *)workspace + wl.off_tk);
                const int tail = per / 10 > 3 ? per / 10 : 3;
                p.p_static = per - tail;   // >= 21
            }
        }
        if (int st = qsw_launch(dtype, pl.K, walk_mfma, p, pl.grid_x, pl.grid_y, s)) return st;
        MMRAG_CHECK_HIP(hipGetLastError());
        return MMRAG_OK;
    }
    if (pl.qs_ok && qs_supported(dtype, p.row_bytes, pl.K)) {
        // query-stationary kernel.  Long shards: (1) a sample pass over the first pre_tiles * 256 rows that keeps
        // only each workgroup's best score per query, (2) thr0[q] = K-th largest of those bests, (3) the walk over
        // ALL rows (the sample again: it kept no candidates) with every lane's threshold warm from the first tile.
        const int tiles = (int)((n + QS_TILE_ROWS - 1) / QS_TILE_ROWS);
        float *thr0 = nullptr;
        if (pl.pre_tiles > 0) {
            KParams ks = p;
            ks.n_tiles = pl.pre_tiles * (TM / QS_TILE_ROWS);
            ks.sample_best = (float *)((char *)workspace + wl.off_pub);
            if (int st = qs_launch(dtype, pl.K, ks, pl.grid_x, pl.grid_y, s)) return st;
            thr0 = (float *)((char *)workspace + wl.off_thr);
            if (int st = qs_seed_thresholds(pl.K, ks.sample_best, pl.grid_x, pl.b_pad, thr0, s)) return st;
        }
        p.n_tiles = tiles;
        p.thr0 = thr0;
        if (int st = qs_launch(dtype, pl.K, p, pl.grid_x, pl.grid_y, s)) return st;
        MMRAG_CHECK_HIP(hipGetLastError());
        return MMRAG_OK;
    }
    auto run = [&](int tile0, int n_tiles, int grid_x, const float *thr0) -> int {
        KParams kp = p;
        kp.tile0 = tile0;
        kp.n_tiles = n_tiles;
        kp.thr0 = thr0;
        Plan lp = pl;
        lp.grid_x = grid_x;
        if (dtype == MMRAG_F32) return dispatch_main<MMRAG_F32>(lp, kp, s);
        if (dtype == MMRAG_F16) return dispatch_main<MMRAG_F16>(lp, kp, s);
        return dispatch_main<MMRAG_BF16>(lp, kp, s);
    };
    if (pl.pre_tiles > 0) {
        // 1. sample pre-pass over the first pre_tiles tiles, one per workgroup
        float *thr0 = (float *)((char *)workspace + wl.off_thr);
        float *top_s = (float *)((char *)workspace + wl.off_ts);
        long long *top_r = (long long *)((char *)workspace + wl.off_tr);
        const int pre = pl.pre_tiles;
        const int pre_grid = pre < pl.grid_x ? pre : pl.grid_x;
        if (int st = run(0, pre, pre_grid, nullptr)) return st;
        MMRAG_CHECK_HIP(hipGetLastError());
        // 2. its exact top-K per query, 3. seed thresholds + keep it as the last candidate list
        const long long n_cand = (long long)pre_grid * pl.K;
        launch_merge<int>(pl.K, p.cand_s, p.cand_r, n_cand, n_cand, 0, (long long)pl.n_lists * pl.K, B, pl.K, 0, top_s,
                          top_r, s, thr0, p.cand_s + seed_off, p.cand_r + seed_off, (long long)pl.n_lists * pl.K);
        MMRAG_CHECK_HIP(hipGetLastError());
        // 4. main pass over the remaining tiles, selection armed with the sample thresholds
        if (int st = run(pre, pl.n_tiles - pre, pl.grid_x, thr0)) return st;
    } else {
        if (pl.n_lists > pl.grid_x) {
            // (debug: pre-pass switched off) the sample's list slot exists but nothing fills it
            const long long total = (long long)pl.b_pad;
            fill_seed_empty_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(
                p.cand_s + seed_off, p.cand_r + seed_off, (long long)pl.n_lists * pl.K, pl.K, total);
        }
        if (int st = run(0, pl.n_tiles, pl.grid_x, nullptr)) return st;
    }
    MMRAG_CHECK_HIP(hipGetLastError());
    return MMRAG_OK;
}

// which kernel a (B, n, ld, dtype, k) search runs on: 2 = single-launch walk (search_qsw.hip), 1 = three-launch
// query-stationary (search_qs.hip), 0 = slab-ring.  For the
// bench's roofline label and the tools; not part of the public ABI.
int mmrag_internal_search_uses_qs(int B, int64_t n, int64_t ld, int dtype, int k) {
    if (B <= 0 || n <= 0 || k <= 0 || k > MMRAG_MAX_K) return 0;
    const Plan pl = make_plan(B, n, k, 0u);
    const unsigned rb = (unsigned)(ld * esize(dtype));
    if (!pl.qs_ok) return 0;
    if (qsw_supported(dtype, rb, pl.K, QSW_DEFAULT_MFMA)) return 2;   // the single-launch walk (search_qsw.hip)
    return qs_supported(dtype, rb, pl.K) ? 1 : 0;
}

int mmrag_cosine_topk_lists(const void *q, const void *corpus, int B, int64_t n, int d, int64_t ld, int dtype,
                            int k, const uint32_t *alive_bits, void *workspace, size_t workspace_bytes,
                            void *stream) {
    return mmrag_internal_cosine_topk_lists_ex(q, corpus, B, n, d, ld, dtype, k, alive_bits, workspace,
                                               workspace_bytes, stream, 0u);
}

int mmrag_cosine_topk_select(int B, int64_t n, int k, int64_t row_offset, const void *workspace,
                             float *out_scores, int64_t *out_rows, void *stream) {
    MMRAG_CHECK_ARG(B > 0 && n >= 0 && k >= 1 && k <= MMRAG_MAX_K, "cosine_topk_select: bad shape");
    MMRAG_CHECK_ARG(out_scores && out_rows, "cosine_topk_select: null output");
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        const long long total = (long long)B * k;
        fill_empty_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(out_scores, (long long *)out_rows, total);
        MMRAG_CHECK_HIP(hipGetLastError());
        return MMRAG_OK;
    }
    MMRAG_CHECK_ARG(workspace, "cosine_topk_select: null workspace");
    const Plan pl = make_plan(B, n, k);
    const WsLayout wl = ws_layout(pl);
    const float *cand_s = (const float *)workspace;
    const int *cand_r = (const int *)((const char *)workspace + wl.off_r);
    const long long n_cand = (long long)pl.n_lists * pl.K;
    launch_merge<int>(pl.K, cand_s, cand_r, n_cand, n_cand, 0, n_cand, B, k, row_offset, out_scores,
                      (long long *)out_rows, s);
    MMRAG_CHECK_HIP(hipGetLastError());
    return MMRAG_OK;
}

int mmrag_cosine_topk(const void *q, const void *corpus, int B, int64_t n, int d, int64_t ld, int dtype,
                      int k, int64_t row_offset, const uint32_t *alive_bits, float *out_scores,
                      int64_t *out_rows, void *workspace, size_t workspace_bytes, void *stream) {
    if (int st = mmrag_cosine_topk_lists(q, corpus, B, n, d, ld, dtype, k, alive_bits, workspace, workspace_bytes,
                                         stream))
        return st;
    return mmrag_cosine_topk_select(B, n, k, row_offset, workspace, out_scores, out_rows, stream);
}

int mmrag_merge_topk(const float *scores, const int64_t *rows, int G, int B, int k_in, int k,
                     float *out_scores, int64_t *out_rows, void *stream) {
    MMRAG_CHECK_ARG(scores && rows && out_scores && out_rows, "merge_topk: null pointer");
    MMRAG_CHECK_ARG(G > 0 && B > 0 && k_in > 0, "merge_topk: bad shape G=%d B=%d k_in=%d", G, B, k_in);
    MMRAG_CHECK_ARG(k >= 1 && k <= MMRAG_MAX_K, "merge_topk: k=%d outside 1..%d", k, MMRAG_MAX_K);
    const int K = k <= 5 ? 5 : (k <= 10 ? 10 : 20);
    launch_merge<long long>(K, scores, (const long long *)rows, (long long)G * k_in, k_in, (long long)B * k_in,
                            k_in, B, k, 0, out_scores, (long long *)out_rows, (hipStream_t)stream);
    MMRAG_CHECK_HIP(hipGetLastError());
    return MMRAG_OK;
}

}  // extern "C"
