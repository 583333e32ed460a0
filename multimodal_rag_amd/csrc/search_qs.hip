// Query-stationary fused cosine GEMM + top-k for gfx950 (MI355X): the retrieve kernel for big query batches.
//
// Replaces chromadb's collection.query (reference call site app/utils/embedder.py:595-601) for batches of
// more than 128 queries (BASELINE configs 3-5).  Same contract as search.hip's slab-ring kernel (exact scores,
// per-workgroup candidate lists, ties -> lower row) with a different data flow:
//
//   * the slab-ring kernel streams a 32 KiB corpus slab AND a 32 KiB query slab per K-step through LDS: the
//     query matrix is re-read from L2 once per 256 corpus rows, and that refill costs as much load-path time
//     as the corpus itself;
//   * here the queries never move: a workgroup is 4 waves, ONE per SIMD, each with the whole 512-register
//     file; a wave keeps its 64 queries x full K as MFMA B-operand fragments in registers (d=768 fp16:
//     96 fragments = 384 registers) for the life of the kernel, so the only stream is the corpus, read
//     once from HBM, 128-byte K-slabs of 64-row tiles through a deep LDS-DMA ring (up to 96 KiB in flight);
//   * every wave reads every corpus fragment from LDS (ds_read_b128, XOR-swizzled, conflict-free) and feeds it
//     to two MFMAs (its two 32-query blocks): 1 KiB of LDS read per 64 matrix-pipe cycles per wave;
//   * orientation D[corpus row][query] as before: the query sits on the lane, corpus rows in the accumulator
//     registers, selection is lane-local.  The per-lane top-K lists live in LDS (they are touched only when a
//     score beats the lane's threshold, which after seeding is rare) to leave the registers to Q.
#include "search_shared.h"

#include <type_traits>

using namespace mmrag;

namespace mmrag_impl {

template <int DT>
struct Frag;
template <>
struct Frag<MMRAG_F16> {
    using T = half8_t;
    static __device__ inline f32x16_t mfma(T a, T b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};
template <>
struct Frag<MMRAG_BF16> {
    using T = bf16x8_t;
    static __device__ inline f32x16_t mfma(T a, T b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};

constexpr int qs_lists_bytes(int K) { return K * 2 * QS_QROWS * 8; }  // (score f32, row i32) x 2 lists per query

struct QsRing {
    int G;    // K-slabs per ring stage (one s_barrier per stage)
    int NST;  // ring stages
};
// biggest stage (fewest barriers) that still leaves >= 4 stages in the LDS left over by the lists
constexpr QsRing qs_ring(int NK, int K) {
    const int budget = 160 * 1024 - qs_lists_bytes(K);
    for (int need = 4; need >= 3; --need)
        for (int g = 4; g >= 1; --g) {
            if (NK % g) continue;
            int nst = budget / (g * QS_TILE_ROWS * SLAB);
            if (nst > 6) nst = 6;
            if (nst >= need) return QsRing{g, nst};
        }
    return QsRing{1, 2};
}

// One 16-deep k-step of a wave's 64-row x 64-query tile, as ONE asm statement so that the instruction order is
// exactly this: the two A-fragment reads of the NEXT k-step go out first, the four MFMAs of this k-step cover
// their latency, and the wait for them closes the statement (nothing else of this wave is outstanding on
// lgkmcnt inside the main loop).  The stationary Q fragments are taken from the accumulator file ("a") for the
// first 64 fragments and from arch VGPRs ("v") for the rest: a wave alone on its SIMD owns all 512 registers.
#define MMRAG_QS_KSTEP(MNEMONIC, QC)                                                                         \
    asm volatile("ds_read_b128 %4, %10 offset:%11\n\t"                                                       \
                 "ds_read_b128 %5, %10 offset:%12\n\t" MNEMONIC " %0, %6, %8, %0\n\t" MNEMONIC               \
                 " %1, %6, %9, %1\n\t" MNEMONIC " %2, %7, %8, %2\n\t" MNEMONIC " %3, %7, %9, %3\n\t"         \
                 "s_waitcnt lgkmcnt(0)"                                                                      \
                 : "+v"(c00), "+v"(c01), "+v"(c10), "+v"(c11), "=&v"(n0), "=&v"(n1)                          \
                 : "v"(a0), "v"(a1), QC(q0), QC(q1), "v"(addr), "i"(OFF0), "i"(OFF1))
// first k-step of a tile: the accumulators start from the inline constant 0
#define MMRAG_QS_KSTEP0(MNEMONIC, QC)                                                                        \
    asm volatile("ds_read_b128 %4, %10 offset:%11\n\t"                                                       \
                 "ds_read_b128 %5, %10 offset:%12\n\t" MNEMONIC " %0, %6, %8, 0\n\t" MNEMONIC                \
                 " %1, %6, %9, 0\n\t" MNEMONIC " %2, %7, %8, 0\n\t" MNEMONIC " %3, %7, %9, 0\n\t"            \
                 "s_waitcnt lgkmcnt(0)"                                                                      \
                 : "=&v"(c00), "=&v"(c01), "=&v"(c10), "=&v"(c11), "=&v"(n0), "=&v"(n1)                      \
                 : "v"(a0), "v"(a1), QC(q0), QC(q1), "v"(addr), "i"(OFF0), "i"(OFF1))

template <int DT, bool QA, bool FIRST, int OFF0, int OFF1, typename FT>
__device__ __forceinline__ void qs_kstep(f32x16_t &c00, f32x16_t &c01, f32x16_t &c10, f32x16_t &c11, const FT a0,
                                         const FT a1, const FT q0, const FT q1, FT &n0, FT &n1,
                                         const unsigned addr) {
    if constexpr (FIRST) {
        static_assert(QA, "the first k-step's Q fragments live in the accumulator file");
        if constexpr (DT == MMRAG_F16) MMRAG_QS_KSTEP0("v_mfma_f32_32x32x16_f16", "a");
        else MMRAG_QS_KSTEP0("v_mfma_f32_32x32x16_bf16", "a");
    } else if constexpr (DT == MMRAG_F16) {
        if constexpr (QA) MMRAG_QS_KSTEP("v_mfma_f32_32x32x16_f16", "a");
        else MMRAG_QS_KSTEP("v_mfma_f32_32x32x16_f16", "v");
    } else {
        if constexpr (QA) MMRAG_QS_KSTEP("v_mfma_f32_32x32x16_bf16", "a");
        else MMRAG_QS_KSTEP("v_mfma_f32_32x32x16_bf16", "v");
    }
}

template <int DT, int NK, int K, bool NT>
__global__ __launch_bounds__(256, 1) void cosine_topk_qs_kernel(const KParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using FT = typename Frag<DT>::T;
    constexpr int R = QS_TILE_ROWS;
    constexpr int RB = R / 32;             // 32-row MFMA blocks per tile
    static_assert(RB == 2, "the k-step statement is written for two row blocks");
    constexpr int SLABB = R * SLAB;        // one K-slab of a tile: 8 KiB
    constexpr QsRing RING = qs_ring(NK, K);
    constexpr int G = RING.G, NST = RING.NST;
    constexpr int SPT = NK / G;            // ring stages per tile
    constexpr int STAGE = G * SLABB;
    constexpr int PPS = G * RB;            // 1 KiB DMA pieces per wave per stage (R/8 pieces per slab over 4 waves)
    constexpr int KST = NK * 4;            // 16-deep MFMA k-steps per row
    constexpr int QA_STEPS = 32;           // k-steps whose two Q fragments live in the accumulator file (256 regs)
    constexpr int LISTS = qs_lists_bytes(K);
    static_assert(NST >= 3 && NST * STAGE + LISTS <= 160 * 1024, "LDS");
    static_assert((NST - 1) * PPS <= 56, "vmcnt range");
    static_assert((G - 1) * SLABB + 32 * SLAB < 65536, "ds_read immediate offset");

    __shared__ __attribute__((aligned(1024))) char smem[NST * STAGE + LISTS];
    // per-lane top-K lists: entry e of thread t at [e][t], e = qb * 2K + i (scores) / qb * 2K + K + i (rows):
    // ONE base address register per lane, everything else in the instruction's immediate offset
    float *lists = (float *)(smem + NST * STAGE);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31;
    const int h = lane >> 5;
    const unsigned long long t_entry = (p.dbg & DBG_QS_CLOCK) ? __builtin_amdgcn_s_memrealtime() : 0ull;

    const unsigned RBy = p.row_bytes;
    const int walkers = p.walkers;
    const int bx = (int)blockIdx.x % walkers;
    const int by = (int)blockIdx.x / walkers;
    const int q0 = by * QS_QROWS;
    const int my_tiles = (p.n_tiles - bx + walkers - 1) / walkers;

    // ---- corpus DMA: per-lane source offsets of this wave's RB pieces of a slab -----------------------
    // piece i = 1 is 8 rows below piece 0: same lanes, row + 8 flips bit 2 of the swizzle term
    unsigned c_off0;
    {
        const int row = (wave * RB) * 8 + (lane >> 3);
        c_off0 = (unsigned)row * RBy + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
    }
    auto c_off = [&](int i) -> unsigned { return i == 0 ? c_off0 : (c_off0 ^ 64u) + 8u * RBy; };
    // A ring stage is refilled piece by piece, one 1 KiB piece after every other k-step: four waves alone on
    // their SIMDs pay every DMA instruction with matrix-pipe time, and a burst of eight backs up in the
    // texture-address queue.  `fill` describes the stage being refilled; stages past the end of this
    // workgroup's walk are refilled through a zero-length descriptor (nothing is fetched), so the instruction
    // stream and the vmcnt arithmetic have no tail cases.
    const int n_items = my_tiles * SPT;
    auto tile_at = [&](int pos) -> long long { return (long long)p.tile0 + bx + (long long)pos * walkers; };
    int is_tile = 0, is_sg = 0, is_item = 0;
    __amdgpu_buffer_rsrc_t fill_rsrc;
    char *fill_lds;
    int fill_k0;
    auto next_fill = [&](int buf) {
        const bool real = is_item < n_items && !(p.dbg & DBG_QS_NO_DMA);
        const long long row0 = real ? ((p.dbg & DBG_QS_DMA_L2) ? (long long)bx * R : tile_at(is_tile) * R) : 0;
        const long long rows_left = p.n - row0;
        const unsigned c_bytes = real ? (unsigned)((rows_left < R ? rows_left : (long long)R) * (long long)RBy) : 0u;
        fill_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(p.corpus + (size_t)row0 * RBy), 0, c_bytes, 0x00020000);
        fill_lds = smem + buf * STAGE + wave * (RB * 1024);
        fill_k0 = is_sg * (G * SLAB);
        ++is_item;
        if (++is_sg == SPT) {
            is_sg = 0;
            ++is_tile;
        }
    };
    auto issue_piece = [&](auto pi_c) {
        constexpr int PI = decltype(pi_c)::value;
        constexpr int g = PI / RB, i = PI % RB;
        // NT: the corpus is read once, by this CU only: non-temporal.  Several query groups share the tiles: keep
        // them in L2 for the siblings.
        __builtin_amdgcn_raw_ptr_buffer_load_lds(fill_rsrc, (lds_ptr_t)(fill_lds + g * SLABB + i * 1024), 16, c_off(i),
                                                 fill_k0 + g * SLAB, 0, NT ? 2 : 0);
    };
    auto issue_all = [&]() {
        issue_piece(std::integral_constant<int, 0>{});
        issue_piece(std::integral_constant<int, 1>{});
        issue_piece(std::integral_constant<int, 2>{});
        issue_piece(std::integral_constant<int, 3>{});
        if constexpr (PPS > 4) {
            issue_piece(std::integral_constant<int, 4>{});
            issue_piece(std::integral_constant<int, 5>{});
        }
        if constexpr (PPS > 6) {
            issue_piece(std::integral_constant<int, 6>{});
            issue_piece(std::integral_constant<int, 7>{});
        }
    };
    static_assert(PPS == 4 || PPS == 6 || PPS == 8, "pieces per wave per stage");
    static_assert(2 * PPS == G * 4, "one piece after every other k-step");

    // the ring is filled first: its HBM latency runs under the loading of Q below
    if (my_tiles > 0)
        for (int st = 0; st < NST; ++st) {
            next_fill(st);
            issue_all();
        }

    // ---- the stationary operand: this wave's 64 queries, all of K, as B fragments ----------------------
    FT qf[2][KST];
    float thr[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qrow = q0 + wave * 64 + qb * 32 + r32;
        const bool live = qrow < p.B;
        const char *src = p.q + (size_t)(live ? qrow : 0) * RBy + h * 16;
#pragma unroll
        for (int s = 0; s < KST; ++s) {
            FT v = *(const FT *)(src + s * 32);
            if (!live) v = FT{};
            qf[qb][s] = v;
        }
        // padding query slot: its all-zero scores must never open the insertion path
        thr[qb] = !live ? INFINITY : (p.thr0 != nullptr ? p.thr0[qrow] : NEG_INF);
    }
    float *const my_lists = lists + tid;
    auto lst_v = [&](int qb, int i) -> float & { return my_lists[(qb * 2 * K + i) * QS_QROWS]; };
    auto lst_r = [&](int qb, int i) -> int & { return *(int *)&my_lists[(qb * 2 * K + K + i) * QS_QROWS]; };
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int i = 0; i < K; ++i) {
            lst_v(qb, i) = NEG_INF;
            lst_r(qb, i) = INT_MAX;
        }

    // ---- fragment addresses: lane part; the ring stage's base is added per stage.  ONE lane constant: k-step m of
    // a slab reads chunk (2m + h) ^ sw of row r32, and (2m + h) ^ sw == (h ^ sw) ^ 2m, so the four k-steps of a slab
    // sit at byte offsets base ^ 0, ^ 32, ^ 64, ^ 96 -- registers go to Q, not to addresses.
    const int sw = (r32 >> 1) & 7;
    const unsigned lane_off0 = (unsigned)(r32 * SLAB + ((h ^ sw) * 16));
    auto lane_off = [&](int m) -> unsigned { return lane_off0 ^ (unsigned)(32 * m); };
    const unsigned smem_base = (unsigned)(size_t)(lds_ptr_t)smem;

    f32x16_t c00, c01, c10, c11;  // [row block][query block]
    // rows past the end of the shard and dead rows (tombstones, `where` filters) are struck out of the finished
    // tile, not out of the accumulator init: the common tile (full, no mask) pays nothing
    auto strike = [&](f32x16_t &x, f32x16_t &y, unsigned mask) {
        const unsigned mh = mask >> (4 * h);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const bool ok = (mh >> ((j & 3) + 8 * (j >> 2))) & 1u;
            x[j] = ok ? x[j] : NEG_INF;
            y[j] = ok ? y[j] : NEG_INF;
        }
    };

    auto tile_max = [&](const f32x16_t &lo, const f32x16_t &hi) -> float {
        float mx = lo[0];
#pragma unroll
        for (int j = 1; j < 16; ++j) mx = fmaxf(mx, lo[j]);
#pragma unroll
        for (int j = 0; j < 16; ++j) mx = fmaxf(mx, hi[j]);
        return mx;
    };
    float best[2] = {NEG_INF, NEG_INF};  // sample pass only

    // ---- selection: lane-local, lists in LDS, entered only when some lane's score reaches its threshold ---
    auto select = [&](const int qb, const f32x16_t &lo, const f32x16_t &hi, const int row_base) {
        const float mx = tile_max(lo, hi);
        float t = thr[qb];
        if (__builtin_amdgcn_ballot_w64(mx >= t) == 0ull) return;
        TopList<K> L;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            L.v[i] = lst_v(qb, i);
            L.r[i] = lst_r(qb, i);
        }
#pragma unroll
        for (int b = 0; b < RB; ++b) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float a0 = b ? hi[4 * g] : lo[4 * g], a1 = b ? hi[4 * g + 1] : lo[4 * g + 1];
                const float a2 = b ? hi[4 * g + 2] : lo[4 * g + 2], a3 = b ? hi[4 * g + 3] : lo[4 * g + 3];
                if (__builtin_amdgcn_ballot_w64(fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)) >= t) != 0ull) {
#pragma clang loop unroll(disable)
                    for (int i = 0; i < 4; ++i) {
                        const float s = i == 0 ? a0 : (i == 1 ? a1 : (i == 2 ? a2 : a3));
                        const bool pass = s >= t;
                        if (__builtin_amdgcn_ballot_w64(pass) != 0ull) {
                            L.insert_strict(pass ? s : NEG_INF, row_base + b * 32 + i + 8 * g);
                            t = fmaxf(t, L.v[K - 1]);
                        }
                    }
                }
            }
        }
        // k-th best of the union of the two half-wave lists of this query: a lower bound of the final
        // k-th score, shared by both lanes
        float u = fmaxf(L.v[K - 1], __shfl_xor(L.v[K - 1], 32));
#pragma unroll
        for (int i = 0; i + 1 < K; ++i) u = fmaxf(u, fminf(L.v[i], __shfl_xor(L.v[K - 2 - i], 32)));
        thr[qb] = fmaxf(t, u);
#pragma unroll
        for (int i = 0; i < K; ++i) {
            lst_v(qb, i) = L.v[i];
            lst_r(qb, i) = L.r[i];
        }
    };

    // ---- main loop.  One continuous software pipeline over k-steps: step s issues the LDS reads of step s+1
    // (which may belong to the next ring stage or the next tile) before its own MFMAs.  The stage hand-over
    // (counted vmcnt for the next stage's DMA + s_barrier + refill of the stage just drained) sits after the
    // SECOND-TO-LAST step of a stage: by then every LDS read of the stage has returned in every wave. --------
    unsigned long long t0c = 0, t0r = 0;
    if (p.dbg & DBG_QS_CLOCK) {
        t0c = __builtin_amdgcn_s_memtime();
        t0r = __builtin_amdgcn_s_memrealtime();
    }
    if (my_tiles > 0) {
        FT fa0, fa1, fb0, fb1;  // A fragments of the current / next k-step (two row blocks each)
        wait_vmcnt<(NST - 1) * PPS>();
        __builtin_amdgcn_s_barrier();
        {
            const char *st = smem + lane_off(0);
            fa0 = *(const FT *)(st);
            fa1 = *(const FT *)(st + 32 * SLAB);
        }
        for (int ti = 0; ti < my_tiles; ++ti) {
#pragma unroll
            for (int sg = 0; sg < SPT; ++sg) {
                const int it = ti * SPT + sg;
                const unsigned st_cur = smem_base + (unsigned)((it % NST) * STAGE);
                const unsigned st_nxt = smem_base + (unsigned)(((it + 1) % NST) * STAGE);
#pragma unroll
                for (int g = 0; g < G; ++g) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const int ks = (sg * G + g) * 4 + m;          // this step's Q fragments
                        const bool last = g == G - 1 && m == 3;        // next step opens the next stage
                        const int ng = last ? 0 : (m == 3 ? g + 1 : g);
                        const int nm = (m + 1) & 3;
                        const unsigned addr = (last ? st_nxt : st_cur) + lane_off(nm);
                        auto step = [&](auto ng_c, auto qa_c, FT &x0, FT &x1, FT &y0, FT &y1) {
                            constexpr int NG = decltype(ng_c)::value;
                            constexpr bool QA = decltype(qa_c)::value;
                            if (ks == 0)
                                qs_kstep<DT, true, true, NG * SLABB, NG * SLABB + 32 * SLAB, FT>(
                                    c00, c01, c10, c11, x0, x1, qf[0][ks], qf[1][ks], y0, y1, addr);
                            else
                                qs_kstep<DT, QA, false, NG * SLABB, NG * SLABB + 32 * SLAB, FT>(
                                    c00, c01, c10, c11, x0, x1, qf[0][ks], qf[1][ks], y0, y1, addr);
                        };
                        auto with_ng = [&](auto qa_c, FT &x0, FT &x1, FT &y0, FT &y1) {
                            if (ng == 0) step(std::integral_constant<int, 0>{}, qa_c, x0, x1, y0, y1);
                            else if (ng == 1) step(std::integral_constant<int, 1>{}, qa_c, x0, x1, y0, y1);
                            else if (ng == 2) step(std::integral_constant<int, 2>{}, qa_c, x0, x1, y0, y1);
                            else step(std::integral_constant<int, 3>{}, qa_c, x0, x1, y0, y1);
                        };
                        // A fragments alternate between (fa0, fa1) and (fb0, fb1); a stage has an even number of steps
                        if ((m & 1) == 0) {
                            if (ks < QA_STEPS) with_ng(std::true_type{}, fa0, fa1, fb0, fb1);
                            else with_ng(std::false_type{}, fa0, fa1, fb0, fb1);
                        } else {
                            if (ks < QA_STEPS) with_ng(std::true_type{}, fb0, fb1, fa0, fa1);
                            else with_ng(std::false_type{}, fb0, fb1, fa0, fa1);
                        }
                        constexpr int KSTG = G * 4;          // k-steps per stage
                        const int j = g * 4 + m;
                        if (j == KSTG - 2) {
                            // every read of this stage is back: hand the ring over.  The stage just drained is
                            // refilled over the next KSTG steps (pieces after steps KSTG-1, 1, 3, ...)
                            wait_vmcnt<(NST - 2) * PPS>();
                            if (!(p.dbg & DBG_QS_NO_BARRIER)) __builtin_amdgcn_s_barrier();
                            next_fill(it % NST);
                        }
                        if (j == KSTG - 1) issue_piece(std::integral_constant<int, 0>{});
                        else if (j == 1) issue_piece(std::integral_constant<int, 1>{});
                        else if (j == 3) issue_piece(std::integral_constant<int, 2>{});
                        else if (j == 5) issue_piece(std::integral_constant<int, 3>{});
                        else if (j == 7 && PPS > 4) issue_piece(std::integral_constant<int, (PPS > 4 ? 4 : 0)>{});
                        else if (j == 9 && PPS > 4) issue_piece(std::integral_constant<int, (PPS > 4 ? 5 : 0)>{});
                        else if (j == 11 && PPS > 6) issue_piece(std::integral_constant<int, (PPS > 6 ? 6 : 0)>{});
                        else if (j == 13 && PPS > 6) issue_piece(std::integral_constant<int, (PPS > 6 ? 7 : 0)>{});
                    }
                }
            }
            asm volatile("s_nop 15\n\ts_nop 7");  // last MFMA's D -> VALU readers
            const long long row0 = tile_at(ti) * R;
            if (row0 + R > p.n || p.alive_bits != nullptr) {
                const long long left = p.n - row0;  // >= 1
                unsigned m0 = left >= 32 ? 0xffffffffu : ((1u << (int)left) - 1u);
                unsigned m1 = left >= 64 ? 0xffffffffu : (left > 32 ? ((1u << (int)(left - 32)) - 1u) : 0u);
                if (p.alive_bits != nullptr) {
                    // scalar loads (wave-uniform words): a vector load here would make hipcc drain vmcnt, i.e. the
                    // whole DMA ring, once per tile
                    typedef const __attribute__((address_space(4))) uint32_t *scalar_words_t;
                    const scalar_words_t words = (scalar_words_t)p.alive_bits;
                    m0 &= words[row0 >> 5];
                    if (left > 32) m1 &= words[(row0 >> 5) + 1];
                }
                strike(c00, c01, m0);
                strike(c10, c11, m1);
            }
            const int row_base = (int)row0 + 4 * h;
            if (p.sample_best != nullptr) {
                // sample pass: only the best score of each query matters (K workgroups' bests bound the K-th score)
                best[0] = fmaxf(best[0], tile_max(c00, c10));
                best[1] = fmaxf(best[1], tile_max(c01, c11));
            } else if (!(p.dbg & DBG_QS_NO_SELECT)) {
                select(0, c00, c10, row_base);
                select(1, c01, c11, row_base);
            } else {
                asm volatile("" ::"v"(c00), "v"(c01), "v"(c10), "v"(c11));
            }
        }
    }
    if ((p.dbg & DBG_QS_CLOCK) && p.stamps != nullptr && tid == 0) {
        p.stamps[8 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0c;
        p.stamps[8 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - t0r;
        p.stamps[8 * blockIdx.x + 2] = t0r - t_entry;
        p.stamps[8 * blockIdx.x + 3] = t_entry;
    }
    if (p.sample_best != nullptr) {
        // sample pass: best[q][bx] for the threshold kernel; padding queries publish -inf
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float m = fmaxf(best[qb], __shfl_xor(best[qb], 32));
            const int q = q0 + wave * 64 + qb * 32 + r32;
            if (h == 0) p.sample_best[(size_t)q * walkers + bx] = q < p.B ? m : NEG_INF;
        }
        return;
    }
    // ---- merge the two half-wave lists of every query, write ONE list per query per workgroup ------------
    __syncthreads();
    if (q0 + tid < p.B) {
        const int w = tid >> 6, qb = (tid >> 5) & 1, r = tid & 31;
        const float *la = lists + w * 64 + r + (qb * 2 * K) * QS_QROWS, *lb = la + 32;
        TopList<K> m;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            m.v[i] = la[i * QS_QROWS];
            m.r[i] = __float_as_int(la[(K + i) * QS_QROWS]);
        }
        for (int i = 0; i < K; ++i) {
            const float x = lb[i * QS_QROWS];
            const int xr = __float_as_int(lb[(K + i) * QS_QROWS]);
            if (xr == INT_MAX || !better(x, xr, m.v[K - 1], m.r[K - 1])) break;
            m.insert_ordered(x, xr);
        }
        const size_t base = ((size_t)(q0 + tid) * p.n_lists + bx) * K;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            p.cand_s[base + i] = m.v[i];
            p.cand_r[base + i] = m.r[i];
        }
        if (bx == 0 && p.n_lists > walkers) {
            // the list slot the two-launch plan keeps for its sample pass: nothing to put there
            const size_t seed = ((size_t)(q0 + tid) * p.n_lists + (p.n_lists - 1)) * K;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                p.cand_s[seed + i] = NEG_INF;
                p.cand_r[seed + i] = INT_MAX;
            }
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

// thr0[q] = K-th largest of the `walkers` per-workgroup bests of query q (scores of distinct rows of the sample, so
// at least K rows of the shard reach it: a valid lower bound of the final K-th score).  One wave per query.
template <int K>
__global__ __launch_bounds__(256) void qs_seed_thr_kernel(const float *__restrict__ best, int walkers, int n_queries,
                                                          float *__restrict__ thr0) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= n_queries) return;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int w = lane + 64 * j;
        v[j] = w < walkers ? best[(size_t)q * walkers + w] : NEG_INF;
    }
    const float t = wave_kth_largest<K>(v[0], v[1], v[2], v[3], lane);
    if (lane == 0) thr0[q] = t;
}

int qs_seed_thresholds(int K, const float *best, int walkers, int n_queries, float *thr0, hipStream_t s) {
    if (walkers > 256) return MMRAG_EUNSUPPORTED;   // (fewer bests than K: the K-th largest is -inf, i.e. no threshold)
    const unsigned grid = (unsigned)((n_queries + 3) / 4);
    if (K == 5) qs_seed_thr_kernel<5><<<grid, 256, 0, s>>>(best, walkers, n_queries, thr0);
    else if (K == 10) qs_seed_thr_kernel<10><<<grid, 256, 0, s>>>(best, walkers, n_queries, thr0);
    else return MMRAG_EUNSUPPORTED;
    return MMRAG_OK;
}

bool qs_supported(int dtype, unsigned row_bytes, int K) {
    if (dtype != MMRAG_F16 && dtype != MMRAG_BF16) return false;
    // k = 20 was built and measured too: 911 us against 772 for the slab-ring kernel at 1M x 768, B = 256 (the 20-th best
    // of the sample is a weak threshold, the lists take 80 KB of LDS from the ring and the insertion path spills)
    if (K != 5 && K != 10) return false;
    const unsigned nk = row_bytes / SLAB;
    return row_bytes % SLAB == 0 && (nk == 6 || nk == 8 || nk == 12);
}

template <int DT, int K>
static int qs_launch_nk(const KParams &p, int grid, hipStream_t s) {
    const bool nt = !p.share_l2;
    switch (p.row_bytes / SLAB) {
        case 6:
            if (nt) cosine_topk_qs_kernel<DT, 6, K, true><<<grid, 256, 0, s>>>(p);
            else cosine_topk_qs_kernel<DT, 6, K, false><<<grid, 256, 0, s>>>(p);
            break;
        case 8:
            if (nt) cosine_topk_qs_kernel<DT, 8, K, true><<<grid, 256, 0, s>>>(p);
            else cosine_topk_qs_kernel<DT, 8, K, false><<<grid, 256, 0, s>>>(p);
            break;
        case 12:
            if (nt) cosine_topk_qs_kernel<DT, 12, K, true><<<grid, 256, 0, s>>>(p);
            else cosine_topk_qs_kernel<DT, 12, K, false><<<grid, 256, 0, s>>>(p);
            break;
        default: return MMRAG_EUNSUPPORTED;
    }
    return MMRAG_OK;
}

int qs_launch(int dtype, int K, const KParams &p, int grid_x, int grid_y, hipStream_t s) {
    KParams kp = p;
    kp.walkers = grid_x;
    kp.share_l2 = grid_y > 1;
    const int grid = grid_x * grid_y;
    if (dtype == MMRAG_F16) {
        if (K == 5) return qs_launch_nk<MMRAG_F16, 5>(kp, grid, s);
        if (K == 10) return qs_launch_nk<MMRAG_F16, 10>(kp, grid, s);
    } else if (dtype == MMRAG_BF16) {
        if (K == 5) return qs_launch_nk<MMRAG_BF16, 5>(kp, grid, s);
        if (K == 10) return qs_launch_nk<MMRAG_BF16, 10>(kp, grid, s);
    }
    return MMRAG_EUNSUPPORTED;
}

}  // namespace mmrag_impl
