// Single-launch query-stationary walk: fused cosine GEMM + top-k for big query batches on gfx950 (MI355X).
//
// Replaces chromadb's collection.query (reference call site app/utils/embedder.py:595-601) for batches of more than
// 128 queries on long shards (BASELINE configs 3-5).  Same data flow as search_qs.hip's kernel -- the queries never
// move (a wave keeps its 64 queries x all of K as MFMA B fragments in registers), the corpus streams once from HBM
// through an LDS-DMA ring, selection is lane-local with the lists in LDS -- with three differences:
//
//   * ONE launch.  The three-launch plan (sample pass, threshold kernel, walk) read the 65 536 sample rows twice and
//     paid two launch boundaries and two Q loads.  Here every workgroup starts its walk cold, and the thresholds
//     arrive while it walks: after its first tiles a workgroup publishes its best score per query
//     (pub_best[group][walker][query]); the workgroup responsible for a query gathers that query's bests -- whatever
//     has been published by then, the rest still reads -inf -- takes their K-th largest and publishes it
//     (pub_thr[group][query]); every workgroup picks the thresholds up at tile boundaries.  Nobody waits for anybody:
//     bests are scores of distinct rows, so the K-th largest of ANY subset of them is a valid lower bound of the
//     query's final K-th score, and a late or missing value only leaves a threshold colder.  All of it rides the
//     ring's own machinery (LDS-DMA pieces with `sc1`, landed by the counted vmcnt waits the ring already makes), so
//     no access in the tile loop waits for memory.  The host fills the exchange block with -inf before the launch.
//   * the last ~10 % of the tiles are handed out by ticket (one returning atomic per tile, requested three tiles
//     ahead, spread over the workgroup through LDS) instead of by the static rule: the slowest CU no longer sets the
//     time of the launch.
//   * the MFMA shape is a template parameter: v_mfma_f32_32x32x16 (two 32-query blocks per wave) or
//     v_mfma_f32_16x16x32 (four 16-query blocks; one 1 KiB fragment read feeds four MFMAs, identical LDS bytes, Q and
//     accumulator registers per flop) -- MI355X_MICROARCH.md 'DVFS give-back' item 7 / cdna_hip_programming.md rule 28:
//     the chip holds a different clock on the two shapes, the faster by wall is the one dispatched.
#include "search_shared.h"

#include <type_traits>
#include <utility>

using namespace mmrag;

namespace mmrag_impl {

namespace {

template <int DT>
struct WFrag;
template <>
struct WFrag<MMRAG_F16> {
    using T = half8_t;
};
template <>
struct WFrag<MMRAG_BF16> {
    using T = bf16x8_t;
};

template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

constexpr int QSW_XCH = 2048 + 64;   // gather buffer 1 KiB + poll buffer 1 KiB + 8 ticket slots (padded)
constexpr int qsw_lists_bytes(int K, int MS) { return K * (MS == 32 ? 2 : 4) * QS_QROWS * 8; }

struct WRing {
    int G;    // K-slabs per ring stage (one s_barrier per stage)
    int NST;  // ring stages
};
// biggest stage (fewest barriers) that still leaves >= 4 stages in the LDS left over by the lists
constexpr WRing qsw_ring(int NK, int K, int MS) {
#if defined(MMRAG_QSW_RING16)   // developer builds: ring of the 16x16x32 shape as G * 10 + NST (A/B of stage size vs depth)
    if (MS == 16 && NK % (MMRAG_QSW_RING16 / 10) == 0) return WRing{MMRAG_QSW_RING16 / 10, MMRAG_QSW_RING16 % 10};
#endif
    const int budget = 160 * 1024 - qsw_lists_bytes(K, MS) - QSW_XCH;
    for (int need = 4; need >= 3; --need)
        for (int g = 4; g >= 1; --g) {
            if (NK % g) continue;
            int nst = budget / (g * QS_TILE_ROWS * SLAB);
            if (nst > 6) nst = 6;
            if (nst >= need) return WRing{g, nst};
        }
    return WRing{1, 2};
}

// ---- one statement of the k loop = ONE asm statement: the two A-fragment reads of the NEXT statement go out first,
// this statement's MFMAs (128 matrix-pipe cycles) cover their latency, the wait for them closes the statement.  The
// stationary Q fragments sit in the accumulator file ("a") for the first 64 fragments and in arch VGPRs ("v") for the
// rest: a wave alone on its SIMD owns all 512 registers.
//
// 32x32x16: the fragment address is made inside the statement: v_xad_u32 = (lane constant ^ k-step term) + stage base
// (an SGPR); 16x16x32: two addresses per ring stage (qsw_stage_addr), made with the stage's last statement.  Left to hipcc, the four XOR variants of the lane constant and four stage addresses stay live across the tile
// (eight VGPRs that the Q fragments need: it spilled two of those instead, and a spill reload waits vmcnt(0)).
//
// 32x32x16: a statement is one 16-deep k-step of the wave's 64 rows x 64 queries: 2 row blocks x 2 query blocks.
#define MMRAG_QSW_S32(MNEMONIC, QC, C00, C01, C10, C11, ACC)                                                  \
    asm volatile("v_xad_u32 %[t], %[lo], %[xc], %[st]\n\t"                                                    \
                 "ds_read_b128 %[n0], %[t] offset:%[o0]\n\t"                                                  \
                 "ds_read_b128 %[n1], %[t] offset:%[o1]\n\t" MNEMONIC " %[c00], %[a0], %[q0], " C00 "\n\t"    \
                 MNEMONIC " %[c01], %[a0], %[q1], " C01 "\n\t" MNEMONIC " %[c10], %[a1], %[q0], " C10 "\n\t"  \
                 MNEMONIC " %[c11], %[a1], %[q1], " C11 "\n\t"                                                \
                 "s_waitcnt lgkmcnt(0)"                                                                       \
                 : [c00] ACC(c00), [c01] ACC(c01), [c10] ACC(c10), [c11] ACC(c11), [n0] "=&v"(n0),            \
                   [n1] "=&v"(n1), [t] "=&v"(tmp)                                                             \
                 : [a0] "v"(a0), [a1] "v"(a1), [q0] QC(q0), [q1] QC(q1), [lo] "v"(lo), [st] "s"(st),          \
                   [xc] "i"(XC), [o0] "i"(OFF0), [o1] "i"(OFF1))

template <int DT, bool QA, bool FIRST, int XC, int OFF0, int OFF1, typename FT>
__device__ __forceinline__ void qsw_stmt32(f32x16_t &c00, f32x16_t &c01, f32x16_t &c10, f32x16_t &c11, const FT a0,
                                           const FT a1, const FT q0, const FT q1, FT &n0, FT &n1, const unsigned lo,
                                           const unsigned st) {
    unsigned tmp;
    if constexpr (FIRST) {   // the accumulators start from the inline constant 0
        static_assert(QA, "the first k-step's Q fragments live in the accumulator file");
        if constexpr (DT == MMRAG_F16) MMRAG_QSW_S32("v_mfma_f32_32x32x16_f16", "a", "0", "0", "0", "0", "=&v");
        else MMRAG_QSW_S32("v_mfma_f32_32x32x16_bf16", "a", "0", "0", "0", "0", "=&v");
    } else if constexpr (DT == MMRAG_F16) {
        if constexpr (QA) MMRAG_QSW_S32("v_mfma_f32_32x32x16_f16", "a", "%[c00]", "%[c01]", "%[c10]", "%[c11]", "+v");
        else MMRAG_QSW_S32("v_mfma_f32_32x32x16_f16", "v", "%[c00]", "%[c01]", "%[c10]", "%[c11]", "+v");
    } else {
        if constexpr (QA) MMRAG_QSW_S32("v_mfma_f32_32x32x16_bf16", "a", "%[c00]", "%[c01]", "%[c10]", "%[c11]", "+v");
        else MMRAG_QSW_S32("v_mfma_f32_32x32x16_bf16", "v", "%[c00]", "%[c01]", "%[c10]", "%[c11]", "+v");
    }
}

// 16x16x32: a statement is HALF of a 32-deep k-step: two of the four 16-row blocks x all four 16-query blocks.  Each
// 1 KiB A fragment (16 rows x 32 k) feeds four MFMAs; eight MFMAs of 16 cycles = the same 128 cycles per two reads.
// The two reads sit in the gaps after the first two MFMAs, one per gap: a 16x16x32 MFMA holds the SIMD's issue for 8
// of its 16 cycles, what is issued in the other 8 is free, what does not fit adds its full cost (MI355X_MICROARCH.md,
// issue costs) -- and this wave has the SIMD to itself.
#define MMRAG_QSW_S16(MNEMONIC, QC, D0, D1, D2, D3, D4, D5, D6, D7, ACC)                                      \
    asm volatile(MNEMONIC " %[d0], %[a0], %[q0], " D0 "\n\t"                                                  \
                 "ds_read_b128 %[n0], %[ad] offset:%[o0]\n\t" MNEMONIC " %[d1], %[a0], %[q1], " D1 "\n\t"     \
                 "ds_read_b128 %[n1], %[ad] offset:%[o1]\n\t" MNEMONIC " %[d2], %[a0], %[q2], " D2 "\n\t"     \
                 MNEMONIC " %[d3], %[a0], %[q3], " D3 "\n\t" MNEMONIC " %[d4], %[a1], %[q0], " D4 "\n\t"      \
                 MNEMONIC " %[d5], %[a1], %[q1], " D5 "\n\t" MNEMONIC " %[d6], %[a1], %[q2], " D6 "\n\t"      \
                 MNEMONIC " %[d7], %[a1], %[q3], " D7 "\n\t"                                                  \
                 "s_waitcnt lgkmcnt(0)"                                                                       \
                 : [d0] ACC(d0), [d1] ACC(d1), [d2] ACC(d2), [d3] ACC(d3), [d4] ACC(d4), [d5] ACC(d5),        \
                   [d6] ACC(d6), [d7] ACC(d7), [n0] "=&v"(n0), [n1] "=&v"(n1)                                 \
                 : [a0] "v"(a0), [a1] "v"(a1), [q0] QC(q0), [q1] QC(q1), [q2] QC(q2), [q3] QC(q3),            \
                   [ad] "v"(adr), [o0] "i"(OFF0), [o1] "i"(OFF1))
#define MMRAG_QSW_S16_ACC(MNEMONIC, QC)                                                                       \
    MMRAG_QSW_S16(MNEMONIC, QC, "%[d0]", "%[d1]", "%[d2]", "%[d3]", "%[d4]", "%[d5]", "%[d6]", "%[d7]", "+v")
#define MMRAG_QSW_S16_FIRST(MNEMONIC, QC) MMRAG_QSW_S16(MNEMONIC, QC, "0", "0", "0", "0", "0", "0", "0", "0", "=&v")

template <int DT, bool QA, bool FIRST, int OFF0, int OFF1, typename FT>
__device__ __forceinline__ void qsw_stmt16(f32x4_t &d0, f32x4_t &d1, f32x4_t &d2, f32x4_t &d3, f32x4_t &d4,
                                           f32x4_t &d5, f32x4_t &d6, f32x4_t &d7, const FT a0, const FT a1,
                                           const FT q0, const FT q1, const FT q2, const FT q3, FT &n0, FT &n1,
                                           const unsigned adr) {
    if constexpr (FIRST) {
        static_assert(QA, "the first k-step's Q fragments live in the accumulator file");
        if constexpr (DT == MMRAG_F16) MMRAG_QSW_S16_FIRST("v_mfma_f32_16x16x32_f16", "a");
        else MMRAG_QSW_S16_FIRST("v_mfma_f32_16x16x32_bf16", "a");
    } else if constexpr (DT == MMRAG_F16) {
        if constexpr (QA) MMRAG_QSW_S16_ACC("v_mfma_f32_16x16x32_f16", "a");
        else MMRAG_QSW_S16_ACC("v_mfma_f32_16x16x32_f16", "v");
    } else {
        if constexpr (QA) MMRAG_QSW_S16_ACC("v_mfma_f32_16x16x32_bf16", "a");
        else MMRAG_QSW_S16_ACC("v_mfma_f32_16x16x32_bf16", "v");
    }
}

// the two fragment addresses of a ring stage on the 16x16x32 shape (the statements' 64-byte halves of a K-slab differ by
// an XOR on the lane constant, everything else is in the read's immediate offset): two instructions per stage, not one
// per statement -- every instruction between the MFMAs of a wave that is alone on its SIMD is matrix-pipe idle time
__device__ __forceinline__ void qsw_stage_addr(unsigned &a_lo, unsigned &a_hi, const unsigned lo, const unsigned st) {
    asm volatile("v_add_u32 %0, %2, %3\n\tv_xad_u32 %1, %2, 64, %3" : "=&v"(a_lo), "=&v"(a_hi) : "v"(lo), "s"(st));
}

// max of sixteen accumulator values as they are (finite or -inf): ONE statement -- between separate asm statements hipcc
// puts a wait state each
__device__ __forceinline__ float max16_raw(float x0, float x1, float x2, float x3, float x4, float x5, float x6, float x7,
                                           float x8, float x9, float x10, float x11, float x12, float x13, float x14,
                                           float x15) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3\n\t"
        "v_max3_f32 %0, %0, %4, %5\n\t"
        "v_max3_f32 %0, %0, %6, %7\n\t"
        "v_max3_f32 %0, %0, %8, %9\n\t"
        "v_max3_f32 %0, %0, %10, %11\n\t"
        "v_max3_f32 %0, %0, %12, %13\n\t"
        "v_max3_f32 %0, %0, %14, %15\n\t"
        "v_max_f32 %0, %0, %16"
        : "=&v"(r)
        : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7), "v"(x8), "v"(x9), "v"(x10), "v"(x11),
          "v"(x12), "v"(x13), "v"(x14), "v"(x15));
    return r;
}

constexpr int CP_SC1 = 16;   // cache-policy bit of the buffer builtins' aux operand: sc1 (L1 bypassed, L2 coherent at agent scope)

}  // namespace

template <int DT, int NK, int K, int MS, bool NT>
__global__ __launch_bounds__(256, 1) void cosine_topk_walk_kernel(const KParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using FT = typename WFrag<DT>::T;
    static_assert(MS == 32 || MS == 16, "MFMA shape");
    constexpr int R = QS_TILE_ROWS;
    static_assert(R == 64, "statements are written for 64-row tiles");
    constexpr int RB = R / 32;             // 1 KiB DMA pieces per wave per K-slab (8 pieces of 8 rows over 4 waves)
    constexpr int SLABB = R * SLAB;        // one K-slab of a tile: 8 KiB
    constexpr int NQB = MS == 32 ? 2 : 4;  // query blocks per wave
    constexpr int QW = MS;                 // queries per block = lanes that hold different queries
    constexpr int NSUB = 64 / QW;          // lanes that share a query (they hold different rows)
    constexpr int NGRP = 16 / NSUB;        // groups of 4 consecutive rows per lane per query block and tile
    constexpr WRing RING = qsw_ring(NK, K, MS);
    constexpr int G = RING.G, NST = RING.NST;
    constexpr int SPT = NK / G;            // ring stages per tile
    constexpr int STAGE = G * SLABB;
    constexpr int PPS = G * RB;            // DMA pieces per wave per stage
    constexpr int KSTG = G * 4;            // statements per stage
    constexpr int KST = MS == 32 ? NK * 4 : NK * 2;   // Q k-steps (16- or 32-deep), NQB fragments each
    constexpr int QA_STEPS = 64 / NQB;     // k-steps whose Q fragments live in the accumulator file (256 registers)
    constexpr int LISTS = qsw_lists_bytes(K, MS);
    // a ticket for position P must be in LDS one barrier before the fill pipeline first asks for it
    constexpr int LOOK = (NST + 1 + SPT - 1) / SPT + 1;
    static_assert(NST >= 3 && NST * STAGE + LISTS + QSW_XCH <= 160 * 1024, "LDS");
    static_assert((NST - 1) * PPS <= 56, "vmcnt range");
    static_assert((G - 1) * SLABB + 48 * SLAB + 2048 < 65536, "ds_read immediate offset");
    static_assert(2 * PPS == KSTG, "one piece after every other statement");
    // Exchange fetches (LDS-DMA, one wave) are issued at the end of a tile and used at the end of a later one: they
    // have landed once the wave has passed NST - 2 hand-over waits with nothing but ring pieces in between, i.e.
    // LANDT tiles later (1 for 768-d rows; short rows make a tile only one or two ring stages).  Issuing them among the
    // k-steps (right after a hand-over, used at the end of the same tile: thresholds a tile earlier) was built too:
    // hipcc's register allocation there does not survive any extra code, it spilled Q fragments into the tile loop.
    constexpr int LANDT = ((NST - 2) * PPS + 1 + SPT * PPS - 1) / (SPT * PPS);
    static_assert(LANDT >= 1 && LANDT * SPT * PPS - 1 >= (NST - 2) * PPS, "exchange pieces land before they are used");
    static_assert(LOOK <= 6, "ticket slots");

    __shared__ __attribute__((aligned(1024))) char smem[NST * STAGE + LISTS + QSW_XCH];
    // per-lane top-K lists: entry e of thread t at [e][t], e = qb * 2K + i (scores) / qb * 2K + K + i (rows):
    // ONE base address register per lane, everything else in the instruction's immediate offset
    float *lists = (float *)(smem + NST * STAGE);
    float *const gather_lds = (float *)(smem + NST * STAGE + LISTS);          // 256 bests of one query
    float *const poll_lds = (float *)(smem + NST * STAGE + LISTS + 1024);     // 256 thresholds of this query group
    int *const tk_lds = (int *)(smem + NST * STAGE + LISTS + 2048);           // tickets of the next positions

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & (QW - 1);        // query within a block
    const int sub = lane / QW;             // which rows of a group of 4 x NSUB this lane holds
    // ablation switches that sit among the k-steps exist in developer builds only (-DMMRAG_QSW_DEV): a scalar test per
    // ring stage is matrix-pipe time too
    auto dev_dbg = [&](unsigned flag) -> bool {
#if defined(MMRAG_QSW_DEV)
        return (p.dbg & flag) != 0;
#else
        (void)flag;
        return false;
#endif
    };
    const unsigned long long t_entry = (p.dbg & DBG_QS_CLOCK) ? __builtin_amdgcn_s_memrealtime() : 0ull;

    const unsigned RBy = p.row_bytes;
    const int walkers = p.walkers;
    const int bx = (int)blockIdx.x % walkers;
    const int by = (int)blockIdx.x / walkers;
    const int q0 = by * QS_QROWS;
    const int p_static = p.p_static;
    // tile of walk position `pos` (wave-uniform): static rule first, tickets for the tail
    auto pos_tile = [&](int pos) -> int {
        if (pos < p_static) return bx + pos * walkers;
        return p_static * walkers + (int)((unsigned)__builtin_amdgcn_readfirstlane(tk_lds[pos & 7]) - TICKET0);
    };

    // ---- corpus DMA: per-lane source offsets of this wave's RB pieces of a slab -----------------------
    // piece i = 1 is 8 rows below piece 0: same lanes, row + 8 flips bit 2 of the swizzle term
    unsigned c_off0;
    {
        const int row = (wave * RB) * 8 + (lane >> 3);
        c_off0 = (unsigned)row * RBy + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
    }
    auto c_off = [&](int i) -> unsigned { return i == 0 ? c_off0 : (c_off0 ^ 64u) + 8u * RBy; };
    // A ring stage is refilled piece by piece, one 1 KiB piece after every other statement (a burst backs up in the
    // texture-address queue and four waves alone on their SIMDs pay every cycle of it).  Stages past the end of this
    // workgroup's walk are refilled through a zero-length descriptor (nothing is fetched), so the instruction stream and
    // the vmcnt arithmetic have no tail cases.
    // Everything scalar here costs matrix-pipe time (one wave per SIMD: nothing else to issue from), so it is kept to
    // one s_add (M0) and one s_movk (the K offset) per piece: which K-slabs a refill fetches (IS_SG) is a compile-time
    // property of the statement it follows, the descriptor is made once per tile, and both constants are made where
    // they are used -- left to itself hipcc hoists two dozen of them out of the tile loop into SGPRs, spills those into
    // VGPR lanes and pays a v_readlane per piece.
    int is_pos = 0;
    int fill_tile = bx;
    __amdgpu_buffer_rsrc_t fill_rsrc;
    typedef __attribute__((address_space(3))) char *lds_char_t;
    const lds_char_t wave_lds = (lds_char_t)(lds_ptr_t)(smem + wave * (RB * 1024));
    lds_char_t stage_lds = wave_lds;   // this wave's part of the ring stage being refilled
    auto fill_open = [&]() {   // descriptor of the tile the next SPT refills fetch
        const bool real = (unsigned)fill_tile < (unsigned)p.n_tiles && !dev_dbg(DBG_QS_NO_DMA);   // (a ticket past the end: no tile)
        const long long row0 = real ? (dev_dbg(DBG_QS_DMA_L2) ? (long long)bx * R : (long long)fill_tile * R) : 0;
        const long long rows_left = p.n - row0;
        const unsigned c_bytes = real ? (unsigned)((rows_left < R ? rows_left : (long long)R) * (long long)RBy) : 0u;
        fill_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(p.corpus + (size_t)row0 * RBy), 0, c_bytes, 0x00020000);
    };
    auto next_fill = [&](auto is_sg_c, int buf) {
        constexpr int IS_SG = decltype(is_sg_c)::value;
        if constexpr (IS_SG == 0) {
            if (is_pos != 0) fill_tile = pos_tile(is_pos);
            fill_open();
            ++is_pos;
        }
        // (made opaque once per stage: hipcc then forms each piece's M0 as stage_lds + constant in one s_add and
        // cannot turn the 24 sums into loop invariants)
        stage_lds = wave_lds + buf * STAGE;
        asm volatile("" : "+s"(stage_lds));
    };
    unsigned koff = 0;   // K offset of the piece pair being issued
    auto issue_piece = [&](auto is_sg_c, auto pi_c) {
        constexpr int IS_SG = decltype(is_sg_c)::value, PI = decltype(pi_c)::value;
        constexpr int g = PI / RB, i = PI % RB;
        if constexpr (i == 0) asm volatile("s_movk_i32 %0, %1" : "=s"(koff) : "n"((IS_SG * G + g) * SLAB));
        // NT: the corpus is read once, by this CU only: non-temporal.  Several query groups share the tiles: keep
        // them in L2 for the siblings.
        __builtin_amdgcn_raw_ptr_buffer_load_lds(fill_rsrc, (lds_ptr_t)(stage_lds + g * SLABB + i * 1024), 16, c_off(i),
                                                 koff, 0, NT ? 2 : 0);
    };

    // the ring is filled first: its HBM latency runs under the loading of Q below
    const bool any_tile = bx < p.n_tiles;
    if (any_tile)
        static_for<NST>([&](auto st_c) {
            constexpr int st = decltype(st_c)::value;
            next_fill(std::integral_constant<int, st % SPT>{}, st);
            static_for<PPS>([&](auto pi_c) { issue_piece(std::integral_constant<int, st % SPT>{}, pi_c); });
        });
    // (the first stage of the main loop issues pieces 1.. of the last refill above once more -- same bytes to the same
    // place, it keeps the vmcnt arithmetic uniform -- and piece 1 takes its K offset from piece 0)
    asm volatile("s_movk_i32 %0, %1" : "=s"(koff) : "n"(((NST - 1) % SPT) * G * SLAB));

    // ---- the stationary operand: this wave's 64 queries, all of K, as B fragments ----------------------
    // 32x32x16: fragment s of a query = halfs 16 s + 8 (lane >> 5) ...; 16x16x32: halfs 32 s + 8 (lane >> 4) ...
    FT qf[NQB][KST];
    float thr[NQB];
    static_for<NQB>([&](auto qb_c) {
        constexpr int qb = decltype(qb_c)::value;
        const int qrow = q0 + wave * 64 + qb * QW + ql;
        const bool live = qrow < p.B;
        const char *src = p.q + (size_t)(live ? qrow : 0) * RBy + sub * 16;
#pragma unroll
        for (int s = 0; s < KST; ++s) {
            FT v = *(const FT *)(src + s * (16 * NSUB));
            if (!live) v = FT{};
            qf[qb][s] = v;
        }
        // padding query slot: its all-zero scores must never open the insertion path
        thr[qb] = !live ? INFINITY : NEG_INF;
    });
    float *const my_lists = lists + tid;
    auto lst_v = [&](int qb, int i) -> float & { return my_lists[(qb * 2 * K + i) * QS_QROWS]; };
    auto lst_r = [&](int qb, int i) -> int & { return *(int *)&my_lists[(qb * 2 * K + K + i) * QS_QROWS]; };
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
        for (int i = 0; i < K; ++i) {
            lst_v(qb, i) = NEG_INF;
            lst_r(qb, i) = INT_MAX;
        }

    // ---- fragment addresses: lane part; the ring stage's base is added per stage.  ONE lane constant per shape:
    // 32x32x16: statement m of a slab reads chunk (2m + h) ^ sw of row (lane & 31) = (h ^ sw) ^ 2m;
    // 16x16x32: statement m reads chunk (4 (m >> 1) + g4) ^ sw of row (lane & 15) = (g4 ^ sw) ^ 4 (m >> 1), the row
    // block pair (m & 1) in the immediate offset.  sw = (row >> 1) & 7 is the same for every 16-row block.
    const int sw = (ql >> 1) & 7;
    const unsigned lane_off0 = (unsigned)(ql * SLAB + ((sub ^ sw) * 16));
    const unsigned lane_off64 = lane_off0 ^ 64u;   // (32x32x16 only)
    const unsigned smem_base = (unsigned)(size_t)(lds_ptr_t)smem;

    // the lane id again, from the hardware: a value kept live across the k-steps for these rare steps would be spilled,
    // and a spill reload in the tile loop waits vmcnt(0), i.e. drains the ring
    // (volatile asm: a builtin would be hoisted out of the tile loop and kept live just the same)
    auto lane_now = []() -> int {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    f32x16_t c32[MS == 32 ? 4 : 1];   // [row block][query block]
    f32x4_t c16[MS == 16 ? 16 : 1];   // [row block][query block]
    // group gi (4 consecutive rows) of query block qb: score i and its row inside the tile (without the lane's 4 * sub)
    auto score = [&](auto qb_c, auto gi_c, int i) -> float {
        constexpr int qb = decltype(qb_c)::value, gi = decltype(gi_c)::value;
        if constexpr (MS == 32) return c32[(gi >> 2) * 2 + qb][4 * (gi & 3) + i];
        else return c16[gi * 4 + qb][i];
    };
    auto grp_row = [](int gi) -> int { return MS == 32 ? (gi >> 2) * 32 + 8 * (gi & 3) : gi * 16; };
    // rows past the end of the shard and dead rows (tombstones, `where` filters) are struck out of the finished
    // tile, not out of the accumulator init: the common tile (full, no mask) pays nothing
    auto strike = [&](unsigned m0, unsigned m1, const int sub) {
        static_for<NQB>([&](auto qb_c) {
            static_for<NGRP>([&](auto gi_c) {
                constexpr int qb = decltype(qb_c)::value, gi = decltype(gi_c)::value;
                constexpr int r0 = MS == 32 ? (gi >> 2) * 32 + 8 * (gi & 3) : gi * 16;
                const unsigned m = (r0 < 32 ? m0 : m1) >> ((r0 & 31) + 4 * sub);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ok = (m >> i) & 1u;
                    if constexpr (MS == 32) {
                        constexpr int a = (gi >> 2) * 2 + qb, e = 4 * (gi & 3);
                        c32[a][e + i] = ok ? c32[a][e + i] : NEG_INF;
                    } else {
                        c16[gi * 4 + qb][i] = ok ? c16[gi * 4 + qb][i] : NEG_INF;
                    }
                }
            });
        });
    };

    // ---- selection: lane-local, lists in LDS, entered only when some lane's score reaches its threshold ---
    auto select = [&](auto qb_c, const int row_base, float *const my_lists) {
        auto lst_v = [&](int qb, int i) -> float & { return my_lists[(qb * 2 * K + i) * QS_QROWS]; };
        auto lst_r = [&](int qb, int i) -> int & { return *(int *)&my_lists[(qb * 2 * K + K + i) * QS_QROWS]; };
        constexpr int qb = decltype(qb_c)::value;
        // (the common case ends here: a wave alone on its SIMD pays every instruction of this test in matrix-pipe idle
        // time -- v_max3 on the raw accumulators in one statement, not fmaxf, which quiets each MFMA result first: 9 instructions instead of 21)
        auto max16 = [&](auto g0_c) -> float {   // groups g0 .. g0 + 3
            constexpr int g0 = decltype(g0_c)::value;
            using G0 = std::integral_constant<int, g0>;
            using G1 = std::integral_constant<int, g0 + 1>;
            using G2 = std::integral_constant<int, g0 + 2>;
            using G3 = std::integral_constant<int, g0 + 3>;
            return max16_raw(score(qb_c, G0{}, 0), score(qb_c, G0{}, 1), score(qb_c, G0{}, 2), score(qb_c, G0{}, 3),
                             score(qb_c, G1{}, 0), score(qb_c, G1{}, 1), score(qb_c, G1{}, 2), score(qb_c, G1{}, 3),
                             score(qb_c, G2{}, 0), score(qb_c, G2{}, 1), score(qb_c, G2{}, 2), score(qb_c, G2{}, 3),
                             score(qb_c, G3{}, 0), score(qb_c, G3{}, 1), score(qb_c, G3{}, 2), score(qb_c, G3{}, 3));
        };
        float mx = max16(std::integral_constant<int, 0>{});
        if constexpr (NGRP == 8) mx = fmaxf(mx, max16(std::integral_constant<int, 4>{}));
        float t = thr[qb];
        if (__builtin_amdgcn_ballot_w64(mx >= t) == 0ull) return;
        // Which of this lane's NE scores reach its threshold.  While the thresholds are cold almost every element has
        // SOME lane of the wave above its threshold, but a lane has only a few: walking the elements (below, `dense`)
        // pays an insertion per element, walking each lane's own hits pays one per round of the slowest lane.
        constexpr int NE = NGRP * 4;
        // A lane that has no threshold yet (the first tiles of the walk) would pass every score.  Only its K best of
        // this tile can enter its list: raise the bar to (a lower bound of) the K-th largest of the lane's NE scores --
        // K rounds of "largest score below the last one" (equal scores count once: the bar only gets lower), trees of
        // independent max operations: issue-bound, where insertions are latency-bound, and three live registers (a
        // sorting network on copies was faster still but took sixteen, and hipcc then spilled a Q fragment).
        // Scores equal to the bar still pass and are settled in row order by the insertions.
        float bar = t;
        if (__builtin_amdgcn_ballot_w64(t == NEG_INF) != 0ull) {
            float prev = INFINITY;
#pragma unroll
            for (int r = 0; r < K; ++r) {
                float m = NEG_INF;
                static_for<NGRP>([&](auto gi_c) {
                    const float x0 = score(qb_c, gi_c, 0), x1 = score(qb_c, gi_c, 1);
                    const float x2 = score(qb_c, gi_c, 2), x3 = score(qb_c, gi_c, 3);
                    const float y0 = x0 < prev ? x0 : NEG_INF, y1 = x1 < prev ? x1 : NEG_INF;
                    const float y2 = x2 < prev ? x2 : NEG_INF, y3 = x3 < prev ? x3 : NEG_INF;
                    m = fmaxf(m, fmaxf(fmaxf(y0, y1), fmaxf(y2, y3)));
                });
                prev = m;
            }
            bar = fmaxf(t, prev);
        }
        unsigned mask = 0;
        static_for<NE>([&](auto e_c) {
            constexpr int e = decltype(e_c)::value;
            mask |= score(qb_c, std::integral_constant<int, e / 4>{}, e % 4) >= bar ? (1u << e) : 0u;
        });
        TopList<K> L;   // (loaded only now: the network above wants the registers)
#pragma unroll
        for (int i = 0; i < K; ++i) {
            L.v[i] = lst_v(qb, i);
            L.r[i] = lst_r(qb, i);
        }
        constexpr int DENSE_AT = NE == 16 ? 8 : 12;
        if (__builtin_amdgcn_ballot_w64(__builtin_popcount(mask) > DENSE_AT) == 0ull) {
            while (__builtin_amdgcn_ballot_w64(mask != 0u) != 0ull) {
                const bool valid = mask != 0u;
                const unsigned e = (unsigned)__builtin_ctz(mask | (1u << (NE - 1)));   // lowest row first
                mask &= mask - 1u;
                float lvl[NE];   // score e of this lane by a binary tree of selects (every index a constant)
                static_for<NE>([&](auto e_c) {
                    constexpr int x = decltype(e_c)::value;
                    lvl[x] = score(qb_c, std::integral_constant<int, x / 4>{}, x % 4);
                });
#pragma unroll
                for (int w = NE / 2, b = 0; w >= 1; w >>= 1, ++b) {
                    const bool bit = (e >> b) & 1u;
#pragma unroll
                    for (int i = 0; i < w; ++i) lvl[i] = bit ? lvl[2 * i + 1] : lvl[2 * i];
                }
                const float sc = lvl[0];
                const int row = MS == 32 ? (int)(((e >> 4) << 5) + (((e >> 2) & 3u) << 3) + (e & 3u))
                                         : (int)(((e >> 2) << 4) + (e & 3u));
                L.insert_strict_flat(valid && sc >= t ? sc : NEG_INF, row_base + row);
                t = fmaxf(t, L.v[K - 1]);
            }
        } else {
            static_for<NGRP>([&](auto gi_c) {
                constexpr int gi = decltype(gi_c)::value;
                const float a0 = score(qb_c, gi_c, 0), a1 = score(qb_c, gi_c, 1);
                const float a2 = score(qb_c, gi_c, 2), a3 = score(qb_c, gi_c, 3);
                if (__builtin_amdgcn_ballot_w64(fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)) >= t) != 0ull) {
#pragma clang loop unroll(disable)
                    for (int i = 0; i < 4; ++i) {
                        const float s = i == 0 ? a0 : (i == 1 ? a1 : (i == 2 ? a2 : a3));
                        const bool pass = s >= t;
                        if (__builtin_amdgcn_ballot_w64(pass) != 0ull) {
                            L.insert_strict(pass ? s : NEG_INF, row_base + grp_row(gi) + i);
                            t = fmaxf(t, L.v[K - 1]);
                        }
                    }
                }
            });
        }
        // k-th best of the union of this lane's list and its neighbour's (the lanes that share the query hold
        // different rows): a lower bound of the final k-th score; with four lanes per query, the better of the two pairs
        float u = fmaxf(L.v[K - 1], __shfl_xor(L.v[K - 1], QW));
#pragma unroll
        for (int i = 0; i + 1 < K; ++i) u = fmaxf(u, fminf(L.v[i], __shfl_xor(L.v[K - 2 - i], QW)));
        if constexpr (NSUB == 4) u = fmaxf(u, __shfl_xor(u, 32));
        thr[qb] = fmaxf(t, u);
#pragma unroll
        for (int i = 0; i < K; ++i) {
            lst_v(qb, i) = L.v[i];
            lst_r(qb, i) = L.r[i];
        }
    };

    // ---- threshold exchange (see the head of the file).  Everything here is issued at tile boundaries; what it reads
    // from other workgroups comes in by LDS-DMA and is consumed one tile later. --------------------------------
    const bool seeding = p.pub_best != nullptr;
    const int n_resp = (QS_QROWS + walkers - 1) / walkers;   // queries this workgroup computes the threshold of
    auto publish_bests = [&](const int ql, const int sub, float *const my_lists) {
        auto lst_v = [&](int qb, int i) -> float & { return my_lists[(qb * 2 * K + i) * QS_QROWS]; };
        float *dst = p.pub_best + ((size_t)by * walkers + bx) * QS_QROWS + wave * 64 + ql;
        static_for<NQB>([&](auto qb_c) {
            constexpr int qb = decltype(qb_c)::value;
            float v = lst_v(qb, 0);   // the workgroup's best row of a query is in one of its lanes' lists
            v = fmaxf(v, __shfl_xor(v, QW));
            if constexpr (NSUB == 4) v = fmaxf(v, __shfl_xor(v, 32));
            if (sub == 0) __hip_atomic_store(dst + qb * QW, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        });
    };
    auto gather_issue = [&](int q) {   // one wave: the bests of query q, walker w at gather_lds[w]
        const int ln = lane_now();
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(p.pub_best + (size_t)by * walkers * QS_QROWS), 0, (unsigned)(walkers * QS_QROWS * 4), 0x00020000);
#pragma unroll
        for (int c = 0; c < 4; ++c)   // (walkers past the group's last: out of the descriptor's range, nothing fetched)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)((char *)gather_lds + c * 256), 4,
                                                     (unsigned)(((c * 64 + ln) * QS_QROWS + q) * 4), 0, 0, CP_SC1);
    };
    auto compute_thr = [&](int q) {    // the same wave, one tile later
        const int ln = lane_now();
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = (c * 64 + ln) < walkers ? gather_lds[c * 64 + ln] : NEG_INF;
        const float t = wave_kth_largest<K>(v[0], v[1], v[2], v[3], ln);
        if (ln == 0)
            __hip_atomic_store(p.pub_thr + (size_t)by * QS_QROWS + q, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto poll_issue = [&]() {          // one wave: the 256 thresholds of this query group
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(p.pub_thr + (size_t)by * QS_QROWS), 0, QS_QROWS * 4, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)poll_lds, 16, (unsigned)(lane_now() * 16), 0, 0, CP_SC1);
    };
    auto poll_consume = [&](const int ql) {
        static_for<NQB>([&](auto qb_c) {
            constexpr int qb = decltype(qb_c)::value;
            thr[qb] = fmaxf(thr[qb], poll_lds[wave * 64 + qb * QW + ql]);
        });
    };
    // Two rounds: bests after the first tile (a 64-row sample per workgroup) and again after tile pub1 (stragglers, and
    // lists that have seen several tiles).  The responsible wave fetches the bests of its j-th query STEP tiles apart,
    // starting the tile after the publication, and reduces them LANDT tiles later; thresholds are polled every tile
    // while the rounds are fresh, every fourth tile afterwards.
    constexpr int STEP = LANDT;
    const int span = n_resp * STEP + LANDT + 1;     // tiles a round keeps the gather buffer busy
    const int pub1 = span + 1 > 5 ? span + 1 : 5;
    auto polled_at = [&](int t) -> bool { return t >= 1 + LANDT && (t < pub1 + span + 4 || (t & 3) == 3); };
    const int quiet = pub1 + span + 4 + LANDT;      // from this tile on nothing but the periodic poll happens

    // ---- main loop.  One continuous software pipeline over statements: statement s issues the LDS reads of
    // statement s+1 (which may belong to the next ring stage or the next tile) before its own MFMAs.  The stage
    // hand-over (counted vmcnt for the next stage's DMA + s_barrier + refill of the stage just drained) sits after
    // the SECOND-TO-LAST statement of a stage: by then every LDS read of the stage has returned in every wave. ----
    unsigned long long t0c = 0, t0r = 0;
    if (p.dbg & DBG_QS_CLOCK) {
        t0c = __builtin_amdgcn_s_memtime();
        t0r = __builtin_amdgcn_s_memrealtime();
    }
    if (any_tile) {
        FT fa0, fa1, fb0, fb1;  // A fragments of the current / next statement
        wait_vmcnt<(NST - 1) * PPS>();
        __builtin_amdgcn_s_barrier();
        {
            const char *st = smem + lane_off0;
            fa0 = *(const FT *)(st);
            fa1 = *(const FT *)(st + (MS == 32 ? 32 * SLAB : 16 * SLAB));
        }
        int cur_tile = bx;
        for (int ti = 0; (unsigned)cur_tile < (unsigned)p.n_tiles; ++ti) {
            // (16x16x32) fragment addresses of the stage being read; made again here so that nothing more than the lane
            // constant is live across the selection at the end of a tile
            unsigned adr_lo = 0, adr_hi = 0;
            if constexpr (MS == 16)
                qsw_stage_addr(adr_lo, adr_hi, lane_off0, smem_base + (unsigned)(((ti * SPT) % NST) * STAGE));
            // ticket for the position LOOK tiles ahead: the atomic's return value is picked up at the end of this tile
            // (hipcc waits for it with a counted vmcnt: two dozen younger DMA pieces by then)
            // As asm, lane 0 only: hipcc would wait vmcnt(0) for a builtin atomic's result at once (it feeds a phi
            // behind the lane-0 branch), i.e. drain the DMA ring.  The value lands in lane 0 of `tkv` before the last
            // hand-over wait of this tile (counted vmcnt: two dozen younger DMA pieces by then); it is read only by
            // the asm store at the end of the tile (tests/test_kernel_codegen.py checks that the register is not
            // touched in between).
            unsigned tkv;
            const bool want_ticket = ti + LOOK >= p_static && wave == 0;
            if (want_ticket) {
                unsigned long long saved_exec;
                asm volatile("s_mov_b64 %1, exec\n\t"
                             "s_mov_b64 exec, 1\n\t"
                             "global_atomic_add %0, %2, %3, %4 sc0\n\t"
                             "s_mov_b64 exec, %1"
                             : "=&v"(tkv), "=&s"(saved_exec)
                             : "v"(0u), "v"(1u), "s"(p.tickets + by)
                             : "memory");
            }
            static_for<SPT>([&](auto sg_c) {
                constexpr int sg = decltype(sg_c)::value;
                const int it = ti * SPT + sg;
                const unsigned st_cur = smem_base + (unsigned)((it % NST) * STAGE);
                const unsigned st_nxt = smem_base + (unsigned)(((it + 1) % NST) * STAGE);
                static_for<KSTG>([&](auto j_c) {
                    constexpr int j = decltype(j_c)::value;       // statement of the stage
                    constexpr int g = j / 4, m = j % 4;
                    constexpr bool last = j == KSTG - 1;           // the next statement opens the next stage
                    constexpr int nj = last ? 0 : j + 1;
                    constexpr int ng = nj / 4, nm = nj % 4;
                    constexpr int slab = sg * G + g;               // K-slab of the tile
                    const unsigned st = last ? st_nxt : st_cur;
                    // A fragments alternate between (fa0, fa1) and (fb0, fb1); a stage has an even number of statements
                    FT &x0 = (j & 1) ? fb0 : fa0, &x1 = (j & 1) ? fb1 : fa1;
                    FT &y0 = (j & 1) ? fa0 : fb0, &y1 = (j & 1) ? fa1 : fb1;
                    if constexpr (MS == 32) {
                        constexpr int ks = slab * 4 + m;
                        // chunk (2 nm + h) ^ sw = (h ^ sw) ^ 2 nm: bit 1 of nm in the lane constant, bit 0 in the statement
                        qsw_stmt32<DT, (ks < QA_STEPS), ks == 0, (nm & 1) * 32, ng * SLABB, ng * SLABB + 32 * SLAB, FT>(
                            c32[0], c32[1], c32[2], c32[3], x0, x1, qf[0][ks], qf[1][ks], y0, y1,
                            (nm & 2) ? lane_off64 : lane_off0, st);
                    } else {
                        constexpr int ks = slab * 2 + (m >> 1), rp = m & 1;
                        constexpr int o0 = ng * SLABB + (nm & 1) * (32 * SLAB);
                        if constexpr (last) qsw_stage_addr(adr_lo, adr_hi, lane_off0, st_nxt);
                        qsw_stmt16<DT, (ks < QA_STEPS), ks == 0, o0, o0 + 16 * SLAB, FT>(
                            c16[rp * 8 + 0], c16[rp * 8 + 1], c16[rp * 8 + 2], c16[rp * 8 + 3], c16[rp * 8 + 4],
                            c16[rp * 8 + 5], c16[rp * 8 + 6], c16[rp * 8 + 7], x0, x1, qf[0][ks], qf[1][ks], qf[2][ks],
                            qf[3][ks], y0, y1, (nm >> 1) ? adr_hi : adr_lo);
                    }
                    if constexpr (j == KSTG - 2) {
                        // every read of this stage is back: hand the ring over.  The stage just drained is
                        // refilled over the next KSTG statements (pieces after statements KSTG-1, 1, 3, ...)
                        wait_vmcnt<(NST - 2) * PPS>();
                        if (!dev_dbg(DBG_QS_NO_BARRIER)) __builtin_amdgcn_s_barrier();
                        next_fill(std::integral_constant<int, (NST + sg) % SPT>{}, it % NST);
                    }
                    // (the refill opened in this stage fetches K-slabs (NST + sg) % SPT of its tile; pieces 1.. of the
                    // one opened a stage earlier are still going out)
                    if constexpr (j == KSTG - 1)
                        issue_piece(std::integral_constant<int, (NST + sg) % SPT>{}, std::integral_constant<int, 0>{});
                    else if constexpr ((j & 1) == 1)
                        issue_piece(std::integral_constant<int, (NST + sg + SPT - 1) % SPT>{},
                                    std::integral_constant<int, (j + 1) / 2>{});
                });
            });
            asm volatile("s_nop 15\n\ts_nop 7");  // last MFMA's D -> VALU readers
            const long long row0 = (long long)cur_tile * R;
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
                strike(m0, m1, lane_now() / QW);
            }
            // the lane's place in the wave, again (see lane_now)
            const int ln = lane_now();
            const int ql_t = ln & (QW - 1), sub_t = ln / QW;
            float *const lists_t = lists + wave * 64 + ln;
            const int row_base = (int)row0 + 4 * sub_t;
            if (!(p.dbg & DBG_QS_NO_SELECT)) {
                static_for<NQB>([&](auto qb_c) { select(qb_c, row_base, lists_t); });
            } else {
                if constexpr (MS == 32) asm volatile("" ::"v"(c32[0]), "v"(c32[1]), "v"(c32[2]), "v"(c32[3]));
                else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) asm volatile("" ::"v"(c16[e]));
                }
            }
            if (seeding && ti >= quiet) {
                // both rounds are over: what is left is the poll every fourth tile (a workgroup that ran late publishes
                // late) -- the same schedule as below, spelled out because the tests below cost a dozen scalar
                // instructions per tile
                if (((ti - LANDT) & 3) == 3) poll_consume(ql_t);
                if ((ti & 3) == 3 && wave == 0) poll_issue();
            } else if (seeding) {
                if (ti >= LANDT && polled_at(ti - LANDT)) poll_consume(ql_t);
                if (ti == 0 || ti == pub1) publish_bests(ql_t, sub_t, lists_t);
                if (wave == 0) {
                    // responsible query j of a round: fetched at tile (round's publication) + 1 + j STEP, reduced LANDT
                    // tiles later (STEP = LANDT: the reduction of query j and the fetch of query j + 1 share a tile end,
                    // reduction first)
                    const int rel = ti > pub1 ? ti - (pub1 + 1) : ti - 1;
                    const int jc = rel - LANDT, jg = rel;
                    if (jc >= 0 && jc % STEP == 0 && jc / STEP < n_resp && bx + (jc / STEP) * walkers < QS_QROWS)
                        compute_thr(bx + (jc / STEP) * walkers);
                    if (jg >= 0 && jg % STEP == 0 && jg / STEP < n_resp && bx + (jg / STEP) * walkers < QS_QROWS)
                        gather_issue(bx + (jg / STEP) * walkers);
                    if (polled_at(ti)) poll_issue();
                }
            }
            if ((p.dbg & DBG_QS_CLOCK) && p.stamps != nullptr && ti < 28 && wave == 0 && ln == 0)
                p.stamps[32 * blockIdx.x + 4 + ti] = __builtin_amdgcn_s_memrealtime();   // end of tile ti
            if (want_ticket) {
                // the atomic is older than the tile's SPT * PPS ring pieces and nothing else: this wait returns at once
                // unless the ticket is late (the hand-over waits alone do not cover it when a tile is only a dozen
                // pieces: 384-d rows lost a tail tile once in a few hundred launches)
                wait_vmcnt<SPT * PPS>();
                unsigned long long saved_exec;
                const unsigned slot = (unsigned)(size_t)(lds_ptr_t)(tk_lds + ((ti + LOOK) & 7));
                asm volatile("s_mov_b64 %0, exec\n\t"
                             "s_mov_b64 exec, 1\n\t"
                             "ds_write_b32 %1, %2\n\t"
                             "s_mov_b64 exec, %0"
                             : "=&s"(saved_exec)
                             : "v"(slot), "v"(tkv)
                             : "memory");
            }
            cur_tile = pos_tile(ti + 1);
        }
    }
    const int tid_e = wave * 64 + lane_now();   // (not `tid`: nothing per-lane stays live across the tile loop for the end)
    if ((p.dbg & DBG_QS_CLOCK) && p.stamps != nullptr && tid_e == 0) {
        p.stamps[32 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0c;
        p.stamps[32 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - t0r;
        p.stamps[32 * blockIdx.x + 2] = t0r - t_entry;
        p.stamps[32 * blockIdx.x + 3] = t_entry;
    }
    // ---- merge the lists of the lanes that share a query, write ONE list per query per workgroup ------------
    __syncthreads();
    if (q0 + tid_e < p.B) {
        // query t of the workgroup sits in wave t >> 6, block (t / QW) & (NQB - 1), lane column t & (QW - 1)
        const int w = tid_e >> 6, qb = (tid_e / QW) & (NQB - 1), c = tid_e & (QW - 1);
        const float *la = lists + w * 64 + c + (qb * 2 * K) * QS_QROWS;
        TopList<K> m;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            m.v[i] = la[i * QS_QROWS];
            m.r[i] = __float_as_int(la[(K + i) * QS_QROWS]);
        }
        for (int s = 1; s < NSUB; ++s) {
            const float *lb = la + s * QW;
            for (int i = 0; i < K; ++i) {
                const float x = lb[i * QS_QROWS];
                const int xr = __float_as_int(lb[(K + i) * QS_QROWS]);
                if (xr == INT_MAX || !better(x, xr, m.v[K - 1], m.r[K - 1])) break;
                m.insert_ordered(x, xr);
            }
        }
        const size_t base = ((size_t)(q0 + tid_e) * p.n_lists + bx) * K;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            p.cand_s[base + i] = m.v[i];
            p.cand_r[base + i] = m.r[i];
        }
        if (bx == 0 && p.n_lists > walkers) {
            // the list slot the slab-ring plan keeps for its sample pass: nothing to put there
            const size_t seed = ((size_t)(q0 + tid_e) * p.n_lists + (p.n_lists - 1)) * K;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                p.cand_s[seed + i] = NEG_INF;
                p.cand_r[seed + i] = INT_MAX;
            }
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

// The product dispatches the 16x16x32 shape at list depth 5 (3.7 % ahead of 32x32x16 on the bare
// walk in steady state, and hipcc keeps its tile loop free of spills; with the f32x16 accumulator tuples of the 32x32
// shape it does not).  MMRAG_QSW_DEV builds (tools/ab_search.py) instantiate both shapes for 768-d fp16 only.
bool qsw_supported(int dtype, unsigned row_bytes, int K, int mfma) {
    if (dtype != MMRAG_F16 && dtype != MMRAG_BF16) return false;
    if (K != 5) return false;   // four lists per query at K = 10 leave no LDS to the ring: search_qs.hip keeps that depth
    const unsigned nk = row_bytes / SLAB;
#if defined(MMRAG_QSW_DEV)
    return (mfma == 16 || mfma == 32) && row_bytes % SLAB == 0 && nk == 12 && dtype == MMRAG_F16;
#else
    return mfma == 16 && row_bytes % SLAB == 0 && (nk == 6 || nk == 8 || nk == 12);
#endif
}

template <int DT, int K, int MS>
static int qsw_launch_nk(const KParams &p, int grid, hipStream_t s) {
    const bool nt = !p.share_l2;
    switch (p.row_bytes / SLAB) {
#if !defined(MMRAG_QSW_DEV)
        case 6:
            if (nt) cosine_topk_walk_kernel<DT, 6, K, MS, true><<<grid, 256, 0, s>>>(p);
            else cosine_topk_walk_kernel<DT, 6, K, MS, false><<<grid, 256, 0, s>>>(p);
            break;
        case 8:
            if (nt) cosine_topk_walk_kernel<DT, 8, K, MS, true><<<grid, 256, 0, s>>>(p);
            else cosine_topk_walk_kernel<DT, 8, K, MS, false><<<grid, 256, 0, s>>>(p);
            break;
#endif
        case 12:
            if (nt) cosine_topk_walk_kernel<DT, 12, K, MS, true><<<grid, 256, 0, s>>>(p);
            else cosine_topk_walk_kernel<DT, 12, K, MS, false><<<grid, 256, 0, s>>>(p);
            break;
        default: return MMRAG_EUNSUPPORTED;
    }
    return MMRAG_OK;
}

int qsw_launch(int dtype, int K, int mfma, const KParams &p, int grid_x, int grid_y, hipStream_t s) {
    if (!qsw_supported(dtype, p.row_bytes, K, mfma)) return MMRAG_EUNSUPPORTED;
    KParams kp = p;
    kp.walkers = grid_x;
    kp.share_l2 = grid_y > 1;
    const int grid = grid_x * grid_y;
#if defined(MMRAG_QSW_DEV)
    if (mfma == 32) return qsw_launch_nk<MMRAG_F16, 5, 32>(kp, grid, s);
    return qsw_launch_nk<MMRAG_F16, 5, 16>(kp, grid, s);
#else
    if (dtype == MMRAG_F16) return qsw_launch_nk<MMRAG_F16, 5, 16>(kp, grid, s);
    return qsw_launch_nk<MMRAG_BF16, 5, 16>(kp, grid, s);
#endif
}

}  // namespace mmrag_impl
