// Transformer encoder forward for gfx950 (MI355X): the embed half of the hot path.
//
// Replaces SentenceTransformer.encode(texts, normalize_embeddings=True)
// (reference call site app/utils/embedder.py:397-403; model all-MiniLM-L6-v2 / bge-base /
// CLIP towers per config.py:102-106).  fp16 weights and activations, fp32 accumulation and
// fp32 LayerNorm / softmax / pooling statistics.  Sequences are packed (no padded tokens):
// x is [T, H] with sequence b owning rows cu_seqlens[b] .. cu_seqlens[b+1].
//
// Kernels
//   embed_ln_kernel   gather tok+pos(+type) rows, LayerNorm                    (HBM-bound)
//   linear_kernel     out = act(x . Wt^T + bias) (+ residual); MFMA 32x32x16 f16, LDS-DMA
//                     slab ring shared with the search kernel (tile_dma.h); epilogue
//                     transposes the tile through LDS for whole-row 512-byte stores  (MFMA-bound)
//   attention_kernel  flash-style, per (sequence, head, 128-query tile): S^T = K.Q^T so the
//                     query sits on the lane and softmax state is lane-local; P feeds the
//                     P.V MFMA straight from the accumulator registers; V is transposed
//                     while it is staged into LDS                               (MFMA-bound)
//   layernorm_kernel  row LayerNorm (residual already added by linear_kernel)   (HBM-bound)
//   pool_norm_kernel  masked mean / CLS / last-token pooling (+ LayerNorm for CLIP), L2 norm
#include "mmrag_internal.h"
#include "tile_dma.h"

#include <atomic>
#include <chrono>
#include <type_traits>

#include <limits.h>
#include <stdlib.h>

using namespace mmrag;

namespace mmrag_impl {

constexpr float NEG_INF_F = -__builtin_inff();
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
__device__ inline float2 h2f(half2_t t) { return make_float2((float)t[0], (float)t[1]); }
__device__ inline half2_t f2h(float a, float b) {
    half2_t r;
    r[0] = (_Float16)a;
    r[1] = (_Float16)b;
    return r;
}

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

template <int ACT>
__device__ inline float act_apply(float x) {
    if constexpr (ACT == MMRAG_ACT_GELU) {
        // erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the fp16 output step): one rcp, one
        // exp2 and six fma instead of libm's branchy erff, which cost 12 us per 256x256 tile -- more than
        // half of the K=768 main loop
        const float z = fabsf(x) * 0.70710678118654752f;
        const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
        const float poly =
            t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
        const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
        const float erf_abs = fmaf(-poly, e, 1.0f);
        return 0.5f * x * (1.0f + copysignf(erf_abs, x));
    }
    if constexpr (ACT == MMRAG_ACT_QUICK_GELU) return x / (1.0f + __expf(-1.702f * x));
    return x;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over rows of H fp16 (one wave per row, fp32 statistics)
// ---------------------------------------------------------------------------------------------
template <int MAXV>  // MAXV 16-byte chunks (8 fp16) per lane: H <= 512 * MAXV, H % 8 == 0
__device__ inline void ln_row(const _Float16 *src, _Float16 *dst, const float *g, const float *b, int H, float eps,
                              int lane, const _Float16 *add1 = nullptr, const _Float16 *add2 = nullptr) {
    float v[MAXV][8];
    const int nchunk = H >> 3;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
        if (c < nchunk) {
            const half8_t t = ((const half8_t *)src)[c];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[i][e] = (float)t[e];
            if (add1) {
                const half8_t a = ((const half8_t *)add1)[c];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[i][e] += (float)a[e];
            }
            if (add2) {
                const half8_t a = ((const half8_t *)add2)[c];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[i][e] += (float)a[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[i][e];
        }
    }
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nchunk) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float dx = v[i][e] - mean;
                q += dx * dx;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nchunk) {
            const f32x4_t g0 = ((const f32x4_t *)g)[2 * c], g1 = ((const f32x4_t *)g)[2 * c + 1];
            const f32x4_t b0 = ((const f32x4_t *)b)[2 * c], b1 = ((const f32x4_t *)b)[2 * c + 1];
            half8_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (_Float16)((v[i][e] - mean) * rstd * g0[e] + b0[e]);
                o[4 + e] = (_Float16)((v[i][4 + e] - mean) * rstd * g1[e] + b1[e]);
            }
            ((half8_t *)dst)[c] = o;
        }
    }
}

__global__ __launch_bounds__(256) void layernorm_kernel(const _Float16 *__restrict__ x, _Float16 *__restrict__ out,
                                                         const float *__restrict__ g, const float *__restrict__ b,
                                                         int T, int H, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    ln_row<2>(x + (size_t)row * H, out + (size_t)row * H, g, b, H, eps, lane);
}

// out[t] = LN(tok[ids[t]] + pos[pos_ids[t]] (+ type[0]))      (BERT embeddings, post-LN)
// with g == nullptr: plain sum (CLIP text: no embedding LayerNorm)
__global__ __launch_bounds__(256) void embed_ln_kernel(const int *__restrict__ ids, const int *__restrict__ pos_ids,
                                                        const _Float16 *__restrict__ tok,
                                                        const _Float16 *__restrict__ pos,
                                                        const _Float16 *__restrict__ type0,
                                                        const float *__restrict__ g, const float *__restrict__ b,
                                                        _Float16 *__restrict__ out, int T, int H, int vocab,
                                                        int max_pos, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    int id = ids[row];
    int ps = pos_ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    ps = ps < 0 ? 0 : (ps >= max_pos ? max_pos - 1 : ps);
    const _Float16 *a = tok + (size_t)id * H;
    const _Float16 *p2 = pos + (size_t)ps * H;
    if (g != nullptr) {
        ln_row<2>(a, out + (size_t)row * H, g, b, H, eps, lane, p2, type0);
    } else {
        for (int c = lane; c < (H >> 1); c += 64) {
            float2 v = h2f(((const half2_t *)a)[c]);
            const float2 w = h2f(((const half2_t *)p2)[c]);
            ((half2_t *)(out + (size_t)row * H))[c] = f2h(v.x + w.x, v.y + w.y);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// linear: out[M, N] = act(x[M, K] . wt[N, K]^T + bias[N]) (+ resid[M, N])
// MFMA orientation: D[feature][token]: features in the accumulator registers (MFMA A = weight
// rows), tokens on the lanes (MFMA B = activation rows).
// ---------------------------------------------------------------------------------------------
struct LinearParams {
    const char *x;
    const char *wt;
    const float *bias;
    const _Float16 *resid;
    _Float16 *out;
    int M, N, K;
    int act;
    int xcd_order;
    unsigned dbg;
    // LayerNorm folded into the single-query GEMMs (linear_small16_kernel only; "the single-query path without LayerNorm
    // launches" below).  ln_gamma != null: x holds UN-normalised rows; the kernel normalises them (gamma, beta, ln_eps) as
    // it loads them and workgroup 0 leaves (mean, rstd) per row in ln_stats_out.  res_stats != null: resid holds
    // un-normalised rows whose (mean, rstd) are in res_stats; they are normalised (res_gamma, res_beta) on their way
    // into the add.
    const float *ln_gamma, *ln_beta;
    float ln_eps;
    float2 *ln_stats_out;
    const float2 *res_stats;
    const float *res_gamma, *res_beta;
};

template <int BF, int BT, int WF, int WT, int NSTAGE, bool PIPE = false>
__global__ __launch_bounds__(64 * WF *WT, (WF * WT) >= 4 ? (WF * WT) / 4 : 1) void linear_kernel(const LinearParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NW = WF * WT;
    constexpr int RF = BF / (32 * WF);
    constexpr int RT = BT / (32 * WT);
    constexpr int STAGE = (BF + BT) * SLAB;
    constexpr int PIECES = (BF + BT) / 8;
    constexpr int PW = PIECES / NW;  // DMA pieces per wave per ring item
    constexpr int W_PIECES = BF / 8;
    constexpr int STG_ROW = BF * 2 + 8;  // staged C tile: [token][feature] fp16, padded rows
    constexpr int STG_BYTES = BT * STG_ROW;
    constexpr int RING_BYTES = NSTAGE * STAGE;
    constexpr int LDS_BYTES = RING_BYTES > STG_BYTES ? RING_BYTES : STG_BYTES;
    static_assert(PIECES % NW == 0 && (W_PIECES % PW == 0 || PW % W_PIECES == 0), "piece split");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wf = wave / WT;
    const int wt_ = wave % WT;
    const int r32 = lane & 31;
    const int h = lane >> 5;

    const int tiles_f = (p.N + BF - 1) / BF;
    // XCD-aware order: workgroup i runs on XCD i % 8, each with its own L2.  Give every XCD a contiguous
    // band of tiles, so the tiles_f workgroups that share an x row-tile (and run at about the same time)
    // share one L2 instead of fetching that tile through eight.
    int lid = blockIdx.x;
    if (p.xcd_order) {
        const int total = gridDim.x, per = total / 8, rem = total % 8;
        const int xcd = blockIdx.x % 8, idx = blockIdx.x / 8;
        lid = xcd * per + (xcd < rem ? xcd : rem) + idx;
    }
    const int tile_f = lid % tiles_f;
    const int tile_t = lid / tiles_f;
    const int f0 = tile_f * BF;
    const int t0 = tile_t * BT;
    const unsigned RB = (unsigned)p.K * 2u;
    const int nk = (int)(RB / SLAB);

    const int f_left = p.N - f0, t_left = p.M - t0;
    const __amdgpu_buffer_rsrc_t rsrc_w =
        make_rsrc(p.wt + (size_t)f0 * RB, (unsigned)((f_left < BF ? f_left : BF) * (long long)RB));
    const __amdgpu_buffer_rsrc_t rsrc_x =
        make_rsrc(p.x + (size_t)t0 * RB, (unsigned)((t_left < BT ? t_left : BT) * (long long)RB));

    unsigned src_off[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int piece = wave * PW + i;
        src_off[i] = dma_src_offset(piece < W_PIECES ? piece : piece - W_PIECES, lane, RB);
    }
    auto issue = [&](int item) {
        char *st = smem + (item % NSTAGE) * STAGE;
        const unsigned koff = (unsigned)item * SLAB;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int piece = wave * PW + i;
            if (piece < W_PIECES)
                dma_piece(rsrc_w, st, piece, src_off[i] + koff);
            else
                dma_piece(rsrc_x, st + BF * SLAB, piece - W_PIECES, src_off[i] + koff);
        }
    };

    f32x16_t acc[RF][RT];
#pragma unroll
    for (int a = 0; a < RF; ++a)
#pragma unroll
        for (int b = 0; b < RT; ++b)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.0f;

    const int sw = (r32 >> 1) & 7;
    const int a_base = (wf * RF * 32 + r32) * SLAB;
    const int b_base = BF * SLAB + (wt_ * RT * 32 + r32) * SLAB;

    int issued = 0;
    for (; issued < NSTAGE - 1 && issued < nk; ++issued) issue(issued);
    if constexpr (PIPE) {
        // Software-pipelined form: the fragments of k-step m+1 are read while the MFMAs of step m issue,
        // and the ring barrier sits BEFORE the last k-step's MFMAs of a slab -- those need no LDS, so
        // they cover the barrier wake-up and the read latency of the next slab's first fragments.
        half8_t fa[2][RF], fb[2][RT];
        auto load = [&](half8_t(&a_)[RF], half8_t(&b_)[RT], const char *st, int m) {
            const int off = ((2 * m + h) ^ sw) * 16;
#pragma unroll
            for (int b = 0; b < RT; ++b) b_[b] = *(const half8_t *)(st + b_base + b * (32 * SLAB) + off);
#pragma unroll
            for (int a = 0; a < RF; ++a) a_[a] = *(const half8_t *)(st + a_base + a * (32 * SLAB) + off);
        };
        auto mma = [&](const half8_t(&a_)[RF], const half8_t(&b_)[RT]) {
#pragma unroll
            for (int a = 0; a < RF; ++a)
#pragma unroll
                for (int b = 0; b < RT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_[a], b_[b], acc[a][b], 0, 0, 0);
        };
        // pin the order inside a (reads of the next step, MFMAs of this step) group: one read, one MFMA, ...
        auto interleave = [&]() {
#pragma unroll
            for (int i = 0; i < RF + RT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
                __builtin_amdgcn_sched_group_barrier(0x008, (RF * RT) / (RF + RT), 0);  // MFMA(s)
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        wait_items<PW, NSTAGE - 2>(issued - 1);
        __builtin_amdgcn_s_barrier();
        if (issued < nk) {
            issue(issued);
            ++issued;
        }
        load(fa[0], fb[0], smem, 0);
        for (int it = 0; it < nk; ++it) {
            const char *st = smem + (it % NSTAGE) * STAGE;
            load(fa[1], fb[1], st, 1);
            mma(fa[0], fb[0]);
            interleave();
            load(fa[0], fb[0], st, 2);
            mma(fa[1], fb[1]);
            interleave();
            load(fa[1], fb[1], st, 3);
            mma(fa[0], fb[0]);
            interleave();
            if (it + 1 < nk) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my reads of this stage are complete
                wait_items<PW, NSTAGE - 2>(issued - it - 2);
                __builtin_amdgcn_s_barrier();
                if (issued < nk) {
                    issue(issued);  // into the stage every wave has just finished reading
                    ++issued;
                }
                load(fa[0], fb[0], smem + ((it + 1) % NSTAGE) * STAGE, 0);
            }
            mma(fa[1], fb[1]);
            interleave();
        }
    } else
    for (int it = 0; it < nk; ++it) {
        wait_items<PW, NSTAGE - 2>(issued - it - 1);
        __builtin_amdgcn_s_barrier();
        if (issued < nk) {
            issue(issued);
            ++issued;
        }
        const char *st = smem + (it % NSTAGE) * STAGE;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int off = ((2 * m + h) ^ sw) * 16;
            half8_t bt[RT];
#pragma unroll
            for (int b = 0; b < RT; ++b) bt[b] = *(const half8_t *)(st + b_base + b * (32 * SLAB) + off);
#pragma unroll
            for (int a = 0; a < RF; ++a) {
                const half8_t af = *(const half8_t *)(st + a_base + a * (32 * SLAB) + off);
#pragma unroll
                for (int b = 0; b < RT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bt[b], acc[a][b], 0, 0, 0);
            }
        }
    }

    // ---- epilogue ---------------------------------------------------------------------------
    // residual rows are fetched FIRST (their latency hides under phase 1), 16 bytes per lane
    constexpr int LPR = BF / 8;              // lanes per output row
    constexpr int RPI = 64 / LPR;            // rows per wave instruction
    constexpr int ITERS = BT / (NW * RPI);
    static_assert(BT % (NW * RPI) == 0, "row split");
    const int sub = lane / LPR;
    const int col = (lane % LPR) * 8;
    const bool wide = (p.N & 7) == 0;        // 16-byte rows possible (always for the encoder shapes)
    half8_t rv[ITERS];
    if (wide && p.resid != nullptr) {
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int t = t0 + wave * RPI + sub + i * NW * RPI;
            const int f = f0 + col;
#pragma unroll
            for (int e = 0; e < 8; ++e) rv[i][e] = (_Float16)0.f;
            if (t < p.M && f < p.N) rv[i] = *(const half8_t *)(p.resid + (size_t)t * p.N + f);
        }
    }
    // phase 1: bias + activation in fp32, stage the tile as [token][feature] fp16
    __builtin_amdgcn_s_barrier();  // every wave is done reading the ring
    // (the activation is a compile-time constant inside the unrolled loops: a run-time test per element
    // cost more than the arithmetic)
    auto stage = [&](auto act_tag) {
        constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
        for (int a = 0; a < RF; ++a) {
            const int fl = wf * RF * 32 + a * 32 + 4 * h;  // local feature of register group 0
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f = fl + 8 * g;  // features f .. f+3 live in registers 4g .. 4g+3
                float bias4[4] = {0.f, 0.f, 0.f, 0.f};
                if (p.bias != nullptr && f0 + f + 3 < p.N) {
                    const float4 bv = *(const float4 *)(p.bias + f0 + f);
                    bias4[0] = bv.x, bias4[1] = bv.y, bias4[2] = bv.z, bias4[3] = bv.w;
                }
#pragma unroll
                for (int b = 0; b < RT; ++b) {
                    const int tl = wt_ * RT * 32 + b * 32 + r32;
                    half4_t o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (_Float16)act_apply<ACT>(acc[a][b][4 * g + i] + bias4[i]);
                    *(half4_t *)(smem + tl * STG_ROW + f * 2) = o;
                }
            }
        }
    };
    if (p.act == MMRAG_ACT_GELU) stage(std::integral_constant<int, MMRAG_ACT_GELU>{});
    else if (p.act == MMRAG_ACT_QUICK_GELU) stage(std::integral_constant<int, MMRAG_ACT_QUICK_GELU>{});
    else stage(std::integral_constant<int, MMRAG_ACT_NONE>{});
    __syncthreads();
    // phase 2: whole-row stores
    if (wide) {
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int r = wave * RPI + sub + i * NW * RPI;
            const int t = t0 + r;
            const int f = f0 + col;
            if (t < p.M && f < p.N) {
                const half4_t lo = *(const half4_t *)(smem + r * STG_ROW + col * 2);
                const half4_t hi = *(const half4_t *)(smem + r * STG_ROW + col * 2 + 8);
                half8_t v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = lo[e];
                    v[4 + e] = hi[e];
                }
                if (p.resid != nullptr) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (_Float16)((float)v[e] + (float)rv[i][e]);
                }
                *(half8_t *)(p.out + (size_t)t * p.N + f) = v;
            }
        }
    } else {
        constexpr int LANES_PER_ROW = BF / 4;  // 8 bytes per lane
        constexpr int ROWS_PER_INSTR = 64 / LANES_PER_ROW;
        const int sub4 = lane / LANES_PER_ROW;
        const int col4 = (lane % LANES_PER_ROW) * 4;
        for (int r = wave * ROWS_PER_INSTR + sub4; r < BT; r += NW * ROWS_PER_INSTR) {
            const int t = t0 + r;
            const int f = f0 + col4;
            if (t < p.M && f < p.N) {
                half4_t v = *(const half4_t *)(smem + r * STG_ROW + col4 * 2);
                if (p.resid != nullptr) {
                    const half4_t rv4 = *(const half4_t *)(p.resid + (size_t)t * p.N + f);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (_Float16)((float)v[e] + (float)rv4[e]);
                }
                *(half4_t *)(p.out + (size_t)t * p.N + f) = v;
            }
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// linear, persistent form for big problems (>= one 256x256 tile per CU): one workgroup per CU walks its tiles.
//
// Why: in linear_kernel a CU sits idle from the last MFMA of a tile until its 128 KB of output has been
// accepted by the memory system AND the next workgroup has filled its ring (8.7 us of 29 us per tile at K = 768:
// every CU stores at the same moment, 32 MB in a burst).  Here
//   * the ring never stops: the first K-slabs of the NEXT tile are requested before the last MFMAs of this one,
//   * MFMA orientation is D[token][feature] with the weight rows of a wave's 64-feature block interleaved at
//     DMA time (LDS row j of the block = feature 2j, row 32+j = feature 2j+1), so a lane owns two ADJACENT
//     features of a token: the tile goes from the accumulators to memory as 4-byte stores that cover whole
//     128-byte lines per half-wave -- no LDS staging, no barrier, and the stores drain under the next main loop.
// Same ring, swizzle and fragment reads as linear_kernel<256, 256, 4, 4, 2, true>.
// ---------------------------------------------------------------------------------------------
// pin the order inside one (fragment reads of the next k-step, MFMAs of this k-step) group: a read, its share of the MFMAs, ...
template <int READS, int MFMAS, int I = 0>
__device__ inline void interleave_reads_mfmas() {
    if constexpr (I < READS) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, (MFMAS * (I + 1)) / READS - (MFMAS * I) / READS, 0);
        interleave_reads_mfmas<READS, MFMAS, I + 1>();
    } else {
        __builtin_amdgcn_sched_barrier(0);
    }
}

// MS: MFMA shape.  32 = v_mfma_f32_32x32x16_f16 (a wave's 128 x 64 outputs as 4 x 2 tiles of 32 x 32), 16 =
// v_mfma_f32_16x16x32_f16 (8 x 4 tiles of 16 x 16: the same output tile, accumulator registers and LDS bytes per flop;
// the chip holds a higher clock on it under load -- MI355X_MICROARCH.md, DVFS give-back item 7; bare loops on this
// part: 1.88 vs 1.71 PFLOP/s).  The fragment reads of a 32-deep k-step are split over its two halves (features 0-63
// first, 64-127 second), so the two sets of fragments in flight take 64 registers, not 96.
template <int NWF, int NWT, int MS = 32>  // waves along the features x waves along the tokens of the 256x256 tile
__global__ __launch_bounds__(64 * NWF *NWT, 1) void linear_persistent_kernel(const LinearParams p, const int n_tiles,
                                                                            const int tiles_f) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(MS == 32 || MS == 16, "MFMA shape");
    constexpr int BF = 256, BT = 256;
    constexpr int NW = NWF * NWT;
    constexpr int FT = BF / (32 * NWF);   // MFMA tiles along the features per wave (pairs of them interleave)
    constexpr int TT = BT / (32 * NWT);   // MFMA tiles along the tokens per wave
    constexpr int WAVE_F = BF / NWF, WAVE_T = BT / NWT;
    constexpr int STAGE = (BF + BT) * SLAB;
    constexpr int PW = 64 / NW;           // DMA pieces per wave per ring item: first half of the waves fetch W, second half x
    constexpr int STORES = MS == 32 ? (FT / 2) * TT * 16 : 2 * 4 * 4;   // store instructions of one epilogue, per wave
    static_assert(FT % 2 == 0 && PW >= 4, "layout");
    static_assert(MS == 32 || (WAVE_F == 128 && WAVE_T == 64), "the 16x16x32 form is written for 128 x 64 per wave");
    constexpr int STORES_WAIT = STORES < 63 ? STORES : 63;  // vmcnt is a 6-bit counter
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wf = wave / NWT, wt_ = wave % NWT;
    const int r32 = lane & 31, h = lane >> 5;
    const unsigned RB = (unsigned)p.K * 2u;
    const int nk = (int)(RB / SLAB);
    const bool loads_w = wave < NW / 2;
    const int lw = wave % (NW / 2);       // index among the waves that fetch the same operand

    // LDS row prow of the W tile holds feature perm(prow): inside each block of 64 rows, row j <-> feature 2j,
    // row 32+j <-> feature 2j+1.  Pieces of a wave: rows lw*PW*8 + i*8 + lane/8; the source offset of piece i
    // differs from that of piece (i & 1) by a multiple of 16 rows (swizzle period) -> two VGPRs + scalar steps.
    unsigned src_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int prow = (lw * PW + i) * 8 + (lane >> 3);
        // 16x16x32: FOUR features per lane: inside each block of 64 rows, row 16 b + j <-> feature 4 j + b
        const int srow = !loads_w ? prow
                         : (MS == 32 ? ((prow & ~63) | ((prow & 31) << 1) | ((prow >> 5) & 1))
                                     : ((prow & ~63) | ((prow & 15) << 2) | ((prow >> 4) & 3)));
        src_off[i] = __umul24((unsigned)srow, RB) + (unsigned)(((lane & 7) ^ ((prow >> 1) & 7)) * 16);   // (K < 2^23)
    }
    // piece i = 2q + (i&1): LDS rows advance by 16q
    auto piece_src_rows = [&](int i) -> int {
        const int q = i >> 1;
        if (!loads_w) return 16 * q;
        if constexpr (MS == 16) return (q >> 2) * 64 + (q & 3);   // 16 LDS rows on = the next feature of the four; 64 = next block
        return (q >> 2) * 64 + ((q >> 1) & 1) + (q & 1) * 32;   // 16 LDS rows on = 32 features on; 32 rows on = the odd features; 64 = next block
    };
    char *const my_dst = smem + (loads_w ? 0 : BF * SLAB) + lw * PW * 1024;

    auto tile_origin = [&](int v, int &f0, int &t0) {   // virtual block id -> tile, XCD-banded like linear_kernel
        int lid = v;
        if (p.xcd_order) {
            const int per = n_tiles / 8, rem = n_tiles % 8;
            const int xcd = v % 8, idx = v / 8;
            lid = xcd * per + (xcd < rem ? xcd : rem) + idx;
        }
        f0 = (lid % tiles_f) * BF;
        t0 = (lid / tiles_f) * BT;
    };
    auto my_source = [&](int v, const char *&base, unsigned &bytes) {   // what this wave fetches of tile v
        if (v >= n_tiles) {
            base = p.x;
            bytes = 0;
            return;
        }
        int f0, t0;
        tile_origin(v, f0, t0);
        if (loads_w) {
            const int left = p.N - f0;
            base = p.wt + (size_t)f0 * RB;
            bytes = (unsigned)((left < BF ? left : BF) * (long long)RB);
        } else {
            const int left = p.M - t0;
            base = p.x + ((p.dbg & DBG_LINEAR_X_SAME) ? (size_t)0 : (size_t)t0 * RB);
            bytes = (unsigned)((left < BT ? left : BT) * (long long)RB);
        }
    };
    auto issue = [&](const char *base, unsigned bytes, int stage, int kslab) {
        const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(base, bytes);
        char *dst = my_dst + stage * STAGE;
        const unsigned koff = (unsigned)kslab * SLAB;
#pragma unroll
        for (int i = 0; i < PW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + i * 1024), 16, src_off[i & 1],
                                                     koff + (unsigned)piece_src_rows(i) * RB, 0, 0);
    };

    const int sw = (r32 >> 1) & 7;
    const int w_base = (wf * WAVE_F + r32) * SLAB;               // + 32 rows per MFMA tile
    const int x_base = BF * SLAB + (wt_ * WAVE_T + r32) * SLAB;

    half8_t fw[2][FT], fx[2][TT];
    auto load = [&](half8_t(&w_)[FT], half8_t(&x_)[TT], const char *st, int m) {
        const int off = ((2 * m + h) ^ sw) * 16;
#pragma unroll
        for (int b = 0; b < TT; ++b) x_[b] = *(const half8_t *)(st + x_base + b * (32 * SLAB) + off);
#pragma unroll
        for (int a = 0; a < FT; ++a) w_[a] = *(const half8_t *)(st + w_base + a * (32 * SLAB) + off);
    };
    // acc[a][b]: a = 2*block + parity; register r <-> token 8*(r/4) + 4*h + r%4 of token tile b; lane j <-> features
    // 64*block + 2j + parity
    f32x16_t acc[MS == 32 ? FT : 1][MS == 32 ? TT : 1];
    auto mma = [&](const half8_t(&w_)[FT], const half8_t(&x_)[TT]) {
#pragma unroll
        for (int a = 0; a < FT; ++a)
#pragma unroll
            for (int b = 0; b < TT; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x_[b], w_[a], acc[a][b], 0, 0, 0);
    };
    auto interleave = [&]() { interleave_reads_mfmas<FT + TT, FT * TT>(); };

    // ---- 16x16x32 form: lane (c = lane & 15, g4 = lane >> 4); a 32-deep k-step s of a slab reads chunk (4 s + g4) ^ sw
    // of row c of every 16-row block.  acc16[fb][tb]: feature block fb (LDS rows wf*128 + 16 fb ...), token block tb;
    // register r <-> token 16 tb + 4 g4 + r, lane c <-> LDS row c of the block (feature 64 (fb/4) + 4c + fb%4).
    const int c16 = lane & 15, g4 = lane >> 4;
    const int sw16 = (c16 >> 1) & 7;
    const int w16_base = (wf * WAVE_F + c16) * SLAB, x16_base = BF * SLAB + (wt_ * WAVE_T + c16) * SLAB;
    half8_t gw[2][4], gx[2][4];           // two sets of four W fragments (half a k-step) and of four x fragments (a k-step)
    f32x4_t acc16[MS == 16 ? 8 : 1][MS == 16 ? 4 : 1];
    auto load_w16 = [&](half8_t(&w_)[4], const char *st, int step, int half) {
        const int off = ((4 * step + g4) ^ sw16) * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) w_[i] = *(const half8_t *)(st + w16_base + (4 * half + i) * (16 * SLAB) + off);
    };
    auto load_x16 = [&](half8_t(&x_)[4], const char *st, int step) {
        const int off = ((4 * step + g4) ^ sw16) * 16;
#pragma unroll
        for (int b = 0; b < 4; ++b) x_[b] = *(const half8_t *)(st + x16_base + b * (16 * SLAB) + off);
    };
    auto mma16 = [&](const half8_t(&w_)[4], const half8_t(&x_)[4], auto half_c) {
        constexpr int half = decltype(half_c)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                acc16[4 * half + i][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x_[b], w_[i], acc16[4 * half + i][b], 0, 0, 0);
    };

    int v = blockIdx.x;
    const char *cur_base, *nxt_base;
    unsigned cur_bytes, nxt_bytes;
    my_source(v, cur_base, cur_bytes);
    my_source(v + (int)gridDim.x, nxt_base, nxt_bytes);
    // ring prologue: items 0 and 1 of the first tile (nk >= 2 is the launcher's condition)
    issue(cur_base, cur_bytes, 0, 0);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    issue(cur_base, cur_bytes, 1, 1);
    int g = 0;  // ring items consumed so far (stage = g & 1)

    for (; v < n_tiles; v += (int)gridDim.x) {
        int f0, t0;
        tile_origin(v, f0, t0);
        if constexpr (MS == 32) {
#pragma unroll
            for (int a = 0; a < FT; ++a)
#pragma unroll
                for (int b = 0; b < TT; ++b)
#pragma unroll
                    for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.0f;
        } else {
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc16[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }

        if constexpr (MS == 32) load(fw[0], fx[0], smem + (g & 1) * STAGE, 0);
        else {
            load_x16(gx[0], smem + (g & 1) * STAGE, 0);
            load_w16(gw[0], smem + (g & 1) * STAGE, 0, 0);
        }
        for (int it = 0; it < nk; ++it, ++g) {
            const char *st = smem + (g & 1) * STAGE;
            if constexpr (MS == 32) {
                load(fw[1], fx[1], st, 1);
                mma(fw[0], fx[0]);
                interleave();
                load(fw[0], fx[0], st, 2);
                mma(fw[1], fx[1]);
                interleave();
                load(fw[1], fx[1], st, 3);
                mma(fw[0], fx[0]);
                interleave();
            } else {
                // quarter 0: k-step 0, features 0-63; fetch the W fragments of features 64-127
                load_w16(gw[1], st, 0, 1);
                mma16(gw[0], gx[0], std::integral_constant<int, 0>{});
                interleave_reads_mfmas<4, 16>();
                // quarter 1: k-step 0, features 64-127; fetch k-step 1 (x and the first W half)
                load_x16(gx[1], st, 1);
                load_w16(gw[0], st, 1, 0);
                mma16(gw[1], gx[0], std::integral_constant<int, 1>{});
                interleave_reads_mfmas<8, 16>();
                // quarter 2: k-step 1, features 0-63
                load_w16(gw[1], st, 1, 1);
                mma16(gw[0], gx[1], std::integral_constant<int, 0>{});
                interleave_reads_mfmas<4, 16>();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my reads of this stage are complete
            // item g+1 has landed.  vmcnt retires in issue order: in the first iteration after an epilogue the
            // stores of the previous tile are YOUNGER than item g+1 and may stay in flight
            if (it == 0 && g != 0) wait_vmcnt<STORES_WAIT>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            {   // item g+2 goes into the stage every wave has just finished reading
                const int k2 = it + 2;
                if (k2 < nk) issue(cur_base, cur_bytes, g & 1, k2);
                else if (nxt_bytes != 0) issue(nxt_base, nxt_bytes, g & 1, k2 - nk);
            }
            if constexpr (MS == 32) {
                if (it + 1 < nk) load(fw[0], fx[0], smem + ((g + 1) & 1) * STAGE, 0);
                mma(fw[1], fx[1]);
                interleave();
            } else {
                // quarter 3: k-step 1, features 64-127; fetch k-step 0 of the next slab
                if (it + 1 < nk) {
                    load_x16(gx[0], smem + ((g + 1) & 1) * STAGE, 0);
                    load_w16(gw[0], smem + ((g + 1) & 1) * STAGE, 0, 0);
                }
                mma16(gw[1], gx[1], std::integral_constant<int, 1>{});
                interleave_reads_mfmas<8, 16>();
            }
        }
        cur_base = nxt_base;
        cur_bytes = nxt_bytes;
        my_source(v + 2 * (int)gridDim.x, nxt_base, nxt_bytes);

        // ---- epilogue: accumulators -> memory, no LDS ------------------------------------------
        // address = per-lane part (feature pair, +4 rows for the upper half-wave) in the VGPR offset, the
        // register's row in the scalar offset.  Only the VGPR offset is range-checked by the buffer unit, so a
        // ragged last token tile takes the form that adds the row into the VGPR offset.  Stores are non-temporal:
        // default-policy stores of 32 MB per round push the W and x tiles out of the L2s (-6 % on the K = 768 shapes).
        if (p.dbg & DBG_LINEAR_SKIP_EPILOGUE) continue;   // timing only
        // (lane-derived values are recomputed here behind an opaque copy: hoisted out of the tile loop they would
        // sit in VGPRs through the main loop, which has none to spare at 16 waves)
        unsigned lane_e = (unsigned)lane;
        asm volatile("" : "+v"(lane_e));
        const int r32e = (int)(lane_e & 31u), he = (int)(lane_e >> 5);
        const int rows = p.M - t0 < BT ? p.M - t0 : BT;
        const unsigned row_b = (unsigned)p.N * 2u;
        const unsigned valid = (unsigned)rows * row_b;
        const unsigned wrow_b = (unsigned)(wt_ * WAVE_T) * row_b;
        const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(p.out + (size_t)t0 * p.N, (p.dbg & DBG_LINEAR_DROP_STORES) ? 0u : valid);
        const __amdgpu_buffer_rsrc_t rs_res = make_rsrc(p.resid != nullptr ? p.resid + (size_t)t0 * p.N : p.out, p.resid != nullptr ? valid : 0u);
        auto finish = [&](auto act_tag, auto res_tag, auto ragged_tag) {
            constexpr int ACT = decltype(act_tag)::value;
            constexpr bool RES = decltype(res_tag)::value;
            constexpr bool RAGGED = decltype(ragged_tag)::value;
            if constexpr (MS == 16) {
                // lane (c, g4) holds, in the four feature blocks of a 64-feature group, the ADJACENT features 4c .. 4c+3
                // (the W rows were interleaved that way at DMA time) of tokens 16 tb + 4 g4 + r: an 8-byte store per lane,
                // sixteen lanes = one whole 128-byte line, four token rows per instruction -- half the store instructions
                // of the 32x32x16 form (with two features per lane and 64-byte segments this epilogue cost 8 us more)
                typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                const int c16e = (int)(lane_e & 15u), g4e = (int)(lane_e >> 4);
#pragma unroll
                for (int grp = 0; grp < 2; ++grp) {
                    const int fcol = f0 + wf * WAVE_F + grp * 64 + 4 * c16e;
                    const bool f_ok = fcol < p.N;   // N % 4 == 0 (the launcher's condition)
                    const unsigned voff = f_ok ? (unsigned)(4 * g4e) * row_b + (unsigned)fcol * 2u : 0x80000000u;
                    f32x4_t bias4 = {0.f, 0.f, 0.f, 0.f};
                    if (p.bias != nullptr && f_ok) bias4 = *(const f32x4_t *)(p.bias + fcol);
                    unsigned rv0[4][4], rv1[4][4];   // the residual's two feature pairs (as scalars: see below)
                    if constexpr (RES) {
#pragma unroll
                        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const unsigned ro = wrow_b + (unsigned)(tb * 16 + r) * row_b;
                                // (element by element into scalars: through a 2-vector -- __builtin_bit_cast of the builtin's
                                // result, or element stores into one -- hipcc loaded ONE dword and used it for both halves)
                                const auto raw = RAGGED ? __builtin_amdgcn_raw_buffer_load_b64(rs_res, voff + ro, 0, 0)
                                                        : __builtin_amdgcn_raw_buffer_load_b64(rs_res, voff, ro, 0);
                                rv0[tb][r] = raw[0];
                                rv1[tb][r] = raw[1];
                            }
                    }
#pragma unroll
                    for (int tb = 0; tb < 4; ++tb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const unsigned ro = wrow_b + (unsigned)(tb * 16 + r) * row_b;
                            half2_t lo = f2h(act_apply<ACT>(acc16[4 * grp][tb][r] + bias4[0]),
                                             act_apply<ACT>(acc16[4 * grp + 1][tb][r] + bias4[1]));
                            half2_t hi = f2h(act_apply<ACT>(acc16[4 * grp + 2][tb][r] + bias4[2]),
                                             act_apply<ACT>(acc16[4 * grp + 3][tb][r] + bias4[3]));
                            if constexpr (RES) {
                                lo = lo + __builtin_bit_cast(half2_t, rv0[tb][r]);
                                hi = hi + __builtin_bit_cast(half2_t, rv1[tb][r]);
                            }
                            const u32x2_t o = {__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
                            if constexpr (RAGGED) __builtin_amdgcn_raw_buffer_store_b64(o, rs_out, voff + ro, 0, 2);
                            else __builtin_amdgcn_raw_buffer_store_b64(o, rs_out, voff, ro, 2);
                        }
                }
                return;
            }
#pragma unroll
            for (int blk = 0; blk < FT / 2; ++blk) {
                const int fcol = f0 + wf * WAVE_F + blk * 64 + 2 * r32e;   // this lane's feature pair
                const bool f_ok = fcol < p.N;                               // N is even
                // (an out-of-range pair gets an offset that stays out of range with any row added)
                const unsigned voff = f_ok ? (unsigned)(4 * he) * row_b + (unsigned)fcol * 2u : 0x80000000u;
                float2 bias2 = make_float2(0.f, 0.f);
                if (p.bias != nullptr && f_ok) bias2 = *(const float2 *)(p.bias + fcol);
                unsigned rv[TT][16];
                if constexpr (RES) {
#pragma unroll
                    for (int b = 0; b < TT; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const unsigned ro = wrow_b + (unsigned)(b * 32 + 8 * (r >> 2) + (r & 3)) * row_b;
                            rv[b][r] = RAGGED ? __builtin_amdgcn_raw_buffer_load_b32(rs_res, voff + ro, 0, 0)
                                              : __builtin_amdgcn_raw_buffer_load_b32(rs_res, voff, ro, 0);
                        }
                }
#pragma unroll
                for (int b = 0; b < TT; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned ro = wrow_b + (unsigned)(b * 32 + 8 * (r >> 2) + (r & 3)) * row_b;
                        // (the activated value is rounded to fp16 before the residual add, as in linear_kernel)
                        half2_t o;
                        if constexpr (MS == 32)
                            o = f2h(act_apply<ACT>(acc[2 * blk][b][r] + bias2.x), act_apply<ACT>(acc[2 * blk + 1][b][r] + bias2.y));
                        if constexpr (RES) o = o + __builtin_bit_cast(half2_t, rv[b][r]);
                        if constexpr (RAGGED)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rs_out, voff + ro, 0, 2);
                        else
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rs_out, voff, ro, 2);
                    }
            }
        };
        auto with_res = [&](auto act_tag) {
            if (rows < BT) {
                if (p.resid != nullptr) finish(act_tag, std::true_type{}, std::true_type{});
                else finish(act_tag, std::false_type{}, std::true_type{});
            } else {
                if (p.resid != nullptr) finish(act_tag, std::true_type{}, std::false_type{});
                else finish(act_tag, std::false_type{}, std::false_type{});
            }
        };
        if (p.act == MMRAG_ACT_GELU) with_res(std::integral_constant<int, MMRAG_ACT_GELU>{});
        else if (p.act == MMRAG_ACT_QUICK_GELU) with_res(std::integral_constant<int, MMRAG_ACT_QUICK_GELU>{});
        else with_res(std::integral_constant<int, MMRAG_ACT_NONE>{});
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// attention (packed sequences): ctx[t, head*DH + d] = softmax(Q K^T * scale + mask) V
// qkv is [T, 3H] fp16 (Q | K | V column blocks).  One workgroup = (128-query tile, head, seq),
// 4 waves x 32 queries; keys are consumed in tiles of 64.
// ---------------------------------------------------------------------------------------------
// RESIDENT (sequences of at most 256 keys): all of K and V^T of the (sequence, head) is staged once, ONE barrier, and the
// waves then walk the key tiles on their own -- no per-tile staging, no per-tile barrier.
template <int DH, int OCC = 1, bool RESIDENT = false, int NWAVES = 4, int MAXT = 4>   // NWAVES x 32 queries per workgroup; RESIDENT: <= MAXT key tiles
__global__ __launch_bounds__(64 * NWAVES, OCC) void attention_kernel(const _Float16 *__restrict__ qkv,
                                                            const int *__restrict__ cu_seqlens,
                                                            _Float16 *__restrict__ ctx, int H, float scale,
                                                            int causal) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(DH == 32 || DH == 64, "head dim");
    constexpr int KT = 64;                 // keys per tile
    constexpr int CH = DH / 8;             // 16-byte chunks per K row
    constexpr int KROW = DH * 2;           // K tile row bytes
    constexpr int VROW = KT * 2 + 8;       // V^T tile row bytes (padded: conflict-free b64 reads)
    constexpr int K_BYTES = KT * KROW;
    constexpr int V_BYTES = DH * VROW;
    constexpr int NQK = DH / 16;           // MFMA k-steps for Q.K^T
    constexpr int NDB = DH / 32;           // 32-row blocks of O^T
    constexpr int NBUF = RESIDENT ? MAXT : 2;  // key tiles held in LDS
    static_assert(NBUF * (K_BYTES + V_BYTES) <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) char smem[NBUF * (K_BYTES + V_BYTES)];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r32 = lane & 31;
    const int hh = lane >> 5;
    const int head = blockIdx.y;
    const int seq0 = cu_seqlens[blockIdx.z];
    const int len = cu_seqlens[blockIdx.z + 1] - seq0;
    constexpr int QT = 32 * NWAVES, NT = 64 * NWAVES;
    const int q_tile0 = blockIdx.x * QT;
    if (q_tile0 >= len) return;
    const int ld = 3 * H;
    const _Float16 *Qp = qkv + (size_t)seq0 * ld + head * DH;
    const _Float16 *Kp = Qp + H;
    const _Float16 *Vp = Qp + 2 * H;

    // this lane's query row (both half-waves hold the same 32 queries)
    const int qi = q_tile0 + wave * 32 + r32;
    half8_t qf[NQK];
#pragma unroll
    for (int ks = 0; ks < NQK; ++ks) {
        if (qi < len)
            qf[ks] = *(const half8_t *)(Qp + (size_t)qi * ld + ks * 16 + hh * 8);
        else
#pragma unroll
            for (int i = 0; i < 8; ++i) qf[ks][i] = (_Float16)0.f;
    }

    const int kv_end = causal ? ((q_tile0 + QT < len) ? q_tile0 + QT : len) : len;
    const int n_tiles = (kv_end + KT - 1) / KT;

    // staging assignment.  K: 16-byte chunk (key, c) per thread-iteration, coalesced.
    // V: lane = key, so the transposed 2-byte LDS writes of a wave are contiguous.
    constexpr int K_ITERS = (KT * CH) / NT > 0 ? (KT * CH) / NT : 1;
    constexpr int V_ITERS = CH / NWAVES > 0 ? CH / NWAVES : 1;  // d-chunks per wave
    uint4 kreg0[K_ITERS], vreg0[V_ITERS];
    auto load_tile_into = [&](int kt, uint4(&kreg)[K_ITERS], uint4(&vreg)[V_ITERS]) {
        const int k0 = kt * KT;
#pragma unroll
        for (int i = 0; i < K_ITERS; ++i) {
            const int idx = threadIdx.x + i * NT;
            const int key = idx / CH, c = idx % CH;
            kreg[i] = make_uint4(0, 0, 0, 0);
            if (idx < KT * CH && k0 + key < len) kreg[i] = *(const uint4 *)(Kp + (size_t)(k0 + key) * ld + c * 8);
        }
#pragma unroll
        for (int i = 0; i < V_ITERS; ++i) {
            const int c = wave + i * NWAVES;
            vreg[i] = make_uint4(0, 0, 0, 0);
            if (c < CH && k0 + lane < len) vreg[i] = *(const uint4 *)(Vp + (size_t)(k0 + lane) * ld + c * 8);
        }
    };
    auto store_tile_from = [&](int buf, const uint4(&kreg)[K_ITERS], const uint4(&vreg)[V_ITERS]) {
        char *kb = smem + buf * (K_BYTES + V_BYTES);
        char *vb = kb + K_BYTES;
#pragma unroll
        for (int i = 0; i < K_ITERS; ++i) {
            const int idx = threadIdx.x + i * NT;
            const int key = idx / CH, c = idx % CH;
            if (idx < KT * CH) {
                const int cs = DH == 64 ? (c ^ ((key >> 1) & 7)) : (c ^ ((key >> 2) & 3));
                *(uint4 *)(kb + key * KROW + cs * 16) = kreg[i];
            }
        }
#pragma unroll
        for (int i = 0; i < V_ITERS; ++i) {
            const int c = wave + i * NWAVES;
            if (c < CH) {
                const unsigned w[4] = {vreg[i].x, vreg[i].y, vreg[i].z, vreg[i].w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned short bits = (unsigned short)(w[e >> 1] >> ((e & 1) * 16));
                    *(unsigned short *)(vb + (c * 8 + e) * VROW + lane * 2) = bits;
                }
            }
        }
    };

    f32x16_t o[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int j = 0; j < 16; ++j) o[d][j] = 0.f;
    float m_run = NEG_INF_F, l_run = 0.f;

    auto load_tile = [&](int kt) { load_tile_into(kt, kreg0, vreg0); };
    auto store_tile = [&](int buf) { store_tile_from(buf, kreg0, vreg0); };
    if constexpr (RESIDENT) {
        // every tile's loads in flight together (one memory latency), then the LDS writes, then the only barrier
        uint4 kr[MAXT - 1][K_ITERS], vr[MAXT - 1][V_ITERS];
        load_tile(0);
#pragma unroll
        for (int t = 1; t < MAXT; ++t)
            if (t < n_tiles) load_tile_into(t, kr[t - 1], vr[t - 1]);
        store_tile(0);
#pragma unroll
        for (int t = 1; t < MAXT; ++t)
            if (t < n_tiles) store_tile_from(t, kr[t - 1], vr[t - 1]);
    } else {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    for (int kt = 0; kt < n_tiles; ++kt) {
        const int buf = RESIDENT ? kt : (kt & 1);
        if constexpr (!RESIDENT)
            if (kt + 1 < n_tiles) load_tile(kt + 1);
        const char *kb = smem + buf * (K_BYTES + V_BYTES);
        const char *vb = kb + K_BYTES;

        // S^T[key][query] for the two 32-key blocks
        f32x16_t s[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int j = 0; j < 16; ++j) s[b][j] = 0.f;
            const int key = b * 32 + r32;
#pragma unroll
            for (int ks = 0; ks < NQK; ++ks) {
                const int c = 2 * ks + hh;
                const int cs = DH == 64 ? (c ^ ((key >> 1) & 7)) : (c ^ ((key >> 2) & 3));
                const half8_t kf = *(const half8_t *)(kb + key * KROW + cs * 16);
                s[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[b], 0, 0, 0);
            }
        }
        // softmax in the log2 domain on RAW scores: p = exp2(s*c - m*c), c = scale*log2(e) > 0, so the
        // running max is taken on raw scores; masking only on tiles that touch the sequence end or the
        // causal diagonal (wave-uniform test)
        const int k0 = kt * KT;
        const bool edge = (k0 + KT > len) || (causal && k0 + KT - 1 > q_tile0 + wave * 32);
        if (edge) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int key = k0 + b * 32 + (j & 3) + 8 * (j >> 2) + 4 * hh;
                    const bool ok = key < len && (!causal || key <= qi);
                    s[b][j] = ok ? s[b][j] : NEG_INF_F;
                }
        }
        float m_tile = NEG_INF_F;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 16; j += 2) m_tile = fmaxf(m_tile, fmaxf(s[b][j], s[b][j + 1]));
        m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32));
        const float m_new = fmaxf(m_run, m_tile);
        const float m_use = m_new == NEG_INF_F ? 0.f : m_new;  // fully masked row (padding query)
        const float c2 = scale * 1.4426950408889634f;
        const float mc = m_use * c2;
        const float alpha = __builtin_amdgcn_exp2f(m_run * c2 - mc);
        float psum = 0.f;
        half8_t pf[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(s[b][j], c2, -mc));
                psum += pv;
                pf[b][j >> 3][j & 7] = (_Float16)pv;
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int j = 0; j < 16; ++j) o[d][j] *= alpha;
        // O^T[d][query] += V^T . P^T   (P straight from the accumulator registers; element j of
        // k-step st of key block b is key b*32 + 16*st + 8*(j>>2) + 4*hh + (j&3))
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
            const char *vrow = vb + (d * 32 + r32) * VROW;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const int kk = b * 32 + st * 16 + 4 * hh;
                    const half4_t lo = *(const half4_t *)(vrow + kk * 2);
                    const half4_t hi = *(const half4_t *)(vrow + (kk + 8) * 2);
                    half8_t vf;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        vf[i] = lo[i];
                        vf[4 + i] = hi[i];
                    }
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[b][st], o[d], 0, 0, 0);
                }
        }
        if constexpr (!RESIDENT) {
            if (kt + 1 < n_tiles) store_tile(buf ^ 1);
            __syncthreads();
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (qi < len) {
        _Float16 *dst = ctx + (size_t)(seq0 + qi) * H + head * DH;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                half4_t v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = (_Float16)(o[d][4 * g + i] * inv);
                *(half4_t *)(dst + d * 32 + 8 * g + 4 * hh) = v;
            }
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// pooling + (optional LayerNorm + projection handled by caller) + L2 normalisation
// pool: 0 = mean over the sequence's tokens, 1 = first token (CLS), 2 = token index sel[b]
// out fp32 [B, H]; x / max(||x||, 1e-12)
// ---------------------------------------------------------------------------------------------
template <typename OutT>
__global__ __launch_bounds__(256) void pool_norm_kernel(const _Float16 *__restrict__ x,
                                                         const int *__restrict__ cu_seqlens,
                                                         const int *__restrict__ sel, OutT *__restrict__ out,
                                                         int H, int pool, int normalize) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const int s0 = cu_seqlens[b], s1 = cu_seqlens[b + 1];
    float vals[4];  // H <= 1024
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = threadIdx.x + i * 256;
        float v = 0.f;
        if (c < H) {
            if (pool == 0) {
                for (int t = s0; t < s1; ++t) v += (float)x[(size_t)t * H + c];
                const float cnt = (float)(s1 - s0);
                v = v / fmaxf(cnt, 1e-9f);
            } else {
                const int t = pool == 1 ? s0 : s0 + sel[b];
                v = (float)x[(size_t)t * H + c];
            }
        }
        vals[i] = v;
        sq += v * v;
    }
    sq = wave_sum(sq);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
    __syncthreads();
    const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]);
    const float inv = normalize ? 1.0f / fmaxf(nrm, 1e-12f) : 1.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = threadIdx.x + i * 256;
        if (c < H) out[(size_t)b * H + c] = (OutT)(vals[i] * inv);
    }
}

// ---------------------------------------------------------------------------------------------
// ViT front end.  patchify: image -> rows of the patch-embedding GEMM, vector order (c, ph, pw) =
// the conv kernel's order.  Input either fp16 CHW already normalised, or raw uint8 HWC crops, in
// which case CLIP's (x/255 - mean)/std normalisation is fused here (image preprocessing on the GPU).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void patchify_kernel(const void *__restrict__ pixels, int kind,
                                                        _Float16 *__restrict__ patches, int B, int IMG, int P) {
    const int G = IMG / P, NP = G * G, PK = 3 * P * P, CH = PK / 8;
    const long long total = (long long)B * NP * CH;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ch = (int)(i % CH);
    const long long prow = i / CH;
    const int pidx = (int)(prow % NP), b = (int)(prow / NP);
    const int e0 = ch * 8;  // element offset inside the patch vector: 8 consecutive pw of one (c, ph)
    const int c = e0 / (P * P), ph = (e0 / P) % P, pw = e0 % P;
    const int y = (pidx / G) * P + ph, x0 = (pidx % G) * P + pw;
    half8_t o;
    if (kind == MMRAG_PIXELS_F16_CHW) {
        o = *(const half8_t *)((const _Float16 *)pixels + (((size_t)b * 3 + c) * IMG + y) * IMG + x0);
    } else {
        const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f};
        const float stdv[3] = {0.26862954f, 0.26130258f, 0.27577711f};
        const unsigned char *src = (const unsigned char *)pixels + (((size_t)b * IMG + y) * IMG + x0) * 3 + c;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)(((float)src[e * 3] / 255.0f - mean[c]) / stdv[c]);
    }
    *(half8_t *)(patches + (size_t)prow * PK + e0) = o;
}

// x[b*S + t] = LN_pre((t == 0 ? class_embedding : patch_emb[b*(S-1) + t-1]) + pos[t])
__global__ __launch_bounds__(256) void vit_assemble_ln_kernel(const _Float16 *__restrict__ emb,
                                                               const _Float16 *__restrict__ cls,
                                                               const _Float16 *__restrict__ pos,
                                                               const float *__restrict__ g, const float *__restrict__ b,
                                                               _Float16 *__restrict__ out, int T, int H, int S,
                                                               float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const int t = row % S, img = row / S;
    const _Float16 *src = t == 0 ? cls : emb + ((size_t)img * (S - 1) + (t - 1)) * H;
    ln_row<2>(src, out + (size_t)row * H, g, b, H, eps, lane, pos + (size_t)t * H);
}

// out[b, :] = x[b, :] / max(||x[b, :]||, 1e-12)  (fp16 in, fp32 out), one wave per row
__global__ __launch_bounds__(256) void normalize_rows_kernel(const _Float16 *__restrict__ x, float *__restrict__ out,
                                                              int B, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    float sq = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float v = (float)x[(size_t)row * D + c];
        sq += v * v;
    }
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(sq)), 1e-12f);
    for (int c = lane; c < D; c += 64) out[(size_t)row * D + c] = (float)x[(size_t)row * D + c] * inv;
}

// ---------------------------------------------------------------------------------------------
// Few tokens (M <= 64): the online /query path embeds ONE short text (reference embedder.py:539-583),
// where the tiled kernel above runs a handful of workgroups through a long serial K loop (20 us for
// MiniLM's FFN2).  Here one workgroup owns 32 output features; its four waves split K four ways, read
// the MFMA fragments straight from global memory (the operands are a few hundred KB, L2-resident),
// and the partial 32x32 tiles are summed through LDS.  Same arithmetic order inside a k-step as the tiled
// kernel, but a different split of K, so results differ from it in the last fp32 bit -- parity is against
// the oracle, as for every kernel.
// ---------------------------------------------------------------------------------------------
template <int NWC>  // waves that split K (4, 8 or 16); the first four also reduce and store
__global__ __launch_bounds__(64 * NWC) void linear_small_kernel(const LinearParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ float part[NWC][16][64];  // [wave][register][lane]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r32 = lane & 31, h = lane >> 5;
    const int f0 = blockIdx.x * 32;
    const int kw = p.K / NWC;               // this wave's share of K (a multiple of 16, checked by the launcher)
    const int k_lo = wave * kw;
    const int fr = f0 + r32 < p.N ? f0 + r32 : p.N - 1;   // clamped: out-of-range rows are computed and dropped
    const _Float16 *wrow = (const _Float16 *)p.wt + (size_t)fr * p.K + k_lo + 8 * h;
    for (int t0 = 0; t0 < p.M; t0 += 32) {
        const int tr = t0 + r32 < p.M ? t0 + r32 : p.M - 1;
        const _Float16 *xrow = (const _Float16 *)p.x + (size_t)tr * p.K + k_lo + 8 * h;
        f32x16_t acc;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
        for (int k = 0; k < kw; k += 64) {   // 4 k-steps per trip: 8 independent 16-byte loads in flight
            half8_t a[4], b[4];
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                const bool in = k + 16 * s_ < kw;
                a[s_] = *(const half8_t *)(wrow + (in ? k + 16 * s_ : 0));
                b[s_] = *(const half8_t *)(xrow + (in ? k + 16 * s_ : 0));
            }
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
                if (k + 16 * s_ < kw) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s_], b[s_], acc, 0, 0, 0);
        }
        if (t0 > 0) __syncthreads();  // the previous token block's partials have been consumed
#pragma unroll
        for (int j = 0; j < 16; ++j) part[wave][j][lane] = acc[j];
        __syncthreads();
        // wave g < 4 reduces register group g: features f0 + 8 g + 4 h + (0..3) of token t0 + r32
        const int g = wave & 3;
        const int t = t0 + r32;
        const int f = f0 + 8 * g + 4 * h;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (wave < 4) {
#pragma unroll
            for (int w = 0; w < NWC; ++w)
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] += part[w][4 * g + i][lane];
        }
        if (wave < 4 && t < p.M && f < p.N) {   // N % 4 == 0: a group of four features is in range together
            if (p.bias != nullptr) {
                const float4 bv = *(const float4 *)(p.bias + f);
                v[0] += bv.x, v[1] += bv.y, v[2] += bv.z, v[3] += bv.w;
            }
            half4_t o;
            if (p.act == MMRAG_ACT_GELU) {
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (_Float16)act_apply<MMRAG_ACT_GELU>(v[i]);
            } else if (p.act == MMRAG_ACT_QUICK_GELU) {
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (_Float16)act_apply<MMRAG_ACT_QUICK_GELU>(v[i]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (_Float16)v[i];
            }
            if (p.resid != nullptr) {
                const half4_t rv = *(const half4_t *)(p.resid + (size_t)t * p.N + f);
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (_Float16)((float)o[i] + (float)rv[i]);
            }
            *(half4_t *)(p.out + (size_t)t * p.N + f) = o;
        }
    }
#endif
}

// M <= 64, second form: 16 features per workgroup (v_mfma_f32_16x16x32_f16), i.e. twice the workgroups of
// linear_small_kernel.  One query is 3-30 token rows against N = 384-3072 features: N / 32 workgroups leave most of the
// 256 CUs idle and each busy CU streams its weight rows alone (4.9 us at K = 384, 9.2 us at K = 1536, rocprof).
// NWC waves split K in shares of 32 * STEPS; every wave has all its loads in flight at once; wave 0 reduces.
template <int NWC, int STEPS, bool LN_IN = false>
__global__ __launch_bounds__(64 * NWC) void linear_small16_kernel(const LinearParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ float part[NWC][4][64];  // [wave][register][lane]
    __shared__ float2 rowsum[NWC][16];  // LN_IN: (sum, sum of squares) of each wave's share of the 16 token rows
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int f0 = blockIdx.x * 16;
    const int k_lo = wave * (32 * STEPS);
    const int fr = f0 + r16 < p.N ? f0 + r16 : p.N - 1;   // clamped: out-of-range rows are computed and dropped
    const _Float16 *wrow = (const _Float16 *)p.wt + (size_t)fr * p.K + k_lo + 8 * kq;
    half8_t a[STEPS];
#pragma unroll
    for (int s_ = 0; s_ < STEPS; ++s_) a[s_] = *(const half8_t *)(wrow + 32 * s_);
    float gam[LN_IN ? STEPS : 1][8], bet[LN_IN ? STEPS : 1][8];
    if constexpr (LN_IN) {
#pragma unroll
        for (int s_ = 0; s_ < STEPS; ++s_) {
            const int k = k_lo + 32 * s_ + 8 * kq;
            const f32x4_t g0 = *(const f32x4_t *)(p.ln_gamma + k), g1 = *(const f32x4_t *)(p.ln_gamma + k + 4);
            const f32x4_t b0 = *(const f32x4_t *)(p.ln_beta + k), b1 = *(const f32x4_t *)(p.ln_beta + k + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) gam[s_][e] = g0[e], gam[s_][4 + e] = g1[e], bet[s_][e] = b0[e], bet[s_][4 + e] = b1[e];
        }
    }
    for (int t0 = 0; t0 < p.M; t0 += 16) {
        const int tr = t0 + r16 < p.M ? t0 + r16 : p.M - 1;
        const _Float16 *xrow = (const _Float16 *)p.x + (size_t)tr * p.K + k_lo + 8 * kq;
        half8_t b[STEPS];
#pragma unroll
        for (int s_ = 0; s_ < STEPS; ++s_) b[s_] = *(const half8_t *)(xrow + 32 * s_);
        if constexpr (LN_IN) {
            // the workgroup's waves hold the 16 token rows between them (wave = K share, lane group kq = 8 of every 32):
            // row sums over the lane's values, over the four lane groups, over the waves (LDS), then normalise in place,
            // rounding to fp16 as the LayerNorm pass would have stored the row
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int s_ = 0; s_ < STEPS; ++s_)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = (float)b[s_][e];
                    s1 += v;
                    s2 = fmaf(v, v, s2);
                }
            s1 += __shfl_xor(s1, 16), s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32), s2 += __shfl_xor(s2, 32);
            if (t0 > 0) __syncthreads();   // rowsum of the previous token block has been read
            if (kq == 0) rowsum[wave][r16] = make_float2(s1, s2);
            __syncthreads();
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < NWC; ++w) {
                const float2 v = rowsum[w][r16];
                t1 += v.x, t2 += v.y;
            }
            const float inv_k = 1.0f / (float)p.K, mean = t1 * inv_k;
            const float rstd = rsqrtf(fmaxf(fmaf(t2, inv_k, -mean * mean), 0.f) + p.ln_eps);
            if (blockIdx.x == 0 && wave == 0 && kq == 0 && t0 + r16 < p.M && p.ln_stats_out != nullptr)
                p.ln_stats_out[t0 + r16] = make_float2(mean, rstd);
#pragma unroll
            for (int s_ = 0; s_ < STEPS; ++s_)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    b[s_][e] = (_Float16)fmaf(((float)b[s_][e] - mean) * rstd, gam[s_][e], bet[s_][e]);
        }
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s_ = 0; s_ < STEPS; ++s_) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s_], b[s_], acc, 0, 0, 0);
        if (t0 > 0) __syncthreads();  // the previous token block's partials have been consumed
#pragma unroll
        for (int j = 0; j < 4; ++j) part[wave][j][lane] = acc[j];
        __syncthreads();
        // wave 0: lane (token r16, feature group kq) owns features f0 + 4 kq + (0..3) of token t0 + r16
        const int t = t0 + r16;
        const int f = f0 + 4 * kq;
        if (wave == 0 && t < p.M && f < p.N) {   // N % 4 == 0: a group of four features is in range together
            float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < NWC; ++w)
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] += part[w][i][lane];
            if (p.bias != nullptr) {
                const float4 bv = *(const float4 *)(p.bias + f);
                v[0] += bv.x, v[1] += bv.y, v[2] += bv.z, v[3] += bv.w;
            }
            half4_t o;
            if (p.act == MMRAG_ACT_GELU) {
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (_Float16)act_apply<MMRAG_ACT_GELU>(v[i]);
            } else if (p.act == MMRAG_ACT_QUICK_GELU) {
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (_Float16)act_apply<MMRAG_ACT_QUICK_GELU>(v[i]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (_Float16)v[i];
            }
            if (p.resid != nullptr) {
                half4_t rv = *(const half4_t *)(p.resid + (size_t)t * p.N + f);
                if (p.res_stats != nullptr) {   // un-normalised residual rows: normalise, round as the pass would have
                    const float2 st = p.res_stats[t];
                    const float4 g = *(const float4 *)(p.res_gamma + f), be = *(const float4 *)(p.res_beta + f);
                    rv[0] = (_Float16)fmaf(((float)rv[0] - st.x) * st.y, g.x, be.x);
                    rv[1] = (_Float16)fmaf(((float)rv[1] - st.x) * st.y, g.y, be.y);
                    rv[2] = (_Float16)fmaf(((float)rv[2] - st.x) * st.y, g.z, be.z);
                    rv[3] = (_Float16)fmaf(((float)rv[3] - st.x) * st.y, g.w, be.w);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (_Float16)((float)o[i] + (float)rv[i]);
            }
            *(half4_t *)(p.out + (size_t)t * p.N + f) = o;
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
// what the single-query forward adds to a GEMM call (LinearParams explains the fields)
struct SmallNorms {
    const float *ln_gamma = nullptr, *ln_beta = nullptr;
    float ln_eps = 0.f;
    float2 *ln_stats_out = nullptr;
    const float2 *res_stats = nullptr;
    const float *res_gamma = nullptr, *res_beta = nullptr;
};

// does (M, K) take linear_small16_kernel, and can that kernel normalise its input rows (K = 32 or 64 per wave)?
static bool takes_small16_kernel(long long M, int K, bool ln_in) {
    if (M > 64 || (debug_flags() & (DBG_LINEAR_NO_SMALL | DBG_LINEAR_SMALL32))) return false;
    if (ln_in) return K == 384 || K == 512 || K == 768 || K == 1024;
    return K == 384 || K == 512 || K == 768 || K == 1024 || K == 1536 || K == 2048 || K == 3072;
}

int launch_linear(const void *x, int M, int K, const void *wt, int N, const float *bias, int act,
                  const void *resid, void *out, hipStream_t s, const SmallNorms *norms = nullptr) {
    LinearParams p;
    p.ln_gamma = p.ln_beta = nullptr, p.ln_eps = 0.f, p.ln_stats_out = nullptr;
    p.res_stats = nullptr, p.res_gamma = p.res_beta = nullptr;
    if (norms != nullptr) {
        if (!takes_small16_kernel(M, K, norms->ln_gamma != nullptr)) {
            set_error("linear: folded LayerNorm needs the 16-feature single-query kernel (M=%d K=%d)", M, K);
            return MMRAG_EUNSUPPORTED;
        }
        p.ln_gamma = norms->ln_gamma, p.ln_beta = norms->ln_beta, p.ln_eps = norms->ln_eps, p.ln_stats_out = norms->ln_stats_out;
        p.res_stats = norms->res_stats, p.res_gamma = norms->res_gamma, p.res_beta = norms->res_beta;
    }
    p.x = (const char *)x;
    p.wt = (const char *)wt;
    p.bias = bias;
    p.resid = (const _Float16 *)resid;
    p.out = (_Float16 *)out;
    p.M = M, p.N = N, p.K = K, p.act = act;
    const unsigned dbg = debug_flags();
    p.xcd_order = (dbg & DBG_LINEAR_PLAIN) ? 0 : 1;
    p.dbg = dbg;
    if (M <= 64 && !(dbg & (DBG_LINEAR_NO_SMALL | DBG_LINEAR_SMALL32))) {
        // the online single-query path, 16 features per workgroup: K in shares of 32 * STEPS over up to 16 waves
        const unsigned g16 = (unsigned)((N + 15) / 16);
        bool done = true;
        if (p.ln_gamma != nullptr) {
            if (K == 384) linear_small16_kernel<12, 1, true><<<g16, 768, 0, s>>>(p);
            else if (K == 512) linear_small16_kernel<16, 1, true><<<g16, 1024, 0, s>>>(p);
            else if (K == 768) linear_small16_kernel<12, 2, true><<<g16, 768, 0, s>>>(p);
            else linear_small16_kernel<16, 2, true><<<g16, 1024, 0, s>>>(p);   // K == 1024
        } else if (K == 384) linear_small16_kernel<12, 1><<<g16, 768, 0, s>>>(p);
        else if (K == 512) linear_small16_kernel<16, 1><<<g16, 1024, 0, s>>>(p);
        else if (K == 768) linear_small16_kernel<12, 2><<<g16, 768, 0, s>>>(p);
        else if (K == 1024) linear_small16_kernel<16, 2><<<g16, 1024, 0, s>>>(p);
        else if (K == 1536) linear_small16_kernel<16, 3><<<g16, 1024, 0, s>>>(p);
        else if (K == 2048) linear_small16_kernel<16, 4><<<g16, 1024, 0, s>>>(p);
        else if (K == 3072) linear_small16_kernel<16, 6><<<g16, 1024, 0, s>>>(p);
        else done = false;
        if (done) return MMRAG_OK;
    }
    if (M <= 64 && (K / 4) % 16 == 0 && !(dbg & DBG_LINEAR_NO_SMALL)) {
        // other K: split-K over the 16 / 8 / 4 waves of a 32-feature workgroup
        const unsigned g = (unsigned)((N + 31) / 32);
        if (K % 256 == 0) linear_small_kernel<16><<<g, 1024, 0, s>>>(p);
        else if (K % 128 == 0) linear_small_kernel<8><<<g, 512, 0, s>>>(p);
        else linear_small_kernel<4><<<g, 256, 0, s>>>(p);
        return MMRAG_OK;
    }
    const int cus = num_cus();
    const long long big_tiles = (long long)((M + 255) / 256) * ((N + 255) / 256);
    // a ragged last feature tile (N % 256 != 0) wastes part of its MFMAs but still beats the small tile while
    // at most about a third is padding (MiniLM: N = 384 and 1152 -> +21 % on the whole forward)
    const long long tiles_f = (N + 255) / 256;
    const bool big_pays = (double)N >= 0.65 * (double)(tiles_f * 256);
    if (big_tiles >= cus && big_pays) {
        // 16 waves (4 per SIMD): +10-20 % over 8 waves on the encoder shapes (A/B in one process)
        // 16 waves (4 per SIMD): +10-20 % over 8 waves on the encoder shapes; software-pipelined fragment
        // reads + XCD-aware tile order: another +4-15 % (A/B in one process, tools/linear_vs_rocblas.py)
        if (dbg & DBG_LINEAR_PLAIN)
            linear_kernel<256, 256, 4, 4, 2><<<(unsigned)big_tiles, 1024, 0, s>>>(p);
        else if (K >= 128 && N % 2 == 0 && !(dbg & DBG_LINEAR_NO_PERSIST)) {
            const unsigned grid = (unsigned)(big_tiles < cus ? big_tiles : cus);
            // 16x16x32 MFMAs: 5.3-5.9 % over the 32x32x16 form on every encoder shape (tools/linear_vs_rocblas.py, one
            // process; main loop alone 6-7.6 %), as fast as hipBLASLt at K = 768 with the bias fused.  Its stores take four
            // features per lane, hence N % 4.
            if (!(dbg & DBG_LINEAR_MFMA32) && N % 4 == 0) linear_persistent_kernel<2, 4, 16><<<grid, 512, 0, s>>>(p, (int)big_tiles, (int)tiles_f);
            else linear_persistent_kernel<2, 4, 32><<<grid, 512, 0, s>>>(p, (int)big_tiles, (int)tiles_f);
        } else
            linear_kernel<256, 256, 4, 4, 2, true><<<(unsigned)big_tiles, 1024, 0, s>>>(p);
    } else {
        const long long tiles = (long long)((M + 127) / 128) * ((N + 127) / 128);
        if ((dbg & DBG_LINEAR_TILE64) || (!(dbg & DBG_LINEAR_TILE128) && tiles < cus / 2)) {
            // a few hundred token rows (a dispatcher batch of queries): 128x128 tiles leave most CUs idle, 64x64 tiles of one
            // wave each are four times as many workgroups
            const long long t64 = (long long)((M + 63) / 64) * ((N + 63) / 64);
            linear_kernel<64, 64, 1, 1, 2, true><<<(unsigned)t64, 64, 0, s>>>(p);
            return MMRAG_OK;
        }
        // 2 ring stages (64 KB) so two workgroups share a CU, pipelined fragment reads: +5 % on the ViT forward
        linear_kernel<128, 128, 2, 2, 2, true><<<(unsigned)tiles, 256, 0, s>>>(p);
    }
    return MMRAG_OK;
}

}  // namespace mmrag_impl
using namespace mmrag_impl;

static std::atomic<long long> g_last_forward_us{0};

extern "C" {

int mmrag_linear_f16(const void *x, int64_t M, int K, const void *wt, int N, const float *bias, int act,
                     const void *resid, void *out, void *stream) {
    MMRAG_CHECK_ARG(x && wt && out, "linear: null pointer");
    MMRAG_CHECK_ARG(M > 0 && M < INT_MAX && N > 0 && K > 0, "linear: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
    MMRAG_CHECK_ARG(K % 64 == 0, "linear: K=%d must be a multiple of 64 (128-byte fp16 slabs)", K);
    MMRAG_CHECK_ARG(N % 4 == 0, "linear: N=%d must be a multiple of 4", N);
    MMRAG_CHECK_ARG(act >= 0 && act <= 2, "linear: bad activation %d", act);
    MMRAG_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)wt % 16) == 0 && ((uintptr_t)out % 8) == 0 &&
                        ((uintptr_t)resid % 8) == 0 && ((uintptr_t)bias % 16) == 0,
                    "linear: misaligned pointer");
    launch_linear(x, (int)M, K, wt, N, bias, act, resid, out, (hipStream_t)stream);
    MMRAG_CHECK_HIP(hipGetLastError());
    return MMRAG_OK;
}

int mmrag_layernorm_f16(const void *x, void *out, const float *gamma, const float *beta, int64_t T, int H,
                        float eps, void *stream) {
    MMRAG_CHECK_ARG(x && out && gamma && beta, "layernorm: null pointer");
    MMRAG_CHECK_ARG(T > 0 && H > 0 && H % 8 == 0 && H <= 1024, "layernorm: bad shape T=%lld H=%d (H % 8 == 0, <= 1024)", (long long)T, H);
    MMRAG_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)gamma % 16) == 0 && ((uintptr_t)beta % 16) == 0, "layernorm: pointers must be 16-byte aligned");
    layernorm_kernel<<<(unsigned)((T + 3) / 4), 256, 0, (hipStream_t)stream>>>(
        (const _Float16 *)x, (_Float16 *)out, gamma, beta, (int)T, H, eps);
    MMRAG_CHECK_HIP(hipGetLastError());
    return MMRAG_OK;
}

int mmrag_embed_ln_f16(const int32_t *ids, const int32_t *pos_ids, const void *tok, const void *pos,
                       const void *type0, const float *gamma, const float *beta, void *out, int64_t T, int H,
                       int vocab, int max_pos, float eps, void *stream) {
    MMRAG_CHECK_ARG(ids && pos_ids && tok && pos && out, "embed_ln: null pointer");
    MMRAG_CHECK_ARG((gamma == nullptr) == (beta == nullptr), "embed_ln: gamma/beta must both be given or both null");
    MMRAG_CHECK_ARG(T > 0 && H > 0 && H % 8 == 0 && H <= 1024, "embed_ln: bad shape T=%lld H=%d", (long long)T, H);
    embed_ln_kernel<<<(unsigned)((T + 3) / 4), 256, 0, (hipStream_t)stream>>>(
        ids, pos_ids, (const _Float16 *)tok, (const _Float16 *)pos, (const _Float16 *)type0, gamma, beta,
        (_Float16 *)out, (int)T, H, vocab, max_pos, eps);
    MMRAG_CHECK_HIP(hipGetLastError());
    return MMRAG_OK;
}

int mmrag_attention_f16(const void *qkv, const int32_t *cu_seqlens, void *ctx, int B, int max_len, int H,
                        int n_heads, int causal, void *stream) {
    MMRAG_CHECK_ARG(qkv && cu_seqlens && ctx, "attention: null pointer");
    MMRAG_CHECK_ARG(B > 0 && max_len > 0 && n_heads > 0 && H % n_heads == 0, "attention: bad shape");
    const int dh = H / n_heads;
    MMRAG_CHECK_ARG(dh == 32 || dh == 64, "attention: head dim %d unsupported (32 or 64)", dh);
    MMRAG_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)ctx % 8) == 0, "attention: misaligned pointer");
    const dim3 grid((unsigned)((max_len + 127) / 128), (unsigned)n_heads, (unsigned)B);
    const float scale = 1.0f / sqrtf((float)dh);
    // 3 waves per SIMD (<= 168 registers): -25 % vs the unconstrained 194-register build, which only fits 2
    // (A/B in one process, tools/attn_bench.py; 4 per SIMD spills and is slower)
    if (dh == 64 && max_len <= 256 && max_len > 128 && !(debug_flags() & DBG_ATTENTION_STREAMED)) {
        // at most four key tiles: K and V^T of the whole sequence stay in LDS (66 KB: two workgroups per CU), shared by
        // all 256 queries of the (sequence, head): 8 waves per workgroup, 4 per SIMD
        const dim3 grid8((unsigned)((max_len + 255) / 256), (unsigned)n_heads, (unsigned)B);
        attention_kernel<64, 2, true, 8><<<grid8, 512, 0, (hipStream_t)stream>>>((const _Float16 *)qkv, cu_seqlens,
                                                                                 (_Float16 *)ctx, H, scale, causal);
    } else if (dh == 64 && max_len > 256 && max_len <= 512 && !(debug_flags() & DBG_ATTENTION_STREAMED)) {
        // up to eight key tiles resident (132 KB): one workgroup of 16 waves (512 queries) per CU
        const dim3 grid16((unsigned)((max_len + 511) / 512), (unsigned)n_heads, (unsigned)B);
        attention_kernel<64, 1, true, 16, 8><<<grid16, 1024, 0, (hipStream_t)stream>>>((const _Float16 *)qkv, cu_seqlens,
                                                                                       (_Float16 *)ctx, H, scale, causal);
    } else if (dh == 64)
        attention_kernel<64, 3><<<grid, 256, 0, (hipStream_t)stream>>>((const _Float16 *)qkv, cu_seqlens,
                                                                       (_Float16 *)ctx, H, scale, causal);
    else if (max_len <= 256 && max_len > 128 && !(debug_flags() & DBG_ATTENTION_STREAMED)) {
        const dim3 grid8((unsigned)((max_len + 255) / 256), (unsigned)n_heads, (unsigned)B);
        attention_kernel<32, 2, true, 8><<<grid8, 512, 0, (hipStream_t)stream>>>((const _Float16 *)qkv, cu_seqlens,
                                                                                 (_Float16 *)ctx, H, scale, causal);
    } else
        attention_kernel<32, 3><<<grid, 256, 0, (hipStream_t)stream>>>((const _Float16 *)qkv, cu_seqlens,
                                                                       (_Float16 *)ctx, H, scale, causal);
    MMRAG_CHECK_HIP(hipGetLastError());
    return MMRAG_OK;
}

int mmrag_pool_normalize_f16(const void *x, const int32_t *cu_seqlens, const int32_t *sel, float *out, int B,
                             int H, int pool, int normalize, void *stream) {
    MMRAG_CHECK_ARG(x && cu_seqlens && out, "pool: null pointer");
    MMRAG_CHECK_ARG(B > 0 && H > 0 && H <= 1024, "pool: bad shape B=%d H=%d", B, H);
    MMRAG_CHECK_ARG(pool >= 0 && pool <= 2 && (pool != 2 || sel), "pool: bad mode %d", pool);
    pool_norm_kernel<float><<<(unsigned)B, 256, 0, (hipStream_t)stream>>>((const _Float16 *)x, cu_seqlens, sel, out, H,
                                                                         pool, normalize);
    MMRAG_CHECK_HIP(hipGetLastError());
    return MMRAG_OK;
}


// ---------------------------------------------------------------------------------------------
// composite forward: all launches of one encoder pass, stream-ordered, no host sync
// ---------------------------------------------------------------------------------------------
static size_t align256(size_t x) { return (x + 255) / 256 * 256; }

static size_t hm_elems(const mmrag_encoder_desc *d, int64_t T, int B) {
    size_t n = (size_t)T * (size_t)d->intermediate;
    if (d->patch > 0 && d->image > 0) {
        const size_t g = (size_t)(d->image / d->patch);
        const size_t patches = (size_t)B * g * g * 3 * (size_t)d->patch * (size_t)d->patch;
        if (patches > n) n = patches;
    }
    return n;
}

size_t mmrag_encoder_workspace_bytes(const mmrag_encoder_desc *d, int64_t T, int B) {
    if (!d || T <= 0 || B <= 0) return 0;
    const size_t H = (size_t)d->hidden, I = (size_t)d->intermediate;
    size_t bytes = 0;
    bytes += 2 * align256((size_t)T * H * 2);      // x, y
    bytes += align256((size_t)T * 3 * H * 2);      // qkv
    bytes += align256((size_t)T * H * 2);          // ctx
    bytes += align256(hm_elems(d, T, B) * 2);      // mlp hidden (also holds the ViT patch rows)
    bytes += 2 * align256((size_t)B * (H > (size_t)d->out_dim ? H : (size_t)d->out_dim) * 2);  // pooled, projected
    bytes += 2 * align256(64 * 8);                 // (mean, rstd) of up to 64 rows, twice (single-query path)
    return bytes + 256;
}

struct EncBuffers {
    void *x, *y, *qkv, *ctx, *hm, *pooled, *proj;
    float2 *stats_a, *stats_b;
};

static int carve(const mmrag_encoder_desc *d, int64_t T, int B, void *workspace, size_t workspace_bytes,
                 EncBuffers *b) {
    const size_t need = mmrag_encoder_workspace_bytes(d, T, B);
    if (!workspace || workspace_bytes < need) {
        set_error("encoder_forward: workspace %zu bytes < required %zu", workspace_bytes, need);
        return MMRAG_EWORKSPACE;
    }
    const size_t H = (size_t)d->hidden, Tz = (size_t)T;
    char *p = (char *)(((uintptr_t)workspace + 255) / 256 * 256);
    auto take = [&](size_t bytes) { char *r = p; p += align256(bytes); return (void *)r; };
    b->x = take(Tz * H * 2), b->y = take(Tz * H * 2), b->qkv = take(Tz * 3 * H * 2), b->ctx = take(Tz * H * 2);
    b->hm = take(hm_elems(d, T, B) * 2);
    const size_t pd = (size_t)(d->hidden > d->out_dim ? d->hidden : d->out_dim);
    b->pooled = take((size_t)B * pd * 2), b->proj = take((size_t)B * pd * 2);
    b->stats_a = (float2 *)take(64 * 8), b->stats_b = (float2 *)take(64 * 8);
    return MMRAG_OK;
}

#define RUN(call) do { if ((st = (call)) != MMRAG_OK) return st; } while (0)

// transformer blocks + head over the packed activations already in b.x
static int encoder_body(const mmrag_encoder_desc *d, const void *const *lw, const EncBuffers &b,
                        const int32_t *cu_seqlens, const int32_t *sel, int64_t T, int B, int max_len, float *out,
                        void *stream) {
    const int H = d->hidden, I = d->intermediate, L = d->n_layers;
    void *x = b.x, *y = b.y, *qkv = b.qkv, *ctx = b.ctx, *hm = b.hm;
    int st;
    const int causal = d->causal;
    hipStream_t s = (hipStream_t)stream;
    // The single-query path without LayerNorm launches (BERT, T <= 64): a query is a chain of ~45 dependent kernels of
    // 3-5 us, 12 of them LayerNorms over a few rows.  Here the GEMM that consumes a normalised tensor normalises the rows
    // itself as it loads them (its workgroup holds whole rows between its waves) and leaves (mean, rstd) behind for the
    // residual add two kernels later, which normalises the rows it adds.  x and y alternate as the UN-normalised outputs
    // of FFN2 / O-proj; only the last layer's output meets a real LayerNorm (the pooling reads it).
    const bool folded_ln = d->arch == MMRAG_ARCH_BERT && !(debug_flags() & DBG_ENCODER_LN_PASSES) &&
                           takes_small16_kernel(T, H, true) && takes_small16_kernel(T, I, false);
    if (folded_ln) {
        const float *g_prev = nullptr, *b_prev = nullptr;   // the previous layer's output norm (none before layer 0)
        for (int l = 0; l < L; ++l, lw += 12) {
            const float *bqkv = (const float *)lw[1], *bo = (const float *)lw[3];
            const float *g1 = (const float *)lw[4], *b1n = (const float *)lw[5];
            const float *bi = (const float *)lw[7], *b2 = (const float *)lw[9];
            SmallNorms nq, no, n1, n2;
            if (l > 0) {   // x holds the previous layer's un-normalised output
                nq.ln_gamma = g_prev, nq.ln_beta = b_prev, nq.ln_eps = d->ln_eps, nq.ln_stats_out = b.stats_b;
                no.res_stats = b.stats_b, no.res_gamma = g_prev, no.res_beta = b_prev;
            }
            RUN(launch_linear(x, (int)T, H, lw[0], 3 * H, bqkv, MMRAG_ACT_NONE, nullptr, qkv, s, &nq));
            RUN(mmrag_attention_f16(qkv, cu_seqlens, ctx, B, max_len, H, d->n_heads, causal, stream));
            RUN(launch_linear(ctx, (int)T, H, lw[2], H, bo, MMRAG_ACT_NONE, x, y, s, &no));
            n1.ln_gamma = g1, n1.ln_beta = b1n, n1.ln_eps = d->ln_eps, n1.ln_stats_out = b.stats_a;
            RUN(launch_linear(y, (int)T, H, lw[6], I, bi, d->act, nullptr, hm, s, &n1));
            n2.res_stats = b.stats_a, n2.res_gamma = g1, n2.res_beta = b1n;
            RUN(launch_linear(hm, (int)T, I, lw[8], H, b2, MMRAG_ACT_NONE, y, x, s, &n2));
            g_prev = (const float *)lw[10], b_prev = (const float *)lw[11];
        }
        MMRAG_CHECK_HIP(hipGetLastError());
        RUN(mmrag_layernorm_f16(x, y, g_prev, b_prev, T, H, d->ln_eps, stream));
        return mmrag_pool_normalize_f16(y, cu_seqlens, sel, out, B, H, d->pool, d->normalize, stream);
    }
    for (int l = 0; l < L; ++l, lw += 12) {
        const float *bqkv = (const float *)lw[1], *bo = (const float *)lw[3];
        const float *g1 = (const float *)lw[4], *b1n = (const float *)lw[5];
        const float *bi = (const float *)lw[7], *b2 = (const float *)lw[9];
        const float *g2 = (const float *)lw[10], *b2n = (const float *)lw[11];
        if (d->arch == MMRAG_ARCH_BERT) {
            RUN(mmrag_linear_f16(x, T, H, lw[0], 3 * H, bqkv, MMRAG_ACT_NONE, nullptr, qkv, stream));
            RUN(mmrag_attention_f16(qkv, cu_seqlens, ctx, B, max_len, H, d->n_heads, causal, stream));
            RUN(mmrag_linear_f16(ctx, T, H, lw[2], H, bo, MMRAG_ACT_NONE, x, y, stream));
            RUN(mmrag_layernorm_f16(y, x, g1, b1n, T, H, d->ln_eps, stream));
            RUN(mmrag_linear_f16(x, T, H, lw[6], I, bi, d->act, nullptr, hm, stream));
            RUN(mmrag_linear_f16(hm, T, I, lw[8], H, b2, MMRAG_ACT_NONE, x, y, stream));
            RUN(mmrag_layernorm_f16(y, x, g2, b2n, T, H, d->ln_eps, stream));
        } else {
            RUN(mmrag_layernorm_f16(x, y, g1, b1n, T, H, d->ln_eps, stream));
            RUN(mmrag_linear_f16(y, T, H, lw[0], 3 * H, bqkv, MMRAG_ACT_NONE, nullptr, qkv, stream));
            RUN(mmrag_attention_f16(qkv, cu_seqlens, ctx, B, max_len, H, d->n_heads, causal, stream));
            RUN(mmrag_linear_f16(ctx, T, H, lw[2], H, bo, MMRAG_ACT_NONE, x, x, stream));
            RUN(mmrag_layernorm_f16(x, y, g2, b2n, T, H, d->ln_eps, stream));
            RUN(mmrag_linear_f16(y, T, H, lw[6], I, bi, d->act, nullptr, hm, stream));
            RUN(mmrag_linear_f16(hm, T, I, lw[8], H, b2, MMRAG_ACT_NONE, x, x, stream));
        }
    }
    if (d->arch == MMRAG_ARCH_BERT) {
        RUN(mmrag_pool_normalize_f16(x, cu_seqlens, sel, out, B, H, d->pool, d->normalize, stream));
    } else {
        // final LayerNorm (tail[0..1]), pooled token, bias-free projection (tail[2]), L2 normalise
        RUN(mmrag_layernorm_f16(x, y, (const float *)lw[0], (const float *)lw[1], T, H, d->ln_eps, stream));
        pool_norm_kernel<_Float16><<<(unsigned)B, 256, 0, s>>>((const _Float16 *)y, cu_seqlens, sel,
                                                               (_Float16 *)b.pooled, H, d->pool, 0);
        MMRAG_CHECK_HIP(hipGetLastError());
        RUN(mmrag_linear_f16(b.pooled, B, H, lw[2], d->out_dim, nullptr, MMRAG_ACT_NONE, nullptr, b.proj, stream));
        normalize_rows_kernel<<<(unsigned)((B + 3) / 4), 256, 0, s>>>((const _Float16 *)b.proj, out, B, d->out_dim);
        MMRAG_CHECK_HIP(hipGetLastError());
    }
    return MMRAG_OK;
}

static int check_desc(const mmrag_encoder_desc *d) {
    MMRAG_CHECK_ARG(d->arch == MMRAG_ARCH_BERT || d->arch == MMRAG_ARCH_PRELN, "encoder_forward: bad arch %d", d->arch);
    MMRAG_CHECK_ARG(d->hidden % 64 == 0 && d->intermediate % 64 == 0 && d->hidden <= 1024,
                    "encoder_forward: hidden/intermediate must be multiples of 64 (hidden <= 1024)");
    MMRAG_CHECK_ARG(d->pool >= 0 && d->pool <= 2, "encoder_forward: bad pool mode");
    return MMRAG_OK;
}

long long mmrag_internal_last_forward_us(void) { return g_last_forward_us.load(std::memory_order_relaxed); }

int mmrag_encoder_forward(const mmrag_encoder_desc *d, const void *const *w, const int32_t *ids,
                          const int32_t *pos_ids, const int32_t *cu_seqlens, const int32_t *sel, int64_t T, int B,
                          int max_len, float *out, void *workspace, size_t workspace_bytes, void *stream) {
    MMRAG_CHECK_ARG(d && w && ids && pos_ids && cu_seqlens && out, "encoder_forward: null pointer");
    MMRAG_CHECK_ARG(T > 0 && T < INT_MAX && B > 0 && max_len > 0, "encoder_forward: bad shape T=%lld B=%d", (long long)T, B);
    int st;
    RUN(check_desc(d));
    MMRAG_CHECK_ARG(d->pool != 2 || sel, "encoder_forward: MMRAG_POOL_SELECT needs sel");
    EncBuffers b;
    RUN(carve(d, T, B, workspace, workspace_bytes, &b));
    const auto t0 = std::chrono::steady_clock::now();
    RUN(mmrag_embed_ln_f16(ids, pos_ids, w[0], w[1], w[2], (const float *)w[3], (const float *)w[4], b.x, T,
                           d->hidden, d->vocab, d->max_pos, d->ln_eps, stream));
    st = encoder_body(d, w + 5, b, cu_seqlens, sel, T, B, max_len, out, stream);
    // (developer read-out: how long the launches of the last forward took inside the library, as opposed to around it)
    g_last_forward_us.store((long long)std::chrono::duration_cast<std::chrono::microseconds>(
                                std::chrono::steady_clock::now() - t0).count(), std::memory_order_relaxed);
    return st;
}

int mmrag_vit_forward(const mmrag_encoder_desc *d, const void *const *w, const void *pixels, int pixel_kind,
                      const int32_t *cu_seqlens, int B, float *out, void *workspace, size_t workspace_bytes,
                      void *stream) {
    MMRAG_CHECK_ARG(d && w && pixels && cu_seqlens && out, "vit_forward: null pointer");
    MMRAG_CHECK_ARG(B > 0, "vit_forward: bad batch %d", B);
    int st;
    RUN(check_desc(d));
    MMRAG_CHECK_ARG(d->arch == MMRAG_ARCH_PRELN && d->pool == MMRAG_POOL_FIRST, "vit_forward: needs a pre-LN CLS-pooled desc");
    MMRAG_CHECK_ARG(d->patch > 0 && d->image > 0 && d->image % d->patch == 0 && d->patch % 8 == 0,
                    "vit_forward: bad image/patch %d/%d", d->image, d->patch);
    MMRAG_CHECK_ARG(pixel_kind == MMRAG_PIXELS_F16_CHW || pixel_kind == MMRAG_PIXELS_U8_HWC, "vit_forward: bad pixel kind");
    const int G = d->image / d->patch, NP = G * G, S = NP + 1;
    const int PK = 3 * d->patch * d->patch;
    MMRAG_CHECK_ARG(PK % 64 == 0, "vit_forward: 3*patch*patch must be a multiple of 64");
    const int64_t T = (int64_t)B * S;
    EncBuffers b;
    RUN(carve(d, T, B, workspace, workspace_bytes, &b));
    hipStream_t s = (hipStream_t)stream;
    // patches -> b.hm [B*NP, 3*P*P]; patch embeddings -> b.qkv [B*NP, H]
    const long long chunks = (long long)B * NP * (PK / 8);
    patchify_kernel<<<(unsigned)((chunks + 255) / 256), 256, 0, s>>>(pixels, pixel_kind, (_Float16 *)b.hm, B, d->image,
                                                                      d->patch);
    MMRAG_CHECK_HIP(hipGetLastError());
    RUN(mmrag_linear_f16(b.hm, (int64_t)B * NP, PK, w[0], d->hidden, nullptr, MMRAG_ACT_NONE, nullptr, b.qkv, stream));
    vit_assemble_ln_kernel<<<(unsigned)((T + 3) / 4), 256, 0, s>>>(
        (const _Float16 *)b.qkv, (const _Float16 *)w[2], (const _Float16 *)w[1], (const float *)w[3],
        (const float *)w[4], (_Float16 *)b.x, (int)T, d->hidden, S, d->ln_eps);
    MMRAG_CHECK_HIP(hipGetLastError());
    return encoder_body(d, w + 5, b, cu_seqlens, nullptr, T, B, S, out, stream);
}
#undef RUN

}  // extern "C"
