// The encoder at the REFERENCE's precision: float32 weights, activations and arithmetic, for gfx950 (MI355X).
//
// SentenceTransformer.encode (app/utils/embedder.py:397-403) runs in float32 -- the reference picks a device
// (embedder.py:204-210) and never casts or autocasts -- so a drop-in that must reproduce its scores to 1e-4 needs a
// mode that computes as it does.  encoder.hip is the throughput path (fp16 storage, fp32 accumulation: embeddings
// within ~2e-4 of the float32 model); this file is the faithful one, opt-in (MMRAG_ENCODER_PRECISION=fp32):
//   * every contraction on the exact float32 matrix instruction v_mfma_f32_32x32x2_f32 (a k-ordered fmaf chain, one
//     rounding per product: cdna_hip_programming.md section 3, "FP32-input MFMA"; 1/16 of the fp16 rate, so this mode
//     is MFMA-bound at ~150 TFLOP/s where the fp16 path has 2.5 PFLOP/s);
//   * LayerNorm, softmax, GELU (libm erff) and pooling in float32, as torch computes them;
//   * BERT family only (all-MiniLM-L6-v2, bge-base-en-v1.5: the reference's models); the CLIP towers have no
//     reference behaviour to be faithful to.
// Same packed-sequence layout, weight-table order and pooling semantics as mmrag_encoder_forward.
#include "mmrag_internal.h"
#include "tile_dma.h"

#include <limits.h>
#include <math.h>

using namespace mmrag;

namespace mmrag_impl {

namespace {

__device__ inline float wave_sum_f(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// ---- rows of H floats, one wave per row (H <= 1024, H % 4 == 0): v = src (+ add), LayerNorm, two-pass statistics ----
__device__ inline void ln_row_f32(const float *src, const float *add1, const float *add2, float *dst, const float *g,
                                  const float *b, int H, float eps, int lane) {
    f32x4_t v[4];
    const int nch = H >> 2;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        v[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (c < nch) {
            v[i] = ((const f32x4_t *)src)[c];
            if (add1) v[i] += ((const f32x4_t *)add1)[c];
            if (add2) v[i] += ((const f32x4_t *)add2)[c];
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
    }
    const float mean = wave_sum_f(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (lane + 64 * i < nch) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dx = v[i][e] - mean;
                q += dx * dx;
            }
        }
    const float rstd = 1.0f / sqrtf(wave_sum_f(q) / (float)H + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            const f32x4_t gg = ((const f32x4_t *)g)[c], bb = ((const f32x4_t *)b)[c];
            f32x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * gg[e] + bb[e];
            ((f32x4_t *)dst)[c] = o;
        }
    }
}

__global__ __launch_bounds__(256) void layernorm_f32_kernel(const float *__restrict__ x, float *__restrict__ out,
                                                            const float *__restrict__ g, const float *__restrict__ b,
                                                            int T, int H, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    ln_row_f32(x + (size_t)row * H, nullptr, nullptr, out + (size_t)row * H, g, b, H, eps, threadIdx.x & 63);
}

__global__ __launch_bounds__(256) void embed_ln_f32_kernel(const int *__restrict__ ids, const int *__restrict__ pos_ids,
                                                           const float *__restrict__ tok, const float *__restrict__ pos,
                                                           const float *__restrict__ type0, const float *__restrict__ g,
                                                           const float *__restrict__ b, float *__restrict__ out, int T,
                                                           int H, int vocab, int max_pos, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    int id = ids[row], ps = pos_ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);       // (out-of-range ids are clamped, as the fp16 path does)
    ps = ps < 0 ? 0 : (ps >= max_pos ? max_pos - 1 : ps);
    ln_row_f32(tok + (size_t)id * H, pos + (size_t)ps * H, type0, out + (size_t)row * H, g, b, H, eps, threadIdx.x & 63);
}

// ---- out = act(x . Wt^T + bias) (+ resid), all float32.  128 tokens x 128 features per workgroup, 4 waves of 64 x 64
// (2 x 2 MFMA tiles of 32 x 32), K in 128-byte slabs (32 floats) through a two-stage LDS-DMA ring, XOR-swizzled like
// every slab tile of this library.  v_mfma_f32_32x32x2_f32 takes ONE float per lane and operand: lane (r, h) feeds
// element 16 h + s of its row in step s of a slab -- A and B use the same map, and a dot product does not care in which
// order its terms arrive -- so a lane's sixteen steps are four 16-byte LDS reads. ------------------------------------
struct LinF32 {
    const float *x, *wt, *bias, *resid;
    float *out;
    int M, N, K, act;
};

template <int ACT>
__device__ inline float act_f32(float v) {
    if constexpr (ACT == MMRAG_ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    if constexpr (ACT == MMRAG_ACT_QUICK_GELU) return v / (1.0f + expf(-1.702f * v));
    return v;
}

__global__ __launch_bounds__(256, 2) void linear_f32_kernel(const LinF32 p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BT = 128, BF = 128;
    constexpr int STAGE = (BT + BF) * SLAB;     // 32 KiB
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wt_ = wave >> 1, wf = wave & 1;   // the wave's 64 tokens x 64 features
    const int r32 = lane & 31, h = lane >> 5;
    const int t0 = blockIdx.y * BT, f0 = blockIdx.x * BF;
    const unsigned RB = (unsigned)p.K * 4u;
    const int nk = (int)(RB / SLAB);

    // DMA: 32 pieces of 8 rows per slab (16 of x, 16 of W), 8 per wave; out-of-range rows read as zero
    const int rows_x = p.M - t0 < BT ? p.M - t0 : BT, rows_w = p.N - f0 < BF ? p.N - f0 : BF;
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x + (size_t)t0 * p.K, (unsigned)rows_x * RB);
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.wt + (size_t)f0 * p.K, (unsigned)rows_w * RB);
    unsigned src_off[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) src_off[i] = dma_src_offset((wave * 8 + i) & 15, lane, RB);
    auto issue = [&](int stage, int ks) {
        char *base = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pc = wave * 8 + i;            // pieces 0..15: x rows, 16..31: W rows
            if (pc < 16)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(base + pc * 1024), 16, src_off[i],
                                                         (unsigned)ks * SLAB, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(base + BT * SLAB + (pc - 16) * 1024), 16,
                                                         src_off[i], (unsigned)ks * SLAB, 0, 0);
        }
    };

    f32x16_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;

    issue(0, 0);
    for (int ks = 0; ks < nk; ++ks) {
        if (ks + 1 < nk) {
            issue((ks + 1) & 1, ks + 1);
            wait_vmcnt<8>();        // this slab's eight pieces (the next slab's stay in flight)
        } else {
            wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        const char *st = smem + (ks & 1) * STAGE;
        f32x4_t xa[2][4], wb[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int rx = wt_ * 64 + a * 32 + r32, rw = wf * 64 + a * 32 + r32;
                xa[a][c] = *(const f32x4_t *)(st + frag_offset(rx, 4 * h + c));
                wb[a][c] = *(const f32x4_t *)(st + BT * SLAB + frag_offset(rw, 4 * h + c));
            }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[a][c][e], wb[b][c][e], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_s_barrier();   // every wave has read this stage before the slab after next overwrites it
    }
    // D[token][feature]: the feature on the lane, the token rows (r & 3) + 8 (r >> 2) + 4 h in the registers: a
    // half-wave stores 32 consecutive floats of one token row
    auto finish = [&](auto act_tag) {
        constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int f = f0 + wf * 64 + b * 32 + r32;
            if (f >= p.N) continue;
            const float bias = p.bias ? p.bias[f] : 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int t = t0 + wt_ * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (t < p.M) {
                        float v = act_f32<ACT>(acc[a][b][r] + bias);
                        if (p.resid) v += p.resid[(size_t)t * p.N + f];
                        p.out[(size_t)t * p.N + f] = v;
                    }
                }
        }
    };
    if (p.act == MMRAG_ACT_GELU) finish(std::integral_constant<int, MMRAG_ACT_GELU>{});
    else if (p.act == MMRAG_ACT_QUICK_GELU) finish(std::integral_constant<int, MMRAG_ACT_QUICK_GELU>{});
    else finish(std::integral_constant<int, MMRAG_ACT_NONE>{});
#endif
}

// ---- attention, float32: ctx[t, head*DH + d] = softmax(Q K^T / sqrt(DH), keys of the same sequence) V.  One workgroup =
// (128-query tile, head, sequence), a wave = 32 queries; keys in blocks of 32 staged in LDS.  S^T = K Q^T, so the query
// sits on the lane and the online-softmax state is lane-local; the finished P tile feeds the P V product straight from
// the accumulator registers (step j of the product takes register j: keys (j & 3) + 8 (j >> 2) + 4 h), the matching V
// rows come from LDS.  Same scheme as encoder.hip's attention_kernel, on the float32 matrix instruction. ---------------
template <int DH>
__global__ __launch_bounds__(256) void attention_f32_kernel(const float *__restrict__ qkv, const int *__restrict__ cu,
                                                            float *__restrict__ ctx, int H, float scale) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int KB = 32;              // keys per block
    constexpr int KS = DH + 4;          // K row stride in LDS (floats): 16-byte aligned rows, banks spread
    constexpr int ND = DH / 32;         // 32-row blocks of the output's head dimension
    __shared__ __attribute__((aligned(16))) float k_lds[KB * KS];
    __shared__ __attribute__((aligned(16))) float v_lds[KB * DH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r32 = lane & 31, h = lane >> 5;
    const int seq = blockIdx.z, head = blockIdx.y;
    const int s0 = cu[seq], len = cu[seq + 1] - s0;
    const int q_tile = blockIdx.x * 128;
    if (q_tile >= len) return;
    const int qi = q_tile + wave * 32 + r32;            // this lane's query
    const bool q_ok = qi < len;
    const size_t row3 = (size_t)3 * H;
    // B operand of S^T = K Q^T: lane (query, h) feeds Q[query][16 h' ...]: its half row (DH / 2 floats), scaled
    float qv[DH / 2];
    {
        const float *qp = qkv + (size_t)(s0 + (q_ok ? qi : 0)) * row3 + head * DH + h * (DH / 2);
#pragma unroll
        for (int i = 0; i < DH / 2; i += 4) {
            const f32x4_t t = *(const f32x4_t *)(qp + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) qv[i + e] = q_ok ? t[e] * scale : 0.f;
        }
    }
    f32x16_t o[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int j = 0; j < 16; ++j) o[d][j] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    for (int k0 = 0; k0 < len; k0 += KB) {
        __syncthreads();   // the previous block has been consumed
        // stage K and V rows k0 .. k0+31 (rows past the sequence: zeros, masked below)
        for (int i = threadIdx.x; i < KB * (DH / 4); i += 256) {
            const int kr = i / (DH / 4), c4 = i % (DH / 4);
            f32x4_t kk = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (k0 + kr < len) {
                const float *base = qkv + (size_t)(s0 + k0 + kr) * row3 + head * DH + c4 * 4;
                kk = *(const f32x4_t *)(base + H);
                vv = *(const f32x4_t *)(base + 2 * H);
            }
            *(f32x4_t *)(k_lds + kr * KS + c4 * 4) = kk;
            *(f32x4_t *)(v_lds + kr * DH + c4 * 4) = vv;
        }
        __syncthreads();
        // S^T tile [32 keys][32 queries]: A = K (lane (key, h) feeds K[key][(DH/2) h + s]), B = Q
        f32x16_t s;
#pragma unroll
        for (int j = 0; j < 16; ++j) s[j] = 0.f;
        {
            const float *kp = k_lds + r32 * KS + h * (DH / 2);
#pragma unroll
            for (int i = 0; i < DH / 2; i += 4) {
                const f32x4_t kk = *(const f32x4_t *)(kp + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk[e], qv[i + e], s, 0, 0, 0);
            }
        }
        // register j <-> key k0 + (j & 3) + 8 (j >> 2) + 4 h; the query is this lane's
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int key = k0 + (j & 3) + 8 * (j >> 2) + 4 * h;
            s[j] = key < len ? s[j] : -INFINITY;
            mx = fmaxf(mx, s[j]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));                 // both half-waves hold keys of the same query
        const float m_new = fmaxf(m_run, mx);               // finite: key k0 is always valid
        const float corr = expf(m_run - m_new);             // exp(-inf) = 0 on the first block
        float ps = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            s[j] = expf(s[j] - m_new);
            ps += s[j];
        }
        ps += __shfl_xor(ps, 32);
        l_run = l_run * corr + ps;
        m_run = m_new;
        // O^T[d][query] = corr * O^T + V^T P: A = V^T (lane (d, h) feeds V[key(j, h)][d]), B = P = register j as it stands
#pragma unroll
        for (int d = 0; d < ND; ++d) {
#pragma unroll
            for (int j = 0; j < 16; ++j) o[d][j] *= corr;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int key = (j & 3) + 8 * (j >> 2) + 4 * h;
                o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(v_lds[key * DH + d * 32 + r32], s[j], o[d], 0, 0, 0);
            }
        }
    }
    if (q_ok) {
        const float inv = 1.0f / l_run;
        float *dst = ctx + (size_t)(s0 + qi) * H + head * DH;
        // register j of block d <-> head dimension d * 32 + (j & 3) + 8 (j >> 2) + 4 h: four consecutive floats per store
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4_t v4;
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = o[d][4 * g + e] * inv;
                *(f32x4_t *)(dst + d * 32 + 8 * g + 4 * h) = v4;
            }
    }
#endif
}

// ---- pooling (masked mean / first / selected token) + L2 normalise, one workgroup per sequence ----------------------
__global__ __launch_bounds__(256) void pool_norm_f32_kernel(const float *__restrict__ x, const int *__restrict__ cu,
                                                            const int *__restrict__ sel, float *__restrict__ out, int H,
                                                            int pool, int normalize) {
    __shared__ float red[4];
    const int b = blockIdx.x, s0 = cu[b], len = cu[b + 1] - s0;
    float v[4] = {0.f, 0.f, 0.f, 0.f};   // features threadIdx.x + 256 i, H <= 1024
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + 256 * i;
        if (f >= H) continue;
        if (pool == MMRAG_POOL_MEAN) {
            float a = 0.f;
            for (int t = 0; t < len; ++t) a += x[(size_t)(s0 + t) * H + f];
            v[i] = a / fmaxf((float)len, 1e-9f);
        } else {
            const int t = pool == MMRAG_POOL_SELECT ? sel[b] : 0;
            v[i] = x[(size_t)(s0 + t) * H + f];
        }
        ss += v[i] * v[i];
    }
    ss = wave_sum_f(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]);
    const float inv = normalize ? 1.0f / fmaxf(nrm, 1e-12f) : 1.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + 256 * i;
        if (f < H) out[(size_t)b * H + f] = v[i] * inv;
    }
}

size_t align256f(size_t x) { return (x + 255) / 256 * 256; }

int launch_linear_f32(const float *x, int64_t M, int K, const float *wt, int N, const float *bias, int act,
                      const float *resid, float *out, hipStream_t s) {
    MMRAG_CHECK_ARG(x && wt && out && M > 0 && N > 0 && K > 0, "linear_f32: bad arguments");
    MMRAG_CHECK_ARG(K % 32 == 0, "linear_f32: K = %d must be a multiple of 32 (128-byte K-slabs)", K);
    MMRAG_CHECK_ARG(M < INT_MAX && (long long)128 * K * 4 < (long long)UINT_MAX, "linear_f32: shape out of range");
    LinF32 p{x, wt, bias, resid, out, (int)M, N, K, act};
    dim3 grid((unsigned)((N + 127) / 128), (unsigned)((M + 127) / 128));
    linear_f32_kernel<<<grid, 256, 0, s>>>(p);
    MMRAG_CHECK_HIP(hipGetLastError());
    return MMRAG_OK;
}

}  // namespace

}  // namespace mmrag_impl
using namespace mmrag_impl;

extern "C" {

int mmrag_linear_f32(const float *x, int64_t M, int K, const float *wt, int N, const float *bias, int act,
                     const float *resid, float *out, void *stream) {
    return launch_linear_f32(x, M, K, wt, N, bias, act, resid, out, (hipStream_t)stream);
}

size_t mmrag_encoder_f32_workspace_bytes(const mmrag_encoder_desc *d, int64_t T, int B) {
    if (!d || T <= 0 || B <= 0) return 0;
    const size_t H = (size_t)d->hidden, I = (size_t)d->intermediate, Tz = (size_t)T;
    return 2 * align256f(Tz * H * 4) + align256f(Tz * 3 * H * 4) + align256f(Tz * H * 4) + align256f(Tz * I * 4) + 256;
}

#define RUN(call) do { if ((st = (call)) != MMRAG_OK) return st; } while (0)

int mmrag_encoder_forward_f32(const mmrag_encoder_desc *d, const void *const *w, const int32_t *ids,
                              const int32_t *pos_ids, const int32_t *cu_seqlens, const int32_t *sel, int64_t T, int B,
                              int max_len, float *out, void *workspace, size_t workspace_bytes, void *stream) {
    MMRAG_CHECK_ARG(d && w && ids && pos_ids && cu_seqlens && out, "encoder_forward_f32: null pointer");
    MMRAG_CHECK_ARG(T > 0 && T < INT_MAX && B > 0 && max_len > 0, "encoder_forward_f32: bad shape T=%lld B=%d", (long long)T, B);
    MMRAG_CHECK_ARG(d->arch == MMRAG_ARCH_BERT, "encoder_forward_f32: the float32 mode covers the BERT family only");
    MMRAG_CHECK_ARG(d->hidden % 64 == 0 && d->intermediate % 64 == 0 && d->hidden <= 1024,
                    "encoder_forward_f32: hidden/intermediate must be multiples of 64 (hidden <= 1024)");
    MMRAG_CHECK_ARG(d->n_heads > 0 && d->hidden % d->n_heads == 0, "encoder_forward_f32: bad head count");
    const int H = d->hidden, I = d->intermediate, DH = H / d->n_heads;
    MMRAG_CHECK_ARG(DH == 32 || DH == 64, "encoder_forward_f32: head dimension %d (32 and 64 are built)", DH);
    MMRAG_CHECK_ARG(d->pool >= 0 && d->pool <= 2 && (d->pool != 2 || sel), "encoder_forward_f32: bad pool mode");
    MMRAG_CHECK_ARG(!d->causal, "encoder_forward_f32: causal attention is not built (BERT is bidirectional)");
    const size_t need = mmrag_encoder_f32_workspace_bytes(d, T, B);
    if (!workspace || workspace_bytes < need) {
        set_error("encoder_forward_f32: workspace %zu bytes < required %zu", workspace_bytes, need);
        return MMRAG_EWORKSPACE;
    }
    const size_t Tz = (size_t)T;
    char *pw = (char *)(((uintptr_t)workspace + 255) / 256 * 256);
    auto take = [&](size_t bytes) { char *r = pw; pw += align256f(bytes); return (float *)r; };
    float *x = take(Tz * H * 4), *y = take(Tz * H * 4), *qkv = take(Tz * 3 * H * 4), *ctx = take(Tz * H * 4);
    float *hm = take(Tz * (size_t)I * 4);
    hipStream_t s = (hipStream_t)stream;
    const unsigned row_grid = (unsigned)((T + 3) / 4);
    int st;
    embed_ln_f32_kernel<<<row_grid, 256, 0, s>>>(ids, pos_ids, (const float *)w[0], (const float *)w[1],
                                                 (const float *)w[2], (const float *)w[3], (const float *)w[4], x,
                                                 (int)T, H, d->vocab, d->max_pos, d->ln_eps);
    MMRAG_CHECK_HIP(hipGetLastError());
    const void *const *lw = w + 5;
    const float scale = 1.0f / sqrtf((float)DH);
    const dim3 agrid((unsigned)((max_len + 127) / 128), (unsigned)d->n_heads, (unsigned)B);
    for (int l = 0; l < d->n_layers; ++l, lw += 12) {
        RUN(launch_linear_f32(x, T, H, (const float *)lw[0], 3 * H, (const float *)lw[1], MMRAG_ACT_NONE, nullptr, qkv, s));
        if (DH == 32) attention_f32_kernel<32><<<agrid, 256, 0, s>>>(qkv, cu_seqlens, ctx, H, scale);
        else attention_f32_kernel<64><<<agrid, 256, 0, s>>>(qkv, cu_seqlens, ctx, H, scale);
        MMRAG_CHECK_HIP(hipGetLastError());
        RUN(launch_linear_f32(ctx, T, H, (const float *)lw[2], H, (const float *)lw[3], MMRAG_ACT_NONE, x, y, s));
        layernorm_f32_kernel<<<row_grid, 256, 0, s>>>(y, x, (const float *)lw[4], (const float *)lw[5], (int)T, H, d->ln_eps);
        RUN(launch_linear_f32(x, T, H, (const float *)lw[6], I, (const float *)lw[7], d->act, nullptr, hm, s));
        RUN(launch_linear_f32(hm, T, I, (const float *)lw[8], H, (const float *)lw[9], MMRAG_ACT_NONE, x, y, s));
        layernorm_f32_kernel<<<row_grid, 256, 0, s>>>(y, x, (const float *)lw[10], (const float *)lw[11], (int)T, H, d->ln_eps);
        MMRAG_CHECK_HIP(hipGetLastError());
    }
    pool_norm_f32_kernel<<<(unsigned)B, 256, 0, s>>>(x, cu_seqlens, sel, out, H, d->pool, d->normalize);
    MMRAG_CHECK_HIP(hipGetLastError());
    return MMRAG_OK;
}
#undef RUN

}  // extern "C"
