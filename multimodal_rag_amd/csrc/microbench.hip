// Measured peaks for the roofline report ("measure a stream-copy bandwidth and an
// MFMA micro-benchmark peak on the box; fractions against both vendor and measured").  Two tiny kernels and a device
// query, exported through the C-ABI so bench.py measures them in the same run as the hot path.
#include "mmrag_internal.h"
#include "tile_dma.h"

using namespace mmrag;

namespace mmrag_impl {

// 16 bytes per lane, grid-stride: the copy the 6.3 TB/s figure of MI355X_MICROARCH.md is quoted on
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_copy_kernel(u32x4_t *__restrict__ dst, const u32x4_t *__restrict__ src,
                                                          long long n_vec) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

// read-only: the sum keeps the loads alive; one store per thread at the end
__global__ __launch_bounds__(256) void stream_read_kernel(const u32x4_t *__restrict__ src, long long n_vec,
                                                          float *__restrict__ out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    u32x4_t acc = {0u, 0u, 0u, 0u};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
        const u32x4_t v = __builtin_nontemporal_load(src + i);
        acc.x ^= v.x;
        acc.y ^= v.y;
        acc.z ^= v.z;
        acc.w ^= v.w;
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float((acc.x ^ acc.y ^ acc.z ^ acc.w) & 0x3fffffffu);
}

__global__ __launch_bounds__(256) void stream_write_kernel(u32x4_t *__restrict__ dst, long long n_vec) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const u32x4_t v = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride)
        __builtin_nontemporal_store(v, dst + i);
}

// Back-to-back v_mfma_f32_32x32x16_f16 on four independent accumulators per wave, operands from `seed` (random data:
// the clock the chip holds under an MFMA load depends on the operand values, so zeros would flatter the peak).
// One wave per SIMD at 256 threads per workgroup and one workgroup per CU.
__global__ __launch_bounds__(256) void mfma_peak_kernel(const half8_t *__restrict__ seed, float *__restrict__ out,
                                                        int iters) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int lane = threadIdx.x & 63;
    const half8_t a0 = seed[lane], a1 = seed[64 + lane], b0 = seed[128 + lane], b1 = seed[192 + lane];
    f32x16_t c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c3, 0, 0, 0);
    }
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += c0[j] + c1[j] + c2[j] + c3[j];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
#endif
}

// the same 64 x 64 output tile per wave on the 16x16x32 shape: 4 A x 4 B fragments, sixteen accumulators of 4.  Which
// shape the chip clocks higher under load is measured, not assumed (MI355X_MICROARCH.md, DVFS give-back item 7)
__global__ __launch_bounds__(256) void mfma_peak16_kernel(const half8_t *__restrict__ seed, float *__restrict__ out,
                                                          int iters) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int lane = threadIdx.x & 63;
    // sixteen NAMED accumulators and one asm statement per step: from an indexed array hipcc made a loop of
    // accumulator-file moves and s_nops around the MFMAs (0.84 of the 32x32x16 loop's rate, all of it that)
    const half8_t a0 = seed[lane], a1 = seed[(64 + lane) & 255], a2 = seed[(128 + lane) & 255], a3 = seed[(192 + lane) & 255];
    const half8_t b0 = seed[(17 * lane) & 255], b1 = seed[(17 * lane + 64) & 255], b2 = seed[(17 * lane + 128) & 255],
                  b3 = seed[(17 * lane + 192) & 255];
    f32x4_t c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {}, c6 = {}, c7 = {}, c8 = {}, c9 = {}, c10 = {}, c11 = {},
            c12 = {}, c13 = {}, c14 = {}, c15 = {};
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %16, %20, %0\n\t"
                     "v_mfma_f32_16x16x32_f16 %1, %16, %21, %1\n\t"
                     "v_mfma_f32_16x16x32_f16 %2, %16, %22, %2\n\t"
                     "v_mfma_f32_16x16x32_f16 %3, %16, %23, %3\n\t"
                     "v_mfma_f32_16x16x32_f16 %4, %17, %20, %4\n\t"
                     "v_mfma_f32_16x16x32_f16 %5, %17, %21, %5\n\t"
                     "v_mfma_f32_16x16x32_f16 %6, %17, %22, %6\n\t"
                     "v_mfma_f32_16x16x32_f16 %7, %17, %23, %7\n\t"
                     "v_mfma_f32_16x16x32_f16 %8, %18, %20, %8\n\t"
                     "v_mfma_f32_16x16x32_f16 %9, %18, %21, %9\n\t"
                     "v_mfma_f32_16x16x32_f16 %10, %18, %22, %10\n\t"
                     "v_mfma_f32_16x16x32_f16 %11, %18, %23, %11\n\t"
                     "v_mfma_f32_16x16x32_f16 %12, %19, %20, %12\n\t"
                     "v_mfma_f32_16x16x32_f16 %13, %19, %21, %13\n\t"
                     "v_mfma_f32_16x16x32_f16 %14, %19, %22, %14\n\t"
                     "v_mfma_f32_16x16x32_f16 %15, %19, %23, %15"
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7), "+v"(c8), "+v"(c9),
                       "+v"(c10), "+v"(c11), "+v"(c12), "+v"(c13), "+v"(c14), "+v"(c15)
                     : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
    }
    asm volatile("s_nop 15\n\ts_nop 7");   // last MFMA's D -> VALU readers
    const f32x4_t t = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + c8 + c9 + c10 + c11 + c12 + c13 + c14 + c15;
    const float s = t[0] + t[1] + t[2] + t[3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
#endif
}


// Developer probe: cycles per v_mfma_f32_16x16x32_f16 in loops that add, one at a time, what the search walk's k-step has
// around its MFMAs.  MODE 0: eight MFMAs per statement, A and B in arch VGPRs; 1: B in the accumulator file; 2: B rotating
// over 16 fragments (64 registers) of the accumulator file; 3: (2) + the statement's v_xad and two ds_read_b128 of the
// NEXT statement's A fragments; 4: (3) + the closing s_waitcnt lgkmcnt(0); 5: (4) + one 1 KiB LDS-DMA piece after every
// other statement (per-lane offsets, the walk's form), 6: the same piece without a per-lane offset (timing only),
// 7: as 5 with the two pieces of an iteration back to back; 8: the statement of 4 with its fillers spread over the MFMA
// gaps; 9: as 5 with global_load_lds_dwordx4 instead of buffer_load ... lds.  stamps[2 wg] = shader cycles (s_memtime),
// stamps[2 wg + 1] = 100 MHz ticks (s_memrealtime) of wave 0's loop.
#define MMRAG_PROBE_8(QC, Q0, Q1, Q2, Q3, PRE, POST)                                                                  \
    asm volatile(PRE "v_mfma_f32_16x16x32_f16 %[d0], %[a0], %[" Q0 "], %[d0]\n\t"                                     \
                     "v_mfma_f32_16x16x32_f16 %[d1], %[a0], %[" Q1 "], %[d1]\n\t"                                     \
                     "v_mfma_f32_16x16x32_f16 %[d2], %[a0], %[" Q2 "], %[d2]\n\t"                                     \
                     "v_mfma_f32_16x16x32_f16 %[d3], %[a0], %[" Q3 "], %[d3]\n\t"                                     \
                     "v_mfma_f32_16x16x32_f16 %[d4], %[a1], %[" Q0 "], %[d4]\n\t"                                     \
                     "v_mfma_f32_16x16x32_f16 %[d5], %[a1], %[" Q1 "], %[d5]\n\t"                                     \
                     "v_mfma_f32_16x16x32_f16 %[d6], %[a1], %[" Q2 "], %[d6]\n\t"                                     \
                     "v_mfma_f32_16x16x32_f16 %[d7], %[a1], %[" Q3 "], %[d7]" POST                                    \
                 : [d0] "+v"(d0), [d1] "+v"(d1), [d2] "+v"(d2), [d3] "+v"(d3), [d4] "+v"(d4), [d5] "+v"(d5),           \
                   [d6] "+v"(d6), [d7] "+v"(d7), [n0] "=&v"(n0), [n1] "=&v"(n1), [t] "=&v"(tmp)                        \
                 : [a0] "v"(a0), [a1] "v"(a1), [q0] QC(q0), [q1] QC(q1), [q2] QC(q2), [q3] QC(q3), [lo] "v"(lo),       \
                   [st] "s"(st))
// mode 8: the statement of mode 4 with its fillers spread over the MFMA gaps (one per gap) instead of bunched in front
#define MMRAG_PROBE_8I(QC)                                                                                            \
    asm volatile("v_mfma_f32_16x16x32_f16 %[d0], %[a0], %[q0], %[d0]\n\t"                                             \
                 "v_xad_u32 %[t], %[lo], 64, %[st]\n\t"                                                               \
                 "v_mfma_f32_16x16x32_f16 %[d1], %[a0], %[q1], %[d1]\n\t"                                             \
                 "ds_read_b128 %[n0], %[t]\n\t"                                                                       \
                 "v_mfma_f32_16x16x32_f16 %[d2], %[a0], %[q2], %[d2]\n\t"                                             \
                 "ds_read_b128 %[n1], %[t] offset:2048\n\t"                                                           \
                 "v_mfma_f32_16x16x32_f16 %[d3], %[a0], %[q3], %[d3]\n\t"                                             \
                 "v_mfma_f32_16x16x32_f16 %[d4], %[a1], %[q0], %[d4]\n\t"                                             \
                 "v_mfma_f32_16x16x32_f16 %[d5], %[a1], %[q1], %[d5]\n\t"                                             \
                 "v_mfma_f32_16x16x32_f16 %[d6], %[a1], %[q2], %[d6]\n\t"                                             \
                 "v_mfma_f32_16x16x32_f16 %[d7], %[a1], %[q3], %[d7]\n\t"                                             \
                 "s_waitcnt lgkmcnt(0)"                                                                               \
                 : [d0] "+v"(d0), [d1] "+v"(d1), [d2] "+v"(d2), [d3] "+v"(d3), [d4] "+v"(d4), [d5] "+v"(d5),           \
                   [d6] "+v"(d6), [d7] "+v"(d7), [n0] "=&v"(n0), [n1] "=&v"(n1), [t] "=&v"(tmp)                        \
                 : [a0] "v"(a0), [a1] "v"(a1), [q0] QC(q0), [q1] QC(q1), [q2] QC(q2), [q3] QC(q3), [lo] "v"(lo),       \
                   [st] "s"(st))
template <int MODE>
__device__ __forceinline__ void probe_stmt(f32x4_t &d0, f32x4_t &d1, f32x4_t &d2, f32x4_t &d3, f32x4_t &d4, f32x4_t &d5,
                                           f32x4_t &d6, f32x4_t &d7, const half8_t a0, const half8_t a1, const half8_t q0,
                                           const half8_t q1, const half8_t q2, const half8_t q3, half8_t &n0, half8_t &n1,
                                           const unsigned lo, const unsigned st) {
    unsigned tmp;
    if constexpr (MODE == 0) MMRAG_PROBE_8("v", "q0", "q1", "q2", "q3", "", "");
    else if constexpr (MODE <= 2) MMRAG_PROBE_8("a", "q0", "q1", "q2", "q3", "", "");
    else if constexpr (MODE == 3)
        MMRAG_PROBE_8("a", "q0", "q1", "q2", "q3",
                      "v_xad_u32 %[t], %[lo], 64, %[st]\n\tds_read_b128 %[n0], %[t]\n\tds_read_b128 %[n1], %[t] offset:2048\n\t", "");
    else if constexpr (MODE == 8) MMRAG_PROBE_8I("a");
    else   // 4 and the LDS-DMA modes
        MMRAG_PROBE_8("a", "q0", "q1", "q2", "q3",
                      "v_xad_u32 %[t], %[lo], 64, %[st]\n\tds_read_b128 %[n0], %[t]\n\tds_read_b128 %[n1], %[t] offset:2048\n\t",
                      "\n\ts_waitcnt lgkmcnt(0)");
    if constexpr (MODE < 3) { n0 = a0; n1 = a1; }
}
template <int MODE>
__global__ __launch_bounds__(256, 1) void mfma_probe_kernel(const half8_t *__restrict__ seed, float *__restrict__ out,
                                                            int iters, long long *__restrict__ stamps) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(1024))) half8_t tile[4096];   // 64 KiB of fragments
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) tile[i] = seed[i & 255];
    __syncthreads();
    half8_t q[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) q[j] = seed[(17 * lane + 16 * j) & 255];
    half8_t a0 = seed[lane], a1 = seed[(64 + lane) & 255], b0 = a1, b1 = a0;
    f32x4_t c[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) c[j] = f32x4_t{};
    const unsigned lo = (unsigned)(size_t)(__attribute__((address_space(3))) void *)tile + lane * 16;
    // (LDS-DMA modes) pieces land in the upper half of `tile`, which the statements do not read; source: the seed (L2)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)seed, 0, 4096, 0x00020000);
    const unsigned voff = (unsigned)((lane >> 3) * 128 + ((lane & 7) ^ (lane >> 4)) * 16);
    auto piece = [&](int it, int h) {
        typedef __attribute__((address_space(3))) void *lds_t;
        char *dst = (char *)tile + 32768 + (threadIdx.x >> 6) * 8192 + (((it << 1) + h) & 7) * 1024;
        if constexpr (MODE == 9)   // the same piece as a global (not buffer) LDS-DMA load: per-lane 64-bit addresses
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)seed + voff + (h + (it & 1) * 2) * 1024),
                                             (lds_t)dst, 16, 0, 2);
        else if constexpr (MODE == 6) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_t)dst, 16, 0, (h + (it & 1) * 2) * 1024, 0, 2);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_t)dst, 16, voff, (h + (it & 1) * 2) * 1024, 0, 2);
    };
    const unsigned long long t0c = __builtin_amdgcn_s_memtime(), t0r = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        const unsigned st = (unsigned)((it & 7) * 4096);
        constexpr int R = MODE >= 2 ? 4 : 0;   // B fragments rotate over q[0..15] / stay q[0..3]
        probe_stmt<MODE>(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], a0, a1, q[0], q[1], q[2], q[3], b0, b1, lo, st);
        probe_stmt<MODE>(c[8], c[9], c[10], c[11], c[12], c[13], c[14], c[15], b0, b1, q[R], q[R + 1], q[R + 2], q[R + 3], a0, a1, lo, st + 1024);
        if constexpr (MODE == 5 || MODE == 6 || MODE == 9) piece(it, 0);
        probe_stmt<MODE>(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], a0, a1, q[2 * R], q[2 * R + 1], q[2 * R + 2], q[2 * R + 3], b0, b1, lo, st + 2048);
        probe_stmt<MODE>(c[8], c[9], c[10], c[11], c[12], c[13], c[14], c[15], b0, b1, q[3 * R], q[3 * R + 1], q[3 * R + 2], q[3 * R + 3], a0, a1, lo, st + 3072);
        if constexpr ((MODE >= 5 && MODE <= 7) || MODE == 9) piece(it, 1);
        if constexpr (MODE == 7) piece(it, 0);
        if constexpr ((MODE >= 5 && MODE <= 7) || MODE == 9) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    if constexpr ((MODE >= 5 && MODE <= 7) || MODE == 9) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1c = __builtin_amdgcn_s_memtime(), t1r = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_nop 15\n\ts_nop 7");
    f32x4_t t = c[0];
#pragma unroll
    for (int j = 1; j < 16; ++j) t += c[j];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = t[0] + t[1] + t[2] + t[3];
    if (threadIdx.x == 0 && stamps) {
        stamps[2 * blockIdx.x] = (long long)(t1c - t0c);
        stamps[2 * blockIdx.x + 1] = (long long)(t1r - t0r);
    }
#endif
}

}  // namespace mmrag_impl
using namespace mmrag_impl;

extern "C" {

int mmrag_device_info(int *n_cus, int *max_clock_mhz, int64_t *hbm_bytes) {
    int dev = 0;
    MMRAG_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    MMRAG_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    if (n_cus) *n_cus = prop.multiProcessorCount;
    if (max_clock_mhz) *max_clock_mhz = prop.clockRate / 1000;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return MMRAG_OK;
}

int mmrag_bench_stream_copy(void *dst, const void *src, int64_t bytes, void *stream) {
    MMRAG_CHECK_ARG(dst && src && bytes > 0 && bytes % 16 == 0, "bench_stream_copy: need 16-byte multiples");
    MMRAG_CHECK_ARG(((uintptr_t)dst % 16) == 0 && ((uintptr_t)src % 16) == 0, "bench_stream_copy: 16-byte alignment");
    stream_copy_kernel<<<num_cus() * 8, 256, 0, (hipStream_t)stream>>>((u32x4_t *)dst, (const u32x4_t *)src, bytes / 16);
    MMRAG_CHECK_HIP(hipGetLastError());
    return MMRAG_OK;
}

int mmrag_bench_stream_read(const void *src, int64_t bytes, float *out, void *stream) {
    MMRAG_CHECK_ARG(src && out && bytes > 0 && bytes % 16 == 0, "bench_stream_read: need 16-byte multiples");
    MMRAG_CHECK_ARG(((uintptr_t)src % 16) == 0, "bench_stream_read: 16-byte alignment");
    stream_read_kernel<<<num_cus() * 8, 256, 0, (hipStream_t)stream>>>((const u32x4_t *)src, bytes / 16, out);
    MMRAG_CHECK_HIP(hipGetLastError());
    return MMRAG_OK;
}

int mmrag_bench_stream_write(void *dst, int64_t bytes, void *stream) {
    MMRAG_CHECK_ARG(dst && bytes > 0 && bytes % 16 == 0, "bench_stream_write: need 16-byte multiples");
    MMRAG_CHECK_ARG(((uintptr_t)dst % 16) == 0, "bench_stream_write: 16-byte alignment");
    stream_write_kernel<<<num_cus() * 8, 256, 0, (hipStream_t)stream>>>((u32x4_t *)dst, bytes / 16);
    MMRAG_CHECK_HIP(hipGetLastError());
    return MMRAG_OK;
}

int mmrag_bench_mfma_f16_16x16x32(const void *seed, float *out, int iters, int64_t *flops, void *stream) {
    MMRAG_CHECK_ARG(seed && out && iters > 0, "bench_mfma_f16_16x16x32: bad arguments");
    const int grid = num_cus();
    mfma_peak16_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const half8_t *)seed, out, iters);
    MMRAG_CHECK_HIP(hipGetLastError());
    if (flops) *flops = (int64_t)grid * 4 /*waves*/ * (int64_t)iters * 16 /*MFMAs*/ * (2LL * 16 * 16 * 32);
    return MMRAG_OK;
}

int mmrag_bench_mfma_f16(const void *seed, float *out, int iters, int64_t *flops, void *stream) {
    MMRAG_CHECK_ARG(seed && out && iters > 0, "bench_mfma_f16: bad arguments");
    const int grid = num_cus();
    mfma_peak_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const half8_t *)seed, out, iters);
    MMRAG_CHECK_HIP(hipGetLastError());
    if (flops) *flops = (int64_t)grid * 4 /*waves*/ * (int64_t)iters * 4 /*MFMAs*/ * (2LL * 32 * 32 * 16);
    return MMRAG_OK;
}

// developer probe (tools/mfma_probe.py): 32 MFMAs per iteration and wave; not part of the reference-facing surface
int mmrag_internal_mfma_probe(const void *seed, float *out, int iters, int mode, long long *stamps, void *stream) {
    MMRAG_CHECK_ARG(seed && out && iters > 0 && mode >= 0 && mode <= 9, "mfma_probe: bad arguments");
    const int grid = num_cus();
    hipStream_t s = (hipStream_t)stream;
    switch (mode) {
    case 0: mfma_probe_kernel<0><<<grid, 256, 0, s>>>((const half8_t *)seed, out, iters, stamps); break;
    case 1: mfma_probe_kernel<1><<<grid, 256, 0, s>>>((const half8_t *)seed, out, iters, stamps); break;
    case 2: mfma_probe_kernel<2><<<grid, 256, 0, s>>>((const half8_t *)seed, out, iters, stamps); break;
    case 3: mfma_probe_kernel<3><<<grid, 256, 0, s>>>((const half8_t *)seed, out, iters, stamps); break;
    case 4: mfma_probe_kernel<4><<<grid, 256, 0, s>>>((const half8_t *)seed, out, iters, stamps); break;
    case 5: mfma_probe_kernel<5><<<grid, 256, 0, s>>>((const half8_t *)seed, out, iters, stamps); break;
    case 6: mfma_probe_kernel<6><<<grid, 256, 0, s>>>((const half8_t *)seed, out, iters, stamps); break;
    case 7: mfma_probe_kernel<7><<<grid, 256, 0, s>>>((const half8_t *)seed, out, iters, stamps); break;
    case 8: mfma_probe_kernel<8><<<grid, 256, 0, s>>>((const half8_t *)seed, out, iters, stamps); break;
    default: mfma_probe_kernel<9><<<grid, 256, 0, s>>>((const half8_t *)seed, out, iters, stamps); break;
    }
    MMRAG_CHECK_HIP(hipGetLastError());
    return MMRAG_OK;
}

}  // extern "C"
