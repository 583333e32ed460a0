// Error reporting, device queries and the small row-movement kernels of libmmrag.so.
#include "mmrag_internal.h"

#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <string.h>

#include <atomic>

namespace mmrag {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// developer switches for A/B runs (mmrag_internal_set_debug; never set by the product): one relaxed word, no getenv
static std::atomic<unsigned> g_debug{0};
unsigned debug_flags() { return g_debug.load(std::memory_order_relaxed); }

int num_cus() {
    static int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

}  // namespace mmrag

using namespace mmrag;

namespace mmrag_impl {

template <typename T>
__device__ inline T from_f32(float x);
template <>
__device__ inline float from_f32<float>(float x) { return x; }
template <>
__device__ inline _Float16 from_f32<_Float16>(float x) { return (_Float16)x; }
template <>
__device__ inline __bf16 from_f32<__bf16>(float x) { return (__bf16)x; }

// dst[n_used + i, :d] = cast(src[i, :d]); pad columns [d, ld) = 0.  One wave per row slab.
template <typename T>
__global__ void append_rows_kernel(T *__restrict__ corpus, int64_t ld, int64_t n_used,
                                   const float *__restrict__ src, int64_t m, int d) {
    const int64_t total = m * ld;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ld;
        const int c = (int)(i - r * ld);
        const float v = c < d ? src[r * d + c] : 0.0f;
        corpus[(n_used + r) * ld + c] = from_f32<T>(v);
    }
}

// dst[i, :] = src[keep[i], :], 16 bytes per lane (rows are multiples of 128 bytes)
__global__ void gather_rows_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src,
                                   int64_t row_vec, const int64_t *__restrict__ keep, int64_t m) {
    const int64_t total = m * row_vec;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / row_vec;
        const int64_t c = i - r * row_vec;
        dst[i] = src[keep[r] * row_vec + c];
    }
}

template <typename T>
__global__ void fetch_rows_kernel(const T *__restrict__ corpus, int64_t ld,
                                  const int64_t *__restrict__ rows, int64_t m, int d,
                                  float *__restrict__ out) {
    const int64_t total = m * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / d;
        const int c = (int)(i - r * d);
        out[i] = (float)corpus[rows[r] * ld + c];
    }
}

inline int grid_for(int64_t total, int block) {
    int64_t g = (total + block - 1) / block;
    int64_t cap = (int64_t)num_cus() * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace mmrag_impl
using namespace mmrag_impl;

extern "C" {

int mmrag_abi_version(void) { return 1; }

// A/B switches of the encoder GEMM (tools/linear_vs_rocblas.py); exported for the tools, absent from include/mmrag.h
void mmrag_internal_set_debug(unsigned flags) { mmrag::g_debug.store(flags, std::memory_order_relaxed); }

const char *mmrag_last_error(void) { return mmrag::g_err; }

int64_t mmrag_padded_dim(int d, int dtype) {
    if (d <= 0 || dtype < 0 || dtype > 2) return -1;
    const int per = 128 / esize(dtype);
    return ((int64_t)d + per - 1) / per * per;
}

int mmrag_append_rows(void *corpus, int64_t capacity, int64_t ld, int dtype, int64_t n_used,
                      const float *new_rows, int64_t m, int d, void *stream) {
    MMRAG_CHECK_ARG(corpus && new_rows, "append_rows: null pointer");
    MMRAG_CHECK_ARG(dtype >= 0 && dtype <= 2, "append_rows: bad dtype %d", dtype);
    MMRAG_CHECK_ARG(d > 0 && ld >= d, "append_rows: need 0 < d <= ld (d=%d ld=%lld)", d, (long long)ld);
    MMRAG_CHECK_ARG(m >= 0 && n_used >= 0 && n_used + m <= capacity,
                    "append_rows: rows [%lld, %lld) exceed capacity %lld", (long long)n_used,
                    (long long)(n_used + m), (long long)capacity);
    if (m == 0) return MMRAG_OK;
    hipStream_t s = (hipStream_t)stream;
    const int block = 256;
    const int grid = grid_for(m * ld, block);
    if (dtype == MMRAG_F32)
        append_rows_kernel<float><<<grid, block, 0, s>>>((float *)corpus, ld, n_used, new_rows, m, d);
    else if (dtype == MMRAG_F16)
        append_rows_kernel<_Float16><<<grid, block, 0, s>>>((_Float16 *)corpus, ld, n_used, new_rows, m, d);
    else
        append_rows_kernel<__bf16><<<grid, block, 0, s>>>((__bf16 *)corpus, ld, n_used, new_rows, m, d);
    MMRAG_CHECK_HIP(hipGetLastError());
    return MMRAG_OK;
}

int mmrag_gather_rows(void *dst, const void *src, int64_t ld, int dtype, const int64_t *keep_rows,
                      int64_t m, void *stream) {
    MMRAG_CHECK_ARG(dst && src && (keep_rows || m == 0), "gather_rows: null pointer");
    MMRAG_CHECK_ARG(dtype >= 0 && dtype <= 2, "gather_rows: bad dtype %d", dtype);
    const int64_t row_bytes = ld * esize(dtype);
    MMRAG_CHECK_ARG(ld > 0 && row_bytes % 16 == 0, "gather_rows: row bytes %lld not a multiple of 16",
                    (long long)row_bytes);
    MMRAG_CHECK_ARG(((uintptr_t)dst % 16) == 0 && ((uintptr_t)src % 16) == 0, "gather_rows: unaligned");
    if (m == 0) return MMRAG_OK;
    const int block = 256;
    gather_rows_kernel<<<grid_for(m * (row_bytes / 16), block), block, 0, (hipStream_t)stream>>>(
        (uint4 *)dst, (const uint4 *)src, row_bytes / 16, keep_rows, m);
    MMRAG_CHECK_HIP(hipGetLastError());
    return MMRAG_OK;
}

int mmrag_fetch_rows_f32(const void *corpus, int64_t ld, int dtype, const int64_t *rows, int64_t m,
                         int d, float *out, void *stream) {
    MMRAG_CHECK_ARG(corpus && out && (rows || m == 0), "fetch_rows: null pointer");
    MMRAG_CHECK_ARG(dtype >= 0 && dtype <= 2, "fetch_rows: bad dtype %d", dtype);
    MMRAG_CHECK_ARG(d > 0 && ld >= d, "fetch_rows: need 0 < d <= ld");
    if (m == 0) return MMRAG_OK;
    hipStream_t s = (hipStream_t)stream;
    const int block = 256;
    const int grid = grid_for(m * d, block);
    if (dtype == MMRAG_F32)
        fetch_rows_kernel<float><<<grid, block, 0, s>>>((const float *)corpus, ld, rows, m, d, out);
    else if (dtype == MMRAG_F16)
        fetch_rows_kernel<_Float16><<<grid, block, 0, s>>>((const _Float16 *)corpus, ld, rows, m, d, out);
    else
        fetch_rows_kernel<__bf16><<<grid, block, 0, s>>>((const __bf16 *)corpus, ld, rows, m, d, out);
    MMRAG_CHECK_HIP(hipGetLastError());
    return MMRAG_OK;
}

int mmrag_copy_to_host_async(void *dst_host, const void *src_dev, size_t bytes, void *stream) {
    MMRAG_CHECK_ARG(dst_host && src_dev, "copy_to_host_async: null pointer");
    MMRAG_CHECK_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return MMRAG_OK;
}

}  // extern "C"
