// CLIP image front end of libmmrag.so: shortest-edge bicubic resize + centre crop on uint8 HWC
// images, bit-exact with Pillow's 8-bit resampler (what transformers' CLIP processor runs on the
// host).  Two integer passes -- horizontal, round to uint8, vertical, round to uint8 -- with
// 22-bit fixed-point taps computed on the host in double precision exactly as Pillow does.
// BASELINE config 4 has no reference behaviour; the oracle is
// oracle/clip_oracle.py:resize_u8, pinned against PIL.Image.resize.
#include <math.h>

#include "mmrag_internal.h"

using namespace mmrag;

namespace mmrag_impl {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ inline uint8_t clip8(int acc) {
    const int v = acc >> PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// tmp[(y - y_lo), ox, c] = clip8(half + sum_t src[y, bx[ox].first + t, c] * kx[ox, t])
// one workgroup per source row; threads stride over output columns
__global__ void resize_rows_kernel(const uint8_t *__restrict__ src, int64_t src_row_bytes, int y_lo,
                                   const int32_t *__restrict__ bx, const int32_t *__restrict__ kx, int ksx,
                                   int out_w, uint8_t *__restrict__ tmp) {
    const int y = y_lo + blockIdx.x;
    const uint8_t *row = src + (int64_t)y * src_row_bytes;
    uint8_t *out = tmp + (int64_t)blockIdx.x * out_w * 3;
    for (int ox = threadIdx.x; ox < out_w; ox += blockDim.x) {
        const int x0 = bx[2 * ox], n = bx[2 * ox + 1];
        const int32_t *k = kx + (int64_t)ox * ksx;
        int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
        const uint8_t *p = row + (int64_t)x0 * 3;
        for (int t = 0; t < n; ++t) {
            const int w = k[t];
            a0 += p[3 * t] * w;
            a1 += p[3 * t + 1] * w;
            a2 += p[3 * t + 2] * w;
        }
        out[3 * ox] = clip8(a0);
        out[3 * ox + 1] = clip8(a1);
        out[3 * ox + 2] = clip8(a2);
    }
}

// dst[oy, i] = clip8(half + sum_t tmp[by[oy].first + t - y_lo, i] * ky[oy, t]),  i over out_w*3 bytes
__global__ void resize_cols_kernel(const uint8_t *__restrict__ tmp, int y_lo, const int32_t *__restrict__ by,
                                   const int32_t *__restrict__ ky, int ksy, int row_elems,
                                   uint8_t *__restrict__ dst) {
    const int oy = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= row_elems) return;
    const int y0 = by[2 * oy], n = by[2 * oy + 1];
    const int32_t *k = ky + (int64_t)oy * ksy;
    const uint8_t *p = tmp + (int64_t)(y0 - y_lo) * row_elems + i;
    int acc = 1 << (PRECISION_BITS - 1);
    for (int t = 0; t < n; ++t) acc += p[(int64_t)t * row_elems] * k[t];
    dst[(int64_t)oy * row_elems + i] = clip8(acc);
}

static inline double bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

}  // namespace mmrag_impl
using namespace mmrag_impl;

extern "C" {

int mmrag_resample_ksize(int in_size, int out_size) {
    if (in_size <= 0 || out_size <= 0) return 0;
    if (in_size == out_size) return 1;
    double filterscale = (double)in_size / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    return (int)ceil(2.0 * filterscale) * 2 + 1;
}

// Host: fixed-point bicubic taps of output indices [first, first + count) of an in_size -> out_size
// resample (Pillow: precompute_coeffs + normalize_coeffs_8bpc).  in_size == out_size gives identity
// taps, as Pillow skips that pass.
int mmrag_resample_coeffs(int in_size, int out_size, int first, int count, int32_t *bounds, int32_t *taps) {
    MMRAG_CHECK_ARG(in_size > 0 && out_size > 0 && first >= 0 && count >= 0 && first + count <= out_size && bounds && taps,
                    "resample_coeffs: bad arguments (in=%d out=%d first=%d count=%d)", in_size, out_size, first, count);
    const int ksize = mmrag_resample_ksize(in_size, out_size);
    if (in_size == out_size) {
        for (int i = 0; i < count; ++i) {
            bounds[2 * i] = first + i;
            bounds[2 * i + 1] = 1;
            taps[i] = 1 << PRECISION_BITS;
        }
        return MMRAG_OK;
    }
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const double ss = 1.0 / filterscale;
    double *k = new double[ksize];
    for (int i = 0; i < count; ++i) {
        const int xx = first + i;
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            const double w = bicubic((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        int32_t *t = taps + (size_t)i * ksize;
        for (int x = 0; x < ksize; ++x) {
            double w = 0.0;
            if (x < xmax) w = ww != 0.0 ? k[x] / ww : k[x];
            t[x] = w < 0 ? (int32_t)(-0.5 + w * (1 << PRECISION_BITS)) : (int32_t)(0.5 + w * (1 << PRECISION_BITS));
        }
        bounds[2 * i] = xmin;
        bounds[2 * i + 1] = xmax;
    }
    delete[] k;
    return MMRAG_OK;
}

int mmrag_resize_crop_u8(const uint8_t *src, int H, int W, int64_t src_row_bytes, const int32_t *bx,
                         const int32_t *kx, int ksx, const int32_t *by, const int32_t *ky, int ksy, int out_h,
                         int out_w, int y_lo, int y_hi, uint8_t *tmp, uint8_t *dst, void *stream) {
    MMRAG_CHECK_ARG(src && bx && kx && by && ky && tmp && dst, "resize_crop_u8: null pointer");
    MMRAG_CHECK_ARG(H > 0 && W > 0 && src_row_bytes >= (int64_t)W * 3, "resize_crop_u8: bad source shape %dx%d", H, W);
    MMRAG_CHECK_ARG(out_h > 0 && out_w > 0 && ksx > 0 && ksy > 0, "resize_crop_u8: bad output shape");
    MMRAG_CHECK_ARG(0 <= y_lo && y_lo < y_hi && y_hi <= H, "resize_crop_u8: row window [%d,%d) outside 0..%d", y_lo, y_hi, H);
    hipStream_t s = (hipStream_t)stream;
    resize_rows_kernel<<<(unsigned)(y_hi - y_lo), 256, 0, s>>>(src, src_row_bytes, y_lo, bx, kx, ksx, out_w, tmp);
    MMRAG_CHECK_HIP(hipGetLastError());
    const int row_elems = out_w * 3;
    dim3 grid((unsigned)((row_elems + 255) / 256), (unsigned)out_h);
    resize_cols_kernel<<<grid, 256, 0, s>>>(tmp, y_lo, by, ky, ksy, row_elems, dst);
    MMRAG_CHECK_HIP(hipGetLastError());
    return MMRAG_OK;
}

}  // extern "C"
