// Internal helpers shared by the libmmrag.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/mmrag.h"

namespace mmrag {

void set_error(const char *fmt, ...);

inline int esize(int dtype) { return dtype == MMRAG_F32 ? 4 : 2; }

#define MMRAG_CHECK_ARG(cond, ...)            \
    do {                                      \
        if (!(cond)) {                        \
            mmrag::set_error(__VA_ARGS__);    \
            return MMRAG_EINVAL;              \
        }                                     \
    } while (0)

#define MMRAG_CHECK_HIP(expr)                                                     \
    do {                                                                          \
        hipError_t e_ = (expr);                                                   \
        if (e_ != hipSuccess) {                                                   \
            mmrag::set_error("%s failed: %s", #expr, hipGetErrorString(e_));      \
            return MMRAG_EHIP;                                                    \
        }                                                                         \
    } while (0)

// number of CUs of the current device (cached per device id)
int num_cus();

// developer switches (set through mmrag_internal_set_debug by A/B tools only)
unsigned debug_flags();
constexpr unsigned DBG_LINEAR_PLAIN = 1u, DBG_LINEAR_NO_SMALL = 2u, DBG_LINEAR_NO_PERSIST = 4u;
// timing-only ablations of the persistent linear kernel (results are wrong): stores dropped by the buffer unit, no epilogue at all
constexpr unsigned DBG_LINEAR_DROP_STORES = 8u, DBG_LINEAR_SKIP_EPILOGUE = 16u;
constexpr unsigned DBG_LINEAR_X_SAME = 128u;
constexpr unsigned DBG_LINEAR_MFMA32 = 8192u;   // persistent GEMM on v_mfma_f32_32x32x16_f16 instead of 16x16x32 (A/B of the MFMA shape)
constexpr unsigned DBG_LINEAR_SMALL32 = 256u;
constexpr unsigned DBG_ENCODER_LN_PASSES = 512u;
constexpr unsigned DBG_ATTENTION_STREAMED = 4096u;   // attention: the double-buffered key-tile loop also for short sequences (A/B)
constexpr unsigned DBG_LINEAR_TILE64 = 1024u, DBG_LINEAR_TILE128 = 2048u;   // mid-size M: force 64x64 / 128x128 tiles (A/B)   // BERT single-query forward with LayerNorm launches (A/B, tests)   // M <= 64: the 32-feature workgroups for every K (A/B)   // timing only: every tile reads the first token tile

}  // namespace mmrag
