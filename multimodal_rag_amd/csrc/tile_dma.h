// Shared LDS-DMA tile machinery for the gfx950 kernels of libmmrag.so.
//
// A "slab tile" is ROWS x 128 bytes in LDS: one 128-byte K-slab (64 fp16 / 32 fp32) of ROWS
// matrix rows.  It is filled by `buffer_load_dwordx4 ... lds` (1 KiB = 8 rows per wave
// instruction, lane i -> LDS byte i*16), with the 16-byte chunk index XOR-swizzled by
// (row >> 1) & 7 on the *source* address; MFMA fragment reads (ds_read_b128, lane = row) apply
// the same XOR and are bank-conflict free.  Out-of-range rows read as zero through the buffer
// descriptor's bounds check, so ragged edges need no branches.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mmrag_impl {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;

constexpr int SLAB = 128;  // bytes of K per row per ring stage

template <int N>
__device__ inline void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// wait until at most `items` later ring items (LOADS wave-instructions each) are outstanding
template <int LOADS, int MAXITEMS>
__device__ inline void wait_items(int items) {
    if constexpr (MAXITEMS >= 3) {
        if (items >= 3) {
            wait_vmcnt<3 * LOADS>();
            return;
        }
    }
    if constexpr (MAXITEMS >= 2) {
        if (items >= 2) {
            wait_vmcnt<2 * LOADS>();
            return;
        }
    }
    if constexpr (MAXITEMS >= 1) {
        if (items >= 1) {
            wait_vmcnt<LOADS>();
            return;
        }
    }
    wait_vmcnt<0>();
}

__device__ inline __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}

// per-lane source byte offset of DMA piece `piece` (8 rows) inside a tile whose rows are
// `row_bytes` apart: row * row_bytes + swizzled chunk * 16   (add the K-slab offset at issue)
__device__ inline unsigned dma_src_offset(int piece, int lane, unsigned row_bytes) {
    const int row = piece * 8 + (lane >> 3);
    return (unsigned)row * row_bytes + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
}

__device__ inline void dma_piece(__amdgpu_buffer_rsrc_t rsrc, char *lds_tile, int piece, unsigned src_off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds_tile + piece * 1024), 16, src_off, 0, 0, 0);
}

// byte offset inside a slab tile of the 16-byte chunk `chunk` (0..7) of row `row`
__device__ inline int frag_offset(int row, int chunk) { return row * SLAB + ((chunk ^ ((row >> 1) & 7)) * 16); }


}  // namespace mmrag_impl
