"""ctypes binding of libmmrag.so (include/mmrag.h) -- the only way the package computes.

There is no CPU / PyTorch-eager fallback: if the shared library is missing or a call
returns a non-zero status this module raises.  torch is used for device memory and the
current HIP stream only.
"""
from __future__ import annotations

import ctypes
import os
import threading
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p
from typing import Optional, Tuple

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libmmrag.so")

F32, F16, BF16 = 0, 1, 2
MAX_K = 20
_TORCH2DT = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}
_DT2TORCH = {v: k for k, v in _TORCH2DT.items()}

_lib = None
_lock = threading.Lock()


class MMRagNativeError(RuntimeError):
    pass


def _declare(lib):
    lib.mmrag_abi_version.restype = c_int
    lib.mmrag_last_error.restype = c_char_p
    lib.mmrag_padded_dim.restype = c_int64
    lib.mmrag_padded_dim.argtypes = [c_int, c_int]
    lib.mmrag_cosine_topk_workspace_bytes.restype = c_size_t
    lib.mmrag_cosine_topk_workspace_bytes.argtypes = [c_int, c_int64, c_int]
    lib.mmrag_cosine_topk.restype = c_int
    lib.mmrag_cosine_topk.argtypes = [c_void_p, c_void_p, c_int, c_int64, c_int, c_int64, c_int, c_int,
                                      c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.mmrag_cosine_topk_lists.restype = c_int
    lib.mmrag_cosine_topk_lists.argtypes = [c_void_p, c_void_p, c_int, c_int64, c_int, c_int64, c_int, c_int,
                                            c_void_p, c_void_p, c_size_t, c_void_p]
    # debug form of mmrag_cosine_topk_lists (kernel-shape switches; csrc/search.hip, not in include/mmrag.h)
    lib.mmrag_internal_search_uses_qs.restype = c_int
    lib.mmrag_internal_search_uses_qs.argtypes = [c_int, c_int64, c_int64, c_int, c_int]
    lib.mmrag_internal_cosine_topk_lists_ex.restype = c_int
    lib.mmrag_internal_cosine_topk_lists_ex.argtypes = [c_void_p, c_void_p, c_int, c_int64, c_int, c_int64, c_int,
                                                        c_int, c_void_p, c_void_p, c_size_t, c_void_p, ctypes.c_uint]
    lib.mmrag_cosine_topk_select.restype = c_int
    lib.mmrag_cosine_topk_select.argtypes = [c_int, c_int64, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.mmrag_merge_topk.restype = c_int
    lib.mmrag_merge_topk.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]
    lib.mmrag_merge_topk_host.restype = c_int
    lib.mmrag_merge_topk_host.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.mmrag_merge_topk_host_packed.restype = c_int
    lib.mmrag_merge_topk_host_packed.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.mmrag_append_rows.restype = c_int
    lib.mmrag_append_rows.argtypes = [c_void_p, c_int64, c_int64, c_int, c_int64, c_void_p, c_int64, c_int, c_void_p]
    lib.mmrag_gather_rows.restype = c_int
    lib.mmrag_gather_rows.argtypes = [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p]
    lib.mmrag_fetch_rows_f32.restype = c_int
    lib.mmrag_fetch_rows_f32.argtypes = [c_void_p, c_int64, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p]
    lib.mmrag_device_info.restype = c_int
    lib.mmrag_device_info.argtypes = [ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int64)]
    lib.mmrag_bench_stream_copy.restype = c_int
    lib.mmrag_bench_stream_copy.argtypes = [c_void_p, c_void_p, c_int64, c_void_p]
    lib.mmrag_bench_mfma_f16.restype = c_int
    lib.mmrag_bench_mfma_f16.argtypes = [c_void_p, c_void_p, c_int, ctypes.POINTER(c_int64), c_void_p]
    lib.mmrag_bench_mfma_f16_16x16x32.restype = c_int
    lib.mmrag_bench_mfma_f16_16x16x32.argtypes = [c_void_p, c_void_p, c_int, ctypes.POINTER(c_int64), c_void_p]
    lib.mmrag_bench_stream_read.restype = c_int
    lib.mmrag_bench_stream_read.argtypes = [c_void_p, c_int64, c_void_p, c_void_p]
    lib.mmrag_bench_stream_write.restype = c_int
    lib.mmrag_bench_stream_write.argtypes = [c_void_p, c_int64, c_void_p]
    lib.mmrag_copy_to_host_async.restype = c_int
    lib.mmrag_copy_to_host_async.argtypes = [c_void_p, c_void_p, c_size_t, c_void_p]
    lib.mmrag_encoder_workspace_bytes.restype = c_size_t
    lib.mmrag_encoder_workspace_bytes.argtypes = [c_void_p, c_int64, c_int]
    lib.mmrag_encoder_forward.restype = c_int
    lib.mmrag_encoder_forward.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                          c_int, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.mmrag_encoder_f32_workspace_bytes.restype = c_size_t
    lib.mmrag_encoder_f32_workspace_bytes.argtypes = [c_void_p, c_int64, c_int]
    lib.mmrag_encoder_forward_f32.restype = c_int
    lib.mmrag_encoder_forward_f32.argtypes = lib.mmrag_encoder_forward.argtypes
    lib.mmrag_linear_f32.restype = c_int
    lib.mmrag_linear_f32.argtypes = [c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]
    lib.mmrag_vit_forward.restype = c_int
    lib.mmrag_vit_forward.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t,
                                      c_void_p]
    lib.mmrag_wordpiece_create.restype = c_void_p
    lib.mmrag_wordpiece_create.argtypes = [c_void_p, c_void_p, c_int, c_int]
    lib.mmrag_wordpiece_destroy.restype = None
    lib.mmrag_wordpiece_destroy.argtypes = [c_void_p]
    lib.mmrag_wordpiece_encode_batch.restype = c_int
    lib.mmrag_wordpiece_encode_batch.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int]
    lib.mmrag_clip_bpe_create.restype = c_void_p
    lib.mmrag_clip_bpe_create.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int]
    lib.mmrag_clip_bpe_destroy.restype = None
    lib.mmrag_clip_bpe_destroy.argtypes = [c_void_p]
    lib.mmrag_clip_bpe_encode_batch.restype = c_int
    lib.mmrag_clip_bpe_encode_batch.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int]
    lib.mmrag_resample_ksize.restype = c_int
    lib.mmrag_resample_ksize.argtypes = [c_int, c_int]
    lib.mmrag_resample_coeffs.restype = c_int
    lib.mmrag_resample_coeffs.argtypes = [c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.mmrag_resize_crop_u8.restype = c_int
    lib.mmrag_resize_crop_u8.argtypes = [c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                         c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]
    lib.mmrag_linear_f16.restype = c_int
    lib.mmrag_linear_f16.argtypes = [c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                     c_void_p]
    lib.mmrag_layernorm_f16.restype = c_int
    lib.mmrag_layernorm_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p]
    lib.mmrag_embed_ln_f16.restype = c_int
    lib.mmrag_embed_ln_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_int64, c_int, c_int, c_int, c_float, c_void_p]
    lib.mmrag_attention_f16.restype = c_int
    lib.mmrag_attention_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.mmrag_pool_normalize_f16.restype = c_int
    lib.mmrag_pool_normalize_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                             c_void_p]


def lib():
    """Load libmmrag.so once (after torch, so both share one HIP runtime)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise MMRagNativeError(
                        f"{LIB_PATH} is missing: build it with `python -m multimodal_rag_amd.build` "
                        "(hipcc, gfx950).  There is no CPU fallback.")
                handle = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
                _declare(handle)
                _lib = handle
    return _lib


def _check(status: int, what: str):
    if status != 0:
        msg = lib().mmrag_last_error().decode("utf-8", "replace")
        raise MMRagNativeError(f"{what} failed (status {status}): {msg}")


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _dev_check(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise MMRagNativeError("libmmrag entry points take device (HIP) tensors only; got a CPU tensor")


def search_uses_query_stationary(B: int, n: int, d: int, k: int, dtype: torch.dtype) -> bool:
    """which kernel a search of this shape runs on (True: a query-stationary kernel, False: the slab-ring cosine_topk_kernel)"""
    return bool(lib().mmrag_internal_search_uses_qs(B, n, padded_dim(d, dtype), _TORCH2DT[dtype], k))


def search_kernel_name(B: int, n: int, d: int, k: int, dtype: torch.dtype) -> str:
    """name of the dominant kernel of a search of this shape (for the bench's roofline label)"""
    v = lib().mmrag_internal_search_uses_qs(B, n, padded_dim(d, dtype), _TORCH2DT[dtype], k)
    return {0: "cosine_topk_kernel (slab-ring)", 1: "cosine_topk_qs_kernel", 2: "cosine_topk_walk_kernel"}[v]


def padded_dim(d: int, dtype: torch.dtype) -> int:
    v = lib().mmrag_padded_dim(int(d), _TORCH2DT[dtype])
    if v < 0:
        raise MMRagNativeError(f"padded_dim: bad arguments d={d} dtype={dtype}")
    return int(v)


# kernel-shape switches of the debug entry point (tests / A-B tools; the product always passes 0)
DBG_NO_PREPASS, DBG_8_WAVES, DBG_NO_QS, DBG_FORCE_QS = 1, 2, 4, 8
# query-stationary shapes: three-launch plan instead of the single-launch walk; the walk's MFMA shape forced; static tile
# assignment only; no in-kernel threshold seeding (A/B tools and the tests that pin every code path)
DBG_OLD_QS, DBG_MFMA32, DBG_MFMA16, DBG_NO_DYN, DBG_NO_SEED = 0x10000, 0x20000, 0x40000, 0x80000, 0x100000


def cosine_topk(q: torch.Tensor, corpus: torch.Tensor, n: int, d: int, k: int, row_offset: int = 0,
                alive_bits: Optional[torch.Tensor] = None,
                workspace: Optional[torch.Tensor] = None, dbg: int = 0,
                packed_out: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Exact cosine top-k of q [B, ld] against the first n rows of corpus [cap, ld].

    Returns (scores [B, k] float32 descending, rows [B, k] int64 global).  Both inputs must be
    contiguous, same dtype, same padded leading dimension (pad columns zero).
    """
    _dev_check(q, corpus, alive_bits)
    if q.dim() != 2 or corpus.dim() != 2 or not q.is_contiguous() or not corpus.is_contiguous():
        raise MMRagNativeError("cosine_topk: q and corpus must be contiguous 2-D tensors")
    if q.dtype != corpus.dtype or q.shape[1] != corpus.shape[1]:
        raise MMRagNativeError("cosine_topk: q and corpus must share dtype and padded width")
    if n > corpus.shape[0]:
        raise MMRagNativeError(f"cosine_topk: n={n} exceeds corpus capacity {corpus.shape[0]}")
    B, ld = q.shape
    L = lib()
    need = L.mmrag_cosine_topk_workspace_bytes(B, n, k)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(max(need, 16), dtype=torch.uint8, device=q.device)
    if packed_out:   # ONE buffer [rows B*k int64 | scores B*k float32]: a caller that wants both on the host copies once
        buf = torch.empty(B * k * 12, dtype=torch.uint8, device=q.device)
        out_r = buf[: B * k * 8].view(torch.int64).view(B, k)
        out_s = buf[B * k * 8:].view(torch.float32).view(B, k)
    else:
        out_s = torch.empty((B, k), dtype=torch.float32, device=q.device)
        out_r = torch.empty((B, k), dtype=torch.int64, device=q.device)
    if dbg:
        cosine_topk_lists(q, corpus, n, d, k, workspace, alive_bits=alive_bits, dbg=dbg)
        return cosine_topk_select(B, n, k, row_offset, workspace, out_s, out_r)
    with torch.cuda.device(q.device):
        st = L.mmrag_cosine_topk(q.data_ptr(), corpus.data_ptr(), B, n, d, ld, _TORCH2DT[q.dtype], k, row_offset,
                                 alive_bits.data_ptr() if alive_bits is not None else None,
                                 out_s.data_ptr(), out_r.data_ptr(), workspace.data_ptr(),
                                 workspace.numel() * workspace.element_size(), _stream_ptr(q.device))
    _check(st, "mmrag_cosine_topk")
    return out_s, out_r


def device_info() -> dict:
    """CU count, maximum shader clock and HBM size of the current device (mmrag_device_info)"""
    cus, mhz, mem = c_int(0), c_int(0), c_int64(0)
    _check(lib().mmrag_device_info(ctypes.byref(cus), ctypes.byref(mhz), ctypes.byref(mem)), "mmrag_device_info")
    return {"compute_units": cus.value, "max_clock_mhz": mhz.value, "hbm_bytes": mem.value}


def measure_peaks(device: torch.device, seconds: float = 0.4) -> dict:
    """Measured stream-copy bandwidth and fp16 MFMA rate of this device (the library's own micro-kernels,
    each held for `seconds` so the chip reaches the clock it sustains): the second set of roofline peaks."""
    L = lib()
    out = {}
    with torch.cuda.device(device):
        stream = _stream_ptr(device)
        nbytes = 1 << 30
        src = torch.empty(nbytes, dtype=torch.uint8, device=device)
        dst = torch.empty(nbytes, dtype=torch.uint8, device=device)
        src.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def timed(fn, min_s):
            import time
            fn(); torch.cuda.synchronize(device)
            t_end = time.time() + min_s / 2
            while time.time() < t_end:   # reach the sustained clock first
                for _ in range(4): fn()
                torch.cuda.synchronize(device)
            n = 0
            e0.record()
            t_end = time.time() + min_s / 2
            while time.time() < t_end:
                for _ in range(4): fn()
                n += 4
                torch.cuda.synchronize(device)
            e1.record(); torch.cuda.synchronize(device)
            return e0.elapsed_time(e1) * 1e-3 / n

        t = timed(lambda: _check(L.mmrag_bench_stream_copy(dst.data_ptr(), src.data_ptr(), nbytes, stream), "copy"), seconds)
        out["stream_copy_GBps"] = round(2 * nbytes / t / 1e9, 1)      # bytes read + bytes written
        sink = torch.empty(256 * 8 * 256, dtype=torch.float32, device=device)
        t = timed(lambda: _check(L.mmrag_bench_stream_read(src.data_ptr(), nbytes, sink.data_ptr(), stream), "read"), seconds)
        out["stream_read_GBps"] = round(nbytes / t / 1e9, 1)           # a read-only stream: what a corpus scan is
        t = timed(lambda: _check(L.mmrag_bench_stream_write(dst.data_ptr(), nbytes, stream), "write"), seconds)
        out["stream_write_GBps"] = round(nbytes / t / 1e9, 1)
        del src, dst, sink
        seed = (torch.randn(256 * 8, device=device) * 0.5).to(torch.float16)
        res = torch.empty(256 * 1024, dtype=torch.float32, device=device)
        flops = c_int64(0)
        iters = 20000
        t = timed(lambda: _check(L.mmrag_bench_mfma_f16(seed.data_ptr(), res.data_ptr(), iters, ctypes.byref(flops), stream),
                                 "mfma"), seconds)
        out["mfma_f16_TFLOPs"] = round(flops.value / t / 1e12, 1)
        iters = 5000
        t = timed(lambda: _check(L.mmrag_bench_mfma_f16_16x16x32(seed.data_ptr(), res.data_ptr(), iters,
                                                                 ctypes.byref(flops), stream), "mfma16"), seconds)
        out["mfma_f16_16x16x32_TFLOPs"] = round(flops.value / t / 1e12, 1)
    return out


def cosine_topk_workspace_bytes(B: int, n: int, k: int) -> int:
    return int(lib().mmrag_cosine_topk_workspace_bytes(B, n, k))


def merge_topk(scores: torch.Tensor, rows: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Device merge of [G, B, k_in] shard results into [B, k]."""
    _dev_check(scores, rows)
    G, B, k_in = scores.shape
    scores = scores.contiguous()
    rows = rows.contiguous()
    out_s = torch.empty((B, k), dtype=torch.float32, device=scores.device)
    out_r = torch.empty((B, k), dtype=torch.int64, device=scores.device)
    with torch.cuda.device(scores.device):
        st = lib().mmrag_merge_topk(scores.data_ptr(), rows.data_ptr(), G, B, k_in, k, out_s.data_ptr(),
                                    out_r.data_ptr(), _stream_ptr(scores.device))
    _check(st, "mmrag_merge_topk")
    return out_s, out_r


def merge_topk_host(scores: torch.Tensor, rows: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Host (C++) merge of [G, B, k_in] shard results held in CPU tensors into [B, k]."""
    if scores.is_cuda or rows.is_cuda:
        raise MMRagNativeError("merge_topk_host takes CPU tensors")
    G, B, k_in = scores.shape
    scores = scores.contiguous().to(torch.float32)
    rows = rows.contiguous().to(torch.int64)
    out_s = torch.empty((B, k), dtype=torch.float32)
    out_r = torch.empty((B, k), dtype=torch.int64)
    st = lib().mmrag_merge_topk_host(scores.data_ptr(), rows.data_ptr(), G, B, k_in, k, out_s.data_ptr(),
                                     out_r.data_ptr())
    _check(st, "mmrag_merge_topk_host")
    return out_s, out_r


def packed_block_bytes(B: int, k: int) -> int:
    return (B * k * 12 + 7) // 8 * 8


def merge_topk_host_packed(blocks: torch.Tensor, G: int, B: int, k_in: int, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Host merge of G packed rank blocks [rows B*k_in i64 | scores B*k_in f32 | pad to 8 B] (CPU uint8)."""
    if blocks.is_cuda or blocks.dtype != torch.uint8 or blocks.numel() != G * packed_block_bytes(B, k_in):
        raise MMRagNativeError("merge_topk_host_packed takes a CPU uint8 tensor of G * packed_block_bytes(B, k_in) bytes")
    out_s = torch.empty((B, k), dtype=torch.float32)
    out_r = torch.empty((B, k), dtype=torch.int64)
    st = lib().mmrag_merge_topk_host_packed(blocks.data_ptr(), G, B, k_in, k, out_s.data_ptr(), out_r.data_ptr())
    _check(st, "mmrag_merge_topk_host_packed")
    return out_s, out_r


def append_rows(corpus: torch.Tensor, n_used: int, new_rows: torch.Tensor, d: int) -> None:
    """corpus[n_used : n_used+m, :d] = cast(new_rows) with zero pad columns."""
    _dev_check(corpus, new_rows)
    new_rows = new_rows.to(torch.float32).contiguous()
    m = new_rows.shape[0]
    with torch.cuda.device(corpus.device):
        st = lib().mmrag_append_rows(corpus.data_ptr(), corpus.shape[0], corpus.shape[1], _TORCH2DT[corpus.dtype],
                                     n_used, new_rows.data_ptr(), m, d, _stream_ptr(corpus.device))
    _check(st, "mmrag_append_rows")


def gather_rows(dst: torch.Tensor, src: torch.Tensor, keep_rows: torch.Tensor) -> None:
    _dev_check(dst, src, keep_rows)
    keep_rows = keep_rows.to(torch.int64).contiguous()
    with torch.cuda.device(src.device):
        st = lib().mmrag_gather_rows(dst.data_ptr(), src.data_ptr(), src.shape[1], _TORCH2DT[src.dtype],
                                     keep_rows.data_ptr(), keep_rows.numel(), _stream_ptr(src.device))
    _check(st, "mmrag_gather_rows")


def fetch_rows_f32(corpus: torch.Tensor, rows: torch.Tensor, d: int) -> torch.Tensor:
    _dev_check(corpus, rows)
    rows = rows.to(torch.int64).contiguous()
    out = torch.empty((rows.numel(), d), dtype=torch.float32, device=corpus.device)
    with torch.cuda.device(corpus.device):
        st = lib().mmrag_fetch_rows_f32(corpus.data_ptr(), corpus.shape[1], _TORCH2DT[corpus.dtype], rows.data_ptr(),
                                        rows.numel(), d, out.data_ptr(), _stream_ptr(corpus.device))
    _check(st, "mmrag_fetch_rows_f32")
    return out


def cosine_topk_lists(q: torch.Tensor, corpus: torch.Tensor, n: int, d: int, k: int, workspace: torch.Tensor,
                      alive_bits: Optional[torch.Tensor] = None, dbg: int = 0) -> None:
    """Phase 1 of cosine_topk: the fused GEMM + selection kernel; candidates stay in `workspace`."""
    _dev_check(q, corpus, workspace, alive_bits)
    if q.dtype != corpus.dtype or q.shape[1] != corpus.shape[1] or not q.is_contiguous() or not corpus.is_contiguous():
        raise MMRagNativeError("cosine_topk_lists: q and corpus must be contiguous and share dtype and padded width")
    if n > corpus.shape[0]:
        raise MMRagNativeError(f"cosine_topk_lists: n={n} exceeds corpus capacity {corpus.shape[0]}")
    B, ld = q.shape
    with torch.cuda.device(q.device):
        st = lib().mmrag_internal_cosine_topk_lists_ex(
            q.data_ptr(), corpus.data_ptr(), B, n, d, ld, _TORCH2DT[q.dtype], k,
            alive_bits.data_ptr() if alive_bits is not None else None, workspace.data_ptr(),
            workspace.numel() * workspace.element_size(), _stream_ptr(q.device), int(dbg))
    _check(st, "mmrag_cosine_topk_lists")


def cosine_topk_select(B: int, n: int, k: int, row_offset: int, workspace: torch.Tensor,
                       out_scores: Optional[torch.Tensor] = None,
                       out_rows: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Phase 2 of cosine_topk: merge the candidate lists in `workspace` into [B, k]."""
    _dev_check(workspace)
    dev = workspace.device
    if out_scores is None:
        out_scores = torch.empty((B, k), dtype=torch.float32, device=dev)
    if out_rows is None:
        out_rows = torch.empty((B, k), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        st = lib().mmrag_cosine_topk_select(B, n, k, row_offset, workspace.data_ptr(), out_scores.data_ptr(),
                                            out_rows.data_ptr(), _stream_ptr(dev))
    _check(st, "mmrag_cosine_topk_select")
    return out_scores, out_rows


# ----------------------------------------------------------------------------------------------
# encoder building blocks (fp16 tensors on the device)
# ----------------------------------------------------------------------------------------------
ACT_NONE, ACT_GELU, ACT_QUICK_GELU = 0, 1, 2
ARCH_BERT, ARCH_PRELN = 0, 1
POOL_MEAN, POOL_FIRST, POOL_SELECT = 0, 1, 2


class EncoderDesc(ctypes.Structure):
    """mirror of `mmrag_encoder_desc` (include/mmrag.h)"""
    _fields_ = [("arch", c_int32), ("n_layers", c_int32), ("hidden", c_int32), ("n_heads", c_int32),
                ("intermediate", c_int32), ("vocab", c_int32), ("max_pos", c_int32), ("pool", c_int32),
                ("act", c_int32), ("causal", c_int32), ("normalize", c_int32), ("out_dim", c_int32),
                ("ln_eps", c_float), ("image", c_int32), ("patch", c_int32)]


def _ptr(t: Optional[torch.Tensor]):
    return t.data_ptr() if t is not None else None


def linear_f16(x: torch.Tensor, wt: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE,
               resid: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M, N] = act(x[M, K] @ wt[N, K].T + bias) (+ resid)"""
    _dev_check(x, wt, bias, resid, out)
    M, K = x.shape
    Nf = wt.shape[0]
    if x.dtype != torch.float16 or wt.dtype != torch.float16 or wt.shape[1] != K:
        raise MMRagNativeError("linear_f16: x [M,K] and wt [N,K] must be fp16 with matching K")
    if out is None:
        out = torch.empty((M, Nf), dtype=torch.float16, device=x.device)
    with torch.cuda.device(x.device):
        st = lib().mmrag_linear_f16(x.data_ptr(), M, K, wt.data_ptr(), Nf, _ptr(bias), act, _ptr(resid),
                                    out.data_ptr(), _stream_ptr(x.device))
    _check(st, "mmrag_linear_f16")
    return out


def layernorm_f16(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float) -> torch.Tensor:
    _dev_check(x, gamma, beta)
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        st = lib().mmrag_layernorm_f16(x.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(), x.shape[0],
                                       x.shape[1], eps, _stream_ptr(x.device))
    _check(st, "mmrag_layernorm_f16")
    return out


def embed_ln_f16(ids, pos_ids, tok, pos, type0, gamma, beta, eps: float) -> torch.Tensor:
    _dev_check(ids, pos_ids, tok, pos, type0, gamma, beta)
    T, H = ids.numel(), tok.shape[1]
    out = torch.empty((T, H), dtype=torch.float16, device=tok.device)
    with torch.cuda.device(tok.device):
        st = lib().mmrag_embed_ln_f16(ids.data_ptr(), pos_ids.data_ptr(), tok.data_ptr(), pos.data_ptr(), _ptr(type0),
                                      _ptr(gamma), _ptr(beta), out.data_ptr(), T, H, tok.shape[0], pos.shape[0], eps,
                                      _stream_ptr(tok.device))
    _check(st, "mmrag_embed_ln_f16")
    return out


def attention_f16(qkv: torch.Tensor, cu_seqlens: torch.Tensor, max_len: int, n_heads: int,
                  causal: bool = False) -> torch.Tensor:
    _dev_check(qkv, cu_seqlens)
    T, H3 = qkv.shape
    H = H3 // 3
    ctx = torch.empty((T, H), dtype=torch.float16, device=qkv.device)
    with torch.cuda.device(qkv.device):
        st = lib().mmrag_attention_f16(qkv.data_ptr(), cu_seqlens.data_ptr(), ctx.data_ptr(), cu_seqlens.numel() - 1,
                                       max_len, H, n_heads, int(causal), _stream_ptr(qkv.device))
    _check(st, "mmrag_attention_f16")
    return ctx


def pool_normalize_f16(x: torch.Tensor, cu_seqlens: torch.Tensor, pool: int, normalize: bool = True,
                       sel: Optional[torch.Tensor] = None) -> torch.Tensor:
    _dev_check(x, cu_seqlens, sel)
    B, H = cu_seqlens.numel() - 1, x.shape[1]
    out = torch.empty((B, H), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        st = lib().mmrag_pool_normalize_f16(x.data_ptr(), cu_seqlens.data_ptr(), _ptr(sel), out.data_ptr(), B, H, pool,
                                            int(normalize), _stream_ptr(x.device))
    _check(st, "mmrag_pool_normalize_f16")
    return out


def encoder_workspace_bytes(desc: EncoderDesc, T: int, B: int, f32: bool = False) -> int:
    if f32:
        return int(lib().mmrag_encoder_f32_workspace_bytes(ctypes.byref(desc), T, B))
    return int(lib().mmrag_encoder_workspace_bytes(ctypes.byref(desc), T, B))


def linear_f32(x: torch.Tensor, wt: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE,
               resid: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = act(x . wt^T + bias) (+ resid), float32 on the exact float32 matrix instruction (the fp32 encoder's GEMM)"""
    _dev_check(x, wt, bias, resid)
    M, K = x.shape
    N = wt.shape[0]
    out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        st = lib().mmrag_linear_f32(x.data_ptr(), M, K, wt.data_ptr(), N, _ptr(bias), act, _ptr(resid), out.data_ptr(),
                                    _stream_ptr(x.device))
    _check(st, "mmrag_linear_f32")
    return out


def encoder_forward(desc: EncoderDesc, weight_ptrs, ids: torch.Tensor, pos_ids: torch.Tensor,
                    cu_seqlens: torch.Tensor, max_len: int, sel: Optional[torch.Tensor] = None,
                    workspace: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                    f32: bool = False) -> torch.Tensor:
    """One encoder pass over packed token ids -> [B, out_dim] float32.  `weight_ptrs` is a ctypes
    array of c_void_p in the order include/mmrag.h documents (see encoder.DeviceEncoder).  `f32`: the float32 mode
    (mmrag_encoder_forward_f32; every weight float32)."""
    _dev_check(ids, pos_ids, cu_seqlens, sel, workspace, out)
    T, B = ids.numel(), cu_seqlens.numel() - 1
    need = encoder_workspace_bytes(desc, T, B, f32)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=ids.device)
    if out is None:
        out = torch.empty((B, desc.out_dim), dtype=torch.float32, device=ids.device)
    fn = lib().mmrag_encoder_forward_f32 if f32 else lib().mmrag_encoder_forward
    with torch.cuda.device(ids.device):
        st = fn(ctypes.byref(desc), weight_ptrs, ids.data_ptr(), pos_ids.data_ptr(),
                                         cu_seqlens.data_ptr(), _ptr(sel), T, B, max_len, out.data_ptr(),
                                         workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                                         _stream_ptr(ids.device))
    _check(st, "mmrag_encoder_forward")
    return out


PIXELS_F16_CHW, PIXELS_U8_HWC = 0, 1


def vit_forward(desc: EncoderDesc, weight_ptrs, pixels: torch.Tensor, pixel_kind: int, cu_seqlens: torch.Tensor,
                workspace: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """CLIP-style vision tower: images -> [B, out_dim] float32 L2-normalised."""
    _dev_check(pixels, cu_seqlens, workspace, out)
    B = pixels.shape[0]
    S = (desc.image // desc.patch) ** 2 + 1
    need = encoder_workspace_bytes(desc, B * S, B)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=pixels.device)
    if out is None:
        out = torch.empty((B, desc.out_dim), dtype=torch.float32, device=pixels.device)
    with torch.cuda.device(pixels.device):
        st = lib().mmrag_vit_forward(ctypes.byref(desc), weight_ptrs, pixels.data_ptr(), pixel_kind,
                                     cu_seqlens.data_ptr(), B, out.data_ptr(), workspace.data_ptr(),
                                     workspace.numel() * workspace.element_size(), _stream_ptr(pixels.device))
    _check(st, "mmrag_vit_forward")
    return out


def resample_coeffs(in_size: int, out_size: int, first: int, count: int):
    """HOST: Pillow's fixed-point bicubic taps of output indices [first, first+count).  Returns numpy
    (bounds [count,2] int32, taps [count,ksize] int32).  No GPU needed."""
    import numpy as np

    ks = lib().mmrag_resample_ksize(in_size, out_size)
    bounds = np.zeros((count, 2), np.int32)
    taps = np.zeros((count, max(ks, 1)), np.int32)
    _check(lib().mmrag_resample_coeffs(in_size, out_size, first, count, bounds.ctypes.data, taps.ctypes.data),
           "mmrag_resample_coeffs")
    return bounds, taps


def resize_crop_u8(src: torch.Tensor, bx: torch.Tensor, kx: torch.Tensor, by: torch.Tensor, ky: torch.Tensor,
                   y_lo: int, y_hi: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """src [H,W,3] uint8 (device) -> [len(by), len(bx), 3] uint8: horizontal taps (bx,kx) per output column,
    then vertical taps (by,ky) per output row; bit-exact with Pillow's 8-bit bicubic resize."""
    _dev_check(src, bx, kx, by, ky, out)
    if src.dtype != torch.uint8 or src.dim() != 3 or src.shape[2] != 3 or src.stride(2) != 1 or src.stride(1) != 3:
        raise MMRagNativeError("resize_crop_u8: src must be [H, W, 3] uint8 with packed pixels")
    H, W = int(src.shape[0]), int(src.shape[1])
    out_h, out_w = int(by.shape[0]), int(bx.shape[0])
    if out is None:
        out = torch.empty((out_h, out_w, 3), dtype=torch.uint8, device=src.device)
    tmp = torch.empty(((y_hi - y_lo) * out_w * 3,), dtype=torch.uint8, device=src.device)
    with torch.cuda.device(src.device):
        st = lib().mmrag_resize_crop_u8(src.data_ptr(), H, W, src.stride(0), bx.data_ptr(), kx.data_ptr(),
                                        int(kx.shape[1]), by.data_ptr(), ky.data_ptr(), int(ky.shape[1]), out_h, out_w,
                                        y_lo, y_hi, tmp.data_ptr(), out.data_ptr(), _stream_ptr(src.device))
    _check(st, "mmrag_resize_crop_u8")
    return out


def copy_to_host_async(dst_host: torch.Tensor, src_dev: torch.Tensor, stream: int) -> None:
    """Stream-ordered device -> pinned host copy on a raw hipStream_t (no torch stream context)."""
    nbytes = src_dev.numel() * src_dev.element_size()
    if dst_host.is_cuda or not src_dev.is_cuda or dst_host.numel() * dst_host.element_size() < nbytes:
        raise MMRagNativeError("copy_to_host_async: need a host destination at least as large as the device source")
    _check(lib().mmrag_copy_to_host_async(dst_host.data_ptr(), src_dev.data_ptr(), nbytes, stream),
           "mmrag_copy_to_host_async")


class SearchPlan:
    """Pre-validated, pointer-cached form of cosine_topk_lists / cosine_topk_select for hot loops
    (a serving loop or bench.py issues the same shapes thousands of times; argument checking and
    torch context managers would otherwise dominate the host time per batch at small shard sizes).
    The tensors are kept alive by the plan; `stream` arguments are raw hipStream_t values."""

    def __init__(self, q: torch.Tensor, corpus: torch.Tensor, n: int, d: int, k: int, workspace: torch.Tensor,
                 alive_bits: Optional[torch.Tensor] = None):
        _dev_check(q, corpus, workspace, alive_bits)
        if q.dtype != corpus.dtype or q.shape[1] != corpus.shape[1] or not q.is_contiguous() or not corpus.is_contiguous():
            raise MMRagNativeError("SearchPlan: q and corpus must be contiguous and share dtype and padded width")
        if n > corpus.shape[0]:
            raise MMRagNativeError(f"SearchPlan: n={n} exceeds corpus capacity {corpus.shape[0]}")
        need = lib().mmrag_cosine_topk_workspace_bytes(q.shape[0], n, k)
        if workspace.numel() * workspace.element_size() < need:
            raise MMRagNativeError("SearchPlan: workspace too small")
        self._keep = (q, corpus, workspace, alive_bits)
        self.B, self.n, self.k = q.shape[0], n, k
        self._scan_args = (q.data_ptr(), corpus.data_ptr(), q.shape[0], n, d, q.shape[1], _TORCH2DT[q.dtype], k,
                           alive_bits.data_ptr() if alive_bits is not None else None, workspace.data_ptr(),
                           workspace.numel() * workspace.element_size())
        self._ws = workspace.data_ptr()
        self._scan = lib().mmrag_cosine_topk_lists
        self._select = lib().mmrag_cosine_topk_select

    def scan(self, stream: int) -> None:
        st = self._scan(*self._scan_args, stream)
        if st:
            _check(st, "mmrag_cosine_topk_lists")

    def select(self, row_offset: int, out_scores_ptr: int, out_rows_ptr: int, stream: int) -> None:
        st = self._select(self.B, self.n, self.k, row_offset, self._ws, out_scores_ptr, out_rows_ptr, stream)
        if st:
            _check(st, "mmrag_cosine_topk_select")
