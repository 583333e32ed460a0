"""CPU: libmmrag.so builds in-tree, loads, and exports every symbol include/mmrag.h declares.
No compute entry point is called here (no GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mmrag.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mmrag_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def native():
    from multimodal_rag_amd import _native, build

    build.build(verbose=False)
    _native.lib()
    return _native


def test_header_symbols_all_exported(native):
    syms = declared_symbols()
    assert "mmrag_cosine_topk" in syms and "mmrag_merge_topk_host" in syms
    lib = ctypes.CDLL(native.LIB_PATH)
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_pure_host_entry_points(native):
    assert native.lib().mmrag_abi_version() == 1
    assert native.padded_dim(384, torch.float32) == 384
    assert native.padded_dim(768, torch.float16) == 768
    assert native.padded_dim(100, torch.float16) == 128
    assert native.padded_dim(33, torch.float32) == 64
    assert native.cosine_topk_workspace_bytes(256, 1_000_000, 5) > 0
    assert native.cosine_topk_workspace_bytes(256, 1_000_000, 21) == 0


def test_host_merge_matches_oracle(native):
    from oracle import search_oracle as O

    g = np.random.default_rng(0)
    s = -np.sort(-g.standard_normal((4, 9, 5)).astype(np.float32), axis=2)
    r = g.permutation(4 * 9 * 5).reshape(4, 9, 5).astype(np.int64)
    s[2, :, 2:] = -np.inf
    r[2, :, 2:] = -1
    es, er = O.merge_topk(s, r, 5)
    hs, hr = native.merge_topk_host(torch.from_numpy(s), torch.from_numpy(r), 5)
    assert np.array_equal(hs.numpy(), es) and np.array_equal(hr.numpy(), er)


def test_no_cpu_fallback(native):
    """the product path refuses host tensors instead of computing on the CPU"""
    q = torch.zeros((2, 384))
    with pytest.raises(native.MMRagNativeError):
        native.cosine_topk(q, q, 2, 384, 1)


def test_packed_host_merge_matches_oracle(native):
    from oracle import search_oracle as O

    g = np.random.default_rng(1)
    G, B, k = 3, 7, 5                      # B*k odd: exercises the 8-byte block padding
    s = -np.sort(-g.standard_normal((G, B, k)).astype(np.float32), axis=2)
    r = g.permutation(G * B * k).reshape(G, B, k).astype(np.int64)
    bb = native.packed_block_bytes(B, k)
    buf = np.zeros(G * bb, np.uint8)
    for i in range(G):
        buf[i * bb: i * bb + B * k * 8] = r[i].reshape(-1).view(np.uint8)
        buf[i * bb + B * k * 8: i * bb + B * k * 12] = s[i].reshape(-1).view(np.uint8)
    es, er = O.merge_topk(s, r, k)
    hs, hr = native.merge_topk_host_packed(torch.from_numpy(buf), G, B, k, k)
    assert np.array_equal(hs.numpy(), es) and np.array_equal(hr.numpy(), er)
