"""TEST / MEASUREMENT INFRASTRUCTURE ONLY.  ctypes wrapper of oracle/hnsw_oracle.cpp (the HNSW restatement of the
approximate index chromadb queries; see that file's header).  Compiled on first use with g++ on the machine it
runs on (`-O3 -mavx2 -mfma`), into oracle/_build/ (git-ignored).  Used by bench.py's optional `--hnsw-baseline`
and by tests/test_oracle_hnsw.py; never by the product path."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libhnsw_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "hnsw_oracle.cpp")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            os.makedirs(os.path.dirname(_SO), exist_ok=True)
            subprocess.run(["g++", "-O3", "-mavx2", "-mfma", "-std=c++17", "-shared", "-fPIC", "-pthread", src, "-o", _SO],
                           check=True)
        _lib = ctypes.CDLL(_SO)
        _lib.hnsw_build.restype = ctypes.c_void_p
        _lib.hnsw_build.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_uint64, ctypes.c_int]
        _lib.hnsw_search.restype = None
        _lib.hnsw_search.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        _lib.hnsw_free.restype = None
        _lib.hnsw_free.argtypes = [ctypes.c_void_p]
    return _lib


class HnswIndex:
    """Chroma's defaults: M=16, ef_construction=100, search ef = max(10, k); cosine space on unit-norm rows."""

    def __init__(self, vectors: np.ndarray, M: int = 16, ef_construction: int = 100, seed: int = 100, n_threads: int = 0):
        self.x = np.ascontiguousarray(vectors, dtype=np.float32)      # borrowed by the C side: keep alive
        self.n_threads = n_threads or min(64, os.cpu_count() or 1)
        n, d = self.x.shape
        self._h = lib().hnsw_build(self.x.ctypes.data, n, d, M, ef_construction, seed, self.n_threads)

    def search(self, queries: np.ndarray, k: int = 5, ef: int = 10, n_threads: int = 0):
        q = np.ascontiguousarray(queries, dtype=np.float32)
        rows = np.empty((q.shape[0], k), np.int64)
        scores = np.empty((q.shape[0], k), np.float32)
        lib().hnsw_search(self._h, q.ctypes.data, q.shape[0], k, max(ef, k), rows.ctypes.data, scores.ctypes.data,
                          n_threads or self.n_threads)
        return scores, rows

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.hnsw_free(h)
