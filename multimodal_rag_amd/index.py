"""Device-resident vector collection: what `chromadb`'s collection is to the reference.

Replaces the engine calls made by app/utils/embedder.py:
    collection.add(embeddings, documents, metadatas, ids)          :518
    collection.query(query_embeddings, n_results, where, include)  :596-601, :900-905
    collection.get(where|ids, include)                             :632-635, :887-891
    collection.delete(ids)                                         :639-642
    collection.count()                                             :700
The vectors live in one [capacity, ld] matrix in HBM (fp16 by default, rows padded to the
kernel's 128-byte slabs); ids / documents / metadata stay in host tables indexed by row, as
SURVEY.md section 8b "Ownership" lays out.  Search is exact (fused MFMA GEMM + top-k in
libmmrag.so), distance = 1 - cos (the committed collection's hnsw:space=cosine).

No vector arithmetic happens in this file: torch provides device buffers and copies only.
"""
from __future__ import annotations

import gc
import logging
import operator
import threading
from typing import Any, Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _native
from .hostutil import load_hostrows
from .tracing import stage

_HOSTROWS = load_hostrows()

logger = logging.getLogger(__name__)


# --------------------------------------------------------------------------------------------
# `where` filters (the subset of Chroma's grammar that is expressible on flat metadata)
# --------------------------------------------------------------------------------------------
_CMP: Dict[str, Callable[[Any, Any], bool]] = {
    "$eq": lambda a, b: a == b,
    "$ne": lambda a, b: a != b,
    "$gt": lambda a, b: a is not None and a > b,
    "$gte": lambda a, b: a is not None and a >= b,
    "$lt": lambda a, b: a is not None and a < b,
    "$lte": lambda a, b: a is not None and a <= b,
    "$in": lambda a, b: a in b,
    "$nin": lambda a, b: a not in b,
}


def match_where(meta: Dict[str, Any], where: Optional[Dict[str, Any]]) -> bool:
    if not where:
        return True
    for key, cond in where.items():
        if key == "$and":
            if not all(match_where(meta, w) for w in cond):
                return False
        elif key == "$or":
            if not any(match_where(meta, w) for w in cond):
                return False
        elif isinstance(cond, dict):
            for op, val in cond.items():
                if op not in _CMP:
                    raise ValueError(f"unsupported where operator {op!r}")
                if not _CMP[op](meta.get(key), val):
                    return False
        else:
            if meta.get(key) != cond:
                return False
    return True


class MetaIndex:
    """Inverted index over the rows' metadata: (key, value) -> ascending row list.  Answers the `where`
    forms the service actually sends -- equality, `$eq`, `$in`, `$and`, `$or` over hashable scalars -- in time
    proportional to the matches instead of one Python predicate call per stored row (a filtered query or a
    `delete_document` on a 1M-row shard was a second of host time); anything else returns None and the caller
    falls back to the full scan.  Rows are only ever appended; after a compaction the owner rebuilds it."""

    def __init__(self):
        self.n = 0
        self._kv: Dict[Any, Dict[Any, List[int]]] = {}

    def append(self, metadatas: Sequence[Dict[str, Any]]):
        for meta in metadatas:
            for k, v in meta.items():
                try:
                    self._kv.setdefault(k, {}).setdefault(v, []).append(self.n)
                except TypeError:        # unhashable value: only the full scan can match it
                    self._kv.setdefault(k, {}).setdefault(_UNHASHABLE, []).append(self.n)
            self.n += 1

    def _eq(self, key, val) -> Optional[np.ndarray]:
        by_val = self._kv.get(key)
        if by_val is None:
            return np.zeros(0, np.int64) if val is not None else None   # meta.get(key) is None for every row
        if _UNHASHABLE in by_val or val is None:
            return None
        try:
            return np.asarray(by_val.get(val, ()), dtype=np.int64)
        except TypeError:
            return None

    def rows(self, where: Optional[Dict[str, Any]]) -> Optional[np.ndarray]:
        """Ascending rows matching `where`, or None when the form is not covered."""
        if not where:
            return np.arange(self.n, dtype=np.int64)
        acc: Optional[np.ndarray] = None
        for key, cond in where.items():
            if key == "$and":
                parts = [self.rows(w) for w in cond]
                if any(p is None for p in parts):
                    return None
                cur = np.arange(self.n, dtype=np.int64)
                for p in parts:
                    cur = np.intersect1d(cur, p, assume_unique=True)
            elif key == "$or":
                parts = [self.rows(w) for w in cond]
                if any(p is None for p in parts):
                    return None
                cur = np.unique(np.concatenate(parts)) if parts else np.zeros(0, np.int64)
            elif isinstance(cond, dict):
                cur = np.arange(self.n, dtype=np.int64)
                for op, val in cond.items():
                    if op == "$eq":
                        part = self._eq(key, val)
                    elif op == "$in" and isinstance(val, (list, tuple, set)):
                        sub = [self._eq(key, v) for v in val]
                        part = None if any(x is None for x in sub) else (
                            np.unique(np.concatenate(sub)) if sub else np.zeros(0, np.int64))
                    else:
                        return None
                    if part is None:
                        return None
                    cur = np.intersect1d(cur, part, assume_unique=True)
            else:
                cur = self._eq(key, cond)
                if cur is None:
                    return None
            acc = cur if acc is None else np.intersect1d(acc, cur, assume_unique=True)
        return acc


_UNHASHABLE = object()


class VectorIndex:
    """One shard of the corpus matrix on one GPU plus its host-side row tables.

    Rows are appended in insertion order and never move while they are alive, so "lower row" always means
    "earlier insert" (the tie rule).  A delete is a TOMBSTONE: the row's bit is cleared in a device-resident alive
    bitmap that the search kernels consume (masked rows start at -inf, csrc/search*.hip), its id leaves the id map,
    and nothing else is touched -- `delete_document` on a 1M-row shard is O(victims) on the host and one small
    scatter on the device.  The matrix is compacted (stably) only when more than COMPACT_DEAD_FRACTION of the rows
    are dead, or on `compact()` / save.

    Vectors must be unit-norm (cosine = inner product of unit vectors; `distance = 1 - cos`).  The encoders of this
    package normalise; `add` / `query` reject rows whose norm is off by more than 1e-2 instead of silently ranking
    by raw inner product (Chroma's cosine space would have normalised them)."""

    COMPACT_DEAD_FRACTION = 0.25
    COMPACT_MIN_DEAD = 4096

    def __init__(self, dim: int, dtype: torch.dtype = torch.float16, device: str = "cuda:0",
                 capacity: int = 4096, name: str = "multimodal_rag",
                 metadata: Optional[Dict[str, Any]] = None):
        if dtype not in (torch.float16, torch.float32, torch.bfloat16):
            raise ValueError(f"unsupported storage dtype {dtype}")
        _native.lib()  # fail loudly if the HIP library is absent
        self.name = name
        self.metadata = dict(metadata or {})
        self.dim = int(dim)
        self.dtype = dtype
        self.device = torch.device(device)
        self.ld = _native.padded_dim(self.dim, dtype)
        cap = max(int(capacity), 256)
        self._matrix = torch.zeros((cap, self.ld), dtype=dtype, device=self.device)
        self._alive_dev = torch.zeros(self._n_words(cap), dtype=torch.int32, device=self.device)
        self._alive_host = np.zeros(self._n_words(cap), dtype=np.uint32)
        self._n = 0            # rows in use (alive + dead)
        self._n_dead = 0
        self._ids: List[Optional[str]] = []
        self._documents: List[Optional[str]] = []
        self._metadatas: List[Dict[str, Any]] = []
        self._meta_index = MetaIndex()
        self._row_of: Dict[str, int] = {}
        self._lock = threading.RLock()
        self._search_ws: Optional[torch.Tensor] = None    # candidate-list workspace of the search kernels, reused
        from .config import settings

        self.f32_exact = bool(settings.MMRAG_F32_EXACT_SEARCH)   # float32 collections only (see config.py)

    @staticmethod
    def _n_words(rows: int) -> int:
        return (rows + 31) // 32 + 8   # the kernels read whole words of the last tile

    # ------------------------------------------------------------------ storage ----------
    @property
    def matrix(self) -> torch.Tensor:
        return self._matrix

    def count(self) -> int:
        return self._n - self._n_dead

    @property
    def rows_in_use(self) -> int:
        """rows of the matrix that hold a vector, dead ones included (what the kernels scan)"""
        return self._n

    def _reserve(self, rows: int):
        cap = self._matrix.shape[0]
        if rows <= cap:
            return
        new_cap = max(rows, cap * 2)
        grown = torch.empty((new_cap, self.ld), dtype=self.dtype, device=self.device)
        grown[: self._n].copy_(self._matrix[: self._n])
        grown[self._n:].zero_()
        self._matrix = grown
        words = torch.zeros(self._n_words(new_cap), dtype=torch.int32, device=self.device)
        words[: self._alive_dev.numel()].copy_(self._alive_dev)
        self._alive_dev = words
        host = np.zeros(self._n_words(new_cap), dtype=np.uint32)
        host[: self._alive_host.size] = self._alive_host
        self._alive_host = host

    def _set_alive(self, lo: int, hi: int):
        """mark rows [lo, hi) alive (appends): touch only the words they fall in"""
        idx = np.arange(lo, hi, dtype=np.int64)
        np.bitwise_or.at(self._alive_host, idx >> 5, np.uint32(1) << (idx & 31).astype(np.uint32))
        w0, w1 = lo >> 5, ((hi - 1) >> 5) + 1
        self._alive_dev[w0:w1].copy_(torch.from_numpy(self._alive_host[w0:w1].view(np.int32)), non_blocking=False)

    def _clear_alive(self, rows: np.ndarray):
        np.bitwise_and.at(self._alive_host, rows >> 5, ~(np.uint32(1) << (rows & 31).astype(np.uint32)))
        touched = np.unique(rows >> 5)
        self._alive_dev[torch.from_numpy(touched).to(self.device)] = torch.from_numpy(
            self._alive_host[touched].view(np.int32)).to(self.device)

    @staticmethod
    def _bad_norm(what: str, i: int, nrm: float):
        return ValueError(f"{what}: row {i} has norm {nrm:.4f}; this collection is cosine (inner product of unit "
                          f"vectors) -- L2-normalise the vectors first")

    def _to_device_f32(self, x, what: str = "vectors", check_norm: bool = True) -> torch.Tensor:
        """[m, dim] float32 on the device; unit norm is checked where the data already is (host arrays on the host:
        a device-side check would put a synchronisation into every single-query call)"""
        if isinstance(x, torch.Tensor):
            t = x
            if t.dim() == 1:
                t = t.unsqueeze(0)
            if t.dim() == 2 and t.shape[0] and t.is_cuda and check_norm:
                nrm = torch.linalg.vector_norm(t.float(), dim=1)
                bad = (nrm - 1.0).abs() > 1e-2
                if bool(bad.any()):
                    i = int(torch.nonzero(bad)[0])
                    raise self._bad_norm(what, i, float(nrm[i]))
            elif t.dim() == 2 and t.shape[0] and not t.is_cuda:
                x = t.numpy()
        if not isinstance(x, torch.Tensor):
            a = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
            if a.ndim == 1:
                a = a[None, :]
            if a.ndim == 2 and a.shape[0]:
                nrm = np.sqrt(np.einsum("ij,ij->i", a, a))
                bad = np.abs(nrm - 1.0) > 1e-2
                if bad.any():
                    i = int(np.argmax(bad))
                    raise self._bad_norm(what, i, float(nrm[i]))
            t = torch.from_numpy(a)
        if t.dim() != 2 or t.shape[1] != self.dim:
            raise ValueError(f"embedding dimension {tuple(t.shape)} does not match collection dimensionality {self.dim}")
        return t.to(device=self.device, dtype=torch.float32, non_blocking=True).contiguous()

    def _pack_queries(self, q, check_norm: bool = True) -> torch.Tensor:
        """float32 [B, d] -> storage dtype [B, ld] with zero pad columns (device-side cast kernel)."""
        qf = self._to_device_f32(q, "query", check_norm)
        if self.dtype == torch.float32 and self.ld == self.dim:
            return qf          # already the stored form: no cast pass (one launch less on the single-query path)
        packed = torch.empty((qf.shape[0], self.ld), dtype=self.dtype, device=self.device)
        _native.append_rows(packed, 0, qf, self.dim)
        return packed

    # ------------------------------------------------------------------ collection API ----
    def add(self, embeddings, documents: Optional[Sequence[Optional[str]]] = None,
            metadatas: Optional[Sequence[Dict[str, Any]]] = None, ids: Optional[Sequence[str]] = None):
        if ids is None:
            raise ValueError("ids are required")
        emb = self._to_device_f32(embeddings, "add")
        m = emb.shape[0]
        if len(ids) != m:
            raise ValueError(f"{len(ids)} ids for {m} embeddings")
        documents = list(documents) if documents is not None else [None] * m
        metadatas = [dict(x) if x else {} for x in metadatas] if metadatas is not None else [{} for _ in range(m)]
        if len(documents) != m or len(metadatas) != m:
            raise ValueError("documents / metadatas length mismatch")
        with self._lock:
            fresh = [i for i, s in enumerate(ids) if s not in self._row_of]
            seen = set()
            keep = []
            for i in fresh:
                if ids[i] not in seen:
                    seen.add(ids[i])
                    keep.append(i)
            if len(keep) != m:
                logger.warning("Add of existing embedding ID ignored for %d of %d items", m - len(keep), m)
                if not keep:
                    return
                emb = emb[torch.tensor(keep, device=self.device)].contiguous()
            self._reserve(self._n + len(keep))
            _native.append_rows(self._matrix, self._n, emb, self.dim)
            for j, i in enumerate(keep):
                self._row_of[ids[i]] = self._n + j
                self._ids.append(ids[i])
                self._documents.append(documents[i])
                self._metadatas.append(metadatas[i])
            self._meta_index.append([metadatas[i] for i in keep])
            self._set_alive(self._n, self._n + len(keep))
            self._n += len(keep)
            self._grown(len(keep))

    def add_rows_device(self, rows_packed: torch.Tensor, documents, metadatas, ids):
        """Append rows that are already in storage layout [m, ld] (bulk loads, benchmarks)."""
        m = rows_packed.shape[0]
        with self._lock:
            self._reserve(self._n + m)
            self._matrix[self._n: self._n + m].copy_(rows_packed)
            for i in range(m):
                self._row_of[ids[i]] = self._n + i
            self._ids.extend(ids)
            self._documents.extend(documents if documents is not None else [None] * m)
            self._metadatas.extend(metadatas if metadatas is not None else [{} for _ in range(m)])
            self._meta_index.append(self._metadatas[self._n: self._n + m])
            self._set_alive(self._n, self._n + m)
            self._n += m
            self._grown(m)

    def _grown(self, m: int):
        """row tables grew by m (caller holds the lock): see config.MMRAG_GC_FREEZE_ROWS"""
        self._unfrozen_rows = getattr(self, "_unfrozen_rows", 0) + m
        from .config import settings

        every = settings.MMRAG_GC_FREEZE_ROWS
        if every > 0 and self._unfrozen_rows >= every:
            gc.freeze()
            self._unfrozen_rows = 0

    def _is_dead(self, rows: np.ndarray) -> np.ndarray:
        return ((self._alive_host[rows >> 5] >> (rows & 31).astype(np.uint32)) & 1) == 0

    def _rows_where(self, where: Optional[Dict[str, Any]]) -> np.ndarray:
        """Ascending LIVE rows whose metadata matches `where` (inverted index when the form allows, else a scan)."""
        fast = self._meta_index.rows(where)
        if fast is None:
            fast = np.fromiter((i for i in range(self._n) if match_where(self._metadatas[i], where)), dtype=np.int64)
        if self._n_dead and fast.size:
            fast = fast[~self._is_dead(fast)]
        return fast

    def _where_bits(self, where: Optional[Dict[str, Any]]) -> Optional[torch.Tensor]:
        """device alive bitmap for one search: tombstones AND `where` (ANDed on the device), or None = every row"""
        if not where:
            return self._alive_dev if self._n_dead else None
        flags = np.zeros(self._alive_host.size * 32, dtype=bool)
        flags[self._rows_where(where)] = True
        words = torch.from_numpy(np.packbits(flags, bitorder="little").view(np.int32)).to(self.device)
        return torch.bitwise_and(words, self._alive_dev) if self._n_dead else words

    def _launch_search(self, query_embeddings, n_results: int, where, check_norm: bool = True):
        """enqueue the search (caller holds the lock); returns device tensors, no host sync"""
        if n_results < 1:
            raise ValueError("n_results must be >= 1")
        q = self._pack_queries(query_embeddings, check_norm)
        bits = self._where_bits(where)
        if n_results <= _native.MAX_K and self.f32_exact and self.dtype == torch.float32 and q.shape[0] > 64:
            # exact float32 scores whatever the batch size (MMRAG_F32_EXACT_SEARCH): 64 queries per scan keep the exact
            # float32 matrix instruction; bigger batches would take the bf16-split path of csrc/search.hip
            parts = [_native.cosine_topk(q[i:i + 64], self._matrix, self._n, self.dim, n_results, alive_bits=bits)
                     for i in range(0, q.shape[0], 64)]
            return torch.cat([p[0] for p in parts], 0), torch.cat([p[1] for p in parts], 0)
        if n_results <= _native.MAX_K:
            need = _native.cosine_topk_workspace_bytes(q.shape[0], self._n, n_results)
            if self._search_ws is None or self._search_ws.numel() < need:
                self._search_ws = torch.empty(max(need, 16), dtype=torch.uint8, device=self.device)
            # (one workspace per index: searches are enqueued under the lock on the caller's current stream, and every
            # caller thread of the service uses the default stream, so consecutive scans are ordered on the device)
            return _native.cosine_topk(q, self._matrix, self._n, self.dim, n_results, alive_bits=bits,
                                       workspace=self._search_ws if torch.cuda.current_stream(self.device) == torch.cuda.default_stream(self.device) else None,
                                       packed_out=True)
        if q.shape[0] != 1:
            raise ValueError(f"n_results > {_native.MAX_K} is supported for single queries only")
        # deeper than the kernel's lists (get_similar_documents asks for n_results + 1): further passes with the rows
        # already returned masked out -- still exact, still ordered
        if bits is None:
            bits = self._alive_dev
        bits = bits.clone()
        out_s, out_r, left = [], [], n_results
        while left > 0:
            k = min(left, _native.MAX_K)
            s, r = _native.cosine_topk(q, self._matrix, self._n, self.dim, k, alive_bits=bits)
            out_s.append(s)
            out_r.append(r)
            got = r[0]
            got = got[got >= 0]
            if got.numel() < k:
                break
            host = got.cpu().numpy()
            w = bits.cpu().numpy().view(np.uint32).copy()
            np.bitwise_and.at(w, host >> 5, ~(np.uint32(1) << (host & 31).astype(np.uint32)))
            bits = torch.from_numpy(w.view(np.int32)).to(self.device)
            left -= k
        return torch.cat(out_s, 1), torch.cat(out_r, 1)

    def search(self, query_embeddings, n_results: int, where: Optional[Dict[str, Any]] = None):
        """Raw device search: (scores [B, k] float32 desc, rows [B, k] int64, -1 = none).

        The kernel selects up to MAX_K = 20 per pass (api.py:163 caps top_k at 20)."""
        with self._lock:
            return self._launch_search(query_embeddings, n_results, where)

    accepts_device_queries = True   # query() takes a device tensor as it is (EmbeddingManager's no-round-trip path)

    def query(self, query_embeddings, n_results: int = 10, where: Optional[Dict[str, Any]] = None,
              include: Sequence[str] = ("metadatas", "documents", "distances"), check_norm: bool = True) -> Dict[str, Any]:
        """Chroma-shaped result: lists of lists, ascending distance = 1 - cos, at most count() hits.
        `check_norm=False` (not in Chroma): the caller vouches for unit-norm rows -- the engine's own embeddings on the
        device, where the check would cost a host synchronisation per call.

        The lock is held only while the kernels are enqueued: concurrent callers (asyncio.to_thread workers,
        embedder.py:595) overlap their host waits and result building.  Row tables are append-only between
        compactions (a delete only clears alive bits and the id map) and a compaction swaps in NEW lists, so the
        snapshot taken under the lock stays valid: a hit whose row is deleted after the launch is still returned
        whole, exactly as if the delete had come a moment later."""
        with self._lock, stage("search"):
            scores, rows = self._launch_search(query_embeddings, n_results, where, check_norm)
            ids_t, docs_t, metas_t = self._ids, self._documents, self._metadatas
            emb_src = self._matrix if "embeddings" in include else None
        with stage("collect"):
            return self._collect(scores, rows, include, ids_t, docs_t, metas_t, emb_src)

    def _collect(self, scores, rows, include, ids_t, docs_t, metas_t, emb_src) -> Dict[str, Any]:
        # one device -> host copy each, then plain Python lists: per-element numpy scalars cost 10x a list item
        with stage("collect.wait"):               # (the device's share of the call: encoder + search finish here)
            if rows.is_cuda and rows._base is not None and scores._base is not None and rows._base.data_ptr() == scores._base.data_ptr():
                # packed [rows | scores] (cosine_topk(packed_out=True)): ONE copy into pinned memory and a wait for
                # ITS event -- a blocking copy to pageable memory (`.cpu()`) holds the stream inside the runtime
                # until it is through, and other callers' launches queue up behind it
                host = torch.empty(rows._base.shape, dtype=rows._base.dtype).pin_memory()
                host.copy_(rows._base, non_blocking=True)
                done = torch.cuda.Event()
                done.record()
                done.synchronize()
                nb = rows.numel()
                rows_h = host[: nb * 8].view(torch.int64).view(rows.shape)
                scores_h = host[nb * 8:].view(torch.float32).view(scores.shape)
            else:
                rows_h, scores_h = rows.cpu(), scores.cpu()
        dist_l = (1.0 - scores_h).tolist() if "distances" in include else None        # float32 arithmetic, as before
        want_m, want_d, want_e = "metadatas" in include, "documents" in include, "embeddings" in include
        out: Dict[str, Any] = {"ids": [], "distances": [] if dist_l is not None else None,
                               "metadatas": [] if want_m else None, "documents": [] if want_d else None,
                               "embeddings": [] if want_e else None}
        # every table is read with ONE itemgetter call over all hits of the batch (a C loop), then cut per query:
        # B x k Python-level index operations and dict() calls were most of the host time of a 256-query batch
        no_miss = bool(rows_h.numel()) and int(rows_h.min()) >= 0
        if no_miss and _HOSTROWS is not None and not want_e:
            # every query has all its k hits: one pass in C over the row numbers (csrc/hostrows.c) -- the same objects
            # in the same order as the comprehensions below, without the interpreter in the loop
            ids_ll, docs_ll, metas_ll = _HOSTROWS.gather(rows_h.contiguous().numpy(), rows_h.shape[1], ids_t,
                                                         docs_t if want_d else None, metas_t if want_m else None)
            out["ids"], out["documents"], out["metadatas"] = ids_ll, docs_ll, metas_ll
            if dist_l is not None:
                out["distances"] = dist_l
            return out
        if no_miss:                                                              # no misses at all (the usual batch)
            counts = None
            flat = rows_h.reshape(-1).tolist()
        else:
            rows_l = rows_h.tolist()
            counts = [k_ if row[-1] >= 0 else sum(1 for r in row if r >= 0) for row in rows_l for k_ in (len(row),)]
            flat = [r for row, c in zip(rows_l, counts) for r in row[:c]]        # misses (-1) only trail
        if len(flat) == 1:
            pick = lambda table: (table[flat[0]],)                               # noqa: E731  (itemgetter(x) alone returns the item)
        elif flat:
            getter = operator.itemgetter(*flat)
            pick = lambda table: getter(table)                                   # noqa: E731
        else:
            pick = lambda table: ()                                              # noqa: E731
        ids_f = pick(ids_t)
        metas_f = list(map(dict, pick(metas_t))) if want_m else None
        docs_f = pick(docs_t) if want_d else None
        if counts is None:
            # the common case, every query has all its k hits: cut the flat columns with one comprehension each
            k_, nf = rows_h.shape[1], len(flat)
            out["ids"] = [list(ids_f[lo:lo + k_]) for lo in range(0, nf, k_)]
            if dist_l is not None:
                out["distances"] = dist_l
            if want_m:
                out["metadatas"] = [metas_f[lo:lo + k_] for lo in range(0, nf, k_)]
            if want_d:
                out["documents"] = [list(docs_f[lo:lo + k_]) for lo in range(0, nf, k_)]
            if want_e:
                out["embeddings"] = [self._fetch(flat[lo:lo + k_], emb_src) for lo in range(0, nf, k_)]
            return out
        lo = 0
        for b, c in enumerate(counts):
            hi = lo + c
            out["ids"].append(list(ids_f[lo:hi]))
            if dist_l is not None:
                out["distances"].append(dist_l[b][:c])
            if want_m:
                out["metadatas"].append(metas_f[lo:hi])
            if want_d:
                out["documents"].append(list(docs_f[lo:hi]))
            if want_e:
                out["embeddings"].append(self._fetch(flat[lo:hi], emb_src))
            lo = hi
        return out

    def ids_of_rows(self, rows: Sequence[int]) -> List[str]:
        """ids of the given local rows (row numbers are stable until the next compaction)."""
        with self._lock:
            return [self._ids[int(r)] for r in rows]

    def _fetch(self, rows: List[int], matrix: Optional[torch.Tensor] = None) -> List[List[float]]:
        if not rows:
            return []
        m = self._matrix if matrix is None else matrix
        t = _native.fetch_rows_f32(m, torch.tensor(rows, dtype=torch.int64, device=self.device), self.dim)
        return t.cpu().numpy().tolist()

    def get(self, ids: Optional[Sequence[str]] = None, where: Optional[Dict[str, Any]] = None,
            include: Sequence[str] = ("metadatas", "documents")) -> Dict[str, Any]:
        with self._lock:
            if ids is not None:
                rows = [self._row_of[i] for i in ids if i in self._row_of]
                rows = [r for r in rows if match_where(self._metadatas[r], where)]
            else:
                rows = self._rows_where(where).tolist()
            out: Dict[str, Any] = {"ids": [self._ids[r] for r in rows]}
            out["metadatas"] = [dict(self._metadatas[r]) for r in rows] if "metadatas" in include else None
            out["documents"] = [self._documents[r] for r in rows] if "documents" in include else None
            out["embeddings"] = self._fetch(rows) if "embeddings" in include else None
            return out

    def delete(self, ids: Optional[Sequence[str]] = None, where: Optional[Dict[str, Any]] = None) -> List[str]:
        """Tombstone the matching rows (collection.delete, embedder.py:639-642): clear their alive bits on the
        device, drop them from the id map.  O(victims); the matrix is compacted only past COMPACT_DEAD_FRACTION."""
        with self._lock:
            if ids is not None:
                rows = [self._row_of[i] for i in ids if i in self._row_of]
                if where:
                    rows = [r for r in rows if match_where(self._metadatas[r], where)]
                rows = np.asarray(sorted(set(rows)), dtype=np.int64)
            else:
                rows = self._rows_where(where)
            if rows.size == 0:
                return []
            gone = [self._ids[int(r)] for r in rows]
            for s in gone:
                del self._row_of[s]
            # The row tables (_ids, _documents, _metadatas) are NOT touched: a query() that enqueued its search before
            # this delete builds its result from them after dropping the lock, and must still find the hit's id and
            # text there.  The alive bitmap and _row_of are what say "dead"; compact() drops the entries for good.
            self._clear_alive(rows)
            self._n_dead += int(rows.size)
            if self._n_dead >= self.COMPACT_MIN_DEAD and self._n_dead > self.COMPACT_DEAD_FRACTION * self._n:
                self.compact()
            return sorted(gone)

    def compact(self):
        """Drop the dead rows (stable: survivors keep their order).  New tables and a new matrix are swapped in, so
        result-building threads that still hold the old ones are unaffected."""
        with self._lock:
            if self._n_dead == 0:
                return
            keep = np.nonzero(~self._is_dead(np.arange(self._n, dtype=np.int64)))[0]
            cap = max(256, int(keep.size), self._matrix.shape[0] // 2 if keep.size < self._matrix.shape[0] // 4 else self._matrix.shape[0])
            dst = torch.zeros((cap, self.ld), dtype=self.dtype, device=self.device)
            if keep.size:
                _native.gather_rows(dst, self._matrix, torch.from_numpy(keep).to(self.device))
            self._matrix = dst
            self._ids = [self._ids[r] for r in keep]
            self._documents = [self._documents[r] for r in keep]
            self._metadatas = [self._metadatas[r] for r in keep]
            self._row_of = {s: i for i, s in enumerate(self._ids)}
            self._n = int(keep.size)
            self._n_dead = 0
            self._meta_index = MetaIndex()
            self._meta_index.append(self._metadatas)
            self._alive_host = np.zeros(self._n_words(cap), dtype=np.uint32)
            self._alive_dev = torch.zeros(self._n_words(cap), dtype=torch.int32, device=self.device)
            if self._n:
                self._set_alive(0, self._n)

    def reset(self):
        with self._lock:
            self._n = 0
            self._n_dead = 0
            self._ids, self._documents, self._metadatas, self._row_of = [], [], [], {}
            self._meta_index = MetaIndex()
            self._alive_host[:] = 0
            self._alive_dev.zero_()
