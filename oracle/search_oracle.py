"""CPU oracle for the retrieve half of the hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package (multimodal_rag_amd/) never does.

What it restates
----------------
The reference delegates search to chromadb 0.4.22 -> chroma-hnswlib (C++ HNSW,
M=16, ef_construction=100), neither of which is present under /root/reference
(requirements.txt:21) nor installable here.  HNSW is an *approximation* of exact
k-NN, so the oracle restates the exact definition the approximation targets, anchored
on the reference's own call site and result handling:

  * embedder.py:595-601  collection.query(query_embeddings=[v], n_results=k,
                         include=['metadatas','documents','distances'])
  * embedder.py:604-609  flatten [0] -> {'ids','distances','metadatas','documents'},
                         ascending distance, at most k entries (fewer if N < k)
  * embedder.py:402      vectors are L2-normalised at encode time
  * chroma_db/chroma.sqlite3 collection_metadata: hnsw:space = cosine
                         => distance = 1 - cos
  * api.py:390-394       relevance_score = round(1 - min(distance, 1.0), 3)

Arithmetic: score = sum_j q[j]*c[j] accumulated in float32 over float32 inputs
(fp16/bf16 storage is up-cast exactly first), distance = 1 - score.  Order:
descending score, ties -> lower insertion row first (the build's documented rule;
hnswlib gives no tie guarantee).

Parity pin: the reference has no tests; the oracle is pinned against
the only reference-produced numeric artefact, the 70 WAL vectors
(tests/golden/wal70_*), whose float64 all-pairs top-5 it must reproduce
(tests/test_oracle_search.py).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

NEG_INF = np.float32(-np.inf)


def _as_f32(x: np.ndarray) -> np.ndarray:
    """Exact up-cast of fp16 / fp32 storage to float32 (embedder.py:400 convert_to_numpy)."""
    return np.ascontiguousarray(x, dtype=np.float32)


def _select(scores: np.ndarray, rows: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """Exact top-k of one query's candidate set under (score desc, row asc)."""
    n = scores.shape[0]
    if n > k:
        # keep every candidate tied with the k-th score so the tie rule decides, not argpartition
        kth = np.partition(scores, n - k)[n - k]
        keep = scores >= kth
        scores, rows = scores[keep], rows[keep]
    order = np.lexsort((rows, -scores.astype(np.float64)))[:k]
    return scores[order], rows[order]


def cosine_topk(
    q: np.ndarray,
    corpus: np.ndarray,
    k: int,
    row_offset: int = 0,
    alive: Optional[np.ndarray] = None,
    chunk_rows: int = 131072,
) -> Tuple[np.ndarray, np.ndarray]:
    """Exact batched k-NN by inner product (rows are unit-norm => cosine).

    Restates what `collection.query` (embedder.py:595-601) approximates.

    q       [B, d]  fp32 or fp16
    corpus  [n, d]  fp32 or fp16
    alive   [n] bool, optional: dead rows are never returned (delete_document,
            embedder.py:619-656)
    returns scores [B, k] float32 descending, rows [B, k] int64 (global = row_offset +
            local); when fewer than k live rows exist the tail is (-inf, -1).
    """
    qf = _as_f32(q)
    B = qf.shape[0]
    n = corpus.shape[0]
    best_s = np.full((B, k), NEG_INF, dtype=np.float32)
    best_r = np.full((B, k), -1, dtype=np.int64)
    for lo in range(0, n, chunk_rows):
        hi = min(n, lo + chunk_rows)
        c = _as_f32(corpus[lo:hi])
        s = qf @ c.T  # sgemm: float32 accumulate
        if alive is not None:
            s[:, ~np.asarray(alive[lo:hi], dtype=bool)] = NEG_INF
        kk = min(k, hi - lo)
        # per-row candidate reduction: anything >= the chunk's kk-th score can still win
        part = np.partition(s, (hi - lo) - kk, axis=1)[:, (hi - lo) - kk]
        for b in range(B):
            idx = np.nonzero(s[b] >= part[b])[0]
            cs = np.concatenate([best_s[b], s[b, idx]])
            cr = np.concatenate([best_r[b], idx.astype(np.int64) + lo])
            live = cs > NEG_INF
            cs, cr = cs[live], cr[live]
            ss, rr = _select(cs, cr, k)
            best_s[b, : ss.shape[0]] = ss
            best_r[b, : rr.shape[0]] = rr
            best_s[b, ss.shape[0]:] = NEG_INF
            best_r[b, rr.shape[0]:] = -1
    out_r = np.where(best_r >= 0, best_r + row_offset, -1)
    return best_s, out_r


def merge_topk(scores: np.ndarray, rows: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """Merge G shards' local top-k into the global top-k (north star: final host merge).

    scores [G, B, k'] float32, rows [G, B, k'] int64 (global ids, -1 = padding).
    Global top-k is a subset of the union of local top-k, so this equals the single-shard
    answer bit for bit given the (score desc, row asc) rule.
    """
    G, B, kp = scores.shape
    out_s = np.full((B, k), NEG_INF, dtype=np.float32)
    out_r = np.full((B, k), -1, dtype=np.int64)
    for b in range(B):
        cs = scores[:, b, :].reshape(-1)
        cr = rows[:, b, :].reshape(-1)
        live = cr >= 0
        ss, rr = _select(cs[live], cr[live], k)
        out_s[b, : ss.shape[0]] = ss
        out_r[b, : rr.shape[0]] = rr
    return out_s, out_r


def distances_from_scores(scores: np.ndarray) -> np.ndarray:
    """cosine-space distance as Chroma returns it: 1 - cos."""
    return (np.float32(1.0) - scores.astype(np.float32)).astype(np.float32)


def relevance_score(distance: float) -> float:
    """api.py:390-394: round(float(1.0 - min(distance, 1.0)), 3)."""
    return round(float(1.0 - min(distance, 1.0)), 3)


def same_topk_sets(rows_a, scores_a, rows_b, scores_b, margin: float = 2e-4) -> bool:
    """Parity rule of BASELINE.md section 4: identical id sets, except that candidates whose
    score lies within `margin` of the k-th score are interchangeable (fp16 storage +
    different fp32 summation orders can flip near-ties)."""
    rows_a, rows_b = np.asarray(rows_a), np.asarray(rows_b)
    scores_a, scores_b = np.asarray(scores_a), np.asarray(scores_b)
    for b in range(rows_a.shape[0]):
        sa, sb = set(rows_a[b].tolist()), set(rows_b[b].tolist())
        if sa == sb:
            continue
        kth = min(scores_a[b].min(), scores_b[b].min())
        for r, s in list(zip(rows_a[b], scores_a[b])) + list(zip(rows_b[b], scores_b[b])):
            if (r in sa) != (r in sb) and abs(float(s) - float(kth)) > margin:
                return False
    return True
