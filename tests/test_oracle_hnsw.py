"""CPU: the HNSW restatement (oracle/hnsw_oracle.cpp: what chromadb's approximate index does, Chroma's defaults)
behaves like the published algorithm: high recall where neighbourhoods exist, exact scores for what it returns,
monotone in ef.  Measurement infrastructure only (bench.py --hnsw-baseline); parity with hnswlib is unpinned."""
import shutil

import numpy as np
import pytest

from oracle import search_oracle as O

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="g++ needed to build the restatement")


def clustered(n, d, n_q, seed):
    g = np.random.default_rng(seed)
    cent = g.standard_normal((100, d)).astype(np.float32)
    x = cent[g.integers(100, size=n)] + 0.3 * g.standard_normal((n, d)).astype(np.float32)
    q = cent[g.integers(100, size=n_q)] + 0.3 * g.standard_normal((n_q, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True), q / np.linalg.norm(q, axis=1, keepdims=True)


def recall(r, er):
    return float(np.mean([len(set(a) & set(b)) / len(b) for a, b in zip(r.tolist(), er.tolist())]))


def test_hnsw_restatement_recall_and_scores():
    from oracle.hnsw_oracle import HnswIndex

    x, q = clustered(6000, 64, 100, 3)
    idx = HnswIndex(x, n_threads=4)
    es, er = O.cosine_topk(q, x, 5)
    rec = {}
    for ef in (10, 40, 160):
        s, r = idx.search(q, 5, ef, n_threads=2)
        rec[ef] = recall(r, er)
        assert np.all(r >= 0) and np.all(np.diff(s, axis=1) <= 1e-6)                 # valid rows, descending
        exact = np.einsum("bd,bkd->bk", q, x[r])
        assert np.abs(exact - s).max() <= 1e-5                                       # the scores it reports are exact
        assert all(len(set(row)) == 5 for row in r.tolist())
    assert rec[10] >= 0.85 and rec[40] >= 0.97 and rec[160] >= rec[40] - 0.01, rec


def test_hnsw_restatement_tiny_and_duplicates():
    from oracle.hnsw_oracle import HnswIndex

    x, q = clustered(7, 16, 3, 4)
    x[5] = x[2]
    idx = HnswIndex(x, n_threads=1)
    s, r = idx.search(q, 5, 10)
    es, er = O.cosine_topk(q, x, 5)
    assert recall(r, er) == 1.0 and np.allclose(np.sort(s, 1), np.sort(es, 1), atol=1e-6)
    s9, r9 = idx.search(q, 9, 10)                      # more than stored: padded with -1 / -inf
    assert np.all((r9 >= 0).sum(1) == 7) and np.all(np.isneginf(s9[r9 < 0]))
