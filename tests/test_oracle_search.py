"""CPU: pin the search oracle against the reference's only numeric artefact (the 70 WAL
vectors) and check its own invariants."""
import numpy as np

from oracle import search_oracle as O


def test_wal70_fixture_shape_and_norms(wal70):
    V = wal70["vectors"]
    assert V.shape == (70, 384) and V.dtype == np.float32
    assert np.all(np.abs(np.linalg.norm(V.astype(np.float64), axis=1) - 1.0) < 1e-6)
    assert len(wal70["ids"]) == 70 and len(wal70["log"]) == 140
    assert sum(m["type"] == "text" for m in wal70["metadatas"]) == 48
    assert sum(m["type"] == "image" for m in wal70["metadatas"]) == 22


def test_oracle_reproduces_wal70_float64_top5(wal70):
    V = wal70["vectors"]
    s, r = O.cosine_topk(V, V, 5)
    assert np.array_equal(r, wal70["top_rows"])
    assert np.all(np.abs(s - wal70["top_cos"]) < 1e-6)
    # every vector's nearest neighbour is itself (or an exact duplicate with a lower row)
    assert np.all(s[:, 0] > 0.999999)
    d = O.distances_from_scores(s)
    assert np.all(np.diff(d, axis=1) >= 0)  # ascending distance, embedder.py:604-609


def test_oracle_chunking_is_invisible():
    g = np.random.default_rng(0)
    c = g.standard_normal((1000, 64)).astype(np.float32)
    q = g.standard_normal((9, 64)).astype(np.float32)
    a = O.cosine_topk(q, c, 7, chunk_rows=1000)
    b = O.cosine_topk(q, c, 7, chunk_rows=37)
    # sgemm may block differently per chunk shape: scores agree to rounding, ids exactly
    assert np.allclose(a[0], b[0], atol=1e-5) and np.array_equal(a[1], b[1])
    brute = np.argsort(-(q @ c.T), axis=1, kind="stable")[:, :7]
    assert np.array_equal(a[1], brute)


def test_oracle_ties_lower_row_first_and_padding():
    c = np.tile(np.eye(4, dtype=np.float32)[:1], (6, 1))  # six identical rows
    q = np.eye(4, dtype=np.float32)[:1]
    s, r = O.cosine_topk(q, c, 4)
    assert r.tolist() == [[0, 1, 2, 3]]
    s, r = O.cosine_topk(q, c[:2], 4, row_offset=10)
    assert r.tolist() == [[10, 11, -1, -1]] and np.isinf(s[0, 2])
    alive = np.array([0, 1, 1, 0, 1, 1], bool)
    s, r = O.cosine_topk(q, c, 3, alive=alive)
    assert r.tolist() == [[1, 2, 4]]


def test_sharded_merge_equals_single_shard():
    g = np.random.default_rng(1)
    c = g.standard_normal((999, 32)).astype(np.float32)
    c[500] = c[3]  # cross-shard tie
    q = g.standard_normal((11, 32)).astype(np.float32)
    q[0] = c[3]
    full = O.cosine_topk(q, c, 5)
    for G in (2, 4, 8):
        per = -(-999 // G)
        parts = [O.cosine_topk(q, c[g0 * per:(g0 + 1) * per], 5, row_offset=g0 * per) for g0 in range(G)]
        s = np.stack([p[0] for p in parts])
        r = np.stack([p[1] for p in parts])
        ms, mr = O.merge_topk(s, r, 5)
        assert np.array_equal(ms, full[0]) and np.array_equal(mr, full[1])


def test_relevance_score_matches_api():
    assert O.relevance_score(0.25) == 0.75
    assert O.relevance_score(1.7) == 0.0
    assert O.relevance_score(0.12345) == 0.877
