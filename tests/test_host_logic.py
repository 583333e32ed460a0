"""CPU tests of the host-side mirror of the reference interface (no GPU, no kernels): cache and
ordering semantics, retry/error conventions, key mapping, chunker, payload format."""
import asyncio
import gzip
import json
import os

import numpy as np
import pytest

from multimodal_rag_amd import ingest
from multimodal_rag_amd.embedder import EmbeddingManager, LRUCache
from multimodal_rag_amd.retriever import DocumentCache, InProcessKV, MultiVectorRetriever
from oracle import host_oracle as H
from tests.fakes import FakeEngine


def run(coro):
    return asyncio.run(coro)


async def no_sleep(_):
    return None


def manager(**kw):
    eng = FakeEngine()
    m = EmbeddingManager(engine=eng, **kw)
    m._sleep = no_sleep
    return m, eng


def summaries(n, kinds=("text",)):
    return [{"id": f"{kinds[i % len(kinds)]}_{i}", "summary": f"summary number {i}", "raw": f"raw {i}",
             "type": kinds[i % len(kinds)]} for i in range(n)]


# ---------------------------------------------------------------- LRU cache (embedder.py:26-80)
def test_lru_cache_semantics():
    c = LRUCache(maxsize=2)
    assert c.get("a") is None and c.misses == 1
    c.put("a", [1.0]); c.put("b", [2.0])
    assert c.get("a") == [1.0]           # refreshes a
    c.put("c", [3.0])                    # evicts b (least recently used)
    assert c.get("b") is None and c.get("c") == [3.0]
    assert c.get_stats() == {"size": 2, "maxsize": 2, "hits": 2, "misses": 2, "hit_rate": 0.5}
    c.clear()
    assert c.get_stats()["hits"] == 0 and c.get_stats()["hit_rate"] == 0.0
    assert DocumentCache(1).get_stats()["hit_rate"] == 0.0


# ---------------------------------------------------------------- embed_texts_batch (:266-347)
def test_embed_batch_order_cache_and_slicing():
    m, eng = manager(batch_size=4)
    texts = [f"t{i}" for i in range(10)]
    a = run(m.embed_texts_batch(texts))
    assert eng.calls == [4, 4, 2]                       # sequential slices of batch_size (:359-373)
    assert len(a) == 10 and all(isinstance(x, list) and isinstance(x[0], float) for x in a)
    b = run(m.embed_texts_batch(["new"] + texts[::-1]))  # 10 hits, 1 miss, order restored (:330-332)
    assert eng.calls[-1] == 1
    assert b[1:] == a[::-1]
    assert m.stats["total_embeddings_created"] == 11
    assert m.stats["cache_hits"] == 10 and m.stats["cache_misses"] == 11
    assert run(m.embed_texts_batch([])) == []
    assert m._get_cache_key("x") == H.cache_key("x")


def test_embed_batch_without_cache():
    m, eng = manager(enable_cache=False)
    run(m.embed_texts_batch(["a", "a"]))
    assert eng.calls == [2] and run(m.get_cache_stats()) == {"enabled": False}


# ---------------------------------------------------------------- embed_and_store (:428-500)
def test_embed_and_store_counts_ids_metadata():
    m, eng = manager()
    s = summaries(7, ("text", "table", "image")) + [{"id": "x_9", "summary": "odd", "raw": "", "type": "audio"}]
    counts = run(m.embed_and_store(s, "doc_abc123def456"))
    assert counts == {"text": 3, "table": 2, "image": 2}            # unknown types uncounted (:477-479)
    col = eng.collections[0]
    assert col.ids[0] == "doc_abc123def456_text_0" == H.chroma_id("doc_abc123def456", "text_0")
    assert col.metas[1] == {"doc_id": "doc_abc123def456", "item_id": "table_1", "type": "table"}
    assert col.docs[2] == "summary number 2"
    assert m.stats["total_items_stored"] == 8
    assert run(m.embed_and_store([], "doc_x")) == {"text": 0, "table": 0, "image": 0}   # (:448-450)


def test_store_retry_then_success_and_exhaustion():
    m, eng = manager()
    run(m.initialize())
    eng.collections[0].fail_next = 2
    assert run(m.embed_and_store(summaries(2), "doc_aaaaaaaaaaaa"))["text"] == 2   # 3rd attempt succeeds
    eng.collections[0].fail_next = 3
    with pytest.raises(RuntimeError):
        run(m.embed_and_store(summaries(2), "doc_bbbbbbbbbbbb"))


# ---------------------------------------------------------------- query (:539-617)
def test_query_shape_order_and_errors():
    m, eng = manager()
    run(m.embed_and_store(summaries(12), "doc_aaaaaaaaaaaa"))
    r = run(m.query("summary number 3", n_results=5))
    assert set(r) == {"ids", "distances", "metadatas", "documents"}
    assert r["ids"][0] == "doc_aaaaaaaaaaaa_text_3" and abs(r["distances"][0]) < 1e-6
    assert r["distances"] == sorted(r["distances"]) and len(r["ids"]) == 5
    assert len(run(m.query("anything", n_results=20))["ids"]) == 12         # fewer than k rows
    for bad in ("", "   "):
        with pytest.raises(ValueError, match="Query text cannot be empty"):
            run(m.query(bad))
    assert m.stats["total_queries"] == 2
    eng.collections[0].fail_next = 3
    with pytest.raises(RuntimeError):
        run(m.query("x"))


def test_batch_query_is_one_encode_and_keeps_error_dicts():
    m, eng = manager()
    run(m.embed_and_store(summaries(6), "doc_aaaaaaaaaaaa"))
    eng.calls.clear()
    run(m.clear_cache())
    out = run(m.batch_query(["summary number 1", "", "summary number 4"], n_results=2))
    assert eng.calls == [2]                                # one batch for the two valid queries
    assert out[0]["ids"][0].endswith("text_1") and out[2]["ids"][0].endswith("text_4")
    assert out[1] == {"ids": [], "distances": [], "metadatas": [], "documents": [],
                      "error": "Query text cannot be empty"}
    assert run(m.batch_query([])) == []
    singles = [run(m.query(q, 2)) for q in ("summary number 1", "summary number 4")]
    assert [o["ids"] for o in (out[0], out[2])] == [s["ids"] for s in singles]


def test_delete_similar_stats_rerank():
    m, eng = manager()
    run(m.embed_and_store(summaries(5), "doc_aaaaaaaaaaaa"))
    run(m.embed_and_store(summaries(3), "doc_bbbbbbbbbbbb"))
    sim = run(m.get_similar_documents("doc_aaaaaaaaaaaa", "text_2", n_results=3))
    assert "doc_aaaaaaaaaaaa_text_2" not in sim["ids"] and len(sim["ids"]) == 3
    assert sim["ids"][0] == "doc_bbbbbbbbbbbb_text_2"     # identical summary text in the other doc
    with pytest.raises(ValueError, match="Item not found"):
        run(m.get_similar_documents("doc_zz", "nope"))
    run(m.delete_document("doc_aaaaaaaaaaaa"))
    st = run(m.get_collection_stats())
    assert st["count"] == 3 and st["embedding_dim"] == 32 and st["batch_size"] == 32
    assert set(st) == {"name", "count", "model", "device", "embedding_dim", "batch_size", "stats", "cache"}
    assert run(m.get_stats()) == st
    res = run(m.query("summary number 1", 3))
    assert run(m.rerank_results("q", res, top_k=1))["ids"] == res["ids"][:1]
    run(m.delete_all_documents())
    assert run(m.get_collection_stats())["count"] == 0 and m.cache.get_stats()["size"] == 0
    assert run(m.query("summary number 1"))["ids"] == []
    run(m.cleanup())
    assert not m.is_initialized and m.get_embedding_dimension() == 384     # default (:734)


# ---------------------------------------------------------------- retriever
def test_item_id_to_key_kats():
    r = MultiVectorRetriever()
    assert r._item_id_to_redis_key("doc_abc123_chunk_0_a1b2c3") == "doc:doc_abc123:chunk_0_a1b2c3"  # docstring KAT
    assert r._item_id_to_redis_key("doc_abc") == "doc:doc_abc"
    assert r._item_id_to_redis_key("doc_5a0ed15f2e79_text_0") == "doc:doc_5a0ed15f2e79:text_0"
    for s in ("a", "a_b", "a_b_c", "doc_x_y_z_w"):
        assert r._item_id_to_redis_key(s) == H.item_id_to_store_key(s)


def test_retriever_store_retrieve_delete_payload_format():
    kv = InProcessKV()
    r = MultiVectorRetriever(store=kv)
    s = [{"id": "text_0", "summary": "s0", "raw": "RAW TEXT", "type": "text", "metadata": {"x": 1}},
         {"id": "image_1", "summary": "an image", "raw": "QkFTRTY0", "type": "image", "path": "/f/p.png"},
         {"id": "table_0", "summary": "tbl", "raw": "|a|b|", "type": "table"}]
    run(r.store_raw_documents("doc_abcdefabcdef", s, "f.txt"))
    stored = json.loads(gzip.decompress(kv.get("doc:doc_abcdefabcdef:image_1")))
    assert stored == {"id": "image_1", "type": "image", "raw": "QkFTRTY0", "summary": "an image", "path": "/f/p.png"}
    assert json.loads(kv.get("doc_index:doc_abcdefabcdef")) == ["text_0", "image_1", "table_0"]
    meta = json.loads(kv.get("doc_meta:doc_abcdefabcdef"))
    assert meta["chunks"] == {"text": 1, "table": 1, "image": 1} and meta["compressed"] is True
    ids = ["doc_abcdefabcdef_table_0", "doc_abcdefabcdef_text_0", "doc_missing_x", "doc_abcdefabcdef_image_1"]
    out = run(r.retrieve_raw_documents(ids))
    assert out == {"text_chunks": ["RAW TEXT"], "table_chunks": ["|a|b|"], "image_chunks": ["QkFTRTY0"]}
    run(r.retrieve_raw_documents(ids))
    assert r.stats["cache_hits"] == 3 and r.stats["total_retrieved"] == 8
    assert run(r.retrieve_raw_documents([])) == {"text_chunks": [], "table_chunks": [], "image_chunks": []}
    assert [d["doc_id"] for d in run(r.list_all_documents())] == ["doc_abcdefabcdef"]
    assert run(r.health_check())["healthy"] is True
    st = run(r.get_stats())
    assert st["features"] == {"compression": True, "cache": True} and 0 < st["compression"]["ratio"]
    run(r.delete_document("doc_abcdefabcdef"))
    assert kv.scan_iter("doc*") == [] and run(r.retrieve_raw_documents(ids))["text_chunks"] == []


# ---------------------------------------------------------------- chunker (parser.py:1702-1736)
def test_chunker_sample_document_is_one_chunk(golden_dir):
    raw = open(os.path.join(golden_dir, "sample_document.txt"), "rb").read()
    text = raw.decode("utf-8")
    assert len(text) == 579 and len(text.replace("\r\n", "\n")) == 563
    chunks = ingest.basic_chunk_text(text, 1000, 200)
    assert len(chunks) == 1 and chunks[0] == text.strip()
    assert chunks == H.basic_chunk_text(text)


def test_chunker_boundaries_hand_computed():
    s = ("a" * 598 + ". ") + ("b" * 598 + ". ") + ("c" * 400)      # sentence ends at 600 and 1200
    chunks = ingest.basic_chunk_text(s, 1000, 200)
    # window 0 = [0,1000): last '. ' at index 598 > 500 -> chunk = s[:599], next start = 599 - 200
    assert chunks[0] == "a" * 598 + "."
    assert chunks[1].startswith("a" * 199 + ". " + "b" * 10) and chunks[1].endswith("b.")
    assert chunks == H.basic_chunk_text(s, 1000, 200)
    assert ingest.basic_chunk_text("   \n ") == [] and ingest.basic_chunk_text("") == []
    no_boundary = "x" * 2500
    got = ingest.basic_chunk_text(no_boundary, 1000, 200)
    assert [len(c) for c in got] == [1000, 1000, 900, 100] and got == H.basic_chunk_text(no_boundary)
    assert ingest.fallback_summary("x" * 10, 300) == "x" * 10 == H.fallback_summary("x" * 10, 300)
    long = "First sentence here. " * 30
    assert ingest.fallback_summary(long, 300) == H.fallback_summary(long, 300) and ingest.fallback_summary(long, 300).endswith(".")
    assert ingest.fallback_summary("", 10) == "Content unavailable"


# ---------------------------------------------------------------- dynamic batching dispatcher
def test_dispatcher_batches_concurrent_queries():
    async def main():
        m, eng = manager(enable_cache=False)
        await m.embed_and_store(summaries(40), "doc_aaaaaaaaaaaa")
        singles = [await m.query(f"summary number {i}", 3) for i in range(40)]
        eng.calls.clear()
        d = m.enable_dynamic_batching(max_batch=64, max_wait_ms=20)
        outs = await asyncio.gather(*[m.query(f"summary number {i}", 3) for i in range(40)])
        assert [o["ids"] for o in outs] == [s["ids"] for s in singles]
        assert len(eng.calls) <= 3 and sum(eng.calls) == 40          # batched encodes instead of 40
        assert d.stats["requests"] == 40 and d.stats["max_batch_seen"] >= 20
        mixed = await asyncio.gather(m.query("summary number 1", 2), m.query("summary number 2", 5),
                                     return_exceptions=True)
        assert len(mixed[0]["ids"]) == 2 and len(mixed[1]["ids"]) == 5   # grouped per k
        with pytest.raises(ValueError):
            await m.query("  ")
        await m.cleanup()
    run(main())


# ---------------------------------------------------------------- joint-space engines: image items by pixels
class FakeJointEngine(FakeEngine):
    """adds encode_images: the vector is keyed on the mean pixel, so tests can tell which path was taken"""

    def __init__(self):
        super().__init__()
        self.image_calls = []

    def encode_images(self, images):
        self.image_calls.append([im.shape for im in images])
        out = np.zeros((len(images), self.dim), np.float32)
        for i, im in enumerate(images):
            out[i, int(im.mean()) % self.dim] = 1.0
        return out


def test_image_items_are_embedded_from_pixels_on_joint_engines(tmp_path):
    import base64
    import io

    Image = pytest.importorskip("PIL.Image")
    eng = FakeJointEngine()
    m = EmbeddingManager(engine=eng)
    m._sleep = no_sleep
    png = tmp_path / "fig.png"
    Image.fromarray(np.full((40, 60, 3), 7, np.uint8)).save(png)
    buf = io.BytesIO()
    Image.fromarray(np.full((30, 30, 3), 9, np.uint8)).save(buf, format="PNG")
    items = [
        {"id": "text_0", "summary": "plain text", "raw": "plain text", "type": "text"},
        {"id": "image_0", "summary": "a figure", "raw": "", "type": "image", "path": str(png)},
        {"id": "image_1", "summary": "inline figure", "raw": base64.b64encode(buf.getvalue()).decode(), "type": "image"},
        {"id": "image_2", "summary": "undecodable figure", "raw": "not an image", "type": "image"},
    ]
    counts = run(m.embed_and_store(items, "doc_aaaaaaaaaaaa"))
    assert counts == {"text": 1, "table": 0, "image": 3}
    assert eng.image_calls == [[(40, 60, 3), (30, 30, 3)]]
    got = m.collection.get(ids=[f"doc_aaaaaaaaaaaa_{it['id']}" for it in items], include=["embeddings", "documents"])
    E = np.asarray(got["embeddings"], np.float32)
    assert E[1, 7] == 1.0 and E[2, 9] == 1.0                       # pixel path
    assert np.allclose(E[0], eng.encode(["plain text"])[0]) and np.allclose(E[3], eng.encode(["undecodable figure"])[0])
    assert got["documents"][1] == "a figure"                        # the stored document is still the summary


def test_text_only_engines_keep_reference_behaviour_for_images():
    m, eng = manager()
    items = summaries(3, kinds=("image",))
    run(m.embed_and_store(items, "doc_bbbbbbbbbbbb"))
    assert not hasattr(eng, "encode_images") and eng.calls == [3]   # embedder.py:452: summaries are what is embedded


# ---------------------------------------------------------------- metadata inverted index (index.MetaIndex)
def test_meta_index_equals_full_scan():
    from multimodal_rag_amd.index import MetaIndex, match_where

    g = np.random.default_rng(3)
    metas = []
    for i in range(500):
        m = {"doc_id": f"doc_{int(g.integers(7))}", "type": ["text", "table", "image"][int(g.integers(3))], "page": int(g.integers(5))}
        if i % 50 == 0:
            m["tags"] = ["a", "b"]          # unhashable value under another key
        if i % 11 == 0:
            del m["page"]
        metas.append(m)
    idx = MetaIndex()
    idx.append(metas[:200])
    idx.append(metas[200:])
    wheres = [
        None, {}, {"doc_id": "doc_3"}, {"doc_id": "nope"}, {"missing_key": "x"}, {"type": {"$eq": "image"}},
        {"type": {"$in": ["text", "image"]}}, {"type": {"$in": []}}, {"doc_id": "doc_1", "type": "table"},
        {"$and": [{"doc_id": "doc_2"}, {"page": 3}]}, {"$or": [{"doc_id": "doc_2"}, {"type": "image"}, {"page": 4}]},
        {"$and": [{"$or": [{"page": 1}, {"page": 2}]}, {"type": "text"}]}, {"page": 0},
    ]
    for w in wheres:
        want = [i for i, m in enumerate(metas) if match_where(m, w)]
        got = idx.rows(w)
        assert got is not None and got.tolist() == want, w
    # forms outside the index fall back (None), never a wrong answer
    for w in [{"page": {"$gt": 2}}, {"page": {"$ne": 1}}, {"tags": ["a", "b"]}, {"page": None}, {"type": {"$nin": ["text"]}}]:
        got = idx.rows(w)
        assert got is None or got.tolist() == [i for i, m in enumerate(metas) if match_where(m, w)], w


def test_stage_timers_and_stats_key(monkeypatch):
    """tracing.stage: per-stage wall clock (always on), read through EmbeddingManager.get_stage_timers(); roctx stays off unless
    MMRAG_ROCTX=1, and a missing library only turns the ranges off"""
    import asyncio
    import time as _time

    from multimodal_rag_amd import tracing

    tracing.reset()
    with tracing.stage("unit.a"):
        _time.sleep(0.01)
    with tracing.stage("unit.a"):
        pass
    try:
        with tracing.stage("unit.b"):
            raise RuntimeError("stage bodies may raise")
    except RuntimeError:
        pass
    snap = tracing.snapshot()
    assert snap["unit.a"]["calls"] == 2 and snap["unit.a"]["total_s"] >= 0.009 and snap["unit.a"]["max_ms"] >= 9.0
    assert snap["unit.b"]["calls"] == 1
    assert not tracing.roctx_enabled()

    m = EmbeddingManager(engine=FakeEngine(), enable_cache=False)
    asyncio.run(m.embed_and_store([{"id": "text_0", "summary": "alpha beta", "raw": "", "type": "text"}], "doc_aaaaaaaaaaaa"))
    asyncio.run(m.query("alpha"))
    assert "unit.a" in m.get_stage_timers()
    tracing.reset()
    assert tracing.snapshot() == {}


def test_delete_between_search_launch_and_result_building_keeps_hits_whole():
    """query() drops the index lock after enqueueing the kernels and builds its result from the row tables;
    a delete() landing in between used to blank `_ids[r]` / `_documents[r]` in place, so the result carried id None
    (and `retriever._item_id_to_redis_key(None)` raised -> HTTP 500).  The tables now stay whole until compaction.
    Host tables only: the index lives on the CPU device here, no kernel is called."""
    import numpy as np
    import torch
    from multimodal_rag_amd.index import VectorIndex

    ix = VectorIndex(dim=8, dtype=torch.float32, device="cpu")
    ids = [f"doc_a_text_{i}" for i in range(6)]
    rows = torch.zeros((6, ix.ld), dtype=torch.float32)
    rows[:, 0] = 1.0
    ix.add_rows_device(rows, [f"text {i}" for i in range(6)], [{"doc_id": "doc_a", "type": "text"} for _ in ids], ids)
    # what query() holds when it leaves the lock: device results (here: rows 1, 4 and a miss) and the table snapshot
    scores = torch.tensor([[0.9, 0.8, float("-inf")]])
    hit_rows = torch.tensor([[1, 4, -1]], dtype=torch.int64)
    snap = (ix._ids, ix._documents, ix._metadatas)
    assert ix.delete(ids=[ids[1]]) == [ids[1]]            # lands before the result is built
    out = ix._collect(scores, hit_rows, ("metadatas", "documents", "distances"), *snap, None)
    assert out["ids"] == [[ids[1], ids[4]]] and None not in out["ids"][0]
    assert out["documents"] == [["text 1", "text 4"]]
    assert out["metadatas"][0][0]["doc_id"] == "doc_a"
    # the row is dead for everything that starts after the delete
    assert ix.count() == 5 and ids[1] not in ix._row_of
    assert ix.get(ids=[ids[1]])["ids"] == []
    assert ids[1] not in ix.get(where={"doc_id": "doc_a"})["ids"]
    assert ix._rows_where({"$or": [{"doc_id": "doc_a"}, {"type": "text"}]}).tolist() == [0, 2, 3, 4, 5]   # scan fallback
    # (compact() moves rows with a HIP kernel: tests/test_pipeline_gpu.py covers it)


def test_result_rows_built_in_c_equal_the_python_form():
    """csrc/hostrows.c (CPython extension built by `python -m multimodal_rag_amd.build`): the per-query lists of a
    batch's hits -- same objects, same order, metadata dicts copied -- and the per-query dicts, against the plain
    Python loops they replace; rows outside the tables are an IndexError, not a read past the end"""
    import numpy as np
    import torch

    import multimodal_rag_amd.embedder as E
    import multimodal_rag_amd.index as I
    from multimodal_rag_amd.hostutil import load_hostrows

    H = load_hostrows()
    assert H is not None, "lib/_hostrows*.so is missing: run python -m multimodal_rag_amd.build"
    n, B, k = 5000, 37, 5
    ids = [f"id{i}" for i in range(n)]
    docs = [None if i % 3 else f"text {i}" for i in range(n)]
    metas = [{"doc_id": f"d{i // 7}", "n": i} for i in range(n)]
    g = np.random.default_rng(5)
    rows = torch.from_numpy(g.integers(0, n, size=(B, k)).astype(np.int64))
    scores = torch.from_numpy(g.random((B, k)).astype(np.float32))
    idx = I.VectorIndex.__new__(I.VectorIndex)
    inc = ["metadatas", "documents", "distances"]

    def both(r, include):
        saved = I._HOSTROWS, E._HOSTROWS
        try:
            I._HOSTROWS = E._HOSTROWS = H
            c = E.EmbeddingManager._split(idx._collect(scores, r, include, ids, docs, metas, None), B)
            I._HOSTROWS = E._HOSTROWS = None
            p = E.EmbeddingManager._split(idx._collect(scores, r, include, ids, docs, metas, None), B)
        finally:
            I._HOSTROWS, E._HOSTROWS = saved
        return c, p

    c, p = both(rows, inc)
    assert c == p and len(c) == B and c[0]["ids"] == [ids[int(r)] for r in rows[0]]
    assert c[3]["metadatas"][1] == metas[int(rows[3, 1])] and c[3]["metadatas"][1] is not metas[int(rows[3, 1])]
    assert c[3]["documents"][2] is docs[int(rows[3, 2])]
    c, p = both(rows, ["distances"])                       # columns not asked for
    assert c == p and c[0]["metadatas"] == [] and c[0]["documents"] == []
    miss = rows.clone()
    miss[4, 2:] = -1
    miss[9, :] = -1
    c, p = both(miss, inc)                                 # misses take the Python form either way
    assert c == p and len(c[4]["ids"]) == 2 and c[9]["ids"] == []
    with pytest.raises(IndexError):
        H.gather(np.array([[0, n]], dtype=np.int64), 2, ids, docs, metas)
    with pytest.raises(ValueError):
        H.gather(np.zeros(7, dtype=np.int64), 2, ids, None, None)
    assert H.split(("a", "b"), ([1, 2], [[3], [4]])) == [{"a": 1, "b": [3]}, {"a": 2, "b": [4]}]


def test_row_tables_leave_the_cyclic_collector_after_bulk_growth(monkeypatch):
    """config.MMRAG_GC_FREEZE_ROWS: once that many rows have been added, everything alive is frozen out of the cyclic
    collector's generations (gc.freeze); 0 turns the policy off"""
    import gc

    import multimodal_rag_amd.index as I
    from multimodal_rag_amd.config import settings

    idx = I.VectorIndex.__new__(I.VectorIndex)
    try:
        gc.unfreeze()
        monkeypatch.setattr(settings, "MMRAG_GC_FREEZE_ROWS", 1000)
        idx._grown(999)
        assert gc.get_freeze_count() == 0
        idx._grown(1)
        assert gc.get_freeze_count() > 0 and idx._unfrozen_rows == 0
        gc.unfreeze()
        monkeypatch.setattr(settings, "MMRAG_GC_FREEZE_ROWS", 0)
        idx._grown(10 ** 7)
        assert gc.get_freeze_count() == 0
    finally:
        gc.unfreeze()
