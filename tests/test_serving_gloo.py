"""CPU, multi-process (gloo): the multi-GPU serving loop (serving.ShardedCollection).  Rank 0 drives the
collection -- directly and through EmbeddingManager -- while the other ranks run worker_loop(); every
answer must equal what ONE collection holding all rows gives (the oracle stands in for the shard kernel)."""
import asyncio
import json
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from multimodal_rag_amd.serving import ShardedCollection
from tests.fakes import FakeCollection, FakeEngine

D = 32


def data(n=157):
    """small-integer vectors: every dot product is exact in float32 whatever the shard split, and equal
    scores are frequent, so the order of the answers exercises the insertion-order tie rule"""
    g = np.random.default_rng(21)
    v = g.integers(-2, 3, size=(n, D)).astype(np.float32)
    v[100] = v[3]                                   # duplicate vector, lands on another shard
    ids = [f"doc_{i // 50:012x}_text_{i}" for i in range(n)]
    metas = [{"doc_id": i_[:16], "item_id": i_[17:], "type": "image" if i % 4 == 0 else "text"} for i, i_ in enumerate(ids)]
    docs = [f"document {i}" for i in range(n)]
    q = g.integers(-2, 3, size=(9, D)).astype(np.float32)
    q[0] = v[3]
    return v, ids, metas, docs, q


def script(col):
    """the same calls against a sharded and a single collection; returns a JSON-able transcript"""
    v, ids, metas, docs, q = data()
    out = {}
    for lo in range(0, len(ids), 40):               # several adds, one duplicate id batch
        col.add(v[lo:lo + 40].tolist(), documents=docs[lo:lo + 40], metadatas=metas[lo:lo + 40], ids=ids[lo:lo + 40])
    col.add(v[:2].tolist(), documents=["dup", "dup"], metadatas=metas[:2], ids=ids[:2])
    out["count"] = col.count()
    out["q5"] = col.query(q.tolist(), n_results=5)
    out["q5_img"] = col.query(q.tolist(), n_results=5, where={"type": "image"})
    out["q1_deep"] = col.query(q[:1].tolist(), n_results=31, include=("distances",))
    out["q_more_than_rows"] = col.query(q[:2].tolist(), n_results=20, where={"doc_id": "doc_000000000003"})
    out["get_ids"] = col.get(ids=[ids[77], ids[3], "missing", ids[150]], include=("documents", "metadatas"))
    out["deleted"] = col.delete(where={"doc_id": "doc_000000000001"})
    out["count_after"] = col.count()
    out["q5_after"] = col.query(q.tolist(), n_results=5)
    out["get_where"] = sorted(col.get(where={"type": "image"}, include=())["ids"])
    return out


def fake_io():
    """save / load of the numpy stand-in shard (the GPU build uses persistence.save_index / load_index)"""
    def save(shard, d):
        np.save(os.path.join(d, "vecs.npy"), shard.vecs, allow_pickle=False)
        json.dump({"ids": shard.ids, "docs": shard.docs, "metas": shard.metas}, open(os.path.join(d, "t.json"), "w"))

    def load(d):
        c = FakeCollection(D)
        t = json.load(open(os.path.join(d, "t.json")))
        c.vecs, c.ids, c.docs, c.metas = np.load(os.path.join(d, "vecs.npy")), t["ids"], t["docs"], t["metas"]
        return c

    return save, load


def manager_script(m):
    async def go():
        await m.initialize()
        items = [{"id": f"text_{i}", "summary": f"summary number {i}", "raw": "", "type": "text"} for i in range(60)]
        counts = await m.embed_and_store(items, "doc_feedfeedfeed")
        r = await m.query("summary number 17", n_results=5)
        b = await m.batch_query(["summary number 3", "summary number 44"], n_results=3)
        sim = await m.get_similar_documents("doc_feedfeedfeed", "text_5", n_results=4)
        stats = await m.get_collection_stats()
        await m.delete_document("doc_feedfeedfeed")
        left = (await m.get_collection_stats())["count"]
        return {"counts": counts, "r": r, "b": b, "sim": sim, "count": stats["count"], "left": left}

    return asyncio.run(go())


class ShardedEngine(FakeEngine):
    """FakeEngine whose collections are sharded over the process group; with dp=True every rank embeds its own
    share of an ingest (the encoder is replicated)"""

    def __init__(self, dim, dp=False):
        super().__init__(dim)
        self.dp = dp

    def new_collection(self, name, metadata=None):
        return ShardedCollection(FakeCollection(self.dim, name, metadata), encode_fn=self.encode if self.dp else None)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        col = ShardedCollection(FakeCollection(D), shard_io=fake_io())
        if rank == 0:
            out = script(col)
            out["shard_counts"] = list(col._counts)
            # save, wipe, load: answers and later inserts behave as if nothing happened
            v, ids, metas, docs, q = data()
            before = col.query(q.tolist(), n_results=5)
            col.save(os.path.join(out_dir, f"saved_{world}"))
            col.reset()
            assert col.count() == 0
            col.load(os.path.join(out_dir, f"saved_{world}"))
            out["reload_same"] = col.query(q.tolist(), n_results=5) == before and col.count() == out["count_after"]
            col.add(v[120:122].tolist(), documents=["again", "again"], metadatas=metas[120:122], ids=["new_a", ids[130]])
            out["reload_add"] = [col.count(), col.query(v[120:121].tolist(), n_results=2)["ids"][0]]
            col.stop()
        else:
            col.worker_loop()
        # second phase: the collection behind EmbeddingManager
        from multimodal_rag_amd.embedder import EmbeddingManager

        for dp in (False, True):      # rank 0 embeds everything / every rank embeds its share
            eng = ShardedEngine(D, dp=dp)
            m = EmbeddingManager(engine=eng)
            if rank == 0:
                out["manager_dp" if dp else "manager"] = manager_script(m)
                out["encode_calls_dp" if dp else "encode_calls"] = sum(eng.calls)
                m.collection.stop()
            else:
                eng.new_collection("multimodal_rag").worker_loop()
        if rank == 0:
            json.dump(out, open(os.path.join(out_dir, f"sharded_{world}.json"), "w"))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _close(a, b):
    if isinstance(a, float) or isinstance(b, float):
        return abs(a - b) <= 1e-6
    if isinstance(a, dict):
        # include-dependent keys may be None or absent (the numpy stand-in ignores `include`)
        common = [k for k in a if a[k] is not None and b.get(k) is not None]
        return "ids" not in a or ("ids" in common and all(_close(a[k], b[k]) for k in common))
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(_close(x, y) for x, y in zip(a, b))
    return a == b


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_collection_equals_single_collection(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = json.load(open(tmp_path / f"sharded_{world}.json"))

    single = FakeCollection(D)
    want = json.loads(json.dumps(script(single)))           # same JSON round trip
    assert sum(got["shard_counts"]) == want["count_after"] and max(got["shard_counts"]) - min(got["shard_counts"]) <= 50
    for key in want:
        assert _close(got[key], want[key]), key
    assert got["reload_same"] is True
    # after the reload: one new id accepted (an existing one ignored), and the new row ties with its twin by insertion order
    assert got["reload_add"][0] == want["count_after"] + 1 and got["reload_add"][1] == ["doc_000000000002_text_120", "new_a"]            # ids in the same ORDER: ties break by insertion order
    assert got["q5"]["ids"][0][:2] == ["doc_000000000000_text_3", "doc_000000000002_text_100"]

    from multimodal_rag_amd.embedder import EmbeddingManager

    m = EmbeddingManager(engine=FakeEngine(D))
    want_m = json.loads(json.dumps(manager_script(m)))
    assert _close(got["manager"], want_m) and _close(got["manager_dp"], want_m)
    # data-parallel ingest: rank 0 encoded only its share of the 60 chunks (plus the queries)
    assert got["encode_calls_dp"] < got["encode_calls"] - 60 // world // 2


# ---- a rank whose local step raises must not wedge the service (worker_loop had no error handling) -------
class FlakyCollection(FakeCollection):
    """FakeCollection whose search / add raise while `boom` is set (set through a special `where` / id)"""

    def search(self, query_embeddings, n_results, where=None):
        if where and where.get("boom") == self.rank_tag:
            raise ValueError("unsupported where operator '$boom'")
        return super().search(query_embeddings, n_results, where)

    def add(self, embeddings, documents=None, metadatas=None, ids=None):
        if any(i.startswith(f"explode_on_{self.rank_tag}") for i in ids):
            raise MemoryError("shard is full")
        return super().add(embeddings, documents, metadatas, ids)

    def get(self, ids=None, where=None, include=("metadatas", "documents")):
        # the payload step of a query (after the all-gather): include=embeddings is a GPU fetch on a real index
        if "embeddings" in include and getattr(self, "payload_boom", False):
            raise RuntimeError("fetch_rows failed")
        return super().get(ids=ids, where=where, include=include)


def _flaky_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from multimodal_rag_amd.serving import ShardError

        shard = FlakyCollection(D)
        shard.rank_tag = rank
        shard.payload_boom = rank == 1        # the worker's payload step of an include=embeddings query raises
        import datetime

        col = ShardedCollection(shard, control_group=dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=60)))
        if rank != 0:
            col.worker_loop()
            return
        v, ids, metas, docs, q = data()
        col.add(v[:80].tolist(), documents=docs[:80], metadatas=metas[:80], ids=ids[:80])
        good = col.query(q.tolist(), n_results=5)
        out = {}
        # a worker raising AFTER the all-gather (merge / payload) used to skip the payload gather and leave
        # rank 0 waiting in it for ever; now it reports, rank 0 raises, and the service goes on
        try:
            col.query(q.tolist(), n_results=5, include=("embeddings", "distances"))
            out["payload"] = "no error"
        except ShardError as e:
            out["payload"] = str(e)
        out["after_payload"] = col.query(q.tolist(), n_results=5) == good
        col.start_keepalive(0.2)              # idle pings: the workers just keep waiting
        import time as _t
        _t.sleep(0.8)
        out["after_pings"] = col.query(q.tolist(), n_results=5) == good
        for bad_rank in (0, 1):              # the driving rank's shard fails / a worker's shard fails
            try:
                col.query(q.tolist(), n_results=5, where={"boom": bad_rank})
                out[f"query_{bad_rank}"] = "no error"
            except ShardError as e:
                out[f"query_{bad_rank}"] = str(e)
            out[f"after_query_{bad_rank}"] = col.query(q.tolist(), n_results=5) == good      # still serving
        # a failed add leaves nothing behind and can be retried (embedder.py:514-537 retries store calls)
        n0 = col.count()
        batch_ids = [f"explode_on_1_{i}" if i == 3 else f"fresh_{i}" for i in range(8)]
        try:
            col.add(v[80:88].tolist(), documents=docs[80:88], metadatas=metas[80:88], ids=batch_ids)
            out["add"] = "no error"
        except ShardError as e:
            out["add"] = str(e)
        out["count_after_failed_add"] = col.count() - n0
        retry_ids = [f"fresh_{i}" for i in range(8)]
        col.add(v[80:88].tolist(), documents=docs[80:88], metadatas=metas[80:88], ids=retry_ids)
        out["count_after_retry"] = col.count() - n0
        out["retry_rows_found"] = sorted(col.get(ids=retry_ids, include=())["ids"]) == sorted(retry_ids)
        # interleaved add / delete / query: equal to ONE collection fed the same calls
        single = FakeCollection(D)
        single.add(v[:80].tolist(), documents=docs[:80], metadatas=metas[:80], ids=ids[:80])
        single.add(v[80:88].tolist(), documents=docs[80:88], metadatas=metas[80:88], ids=retry_ids)
        same = True
        for step in range(6):
            lo = 88 + 10 * step
            for c in (col, single):
                c.add(v[lo:lo + 10].tolist(), documents=docs[lo:lo + 10], metadatas=metas[lo:lo + 10], ids=ids[lo:lo + 10])
                c.delete(ids=[ids[step * 7], ids[lo + 3]])
            a, b = col.query(q.tolist(), n_results=5), single.query(q.tolist(), n_results=5)
            same = same and a["ids"] == b["ids"] and np.allclose(a["distances"], b["distances"], atol=1e-6)
            same = same and col.count() == single.count()
        out["interleaved_equal"] = same
        col.stop()
        json.dump(out, open(os.path.join(out_dir, "flaky.json"), "w"))
    finally:
        dist.destroy_process_group()


def test_failing_rank_reports_and_service_survives(tmp_path):
    mp.spawn(_flaky_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    out = json.load(open(tmp_path / "flaky.json"))
    assert "rank 0: ValueError" in out["query_0"] and "rank 1" not in out["query_0"]
    assert "rank 1: ValueError" in out["query_1"]
    assert out["after_query_0"] and out["after_query_1"]
    assert "rank 1: RuntimeError: fetch_rows failed" in out["payload"] and "rank 0" not in out["payload"]
    assert out["after_payload"] and out["after_pings"]
    assert "rank 1: MemoryError" in out["add"]
    assert out["count_after_failed_add"] == 0           # the half-stored batch was taken out again
    assert out["count_after_retry"] == 8 and out["retry_rows_found"]
    assert out["interleaved_equal"]
