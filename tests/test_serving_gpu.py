"""GPU: the serving loop's collection over a real VectorIndex shard and a real RCCL communicator (one rank on
the one-GPU box: ncclAllGather, host merge, payload gather all run; N > 1 is covered by tests/test_serving_gloo.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pg():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield True
    dist.destroy_process_group()


def test_sharded_collection_over_vector_index_matches_plain_index(pg, tmp_path):
    from multimodal_rag_amd.index import VectorIndex
    from multimodal_rag_amd.serving import ShardedCollection

    d, n = 384, 3000
    g = np.random.default_rng(4)
    v = g.standard_normal((n, d)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    v[2000] = v[7]
    ids = [f"doc_{i // 1000:012x}_text_{i}" for i in range(n)]
    metas = [{"doc_id": s[:16], "item_id": s[17:], "type": "image" if i % 5 == 0 else "text"} for i, s in enumerate(ids)]
    docs = [f"d{i}" for i in range(n)]
    q = g.standard_normal((17, d)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[0] = v[7]

    plain = VectorIndex(d, device="cuda:0")
    col = ShardedCollection(VectorIndex(d, device="cuda:0"), device=torch.device("cuda", 0))
    for c in (plain, col):
        for lo in range(0, n, 1000):
            c.add(v[lo:lo + 1000].tolist(), documents=docs[lo:lo + 1000], metadatas=metas[lo:lo + 1000], ids=ids[lo:lo + 1000])
    assert col.count() == plain.count() == n
    for kw in ({"n_results": 5}, {"n_results": 20, "where": {"type": "image"}}, {"n_results": 3, "where": {"doc_id": "doc_000000000001"}}):
        a, b = col.query(q.tolist(), **kw), plain.query(q.tolist(), **kw)
        assert a["ids"] == b["ids"] and a["documents"] == b["documents"] and a["metadatas"] == b["metadatas"]
        assert np.array_equal(np.array(a["distances"]), np.array(b["distances"]))
    assert col.query(q[:1].tolist(), n_results=5)["ids"][0][:2] == [ids[7], ids[2000]]     # tie: earlier insert first
    deep_a = col.query(q[:1].tolist(), n_results=45, include=("distances",))
    deep_b = plain.query(q[:1].tolist(), n_results=45, include=("distances",))
    assert deep_a["ids"] == deep_b["ids"] and len(deep_a["ids"][0]) == 45
    assert col.delete(where={"doc_id": "doc_000000000000"}) == plain.delete(where={"doc_id": "doc_000000000000"})
    assert col.count() == plain.count() == 2000
    assert col.query(q.tolist(), n_results=5)["ids"] == plain.query(q.tolist(), n_results=5)["ids"]
    got = col.get(ids=[ids[2500], ids[1001]], include=("documents", "embeddings"))
    want = plain.get(ids=[ids[2500], ids[1001]], include=("documents", "embeddings"))
    assert got["ids"] == want["ids"] and got["documents"] == want["documents"] and got["embeddings"] == want["embeddings"]
    # save / wipe / load through persistence.save_index / load_index
    before = col.query(q.tolist(), n_results=5)
    col.save(str(tmp_path / "saved"))
    col.reset()
    assert col.count() == 0
    col.load(str(tmp_path / "saved"))
    assert col.count() == 2000 and col.query(q.tolist(), n_results=5) == before
    col.add(v[:1].tolist(), documents=["back"], metadatas=metas[:1], ids=[ids[0]])
    assert col.count() == 2001 and col.query(v[:1].tolist(), n_results=1)["ids"][0] == [ids[0]]
    col.reset()
    assert col.count() == 0
    col.stop()


def test_manager_over_sharded_engine_with_data_parallel_ingest(pg):
    """serve_sharded.ShardedEngine (rank-0 side of the launcher) behind EmbeddingManager: ingest goes through
    add_texts (each rank embeds its own share -- here the single rank), answers equal the plain engine's."""
    import asyncio

    from multimodal_rag_amd.embedder import EmbeddingManager, HipEngine
    from multimodal_rag_amd.serve_sharded import ShardedEngine

    items = [{"id": f"text_{i}", "summary": f"chunk {i} about subject {i % 9} " + "word " * (i % 13), "raw": "", "type": "text"}
             for i in range(80)]

    async def go(m):
        await m.initialize()
        counts = await m.embed_and_store(items, "doc_5eed5eed5eed")
        r = await m.query("chunk 17 about subject 8", n_results=5)
        b = await m.batch_query(["chunk 3 about subject 3", "chunk 44 about subject 8"], n_results=3)
        n = (await m.get_collection_stats())["count"]
        await m.delete_document("doc_5eed5eed5eed")
        left = (await m.get_collection_stats())["count"]
        return counts, r, b, n, left

    plain = asyncio.run(go(EmbeddingManager(engine=HipEngine("all-MiniLM-L6-v2"))))
    eng = ShardedEngine(HipEngine("all-MiniLM-L6-v2"), torch.device("cuda", 0))
    m = EmbeddingManager(engine=eng)
    sharded = asyncio.run(go(m))
    assert eng._col is not None and eng._col.encode_fn is not None
    assert sharded[0] == plain[0] == {"text": 80, "table": 0, "image": 0}
    assert sharded[1]["ids"] == plain[1]["ids"] and np.allclose(sharded[1]["distances"], plain[1]["distances"], atol=1e-6)
    assert [x["ids"] for x in sharded[2]] == [x["ids"] for x in plain[2]]
    assert sharded[3:] == plain[3:] == (80, 0)
    eng._col.stop()


@pytest.mark.parametrize("merge", ["host", "device"])
def test_sharded_search_pipeline_over_rccl_equals_plain_search(pg, merge):
    """The exact path `bench.py --gpus N` takes (round-2 verdict: it had no -m gpu test): ShardedSearch with the
    corpus scan on the main stream (`local_scan`), the candidate merge + RCCL all-gather + copy-out on the
    high-priority side stream, four slots in flight -- on a one-rank RCCL group with `force_exchange=True`, so
    ncclAllGather, the packed block and the G*k -> k merge all run.  Several different query batches through the
    pipeline, each bit for bit equal to a plain mmrag_cosine_topk of the same batch."""
    from multimodal_rag_amd import _native as N
    from multimodal_rag_amd.sharded import ShardedSearch

    dev = torch.device("cuda", 0)
    d, n, B, k, SLOTS, lo = 768, 400_000, 256, 5, 4, 1_000_000     # lo: this rank's first global row
    dtype = torch.float16
    ld = N.padded_dim(d, dtype)
    g = torch.Generator(device=dev).manual_seed(12)
    corpus = torch.randn((n, ld), device=dev, generator=g)
    corpus[:, d:] = 0
    corpus = (corpus / corpus.norm(dim=1, keepdim=True)).to(dtype)
    batches = []
    for i in range(7):
        q = torch.randn((B, ld), device=dev, generator=g)
        q[:, d:] = 0
        batches.append((q / q.norm(dim=1, keepdim=True)).to(dtype))
    want = [N.cosine_topk(q, corpus, n, d, k, row_offset=lo) for q in batches]
    torch.cuda.synchronize()

    qbuf = [torch.empty_like(batches[0]) for _ in range(SLOTS)]
    ws = [torch.empty(N.cosine_topk_workspace_bytes(B, n, k) + 16, dtype=torch.uint8, device=dev) for _ in range(SLOTS)]
    plans = [N.SearchPlan(qbuf[i], corpus, n, d, k, ws[i]) for i in range(SLOTS)]
    main = torch.cuda.current_stream(dev)
    side_ptr = [main.cuda_stream]

    def local_scan(slot):
        qbuf[slot % SLOTS].copy_(batches[slot], non_blocking=True)     # (main stream: ordered before the scan)
        plans[slot % SLOTS].scan(main.cuda_stream)

    def local_finish(slot, out_s, out_r):
        plans[slot % SLOTS].select(lo, out_s.data_ptr(), out_r.data_ptr(), side_ptr[0])

    ss = ShardedSearch(B, k, 1, 0, dev, local_finish, merge=merge, local_scan=local_scan, force_exchange=True,
                       n_slots=SLOTS)
    assert ss.exchange and ss.side is not None
    side_ptr[0] = ss.side.cuda_stream
    got = {}
    LAG = SLOTS - 1
    for i in range(len(batches)):
        ss.launch(i)
        if i >= LAG:
            s, r = ss.finish(i - LAG)
            got[i - LAG] = (s.clone(), r.clone())
    for j in range(len(batches) - LAG, len(batches)):
        s, r = ss.finish(j)
        got[j] = (s.clone(), r.clone())
    torch.cuda.synchronize()
    for i, (ws_, wr_) in enumerate(want):
        assert torch.equal(got[i][1], wr_.cpu()) and torch.equal(got[i][0], ws_.cpu()), i
    assert int(want[0][1].min()) >= lo
