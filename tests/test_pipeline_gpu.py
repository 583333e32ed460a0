"""GPU: the host mirror driving the real HIP engine end to end (VectorIndex, EmbeddingManager,
FastAPI surface), checked against the CPU oracle on the embeddings the device produced."""
import asyncio
import os

import numpy as np
import pytest
import torch

from oracle import search_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from multimodal_rag_amd import _native

    _native.lib()
    return True


def run(coro):
    return asyncio.run(coro)


def unit(n, d, seed):
    x = np.random.default_rng(seed).standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_vector_index_collection_semantics(gpu, dtype):
    from multimodal_rag_amd.index import VectorIndex

    d = 384
    idx = VectorIndex(d, dtype=dtype, capacity=256)
    V = unit(1500, d, 1)
    ids = [f"doc_{i // 100:012x}_text_{i % 100}" for i in range(1500)]
    metas = [{"doc_id": s[:16], "item_id": s[17:], "type": "text" if i % 3 else "image"} for i, s in enumerate(ids)]
    for lo in range(0, 1500, 500):   # growth across capacity doublings
        idx.add(V[lo:lo + 500].tolist(), [f"d{i}" for i in range(lo, lo + 500)], metas[lo:lo + 500], ids[lo:lo + 500])
    assert idx.count() == 1500
    idx.add(V[:2].tolist(), ["dup", "dup"], metas[:2], ids[:2])      # duplicate ids ignored
    assert idx.count() == 1500
    stored = np.asarray(idx.get(ids=ids, include=["embeddings"])["embeddings"], np.float32)
    assert np.abs(stored - V).max() <= (1e-3 if dtype == torch.float16 else 0)

    q = unit(9, d, 2)
    res = idx.query(q.tolist(), n_results=5)
    es, er = O.cosine_topk(q.astype(np.float16).astype(np.float32) if dtype == torch.float16 else q, stored, 5)
    assert [[ids[r] for r in row] for row in er] == res["ids"]
    assert np.allclose(np.array(res["distances"]), 1.0 - es, atol=1e-4)
    assert res["documents"][0][0] == f"d{er[0][0]}" and res["metadatas"][0][0] == metas[er[0][0]]
    assert all(a <= b + 1e-7 for row in res["distances"] for a, b in zip(row, row[1:]))

    # where filter == alive mask in the kernel epilogue
    res_f = idx.query(q.tolist(), n_results=5, where={"type": "image"})
    alive = np.array([m["type"] == "image" for m in metas])
    _, er_f = O.cosine_topk(q.astype(np.float16).astype(np.float32) if dtype == torch.float16 else q, stored, 5, alive=alive)
    assert [[ids[r] for r in row] for row in er_f] == res_f["ids"]

    # deep k (> MAX_K) by masked passes
    deep = idx.query(q[:1].tolist(), n_results=45)
    _, er_d = O.cosine_topk(q[:1].astype(np.float16).astype(np.float32) if dtype == torch.float16 else q[:1], stored, 45)
    assert deep["ids"][0] == [ids[r] for r in er_d[0]]

    # delete one document (stable compaction) and search again
    gone = idx.delete(where={"doc_id": ids[300][:16]})
    assert len(gone) == 100 and idx.count() == 1400
    keep = np.array([i for i in range(1500) if not (300 <= i < 400)])
    res2 = idx.query(q.tolist(), n_results=5)
    _, er2 = O.cosine_topk(q.astype(np.float16).astype(np.float32) if dtype == torch.float16 else q, stored[keep], 5)
    assert [[ids[keep[r]] for r in row] for row in er2] == res2["ids"]
    assert idx.query(q.tolist(), n_results=20)["ids"][0][0] in ids
    idx.reset()
    assert idx.query(q.tolist(), n_results=5)["ids"] == [[] for _ in range(9)]


def test_embedding_manager_on_hip_engine(gpu, golden_dir):
    from multimodal_rag_amd import ingest
    from multimodal_rag_amd.embedder import EmbeddingManager

    m = EmbeddingManager(batch_size=32, enable_cache=True)
    run(m.initialize())
    assert m.device == "cuda" and m.get_embedding_dimension() == 384
    text = open(os.path.join(golden_dir, "sample_document.txt"), encoding="utf-8").read()
    corpus = ingest.basic_chunk_text(text) + [f"Paragraph {i} about topic {i % 7}: " + "word " * (5 + i % 40) for i in range(120)]
    summ = [{"id": f"text_{i}", "summary": t, "raw": t, "type": "text"} for i, t in enumerate(corpus)]
    assert run(m.embed_and_store(summ, "doc_0123456789ab")) == {"text": 121, "table": 0, "image": 0}
    embs = np.asarray(run(m.embed_texts_batch(corpus)), np.float32)        # all cache hits
    assert m.cache.get_stats()["hits"] >= 121 - 32 and np.allclose(np.linalg.norm(embs, axis=1), 1, atol=1e-4)

    queries = [corpus[0][:80], corpus[17], "Paragraph 33 about topic 5"]
    for qtext in queries:
        r = run(m.query(qtext, n_results=5))
        qe = np.asarray(run(m.embed_texts_batch([qtext])), np.float32)
        stored = embs.astype(np.float16).astype(np.float32)               # index stores fp16
        es, er = O.cosine_topk(qe.astype(np.float16).astype(np.float32), stored, 5)
        got_rows = np.array([[int(i.rsplit("_", 1)[1]) for i in r["ids"]]])
        assert O.same_topk_sets(got_rows, 1 - np.array([r["distances"]]), er, es)
        assert np.allclose(1 - np.array(r["distances"]), es[0], atol=1e-4)
    assert run(m.query(corpus[17]))["ids"][0] == "doc_0123456789ab_text_17"
    out = run(m.batch_query([corpus[3], " ", corpus[90]], n_results=3))
    assert out[0]["ids"][0].endswith("text_3") and "error" in out[1] and out[2]["ids"][0].endswith("text_90")
    sim = run(m.get_similar_documents("doc_0123456789ab", "text_5", n_results=20))   # k+1 = 21 > MAX_K
    assert len(sim["ids"]) == 20 and "doc_0123456789ab_text_5" not in sim["ids"]
    run(m.delete_document("doc_0123456789ab"))
    assert run(m.get_collection_stats())["count"] == 0
    run(m.cleanup())


def test_vector_index_under_concurrent_callers(gpu):
    """the reference calls its engines from asyncio.to_thread workers (embedder.py:368, 517, 595) and batch_query
    makes those calls concurrent: queries racing adds must each see a consistent index (every answer is the exact
    top-k of SOME prefix of the inserts) and nothing may be lost"""
    import threading

    from multimodal_rag_amd.index import VectorIndex

    d, n_batches, per = 128, 24, 250
    V = unit(n_batches * per, d, 5)
    q = unit(6, d, 6)
    idx = VectorIndex(d, capacity=256)
    idx.add(V[:per].tolist(), [f"d{i}" for i in range(per)], [{"type": "text"}] * per, [f"id{i}" for i in range(per)])
    answers, errors = [], []

    def writer():
        try:
            for b in range(1, n_batches):
                lo = b * per
                idx.add(V[lo:lo + per].tolist(), [f"d{i}" for i in range(lo, lo + per)], [{"type": "text"}] * per,
                        [f"id{i}" for i in range(lo, lo + per)])
        except Exception as e:  # pragma: no cover
            errors.append(e)

    def reader():
        try:
            for _ in range(40):
                r = idx.query(q.tolist(), n_results=5)
                answers.append((idx.count(), r["ids"], r["distances"]))
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=writer)] + [threading.Thread(target=reader) for _ in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert idx.count() == n_batches * per
    stored = V.astype(np.float16).astype(np.float32)
    qs = q.astype(np.float16).astype(np.float32)
    prefix_answers = {}
    for m in range(per, n_batches * per + 1, per):
        es, er = O.cosine_topk(qs, stored[:m], 5)
        prefix_answers[m] = [[f"id{r}" for r in row] for row in er]
    for count_after, ids, dist in answers:
        assert any(ids == prefix_answers[m] for m in prefix_answers if m <= count_after), "answer matches no insert prefix"
        assert all(a <= b + 1e-7 for row in dist for a, b in zip(row, row[1:]))
    final = idx.query(q.tolist(), n_results=5)
    assert final["ids"] == prefix_answers[n_batches * per]


def test_hip_engine_from_local_checkpoint_dir(gpu, tmp_path, monkeypatch):
    """MMRAG_MODEL_DIR: config.json + model.safetensors + vocab.txt (+ sentence-transformers side files) -> native
    tokenizer + DeviceEncoder; embeddings must equal the oracle fed by the Python tokenizer."""
    import json

    from safetensors.numpy import save_file

    from multimodal_rag_amd import embedder as EM
    from multimodal_rag_amd.tokenizer import WordPieceTokenizer
    from oracle import encoder_oracle as E
    from tests.test_tokenizer import VOCAB

    shape = E.BertShape(2, 128, 4, 256, len(VOCAB), 64, 1e-12, "mean")
    w = E.make_bert_weights(shape, 3)
    d = tmp_path / "tiny-bert"
    (d / "1_Pooling").mkdir(parents=True)
    save_file({"bert." + k: v for k, v in w.items()}, str(d / "model.safetensors"))
    json.dump({"model_type": "bert", "num_hidden_layers": 2, "hidden_size": 128, "num_attention_heads": 4,
               "intermediate_size": 256, "vocab_size": len(VOCAB), "max_position_embeddings": 64,
               "layer_norm_eps": 1e-12}, open(d / "config.json", "w"))
    json.dump({"pooling_mode_mean_tokens": True, "pooling_mode_cls_token": False}, open(d / "1_Pooling" / "config.json", "w"))
    json.dump({"max_seq_length": 48}, open(d / "sentence_bert_config.json", "w"))
    (d / "vocab.txt").write_text("\n".join(VOCAB) + "\n", encoding="utf-8")
    monkeypatch.setattr(EM.settings, "MMRAG_MODEL_DIR", str(d))

    eng = EM.HipEngine("whatever-name")
    assert type(eng.tokenizer).__name__ == "NativeWordPieceTokenizer" and eng.max_seq_length == 48 and eng.dim == 128
    texts = ["Machine learning là gì?", "The quick brown fox runs.", "unaffable running, naïve café! " * 20, "你好 state-of-the-art"]
    got = eng.encode(texts)
    py = WordPieceTokenizer({t: i for i, t in enumerate(VOCAB)})
    want = E.bert_encode(shape, E.round_weights_fp16(w), [py.encode(t, 48) for t in texts])
    assert np.abs(got - want).max() <= 4e-3 and (got * want).sum(1).min() >= 0.9999
    # the list path and the array path of the encoder agree bit for bit
    ids, lens = eng.tokenizer.encode_batch_arrays(texts, 48)
    a = eng.encoder.encode_id_rows(ids, lens).cpu().numpy()
    b = eng.encoder.encode_ids([ids[i, : lens[i]].tolist() for i in range(len(texts))]).cpu().numpy()
    assert np.array_equal(a, b)


def test_clip_engine_joint_space(gpu, golden_dir, tmp_path):
    """BASELINE config 4: text chunks and images of arbitrary size in ONE index (CLIP ViT-B/32 shape, random
    weights); vectors must equal the towers driven directly, and search over the mixed index matches the oracle."""
    from PIL import Image

    from multimodal_rag_amd.embedder import ClipEngine, EmbeddingManager
    from tests.golden.make_resize_golden import formula_image

    eng = ClipEngine("openai/clip-vit-base-patch32")
    m = EmbeddingManager(engine=eng)
    run(m.initialize())
    assert m.get_embedding_dimension() == 512
    sizes = [(300, 420), (640, 480), (224, 224), (90, 700)]
    items = [{"id": f"text_{i}", "summary": f"chunk {i} about topic {i % 5}", "raw": "", "type": "text"} for i in range(40)]
    for j, (H, W) in enumerate(sizes):
        p = tmp_path / f"img{j}.png"
        Image.fromarray(formula_image(H, W)).save(p)
        items.append({"id": f"image_{j}", "summary": f"figure {j}", "raw": "", "type": "image", "path": str(p)})
    assert run(m.embed_and_store(items, "doc_c11bc11bc11b")) == {"text": 40, "table": 0, "image": 4}

    ids = [f"doc_c11bc11bc11b_{it['id']}" for it in items]
    stored = np.asarray(m.collection.get(ids=ids, include=["embeddings"])["embeddings"], np.float32)
    want_img = eng.clip.encode_images(eng.preprocess([formula_image(H, W) for H, W in sizes])).cpu().numpy()
    want_txt = eng.encode([it["summary"] for it in items[:40]])
    assert np.abs(stored[40:] - want_img).max() <= 1e-3 and np.abs(stored[:40] - want_txt).max() <= 1e-3
    assert np.allclose(np.linalg.norm(stored, axis=1), 1, atol=2e-3)

    q = "figure 2"
    r = run(m.query(q, n_results=5))
    qe = np.asarray(run(m.embed_texts_batch([q])), np.float32).astype(np.float16).astype(np.float32)
    es, er = O.cosine_topk(qe, stored, 5)
    assert O.same_topk_sets(np.array([[ids.index(i) for i in r["ids"]]]), 1 - np.array([r["distances"]]), er, es)
    only_img = run(m.query(q, n_results=3, filter_dict={"type": "image"}))
    assert all(md["type"] == "image" for md in only_img["metadatas"])
    run(m.cleanup())


def test_fastapi_surface_on_gpu(gpu, golden_dir):
    from starlette.testclient import TestClient

    from multimodal_rag_amd.server import create_app

    with TestClient(create_app()) as client:
        data = open(os.path.join(golden_dir, "sample_document.txt"), "rb").read()
        up = client.post("/upload", files={"file": ("sample_document.txt", data, "text/plain")}).json()
        assert up["chunks_processed"] == {"text": 1, "table": 0, "image": 0}        # BASELINE config 1
        q = client.post("/query", json={"query": "Machine Learning", "top_k": 5}).json()
        assert len(q["sources"]) == 1 and q["sources"][0]["doc_id"] == up["doc_id"] + "_text_0"
        assert 0 <= q["sources"][0]["relevance_score"] <= 1
        assert client.get("/health").json()["components"]["embedder"]["documents"] == 1
