"""Single-query latency through EmbeddingManager (all-MiniLM-L6-v2 shape, random weights) and its parts."""
import asyncio, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multimodal_rag_amd.embedder import EmbeddingManager

async def main():
    if os.environ.get("MMRAG_DEBUG"):   # developer switches of the library (csrc/mmrag_internal.h), e.g. 256 = 32-feature small-M GEMM
        import ctypes
        from multimodal_rag_amd import _native
        _native.lib().mmrag_internal_set_debug(ctypes.c_uint(int(os.environ["MMRAG_DEBUG"])))
    m = EmbeddingManager(batch_size=32, enable_cache=False)
    await m.initialize()
    if os.environ.get("TOKENIZER", "native") == "native":
        # a deployment has vocab.txt and therefore the native WordPiece tokenizer; without a checkpoint the engine falls back to
        # the pure-Python hash tokenizer (TOKENIZER=hash keeps it).  Same synthetic vocabulary as bench.py's served leg.
        from bench_embed import synthetic_vocab
        from multimodal_rag_amd.tokenizer import NativeWordPieceTokenizer
        vocab, words = synthetic_vocab(m._engine.encoder.cfg.vocab)
        m._engine.tokenizer = NativeWordPieceTokenizer(vocab)
    n = int(os.environ.get("ROWS", "100000"))
    g = np.random.default_rng(0)
    v = g.standard_normal((n, 384)).astype(np.float32); v /= np.linalg.norm(v, axis=1, keepdims=True)
    for lo in range(0, n, 20000):
        m.collection.add(v[lo:lo + 20000], documents=[f"d{i}" for i in range(lo, lo + 20000)],
                         metadatas=[{"type": "text"}] * 20000, ids=[f"doc_000000000000_t{i}" for i in range(lo, lo + 20000)])
    qs = [f"what is the meaning of topic number {i} in the retrieval pipeline" for i in range(60)]
    if os.environ.get("TOKENIZER", "native") == "native":
        qs = [" ".join(words[(i * 13 + j * 101) % len(words)] for j in range(11)) for i in range(60)]
    for q in qs[:10]: await m.query(q)
    from multimodal_rag_amd import tracing
    tracing.reset()
    t0 = time.perf_counter()
    for q in qs[10:]: await m.query(q)
    dt = (time.perf_counter() - t0) / 50
    stages = {k: v["mean_ms"] for k, v in m.get_stage_timers().items()}
    eng = m._engine
    t0 = time.perf_counter()
    for q in qs[10:]: e = eng.encode([q])
    te = (time.perf_counter() - t0) / 50
    t0 = time.perf_counter()
    for q in qs[10:]: ids = eng.tokenizer.encode(q, 256)
    tt = (time.perf_counter() - t0) / 50
    t0 = time.perf_counter()
    for _ in range(50): r = m.collection.query([e[0].tolist()], n_results=5)
    ts = (time.perf_counter() - t0) / 50
    print("stage means (ms) inside query():", stages)
    print(f"rows {n}: query() {dt*1e3:.2f} ms | encode {te*1e3:.2f} ms (tokenize {tt*1e3:.3f}) | collection.query {ts*1e3:.2f} ms")
    res = await m.batch_query(qs[:32]); t0 = time.perf_counter(); res = await m.batch_query(qs[:32]); tb = time.perf_counter() - t0
    print(f"batch_query(32): {tb*1e3:.2f} ms total")
asyncio.run(main())
