"""Developer check: EmbeddingManager.batch_query (256 queries per call, 1M x 768 index) from 1..4 concurrent callers:
queries/s and the mean time of every stage, to see what the callers contend for."""
import asyncio, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_embed import synthetic_vocab
from multimodal_rag_amd import tracing
from multimodal_rag_amd.embedder import EmbeddingManager, HipEngine
from multimodal_rag_amd.tokenizer import NativeWordPieceTokenizer
n_rows = int(os.environ.get("ROWS", "1000000"))
if os.environ.get("SWITCH"): sys.setswitchinterval(float(os.environ["SWITCH"]))   # (the interpreter's default is 5 ms)
print("switch interval", sys.getswitchinterval())
dev = torch.device("cuda:0")
g = torch.Generator(device="cuda").manual_seed(1)
corpus = torch.empty((n_rows, 768), dtype=torch.float16, device=dev)
for lo in range(0, n_rows, 1 << 18):
    x = torch.randn((min(n_rows, lo + (1 << 18)) - lo, 768), device=dev, generator=g); corpus[lo:lo + x.shape[0]] = (x / x.norm(dim=1, keepdim=True)).half()
eng = HipEngine("BAAI/bge-base-en-v1.5", "cuda:0")
vocab, words = synthetic_vocab(eng.encoder.cfg.vocab)
eng.tokenizer = NativeWordPieceTokenizer(vocab)
m = EmbeddingManager(engine=eng, enable_cache=False)
import ctypes
from multimodal_rag_amd import _native, encoder as _enc
_L = _native.lib(); _L.mmrag_internal_last_forward_us.restype = ctypes.c_longlong
_inside = []
_orig = _native.encoder_forward
def _timed(*a, **k):
    r = _orig(*a, **k); _inside.append(_L.mmrag_internal_last_forward_us()); return r
_native.encoder_forward = _timed
async def go():
    await m.initialize()
    ids = [f"doc_{i // 64:012x}_text_{i % 64}" for i in range(n_rows)]
    metas = [{"doc_id": s[:16], "item_id": s[17:], "type": "text"} for s in ids]
    m.collection.add_rows_device(corpus, None, metas, ids)
    texts = [" ".join(words[(i * 7 + j * 131) % len(words)] for j in range(7)) + f" {i % 97}" for i in range(4096)]
    await m.batch_query(texts[:256], n_results=5)
    for callers in (1, 2, 3, 4):
        tracing.reset(); _inside.clear()
        tracing.timeline = None
        t_end, t0 = time.perf_counter() + 2.0, time.perf_counter()
        done = [0]
        async def caller(j):
            k = (j * 1280) % 3840
            while time.perf_counter() < t_end:
                res = await m.batch_query(texts[k:k + 256], n_results=5)
                assert len(res) == 256 and len(res[0]["ids"]) == 5
                done[0] += 256; k = (k + 256) % 3840
        await asyncio.gather(*[caller(j) for j in range(callers)])
        dt = time.perf_counter() - t0
        st = {k: v["mean_ms"] for k, v in tracing.snapshot().items()}
        print(f"{callers} callers: {done[0] / dt:9.0f} queries/s   launches inside the library: mean {sum(_inside) / max(1, len(_inside)) / 1e3:.3f} ms   stages (mean ms): {st}", flush=True)
        if tracing.timeline:
            tl = sorted(tracing.timeline, key=lambda e: e[2]); base = tl[len(tl) // 2][2]; tids = {}
            for tid, name, a, b in tl[len(tl) // 2: len(tl) // 2 + 60]:
                col = tids.setdefault(tid, len(tids))
                print(f"   {'                                  ' * col}[t{col}] {name:12s} {1e3 * (a - base):7.2f} -> {1e3 * (b - base):7.2f} ms")
            tracing.timeline = None
asyncio.run(go())
