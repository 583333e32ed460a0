"""The encoder half of one served batch_query call (256 queries of 8 words: ~2.5 k tokens), 40 times: for
rocprofv3 --kernel-trace --stats (which kernels the mid-size forward spends its time in) and a device-time figure."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_embed import synthetic_vocab
from multimodal_rag_amd.embedder import HipEngine
from multimodal_rag_amd.tokenizer import NativeWordPieceTokenizer
eng = HipEngine("BAAI/bge-base-en-v1.5", "cuda:0")
vocab, words = synthetic_vocab(eng.encoder.cfg.vocab)
eng.tokenizer = NativeWordPieceTokenizer(vocab)
texts = [" ".join(words[(i * 7 + j * 131) % len(words)] for j in range(7)) + f" {i % 97}" for i in range(256)]
out = eng.encode_device(texts); torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
iters = int(os.environ.get("ITERS", "40"))
e0.record()
for _ in range(iters): out = eng.encode_device(texts)
e1.record(); torch.cuda.synchronize()
rows = eng.tokenizer.encode_batch_arrays(texts, eng.max_seq_length); toks = int(rows[1].sum())
print(f"encode_device(256 queries, {toks} tokens): {e0.elapsed_time(e1) / iters:.3f} ms per call (wall incl. host enqueue)")
