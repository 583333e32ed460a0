"""Developer check: all-MiniLM-L6-v2 shape (configs 1-2), 256 chunks x 256 tokens per step, chunks/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multimodal_rag_amd.encoder import PRESETS, DeviceEncoder, random_bert_weights
cfg = PRESETS["sentence-transformers/all-MiniLM-L6-v2"] if "sentence-transformers/all-MiniLM-L6-v2" in PRESETS else PRESETS[sorted(PRESETS)[0]]
print(cfg)
dev = torch.device("cuda:0")
enc = DeviceEncoder(cfg, random_bert_weights(cfg, seed=1, device=dev), dev)
B, S = 256, 256
g = np.random.default_rng(0)
ids = torch.from_numpy(g.integers(1000, cfg.vocab, size=B * S).astype(np.int32)).to(dev)
pos = torch.arange(S, dtype=torch.int32, device=dev).repeat(B)
cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device=dev)
out = torch.empty((B, cfg.dim), dtype=torch.float32, device=dev)
for _ in range(3): enc.forward_packed(ids, pos, cu, S, out=out)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): enc.forward_packed(ids, pos, cu, S, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"MiniLM-L6 shape: {ms:.3f} ms/step, {B/ms*1e3:.0f} chunks/s, {B*enc.flops_per_sequence(S)/ms/1e9:.0f} TFLOP/s  checksum {float(out.sum()):.6f}")
