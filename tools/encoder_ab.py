"""Developer A/B of the BERT forward at the bench shape (bge-base, 256 chunks x 256 tokens): deferred LayerNorm (default)
against LayerNorm passes (debug switch 512), interleaved in one process, about 0.5 s per measurement."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multimodal_rag_amd import _native as N
from multimodal_rag_amd.encoder import PRESETS, DeviceEncoder, random_bert_weights
import bench_embed
L = N.lib(); L.mmrag_internal_set_debug.argtypes = [ctypes.c_uint]
dev = torch.device("cuda:0")
cfg = PRESETS["BAAI/bge-base-en-v1.5"]
enc = DeviceEncoder(cfg, random_bert_weights(cfg, seed=4321, device=dev), dev)
S, C = int(os.environ.get("S", "256")), int(os.environ.get("CHUNKS", "256"))
if os.environ.get("NO_ACT"): enc.desc.act = 0   # timing only: GEMMs without the GELU
ids = torch.from_numpy(bench_embed.synthetic_ids(C, S, cfg.vocab, 100).reshape(-1)).to(dev)
pos = torch.arange(S, dtype=torch.int32, device=dev).repeat(C)
cu = torch.arange(0, (C + 1) * S, S, dtype=torch.int32, device=dev)
out = torch.empty((C, cfg.dim), dtype=torch.float32, device=dev)
def t(iters=30):
    enc.forward_packed(ids, pos, cu, S, out=out); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): enc.forward_packed(ids, pos, cu, S, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
outs = {}
for rnd in range(3):
    for name, flag in [v for v in (("deferred LayerNorm", 0), ("LayerNorm passes", 512)) if os.environ.get("ONLY", str(v[1])) == str(v[1])]:
        L.mmrag_internal_set_debug(flag)
        ms = t()
        outs[name] = out.clone()
        print(f"{name}: {ms:.3f} ms/step  {C / ms * 1e3:.0f} chunks/s", flush=True)
L.mmrag_internal_set_debug(0)
if len(outs) == 2:
    a, b = outs["deferred LayerNorm"], outs["LayerNorm passes"]
    print("max |d embedding|:", float((a - b).abs().max()), " min cosine:", float((a * b).sum(1).min()))
