"""Developer micro-benchmark of mmrag_attention_f16 at the bge-base shape (256 seqs x 256 tokens, 12 heads x 64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_rag_amd import _native as N
B, S, H, nh = 256, 256, 768, 12
qkv = (torch.randn((B * S, 3 * H), device="cuda") * 0.5).half()
cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device="cuda")
def t(iters=20):
    global ctx
    ctx = N.attention_f16(qkv, cu, S, nh, False); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): N.attention_f16(qkv, cu, S, nh, False)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
fl = 4.0 * S * H * B * S
ref = None
for v in os.environ.get("VARIANTS", "1,3,4").split(","):
    os.environ["MMRAG_ATTN_OCC"] = v
    us = sorted(t() for _ in range(3))[1]
    same = "" if ref is None else (" same" if torch.equal(ref, ctx) else " DIFF")
    if ref is None: ref = ctx.clone()
    print(f"occ {v}: {us:.1f} us  {fl/us/1e6:.0f} TFLOP/s{same}", flush=True)
