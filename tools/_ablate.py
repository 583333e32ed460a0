import os, sys
sys.path.insert(0, "/root/repo")
import torch
from multimodal_rag_amd import _native as N
def bench(B, n, d, dtype, k=5, iters=20):
    ld = N.padded_dim(d, dtype)
    c = torch.randn((n, ld), device="cuda").to(dtype)
    q = torch.randn((B, ld), device="cuda").to(dtype)
    ws = torch.empty(N.cosine_topk_workspace_bytes(B, n, k) + 16, dtype=torch.uint8, device="cuda")
    for _ in range(3): N.cosine_topk_lists(q, c, n, d, k, ws)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): N.cosine_topk_lists(q, c, n, d, k, ws)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
print("dbg", os.environ.get("MMRAG_DBG"), "B=256: %.1f us" % bench(256, 1_000_000, 768, torch.float16), " B=128: %.1f us" % bench(128, 1_000_000, 768, torch.float16))
