"""A/B kernel variants interleaved in ONE process (devices differ by >10%: never compare across runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_rag_amd import _native as N
B = int(os.environ.get("PB", "256")); n = int(os.environ.get("PN", "1000000")); d = 768; k = 5; dtype = torch.float16
variants = [int(x) for x in os.environ.get("VARIANTS", "0,1,2,3,4,7").split(",")]
ld = N.padded_dim(d, dtype)
c = torch.randn((n, ld), device="cuda"); c = (c / c.norm(dim=1, keepdim=True)).to(dtype)
q = torch.randn((B, ld), device="cuda"); q = (q / q.norm(dim=1, keepdim=True)).to(dtype)
ws = torch.empty(N.cosine_topk_workspace_bytes(B, n, k) + 16, dtype=torch.uint8, device="cuda")
def run(v, iters=10):
    os.environ["MMRAG_VARIANT"] = str(v)
    N.cosine_topk_lists(q, c, n, d, k, ws)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): N.cosine_topk_lists(q, c, n, d, k, ws)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for v in variants: run(v, 3)
res = {v: [] for v in variants}
for rnd in range(6):
    for v in variants: res[v].append(run(v))
for v in variants:
    r = sorted(res[v]); print(f"variant {v}: median {r[len(r)//2]:.1f} us  min {r[0]:.1f}  max {r[-1]:.1f}")
