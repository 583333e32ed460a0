import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_rag_amd import _native as N
B = int(os.environ.get("PB", "256")); n = 1_000_000; d = 768; k = 5; dtype = torch.float16
ld = N.padded_dim(d, dtype)
c = torch.randn((n, ld), device="cuda"); c = (c / c.norm(dim=1, keepdim=True)).to(dtype)
q = torch.randn((B, ld), device="cuda"); q = (q / q.norm(dim=1, keepdim=True)).to(dtype)
ws = torch.empty(N.cosine_topk_workspace_bytes(B, n, k) + 16, dtype=torch.uint8, device="cuda")
prof = torch.zeros((256, 8, 4), dtype=torch.int64, device="cuda")
for _ in range(3): N.cosine_topk_lists(q, c, n, d, k, ws)
torch.cuda.synchronize()
os.environ["MMRAG_PROF_PTR"] = str(prof.data_ptr())
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); N.cosine_topk_lists(q, c, n, d, k, ws); e1.record(); torch.cuda.synchronize()
p = prof.cpu().double()
tot = p.sum(-1)
print("kernel us", e0.elapsed_time(e1) * 1e3)
print("per-wave total cycles: mean %.0f min %.0f max %.0f" % (tot.mean(), tot.min(), tot.max()))
names = ["vmcnt wait", "barrier", "compute(frag+mfma+dma issue)", "epilogue"]
for i, nm in enumerate(names):
    print("%-30s mean %10.0f  (%.1f%%)   min %10.0f max %10.0f" % (nm, p[..., i].mean(), 100 * p[..., i].mean() / tot.mean(), p[..., i].min(), p[..., i].max()))
print("by wave (mean over WGs):")
for w in range(8): print(w, [int(x) for x in p[:, w, :].mean(0)])
