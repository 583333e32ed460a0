import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_rag_amd import _native as N
B = 256; n = 1_000_000; d = 768; k = 5; dtype = torch.float16
ld = N.padded_dim(d, dtype)
c = torch.randn((n, ld), device="cuda"); c = (c / c.norm(dim=1, keepdim=True)).to(dtype)
q = torch.randn((B, ld), device="cuda"); q = (q / q.norm(dim=1, keepdim=True)).to(dtype)
ws = torch.empty(2 * N.cosine_topk_workspace_bytes(B, n, k) + 16, dtype=torch.uint8, device="cuda")
for nw in (16, 8):
    if nw == 8: os.environ["MMRAG_NW8"] = "1"
    os.environ.pop("MMRAG_PROF_PTR", None)
    prof = torch.zeros((256, nw, 4), dtype=torch.int64, device="cuda")
    for _ in range(3): N.cosine_topk_lists(q, c, n, d, k, ws)
    torch.cuda.synchronize()
    os.environ["MMRAG_PROF_PTR"] = str(prof.data_ptr())
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); N.cosine_topk_lists(q, c, n, d, k, ws); e1.record(); torch.cuda.synchronize()
    p = prof.cpu().double(); tot = p.sum(-1)
    print(f"NW={nw}: kernel {e0.elapsed_time(e1)*1e3:.0f} us; per-wave cycles mean {tot.mean():.0f}")
    for i, nm in enumerate(["vmcnt wait", "barrier", "compute", "epilogue"]):
        print("   %-10s mean %9.0f (%.1f%%)  min %9.0f max %9.0f" % (nm, p[..., i].mean(), 100*p[..., i].mean()/tot.mean(), p[..., i].min(), p[..., i].max()))
    print("   by wave:", [[int(x) for x in p[:, w, :].mean(0)] for w in range(0, nw, max(1, nw // 4))])
