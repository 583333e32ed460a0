import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_rag_amd import _native as N
B = int(os.environ.get("PB", "256")); n = 1_000_000; d = 768; k = 5; dtype = torch.float16
ld = N.padded_dim(d, dtype)
c = torch.randn((n, ld), device="cuda"); c = (c / c.norm(dim=1, keepdim=True)).to(dtype)
q = torch.randn((B, ld), device="cuda"); q = (q / q.norm(dim=1, keepdim=True)).to(dtype)
ws = torch.empty(N.cosine_topk_workspace_bytes(B, n, k) + 16, dtype=torch.uint8, device="cuda")
for _ in range(6): N.cosine_topk_lists(q, c, n, d, k, ws)
torch.cuda.synchronize()
