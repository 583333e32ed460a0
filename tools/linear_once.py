"""Developer helper for profiler passes: `python tools/linear_once.py K N [M] [launches]` runs mmrag_linear_f16 on one shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_rag_amd import _native as N
K, Nf = int(sys.argv[1]), int(sys.argv[2])
M = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
n = int(sys.argv[4]) if len(sys.argv) > 4 else 20
x = (torch.randn((M, K), device="cuda") * 0.5).half(); w = (torch.randn((Nf, K), device="cuda") * 0.05).half()
b = torch.randn(Nf, device="cuda"); out = torch.empty((M, Nf), dtype=torch.float16, device="cuda")
for _ in range(n): N.linear_f16(x, w, b, 0, None, out)
torch.cuda.synchronize()
print("done", M, K, Nf)
