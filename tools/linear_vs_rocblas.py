"""Developer check: mmrag_linear_f16 vs torch.matmul (hipBLASLt) on the encoder GEMM shapes, one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_rag_amd import _native as N
def t(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
M = 65536
for K, Nf in [(768, 2304), (768, 768), (768, 3072), (3072, 768), (8192, 2304), (8192, 768)]:
    x = (torch.randn((M, K), device="cuda") * 0.5).half(); w = (torch.randn((Nf, K), device="cuda") * 0.05).half()
    b = torch.randn(Nf, device="cuda"); out = torch.empty((M, Nf), dtype=torch.float16, device="cuda")
    fl = 2.0 * M * K * Nf
    os.environ["MMRAG_LINEAR_PLAIN"] = "1"
    plain = sorted(t(lambda: N.linear_f16(x, w, b, 0, None, out)) for _ in range(3))[1]
    o0 = out.clone()
    os.environ.pop("MMRAG_LINEAR_PLAIN", None)
    mine = sorted(t(lambda: N.linear_f16(x, w, b, 0, None, out)) for _ in range(3))[1]
    same = f"plain {plain:.1f} us {fl/plain/1e6:.0f} TF ({'same bits' if torch.equal(o0, out) else 'DIFF'})"
    ref = sorted(t(lambda: torch.matmul(x, w.t())) for _ in range(3))[1]
    print(f"K={K} N={Nf}: mine {mine:.1f} us {fl/mine/1e6:.0f} TF | {same} | rocBLAS {ref:.1f} us {fl/ref/1e6:.0f} TF", flush=True)
