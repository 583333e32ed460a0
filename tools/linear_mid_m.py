"""Developer check: mmrag_linear_f16 at a few hundred to a few thousand token rows (dispatcher batches of queries):
64x64 tiles (switch 1024) against 128x128 tiles (switch 2048), interleaved in one process."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_rag_amd import _native as N
L = N.lib(); L.mmrag_internal_set_debug.argtypes = [ctypes.c_uint]
def t(fn, iters=400):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for M in (160, 640, 1280, 2560, 5120):
    line = f"M={M}:"
    for K, Nf in [(768, 2304), (768, 768), (768, 3072), (3072, 768), (384, 1152), (1536, 384)]:
        x = (torch.randn((M, K), device="cuda") * 0.5).half(); w = (torch.randn((Nf, K), device="cuda") * 0.05).half()
        b = torch.randn(Nf, device="cuda"); out = torch.empty((M, Nf), dtype=torch.float16, device="cuda")
        r = {}
        for name, flag in (("64", 1024), ("128", 2048), ("auto", 0)):
            L.mmrag_internal_set_debug(flag); r[name] = t(lambda: N.linear_f16(x, w, b, 0, None, out))
        L.mmrag_internal_set_debug(0)
        line += f"  K{K}N{Nf} 64:{r['64']:.1f} 128:{r['128']:.1f} auto:{r['auto']:.1f}"
    print(line, flush=True)
