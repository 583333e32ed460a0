"""Developer check: mmrag_linear_f16 variants (debug switches) vs torch.matmul (hipBLASLt) on the encoder GEMM shapes,
interleaved in one process."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_rag_amd import _native as N
L = N.lib()
L.mmrag_internal_set_debug.argtypes = [ctypes.c_uint]
VARIANTS = {"default": 0, "plain": 1}
def t(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
M = 65536
for K, Nf in [(768, 2304), (768, 768), (768, 3072), (3072, 768)]:
    x = (torch.randn((M, K), device="cuda") * 0.5).half(); w = (torch.randn((Nf, K), device="cuda") * 0.05).half()
    b = torch.randn(Nf, device="cuda"); out = torch.empty((M, Nf), dtype=torch.float16, device="cuda")
    fl = 2.0 * M * K * Nf
    res, ref_out = {}, None
    for rnd in range(4):
        for name, flag in VARIANTS.items():
            L.mmrag_internal_set_debug(flag)
            res.setdefault(name, []).append(t(lambda: N.linear_f16(x, w, b, 0, None, out)))
            if ref_out is None: ref_out = out.clone()
            else: assert torch.equal(ref_out, out), name
        L.mmrag_internal_set_debug(0)
        res.setdefault("hipBLASLt", []).append(t(lambda: torch.matmul(x, w.t())))
    line = " | ".join(f"{k} {sorted(v)[len(v)//2]:.1f} us {fl/sorted(v)[len(v)//2]/1e6:.0f} TF" for k, v in res.items())
    print(f"K={K} N={Nf}: {line}", flush=True)
