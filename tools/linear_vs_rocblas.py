"""Developer check: mmrag_linear_f16 variants (debug switches) vs torch.matmul (hipBLASLt) on the encoder GEMM shapes,
interleaved in one process."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_rag_amd import _native as N
L = N.lib()
L.mmrag_internal_set_debug.argtypes = [ctypes.c_uint]
VARIANTS = {"default": 0, "mfma32": 8192, "one-tile-per-wg": 4, "no-epilogue": 16, "mfma32+no-epilogue": 8192 + 16}
CHECKED = ("default", "mfma32", "one-tile-per-wg")
def t(fn, iters=800):   # ~0.2 s per measurement: the board throttles after a few ms of this load
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
M = 65536
for K, Nf in [(768, 2304), (768, 768), (768, 3072), (3072, 768), (1536, 768)]:
    x = (torch.randn((M, K), device="cuda") * 0.5).half(); w = (torch.randn((Nf, K), device="cuda") * 0.05).half()
    b = torch.randn(Nf, device="cuda"); out = torch.empty((M, Nf), dtype=torch.float16, device="cuda")
    fl = 2.0 * M * K * Nf
    res, ref_out = {}, None
    for rnd in range(4):
        for name, flag in VARIANTS.items():
            L.mmrag_internal_set_debug(flag)
            res.setdefault(name, []).append(t(lambda: N.linear_f16(x, w, b, 0, None, out)))
            if name in CHECKED:
                if ref_out is None: ref_out = out.clone()
                else: assert (ref_out.float() - out.float()).abs().max().item() < 0.02, name   # (K is walked from different slabs)
        L.mmrag_internal_set_debug(0)
        res.setdefault("hipBLASLt", []).append(t(lambda: torch.matmul(x, w.t())))
    line = " | ".join(f"{k} {sorted(v)[len(v)//2]:.1f} us {fl/sorted(v)[len(v)//2]/1e6:.0f} TF" for k, v in res.items())
    print(f"K={K} N={Nf}: {line}", flush=True)
