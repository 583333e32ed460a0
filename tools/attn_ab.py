"""Developer A/B of mmrag_attention_f16 between the built library and another build of it (path in argv[1]), bge shape,
interleaved in one process, 0.2 s per measurement."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_rag_amd import _native as N
B, S, H, nh = 256, int(os.environ.get("S", "256")), int(os.environ.get("H", "768")), 12
qkv = (torch.randn((B * S, 3 * H), device="cuda") * 0.5).half()
cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device="cuda")
ctx = torch.empty((B * S, H), dtype=torch.float16, device="cuda")
libs = {"built": N.lib()}
if len(sys.argv) > 1:
    libs["other"] = ctypes.CDLL(sys.argv[1])
def run(lib):
    st = lib.mmrag_attention_f16(ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(cu.data_ptr()), ctypes.c_void_p(ctx.data_ptr()),
                                 B, S, H, nh, 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0
def t(lib, iters=1500):
    run(lib); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run(lib)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
outs = {}
for rnd in range(3):
    for name, lib in libs.items():
        us = t(lib)
        outs.setdefault(name, ctx.clone())
        print(f"{name}: {us:.1f} us  {4.0 * S * H * B * S / us / 1e6:.0f} TFLOP/s", flush=True)
if len(outs) == 2:
    print("same bits:", torch.equal(outs["built"], outs["other"]))
