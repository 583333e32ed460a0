"""Developer micro-benchmark of mmrag_cosine_topk (not the judged bench.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_rag_amd import _native as N

def bench(B, n, d, dtype, k=5, iters=20, dbg=0):
    ld = N.padded_dim(d, dtype)
    g = torch.Generator(device="cuda").manual_seed(1)
    c = torch.empty((n, ld), dtype=dtype, device="cuda")
    step = 1 << 18
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        x = torch.randn((hi - lo, ld), device="cuda", generator=g)
        x[:, d:] = 0
        x /= x.norm(dim=1, keepdim=True)
        c[lo:hi] = x.to(dtype)
    q = torch.randn((B, ld), device="cuda", generator=g); q[:, d:] = 0; q /= q.norm(dim=1, keepdim=True); q = q.to(dtype)
    ws = torch.empty(N.cosine_topk_workspace_bytes(B, n, k) + 16, dtype=torch.uint8, device="cuda")
    t_end = time.time() + 0.25            # steady state: hold the kernel for a while before timing it
    while time.time() < t_end:
        for _ in range(10): N.cosine_topk(q, c, n, d, k, workspace=ws, dbg=dbg)
        torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    iters = 0; e0.record(); t_end = time.time() + 0.25
    while time.time() < t_end:
        for _ in range(10): N.cosine_topk(q, c, n, d, k, workspace=ws, dbg=dbg)
        iters += 10; torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    byt = n * ld * c.element_size()
    fl = 2.0 * B * n * d
    tag = ""
    if dbg & N.DBG_NO_QS: tag = " [slab-ring kernel]"
    elif dbg & N.DBG_OLD_QS: tag = " [three-launch query-stationary kernel]"
    elif dbg & N.DBG_FORCE_QS: tag = " [walk kernel forced]"
    print(f"B={B} n={n} d={d} {dtype} k={k}{tag}: {ms*1e3:.1f} us  {byt/ms/1e6:.1f} GB/s  {fl/ms/1e9:.1f} TFLOP/s  {B/ms*1e3:.0f} q/s", flush=True)

if __name__ == "__main__":
    torch.cuda.init()
    f16, f32 = torch.float16, torch.float32
    print("# C3 (1M x 768 fp16): whole corpus, then the per-GPU shards of N=2,4,8")
    for n in (1_000_000, 500_000, 250_000, 125_000):
        bench(256, n, 768, f16); bench(256, n, 768, f16, dbg=N.DBG_NO_QS)
        bench(256, n, 768, f16, dbg=N.DBG_FORCE_QS); bench(256, n, 768, f16, dbg=N.DBG_FORCE_QS | N.DBG_OLD_QS)
    print("# C2 (100k x 384 fp32, one GPU)")
    for B in (1, 32, 256): bench(B, 100_000, 384, f32)
    print("# C4 (600k x 512 fp16): whole and 1/8")
    for n in (600_000, 75_000):
        bench(256, n, 512, f16); bench(256, n, 512, f16, dbg=N.DBG_NO_QS); bench(256, n, 512, f16, dbg=N.DBG_FORCE_QS)
    print("# C5 (10M x 768 fp16, B=1024): the 1/8 shard")
    bench(1024, 1_250_000, 768, f16); bench(1024, 1_250_000, 768, f16, dbg=N.DBG_NO_QS)
    bench(1024, 1_250_000, 768, f16, dbg=N.DBG_OLD_QS)
    print("# other batch sizes / depths at 1M x 768")
    for B in (1, 32, 128): bench(B, 1_000_000, 768, f16)
    for B in (32, 128, 256):
        bench(B, 1_000_000, 768, f16, k=10); bench(B, 1_000_000, 768, f16, k=20)
    bench(256, 1_000_000, 768, torch.bfloat16)
