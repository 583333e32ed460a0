"""compare current libmmrag.so variants against the committed v1 kernel in ONE process"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_rag_amd import _native as N
from ctypes import c_void_p, c_int, c_int64, c_size_t
old = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_old", "libmmrag_old.so"), mode=ctypes.RTLD_LOCAL)
old.mmrag_cosine_topk_lists.restype = c_int
old.mmrag_cosine_topk_lists.argtypes = [c_void_p, c_void_p, c_int, c_int64, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]
d = 768; k = 5; dtype = torch.float16
ld = N.padded_dim(d, dtype)
def setup(B, n):
    c = torch.randn((n, ld), device="cuda"); c = (c / c.norm(dim=1, keepdim=True)).to(dtype)
    q = torch.randn((B, ld), device="cuda"); q = (q / q.norm(dim=1, keepdim=True)).to(dtype)
    ws = torch.empty(2 * N.cosine_topk_workspace_bytes(B, n, k) + 4096, dtype=torch.uint8, device="cuda")
    return c, q, ws
def timeit(fn, iters=10):
    fn(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for B, n in [(256, 1_000_000)]:
    c, q, ws = setup(B, n)
    st = torch.cuda.current_stream().cuda_stream
    f_old = lambda: old.mmrag_cosine_topk_lists(q.data_ptr(), c.data_ptr(), B, n, d, ld, 1, k, None, ws.data_ptr(), ws.numel(), st)
    def f_new(v):
        def f():
            os.environ["MMRAG_DBG"] = str(v)
            N.cosine_topk_lists(q, c, n, d, k, ws)
        return f
    cands = {"old": f_old}
    for v in [int(x) for x in os.environ.get("DBGS", "0,1,2,3,4,5,7").split(",")]: cands[f"dbg{v}"] = f_new(v)
    res = {k_: [] for k_ in cands}
    for r in range(5):
        for k_, f in cands.items(): res[k_].append(timeit(f))
    print(f"B={B} n={n}: " + "  ".join(f"{k_}: {sorted(v)[len(v)//2]:.1f}" for k_, v in res.items()), flush=True)
