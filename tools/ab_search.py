"""A/B search-kernel variants (MMRAG_VARIANT / MMRAG_NW8, read per call) interleaved in ONE process:
devices differ by >10 %, never compare across runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_rag_amd import _native as N
B = int(os.environ.get("PB", "256")); n = int(os.environ.get("PN", "1000000")); d = 768; k = 5; dtype = torch.float16
cands = os.environ.get("CANDS", "default,nopre").split(",")
ld = N.padded_dim(d, dtype)
c = torch.randn((n, ld), device="cuda"); c = (c / c.norm(dim=1, keepdim=True)).to(dtype)
q = torch.randn((B, ld), device="cuda"); q = (q / q.norm(dim=1, keepdim=True)).to(dtype)
ws = torch.empty(2 * N.cosine_topk_workspace_bytes(B, n, k) + 4096, dtype=torch.uint8, device="cuda")
def setenv(c_):
    for k_ in ("MMRAG_NW8", "MMRAG_NO_PREPASS"): os.environ.pop(k_, None)
    if "nw8" in c_: os.environ["MMRAG_NW8"] = "1"
    if "nopre" in c_: os.environ["MMRAG_NO_PREPASS"] = "1"
def run(c_, iters=10):
    setenv(c_)
    N.cosine_topk_lists(q, c, n, d, k, ws)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): N.cosine_topk_lists(q, c, n, d, k, ws)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
ref = None
for c_ in cands:
    setenv(c_); s_, r_ = N.cosine_topk(q, c, n, d, k, workspace=ws)
    if ref is None: ref = (s_.clone(), r_.clone())
    else: assert torch.equal(r_, ref[1]) and torch.equal(s_, ref[0]), c_
res = {c_: [] for c_ in cands}
for rnd in range(6):
    for c_ in cands: res[c_].append(run(c_))
for c_ in cands:
    r = sorted(res[c_]); print(f"{c_:8s}: median {r[len(r)//2]:.1f} us  min {r[0]:.1f}  max {r[-1]:.1f}")
