"""A/B search-kernel variants (debug switches of the internal entry point) interleaved in ONE process:
devices differ by >10 %, never compare across runs.  CANDS=default,noqs,nopre,noqs+nopre,noqs+nw8,old,m32,m16,nodyn,noseed (joined with +)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_rag_amd import _native as N
if os.environ.get("MMRAG_AB_LIB"):   # a developer build of the library (tools/build_variant.sh)
    N.LIB_PATH = os.environ["MMRAG_AB_LIB"]
B = int(os.environ.get("PB", "256")); n = int(os.environ.get("PN", "1000000")); d = 768; k = 5; dtype = torch.float16
cands = os.environ.get("CANDS", "default,noqs,nopre").split(",")
ld = N.padded_dim(d, dtype)
c = torch.randn((n, ld), device="cuda"); c = (c / c.norm(dim=1, keepdim=True)).to(dtype)
q = torch.randn((B, ld), device="cuda"); q = (q / q.norm(dim=1, keepdim=True)).to(dtype)
ws = torch.empty(2 * N.cosine_topk_workspace_bytes(B, n, k) + 65536, dtype=torch.uint8, device="cuda")
def flags(c_):
    return (8 if "qs4" in c_ else 0) | (16 if "nosel" in c_ else 0) | (32 if "nodma" in c_ else 0) | (64 if "nobar" in c_ else 0) | (128 if "nomfma" in c_ else 0) | (256 if "clk" in c_ else 0) | (128 if "fastonly" in c_ else 0) | (0x4000 if "pre1" in c_ else 0) | (0x8000 if "pre2" in c_ else 0) | (0x1000 if "pd2" in c_ else 0) | (0x2000 if "pd5" in c_ else 0) | (0x3000 if "pd7" in c_ else 0) | (512 if "dmal2" in c_ else 0) | (1024 if "nowait" in c_ else 0) | (N.DBG_OLD_QS if "old" in c_ else 0) | (N.DBG_MFMA32 if "m32" in c_ else 0) | (N.DBG_MFMA16 if "m16" in c_ else 0) | (N.DBG_NO_DYN if "nodyn" in c_ else 0) | (N.DBG_NO_SEED if "noseed" in c_ else 0) | (0x200000 if "pub1" in c_ else 0) | (N.DBG_8_WAVES if "nw8" in c_ else 0) | (N.DBG_NO_PREPASS if "nopre" in c_ else 0) | (N.DBG_NO_QS if "noqs" in c_ else 0)
def run(c_, iters=10):
    f = flags(c_)
    N.cosine_topk_lists(q, c, n, d, k, ws, dbg=f)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): N.cosine_topk_lists(q, c, n, d, k, ws, dbg=f)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
ref = None
for c_ in cands:
    if flags(c_) & 0x6f0: continue   # timing-only ablation: results are wrong by construction
    s_, r_ = N.cosine_topk(q, c, n, d, k, workspace=ws, dbg=flags(c_))
    if ref is None: ref = (s_.clone(), r_.clone())
    else: assert torch.equal(r_, ref[1]) and torch.equal(s_, ref[0]), c_
res = {c_: [] for c_ in cands}
for rnd in range(6):
    for c_ in cands: res[c_].append(run(c_))
for c_ in cands:
    r = sorted(res[c_]); print(f"{c_:8s}: median {r[len(r)//2]:.1f} us  min {r[0]:.1f}  max {r[-1]:.1f}")

need = N.cosine_topk_workspace_bytes(B, n, k) - 256
import numpy as np
for c_ in cands:
    if "clk" not in c_: continue
    walk = "old" not in c_ and "noqs" not in c_     # the walk kernel writes 32 words per workgroup, the others 8
    W = 32 if walk else 8
    ws[need:need + 256 * W * 8].zero_()
    run(c_, iters=int(os.environ.get('CLK_ITERS', '3'))); torch.cuda.synchronize()   # (CLK_ITERS=3000: stamps of a throttled launch)
    st = ws[need:need + 256 * W * 8].view(torch.int64).cpu().numpy().reshape(256, W).astype(np.float64)
    mhz = st[:, 0] / st[:, 1] * 100.0
    us = st[:, 1] / 100.0
    print(f"{c_}: clock MHz min/med/max {mhz.min():.0f}/{np.median(mhz):.0f}/{mhz.max():.0f}; main loop us min/med/max "
          f"{us.min():.1f}/{np.median(us):.1f}/{us.max():.1f}; per XCD-label median us "
          + " ".join(f"{np.median(us[i::8]):.0f}" for i in range(8)))
    print(f"   entry -> main loop: median {np.median(st[:,2])/100:.1f} us (max {st[:,2].max()/100:.1f}); "
          f"first entry -> last loop end: {(st[:,3] + st[:,2] + st[:,1]).max()/100 - st[:,3].min()/100:.1f} us; "
          f"entry skew {(st[:,3].max() - st[:,3].min())/100:.1f} us")
    if walk:
        t0 = st[:, 3] + st[:, 2]                      # main loop start (100 MHz ticks)
        ends = st[:, 4:32]
        ok = ends > 0
        prev = np.concatenate([t0[:, None], ends[:, :-1]], 1)
        dur = np.where(ok, (ends - prev) / 100.0, np.nan)
        print("   per-tile us (median over workgroups), tiles 0..27: " + " ".join(f"{x:.1f}" for x in np.nanmedian(dur, 0)))
        print("   per-tile us (max over workgroups):               " + " ".join(f"{x:.1f}" for x in np.nanmax(dur, 0)))

# steady state: the chip throttles under sustained MFMA + HBM load (a kernel can run 350 us for the first five
# launches and 500 us afterwards), so each candidate is also held for STEADY seconds and timed over the second half
steady = float(os.environ.get("STEADY", "0"))
if steady > 0:
    import time
    for c_ in cands:
        if "clk" in c_: continue
        f = flags(c_)
        t_end = time.time() + steady / 2
        while time.time() < t_end:
            for _ in range(20): N.cosine_topk_lists(q, c, n, d, k, ws, dbg=f)
            torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        iters = 0; e0.record(); t_end = time.time() + steady / 2
        while time.time() < t_end:
            for _ in range(20): N.cosine_topk_lists(q, c, n, d, k, ws, dbg=f)
            iters += 20
            if iters % 200 == 0: torch.cuda.synchronize()
        e1.record(); torch.cuda.synchronize()
        print(f"{c_:24s}: steady {e0.elapsed_time(e1) / iters * 1e3:.1f} us/scan over {iters} scans")
