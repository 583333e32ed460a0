"""Developer soak of the single-launch search kernel: the same answer from every one of many launches, on shapes that take
the ticketed tail (>= 24 tiles per workgroup) at the three row widths, full and ragged batches, with an alive mask."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multimodal_rag_amd import _native as N
launches = int(os.environ.get("LAUNCHES", "300"))
g = torch.Generator(device="cuda").manual_seed(7)
for d, n, B in ((384, 400_003, 256), (512, 420_000, 200), (768, 400_003, 256), (768, 1_000_000, 256), (384, 900_001, 129)):
    ld = N.padded_dim(d, torch.float16)
    c = torch.randint(-2, 3, (n, ld), device="cuda", generator=g).half(); c[:, d:] = 0      # exact scores, ties everywhere
    q = torch.randint(-2, 3, (B, ld), device="cuda", generator=g).half(); q[:, d:] = 0
    alive = (torch.rand(n, device="cuda", generator=g) < 0.7)
    bits = torch.from_numpy(np.packbits(alive.cpu().numpy(), bitorder="little").view(np.int32).copy() if alive.numel() % 32 == 0 else
                            np.packbits(np.concatenate([alive.cpu().numpy(), np.zeros(32 - alive.numel() % 32, bool)]), bitorder="little").view(np.int32).copy()).cuda()
    for mask in (None, bits):
        s0, r0 = N.cosine_topk(q, c, n, d, 5, alive_bits=mask, dbg=N.DBG_FORCE_QS)
        # reference: exact integer scores on the device, ties to the lower row
        full = q.float() @ c.float().t()
        if mask is not None: full[:, ~alive] = float("-inf")
        key = full.double() * 4_000_000 - torch.arange(n, device="cuda").double()[None, :]
        want_r = key.topk(5, dim=1).indices
        assert torch.equal(r0, want_r), (d, n, B, mask is not None)
        bad = 0
        for _ in range(launches):
            s, r = N.cosine_topk(q, c, n, d, 5, alive_bits=mask, dbg=N.DBG_FORCE_QS)
            bad += int(not (torch.equal(r, r0) and torch.equal(s, s0)))
        print(f"d={d} n={n} B={B} mask={mask is not None}: {launches} launches, {bad} differ", flush=True)
        assert bad == 0
print("soak ok")
