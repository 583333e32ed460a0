"""Row-sharded search across the GPUs of one node.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI).  The corpus matrix
is split row-wise: rank g owns global rows [g*ceil(N/G), (g+1)*ceil(N/G)).  A query batch is
present on every rank; each rank runs the fused GEMM + top-k kernel on its shard, the per-rank
[B, k] (score, global row) blocks are packed into one buffer and exchanged by ONE all-gather (15-61 KB per rank:
latency-bound, not link-bound), and the G*k candidates per query are merged to k on the host
(north star; identical C++ ordering rule as the device merge: score desc, lower global row on
ties).  Because the global top-k is a subset of the union of shard top-k, the result is
identical for every G.

The reference is single-process (no collective anywhere), so nothing here
replaces reference code; it is the exchange step the row-sharded index needs.

`launch()` enqueues the device work of one query batch and returns immediately; `finish()`
waits for that batch's copy and merges on the host, so callers can keep `n_slots - 1` batches
in flight behind the one being merged (`n_slots` buffer sets; a batch's scan + tail latency is
longer than one host launch at small shards, so two sets leave the host waiting).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

from . import _native


def shard_range(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row range [lo, hi) of `rank` (SURVEY.md section 8e partitioning)."""
    per = (n_total + world - 1) // world
    return min(n_total, rank * per), min(n_total, (rank + 1) * per)


class ShardedSearch:
    def __init__(self, batch: int, k: int, world: int, rank: int, device: torch.device,
                 local_search: Callable[[int, torch.Tensor, torch.Tensor], None],
                 group: Optional[dist.ProcessGroup] = None, merge: str = "host",
                 collective_on_host: bool = False,
                 local_scan: Optional[Callable[[int], None]] = None, force_exchange: bool = False,
                 n_slots: int = 2):
        """local_search(slot, out_scores [B,k] f32, out_rows [B,k] i64) must enqueue / perform this
        rank's shard search with GLOBAL row ids (-1 / -inf padding).

        Two-phase form (GPU): if `local_scan(slot)` is given it is the corpus scan (phase 1 of
        mmrag_cosine_topk, into a per-slot workspace) and runs on the caller's stream, while
        local_search(slot, ...) is then only the small finishing step (phase 2).  Everything after
        the scan -- finish, all-gather, copy to the host -- runs on a side stream, so the next
        batch's scan does not wait for it (slots alternate between two buffer sets).

        force_exchange: run the all-gather and the G*k -> k merge even when world == 1 (rehearses the
        RCCL path on a one-GPU box; needs an initialised process group)."""
        if merge not in ("host", "device"):
            raise ValueError("merge must be 'host' or 'device'")
        if n_slots < 1:
            raise ValueError("n_slots must be >= 1")
        self.n_slots = n_slots
        self.B, self.k, self.world, self.rank = batch, k, world, rank
        self.device = torch.device(device)
        self.group = group
        self.merge = merge
        self.local_search = local_search
        self.local_scan = local_scan
        # rehearsal mode (gloo on a one-GPU box): the all-gather runs on host copies
        self.exchange = world > 1 or force_exchange
        self.collective_on_host = collective_on_host and self.exchange
        on_gpu = self.device.type == "cuda"
        # ONE packed exchange buffer per rank: [rows B*k i64 | scores B*k f32 | pad] -> one all-gather per batch
        nb = batch * k
        self.block_bytes = _native.packed_block_bytes(batch, k)
        self.locs, self.loc_rs, self.loc_ss, self.alls = [], [], [], []
        for _ in range(n_slots):  # buffer sets: later batches may start while batch i is still in its tail
            loc = torch.empty(self.block_bytes, dtype=torch.uint8, device=self.device)
            self.locs.append(loc)
            self.loc_rs.append(loc[: nb * 8].view(torch.int64).view(batch, k))
            self.loc_ss.append(loc[nb * 8: nb * 12].view(torch.float32).view(batch, k))
            self.alls.append(torch.empty(world * self.block_bytes, dtype=torch.uint8, device=self.device)
                             if self.exchange else loc)
        mk = (lambda: torch.empty(world * self.block_bytes, dtype=torch.uint8).pin_memory()) if on_gpu else (
            lambda: torch.empty(world * self.block_bytes, dtype=torch.uint8))
        self.host = [mk() for _ in range(n_slots)]
        self.copied = [torch.cuda.Event() for _ in range(n_slots)] if on_gpu else None
        # High priority: the persistent scan kernel fills every CU, so tail kernels only get CUs as scan
        # workgroups retire; at equal priority the next batch's scan is dispatched first and each tail op
        # waits out a whole scan.
        self.side = torch.cuda.Stream(self.device, priority=-1) if (on_gpu and local_scan is not None) else None
        self.scanned = [torch.cuda.Event() for _ in range(n_slots)] if self.side is not None else None
        # stream the scans run on (the one current at construction); with `side_is_current` the caller then
        # keeps `self.side` current for the whole loop, which saves a stream-context switch per batch
        self.main = torch.cuda.current_stream(self.device) if self.side is not None else None
        self.side_is_current = False
        self._tail_used = [False] * n_slots

    def _views(self, buf: torch.Tensor):
        """[G, B, k] score / row views of a packed exchange buffer (strided over the rank blocks)."""
        nb = self.B * self.k
        blocks = buf.view(self.world, self.block_bytes)
        return (blocks[:, nb * 8: nb * 12].contiguous().view(torch.float32).view(self.world, self.B, self.k),
                blocks[:, : nb * 8].contiguous().view(torch.int64).view(self.world, self.B, self.k))

    def launch(self, slot: int):
        """Device phase of one query batch: shard search, all-gather, async copy (or device merge)."""
        b = slot % self.n_slots
        if self.side is None:
            self._tail(slot, b)
            return
        main = self.main
        if self._tail_used[b]:
            main.wait_event(self.copied[b])      # the slot's workspace / buffers are free again
        self.local_scan(slot)
        self.scanned[b].record(main)
        if self.side_is_current:                 # caller made the side stream current for the whole loop
            self.side.wait_event(self.scanned[b])
            self._tail(slot, b)
        else:
            with torch.cuda.stream(self.side):
                self.side.wait_event(self.scanned[b])
                self._tail(slot, b)
        self._tail_used[b] = True

    def _tail(self, slot: int, b: int):
        loc, all_ = self.locs[b], self.alls[b]
        self.local_search(slot, self.loc_ss[b], self.loc_rs[b])
        if self.collective_on_host:
            gathered = torch.empty(self.world * self.block_bytes, dtype=torch.uint8)
            dist.all_gather_into_tensor(gathered, loc.cpu(), group=self.group)
            all_.copy_(gathered)
        elif self.exchange:
            dist.all_gather_into_tensor(all_, loc, group=self.group)
        if self.merge == "device" and self.exchange:
            s, r = _native.merge_topk(*self._views(all_), self.k)
            nb = self.B * self.k
            self.host[b][nb * 8: nb * 12].view(torch.float32).view(self.B, self.k).copy_(s, non_blocking=True)
            self.host[b][: nb * 8].view(torch.int64).view(self.B, self.k).copy_(r, non_blocking=True)
        elif self.side is not None:
            _native.copy_to_host_async(self.host[b], all_, self.side.cuda_stream)
        else:
            self.host[b].copy_(all_, non_blocking=True)
        if self.copied is not None:
            self.copied[b].record(self.side) if self.side is not None else self.copied[b].record()

    def finish(self, slot: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """Host phase: wait for the batch's copy, merge G*k -> k per query.  Returns CPU tensors."""
        b = slot % self.n_slots
        if self.copied is not None:
            self.copied[b].synchronize()
        if self.exchange and self.merge == "host":
            return _native.merge_topk_host_packed(self.host[b], self.world, self.B, self.k, self.k)
        nb = self.B * self.k
        return (self.host[b][nb * 8: nb * 12].view(torch.float32).view(self.B, self.k),
                self.host[b][: nb * 8].view(torch.int64).view(self.B, self.k))

    def search(self, slot: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
        self.launch(slot)
        return self.finish(slot)
