"""Dynamic batching in front of `EmbeddingManager.query`.

The service issues batch-of-1 queries (api.py:338).  The dispatcher collects concurrent requests
for up to `max_wait_ms` (or `max_batch` requests), runs ONE batched encode + ONE
[B, d] x [N, d]^T search for them, and resolves each caller's future -- this is what converts the
kernel's batched throughput into served queries/s, and it also serialises device work (and, in a
multi-GPU deployment, the RCCL collectives) through a single task.
"""
from __future__ import annotations

import asyncio
import time
from typing import Any, Awaitable, Callable, Dict, List, Optional, Tuple

BatchFn = Callable[[List[str], int, Optional[Dict]], Awaitable[List[Dict[str, Any]]]]


class QueryDispatcher:
    def __init__(self, batch_fn: BatchFn, max_batch: int = 256, max_wait_ms: float = 2.0, idle_ms: float = 0.25):
        self.batch_fn = batch_fn
        self.max_batch = max_batch
        self.max_wait = max_wait_ms / 1e3
        # a batch also closes when nothing new has arrived for `idle_ms`: with a fixed set of callers that all come back
        # right after their answers, waiting out the whole window for requests that cannot exist is a quarter of the cycle
        self.idle = idle_ms / 1e3
        self._queue: "asyncio.Queue[Tuple[str, int, Optional[Dict], asyncio.Future]]" = asyncio.Queue()
        self._task: Optional[asyncio.Task] = None
        self.stats = {"requests": 0, "batches": 0, "max_batch_seen": 0}

    def start(self):
        if self._task is None or self._task.done():
            self._task = asyncio.get_running_loop().create_task(self._run())

    async def stop(self):
        if self._task is not None:
            self._task.cancel()
            try:
                await self._task
            except asyncio.CancelledError:
                pass
            self._task = None

    async def submit(self, text: str, n_results: int = 5, filter_dict: Optional[Dict] = None) -> Dict[str, Any]:
        self.start()
        fut: asyncio.Future = asyncio.get_running_loop().create_future()
        await self._queue.put((text, n_results, filter_dict, fut))
        return await fut

    async def _run(self):
        while True:
            first = await self._queue.get()
            batch = [first]
            deadline = time.monotonic() + self.max_wait
            while len(batch) < self.max_batch:
                try:                                  # whatever is already queued costs no await
                    batch.append(self._queue.get_nowait())
                    continue
                except asyncio.QueueEmpty:
                    pass
                timeout = min(deadline - time.monotonic(), self.idle)
                if timeout <= 0:
                    break
                try:
                    batch.append(await asyncio.wait_for(self._queue.get(), timeout))
                except asyncio.TimeoutError:
                    break
            # one kernel batch per (k, filter) group
            groups: Dict[Any, List[int]] = {}
            for i, (_, k, flt, _) in enumerate(batch):
                groups.setdefault((k, repr(flt)), []).append(i)
            for (_k, _f), idxs in groups.items():
                k, flt = batch[idxs[0]][1], batch[idxs[0]][2]
                try:
                    results = await self.batch_fn([batch[i][0] for i in idxs], k, flt)
                    for i, res in zip(idxs, results):
                        fut = batch[i][3]
                        if fut.done():
                            continue
                        if isinstance(res, dict) and "error" in res:
                            fut.set_exception(ValueError(res["error"]) if "empty" in res["error"]
                                              else RuntimeError(res["error"]))
                        else:
                            fut.set_result(res)
                except Exception as e:  # the whole batch failed
                    for i in idxs:
                        if not batch[i][3].done():
                            batch[i][3].set_exception(e)
                self.stats["batches"] += 1
                self.stats["max_batch_seen"] = max(self.stats["max_batch_seen"], len(idxs))
            self.stats["requests"] += len(batch)
