"""Multi-GPU serving loop around the row-sharded index.

One process per GPU.  Rank 0 owns the public surface (`EmbeddingManager`, FastAPI) and drives a
`ShardedCollection`, which has the collection interface the embedder expects (add / query / get /
delete / count / reset); every other rank sits in `worker_loop()` and executes the same command
on its own shard.  A shard is anything with `VectorIndex`'s methods -- on the GPU it IS a
`VectorIndex` (rows in HBM, fused cosine top-k kernel), in the CPU tests a numpy stand-in.

Query data path (the part BASELINE measures, same sequence as `ShardedSearch`):
    rank 0 broadcasts the command and the query matrix
    every rank:  shard.search(q, k, where)            local top-k, local row numbers
                 pack [sequence no. i64 | score f32]  sequence no. = global insertion order of the row
                 ONE all-gather of the packed blocks  (RCCL on GPUs, gloo in the CPU tests)
                 host merge G*k -> k  (mmrag_merge_topk_host_packed; every rank runs it: the
                                       winners tell each rank which of its rows to describe)
                 gather_object(payload of my winning rows) -> rank 0
    rank 0 assembles the Chroma-shaped result.
Ties are broken by (score desc, sequence number asc) = earlier insert first, the rule of a single
`VectorIndex`, so the answer does not depend on how many ranks share the corpus.
Ingest: rank 0 assigns each new item to the emptiest shard and broadcasts the batch; each rank
keeps its part.  Ids, documents and metadata live with their rows, on the owning rank's host.

The reference is single-process; nothing here replaces reference code.
"""
from __future__ import annotations

import logging
import threading
import time
from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from . import _native
from .tracing import stage

logger = logging.getLogger(__name__)


class ShardError(RuntimeError):
    """a rank's local step of a collection command failed; raised on rank 0 with every failing rank's message"""


OP_STOP, OP_QUERY, OP_OBJECT, OP_PING = 0, 1, 2, 3
_INCLUDE_BITS = {"distances": 1, "metadatas": 2, "documents": 4, "embeddings": 8}
_FAILED_ROW = -2      # first sequence number of a rank's packed block when its local search raised


class ShardedCollection:
    """Control plane and data plane are separate:

    * control (`control_group`, default = `group`): a fixed 8-word int64 header per command, and pickled objects
      only for what is not a tensor (ingest batches, get / delete arguments, `where` dicts, the winners' payload).
      Give it a gloo group whose timeout is longer than the keep-alive interval (`start_keepalive`): workers block in
      it between requests, and a collective watchdog (the default 10 minutes of an RCCL group) would abort an idle
      service; with the keep-alive a finite timeout (hours) still turns a lost collective step into an error.
    * data (`group`, buffers on `device`): a query = header + the [B, d] float32 query matrix as a tensor broadcast,
      ONE all-gather of the packed per-rank candidates, the C++ host merge.  Nothing is pickled on that path.

    Every command is: local step on every rank (may raise) -> status exchange -> finish on rank 0.  A rank whose
    local step raised reports it instead of leaving the collective sequence, so rank 0 raises `ShardError` and the
    service keeps running."""

    def __init__(self, shard, group: Optional[dist.ProcessGroup] = None, device: Optional[torch.device] = None,
                 shard_io=None, encode_fn=None, control_group: Optional[dist.ProcessGroup] = None,
                 encode_batch: int = 32):
        """`shard`: this rank's VectorIndex (or stand-in).  `device`: where collective buffers live
        (the shard's GPU with the nccl backend, CPU with gloo).  `shard_io`: (save(shard, directory),
        load(directory) -> shard) used by save() / load(); default = persistence.save_index / load_index."""
        self.shard = shard
        self._shard_io = shard_io
        # data-parallel ingest: encode_fn(texts) -> [n, d] float32 on THIS rank (every rank holds the encoder);
        # with it add_texts() ships strings, each rank embeds the items it will own and appends them locally --
        # no vector leaves its GPU (SURVEY.md section 8e "Embed: no collective at all")
        self.encode_fn = encode_fn
        self.encode_batch = max(1, int(encode_batch))
        self.group = group
        self.control = control_group if control_group is not None else group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self._ctl_device = self.device if dist.get_backend(self.control) == "nccl" else torch.device("cpu")
        self.name = getattr(shard, "name", "sharded")
        self.metadata = getattr(shard, "metadata", None)
        self._owner: Dict[str, int] = {}          # rank 0 only: id -> owning rank
        self._counts = [0] * self.world           # rank 0 only: rows per shard
        self._next_seq = 0                        # rank 0 only: global insertion counter
        self._seq_of: Dict[str, int] = {}         # this rank's rows: id -> sequence number
        self._id_of: Dict[int, str] = {}          # ... and back
        # the embedder calls the collection from asyncio.to_thread workers (embedder.py:517, 595): one
        # command (= one sequence of collectives) at a time per communicator
        self._lock = threading.Lock()
        self._last_header = time.monotonic()      # rank 0: when the workers last heard from it
        self._keepalive: Optional[threading.Thread] = None
        self._stopped = threading.Event()

    # ------------------------------------------------------------------ control plane -------
    def _header(self, words: Optional[Sequence[int]]) -> List[int]:
        self._last_header = time.monotonic()
        t = torch.zeros(8, dtype=torch.int64, device=self._ctl_device)
        if words is not None:
            t[: len(words)] = torch.tensor(list(words), dtype=torch.int64)
        dist.broadcast(t, src=0, group=self.control)
        return [int(x) for x in t.cpu()]

    def _object(self, obj: Any) -> Any:
        box = [obj]
        dist.broadcast_object_list(box, src=0, group=self.control)
        return box[0]

    def _gather(self, obj: Any) -> Optional[List[Any]]:
        out = [None] * self.world if self.rank == 0 else None
        dist.gather_object(obj, out, dst=0, group=self.control)
        return out

    def _require_rank0(self):
        if self.rank != 0:
            raise RuntimeError("collection methods are driven from rank 0; other ranks run worker_loop()")

    def worker_loop(self):
        """Ranks != 0: execute rank 0's commands until it sends 'stop'.  A failing command has already reported its
        error through the command's own status exchange; the loop goes on."""
        if self.rank == 0:
            raise RuntimeError("rank 0 drives the collection; worker_loop() is for the other ranks")
        while True:
            hdr = self._header(None)
            if hdr[0] == OP_STOP:
                return
            if hdr[0] == OP_PING:
                continue
            try:
                if hdr[0] == OP_QUERY:
                    self._query_path(hdr, None, None)
                else:
                    self._execute(self._object(None))
            except Exception:   # noqa: BLE001 -- the service must survive a bad request
                logger.exception("rank %d: command failed", self.rank)

    def stop(self):
        self._require_rank0()
        self._stopped.set()
        with self._lock:
            self._header([OP_STOP])

    def ping(self):
        """rank 0: a header nobody acts on; resets the workers' wait in the control group"""
        self._require_rank0()
        with self._lock:
            if not self._stopped.is_set():
                self._header([OP_PING])

    def start_keepalive(self, interval_s: float = 600.0):
        """rank 0: ping the workers whenever the service has been idle for `interval_s`, so that the control group can
        have a finite timeout (a lost collective step then surfaces as an error instead of a hang for ever)"""
        self._require_rank0()
        if self._keepalive is not None:
            return

        def loop():
            while not self._stopped.wait(min(interval_s / 4.0, 30.0)):
                if time.monotonic() - self._last_header >= interval_s:
                    try:
                        self.ping()
                    except Exception:   # noqa: BLE001 -- the next real command will report the broken group
                        logger.exception("keep-alive ping failed")
                        return

        self._keepalive = threading.Thread(target=loop, name="mmrag-shard-keepalive", daemon=True)
        self._keepalive.start()

    def _execute(self, cmd: Dict[str, Any]):
        """local step on this rank -> one status gather -> finish on rank 0"""
        try:
            status = (True, getattr(self, "_local_" + cmd["op"])(cmd))
        except Exception as e:   # noqa: BLE001
            logger.exception("rank %d: local step of %r failed", self.rank, cmd["op"])
            status = (False, f"{type(e).__name__}: {e}")
        parts = self._gather(status)
        if self.rank != 0:
            return None
        bad = [f"rank {g}: {p[1]}" for g, p in enumerate(parts) if not p[0]]
        if bad:
            raise ShardError(f"{cmd['op']} failed on " + "; ".join(bad))
        return getattr(self, "_finish_" + cmd["op"])(cmd, [p[1] for p in parts])

    def _run(self, cmd: Dict[str, Any]):
        """rank 0: broadcast a command and take part in it, one at a time"""
        with self._lock:
            self._header([OP_OBJECT])
            return self._execute(self._object(cmd))

    # ------------------------------------------------------------------ add -----------------
    def _assign(self, ids: Sequence[str]):
        """shard assignment of a batch (emptiest shard first) WITHOUT touching the books: they are committed only
        when every rank has stored its part, so a failed add can be retried (embedder.py:514-537 retries)"""
        counts = list(self._counts)
        seq = self._next_seq
        seen, plan = set(), []
        for i, s in enumerate(ids):
            if s in self._owner or s in seen:    # duplicate id: ignored, as VectorIndex.add does
                continue
            seen.add(s)
            r = int(np.argmin(counts))
            counts[r] += 1
            plan.append((i, r, seq))
            seq += 1
        return plan

    def _add(self, payload_key: str, payload, documents, metadatas, ids):
        self._require_rank0()
        n = len(payload)
        if ids is None or len(ids) != n:
            raise ValueError("ids are required, one per item")
        documents = list(documents) if documents is not None else [None] * n
        metadatas = list(metadatas) if metadatas is not None else [{} for _ in range(n)]
        with self._lock:
            plan = self._assign(ids)
            parts: Dict[int, Dict[str, Any]] = {}
            for i, r, seq in plan:
                p = parts.setdefault(r, {payload_key: [], "documents": [], "metadatas": [], "ids": [], "seqs": []})
                p["seqs"].append(seq)
                p[payload_key].append(payload[i])
                p["documents"].append(documents[i])
                p["metadatas"].append(metadatas[i])
                p["ids"].append(ids[i])
            if payload_key == "embeddings":
                for p in parts.values():
                    p["embeddings"] = np.asarray(p["embeddings"], dtype=np.float32)
            self._header([OP_OBJECT])
            try:
                self._execute(self._object({"op": "add", "parts": parts}))
            except ShardError:
                # some ranks stored their part: take the whole batch out again, then report
                self._header([OP_OBJECT])
                try:
                    self._execute(self._object({"op": "delete", "ids": [ids[i] for i, _, _ in plan], "where": None,
                                                "uncounted": True}))
                finally:
                    pass
                raise
            for i, r, seq in plan:
                self._owner[ids[i]] = r
                self._counts[r] += 1
            if plan:
                self._next_seq = plan[-1][2] + 1

    def add(self, embeddings, documents=None, metadatas=None, ids: Optional[Sequence[str]] = None):
        self._add("embeddings", embeddings, documents, metadatas, ids)

    def add_texts(self, texts: Sequence[str], documents=None, metadatas=None, ids: Optional[Sequence[str]] = None):
        """Ingest by text: every rank embeds and stores its own share (needs encode_fn on every rank)."""
        if self.encode_fn is None:
            raise RuntimeError("add_texts needs an encode_fn on every rank")
        self._add("texts", texts, documents, metadatas, ids)

    def _local_add(self, cmd):
        p = cmd["parts"].get(self.rank)
        if p:
            if "embeddings" in p:
                emb = p["embeddings"]
            else:   # this rank's share, in slices of the service's batch size (api.py:93): bounded workspace
                texts = p["texts"]
                emb = np.concatenate([np.asarray(self.encode_fn(texts[i:i + self.encode_batch]), dtype=np.float32)
                                      for i in range(0, len(texts), self.encode_batch)])
            self.shard.add(emb, documents=p["documents"], metadatas=p["metadatas"], ids=p["ids"])
            for i, sq in zip(p["ids"], p["seqs"]):
                self._seq_of[i] = sq
                self._id_of[sq] = i
        return None

    def _finish_add(self, cmd, parts):
        return None

    # ------------------------------------------------------------------ query ---------------
    def query(self, query_embeddings, n_results: int = 10, where: Optional[Dict[str, Any]] = None,
              include: Sequence[str] = ("metadatas", "documents", "distances")) -> Dict[str, Any]:
        self._require_rank0()
        q = torch.as_tensor(np.ascontiguousarray(np.asarray(query_embeddings, dtype=np.float32)))
        if q.dim() == 1:
            q = q.unsqueeze(0)
        mask = 0
        for x in include:
            mask |= _INCLUDE_BITS[x]
        with self._lock:
            hdr = self._header([OP_QUERY, q.shape[0], int(n_results), q.shape[1], 1 if where else 0, mask])
            return self._query_path(hdr, q, where)

    def _query_path(self, hdr, q, where):
        _, B, k, d, has_where, mask = hdr[:6]
        include = tuple(x for x, bit in _INCLUDE_BITS.items() if mask & bit)
        # the query matrix: a tensor broadcast on the control group's device (no pickling)
        qt = (q.to(self._ctl_device) if self.rank == 0 else torch.empty((B, d), dtype=torch.float32,
                                                                       device=self._ctl_device))
        with stage("shard.broadcast"):
            dist.broadcast(qt, src=0, group=self.control)
            if has_where:
                where = self._object(where)
        nb = B * k
        nbytes = _native.packed_block_bytes(B, k)
        loc = torch.zeros(nbytes, dtype=torch.uint8)
        failure = None
        try:
            with stage("shard.search"):
                scores, rows = self.shard.search(qt.cpu().numpy() if self._ctl_device.type == "cpu" else qt, k, where)
            scores = torch.as_tensor(scores).to(torch.float32)
            rows = torch.as_tensor(rows).to(torch.int64)
            kk = scores.shape[1]
            if kk < k:                                           # shard returned fewer columns than asked
                pad = k - kk
                scores = torch.cat([scores, torch.full((B, pad), float("-inf"), dtype=scores.dtype, device=scores.device)], 1)
                rows = torch.cat([rows, torch.full((B, pad), -1, dtype=rows.dtype, device=rows.device)], 1)
            # local row -> global sequence number (host tables)
            rows_h = rows.cpu().numpy()
            live = sorted({int(r) for r in rows_h.reshape(-1) if r >= 0})
            seq_of_row = {r: self._seq_of[i] for r, i in zip(live, self.shard.ids_of_rows(live))}
            seqs = np.array([[seq_of_row[int(r)] if r >= 0 else -1 for r in row] for row in rows_h], np.int64).reshape(B, k)
            # ONE packed block per rank: [sequence numbers B*k i64 | scores B*k f32 | pad]
            loc[: nb * 8].view(torch.int64).copy_(torch.from_numpy(seqs).reshape(-1))
            loc[nb * 8: nb * 12].view(torch.float32).copy_(scores.reshape(-1).cpu())
        except Exception as e:   # noqa: BLE001 -- report through the data collective, stay in step
            logger.exception("rank %d: local search failed", self.rank)
            failure = f"{type(e).__name__}: {e}"
            loc.zero_()
            loc[: nb * 8].view(torch.int64).fill_(-1)
            loc[nb * 8: nb * 12].view(torch.float32).fill_(float("-inf"))
            loc[:8].view(torch.int64)[0] = _FAILED_ROW
        loc = loc.to(self.device)
        all_ = torch.empty(self.world * nbytes, dtype=torch.uint8, device=self.device)
        with stage("shard.exchange"):
            dist.all_gather_into_tensor(all_, loc, group=self.group)
            host = all_.cpu()
        heads = host.view(self.world, nbytes)[:, :8].contiguous().view(torch.int64).reshape(-1)
        if bool((heads == _FAILED_ROW).any()):
            msgs = self._gather(failure)        # every rank saw the marker: they all come here
            if self.rank != 0:
                return None
            raise ShardError("query failed on " + "; ".join(f"rank {g}: {m}" for g, m in enumerate(msgs) if m))
        # Merge and payload: a rank that raises here (a payload fetch is a GPU call for include=embeddings) must still
        # take part in the payload gather, or rank 0 waits in it for ever while holding the service lock.  Every rank
        # gathers an (ok, payload-or-message) pair, as `_execute` does for the other commands.
        mine: Dict[int, Dict[str, Any]] = {}
        failure = None
        top_s = top_r = None
        try:
            with stage("shard.merge"):
                if k <= _native.MAX_K:
                    top_s, top_r = _native.merge_topk_host_packed(host, self.world, B, k, k)
                    top_s, top_r = top_s.numpy(), top_r.numpy()
                else:
                    top_s, top_r = _merge_deep(host, self.world, B, k)
            # describe the winning rows this rank owns (the only pickled part of a query: k rows' payload per query)
            owned = sorted({int(sq) for sq in top_r.reshape(-1) if sq >= 0 and int(sq) in self._id_of})
            if owned:
                got = self.shard.get(ids=[self._id_of[sq] for sq in owned],
                                     include=tuple(x for x in include if x != "distances"))
                at = {i: j for j, i in enumerate(got["ids"])}
                for sq in owned:
                    j = at[self._id_of[sq]]
                    mine[sq] = {"id": self._id_of[sq],
                                "metadata": got["metadatas"][j] if got.get("metadatas") is not None else None,
                                "document": got["documents"][j] if got.get("documents") is not None else None,
                                "embedding": got["embeddings"][j] if got.get("embeddings") is not None else None}
        except Exception as e:   # noqa: BLE001 -- reported through the gather below, the rank stays in step
            logger.exception("rank %d: building the query payload failed", self.rank)
            failure = f"{type(e).__name__}: {e}"
        with stage("shard.payload"):
            answers = self._gather((failure is None, mine if failure is None else failure))
        if self.rank != 0:
            return None
        bad = [(g, a[1]) for g, a in enumerate(answers) if not a[0]]
        if bad:
            raise ShardError("query failed on " + "; ".join(f"rank {g}: {m}" for g, m in bad))
        payloads = [a[1] for a in answers]
        out: Dict[str, Any] = {"ids": []}
        for key in ("distances", "metadatas", "documents", "embeddings"):
            out[key] = [] if key in include else None
        where_is = {}
        for g, p in enumerate(payloads):
            for sq in p:
                where_is[sq] = g
        for b in range(B):
            hits = [(where_is[int(sq)], int(sq), float(sc)) for sq, sc in zip(top_r[b], top_s[b]) if sq >= 0]
            out["ids"].append([payloads[g][sq]["id"] for g, sq, _ in hits])
            if "distances" in include:
                out["distances"].append([float(np.float32(1.0) - np.float32(sc)) for _, _, sc in hits])
            for key, field in (("metadatas", "metadata"), ("documents", "document"), ("embeddings", "embedding")):
                if key in include:
                    out[key].append([payloads[g][sq][field] for g, sq, _ in hits])
        return out

    # ------------------------------------------------------------------ get / delete / count -
    def get(self, ids: Optional[Sequence[str]] = None, where: Optional[Dict[str, Any]] = None,
            include: Sequence[str] = ("metadatas", "documents")) -> Dict[str, Any]:
        self._require_rank0()
        return self._run({"op": "get", "ids": list(ids) if ids is not None else None, "where": where,
                          "include": tuple(include)})

    def _local_get(self, cmd):
        return self.shard.get(ids=cmd["ids"], where=cmd["where"], include=cmd["include"])

    def _finish_get(self, cmd, parts):
        rows = []
        for p in parts:
            for j, i in enumerate(p["ids"]):
                rows.append((i, {k: (p[k][j] if p.get(k) is not None else None)
                                 for k in ("metadatas", "documents", "embeddings")}))
        if cmd["ids"] is not None:                      # requested order, as a single index answers
            order = {s: n for n, s in enumerate(cmd["ids"])}
            rows.sort(key=lambda t: order.get(t[0], len(order)))
        out: Dict[str, Any] = {"ids": [r[0] for r in rows]}
        for k in ("metadatas", "documents", "embeddings"):
            out[k] = [r[1][k] for r in rows] if k in cmd["include"] else None
        return out

    def delete(self, ids: Optional[Sequence[str]] = None, where: Optional[Dict[str, Any]] = None) -> List[str]:
        self._require_rank0()
        return self._run({"op": "delete", "ids": list(ids) if ids is not None else None, "where": where})

    def _local_delete(self, cmd):
        gone = self.shard.delete(ids=cmd["ids"], where=cmd["where"])
        for i in gone:
            self._id_of.pop(self._seq_of.pop(i), None)
        return gone

    def _finish_delete(self, cmd, parts):
        out = []
        for g, p in enumerate(parts):
            for i in p:
                if not cmd.get("uncounted"):      # (roll-back of a failed add: those ids never reached the books)
                    self._owner.pop(i, None)
                    self._counts[g] -= 1
                out.append(i)
        return sorted(out)

    def count(self) -> int:
        self._require_rank0()
        return self._run({"op": "count"})

    def _local_count(self, cmd):
        return self.shard.count()

    def _finish_count(self, cmd, parts):
        return sum(parts)

    def reset(self):
        self._require_rank0()
        self._run({"op": "reset"})

    def _local_reset(self, cmd):
        self.shard.reset()
        self._seq_of.clear()
        self._id_of.clear()
        return None

    def _finish_reset(self, cmd, parts):
        self._owner.clear()
        self._counts = [0] * self.world
        self._next_seq = 0

    # ------------------------------------------------------------------ save / load ---------
    def save(self, directory: str):
        """Persist every rank's shard (rows + tables + sequence numbers) under `directory`/rank_<r>."""
        self._require_rank0()
        self._run({"op": "save", "dir": directory})

    def load(self, directory: str):
        """Replace every rank's shard by what save() wrote (same world size)."""
        self._require_rank0()
        self._run({"op": "load", "dir": directory})

    def _io(self):
        if self._shard_io is not None:
            return self._shard_io
        from . import persistence

        dev = str(getattr(self.shard, "device", "cuda:0"))
        return persistence.save_index, (lambda d: persistence.load_index(d, device=dev))

    def _local_save(self, cmd):
        import json
        import os

        d = os.path.join(cmd["dir"], f"rank_{self.rank}")
        os.makedirs(d, exist_ok=True)
        self._io()[0](self.shard, d)
        with open(os.path.join(d, "sequence.json"), "w", encoding="utf-8") as f:
            json.dump({"world": self.world, "seq_of": self._seq_of}, f)
        return True

    def _finish_save(self, cmd, parts):      # every rank has written before rank 0 returns
        import json
        import os

        with open(os.path.join(cmd["dir"], "sharded.json"), "w", encoding="utf-8") as f:
            json.dump({"world": self.world, "next_seq": self._next_seq}, f)

    def _local_load(self, cmd):
        import json
        import os

        d = os.path.join(cmd["dir"], f"rank_{self.rank}")
        with open(os.path.join(d, "sequence.json"), encoding="utf-8") as f:
            t = json.load(f)
        if t["world"] != self.world:
            raise ValueError(f"saved with {t['world']} ranks, loading with {self.world}")
        self.shard = self._io()[1](d)
        self._seq_of = {i: int(sq) for i, sq in t["seq_of"].items()}
        self._id_of = {sq: i for i, sq in self._seq_of.items()}
        return list(self._seq_of)

    def _finish_load(self, cmd, parts):
        import json
        import os

        with open(os.path.join(cmd["dir"], "sharded.json"), encoding="utf-8") as f:
            self._next_seq = int(json.load(f)["next_seq"])
        self._owner = {i: g for g, p in enumerate(parts) for i in p}
        self._counts = [len(p) for p in parts]


def _merge_deep(host: torch.Tensor, G: int, B: int, k: int):
    """k > MAX_K (single deep queries, get_similar_documents): same order rule as the C++ merge
    (score desc, lower sequence number on ties) over the G*k candidates, in numpy."""
    nb = B * k
    nbytes = _native.packed_block_bytes(B, k)
    blocks = host.view(G, nbytes)
    rows = np.stack([blocks[g, : nb * 8].view(torch.int64).view(B, k).numpy() for g in range(G)], 1).reshape(B, G * k)
    scores = np.stack([blocks[g, nb * 8: nb * 12].view(torch.float32).view(B, k).numpy() for g in range(G)], 1).reshape(B, G * k)
    out_s = np.full((B, k), -np.inf, np.float32)
    out_r = np.full((B, k), -1, np.int64)
    for b in range(B):
        ok = np.nonzero((rows[b] >= 0) & np.isfinite(scores[b]))[0]
        order = ok[np.lexsort((rows[b][ok], -scores[b][ok]))][:k]
        out_s[b, : len(order)] = scores[b][order]
        out_r[b, : len(order)] = rows[b][order]
    return out_s, out_r
