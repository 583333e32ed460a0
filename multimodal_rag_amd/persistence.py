"""Restart-ability for the GPU index.

* `save_index` / `load_index`: the shard matrix (.npy, storage dtype as raw uint16/float32) plus the
  host row tables (JSON).  Files written by this module are read back with `numpy.load(...,
  allow_pickle=False)` and `json`.
* `read_chroma_wal` / `replay_wal`: ingest an existing reference deployment directly from Chroma's
  write-ahead log, the `embeddings_queue` table of `chroma.sqlite3` (schema verified on the
  reference's committed DB: seq_id, operation {0 ADD, 1 UPDATE, 2 UPSERT, 3 DELETE}, id, vector BLOB
  with encoding 'FLOAT32' (little-endian), metadata JSON with the document under
  'chroma:document').  The database is opened read-only / immutable; nothing in it is executed.
"""
from __future__ import annotations

import json
import os
import sqlite3
from dataclasses import dataclass
from typing import Any, Dict, Iterator, List, Optional

import numpy as np

OP_ADD, OP_UPDATE, OP_UPSERT, OP_DELETE = 0, 1, 2, 3


@dataclass
class WalRecord:
    seq_id: int
    operation: int
    id: str
    vector: Optional[np.ndarray]
    metadata: Optional[Dict[str, Any]]
    document: Optional[str]


def read_chroma_wal(sqlite_path: str, topic: Optional[str] = None) -> Iterator[WalRecord]:
    uri = f"file:{os.path.abspath(sqlite_path)}?mode=ro&immutable=1"
    con = sqlite3.connect(uri, uri=True)
    try:
        sql = "select seq_id, operation, id, vector, encoding, metadata, topic from embeddings_queue"
        args: tuple = ()
        if topic is not None:
            sql += " where topic = ?"
            args = (topic,)
        for seq, op, rid, blob, enc, meta, _t in con.execute(sql + " order by seq_id", args):
            vec = None
            if blob is not None:
                if enc == "FLOAT32":
                    vec = np.frombuffer(blob, dtype="<f4").astype(np.float32)
                elif enc == "INT32":
                    vec = np.frombuffer(blob, dtype="<i4").astype(np.float32)
                else:
                    raise ValueError(f"unsupported vector encoding {enc!r} at seq_id {seq}")
            md = json.loads(meta) if meta else None
            doc = md.pop("chroma:document", None) if md else None
            yield WalRecord(int(seq), int(op), rid, vec, md, doc)
    finally:
        con.close()


def replay_wal(collection, records, batch: int = 512) -> Dict[str, int]:
    """Apply WAL records in order to any collection with add/delete (VectorIndex).  Consecutive
    ADDs are batched into one device append."""
    counts = {"add": 0, "delete": 0, "update": 0, "skipped": 0}
    pend: List[WalRecord] = []

    def flush():
        if pend:
            collection.add(np.stack([r.vector for r in pend]), [r.document for r in pend],
                           [r.metadata or {} for r in pend], [r.id for r in pend])
            counts["add"] += len(pend)
            pend.clear()

    for r in records:
        if r.operation == OP_ADD and r.vector is not None:
            if any(p.id == r.id for p in pend):
                flush()
            pend.append(r)
            if len(pend) >= batch:
                flush()
        elif r.operation == OP_DELETE:
            flush()
            collection.delete(ids=[r.id])
            counts["delete"] += 1
        elif r.operation in (OP_UPDATE, OP_UPSERT) and r.vector is not None:
            flush()
            collection.delete(ids=[r.id])
            collection.add(r.vector[None, :], [r.document], [r.metadata or {}], [r.id])
            counts["update"] += 1
        else:
            counts["skipped"] += 1
    flush()
    return counts


def save_index(index, directory: str) -> None:
    """Persist a VectorIndex: matrix rows [0, n) + id/document/metadata tables."""
    import torch

    os.makedirs(directory, exist_ok=True)
    index.compact()   # tombstoned rows are not persisted
    n = index.count()
    m = index.matrix[:n].cpu()
    raw = m.view(torch.int16).numpy() if m.dtype in (torch.float16, torch.bfloat16) else m.numpy()
    np.save(os.path.join(directory, "matrix.npy"), raw, allow_pickle=False)
    with open(os.path.join(directory, "tables.json"), "w", encoding="utf-8") as f:
        json.dump({"name": index.name, "dim": index.dim, "ld": index.ld, "dtype": str(index.dtype).split(".")[-1],
                   "count": n, "metadata": index.metadata, "ids": index._ids, "documents": index._documents,
                   "metadatas": index._metadatas}, f)


def load_index(directory: str, device: str = "cuda:0"):
    import torch

    from .index import VectorIndex

    with open(os.path.join(directory, "tables.json"), encoding="utf-8") as f:
        t = json.load(f)
    dtype = getattr(torch, t["dtype"])
    raw = np.load(os.path.join(directory, "matrix.npy"), allow_pickle=False)
    idx = VectorIndex(t["dim"], dtype=dtype, device=device, capacity=max(t["count"], 256), name=t["name"],
                      metadata=t.get("metadata"))
    rows = torch.from_numpy(raw)
    if dtype in (torch.float16, torch.bfloat16):
        rows = rows.view(dtype)
    if rows.shape != (t["count"], idx.ld):
        raise ValueError(f"matrix shape {tuple(rows.shape)} does not match tables ({t['count']}, {idx.ld})")
    if t["count"]:
        idx.add_rows_device(rows.to(device), t["documents"], t["metadatas"], t["ids"])
    return idx
