"""Test doubles for CPU-only tests of the host logic (never used by the product path).

FakeEngine / FakeCollection stand where HipEngine / VectorIndex stand on a GPU box; the
collection answers queries with the CPU oracle so host-side plumbing can be checked end to end
without a device."""
import hashlib
from typing import Any, Dict, List, Optional

import numpy as np

from multimodal_rag_amd.index import match_where
from oracle import search_oracle as O


class FakeCollection:
    def __init__(self, dim, name="multimodal_rag", metadata=None):
        self.dim, self.name, self.metadata = dim, name, metadata or {}
        self.ids: List[str] = []
        self.docs: List[Optional[str]] = []
        self.metas: List[Dict[str, Any]] = []
        self.vecs = np.zeros((0, dim), np.float32)
        self.fail_next = 0

    def _maybe_fail(self):
        if self.fail_next > 0:
            self.fail_next -= 1
            raise RuntimeError("injected engine failure")

    def count(self):
        return len(self.ids)

    def add(self, embeddings, documents=None, metadatas=None, ids=None):
        self._maybe_fail()
        e = np.asarray(embeddings, np.float32).reshape(-1, self.dim)
        documents = list(documents or [None] * len(ids))
        metadatas = list(metadatas or [{}] * len(ids))
        have = set(self.ids)
        new = [j for j, i in enumerate(ids) if not (i in have or have.add(i))]   # duplicate ids ignored, as VectorIndex
        self.vecs = np.concatenate([self.vecs, e[new]])
        self.ids += [ids[j] for j in new]
        self.docs += [documents[j] for j in new]
        self.metas += [dict(metadatas[j]) for j in new]

    def query(self, query_embeddings, n_results=10, where=None, include=("metadatas", "documents", "distances")):
        self._maybe_fail()
        q = np.asarray(query_embeddings, np.float32).reshape(-1, self.dim)
        alive = np.array([match_where(m, where) for m in self.metas], bool) if where else None
        s, r = O.cosine_topk(q, self.vecs, n_results, alive=alive) if len(self.ids) else (
            np.zeros((len(q), 0)), np.zeros((len(q), 0), np.int64))
        out = {"ids": [], "distances": [], "metadatas": [], "documents": []}
        for b in range(len(q)):
            hit = [int(x) for x in r[b] if x >= 0]
            out["ids"].append([self.ids[i] for i in hit])
            out["distances"].append([float(1.0 - s[b, j]) for j in range(len(hit))])
            out["metadatas"].append([dict(self.metas[i]) for i in hit])
            out["documents"].append([self.docs[i] for i in hit])
        return out

    # ---- the row-level half of VectorIndex (what serving.ShardedCollection drives on each rank)
    def search(self, query_embeddings, n_results, where=None):
        q = np.asarray(query_embeddings, np.float32).reshape(-1, self.dim)
        if not self.ids:
            return np.full((len(q), n_results), -np.inf, np.float32), np.full((len(q), n_results), -1, np.int64)
        alive = np.array([match_where(m, where) for m in self.metas], bool) if where else None
        return O.cosine_topk(q, self.vecs, n_results, alive=alive)

    def ids_of_rows(self, rows):
        return [self.ids[int(r)] for r in rows]

    def reset(self):
        self.__init__(self.dim, self.name, self.metadata)

    def get(self, ids=None, where=None, include=("metadatas", "documents")):
        if ids is not None:      # requested order, as VectorIndex.get
            at = {s: i for i, s in enumerate(self.ids)}
            rows = [at[s] for s in ids if s in at and match_where(self.metas[at[s]], where)]
        else:
            rows = [i for i in range(len(self.ids)) if match_where(self.metas[i], where)]
        return {"ids": [self.ids[i] for i in rows], "metadatas": [self.metas[i] for i in rows],
                "documents": [self.docs[i] for i in rows],
                "embeddings": [self.vecs[i].tolist() for i in rows] if "embeddings" in include else None}

    def delete(self, ids=None, where=None):
        kill = set(self.get(ids=ids, where=where)["ids"])
        keep = [i for i, s in enumerate(self.ids) if s not in kill]
        self.vecs = self.vecs[keep]
        self.ids = [self.ids[i] for i in keep]
        self.docs = [self.docs[i] for i in keep]
        self.metas = [self.metas[i] for i in keep]
        return sorted(kill)


class FakeEngine:
    """Deterministic unit vectors keyed on the text; counts encode() calls and batch sizes."""
    device_name = "fake"

    def __init__(self, dim=32):
        self.dim = dim
        self.max_seq_length = 256
        self.calls: List[int] = []
        self.collections: List[FakeCollection] = []

    def encode(self, texts):
        self.calls.append(len(texts))
        out = []
        for t in texts:
            seed = int.from_bytes(hashlib.md5(t.encode()).digest()[:8], "little")
            v = np.random.default_rng(seed).standard_normal(self.dim).astype(np.float32)
            out.append(v / np.linalg.norm(v))
        return np.stack(out)

    def new_collection(self, name, metadata=None):
        c = FakeCollection(self.dim, name, metadata)
        self.collections.append(c)
        return c

    def release(self):
        pass
