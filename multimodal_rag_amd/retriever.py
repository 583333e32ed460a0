"""`MultiVectorRetriever`: the reference's app/utils/retriever.py surface at the boundary of the hot path (the
id -> raw-content hop after search, api.py:348).

Signatures, key layout, stored payload (gzip level 6 of the item's JSON), bucketing by type and statistics keys are the
reference's.  The body is written against a five-command key-value protocol (ping / mset / mget / get / delete /
scan_iter) instead of a Redis client: the reference talks to Redis (retriever.py:168-213), which is not available
here, so the default engine is `InProcessKV`; any object with those commands (a Redis client adapter included) plugs
in through `store=`.  Host-only code: nothing here touches the GPU.

Key layout (retriever.py:97-100):  doc:{doc_id}:{item_id} -> item payload,  doc_index:{doc_id} -> JSON list of item
ids,  doc_meta:{doc_id} -> JSON document record.
"""
from __future__ import annotations

import asyncio
import fnmatch
import gzip
import json
import logging
import threading
import time
from datetime import datetime
from typing import Any, Dict, Iterable, List, Optional

from .hostutil import CountingLRU, call_with_retry

logger = logging.getLogger(__name__)

BUCKET_OF_TYPE = {"text": "text_chunks", "table": "table_chunks", "image": "image_chunks"}   # retriever.py:500-512


class DocumentCache(CountingLRU):
    """item id -> decoded payload (retriever.py:35-90: maxsize 100, same counters and rounding as the embedding cache)"""

    def __init__(self, maxsize: int = 100):
        super().__init__(maxsize)


class InProcessKV:
    """The Redis commands retriever.py issues, over a dict (bytes values, str keys)."""

    def __init__(self):
        self._d: Dict[str, bytes] = {}
        self._lock = threading.Lock()

    @staticmethod
    def _k(key) -> str:
        return key.decode("utf-8") if isinstance(key, (bytes, bytearray)) else key

    def ping(self) -> bool:
        return True

    def set(self, key, value):
        with self._lock:
            self._d[self._k(key)] = value if isinstance(value, (bytes, bytearray)) else str(value).encode("utf-8")

    def mset(self, items: Dict[str, bytes]):
        with self._lock:  # one lock hold == the reference's MULTI/EXEC pipeline (retriever.py:319-369)
            for k, v in items.items():
                self._d[self._k(k)] = v if isinstance(v, (bytes, bytearray)) else str(v).encode("utf-8")

    def get(self, key) -> Optional[bytes]:
        with self._lock:
            return self._d.get(self._k(key))

    def mget(self, keys) -> List[Optional[bytes]]:
        with self._lock:
            return [self._d.get(self._k(k)) for k in keys]

    def delete(self, *keys) -> int:
        n = 0
        with self._lock:
            for k in keys:
                if self._d.pop(self._k(k), None) is not None:
                    n += 1
        return n

    def scan_iter(self, match: str = "*"):
        with self._lock:
            return [k for k in self._d if fnmatch.fnmatchcase(k, match)]

    def close(self):
        pass


def _item_key(doc_id: str, item_id: str) -> str:
    return f"doc:{doc_id}:{item_id}"


def _index_key(doc_id: str) -> str:
    return f"doc_index:{doc_id}"


def _meta_key(doc_id: str) -> str:
    return f"doc_meta:{doc_id}"


class MultiVectorRetriever:
    """retriever.py:93-1015."""

    def __init__(self, enable_compression: bool = True, enable_cache: bool = True, cache_size: int = 100,
                 max_retries: int = 3, connection_pool_size: int = 10, batch_size: int = 100, *, store: Any = None):
        self.enable_compression = enable_compression
        self.enable_cache = enable_cache
        self.max_retries = max_retries
        self.batch_size = batch_size               # keys per DELETE command
        self.redis_client = store                  # (the reference's attribute name; any KV engine)
        self.connection_pool = None
        self.is_initialized = False
        self.cache = DocumentCache(maxsize=cache_size) if enable_cache else None
        self.stats = {"total_stored": 0, "total_retrieved": 0, "total_deleted": 0, "compression_ratio": 0.0,
                      "cache_hits": 0, "cache_misses": 0}
        self._sleep = asyncio.sleep

    # ------------------------------------------------------------------ lifecycle -----------
    async def initialize(self):
        """retriever.py:168-213: connect and ping."""
        if not self.is_initialized:
            if self.redis_client is None:
                self.redis_client = InProcessKV()
            self.redis_client.ping()
            self.is_initialized = True

    async def cleanup(self):
        """retriever.py:215-236."""
        if self.redis_client is not None:
            self.redis_client.close()
        self.redis_client = None
        if self.cache:
            self.cache.clear()
        self.is_initialized = False

    def _kv_call(self, what: str, fn, *args):
        return call_with_retry(what, fn, *args, attempts=self.max_retries, sleep=self._sleep, log=logger)

    # ------------------------------------------------------------------ codec ---------------
    def _compress(self, data: str) -> bytes:
        """retriever.py:1008-1010."""
        return gzip.compress(data.encode("utf-8"), compresslevel=6)

    def _decompress(self, data: bytes) -> str:
        """retriever.py:1012-1014."""
        return gzip.decompress(data).decode("utf-8")

    def _encode(self, record: Dict[str, Any]) -> bytes:
        text = json.dumps(record)
        return self._compress(text) if self.enable_compression else text.encode("utf-8")

    def _decode(self, blob: bytes) -> Dict[str, Any]:
        return json.loads(self._decompress(blob) if self.enable_compression else blob.decode("utf-8"))

    def _item_id_to_redis_key(self, item_id: str) -> str:
        """retriever.py:610-637: a collection id "doc_<12 hex>_<item id>" back to its store key -- the first two
        '_'-separated fields are the document id ("doc_abc123_chunk_0_a1b2c3" -> "doc:doc_abc123:chunk_0_a1b2c3");
        an id with fewer than three fields maps to "doc:" + id."""
        head = item_id.split("_", 2)
        return _item_key("_".join(head[:2]), head[2]) if len(head) == 3 else f"doc:{item_id}"

    # ------------------------------------------------------------------ store ---------------
    async def store_raw_documents(self, doc_id: str, summaries: List[Dict[str, Any]], filename: str):
        """retriever.py:238-426: one atomic multi-set of the item payloads (fields :384-393), the item-id index
        (:404-405) and the document record (:408-421)."""
        await self.initialize()
        records: Dict[str, bytes] = {}
        plain_bytes = packed_bytes = 0
        for item in summaries:
            payload = {field: item[field] for field in ("id", "type", "raw", "summary")}
            if item["type"] == "image" and "path" in item:
                payload["path"] = item["path"]
            blob = self._encode(payload)
            records[_item_key(doc_id, item["id"])] = blob
            plain_bytes += len(json.dumps(payload))
            packed_bytes += len(blob)
        records[_index_key(doc_id)] = json.dumps([item["id"] for item in summaries]).encode("utf-8")
        kinds = [item["type"] for item in summaries]
        records[_meta_key(doc_id)] = json.dumps({
            "doc_id": doc_id, "filename": filename, "item_count": len(summaries),
            "chunks": {kind: kinds.count(kind) for kind in BUCKET_OF_TYPE},
            "timestamp": datetime.utcnow().isoformat(), "compressed": self.enable_compression,
        }).encode("utf-8")
        await self._kv_call("Store", self.redis_client.mset, records)
        if self.enable_compression:
            self.stats["compression_ratio"] = packed_bytes / plain_bytes if plain_bytes else 1.0
        self.stats["total_stored"] += len(summaries)

    # ------------------------------------------------------------------ retrieve ------------
    def _fetch(self, ids: List[str]) -> Dict[str, Dict[str, Any]]:
        found = {}
        for item_id, blob in zip(ids, self.redis_client.mget([self._item_id_to_redis_key(i) for i in ids])):
            if blob:
                try:
                    found[item_id] = self._decode(blob)
                except Exception as e:   # noqa: BLE001 -- a damaged record must not fail the whole answer
                    logger.warning("Failed to decode item %s: %s", item_id, e)
        return found

    async def retrieve_raw_documents(self, ids: List[str]) -> Dict[str, List[str]]:
        """retriever.py:428-608: the `raw` field of every id that exists, bucketed by item type, in the order of
        `ids`; cache first, one multi-get for the rest."""
        await self.initialize()
        buckets: Dict[str, List[str]] = {name: [] for name in BUCKET_OF_TYPE.values()}
        if not ids:
            return buckets
        known: Dict[str, Dict[str, Any]] = {}
        if self.cache:
            for item_id in ids:
                hit = self.cache.get(item_id)
                if hit:
                    known[item_id] = hit
        missing = [item_id for item_id in ids if item_id not in known]
        if missing:
            fetched = await self._kv_call("Retrieve", self._fetch, missing)
            if self.cache:
                for item_id, record in fetched.items():
                    self.cache.put(item_id, record)
            known.update(fetched)
        for item_id in ids:
            record = known.get(item_id)
            if record and record["type"] in BUCKET_OF_TYPE:
                buckets[BUCKET_OF_TYPE[record["type"]]].append(record["raw"])
        self.stats["total_retrieved"] += len(ids)
        if self.cache:
            self.stats["cache_hits"], self.stats["cache_misses"] = self.cache.hits, self.cache.misses
        return buckets

    # ------------------------------------------------------------------ delete / list -------
    def _delete_keys(self, keys: Iterable[str]):
        keys = list(keys)
        for lo in range(0, len(keys), self.batch_size):
            self.redis_client.delete(*keys[lo: lo + self.batch_size])

    def _drop_document(self, doc_id: str):
        listed = self.redis_client.get(_index_key(doc_id))
        if listed:
            keys = [_item_key(doc_id, item_id) for item_id in json.loads(listed.decode("utf-8"))] + [_index_key(doc_id)]
        else:                                       # no index record: find the items by pattern (retriever.py:748-752)
            keys = list(self.redis_client.scan_iter(match=_item_key(doc_id, "*")))
        self._delete_keys(keys + [_meta_key(doc_id)])

    async def delete_document(self, doc_id: str):
        """retriever.py:639-763."""
        await self.initialize()
        await self._kv_call(f"Delete of document {doc_id}", self._drop_document, doc_id)
        if self.cache:
            self.cache.clear()
        self.stats["total_deleted"] += 1

    async def delete_all_documents(self):
        """retriever.py:765-787."""
        await self.initialize()
        self._delete_keys(key for pattern in ("doc:*", "doc_meta:*", "doc_index:*")
                          for key in self.redis_client.scan_iter(match=pattern))
        if self.cache:
            self.cache.clear()

    def _json_at(self, key: str) -> Optional[Dict[str, Any]]:
        blob = self.redis_client.get(key)
        return json.loads(blob.decode("utf-8")) if blob else None

    async def list_all_documents(self) -> List[Dict[str, Any]]:
        """retriever.py:832-913: every document record, newest first."""
        await self.initialize()
        found = []
        for key in self.redis_client.scan_iter(match="doc_meta:*"):
            try:
                record = self._json_at(key)
            except Exception as e:   # noqa: BLE001
                logger.warning("Failed to decode metadata: %s", e)
                continue
            if record:
                found.append(record)
        return sorted(found, key=lambda record: record.get("timestamp", ""), reverse=True)

    async def get_document_metadata(self, doc_id: str) -> Optional[Dict[str, Any]]:
        """retriever.py:915-934."""
        await self.initialize()
        try:
            return self._json_at(_meta_key(doc_id))
        except Exception as e:   # noqa: BLE001
            logger.error("Failed to get metadata for %s: %s", doc_id, e)
            return None

    async def get_stats(self) -> Dict[str, Any]:
        """retriever.py:936-968 (same keys)."""
        report: Dict[str, Any] = {
            "redis": {"connected": self.is_initialized, "async": False},
            "features": {"compression": self.enable_compression, "cache": self.enable_cache},
            "operations": {key: self.stats[key] for key in ("total_stored", "total_retrieved", "total_deleted")},
        }
        if self.enable_compression:
            ratio = self.stats["compression_ratio"]
            report["compression"] = {"ratio": ratio, "savings_percent": (1 - ratio) * 100}
        if self.cache:
            report["cache"] = self.cache.get_stats()
        return report

    async def health_check(self) -> Dict[str, Any]:
        """retriever.py:970-1004."""
        report = {"healthy": False, "redis_connected": False, "latency_ms": None, "error": None}
        try:
            await self.initialize()
            t0 = time.time()
            self.redis_client.ping()
            report.update(healthy=True, redis_connected=True, latency_ms=round((time.time() - t0) * 1000, 2))
        except Exception as e:   # noqa: BLE001
            report["error"] = str(e)
        return report
