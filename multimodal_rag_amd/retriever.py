"""`MultiVectorRetriever`: mirror of the reference's app/utils/retriever.py at the boundary of
the hot path (the id -> raw-content hop after search, api.py:348).

Same signatures, key layout, payload format (gzip level 6 of the JSON the reference writes),
bucketing and statistics.  The key-value engine is pluggable: the reference talks to Redis
(retriever.py:168-213), which is not available here, so the default engine is an in-process
store with the same commands the reference issues (SET/GET/DELETE/SCAN-by-pattern/pipeline).
Host-only code: nothing here touches the GPU.
"""
from __future__ import annotations

import asyncio
import fnmatch
import gzip
import json
import logging
import threading
import time
from collections import OrderedDict
from datetime import datetime
from typing import Any, Dict, List, Optional

logger = logging.getLogger(__name__)


class DocumentCache:
    """retriever.py:35-90."""

    def __init__(self, maxsize: int = 100):
        self.cache: "OrderedDict[str, Any]" = OrderedDict()
        self.maxsize = maxsize
        self.hits = 0
        self.misses = 0

    def get(self, key: str) -> Optional[Any]:
        if key in self.cache:
            self.cache.move_to_end(key)
            self.hits += 1
            return self.cache[key]
        self.misses += 1
        return None

    def put(self, key: str, value: Any):
        if key in self.cache:
            self.cache.move_to_end(key)
        elif len(self.cache) >= self.maxsize:
            self.cache.popitem(last=False)
        self.cache[key] = value

    def invalidate(self, key: str):
        if key in self.cache:
            del self.cache[key]

    def clear(self):
        self.cache.clear()
        self.hits = 0
        self.misses = 0

    def get_stats(self) -> Dict[str, Any]:
        total = self.hits + self.misses
        return {"size": len(self.cache), "maxsize": self.maxsize, "hits": self.hits, "misses": self.misses,
                "hit_rate": round(self.hits / total if total > 0 else 0.0, 3)}


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


class MultiVectorRetriever:
    """retriever.py:93-1015.

    Key format (retriever.py:97-100):
      doc:{doc_id}:{item_id} -> compressed raw content; doc_meta:{doc_id}; doc_index:{doc_id}
    """

    def __init__(
        self,
        enable_compression: bool = True,
        enable_cache: bool = True,
        cache_size: int = 100,
        max_retries: int = 3,
        connection_pool_size: int = 10,
        batch_size: int = 100,
        *,
        store: Any = None,
    ):
        self.enable_compression = enable_compression
        self.enable_cache = enable_cache
        self.max_retries = max_retries
        self.batch_size = batch_size
        self.redis_client = store
        self.connection_pool = None
        self.is_initialized = False
        self.cache = DocumentCache(maxsize=cache_size) if enable_cache else None
        self._sleep = asyncio.sleep
        self.stats = {"total_stored": 0, "total_retrieved": 0, "total_deleted": 0, "compression_ratio": 0.0,
                      "cache_hits": 0, "cache_misses": 0}

    async def initialize(self):
        """retriever.py:168-213 (connect + ping)."""
        if self.is_initialized:
            return
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

    # ------------------------------------------------------------------ store ---------------
    async def store_raw_documents(self, doc_id: str, summaries: List[Dict[str, Any]], filename: str):
        """retriever.py:238-309."""
        if not self.is_initialized:
            await self.initialize()
        total_uncompressed = 0
        total_compressed = 0
        for attempt in range(self.max_retries):
            try:
                await asyncio.to_thread(self._store_sync, doc_id, summaries, filename)
                if self.enable_compression:
                    for item in summaries:
                        raw_data = json.dumps(item)
                        total_uncompressed += len(raw_data)
                        total_compressed += len(self._compress(raw_data))
                    self.stats["compression_ratio"] = (total_compressed / total_uncompressed
                                                       if total_uncompressed > 0 else 1.0)
                self.stats["total_stored"] += len(summaries)
                return
            except Exception as e:
                if attempt == self.max_retries - 1:
                    logger.error("Failed to store after %d attempts: %s", self.max_retries, e)
                    raise
                await self._sleep(2 ** attempt)

    def _store_sync(self, doc_id: str, summaries: List[Dict[str, Any]], filename: str):
        """retriever.py:371-426 (payload fields :384-393, index :404-405, metadata :408-421)."""
        batch: Dict[str, bytes] = {}
        item_ids = []
        for item in summaries:
            data = {"id": item["id"], "type": item["type"], "raw": item["raw"], "summary": item["summary"]}
            if item["type"] == "image" and "path" in item:
                data["path"] = item["path"]
            json_data = json.dumps(data)
            batch[f"doc:{doc_id}:{item['id']}"] = (self._compress(json_data) if self.enable_compression
                                                   else json_data.encode("utf-8"))
            item_ids.append(item["id"])
        batch[f"doc_index:{doc_id}"] = json.dumps(item_ids).encode("utf-8")
        meta_data = {
            "doc_id": doc_id,
            "filename": filename,
            "item_count": len(summaries),
            "chunks": {t: sum(1 for s in summaries if s["type"] == t) for t in ("text", "table", "image")},
            "timestamp": datetime.utcnow().isoformat(),
            "compressed": self.enable_compression,
        }
        batch[f"doc_meta:{doc_id}"] = json.dumps(meta_data).encode("utf-8")
        self.redis_client.mset(batch)

    # ------------------------------------------------------------------ retrieve ------------
    async def retrieve_raw_documents(self, ids: List[str]) -> Dict[str, List[str]]:
        """retriever.py:428-531: cache, fetch, bucket raw content by type in input-id order."""
        if not self.is_initialized:
            await self.initialize()
        if not ids:
            return {"text_chunks": [], "table_chunks": [], "image_chunks": []}

        cached_items: Dict[str, Any] = {}
        ids_to_fetch: List[str] = []
        if self.cache:
            for item_id in ids:
                cached = self.cache.get(item_id)
                if cached:
                    cached_items[item_id] = cached
                else:
                    ids_to_fetch.append(item_id)
        else:
            ids_to_fetch = ids

        fetched_items: Dict[str, Any] = {}
        if ids_to_fetch:
            for attempt in range(self.max_retries):
                try:
                    fetched_items = await asyncio.to_thread(self._retrieve_sync, ids_to_fetch)
                    if self.cache:
                        for item_id, item_data in fetched_items.items():
                            self.cache.put(item_id, item_data)
                    break
                except Exception as e:
                    if attempt == self.max_retries - 1:
                        logger.error("Failed to retrieve after %d attempts: %s", self.max_retries, e)
                        raise
                    await self._sleep(2 ** attempt)

        all_items = {**cached_items, **fetched_items}
        text_chunks, table_chunks, image_chunks = [], [], []
        for item_id in ids:
            item = all_items.get(item_id)
            if item:
                if item["type"] == "text":
                    text_chunks.append(item["raw"])
                elif item["type"] == "table":
                    table_chunks.append(item["raw"])
                elif item["type"] == "image":
                    image_chunks.append(item["raw"])

        self.stats["total_retrieved"] += len(ids)
        if self.cache:
            cs = self.cache.get_stats()
            self.stats["cache_hits"] = cs["hits"]
            self.stats["cache_misses"] = cs["misses"]
        return {"text_chunks": text_chunks, "table_chunks": table_chunks, "image_chunks": image_chunks}

    def _retrieve_sync(self, ids: List[str]) -> Dict[str, Dict[str, Any]]:
        """retriever.py:576-608."""
        items: Dict[str, Dict[str, Any]] = {}
        keys = [(item_id, self._item_id_to_redis_key(item_id)) for item_id in ids]
        results = self.redis_client.mget([k for _, k in keys])
        for (item_id, _), data_bytes in zip(keys, results):
            if data_bytes:
                try:
                    json_str = self._decompress(data_bytes) if self.enable_compression else data_bytes.decode("utf-8")
                    items[item_id] = json.loads(json_str)
                except Exception as e:
                    logger.warning("Failed to decode item %s: %s", item_id, e)
        return items

    def _item_id_to_redis_key(self, item_id: str) -> str:
        """retriever.py:610-637: "doc_abc123_chunk_0_a1b2c3" -> "doc:doc_abc123:chunk_0_a1b2c3"."""
        parts = item_id.split("_")
        if len(parts) < 3:
            return f"doc:{item_id}"
        return f"doc:{'_'.join(parts[:2])}:{'_'.join(parts[2:])}"

    # ------------------------------------------------------------------ delete / list -------
    async def delete_document(self, doc_id: str):
        """retriever.py:639-675."""
        if not self.is_initialized:
            await self.initialize()
        for attempt in range(self.max_retries):
            try:
                await asyncio.to_thread(self._delete_sync, doc_id)
                if self.cache:
                    self.cache.clear()
                self.stats["total_deleted"] += 1
                return
            except Exception as e:
                if attempt == self.max_retries - 1:
                    logger.error("Failed to delete document %s: %s", doc_id, e)
                    raise
                await self._sleep(2 ** attempt)

    def _delete_sync(self, doc_id: str):
        """retriever.py:728-763."""
        index_key = f"doc_index:{doc_id}"
        index_data = self.redis_client.get(index_key)
        keys_to_delete: List[str] = []
        if index_data:
            for item_id in json.loads(index_data.decode("utf-8")):
                keys_to_delete.append(f"doc:{doc_id}:{item_id}")
            keys_to_delete.append(index_key)
        else:
            keys_to_delete.extend(self.redis_client.scan_iter(match=f"doc:{doc_id}:*"))
        keys_to_delete.append(f"doc_meta:{doc_id}")
        for i in range(0, len(keys_to_delete), self.batch_size):
            self.redis_client.delete(*keys_to_delete[i: i + self.batch_size])

    async def delete_all_documents(self):
        """retriever.py:765-787."""
        if not self.is_initialized:
            await self.initialize()
        all_keys: List[str] = []
        for pattern in ("doc:*", "doc_meta:*", "doc_index:*"):
            all_keys.extend(self.redis_client.scan_iter(match=pattern))
        for i in range(0, len(all_keys), self.batch_size):
            self.redis_client.delete(*all_keys[i: i + self.batch_size])
        if self.cache:
            self.cache.clear()

    async def list_all_documents(self) -> List[Dict[str, Any]]:
        """retriever.py:832-913: all doc_meta records, newest first."""
        if not self.is_initialized:
            await self.initialize()
        documents = []
        for key in self.redis_client.scan_iter(match="doc_meta:*"):
            meta_bytes = self.redis_client.get(key)
            if meta_bytes:
                try:
                    documents.append(json.loads(meta_bytes.decode("utf-8")))
                except Exception as e:
                    logger.warning("Failed to decode metadata: %s", e)
        documents.sort(key=lambda x: x.get("timestamp", ""), reverse=True)
        return documents

    async def get_document_metadata(self, doc_id: str) -> Optional[Dict[str, Any]]:
        """retriever.py:915-934."""
        if not self.is_initialized:
            await self.initialize()
        try:
            meta_bytes = self.redis_client.get(f"doc_meta:{doc_id}")
            return json.loads(meta_bytes.decode("utf-8")) if meta_bytes else None
        except Exception as e:
            logger.error("Failed to get metadata for %s: %s", doc_id, e)
            return None

    async def get_stats(self) -> Dict[str, Any]:
        """retriever.py:936-968 (same keys)."""
        stats = {
            "redis": {"connected": self.is_initialized, "async": False},
            "features": {"compression": self.enable_compression, "cache": self.enable_cache},
            "operations": {k: self.stats[k] for k in ("total_stored", "total_retrieved", "total_deleted")},
        }
        if self.enable_compression:
            stats["compression"] = {"ratio": self.stats["compression_ratio"],
                                    "savings_percent": (1 - self.stats["compression_ratio"]) * 100}
        if self.cache:
            stats["cache"] = self.cache.get_stats()
        return stats

    async def health_check(self) -> Dict[str, Any]:
        """retriever.py:970-1004."""
        health = {"healthy": False, "redis_connected": False, "latency_ms": None, "error": None}
        try:
            if not self.is_initialized:
                await self.initialize()
            start = time.time()
            self.redis_client.ping()
            health.update(healthy=True, redis_connected=True, latency_ms=round((time.time() - start) * 1000, 2))
        except Exception as e:
            health["error"] = str(e)
        return health

    def _compress(self, data: str) -> bytes:
        """retriever.py:1008-1010."""
        return gzip.compress(data.encode("utf-8"), compresslevel=6)

    def _decompress(self, data: bytes) -> str:
        """retriever.py:1012-1014."""
        return gzip.decompress(data).decode("utf-8")
