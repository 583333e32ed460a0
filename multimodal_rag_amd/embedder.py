"""`EmbeddingManager`: the reference's app/utils/embedder.py surface over the MI355X engines.

Class names, constructor and method signatures, return shapes, statistics keys, messages and the retry / error
conventions are the reference's (each method names the lines it answers to).  The bodies are this package's own:
vectors travel as numpy matrices between the encoder and the collection (Python float lists exist only where a
signature returns them), every engine call goes through one retry helper, and both caches are one LRU class.

    reference                               here
    SentenceTransformer(...).encode    ->   DeviceEncoder (HIP kernels, libmmrag.so)            engines.HipEngine
    chromadb collection (HNSW, CPU)    ->   VectorIndex  (fused MFMA GEMM + exact top-k, HBM)   index.VectorIndex

Deliberate differences:
  * no "CUDA OOM -> fall back to CPU" path (embedder.py:231-243, :407-426): there is no second backend, an
    out-of-memory error propagates;
  * `batch_query` embeds the whole list in one pass and runs ONE [B, d] x [N, d]^T search instead of N concurrent
    single queries (embedder.py:808-815); same result list, per-query failures still become dicts carrying 'error';
  * distances are cosine distances (1 - cos), the space of the collection the reference committed;
  * `get_stats()` exists because api.py:472 calls it.
"""
from __future__ import annotations

import asyncio
import hashlib
import logging
import time
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from . import tracing
from .config import settings
from .engines import CLIP_MODEL_NAMES, ClipEngine, HipEngine, _is_clip_dir, load_item_image  # noqa: F401 (re-exported)
from .hostutil import CountingLRU, call_with_retry, load_hostrows

logger = logging.getLogger(__name__)

ITEM_KINDS = ("text", "table", "image")           # the kinds embed_and_store counts (embedder.py:477-479)
RESULT_KEYS = ("ids", "distances", "metadatas", "documents")
_HOSTROWS = load_hostrows()
_COLLECTION_NOTE = {"description": "Multi-modal RAG embeddings"}


class LRUCache(CountingLRU):
    """text -> embedding cache (embedder.py:26-80: maxsize 1000, hit/miss counters, hit rate rounded to 3 places)"""

    def __init__(self, maxsize: int = 1000):
        super().__init__(maxsize)


class EmbeddingManager:
    """embedder.py:83-930."""

    def __init__(self, batch_size: int = 32, enable_cache: bool = True, cache_size: int = 1000,
                 device: Optional[str] = None, max_retries: int = 3, enable_progress_logging: bool = True,
                 *, engine: Any = None):
        self.batch_size = batch_size
        self.enable_cache = enable_cache
        self.max_retries = max_retries
        self.enable_progress_logging = enable_progress_logging
        self.device = device
        self.client = None
        self.collection = None
        self.text_model = None
        self.is_initialized = False
        self.cache = LRUCache(maxsize=cache_size) if enable_cache else None
        self.stats = dict.fromkeys(("total_embeddings_created", "total_items_stored", "total_queries", "cache_hits",
                                    "cache_misses"), 0)
        self._engine = engine
        self._dispatcher = None
        self._encode_lock = asyncio.Lock()       # one encoder pass at a time from this event loop
        self._sleep = asyncio.sleep              # (tests swap the back-off sleep out)

    # ------------------------------------------------------------------ lifecycle -----------
    async def initialize(self):
        """embedder.py:152-248: bring up the model and the collection (idempotent)."""
        if self.is_initialized:
            return
        try:
            if self._engine is None:
                name = settings.SENTENCE_TRANSFORMER_MODEL
                joint = name in CLIP_MODEL_NAMES or _is_clip_dir(settings.MMRAG_MODEL_DIR)
                self._engine = await asyncio.to_thread(ClipEngine if joint else HipEngine, name, self.device)
            self.text_model = self.client = self._engine
            self.device = getattr(self._engine, "device_name", self.device or "cuda")
            self.collection = self._engine.new_collection(settings.CHROMA_COLLECTION_NAME, dict(_COLLECTION_NOTE))
            if settings.MMRAG_PERSIST:
                self._restore_collection()
            self.is_initialized = True
            logger.info("EmbeddingManager initialized (device=%s, dim=%d)", self.device, self.get_embedding_dimension())
        except Exception as e:
            logger.error("Failed to initialize EmbeddingManager: %s", e, exc_info=True)
            raise

    async def _ready(self):
        if not self.is_initialized:
            await self.initialize()

    def _persist_dir(self) -> str:
        import os

        return os.path.join(settings.CHROMA_PERSIST_DIR, "mmrag_index")

    def _restore_collection(self):
        """MMRAG_PERSIST: pick up what the last cleanup() saved (a collection with load(), or a VectorIndex directory)"""
        import os

        where = self._persist_dir()
        if not os.path.isdir(where):
            return
        if hasattr(self.collection, "load"):                 # serving.ShardedCollection
            self.collection.load(where)
        elif os.path.exists(os.path.join(where, "tables.json")):
            from .persistence import load_index

            self.collection = load_index(where, device=str(getattr(self.collection, "device", "cuda:0")))
        logger.info("Restored %d rows from %s", self.collection.count(), where)

    def _save_collection(self):
        import os

        where = self._persist_dir()
        os.makedirs(where, exist_ok=True)
        if hasattr(self.collection, "save"):
            self.collection.save(where)
        elif hasattr(self.collection, "matrix"):
            from .persistence import save_index

            save_index(self.collection, where)

    async def cleanup(self):
        """embedder.py:250-264."""
        if settings.MMRAG_PERSIST and self.collection is not None:
            try:
                await asyncio.to_thread(self._save_collection)
            except Exception as e:   # noqa: BLE001 -- shutting down: report, do not mask the shutdown
                logger.error("Could not save the collection: %s", e)
        if self._dispatcher is not None:
            await self._dispatcher.stop()
            self._dispatcher = None
        release = getattr(self._engine, "release", None)
        if release is not None:
            release()
        self.client = self.collection = self.text_model = None
        self.is_initialized = False
        if self.cache:
            self.cache.clear()

    def _engine_call(self, what: str, fn, *args, **kwargs):
        return call_with_retry(what, fn, *args, attempts=self.max_retries, sleep=self._sleep, log=logger, **kwargs)

    # ------------------------------------------------------------------ embed ---------------
    def _get_cache_key(self, text: str) -> str:
        """embedder.py:736-742."""
        return hashlib.md5(text.encode("utf-8")).hexdigest()

    def _lookup(self, texts: Sequence[str]):
        """cache pass over `texts`: (rows with the hits filled in, positions still to encode, their cache keys).
        A cached value of length 0 counts as a miss, as the reference's truthiness test does (embedder.py:306)."""
        rows: List[Optional[np.ndarray]] = [None] * len(texts)
        todo: List[int] = []
        keys: Dict[int, str] = {}
        for at, text in enumerate(texts):
            if self.cache:
                keys[at] = self._get_cache_key(text)
                seen = self.cache.get(keys[at])
                if seen is not None and len(seen):
                    rows[at] = np.asarray(seen, dtype=np.float32)
                    continue
            todo.append(at)
        return rows, todo, keys

    def _encode_into(self, texts: Sequence[str], rows, part: List[int], keys: Dict[int, str]):
        """one encoder pass (blocking): the texts at positions `part` -> their rows, and into the cache"""
        fresh = np.asarray(self.text_model.encode([texts[at] for at in part]), dtype=np.float32)
        for at, row in zip(part, fresh):
            rows[at] = row
            if self.cache:
                self.cache.put(keys[at], row)

    def _stack(self, rows, n_new: int) -> np.ndarray:
        self.stats["total_embeddings_created"] += n_new
        if self.cache:
            self.stats["cache_hits"], self.stats["cache_misses"] = self.cache.hits, self.cache.misses
        if not rows:
            return np.zeros((0, self.get_embedding_dimension()), dtype=np.float32)
        return np.stack(rows)

    async def _embed_matrix(self, texts: Sequence[str], rows_per_pass: int) -> np.ndarray:
        """[len(texts), dim] float32: cached rows reused, the rest encoded `rows_per_pass` texts at a time (each pass
        in a worker thread), every row at its text's position (embedder.py:301-332 partitions, encodes the misses,
        caches them, restores the order)."""
        rows, todo, keys = self._lookup(texts)
        step = max(1, rows_per_pass)
        for lo in range(0, len(todo), step):
            async with self._encode_lock:
                await asyncio.to_thread(self._encode_into, texts, rows, todo[lo: lo + step], keys)
        return self._stack(rows, len(todo))

    async def embed_texts_batch(self, texts: List[str], show_progress: bool = None) -> List[List[float]]:
        """embedder.py:266-383: lists of Python floats, input order, misses encoded in slices of `batch_size`."""
        await self._ready()
        if not texts:
            return []
        loud = (len(texts) > 100 and self.enable_progress_logging) if show_progress is None else show_progress
        if loud:
            logger.info("Creating embeddings for %d texts...", len(texts))
        before = self.stats["total_embeddings_created"]
        matrix = await self._embed_matrix(texts, self.batch_size)
        if loud:
            made = self.stats["total_embeddings_created"] - before
            logger.info("Created %d new embeddings, %d from cache", made, len(texts) - made)
        return matrix.tolist()

    # ------------------------------------------------------------------ store ---------------
    async def embed_and_store(self, summaries: List[Dict[str, Any]], doc_id: str) -> Dict[str, int]:
        """embedder.py:428-500: embed every item's `summary`, store it under f"{doc_id}_{item id}"."""
        await self._ready()
        counts = dict.fromkeys(ITEM_KINDS, 0)
        if not summaries:
            logger.warning("No summaries provided for embedding")
            return counts
        t0 = time.time()
        texts = [item["summary"] for item in summaries]
        ids = [f"{doc_id}_{item['id']}" for item in summaries]
        metas = [{"doc_id": doc_id, "item_id": item["id"], "type": item["type"]} for item in summaries]
        for item in summaries:
            if item["type"] in counts:           # other kinds are stored but not counted (:477-479)
                counts[item["type"]] += 1
        joint = hasattr(self._engine, "encode_images")
        if getattr(self.collection, "encode_fn", None) is not None and not joint:
            # multi-GPU serving loop with an encoder on every rank: ship the strings, each rank embeds and stores the
            # items it owns (serving.ShardedCollection.add_texts) -- no vector leaves its GPU
            await asyncio.to_thread(self.collection.add_texts, texts, documents=texts, metadatas=metas, ids=ids)
        else:
            matrix = await self._embed_matrix(texts, self.batch_size)
            if joint and settings.MMRAG_EMBED_IMAGE_PIXELS:
                # joint-space engines (CLIP, BASELINE config 4): image items are embedded from their pixels
                pixels = {at: load_item_image(item) for at, item in enumerate(summaries) if item.get("type") == "image"}
                pixels = {at: px for at, px in pixels.items() if px is not None}
                if pixels:
                    async with self._encode_lock:
                        vecs = await asyncio.to_thread(self._engine.encode_images, list(pixels.values()))
                    matrix[list(pixels)] = np.asarray(vecs, dtype=np.float32)
            await self._store_with_retry(embeddings=matrix, documents=texts, metadatas=metas, ids=ids)
        self.stats["total_items_stored"] += len(summaries)
        logger.info("Stored %d embeddings for doc %s (text: %d, table: %d, image: %d) in %.2fs", len(summaries), doc_id,
                    counts["text"], counts["table"], counts["image"], time.time() - t0)
        return counts

    async def _store_with_retry(self, embeddings, documents, metadatas, ids):
        """embedder.py:502-537."""
        await self._engine_call("Store", self.collection.add, embeddings=embeddings, documents=documents,
                                metadatas=metadatas, ids=ids)

    # ------------------------------------------------------------------ query ---------------
    def enable_dynamic_batching(self, max_batch: int = 256, max_wait_ms: float = 2.0):
        """Route query() through a micro-batching dispatcher (not in the reference): concurrent single
        queries are served by one batched encode + one batched search."""
        from .dispatcher import QueryDispatcher

        self._dispatcher = QueryDispatcher(self.batch_query, max_batch=max_batch, max_wait_ms=max_wait_ms)
        return self._dispatcher

    _INCLUDE = ["metadatas", "documents", "distances"]

    @staticmethod
    def _split(res: Dict[str, Any], n: int) -> List[Dict[str, Any]]:
        """collection.query's lists of lists -> one dict per query (embedder.py:604-609)"""
        cols = [res[key] if res.get(key) else [[] for _ in range(n)] for key in RESULT_KEYS]
        if _HOSTROWS is not None and all(type(col) is list and len(col) == n for col in cols):
            return _HOSTROWS.split(RESULT_KEYS, tuple(cols))
        return [dict(zip(RESULT_KEYS, per_query)) for per_query in zip(*(col[:n] for col in cols))]

    def _answer(self, texts: Sequence[str], n_results: int, filter_dict: Optional[Dict]) -> List[Dict[str, Any]]:
        """Blocking, runs in ONE worker thread: cache lookups, ONE encoder pass for the misses, ONE collection.query
        for all of them (embedder.py:566 + :595-601).  One thread hop per request instead of one per stage."""
        rows, todo, keys = self._lookup(texts)
        if (len(todo) == len(texts) and texts and hasattr(self.text_model, "encode_device")
                and getattr(self.collection, "accepts_device_queries", False)):
            # nothing cached: the embeddings go from the encoder to the search kernel on the device; the host does not
            # wait in between and reads them back (for the cache) only after the answer is there
            dev = self.text_model.encode_device(list(texts))
            res = self.collection.query(query_embeddings=dev, n_results=n_results, where=filter_dict,
                                        include=self._INCLUDE, check_norm=False)
            if self.cache:
                for at, row in zip(todo, dev.cpu().numpy()):
                    self.cache.put(keys[at], row)
            self._stack([], len(todo))
            with tracing.stage("split"):
                return self._split(res, len(texts))
        if todo:
            self._encode_into(texts, rows, todo, keys)
        matrix = self._stack(rows, len(todo))
        res = self.collection.query(query_embeddings=matrix, n_results=n_results, where=filter_dict,
                                    include=self._INCLUDE)
        return self._split(res, len(texts))

    async def _query_with_retry(self, query_embedding: List[float], n_results: int,
                                filter_dict: Optional[Dict]) -> Dict[str, Any]:
        """embedder.py:585-617: search for one ready-made vector."""
        matrix = np.asarray([query_embedding], dtype=np.float32)
        res = await self._engine_call("Query", self.collection.query, query_embeddings=matrix, n_results=n_results,
                                      where=filter_dict, include=self._INCLUDE)
        return self._split(res, 1)[0]

    async def query(self, query_text: str, n_results: int = 5, filter_dict: Optional[Dict] = None) -> Dict[str, Any]:
        """embedder.py:539-583."""
        await self._ready()
        if not query_text or not query_text.strip():
            raise ValueError("Query text cannot be empty")
        if self._dispatcher is not None:
            return await self._dispatcher.submit(query_text, n_results, filter_dict)
        try:
            hit = (await self._engine_call("Query", self._answer, [query_text], n_results, filter_dict))[0]
        except Exception as e:
            logger.error("Query failed: %s", e, exc_info=True)
            raise
        self.stats["total_queries"] += 1
        return hit

    async def batch_query(self, queries: List[str], n_results: int = 5,
                          filter_dict: Optional[Dict] = None) -> List[Dict[str, Any]]:
        """embedder.py:784-832: one result dict per query, input order; a query that cannot be answered gets a dict
        with empty lists and an 'error' message instead of an exception (:817-830)."""
        await self._ready()

        def failed(why: str) -> Dict[str, Any]:
            return {**{key: [] for key in RESULT_KEYS}, "error": why}

        answers: List[Optional[Dict[str, Any]]] = [None] * len(queries)
        live = [at for at, q in enumerate(queries) if q and q.strip()]
        for at in set(range(len(queries))) - set(live):
            answers[at] = failed("Query text cannot be empty")
        if live:
            try:
                hits = await self._engine_call("Batch query", self._answer, [queries[at] for at in live], n_results,
                                               filter_dict)
                for at, hit in zip(live, hits):
                    answers[at] = hit
                self.stats["total_queries"] += len(live)
            except Exception as e:
                logger.error("Batch query failed: %s", e)
                for at in live:
                    answers[at] = failed(str(e))
        return answers  # type: ignore[return-value]

    async def get_similar_documents(self, doc_id: str, item_id: str, n_results: int = 5) -> Dict[str, Any]:
        """embedder.py:861-930: the stored vector of one item searched for its n nearest OTHER items."""
        await self._ready()
        try:
            me = f"{doc_id}_{item_id}"
            stored = await asyncio.to_thread(self.collection.get, ids=[me], include=["embeddings", "documents"])
            if not stored["ids"]:
                raise ValueError(f"Item not found: {me}")
            near = await asyncio.to_thread(self.collection.query, query_embeddings=[stored["embeddings"][0]],
                                           n_results=n_results + 1, include=["metadatas", "documents", "distances"])
            others = [j for j, found in enumerate(near["ids"][0]) if found != me][:n_results]
            return {key: [near[key][0][j] for j in others] for key in RESULT_KEYS}
        except Exception as e:
            logger.error("Failed to find similar documents: %s", e)
            raise

    async def rerank_results(self, query_text: str, results: Dict[str, Any],
                             top_k: Optional[int] = None) -> Dict[str, Any]:
        """embedder.py:834-859: the reference's placeholder -- no re-ranking, only truncation to top_k."""
        logger.warning("Re-ranking not implemented yet")
        if top_k and top_k < len(results["ids"]):
            return {key: results[key][:top_k] for key in RESULT_KEYS}
        return results

    # ------------------------------------------------------------------ maintenance ---------
    async def delete_document(self, doc_id: str):
        """embedder.py:619-656: drop every row whose metadata carries this doc_id."""
        await self._ready()
        gone = await self._engine_call(f"Delete of document {doc_id}", self.collection.delete, where={"doc_id": doc_id})
        if gone:
            logger.info("Deleted %d embeddings for doc %s", len(gone), doc_id)
        else:
            logger.warning("No embeddings found for doc %s", doc_id)

    async def delete_all_documents(self):
        """embedder.py:658-688: a fresh collection under the same name, and an empty cache."""
        await self._ready()
        try:
            self.collection = await asyncio.to_thread(self._engine.new_collection, settings.CHROMA_COLLECTION_NAME,
                                                      dict(_COLLECTION_NOTE))
        except Exception as e:
            logger.error("Failed to delete all documents: %s", e)
            raise
        if self.cache:
            self.cache.clear()

    async def get_collection_stats(self) -> Dict[str, Any]:
        """embedder.py:690-728 (same keys; an engine failure is reported in the dict, not raised)."""
        await self._ready()
        try:
            report = {
                "name": settings.CHROMA_COLLECTION_NAME,
                "count": await asyncio.to_thread(self.collection.count),
                "model": settings.SENTENCE_TRANSFORMER_MODEL,
                "device": self.device,
                "embedding_dim": self.get_embedding_dimension(),
                "batch_size": self.batch_size,
                "stats": {key: self.stats[key] for key in ("total_embeddings_created", "total_items_stored",
                                                           "total_queries")},
            }
        except Exception as e:
            logger.error("Failed to get collection stats: %s", e)
            return {"name": settings.CHROMA_COLLECTION_NAME, "count": 0, "error": str(e)}
        if self.cache:
            report["cache"] = self.cache.get_stats()
        return report

    def get_stage_timers(self) -> Dict[str, Dict[str, float]]:
        """Not in the reference (it only logs time.time() deltas): wall clock per stage of this process' requests --
        tokenize / encode / search / collect, shard.* in the sharded service -- see tracing.py."""
        return tracing.snapshot()

    async def get_stats(self) -> Dict[str, Any]:
        """api.py:472 calls this name; the reference class only defines get_collection_stats."""
        return await self.get_collection_stats()

    def get_embedding_dimension(self) -> int:
        """embedder.py:730-734 (384 while no model is loaded)."""
        return int(self.text_model.dim) if self.text_model else 384

    async def warmup_cache(self, common_queries: List[str]):
        """embedder.py:744-762."""
        if not self.cache:
            logger.warning("Cache not enabled, skipping warmup")
            return
        await self.embed_texts_batch(common_queries, show_progress=False)

    async def get_cache_stats(self) -> Dict[str, Any]:
        """embedder.py:764-772."""
        return {"enabled": True, **self.cache.get_stats()} if self.cache else {"enabled": False}

    async def clear_cache(self):
        """embedder.py:774-780."""
        if self.cache:
            self.cache.clear()
        else:
            logger.warning("Cache not enabled")
