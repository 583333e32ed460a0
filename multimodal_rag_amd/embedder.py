"""`EmbeddingManager`: drop-in mirror of the reference's app/utils/embedder.py for the hot path.

Same class names, constructor and method signatures, return shapes, statistics keys, retry
and error behaviour (each method cites the reference lines it follows).  What differs is what
sits underneath:

    reference                               here
    SentenceTransformer(...).encode    ->   DeviceEncoder (HIP kernels, libmmrag.so)
    chromadb collection (HNSW, CPU)    ->   VectorIndex  (fused MFMA GEMM + exact top-k, HBM)

Deliberate differences:
  * no "CUDA OOM -> fall back to CPU" path (embedder.py:231-243, :407-426): there is no second
    backend, an out-of-memory error propagates;
  * `batch_query` embeds the whole list as one batch and runs ONE [B, d] x [N, d]^T search
    instead of N concurrent single queries (embedder.py:808-815); the result list has the same
    shape, per-query failures are still reported as dicts carrying 'error';
  * distances are cosine distances (1 - cos), the space of the collection the reference
    committed;
  * `get_stats()` is provided because api.py:472 calls it.
"""
from __future__ import annotations

import asyncio
import hashlib
import logging
import time
from collections import OrderedDict
from typing import Any, Dict, List, Optional

import numpy as np

from .config import settings

logger = logging.getLogger(__name__)


class LRUCache:
    """embedder.py:26-80 (same counters, same rounding)."""

    def __init__(self, maxsize: int = 1000):
        self.cache: "OrderedDict[str, List[float]]" = OrderedDict()
        self.maxsize = maxsize
        self.hits = 0
        self.misses = 0

    def get(self, key: str) -> Optional[List[float]]:
        if key in self.cache:
            self.cache.move_to_end(key)
            self.hits += 1
            return self.cache[key]
        self.misses += 1
        return None

    def put(self, key: str, value: List[float]):
        if key in self.cache:
            self.cache.move_to_end(key)
        elif len(self.cache) >= self.maxsize:
            self.cache.popitem(last=False)
        self.cache[key] = value

    def clear(self):
        self.cache.clear()
        self.hits = 0
        self.misses = 0

    def get_stats(self) -> Dict[str, Any]:
        total = self.hits + self.misses
        hit_rate = self.hits / total if total > 0 else 0.0
        return {"size": len(self.cache), "maxsize": self.maxsize, "hits": self.hits, "misses": self.misses,
                "hit_rate": round(hit_rate, 3)}


class HipEngine:
    """The MI355X engine pair behind EmbeddingManager: tokenizer + DeviceEncoder + VectorIndex
    factory.  Constructing it without a GPU / without libmmrag.so raises."""

    def __init__(self, model_name: str, device: Optional[str] = None):
        import torch

        from . import _native
        from .encoder import PRESETS, DeviceEncoder
        from .tokenizer import HashTokenizer, NativeWordPieceTokenizer

        _native.lib()
        if not torch.cuda.is_available():
            raise RuntimeError("multimodal_rag_amd needs an MI355X (no HIP device visible); there is no CPU fallback")
        self.device = device if device and device != "cuda" else "cuda:0"
        self.device_name = "cuda"
        import os

        model_dir = settings.MMRAG_MODEL_DIR
        if model_dir:
            self.encoder = DeviceEncoder.from_local_dir(model_dir, self.device)
            vocab = os.path.join(model_dir, "vocab.txt")
            self.tokenizer = (NativeWordPieceTokenizer.from_vocab_file(vocab) if os.path.exists(vocab)
                              else HashTokenizer(self.encoder.cfg.vocab))
        else:
            if model_name not in PRESETS:
                raise ValueError(f"unknown model {model_name!r}: give MMRAG_MODEL_DIR or one of {sorted(PRESETS)}")
            logger.warning("No local checkpoint (MMRAG_MODEL_DIR unset): %s architecture with seeded random "
                           "weights and the stand-in hash tokenizer", model_name)
            self.encoder = DeviceEncoder.random_init(PRESETS[model_name], settings.MMRAG_WEIGHT_SEED, self.device)
            self.tokenizer = HashTokenizer(self.encoder.cfg.vocab)
        self.dim = self.encoder.dim
        self.max_seq_length = self.encoder.cfg.max_seq_length
        self._torch = torch

    def encode(self, texts: List[str]) -> np.ndarray:
        if hasattr(self.tokenizer, "encode_batch_arrays"):   # native, multi-threaded tokenizer
            out = self.encoder.encode_id_rows(*self.tokenizer.encode_batch_arrays(texts, self.max_seq_length))
        else:
            out = self.encoder.encode_ids([self.tokenizer.encode(t, self.max_seq_length) for t in texts])
        return out.cpu().numpy()

    def new_collection(self, name: str, metadata: Optional[Dict[str, Any]] = None):
        from .index import VectorIndex

        dtype = {"float16": self._torch.float16, "float32": self._torch.float32,
                 "bfloat16": self._torch.bfloat16}[settings.MMRAG_INDEX_DTYPE]
        return VectorIndex(self.dim, dtype=dtype, device=self.device, name=name, metadata=metadata)

    def release(self):
        self._torch.cuda.empty_cache()


CLIP_MODEL_NAMES = ("openai/clip-vit-base-patch32", "clip-ViT-B-32", "ViT-B/32")


class ClipEngine:
    """BASELINE config 4 (joint text/image space; the reference never loads CLIP): the same
    engine interface as HipEngine over both CLIP towers.  Text goes through the byte-level BPE when the
    model directory holds vocab.json + merges.txt; `encode_images` takes decoded RGB arrays of any size
    (HIP resize + centre crop, then the vision tower)."""

    def __init__(self, model_name: str, device: Optional[str] = None):
        import os

        import torch

        from . import _native
        from .clip import VIT_B32, ClipImagePreprocessor, DeviceClip
        from .tokenizer import ClipBpeTokenizer

        _native.lib()
        if not torch.cuda.is_available():
            raise RuntimeError("multimodal_rag_amd needs an MI355X (no HIP device visible); there is no CPU fallback")
        self.device = device if device and device != "cuda" else "cuda:0"
        self.device_name = "cuda"
        model_dir = settings.MMRAG_MODEL_DIR
        self.tokenizer = None
        if model_dir:
            self.clip = DeviceClip.from_local_dir(model_dir, self.device)
            vj, mt = os.path.join(model_dir, "vocab.json"), os.path.join(model_dir, "merges.txt")
            if os.path.exists(vj) and os.path.exists(mt):
                self.tokenizer = ClipBpeTokenizer.from_files(vj, mt, self.clip.cfg.t_max_pos)
        else:
            logger.warning("No local checkpoint (MMRAG_MODEL_DIR unset): CLIP ViT-B/32 architecture with seeded "
                           "random weights and stand-in token ids (%s)", model_name)
            self.clip = DeviceClip.random_init(VIT_B32, settings.MMRAG_WEIGHT_SEED, self.device)
        self.preprocess = ClipImagePreprocessor(self.device, self.clip.cfg.image)
        self.dim = self.clip.dim
        self.max_seq_length = self.clip.cfg.t_max_pos
        self._torch = torch

    def _ids(self, text: str) -> List[int]:
        if self.tokenizer is not None:
            return self.tokenizer.encode(text, self.max_seq_length)
        # stand-in (no vocabulary available): hashed word ids below the two special tokens
        c = self.clip.cfg
        words = text.lower().split()[: self.max_seq_length - 2]
        body = [int.from_bytes(hashlib.md5(w.encode()).digest()[:4], "little") % (c.vocab - 2) for w in words]
        return [c.vocab - 2] + body + [c.eos_id]

    def encode(self, texts: List[str]) -> np.ndarray:
        return self.clip.encode_text_ids([self._ids(t) for t in texts]).cpu().numpy()

    def encode_images(self, images: List[np.ndarray]) -> np.ndarray:
        return self.clip.encode_images(self.preprocess(images)).cpu().numpy()

    def new_collection(self, name: str, metadata: Optional[Dict[str, Any]] = None):
        from .index import VectorIndex

        dtype = {"float16": self._torch.float16, "float32": self._torch.float32,
                 "bfloat16": self._torch.bfloat16}[settings.MMRAG_INDEX_DTYPE]
        return VectorIndex(self.dim, dtype=dtype, device=self.device, name=name, metadata=metadata)

    def release(self):
        self._torch.cuda.empty_cache()


def _is_clip_dir(model_dir: str) -> bool:
    import json
    import os

    cfg = os.path.join(model_dir, "config.json") if model_dir else ""
    if not cfg or not os.path.exists(cfg):
        return False
    with open(cfg) as f:
        return json.load(f).get("model_type") == "clip"


def load_item_image(item: Dict[str, Any]) -> Optional[np.ndarray]:
    """RGB uint8 [H, W, 3] pixels of an image item: `path` (parser output, reference parser.py image items) or
    `raw` holding a base64 PNG/JPEG (summarizer.py:629-655 schema).  None when neither decodes."""
    import base64
    import io
    import os

    try:
        from PIL import Image
    except ImportError:
        return None
    try:
        path = item.get("path")
        if path and os.path.exists(path):
            return np.asarray(Image.open(path).convert("RGB"))
        raw = item.get("raw")
        if isinstance(raw, str) and len(raw) > 64:
            data = raw.split(",", 1)[1] if raw.startswith("data:") else raw
            return np.asarray(Image.open(io.BytesIO(base64.b64decode(data))).convert("RGB"))
    except Exception as e:  # undecodable image: fall back to embedding its summary text
        logger.warning("image item %s: cannot decode pixels (%s); embedding its summary instead", item.get("id"), e)
    return None


class EmbeddingManager:
    """embedder.py:83-930."""

    def __init__(
        self,
        batch_size: int = 32,
        enable_cache: bool = True,
        cache_size: int = 1000,
        device: Optional[str] = None,
        max_retries: int = 3,
        enable_progress_logging: bool = True,
        *,
        engine: Any = None,
    ):
        self.batch_size = batch_size
        self.enable_cache = enable_cache
        self.max_retries = max_retries
        self.enable_progress_logging = enable_progress_logging

        self.client = None
        self.collection = None
        self.text_model = None
        self.device = device
        self.is_initialized = False
        self._engine = engine
        self._encode_lock = asyncio.Lock()
        self._sleep = asyncio.sleep

        self.cache = LRUCache(maxsize=cache_size) if enable_cache else None
        self.stats = {
            "total_embeddings_created": 0,
            "total_items_stored": 0,
            "total_queries": 0,
            "cache_hits": 0,
            "cache_misses": 0,
        }

    # ------------------------------------------------------------------ lifecycle -----------
    async def initialize(self):
        """embedder.py:152-193: collection first, then the model."""
        if self.is_initialized:
            return
        try:
            await self._initialize_model()
            self.client = self._engine
            self.collection = self._engine.new_collection(
                settings.CHROMA_COLLECTION_NAME, {"description": "Multi-modal RAG embeddings"})
            self.is_initialized = True
            logger.info("EmbeddingManager initialized (device=%s, dim=%d)", self.device, self.get_embedding_dimension())
        except Exception as e:
            logger.error("Failed to initialize EmbeddingManager: %s", e, exc_info=True)
            raise

    async def _initialize_model(self):
        """embedder.py:195-243 minus the CPU fallback."""
        if self._engine is None:
            name = settings.SENTENCE_TRANSFORMER_MODEL
            factory = ClipEngine if (name in CLIP_MODEL_NAMES or _is_clip_dir(settings.MMRAG_MODEL_DIR)) else HipEngine
            self._engine = await asyncio.to_thread(factory, name, self.device)
        self.text_model = self._engine
        self.device = getattr(self._engine, "device_name", self.device or "cuda")

    async def cleanup(self):
        """embedder.py:250-264."""
        if getattr(self, "_dispatcher", None) is not None:
            await self._dispatcher.stop()
            self._dispatcher = None
        if self._engine is not None and hasattr(self._engine, "release"):
            self._engine.release()
        self.client = None
        self.collection = None
        self.text_model = None
        self.is_initialized = False
        if self.cache:
            self.cache.clear()

    # ------------------------------------------------------------------ embed ---------------
    async def embed_texts_batch(self, texts: List[str], show_progress: bool = None) -> List[List[float]]:
        """embedder.py:266-347."""
        if not self.is_initialized:
            await self.initialize()
        if not texts:
            return []
        if show_progress is None:
            show_progress = len(texts) > 100 and self.enable_progress_logging
        if show_progress:
            logger.info("Creating embeddings for %d texts...", len(texts))

        embeddings = []
        texts_to_embed = []
        cache_indices = []
        if self.cache:
            for idx, text in enumerate(texts):
                cached_embedding = self.cache.get(self._get_cache_key(text))
                if cached_embedding:  # (an empty list is treated as a miss, as in the reference :306)
                    embeddings.append((idx, cached_embedding))
                else:
                    texts_to_embed.append(text)
                    cache_indices.append(idx)
        else:
            texts_to_embed = texts
            cache_indices = list(range(len(texts)))

        if texts_to_embed:
            new_embeddings = await self._encode_batch(texts_to_embed, show_progress=show_progress)
            for idx, text, embedding in zip(cache_indices, texts_to_embed, new_embeddings):
                if self.cache:
                    self.cache.put(self._get_cache_key(text), embedding)
                embeddings.append((idx, embedding))

        embeddings.sort(key=lambda x: x[0])
        result = [emb for _, emb in embeddings]

        self.stats["total_embeddings_created"] += len(texts_to_embed)
        if self.cache:
            cache_stats = self.cache.get_stats()
            self.stats["cache_hits"] = cache_stats["hits"]
            self.stats["cache_misses"] = cache_stats["misses"]
        if show_progress:
            logger.info("Created %d new embeddings, %d from cache", len(texts_to_embed),
                        len(texts) - len(texts_to_embed))
        return result

    async def _encode_batch(self, texts: List[str], show_progress: bool = False) -> List[List[float]]:
        """embedder.py:349-383: sequential slices of `batch_size`, each in a worker thread."""
        total_batches = (len(texts) + self.batch_size - 1) // self.batch_size
        all_embeddings: List[List[float]] = []
        for batch_idx in range(total_batches):
            start_idx = batch_idx * self.batch_size
            batch_texts = texts[start_idx: min(start_idx + self.batch_size, len(texts))]
            async with self._encode_lock:  # one encoder pass at a time (shared workspace)
                batch_embeddings = await asyncio.to_thread(self._encode_sync, batch_texts)
            all_embeddings.extend(batch_embeddings)
        return all_embeddings

    def _encode_sync(self, texts: List[str]) -> List[List[float]]:
        """embedder.py:385-405; OOM fallback (:407-426) deliberately absent."""
        embeddings = self.text_model.encode(texts)
        return np.asarray(embeddings, dtype=np.float32).tolist()

    # ------------------------------------------------------------------ store ---------------
    async def embed_and_store(self, summaries: List[Dict[str, Any]], doc_id: str) -> Dict[str, int]:
        """embedder.py:428-500."""
        if not self.is_initialized:
            await self.initialize()
        if not summaries:
            logger.warning("No summaries provided for embedding")
            return {"text": 0, "table": 0, "image": 0}

        start_time = time.time()
        texts = [item["summary"] for item in summaries]
        if getattr(self.collection, "encode_fn", None) is not None and not hasattr(self._engine, "encode_images"):
            # multi-GPU serving loop with an encoder on every rank: ship the strings, each rank embeds and stores
            # the items it owns (serving.ShardedCollection.add_texts)
            counts = {"text": 0, "table": 0, "image": 0}
            for item in summaries:
                if item["type"] in counts:
                    counts[item["type"]] += 1
            await asyncio.to_thread(
                self.collection.add_texts, texts, documents=texts,
                metadatas=[{"doc_id": doc_id, "item_id": it["id"], "type": it["type"]} for it in summaries],
                ids=[f"{doc_id}_{it['id']}" for it in summaries])
            self.stats["total_items_stored"] += len(summaries)
            logger.info("Stored %d embeddings for doc %s (data-parallel ingest) in %.2fs", len(summaries), doc_id,
                        time.time() - start_time)
            return counts
        embeddings = await self.embed_texts_batch(texts, show_progress=True)
        if hasattr(self._engine, "encode_images") and settings.MMRAG_EMBED_IMAGE_PIXELS:
            # joint-space engines (CLIP, BASELINE config 4): image items are embedded from their pixels
            pix = [(i, load_item_image(it)) for i, it in enumerate(summaries) if it.get("type") == "image"]
            pix = [(i, p) for i, p in pix if p is not None]
            if pix:
                async with self._encode_lock:
                    vecs = await asyncio.to_thread(self._engine.encode_images, [p for _, p in pix])
                for (i, _), v in zip(pix, vecs):
                    embeddings[i] = v.tolist()

        documents, metadatas, ids = [], [], []
        counts = {"text": 0, "table": 0, "image": 0}
        for item, _ in zip(summaries, embeddings):
            documents.append(item["summary"])
            metadatas.append({"doc_id": doc_id, "item_id": item["id"], "type": item["type"]})
            ids.append(f"{doc_id}_{item['id']}")
            if item["type"] in counts:
                counts[item["type"]] += 1

        await self._store_with_retry(embeddings=embeddings, documents=documents, metadatas=metadatas, ids=ids)
        self.stats["total_items_stored"] += len(summaries)
        logger.info("Stored %d embeddings for doc %s (text: %d, table: %d, image: %d) in %.2fs", len(embeddings),
                    doc_id, counts["text"], counts["table"], counts["image"], time.time() - start_time)
        return counts

    async def _store_with_retry(self, embeddings, documents, metadatas, ids):
        """embedder.py:502-537: 3 attempts, 2**attempt seconds apart."""
        for attempt in range(self.max_retries):
            try:
                await asyncio.to_thread(self.collection.add, embeddings=embeddings, documents=documents,
                                        metadatas=metadatas, ids=ids)
                return
            except Exception as e:
                if attempt == self.max_retries - 1:
                    logger.error("Failed to store after %d attempts: %s", self.max_retries, e)
                    raise
                wait_time = 2 ** attempt
                logger.warning("Store attempt %d failed: %s. Retrying in %ds...", attempt + 1, e, wait_time)
                await self._sleep(wait_time)

    # ------------------------------------------------------------------ query ---------------
    def enable_dynamic_batching(self, max_batch: int = 256, max_wait_ms: float = 2.0):
        """Route query() through a micro-batching dispatcher (not in the reference):
        concurrent single queries are served by one batched encode + one batched search."""
        from .dispatcher import QueryDispatcher

        self._dispatcher = QueryDispatcher(self.batch_query, max_batch=max_batch, max_wait_ms=max_wait_ms)
        return self._dispatcher

    async def query(self, query_text: str, n_results: int = 5, filter_dict: Optional[Dict] = None) -> Dict[str, Any]:
        """embedder.py:539-583."""
        if not self.is_initialized:
            await self.initialize()
        if not query_text or not query_text.strip():
            raise ValueError("Query text cannot be empty")
        if getattr(self, "_dispatcher", None) is not None:
            return await self._dispatcher.submit(query_text, n_results, filter_dict)
        try:
            query_embeddings = await self.embed_texts_batch([query_text])
            results = await self._query_with_retry(query_embedding=query_embeddings[0], n_results=n_results,
                                                   filter_dict=filter_dict)
            self.stats["total_queries"] += 1
            return results
        except Exception as e:
            logger.error("Query failed: %s", e, exc_info=True)
            raise

    async def _query_with_retry(self, query_embedding: List[float], n_results: int,
                                filter_dict: Optional[Dict]) -> Dict[str, Any]:
        """embedder.py:585-617."""
        for attempt in range(self.max_retries):
            try:
                results = await asyncio.to_thread(
                    self.collection.query, query_embeddings=[query_embedding], n_results=n_results,
                    where=filter_dict, include=["metadatas", "documents", "distances"])
                return {
                    "ids": results["ids"][0] if results["ids"] else [],
                    "distances": results["distances"][0] if results["distances"] else [],
                    "metadatas": results["metadatas"][0] if results["metadatas"] else [],
                    "documents": results["documents"][0] if results["documents"] else [],
                }
            except Exception as e:
                if attempt == self.max_retries - 1:
                    raise
                wait_time = 2 ** attempt
                logger.warning("Query attempt %d failed: %s. Retrying in %ds...", attempt + 1, e, wait_time)
                await self._sleep(wait_time)

    async def batch_query(self, queries: List[str], n_results: int = 5,
                          filter_dict: Optional[Dict] = None) -> List[Dict[str, Any]]:
        """embedder.py:784-832, as ONE batched search (see module docstring)."""
        if not queries:
            return []
        if not self.is_initialized:
            await self.initialize()
        empty = {"ids": [], "distances": [], "metadatas": [], "documents": []}
        final: List[Optional[Dict[str, Any]]] = [None] * len(queries)
        good = []
        for i, q in enumerate(queries):
            if not q or not q.strip():
                final[i] = {**empty, "error": "Query text cannot be empty"}
            else:
                good.append(i)
        if good:
            try:
                embs = await self.embed_texts_batch([queries[i] for i in good])
                res = None
                for attempt in range(self.max_retries):
                    try:
                        res = await asyncio.to_thread(
                            self.collection.query, query_embeddings=embs, n_results=n_results, where=filter_dict,
                            include=["metadatas", "documents", "distances"])
                        break
                    except Exception:
                        if attempt == self.max_retries - 1:
                            raise
                        await self._sleep(2 ** attempt)
                for j, i in enumerate(good):
                    final[i] = {"ids": res["ids"][j], "distances": res["distances"][j],
                                "metadatas": res["metadatas"][j], "documents": res["documents"][j]}
                    self.stats["total_queries"] += 1
            except Exception as e:
                logger.error("Batch query failed: %s", e)
                for i in good:
                    final[i] = {**empty, "error": str(e)}
        return final  # type: ignore[return-value]

    # ------------------------------------------------------------------ maintenance ---------
    async def delete_document(self, doc_id: str):
        """embedder.py:619-656."""
        if not self.is_initialized:
            await self.initialize()
        for attempt in range(self.max_retries):
            try:
                results = await asyncio.to_thread(self.collection.get, where={"doc_id": doc_id}, include=[])
                if results["ids"]:
                    await asyncio.to_thread(self.collection.delete, ids=results["ids"])
                    logger.info("Deleted %d embeddings for doc %s", len(results["ids"]), doc_id)
                else:
                    logger.warning("No embeddings found for doc %s", doc_id)
                return
            except Exception as e:
                if attempt == self.max_retries - 1:
                    logger.error("Failed to delete document %s: %s", doc_id, e)
                    raise
                await self._sleep(2 ** attempt)

    async def delete_all_documents(self):
        """embedder.py:658-688: drop and re-create the collection, clear the cache."""
        if not self.is_initialized:
            await self.initialize()
        try:
            self.collection = await asyncio.to_thread(
                self._engine.new_collection, settings.CHROMA_COLLECTION_NAME,
                {"description": "Multi-modal RAG embeddings"})
            if self.cache:
                self.cache.clear()
        except Exception as e:
            logger.error("Failed to delete all documents: %s", e)
            raise

    async def get_collection_stats(self) -> Dict[str, Any]:
        """embedder.py:690-728 (same keys)."""
        if not self.is_initialized:
            await self.initialize()
        try:
            count = await asyncio.to_thread(self.collection.count)
            stats = {
                "name": settings.CHROMA_COLLECTION_NAME,
                "count": count,
                "model": settings.SENTENCE_TRANSFORMER_MODEL,
                "device": self.device,
                "embedding_dim": self.get_embedding_dimension(),
                "batch_size": self.batch_size,
                "stats": {
                    "total_embeddings_created": self.stats["total_embeddings_created"],
                    "total_items_stored": self.stats["total_items_stored"],
                    "total_queries": self.stats["total_queries"],
                },
            }
            if self.cache:
                stats["cache"] = self.cache.get_stats()
            return stats
        except Exception as e:
            logger.error("Failed to get collection stats: %s", e)
            return {"name": settings.CHROMA_COLLECTION_NAME, "count": 0, "error": str(e)}

    async def get_stats(self) -> Dict[str, Any]:
        """api.py:472 calls this name; the reference class only defines get_collection_stats."""
        return await self.get_collection_stats()

    def get_embedding_dimension(self) -> int:
        """embedder.py:730-734."""
        if self.text_model:
            return int(self.text_model.dim)
        return 384

    def _get_cache_key(self, text: str) -> str:
        """embedder.py:736-742."""
        return hashlib.md5(text.encode("utf-8")).hexdigest()

    async def warmup_cache(self, common_queries: List[str]):
        """embedder.py:744-762."""
        if not self.cache:
            logger.warning("Cache not enabled, skipping warmup")
            return
        await self.embed_texts_batch(common_queries, show_progress=False)

    async def get_cache_stats(self) -> Dict[str, Any]:
        """embedder.py:764-772."""
        if not self.cache:
            return {"enabled": False}
        return {"enabled": True, **self.cache.get_stats()}

    async def clear_cache(self):
        """embedder.py:774-780."""
        if self.cache:
            self.cache.clear()
        else:
            logger.warning("Cache not enabled")

    async def rerank_results(self, query_text: str, results: Dict[str, Any],
                             top_k: Optional[int] = None) -> Dict[str, Any]:
        """embedder.py:834-859: the reference's placeholder (truncation only)."""
        logger.warning("Re-ranking not implemented yet")
        if top_k and top_k < len(results["ids"]):
            return {k: results[k][:top_k] for k in ("ids", "distances", "metadatas", "documents")}
        return results

    async def get_similar_documents(self, doc_id: str, item_id: str, n_results: int = 5) -> Dict[str, Any]:
        """embedder.py:861-930: stored vector -> k+1 search -> drop self -> truncate."""
        if not self.is_initialized:
            await self.initialize()
        try:
            source_id = f"{doc_id}_{item_id}"
            source_data = await asyncio.to_thread(self.collection.get, ids=[source_id],
                                                  include=["embeddings", "documents"])
            if not source_data["ids"]:
                raise ValueError(f"Item not found: {source_id}")
            results = await asyncio.to_thread(
                self.collection.query, query_embeddings=[source_data["embeddings"][0]], n_results=n_results + 1,
                include=["metadatas", "documents", "distances"])
            filtered = {"ids": [], "distances": [], "metadatas": [], "documents": []}
            for i in range(len(results["ids"][0])):
                if results["ids"][0][i] != source_id:
                    for key in filtered:
                        filtered[key].append(results[key][0][i])
            for key in filtered:
                filtered[key] = filtered[key][:n_results]
            return filtered
        except Exception as e:
            logger.error("Failed to find similar documents: %s", e)
            raise
