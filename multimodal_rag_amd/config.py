"""Settings for the hot path: the subset of the reference's `config.Settings`
(config.py:18-132) that the embed/retrieve path reads, with the same key names, defaults and
environment-variable overrides.  Extra keys (MMRAG_*) select the MI355X engine's options."""
from __future__ import annotations

import os
from dataclasses import dataclass, field


def _b(name: str, default: str) -> bool:
    return os.getenv(name, default).lower() == "true"


@dataclass
class Settings:
    # config.py:58-59
    CHROMA_PERSIST_DIR: str = field(default_factory=lambda: os.getenv("CHROMA_PERSIST_DIR", "./chroma_db"))
    CHROMA_COLLECTION_NAME: str = field(default_factory=lambda: os.getenv("CHROMA_COLLECTION_NAME", "multimodal_rag"))
    # config.py:64-66
    CHUNK_SIZE: int = field(default_factory=lambda: int(os.getenv("CHUNK_SIZE", "1000")))
    CHUNK_OVERLAP: int = field(default_factory=lambda: int(os.getenv("CHUNK_OVERLAP", "200")))
    TOP_K_RESULTS: int = field(default_factory=lambda: int(os.getenv("TOP_K_RESULTS", "5")))
    # config.py:79-81 (api.py:92-95 does not read them; kept for parity)
    EMBEDDER_BATCH_SIZE: int = field(default_factory=lambda: int(os.getenv("EMBEDDER_BATCH_SIZE", "32")))
    EMBEDDER_CACHE_SIZE: int = field(default_factory=lambda: int(os.getenv("EMBEDDER_CACHE_SIZE", "1000")))
    EMBEDDER_ENABLE_CACHE: bool = field(default_factory=lambda: _b("EMBEDDER_ENABLE_CACHE", "true"))
    # config.py:86-89
    RETRIEVER_ENABLE_COMPRESSION: bool = field(default_factory=lambda: _b("RETRIEVER_ENABLE_COMPRESSION", "true"))
    RETRIEVER_ENABLE_CACHE: bool = field(default_factory=lambda: _b("RETRIEVER_ENABLE_CACHE", "true"))
    RETRIEVER_CACHE_SIZE: int = field(default_factory=lambda: int(os.getenv("RETRIEVER_CACHE_SIZE", "100")))
    # config.py:102-106
    SENTENCE_TRANSFORMER_MODEL: str = field(
        default_factory=lambda: os.getenv("SENTENCE_TRANSFORMER_MODEL", "all-MiniLM-L6-v2"))
    CLIP_MODEL: str = field(default_factory=lambda: os.getenv("CLIP_MODEL", "ViT-B/32"))
    # config.py:117-119
    LOG_LEVEL: str = field(default_factory=lambda: os.getenv("LOG_LEVEL", "INFO"))
    ENABLE_CORS: bool = field(default_factory=lambda: _b("ENABLE_CORS", "true"))
    MAX_UPLOAD_SIZE: int = field(default_factory=lambda: int(os.getenv("MAX_UPLOAD_SIZE", "50")))  # MB

    # ---- engine options (not in the reference) ----
    # local Hugging Face style directory with config.json + model.safetensors + vocab.txt; when
    # empty the named architecture is random-initialised and a stand-in tokenizer is used
    MMRAG_MODEL_DIR: str = field(default_factory=lambda: os.getenv("MMRAG_MODEL_DIR", ""))
    MMRAG_INDEX_DTYPE: str = field(default_factory=lambda: os.getenv("MMRAG_INDEX_DTYPE", "float16"))
    # float32 collections: batches of more than 64 queries are scored on the bf16 matrix pipe from a three-term split
    # of the float32 operands (|score error| <= 4e-5: approximate, and a query's score then depends on how many
    # requests the dispatcher batched it with).  "true" keeps the exact float32 matrix instruction for every batch
    # size (the batch is scanned 64 queries at a time: identical scores whatever the batch, ~2.5x the scan time)
    MMRAG_F32_EXACT_SEARCH: bool = field(default_factory=lambda: _b("MMRAG_F32_EXACT_SEARCH", "false"))
    # Row tables (ids, documents, metadata dicts) are millions of long-lived Python objects; every older-generation pass
    # of the cyclic collector walks them, and the lists a query batch builds trigger such passes: half of the 1 ms a
    # 256-query answer takes to build.  After this many rows have been added since the last time, the index calls
    # gc.freeze(): everything alive moves to the permanent generation (still freed by reference counting; no longer
    # scanned for cycles).  0 turns it off.
    MMRAG_GC_FREEZE_ROWS: int = field(default_factory=lambda: int(os.getenv("MMRAG_GC_FREEZE_ROWS", "100000")))
    MMRAG_WEIGHT_SEED: int = field(default_factory=lambda: int(os.getenv("MMRAG_WEIGHT_SEED", "0")))
    # "fp16" (throughput path) or "fp32": the reference's own arithmetic (SentenceTransformer.encode is float32,
    # embedder.py:397-403) -- scores within 1e-4 of the float32 model; pair it with MMRAG_INDEX_DTYPE=float32
    MMRAG_ENCODER_PRECISION: str = field(default_factory=lambda: os.getenv("MMRAG_ENCODER_PRECISION", "fp16"))
    # keep the collection across restarts: load it from CHROMA_PERSIST_DIR/mmrag_index in initialize(), save it there in
    # cleanup().  Off by default, as in the reference's current code: its chromadb.Client(Settings(persist_directory=...))
    # lacks is_persistent=True, i.e. the reference's collection is in-memory too
    MMRAG_PERSIST: bool = field(default_factory=lambda: _b("MMRAG_PERSIST", "false"))
    # CLIP engines only: embed image items from their pixels (vision tower) instead of their summary text
    MMRAG_EMBED_IMAGE_PIXELS: bool = field(default_factory=lambda: _b("MMRAG_EMBED_IMAGE_PIXELS", "true"))


settings = Settings()
