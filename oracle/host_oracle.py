"""CPU oracle for the host-side (non-arithmetic) pieces of the hot path.  TEST
INFRASTRUCTURE ONLY: imported by tests/ (and nothing in multimodal_rag_amd/).

Each function is a literal restatement of one reference routine, cited per function.
Pinned by the reference's own docstring examples and committed data:
  * retriever.py:617-620 docstring: "doc_abc123_chunk_0_a1b2c3" -> "doc:doc_abc123:chunk_0_a1b2c3"
  * data/sample_document.txt (563 chars after newline translation) -> exactly one chunk
    
"""
from __future__ import annotations

import hashlib
from typing import List


def basic_chunk_text(text: str, chunk_size: int = 1000, chunk_overlap: int = 200) -> List[str]:
    """parser.py:1702-1736 `_basic_chunk_text` (defaults: config.py:64-65)."""
    if not text or not text.strip():
        return []
    chunks: List[str] = []
    start = 0
    length = len(text)
    while start < length:
        end = start + chunk_size
        chunk = text[start:end]
        if end < length:
            boundary = max(
                chunk.rfind(". "),
                chunk.rfind(".\n"),
                chunk.rfind("? "),
                chunk.rfind("! "),
                chunk.rfind("\n\n"),
            )
            if boundary > chunk_size // 2:
                chunk = chunk[: boundary + 1]
                end = start + boundary + 1
        chunk = chunk.strip()
        if chunk:
            chunks.append(chunk)
        start = end - chunk_overlap
    return chunks


def item_id_to_store_key(item_id: str) -> str:
    """retriever.py:610-637 `_item_id_to_redis_key`."""
    parts = item_id.split("_")
    if len(parts) < 3:
        return f"doc:{item_id}"
    return f"doc:{'_'.join(parts[:2])}:{'_'.join(parts[2:])}"


def fallback_summary(content: str, max_length: int) -> str:
    """summarizer.py:743-771 `_generate_fallback_summary`."""
    if not content or not content.strip():
        return "Content unavailable"
    clean = content.strip()
    if len(clean) <= max_length:
        return clean
    truncated = clean[:max_length]
    boundary = max(truncated.rfind(". "), truncated.rfind("? "), truncated.rfind("! "))
    if boundary > max_length // 2:
        return truncated[: boundary + 1]
    return truncated + "..."


def cache_key(text: str) -> str:
    """embedder.py:736-742 `_get_cache_key`."""
    return hashlib.md5(text.encode("utf-8")).hexdigest()


def chroma_id(doc_id: str, item_id: str) -> str:
    """embedder.py:474."""
    return f"{doc_id}_{item_id}"
