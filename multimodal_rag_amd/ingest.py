"""The stages on either side of the hot path, reduced to what BASELINE config 1 needs
(data/sample_document.txt -> chunk -> embed -> top-5) so the FastAPI surface can run end to end
without the reference's out-of-scope subsystems (document parsers, LLM summariser, LLM answer
generation: SURVEY.md section 2 rows 8-10).  Each class keeps the reference's call signature so
the real component can be plugged back in.
"""
from __future__ import annotations

import uuid
from typing import Any, Dict, List, Optional

from .config import settings


def basic_chunk_text(text: str, chunk_size: Optional[int] = None, chunk_overlap: Optional[int] = None) -> List[str]:
    """parser.py:1702-1736 `_basic_chunk_text`: window of chunk_size characters, backed off to the
    last sentence boundary if it lies in the second half; next window starts chunk_overlap before
    the previous end."""
    chunk_size = settings.CHUNK_SIZE if chunk_size is None else chunk_size
    chunk_overlap = settings.CHUNK_OVERLAP if chunk_overlap is None else chunk_overlap
    if not text or not text.strip():
        return []
    chunks: List[str] = []
    start = 0
    length = len(text)
    while start < length:
        end = start + chunk_size
        chunk = text[start:end]
        if end < length:
            boundary = max(chunk.rfind(". "), chunk.rfind(".\n"), chunk.rfind("? "), chunk.rfind("! "),
                           chunk.rfind("\n\n"))
            if boundary > chunk_size // 2:
                chunk = chunk[: boundary + 1]
                end = start + boundary + 1
        chunk = chunk.strip()
        if chunk:
            chunks.append(chunk)
        start = end - chunk_overlap
    return chunks


class TextDocumentParser:
    """Plain-text / markdown slice of `DocumentParser.parse_document` (parser.py:188, _parse_text
    :1444-1494, _chunk_text_simple :1672-1700, _enrich_chunks_metadata :1740-1760)."""

    def __init__(self, **_ignored):
        self.chunk_size = settings.CHUNK_SIZE
        self.chunk_overlap = settings.CHUNK_OVERLAP

    async def parse_document(self, content: bytes, filename: str, content_type: Optional[str] = None,
                             doc_id: Optional[str] = None) -> Dict[str, Any]:
        name = (filename or "").lower()
        if not name.endswith((".txt", ".md", ".markdown", ".text")) and not (content_type or "").startswith("text/"):
            raise ValueError(f"Unsupported file type for this build: {filename} (text/markdown only)")
        try:
            text = content.decode("utf-8")
        except UnicodeDecodeError:
            text = content.decode("latin-1", errors="ignore")
        chunks = []
        for chunk_id, chunk_text in enumerate(basic_chunk_text(text, self.chunk_size, self.chunk_overlap)):
            unique_id = str(uuid.uuid4())[:8]
            chunks.append({
                "chunk_id": f"{doc_id}_chunk_{chunk_id}_{unique_id}",
                "content": chunk_text.strip(),
                "metadata": {"char_count": len(chunk_text), "filename": filename, "doc_type": "text",
                             "doc_id": doc_id},
            })
        return {"doc_type": "text", "text_chunks": chunks, "tables": [], "images": [], "document_structure": {}}


def fallback_summary(content: str, max_length: int) -> str:
    """summarizer.py:743-771 `_generate_fallback_summary` (the LLM-free path)."""
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


class PassthroughSummarizer:
    """Emits the summariser's output contract (summarizer.py:629-655, :668-706) with the
    reference's own LLM-free fallback as the summary text."""

    def __init__(self, *_args, **_kwargs):
        self.stats = {"total_summaries": 0}

    async def summarize_parsed_document(self, parsed_result: Dict[str, Any], max_length: int = 300,
                                        show_progress: bool = True) -> List[Dict[str, Any]]:
        out: List[Dict[str, Any]] = []
        for idx, chunk in enumerate(parsed_result.get("text_chunks", [])):
            out.append({"id": f"text_{idx}", "summary": fallback_summary(chunk["content"], max_length),
                        "raw": chunk["content"], "type": "text", "metadata": chunk.get("metadata", {})})
        for table in parsed_result.get("tables", []):
            out.append({"id": table.get("id", "table_0"), "summary": fallback_summary(table.get("content", ""), max_length),
                        "raw": table.get("content", ""), "type": "table"})
        for image in parsed_result.get("images", []):
            out.append({"id": image.get("id", "image_0"),
                        "summary": image.get("description") or f"Image: {image.get('id', 'image_0')}",
                        "raw": image.get("base64", ""), "path": image.get("path", ""), "type": "image"})
        self.stats["total_summaries"] += len(out)
        return out

    async def get_stats(self) -> Dict[str, Any]:
        return {"total_summaries": self.stats["total_summaries"], "cache": {"hit_rate": 0}}


class ExtractiveAnswerer:
    """Stand-in for LLMAdapter/MLLMAdapter (llm_adapter.py:96, mllm_adapter.py:127): no network,
    returns the retrieved context so the /query response schema is exercised."""

    async def initialize(self):
        return None

    async def cleanup(self):
        return None

    async def health_check(self) -> Dict[str, Any]:
        return {"status": "healthy", "model": "extractive-stub"}

    async def generate_text(self, prompt: str, max_tokens: int = 1000, temperature: float = 0.7) -> str:
        body = prompt.split("Câu hỏi:")[0].replace("Context:", "", 1).strip()
        return body[: max_tokens * 4]

    async def generate_multimodal(self, text: str = "", tables=None, images=None, max_tokens: int = 1000,
                                  temperature: float = 0.7) -> str:
        parts = [text] + list(tables or [])
        return "\n\n".join(p for p in parts if p)[: max_tokens * 4]
