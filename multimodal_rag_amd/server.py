"""FastAPI surface of the hot path: POST /upload and POST /query with the reference's request /
response schemas, status codes and messages (app/server/api.py:161-179, :244-413), plus the
maintenance routes that call into the embedder / retriever (:202-241, :416-508).

    uvicorn multimodal_rag_amd.server:app            # needs an MI355X

`create_app()` accepts replacement components so the out-of-scope stages (parser, summariser,
LLM) can be the reference's real ones; defaults are the minimal stand-ins in ingest.py.
"""
from __future__ import annotations

import logging
import time
import uuid
from contextlib import asynccontextmanager
from datetime import datetime
from typing import Any, List, Optional

from fastapi import FastAPI, HTTPException, Request, status
from pydantic import BaseModel, Field

from .config import settings
from .embedder import EmbeddingManager
from .ingest import ExtractiveAnswerer, PassthroughSummarizer, TextDocumentParser
from .retriever import MultiVectorRetriever

logger = logging.getLogger(__name__)

NO_DOCS_ANSWER = "Không tìm thấy tài liệu liên quan. Vui lòng upload tài liệu hoặc thử câu hỏi khác."  # api.py:342


class QueryRequest(BaseModel):  # api.py:161-164
    query: str = Field(..., min_length=1, max_length=2000)
    top_k: int = Field(5, ge=1, le=20)
    use_multimodal: bool = Field(False)


class QueryResponse(BaseModel):  # api.py:167-170
    answer: str
    sources: List[dict]
    processing_time: float


class UploadResponse(BaseModel):  # api.py:173-179
    doc_id: str
    filename: str
    doc_type: str
    chunks_processed: dict
    message: str
    processing_time: float


def parse_multipart_file(content_type: str, body: bytes, field: str = "file"):
    """Minimal multipart/form-data reader for the single `file` part /upload takes (api.py:245
    `file: UploadFile = File(...)`); python-multipart is not available in this image, so the
    stdlib MIME parser does the work.  Returns (filename, part content type, bytes) or None."""
    from email.parser import BytesParser
    from email.policy import HTTP

    if not content_type or "multipart/form-data" not in content_type.lower():
        return None
    msg = BytesParser(policy=HTTP).parsebytes(
        b"Content-Type: " + content_type.encode("latin-1") + b"\r\nMIME-Version: 1.0\r\n\r\n" + body)
    if not msg.is_multipart():
        return None
    for part in msg.iter_parts():
        if part.get_content_disposition() != "form-data":
            continue
        if part.get_param("name", header="content-disposition") != field:
            continue
        return part.get_filename(), part.get_content_type(), part.get_payload(decode=True) or b""
    return None


class _Upload:
    def __init__(self, filename, content_type, data):
        self.filename, self.content_type, self._data = filename, content_type, data

    async def read(self) -> bytes:
        return self._data


def create_app(embedder: Optional[Any] = None, retriever: Optional[Any] = None, parser: Optional[Any] = None,
               summarizer: Optional[Any] = None, llm_adapter: Optional[Any] = None,
               mllm_adapter: Optional[Any] = None) -> FastAPI:
    c = {"embedder": embedder, "retriever": retriever, "parser": parser, "summarizer": summarizer,
         "llm": llm_adapter, "mllm": mllm_adapter}

    @asynccontextmanager
    async def lifespan(app: FastAPI):  # api.py:65-128
        c["parser"] = c["parser"] or TextDocumentParser()
        c["llm"] = c["llm"] or ExtractiveAnswerer()
        c["mllm"] = c["mllm"] or c["llm"]
        await c["llm"].initialize()
        c["summarizer"] = c["summarizer"] or PassthroughSummarizer(c["mllm"])
        c["embedder"] = c["embedder"] or EmbeddingManager(batch_size=32, enable_cache=True)
        await c["embedder"].initialize()
        c["retriever"] = c["retriever"] or MultiVectorRetriever(enable_compression=True, enable_cache=True)
        await c["retriever"].initialize()
        yield
        for name in ("llm", "embedder", "retriever"):
            try:
                await c[name].cleanup()
            except Exception as e:  # pragma: no cover
                logger.error("Cleanup error: %s", e)

    app = FastAPI(title="Multi-modal RAG System (MI355X hot path)", version="2.0.0", lifespan=lifespan)
    app.state.components = c

    @app.get("/health")
    async def health_check():  # api.py:202-241
        try:
            comp = {"llm_adapter": await c["llm"].health_check()}
            stats = await c["embedder"].get_collection_stats()
            comp["embedder"] = {"status": "healthy", "documents": stats.get("count", 0)}
            comp["retriever"] = await c["retriever"].health_check()
            ok = all(x.get("status") == "healthy" or x.get("healthy") is True for x in comp.values())
            return {"status": "healthy" if ok else "degraded", "components": comp,
                    "timestamp": datetime.utcnow().isoformat(), "auth": "disabled"}
        except Exception as e:
            return {"status": "unhealthy", "error": str(e)}

    @app.post("/upload", response_model=UploadResponse)
    async def upload_document(request: Request):  # api.py:244-322
        start_time = time.time()
        parsed_form = parse_multipart_file(request.headers.get("content-type", ""), await request.body())
        if parsed_form is None:  # FastAPI's own answer to a missing File(...) field
            raise HTTPException(status_code=422, detail=[{"loc": ["body", "file"], "msg": "Field required",
                                                          "type": "missing"}])
        file = _Upload(*parsed_form)
        if not file.filename:
            raise HTTPException(status_code=status.HTTP_400_BAD_REQUEST, detail="Filename is required")
        content = await file.read()
        if len(content) / (1024 * 1024) > settings.MAX_UPLOAD_SIZE:
            raise HTTPException(status_code=413,
                                detail=f"File too large. Max: {settings.MAX_UPLOAD_SIZE}MB")
        try:
            doc_id = f"doc_{uuid.uuid4().hex[:12]}"
            parsed = await c["parser"].parse_document(content, file.filename, file.content_type, doc_id=doc_id)
            doc_type = parsed.get("doc_type", "unknown")
            all_summaries = await c["summarizer"].summarize_parsed_document(parsed, max_length=300, show_progress=True)
            if not all_summaries:
                raise HTTPException(status_code=status.HTTP_400_BAD_REQUEST, detail="No content extracted")
            counts = await c["embedder"].embed_and_store(all_summaries, doc_id)
            await c["retriever"].store_raw_documents(doc_id, all_summaries, file.filename)
            total_time = time.time() - start_time
            return {"doc_id": doc_id, "filename": file.filename, "doc_type": doc_type, "chunks_processed": counts,
                    "message": f"Processed in {total_time:.2f}s", "processing_time": total_time}
        except HTTPException:
            raise
        except Exception as e:
            logger.error("Upload failed: %s", e, exc_info=True)
            raise HTTPException(status_code=status.HTTP_500_INTERNAL_SERVER_ERROR, detail=str(e))

    @app.post("/query", response_model=QueryResponse)
    async def query_documents(request: QueryRequest):  # api.py:325-413
        start_time = time.time()
        try:
            search_results = await c["embedder"].query(request.query, n_results=request.top_k)
            if not search_results["ids"]:
                return {"answer": NO_DOCS_ANSWER, "sources": [], "processing_time": time.time() - start_time}
            raw_docs = await c["retriever"].retrieve_raw_documents(search_results["ids"])
            text_context = "\n\n".join(raw_docs["text_chunks"]) if raw_docs["text_chunks"] else ""
            table_context = raw_docs["table_chunks"]
            image_context = raw_docs["image_chunks"]
            if request.use_multimodal and (image_context or table_context):
                answer = await c["mllm"].generate_multimodal(text=text_context, tables=table_context,
                                                             images=image_context, max_tokens=1000, temperature=0.7)
            else:
                full_context = text_context
                if table_context:
                    full_context += "\n\nBảng:\n" + "\n\n".join(table_context)
                prompt = f"Context:\n{full_context}\n\nCâu hỏi: {request.query}\n\nTrả lời:"
                answer = await c["llm"].generate_text(prompt, max_tokens=1000, temperature=0.7)
            sources = []
            for i, (doc_id, distance, metadata) in enumerate(zip(search_results["ids"], search_results["distances"],
                                                                 search_results["metadatas"])):
                relevance_score = float(1.0 - min(distance, 1.0))  # api.py:390
                sources.append({"rank": i + 1, "doc_id": doc_id, "relevance_score": round(relevance_score, 3),
                                "type": metadata.get("type", "unknown")})
            return {"answer": answer, "sources": sources, "processing_time": time.time() - start_time}
        except Exception as e:
            logger.error("Query failed: %s", e, exc_info=True)
            raise HTTPException(status_code=status.HTTP_500_INTERNAL_SERVER_ERROR, detail=str(e))

    @app.get("/documents")
    async def list_documents():  # api.py:416-429
        try:
            documents = await c["retriever"].list_all_documents()
            return {"total": len(documents), "documents": documents}
        except Exception as e:
            raise HTTPException(status_code=status.HTTP_500_INTERNAL_SERVER_ERROR, detail=str(e))

    @app.delete("/documents/{doc_id}")
    async def delete_document(doc_id: str):  # api.py:432-445
        try:
            await c["embedder"].delete_document(doc_id)
            await c["retriever"].delete_document(doc_id)
            return {"message": f"Document {doc_id} deleted"}
        except Exception as e:
            raise HTTPException(status_code=status.HTTP_500_INTERNAL_SERVER_ERROR, detail=str(e))

    @app.delete("/documents")
    async def delete_all_documents():  # api.py:448-465
        try:
            count = len(await c["retriever"].list_all_documents())
            await c["embedder"].delete_all_documents()
            await c["retriever"].delete_all_documents()
            return {"message": f"Deleted {count} documents", "count": count}
        except Exception as e:
            raise HTTPException(status_code=status.HTTP_500_INTERNAL_SERVER_ERROR, detail=str(e))

    @app.get("/stats")
    async def get_stats():  # api.py:468-508
        try:
            e = await c["embedder"].get_stats()
            r = await c["retriever"].get_stats()
            s = await c["summarizer"].get_stats()
            documents = await c["retriever"].list_all_documents()
            by = {t: sum(d.get("chunks", {}).get(t, 0) for d in documents) for t in ("text", "table", "image")}
            return {
                "documents": {"total": len(documents), "total_chunks": e.get("count", 0),
                              "by_type": {"text": by["text"], "table": by["table"], "image": by["image"]}},
                "embedder": {"cache_hit_rate": e.get("cache", {}).get("hit_rate", 0)},
                "retriever": {"compression_enabled": r.get("features", {}).get("compression", False),
                              "compression_savings": r.get("compression", {}).get("savings_percent", 0)},
                "summarizer": {"total_summaries": s.get("total_summaries", 0),
                               "cache_hit_rate": s.get("cache", {}).get("hit_rate", 0)},
                "auth": "disabled",
            }
        except Exception as e:
            raise HTTPException(status_code=status.HTTP_500_INTERNAL_SERVER_ERROR, detail=str(e))

    return app


app = create_app()
