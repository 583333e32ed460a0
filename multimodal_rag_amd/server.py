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


class Pipeline:
    """The stages behind the routes.  The stages this package does not own (parser, summariser, text / multimodal
    generators) are whatever the caller passes to `create_app`, else the stand-ins of ingest.py."""

    CONTEXT_KINDS = ("text_chunks", "table_chunks", "image_chunks")

    def __init__(self, parts: dict):
        self.parts = parts

    def __getattr__(self, name):          # pipeline.embedder, pipeline.retriever, ...
        try:
            return self.__dict__["parts"][name]
        except KeyError:
            raise AttributeError(name) from None

    async def start(self):
        p = self.parts
        p["parser"] = p["parser"] or TextDocumentParser()
        p["llm"] = p["llm"] or ExtractiveAnswerer()
        p["mllm"] = p["mllm"] or p["llm"]
        await p["llm"].initialize()
        p["summarizer"] = p["summarizer"] or PassthroughSummarizer(p["mllm"])
        p["embedder"] = p["embedder"] or EmbeddingManager(batch_size=32, enable_cache=True)
        p["retriever"] = p["retriever"] or MultiVectorRetriever(enable_compression=True, enable_cache=True)
        for name in ("embedder", "retriever"):
            await p[name].initialize()

    async def stop(self):
        for name in ("llm", "embedder", "retriever"):
            try:
                await self.parts[name].cleanup()
            except Exception as e:  # pragma: no cover
                logger.error("Cleanup error: %s", e)

    async def ingest(self, filename: str, content_type: Optional[str], data: bytes) -> dict:
        """parse -> summarise -> embed + store vectors -> store raw items (api.py:262-300); returns the UploadResponse
        fields except the timing ones"""
        doc_id = "doc_" + uuid.uuid4().hex[:12]
        tree = await self.parser.parse_document(data, filename, content_type, doc_id=doc_id)
        items = await self.summarizer.summarize_parsed_document(tree, max_length=300, show_progress=True)
        if not items:
            raise HTTPException(status_code=status.HTTP_400_BAD_REQUEST, detail="No content extracted")
        stored = await self.embedder.embed_and_store(items, doc_id)
        await self.retriever.store_raw_documents(doc_id, items, filename)
        return {"doc_id": doc_id, "filename": filename, "doc_type": tree.get("doc_type", "unknown"),
                "chunks_processed": stored}

    async def answer(self, question: str, top_k: int, multimodal: bool) -> Optional[dict]:
        """vector search -> raw items -> generator (api.py:338-400); None when nothing was retrieved"""
        hits = await self.embedder.query(question, n_results=top_k)
        if not hits["ids"]:
            return None
        raw = await self.retriever.retrieve_raw_documents(hits["ids"])
        passages, tables, images = (raw[k] for k in self.CONTEXT_KINDS)
        body = "\n\n".join(passages) if passages else ""
        if multimodal and (images or tables):
            text = await self.mllm.generate_multimodal(text=body, tables=tables, images=images, max_tokens=1000,
                                                       temperature=0.7)
        else:
            if tables:
                body += "\n\nBảng:\n" + "\n\n".join(tables)
            text = await self.llm.generate_text(f"Context:\n{body}\n\nCâu hỏi: {question}\n\nTrả lời:",
                                                max_tokens=1000, temperature=0.7)
        ranked = [{"rank": at, "doc_id": found, "relevance_score": round(float(1.0 - min(dist, 1.0)), 3),   # api.py:390
                   "type": meta.get("type", "unknown")}
                  for at, (found, dist, meta) in enumerate(zip(hits["ids"], hits["distances"], hits["metadatas"]), 1)]
        return {"answer": text, "sources": ranked}

    async def health(self) -> dict:
        parts = {"llm_adapter": await self.llm.health_check(),
                 "embedder": {"status": "healthy",
                              "documents": (await self.embedder.get_collection_stats()).get("count", 0)},
                 "retriever": await self.retriever.health_check()}
        fine = all(v.get("status") == "healthy" or v.get("healthy") is True for v in parts.values())
        return {"status": "healthy" if fine else "degraded", "components": parts,
                "timestamp": datetime.utcnow().isoformat(), "auth": "disabled"}

    async def report(self) -> dict:
        emb, ret, summ = [await self.parts[n].get_stats() for n in ("embedder", "retriever", "summarizer")]
        listing = await self.retriever.list_all_documents()
        per_kind = {kind: sum(doc.get("chunks", {}).get(kind, 0) for doc in listing) for kind in ("text", "table", "image")}
        return {"documents": {"total": len(listing), "total_chunks": emb.get("count", 0), "by_type": per_kind},
                "embedder": {"cache_hit_rate": emb.get("cache", {}).get("hit_rate", 0)},
                "retriever": {"compression_enabled": ret.get("features", {}).get("compression", False),
                              "compression_savings": ret.get("compression", {}).get("savings_percent", 0)},
                "summarizer": {"total_summaries": summ.get("total_summaries", 0),
                               "cache_hit_rate": summ.get("cache", {}).get("hit_rate", 0)},
                "auth": "disabled"}


def _as_http_500(fn):
    """route wrapper: anything but an HTTPException becomes a 500 with the message as detail (the reference's blanket
    `except Exception` around every route body)"""
    import functools

    @functools.wraps(fn)
    async def wrapped(*a, **kw):
        try:
            return await fn(*a, **kw)
        except HTTPException:
            raise
        except Exception as e:
            logger.error("%s failed: %s", fn.__name__, e, exc_info=True)
            raise HTTPException(status_code=status.HTTP_500_INTERNAL_SERVER_ERROR, detail=str(e))

    return wrapped


def create_app(embedder: Optional[Any] = None, retriever: Optional[Any] = None, parser: Optional[Any] = None,
               summarizer: Optional[Any] = None, llm_adapter: Optional[Any] = None,
               mllm_adapter: Optional[Any] = None) -> FastAPI:
    pipe = Pipeline({"embedder": embedder, "retriever": retriever, "parser": parser, "summarizer": summarizer,
                     "llm": llm_adapter, "mllm": mllm_adapter})

    @asynccontextmanager
    async def lifespan(app: FastAPI):  # api.py:65-128
        await pipe.start()
        yield
        await pipe.stop()

    app = FastAPI(title="Multi-modal RAG System (MI355X hot path)", version="2.0.0", lifespan=lifespan)
    app.state.components = pipe.parts

    @app.get("/health")
    async def health_check():  # api.py:202-241: never raises
        try:
            return await pipe.health()
        except Exception as e:
            return {"status": "unhealthy", "error": str(e)}

    @app.post("/upload", response_model=UploadResponse)
    @_as_http_500
    async def upload_document(request: Request):  # api.py:244-322
        t0 = time.time()
        form = parse_multipart_file(request.headers.get("content-type", ""), await request.body())
        if form is None:  # FastAPI's own answer to a missing File(...) field
            raise HTTPException(status_code=422, detail=[{"loc": ["body", "file"], "msg": "Field required",
                                                          "type": "missing"}])
        filename, content_type, data = form
        if not filename:
            raise HTTPException(status_code=status.HTTP_400_BAD_REQUEST, detail="Filename is required")
        if len(data) > settings.MAX_UPLOAD_SIZE * 1024 * 1024:
            raise HTTPException(status_code=413, detail=f"File too large. Max: {settings.MAX_UPLOAD_SIZE}MB")
        out = await pipe.ingest(filename, content_type, data)
        took = time.time() - t0
        return {**out, "message": f"Processed in {took:.2f}s", "processing_time": took}

    @app.post("/query", response_model=QueryResponse)
    @_as_http_500
    async def query_documents(request: QueryRequest):  # api.py:325-413
        t0 = time.time()
        out = await pipe.answer(request.query, request.top_k, request.use_multimodal)
        if out is None:
            out = {"answer": NO_DOCS_ANSWER, "sources": []}
        return {**out, "processing_time": time.time() - t0}

    @app.get("/documents")
    @_as_http_500
    async def list_documents():  # api.py:416-429
        listing = await pipe.retriever.list_all_documents()
        return {"total": len(listing), "documents": listing}

    @app.delete("/documents/{doc_id}")
    @_as_http_500
    async def delete_document(doc_id: str):  # api.py:432-445
        for store in (pipe.embedder, pipe.retriever):
            await store.delete_document(doc_id)
        return {"message": f"Document {doc_id} deleted"}

    @app.delete("/documents")
    @_as_http_500
    async def delete_all_documents():  # api.py:448-465
        n = len(await pipe.retriever.list_all_documents())
        for store in (pipe.embedder, pipe.retriever):
            await store.delete_all_documents()
        return {"message": f"Deleted {n} documents", "count": n}

    @app.get("/stats")
    @_as_http_500
    async def get_stats():  # api.py:468-508
        return await pipe.report()

    return app


app = create_app()
