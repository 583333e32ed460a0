"""CPU: the FastAPI /upload + /query surface (api.py:161-179, :244-413) over the fake engine."""
import os

import pytest
from starlette.testclient import TestClient

from multimodal_rag_amd.embedder import EmbeddingManager
from multimodal_rag_amd.server import NO_DOCS_ANSWER, create_app, parse_multipart_file
from tests.fakes import FakeEngine


@pytest.fixture()
def client():
    app = create_app(embedder=EmbeddingManager(engine=FakeEngine()))
    with TestClient(app) as c:
        yield c


def test_query_on_empty_index_returns_canned_answer(client):
    r = client.post("/query", json={"query": "anything"})
    assert r.status_code == 200
    body = r.json()
    assert body["answer"] == NO_DOCS_ANSWER and body["sources"] == [] and body["processing_time"] >= 0


def test_upload_then_query_roundtrip(client, golden_dir):
    data = open(os.path.join(golden_dir, "sample_document.txt"), "rb").read()
    r = client.post("/upload", files={"file": ("sample_document.txt", data, "text/plain")})
    assert r.status_code == 200, r.text
    up = r.json()
    assert set(up) == {"doc_id", "filename", "doc_type", "chunks_processed", "message", "processing_time"}
    assert up["doc_id"].startswith("doc_") and len(up["doc_id"]) == 16
    assert up["chunks_processed"] == {"text": 1, "table": 0, "image": 0} and up["doc_type"] == "text"
    long_doc = ("Sentence about retrieval. " * 200).encode()
    up2 = client.post("/upload", files={"file": ("long.md", long_doc, "text/markdown")}).json()
    assert up2["chunks_processed"]["text"] > 3

    q = client.post("/query", json={"query": "Machine learning là gì?", "top_k": 3}).json()
    assert len(q["sources"]) == 3
    for i, s in enumerate(q["sources"]):
        assert set(s) == {"rank", "doc_id", "relevance_score", "type"} and s["rank"] == i + 1
        assert 0.0 <= s["relevance_score"] <= 1.0 and s["type"] == "text"
    assert isinstance(q["answer"], str) and q["answer"]

    docs = client.get("/documents").json()
    assert docs["total"] == 2
    st = client.get("/stats").json()
    assert st["documents"]["total_chunks"] == 1 + up2["chunks_processed"]["text"]
    assert client.get("/health").json()["status"] == "healthy"
    assert client.delete(f"/documents/{up['doc_id']}").status_code == 200
    assert client.get("/documents").json()["total"] == 1
    assert client.delete("/documents").json()["count"] == 1
    assert client.post("/query", json={"query": "x"}).json()["sources"] == []


def test_request_validation_and_errors(client):
    assert client.post("/query", json={"query": ""}).status_code == 422          # min_length=1
    assert client.post("/query", json={"query": "x", "top_k": 21}).status_code == 422
    assert client.post("/query", json={"query": "x", "top_k": 0}).status_code == 422
    assert client.post("/query", json={"query": "x" * 2001}).status_code == 422
    assert client.post("/query", json={"query": "   "}).status_code == 500       # ValueError -> 500 detail (api.py:408-413)
    assert client.post("/upload").status_code == 422
    r = client.post("/upload", files={"file": ("empty.txt", b"   ", "text/plain")})
    assert r.status_code == 400 and r.json()["detail"] == "No content extracted"
    r = client.post("/upload", files={"file": ("scan.pdf", b"%PDF-1.4", "application/pdf")})
    assert r.status_code == 500 and "Unsupported" in r.json()["detail"]


def test_upload_size_limit(client, monkeypatch):
    from multimodal_rag_amd import server

    monkeypatch.setattr(server.settings, "MAX_UPLOAD_SIZE", 0)
    r = client.post("/upload", files={"file": ("a.txt", b"hello world", "text/plain")})
    assert r.status_code == 413 and r.json()["detail"] == "File too large. Max: 0MB"


def test_multipart_parser():
    body = (b"--XX\r\nContent-Disposition: form-data; name=\"other\"\r\n\r\nv\r\n"
            b"--XX\r\nContent-Disposition: form-data; name=\"file\"; filename=\"a b.txt\"\r\n"
            b"Content-Type: text/plain\r\n\r\nline1\r\nline2\r\n--XX--\r\n")
    name, ctype, data = parse_multipart_file("multipart/form-data; boundary=XX", body)
    assert (name, ctype, data) == ("a b.txt", "text/plain", b"line1\r\nline2")
    assert parse_multipart_file("application/json", b"{}") is None
