#!/usr/bin/env python3
"""Regenerate the WAL-70 golden fixture from the reference's committed Chroma state.

Run in the BUILD container only (the GPU box has no /root/reference):

    python tests/golden/make_wal_fixture.py

Source: /root/reference/chroma_db/chroma.sqlite3, table `embeddings_queue` (Chroma's
write-ahead log): 70 ADD records (operation=0) each holding a 384 x float32 unit
vector produced by the reference author's all-MiniLM-L6-v2 run, plus 70 DELETE
records (operation=3).  SURVEY.md F8 / Appendix B.

What is written (data only; the embedded document texts are NOT copied):
  wal70_vectors.f32   70*384 little-endian float32, row i = i-th ADD in seq order
  wal70_ids.json      ids, per-row {doc_id,item_id,type}, and the full op log
                      [(seq_id, op, id)] for WAL-replay tests
  wal70_top5.json     all-pairs exact cosine top-5 (self included) computed in
                      float64: ids + cosines, ties -> lower row first
"""
import json
import os
import sqlite3

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DB = "file:/root/reference/chroma_db/chroma.sqlite3?mode=ro&immutable=1"


def main():
    con = sqlite3.connect(DB, uri=True)
    rows = con.execute(
        "select seq_id, operation, id, vector, encoding, metadata "
        "from embeddings_queue order by seq_id"
    ).fetchall()
    log = [(int(s), int(op), i) for s, op, i, _, _, _ in rows]
    adds = [r for r in rows if r[1] == 0]
    assert len(adds) == 70 and all(r[4] == "FLOAT32" for r in adds)
    V = np.stack([np.frombuffer(r[3], dtype="<f4") for r in adds]).astype(np.float32)
    assert V.shape == (70, 384)
    norms = np.linalg.norm(V.astype(np.float64), axis=1)
    assert np.all(np.abs(norms - 1.0) < 1e-6)
    metas = []
    for r in adds:
        m = json.loads(r[5])
        metas.append({k: m[k] for k in ("doc_id", "item_id", "type")})
    ids = [r[2] for r in adds]

    V.astype("<f4").tofile(os.path.join(HERE, "wal70_vectors.f32"))
    with open(os.path.join(HERE, "wal70_ids.json"), "w") as f:
        json.dump({"ids": ids, "metadatas": metas, "dim": 384, "log": log}, f, indent=0)

    S = V.astype(np.float64) @ V.astype(np.float64).T
    top_rows, top_cos = [], []
    for i in range(70):
        order = sorted(range(70), key=lambda j: (-S[i, j], j))[:5]
        top_rows.append(order)
        top_cos.append([float(S[i, j]) for j in order])
    with open(os.path.join(HERE, "wal70_top5.json"), "w") as f:
        json.dump({"rows": top_rows, "cos": top_cos}, f)
    print("wrote wal70 fixture:", V.shape, "min 5th-vs-6th gap:",
          min(sorted(S[i], reverse=True)[4] - sorted(S[i], reverse=True)[5] for i in range(70)))


if __name__ == "__main__":
    main()
