"""Launcher of the multi-GPU service: one process per GPU, rank 0 serves HTTP.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        -m multimodal_rag_amd.serve_sharded --host 0.0.0.0 --port 8000

Rank 0 builds the encoder engine and the FastAPI app of server.py; its collection is a
`serving.ShardedCollection` over this rank's `VectorIndex`.  Ranks 1..N-1 hold a `VectorIndex`
shard each and execute rank 0's commands (`worker_loop`).  The reference is single-process
(SURVEY.md section 2.1); its /upload and /query contract is unchanged.
"""
from __future__ import annotations

import argparse
import os
from typing import Any, Dict, Optional

import torch
import torch.distributed as dist

from .config import settings
from .serving import ShardedCollection


class ShardedEngine:
    """An encoder engine whose collections are sharded over the process group (rank 0 side)."""

    def __init__(self, engine, device: torch.device, group: Optional[dist.ProcessGroup] = None,
                 control_group: Optional[dist.ProcessGroup] = None):
        self._engine = engine
        self._device = device
        self._group = group
        self._control = control_group
        self._col: Optional[ShardedCollection] = None

    def __getattr__(self, name):            # encode / encode_images / dim / max_seq_length / release ...
        return getattr(self._engine, name)

    def new_collection(self, name: str, metadata: Optional[Dict[str, Any]] = None):
        if self._col is None:
            self._col = ShardedCollection(self._engine.new_collection(name, metadata), self._group, self._device,
                                          encode_fn=self._engine.encode, control_group=self._control)
            self._col.start_keepalive(600.0)     # the control group's timeout is finite (main())
        else:                                # delete_all_documents re-creates the collection (embedder.py:670-678)
            self._col.reset()
        return self._col


def index_dtype():
    return {"float16": torch.float16, "float32": torch.float32, "bfloat16": torch.bfloat16}[settings.MMRAG_INDEX_DTYPE]


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=8000)
    args = ap.parse_args(argv)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    if not torch.cuda.is_available():
        raise SystemExit("serve_sharded needs MI355X devices (one per rank); there is no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    # the control plane (command headers, ingest batches, winners' payload) runs on gloo: workers block in it between
    # requests, and the RCCL group's watchdog (10 minutes by default) would abort them.  Its timeout is finite (a lost
    # collective step must surface as an error, not hang rank 0 under the service lock for ever); rank 0 pings the
    # workers every 10 idle minutes (ShardedCollection.start_keepalive).  RCCL carries the per-query all-gather only.
    import datetime

    control = dist.new_group(backend="gloo", timeout=datetime.timedelta(hours=2))
    try:
        if rank == 0:
            import uvicorn

            from .embedder import CLIP_MODEL_NAMES, ClipEngine, EmbeddingManager, HipEngine, _is_clip_dir
            from .server import create_app

            name = settings.SENTENCE_TRANSFORMER_MODEL
            factory = ClipEngine if (name in CLIP_MODEL_NAMES or _is_clip_dir(settings.MMRAG_MODEL_DIR)) else HipEngine
            engine = factory(name, f"cuda:{local}")
            box = [engine.dim]
            dist.broadcast_object_list(box, src=0)
            sharded = ShardedEngine(engine, dev, control_group=control)
            manager = EmbeddingManager(batch_size=32, enable_cache=True, engine=sharded)
            try:
                uvicorn.run(create_app(embedder=manager), host=args.host, port=args.port, log_level=settings.LOG_LEVEL.lower())
            finally:
                if sharded._col is not None:
                    sharded._col.stop()
        else:
            from .embedder import CLIP_MODEL_NAMES, ClipEngine, HipEngine, _is_clip_dir

            box = [None]
            dist.broadcast_object_list(box, src=0)
            name = settings.SENTENCE_TRANSFORMER_MODEL
            factory = ClipEngine if (name in CLIP_MODEL_NAMES or _is_clip_dir(settings.MMRAG_MODEL_DIR)) else HipEngine
            engine = factory(name, f"cuda:{local}")      # every rank holds the encoder: ingest is data-parallel
            assert engine.dim == int(box[0])
            shard = engine.new_collection(settings.CHROMA_COLLECTION_NAME)
            ShardedCollection(shard, device=dev, encode_fn=engine.encode, control_group=control).worker_loop()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
