"""Host side of the text encoder: what `SentenceTransformer(name, device)` is to the reference
(app/utils/embedder.py:245-248) and `.encode(...)` at :397-403.

The forward pass itself is mmrag_encoder_forward() in libmmrag.so (hand-written HIP kernels);
this module only owns the weight table in HBM, packs token ids, and hands pointers over.
There is no eager / CPU forward here.

Weights: the reference loads a checkpoint by hub NAME, which needs a network.  Here a model is
either loaded from a user-supplied LOCAL directory in Hugging Face layout (config.json +
model.safetensors [+ vocab.txt]) or random-initialised with the named architecture (what the
benchmarks use; no checkpoint can be fetched in this environment).
"""
from __future__ import annotations

import ctypes
import json
import logging
import os
import threading
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _native
from .tracing import stage

logger = logging.getLogger(__name__)


@dataclass(frozen=True)
class EncoderConfig:
    name: str
    n_layers: int
    hidden: int
    n_heads: int
    intermediate: int
    vocab: int = 30522
    max_pos: int = 512
    max_seq_length: int = 256       # sentence-transformers truncation length
    pool: str = "mean"              # "mean" | "cls"
    ln_eps: float = 1e-12
    arch: str = "bert"

    @property
    def dim(self) -> int:
        return self.hidden


# config.py:102-105 SENTENCE_TRANSFORMER_MODEL default, plus the BASELINE.json config-3 model
PRESETS: Dict[str, EncoderConfig] = {
    "all-MiniLM-L6-v2": EncoderConfig("all-MiniLM-L6-v2", 6, 384, 12, 1536, max_seq_length=256, pool="mean"),
    "sentence-transformers/all-MiniLM-L6-v2": EncoderConfig("all-MiniLM-L6-v2", 6, 384, 12, 1536,
                                                            max_seq_length=256, pool="mean"),
    "BAAI/bge-base-en-v1.5": EncoderConfig("bge-base-en-v1.5", 12, 768, 12, 3072, max_seq_length=512, pool="cls"),
    "bge-base-en-v1.5": EncoderConfig("bge-base-en-v1.5", 12, 768, 12, 3072, max_seq_length=512, pool="cls"),
}



def random_bert_weights(cfg: EncoderConfig, seed: int = 0, device="cuda:0", std: float = 0.02) -> Dict[str, torch.Tensor]:
    """Seeded random float32 weights (HF BertModel names) generated on `device`: what the
    benchmarks and the checkpoint-less service mode use (no weights can be fetched here)."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    H, I = cfg.hidden, cfg.intermediate

    def mat(o, i):
        return torch.randn((o, i), generator=g, device=dev) * std

    def vec(n, base=0.0, scale=0.02):
        return base + torch.randn((n,), generator=g, device=dev) * scale

    w = {"embeddings.word_embeddings.weight": mat(cfg.vocab, H),
         "embeddings.position_embeddings.weight": mat(cfg.max_pos, H),
         "embeddings.token_type_embeddings.weight": mat(2, H),
         "embeddings.LayerNorm.weight": vec(H, 1.0), "embeddings.LayerNorm.bias": vec(H)}
    for l in range(cfg.n_layers):
        p = f"encoder.layer.{l}."
        for n in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense"):
            w[p + n + ".weight"] = mat(H, H)
            w[p + n + ".bias"] = vec(H)
        w[p + "attention.output.LayerNorm.weight"] = vec(H, 1.0)
        w[p + "attention.output.LayerNorm.bias"] = vec(H)
        w[p + "intermediate.dense.weight"] = mat(I, H)
        w[p + "intermediate.dense.bias"] = vec(I)
        w[p + "output.dense.weight"] = mat(H, I)
        w[p + "output.dense.bias"] = vec(H)
        w[p + "output.LayerNorm.weight"] = vec(H, 1.0)
        w[p + "output.LayerNorm.bias"] = vec(H)
    return w


class DeviceEncoder:
    """BERT-family sentence encoder resident on one GPU."""

    def __init__(self, cfg: EncoderConfig, weights: Dict[str, "np.ndarray | torch.Tensor"], device="cuda:0",
                 precision: str = "fp16"):
        """`precision`: "fp16" (fp16 weights / activations, float32 accumulation and statistics: the throughput
        path) or "fp32" (the reference's own arithmetic -- SentenceTransformer.encode runs in float32,
        embedder.py:397-403: float32 weights and activations on the exact float32 matrix instruction; ~1/10 of the
        fp16 path's throughput, embeddings within 2e-5 of the float32 model)."""
        _native.lib()
        if precision not in ("fp16", "fp32"):
            raise ValueError(f"precision must be 'fp16' or 'fp32', not {precision!r}")
        self.cfg = cfg
        self.precision = precision
        self._f32 = precision == "fp32"
        self.device = torch.device(device)
        self._tensors: List[torch.Tensor] = []   # keeps the HBM weight table alive
        ptrs: List[Optional[int]] = []

        def up(x, dtype):
            t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
            if self._f32:
                dtype = torch.float32
            t = t.to(device=self.device, dtype=dtype).contiguous()
            self._tensors.append(t)
            ptrs.append(t.data_ptr())

        g = lambda k: weights[k]  # noqa: E731
        up(g("embeddings.word_embeddings.weight"), torch.float16)
        up(g("embeddings.position_embeddings.weight"), torch.float16)
        tt = g("embeddings.token_type_embeddings.weight")
        up(tt[0], torch.float16)
        up(g("embeddings.LayerNorm.weight"), torch.float32)
        up(g("embeddings.LayerNorm.bias"), torch.float32)
        cat = (lambda xs: torch.cat([x if isinstance(x, torch.Tensor) else torch.from_numpy(np.asarray(x)) for x in xs], 0))
        for l in range(cfg.n_layers):
            p = f"encoder.layer.{l}."
            up(cat([g(p + f"attention.self.{n}.weight") for n in ("query", "key", "value")]), torch.float16)
            up(cat([g(p + f"attention.self.{n}.bias") for n in ("query", "key", "value")]), torch.float32)
            up(g(p + "attention.output.dense.weight"), torch.float16)
            up(g(p + "attention.output.dense.bias"), torch.float32)
            up(g(p + "attention.output.LayerNorm.weight"), torch.float32)
            up(g(p + "attention.output.LayerNorm.bias"), torch.float32)
            up(g(p + "intermediate.dense.weight"), torch.float16)
            up(g(p + "intermediate.dense.bias"), torch.float32)
            up(g(p + "output.dense.weight"), torch.float16)
            up(g(p + "output.dense.bias"), torch.float32)
            up(g(p + "output.LayerNorm.weight"), torch.float32)
            up(g(p + "output.LayerNorm.bias"), torch.float32)
        self._ptrs = (ctypes.c_void_p * len(ptrs))(*ptrs)
        self.desc = _native.EncoderDesc(
            arch=_native.ARCH_BERT, n_layers=cfg.n_layers, hidden=cfg.hidden, n_heads=cfg.n_heads,
            intermediate=cfg.intermediate, vocab=cfg.vocab, max_pos=cfg.max_pos,
            pool=_native.POOL_MEAN if cfg.pool == "mean" else _native.POOL_FIRST, act=_native.ACT_GELU, causal=0,
            normalize=1, out_dim=cfg.hidden, ln_eps=cfg.ln_eps)
        self._workspace: Optional[torch.Tensor] = None
        # One workspace, one stream: the launches of two forwards must not interleave (the service encodes from
        # asyncio.to_thread workers, embedder.py:368, and ctypes drops the GIL).  Held while the kernels are
        # ENQUEUED; stream order then keeps the forwards apart on the device.
        self._launch_lock = threading.Lock()
        self._graphs = {}                 # token count -> captured single-sequence forward (encode_one)
        self._use_graphs = os.environ.get("MMRAG_ENCODER_GRAPHS", "1") not in ("0", "false", "no") and not self._f32

    # ---------------------------------------------------------------- constructors ---------
    @classmethod
    def random_init(cls, cfg: EncoderConfig, seed: int = 0, device="cuda:0", std: float = 0.02,
                    precision: str = "fp16") -> "DeviceEncoder":
        """Seeded random weights of the named architecture, generated on the device."""
        return cls(cfg, random_bert_weights(cfg, seed, device, std), device, precision)

    @classmethod
    def from_local_dir(cls, path: str, device="cuda:0", max_seq_length: Optional[int] = None,
                       precision: str = "fp16") -> "DeviceEncoder":
        """Load a BERT-family checkpoint from a local Hugging Face style directory
        (config.json + model.safetensors).  Nothing is downloaded."""
        from safetensors.numpy import load_file

        with open(os.path.join(path, "config.json")) as f:
            c = json.load(f)
        pool = "mean"
        pool_cfg = os.path.join(path, "1_Pooling", "config.json")
        if os.path.exists(pool_cfg):
            with open(pool_cfg) as f:
                if json.load(f).get("pooling_mode_cls_token"):
                    pool = "cls"
        st_cfg = os.path.join(path, "sentence_bert_config.json")
        if max_seq_length is None and os.path.exists(st_cfg):
            with open(st_cfg) as f:
                max_seq_length = json.load(f).get("max_seq_length")
        cfg = EncoderConfig(os.path.basename(os.path.normpath(path)), c["num_hidden_layers"], c["hidden_size"],
                            c["num_attention_heads"], c["intermediate_size"], c["vocab_size"],
                            c["max_position_embeddings"], max_seq_length or c["max_position_embeddings"], pool,
                            c.get("layer_norm_eps", 1e-12))
        raw = load_file(os.path.join(path, "model.safetensors"))
        w = {(k[5:] if k.startswith("bert.") else k): v for k, v in raw.items()}
        return cls(cfg, w, device, precision)

    # ---------------------------------------------------------------- forward ---------------
    def pack(self, sequences: Sequence[Sequence[int]]):
        """Truncate to max_seq_length and pack: (ids [T], pos_ids [T], cu_seqlens [B+1], max_len)."""
        L = min(self.cfg.max_seq_length, self.cfg.max_pos)
        seqs = [np.asarray(s[:L], dtype=np.int32) for s in sequences]
        if any(len(s) == 0 for s in seqs):
            raise ValueError("empty token sequence")
        lens = np.array([len(s) for s in seqs], dtype=np.int32)
        cu = np.zeros(len(seqs) + 1, dtype=np.int32)
        np.cumsum(lens, out=cu[1:])
        ids = np.concatenate(seqs)
        pos = np.concatenate([np.arange(n, dtype=np.int32) for n in lens])
        return ids, pos, cu, int(lens.max())

    def encode_ids(self, sequences: Sequence[Sequence[int]]) -> torch.Tensor:
        """Token-id sequences -> L2-normalised embeddings [B, dim] float32 on the device."""
        if len(sequences) == 1 and 0 < len(sequences[0]) <= self.GRAPH_MAX_TOKENS and self._use_graphs:
            L = min(self.cfg.max_seq_length, self.cfg.max_pos)
            return self.encode_one(list(sequences[0])[:L])
        ids, pos, cu, max_len = self.pack(sequences)
        return self.forward_packed(*self._upload(ids, pos, cu), max_len)

    def _upload(self, ids: np.ndarray, pos: np.ndarray, cu: np.ndarray):
        """token ids, positions and sequence starts -> the device in ONE copy from pinned memory.  A copy from
        pageable memory is not asynchronous: the call returns only when everything enqueued on the stream before it
        has run, so a second caller's upload waited for the first caller's whole forward + search (concurrent
        batch_query callers ran no faster than one: tools/served_callers.py).  torch's caching host allocator hands
        the pinned block back only after the copy has run."""
        n, b = ids.size, cu.size
        n4 = (n + 3) & ~3                                   # 16-byte aligned segments
        host = torch.empty(2 * n4 + b, dtype=torch.int32).pin_memory() if self.device.type == "cuda" else \
            torch.empty(2 * n4 + b, dtype=torch.int32)
        h = host.numpy()
        h[:n] = ids
        h[n4:n4 + n] = pos
        h[2 * n4:] = cu
        dev = host.to(self.device, non_blocking=True)
        return dev[:n], dev[n4:n4 + n], dev[2 * n4:]

    def encode_id_rows(self, ids2d: np.ndarray, lens: np.ndarray) -> torch.Tensor:
        """Array form of `encode_ids` (what the native tokenizer returns): row i of `ids2d` [B, W] int32 holds
        lens[i] token ids.  Packs with numpy only -- no per-sequence Python objects."""
        L = min(self.cfg.max_seq_length, self.cfg.max_pos)
        lens = np.minimum(np.asarray(lens, dtype=np.int32), L)
        if lens.size == 0 or lens.min() <= 0:
            raise ValueError("empty token sequence")
        if lens.size == 1 and lens[0] <= self.GRAPH_MAX_TOKENS and self._use_graphs:
            return self.encode_one(ids2d[0, : int(lens[0])])
        W = ids2d.shape[1]
        keep = np.arange(W, dtype=np.int32)[None, :] < lens[:, None]
        ids = np.ascontiguousarray(ids2d[keep], dtype=np.int32)
        pos = np.broadcast_to(np.arange(W, dtype=np.int32)[None, :], ids2d.shape)[keep]
        cu = np.zeros(len(lens) + 1, dtype=np.int32)
        np.cumsum(lens, out=cu[1:])
        with stage("encode.upload"):
            dev = self._upload(ids, pos, cu)
        return self.forward_packed(*dev, int(lens.max()))

    # ---- one sequence of at most 64 tokens (the online /query shape): the forward is a chain of ~33 kernels of 3-5 us and
    # the HOST's launch calls (4 us each) are what paces it.  The chain is captured once per token count into a HIP graph;
    # a query then costs one small host-to-device copy, one graph launch and one copy of the result row.
    GRAPH_MAX_TOKENS = 64

    def _single_sequence_graph(self, T: int):
        g = self._graphs.get(T)
        if g is not None:
            return g
        d = self.device
        ids = torch.full((T,), 5, dtype=torch.int32, device=d)
        pos = torch.arange(T, dtype=torch.int32, device=d)
        cu = torch.tensor([0, T], dtype=torch.int32, device=d)
        need = _native.encoder_workspace_bytes(self.desc, T, 1)
        ws = torch.empty(need, dtype=torch.uint8, device=d)      # the graph's own workspace: replays never share one
        side = torch.cuda.Stream(d)
        side.wait_stream(torch.cuda.current_stream(d))
        with torch.cuda.stream(side):                              # warm-up outside the capture (lazy one-time setup)
            _native.encoder_forward(self.desc, self._ptrs, ids, pos, cu, T, workspace=ws)
        torch.cuda.current_stream(d).wait_stream(side)
        torch.cuda.synchronize(d)
        graph = torch.cuda.CUDAGraph()
        # thread_local: other threads of the service keep allocating and launching while this one captures
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            out = _native.encoder_forward(self.desc, self._ptrs, ids, pos, cu, T, workspace=ws)
        g = self._graphs[T] = (graph, ids, out, ws, pos, cu)
        return g

    def encode_one(self, token_ids) -> torch.Tensor:
        """[1, dim] float32 on the device for ONE sequence of <= GRAPH_MAX_TOKENS token ids, through the captured graph."""
        ids = np.ascontiguousarray(token_ids, dtype=np.int32)
        T = int(ids.size)
        with self._launch_lock:
            if self._graphs.get(T, 0) is None:      # capture failed once for this length: plain launches
                g = None
            else:
                try:
                    g = self._single_sequence_graph(T)
                except RuntimeError as e:           # e.g. a capture invalidated by the runtime: never fatal
                    logger.warning("HIP graph capture for %d tokens failed (%s): plain launches from now on", T, e)
                    g = self._graphs[T] = None
            if g is not None:
                graph, ids_dev, out = g[:3]
                ids_dev.copy_(torch.from_numpy(ids), non_blocking=True)
                graph.replay()
                return out.clone()    # the graph's output row is overwritten by the next replay
        d = self.device
        return self.forward_packed(torch.from_numpy(ids).to(d, non_blocking=True), torch.arange(T, dtype=torch.int32, device=d),
                                   torch.tensor([0, T], dtype=torch.int32, device=d), T)

    def forward_packed(self, ids: torch.Tensor, pos_ids: torch.Tensor, cu_seqlens: torch.Tensor, max_len: int,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
        T, B = ids.numel(), cu_seqlens.numel() - 1
        need = _native.encoder_workspace_bytes(self.desc, T, B, self._f32)
        with stage("encode.lock"):
            self._launch_lock.acquire()
        try:
            with stage("encode.launch"):
                if self._workspace is None or self._workspace.numel() < need:
                    self._workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
                return _native.encoder_forward(self.desc, self._ptrs, ids, pos_ids, cu_seqlens, max_len,
                                               workspace=self._workspace, out=out, f32=self._f32)
        finally:
            self._launch_lock.release()

    @property
    def dim(self) -> int:
        return self.cfg.hidden

    def flops_per_sequence(self, s: int) -> float:
        """L * S * (24 H^2 + 4 S H)"""
        H = self.cfg.hidden
        return float(self.cfg.n_layers) * s * (24.0 * H * H + 4.0 * s * H)
