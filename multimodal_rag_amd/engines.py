"""The MI355X engines behind `EmbeddingManager`: tokenizer + encoder + collection factory.

`HipEngine` stands where the reference has `SentenceTransformer(name)` and `chromadb.Client(...)`
(app/utils/embedder.py:165-182, :245-248); `ClipEngine` is the joint text/image space of BASELINE config 4 (the
reference names CLIP in config.py:106 but never loads it).  Constructing either without a GPU or
without libmmrag.so raises: there is no second backend.
"""
from __future__ import annotations

import hashlib
import logging
from typing import Any, Dict, List, Optional

import numpy as np

from .config import settings
from .tracing import stage

logger = logging.getLogger(__name__)


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
            self.encoder = DeviceEncoder.from_local_dir(model_dir, self.device, precision=settings.MMRAG_ENCODER_PRECISION)
            vocab = os.path.join(model_dir, "vocab.txt")
            self.tokenizer = (NativeWordPieceTokenizer.from_vocab_file(vocab) if os.path.exists(vocab)
                              else HashTokenizer(self.encoder.cfg.vocab))
        else:
            if model_name not in PRESETS:
                raise ValueError(f"unknown model {model_name!r}: give MMRAG_MODEL_DIR or one of {sorted(PRESETS)}")
            logger.warning("No local checkpoint (MMRAG_MODEL_DIR unset): %s architecture with seeded random "
                           "weights and the stand-in hash tokenizer", model_name)
            self.encoder = DeviceEncoder.random_init(PRESETS[model_name], settings.MMRAG_WEIGHT_SEED, self.device,
                                                     precision=settings.MMRAG_ENCODER_PRECISION)
            self.tokenizer = HashTokenizer(self.encoder.cfg.vocab)
        self.dim = self.encoder.dim
        self.max_seq_length = self.encoder.cfg.max_seq_length
        self._torch = torch

    def encode(self, texts: List[str]) -> np.ndarray:
        if hasattr(self.tokenizer, "encode_batch_arrays"):   # native, multi-threaded tokenizer
            with stage("tokenize"):
                rows = self.tokenizer.encode_batch_arrays(texts, self.max_seq_length)
            with stage("encode"):
                return self.encoder.encode_id_rows(*rows).cpu().numpy()
        with stage("tokenize"):
            seqs = [self.tokenizer.encode(t, self.max_seq_length) for t in texts]
        with stage("encode"):
            return self.encoder.encode_ids(seqs).cpu().numpy()

    def encode_device(self, texts: List[str]):
        """the same embeddings as encode(), left on the device ([n, dim] float32): the query path hands them straight
        to the search kernel, with no device -> host -> device round trip and no host wait in between"""
        if hasattr(self.tokenizer, "encode_batch_arrays"):
            with stage("tokenize"):
                rows = self.tokenizer.encode_batch_arrays(texts, self.max_seq_length)
            with stage("encode"):
                return self.encoder.encode_id_rows(*rows)
        with stage("tokenize"):
            seqs = [self.tokenizer.encode(t, self.max_seq_length) for t in texts]
        with stage("encode"):
            return self.encoder.encode_ids(seqs)

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
        from .tokenizer import NativeClipBpeTokenizer

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
                # native (csrc/clip_bpe.cpp): equals tokenizer.ClipBpeTokenizer, at C++ speed over the whole batch
                self.tokenizer = NativeClipBpeTokenizer.from_files(vj, mt, self.clip.cfg.t_max_pos)
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
        if self.tokenizer is not None:   # one native call for the batch
            return self.clip.encode_text_ids(self.tokenizer.encode_batch(list(texts), self.max_seq_length)).cpu().numpy()
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
