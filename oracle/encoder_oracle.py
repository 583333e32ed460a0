"""CPU oracle for the embed half of the hot path.  TEST INFRASTRUCTURE ONLY (imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by multimodal_rag_amd/).

What it restates
----------------
The reference embeds text with `SentenceTransformer(name).encode(texts, batch_size=len(texts),
convert_to_numpy=True, normalize_embeddings=True)` (app/utils/embedder.py:247, 397-403).
sentence-transformers==2.2.2 (requirements.txt:44) and its torch/transformers model code are
third-party and absent from /root/reference, as are the trained weights and the WordPiece
vocabulary, so this file restates the *published architecture*:

  BERT encoder (Devlin et al. 2018; HF `BertModel` semantics): token + learned absolute
  position + token-type(0) embeddings -> LayerNorm(eps 1e-12) -> L x [ self-attention
  (softmax(QK^T/sqrt(dh) + key padding mask) V) -> dense -> +residual -> LayerNorm ->
  dense(4H) -> erf-GELU -> dense -> +residual -> LayerNorm ].
  all-MiniLM-L6-v2: L=6, H=384, 12 heads, I=1536, vocab 30522, pipeline Transformer -> masked
  mean pooling (sum(mask*x)/clamp(sum(mask), 1e-9)) -> x / max(||x||, 1e-12).
  bge-base-en-v1.5: L=12, H=768, 12 heads, I=3072, [CLS] pooling -> normalise.
  CLIP ViT-B/32 towers (Radford et al. 2021; HF `CLIPModel` semantics): pre-LN blocks,
  quick-GELU, causal text tower pooled at the EOS (arg-max id) token, final LayerNorm,
  bias-free projection to 512, L2 normalise.

Parity pin: the reference holds no fixture that pins the trained encoder without its weights
(SURVEY.md section 8c(iii)); the restatement is pinned against `transformers` model classes
built from local configs with seeded random weights (tests/golden/make_encoder_golden.py ->
tests/golden/encoder_*.npz).  Parity of the *trained* model is therefore "unpinned": it would
need the all-MiniLM-L6-v2 checkpoint, which cannot be fetched here.

All arithmetic is float32 numpy.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
from scipy.special import erf


@dataclass(frozen=True)
class BertShape:
    n_layers: int
    hidden: int
    n_heads: int
    intermediate: int
    vocab: int = 30522
    max_pos: int = 512
    ln_eps: float = 1e-12
    pool: str = "mean"  # "mean" (MiniLM) | "cls" (bge)


MINILM_L6 = BertShape(6, 384, 12, 1536, pool="mean")
BGE_BASE = BertShape(12, 768, 12, 3072, pool="cls")
TINY = BertShape(2, 128, 4, 256, vocab=1000, max_pos=64, pool="mean")


def make_bert_weights(shape: BertShape, seed: int, std: float = 0.05) -> Dict[str, np.ndarray]:
    """Seeded random weights with HF BertModel state_dict names (Linear weights are [out, in])."""
    g = np.random.default_rng(seed)
    H, I = shape.hidden, shape.intermediate

    def mat(o, i):
        return (g.standard_normal((o, i)) * std).astype(np.float32)

    def vec(n, scale=0.05, base=0.0):
        return (base + g.standard_normal(n) * scale).astype(np.float32)

    w = {
        "embeddings.word_embeddings.weight": mat(shape.vocab, H),
        "embeddings.position_embeddings.weight": mat(shape.max_pos, H),
        "embeddings.token_type_embeddings.weight": mat(2, H),
        "embeddings.LayerNorm.weight": vec(H, 0.1, 1.0),
        "embeddings.LayerNorm.bias": vec(H, 0.1),
    }
    for l in range(shape.n_layers):
        p = f"encoder.layer.{l}."
        for name in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense"):
            w[p + name + ".weight"] = mat(H, H)
            w[p + name + ".bias"] = vec(H)
        w[p + "attention.output.LayerNorm.weight"] = vec(H, 0.1, 1.0)
        w[p + "attention.output.LayerNorm.bias"] = vec(H, 0.1)
        w[p + "intermediate.dense.weight"] = mat(I, H)
        w[p + "intermediate.dense.bias"] = vec(I)
        w[p + "output.dense.weight"] = mat(H, I)
        w[p + "output.dense.bias"] = vec(H)
        w[p + "output.LayerNorm.weight"] = vec(H, 0.1, 1.0)
        w[p + "output.LayerNorm.bias"] = vec(H, 0.1)
    return w


def round_weights_fp16(w: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """What the device stores: matrices and embedding tables in fp16 (biases / LayerNorm fp32)."""
    out = {}
    for k, v in w.items():
        is_matrix = v.ndim == 2
        out[k] = v.astype(np.float16).astype(np.float32) if is_matrix else v.copy()
    return out


def layer_norm(x: np.ndarray, g: np.ndarray, b: np.ndarray, eps: float) -> np.ndarray:
    mu = x.mean(axis=-1, keepdims=True, dtype=np.float32)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True, dtype=np.float32)
    return ((x - mu) / np.sqrt(var + np.float32(eps)) * g + b).astype(np.float32)


def gelu_erf(x: np.ndarray) -> np.ndarray:
    return (0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))).astype(np.float32)


def quick_gelu(x: np.ndarray) -> np.ndarray:
    return (x / (1.0 + np.exp(-1.702 * x))).astype(np.float32)


def attention(q: np.ndarray, k: np.ndarray, v: np.ndarray, n_heads: int, causal: bool = False) -> np.ndarray:
    """One sequence: q, k, v [S, H] -> [S, H]."""
    S, H = q.shape
    dh = H // n_heads
    out = np.empty_like(q)
    for h in range(n_heads):
        sl = slice(h * dh, (h + 1) * dh)
        s = (q[:, sl] @ k[:, sl].T) / np.float32(np.sqrt(dh))
        if causal:
            s = np.where(np.tril(np.ones((S, S), bool)), s, np.float32(-np.inf))
        s = s - s.max(axis=1, keepdims=True)
        p = np.exp(s)
        p /= p.sum(axis=1, keepdims=True)
        out[:, sl] = p @ v[:, sl]
    return out.astype(np.float32)


def bert_hidden_states(shape: BertShape, w: Dict[str, np.ndarray], ids: Sequence[int]) -> np.ndarray:
    """Last hidden state [S, H] of one un-padded sequence (padding never influences valid tokens)."""
    ids = np.asarray(ids, dtype=np.int64)
    S = ids.shape[0]
    x = (w["embeddings.word_embeddings.weight"][ids] + w["embeddings.position_embeddings.weight"][:S]
         + w["embeddings.token_type_embeddings.weight"][0])
    x = layer_norm(x, w["embeddings.LayerNorm.weight"], w["embeddings.LayerNorm.bias"], shape.ln_eps)
    for l in range(shape.n_layers):
        p = f"encoder.layer.{l}."
        q = x @ w[p + "attention.self.query.weight"].T + w[p + "attention.self.query.bias"]
        k = x @ w[p + "attention.self.key.weight"].T + w[p + "attention.self.key.bias"]
        v = x @ w[p + "attention.self.value.weight"].T + w[p + "attention.self.value.bias"]
        a = attention(q, k, v, shape.n_heads)
        a = a @ w[p + "attention.output.dense.weight"].T + w[p + "attention.output.dense.bias"]
        x = layer_norm(x + a, w[p + "attention.output.LayerNorm.weight"], w[p + "attention.output.LayerNorm.bias"],
                       shape.ln_eps)
        hmid = gelu_erf(x @ w[p + "intermediate.dense.weight"].T + w[p + "intermediate.dense.bias"])
        o = hmid @ w[p + "output.dense.weight"].T + w[p + "output.dense.bias"]
        x = layer_norm(x + o, w[p + "output.LayerNorm.weight"], w[p + "output.LayerNorm.bias"], shape.ln_eps)
    return x


def l2_normalize(x: np.ndarray) -> np.ndarray:
    n = np.sqrt((x.astype(np.float32) ** 2).sum(axis=-1, keepdims=True))
    return (x / np.maximum(n, np.float32(1e-12))).astype(np.float32)


def bert_encode(shape: BertShape, w: Dict[str, np.ndarray], sequences: List[Sequence[int]],
                normalize: bool = True) -> np.ndarray:
    """`SentenceTransformer.encode(..., normalize_embeddings=True)` on already-tokenised input
    (embedder.py:397-403): returns [B, H] float32."""
    out = []
    for ids in sequences:
        hs = bert_hidden_states(shape, w, ids)
        if shape.pool == "mean":
            pooled = hs.sum(axis=0) / np.float32(max(len(ids), 1e-9))
        else:
            pooled = hs[0]
        out.append(pooled)
    e = np.stack(out).astype(np.float32)
    return l2_normalize(e) if normalize else e


# ---- device weight table (same order as include/mmrag.h documents) -------------------------
def bert_weight_table(shape: BertShape, w: Dict[str, np.ndarray]):
    """(name, array, kind) triples in libmmrag's table order; kind 'h' = fp16 matrix, 'f' = fp32."""
    t = [("tok", w["embeddings.word_embeddings.weight"], "h"),
         ("pos", w["embeddings.position_embeddings.weight"], "h"),
         ("type0", w["embeddings.token_type_embeddings.weight"][0], "h"),
         ("emb_ln_g", w["embeddings.LayerNorm.weight"], "f"),
         ("emb_ln_b", w["embeddings.LayerNorm.bias"], "f")]
    for l in range(shape.n_layers):
        p = f"encoder.layer.{l}."
        wqkv = np.concatenate([w[p + f"attention.self.{n}.weight"] for n in ("query", "key", "value")], axis=0)
        bqkv = np.concatenate([w[p + f"attention.self.{n}.bias"] for n in ("query", "key", "value")], axis=0)
        t += [(f"l{l}.wqkv", wqkv, "h"), (f"l{l}.bqkv", bqkv, "f"),
              (f"l{l}.wo", w[p + "attention.output.dense.weight"], "h"),
              (f"l{l}.bo", w[p + "attention.output.dense.bias"], "f"),
              (f"l{l}.ln1_g", w[p + "attention.output.LayerNorm.weight"], "f"),
              (f"l{l}.ln1_b", w[p + "attention.output.LayerNorm.bias"], "f"),
              (f"l{l}.w1", w[p + "intermediate.dense.weight"], "h"),
              (f"l{l}.b1", w[p + "intermediate.dense.bias"], "f"),
              (f"l{l}.w2", w[p + "output.dense.weight"], "h"),
              (f"l{l}.b2", w[p + "output.dense.bias"], "f"),
              (f"l{l}.ln2_g", w[p + "output.LayerNorm.weight"], "f"),
              (f"l{l}.ln2_b", w[p + "output.LayerNorm.bias"], "f")]
    return t
