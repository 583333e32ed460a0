#!/usr/bin/env python3
"""Pin oracle/encoder_oracle.py against `transformers` model classes built from LOCAL configs
with seeded random weights (no download; BUILD container only -- transformers does not need to
exist on the GPU box, only the .npz files written here travel).

    python tests/golden/make_encoder_golden.py

Writes tests/golden/encoder_<name>.npz = {seed, ids (padded), lens, hidden_cls/mean outputs}.
Weights are NOT stored: make_bert_weights(shape, seed) regenerates them bit-identically
(numpy PCG64 is stable across numpy versions).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import encoder_oracle as E  # noqa: E402


def hf_bert(shape: E.BertShape, w):
    from transformers import BertConfig, BertModel

    cfg = BertConfig(vocab_size=shape.vocab, hidden_size=shape.hidden, num_hidden_layers=shape.n_layers,
                     num_attention_heads=shape.n_heads, intermediate_size=shape.intermediate,
                     max_position_embeddings=shape.max_pos, hidden_act="gelu", layer_norm_eps=shape.ln_eps,
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = BertModel(cfg, add_pooling_layer=False)
    sd = m.state_dict()
    for k, v in w.items():
        assert k in sd and tuple(sd[k].shape) == v.shape, k
        sd[k] = torch.from_numpy(v)
    m.load_state_dict(sd, strict=False)
    return m.eval()


def one(name, shape, seed, lens, S):
    w = E.make_bert_weights(shape, seed)
    g = np.random.default_rng(seed + 1)
    B = len(lens)
    ids = np.zeros((B, S), dtype=np.int64)
    mask = np.zeros((B, S), dtype=np.int64)
    for b, n in enumerate(lens):
        ids[b, 0] = 101 % shape.vocab
        ids[b, 1:n - 1] = g.integers(min(1000, shape.vocab // 2), shape.vocab, n - 2)
        ids[b, n - 1] = 102 % shape.vocab
        mask[b, :n] = 1
    m = hf_bert(shape, w)
    with torch.no_grad():
        hs = m(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask)).last_hidden_state.numpy()
    mk = mask[..., None].astype(np.float32)
    mean = (hs * mk).sum(1) / np.clip(mk.sum(1), 1e-9, None)   # sentence-transformers Pooling (mean)
    cls = hs[:, 0]
    norm = lambda x: x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-12)
    np.savez_compressed(os.path.join(HERE, f"encoder_{name}.npz"), seed=seed, ids=ids.astype(np.int32),
                        lens=np.asarray(lens, np.int32), mean=norm(mean).astype(np.float32),
                        cls=norm(cls).astype(np.float32))
    # immediate self-check of the restatement
    seqs = [ids[b, :n] for b, n in enumerate(lens)]
    import dataclasses
    for pool, ref in (("mean", norm(mean)), ("cls", norm(cls))):
        got = E.bert_encode(dataclasses.replace(shape, pool=pool), w, seqs)
        print(f"{name} pool={pool}: max |oracle - transformers| = {np.abs(got - ref).max():.3e}")


if __name__ == "__main__":
    one("tiny", E.TINY, 11, [5, 17, 64, 33], 64)
    one("minilm", E.MINILM_L6, 12, [9, 256, 100, 31], 256)
    one("bge", E.BGE_BASE, 13, [12, 300, 47], 300)
