#!/usr/bin/env python3
"""Pin oracle/clip_oracle.py against transformers.CLIPModel (local configs, seeded random weights;
BUILD container only).  Writes tests/golden/clip_<name>.npz = {seed, ids, lens, pixels(u8 tiles),
text, image}.  Weights are regenerated from the seed by make_clip_weights()."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import clip_oracle as C  # noqa: E402


def hf_clip(s, w):
    from transformers import CLIPConfig, CLIPModel, CLIPTextConfig, CLIPVisionConfig

    tc = CLIPTextConfig(vocab_size=s.vocab, hidden_size=s.t_hidden, intermediate_size=s.t_inter,
                        num_hidden_layers=s.t_layers, num_attention_heads=s.t_heads,
                        max_position_embeddings=s.t_max_pos, eos_token_id=s.eos_id, bos_token_id=s.eos_id - 1,
                        pad_token_id=0, hidden_act="quick_gelu", layer_norm_eps=s.ln_eps, attention_dropout=0.0)
    vc = CLIPVisionConfig(hidden_size=s.v_hidden, intermediate_size=s.v_inter, num_hidden_layers=s.v_layers,
                          num_attention_heads=s.v_heads, image_size=s.image, patch_size=s.patch,
                          hidden_act="quick_gelu", layer_norm_eps=s.ln_eps, attention_dropout=0.0)
    m = CLIPModel(CLIPConfig(text_config=tc.to_dict(), vision_config=vc.to_dict(), projection_dim=s.proj))
    sd = m.state_dict()
    for k, v in w.items():
        assert k in sd and tuple(sd[k].shape) == v.shape, (k, v.shape)
        sd[k] = torch.from_numpy(v)
    m.load_state_dict(sd, strict=False)
    return m.eval()


def one(name, s, seed, lens, n_img):
    w = C.make_clip_weights(s, seed)
    g = np.random.default_rng(seed + 1)
    S = s.t_max_pos
    ids = np.zeros((len(lens), S), np.int64)
    mask = np.zeros((len(lens), S), np.int64)
    for b, n in enumerate(lens):
        ids[b, 0] = s.eos_id - 1
        ids[b, 1:n - 1] = g.integers(1, s.eos_id - 1, n - 2)
        ids[b, n - 1] = s.eos_id
        mask[b, :n] = 1
    tiles = g.integers(0, 256, (n_img, s.image, s.image, 3), dtype=np.uint8)
    px = C.preprocess_tiles(tiles)
    m = hf_clip(s, w)
    norm = lambda x: x / np.linalg.norm(x, axis=1, keepdims=True)
    with torch.no_grad():
        t = m.get_text_features(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask))
        v = m.get_image_features(pixel_values=torch.from_numpy(px))
        t = (t.pooler_output if hasattr(t, "pooler_output") else t).numpy()
        v = (v.pooler_output if hasattr(v, "pooler_output") else v).numpy()
    np.savez_compressed(os.path.join(HERE, f"clip_{name}.npz"), seed=seed, ids=ids.astype(np.int32),
                        lens=np.asarray(lens, np.int32), tiles=tiles, text=norm(t).astype(np.float32),
                        image=norm(v).astype(np.float32))
    seqs = [ids[b, :n] for b, n in enumerate(lens)]
    print(f"{name}: text max|oracle-hf| = {np.abs(C.text_embed(s, w, seqs) - norm(t)).max():.2e}   "
          f"image max|oracle-hf| = {np.abs(C.image_embed(s, w, px) - norm(v)).max():.2e}")


if __name__ == "__main__":
    one("tiny", C.TINY_CLIP, 21, [4, 17, 32, 9], 3)
    one("vitb32", C.VIT_B32, 22, [8, 77, 30], 2)
