"""CPU oracle for the CLIP towers (BASELINE config 4).  TEST INFRASTRUCTURE ONLY.

The reference never runs CLIP (`CLIP_MODEL="ViT-B/32"` is only a config string,
config.py:106, requirements.txt:63-67 commented out), so config 4 is a north-star extension with
no reference behaviour to match.  This file restates the PUBLISHED architecture (Radford et al.
2021; Hugging Face `CLIPModel` semantics) and is pinned against `transformers.CLIPModel` built from
local configs with seeded random weights (tests/golden/make_clip_golden.py -> clip_*.npz):

  text   : token + position embeddings -> 12 x pre-LN block [LN1 -> causal MHA -> +res -> LN2 ->
           fc1 -> quick_gelu -> fc2 -> +res] -> final LN -> hidden state at the EOS token ->
           bias-free projection -> L2 normalise.
  vision : 32x32 stride-32 conv (no bias) as a [49, 3072] x [3072, 768] GEMM, class token, position
           embeddings, pre-LN, 12 x pre-LN block (no mask), post-LN on the class token, bias-free
           projection -> L2 normalise.
Float32 numpy throughout.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Sequence

import numpy as np

from .encoder_oracle import attention, l2_normalize, layer_norm, quick_gelu


@dataclass(frozen=True)
class ClipShape:
    t_layers: int = 12
    t_hidden: int = 512
    t_heads: int = 8
    t_inter: int = 2048
    vocab: int = 49408
    t_max_pos: int = 77
    eos_id: int = 49407
    v_layers: int = 12
    v_hidden: int = 768
    v_heads: int = 12
    v_inter: int = 3072
    image: int = 224
    patch: int = 32
    proj: int = 512
    ln_eps: float = 1e-5

    @property
    def n_patches(self) -> int:
        return (self.image // self.patch) ** 2


VIT_B32 = ClipShape()
TINY_CLIP = ClipShape(2, 128, 4, 256, 1000, 32, 999, 2, 128, 4, 256, 64, 32, 64)


def make_clip_weights(s: ClipShape, seed: int, std: float = 0.05) -> Dict[str, np.ndarray]:
    g = np.random.default_rng(seed)
    mat = lambda *sh: (g.standard_normal(sh) * std).astype(np.float32)  # noqa: E731
    vec = lambda n, sc=0.05, base=0.0: (base + g.standard_normal(n) * sc).astype(np.float32)  # noqa: E731
    w = {"text_model.embeddings.token_embedding.weight": mat(s.vocab, s.t_hidden),
         "text_model.embeddings.position_embedding.weight": mat(s.t_max_pos, s.t_hidden),
         "text_model.final_layer_norm.weight": vec(s.t_hidden, 0.1, 1.0),
         "text_model.final_layer_norm.bias": vec(s.t_hidden, 0.1),
         "vision_model.embeddings.class_embedding": vec(s.v_hidden, std),
         "vision_model.embeddings.patch_embedding.weight": mat(s.v_hidden, 3, s.patch, s.patch) * 0.3,
         "vision_model.embeddings.position_embedding.weight": mat(s.n_patches + 1, s.v_hidden),
         "vision_model.pre_layrnorm.weight": vec(s.v_hidden, 0.1, 1.0),
         "vision_model.pre_layrnorm.bias": vec(s.v_hidden, 0.1),
         "vision_model.post_layernorm.weight": vec(s.v_hidden, 0.1, 1.0),
         "vision_model.post_layernorm.bias": vec(s.v_hidden, 0.1),
         "visual_projection.weight": mat(s.proj, s.v_hidden),
         "text_projection.weight": mat(s.proj, s.t_hidden)}
    for tower, L, H, I in (("text_model", s.t_layers, s.t_hidden, s.t_inter),
                           ("vision_model", s.v_layers, s.v_hidden, s.v_inter)):
        for l in range(L):
            p = f"{tower}.encoder.layers.{l}."
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                w[p + f"self_attn.{n}.weight"] = mat(H, H)
                w[p + f"self_attn.{n}.bias"] = vec(H)
            for n in ("layer_norm1", "layer_norm2"):
                w[p + n + ".weight"] = vec(H, 0.1, 1.0)
                w[p + n + ".bias"] = vec(H, 0.1)
            w[p + "mlp.fc1.weight"] = mat(I, H)
            w[p + "mlp.fc1.bias"] = vec(I)
            w[p + "mlp.fc2.weight"] = mat(H, I)
            w[p + "mlp.fc2.bias"] = vec(H)
    return w


def round_weights_fp16(w: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """device storage: every matrix / embedding table / conv kernel / class token in fp16"""
    out = {}
    for k, v in w.items():
        as16 = v.ndim >= 2 or k.endswith("class_embedding")
        out[k] = v.astype(np.float16).astype(np.float32) if as16 else v.copy()
    return out


def _blocks(x: np.ndarray, w: Dict[str, np.ndarray], tower: str, n_layers: int, heads: int, eps: float,
            causal: bool) -> np.ndarray:
    for l in range(n_layers):
        p = f"{tower}.encoder.layers.{l}."
        a = layer_norm(x, w[p + "layer_norm1.weight"], w[p + "layer_norm1.bias"], eps)
        q = a @ w[p + "self_attn.q_proj.weight"].T + w[p + "self_attn.q_proj.bias"]
        k = a @ w[p + "self_attn.k_proj.weight"].T + w[p + "self_attn.k_proj.bias"]
        v = a @ w[p + "self_attn.v_proj.weight"].T + w[p + "self_attn.v_proj.bias"]
        o = attention(q, k, v, heads, causal=causal)
        x = x + (o @ w[p + "self_attn.out_proj.weight"].T + w[p + "self_attn.out_proj.bias"])
        b = layer_norm(x, w[p + "layer_norm2.weight"], w[p + "layer_norm2.bias"], eps)
        hm = quick_gelu(b @ w[p + "mlp.fc1.weight"].T + w[p + "mlp.fc1.bias"])
        x = x + (hm @ w[p + "mlp.fc2.weight"].T + w[p + "mlp.fc2.bias"])
    return x.astype(np.float32)


def text_embed(s: ClipShape, w: Dict[str, np.ndarray], sequences: List[Sequence[int]]) -> np.ndarray:
    out = []
    for ids in sequences:
        ids = np.asarray(ids, np.int64)
        S = len(ids)
        x = w["text_model.embeddings.token_embedding.weight"][ids] + w["text_model.embeddings.position_embedding.weight"][:S]
        x = _blocks(x, w, "text_model", s.t_layers, s.t_heads, s.ln_eps, causal=True)
        x = layer_norm(x, w["text_model.final_layer_norm.weight"], w["text_model.final_layer_norm.bias"], s.ln_eps)
        eos = int(np.nonzero(ids == s.eos_id)[0][0]) if np.any(ids == s.eos_id) else int(np.argmax(ids))
        out.append(x[eos] @ w["text_projection.weight"].T)
    return l2_normalize(np.stack(out))


def patchify(s: ClipShape, pixels: np.ndarray) -> np.ndarray:
    """[B, 3, H, W] -> [B, n_patches, 3*P*P], patch vector order (c, ph, pw) = conv weight order."""
    B = pixels.shape[0]
    P, G = s.patch, s.image // s.patch
    x = pixels.reshape(B, 3, G, P, G, P).transpose(0, 2, 4, 1, 3, 5)
    return np.ascontiguousarray(x.reshape(B, G * G, 3 * P * P)).astype(np.float32)


def image_embed(s: ClipShape, w: Dict[str, np.ndarray], pixels: np.ndarray) -> np.ndarray:
    """pixels: [B, 3, image, image] float32 (already normalised with CLIP mean/std)."""
    wp = w["vision_model.embeddings.patch_embedding.weight"].reshape(s.v_hidden, -1)
    out = []
    for patches in patchify(s, pixels):
        x = np.concatenate([w["vision_model.embeddings.class_embedding"][None, :], patches @ wp.T], 0)
        x = x + w["vision_model.embeddings.position_embedding.weight"]
        x = layer_norm(x, w["vision_model.pre_layrnorm.weight"], w["vision_model.pre_layrnorm.bias"], s.ln_eps)
        x = _blocks(x, w, "vision_model", s.v_layers, s.v_heads, s.ln_eps, causal=False)
        cls = layer_norm(x[0], w["vision_model.post_layernorm.weight"], w["vision_model.post_layernorm.bias"], s.ln_eps)
        out.append(cls @ w["visual_projection.weight"].T)
    return l2_normalize(np.stack(out))


CLIP_MEAN = np.array([0.48145466, 0.4578275, 0.40821073], np.float32)
CLIP_STD = np.array([0.26862954, 0.26130258, 0.27577711], np.float32)


def preprocess_tiles(tiles_u8: np.ndarray) -> np.ndarray:
    """[B, H, W, 3] uint8 crops -> [B, 3, H, W] float32, (x/255 - mean)/std  (CLIP image normalisation)."""
    x = tiles_u8.astype(np.float32) / np.float32(255.0)
    return np.ascontiguousarray(((x - CLIP_MEAN) / CLIP_STD).transpose(0, 3, 1, 2)).astype(np.float32)


# ---------------------------------------------------------------------------------------------
# Image front end: shortest-edge bicubic resize + centre crop on uint8 (what CLIP's processor does
# with PIL).  Integer arithmetic, restated from Pillow's published 8-bit resampler (Resample.c:
# precompute_coeffs / normalize_coeffs_8bpc / ImagingResampleHorizontal_8bpc / ..Vertical_8bpc);
# pinned bit-exactly against PIL.Image.resize and transformers.CLIPImageProcessor in
# tests/golden/make_resize_golden.py.  Test infrastructure only.
# ---------------------------------------------------------------------------------------------
PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size: int, out_size: int):
    """Per output index: (first input index, tap count) and fixed-point taps, as Pillow computes them
    for the bicubic filter over the whole input extent.  Returns (bounds [out,2] int32, taps [out,ksize] int32)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    taps = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        if ww != 0.0:
            k = [w / ww for w in k]
        for x, w in enumerate(k):
            taps[xx, x] = int(-0.5 + w * (1 << PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, taps


def _clip8(acc: np.ndarray) -> np.ndarray:
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """PIL.Image.resize((out_w, out_h), BICUBIC) of an [H, W, C] uint8 image: horizontal pass, round to
    uint8, vertical pass, round to uint8."""
    H, W, C = img.shape
    src = img.astype(np.int64)
    if out_w != W:
        bx, kx = resample_coeffs(W, out_w)
        tmp = np.empty((H, out_w, C), np.uint8)
        for xx in range(out_w):
            x0, n = bx[xx]
            acc = (src[:, x0:x0 + n, :] * kx[xx, :n].astype(np.int64)[None, :, None]).sum(axis=1) + (1 << (PRECISION_BITS - 1))
            tmp[:, xx, :] = _clip8(acc)
        src = tmp.astype(np.int64)
    else:
        tmp = img
    if out_h != H:
        by, ky = resample_coeffs(H, out_h)
        out = np.empty((out_h, src.shape[1], C), np.uint8)
        for yy in range(out_h):
            y0, n = by[yy]
            acc = (src[y0:y0 + n] * ky[yy, :n].astype(np.int64)[:, None, None]).sum(axis=0) + (1 << (PRECISION_BITS - 1))
            out[yy] = _clip8(acc)
        return out
    return np.ascontiguousarray(tmp)


def clip_resize_geometry(H: int, W: int, size: int = 224):
    """Shortest edge -> `size` keeping aspect (long edge = int(size * long / short)), then the centre
    crop's top-left corner.  Returns (new_h, new_w, top, left)."""
    if H <= W:
        new_h, new_w = size, int(size * W / H)
    else:
        new_h, new_w = int(size * H / W), size
    return new_h, new_w, (new_h - size) // 2, (new_w - size) // 2


def clip_resize_crop_u8(img: np.ndarray, size: int = 224) -> np.ndarray:
    """[H, W, 3] uint8 -> [size, size, 3] uint8: CLIP's resize + centre crop, before rescale/normalise."""
    new_h, new_w, top, left = clip_resize_geometry(img.shape[0], img.shape[1], size)
    r = resize_u8(img, new_h, new_w)
    return np.ascontiguousarray(r[top:top + size, left:left + size])
