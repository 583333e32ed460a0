"""Host side of the CLIP towers (BASELINE config 4: joint text/image space, ViT-B/32 shape).

The reference names `CLIP_MODEL="ViT-B/32"` (config.py:106) but never loads it;
this is the north-star extension: both towers run as HIP kernels (mmrag_encoder_forward for
text with causal attention and EOS pooling, mmrag_vit_forward for images) and produce
L2-normalised vectors in one `proj`-dimensional space that the same VectorIndex searches.
Weights use Hugging Face `CLIPModel` state_dict names (local checkpoint or random init).
"""
from __future__ import annotations

import ctypes
import threading
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _native


@dataclass(frozen=True)
class ClipConfig:
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
    def tokens_per_image(self) -> int:
        return (self.image // self.patch) ** 2 + 1


VIT_B32 = ClipConfig()


def random_clip_weights(cfg: ClipConfig, seed: int = 0, device="cuda:0", std: float = 0.02) -> Dict[str, torch.Tensor]:
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    mat = lambda *sh: torch.randn(sh, generator=g, device=dev) * std  # noqa: E731
    vec = lambda n, base=0.0: base + torch.randn((n,), generator=g, device=dev) * 0.02  # noqa: E731
    w = {"text_model.embeddings.token_embedding.weight": mat(cfg.vocab, cfg.t_hidden),
         "text_model.embeddings.position_embedding.weight": mat(cfg.t_max_pos, cfg.t_hidden),
         "text_model.final_layer_norm.weight": vec(cfg.t_hidden, 1.0), "text_model.final_layer_norm.bias": vec(cfg.t_hidden),
         "vision_model.embeddings.class_embedding": vec(cfg.v_hidden),
         "vision_model.embeddings.patch_embedding.weight": mat(cfg.v_hidden, 3, cfg.patch, cfg.patch),
         "vision_model.embeddings.position_embedding.weight": mat(cfg.tokens_per_image, cfg.v_hidden),
         "vision_model.pre_layrnorm.weight": vec(cfg.v_hidden, 1.0), "vision_model.pre_layrnorm.bias": vec(cfg.v_hidden),
         "vision_model.post_layernorm.weight": vec(cfg.v_hidden, 1.0), "vision_model.post_layernorm.bias": vec(cfg.v_hidden),
         "visual_projection.weight": mat(cfg.proj, cfg.v_hidden), "text_projection.weight": mat(cfg.proj, cfg.t_hidden)}
    for tower, L, H, I in (("text_model", cfg.t_layers, cfg.t_hidden, cfg.t_inter),
                           ("vision_model", cfg.v_layers, cfg.v_hidden, cfg.v_inter)):
        for l in range(L):
            p = f"{tower}.encoder.layers.{l}."
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                w[p + f"self_attn.{n}.weight"] = mat(H, H)
                w[p + f"self_attn.{n}.bias"] = vec(H)
            for n in ("layer_norm1", "layer_norm2"):
                w[p + n + ".weight"] = vec(H, 1.0)
                w[p + n + ".bias"] = vec(H)
            w[p + "mlp.fc1.weight"] = mat(I, H)
            w[p + "mlp.fc1.bias"] = vec(I)
            w[p + "mlp.fc2.weight"] = mat(H, I)
            w[p + "mlp.fc2.bias"] = vec(H)
    return w


class _Tower:
    def __init__(self, device):
        self.device = torch.device(device)
        self.tensors: List[torch.Tensor] = []
        self.ptrs: List[Optional[int]] = []

    def up(self, x, dtype):
        if x is None:
            self.ptrs.append(None)
            return
        t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
        t = t.to(device=self.device, dtype=dtype).contiguous()
        self.tensors.append(t)
        self.ptrs.append(t.data_ptr())

    def layers(self, w, tower, n_layers):
        cat = lambda xs: torch.cat([x if isinstance(x, torch.Tensor) else torch.from_numpy(np.asarray(x)) for x in xs], 0)  # noqa: E731
        for l in range(n_layers):
            p = f"{tower}.encoder.layers.{l}."
            self.up(cat([w[p + f"self_attn.{n}.weight"] for n in ("q_proj", "k_proj", "v_proj")]), torch.float16)
            self.up(cat([w[p + f"self_attn.{n}.bias"] for n in ("q_proj", "k_proj", "v_proj")]), torch.float32)
            self.up(w[p + "self_attn.out_proj.weight"], torch.float16)
            self.up(w[p + "self_attn.out_proj.bias"], torch.float32)
            self.up(w[p + "layer_norm1.weight"], torch.float32)
            self.up(w[p + "layer_norm1.bias"], torch.float32)
            self.up(w[p + "mlp.fc1.weight"], torch.float16)
            self.up(w[p + "mlp.fc1.bias"], torch.float32)
            self.up(w[p + "mlp.fc2.weight"], torch.float16)
            self.up(w[p + "mlp.fc2.bias"], torch.float32)
            self.up(w[p + "layer_norm2.weight"], torch.float32)
            self.up(w[p + "layer_norm2.bias"], torch.float32)

    def table(self):
        return (ctypes.c_void_p * len(self.ptrs))(*self.ptrs)


class DeviceClip:
    """Both CLIP towers resident on one GPU."""

    def __init__(self, cfg: ClipConfig, weights: Dict[str, "np.ndarray | torch.Tensor"], device="cuda:0"):
        _native.lib()
        self.cfg = cfg
        self.device = torch.device(device)
        w = weights
        # ---- text tower table (include/mmrag.h: MMRAG_ARCH_PRELN) ----
        t = _Tower(device)
        t.up(w["text_model.embeddings.token_embedding.weight"], torch.float16)
        t.up(w["text_model.embeddings.position_embedding.weight"], torch.float16)
        t.up(None, None)
        t.up(None, None)
        t.up(None, None)
        t.layers(w, "text_model", cfg.t_layers)
        t.up(w["text_model.final_layer_norm.weight"], torch.float32)
        t.up(w["text_model.final_layer_norm.bias"], torch.float32)
        t.up(w["text_projection.weight"], torch.float16)
        self._text, self._text_tab = t, t.table()
        self.text_desc = _native.EncoderDesc(
            arch=_native.ARCH_PRELN, n_layers=cfg.t_layers, hidden=cfg.t_hidden, n_heads=cfg.t_heads,
            intermediate=cfg.t_inter, vocab=cfg.vocab, max_pos=cfg.t_max_pos, pool=_native.POOL_SELECT,
            act=_native.ACT_QUICK_GELU, causal=1, normalize=1, out_dim=cfg.proj, ln_eps=cfg.ln_eps, image=0, patch=0)
        # ---- vision tower table ----
        v = _Tower(device)
        pw = w["vision_model.embeddings.patch_embedding.weight"]
        pw = pw if isinstance(pw, torch.Tensor) else torch.from_numpy(np.asarray(pw))
        v.up(pw.reshape(cfg.v_hidden, -1), torch.float16)
        v.up(w["vision_model.embeddings.position_embedding.weight"], torch.float16)
        v.up(w["vision_model.embeddings.class_embedding"], torch.float16)
        v.up(w["vision_model.pre_layrnorm.weight"], torch.float32)
        v.up(w["vision_model.pre_layrnorm.bias"], torch.float32)
        v.layers(w, "vision_model", cfg.v_layers)
        v.up(w["vision_model.post_layernorm.weight"], torch.float32)
        v.up(w["vision_model.post_layernorm.bias"], torch.float32)
        v.up(w["visual_projection.weight"], torch.float16)
        self._vis, self._vis_tab = v, v.table()
        self.vis_desc = _native.EncoderDesc(
            arch=_native.ARCH_PRELN, n_layers=cfg.v_layers, hidden=cfg.v_hidden, n_heads=cfg.v_heads,
            intermediate=cfg.v_inter, vocab=0, max_pos=cfg.tokens_per_image, pool=_native.POOL_FIRST,
            act=_native.ACT_QUICK_GELU, causal=0, normalize=1, out_dim=cfg.proj, ln_eps=cfg.ln_eps,
            image=cfg.image, patch=cfg.patch)
        self._ws_t: Optional[torch.Tensor] = None
        self._ws_v: Optional[torch.Tensor] = None
        self._launch_lock = threading.Lock()

    @classmethod
    def random_init(cls, cfg: ClipConfig = VIT_B32, seed: int = 0, device="cuda:0") -> "DeviceClip":
        return cls(cfg, random_clip_weights(cfg, seed, device), device)

    @classmethod
    def from_local_dir(cls, path: str, device="cuda:0") -> "DeviceClip":
        """Load a Hugging Face CLIPModel checkpoint directory (config.json + model.safetensors).
        Nothing is downloaded; safetensors executes nothing from the file."""
        import json
        import os

        from safetensors.numpy import load_file

        with open(os.path.join(path, "config.json")) as f:
            c = json.load(f)
        t, v = c.get("text_config", {}), c.get("vision_config", {})
        d = ClipConfig()
        cfg = ClipConfig(
            t.get("num_hidden_layers", d.t_layers), t.get("hidden_size", d.t_hidden),
            t.get("num_attention_heads", d.t_heads), t.get("intermediate_size", d.t_inter),
            t.get("vocab_size", d.vocab), t.get("max_position_embeddings", d.t_max_pos),
            t.get("eos_token_id", d.eos_id) if t.get("eos_token_id", 2) != 2 else t.get("vocab_size", d.vocab) - 1,
            v.get("num_hidden_layers", d.v_layers), v.get("hidden_size", d.v_hidden),
            v.get("num_attention_heads", d.v_heads), v.get("intermediate_size", d.v_inter),
            v.get("image_size", d.image), v.get("patch_size", d.patch), c.get("projection_dim", d.proj),
            v.get("layer_norm_eps", d.ln_eps))
        return cls(cfg, load_file(os.path.join(path, "model.safetensors")), device)

    @property
    def dim(self) -> int:
        return self.cfg.proj

    def encode_text_ids(self, sequences: Sequence[Sequence[int]]) -> torch.Tensor:
        """Token-id sequences (BOS ... EOS) -> [B, proj] float32, L2-normalised."""
        L = self.cfg.t_max_pos
        seqs = [np.asarray(s[:L], dtype=np.int32) for s in sequences]
        lens = np.array([len(s) for s in seqs], np.int32)
        cu = np.zeros(len(seqs) + 1, np.int32)
        np.cumsum(lens, out=cu[1:])
        eos = np.array([int(np.nonzero(s == self.cfg.eos_id)[0][0]) if np.any(s == self.cfg.eos_id)
                        else int(np.argmax(s)) for s in seqs], np.int32)
        d = self.device
        ids = torch.from_numpy(np.concatenate(seqs)).to(d)
        pos = torch.from_numpy(np.concatenate([np.arange(n, dtype=np.int32) for n in lens])).to(d)
        need = _native.encoder_workspace_bytes(self.text_desc, int(cu[-1]), len(seqs))
        with self._launch_lock:   # one workspace per tower: launches of two forwards must not interleave
            if self._ws_t is None or self._ws_t.numel() < need:
                self._ws_t = torch.empty(need, dtype=torch.uint8, device=d)
            return _native.encoder_forward(self.text_desc, self._text_tab, ids, pos, torch.from_numpy(cu).to(d),
                                           int(lens.max()), sel=torch.from_numpy(eos).to(d), workspace=self._ws_t)

    def encode_images(self, pixels: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """uint8 [B, image, image, 3] raw crops (normalised on the GPU) or fp16 [B, 3, image, image]
        already normalised -> [B, proj] float32, L2-normalised."""
        if pixels.dtype == torch.uint8:
            kind = _native.PIXELS_U8_HWC
            ok = pixels.shape[1:] == (self.cfg.image, self.cfg.image, 3)
        else:
            kind = _native.PIXELS_F16_CHW
            ok = pixels.dtype == torch.float16 and pixels.shape[1:] == (3, self.cfg.image, self.cfg.image)
        if not ok:
            raise ValueError(f"bad image tensor {tuple(pixels.shape)} {pixels.dtype}")
        pixels = pixels.to(self.device).contiguous()
        B, S = pixels.shape[0], self.cfg.tokens_per_image
        cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device=self.device)
        need = _native.encoder_workspace_bytes(self.vis_desc, B * S, B)
        with self._launch_lock:
            if self._ws_v is None or self._ws_v.numel() < need:
                self._ws_v = torch.empty(need, dtype=torch.uint8, device=self.device)
            return _native.vit_forward(self.vis_desc, self._vis_tab, pixels, kind, cu, workspace=self._ws_v, out=out)

    def image_flops(self) -> float:
        c = self.cfg
        s = c.tokens_per_image
        return c.v_layers * s * (24.0 * c.v_hidden ** 2 + 4.0 * s * c.v_hidden) + 2.0 * (s - 1) * 3 * c.patch ** 2 * c.v_hidden


def clip_resize_geometry(H: int, W: int, size: int = 224):
    """CLIP processor geometry: shortest edge -> `size` (long edge = int(size * long / short)), centre crop.
    Returns (new_h, new_w, top, left)."""
    if H <= W:
        new_h, new_w = size, int(size * W / H)
    else:
        new_h, new_w = int(size * H / W), size
    return new_h, new_w, (new_h - size) // 2, (new_w - size) // 2


class ClipImagePreprocessor:
    """uint8 HWC images of any size -> [n, size, size, 3] uint8 on the device: the resize + centre crop of
    CLIP's processor as two integer HIP passes (bit-exact with PIL bicubic); rescale + normalise stay fused
    in the vision tower's patchify kernel.  Tap tables are computed on the host per source size and cached
    on the device."""

    def __init__(self, device="cuda:0", size: int = 224, max_cached_sizes: int = 64):
        self.device = torch.device(device)
        self.size = size
        self._tables: Dict[tuple, tuple] = {}
        self._max = max_cached_sizes

    def tables(self, H: int, W: int):
        key = (H, W)
        t = self._tables.get(key)
        if t is None:
            new_h, new_w, top, left = clip_resize_geometry(H, W, self.size)
            bx, kx = _native.resample_coeffs(W, new_w, left, self.size)
            by, ky = _native.resample_coeffs(H, new_h, top, self.size)
            y_lo = int(by[:, 0].min())
            y_hi = int((by[:, 0] + by[:, 1]).max())
            dev = [torch.from_numpy(a).to(self.device) for a in (bx, kx, by, ky)]
            if len(self._tables) >= self._max:
                self._tables.pop(next(iter(self._tables)))
            t = self._tables[key] = (*dev, y_lo, y_hi)
        return t

    def __call__(self, images, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """images: sequence of [H, W, 3] uint8 numpy arrays / tensors (host or device)."""
        n = len(images)
        if out is None:
            out = torch.empty((n, self.size, self.size, 3), dtype=torch.uint8, device=self.device)
        for i, img in enumerate(images):
            if isinstance(img, np.ndarray) and not img.flags.writeable:
                img = np.array(img)  # e.g. a PIL-backed view; torch wants writable memory
            t = torch.as_tensor(img)
            if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
                raise ValueError("images must be [H, W, 3] uint8 (RGB)")
            t = t.contiguous().to(self.device, non_blocking=True)
            bx, kx, by, ky, y_lo, y_hi = self.tables(int(t.shape[0]), int(t.shape[1]))
            _native.resize_crop_u8(t, bx, kx, by, ky, y_lo, y_hi, out=out[i])
        return out
