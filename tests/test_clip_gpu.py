"""CLIP towers (BASELINE config 4): CPU pin of the oracle against the transformers goldens, GPU
parity of the HIP towers against the oracle on the same fp16-rounded weights."""
import os

import numpy as np
import pytest
import torch

from oracle import clip_oracle as C

SHAPES = {"tiny": C.TINY_CLIP, "vitb32": C.VIT_B32}


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, f"clip_{name}.npz"))
    seqs = [z["ids"][b, :n].tolist() for b, n in enumerate(z["lens"])]
    return z, seqs


@pytest.mark.parametrize("name", ["tiny", "vitb32"])
def test_oracle_matches_transformers_golden(golden_dir, name):
    if name == "vitb32" and os.environ.get("MMRAG_FAST_TESTS"):
        pytest.skip("fast mode")
    s = SHAPES[name]
    z, seqs = load(golden_dir, name)
    w = C.make_clip_weights(s, int(z["seed"]))
    assert np.abs(C.text_embed(s, w, seqs) - z["text"]).max() < 2e-6
    assert np.abs(C.image_embed(s, w, C.preprocess_tiles(z["tiles"])) - z["image"]).max() < 2e-6


def test_preprocess_and_patchify_shapes():
    s = C.TINY_CLIP
    t = np.random.default_rng(0).integers(0, 256, (2, s.image, s.image, 3), dtype=np.uint8)
    px = C.preprocess_tiles(t)
    assert px.shape == (2, 3, s.image, s.image)
    p = C.patchify(s, px)
    assert p.shape == (2, s.n_patches, 3 * s.patch * s.patch)
    assert p[1, 3, 2 * 1024 + 5 * 32 + 7] == px[1, 2, 32 + 5, 32 + 7]   # patch 3 = grid (1,1), (c=2, ph=5, pw=7)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny", "vitb32"])
def test_clip_towers_on_gpu(golden_dir, name):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from multimodal_rag_amd.clip import ClipConfig, DeviceClip

    s = SHAPES[name]
    z, seqs = load(golden_dir, name)
    w = C.make_clip_weights(s, int(z["seed"]))
    cfg = ClipConfig(s.t_layers, s.t_hidden, s.t_heads, s.t_inter, s.vocab, s.t_max_pos, s.eos_id, s.v_layers,
                     s.v_hidden, s.v_heads, s.v_inter, s.image, s.patch, s.proj, s.ln_eps)
    clip = DeviceClip(cfg, w, "cuda:0")
    w16 = C.round_weights_fp16(w)

    got_t = clip.encode_text_ids(seqs).cpu().numpy()
    ref_t = C.text_embed(s, w16, seqs)
    assert np.abs(got_t - ref_t).max() <= 5e-3 and (got_t * ref_t).sum(1).min() >= 0.9999, np.abs(got_t - ref_t).max()
    assert (got_t * z["text"]).sum(1).min() >= 0.999              # vs float32 transformers output

    tiles = torch.from_numpy(z["tiles"])
    got_u8 = clip.encode_images(tiles.cuda()).cpu().numpy()        # fused uint8 preprocessing path
    px = C.preprocess_tiles(z["tiles"])
    px16 = px.astype(np.float16)
    ref_v = C.image_embed(s, w16, px16.astype(np.float32))
    got_f16 = clip.encode_images(torch.from_numpy(px16).cuda()).cpu().numpy()
    for got in (got_u8, got_f16):
        assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-4)
        assert np.abs(got - ref_v).max() <= 5e-3 and (got * ref_v).sum(1).min() >= 0.9999, np.abs(got - ref_v).max()
    assert (got_u8 * z["image"]).sum(1).min() >= 0.999


@pytest.mark.gpu
def test_joint_space_index(golden_dir):
    """texts and images land in one index; search works across modalities (config 4 plumbing)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from multimodal_rag_amd.clip import ClipConfig, DeviceClip
    from multimodal_rag_amd.index import VectorIndex
    from oracle import search_oracle as O

    s = C.TINY_CLIP
    cfg = ClipConfig(s.t_layers, s.t_hidden, s.t_heads, s.t_inter, s.vocab, s.t_max_pos, s.eos_id, s.v_layers,
                     s.v_hidden, s.v_heads, s.v_inter, s.image, s.patch, s.proj, s.ln_eps)
    clip = DeviceClip.random_init(cfg, seed=5)
    g = np.random.default_rng(0)
    seqs = [[s.eos_id - 1] + g.integers(1, 900, int(n)).tolist() + [s.eos_id] for n in g.integers(2, 28, 40)]
    tiles = torch.from_numpy(g.integers(0, 256, (24, s.image, s.image, 3), dtype=np.uint8)).cuda()
    te, ie = clip.encode_text_ids(seqs), clip.encode_images(tiles)
    idx = VectorIndex(cfg.proj, dtype=torch.float16)
    idx.add(te, None, [{"type": "text"}] * 40, [f"doc_aaaaaaaaaaaa_text_{i}" for i in range(40)])
    idx.add(ie, None, [{"type": "image"}] * 24, [f"doc_aaaaaaaaaaaa_image_{i}" for i in range(24)])
    res = idx.query(ie[:3], n_results=5)
    assert [r[0] for r in res["ids"]] == [f"doc_aaaaaaaaaaaa_image_{i}" for i in range(3)]
    stored = np.asarray(idx.get(include=["embeddings"])["embeddings"], np.float32)
    es, er = O.cosine_topk(ie[:3].cpu().numpy().astype(np.float16).astype(np.float32), stored, 5)
    assert O.same_topk_sets(np.array([[idx._row_of[i] for i in r] for r in res["ids"]]), 1 - np.array(res["distances"]), er, es)
    only_text = idx.query(ie[:3], n_results=5, where={"type": "text"})
    assert all(i.split("_")[2] == "text" for r in only_text["ids"] for i in r)
