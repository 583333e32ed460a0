"""CLIP image front end (resize shortest edge -> 224 bicubic + centre crop, uint8): oracle vs the committed
PIL goldens and the host tap routine on CPU; the HIP passes vs the oracle, bit-exact, on the GPU."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
GOLD = os.path.join(ROOT, "tests", "golden")

from oracle import clip_oracle as C  # noqa: E402
from make_resize_golden import SIZES, formula_image  # noqa: E402  (pure numpy helpers)


@pytest.fixture(scope="module")
def expected():
    return np.load(os.path.join(GOLD, "resize_expected.npz"))


def page_region():
    from PIL import Image

    return np.asarray(Image.open(os.path.join(GOLD, "page_region.png")).convert("RGB"))


def test_oracle_matches_pil_goldens(expected):
    for H, W in SIZES:
        got = C.clip_resize_crop_u8(formula_image(H, W))
        assert np.array_equal(got, expected[f"f_{H}x{W}"]), (H, W)


def test_oracle_matches_pil_golden_on_real_page(expected):
    pytest.importorskip("PIL")
    assert np.array_equal(C.clip_resize_crop_u8(page_region()), expected["page_region"])


def test_oracle_matches_pil_live():
    Image = pytest.importorskip("PIL.Image")
    g = np.random.default_rng(3)
    for H, W in [(130, 77), (77, 130), (1000, 700)]:
        img = g.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
        nh, nw, top, left = C.clip_resize_geometry(H, W)
        pil = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BICUBIC))[top:top + 224, left:left + 224]
        assert np.array_equal(C.clip_resize_crop_u8(img), pil)


def test_host_taps_match_oracle():
    from multimodal_rag_amd import _native as N

    for i, o in [(161, 371), (1654, 224), (80, 224), (2339, 316), (1000, 3672), (225, 224)]:
        eb, et = C.resample_coeffs(i, o)
        b, t = N.resample_coeffs(i, o, 0, o)
        assert np.array_equal(b, eb) and np.array_equal(t, et), (i, o)
        b2, t2 = N.resample_coeffs(i, o, o // 3, o // 2)  # a window, as the cropped passes use
        assert np.array_equal(b2, eb[o // 3: o // 3 + o // 2]) and np.array_equal(t2, et[o // 3: o // 3 + o // 2])
    b, t = N.resample_coeffs(224, 224, 0, 224)  # same size: Pillow skips the pass -> identity taps
    assert np.array_equal(b[:, 0], np.arange(224)) and np.all(b[:, 1] == 1) and np.all(t == 1 << 22)


def test_geometry_matches_processor_rules():
    from multimodal_rag_amd.clip import clip_resize_geometry

    for H, W in SIZES + [(2339, 1654), (1654, 2339), (224, 225), (1, 5)]:
        assert clip_resize_geometry(H, W) == C.clip_resize_geometry(H, W)
    assert C.clip_resize_geometry(2339, 1654) == (316, 224, 46, 0)


@pytest.mark.gpu
def test_hip_resize_bit_exact(expected):
    import torch

    from multimodal_rag_amd.clip import ClipImagePreprocessor

    pre = ClipImagePreprocessor("cuda:0")
    imgs = [formula_image(H, W) for H, W in SIZES] + [page_region()]
    out = pre(imgs).cpu().numpy()
    for i, (H, W) in enumerate(SIZES):
        assert np.array_equal(out[i], expected[f"f_{H}x{W}"]), (H, W)
    assert np.array_equal(out[len(SIZES)], expected["page_region"])
    # full reference page size (1654 x 2339, config 4) and extreme aspect ratios, against the oracle
    g = np.random.default_rng(11)
    for H, W in [(2339, 1654), (300, 4000), (4000, 300), (225, 224), (33, 47)]:
        img = g.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
        got = pre([img])[0].cpu().numpy()
        assert np.array_equal(got, C.clip_resize_crop_u8(img)), (H, W)
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_resized_images_through_the_vision_tower():
    """arbitrary-size images -> HIP resize/crop -> vision tower == oracle resize/crop -> oracle tower"""
    import torch

    from multimodal_rag_amd.clip import ClipConfig, ClipImagePreprocessor, DeviceClip

    s = C.TINY_CLIP
    w = C.round_weights_fp16(C.make_clip_weights(s, seed=5))
    cfg = ClipConfig(**{f: getattr(s, f) for f in ClipConfig.__dataclass_fields__})
    clip = DeviceClip(cfg, w, "cuda:0")
    imgs = [formula_image(90, 140), formula_image(200, 64)]
    pre = ClipImagePreprocessor("cuda:0", size=s.image)
    got = clip.encode_images(pre(imgs)).cpu().numpy()
    tiles = np.stack([C.clip_resize_crop_u8(im, s.image) for im in imgs])
    want = C.image_embed(s, w, C.preprocess_tiles(tiles))
    assert np.abs(got - want).max() < 4e-3
