"""Golden outputs for the CLIP image front end (resize shortest edge -> 224, bicubic, centre crop; uint8).

Expected arrays come from PIL.Image.resize(BICUBIC) + crop -- the routine transformers' CLIP processor
runs -- and the script also checks that `CLIPImageProcessor` gives the same pixels after normalisation.
Inputs are formula images (no RNG) rebuilt by the tests, plus one real page region: a 700x520 crop of the
reference's figures/Session01_Khai_niem_co_ban_C_page_0_ab7570e1.png (data file, stored as PNG).
Run in the build container:  python tests/golden/make_resize_golden.py
"""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import clip_oracle as C  # noqa: E402

SIZES = [(97, 161), (300, 200), (100, 80), (224, 224), (500, 224), (480, 640), (61, 1000)]


def formula_image(H, W):
    y, x = np.mgrid[0:H, 0:W]
    chans = [(x * 7 + y * 13 + (x * y) % 251) % 256, (x * x // 3 + y * 5) % 256, ((x ^ y) * 9 + 17) % 256]
    return np.stack(chans, axis=-1).astype(np.uint8)


def pil_resize_crop(img):
    nh, nw, top, left = C.clip_resize_geometry(img.shape[0], img.shape[1])
    r = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BICUBIC))
    return np.ascontiguousarray(r[top:top + 224, left:left + 224])


def main():
    out = {}
    for H, W in SIZES:
        out[f"f_{H}x{W}"] = pil_resize_crop(formula_image(H, W))
    fig = "/root/reference/figures/Session01_Khai_niem_co_ban_C_page_0_ab7570e1.png"
    page = np.asarray(Image.open(fig).convert("RGB"))
    crop = np.ascontiguousarray(page[300:1000, 400:920])
    Image.fromarray(crop).save(os.path.join(HERE, "page_region.png"), optimize=True)
    out["page_region"] = pil_resize_crop(crop)
    np.savez_compressed(os.path.join(HERE, "resize_expected.npz"), **out)

    from transformers import CLIPImageProcessor

    proc = CLIPImageProcessor()
    for H, W in SIZES[:3]:
        img = formula_image(H, W)
        want = proc(images=Image.fromarray(img), return_tensors="np")["pixel_values"][0]
        got = C.preprocess_tiles(out[f"f_{H}x{W}"][None])[0]
        assert np.abs(want - got).max() < 1e-6, (H, W)
    print({k: v.shape for k, v in out.items()}, "page", page.shape)


if __name__ == "__main__":
    main()
