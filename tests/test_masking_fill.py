"""`forge_amd.modules.masking.fill` against the reference's modules/masking.py:80-95 (imported by path when /root/reference is present) and against a
committed fixture generated from it (tests/golden/masking_fill.pt, made by this file: `python tests/test_masking_fill.py`)."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import forge_amd  # noqa: E402,F401
from forge_amd.modules import masking  # noqa: E402

REF = "/root/reference/modules/masking.py"
FIX = os.path.join(HERE, "golden", "masking_fill.pt")


def _case():
    g = torch.Generator().manual_seed(7)
    img = (torch.rand(2, 3, 48, 64, generator=g) * 255).round() / 255.0
    mask = torch.zeros(1, 1, 48, 64)
    mask[..., 10:30, 20:50] = 1.0
    return img, mask


def _ref_fill():
    spec = importlib.util.spec_from_file_location("ref_masking", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.fill


def _pil(img, mask):
    from PIL import Image
    a = (img * 255).round().to(torch.uint8).permute(1, 2, 0).numpy()
    m = (mask[0, 0] * 255).round().to(torch.uint8).numpy()
    return Image.fromarray(a, "RGB"), Image.fromarray(m, "L")


@pytest.mark.skipif(not os.path.exists(REF), reason="needs /root/reference")
def test_fill_equals_the_reference_function():
    img, mask = _case()
    ref_fill = _ref_fill()
    for i in range(img.shape[0]):
        pi, pm = _pil(img[i], mask)
        assert np.array_equal(np.array(masking.fill(pi, pm)), np.array(ref_fill(pi, pm)))


def test_fill_tensor_matches_the_reference_fixture():
    img, mask = _case()
    fx = torch.load(FIX)
    out = masking.fill_tensor(img, mask)
    assert out.shape == img.shape and out.dtype == img.dtype
    assert torch.equal((out * 255).round().to(torch.uint8), fx["filled_u8"])
    keep = (mask == 0).expand_as(img)
    assert torch.equal(out[keep], img[keep]), "pixels outside the mask are the image's own"
    # a latent-resolution mask (8x coarser) is upsampled to the image grid
    out2 = masking.fill_tensor(img, torch.nn.functional.max_pool2d(mask, 8))
    assert out2.shape == img.shape


def test_fill_tensor_takes_one_mask_channel_per_image():
    """a latent mask may be [B or 1, lc, h, w] (processing.py's `latent_mask`): per IMAGE masks with several channels must not be read as B * lc rows"""
    img, mask = _case()
    per_image = torch.cat([mask, torch.zeros_like(mask)], 0)                       # image 0 masked, image 1 not
    want = masking.fill_tensor(img, per_image)
    assert torch.equal(want[1], img[1]) and not torch.equal(want[0], img[0])
    assert torch.equal(masking.fill_tensor(img, per_image.expand(2, 4, -1, -1)), want)
    assert torch.equal(masking.fill_tensor(img, mask.expand(1, 4, -1, -1)), masking.fill_tensor(img, mask))
    assert torch.equal(masking.fill_tensor(img, per_image[:, 0]), want)            # [B, H, W]
    with pytest.raises(ValueError):
        masking.fill_tensor(img, torch.zeros(3, 1, 48, 64))


if __name__ == "__main__":
    img, mask = _case()
    ref_fill = _ref_fill()
    outs = []
    for i in range(img.shape[0]):
        pi, pm = _pil(img[i], mask)
        outs.append(torch.from_numpy(np.array(ref_fill(pi, pm))).permute(2, 0, 1))
    torch.save({"filled_u8": torch.stack(outs), "note": "modules/masking.py:80-95 fill() of the seeded 2 x 3 x 48 x 64 image of tests/test_masking_fill.py::_case"}, FIX)
    print("wrote", FIX)
