"""Image-space mask helpers of the img2img path -- mirror of modules/masking.py:80-95 (`fill`).

Host-side pre-processing of the init image (once per job, before the VAE encoder): no kernel in it.  The reference does it with PIL on the uint8 image;
so does this, on the same PIL calls, for tensor inputs."""
import numpy as np
import torch


def fill(image, mask):
    """masking.py:80-95: fill the masked region of a PIL image with colours bled in from its surroundings (six premultiplied-alpha Gaussian blurs,
    composited coarse to fine).  image: PIL RGB, mask: PIL (white = repaint) -> PIL RGB"""
    from PIL import Image, ImageFilter, ImageOps
    image_mod = Image.new("RGBA", (image.width, image.height))
    image_masked = Image.new("RGBa", (image.width, image.height))
    image_masked.paste(image.convert("RGBA").convert("RGBa"), mask=ImageOps.invert(mask.convert("L")))
    image_masked = image_masked.convert("RGBa")
    for radius, repeats in [(256, 1), (64, 1), (16, 2), (4, 4), (2, 2), (0, 1)]:
        blurred = image_masked.filter(ImageFilter.GaussianBlur(radius)).convert("RGBA")
        for _ in range(repeats):
            image_mod.alpha_composite(blurred)
    return image_mod.convert("RGB")


def fill_tensor(images, mask):
    """`fill` for the tensor form of the job: images [B, 3, H, W] in [0, 1], mask [1 or B, 1, H, W] (or [.., H, W]) with 1 = repaint, any float / bool
    dtype, any device -> filled images, same shape / dtype / device.  Quantises to the uint8 image the reference works on (processing.py:1781-1788 runs
    `fill` on the PIL image before it becomes a float array)."""
    from PIL import Image
    b = images.shape[0]
    # one channel per image: a latent mask may arrive as [B or 1, lc, h, w] (every channel the same), and flattening that to B * lc rows would
    # hand image i channel i of image 0
    m = (mask[:, 0] if mask.dim() == 4 else mask.reshape(-1, mask.shape[-2], mask.shape[-1])).float().cpu()
    if m.shape[0] not in (1, b):
        raise ValueError(f"fill_tensor: a mask of {m.shape[0]} rows for {b} images (1 or {b} expected)")
    out = []
    for i in range(b):
        arr = (images[i].detach().float().cpu().clamp(0, 1) * 255.0).round().to(torch.uint8).permute(1, 2, 0).numpy()
        mi = m[i if m.shape[0] > 1 else 0]
        if mi.shape != arr.shape[:2]:   # a latent-resolution mask: nearest upsample to the image grid
            mi = torch.nn.functional.interpolate(mi[None, None], size=arr.shape[:2], mode="nearest")[0, 0]
        pm = Image.fromarray((mi.clamp(0, 1) * 255.0).round().to(torch.uint8).numpy(), mode="L")
        filled = fill(Image.fromarray(arr, mode="RGB"), pm)
        out.append(torch.from_numpy(np.array(filled).astype(np.float32) / 255.0).permute(2, 0, 1))
    return torch.stack(out).to(device=images.device, dtype=images.dtype)
