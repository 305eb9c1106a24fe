"""`adaptive_resize` -- mirror of backend/misc/image_resize.py:92-113 for the torch.nn.functional.interpolate modes (what ControlNet uses
for its hint: 'nearest-exact' + centre crop, patcher/controlnet.py:303); bislerp / lanczos are image-space upscalers outside the path."""
from ...modules import latent_upscale


def adaptive_resize(samples, width, height, upscale_method, crop):
    if crop == "center":
        old_width, old_height = samples.shape[3], samples.shape[2]
        old_aspect, new_aspect = old_width / old_height, width / height
        x = y = 0
        if old_aspect > new_aspect:
            x = round((old_width - old_width * (new_aspect / old_aspect)) / 2)
        elif old_aspect < new_aspect:
            y = round((old_height - old_height * (old_aspect / new_aspect)) / 2)
        s = samples[:, :, y:old_height - y, x:old_width - x]
    else:
        s = samples
    if upscale_method in ("bislerp", "lanczos"):
        raise NotImplementedError(f"adaptive_resize: '{upscale_method}' is not on the native path")
    if tuple(s.shape[2:]) == (height, width):
        return s.contiguous()
    return latent_upscale.interpolate(s.float().contiguous(), (height, width), mode=upscale_method).to(samples.dtype)
