"""TEST INFRASTRUCTURE ONLY (CPU oracle).

Restates the AutoencoderKL decode path:
  backend/nn/vae.py:305-316 (decode, process_out), :248-271 (Decoder.forward), :77-137 (ResnetBlock,
  AttnBlock), :35-57 (Upsample), backend/patcher/vae.py:128-148 (clamp((y+1)/2) -> NHWC fp32 in [0,1]),
  backend/diffusion_engine/sd15.py:80-84 (decode_first_stage: process_out, decode, *2-1, NCHW),
  modules/processing.py:1012-1040 (clamp, *255, astype(uint8) truncation).
Structure is discovered from the checkpoint keys (decoder.up.{level}.block.{i}, .upsample.conv).
"""
import torch
import torch.nn.functional as F

from .attention import attention_single_head_spatial


def _gn(sd, key, x):
    return F.group_norm(x, 32, sd[key + ".weight"], sd[key + ".bias"], 1e-6)  # vae.py:12-13


def _conv(sd, key, x, padding=1):
    return F.conv2d(x, sd[key + ".weight"], sd[key + ".bias"], padding=padding)


def _resnet(sd, key, x):
    h = _conv(sd, key + ".conv1", F.silu(_gn(sd, key + ".norm1", x)))
    h = _conv(sd, key + ".conv2", F.silu(_gn(sd, key + ".norm2", h)))
    if key + ".nin_shortcut.weight" in sd:
        x = _conv(sd, key + ".nin_shortcut", x, padding=0)
    return x + h


def _attn(sd, key, x):
    h = _gn(sd, key + ".norm", x)
    q = _conv(sd, key + ".q", h, 0)
    k = _conv(sd, key + ".k", h, 0)
    v = _conv(sd, key + ".v", h, 0)
    return x + _conv(sd, key + ".proj_out", attention_single_head_spatial(q, k, v), 0)


@torch.no_grad()
def vae_decode(sd, z):
    """IntegratedAutoencoderKL.decode: z [B,4,h,w] (already /scaling_factor) -> [B,3,8h,8w]."""
    if "post_quant_conv.weight" in sd:
        z = _conv(sd, "post_quant_conv", z, 0)
    h = _conv(sd, "decoder.conv_in", z)
    h = _resnet(sd, "decoder.mid.block_1", h)
    h = _attn(sd, "decoder.mid.attn_1", h)
    h = _resnet(sd, "decoder.mid.block_2", h)
    nlev = 0
    while f"decoder.up.{nlev}.block.0.norm1.weight" in sd:
        nlev += 1
    for lev in reversed(range(nlev)):
        i = 0
        while f"decoder.up.{lev}.block.{i}.norm1.weight" in sd:
            h = _resnet(sd, f"decoder.up.{lev}.block.{i}", h)
            i += 1
        if lev != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, f"decoder.up.{lev}.upsample.conv", h)
    h = _conv(sd, "decoder.conv_out", F.silu(_gn(sd, "decoder.norm_out", h)))
    return h


@torch.no_grad()
def vae_encode_moments(sd, x):
    """Encoder.forward (vae.py:183-200) + quant_conv (:297-298): x [B,3,H,W] in [-1,1] -> moments [B, 2*lc, H/f, W/f]."""
    h = _conv(sd, "encoder.conv_in", x)
    nlev = 0
    while f"encoder.down.{nlev}.block.0.norm1.weight" in sd:
        nlev += 1
    for lev in range(nlev):
        i = 0
        while f"encoder.down.{lev}.block.{i}.norm1.weight" in sd:
            h = _resnet(sd, f"encoder.down.{lev}.block.{i}", h)
            i += 1
        if lev != nlev - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)  # Downsample (vae.py:67-70): pad right/bottom, stride 2
            h = F.conv2d(h, sd[f"encoder.down.{lev}.downsample.conv.weight"], sd[f"encoder.down.{lev}.downsample.conv.bias"], stride=2)
    h = _resnet(sd, "encoder.mid.block_1", h)
    h = _attn(sd, "encoder.mid.attn_1", h)
    h = _resnet(sd, "encoder.mid.block_2", h)
    h = _conv(sd, "encoder.conv_out", F.silu(_gn(sd, "encoder.norm_out", h)))
    if "quant_conv.weight" in sd:
        h = _conv(sd, "quant_conv", h, 0)
    return h


def posterior_sample(moments, noise):
    """DiagonalGaussianDistribution (vae.py:16-29): mean + exp(0.5 * clamp(logvar, -30, 20)) * noise."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    return mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * noise


def encode_first_stage(sd, x, scaling_factor, shift_factor=0.0, noise=None):
    """diffusion_engine/sd15.py:75-78 + patcher/vae.py:172-174: x [B,3,H,W] in [-1,1] -> process_in(sample)."""
    m = vae_encode_moments(sd, x.float())
    if noise is None:
        noise = torch.randn(m.shape[0], m.shape[1] // 2, m.shape[2], m.shape[3])  # vae.py:28: CPU default generator
    return (posterior_sample(m, noise) - shift_factor) * scaling_factor  # process_in (vae.py:312-313)


def process_out(latent, scaling_factor, shift_factor=0.0):
    return latent / scaling_factor + shift_factor  # vae.py:315


def vae_decode_inner(sd, samples_in):
    """patcher/vae.py:128-148: -> [B,8h,8w,3] fp32 in [0,1]."""
    y = vae_decode(sd, samples_in.float())
    return torch.clamp((y + 1.0) / 2.0, 0.0, 1.0).movedim(1, -1)


def decode_first_stage(sd, x, scaling_factor, shift_factor=0.0):
    """diffusion_engine/sd15.py:80-84: -> [B,3,8h,8w] in [-1,1]."""
    sample = vae_decode_inner(sd, process_out(x, scaling_factor, shift_factor)) * 2.0 - 1.0
    return sample.movedim(-1, 1)


def to_uint8_images(decoded):
    """processing.py:1012-1013,1039-1040: clamp((x+1)/2), *255, NHWC, astype(uint8) (truncation)."""
    x = torch.clamp((decoded.float() + 1.0) / 2.0, 0.0, 1.0)
    x = 255.0 * x.movedim(1, -1)
    return x.numpy().astype("uint8")
