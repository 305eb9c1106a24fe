"""Model configs of the BASELINE.json workloads and deterministic synthetic ("random-init") weights.

There are no checkpoints in this environment, so every BASELINE config runs on random-init weights
(SURVEY.md §8d).  The weights are a pure function of (parameter name, shape, seed) so that the CPU
oracle, the real reference (when present) and the MI355X executor can be handed bit-identical fp32
state dicts anywhere, without relying on module construction order.

Init law (documented gain-scaled variant of torch's default):
  * conv / linear weight  ~ N(0, gain^2 / fan_in),   gain = 1/sqrt(3)  (variance of torch's default
    kaiming_uniform(a=sqrt(5)) = U(-1/sqrt(fan_in), 1/sqrt(fan_in)))
  * conv / linear bias    ~ N(0, gain^2 / fan_in)
  * norm weight = 1 + 0.1*N(0,1), norm bias = 0.1*N(0,1)   (so the affine path is exercised)
numpy's PCG64 + ziggurat stream is platform-independent, unlike vectorised torch CPU randn.
"""
import zlib
from collections import OrderedDict

import numpy as np
import torch

from .backend.nn.layout import clip_param_shapes, flux_param_shapes, unet_param_shapes, vae_decoder_param_shapes, vae_encoder_param_shapes

# LDM-style unet_config dicts (parameter counts 859.52 M / 2567.46 M verified against
# the reference module's own parameter shapes, tests/golden/param_shapes.json, in tests/test_oracle_golden.py).
SD15_UNET_CONFIG = dict(
    in_channels=4, model_channels=320, out_channels=4, num_res_blocks=[2, 2, 2, 2], channel_mult=(1, 2, 4, 4),
    num_heads=8, use_spatial_transformer=True, transformer_depth=[1, 1, 1, 1, 1, 1, 0, 0],
    transformer_depth_middle=1, transformer_depth_output=[1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0],
    context_dim=768, use_linear_in_transformer=False)

SDXL_UNET_CONFIG = dict(
    in_channels=4, model_channels=320, out_channels=4, num_res_blocks=[2, 2, 2], channel_mult=(1, 2, 4),
    num_head_channels=64, use_spatial_transformer=True, transformer_depth=[0, 0, 2, 2, 10, 10],
    transformer_depth_middle=10, transformer_depth_output=[0, 0, 0, 2, 2, 2, 10, 10, 10],
    context_dim=2048, use_linear_in_transformer=True, adm_in_channels=2816, num_classes="sequential")

# Small configs with the same block grammar (used by the parity tests; finish in seconds on CPU).
TINY_SD15_UNET_CONFIG = dict(
    in_channels=4, model_channels=64, out_channels=4, num_res_blocks=[1, 1, 1], channel_mult=(1, 2, 2),
    num_heads=4, use_spatial_transformer=True, transformer_depth=[1, 1, 0],
    transformer_depth_middle=1, transformer_depth_output=[1, 1, 1, 1, 0, 0],
    context_dim=128, use_linear_in_transformer=False)
# transformer_depth_output is consumed with pop() from the END (unet.py:649): listed low-res-last.

TINY_SD15_INPAINT_UNET_CONFIG = dict(TINY_SD15_UNET_CONFIG, in_channels=9)  # 4 latent + 1 mask + 4 masked-image latent (sd-v1-5-inpainting)

TINY_SDXL_UNET_CONFIG = dict(
    in_channels=4, model_channels=64, out_channels=4, num_res_blocks=[1, 1], channel_mult=(1, 2),
    num_head_channels=64, use_spatial_transformer=True, transformer_depth=[0, 2],
    transformer_depth_middle=2, transformer_depth_output=[0, 0, 2, 2],
    context_dim=128, use_linear_in_transformer=True, adm_in_channels=192, num_classes="sequential")

SD15_VAE_CONFIG = dict(in_channels=3, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                       latent_channels=4, scaling_factor=0.18215, shift_factor=0.0,
                       use_quant_conv=True, use_post_quant_conv=True)
SDXL_VAE_CONFIG = dict(SD15_VAE_CONFIG, scaling_factor=0.13025)
TINY_VAE_CONFIG = dict(in_channels=3, out_channels=3, block_out_channels=(64, 128), layers_per_block=1,
                       latent_channels=4, scaling_factor=0.18215, shift_factor=0.0,
                       use_quant_conv=True, use_post_quant_conv=True)

# Flux.1-dev VAE (backend/huggingface/black-forest-labs/FLUX.1-dev/vae/config.json): 16 latent channels, no quant convs, shift factor
FLUX_VAE_CONFIG = dict(in_channels=3, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=16,
                       scaling_factor=0.3611, shift_factor=0.1159, use_quant_conv=False, use_post_quant_conv=False)
TINY_FLUX_VAE_CONFIG = dict(TINY_VAE_CONFIG, latent_channels=16, scaling_factor=0.3611, shift_factor=0.1159, use_quant_conv=False,
                            use_post_quant_conv=False)

# Flux.1-dev (backend/huggingface/black-forest-labs/FLUX.1-dev/transformer/config.json) and a tiny twin with the
# same head_dim 128 / axes_dim split (the RoPE table and attention tile shapes are the real ones)
FLUX_DEV_CONFIG = dict(in_channels=16, vec_in_dim=768, context_in_dim=4096, hidden_size=3072, mlp_ratio=4.0, num_heads=24, depth=19,
                       depth_single_blocks=38, axes_dim=[16, 56, 56], theta=10000, qkv_bias=True, guidance_embed=True)
TINY_FLUX_CONFIG = dict(in_channels=16, vec_in_dim=64, context_in_dim=128, hidden_size=256, mlp_ratio=4.0, num_heads=2, depth=2,
                        depth_single_blocks=2, axes_dim=[16, 56, 56], theta=10000, qkv_bias=True, guidance_embed=True)

SCHEDULE = dict(beta_schedule="linear", linear_start=0.00085, linear_end=0.012, timesteps=1000,
                prediction_type="epsilon")

DEFAULT_GAIN = 1.0 / np.sqrt(3.0)


def _is_norm(name, shape):
    if len(shape) != 1:
        return False
    parts = name.split(".")
    return any(p.startswith("norm") for p in parts) or parts[-2] == "0" and (
        "in_layers" in parts or "out_layers" in parts or parts[0] == "out")


def synth_tensor(name, shape, seed=0, gain=DEFAULT_GAIN):
    rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
    z = rng.standard_normal(shape, dtype=np.float32)
    if _is_norm(name, shape):
        if name.endswith((".weight", ".scale")):
            z = 1.0 + 0.1 * z
        else:
            z = 0.1 * z
    else:
        if name.endswith(".bias"):
            # the sibling weight's fan_in is not known from the bias alone; callers pass it via shape hints
            raise ValueError("bias needs fan_in; use synth_state_dict")
        fan_in = int(np.prod(shape[1:]))
        z *= gain / np.sqrt(fan_in)
    return torch.from_numpy(np.ascontiguousarray(z, dtype=np.float32))


_BRANCH_OUT = (".out_layers.3.", ".to_out.0.", ".ff.net.2.", ".proj_out.", ".conv2.")


def synth_entry(shapes, name, seed=0, gain=DEFAULT_GAIN, residual_gain=1.0):
    """one tensor of `synth_state_dict(shapes, ...)`: a pure function of (name, shape, seed) -- and of the sibling weight's shape for a bias"""
    shape = shapes[name]
    if _is_norm(name, shape):
        return synth_tensor(name, shape, seed)
    if name.endswith(".bias"):
        wshape = shapes[name[:-5] + ".weight"]
        fan_in = int(np.prod(wshape[1:]))
        rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
        z = rng.standard_normal(shape, dtype=np.float32) * (gain / np.sqrt(fan_in))
        t = torch.from_numpy(z.astype(np.float32))
    else:
        t = synth_tensor(name, shape, seed, gain)
    if residual_gain != 1.0 and any(b in name for b in _BRANCH_OUT):
        t = t * residual_gain
    return t


def synth_state_dict(shapes, seed=0, gain=DEFAULT_GAIN, residual_gain=1.0):
    """fp32 CPU state dict for `shapes` (name -> shape).  `residual_gain` scales the last projection of
    every residual branch (documented knob to keep random-init nets well conditioned; 1.0 = off)."""
    return OrderedDict((name, synth_entry(shapes, name, seed, gain, residual_gain)) for name in shapes)


class LazySynthStateDict:
    """`synth_state_dict(shapes, seed)` as a read-only mapping that draws a tensor when it is asked for and keeps nothing: a 12 G-parameter
    network (Flux.1 at full depth: 48 GB as fp32) is loaded tensor by tensor without the state dict ever existing in host memory.  `dtype`:
    cast on the way out (round to nearest, what `module.to(dtype)` does to the fp32 tensor)."""

    def __init__(self, shapes, seed=0, dtype=None, **kw):
        self.shapes, self.seed, self.dtype, self.kw = shapes, seed, dtype, kw

    def __getitem__(self, name):
        t = synth_entry(self.shapes, name, self.seed, **self.kw)
        return t if self.dtype is None else t.to(self.dtype)

    def __contains__(self, name):
        return name in self.shapes

    def __iter__(self):
        return iter(self.shapes)

    def __len__(self):
        return len(self.shapes)

    def keys(self):
        return self.shapes.keys()

    def items(self):
        return ((k, self[k]) for k in self.shapes)


def synth_state_dict_threaded(shapes, seed=0, dtype=None, workers=None, **kw):
    """`synth_state_dict` drawn by a thread pool (numpy's generators release the GIL; every tensor is a pure function of its name, so the result does
    not depend on the schedule), each tensor cast to `dtype` as soon as it exists: Flux.1 at full depth is 24 GB of fp16 / bf16 in well under a minute
    on a many-core host instead of five minutes on one core, and the fp32 form never exists as a whole."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    workers = workers or max(1, min(32, (os.cpu_count() or 2) - 1))

    def draw(name):
        t = synth_entry(shapes, name, seed, **kw)
        return t if dtype is None else t.to(dtype)

    names = list(shapes)
    with ThreadPoolExecutor(workers) as pool:
        return OrderedDict(zip(names, pool.map(draw, names)))


def synth_unet_state_dict(cfg, seed=0, **kw):
    return synth_state_dict(unet_param_shapes(cfg), seed=seed, **kw)


def synth_controlnet_state_dict(cfg, hint_channels=3, seed=6, zero_conv_gain=1.0, **kw):
    """cldm.ControlNet weights.  Real checkpoints start from ZERO 1x1 output convs; random ones (same init as every other conv) make the
    residuals non-trivial for parity tests."""
    from .backend.nn.layout import controlnet_param_shapes
    return synth_state_dict(controlnet_param_shapes(cfg, hint_channels), seed=seed, **kw)


def synth_control_lora_state_dict(cfg, hint_channels=3, rank=4, seed=13, delta_gain=0.5, skip_prefixes=("time_embed.",)):
    """A Control-LoRA file (patcher/controlnet.py:360-457): for the trunk's conv / linear weights a low-rank pair `<module>.up` / `<module>.down`
    that is ADDED to the UNet's own weight, everything else (norms, biases, hint block, zero convs) stored directly, plus the `lora_controlnet`
    marker.  Modules under `skip_prefixes` get no entry at all: they run on the UNet's weights unchanged."""
    from .backend.nn.layout import controlnet_param_shapes
    shapes = controlnet_param_shapes(cfg, hint_channels)
    direct = synth_state_dict(shapes, seed=seed)
    trunk = ("input_blocks.", "middle_block.", "time_embed.", "label_emb.")
    sd = OrderedDict()
    for name, shape in shapes.items():
        if name.startswith(skip_prefixes):
            continue
        if name.startswith(trunk) and name.endswith(".weight") and len(shape) >= 2:
            base = name[:-7]
            up_shape = (shape[0], rank) + (1,) * (len(shape) - 2)
            # up ~ N(0, 1/rank), down ~ N(0, (delta_gain * gain)^2 / fan_in)  =>  std(up @ down) = delta_gain * std(a synthetic weight)
            sd[base + ".up"] = synth_tensor(base + ".up", up_shape, seed, 1.0)
            sd[base + ".down"] = synth_tensor(base + ".down", (rank,) + tuple(shape[1:]), seed, delta_gain * DEFAULT_GAIN)
        else:
            sd[name] = direct[name]
    sd["lora_controlnet"] = torch.zeros(1)
    return sd


# an SD1.5-SHAPED small UNet (4 levels x 2 ResBlocks = 12 input blocks): T2I-Adapter features are placed by input-block index, so its tests
# need the real block grammar; channels 64 / 128 / 256 / 256
MINI_SD15_UNET_CONFIG = dict(
    in_channels=4, model_channels=64, out_channels=4, num_res_blocks=[2, 2, 2, 2], channel_mult=(1, 2, 4, 4), num_heads=4,
    use_spatial_transformer=True, transformer_depth=[1, 1, 1, 1, 1, 1, 0, 0], transformer_depth_middle=1,
    transformer_depth_output=[1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0], context_dim=128, use_linear_in_transformer=False)


def synth_t2i_adapter_state_dict(seed=11, **adapter_kw):
    from .backend.nn.cnets.t2i_adapter import adapter_param_shapes
    return synth_state_dict(OrderedDict(adapter_param_shapes(**adapter_kw)), seed=seed)


def synth_t2i_adapter_light_state_dict(seed=12, **adapter_kw):
    from .backend.nn.cnets.t2i_adapter import adapter_light_param_shapes
    return synth_state_dict(OrderedDict(adapter_light_param_shapes(**adapter_kw)), seed=seed)


def synth_flux_state_dict(cfg, seed=2, **kw):
    return synth_state_dict(flux_param_shapes(cfg), seed=seed, **kw)


def synth_vae_decoder_state_dict(cfg, seed=1, **kw):
    return synth_state_dict(vae_decoder_param_shapes(cfg), seed=seed, **kw)


def synth_vae_state_dict(cfg, seed=1, **kw):
    """decoder + encoder + quant convs; the decoder tensors equal synth_vae_decoder_state_dict's (pure function of the name)"""
    shapes = OrderedDict(vae_decoder_param_shapes(cfg))
    shapes.update(vae_encoder_param_shapes(cfg))
    return synth_state_dict(shapes, seed=seed, **kw)


# CLIP-L (SD1.x / SDXL first encoder) and OpenCLIP bigG (SDXL second encoder) text-model configs, and tiny twins with the
# real head width (64) for the parity tests
CLIP_L_CONFIG = dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12, vocab_size=49408,
                     max_position_embeddings=77, hidden_act="quick_gelu")
CLIP_G_CONFIG = dict(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=20, vocab_size=49408,
                     max_position_embeddings=77, hidden_act="gelu", add_text_projection=True)
TINY_CLIP_L_CONFIG = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2, vocab_size=1000,
                          max_position_embeddings=77, hidden_act="quick_gelu")
TINY_CLIP_G_CONFIG = dict(hidden_size=192, intermediate_size=320, num_hidden_layers=4, num_attention_heads=3, vocab_size=1000,
                          max_position_embeddings=77, hidden_act="gelu", add_text_projection=True)


def synth_clip_state_dict(cfg, seed=4, **kw):
    return synth_state_dict(clip_param_shapes(cfg), seed=seed, **kw)


# T5-XXL as Flux uses it (backend/huggingface/black-forest-labs/FLUX.1-dev/text_encoder_2/config.json): 24 layers, 64 heads of 64, gated tanh-GELU
T5_XXL_CONFIG = dict(d_model=4096, d_ff=10240, num_layers=24, num_heads=64, vocab_size=32128, dense_act_fn="gelu_pytorch_tanh", is_gated_act=True, model_type="t5")
TINY_T5_CONFIG = dict(d_model=128, d_ff=320, num_layers=3, num_heads=2, vocab_size=1000, dense_act_fn="gelu_pytorch_tanh", is_gated_act=True, model_type="t5")


def synth_t5_state_dict(cfg, seed=9):
    """random-init T5 encoder: N(0, 1) / sqrt(fan_in) projections, RMS-norm weights 1 + 0.1 N, token embeddings N(0, 1), bias table N(0, 0.5)"""
    from .backend.nn.layout import t5_param_shapes
    sd = OrderedDict()
    for name, shape in t5_param_shapes(cfg).items():
        rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
        if name == "logit_scale":
            sd[name] = torch.tensor(4.6055)
            continue
        z = torch.from_numpy(rng.standard_normal(shape, dtype=np.float32))
        if name.endswith("layer_norm.weight"):
            z = 1.0 + 0.1 * z
        elif name.endswith("relative_attention_bias.weight"):
            z = 0.5 * z
        elif name != "transformer.shared.weight":
            z = z / float(np.sqrt(shape[1]))
        sd[name] = z
    return sd


def synth_conditioning(batch, context_dim, adm_in_channels=None, tokens=77, seed=1234):
    """cond / uncond tensors of the shapes the text encoders would produce."""
    rng = np.random.Generator(np.random.PCG64([seed, 7]))
    c = torch.from_numpy(rng.standard_normal((batch, tokens, context_dim), dtype=np.float32))
    uc = torch.from_numpy(rng.standard_normal((batch, tokens, context_dim), dtype=np.float32))
    if adm_in_channels is None:
        return c, uc
    y = torch.from_numpy(rng.standard_normal((batch, adm_in_channels), dtype=np.float32))
    uy = torch.from_numpy(rng.standard_normal((batch, adm_in_channels), dtype=np.float32))
    return {"crossattn": c, "vector": y}, {"crossattn": uc, "vector": uy}


def synth_state_dict_device(shapes, seed, device, dtype=torch.float16, gain=DEFAULT_GAIN):
    """Same init law as `synth_state_dict`, drawn directly on the device (bench.py: 2.6 G parameters in seconds).
    Values differ from the numpy stream, so this is for throughput runs, not for parity fixtures."""
    g = torch.Generator(device=device).manual_seed(int(seed))
    sd = OrderedDict()
    for name, shape in shapes.items():
        z = torch.randn(shape, generator=g, device=device, dtype=torch.float32)
        if _is_norm(name, shape):
            z = 1.0 + 0.1 * z if name.endswith((".weight", ".scale")) else 0.1 * z
        else:
            wshape = shapes[name[:-5] + ".weight"] if name.endswith(".bias") else shape
            z = z * (gain / float(np.sqrt(int(np.prod(wshape[1:])))))
        sd[name] = z.to(dtype)
    return sd
