"""Single-file checkpoint -> native engine: counterpart of backend/loader.py (`preprocess_state_dict` :442,
`split_state_dict` :449, `forge_loader` :498) for the SD1.x / SD2.x / SDXL-base LDM layouts.

The reference delegates model-family detection to the `huggingface_guess` package (git lllyasviel/huggingface_guess@84826248,
`launch_utils.py:397-404`; absent here): `guess(sd)` reads the UNet hyper-parameters off the tensor shapes
(`detection.detect_unet_config`, the ComfyUI model-detection algorithm) and matches them against the known model list.
`detect_unet_config` below restates that published shape-reading algorithm for the LDM UNet; the head layout, which shapes
cannot reveal, follows the model list (context 768 -> SD1.x: 8 heads; 1024 -> SD2.x, 2048 / 1280 -> SDXL: 64 channels per head).
The native executor binds LDM parameter names directly, so no key conversion is needed for the UNet or the (LDM-layout) VAE of
a single-file checkpoint."""
import torch

from .diffusion_engine.base import build_engine

UNET_PREFIX = "model.diffusion_model."
VAE_PREFIX = "first_stage_model."
SD_VAE_CONFIG = dict(in_channels=3, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4,
                     shift_factor=0.0, use_quant_conv=True, use_post_quant_conv=True)


def load_torch_file(path, device="cpu"):
    if isinstance(path, dict):
        return path
    if str(path).endswith((".safetensors", ".sft")):
        from safetensors.torch import load_file
        return load_file(path, device=str(device))
    sd = torch.load(path, map_location=device, weights_only=True)
    return sd.get("state_dict", sd)


def preprocess_state_dict(sd):
    """loader.py:442-446: a bare UNet state dict gets the checkpoint prefix."""
    if not any(k.startswith("model.diffusion_model") for k in sd.keys()):
        sd = {f"model.diffusion_model.{k}": v for k, v in sd.items()}
    return sd


def _count(sd, fmt):
    n = 0
    while fmt.format(n) in sd:
        n += 1
    return n


def detect_unet_config(sd, prefix=UNET_PREFIX):
    """LDM UNet hyper-parameters from tensor shapes (see the module docstring)."""
    g = lambda k: sd[prefix + k]  # noqa: E731
    has = lambda k: (prefix + k) in sd  # noqa: E731
    mc = g("input_blocks.0.0.weight").shape[0]
    cfg = dict(in_channels=g("input_blocks.0.0.weight").shape[1], model_channels=mc, out_channels=g("out.2.weight").shape[0],
               use_spatial_transformer=True)
    if has("label_emb.0.0.weight"):
        cfg["adm_in_channels"] = g("label_emb.0.0.weight").shape[1]
        cfg["num_classes"] = "sequential"
    channel_mult, num_res_blocks, depth_in = [], [], []
    context_dim, use_linear = None, False

    def tdepth(block):
        nonlocal context_dim, use_linear
        d = 0
        while has(f"{block}.1.transformer_blocks.{d}.norm1.weight"):
            d += 1
        if d:
            context_dim = g(f"{block}.1.transformer_blocks.0.attn2.to_k.weight").shape[1]
            use_linear = g(f"{block}.1.proj_in.weight").dim() == 2
        return d

    i, res_in_level = 1, 0
    while has(f"input_blocks.{i}.0.in_layers.0.weight") or has(f"input_blocks.{i}.0.op.weight"):
        if has(f"input_blocks.{i}.0.op.weight"):
            num_res_blocks.append(res_in_level)
            res_in_level = 0
        else:
            out_ch = g(f"input_blocks.{i}.0.out_layers.3.weight").shape[0]
            if res_in_level == 0:
                channel_mult.append(out_ch // mc)
            res_in_level += 1
            depth_in.append(tdepth(f"input_blocks.{i}"))
        i += 1
    num_res_blocks.append(res_in_level)
    cfg["channel_mult"] = tuple(channel_mult)
    cfg["num_res_blocks"] = num_res_blocks
    cfg["transformer_depth"] = depth_in
    cfg["transformer_depth_middle"] = _count(sd, prefix + "middle_block.1.transformer_blocks.{}.norm1.weight") if has("middle_block.1.norm.weight") else -1
    if cfg["transformer_depth_middle"] > 0 and context_dim is None:
        context_dim = g("middle_block.1.transformer_blocks.0.attn2.to_k.weight").shape[1]
        use_linear = g("middle_block.1.proj_in.weight").dim() == 2
    depth_out = []
    o = 0
    while has(f"output_blocks.{o}.0.in_layers.0.weight"):
        depth_out.append(tdepth(f"output_blocks.{o}"))
        o += 1
    cfg["transformer_depth_output"] = depth_out[::-1]  # consumed with pop() from the end (unet.py:649)
    cfg["context_dim"] = context_dim
    cfg["use_linear_in_transformer"] = bool(use_linear)
    if context_dim == 768:
        cfg["num_heads"] = 8                 # SD1.x
    else:
        cfg["num_head_channels"] = 64        # SD2.x / SDXL / refiner
    return cfg


def split_state_dict(sd):
    """loader.py:449-486 without the text encoders: -> ({'unet': ..., 'vae': ...}, guess dict)."""
    sd = preprocess_state_dict(load_torch_file(sd))
    unet = {k[len(UNET_PREFIX):]: v for k, v in sd.items() if k.startswith(UNET_PREFIX)}
    vae = {k[len(VAE_PREFIX):]: v for k, v in sd.items() if k.startswith(VAE_PREFIX)}
    vae = {k: v for k, v in vae.items() if not k.startswith(("loss.", "model_ema."))}
    unet_config = detect_unet_config(sd)
    is_sdxl = unet_config.get("adm_in_channels") is not None
    vae_config = dict(SD_VAE_CONFIG, scaling_factor=0.13025 if is_sdxl else 0.18215) if vae else None
    # the prediction type is not in the tensor shapes: checkpoints mark it with a 'v_pred' key (and 'ztsnr' for a zero-terminal-SNR schedule,
    # loader.py:462); a yaml next to the file or the caller decides otherwise (loader.py:543-567) -> forge_loader(prediction_type=...)
    pred = "v_prediction" if "v_pred" in sd else "epsilon"
    guess = {"unet_config": unet_config, "vae_config": vae_config, "is_sdxl": is_sdxl, "prediction_type": pred, "ztsnr": "ztsnr" in sd,
             "ignored": sorted({k.split(".")[0] for k in sd if not k.startswith((UNET_PREFIX, VAE_PREFIX))})}
    return {"unet": unet, "vae": vae}, guess


@torch.inference_mode()
def forge_loader(sd, loras=None, device="cuda", prediction_type=None):
    """checkpoint path / state dict (+ optional [(lora_sd, strength)]) -> ForgeDiffusionEngine on the native executors.
    prediction_type: 'epsilon' | 'v_prediction' | 'edm' to override what the checkpoint's marker keys say (SD2.x-768, v-pred SDXL finetunes)."""
    from .patcher.lora import merge_loras_into_state_dict
    parts, guess = split_state_dict(sd)
    if prediction_type is not None:
        guess["prediction_type"] = prediction_type
    unet_sd = parts["unet"]
    report = None
    if loras:
        unet_sd, report = merge_loras_into_state_dict(unet_sd, guess["unet_config"], [(load_torch_file(l), s) for l, s in loras], device=device)
    engine = build_engine(guess["unet_config"], unet_sd, guess["vae_config"], parts["vae"] or None, device=device,
                          prediction_type=guess["prediction_type"], ztsnr=guess["ztsnr"])
    engine.lora_report = report
    engine.model_guess = guess
    return engine
