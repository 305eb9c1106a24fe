"""Single-file checkpoint -> native engine: counterpart of backend/loader.py (`preprocess_state_dict` :442,
`split_state_dict` :449, `forge_loader` :498) for the SD1.x / SD2.x / SDXL-base LDM layouts and Flux.1 (dev / schnell) transformers.

The reference delegates model-family detection to the `huggingface_guess` package (git lllyasviel/huggingface_guess@84826248,
`launch_utils.py:397-404`; absent here): `guess(sd)` reads the UNet hyper-parameters off the tensor shapes
(`detection.detect_unet_config`, the ComfyUI model-detection algorithm) and matches them against the known model list.
`detect_unet_config` below restates that published shape-reading algorithm for the LDM UNet; the head layout, which shapes
cannot reveal, follows the model list (context 768 -> SD1.x: 8 heads; 1024 -> SD2.x, 2048 / 1280 -> SDXL: 64 channels per head).
The native executor binds LDM parameter names directly, so no key conversion is needed for the UNet or the (LDM-layout) VAE of
a single-file checkpoint; a diffusers-keyed VAE goes through `misc.diffusers_state_dict.vae_from_diffusers` (the reference calls
`huggingface_guess.diffusers_convert.convert_vae_state_dict`, loader.py:58-59).  `detect_flux_config` restates the same package's Flux
branch.  Parity of these restatements is UNPINNED against the package itself (it is not in the image): they are anchored on the reference's
call sites, on its engines' use of the resulting configuration (diffusion_engine/flux.py:36-47), on the per-family configuration files
the reference ships under backend/huggingface/ (SD1.5, SD-inpainting, SDXL-base UNets; FLUX.1-dev / -schnell transformers and VAE) and
on round trips through the native parameter-shape tables (tests/test_loader_lora.py)."""
import torch

from .diffusion_engine.base import build_engine

UNET_PREFIX = "model.diffusion_model."
VAE_PREFIX = "first_stage_model."


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


def detect_vae_config(vae_sd, **constants):
    """AutoencoderKL structure from an LDM-keyed VAE state dict (widths per level, ResBlocks per level, latent channels, quant convs); what no
    tensor carries (scaling / shift factor) comes in as `constants`."""
    n = _count(vae_sd, "decoder.up.{}.block.0.conv1.weight")
    return dict(in_channels=vae_sd["encoder.conv_in.weight"].shape[1] if "encoder.conv_in.weight" in vae_sd else 3,
                out_channels=vae_sd["decoder.conv_out.weight"].shape[0],
                block_out_channels=tuple(vae_sd[f"decoder.up.{l}.block.0.conv1.weight"].shape[0] for l in range(n)),
                layers_per_block=_count(vae_sd, "decoder.up.0.block.{}.conv1.weight") - 1, latent_channels=vae_sd["decoder.conv_in.weight"].shape[1],
                use_quant_conv="quant_conv.weight" in vae_sd, use_post_quant_conv="post_quant_conv.weight" in vae_sd, **constants)


def flux_prefix(sd):
    """-> key prefix of a Flux transformer inside `sd` ('model.diffusion_model.' in full checkpoints, '' in transformer-only files), or None."""
    for prefix in (UNET_PREFIX, ""):
        if prefix + "double_blocks.0.img_attn.norm.key_norm.scale" in sd:
            return prefix
    return None


def detect_flux_config(sd, prefix=UNET_PREFIX):
    """Flux hyper-parameters.  huggingface_guess (absent here) keys the family on `double_blocks.0.img_attn.norm.key_norm.scale`, counts the
    double / single blocks, reads `guidance_embed` off the presence of `guidance_in.in_layer.weight` and fills every other field with the
    Flux.1 constants; here the widths are read off the tensors instead (identical for Flux.1 files, and it keeps reduced-size twins
    loadable), head_dim is the norm scale's length, and the rotary split / theta -- which no tensor carries -- are Flux.1's."""
    g = lambda k: sd[prefix + k]  # noqa: E731
    hidden = g("img_in.weight").shape[0]
    head_dim = g("double_blocks.0.img_attn.norm.key_norm.scale").shape[0]
    if head_dim != 128:
        raise NotImplementedError(f"Flux head_dim {head_dim}: the rotary split [16, 56, 56] is defined for 128")
    return dict(in_channels=g("img_in.weight").shape[1] // 4, vec_in_dim=g("vector_in.in_layer.weight").shape[1],
                context_in_dim=g("txt_in.weight").shape[1], hidden_size=hidden, mlp_ratio=g("double_blocks.0.img_mlp.0.weight").shape[0] / hidden,
                num_heads=hidden // head_dim, depth=_count(sd, prefix + "double_blocks.{}.img_attn.qkv.weight"),
                depth_single_blocks=_count(sd, prefix + "single_blocks.{}.linear1.weight"), axes_dim=[16, 56, 56], theta=10000,
                qkv_bias=(prefix + "double_blocks.0.img_attn.qkv.bias") in sd, guidance_embed=(prefix + "guidance_in.in_layer.weight") in sd)


FLUX_VAE_PREFIXES = ("vae.", VAE_PREFIX)  # Forge's own Flux checkpoints store the VAE under 'vae.', converted LDM ones under 'first_stage_model.'


def split_flux_state_dict(sd):
    """Flux counterpart of split_state_dict: -> ({'transformer', 'vae'}, guess).  The compute type follows the stored tensors as the reference's
    loader does (bf16 files -> bf16, fp16 files -> fp16; fp32 files run in bf16, the reference's first choice for Flux); quantised storage
    (fp8 / nf4 / gguf) is outside the native path."""
    prefix = flux_prefix(sd)
    tr = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix) and not k.startswith(FLUX_VAE_PREFIXES + ("text_encoders.",))}
    vae = {}
    from .misc.diffusers_state_dict import vae_from_diffusers
    for vp in FLUX_VAE_PREFIXES:
        vae = {k[len(vp):]: v for k, v in sd.items() if k.startswith(vp)}
        if vae:
            vae = vae_from_diffusers(vae)
            break
    stored = tr["img_in.weight"].dtype
    if stored not in (torch.float16, torch.bfloat16, torch.float32):
        raise NotImplementedError(f"Flux checkpoint stored as {stored}: quantised formats are not on the native path")
    guess = {"flux_config": detect_flux_config(sd, prefix), "vae_config": detect_vae_config(vae, scaling_factor=0.3611, shift_factor=0.1159) if vae else None, "is_flux": True,
             "dtype": torch.float16 if stored == torch.float16 else torch.bfloat16,
             "ignored": sorted({k.split(".")[0] for k in sd if not k.startswith((prefix,) + FLUX_VAE_PREFIXES)} if prefix else set())}
    return {"transformer": tr, "vae": vae}, guess


def split_state_dict(sd):
    """loader.py:449-486 without the text encoders: -> ({'unet': ..., 'vae': ...}, guess dict)."""
    sd = preprocess_state_dict(load_torch_file(sd))
    unet = {k[len(UNET_PREFIX):]: v for k, v in sd.items() if k.startswith(UNET_PREFIX)}
    vae = {k[len(VAE_PREFIX):]: v for k, v in sd.items() if k.startswith(VAE_PREFIX)}
    vae = {k: v for k, v in vae.items() if not k.startswith(("loss.", "model_ema."))}
    from .misc.diffusers_state_dict import vae_from_diffusers
    vae = vae_from_diffusers(vae)   # loader.py:58-59
    unet_config = detect_unet_config(sd)
    is_sdxl = unet_config.get("adm_in_channels") is not None
    vae_config = detect_vae_config(vae, scaling_factor=0.13025 if is_sdxl else 0.18215, shift_factor=0.0) if vae else None
    # the prediction type is not in the tensor shapes: checkpoints mark it with a 'v_pred' key (and 'ztsnr' for a zero-terminal-SNR schedule,
    # loader.py:462); a yaml next to the file or the caller decides otherwise (loader.py:543-567) -> forge_loader(prediction_type=...)
    pred = "v_prediction" if "v_pred" in sd else "epsilon"
    # SD2.x-768 v-prediction checkpoints carry NO marker key.  huggingface_guess (the package the reference's loader.py:567 takes
    # model_type from; absent here, restated) tells them from SD2.x-base (epsilon) by a statistic of one trained tensor: the standard
    # deviation of output_blocks.11.1.transformer_blocks.0.norm1.bias exceeds 0.09 for the v-prediction models (SD2.x: context_dim
    # 1024, 4 input channels).  Loading such a file as epsilon produces garbage images silently, so the heuristic is applied and reported.
    pred_source = "marker key" if "v_pred" in sd else "default"
    probe = UNET_PREFIX + "output_blocks.11.1.transformer_blocks.0.norm1.bias"
    if pred == "epsilon" and unet_config.get("context_dim") == 1024 and unet_config.get("in_channels") == 4 and not is_sdxl and probe in sd:
        if float(sd[probe].float().std(unbiased=False)) > 0.09:   # the population form, as huggingface_guess computes it
            pred, pred_source = "v_prediction", "SD2.x norm1.bias statistic (std > 0.09)"
    guess = {"unet_config": unet_config, "vae_config": vae_config, "is_sdxl": is_sdxl, "prediction_type": pred, "prediction_type_source": pred_source,
             "ztsnr": "ztsnr" in sd, "ignored": sorted({k.split(".")[0] for k in sd if not k.startswith((UNET_PREFIX, VAE_PREFIX))})}
    return {"unet": unet, "vae": vae}, guess


@torch.inference_mode()
def forge_loader(sd, loras=None, device="cuda", prediction_type=None):
    """checkpoint path / state dict (+ optional [(lora_sd, strength)]) -> ForgeDiffusionEngine on the native executors.
    prediction_type: 'epsilon' | 'v_prediction' | 'edm' to override what the checkpoint's marker keys say (SD2.x-768, v-pred SDXL finetunes)."""
    from .patcher.lora import merge_loras_into_state_dict
    sd = load_torch_file(sd)
    if flux_prefix(sd) is not None:
        from .diffusion_engine.base import build_flux_engine
        from .patcher.lora import merge_loras_into_flux_state_dict
        parts, guess = split_flux_state_dict(sd)
        tsd, report = parts["transformer"], None
        if loras:   # native and diffusers-named Flux LoRAs (comfyui_lora_collection/lora.py:286-299, :342-347), merged offline like the UNet's
            tsd, report = merge_loras_into_flux_state_dict(tsd, guess["flux_config"], [(load_torch_file(l), s) for l, s in loras], device=device, dtype=guess["dtype"])
        engine = build_flux_engine(guess["flux_config"], tsd, device=device, vae_config=guess["vae_config"],
                                   vae_state_dict=parts["vae"] or None, dtype=guess["dtype"])
        engine.lora_report = report
        engine.model_guess = guess
        return engine
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
