"""A checkpoint FILE through the loader (VERDICT r5 item 7 / missing 5): synthetic SDXL and Flux checkpoints written in the reference's single-file key
layout (`model.diffusion_model.*` + `first_stage_model.*` / `vae.*` + text-encoder keys the native path ignores; backend/loader.py:442-498,
backend/state_dict.py) to `.safetensors` on disk, read back by `forge_amd.backend.loader.load_torch_file` (:24-31) -- until round 6 no test had ever
handed `forge_loader` a path.

CPU: the file round trip (tensors bit for bit, storage types kept, family / structure detected from the file's tensors, a `.ckpt` pickle with a
`state_dict` wrapper read the same way).  GPU: `forge_loader(path)` -> engine, one network forward against the engine built from the in-memory state
dict, bit for bit -- SDXL at FULL size (2.57 B parameters, a 5.1 GB file: the shapes the family detection is written for) and Flux at its own width."""
import os

import pytest
import torch

import forge_amd  # noqa: F401
from forge_amd import synth
from forge_amd.backend import loader

DEV = "cuda"
BF = torch.bfloat16


def sdxl_single_file(unet_sd, vae_sd, dtype=torch.float16):
    """the layout of an SDXL-base single-file checkpoint: UNet under model.diffusion_model., VAE (LDM names) under first_stage_model., the two text
    encoders under conditioner.embedders.{0,1}. (stand-in tensors: the native path takes conditioning tensors, loader.split_state_dict lists them as ignored)"""
    ck = {loader.UNET_PREFIX + k: v.to(dtype).contiguous() for k, v in unet_sd.items()}
    ck.update({loader.VAE_PREFIX + k: v.to(dtype).contiguous() for k, v in vae_sd.items()})
    ck["conditioner.embedders.0.transformer.text_model.embeddings.position_embedding.weight"] = torch.zeros(77, 768, dtype=dtype)
    ck["conditioner.embedders.1.model.positional_embedding"] = torch.zeros(77, 1280, dtype=dtype)
    return ck


def flux_single_file(tr_sd, vae_sd, dtype=BF):
    """Forge's own Flux layout: transformer under model.diffusion_model., diffusers-keyed VAE under vae., text encoders under text_encoders."""
    from test_loader_lora import _vae_ldm_to_diffusers_names
    ck = {loader.UNET_PREFIX + k: v.to(dtype).contiguous() for k, v in tr_sd.items()}
    nlev = len({k.split(".")[2] for k in vae_sd if k.startswith("decoder.up.")})
    ck.update({"vae." + k: v.to(dtype).contiguous() for k, v in _vae_ldm_to_diffusers_names(vae_sd, nlev).items()})
    ck["text_encoders.clip_l.transformer.text_model.embeddings.position_embedding.weight"] = torch.zeros(77, 768, dtype=dtype)
    return ck


def save(ck, path):
    from safetensors.torch import save_file
    save_file(ck, str(path), metadata={"format": "pt"})
    return str(path)


def test_safetensors_file_round_trip_and_detection(tmp_path):
    cfg, vcfg = synth.TINY_SDXL_UNET_CONFIG, synth.TINY_VAE_CONFIG
    unet = synth.synth_unet_state_dict(cfg, seed=0)
    vae = synth.synth_vae_state_dict(vcfg, seed=1)
    ck = sdxl_single_file(unet, vae)
    path = save(ck, tmp_path / "tiny_sdxl.safetensors")
    back = loader.load_torch_file(path)
    assert set(back) == set(ck) and all(torch.equal(back[k], ck[k]) and back[k].dtype == ck[k].dtype for k in ck)
    parts, guess = loader.split_state_dict(path)                 # a PATH, not a dict
    assert set(parts["unet"]) == set(unet) and all(torch.equal(parts["unet"][k], unet[k].half()) for k in unet)
    assert set(parts["vae"]) == set(vae) and guess["ignored"] == ["conditioner"]
    from_dict = loader.split_state_dict(ck)[1]
    assert guess == from_dict and guess["is_sdxl"] and guess["unet_config"]["adm_in_channels"] == cfg["adm_in_channels"]
    # the other container the reference reads (loader.py load_torch_file): a pickled dict with a `state_dict` wrapper
    torch.save({"state_dict": ck, "global_step": 1}, tmp_path / "tiny_sdxl.ckpt")
    back2 = loader.load_torch_file(str(tmp_path / "tiny_sdxl.ckpt"))
    assert set(back2) == set(ck) and all(torch.equal(back2[k], ck[k]) for k in ck)
    # Flux: bf16 storage survives the file, the family / compute type are read from the FILE's tensors
    fcfg, fvcfg = synth.TINY_FLUX_CONFIG, synth.TINY_FLUX_VAE_CONFIG
    tr = synth.synth_flux_state_dict(fcfg, seed=2)
    fck = flux_single_file(tr, synth.synth_vae_state_dict(fvcfg, seed=1))
    fpath = save(fck, tmp_path / "tiny_flux.safetensors")
    fback = loader.load_torch_file(fpath)
    assert loader.flux_prefix(fback) == loader.UNET_PREFIX
    fparts, fguess = loader.split_flux_state_dict(fback)
    assert fguess["flux_config"] == fcfg and fguess["dtype"] == BF and fguess["ignored"] == ["text_encoders"]
    assert all(fparts["transformer"][k].dtype == BF and torch.equal(fparts["transformer"][k], tr[k].to(BF)) for k in tr)
    assert fguess["vae_config"]["latent_channels"] == fvcfg["latent_channels"]


@pytest.mark.gpu
def test_sdxl_checkpoint_file_through_forge_loader_at_full_size(tmp_path):
    """SDXL-base as a file: 2.57 B-parameter UNet + the VAE, fp16, one 5.1 GB .safetensors.  forge_loader(path) must detect the family from the file's
    tensors (the structure the shipped SDXL configuration names) and produce the executor the in-memory path produces: one forward, bit for bit."""
    from forge_amd.backend.diffusion_engine.base import build_engine
    from forge_amd.backend.nn.layout import unet_param_shapes, vae_decoder_param_shapes, vae_encoder_param_shapes
    cfg, vcfg = synth.SDXL_UNET_CONFIG, synth.SDXL_VAE_CONFIG
    unet = {k: v.cpu() for k, v in synth.synth_state_dict_device(unet_param_shapes(cfg), 0, DEV).items()}       # drawn on the device: seconds
    vshapes = dict(vae_decoder_param_shapes(vcfg))
    vshapes.update(vae_encoder_param_shapes(vcfg))
    vae = {k: v.cpu() for k, v in synth.synth_state_dict_device(vshapes, 1, DEV).items()}
    path = save(sdxl_single_file(unet, vae), tmp_path / "sdxl_base.safetensors")
    assert os.path.getsize(path) > 5.0e9
    eng = loader.forge_loader(path, device=DEV)
    g = eng.model_guess
    assert unet_param_shapes(g["unet_config"]) == unet_param_shapes(cfg)
    for k in ("num_heads", "num_head_channels", "context_dim", "use_linear_in_transformer", "adm_in_channels", "transformer_depth", "transformer_depth_middle"):
        assert g["unet_config"].get(k) == cfg.get(k), k
    assert g["vae_config"] == vcfg
    assert g["is_sdxl"] and g["prediction_type"] == "epsilon" and g["ignored"] == ["conditioner"]
    direct = build_engine(cfg, unet, vcfg, vae, device=DEV)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(2, 4, 128, 128, generator=gen).to(DEV)
    t = torch.tensor([801.0, 23.0], device=DEV)
    ctx = torch.randn(2, 77, cfg["context_dim"], generator=gen).to(DEV)
    y = torch.randn(2, cfg["adm_in_channels"], generator=gen).to(DEV)
    a = eng.forge_objects.unet.model.diffusion_model.forward(x, t, context=ctx, y=y)
    b = direct.forge_objects.unet.model.diffusion_model.forward(x, t, context=ctx, y=y)
    assert torch.isfinite(a.float()).all() and torch.equal(a, b)
    lat = torch.randn(1, 4, 32, 32, generator=gen).to(DEV)
    assert torch.equal(eng.decode_first_stage(lat), direct.decode_first_stage(lat))


@pytest.mark.gpu
def test_flux_checkpoint_file_through_forge_loader_at_its_own_width(tmp_path):
    """Flux as a file at its own width (hidden 3072, 24 x 128, 1 + 1 blocks; bf16 transformer + diffusers-keyed VAE under `vae.`): forge_loader(path) ->
    FluxEngine in the stored compute type, forward bit for bit the directly built executor's."""
    from forge_amd.backend.nn.flux import IntegratedFluxTransformer2DModel
    from oracle.make_floor import FLUX_WIDTH_CONFIG, flux_width_inputs
    cfg = dict(FLUX_WIDTH_CONFIG)
    tr = synth.synth_flux_state_dict(cfg, seed=2)
    path = save(flux_single_file(tr, synth.synth_vae_state_dict(synth.TINY_FLUX_VAE_CONFIG, seed=1)), tmp_path / "flux_width.safetensors")
    eng = loader.forge_loader(path, device=DEV)
    net = eng.forge_objects.unet.model.diffusion_model
    assert eng.is_flux and net.dtype == BF and eng.model_guess["flux_config"] == cfg
    x, t, ctx, y, guid = flux_width_inputs(cfg, seed=33, lat=32, ltxt=64)
    args = (x.to(DEV), t.to(DEV), ctx.to(DEV, BF), y.to(DEV, BF), guid.to(DEV))
    direct = IntegratedFluxTransformer2DModel(cfg, {k: v.to(BF) for k, v in tr.items()}, device=DEV, dtype=BF)
    out = net.forward(*args)
    assert torch.isfinite(out.float()).all() and torch.equal(out, direct.forward(*args))
