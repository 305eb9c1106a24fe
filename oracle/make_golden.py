"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/* by running the REAL reference (imported read-only
from /root/reference, see oracle/ref_import.py) on CPU in fp32 with the deterministic synthetic weights of
forge_amd.synth.  Run in the authoring container only:

    python -m oracle.make_golden [--full]

`--full` additionally runs BASELINE config 0 (SD1.5 512x512, B=1, 20-step Euler, CFG 7) end to end
(~1-2 min) and stores its final latent + decoded uint8 image.
The fixtures are what the GPU box checks against (the reference cannot travel there).
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import forge_amd  # noqa: E402
from forge_amd import synth  # noqa: E402
from oracle import ref_import  # noqa: E402
from oracle.rng import ImageRNG  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def _inputs(cfg, b, hw, seed):
    g = torch.Generator("cpu").manual_seed(seed)
    x = torch.randn(b, cfg["in_channels"], hw, hw, generator=g)
    ctx = torch.randn(b, 77, cfg["context_dim"], generator=g)
    y = torch.randn(b, cfg["adm_in_channels"], generator=g) if cfg.get("adm_in_channels") else None
    t = torch.tensor([981.0, 401.0, 37.0, 3.0][:b])
    return x, t, ctx, y


def gen_keys():
    ref = ref_import.load_reference()
    out = {}
    for name, cfg in (("sd15", synth.SD15_UNET_CONFIG), ("sdxl", synth.SDXL_UNET_CONFIG),
                      ("tiny_sd15", synth.TINY_SD15_UNET_CONFIG), ("tiny_sdxl", synth.TINY_SDXL_UNET_CONFIG)):
        with torch.device("meta"):
            net = ref_import.build_ref_unet(cfg)
        out[name] = {k: list(v.shape) for k, v in net.state_dict().items()}
    for name, cfg in (("vae", synth.SD15_VAE_CONFIG), ("tiny_vae", synth.TINY_VAE_CONFIG)):
        with torch.device("meta"):
            vae = ref_import.build_ref_vae(cfg)
        out[name] = {k: list(v.shape) for k, v in vae.state_dict().items()
                     if k.startswith("decoder.") or k.startswith("post_quant_conv.")}
    with open(os.path.join(GOLD, "param_shapes.json"), "w") as f:
        json.dump(out, f)
    print("param_shapes.json", {k: len(v) for k, v in out.items()})


def gen_unet(name, cfg, b=2, hw=16):
    sd = synth.synth_unet_state_dict(cfg, seed=0)
    net = ref_import.build_ref_unet(cfg, sd)
    x, t, ctx, y = _inputs(cfg, b, hw, seed=11)
    with torch.no_grad():
        eps = net(x.clone(), t, context=ctx, y=y, transformer_options={})
    torch.save({"x": x, "t": t, "ctx": ctx, "y": y, "eps": eps, "hw": hw}, os.path.join(GOLD, f"{name}_unet_fwd.pt"))
    print(name, "unet fwd", tuple(eps.shape), float(eps.std()))
    return net, sd


def synth_control(cfg, b, hw, seed=21):
    """ControlNet-style residuals for every injection point of the UNet (unet.py:714,732,739): 'input' after each input block,
    'middle', 'output' on every skip; lists are consumed from the END, so they are stored in pop() order reversed."""
    from forge_amd.backend.nn.layout import Down, unet_layout
    lay = unet_layout(cfg)
    g = torch.Generator("cpu").manual_seed(seed)
    shapes = []
    ch, res = cfg["model_channels"], hw
    for blk in lay.input_blocks:
        for L in blk:
            if isinstance(L, Down):
                res = (res + 2 - 3) // 2 + 1
            ch = getattr(L, "cout", getattr(L, "ch", ch))
        shapes.append((b, ch, res, res))
    inp = [torch.randn(s, generator=g) * 0.3 for s in shapes]
    inp[2] = None                                             # a None entry is skipped (unet.py:47)
    mid = [torch.randn(shapes[-1], generator=g) * 0.3]
    outp = [torch.randn(s, generator=g) * 0.3 for s in shapes]  # skips are popped last-in-first-out, lists too
    return {"input": inp[::-1], "middle": mid, "output": outp}


def gen_unet_control(name, cfg, net, b=2, hw=16):
    x, t, ctx, y = _inputs(cfg, b, hw, seed=11)
    control = synth_control(cfg, b, hw)
    with torch.no_grad():
        eps = net(x.clone(), t, context=ctx, y=y, control={k: list(v) for k, v in control.items()}, transformer_options={})
    torch.save({"eps": eps, "hw": hw}, os.path.join(GOLD, f"{name}_unet_ctrl.pt"))
    print(name, "unet fwd with control", float(eps.std()))


def clip_test_tokens(cfg, b=2, seed=1):
    """token ids shaped like a tokenised prompt batch: BOS, words, EOS (= largest id), padding with EOS (classic_engine.py:76-83)"""
    g = torch.Generator("cpu").manual_seed(seed)
    v = cfg["vocab_size"]
    ids = torch.randint(3, v - 2, (b, 77), generator=g)
    ids[:, 0] = v - 2
    for i, n in enumerate((10, 76)[:b]):
        ids[i, n:] = v - 1
    return ids


def gen_clip(name, cfg):
    """transformers.CLIPTextModel (the package that executes the reference's text-encoder arithmetic, backend/nn/clip.py:4-12)
    with the synthetic weights: every hidden state the classic engine can select, final LayerNorm, pooled (+ projection)."""
    from transformers import CLIPTextConfig, CLIPTextModel
    sd = synth.synth_clip_state_dict(cfg)
    hc = CLIPTextConfig(vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"], intermediate_size=cfg["intermediate_size"],
                        num_hidden_layers=cfg["num_hidden_layers"], num_attention_heads=cfg["num_attention_heads"], max_position_embeddings=77,
                        hidden_act=cfg["hidden_act"], eos_token_id=2, bos_token_id=0, pad_token_id=1, projection_dim=cfg["hidden_size"])
    m = CLIPTextModel(hc).eval()
    pref = "text_model." if any(k.startswith("text_model.") for k in m.state_dict()) else ""
    miss = m.load_state_dict({pref + k[len("transformer.text_model."):]: v for k, v in sd.items() if k.startswith("transformer.text_model.")}, strict=True)
    ids = clip_test_tokens(cfg)
    with torch.no_grad():
        out = m(ids, output_hidden_states=True)
        fin = m.text_model.final_layer_norm if hasattr(m, "text_model") else m.final_layer_norm
        res = {"ids": ids, "hidden_last": out.hidden_states[-1], "hidden_penultimate": out.hidden_states[-2], "last_hidden_state": out.last_hidden_state,
               "penultimate_final_ln": fin(out.hidden_states[-2]), "pooled": out.pooler_output}
        if "transformer.text_projection.weight" in sd:
            res["pooled_projected"] = torch.nn.functional.linear(out.pooler_output, sd["transformer.text_projection.weight"])
    torch.save(res, os.path.join(GOLD, f"{name}.pt"))
    print(name, tuple(out.last_hidden_state.shape), float(out.last_hidden_state.std()))


def t5_test_tokens(cfg, b=2, t=256, seed=3):
    """token ids shaped like T5TextProcessingEngine's chunks: words, EOS = 1, padding 0 up to min_length 256 (t5_engine.py:74-92)"""
    g = torch.Generator("cpu").manual_seed(seed)
    ids = torch.randint(2, cfg["vocab_size"], (b, t), generator=g)
    for i, n in enumerate((12, 200)[:b]):
        ids[i, n] = 1
        ids[i, n + 1:] = 0
    return ids


def gen_t5(name="tiny_t5", cfg=None):
    """the REAL reference class (backend/nn/t5.py IntegratedT5) on the synthetic weights: the encoder output the T5 text-processing engine feeds Flux"""
    import importlib
    import transformers.activations  # noqa: F401 -- before the reference's import stubs (a spec-less torchvision) are installed: t5.py imports NewGELUActivation
    ref_import.load_reference()
    t5 = importlib.import_module("backend.nn.t5")
    cfg = cfg or synth.TINY_T5_CONFIG
    sd = synth.synth_t5_state_dict(cfg)
    m = t5.IntegratedT5(cfg).eval()
    m.load_state_dict(sd, strict=True)
    ids = t5_test_tokens(cfg)
    with torch.no_grad():
        z = m.transformer(input_ids=ids)
    torch.save({"ids": ids, "z": z}, os.path.join(GOLD, f"{name}.pt"))
    print(name, tuple(z.shape), float(z.std()))


def gen_vae(name, cfg, b=2, hw=8):
    sd = synth.synth_vae_decoder_state_dict(cfg, seed=1)
    vae = ref_import.build_ref_vae(cfg)
    missing = vae.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys, missing.unexpected_keys
    assert all(k.startswith(("encoder.", "quant_conv.")) for k in missing.missing_keys), missing.missing_keys
    g = torch.Generator("cpu").manual_seed(5)
    z = torch.randn(b, cfg["latent_channels"], hw, hw, generator=g)
    with torch.no_grad():
        out = vae.decode(z)
        # decode_first_stage (diffusion_engine/sd15.py:80-84) on latents `z*0.5`
        lat = z * 0.5
        dec = torch.clamp((vae.decode(vae.process_out(lat)) + 1.0) / 2.0, 0.0, 1.0) * 2.0 - 1.0
    torch.save({"z": z, "decode": out, "lat": lat, "decode_first_stage": dec}, os.path.join(GOLD, f"{name}_decode.pt"))
    print(name, "decode", tuple(out.shape), float(out.std()))


def gen_vae_encode(name, cfg, b=2, h=32, w=48):
    """Encoder + quant_conv of the real reference (vae.py:183-200, :296-298) on a seeded image in [-1, 1]; the posterior sample
    with an explicit noise tensor (the reference draws torch.randn on the CPU default generator, vae.py:28)."""
    sd = synth.synth_vae_state_dict(cfg, seed=1)
    vae = ref_import.build_ref_vae(cfg)
    missing = vae.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and not missing.missing_keys, missing
    g = torch.Generator("cpu").manual_seed(9)
    x = torch.rand(b, 3, h, w, generator=g) * 2 - 1
    with torch.no_grad():
        moments = vae.quant_conv(vae.encoder(x))
        torch.manual_seed(123)
        sample = vae.encode(x)          # mean + std * torch.randn(shape) under seed 123
        torch.manual_seed(123)
        noise = torch.randn(sample.shape)
        latent = vae.process_in(sample)
    torch.save({"x": x, "moments": moments, "noise": noise, "sample": sample, "process_in": latent}, os.path.join(GOLD, f"{name}_encode.pt"))
    print(name, "encode", tuple(moments.shape), float(moments.std()))


class _Hijack:
    """modules/sd_samplers_common.py:214-235 TorchHijack stand-in."""

    def __init__(self, rng):
        self.rng = rng

    def __getattr__(self, item):
        if item == "randn_like":
            return lambda x: self.rng.next()
        return getattr(torch, item)


def ref_sample(net, cfg, cond, uncond, seeds, hw, steps, sampler, cfg_scale=7.0, source="CPU", trace=None, sigmas_override=None):
    """Reference k-diffusion loop + reference sampling_function + reference UNet (CPU fp32).  sigmas_override: run the sampler on this schedule
    (the scheduler fixtures) instead of the sampler's default one."""
    ref = ref_import.load_reference()
    pred = ref_import.build_ref_predictor()
    den = ref_import.RefDenoiser(net, pred, seeds)
    rng = ImageRNG((cfg["in_channels"], hw, hw), seeds, source)
    x = rng.next()
    linker = den.inner_model
    if sampler == "DPM++ 2M":
        sigmas = ref.kd_sampling.get_sigmas_karras(n=steps, sigma_min=linker.sigmas[0].item(),
                                                   sigma_max=linker.sigmas[-1].item(), device="cpu")
        fn = ref.kd_sampling.sample_dpmpp_2m
    else:
        sigmas = linker.get_sigmas(steps)
        fn = ref.kd_sampling.sample_euler if sampler == "Euler" else ref.kd_sampling.sample_euler_ancestral
    if sigmas_override is not None:
        sigmas = sigmas_override
    x = pred.noise_scaling(sigmas[0], x, torch.zeros_like(x), max_denoise=False)
    ref.kd_sampling.torch = _Hijack(rng)
    ref.sampling_function.sampling_prepare(den.patcher, x=x)
    extra = {"cond": cond, "uncond": uncond, "cond_scale": cfg_scale, "s_min_uncond": 0.0, "image_cond": None}

    def cb(d):
        if trace is not None:
            trace.append(d["denoised"].clone())
    try:
        out = fn(den, x, sigmas, extra_args=extra, callback=cb, disable=True)
    finally:
        ref.kd_sampling.torch = torch
        ref.sampling_function.sampling_cleanup(den.patcher)
    return out, sigmas


def mask_noise_fn(shape):
    """deterministic stand-in for the torch.randn_like of sd_samplers_cfg_denoiser.py:180 (device RNG in the reference)"""
    def fn(step):
        return torch.randn(shape, generator=torch.Generator("cpu").manual_seed(7000 + step))
    return fn


def ref_sample_img2img(net, cfg, cond, uncond, seeds, init_latent, steps, strength, sampler, cfg_scale=7.0, mask=None, nmask=None):
    """Reference k-diffusion loop + sampling_function + UNet driven the way sample_img2img does
    (modules/sd_samplers_kdiffusion.py:136-194; mask blending of sd_samplers_cfg_denoiser.py:178-181,204-213 and
    processing.py:1865-1866 restated around the reference denoiser)."""
    ref = ref_import.load_reference()
    pred = ref_import.build_ref_predictor()
    den = ref_import.RefDenoiser(net, pred, seeds)
    rng = ImageRNG(tuple(init_latent.shape[1:]), seeds, "CPU")
    noise = rng.next()
    linker = den.inner_model
    t_enc = int(min(strength, 0.999) * steps)                       # sd_samplers_common.py:30-31
    if sampler == "DPM++ 2M":
        sigmas = ref.kd_sampling.get_sigmas_karras(n=steps, sigma_min=linker.sigmas[0].item(), sigma_max=linker.sigmas[-1].item(), device="cpu")
        fn = ref.kd_sampling.sample_dpmpp_2m
    else:
        sigmas = linker.get_sigmas(steps)
        fn = ref.kd_sampling.sample_euler if sampler == "Euler" else ref.kd_sampling.sample_euler_ancestral
    sigma_sched = sigmas[steps - t_enc - 1:]
    xi = pred.noise_scaling(sigma_sched[0], noise, init_latent, max_denoise=False)
    mnoise = mask_noise_fn(tuple(init_latent.shape))
    step = [0]

    class Model:  # the k-diffusion loops read model.inner_model.predictor (sampling.py:143)
        inner_model = den.inner_model

        def __call__(self, x, sigma, **extra):
            if mask is not None:
                noisy = pred.noise_scaling(sigma[:, None, None, None], mnoise(step[0]), init_latent, max_denoise=False)
                x = x * nmask + noisy * mask
            d = den(x, sigma, **extra)
            if mask is not None:
                d = d * nmask + init_latent * mask
            step[0] += 1
            return d

    model = Model()

    ref.kd_sampling.torch = _Hijack(rng)
    ref.sampling_function.sampling_prepare(den.patcher, x=xi)
    extra = {"cond": cond, "uncond": uncond, "cond_scale": cfg_scale, "s_min_uncond": 0.0, "image_cond": None}
    try:
        out = fn(model, xi, sigma_sched, extra_args=extra, disable=True)
    finally:
        ref.kd_sampling.torch = torch
        ref.sampling_function.sampling_cleanup(den.patcher)
    if mask is not None:
        out = out * nmask + init_latent * mask
    return out, sigma_sched


def gen_img2img(name, cfg, net, b=2, hw=16):
    adm = cfg.get("adm_in_channels")
    c, uc = synth.synth_conditioning(b, cfg["context_dim"], adm, seed=1234)
    if adm:
        c, uc = ref_import.SdxlCond(c), ref_import.SdxlCond(uc)
    seeds = [2000 + i for i in range(b)]
    init = torch.randn(b, cfg["in_channels"], hw, hw, generator=torch.Generator("cpu").manual_seed(31)) * 0.8
    res = {"seeds": seeds, "hw": hw, "init_latent": init}
    for sampler, steps, strength in (("Euler", 8, 0.6), ("Euler a", 8, 0.5), ("DPM++ 2M", 9, 0.75)):
        lat, sched = ref_sample_img2img(net, cfg, c, uc, seeds, init, steps, strength, sampler)
        res[sampler] = {"steps": steps, "denoising_strength": strength, "latent": lat, "sigma_sched": sched}
        print(name, "img2img", sampler, float(lat.std()))
    # inpaint-style latent mask: keep the left half (mask = 1 where the original is kept, nmask = 1 - mask)
    nmask = torch.zeros(b, cfg["in_channels"], hw, hw)
    nmask[..., hw // 2:] = 1.0
    nmask[0, :, : hw // 4] = 0.5  # soft values as a blurred mask would give
    mask = 1.0 - nmask
    lat, sched = ref_sample_img2img(net, cfg, c, uc, seeds, init, 8, 0.6, "Euler", mask=mask, nmask=nmask)
    res["Euler_masked"] = {"steps": 8, "denoising_strength": 0.6, "latent": lat, "mask": mask, "nmask": nmask}
    torch.save(res, os.path.join(GOLD, f"{name}_img2img.pt"))


def synth_lora(cfg, seed=3, rank=8):
    """A synthetic LoRA file for the tiny SD1.5-style UNet in the naming styles found in the wild: kohya/diffusers names
    (lora_unet_down_blocks_..), LDM names (lora_unet_input_blocks_..), a LoCon conv with a Tucker mid tensor, a rank that is
    not a multiple of 8, a `.diff` / `.diff_b` pair, a `.set_weight`, plus text-encoder keys that must be ignored."""
    from forge_amd.backend.nn.layout import unet_param_shapes
    shapes = unet_param_shapes(cfg)
    g = torch.Generator("cpu").manual_seed(seed)

    def rn(*shape, scale=0.05):
        return (torch.randn(*shape, generator=g) * scale).half()

    sd = {}

    def lora(name, key, r, alpha=None, conv_mid=False):
        o, i = shapes[key][0], shapes[key][1]
        if len(shapes[key]) == 4 and conv_mid:
            sd[name + ".lora_up.weight"] = rn(o, r, 1, 1)
            sd[name + ".lora_mid.weight"] = rn(r, r, shapes[key][2], shapes[key][3])
            sd[name + ".lora_down.weight"] = rn(r, i, 1, 1)
        elif len(shapes[key]) == 4:
            sd[name + ".lora_up.weight"] = rn(o, r, 1, 1)
            sd[name + ".lora_down.weight"] = rn(r, i, shapes[key][2], shapes[key][3])
        else:
            sd[name + ".lora_up.weight"] = rn(o, r)
            sd[name + ".lora_down.weight"] = rn(r, i)
        if alpha is not None:
            sd[name + ".alpha"] = torch.tensor(float(alpha))

    lora("lora_unet_down_blocks_0_attentions_0_transformer_blocks_0_attn1_to_q", "input_blocks.1.1.transformer_blocks.0.attn1.to_q.weight", rank, alpha=4)
    lora("lora_unet_down_blocks_0_attentions_0_transformer_blocks_0_attn2_to_k", "input_blocks.1.1.transformer_blocks.0.attn2.to_k.weight", rank, alpha=rank)
    lora("lora_unet_down_blocks_0_attentions_0_transformer_blocks_0_ff_net_0_proj", "input_blocks.1.1.transformer_blocks.0.ff.net.0.proj.weight", 5)
    lora("lora_unet_mid_block_attentions_0_proj_in", "middle_block.1.proj_in.weight", rank, alpha=2)
    lora("lora_unet_input_blocks_1_0_in_layers_2", "input_blocks.1.0.in_layers.2.weight", rank, alpha=8)            # LDM naming
    lora("lora_unet_up_blocks_2_resnets_1_conv1", "output_blocks.5.0.in_layers.2.weight", 4, alpha=1, conv_mid=True)
    lora("lora_unet_output_blocks_4_0_out_layers_3", "output_blocks.4.0.out_layers.3.weight", 12, alpha=6)
    k = "output_blocks.5.0.emb_layers.1"
    sd["diffusion_model." + k + ".diff"] = rn(*shapes[k + ".weight"], scale=0.02)
    sd["diffusion_model." + k + ".diff_b"] = rn(*shapes[k + ".bias"], scale=0.02)
    sd["diffusion_model.out.2.set_weight"] = rn(*shapes["out.2.weight"], scale=0.1)
    sd["lora_te_text_model_encoder_layers_0_mlp_fc1.lora_up.weight"] = rn(16, 4)
    sd["lora_te_text_model_encoder_layers_0_mlp_fc1.lora_down.weight"] = rn(4, 16)
    return sd


def synth_lycoris(cfg, seed=8):
    """A synthetic LyCORIS file for the tiny UNet: LoHa (linear; conv through Tucker cores), LoKr (full x low-rank; conv with a 4-d w2; low-rank
    w1 with Tucker w2), GLoRA (square and non-square targets), DoRA scales on a plain LoRA / a LoHa / a LoKr (output- and input-axis), and a
    w_norm / b_norm pair."""
    from forge_amd.backend.nn.layout import unet_param_shapes
    shapes = unet_param_shapes(cfg)
    g = torch.Generator("cpu").manual_seed(seed)
    rn = lambda *shape, scale=0.1: (torch.randn(*shape, generator=g) * scale).half()
    sd = {}
    name = lambda key: "lora_unet_" + key[:-len(".weight")].replace(".", "_")
    r = 4
    # LoHa, linear
    k = "input_blocks.1.1.transformer_blocks.0.attn1.to_v.weight"
    o, i = shapes[k]
    sd.update({name(k) + ".hada_w1_a": rn(o, r), name(k) + ".hada_w1_b": rn(r, i), name(k) + ".hada_w2_a": rn(o, r), name(k) + ".hada_w2_b": rn(r, i),
               name(k) + ".alpha": torch.tensor(2.0)})
    # LoHa, conv 3x3 with Tucker cores, + DoRA
    k = "input_blocks.1.0.out_layers.3.weight"
    o, i, kh, kw = shapes[k]
    sd.update({name(k) + ".hada_t1": rn(r, r, kh, kw), name(k) + ".hada_w1_a": rn(r, o), name(k) + ".hada_w1_b": rn(r, i),
               name(k) + ".hada_t2": rn(r, r, kh, kw), name(k) + ".hada_w2_a": rn(r, o), name(k) + ".hada_w2_b": rn(r, i),
               name(k) + ".alpha": torch.tensor(4.0), name(k) + ".dora_scale": (torch.rand(o, 1, 1, 1, generator=g) + 0.5).half()})
    # LoKr, linear: full w1 x low-rank w2
    k = "middle_block.1.transformer_blocks.0.attn2.to_q.weight"
    o, i = shapes[k]
    sd.update({name(k) + ".lokr_w1": rn(4, 4, scale=0.3), name(k) + ".lokr_w2_a": rn(o // 4, r), name(k) + ".lokr_w2_b": rn(r, i // 4), name(k) + ".alpha": torch.tensor(2.0)})
    # LoKr, conv: low-rank w1, full 4-d w2, + DoRA on the input axis
    k = "output_blocks.4.0.in_layers.2.weight"
    o, i, kh, kw = shapes[k]
    sd.update({name(k) + ".lokr_w1_a": rn(8, 2, scale=0.3), name(k) + ".lokr_w1_b": rn(2, 8, scale=0.3), name(k) + ".lokr_w2": rn(o // 8, i // 8, kh, kw),
               name(k) + ".alpha": torch.tensor(1.0), name(k) + ".dora_scale": (torch.rand(1, i, 1, 1, generator=g) + 0.5).half()})
    # (LoKr with a Tucker-decomposed w2 is left out: torch.kron rejects the einsum's non-contiguous result in the reference itself, torch 2.10)
    # GLoRA: square and non-square linear
    for k in ("input_blocks.3.1.transformer_blocks.0.attn1.to_out.0.weight", "input_blocks.3.1.transformer_blocks.0.attn2.to_k.weight"):
        o, i = shapes[k]
        sd.update({name(k) + ".a1.weight": rn(i, r), name(k) + ".a2.weight": rn(r, i), name(k) + ".b1.weight": rn(o, r), name(k) + ".b2.weight": rn(r, i),
                   name(k) + ".alpha": torch.tensor(2.0)})
    # DoRA on a plain LoRA
    k = "output_blocks.2.1.transformer_blocks.0.ff.net.2.weight"
    o, i = shapes[k]
    sd.update({name(k) + ".lora_up.weight": rn(o, r), name(k) + ".lora_down.weight": rn(r, i), name(k) + ".alpha": torch.tensor(2.0),
               name(k) + ".dora_scale": (torch.rand(o, 1, generator=g) + 0.5).half()})
    # w_norm / b_norm
    k = "out.0"
    sd["diffusion_model." + k + ".w_norm"] = rn(*shapes[k + ".weight"], scale=0.02)
    sd["diffusion_model." + k + ".b_norm"] = rn(*shapes[k + ".bias"], scale=0.02)
    return sd


def synth_flux_lora(cfg, seed=3, rank=4):
    """a diffusers-named Flux LoRA (q / k / v and add_k_proj of a double block, to_q + proj_mlp of a single block, the swapped norm_out), one native-named
    and one OneTrainer-named entry"""
    g = torch.Generator().manual_seed(seed)
    hs = cfg["hidden_size"]
    lo = {}

    def add(name, out_f, in_f, alpha=2.0):
        lo[name + ".lora_up.weight"] = torch.randn(out_f, rank, generator=g) * 0.1
        lo[name + ".lora_down.weight"] = torch.randn(rank, in_f, generator=g) * 0.1
        lo[name + ".alpha"] = torch.tensor(alpha)
    for n in ("to_q", "to_k", "to_v", "add_k_proj"):
        add(f"transformer.transformer_blocks.0.attn.{n}", hs, hs)
    add("transformer.single_transformer_blocks.1.attn.to_q", hs, hs)
    add("transformer.single_transformer_blocks.1.proj_mlp", 4 * hs, hs)
    add("transformer.norm_out.linear", 2 * hs, hs)
    add("lora_unet_double_blocks_1_img_mlp_0", int(hs * cfg["mlp_ratio"]), hs)
    add("lora_transformer_transformer_blocks_1_ff_context_net_2", hs, int(hs * cfg["mlp_ratio"]))
    return lo


def gen_flux_lora():
    """The REAL reference's Flux LoRA handling: comfyui_lora_collection/utils.py flux_to_diffusers (slice / function targets), load_lora, and
    backend/patcher/lora.py merge_lora_to_weight with the offsets / functions ModelPatcher.add_patches attaches (base.py:99-113) -> merged weights of the tiny
    Flux transformer (fp16 weights, fp32 computation)."""
    import importlib
    from forge_amd.backend.nn.layout import flux_param_shapes
    ref_import.load_reference()
    rl = importlib.import_module("backend.patcher.lora")
    cu = importlib.import_module("packages_3rdparty.comfyui_lora_collection.utils")
    cfg = synth.TINY_FLUX_CONFIG
    sd = {k: v.half() for k, v in synth.synth_flux_state_dict(cfg, seed=2).items()}
    key_map = {}
    for k in flux_param_shapes(cfg):           # comfyui_lora_collection/lora.py:286-299 (generic part) + :342-347 (Flux diffusers spellings)
        mk = "diffusion_model." + k
        if k.endswith(".weight"):
            key_map["lora_unet_" + k[:-len(".weight")].replace(".", "_")] = mk
            key_map["diffusion_model." + k[:-len(".weight")]] = mk
        else:
            key_map[mk] = mk
    for dk, to in cu.flux_to_diffusers(dict(cfg), output_prefix="diffusion_model.").items():
        if dk.endswith(".weight"):
            stem = dk[:-len(".weight")]
            key_map["transformer." + stem] = to
            key_map["lycoris_" + stem.replace(".", "_")] = to
            key_map["lora_transformer_" + stem.replace(".", "_")] = to
    patch_dict, remaining = rl.load_lora(synth_flux_lora(cfg), key_map)
    per_key = {}
    for target, pv in patch_dict.items():
        mk, off, fn = (target, None, None) if not isinstance(target, tuple) else (target[0], target[1], target[2] if len(target) > 2 else None)
        per_key.setdefault(mk[len("diffusion_model."):], []).append([0.8, pv, 1.0, off, fn])
    merged = {k: rl.merge_lora_to_weight(patches, sd[k].clone(), key=k, computation_dtype=torch.float32) for k, patches in per_key.items()}
    # the same merge on bfloat16 storage (Flux's own type; round 6): the reference casts to fp32, merges, casts once to the weight's type
    sd_bf = {k: v.bfloat16() for k, v in synth.synth_flux_state_dict(cfg, seed=2).items()}
    merged_bf16 = {k: rl.merge_lora_to_weight(patches, sd_bf[k].clone(), key=k, computation_dtype=torch.float32) for k, patches in per_key.items()}
    assert all(v.dtype == torch.bfloat16 for v in merged_bf16.values())
    torch.save({"strength": 0.8, "merged": merged, "merged_bf16": merged_bf16, "remaining": sorted(remaining),
                "key_map_targets": {k: (v if isinstance(v, str) else (v[0], v[1], v[2].__name__ if len(v) > 2 else None)) for k, v in key_map.items()}},
               os.path.join(GOLD, "tiny_flux_lora_merge.pt"))
    print("tiny_flux lora merge:", sorted(merged), "remaining", sorted(remaining))


def gen_lora(name="tiny_sd15", cfg=None):
    """The REAL reference's key map, patch parser and merge (backend/patcher/lora.py:43,19,85; comfyui_lora_collection/lora.py)
    on a synthetic LoRA for the tiny UNet: merged weights of every patched parameter (fp16 weights, fp32 computation)."""
    import importlib
    from types import SimpleNamespace
    cfg = cfg or synth.TINY_SD15_UNET_CONFIG
    ref_import.load_reference()
    rl = importlib.import_module("backend.patcher.lora")
    sd = {k: v.half() for k, v in synth.synth_unet_state_dict(cfg, seed=0).items()}
    keys = {"diffusion_model." + k: None for k in sd}
    rcfg = {k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in cfg.items()}
    model = SimpleNamespace(state_dict=lambda: keys, diffusion_model=SimpleNamespace(config=rcfg), config=SimpleNamespace(huggingface_repo="sd15"))
    key_map = rl.model_lora_keys_unet(model, {})
    lora_sd = synth_lora(cfg)
    patch_dict, remaining = rl.load_lora(lora_sd, key_map)
    strength = 0.8
    merged = {}
    for mk, pv in patch_dict.items():
        k = mk[len("diffusion_model."):]
        merged[k] = rl.merge_lora_to_weight([(strength, pv, 1.0, None, None)], sd[k].clone(), key=k, computation_dtype=torch.float32)
    ly_dict, ly_remaining = rl.load_lora(synth_lycoris(cfg), key_map)
    ly_merged, ly_kinds = {}, {}
    for mk, pv in ly_dict.items():
        k = mk[len("diffusion_model."):]
        ly_kinds[k] = pv[0]
        ly_merged[k] = rl.merge_lora_to_weight([(0.7, pv, 1.0, None, None)], sd[k].clone(), key=k, computation_dtype=torch.float32)
    torch.save({"strength": 0.7, "merged_every_5th": {k: v.flatten()[::5].clone() for k, v in ly_merged.items()}, "kinds": ly_kinds,
                "remaining": sorted(ly_remaining)}, os.path.join(GOLD, f"{name}_lycoris_merge.pt"))  # a fifth of every merged weight (fixture size)
    print(name, "lycoris merge:", ly_kinds)
    import hashlib
    km = "\n".join(f"{a}\t{b}" for a, b in sorted(key_map.items()))
    torch.save({"strength": strength, "merged": merged, "remaining": sorted(remaining), "key_map_sha256": hashlib.sha256(km.encode()).hexdigest(),
                "key_map_len": len(key_map)}, os.path.join(GOLD, f"{name}_lora_merge.pt"))
    print(name, "lora merge:", len(merged), "patched,", len(remaining), "unused keys, key map", len(key_map))


def gen_samples(name, cfg, net, b=2, hw=16):
    adm = cfg.get("adm_in_channels")
    c, uc = synth.synth_conditioning(b, cfg["context_dim"], adm, seed=1234)
    if adm:
        c, uc = ref_import.SdxlCond(c), ref_import.SdxlCond(uc)
    seeds = [1000 + i for i in range(b)]
    res = {"seeds": seeds, "hw": hw}
    for sampler, steps in (("Euler", 6), ("Euler a", 6), ("DPM++ 2M", 7)):
        trace = []
        lat, sigmas = ref_sample(net, cfg, c, uc, seeds, hw, steps, sampler, trace=trace)
        res[sampler] = {"steps": steps, "latent": lat, "sigmas": sigmas, "denoised0": trace[0], "denoised_last": trace[-1]}
        print(name, sampler, float(lat.std()))
    # cond_scale == 1 shortcut (sampling_function.py:295-298)
    lat, _ = ref_sample(net, cfg, c, uc, seeds, hw, 3, "Euler", cfg_scale=1.0)
    res["Euler_cfg1"] = {"steps": 3, "latent": lat}
    torch.save(res, os.path.join(GOLD, f"{name}_samples.pt"))


REF_SAMPLER_FN = {"Heun": "sample_heun", "DPM2": "sample_dpm_2", "DPM2 a": "sample_dpm_2_ancestral", "DPM++ 2S a": "sample_dpmpp_2s_ancestral",
                  "LMS": "sample_lms", "HeunPP2": "sample_heunpp2", "IPNDM": "sample_ipndm", "IPNDM_V": "sample_ipndm_v", "DEIS": "sample_deis",
                  "Restart": None}


def _ref_restart():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_ref_samplers_extra", os.path.join(ref_import.REFERENCE_ROOT, "modules", "sd_samplers_extra.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)  # imports only torch, tqdm, k_diffusion.sampling
    return mod.restart_sampler


def ref_sampler_sigmas(ref, linker, name, steps):
    """modules/sd_samplers_kdiffusion.py:81-134 for scheduler 'Automatic' with the per-sampler options of :14-34."""
    from oracle.sampling import SAMPLERS_EXTRA
    _, sched, _, _, discard = SAMPLERS_EXTRA[name]
    n = steps + (1 if discard else 0)
    if sched == "karras":
        sig = ref.kd_sampling.get_sigmas_karras(n=n, sigma_min=linker.sigmas[0].item(), sigma_max=linker.sigmas[-1].item(), device="cpu")
    else:
        sig = linker.get_sigmas(n)
    return torch.cat([sig[:-2], sig[-1:]]) if discard else sig


def toy_denoiser(x, sigma, **kw):
    """closed-form stand-in for CFGDenoiser (nonlinear in x and sigma so that solver order and stage placement matter)"""
    s = sigma.view(-1, 1, 1, 1)
    return x / (1 + s * s) + 0.3 * torch.tanh(x) * s / (1 + s)


def toy_inputs():
    g = torch.Generator().manual_seed(77)
    x0 = torch.randn(2, 4, 8, 8, generator=g)
    return x0, [torch.randn(2, 4, 8, 8, generator=g) for _ in range(64)]


def gen_samplers_toy():
    """The reference's sampler FUNCTIONS on the toy denoiser: pins oracle/sampling.py's restatements on CPU (no UNet in the way)."""
    ref = ref_import.load_reference()
    pred = ref_import.build_ref_predictor()
    linker = ref.kd_external.ForgeScheduleLinker(pred)
    restart = _ref_restart()
    x0, noises = toy_inputs()

    class Toy:
        inner_model = SimpleNamespace(predictor=pred)  # sample_dpm_2_ancestral checks for PredictionFlux (sampling.py:251)

        def __call__(self, x, sigma, **kw):
            return toy_denoiser(x, sigma)

    class Seq:
        def __init__(self):
            self.i = 0

        def __getattr__(self, item):
            if item == "randn_like":
                def f(x):
                    self.i += 1
                    return noises[self.i - 1]
                return f
            return getattr(torch, item)
    out = {}
    for name, fn_name in REF_SAMPLER_FN.items():
        for steps in (5, 12, 24, 40):
            sig = ref_sampler_sigmas(ref, linker, name, steps)
            h = Seq()
            ref.kd_sampling.torch = h
            try:
                fn = restart if fn_name is None else getattr(ref.kd_sampling, fn_name)
                lat = fn(Toy(), x0 * sig[0], sig, disable=True)
            finally:
                ref.kd_sampling.torch = torch
            out[(name, steps)] = {"sigmas": sig, "latent": lat, "draws": h.i}
    sig = linker.get_sigmas(12)
    from k_diffusion import deis
    out["deis_tab_3"] = [[float(c) for c in row] for row in deis.get_deis_coeff_list(sig, 3, deis_mode="tab")]
    out["deis_tab_4"] = [[float(c) for c in row] for row in deis.get_deis_coeff_list(sig, 4, deis_mode="tab")]
    out["deis_rhoab_3"] = [[float(c) for c in row] for row in deis.get_deis_coeff_list(sig, 3, deis_mode="rhoab")]
    out["deis_sigmas"] = sig
    torch.save(out, os.path.join(GOLD, "samplers_toy.pt"))
    print("samplers_toy", len(out))


def _ref_timesteps_impl():
    """modules/sd_samplers_timesteps_impl.py imported from the reference; its `modules.*` imports are satisfied by the real
    uni_pc.py / torch_utils.py files and a bare stand-in for modules.shared."""
    import importlib.util
    import types

    def load(name, *parts):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ref_import.REFERENCE_ROOT, *parts))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    ref_import.load_reference()
    names = ["modules", "modules.shared", "modules.models", "modules.models.diffusion", "modules.models.diffusion.uni_pc", "modules.torch_utils"]
    saved = {k: sys.modules.get(k) for k in names}
    try:
        for n in names[:5]:
            sys.modules[n] = types.ModuleType(n)
            sys.modules[n].__path__ = []
        sys.modules["modules.shared"].opts = SimpleNamespace(uni_pc_variant="bh1", uni_pc_skip_type="time_uniform", uni_pc_order=3,
                                                             uni_pc_lower_order_final=True)
        sys.modules["modules"].shared = sys.modules["modules.shared"]
        sys.modules["modules.models.diffusion.uni_pc"].uni_pc = load("_ref_uni_pc", "modules", "models", "diffusion", "uni_pc", "uni_pc.py")
        sys.modules["modules.torch_utils"] = load("_ref_torch_utils", "modules", "torch_utils.py")
        impl = load("_ref_timesteps_impl", "modules", "sd_samplers_timesteps_impl.py")
        return impl
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


class RefEpsDenoiser:
    """CFGDenoiser in classic_ddim_eps_estimation mode (modules/sd_samplers_cfg_denoiser.py:163-169, 201-202, 224-226 restated -- that
    module needs gradio) around `inner(x, sigma, **extra)` -> denoised [, cond_pred, uncond_pred in inner.last]."""

    def __init__(self, inner, alphas_cumprod):
        self.inner = inner
        self.inner_model = SimpleNamespace(inner_model=SimpleNamespace(alphas_cumprod=alphas_cumprod))
        self.need_last_noise_uncond, self.last_noise_uncond = False, None

    def __call__(self, x, t, **extra):
        acd = self.inner_model.inner_model.alphas_cumprod
        fake_sigmas = ((1 - acd) / acd) ** 0.5
        sigma = fake_sigmas[t.round().long().clip(0, int(fake_sigmas.shape[0]))]
        x = x * ((sigma ** 2.0 + 1.0) ** 0.5)[:, None, None, None]
        denoised = self.inner(x, sigma, **extra)
        if self.need_last_noise_uncond:
            self.last_noise_uncond = (x - self.inner.last[2]) / sigma[:, None, None, None]
        return (x - denoised) / sigma[:, None, None, None]


def _ref_lcm_schedule(ref, pred):
    """LCMCompVisDenoiser's schedule methods (modules/sd_samplers_lcm.py:10-49): that module imports modules.shared, so the class
    body is re-assembled here from the reference's own DiscreteEpsDDPMDenoiser (k_diffusion/external.py:76-133)."""
    base = ref.kd_external.DiscreteEpsDDPMDenoiser
    skip = 1000 // 50
    acd = 1.0 / (pred.sigmas ** 2.0 + 1.0)
    valid = torch.zeros(50, dtype=torch.float32)
    for x in range(50):
        valid[50 - 1 - x] = acd[1000 - 1 - x * skip]
    sched = base(None, valid, quantize=None)

    def sigma_to_t(sigma):
        d = sigma.log() - sched.log_sigmas[:, None]
        return d.abs().argmin(dim=0).view(sigma.shape) * skip + (skip - 1)

    def t_to_sigma(timestep):
        t = torch.clamp(((timestep - (skip - 1)) / skip).float(), min=0, max=(len(sched.sigmas) - 1))
        return base.t_to_sigma(sched, t)

    def get_sigmas(n):
        t = torch.linspace(sigma_to_t(sched.sigma_max), sigma_to_t(sched.sigma_min), n)
        return ref.kd_sampling.append_zero(t_to_sigma(t))
    return sched, get_sigmas


def gen_samples_more(name, cfg, net, b=2, hw=16):
    """DDIM / DDIM CFG++ / PLMS (reference functions, eps-mode denoiser), LCM and DDPM (reference loop functions) through the
    reference UNet + sampling_function; plus the same on the toy denoiser for the CPU pin of oracle/sampling.py."""
    import importlib.util
    ref = ref_import.load_reference()
    impl = _ref_timesteps_impl()
    spec = importlib.util.spec_from_file_location("_ref_kd_extra", os.path.join(ref_import.REFERENCE_ROOT, "backend", "modules", "k_diffusion_extra.py"))
    kd_extra = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kd_extra)
    pred = ref_import.build_ref_predictor()
    acd = 1.0 / (pred.sigmas ** 2.0 + 1.0)
    lcm_sched, lcm_get_sigmas = _ref_lcm_schedule(ref, pred)

    def ref_sample_lcm(model, x, sigmas, extra_args=None, noise_sampler=None):  # sd_samplers_lcm.py:69-83 needs modules.shared to import
        s_in = x.new_ones([x.shape[0]])
        for i in range(len(sigmas) - 1):
            denoised = model(x, sigmas[i] * s_in, **(extra_args or {}))
            x = denoised
            if sigmas[i + 1] > 0:
                x = x + sigmas[i + 1] * ref.kd_sampling.torch.randn_like(x)
        return x

    c, uc = synth.synth_conditioning(b, cfg["context_dim"], cfg.get("adm_in_channels"), seed=1234)
    seeds = [1000 + i for i in range(b)]
    extra = {"cond": c, "uncond": uc, "cond_scale": 7.0, "s_min_uncond": 0.0, "image_cond": None}
    x0, noises = toy_inputs()
    res = {"seeds": seeds, "hw": hw, "lcm_sigmas_table": lcm_sched.sigmas.clone(),
           "lcm_get_sigmas": {n: lcm_get_sigmas(n) for n in (1, 4, 8, 50)}}
    steps = 6

    class Seq:
        def __init__(self):
            self.i = 0

        def __getattr__(self, item):
            if item == "randn_like":
                def f(x):
                    self.i += 1
                    return noises[self.i - 1]
                return f
            return getattr(torch, item)

    class ToyInner:
        last = None

        def __call__(self, x, sigma, **kw):
            d = toy_denoiser(x, sigma)
            self.last = (d, d, toy_denoiser(0.5 * x, sigma))
            return d
    timesteps = torch.clip(torch.asarray(list(range(0, 1000, 1000 // steps))) + 1, 0, 999)
    cases = {"DDIM": (impl.ddim, {"eta": 0.0}), "DDIM eta": (impl.ddim, {"eta": 0.7}), "DDIM CFG++": (impl.ddim_cfgpp, {"eta": 0.0}),
             "PLMS": (impl.plms, {})}
    for label, (fn, kw) in cases.items():
        # toy
        h = Seq()
        impl.k_diffusion.sampling.torch = h
        try:
            toy = fn(RefEpsDenoiser(ToyInner(), acd), x0.clone(), timesteps, disable=True, **kw)
        finally:
            impl.k_diffusion.sampling.torch = torch
        # real stack
        den = ref_import.RefDenoiser(net, pred, seeds)
        rng = ImageRNG((cfg["in_channels"], hw, hw), seeds, "CPU")
        x = rng.next()
        impl.k_diffusion.sampling.torch = _Hijack(rng)
        ref.sampling_function.sampling_prepare(den.patcher, x=x)
        try:
            lat = fn(RefEpsDenoiser(den, acd), x, timesteps, extra_args=extra, disable=True, **kw)
        finally:
            impl.k_diffusion.sampling.torch = torch
            ref.sampling_function.sampling_cleanup(den.patcher)
        res[label] = {"steps": steps, "timesteps": timesteps, "latent": lat, "toy": toy, "toy_draws": h.i, **kw}
        print(name, label, float(lat.std()), float(toy.std()))
    for label, fn, sig in (("LCM", ref_sample_lcm, lcm_get_sigmas(steps)),
                           ("DDPM", kd_extra.sample_ddpm, ref.kd_external.ForgeScheduleLinker(pred).get_sigmas(steps))):
        h = Seq()
        ref.kd_sampling.torch = h
        kd_extra.torch = h
        try:
            toy = fn(lambda x, s, **kw: toy_denoiser(x, s), x0 * sig[0], sig)
        finally:
            ref.kd_sampling.torch = torch
            kd_extra.torch = torch
        den = ref_import.RefDenoiser(net, pred, seeds)
        rng = ImageRNG((cfg["in_channels"], hw, hw), seeds, "CPU")
        x = pred.noise_scaling(sig[0], rng.next(), torch.zeros(b, cfg["in_channels"], hw, hw), max_denoise=False)
        hj = _Hijack(rng)
        ref.kd_sampling.torch = hj
        kd_extra.torch = hj
        ref.sampling_function.sampling_prepare(den.patcher, x=x)
        try:
            lat = fn(den, x, sig, extra_args=extra)
        finally:
            ref.kd_sampling.torch = torch
            kd_extra.torch = torch
            ref.sampling_function.sampling_cleanup(den.patcher)
        res[label] = {"steps": steps, "sigmas": sig, "latent": lat, "toy": toy, "toy_draws": h.i}
        print(name, label, float(lat.std()), float(toy.std()))
    torch.save(res, os.path.join(GOLD, f"{name}_samples_more.pt"))


def gen_unipc(name, cfg, net, b=2, hw=16):
    """The reference's `unipc()` wrapper + UniPC / NoiseScheduleVP classes, on the toy denoiser (all variants / skip types / orders)
    and through the reference UNet + sampling_function (default options)."""
    ref = ref_import.load_reference()
    impl = _ref_timesteps_impl()
    pred = ref_import.build_ref_predictor()
    acd = 1.0 / (pred.sigmas ** 2.0 + 1.0)
    x0, _ = toy_inputs()

    class ToyInner:
        last = None

        def __call__(self, x, sigma, **kw):
            return toy_denoiser(x, sigma)
    res = {"seeds": [1000 + i for i in range(b)], "hw": hw, "toy": {}}
    shared_opts = impl.shared.opts
    for variant in ("bh1", "bh2", "vary_coeff"):
        for skip in ("time_uniform", "time_quadratic", "logSNR"):
            for order, lof, steps in ((3, True, 8), (2, True, 5), (1, True, 4), (3, False, 8), (2, False, 6), (6, True, 9)):  # order 4 without lower_order_final is unstable in fp32 (3e-4 vs fp64 in the reference itself)
                shared_opts.uni_pc_variant, shared_opts.uni_pc_skip_type = variant, skip
                shared_opts.uni_pc_order, shared_opts.uni_pc_lower_order_final = order, lof
                xin = x0[:1].clone() if variant == "vary_coeff" else x0.clone()  # the reference's vary_coeff path only runs at batch 1
                ts = torch.clip(torch.asarray(list(range(0, 1000, 1000 // steps))) + 1, 0, 999)
                out = impl.unipc(RefEpsDenoiser(ToyInner(), acd), xin, ts, extra_args={}, callback=lambda d: None, disable=True)
                res["toy"][(variant, skip, order, lof, steps)] = out
    shared_opts.uni_pc_variant, shared_opts.uni_pc_skip_type, shared_opts.uni_pc_order, shared_opts.uni_pc_lower_order_final = "bh1", "time_uniform", 3, True
    ts = torch.clip(torch.asarray(list(range(0, 1000, 1000 // 6))) + 1, 0, 999)
    res["toy_img2img"] = impl.unipc(RefEpsDenoiser(ToyInner(), acd), x0.clone(), ts[:4], extra_args={}, callback=lambda d: None, disable=True, is_img2img=True)
    c, uc = synth.synth_conditioning(b, cfg["context_dim"], cfg.get("adm_in_channels"), seed=1234)
    extra = {"cond": c, "uncond": uc, "cond_scale": 7.0, "s_min_uncond": 0.0, "image_cond": None}
    for steps in (6, 9):
        den = ref_import.RefDenoiser(net, pred, res["seeds"])
        x = ImageRNG((cfg["in_channels"], hw, hw), res["seeds"], "CPU").next()
        ts = torch.clip(torch.asarray(list(range(0, 1000, 1000 // steps))) + 1, 0, 999)
        ref.sampling_function.sampling_prepare(den.patcher, x=x)
        try:
            lat = impl.unipc(RefEpsDenoiser(den, acd), x, ts, extra_args=extra, callback=lambda d: None, disable=True)
        finally:
            ref.sampling_function.sampling_cleanup(den.patcher)
        res[steps] = {"latent": lat, "timesteps": ts}
        print(name, "UniPC", steps, float(lat.std()))
    torch.save(res, os.path.join(GOLD, f"{name}_samples_unipc.pt"))


class ListNoiseSampler:
    """noise_sampler(sigma, sigma_next) stand-in for BrownianTreeNoiseSampler: hands out a fixed list (torchsde is not in this image, and
    its stream could not be reproduced natively anyway); records the (sigma, sigma_next) it was asked for."""

    def __init__(self, noises):
        self.noises, self.i, self.asked = noises, 0, []

    def __call__(self, sigma, sigma_next):
        self.asked.append((float(sigma), float(sigma_next)))
        self.i += 1
        return self.noises[self.i - 1]


def gen_samplers_sde():
    """DPM++ SDE / 2M SDE (midpoint, heun) / 3M SDE with an injected noise sampler, DPM fast and DPM adaptive: the reference's functions
    on the toy denoiser (CPU pin of oracle + product host algebra) and through the reference UNet + sampling_function."""
    ref = ref_import.load_reference()
    kd = ref.kd_sampling
    pred = ref_import.build_ref_predictor()
    linker = ref.kd_external.ForgeScheduleLinker(pred)
    x0, noises = toy_inputs()
    smin, smax = pred.sigmas[0].item(), pred.sigmas[-1].item()

    class Toy:
        inner_model = SimpleNamespace(predictor=pred)

        def __call__(self, x, sigma, **kw):
            return toy_denoiser(x, sigma)

    class Seq:
        def __init__(self):
            self.i = 0

        def __getattr__(self, item):
            if item == "randn_like":
                def f(x):
                    self.i += 1
                    return noises[self.i - 1]
                return f
            return getattr(torch, item)
    out = {}
    karras = lambda n: kd.get_sigmas_karras(n, smin, smax)
    expo = lambda n: kd.get_sigmas_exponential(n, smin, smax)
    disc = lambda sig: torch.cat([sig[:-2], sig[-1:]])
    cases = {"DPM++ SDE": (kd.sample_dpmpp_sde, karras, {}), "DPM++ SDE eta0.5": (kd.sample_dpmpp_sde, karras, {"eta": 0.5, "s_noise": 0.9}),
             "DPM++ 2M SDE": (kd.sample_dpmpp_2m_sde, expo, {}), "DPM++ 2M SDE Heun": (kd.sample_dpmpp_2m_sde, expo, {"solver_type": "heun"}),
             "DPM++ 2M SDE eta0": (kd.sample_dpmpp_2m_sde, expo, {"eta": 0.0}),
             "DPM++ 3M SDE": (kd.sample_dpmpp_3m_sde, lambda n: disc(expo(n + 1)), {}), "DPM++ 3M SDE eta0": (kd.sample_dpmpp_3m_sde, lambda n: disc(expo(n + 1)), {"eta": 0.0})}
    for label, (fn, sched, kw) in cases.items():
        for steps in (5, 12):
            sig = sched(steps)
            ns = ListNoiseSampler(noises)
            lat = fn(Toy(), x0 * sig[0], sig, noise_sampler=ns, disable=True, **kw)
            out[(label, steps)] = {"sigmas": sig, "latent": lat, "asked": ns.asked, "kw": kw}
    for n in (5, 6, 7, 12):
        for eta in (0.0, 0.6):
            h = Seq()
            kd.torch = h
            try:
                lat = kd.sample_dpm_fast(Toy(), x0 * smax, smin, smax, n, disable=True, eta=eta)
            finally:
                kd.torch = torch
            out[("DPM fast", n, eta)] = {"latent": lat, "draws": h.i}
    for order in (2, 3):
        for eta in (0.0, 0.6):
            h = Seq()
            kd.torch = h
            try:
                lat, info = kd.sample_dpm_adaptive(Toy(), x0 * smax, smin, smax, disable=True, order=order, eta=eta, return_info=True)
            finally:
                kd.torch = torch
            out[("DPM adaptive", order, eta)] = {"latent": lat, "draws": h.i, "info": info}
            print("DPM adaptive", order, eta, info)
    # through the reference UNet stack (tiny_sd15): DPM fast 7 evaluations, DPM adaptive order 3 with a loose tolerance (few steps)
    cfg = synth.TINY_SD15_UNET_CONFIG
    net = ref_import.build_ref_unet(cfg, synth.synth_unet_state_dict(cfg, seed=0))
    b, hw = 2, 16
    c, uc = synth.synth_conditioning(b, cfg["context_dim"], None, seed=1234)
    seeds = [1000 + i for i in range(b)]
    extra = {"cond": c, "uncond": uc, "cond_scale": 7.0, "s_min_uncond": 0.0, "image_cond": None}
    stack = {"seeds": seeds, "hw": hw}

    def run(fn, *a, noise_list=None, **kw):
        den = ref_import.RefDenoiser(net, pred, seeds)
        rng = ImageRNG((cfg["in_channels"], hw, hw), seeds, "CPU")
        x = pred.noise_scaling(torch.tensor(smax), rng.next(), torch.zeros(b, cfg["in_channels"], hw, hw), max_denoise=False)
        kd.torch = _Hijack(rng)
        ref.sampling_function.sampling_prepare(den.patcher, x=x)
        try:
            return fn(den, x, *a, extra_args=extra, disable=True, **kw)
        finally:
            kd.torch = torch
            ref.sampling_function.sampling_cleanup(den.patcher)
    stack["DPM fast"] = {"n": 7, "latent": run(kd.sample_dpm_fast, smin, smax, 7)}
    lat, info = run(kd.sample_dpm_adaptive, smin, smax, rtol=0.5, atol=0.5, return_info=True)
    stack["DPM adaptive"] = {"rtol": 0.5, "atol": 0.5, "latent": lat, "info": info}
    print("stack DPM adaptive", info)
    g = torch.Generator().manual_seed(99)
    nz = [torch.randn(b, cfg["in_channels"], hw, hw, generator=g) for _ in range(16)]
    stack["noises_seed"] = 99
    for label, fn, sig, kw in (("DPM++ SDE", kd.sample_dpmpp_sde, karras(5), {}), ("DPM++ 2M SDE", kd.sample_dpmpp_2m_sde, expo(6), {}),
                               ("DPM++ 3M SDE", kd.sample_dpmpp_3m_sde, disc(expo(7)), {})):
        den = ref_import.RefDenoiser(net, pred, seeds)
        x = pred.noise_scaling(sig[0], ImageRNG((cfg["in_channels"], hw, hw), seeds, "CPU").next(), torch.zeros(b, cfg["in_channels"], hw, hw), max_denoise=False)
        ref.sampling_function.sampling_prepare(den.patcher, x=x)
        try:
            lat = fn(den, x, sig, extra_args=extra, disable=True, noise_sampler=ListNoiseSampler(nz), **kw)
        finally:
            ref.sampling_function.sampling_cleanup(den.patcher)
        stack[label] = {"sigmas": sig, "latent": lat}
        print("stack", label, float(lat.std()))
    out["stack"] = stack
    torch.save(out, os.path.join(GOLD, "samplers_sde_dpm.pt"))


def gen_unet_hooks(name, cfg, net, b=2, hw=16):
    """Reference UNet forward with the hook set of oracle/hooks_fixture.py (direct call), and a 3-step Euler run through the reference
    sampling_function with the same hooks installed on the patcher's model_options (exercises the per-call transformer_options keys)."""
    from oracle.hooks_fixture import build_hooks
    fx = torch.load(os.path.join(GOLD, f"{name}_unet_fwd.pt"))
    to, log = build_hooks()
    with torch.no_grad():
        eps = net(fx["x"], fx["t"], context=fx["ctx"], y=fx["y"], transformer_options=to)
    res = {"eps": eps, "log": list(log)}
    ref = ref_import.load_reference()
    pred = ref_import.build_ref_predictor()
    adm = cfg.get("adm_in_channels")
    c, uc = synth.synth_conditioning(b, cfg["context_dim"], adm, seed=1234)
    if adm:
        c, uc = ref_import.SdxlCond(c), ref_import.SdxlCond(uc)
    seeds = [1000 + i for i in range(b)]
    den = ref_import.RefDenoiser(net, pred, seeds)
    to2, log2 = build_hooks(use_call_keys=True)
    den.patcher.model_options["transformer_options"] = to2
    rng = ImageRNG((cfg["in_channels"], hw, hw), seeds, "CPU")
    x = rng.next()
    sigmas = den.inner_model.get_sigmas(3)
    x = pred.noise_scaling(sigmas[0], x, torch.zeros_like(x), max_denoise=False)
    ref.kd_sampling.torch = _Hijack(rng)
    ref.sampling_function.sampling_prepare(den.patcher, x=x)
    try:
        lat = ref.kd_sampling.sample_euler(den, x, sigmas, extra_args={"cond": c, "uncond": uc, "cond_scale": 7.0, "s_min_uncond": 0.0, "image_cond": None},
                                           disable=True)
    finally:
        ref.kd_sampling.torch = torch
        ref.sampling_function.sampling_cleanup(den.patcher)
    res["euler3"] = {"latent": lat, "seeds": seeds, "hw": hw}
    res["log_first_forward_of_run"] = log2[:len(log)]
    torch.save(res, os.path.join(GOLD, f"{name}_unet_hooks.pt"))
    print(name, "hooks:", len(log), "hook calls per forward; eps std", float(eps.std()), "euler3 std", float(lat.std()))


def gen_unet_module_hooks(name, cfg, net):
    """Reference UNet forward with the two module-typed hooks of oracle/hooks_fixture.py build_module_hooks (block_inner_modifiers,
    group_norm_wrapper): output and the sequence of hook calls (class names, layer indices, block lengths, block ids)."""
    from oracle.hooks_fixture import build_module_hooks
    fx = torch.load(os.path.join(GOLD, f"{name}_unet_fwd.pt"))
    to, log = build_module_hooks()
    with torch.no_grad():
        eps = net(fx["x"].clone(), fx["t"], context=fx["ctx"], y=fx["y"], transformer_options=to)
    torch.save({"eps": eps, "log": list(log)}, os.path.join(GOLD, f"{name}_unet_module_hooks.pt"))
    print(name, "module hooks:", len(log), "calls; eps std", float(eps.std()), "moved the output by", float((eps - fx["eps"]).abs().max() / fx["eps"].abs().max()))


def controlnet_case(cfg, b=2, hw=16):
    """Deterministic ControlNet inputs shared by the generator and the tests: hint images (one at the exact 8x size, one that needs the
    nearest-exact resize + centre crop), a mask, per-frame weights."""
    g = torch.Generator().manual_seed(2024)
    return {"hint_a": torch.rand(b, 3, hw * 8, hw * 8, generator=g), "hint_b": torch.rand(1, 3, hw * 6, hw * 10, generator=g),
            "mask": torch.rand(b, 1, 24, 24, generator=g), "frame": [0.6, 1.3][:b],
            "positive": {"output": [1.0, 0.9, 0.8, 1.1], "middle": [0.7]}, "negative": {"output": [0.5, 1.2], "middle": [1.4]}}


def sigma_weight(s):
    return (s / 14.6146) ** 0.5


def gen_controlnet(name, cfg, net, b=2, hw=16):
    """cldm.ControlNet forward (model level) and a 3-step Euler run through the reference sampling_function with a CHAIN of two reference
    patcher-level ControlNets (strength, start / end percent, global average pooling, all five advanced weightings)."""
    import importlib
    ref = ref_import.load_reference()
    cldm = importlib.import_module("backend.nn.cnets.cldm")
    pc = importlib.import_module("backend.patcher.controlnet")
    fx = torch.load(os.path.join(GOLD, f"{name}_unet_fwd.pt"))
    case = controlnet_case(cfg, b, hw)
    kw = {k: v for k, v in cfg.items() if k not in ("out_channels", "transformer_depth_output")}
    kw["transformer_depth"] = list(kw["transformer_depth"])
    models = []
    for seed in (6, 9):
        m = cldm.ControlNet(hint_channels=3, dtype=torch.float32, **kw)
        sd = synth.synth_controlnet_state_dict(cfg, seed=seed)
        missing, unexpected = m.load_state_dict(sd, strict=True), None
        m.eval()
        models.append(m)
    with torch.no_grad():
        outs = models[0](x=fx["x"], hint=case["hint_a"], timesteps=fx["t"], context=fx["ctx"], y=fx["y"])
    res = {"outs_every_4th_channel": [o[:, ::4].clone() for o in outs], "hw": hw}  # all residuals, a quarter of the channels (fixture size)
    pred = ref_import.build_ref_predictor()
    adm = cfg.get("adm_in_channels")
    c, uc = synth.synth_conditioning(b, cfg["context_dim"], adm, seed=1234)
    if adm:
        c, uc = ref_import.SdxlCond(c), ref_import.SdxlCond(uc)
    seeds = [1000 + i for i in range(b)]
    den = ref_import.RefDenoiser(net, pred, seeds)
    cn_a = pc.ControlNet(models[0], load_device=torch.device("cpu"))
    cn_b = pc.ControlNet(models[1], global_average_pooling=True, load_device=torch.device("cpu"))
    unet = pc.apply_controlnet_advanced(den.patcher, cn_a, case["hint_a"], 0.8, 0.0, 0.7, positive_advanced_weighting=case["positive"],
                                        negative_advanced_weighting=case["negative"], advanced_frame_weighting=case["frame"],
                                        advanced_sigma_weighting=sigma_weight, advanced_mask_weighting=case["mask"])
    unet = pc.apply_controlnet_advanced(unet, cn_b, case["hint_b"], 0.5, 0.2, 1.0)
    den.patcher = unet
    den.inner_model.inner_model.forge_objects.unet = unet
    rng = ImageRNG((cfg["in_channels"], hw, hw), seeds, "CPU")
    x = rng.next()
    sigmas = den.inner_model.get_sigmas(4)
    x = pred.noise_scaling(sigmas[0], x, torch.zeros_like(x), max_denoise=False)
    ref.kd_sampling.torch = _Hijack(rng)
    ref.sampling_function.sampling_prepare(unet, x=x)
    try:
        lat = ref.kd_sampling.sample_euler(den, x, sigmas, extra_args={"cond": c, "uncond": uc, "cond_scale": 7.0, "s_min_uncond": 0.0, "image_cond": None},
                                           disable=True)
    finally:
        ref.kd_sampling.torch = torch
        ref.sampling_function.sampling_cleanup(unet)
    res["euler4"] = {"latent": lat, "seeds": seeds, "sigmas": sigmas}
    # the same run without ControlNets, to show the chain matters
    res["euler4_plain_std"] = float(lat.std())
    torch.save(res, os.path.join(GOLD, f"{name}_controlnet.pt"))
    print(name, "controlnet:", len(outs), "residuals; euler4 std", float(lat.std()))


def gen_control_lora(name, cfg, b=2, hw=16):
    """patcher.controlnet.ControlLora (:420-474): the control model is built in pre_run from the UNet's own weights plus the file's low-rank
    pairs.  Fixture: the residuals of one model call (a quarter of the channels) and a 4-step Euler run through the reference stack."""
    import importlib
    ref = ref_import.load_reference()
    pc = importlib.import_module("backend.patcher.controlnet")
    net = ref_import.build_ref_unet(cfg, synth.synth_unet_state_dict(cfg, seed=0))
    # diffusers' ConfigMixin (a stub here) records the constructor arguments as `.config`; pre_run reads them back (:428)
    net.config = {k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in cfg.items()}
    fx = torch.load(os.path.join(GOLD, f"{name}_unet_fwd.pt"))
    case = controlnet_case(cfg, b, hw)
    pred = ref_import.build_ref_predictor()
    adm = cfg.get("adm_in_channels")
    c, uc = synth.synth_conditioning(b, cfg["context_dim"], adm, seed=1234)
    if adm:
        c, uc = ref_import.SdxlCond(c), ref_import.SdxlCond(uc)
    seeds = [1000 + i for i in range(b)]
    den = ref_import.RefDenoiser(net, pred, seeds)
    cl = pc.ControlLora(synth.synth_control_lora_state_dict(cfg))
    unet = pc.apply_controlnet_advanced(den.patcher, cl, case["hint_a"], 0.9, 0.0, 1.0)
    den.patcher = unet
    den.inner_model.inner_model.forge_objects.unet = unet
    rng = ImageRNG((cfg["in_channels"], hw, hw), seeds, "CPU")
    x = rng.next()
    sigmas = den.inner_model.get_sigmas(4)
    x = pred.noise_scaling(sigmas[0], x, torch.zeros_like(x), max_denoise=False)
    ref.kd_sampling.torch = _Hijack(rng)
    ref.sampling_function.sampling_prepare(unet, x=x)
    try:
        linked = unet.controlnet_linked_list
        with torch.no_grad():
            outs = linked.control_model(x=fx["x"], hint=case["hint_a"], timesteps=fx["t"], context=fx["ctx"], y=fx["y"])
        lat = ref.kd_sampling.sample_euler(den, x, sigmas, extra_args={"cond": c, "uncond": uc, "cond_scale": 7.0, "s_min_uncond": 0.0, "image_cond": None},
                                           disable=True)
    finally:
        ref.kd_sampling.torch = torch
        ref.sampling_function.sampling_cleanup(unet)
    res = {"outs_every_8th_channel": [o[:, ::8].clone() for o in outs], "hw": hw, "euler4": {"latent": lat, "seeds": seeds, "sigmas": sigmas}}
    torch.save(res, os.path.join(GOLD, f"{name}_control_lora.pt"))
    print(name, "control-lora:", len(outs), "residuals, std", float(outs[3].std()), "; euler4 std", float(lat.std()))


def gen_prediction_types(name, cfg, net, b=2, hw=16):
    """Prediction(prediction_type='v_prediction' | 'edm') and the non-default beta schedules / zero-terminal-SNR rescale of
    backend/modules/k_prediction.py: sigma tables, calculate_denoised, and a 4-step Euler run through the reference stack per type."""
    ref = ref_import.load_reference()
    kp = ref.k_prediction
    out = {"hw": hw, "seeds": [1000 + i for i in range(b)]}
    for sched, (ls, le) in (("linear", (0.00085, 0.012)), ("cosine", (0.00085, 0.012)), ("sqrt_linear", (0.0001, 0.02)), ("sqrt", (0.0001, 0.0004))):
        p = kp.Prediction(prediction_type="epsilon", beta_schedule=sched, linear_start=ls, linear_end=le, timesteps=1000)
        out[("sigmas", sched)] = p.sigmas.clone()
    base = kp.Prediction(prediction_type="v_prediction", beta_schedule="linear", linear_start=0.00085, linear_end=0.012, timesteps=1000)
    out["ztsnr_sigmas"] = kp.rescale_zero_terminal_snr_sigmas(base.sigmas.clone())
    out["percent_to_sigma"] = {pc: base.percent_to_sigma(pc) for pc in (0.0, 0.1, 0.37, 0.7, 1.0)}
    g = torch.Generator().manual_seed(5)
    x, mo, sg = torch.randn(3, 4, 8, 8, generator=g), torch.randn(3, 4, 8, 8, generator=g), torch.tensor([14.6, 1.3, 0.05])
    c, uc = synth.synth_conditioning(b, cfg["context_dim"], cfg.get("adm_in_channels"), seed=1234)
    for ptype in ("v_prediction", "edm"):
        pred = kp.Prediction(prediction_type=ptype, beta_schedule="linear", linear_start=0.00085, linear_end=0.012, timesteps=1000)
        out[("denoised", ptype)] = pred.calculate_denoised(sg, mo, x)
        den = ref_import.RefDenoiser(net, pred, out["seeds"])
        rng = ImageRNG((cfg["in_channels"], hw, hw), out["seeds"], "CPU")
        xx = rng.next()
        sigmas = den.inner_model.get_sigmas(4)
        xx = pred.noise_scaling(sigmas[0], xx, torch.zeros_like(xx), max_denoise=False)
        ref.kd_sampling.torch = _Hijack(rng)
        ref.sampling_function.sampling_prepare(den.patcher, x=xx)
        try:
            lat = ref.kd_sampling.sample_euler(den, xx, sigmas, extra_args={"cond": c, "uncond": uc, "cond_scale": 7.0, "s_min_uncond": 0.0,
                                                                              "image_cond": None}, disable=True)
        finally:
            ref.kd_sampling.torch = torch
            ref.sampling_function.sampling_cleanup(den.patcher)
        out[("euler4", ptype)] = lat
        print(name, ptype, float(lat.std()))
    out["kat"] = {"x": x, "model_output": mo, "sigma": sg}
    torch.save(out, os.path.join(GOLD, f"{name}_prediction_types.pt"))


def cfg_hooks_fixture():
    """Deterministic model_options hooks shared by the generator (reference sampling_function) and the tests (native sampling_function)."""
    def sampler_cfg_function(args):  # a 'rescale'-style cfg function: returns the noise prediction (x - denoised)
        return args["uncond"] + (args["cond"] - args["uncond"]) * args["cond_scale"] * 0.9

    def post_cfg(args):
        return args["denoised"] * 0.99 + 0.01 * args["cond_denoised"]

    def pre_cfg(model, cond, uncond, x, timestep, model_options):
        return model, cond, uncond, x * 1.0, timestep, model_options

    def wrapper(apply_model, params):
        assert params["cond_or_uncond"][0] == 1 and params["input"].shape[0] == params["timestep"].shape[0]
        return apply_model(params["input"], params["timestep"], **params["c"]) * 1.01
    return {"sampler_cfg_function": sampler_cfg_function, "sampler_post_cfg_function": [post_cfg], "sampler_pre_cfg_function": [pre_cfg],
            "model_function_wrapper": wrapper}


def multicond_case(cfg, b=2, seed=77):
    """AND-composed prompts as tensors: image i has parts (2i, w=1.0) and (2i+1, w=0.6); plus a per-image prompt-editing schedule."""
    adm = cfg.get("adm_in_channels")
    c4, uc = synth.synth_conditioning(2 * b, cfg["context_dim"], adm, seed=seed)
    comp = [[(2 * i, 1.0), (2 * i + 1, 0.6)] for i in range(b)]
    take = lambda t, n: ({k: v[:n] for k, v in t.items()} if isinstance(t, dict) else t[:n])
    return c4, take(uc, b), comp


def gen_cfg_paths(name, cfg, net, b=2, hw=16):
    """sampling_function's general paths through the reference: (1) AND-composed prompts (two weighted conds per image -> edit strength
    1.6), (2) sampler_pre_cfg / sampler_cfg / sampler_post_cfg functions, (3) model_function_wrapper; 3 Euler steps each."""
    ref = ref_import.load_reference()
    pred = ref_import.build_ref_predictor()
    seeds = [1000 + i for i in range(b)]
    c4, uc, comp = multicond_case(cfg, b)
    c1, _ = synth.synth_conditioning(b, cfg["context_dim"], cfg.get("adm_in_channels"), seed=1234)
    wrap = (lambda t: ref_import.SdxlCond(t)) if cfg.get("adm_in_channels") else (lambda t: t)
    hooks = cfg_hooks_fixture()
    res = {"seeds": seeds, "hw": hw}

    class Den(ref_import.RefDenoiser):
        composition = None

        def __call__(self, x, sigma, uncond, cond, cond_scale, s_min_uncond=0.0, image_cond=None):
            params = SimpleNamespace(x=x, sigma=sigma, text_cond=cond, text_uncond=uncond, image_cond=image_cond)
            comp_ = self.composition or [[(i, 1.0)] for i in range(x.shape[0])]
            return ref.sampling_function.sampling_function(self, denoiser_params=params, cond_scale=cond_scale, cond_composition=comp_)[0]

    def run(cond, composition, options):
        den = Den(net, pred, seeds)
        den.composition = composition
        den.patcher.model_options.update(options)
        rng = ImageRNG((cfg["in_channels"], hw, hw), seeds, "CPU")
        x = rng.next()
        sigmas = den.inner_model.get_sigmas(3)
        x = pred.noise_scaling(sigmas[0], x, torch.zeros_like(x), max_denoise=False)
        ref.kd_sampling.torch = _Hijack(rng)
        ref.sampling_function.sampling_prepare(den.patcher, x=x)
        try:
            return ref.kd_sampling.sample_euler(den, x, sigmas, extra_args={"cond": wrap(cond), "uncond": wrap(uc), "cond_scale": 5.0, "s_min_uncond": 0.0,
                                                                            "image_cond": None}, disable=True)
        finally:
            ref.kd_sampling.torch = torch
            ref.sampling_function.sampling_cleanup(den.patcher)
    res["and_composed"] = run(c4, comp, {})
    res["cfg_functions"] = run(c1, None, {k: hooks[k] for k in ("sampler_cfg_function", "sampler_post_cfg_function", "sampler_pre_cfg_function")})
    res["model_function_wrapper"] = run(c1, None, {"model_function_wrapper": hooks["model_function_wrapper"]})
    res["plain"] = run(c1, None, {})
    for k in ("and_composed", "cfg_functions", "model_function_wrapper", "plain"):
        print(name, k, float(res[k].std()))
    torch.save(res, os.path.join(GOLD, f"{name}_cfg_paths.pt"))


def inpaint_case(b=2, hw=16, seed=31):
    g = torch.Generator().manual_seed(seed)
    mask = (torch.rand(b, 1, hw, hw, generator=g) > 0.5).float()
    return torch.cat([mask, torch.randn(b, 4, hw, hw, generator=g)], dim=1)  # [mask | masked-image latent]


def gen_inpaint_model(b=2, hw=16):
    """Inpainting UNet (in_channels 9) in the reference: forward on the concatenated input, KModel-level c_concat via sampling_function's
    image_cond, 3 Euler steps."""
    ref = ref_import.load_reference()
    cfg = synth.TINY_SD15_INPAINT_UNET_CONFIG
    net = ref_import.build_ref_unet(cfg, synth.synth_unet_state_dict(cfg, seed=0))
    fx = torch.load(os.path.join(GOLD, "tiny_sd15_unet_fwd.pt"))
    ic = inpaint_case(b, hw)
    with torch.no_grad():
        eps = net(torch.cat([fx["x"], ic], dim=1), fx["t"], context=fx["ctx"], y=None)
    pred = ref_import.build_ref_predictor()
    c, uc = synth.synth_conditioning(b, cfg["context_dim"], None, seed=1234)
    seeds = [1000 + i for i in range(b)]
    den = ref_import.RefDenoiser(net, pred, seeds)
    rng = ImageRNG((4, hw, hw), seeds, "CPU")
    x = rng.next()
    sigmas = den.inner_model.get_sigmas(3)
    x = pred.noise_scaling(sigmas[0], x, torch.zeros_like(x), max_denoise=False)
    ref.kd_sampling.torch = _Hijack(rng)
    ref.sampling_function.sampling_prepare(den.patcher, x=x)
    try:
        lat = ref.kd_sampling.sample_euler(den, x, sigmas, extra_args={"cond": c, "uncond": uc, "cond_scale": 7.0, "s_min_uncond": 0.0, "image_cond": ic},
                                           disable=True)
    finally:
        ref.kd_sampling.torch = torch
        ref.sampling_function.sampling_cleanup(den.patcher)
    torch.save({"eps": eps, "euler3": lat, "seeds": seeds, "hw": hw}, os.path.join(GOLD, "tiny_sd15_inpaint_model.pt"))
    print("inpaint model: eps std", float(eps.std()), "euler3 std", float(lat.std()))


ADAPTER_VARIANTS = {  # name -> Adapter kwargs (cin = 3 or 1 image channels x unshuffle^2)
    "sd15_k1_pool": dict(channels=[64, 128, 256, 256], nums_rb=2, cin=192, ksize=1, sk=True, use_conv=False, xl=False),
    "sd15_k3_conv": dict(channels=[64, 128, 256, 256], nums_rb=2, cin=64, ksize=3, sk=True, use_conv=True, xl=False),
    "sdxl": dict(channels=[64, 128, 256, 256], nums_rb=2, cin=768, ksize=1, sk=True, use_conv=False, xl=True),
}


def adapter_hint(vname, b=2, hw=16):
    """The hint image of a variant, regenerated from its seed (not stored: 400 KB each)."""
    kw = ADAPTER_VARIANTS[vname]
    g = torch.Generator().manual_seed(55 + list(ADAPTER_VARIANTS).index(vname))
    return torch.rand(b, kw["cin"] // ((16 if kw["xl"] else 8) ** 2), hw * 8, hw * 8, generator=g)


def gen_t2i_adapter(b=2, hw=16):
    """The reference's Adapter (three checkpoint layouts) and a 3-step Euler run of an SD1.5-shaped UNet through the reference sampling_function
    with the reference's patcher-level T2IAdapter attached (strength 0.9, active for the first 60 % of the schedule)."""
    import importlib
    ref = ref_import.load_reference()
    t2i = importlib.import_module("backend.nn.cnets.t2i_adapter")
    pc = importlib.import_module("backend.patcher.controlnet")
    g = torch.Generator().manual_seed(55)
    res = {"hw": hw, "seeds": [1000 + i for i in range(b)], "features": {}}
    models = {}
    for vname, kw in ADAPTER_VARIANTS.items():
        m = t2i.Adapter(**kw)
        sd = synth.synth_t2i_adapter_state_dict(**kw)
        assert set(sd) == set(m.state_dict()) and all(tuple(sd[k].shape) == tuple(v.shape) for k, v in m.state_dict().items()), vname
        m.load_state_dict(sd)
        m.eval()
        models[vname] = m
        hint = adapter_hint(vname, b, hw)
        with torch.no_grad():
            feats = m(hint)
        res["features"][vname] = {"layout": [None if f is None else tuple(f.shape) for f in feats],
                                  "values_every_4th_channel": [f[:, ::4].clone() for f in feats if f is not None]}
    cfg = synth.MINI_SD15_UNET_CONFIG
    net = ref_import.build_ref_unet(cfg, synth.synth_unet_state_dict(cfg, seed=0))
    pred = ref_import.build_ref_predictor()
    c, uc = synth.synth_conditioning(b, cfg["context_dim"], None, seed=1234)
    den = ref_import.RefDenoiser(net, pred, res["seeds"])
    ad = pc.T2IAdapter(models["sd15_k1_pool"], 3, device=torch.device("cpu"))
    hint = adapter_hint("sd15_k1_pool", b, hw)
    unet = den.patcher.clone()
    unet.add_patched_controlnet(ad.copy().set_cond_hint(hint, 0.9, (0.0, 0.6)))
    den.patcher = unet
    den.inner_model.inner_model.forge_objects.unet = unet
    rng = ImageRNG((4, hw, hw), res["seeds"], "CPU")
    x = rng.next()
    sigmas = den.inner_model.get_sigmas(3)
    x = pred.noise_scaling(sigmas[0], x, torch.zeros_like(x), max_denoise=False)
    ref.kd_sampling.torch = _Hijack(rng)
    ref.sampling_function.sampling_prepare(unet, x=x)
    try:
        lat = ref.kd_sampling.sample_euler(den, x, sigmas, extra_args={"cond": c, "uncond": uc, "cond_scale": 7.0, "s_min_uncond": 0.0, "image_cond": None},
                                           disable=True)
    finally:
        ref.kd_sampling.torch = torch
        ref.sampling_function.sampling_cleanup(unet)
    res["euler3"] = lat
    torch.save(res, os.path.join(GOLD, "mini_sd15_t2i_adapter.pt"))
    print("t2i adapter:", {k: len(v["values_every_4th_channel"]) for k, v in res["features"].items()}, "euler3 std", float(lat.std()))


ADAPTER_LIGHT_KW = dict(channels=[64, 128, 256, 256], nums_rb=2, cin=192)   # quarter widths 16 / 32 / 64 / 64: below the GEMM granule


def adapter_light_hint(b=2, hw=16):
    return torch.rand(b, 3, hw * 8, hw * 8, generator=torch.Generator().manual_seed(77))


def gen_adapter_light():
    """The reference's Adapter_light (the colour adapter's network) on synthetic weights: feature maps, every 4th channel."""
    import importlib
    ref_import.load_reference()
    t2i = importlib.import_module("backend.nn.cnets.t2i_adapter")
    m = t2i.Adapter_light(**ADAPTER_LIGHT_KW)
    sd = synth.synth_t2i_adapter_light_state_dict(**ADAPTER_LIGHT_KW)
    assert set(sd) == set(m.state_dict()) and all(tuple(sd[k].shape) == tuple(v.shape) for k, v in m.state_dict().items())
    m.load_state_dict(sd)
    m.eval()
    with torch.no_grad():
        feats = m(adapter_light_hint())
    res = {"layout": [None if f is None else tuple(f.shape) for f in feats], "values_every_4th_channel": [f[:, ::4].clone() for f in feats if f is not None],
           "input_channels": m.input_channels, "unshuffle_amount": m.unshuffle_amount}
    torch.save(res, os.path.join(GOLD, "mini_adapter_light.pt"))
    print("adapter_light:", res["layout"], [float(f.std()) for f in res["values_every_4th_channel"]])


class RegionalToyModel:
    """apply_model stand-in for the regional-conditioning fixtures: depends on x, sigma, the text conditioning AND on the size of the
    rectangle it is evaluated on (a ramp over the crop's width and height), so a wrong crop, weight or placement shows."""

    def memory_required(self, shape):
        return 0

    def apply_model(self, x, t, c_crossattn=None, **kw):
        ramp = torch.linspace(0, 1, x.shape[3]).view(1, 1, 1, -1) + torch.linspace(0, 2, x.shape[2]).view(1, 1, -1, 1)
        return x * (0.3 + 0.01 * t.view(-1, 1, 1, 1)) + c_crossattn.mean(dim=(1, 2)).view(-1, 1, 1, 1) + 0.05 * ramp.to(x)


def regional_case(b=2, c=2, hh=20, ww=24):
    """Seeded inputs shared by the generator and the test: latent, three cond entries (full frame with a mask, an area touching the top-left
    corner, an interior area with a sigma window) and two uncond entries (plain, and one that is only active at high sigma)."""
    g = torch.Generator().manual_seed(77)
    x = torch.randn(b, c, hh, ww, generator=g)
    ctx = [torch.randn(b, 5, 8, generator=g) for _ in range(5)]
    mask = torch.rand(1, hh, ww, generator=g)
    cond = [dict(ctx=0, mask=mask, mask_strength=0.7, strength=1.0), dict(ctx=1, area=(16, 12, 0, 0), strength=0.8),
            dict(ctx=2, area=(10, 14, 9, 6), strength=1.3, timestep_start=8.0, timestep_end=1.0)]
    uncond = [dict(ctx=3), dict(ctx=4, timestep_end=4.0, strength=0.5)]
    return x, ctx, cond, uncond, [11.0, 5.0, 0.7]     # sigmas: third cond inactive / all active / third cond and second uncond inactive


def gen_regional():
    """The reference's calc_cond_uncond_batch on regional / time-ranged entries (get_area_and_mult :17-73, accumulation :154-288)."""
    import importlib
    ref = ref_import.load_reference()
    rc = importlib.import_module("backend.sampling.condition")
    x, ctx, cond, uncond, sigmas = regional_case()

    def build(entries):
        out = []
        for e in entries:
            d = {k: v for k, v in e.items() if k != "ctx"}
            d["model_conds"] = {"c_crossattn": rc.ConditionCrossAttn(ctx[e["ctx"]])}
            out.append(d)
        return out
    res = {}
    model = RegionalToyModel()
    for s in sigmas:
        t = torch.full((x.shape[0],), s)
        c_out, u_out = ref.sampling_function.calc_cond_uncond_batch(model, build(cond), build(uncond), x, t, {})
        res[s] = (c_out, u_out)
    torch.save(res, os.path.join(GOLD, "regional_conds.pt"))
    print("regional:", {s: (float(v[0].std()), float(v[1].std())) for s, v in res.items()})


RNG_VARIATION_CASES = {   # name -> ImageRNG kwargs (modules/rng.py:113-177); latent shape (4, 6, 8), three images
    "subseed": dict(subseeds=[100, 101], subseed_strength=0.35),                                    # fewer subseeds than images: the rest use 0 (:136)
    "resize_grow": dict(seed_resize_from_h=32, seed_resize_from_w=48),                              # 4 x 6 noise pasted into the centre of 6 x 8
    "resize_crop": dict(seed_resize_from_h=80, seed_resize_from_w=48),                              # taller source: cropped rows, pasted columns
    "both": dict(subseeds=[5, 6, 7], subseed_strength=0.8, seed_resize_from_h=32, seed_resize_from_w=104),
}


def gen_rng_variations():
    """The reference's ImageRNG (modules/rng.py) with variation seeds and seed resize, for the two device-independent noise sources; two draws
    each (the second one comes from the per-image generators, re-seeded with eta_noise_seed_delta)."""
    import importlib.util, sys, types
    ref_import.load_reference()
    opts = SimpleNamespace(randn_source="CPU", forge_try_reproduce="None", eta_noise_seed_delta=31337)
    mods = {"modules": types.ModuleType("modules"), "modules.devices": types.ModuleType("modules.devices"), "modules.shared": types.ModuleType("modules.shared")}
    mods["modules.devices"].device = mods["modules.devices"].cpu = torch.device("cpu")
    mods["modules.shared"].opts, mods["modules.shared"].device = opts, torch.device("cpu")
    saved = {k: sys.modules.get(k) for k in list(mods) + ["modules.rng_philox", "modules.rng"]}
    try:
        sys.modules.update(mods)
        for name in ("rng_philox", "rng"):
            spec = importlib.util.spec_from_file_location(f"modules.{name}", os.path.join(ref_import.REFERENCE_ROOT, "modules", f"{name}.py"))
            m = importlib.util.module_from_spec(spec)
            sys.modules[f"modules.{name}"] = m
            setattr(mods["modules"], name, m)
            spec.loader.exec_module(m)
        mods["modules"].devices, mods["modules"].shared = mods["modules.devices"], mods["modules.shared"]
        rng_ref = sys.modules["modules.rng"]
        out = {"shape": (4, 6, 8), "seeds": [7, 8, 9], "eta_noise_seed_delta": 31337}
        for source in ("CPU", "NV"):
            opts.randn_source = source
            for cname, kw in RNG_VARIATION_CASES.items():
                g = rng_ref.ImageRNG(out["shape"], out["seeds"], **kw)
                out[(source, cname)] = [g.next().clone(), g.next().clone()]
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    torch.save(out, os.path.join(GOLD, "rng_variations.pt"))
    print("rng variations:", {k: float(v[0].std()) for k, v in out.items() if isinstance(k, tuple)})


TOKENIZE_PROMPTS = [
    "a photo of a cat", "", "a (cat:1.3) and a [dog], ((very)) detailed \\(literal\\)", "first part BREAK second part, (emphasised BREAK third:1.2)",
    "masterpiece, best quality, " + ", ".join(f"tag number {i}" for i in range(40)),                       # > 75 tokens: comma backtracking
    " ".join(["word"] * 80) + ", tail after a comma",                                                      # no comma within reach: hard cut
    ", ".join(["a"] * 74) + ", boundary, case",
    "an embedding myemb in the middle, and again myemb, (myemb:1.4)", " ".join(["filler"] * 73) + " myemb end",
    "unbalanced (bracket [here", "colon: in text (weight : 0.5 ) x",
]


class ReplayTokenizer:
    """tokenizer(texts)["input_ids"] from a recorded {text: ids} table (the CLIP vocabulary is 1.5 MB and does not travel); the ids of the
    special tokens and of ',</w>' are recorded too."""

    def __init__(self, rec):
        self.table = rec["table"]
        self.bos_token_id, self.eos_token_id, self.pad_token_id = rec["bos"], rec["eos"], rec["pad"]
        self._comma = rec["comma"]

    def get_vocab(self):
        return {",</w>": self._comma}

    def __call__(self, texts, truncation=False, add_special_tokens=False):
        return {"input_ids": [list(self.table[t]) for t in texts]}


class FakeEmbeddingDb:
    """Textual-inversion database stand-in: the token sequence of 'myemb' maps to a 3-vector embedding."""

    def __init__(self, ids, vectors=3):
        self.ids = list(ids)
        self.emb = SimpleNamespace(name="myemb", vectors=vectors, vec=torch.arange(vectors * 8, dtype=torch.float32).reshape(vectors, 8))
        self.fixes = None

    def find_embedding_at_position(self, tokens, offset):
        if self.ids and tokens[offset:offset + len(self.ids)] == self.ids:
            return self.emb, len(self.ids)
        return None, None


T5_PROMPTS = ["a photo of a cat", "a (very:1.3) detailed [painting] of a fox, BREAK with (((emphasis))) and a second chunk", "",
              "plain text with , commas , and . punctuation " * 12]


class ReplayT5Tokenizer:
    """T5TokenizerFast stand-in from a recorded {text: ids} table; get_vocab() carries the recorded bracket tokens (t5_engine.py:36-53)"""

    def __init__(self, rec):
        self.table, self.vocab = rec["table"], rec["vocab"]

    def get_vocab(self):
        return dict(self.vocab)

    def __call__(self, texts, truncation=False, add_special_tokens=False):
        return {"input_ids": [list(self.table[t]) for t in texts]}


def gen_t5_tokenize():
    """The reference's T5TextProcessingEngine.tokenize_line (backend/text_processing/t5_engine.py:68-112) with the REAL T5 tokenizer
    (backend/huggingface/black-forest-labs/FLUX.1-dev/tokenizer_2); every tokenizer call is recorded for replay."""
    import importlib
    import types
    from transformers import T5TokenizerFast  # before the reference's stub modules are installed
    tok = T5TokenizerFast.from_pretrained(os.path.join(ref_import.REFERENCE_ROOT, "backend", "huggingface", "black-forest-labs", "FLUX.1-dev", "tokenizer_2"))
    ref_import.load_reference()
    saved = {k: sys.modules.get(k) for k in ("modules", "modules.shared")}
    try:
        for n in ("modules", "modules.shared"):
            sys.modules[n] = types.ModuleType(n)
        sys.modules["modules"].__path__ = []
        sys.modules["modules"].shared = sys.modules["modules.shared"]
        sys.modules["modules.shared"].opts = SimpleNamespace(emphasis="Original")
        te = importlib.import_module("backend.text_processing.t5_engine")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    table = {}

    class Recording:
        def get_vocab(self):
            return tok.get_vocab()

        def __call__(self, texts, truncation=False, add_special_tokens=False):
            out = tok(texts, truncation=truncation, add_special_tokens=add_special_tokens)
            for t, ids in zip(texts, out["input_ids"]):
                table[t] = list(ids)
            return out
    eng = te.T5TextProcessingEngine(SimpleNamespace(transformer=None), Recording())
    lines = []
    for prompt in T5_PROMPTS:
        chunks, count = eng.tokenize_line(prompt)
        lines.append({"count": count, "chunks": [{"tokens": list(c.tokens), "multipliers": list(c.multipliers)} for c in chunks]})
    vocab = {k: v for k, v in tok.get_vocab().items() if any(ch in k for ch in "()[]") or k == ",</w>"}
    res = {"prompts": T5_PROMPTS, "lines": lines, "tokenizer": {"table": table, "vocab": vocab}, "token_mults": dict(eng.token_mults), "comma_token": eng.comma_token}
    torch.save(res, os.path.join(GOLD, "tokenize_t5.pt"))
    print("tokenize_t5", [(l["count"], len(l["chunks"])) for l in lines], len(vocab))


def gen_tokenize():
    """The reference's parse_prompt_attention / ClassicTextProcessingEngine.tokenize_line / get_multicond_prompt_list on a prompt set, with the
    real CLIP tokenizer (backend/huggingface/runwayml/stable-diffusion-v1-5/tokenizer); every tokenizer call is recorded for replay."""
    import importlib
    import types
    from transformers import CLIPTokenizer  # before the reference's stub modules are installed
    tok = CLIPTokenizer.from_pretrained(os.path.join(ref_import.REFERENCE_ROOT, "backend", "huggingface", "runwayml", "stable-diffusion-v1-5", "tokenizer"))
    ref_import.load_reference()
    saved = {k: sys.modules.get(k) for k in ("PIL", "PIL.Image", "modules", "modules.shared", "lark")}
    try:
        for n in ("PIL", "PIL.Image", "modules", "modules.shared", "lark"):
            sys.modules[n] = types.ModuleType(n)
        sys.modules["PIL"].Image = sys.modules["PIL.Image"]
        sys.modules["modules"].__path__ = []
        sys.modules["modules"].shared = sys.modules["modules.shared"]
        sys.modules["modules.shared"].opts = SimpleNamespace(emphasis="Original")
        sys.modules["lark"].Lark = lambda *a, **k: None   # the schedule grammar is not exercised here
        ce = importlib.import_module("backend.text_processing.classic_engine")
        parsing = importlib.import_module("backend.text_processing.parsing")
        spec = importlib.util.spec_from_file_location("_ref_prompt_parser", os.path.join(ref_import.REFERENCE_ROOT, "modules", "prompt_parser.py"))
        rpp = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(rpp)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    table = {}

    class Recording:
        bos_token_id, eos_token_id, pad_token_id = tok.bos_token_id, tok.eos_token_id, tok.pad_token_id

        def get_vocab(self):
            return tok.get_vocab()

        def __call__(self, texts, **kw):
            out = tok(texts, **kw)
            for t, ids in zip(texts, out["input_ids"]):
                table[t] = list(ids)
            return out
    enc = SimpleNamespace(transformer=SimpleNamespace(text_model=SimpleNamespace(embeddings=SimpleNamespace(token_embedding=SimpleNamespace(weight=None)))))
    eng = ce.ClassicTextProcessingEngine(enc, Recording(), emphasis_name="Original")
    emb_ids = tok(["myemb"], truncation=False, add_special_tokens=False)["input_ids"][0]
    eng.embeddings = FakeEmbeddingDb(emb_ids)
    res = {"prompts": TOKENIZE_PROMPTS, "emb_ids": emb_ids, "lines": [], "parsed": []}
    for pr in TOKENIZE_PROMPTS:
        chunks, count = eng.tokenize_line(pr)
        res["lines"].append({"count": count, "chunks": [{"tokens": c.tokens, "multipliers": c.multipliers, "fixes": [f.offset for f in c.fixes]} for c in chunks]})
        res["parsed"].append(parsing.parse_prompt_attention(pr, "Original"))
    batch_chunks, token_count = eng.process_texts(TOKENIZE_PROMPTS[:4])
    res["process_texts"] = {"token_count": token_count, "n_chunks": [len(c) for c in batch_chunks]}
    and_prompts = ["a cat AND a dog :1.5 AND a bird: 0.25", "plain", "x AND x", "sandy AND candy:2"]
    idx, flat, pidx = rpp.get_multicond_prompt_list(and_prompts)
    res["multicond"] = {"prompts": and_prompts, "indexes": idx, "flat": list(flat), "prompt_indexes": pidx}
    res["tokenizer"] = {"table": table, "bos": tok.bos_token_id, "eos": tok.eos_token_id, "pad": tok.pad_token_id, "comma": tok.get_vocab().get(",</w>")}
    torch.save(res, os.path.join(GOLD, "tokenize_clip_l.pt"))
    print("tokenize:", [(l["count"], len(l["chunks"])) for l in res["lines"]], res["multicond"]["indexes"])


def gen_schedulers():
    """modules/sd_schedulers.py's table, imported from the reference with a two-attribute stand-in for modules.shared."""
    import importlib.util
    import types
    ref = ref_import.load_reference()
    shared = types.ModuleType("modules.shared")
    shared.opts = SimpleNamespace(beta_dist_alpha=0.6, beta_dist_beta=0.6)
    shared.sd_model = SimpleNamespace(is_sdxl=False)
    pkg = types.ModuleType("modules")
    pkg.shared, pkg.__path__ = shared, []
    saved = {k: sys.modules.get(k) for k in ("modules", "modules.shared")}
    sys.modules["modules"], sys.modules["modules.shared"] = pkg, shared
    try:
        spec = importlib.util.spec_from_file_location("_ref_sd_schedulers", os.path.join(ref_import.REFERENCE_ROOT, "modules", "sd_schedulers.py"))
        rs = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(rs)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    pred = ref_import.build_ref_predictor()
    linker = ref.kd_external.ForgeScheduleLinker(pred)
    linker.inner_model = SimpleNamespace(forge_objects=SimpleNamespace(unet=SimpleNamespace(model=SimpleNamespace(predictor=pred))))
    out = {}
    for sdxl in (False, True):
        shared.sd_model.is_sdxl = sdxl
        for sch in rs.schedulers:
            if sch.function is None:
                continue
            for n in (1, 4, 11, 20, 31, 32):
                kw = {"sigma_min": pred.sigmas[0].item(), "sigma_max": pred.sigmas[-1].item()}
                if sch.need_inner_model:
                    kw["inner_model"] = linker
                out[(sch.name, n, sdxl)] = sch.function(n=n, **kw, device="cpu").float()
    out["labels"] = {s.name: s.label for s in rs.schedulers}
    out["rho"] = {"karras_5": rs.schedulers_map["karras"].function(n=10, sigma_min=0.03, sigma_max=14.0, rho=5.0, device="cpu"),
                  "polyexponential_2": rs.schedulers_map["polyexponential"].function(n=10, sigma_min=0.03, sigma_max=14.0, rho=2.0, device="cpu")}
    torch.save(out, os.path.join(GOLD, "schedulers.pt"))
    print("schedulers", len(out))


def gen_samples_extra(name, cfg, net, b=2, hw=16):
    """The rest of the sampler table through the real reference stack (reference UNet + sampling_function + sampler function)."""
    ref = ref_import.load_reference()
    pred = ref_import.build_ref_predictor()
    restart = _ref_restart()
    c, uc = synth.synth_conditioning(b, cfg["context_dim"], cfg.get("adm_in_channels"), seed=1234)
    seeds = [1000 + i for i in range(b)]
    res = {"seeds": seeds, "hw": hw}
    for sampler, fn_name in REF_SAMPLER_FN.items():
        steps = 21 if sampler == "Restart" else 6
        den = ref_import.RefDenoiser(net, pred, seeds)
        rng = ImageRNG((cfg["in_channels"], hw, hw), seeds, "CPU")
        x = rng.next()
        sigmas = ref_sampler_sigmas(ref, den.inner_model, sampler, steps)
        x = pred.noise_scaling(sigmas[0], x, torch.zeros_like(x), max_denoise=False)
        ref.kd_sampling.torch = _Hijack(rng)
        ref.sampling_function.sampling_prepare(den.patcher, x=x)
        extra = {"cond": c, "uncond": uc, "cond_scale": 7.0, "s_min_uncond": 0.0, "image_cond": None}
        try:
            fn = restart if fn_name is None else getattr(ref.kd_sampling, fn_name)
            lat = fn(den, x, sigmas, extra_args=extra, disable=True)
        finally:
            ref.kd_sampling.torch = torch
            ref.sampling_function.sampling_cleanup(den.patcher)
        res[sampler] = {"steps": steps, "latent": lat, "sigmas": sigmas}
        print(name, sampler, float(lat.std()))
    torch.save(res, os.path.join(GOLD, f"{name}_samples_extra.pt"))


def gen_schedules():
    ref = ref_import.load_reference()
    pred = ref_import.build_ref_predictor()
    linker = ref.kd_external.ForgeScheduleLinker(pred)
    sig = torch.tensor([14.6146, 7.3, 1.0, 0.5, 0.0292, 0.03, 100.0])
    out = {
        "table": pred.sigmas.clone(),
        "linker_20": linker.get_sigmas(20), "linker_30": linker.get_sigmas(30), "linker_6": linker.get_sigmas(6),
        "karras_30": ref.kd_sampling.get_sigmas_karras(30, pred.sigmas[0].item(), pred.sigmas[-1].item()),
        "karras_7": ref.kd_sampling.get_sigmas_karras(7, pred.sigmas[0].item(), pred.sigmas[-1].item()),
        "timestep_in": sig, "timestep_out": pred.timestep(sig),
        "sigma_of_t": pred.sigma(torch.tensor([0.0, 0.5, 10.25, 998.9, 999.0])),
        "ancestral": torch.tensor([list(map(float, ref.kd_sampling.get_ancestral_step(torch.tensor(a), torch.tensor(b))))
                                   for a, b in ((14.6, 9.7), (1.0, 0.5), (0.1, 0.0292))]),
        "philox_seed0_3x4": torch.from_numpy(ref.rng_philox.Generator(0).randn((3, 4))),
    }
    g = ref.rng_philox.Generator(12345)
    out["philox_seed12345_a"] = torch.from_numpy(g.randn((4, 8, 8)))
    out["philox_seed12345_b"] = torch.from_numpy(g.randn((4, 8, 8)))
    torch.save(out, os.path.join(GOLD, "schedules.pt"))
    print("schedules", float(pred.sigmas[0]), float(pred.sigmas[-1]))


def gen_full_sd15():
    """BASELINE config 0: SD1.5 random-init, 512x512, B=1, 20-step Euler, CFG 7 (reference, CPU fp32)."""
    cfg = synth.SD15_UNET_CONFIG
    t0 = time.time()
    sd = synth.synth_unet_state_dict(cfg, seed=0)
    net = ref_import.build_ref_unet(cfg, sd)
    del sd
    c, uc = synth.synth_conditioning(1, cfg["context_dim"], None, seed=1234)
    x, t, ctx, _ = _inputs(cfg, 1, 64, seed=11)
    with torch.no_grad():
        eps = net(x.clone(), t, context=ctx, transformer_options={})
    print("sd15 fwd", time.time() - t0, float(eps.std()))
    t1 = time.time()
    trace = []
    lat, sigmas = ref_sample(net, cfg, c, uc, [42], 64, 20, "Euler", trace=trace)
    t_sample = time.time() - t1
    del net
    vcfg = synth.SD15_VAE_CONFIG
    vsd = synth.synth_vae_decoder_state_dict(vcfg, seed=1)
    vae = ref_import.build_ref_vae(vcfg)
    vae.load_state_dict(vsd, strict=False)
    t2 = time.time()
    with torch.no_grad():
        dec = torch.clamp((vae.decode(vae.process_out(lat)) + 1.0) / 2.0, 0.0, 1.0) * 2.0 - 1.0
    t_dec = time.time() - t2
    img = (255.0 * torch.clamp((dec + 1.0) / 2.0, 0.0, 1.0).movedim(1, -1)).numpy().astype(np.uint8)
    torch.save({"x": x, "t": t, "ctx": ctx, "eps": eps, "seed": 42, "latent": lat, "sigmas": sigmas,
                "denoised0": trace[0], "image_u8": torch.from_numpy(img),
                "cpu_seconds": {"sample20": t_sample, "decode": t_dec, "threads": torch.get_num_threads()}},
               os.path.join(GOLD, "sd15_config0.pt"))
    print("sd15 config0: sampler %.1fs (%.3f it/s), decode %.1fs, latent std %.3f" % (t_sample, 20 / t_sample, t_dec, float(lat.std())))


def gen_full_sdxl():
    """The BASELINE bench workload's network at full size: SDXL UNet (2.57 B parameters, random-init), one sample-forward at the 128x128 latent
    (1024x1024) with 77 x 2048 context and the 2816-wide vector conditioning, on the real reference (CPU fp32)."""
    cfg = synth.SDXL_UNET_CONFIG
    t0 = time.time()
    sd = synth.synth_unet_state_dict(cfg, seed=0)
    net = ref_import.build_ref_unet(cfg, sd)
    del sd
    x, t, ctx, y = _inputs(cfg, 1, 128, seed=13)
    with torch.no_grad():
        eps = net(x.clone(), t, context=ctx, y=y, transformer_options={})
    print("sdxl fwd", time.time() - t0, float(eps.std()))
    torch.save({"eps": eps, "inputs_seed": 13}, os.path.join(GOLD, "sdxl_full_fwd.pt"))  # inputs are regenerated from the seed (_inputs)


def gen_flux(name="tiny_flux", cfg=None, b=2, h=16, w=24, ltxt=40):
    """Flux DiT: one forward of the REAL reference and a 4-step Euler flow-sampling run through the reference's KModel +
    PredictionFlux + k_diffusion.sample_euler ('simple' sigmas: modules/sd_schedulers.py:81-87, restated because that module
    imports gradio)."""
    cfg = cfg or synth.TINY_FLUX_CONFIG
    ref = ref_import.load_reference()
    sd = synth.synth_flux_state_dict(cfg, seed=2)
    net = ref_import.build_ref_flux(cfg, sd)
    g = torch.Generator("cpu").manual_seed(21)
    x = torch.randn(b, cfg["in_channels"], h, w, generator=g)
    ctx = torch.randn(b, ltxt, cfg["context_in_dim"], generator=g)
    y = torch.randn(b, cfg["vec_in_dim"], generator=g)
    t = torch.tensor([0.93, 0.41][:b])
    guid = torch.full((b,), 3.5)
    with torch.no_grad():
        out = net(x.clone(), t, context=ctx, y=y, guidance=guid)
    pred = ref.k_prediction.PredictionFlux(seq_len=(h // 2) * (w // 2))
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        km = ref.k_model.KModel(net, None, k_predictor=pred)
    n = 4
    ss = len(pred.sigmas) / n
    sigmas = torch.FloatTensor([float(pred.sigmas[-(1 + int(i * ss))]) for i in range(n)] + [0.0])
    x0 = torch.randn(b, cfg["in_channels"], h, w, generator=g)
    xs = pred.noise_scaling(sigmas[0], x0.clone(), torch.zeros_like(x0))

    def model_fn(xx, sigma, **kw):
        return km.apply_model(xx, sigma, c_crossattn=ctx, y=y, guidance=guid)

    with torch.no_grad():
        lat = ref.kd_sampling.sample_euler(model_fn, xs, sigmas, disable=True)
    torch.save({"x": x, "t": t, "ctx": ctx, "y": y, "guidance": guid, "out": out, "sigma_table": pred.sigmas.clone(), "mu": float(pred.mu),
                "sigmas": sigmas, "noise": x0, "latent": lat, "hw": (h, w)}, os.path.join(GOLD, f"{name}_fwd.pt"))
    keys = {k: list(v.shape) for k, v in net.state_dict().items()}
    with torch.device("meta"):
        big = ref_import.build_ref_flux(synth.FLUX_DEV_CONFIG)
    shapes = json.load(open(os.path.join(GOLD, "param_shapes.json")))
    shapes["tiny_flux"] = keys
    shapes["flux_dev"] = {k: list(v.shape) for k, v in big.state_dict().items()}
    json.dump(shapes, open(os.path.join(GOLD, "param_shapes.json"), "w"))
    print(name, "fwd", tuple(out.shape), float(out.std()), "latent std", float(lat.std()), "params flux_dev",
          sum(int(np.prod(v)) for v in shapes["flux_dev"].values()) / 1e9, "B")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    if a.only in ("", "keys"):
        gen_keys()
    if a.only in ("", "schedules"):
        gen_schedules()
    if a.only in ("", "tiny"):
        net, _ = gen_unet("tiny_sd15", synth.TINY_SD15_UNET_CONFIG)
        gen_samples("tiny_sd15", synth.TINY_SD15_UNET_CONFIG, net)
        gen_samples_extra("tiny_sd15", synth.TINY_SD15_UNET_CONFIG, net)
        gen_samples_more("tiny_sd15", synth.TINY_SD15_UNET_CONFIG, net)
        gen_unipc("tiny_sd15", synth.TINY_SD15_UNET_CONFIG, net)
        gen_img2img("tiny_sd15", synth.TINY_SD15_UNET_CONFIG, net)
        gen_lora("tiny_sd15", synth.TINY_SD15_UNET_CONFIG)
        gen_unet_control("tiny_sd15", synth.TINY_SD15_UNET_CONFIG, net)
        gen_unet_hooks("tiny_sd15", synth.TINY_SD15_UNET_CONFIG, net)
        gen_controlnet("tiny_sd15", synth.TINY_SD15_UNET_CONFIG, net)
        gen_prediction_types("tiny_sd15", synth.TINY_SD15_UNET_CONFIG, net)
        gen_cfg_paths("tiny_sd15", synth.TINY_SD15_UNET_CONFIG, net)
        net, _ = gen_unet("tiny_sdxl", synth.TINY_SDXL_UNET_CONFIG)
        gen_samples("tiny_sdxl", synth.TINY_SDXL_UNET_CONFIG, net)
        gen_unet_hooks("tiny_sdxl", synth.TINY_SDXL_UNET_CONFIG, net)
        gen_controlnet("tiny_sdxl", synth.TINY_SDXL_UNET_CONFIG, net)
        gen_cfg_paths("tiny_sdxl", synth.TINY_SDXL_UNET_CONFIG, net)
        gen_vae("tiny_vae", synth.TINY_VAE_CONFIG)
        gen_vae_encode("tiny_vae", synth.TINY_VAE_CONFIG)
    if a.only == "samplers":
        net, _ = gen_unet("tiny_sd15", synth.TINY_SD15_UNET_CONFIG)
        gen_samples_extra("tiny_sd15", synth.TINY_SD15_UNET_CONFIG, net)
        gen_samples_more("tiny_sd15", synth.TINY_SD15_UNET_CONFIG, net)
    if a.only in ("", "tokenize"):
        gen_tokenize()
    if a.only in ("", "rng"):
        gen_rng_variations()
    if a.only in ("", "regional"):
        gen_regional()
    if a.only in ("", "t2i"):
        gen_t2i_adapter()
    if a.only in ("", "t2i", "adapterlight"):
        gen_adapter_light()
    if a.only in ("", "tiny", "fluxvae"):
        gen_vae("tiny_flux_vae", synth.TINY_FLUX_VAE_CONFIG)
    if a.only == "lora":
        gen_lora("tiny_sd15", synth.TINY_SD15_UNET_CONFIG)
    if a.only in ("", "tiny", "inpaint"):
        gen_inpaint_model()
    if a.only == "cfgpaths":
        for nm, cf in (("tiny_sd15", synth.TINY_SD15_UNET_CONFIG), ("tiny_sdxl", synth.TINY_SDXL_UNET_CONFIG)):
            net, _ = gen_unet(nm, cf)
            gen_cfg_paths(nm, cf, net)
    if a.only == "prediction":
        net, _ = gen_unet("tiny_sd15", synth.TINY_SD15_UNET_CONFIG)
        gen_prediction_types("tiny_sd15", synth.TINY_SD15_UNET_CONFIG, net)
    if a.only == "controlnet":
        for nm, cf in (("tiny_sd15", synth.TINY_SD15_UNET_CONFIG), ("tiny_sdxl", synth.TINY_SDXL_UNET_CONFIG)):
            net, _ = gen_unet(nm, cf)
            gen_controlnet(nm, cf, net)
    if a.only in ("", "tiny", "controllora"):
        gen_control_lora("tiny_sd15", synth.TINY_SD15_UNET_CONFIG)
        gen_control_lora("tiny_sdxl", synth.TINY_SDXL_UNET_CONFIG)
    if a.only == "hooks":
        for nm, cf in (("tiny_sd15", synth.TINY_SD15_UNET_CONFIG), ("tiny_sdxl", synth.TINY_SDXL_UNET_CONFIG)):
            net, _ = gen_unet(nm, cf)
            gen_unet_hooks(nm, cf, net)
    if a.only in ("", "tiny", "module_hooks"):
        for nm, cf in (("tiny_sd15", synth.TINY_SD15_UNET_CONFIG), ("tiny_sdxl", synth.TINY_SDXL_UNET_CONFIG)):
            gen_unet_module_hooks(nm, cf, ref_import.build_ref_unet(cf, synth.synth_unet_state_dict(cf, seed=0)))
    if a.only == "unipc":
        net, _ = gen_unet("tiny_sd15", synth.TINY_SD15_UNET_CONFIG)
        gen_unipc("tiny_sd15", synth.TINY_SD15_UNET_CONFIG, net)
    if a.only == "samplers_more":
        net, _ = gen_unet("tiny_sd15", synth.TINY_SD15_UNET_CONFIG)
        gen_samples_more("tiny_sd15", synth.TINY_SD15_UNET_CONFIG, net)
    if a.only in ("", "samplers", "sde"):
        gen_samplers_sde()
    if a.only in ("", "samplers"):
        gen_samplers_toy()
        gen_schedulers()
    if a.only in ("", "t5"):
        gen_t5()
    if a.only in ("", "fluxlora"):
        gen_flux_lora()
    if a.only in ("", "t5tok"):
        gen_t5_tokenize()
    if a.only in ("", "clip"):
        gen_clip("tiny_clip_l", synth.TINY_CLIP_L_CONFIG)
        gen_clip("tiny_clip_g", synth.TINY_CLIP_G_CONFIG)
    if a.only in ("", "flux"):
        gen_flux()
    if a.full or a.only == "full":
        gen_full_sd15()
    if a.full or a.only == "full_sdxl":
        gen_full_sdxl()


if __name__ == "__main__":
    main()
