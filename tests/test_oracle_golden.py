"""CPU: the torch-fp32 oracle (oracle/) reproduces the fixtures generated from the REAL reference
(tests/golden/*, made by oracle/make_golden.py).  This is what pins the oracle."""
import json
import os

import numpy as np
import pytest
import torch

import forge_amd  # noqa: F401
from forge_amd import synth
from forge_amd.backend.nn.layout import flux_param_shapes, unet_param_shapes, vae_decoder_param_shapes, vae_encoder_param_shapes
from oracle import pipeline, sampling
from oracle.k_prediction import Predictor
from oracle.rng import PhiloxGenerator
from oracle.unet import unet_forward
from oracle.vae import decode_first_stage, encode_first_stage, posterior_sample, vae_decode, vae_encode_moments

from conftest import GOLDEN, load_golden

def max_rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


TINY = {"tiny_sd15": synth.TINY_SD15_UNET_CONFIG, "tiny_sdxl": synth.TINY_SDXL_UNET_CONFIG}


def test_param_shapes_match_reference():
    ref = json.load(open(os.path.join(GOLDEN, "param_shapes.json")))
    for name, cfg in (("sd15", synth.SD15_UNET_CONFIG), ("sdxl", synth.SDXL_UNET_CONFIG), *TINY.items()):
        ours = {k: list(v) for k, v in unet_param_shapes(cfg).items()}
        assert ours == ref[name], name
    for name, cfg in (("vae", synth.SD15_VAE_CONFIG), ("tiny_vae", synth.TINY_VAE_CONFIG)):
        ours = {k: list(v) for k, v in vae_decoder_param_shapes(cfg).items()}
        assert ours == ref[name], name


def test_schedules_and_predictor():
    g = load_golden("schedules.pt")
    p = Predictor()
    assert torch.equal(p.sigmas, g["table"])
    for n in (20, 30, 6):
        torch.testing.assert_close(sampling.get_sigmas_linker(p, n), g[f"linker_{n}"], rtol=0, atol=0)
    for n in (30, 7):
        torch.testing.assert_close(sampling.get_sigmas_karras(n, p.sigmas[0].item(), p.sigmas[-1].item()),
                                   g[f"karras_{n}"], rtol=0, atol=0)
    assert torch.equal(p.timestep(g["timestep_in"]), g["timestep_out"])
    torch.testing.assert_close(p.sigma(torch.tensor([0.0, 0.5, 10.25, 998.9, 999.0])), g["sigma_of_t"], rtol=0, atol=0)
    for (a, b), want in zip(((14.6, 9.7), (1.0, 0.5), (0.1, 0.0292)), g["ancestral"]):
        d, u = sampling.get_ancestral_step(torch.tensor(a), torch.tensor(b))
        assert abs(float(d) - float(want[0])) < 1e-6 and abs(float(u) - float(want[1])) < 1e-6


def test_philox_known_answer():
    # modules/rng_philox.py:10-15 docstring vector (the only KAT the reference ships) + reference outputs
    g = load_golden("schedules.pt")
    doc = np.array([[-0.92466259, -0.42534415, -2.6438457, 0.14518388],
                    [-0.12086647, -0.57972564, -0.62285122, -0.32838709],
                    [-1.07454231, -0.36314407, -1.67105067, 2.26550497]], dtype=np.float32)
    out = PhiloxGenerator(0).randn((3, 4))
    np.testing.assert_allclose(out, doc, rtol=0, atol=2e-6)
    np.testing.assert_array_equal(out, g["philox_seed0_3x4"].numpy())
    gen = PhiloxGenerator(12345)
    np.testing.assert_array_equal(gen.randn((4, 8, 8)), g["philox_seed12345_a"].numpy())
    np.testing.assert_array_equal(gen.randn((4, 8, 8)), g["philox_seed12345_b"].numpy())


@pytest.mark.parametrize("name", list(TINY))
def test_unet_forward(name):
    cfg = TINY[name]
    g = load_golden(f"{name}_unet_fwd.pt")
    sd = synth.synth_unet_state_dict(cfg, seed=0)
    eps = unet_forward(sd, cfg, g["x"], g["t"], g["ctx"], g["y"])
    torch.testing.assert_close(eps, g["eps"], rtol=1e-4, atol=1e-5)


def test_vae_decode():
    g = load_golden("tiny_vae_decode.pt")
    sd = synth.synth_vae_decoder_state_dict(synth.TINY_VAE_CONFIG, seed=1)
    torch.testing.assert_close(vae_decode(sd, g["z"]), g["decode"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(decode_first_stage(sd, g["lat"], 0.18215), g["decode_first_stage"], rtol=1e-4, atol=1e-5)
    # 16-channel latents, no quant convs, shift factor (Flux / SD3 VAE)
    g, cfg = load_golden("tiny_flux_vae_decode.pt"), synth.TINY_FLUX_VAE_CONFIG
    sd = synth.synth_vae_decoder_state_dict(cfg, seed=1)
    torch.testing.assert_close(vae_decode(sd, g["z"]), g["decode"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(decode_first_stage(sd, g["lat"], cfg["scaling_factor"], cfg["shift_factor"]), g["decode_first_stage"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", list(TINY))
@pytest.mark.parametrize("sampler", ["Euler", "Euler a", "DPM++ 2M"])
def test_sampler_runs(name, sampler):
    cfg = TINY[name]
    g = load_golden(f"{name}_samples.pt")
    sd = synth.synth_unet_state_dict(cfg, seed=0)
    b = len(g["seeds"])
    c, uc = synth.synth_conditioning(b, cfg["context_dim"], cfg.get("adm_in_channels"), seed=1234)
    trace = []
    lat, sigmas = pipeline.txt2img_latents(sd, cfg, c, uc, g["seeds"], g["hw"] * 8, g["hw"] * 8,
                                           g[sampler]["steps"], sampler_name=sampler, trace=trace)
    torch.testing.assert_close(sigmas, g[sampler]["sigmas"], rtol=0, atol=0)
    # latents/denoised are O(15) (sigma_max-scaled); fp32 reassociation (SDPA vs explicit softmax) shows at ~1e-5 relative
    assert max_rel(trace[0], g[sampler]["denoised0"]) < 5e-5
    assert max_rel(lat, g[sampler]["latent"]) < 2e-4


def test_cfg_scale_one_shortcut():
    cfg = TINY["tiny_sd15"]
    g = load_golden("tiny_sd15_samples.pt")
    sd = synth.synth_unet_state_dict(cfg, seed=0)
    c, uc = synth.synth_conditioning(2, cfg["context_dim"], None, seed=1234)
    lat, _ = pipeline.txt2img_latents(sd, cfg, c, uc, g["seeds"], 128, 128, 3, sampler_name="Euler", cfg_scale=1.0)
    assert max_rel(lat, g["Euler_cfg1"]["latent"]) < 2e-4


# ---- Flux ---------------------------------------------------------------------------------------
def test_flux_param_shapes_match_reference():
    ref = json.load(open(os.path.join(GOLDEN, "param_shapes.json")))
    for name, cfg in (("tiny_flux", synth.TINY_FLUX_CONFIG), ("flux_dev", synth.FLUX_DEV_CONFIG)):
        ours = {k: list(v) for k, v in flux_param_shapes(cfg).items()}
        assert ours == ref[name], name


def test_flux_forward_and_sampler_restated():
    from oracle import flux as oflux
    g = load_golden("tiny_flux_fwd.pt")
    cfg = synth.TINY_FLUX_CONFIG
    sd = synth.synth_flux_state_dict(cfg, seed=2)
    out = oflux.flux_forward(sd, cfg, g["x"], g["t"], g["ctx"], g["y"], g["guidance"])
    torch.testing.assert_close(out, g["out"], rtol=1e-4, atol=1e-5)
    h, w = g["hw"]
    table = oflux.flux_sigma_table(seq_len=(h // 2) * (w // 2))
    torch.testing.assert_close(table.float(), g["sigma_table"], rtol=1e-6, atol=1e-7)
    sigmas = oflux.flux_sigmas_simple(4, g["sigma_table"])
    torch.testing.assert_close(sigmas, g["sigmas"], rtol=0, atol=0)
    xs = sigmas[0] * g["noise"]  # noise_scaling for 'const' with a zero latent (k_prediction.py:94-96)
    lat = oflux.flux_sample_euler(sd, cfg, xs, sigmas, g["ctx"], g["y"], g["guidance"])
    assert max_rel(lat, g["latent"]) < 2e-4


def test_vae_encoder_oracle_vs_reference_fixture():
    """Encoder + quant_conv + posterior sample + process_in of the real reference (backend/nn/vae.py:183-200, :296-303, :16-29)."""
    g = load_golden("tiny_vae_encode.pt")
    cfg = synth.TINY_VAE_CONFIG
    sd = synth.synth_vae_state_dict(cfg, seed=1)
    assert set(vae_encoder_param_shapes(cfg)) <= set(sd)
    m = vae_encode_moments(sd, g["x"])
    assert max_rel(m, g["moments"]) < 1e-4
    assert max_rel(posterior_sample(m, g["noise"]), g["sample"]) < 1e-4
    lat = encode_first_stage(sd, g["x"], cfg["scaling_factor"], cfg["shift_factor"], noise=g["noise"])
    assert max_rel(lat, g["process_in"]) < 1e-4
    # the decoder half of the joint state dict is the decoder-only one (weights are a pure function of the name)
    dec = synth.synth_vae_decoder_state_dict(cfg, seed=1)
    assert all(torch.equal(sd[k], v) for k, v in dec.items())


def test_img2img_oracle_vs_reference_fixture():
    """sample_img2img (modules/sd_samplers_kdiffusion.py:136-194) + inpaint mask blending (sd_samplers_cfg_denoiser.py:178-213,
    processing.py:1865-1866) restated in oracle/pipeline.py vs the reference's own loops / sampling_function / UNet."""
    from oracle.make_golden import mask_noise_fn
    g = load_golden("tiny_sd15_img2img.pt")
    cfg = synth.TINY_SD15_UNET_CONFIG
    sd = synth.synth_unet_state_dict(cfg, seed=0)
    c, uc = synth.synth_conditioning(len(g["seeds"]), cfg["context_dim"], None, seed=1234)
    for sampler in ("Euler", "Euler a", "DPM++ 2M"):
        r = g[sampler]
        lat, sched = pipeline.img2img_latents(sd, cfg, c, uc, g["seeds"], g["init_latent"], r["steps"], r["denoising_strength"], sampler_name=sampler)
        assert torch.equal(sched, r["sigma_sched"]), sampler
        assert max_rel(lat, r["latent"]) < 2e-4, (sampler, max_rel(lat, r["latent"]))
    r = g["Euler_masked"]
    lat, _ = pipeline.img2img_latents(sd, cfg, c, uc, g["seeds"], g["init_latent"], r["steps"], r["denoising_strength"], sampler_name="Euler",
                                      mask=r["mask"], nmask=r["nmask"], mask_noise=mask_noise_fn(tuple(g["init_latent"].shape)))
    assert max_rel(lat, r["latent"]) < 2e-4
    # the kept region is exactly the original (processing.py:1865-1866)
    keep = r["mask"] == 1.0
    assert torch.equal(lat[keep], g["init_latent"][keep])


def test_unet_control_residuals_oracle_vs_reference_fixture():
    """apply_control (backend/nn/unet.py:44-52) at the three injection points: oracle vs the real reference UNet."""
    from oracle.make_golden import synth_control
    g, fx = load_golden("tiny_sd15_unet_ctrl.pt"), load_golden("tiny_sd15_unet_fwd.pt")
    cfg = synth.TINY_SD15_UNET_CONFIG
    sd = synth.synth_unet_state_dict(cfg, seed=0)
    out = unet_forward(sd, cfg, fx["x"], fx["t"], fx["ctx"], None, control=synth_control(cfg, fx["x"].shape[0], g["hw"]))
    assert max_rel(out, g["eps"]) < 1e-4


@pytest.mark.parametrize("name,cfg", [("tiny_clip_l", synth.TINY_CLIP_L_CONFIG), ("tiny_clip_g", synth.TINY_CLIP_G_CONFIG)])
def test_clip_oracle_vs_transformers_fixture(name, cfg):
    """oracle/clip.py vs transformers.CLIPTextModel outputs (tests/golden/tiny_clip_*.pt)."""
    from oracle import clip as oclip
    g = load_golden(name + ".pt")
    sd = synth.synth_clip_state_dict(cfg)
    hs = oclip.clip_hidden_states(sd, cfg, g["ids"])
    assert max_rel(hs[-1], g["hidden_last"]) < 1e-5 and max_rel(hs[-2], g["hidden_penultimate"]) < 1e-5
    z, pooled = oclip.encode_with_transformers(sd, cfg, g["ids"], return_pooled=True, is_clip_l=True)
    assert max_rel(z, g["last_hidden_state"]) < 1e-5 and max_rel(pooled, g["pooled"]) < 1e-5
    z2, _ = oclip.encode_with_transformers(sd, cfg, g["ids"], clip_skip=2)
    assert max_rel(z2, g["penultimate_final_ln"]) < 1e-5
    if "pooled_projected" in g:
        _, pp = oclip.encode_with_transformers(sd, cfg, g["ids"], return_pooled=True, is_clip_l=False)
        assert max_rel(pp, g["pooled_projected"]) < 1e-5


@pytest.mark.parametrize("name", list(TINY))
def test_unet_hooks_restated(name):
    """oracle/unet.py's hook points vs the reference UNet running the same hook functions (oracle/hooks_fixture.py): identical call
    sequence (hook, block, block_index, transformer_index) and output."""
    from oracle.hooks_fixture import build_hooks
    cfg = TINY[name]
    g, fx = load_golden(f"{name}_unet_hooks.pt"), load_golden(f"{name}_unet_fwd.pt")
    sd = synth.synth_unet_state_dict(cfg, seed=0)
    to, log = build_hooks()
    eps = unet_forward(sd, cfg, fx["x"], fx["t"], fx["ctx"], fx["y"], transformer_options=to)
    assert log == g["log"]
    torch.testing.assert_close(eps, g["eps"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", list(TINY))
def test_controlnet_restated(name):
    """oracle/controlnet.py vs the reference's cldm.ControlNet (model level) and a 4-step Euler run through the reference's
    sampling_function with a chain of two patcher-level ControlNets (strength, percent ranges, global average pooling, all advanced
    weightings, hint resize + centre crop + batch broadcast)."""
    from oracle import controlnet as ocn
    from oracle.make_golden import controlnet_case, sigma_weight
    cfg = TINY[name]
    g, fx = load_golden(f"{name}_controlnet.pt"), load_golden(f"{name}_unet_fwd.pt")
    case = controlnet_case(cfg)
    sd_a, sd_b = synth.synth_controlnet_state_dict(cfg, seed=6), synth.synth_controlnet_state_dict(cfg, seed=9)
    outs = ocn.controlnet_forward(sd_a, cfg, fx["x"], case["hint_a"], fx["t"], fx["ctx"], fx["y"])
    assert len(outs) == len(g["outs_every_4th_channel"])
    for o, want in zip(outs, g["outs_every_4th_channel"]):
        torch.testing.assert_close(o[:, ::4], want, rtol=1e-4, atol=1e-5)
    first = ocn.Control(sd_a, cfg, case["hint_a"], 0.8, (0.0, 0.7), weighting={"positive": case["positive"], "negative": case["negative"],
                                                                                "frame": case["frame"], "sigma": sigma_weight, "mask": case["mask"]})
    chain = ocn.Control(sd_b, cfg, case["hint_b"], 0.5, (0.2, 1.0), global_average_pooling=True, previous=first)
    sd = synth.synth_unet_state_dict(cfg, seed=0)
    c, uc = synth.synth_conditioning(2, cfg["context_dim"], cfg.get("adm_in_channels"), seed=1234)
    lat = pipeline.txt2img_latents_controlnet(sd, cfg, c, uc, g["euler4"]["seeds"], g["hw"] * 8, g["hw"] * 8, 4, chain)
    assert max_rel(lat, g["euler4"]["latent"]) < 2e-4


@pytest.mark.parametrize("name", list(TINY))
def test_control_lora_restated(name):
    """oracle/controlnet.py control_lora_weights (UNet trunk + direct tensors + up @ down pairs) vs the reference's ControlLora: the residuals of
    one control-model call and a 4-step Euler run through the reference's sampling stack."""
    from oracle import controlnet as ocn
    from oracle.make_golden import controlnet_case
    cfg = TINY[name]
    g, fx = load_golden(f"{name}_control_lora.pt"), load_golden(f"{name}_unet_fwd.pt")
    case = controlnet_case(cfg)
    sd = synth.synth_unet_state_dict(cfg, seed=0)
    cl = synth.synth_control_lora_state_dict(cfg)
    assert "lora_controlnet" in cl and not any(k.startswith("time_embed.") for k in cl)   # time_embed runs on the UNet's own weights
    merged = ocn.control_lora_weights(sd, cl)
    assert not torch.equal(merged["input_blocks.1.0.in_layers.2.weight"], sd["input_blocks.1.0.in_layers.2.weight"])
    outs = ocn.controlnet_forward(merged, cfg, fx["x"], case["hint_a"], fx["t"], fx["ctx"], fx["y"])
    assert len(outs) == len(g["outs_every_8th_channel"])
    for o, want in zip(outs, g["outs_every_8th_channel"]):
        torch.testing.assert_close(o[:, ::8], want, rtol=1e-4, atol=1e-5)
    c, uc = synth.synth_conditioning(2, cfg["context_dim"], cfg.get("adm_in_channels"), seed=1234)
    chain = ocn.Control(merged, cfg, case["hint_a"], 0.9, (0.0, 1.0))
    lat = pipeline.txt2img_latents_controlnet(sd, cfg, c, uc, g["euler4"]["seeds"], g["hw"] * 8, g["hw"] * 8, 4, chain)
    assert max_rel(lat, g["euler4"]["latent"]) < 2e-4


@pytest.mark.parametrize("name", list(TINY))
def test_general_cfg_paths_restated(name):
    """oracle/cfg.py cfg_denoise_general vs the reference's sampling_function: AND-composed prompts (edit strength), the three cfg
    function hooks, model_function_wrapper."""
    from oracle.make_golden import cfg_hooks_fixture, multicond_case
    cfg = TINY[name]
    g = load_golden(f"{name}_cfg_paths.pt")
    sd = synth.synth_unet_state_dict(cfg, seed=0)
    c4, uc, comp = multicond_case(cfg)
    c1, _ = synth.synth_conditioning(2, cfg["context_dim"], cfg.get("adm_in_channels"), seed=1234)
    hooks = cfg_hooks_fixture()
    size = g["hw"] * 8
    cases = {"and_composed": (c4, comp, None), "plain": (c1, None, None),
             "cfg_functions": (c1, None, {k: hooks[k] for k in ("sampler_cfg_function", "sampler_post_cfg_function", "sampler_pre_cfg_function")}),
             "model_function_wrapper": (c1, None, {"model_function_wrapper": hooks["model_function_wrapper"]})}
    for key, (cond, composition, options) in cases.items():
        lat = pipeline.txt2img_latents_general_cfg(sd, cfg, cond, uc, composition, g["seeds"], size, size, 3, 5.0, options)
        assert max_rel(lat, g[key]) < 2e-4, key


def test_inpainting_model_restated():
    """in_channels-9 UNet + c_concat (k_model.py:38-39, sampling_function.py:342-350) vs the reference UNet / sampling_function."""
    from oracle.make_golden import inpaint_case
    cfg = synth.TINY_SD15_INPAINT_UNET_CONFIG
    g, fx = load_golden("tiny_sd15_inpaint_model.pt"), load_golden("tiny_sd15_unet_fwd.pt")
    sd = synth.synth_unet_state_dict(cfg, seed=0)
    ic = inpaint_case()
    torch.testing.assert_close(unet_forward(sd, cfg, torch.cat([fx["x"], ic], dim=1), fx["t"], fx["ctx"], None), g["eps"], rtol=1e-4, atol=1e-5)
    c, uc = synth.synth_conditioning(2, cfg["context_dim"], None, seed=1234)
    lat = pipeline.txt2img_latents_inpaint_model(sd, cfg, c, uc, g["seeds"], g["hw"] * 8, g["hw"] * 8, 3, ic)
    assert max_rel(lat, g["euler3"]) < 2e-4


def test_adapter_light_restated():
    """oracle/controlnet.py adapter_light_forward vs the reference's Adapter_light; also pins the product's adapter_light_param_shapes to the
    reference module's state dict (asserted at fixture time)."""
    from oracle import controlnet as ocn
    from oracle.make_golden import ADAPTER_LIGHT_KW, adapter_light_hint
    g = load_golden("mini_adapter_light.pt")
    feats = ocn.adapter_light_forward(synth.synth_t2i_adapter_light_state_dict(**ADAPTER_LIGHT_KW), adapter_light_hint(), ADAPTER_LIGHT_KW["channels"],
                                      ADAPTER_LIGHT_KW["nums_rb"])
    assert [None if f is None else tuple(f.shape) for f in feats] == g["layout"]
    for f, w in zip([f for f in feats if f is not None], g["values_every_4th_channel"]):
        torch.testing.assert_close(f[:, ::4], w, rtol=1e-4, atol=1e-5)


def test_t2i_adapter_restated():
    """oracle/controlnet.py adapter_forward / AdapterControl vs the reference's Adapter (three checkpoint layouts: 1x1 + avg-pool, 3x3 + strided
    conv, SDXL x16 unshuffle) and vs a 3-step Euler run through the reference sampling_function with the reference T2IAdapter attached; also
    pins the product's adapter_param_shapes to the reference module's state dict (checked at fixture time)."""
    from oracle import controlnet as ocn
    from oracle.make_golden import ADAPTER_VARIANTS, adapter_hint
    g = load_golden("mini_sd15_t2i_adapter.pt")
    for vname, kw in ADAPTER_VARIANTS.items():
        sd = synth.synth_t2i_adapter_state_dict(**kw)
        feats = ocn.adapter_forward(sd, adapter_hint(vname), kw["channels"], kw["nums_rb"], kw["ksize"], kw["use_conv"], kw["xl"])
        want = g["features"][vname]
        assert [None if f is None else tuple(f.shape) for f in feats] == want["layout"], vname
        for f, w in zip([f for f in feats if f is not None], want["values_every_4th_channel"]):
            torch.testing.assert_close(f[:, ::4], w, rtol=1e-4, atol=1e-5)
    cfg = synth.MINI_SD15_UNET_CONFIG
    kw = ADAPTER_VARIANTS["sd15_k1_pool"]
    chain = ocn.AdapterControl(synth.synth_t2i_adapter_state_dict(**kw), adapter_hint("sd15_k1_pool"), 0.9, (0.0, 0.6), channels=kw["channels"],
                               nums_rb=kw["nums_rb"], ksize=kw["ksize"], use_conv=kw["use_conv"], xl=kw["xl"])
    sd = synth.synth_unet_state_dict(cfg, seed=0)
    c, uc = synth.synth_conditioning(2, cfg["context_dim"], None, seed=1234)
    lat = pipeline.txt2img_latents_controlnet(sd, cfg, c, uc, g["seeds"], g["hw"] * 8, g["hw"] * 8, 3, chain)
    assert max_rel(lat, g["euler3"]) < 2e-4


def test_image_rng_variation_seeds_and_seed_resize():
    """oracle/rng.py ImageRNG and the product's modules/rng.py ImageRNG (run here on the CPU: the "CPU" noise source is torch-only host
    logic) against the reference's modules/rng.py ImageRNG: variation seeds (slerp), seed resize (centre paste / crop), both, and the second
    draw after the eta_noise_seed_delta re-seed.  Bit-exact for the CPU source; the Philox ("NV") source is checked for the oracle."""
    from oracle.make_golden import RNG_VARIATION_CASES
    from oracle.rng import ImageRNG as OracleRNG
    from forge_amd.modules import rng as prod_rng, shared
    g = load_golden("rng_variations.pt")
    for source in ("CPU", "NV"):
        for cname, kw in RNG_VARIATION_CASES.items():
            o = OracleRNG(g["shape"], g["seeds"], source, eta_noise_seed_delta=g["eta_noise_seed_delta"], **kw)
            for want in g[(source, cname)]:
                got = o.next()
                if source == "CPU":
                    assert torch.equal(got, want), (source, cname)
                else:
                    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6)
    saved = shared.opts.randn_source, shared.opts.eta_noise_seed_delta
    shared.opts.randn_source, shared.opts.eta_noise_seed_delta = "CPU", g["eta_noise_seed_delta"]
    try:
        for cname, kw in RNG_VARIATION_CASES.items():
            r = prod_rng.ImageRNG(g["shape"], g["seeds"], device="cpu", **kw)
            for want in g[("CPU", cname)]:
                assert torch.equal(r.next(), want), cname
        plain = prod_rng.ImageRNG(g["shape"], g["seeds"], device="cpu")   # no variation: the per-image generators' first tensors, as before
        assert torch.equal(plain.next()[1], torch.randn(g["shape"], generator=torch.Generator("cpu").manual_seed(8)))
    finally:
        shared.opts.randn_source, shared.opts.eta_noise_seed_delta = saved


def test_processing_wires_variation_seeds_into_the_noise():
    """process_images_inner (processing.py:863-944): with a subseed strength every image of the job shares the seed and the subseed counts up,
    batch n of n_iter takes its slice, and the ImageRNG it builds draws the reference's values (the sampler is replaced by `rng.next()`)."""
    from types import SimpleNamespace
    from forge_amd.modules import processing, shared
    g = load_golden("rng_variations.pt")
    c, h, w = g["shape"]

    class NoiseOnly(processing.StableDiffusionProcessingTxt2Img):
        def sample(self, conditioning, unconditional_conditioning, seeds, subseeds=None, subseed_strength=0.0, prompts=None):
            return self.rng.next()
    model = SimpleNamespace(device=torch.device("cpu"), forge_objects=SimpleNamespace(vae=None), latent_channels=c)
    saved = shared.opts.randn_source, shared.opts.eta_noise_seed_delta, getattr(shared, "sd_model", None)
    shared.opts.randn_source, shared.opts.eta_noise_seed_delta = "CPU", 0
    try:
        p = NoiseOnly(sd_model=model, c=torch.zeros(4, 1, 1), uc=torch.zeros(4, 1, 1), seed=7, subseed=100, subseed_strength=0.35, batch_size=2,
                      n_iter=2, width=w * 8, height=h * 8, do_decode=False)
        res = processing.process_images_inner(p)
        assert p.all_seeds == [7, 7, 7, 7] and p.all_subseeds == [100, 101, 102, 103] and p.subseeds == [102, 103]
        want = g[("CPU", "subseed")][0][0]   # fixture image 0: seed 7, subseed 100, strength 0.35
        assert torch.equal(res.latents[0], want)
        assert not torch.equal(res.latents[1], res.latents[0])   # same seed, next subseed
        p2 = NoiseOnly(sd_model=model, c=torch.zeros(2, 1, 1), uc=torch.zeros(2, 1, 1), seed=7, batch_size=2, width=w * 8, height=h * 8, do_decode=False)
        res2 = processing.process_images_inner(p2)
        assert p2.all_seeds == [7, 8] and torch.equal(res2.latents[1], torch.randn(g["shape"], generator=torch.Generator("cpu").manual_seed(8)))
    finally:
        shared.opts.randn_source, shared.opts.eta_noise_seed_delta = saved[0], saved[1]
        shared.sd_model = saved[2]


def test_t5_restatement_vs_reference_fixture():
    """oracle/t5.py against the REAL reference class (backend/nn/t5.py IntegratedT5) on the tiny configuration: tests/golden/tiny_t5.pt"""
    from forge_amd import synth
    from oracle import t5 as ot5
    g = load_golden("tiny_t5.pt")
    z = ot5.t5_encode(synth.synth_t5_state_dict(synth.TINY_T5_CONFIG), synth.TINY_T5_CONFIG, g["ids"])
    assert float((z - g["z"]).abs().max() / g["z"].abs().max()) < 1e-4
