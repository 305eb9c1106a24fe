"""GPU: per-block Python hooks (`patches`, `patches_replace`, `block_modifiers`; backend/nn/unet.py:186-279, :696-763)
called by the native executor, against the REAL reference UNet running the same hook functions (oracle/hooks_fixture.py) on CPU fp32
(tests/golden/*_unet_hooks.pt): same call sequence (hook name, block, block_index, transformer_index), same result."""
import pytest

pytestmark = pytest.mark.gpu

import forge_amd  # noqa: E402
from forge_amd import synth  # noqa: E402
from forge_amd.backend.diffusion_engine.base import build_engine  # noqa: E402
from forge_amd.modules import processing, shared  # noqa: E402
from forge_amd.modules.prompt_parser import DictWithShape  # noqa: E402
from oracle.hooks_fixture import build_hooks  # noqa: E402

from conftest import load_golden  # noqa: E402
from parity import check  # noqa: E402

DEV = "cuda"
TINY = {"tiny_sd15": synth.TINY_SD15_UNET_CONFIG, "tiny_sdxl": synth.TINY_SDXL_UNET_CONFIG}


def max_rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max())


@pytest.fixture(scope="module")
def engines():
    return {n: build_engine(cfg, synth.synth_unet_state_dict(cfg, seed=0), None, None, device=DEV) for n, cfg in TINY.items()}


@pytest.mark.parametrize("name", list(TINY))
def test_unet_forward_with_hooks_vs_reference(name, engines):
    g, fx = load_golden(f"{name}_unet_hooks.pt"), load_golden(f"{name}_unet_fwd.pt")
    net = engines[name].forge_objects.unet.model.diffusion_model
    to, log = build_hooks()
    y = fx["y"].to(DEV) if fx["y"] is not None else None
    eps = net.forward(fx["x"].to(DEV), fx["t"].to(DEV), context=fx["ctx"].to(DEV), y=y, transformer_options=to)
    assert log == g["log"], [(a, b) for a, b in zip(log, g["log"]) if a != b][:5]   # every hook, in the reference's order, with its block ids
    check(f"{name} unet forward with {len(log)} hook calls vs reference", eps, g["eps"], floor=f"{name}_unet_hooks.pt:eps")
    plain = net.forward(fx["x"].to(DEV), fx["t"].to(DEV), context=fx["ctx"].to(DEV), y=y)
    check(f"{name} unet forward, hooks off again, vs reference", plain, fx["eps"], floor=f"{name}_unet_fwd.pt:eps")
    assert max_rel(eps, plain) > 0.05  # hooks do change the result


@pytest.mark.parametrize("name", list(TINY))
def test_module_typed_hooks_vs_reference(name, engines):
    """`block_inner_modifiers` and `group_norm_wrapper` (unet.py:73-91, :436-474, :755-757) -- the hooks whose arguments are modules in the
    reference.  The executor hands them stand-ins with the reference's class names and the GroupNorm's parameters (callable: norm(x) runs the
    native kernel); the SAME hook functions ran inside the real reference for the fixture: equal call sequence (class name, layer index, block
    length, block id; GroupNorm geometry), equal result; installed through the UnetPatcher API."""
    from oracle.hooks_fixture import build_module_hooks
    g, fx = load_golden(f"{name}_unet_module_hooks.pt"), load_golden(f"{name}_unet_fwd.pt")
    to, log = build_module_hooks()
    unet = engines[name].forge_objects.unet.clone()
    unet.add_block_inner_modifier(to["block_inner_modifiers"][0])
    unet.set_group_norm_wrapper(to["group_norm_wrapper"])
    net = unet.model.diffusion_model
    y = fx["y"].to(DEV) if fx["y"] is not None else None
    eps = net.forward(fx["x"].to(DEV), fx["t"].to(DEV), context=fx["ctx"].to(DEV), y=y, transformer_options=unet.model_options["transformer_options"])
    assert log == g["log"], [(a, b) for a, b in zip(log, g["log"]) if a != b][:5] + [len(log), len(g["log"])]
    check(f"{name} unet forward with block_inner_modifiers + group_norm_wrapper ({len(log)} calls) vs reference", eps, g["eps"],
          floor=f"{name}_unet_module_hooks.pt:eps")
    assert max_rel(eps, fx["eps"]) > 0.03   # the hooks do change the result
    from forge_amd.backend.nn import unet as native_unet
    seen = []
    net.forward(fx["x"].to(DEV), fx["t"].to(DEV), context=fx["ctx"].to(DEV), y=y,
                transformer_options={"block_inner_modifiers": [lambda x, when, layer, li, block, to_: (seen.append(layer), x)[1]]})
    assert any(isinstance(l, native_unet.SpatialTransformer) for l in seen) and any(isinstance(l, native_unet.ResBlock) for l in seen)


@pytest.mark.parametrize("name", list(TINY))
def test_sampling_with_hooks_installed_on_the_patcher(name, engines):
    """Hooks set through the UnetPatcher API (set_model_*), reaching the executor via sampling_function with the per-call keys
    (cond_or_uncond, sigmas, cond_mark, cond_indices, uncond_indices) the reference adds (sampling_function.py:253-257)."""
    cfg = TINY[name]
    g = load_golden(f"{name}_unet_hooks.pt")
    eng = engines[name]
    to, log = build_hooks(use_call_keys=True)
    unet = eng.forge_objects.unet.clone()
    for k, fns in to["patches"].items():
        for fn in fns:
            unet.set_model_patch(fn, k)
    unet.set_model_attn1_replace(to["patches_replace"]["attn1"][("middle", 0, 0)], "middle", 0, 0)
    unet.set_model_attn2_replace(to["patches_replace"]["attn2"][("input", 3)], "input", 3)
    unet.add_block_modifier(to["block_modifiers"][0])
    saved = eng.forge_objects_after_applying_lora
    eng.forge_objects_after_applying_lora = saved.shallow_copy()
    eng.forge_objects_after_applying_lora.unet = unet
    try:
        b = len(g["euler3"]["seeds"])
        c, uc = synth.synth_conditioning(b, cfg["context_dim"], cfg.get("adm_in_channels"), seed=1234)
        if isinstance(c, dict):
            c, uc = DictWithShape({k: v.to(DEV) for k, v in c.items()}), DictWithShape({k: v.to(DEV) for k, v in uc.items()})
        else:
            c, uc = c.to(DEV), uc.to(DEV)
        shared.opts.randn_source = "CPU"
        hw = g["euler3"]["hw"]
        p = processing.StableDiffusionProcessingTxt2Img(sd_model=eng, c=c, uc=uc, seed=g["euler3"]["seeds"][0], sampler_name="Euler", batch_size=b,
                                                        steps=3, cfg_scale=7.0, width=hw * 8, height=hw * 8, do_decode=False)
        lat = processing.process_images(p).latents
    finally:
        eng.forge_objects_after_applying_lora = saved
        eng.forge_objects = saved.shallow_copy()
    assert log[:len(g["log_first_forward_of_run"])] == g["log_first_forward_of_run"]
    check(f"{name} 3-step Euler with hooks on the patcher vs reference", lat, g["euler3"]["latent"], floor=f"{name}_unet_hooks.pt:euler3/latent")
    # and the same engine, hooks gone, is back on the captured-graph fast path with the unhooked result
    ref = load_golden(f"{name}_samples.pt")
    p = processing.StableDiffusionProcessingTxt2Img(sd_model=eng, c=c, uc=uc, seed=ref["seeds"][0], sampler_name="Euler", batch_size=b,
                                                    steps=ref["Euler"]["steps"], cfg_scale=7.0, width=ref["hw"] * 8, height=ref["hw"] * 8, do_decode=False)
    check(f"{name} Euler after the hooks were removed (graph path) vs reference", processing.process_images(p).latents, ref["Euler"]["latent"],
          floor=f"{name}_samples.pt:Euler/latent")
