"""GPU: the native ControlNet (backend/nn/cnets/cldm.py on the UNet executor's kernels) and its patcher-level wiring
(backend/patcher/controlnet.py, sampling_function) against the REAL reference classes (tests/golden/*_controlnet.pt, made on CPU fp32 by
oracle/make_golden.py gen_controlnet with cldm.ControlNet, patcher.controlnet.ControlNet / apply_controlnet_advanced, sampling_function)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import forge_amd  # noqa: E402
from forge_amd import synth  # noqa: E402
from forge_amd.backend.diffusion_engine.base import build_engine  # noqa: E402
from forge_amd.backend.nn.cnets import cldm  # noqa: E402
from forge_amd.backend.patcher import controlnet as pc  # noqa: E402
from forge_amd.modules import processing, shared  # noqa: E402
from forge_amd.modules.prompt_parser import DictWithShape  # noqa: E402
from oracle.make_golden import controlnet_case, sigma_weight  # noqa: E402

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
def test_controlnet_forward_vs_reference(name):
    cfg = TINY[name]
    g, fx = load_golden(f"{name}_controlnet.pt"), load_golden(f"{name}_unet_fwd.pt")
    case = controlnet_case(cfg)
    net = cldm.ControlNet(cfg, synth.synth_controlnet_state_dict(cfg, seed=6), device=DEV)
    y = fx["y"].to(DEV) if fx["y"] is not None else None
    outs = net(x=fx["x"].to(DEV), hint=case["hint_a"].to(DEV), timesteps=fx["t"].to(DEV), context=fx["ctx"].to(DEV), y=y)
    assert len(outs) == len(g["outs_every_4th_channel"])
    for i, (o, want) in enumerate(zip(outs, g["outs_every_4th_channel"])):
        assert o.dtype == torch.float16 and o.permute(0, 2, 3, 1).is_contiguous()   # NCHW view of channels-last memory
        # every residual against the worst residual of the reference's own fp16 run (oracle/make_floor.py floors_aux)
        check(f"{name} ControlNet forward, residual {i} of {len(outs)} vs reference cldm.ControlNet", o[:, ::4], want, floor=f"{name}_controlnet.pt:outs_worst")
    hint2 = case["hint_a"].to(DEV)
    a = net.guided_hint(hint2)
    assert net.guided_hint(hint2) is a  # cached per hint tensor: the hint block runs once per job, not once per step


@pytest.mark.parametrize("name", list(TINY))
def test_sampling_with_a_controlnet_chain_vs_reference(name, engines):
    cfg = TINY[name]
    g = load_golden(f"{name}_controlnet.pt")
    case = controlnet_case(cfg)
    eng = engines[name]
    cn_a = pc.ControlNet(cldm.ControlNet(cfg, synth.synth_controlnet_state_dict(cfg, seed=6), device=DEV))
    cn_b = pc.ControlNet(cldm.ControlNet(cfg, synth.synth_controlnet_state_dict(cfg, seed=9), device=DEV), global_average_pooling=True)
    unet = pc.apply_controlnet_advanced(eng.forge_objects.unet, cn_a, case["hint_a"].to(DEV), 0.8, 0.0, 0.7, positive_advanced_weighting=case["positive"],
                                        negative_advanced_weighting=case["negative"], advanced_frame_weighting=case["frame"],
                                        advanced_sigma_weighting=sigma_weight, advanced_mask_weighting=case["mask"].to(DEV))
    unet = pc.apply_controlnet_advanced(unet, cn_b, case["hint_b"].to(DEV), 0.5, 0.2, 1.0)
    assert [type(c).__name__ for c in unet.list_controlnets()] == ["ControlNet", "ControlNet"] and eng.forge_objects.unet.controlnet_linked_list is None
    saved = eng.forge_objects_after_applying_lora
    eng.forge_objects_after_applying_lora = saved.shallow_copy()
    eng.forge_objects_after_applying_lora.unet = unet
    try:
        b = len(g["euler4"]["seeds"])
        c, uc = synth.synth_conditioning(b, cfg["context_dim"], cfg.get("adm_in_channels"), seed=1234)
        if isinstance(c, dict):
            c, uc = DictWithShape({k: v.to(DEV) for k, v in c.items()}), DictWithShape({k: v.to(DEV) for k, v in uc.items()})
        else:
            c, uc = c.to(DEV), uc.to(DEV)
        shared.opts.randn_source = "CPU"
        p = processing.StableDiffusionProcessingTxt2Img(sd_model=eng, c=c, uc=uc, seed=g["euler4"]["seeds"][0], sampler_name="Euler", batch_size=b,
                                                        steps=4, cfg_scale=7.0, width=g["hw"] * 8, height=g["hw"] * 8, do_decode=False)
        lat = processing.process_images(p).latents
    finally:
        eng.forge_objects_after_applying_lora = saved
        eng.forge_objects = saved.shallow_copy()
    check(f"{name} 4-step Euler with a 2-ControlNet chain (ranges, pooling, weightings) vs reference", lat, g["euler4"]["latent"],
          floor=f"{name}_controlnet.pt:euler4/latent")


@pytest.mark.parametrize("name", list(TINY))
def test_controlnet_chain_inside_the_captured_graph_matches_eager(name, engines):
    """A chain that needs no Python per step (plain strengths and timestep ranges) runs INSIDE KModel's captured graph: the trunks read the
    UNet's own packed input / timestep buffers, the graph is keyed on the set of links active at the step's sigma.  10 Euler steps with ranges
    (0 - 0.65) and (0.3 - 1.0) of the schedule: the active set goes {a} -> {a, b} -> {b}; three graphs are captured.  Against the same job run eagerly (`use_graph` off: residuals from `get_control` every step); the two differ only in
    where 1 / sqrt(sigma^2 + 1) is rounded (pack kernel vs host), far below fp16 resolution.  A second job with another hint and another
    conditioning on the same engine must not replay anything of the first (validity: conditioning serial, guided hint, arena)."""
    cfg = TINY[name]
    g = load_golden(f"{name}_controlnet.pt")
    case = controlnet_case(cfg)
    eng = engines[name]
    km = eng.forge_objects.unet.model
    cn_a = pc.ControlNet(cldm.ControlNet(cfg, synth.synth_controlnet_state_dict(cfg, seed=6), device=DEV))
    cn_b = pc.ControlNet(cldm.ControlNet(cfg, synth.synth_controlnet_state_dict(cfg, seed=9), device=DEV))

    def job(hint_a, hint_b, cond_seed, use_graph):
        unet = pc.apply_controlnet_advanced(eng.forge_objects.unet, cn_a, hint_a, 0.8, 0.0, 0.65)
        unet = pc.apply_controlnet_advanced(unet, cn_b, hint_b, 0.5, 0.3, 1.0)
        saved = eng.forge_objects_after_applying_lora
        eng.forge_objects_after_applying_lora = saved.shallow_copy()
        eng.forge_objects_after_applying_lora.unet = unet
        was = km.use_graph
        km.use_graph = use_graph
        try:
            b = 2
            c, uc = synth.synth_conditioning(b, cfg["context_dim"], cfg.get("adm_in_channels"), seed=cond_seed)
            if isinstance(c, dict):
                c, uc = DictWithShape({k: v.to(DEV) for k, v in c.items()}), DictWithShape({k: v.to(DEV) for k, v in uc.items()})
            else:
                c, uc = c.to(DEV), uc.to(DEV)
            shared.opts.randn_source = "CPU"
            p = processing.StableDiffusionProcessingTxt2Img(sd_model=eng, c=c, uc=uc, seed=77, sampler_name="Euler", batch_size=b, steps=10,
                                                            cfg_scale=7.0, width=g["hw"] * 8, height=g["hw"] * 8, do_decode=False)
            return processing.process_images(p).latents.clone()
        finally:
            km.use_graph = was
            eng.forge_objects_after_applying_lora = saved
            eng.forge_objects = saved.shallow_copy()

    ha, hb = case["hint_a"].to(DEV), case["hint_b"].to(DEV)
    eager = job(ha, hb, 1234, False)
    n0 = len(km._graphs)
    graphed = job(ha, hb, 1234, True)
    captured = [k for k in km._graphs if "control" in k]
    assert len(km._graphs) > n0 and len({k[k.index("control"):] for k in captured}) == 3, captured   # {a}, {a, b}, {b}
    assert bool(torch.isfinite(graphed).all())
    assert max_rel(graphed, eager) < 2e-3, max_rel(graphed, eager)
    assert float((graphed - job(ha, hb, 1234, True)).abs().max()) == 0.0          # replays are deterministic
    # another job on the same engine: swapped hints, another conditioning -> nothing of the first job may be replayed
    eager2 = job(hb.clone(), ha.clone(), 99, False)
    graphed2 = job(hb.clone(), ha.clone(), 99, True)
    assert max_rel(graphed2, eager2) < 2e-3, max_rel(graphed2, eager2)
    assert max_rel(graphed2, graphed) > 1e-2                                        # (and the two jobs do differ)
    # and the first job again after the second
    assert max_rel(job(ha, hb, 1234, True), eager) < 2e-3


@pytest.mark.parametrize("name", list(TINY))
def test_control_lora_vs_reference(name, engines):
    """Control-LoRA (patcher/controlnet.py:420-474): control model assembled in pre_run from the UNet's trunk + the file's low-rank pairs, against
    the reference's ControlLora -- residuals of one call and a 4-step Euler run."""
    cfg = TINY[name]
    g, fx = load_golden(f"{name}_control_lora.pt"), load_golden(f"{name}_unet_fwd.pt")
    case = controlnet_case(cfg)
    eng = engines[name]
    cl = pc.load_controlnet(synth.synth_control_lora_state_dict(cfg), device=DEV)
    assert type(cl).__name__ == "ControlLora" and cl.control_model is None
    unet = pc.apply_controlnet_advanced(eng.forge_objects.unet, cl, case["hint_a"].to(DEV), 0.9, 0.0, 1.0)
    link = unet.controlnet_linked_list
    from forge_amd.backend.sampling.sampling_function import sampling_cleanup, sampling_prepare
    sampling_prepare(unet, None)
    y = fx["y"].to(DEV) if fx["y"] is not None else None
    outs = link.control_model(x=fx["x"].to(DEV), hint=case["hint_a"].to(DEV), timesteps=fx["t"].to(DEV), context=fx["ctx"].to(DEV), y=y)
    sampling_cleanup(unet)
    assert link.control_model is None and len(outs) == len(g["outs_every_8th_channel"])
    # the reference's ControlLora does not run in half on CPU, so this fixture has no floor of its own: it is the same network as the plain
    # ControlNet of the same size, whose floor is used
    for i, (o, want) in enumerate(zip(outs, g["outs_every_8th_channel"])):
        check(f"{name} Control-LoRA residual {i} vs reference ControlLora", o[:, ::8], want, floor=f"{name}_controlnet.pt:outs_worst")
    saved = eng.forge_objects_after_applying_lora
    eng.forge_objects_after_applying_lora = saved.shallow_copy()
    eng.forge_objects_after_applying_lora.unet = unet
    try:
        b = len(g["euler4"]["seeds"])
        c, uc = synth.synth_conditioning(b, cfg["context_dim"], cfg.get("adm_in_channels"), seed=1234)
        if isinstance(c, dict):
            c, uc = DictWithShape({k: v.to(DEV) for k, v in c.items()}), DictWithShape({k: v.to(DEV) for k, v in uc.items()})
        else:
            c, uc = c.to(DEV), uc.to(DEV)
        shared.opts.randn_source = "CPU"
        p = processing.StableDiffusionProcessingTxt2Img(sd_model=eng, c=c, uc=uc, seed=g["euler4"]["seeds"][0], sampler_name="Euler", batch_size=b,
                                                        steps=4, cfg_scale=7.0, width=g["hw"] * 8, height=g["hw"] * 8, do_decode=False)
        lat = processing.process_images(p).latents
    finally:
        eng.forge_objects_after_applying_lora = saved
        eng.forge_objects = saved.shallow_copy()
    check(f"{name} 4-step Euler with a Control-LoRA vs reference", lat, g["euler4"]["latent"], floor=f"{name}_controlnet.pt:euler4/latent")
    assert link.control_model is None   # built in sampling_prepare, dropped in sampling_cleanup


def test_t2i_adapter_vs_reference():
    """The native T2I-Adapter (three checkpoint layouts) against the reference's Adapter, and a 3-step Euler run of an SD1.5-shaped UNet with the
    adapter attached through the patcher-level T2IAdapter (features computed once, injected as 'input' residuals) against the reference's."""
    from forge_amd.backend.nn.cnets import t2i_adapter
    from oracle.make_golden import ADAPTER_VARIANTS, adapter_hint
    g = load_golden("mini_sd15_t2i_adapter.pt")
    for vname, kw in ADAPTER_VARIANTS.items():
        net = t2i_adapter.Adapter(synth.synth_t2i_adapter_state_dict(**kw), device=DEV, **kw)
        feats = net(adapter_hint(vname).to(DEV))
        want = g["features"][vname]
        assert [None if f is None else tuple(f.shape) for f in feats] == want["layout"], vname
        for i, (f, w) in enumerate(zip([f for f in feats if f is not None], want["values_every_4th_channel"])):
            check(f"T2I-Adapter {vname} feature {i} vs reference Adapter", f[:, ::4], w, floor=f"mini_sd15_t2i_adapter.pt:features/{vname}_worst")
    cfg = synth.MINI_SD15_UNET_CONFIG
    eng = build_engine(cfg, synth.synth_unet_state_dict(cfg, seed=0), None, None, device=DEV)
    kw = ADAPTER_VARIANTS["sd15_k1_pool"]
    ad = pc.load_t2i_adapter(synth.synth_t2i_adapter_state_dict(**kw), device=DEV)
    assert ad.t2i_model.ksize == 1 and not ad.t2i_model.use_conv and not ad.t2i_model.xl and ad.channels_in == 3
    unet = eng.forge_objects.unet.clone()
    unet.add_patched_controlnet(ad.copy().set_cond_hint(adapter_hint("sd15_k1_pool").to(DEV), 0.9, (0.0, 0.6)))
    eng.forge_objects_after_applying_lora = eng.forge_objects_after_applying_lora.shallow_copy()
    eng.forge_objects_after_applying_lora.unet = unet
    c, uc = synth.synth_conditioning(2, cfg["context_dim"], None, seed=1234)
    shared.opts.randn_source = "CPU"
    p = processing.StableDiffusionProcessingTxt2Img(sd_model=eng, c=c.to(DEV), uc=uc.to(DEV), seed=g["seeds"][0], sampler_name="Euler", batch_size=2, steps=3,
                                                    cfg_scale=7.0, width=g["hw"] * 8, height=g["hw"] * 8, do_decode=False)
    check("SD1.5-shaped UNet + T2I-Adapter, 3-step Euler vs reference", processing.process_images(p).latents, g["euler3"],
          floor="mini_sd15_t2i_adapter.pt:euler3")


def test_adapter_light_vs_reference():
    """`Adapter_light` (the colour adapter's network; quarter widths zero-padded to the GEMM granule) against the reference's, through the
    checkpoint loader (layout detected from the keys)."""
    from oracle.make_golden import ADAPTER_LIGHT_KW, adapter_light_hint
    g = load_golden("mini_adapter_light.pt")
    ad = pc.load_t2i_adapter(synth.synth_t2i_adapter_light_state_dict(**ADAPTER_LIGHT_KW), device=DEV)
    net = ad.t2i_model
    assert type(net).__name__ == "Adapter_light" and net.channels == ADAPTER_LIGHT_KW["channels"] and net.nums_rb == ADAPTER_LIGHT_KW["nums_rb"]
    assert (ad.channels_in, net.unshuffle_amount, net.xl) == (g["input_channels"], g["unshuffle_amount"], False)
    feats = net(adapter_light_hint().to(DEV))
    assert [None if f is None else tuple(f.shape) for f in feats] == g["layout"]
    for i, (f, w) in enumerate(zip([f for f in feats if f is not None], g["values_every_4th_channel"])):
        check(f"Adapter_light feature {i} vs reference", f[:, ::4], w, floor="mini_adapter_light.pt:features_worst")
