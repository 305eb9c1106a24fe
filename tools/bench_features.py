#!/usr/bin/env python
"""Full-size (SDXL 1024x1024, batch 8, fp16, CFG 7) timings of the widened paths, one JSON line each: every sampler family, ControlNet,
per-block hooks (eager path), AND-composed prompts, hires-fix second pass.  Not the headline bench (bench.py): this is the check that the
§8f rows hold up at the BASELINE shape -- arena sizing, 32-bit offsets, graph capture with other batch sizes -- and what they cost.

    python tools/bench_features.py [--steps 6] [--only samplers,controlnet,hooks,and,hires]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import conftest  # noqa: F401,E402  (registers the package alias)
import forge_amd  # noqa: E402
from forge_amd import synth  # noqa: E402
from forge_amd.backend.diffusion_engine.base import build_engine  # noqa: E402
from forge_amd.backend.nn.layout import controlnet_param_shapes, unet_param_shapes  # noqa: E402
from forge_amd.modules import processing, prompt_parser as pp, shared  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--only", default="samplers,controlnet,hooks,and,hires")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    width, height = a.width or a.res, a.height or a.res
    cfg = synth.SDXL_UNET_CONFIG
    eng = build_engine(cfg, synth.synth_state_dict_device(unet_param_shapes(cfg), 0, dev), None, None, device=dev)
    b = a.batch
    c, uc = synth.synth_conditioning(2 * b, cfg["context_dim"], cfg["adm_in_channels"], seed=1234)
    half = lambda t, lo, hi: pp.DictWithShape({k: v[lo:hi].to(dev).half() for k, v in t.items()})
    c1, c2, u1 = half(c, 0, b), half(c, b, 2 * b), half(uc, 0, b)
    shared.opts.randn_source = "CPU"
    what = set(a.only.split(","))

    def run(label, sampler="Euler", steps=a.steps, cond=None, unet=None, model_calls_per_step=1, **kw):
        saved = eng.forge_objects_after_applying_lora
        if unet is not None:
            eng.forge_objects_after_applying_lora = saved.shallow_copy()
            eng.forge_objects_after_applying_lora.unet = unet
        try:
            def once(n):
                p = processing.StableDiffusionProcessingTxt2Img(sd_model=eng, c=cond if cond is not None else c1, uc=u1, seed=1, sampler_name=sampler,
                                                                batch_size=b, steps=n, cfg_scale=7.0, width=width, height=height, do_decode=False, **kw)
                return processing.process_images(p).latents
            once(max(3, min(steps, 4)))  # priming (arena, caches, graph)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lat = once(steps)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        finally:
            eng.forge_objects_after_applying_lora = saved
            eng.forge_objects = saved.shallow_copy()
        ok = bool(torch.isfinite(lat).all())
        print(json.dumps({"case": label, "sampler": sampler, "steps": steps, "ms_per_step": round(dt / steps * 1e3, 2),
                          "ms_total": round(dt * 1e3, 1), "finite": ok, "shape": list(lat.shape)}), flush=True)

    if "one" in what:
        run(f"{width}x{height}", sampler="Euler")
    if "samplers" in what:
        for s in ("Euler", "Euler a", "DPM++ 2M", "Heun", "DPM2 a", "DPM++ 2S a", "LMS", "IPNDM_V", "DEIS", "DPM++ SDE", "DPM++ 2M SDE", "DPM++ 3M SDE",
                  "DPM fast", "DDIM", "PLMS", "UniPC", "LCM", "DDPM"):
            run("sampler", sampler=s)
    if "and" in what:
        rows = lambda t, i: {k: v[i] for k, v in t.items()}
        mc = pp.MulticondLearnedConditioning((b,), [[pp.ComposableScheduledPromptConditioning([pp.ScheduledPromptConditioning(10 ** 6, rows(c1, i))], 1.0),
                                                     pp.ComposableScheduledPromptConditioning([pp.ScheduledPromptConditioning(10 ** 6, rows(c2, i))], 0.6)]
                                                    for i in range(b)])
        run("AND-composed prompt (2 conds + uncond = UNet batch 24)", cond=mc)
    if "hooks" in what:
        unet = eng.forge_objects.unet.clone()
        unet.set_model_attn2_output_patch(lambda n, extra: n * 0.98)
        unet.set_model_output_block_patch(lambda h, hsp, to: (h, hsp * 0.95))
        run("per-block hooks installed (eager, general attention path)", unet=unet)
    if "controlnet" in what:
        from forge_amd.backend.nn.cnets import cldm
        from forge_amd.backend.patcher import controlnet as pc
        cn = cldm.ControlNet(cfg, synth.synth_state_dict_device(controlnet_param_shapes(cfg), 6, dev), device=dev)
        hint = torch.rand(1, 3, height, width, device=dev)
        unet = pc.apply_controlnet_advanced(eng.forge_objects.unet, pc.ControlNet(cn), hint, 0.8, 0.0, 1.0)
        run("ControlNet (SDXL-size control model, strength 0.8), trunk inside the captured graph", unet=unet)
        km = eng.forge_objects.unet.model
        km.use_graph = False
        try:
            run("ControlNet (SDXL-size control model, strength 0.8), eager (residuals from get_control every step)", unet=unet)
        finally:
            km.use_graph = True
    if "hires" in what:
        run("hires fix 1024 -> 1536 (Latent bicubic), 6 + 4 steps", steps=6, enable_hr=True, hr_scale=1.5, hr_upscaler="Latent (bicubic)",
            hr_second_pass_steps=4, denoising_strength=0.6, hr_cfg=7.0)


if __name__ == "__main__":
    with torch.inference_mode():
        main()
