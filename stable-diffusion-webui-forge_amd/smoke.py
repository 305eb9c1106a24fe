"""`__graft_entry__.smoke()`: one small invocation of the hot path on cuda:0 (CFG step through the Forge call surface:
CFGDenoiser -> UNet -> sampler update -> VAE decode), checked against the CPU oracle."""
import os
import sys

import torch


def run():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from forge_amd import _lib, synth
    from forge_amd.backend.diffusion_engine.base import build_engine
    from forge_amd.modules import processing
    from oracle import pipeline  # checker only
    assert torch.cuda.is_available(), "smoke() needs the MI355X"
    _lib.lib()
    cfg = synth.TINY_SD15_UNET_CONFIG
    sd = synth.synth_unet_state_dict(cfg, seed=0)
    vsd = synth.synth_vae_decoder_state_dict(synth.TINY_VAE_CONFIG, seed=1)
    eng = build_engine(cfg, sd, synth.TINY_VAE_CONFIG, vsd, device="cuda:0")
    c, uc = synth.synth_conditioning(2, cfg["context_dim"], None, seed=1234)
    p = processing.StableDiffusionProcessingTxt2Img(sd_model=eng, c=c.cuda(), uc=uc.cuda(), seed=3, sampler_name="Euler", batch_size=2,
                                                    steps=3, width=128, height=128)
    res = processing.process_images(p)
    lat, dec, img = pipeline.txt2img(sd, cfg, vsd, synth.TINY_VAE_CONFIG, c, uc, [3, 4], 128, 128, 3, sampler_name="Euler")
    e1 = float((res.latents.cpu() - lat).abs().max() / lat.abs().max())
    e2 = float((res.decoded.cpu() - dec).abs().max() / dec.abs().max())
    # the yard-stick: what the REAL reference loses when it runs this same job in its own fp16 mode instead of fp32
    # (oracle/make_floor.py floors_pipeline -> tests/golden/fp16_floor.json); the north star's 1e-3 applies where that floor is below it
    import json
    fl = json.load(open(os.path.join(root, "tests", "golden", "fp16_floor.json")))
    f1, f2 = fl["pipeline:smoke_euler3/latent"]["max_rel"], fl["pipeline:smoke_euler3/decoded"]["max_rel"]
    l1, l2 = max(1e-3, 1.5 * f1), max(1e-3, 1.5 * f2)
    print(f"smoke: latents max_rel {e1:.3e} (reference fp16-vs-fp32 floor {f1:.3e}, limit {l1:.2e}), "
          f"decoded max_rel {e2:.3e} (floor {f2:.3e}, limit {l2:.2e})")
    assert e1 <= l1 and e2 <= l2, (e1, l1, e2, l2)
    print("smoke OK")
