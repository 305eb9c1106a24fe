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
    print(f"smoke: latents max_rel {e1:.3e}, decoded max_rel {e2:.3e}")
    assert e1 < 1e-2 and e2 < 1e-2, (e1, e2)
    print("smoke OK")
