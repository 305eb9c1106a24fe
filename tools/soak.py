"""Repeatability soak on the bench workload: the same txt2img job (SDXL 1024^2, batch 8 and batch 1, Euler a so that the per-step noise path
runs too) several times on one engine -- graph replay, GroupNorm statistics from GEMM epilogues, folded LayerNorms, split-K hand-overs.
Every run must give finite latents that are BIT-IDENTICAL to the first run of its configuration (all reductions are ordered; there are no
floating-point atomics), also after a run of the other batch size has re-used the arena and the split-K workspace in between.
usage: python tools/soak.py [runs] [steps]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import forge_amd  # noqa
from forge_amd import synth
from forge_amd.backend.diffusion_engine.base import build_engine
from forge_amd.backend.nn.layout import unet_param_shapes
from forge_amd.backend.nn.unet import IntegratedUNet2DConditionModel
from forge_amd.modules import processing, shared
from forge_amd.modules.prompt_parser import DictWithShape

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda", 0)
cfg = synth.SDXL_UNET_CONFIG
IntegratedUNet2DConditionModel.RETAIN_TRUNK_WEIGHTS = False
eng = build_engine(cfg, synth.synth_state_dict_device(unet_param_shapes(cfg), 0, dev), None, None, device=dev)
shared.opts.randn_source = "CPU"
if os.environ.get("SOAK_NO_GRAPH") == "1":
    eng.forge_objects.unet.model.use_graph = False
first = {}
for r in range(runs):
    for b in (8, 1):
        c, uc = synth.synth_conditioning(b, cfg["context_dim"], cfg.get("adm_in_channels"), seed=1234)
        c = DictWithShape({k: v.to(dev).half() for k, v in c.items()})
        uc = DictWithShape({k: v.to(dev).half() for k, v in uc.items()})
        p = processing.StableDiffusionProcessingTxt2Img(sd_model=eng, c=c, uc=uc, seed=4000, sampler_name="Euler a", batch_size=b, steps=steps,
                                                        cfg_scale=7.0, width=1024, height=1024, do_decode=False)
        lat = processing.process_images(p).latents
        torch.cuda.synchronize()
        ok = bool(torch.isfinite(lat).all())
        same = True if b not in first else bool(torch.equal(lat, first[b]))
        first.setdefault(b, lat.clone())
        print(json.dumps({"run": r, "batch": b, "steps": steps, "finite": ok, "bit_identical_to_first_run": same, "latent_std": round(float(lat.std()), 4)}), flush=True)
        assert ok and same
print("soak OK")
