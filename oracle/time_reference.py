"""TEST INFRASTRUCTURE ONLY.  The REAL reference (/root/reference through oracle/ref_import.py) timed on the authoring box's host cores for the headline
configuration's sampler path: SDXL 1024x1024, ONE image, Euler, CFG 7 (cond + uncond per step) through the reference's own CFGDenoiser-level loop
(k_diffusion sample_euler -> sampling_function -> KModel.apply_model -> IntegratedUNet2DConditionModel, CPU fp32).  The GPU box has no /root/reference, so
bench.py times the oracle PORT there (cpu_baseline.kind = "port") and carries THIS file's numbers beside it as `reference_on_authoring_box`.

    python -m oracle.time_reference [steps=2]   ->  profiles/cpu_reference_sdxl_b1.json
"""
import json
import os
import platform
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import forge_amd  # noqa: E402,F401
from forge_amd import synth  # noqa: E402
from oracle import make_golden as mg  # noqa: E402
from oracle import ref_import  # noqa: E402


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    cfg = synth.SDXL_UNET_CONFIG
    net = ref_import.build_ref_unet(cfg, synth.synth_unet_state_dict(cfg, seed=0))
    c, uc = synth.synth_conditioning(1, cfg["context_dim"], cfg["adm_in_channels"], seed=1234)
    c, uc = ref_import.SdxlCond(c), ref_import.SdxlCond(uc)
    mg.ref_sample(net, cfg, c, uc, [1000], 128, 1, "Euler")          # warm-up: one step (allocator, thread pool)
    t0 = time.time()
    mg.ref_sample(net, cfg, c, uc, [1000], 128, steps, "Euler")
    dt = (time.time() - t0) / steps
    out = {"what": "REAL reference (lllyasviel/stable-diffusion-webui-forge: k_diffusion.sample_euler -> sampling_function -> KModel -> UNet) on CPU, fp32",
           "config": "SDXL UNet (2.57 B parameters, random init), 1024x1024 (latent 128x128), batch 1, Euler, CFG 7.0: two sample-forwards per step",
           "steps_timed": steps, "seconds_per_step": round(dt, 2), "it_per_s_b1_cfg": round(1.0 / dt, 5),
           "it_per_s_at_batch_8_extrapolated": round(1.0 / (8 * dt), 6), "cores": os.cpu_count(), "torch_threads": torch.get_num_threads(),
           "cpu": cpu_model(), "torch": torch.__version__,
           "note": "authoring container, not the GPU box: the reference is not present there; bench.py quotes this as cpu_baseline.reference_on_authoring_box"}
    path = os.path.join(ROOT, "profiles", "cpu_reference_sdxl_b1.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
