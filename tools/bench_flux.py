#!/usr/bin/env python
"""Flux-dev (BASELINE config 5 per-GPU shape) at FULL size on one MI355X: 11.9 B parameters fp16 or bf16 (23.8 GB, resident), 1024x1024
(4096 image tokens) + 256 text tokens, batch 2, one transformer forward = one sampler step (distilled guidance: no CFG batch).
Prints one JSON line: ms per forward, achieved TFLOP/s against the 69.47 TFLOP per sample-forward of SURVEY 8d.

    python tools/bench_flux.py [--batch 2] [--steps 4] [--dtype bf16]
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
import conftest  # noqa: F401,E402
from forge_amd import synth  # noqa: E402
from forge_amd.backend.nn.flux import IntegratedFluxTransformer2DModel  # noqa: E402
from forge_amd.backend.nn.layout import flux_param_shapes  # noqa: E402

FLOP_PER_SAMPLE_FWD = 69.47e12  # 1024^2, 256 text tokens


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--ltxt", type=int, default=256)
    ap.add_argument("--dtype", choices=("fp16", "bf16"), default="fp16", help="bf16 = the reference's Flux compute type (bfloat16 build of the kernels)")
    a = ap.parse_args()
    dt16 = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    dev = torch.device("cuda", 0)
    cfg = synth.FLUX_DEV_CONFIG
    t0 = time.time()
    sd = synth.synth_state_dict_device(flux_param_shapes(cfg), 2, dev, dtype=dt16)
    net = IntegratedFluxTransformer2DModel(cfg, sd, device=dev, dtype=dt16)
    del sd
    torch.cuda.synchronize()
    t_build = time.time() - t0
    b = a.batch
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(b, 16, 128, 128, device=dev, generator=g)
    ctx = torch.randn(b, a.ltxt, cfg["context_in_dim"], device=dev, generator=g).to(dt16)
    y = torch.randn(b, cfg["vec_in_dim"], device=dev, generator=g).to(dt16)
    guidance = torch.full((b,), 3.5, device=dev)
    ts = torch.full((b,), 0.7, device=dev)
    with torch.inference_mode():
        out = net.forward(x, ts, ctx, y, guidance)          # warm-up (buffers, rope table)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = net.forward(x, ts, ctx, y, guidance)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"case": f"Flux-dev transformer forward, 1024x1024, {a.dtype}", "batch": b, "text_tokens": a.ltxt, "ms_per_forward": round(dt * 1e3, 2),
                      "ms_per_image_step": round(dt * 1e3 / b, 2), "achieved_tflops": round(FLOP_PER_SAMPLE_FWD * b / dt / 1e12, 1),
                      "frac_of_mfma_peak": round(FLOP_PER_SAMPLE_FWD * b / dt / 2.5e15, 4), "finite": bool(torch.isfinite(out).all()),
                      "out_shape": list(out.shape), "params_GB": 23.8, "build_s": round(t_build, 1),
                      "hbm_GB_after": round(torch.cuda.memory_allocated(dev) / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    main()
