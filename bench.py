#!/usr/bin/env python
"""Headline benchmark: sampler it/s of the txt2img hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (config.workload): SDXL UNet, 1024x1024 (latent 128x128), batch 8 images per GPU, fp16, Euler sampler, CFG 7
(UNet batch 16 = [uncond ; cond]), random-init weights and synthetic conditioning (no checkpoints / datasets here).
A "step" is one sampler iteration through the Forge call surface: CFGDenoiser.forward -> sampling_function ->
KModel (pack, UNet forward, x - eps*sigma, CFG combine) -> sampler update.  W warm-up steps, then EXACTLY K steps timed
between barrier + torch.cuda.synchronize(); max over ranks; `value` = (N * K) / T = batch-steps per second over the job
(weak scaling: 8 images per GPU).  ms/image (20 sampler steps + VAE decode) is reported alongside.

Extra objects on the JSON line:
  roofline      dominant kernel (MFMA implicit-GEMM conv/linear, 88 % of the step's FLOPs): algorithmic FLOP of all its
                launches in one UNet forward / their summed HIP-event time (events recorded on the launch stream), vs the
                2.5 PFLOP/s dense fp16 MFMA peak.  `attention` carries the same for the fused attention kernel.
  cpu_baseline  the CPU oracle (oracle/, a restatement of the reference's torch code; kind "port") timed on this box's
                host cores on a bounded sample: ONE SDXL UNet sample-forward at 128x128 latent (6.76 TFLOP); a step of
                this workload is 16 such forwards, so it/s = 1 / (16 * t).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import forge_amd  # noqa: E402
from forge_amd import distributed as fdist  # noqa: E402
from forge_amd import hipops, synth  # noqa: E402
from forge_amd.backend.diffusion_engine.base import build_engine  # noqa: E402
from forge_amd.backend.nn.layout import unet_param_shapes, vae_decoder_param_shapes  # noqa: E402
from forge_amd.backend.nn.unet import IntegratedUNet2DConditionModel  # noqa: E402
from forge_amd.modules import processing, rng, sd_samplers, shared  # noqa: E402
from forge_amd.modules.prompt_parser import DictWithShape  # noqa: E402

# algorithmic FLOP per UNet sample-forward (FlopCounterMode on the reference modules)
FLOPS_PER_SAMPLE_FWD = {"sdxl": 6.7612e12, "sd15": 0.8033e12}
VAE_FLOPS_PER_IMAGE = {1024: 10.4704e12, 512: 2.5145e12}
MFMA_PEAK = 2.5e15


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="sdxl", choices=["sdxl", "sd15"])
    ap.add_argument("--res", type=int, default=0, help="image size (default 1024 for sdxl, 512 for sd15)")
    ap.add_argument("--batch", type=int, default=0, help="images per GPU (default 8 for sdxl, 4 for sd15)")
    ap.add_argument("--sampler", default="Euler")
    ap.add_argument("--cfg", type=float, default=7.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-vae", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--breakdown", default="", help="write the per-shape kernel-time table of one UNet forward to this file")
    return ap.parse_args()


def cpu_baseline(model, cfg, latent):
    """Oracle timed on host cores (test-infrastructure import allowed for this leg only)."""
    from oracle.unet import unet_forward
    nthreads = torch.get_num_threads()
    t0 = time.time()
    sd = {}
    g = torch.Generator().manual_seed(0)
    for name, shape in unet_param_shapes(cfg).items():
        sd[name] = torch.empty(shape).normal_(0, 0.02, generator=g)
    t_init = time.time() - t0
    x = torch.randn(1, cfg["in_channels"], latent, latent)
    ctx = torch.randn(1, 77, cfg["context_dim"])
    y = torch.randn(1, cfg["adm_in_channels"]) if cfg.get("adm_in_channels") else None
    t1 = time.time()
    unet_forward(sd, cfg, x, torch.tensor([500.0]), ctx, y)
    dt = time.time() - t1
    return dt, nthreads, t_init


def pmc_traffic_per_launch():
    """HBM bytes per launch of the GEMM kernels from the committed rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE
    runs of this same workload, tools/gpu_round.sh pmc -> profiles/*_pmc_fetch_write_summary.json; FETCH_SIZE doubled per
    MI355X_MICROARCH.md).  bench.py cannot run the profiler on itself, so this is the last committed measurement or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_fetch_write_summary.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        n = b = 0.0
        for fam in ("gemm", "gemm256"):
            if fam in d:
                n += d[fam]["launches_FETCH_SIZE"]
                b += d[fam]["hbm_bytes_per_launch"] * d[fam]["launches_FETCH_SIZE"]
        return {"hbm_bytes_per_launch_avg": round(b / n), "source": os.path.basename(files[-1])} if n else None
    except Exception:
        return None


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    model = a.model
    res = a.res or (1024 if model == "sdxl" else 512)
    bpg = a.batch or (8 if model == "sdxl" else 4)
    ucfg = synth.SDXL_UNET_CONFIG if model == "sdxl" else synth.SD15_UNET_CONFIG
    vcfg = synth.SDXL_VAE_CONFIG if model == "sdxl" else synth.SD15_VAE_CONFIG
    latent = res // 8

    # ---- model: random-init weights drawn on the device ---------------------------------------------------------
    t0 = time.time()
    usd = synth.synth_state_dict_device(unet_param_shapes(ucfg), 0, dev)
    vsd = None if a.no_vae else synth.synth_state_dict_device(vae_decoder_param_shapes(vcfg), 1, dev)
    IntegratedUNet2DConditionModel.RETAIN_TRUNK_WEIGHTS = False  # no Control-LoRA in this run: do not keep a second copy of the encoder weights
    eng = build_engine(ucfg, usd, None if a.no_vae else vcfg, vsd, device=dev)
    del usd, vsd
    eng.forge_objects.unet.model.use_graph = not a.no_graph
    torch.cuda.synchronize()
    t_build = time.time() - t0

    # ---- conditioning: rank 0 owns the global batch, RCCL broadcast, each rank keeps its shard -------------------
    total = bpg * world
    t0 = time.time()
    if rank == 0:
        c, uc = synth.synth_conditioning(total, ucfg["context_dim"], ucfg.get("adm_in_channels"), seed=1234)
        to_dev = lambda t: t.to(dev).half()
        c = {k: to_dev(v) for k, v in c.items()} if isinstance(c, dict) else to_dev(c)
        uc = {k: to_dev(v) for k, v in uc.items()} if isinstance(uc, dict) else to_dev(uc)
    else:
        c = uc = None
    c, uc = fdist.broadcast_conditioning(c, uc, dev)
    lo, hi = fdist.shard_range(total, rank, world)
    c, uc = fdist.slice_conditioning(c, lo, hi), fdist.slice_conditioning(uc, lo, hi)
    if isinstance(c, dict):
        c, uc = DictWithShape(c), DictWithShape(uc)
    torch.cuda.synchronize()
    t_bcast = time.time() - t0

    shared.opts.randn_source = "CPU"
    seeds = [1000 + i for i in range(lo, hi)]

    def make_p(steps):
        p = processing.StableDiffusionProcessingTxt2Img(sd_model=eng, c=c, uc=uc, seed=seeds[0], sampler_name=a.sampler, batch_size=bpg,
                                                        steps=steps, cfg_scale=a.cfg, width=res, height=res)
        p.seeds = seeds
        p.all_seeds = seeds
        p.rng = rng.ImageRNG((4, latent, latent), seeds, device=dev)
        return p

    def run_sampler(steps):
        p = make_p(steps)
        sampler = sd_samplers.create_sampler(a.sampler, eng)
        p.sampler = sampler
        x = p.rng.next()
        return sampler.sample(p, x, c, uc, steps=steps, image_conditioning=p.txt2img_image_conditioning(x))

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    with torch.inference_mode():
        run_sampler(3)                       # priming: sizes the arena, builds caches, captures the HIP graph
        if a.warmup > 0:
            run_sampler(a.warmup)            # W untimed warm-up steps
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lat = run_sampler(a.steps)           # exactly K timed steps
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([elapsed], device=dev)
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
            elapsed = float(tmax.item())

        # ---- VAE decode (per-job, outside the step loop) and latent gather -------------------------------------
        vae_ms = None
        if not a.no_vae:
            eng.decode_first_stage(lat)          # untimed: sizes the VAE arena for this batch
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            img = eng.decode_first_stage(lat)
            torch.cuda.synchronize()
            vae_ms = (time.perf_counter() - t0) * 1e3
            del img
        t0 = time.perf_counter()
        gathered = fdist.gather_latents(lat, total)
        torch.cuda.synchronize()
        t_gather = time.perf_counter() - t0

        # ---- roofline of the dominant kernel: HIP events around every launch in one eager UNet forward ----------
        roof = attn_roof = None
        if rank == 0 and not a.no_roofline:
            km = eng.forge_objects.unet.model
            km.use_graph = False
            x = torch.randn(bpg, 4, latent, latent, device=dev)
            sig = torch.full((bpg,), 5.0, device=dev)
            uctx = (uc["crossattn"], uc["vector"]) if isinstance(uc, dict) else (uc, None)
            cctx = (c["crossattn"], c["vector"]) if isinstance(c, dict) else (c, None)
            km.denoise_cfg(x, sig, uctx, cctx, a.cfg)
            torch.cuda.synchronize()
            with hipops.KernelProfiler() as prof:
                km.denoise_cfg(x, sig, uctx, cctx, a.cfg)
                torch.cuda.synchronize()
                summ = prof.summary()
            if a.breakdown:
                rows = sorted(prof.by_tag.items(), key=lambda kv: -kv[1]["seconds"])
                with open(a.breakdown, "w") as f:
                    for (kind, tag), d in rows:
                        f.write(json.dumps({"kind": kind, "shape": tag, "launches": d["launches"], "ms": round(d["seconds"] * 1e3, 3),
                                            "tflops": round(d["flops"] / d["seconds"] / 1e12, 1)}) + "\n")
            km.use_graph = not a.no_graph
            g = summ.get("gemm_conv")
            if g:
                ach = g["flops"] / g["seconds"]
                traffic = pmc_traffic_per_launch()
                roof = {"kernel": "gemm256p_kernel<256x320 | 320x256 | 256x256> + gemm_kernel (fmx_gemm_conv_f16: MFMA implicit-GEMM conv3x3/1x1 + linear, fused epilogues)",
                        "bound": "mfma", "achieved": round(ach / 1e12, 1), "peak": MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                        "frac": round(ach / MFMA_PEAK, 4), "traffic": traffic, "launches_per_forward": g["launches"],
                        "flop_per_launch_avg": round(g["flops"] / g["launches"] / 1e9, 2), "flop_unit": "GFLOP",
                        "us_per_launch_avg": round(g["seconds"] / g["launches"] * 1e6, 1),
                        "kernel_time_per_forward_ms": round(g["seconds"] * 1e3, 2)}
            at = summ.get("attention")
            if at:
                ach = at["flops"] / at["seconds"]
                attn_roof = {"kernel": "attn_q64_kernel (fmx_attention_f16: fused QK^T-softmax-PV)", "bound": "mfma",
                             "achieved": round(ach / 1e12, 1), "peak": MFMA_PEAK / 1e12, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK, 4),
                             "launches_per_forward": at["launches"], "kernel_time_per_forward_ms": round(at["seconds"] * 1e3, 2)}

    ms_per_step = elapsed / a.steps * 1e3
    value = world * a.steps / elapsed
    fl = FLOPS_PER_SAMPLE_FWD[model] * 2 * bpg  # per GPU per step (CFG: 2 sample-forwards per image)
    out = {
        "metric": "sampler it/s (UNet steps/sec)", "value": round(value, 4), "unit": "it/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic (random-init weights, N(0,1) conditioning, per-image seeded CPU noise)",
        "config": {"workload": f"{'SDXL' if model == 'sdxl' else 'SD1.5'} UNet {res}x{res} batch={bpg}/GPU fp16, {a.sampler} sampler, CFG {a.cfg} "
                               f"(UNet batch {2 * bpg}), HIP-graph replay={'on' if not a.no_graph else 'off'}",
                   "global_batch": total, "latent": [latent, latent], "parallelism": f"batch-shard x{world} (RCCL broadcast cond + gather latents)"},
        "step_tflops_per_gpu": round(fl / 1e12, 2),
        "achieved_tflops_per_gpu": round(fl / (ms_per_step * 1e-3) / 1e12, 1),
        "step_frac_of_mfma_peak": round(fl / (ms_per_step * 1e-3) / MFMA_PEAK, 4),
        "vae_decode_ms_per_batch": None if vae_ms is None else round(vae_ms, 1),
        "ms_per_image_20_steps_plus_vae": None if vae_ms is None else round((20 * ms_per_step + vae_ms) / bpg, 1),
        "comm_ms": {"broadcast_cond": round(t_bcast * 1e3, 2), "gather_latents": round(t_gather * 1e3, 2)},
        "build_s": round(t_build, 1),
    }
    if roof:
        out["roofline"] = roof
    if attn_roof:
        out["roofline_attention"] = attn_roof
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            dt, nthreads, t_init = cpu_baseline(model, ucfg, latent)
            out["cpu_baseline"] = {"value": round(1.0 / (2 * bpg * dt), 6), "unit": "it/s", "cores": nthreads, "kind": "port",
                                   "sample": f"one {model.upper()} UNet sample-forward (B=1, {latent}x{latent} latent, fp32, torch CPU, {nthreads} threads) took "
                                             f"{dt:.1f} s; a step of this workload is {2 * bpg} such forwards"}
        except Exception as e:  # the baseline must never take the bench line down
            out["cpu_baseline"] = {"value": None, "error": repr(e)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
