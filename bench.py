#!/usr/bin/env python
"""Headline benchmark: sampler it/s of the txt2img hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config NAME]

N > 1: one rank per GPU over RCCL.  Either the caller launches the ranks (`python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N`, what the driver does: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment) or, when WORLD_SIZE is not set,
bench.py re-executes ITSELF under torch.distributed.run with N ranks on 127.0.0.1.  Either way every rank asserts that the process group it
joined has exactly N members and rank 0 prints the number of ranks an all-reduce actually saw (`ranks_in_collective`); a mismatch is fatal.

Workloads (--config; `config.workload` names the one that ran; BASELINE.json `configs`):
  sdxl-b8-euler20        (default) SDXL UNet 1024x1024 (latent 128x128), batch 8 / GPU, fp16, Euler, CFG 7 (UNet batch 16) -- the
                         configuration BASELINE.json's metric is quoted on (SDXL 1024^2 20-step Euler, 1/2/4/8 GPU)
  sd15-b4-eulera         configs[1]: SD1.5 512x512, batch 4, fp16, Euler a
  sdxl-b8-dpmpp2m30-vae  configs[2]: SDXL 1024x1024, batch 8, fp16, DPM++ 2M (Karras), 30 nominal steps, VAE decode in ms/image
  flux-b2-bf16           configs[4] per-GPU shard (16 images / 8 GPUs): Flux-dev DiT 1024x1024 (4096 + 256 tokens), batch 2, bf16, Euler on
                         the flow schedule, distilled guidance (no CFG batch)
Random-init weights and synthetic conditioning (no checkpoints / datasets here).  A "step" is one sampler iteration through the Forge
call surface: CFGDenoiser.forward -> sampling_function -> KModel (pack, network forward, prediction -> denoised, CFG combine) -> sampler
update.  W warm-up steps, then EXACTLY K steps timed between barrier + torch.cuda.synchronize(); max over ranks; `value` = (N * K) / T =
batch-steps per second over the job (weak scaling: the per-GPU batch is fixed).  ms/image (nominal steps + VAE decode) is reported alongside.

Extra objects on the JSON line:
  roofline      dominant kernel (MFMA implicit-GEMM conv/linear, 88 % of the step's FLOPs): algorithmic FLOP of all its launches in one
                network forward / their summed HIP-event time (events recorded on the launch stream), vs the 2.5 PFLOP/s dense fp16/bf16
                MFMA peak.  `roofline_attention` carries the same for the fused attention kernel, `roofline_groupnorm` the HBM-side one for
                GroupNorm(+SiLU) (algorithmic bytes = 1 read + 1 write of the tensor, vs 8 TB/s).
  torch_rocm_baseline  the reference's own op stack on the SAME GPU (stock PyTorch-ROCm fp16 kernels + F.scaled_dot_product_attention running the oracle's
                walk of the SDXL UNet at batch 16): ms per forward, it/s-equivalent; `vs_baseline` = native / that (context, not the target).
  cpu_baseline  the CPU oracle (oracle/, a restatement of the reference's torch code pinned to the real reference; kind "port") timed on
                this box's host cores on a bounded sample: ONE sample-forward of the workload's network (B = 1; Flux: 2 + 2 blocks of the
                57, extrapolated by block count); a step is 2 * batch (CFG) such forwards.  `reference_on_authoring_box` repeats the figure
                BASELINE.md measured with the reference's own modules (8 vCPU), for scale.

Test hook (tests/test_bench_launcher.py): `--stub-engine` replaces the workload by a few CPU flops over the gloo backend so that the launcher,
rank accounting and JSON contract can be exercised at world_size 2 without a GPU; its line says `"data": "stub"` and is not a measurement.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic FLOP per sample-forward (FlopCounterMode on the reference modules).  `step_tflops_per_gpu` / `step_frac_of_mfma_peak` are quoted on
# THIS count (the work the reference's formulation of the network does per step); `roofline.achieved` counts the multiply-adds the kernels EXECUTE, which since round 6
# are fewer: the Upsample convolutions run as four 2 x 2 phase convolutions = 4 / 9 of the reference form's (SDXL: -1.1 % of a forward's FLOP)
FLOPS_PER_SAMPLE_FWD = {"sdxl": 6.7612e12, "sd15": 0.8033e12, "flux": 69.47e12}
MFMA_PEAK = 2.5e15
HBM_PEAK = 8.0e12

CONFIGS = {
    "sdxl-b8-euler20": dict(model="sdxl", res=1024, batch=8, sampler="Euler", scheduler=None, nominal_steps=20, cfg=7.0, dtype="f16",
                            baseline="metric config: SDXL 1024^2 20-step Euler"),
    "sd15-b4-eulera": dict(model="sd15", res=512, batch=4, sampler="Euler a", scheduler=None, nominal_steps=20, cfg=7.0, dtype="f16",
                           baseline="configs[1]: SD1.5 512x512 batch=4 fp16, 20-step Euler-a"),
    "sdxl-b8-dpmpp2m30-vae": dict(model="sdxl", res=1024, batch=8, sampler="DPM++ 2M", scheduler=None, nominal_steps=30, cfg=7.0, dtype="f16",
                                  baseline="configs[2]: SDXL 1024x1024 batch=8 fp16, 30-step DPM++ 2M, VAE decode"),
    "flux-b2-bf16": dict(model="flux", res=1024, batch=2, sampler="Euler", scheduler="simple", nominal_steps=20, cfg=1.0, dtype="bf16",
                         baseline="configs[4] per-GPU shard: Flux.1-dev DiT 1024x1024 batch=16 over 8 GPUs, bf16"),
}
# reference modules timed on the authoring box (8 vCPU Xeon 2.1 GHz, fp32, torch 2.10): it/s of one UNet step at B = 1 with CFG
REFERENCE_CPU_AUTHORING_BOX = {"sd15": {"it_per_s_b1_cfg": 0.468, "cores": 8, "source": "BASELINE.md section 2 (reference modules, SD1.5 512^2 B=1 20-step Euler)"}}


def reference_on_authoring_box(model):
    """The REAL reference's own CPU path for this workload's network, timed where /root/reference exists (the authoring container; the GPU box has only
    the oracle port): profiles/cpu_reference_sdxl_b1.json is written by oracle/time_reference.py (a sampler run: Euler, CFG 7, batch 1); the batch-8 figure
    is the reference's 5-step DPM++ 2M run that produced tests/golden/sdxl_config3_b8.pt (oracle/make_floor.py gen_config3_b8, 874 s incl. the model build)."""
    if model == "sdxl":
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", "cpu_reference_sdxl_b1.json")))
            return {"it_per_s_b1_cfg": d["it_per_s_b1_cfg"], "seconds_per_step_b1": d["seconds_per_step"], "cores": d["cores"], "cpu": d["cpu"],
                    "it_per_s_b8_cfg_measured": round(5 / 874.0, 5), "source": "profiles/cpu_reference_sdxl_b1.json (oracle/time_reference.py: the real reference, "
                    "SDXL 1024^2 B=1 Euler CFG 7, fp32) + the batch-8 fixture run of oracle/make_floor.py gen_config3_b8"}
        except Exception:  # noqa: BLE001
            return None
    return REFERENCE_CPU_AUTHORING_BOX.get(model)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="sdxl-b8-euler20", choices=sorted(CONFIGS))
    ap.add_argument("--model", default=None, choices=["sdxl", "sd15", "flux"], help="override the workload's network")
    ap.add_argument("--res", type=int, default=0, help="override the image size")
    ap.add_argument("--batch", type=int, default=0, help="override images per GPU")
    ap.add_argument("--sampler", default=None)
    ap.add_argument("--cfg", type=float, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-vae", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--breakdown", default="", help="write the per-shape kernel-time table of one network forward to this file")
    ap.add_argument("--vae-breakdown", default="", help="write the per-shape kernel-time table of one VAE decode to this file")
    ap.add_argument("--rccl-selfcheck", action="store_true", help="(child process of the 1-GPU run) RCCL at world size 1: init, all-reduce, the job's broadcast + gathers on the device")
    ap.add_argument("--no-rccl-selfcheck", action="store_true", help="1-GPU run: skip the RCCL world-1 leg")
    ap.add_argument("--torch-rocm-baseline", action="store_true", help="(child process of the 1-GPU run) the reference's op stack -- stock PyTorch-ROCm fp16 -- on the same workload and GPU")
    ap.add_argument("--no-torch-rocm-baseline", action="store_true", help="1-GPU run: skip the PyTorch-ROCm leg")
    ap.add_argument("--no-other-configs", action="store_true", help="default 1-GPU run: skip the two small-batch legs (BASELINE config 2 and SDXL batch 1) run after the timed region")
    ap.add_argument("--stub-engine", action="store_true", help="launcher self-test without a GPU (gloo, no kernels); not a measurement")
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def write_breakdown(path, prof):
    """Per-(kind, shape) kernel-time table of the launches a hipops.KernelProfiler saw, slowest first, one JSON object per line."""
    rows = sorted(prof.by_tag.items(), key=lambda kv: -kv[1]["seconds"])
    with open(path, "w") as f:
        for (kind, tag), d in rows:
            f.write(json.dumps({"kind": kind, "shape": tag, "launches": d["launches"], "ms": round(d["seconds"] * 1e3, 3),
                                "tflops": round(d["flops"] / d["seconds"] / 1e12, 1) if d["flops"] else None,
                                "GBps": round(d.get("bytes", 0.0) / d["seconds"] / 1e9, 1) if d.get("bytes") else None}) + "\n")


def self_launch(a):
    """--gpus N > 1 without a launcher: become `torch.distributed.run --nproc-per-node N bench.py <same flags>` on 127.0.0.1."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this host driver
    env["FMX_BENCH_SELF_LAUNCHED"] = "1"
    sys.exit(subprocess.call(cmd, env=env))


def cpu_baseline(model, cfg, latent, sampler_name=None, cfg_scale=7.0):
    """Oracle timed on host cores (test-infrastructure import allowed for this leg only).  -> (seconds per sample-forward, threads, note)"""
    import torch
    from forge_amd.backend.nn.layout import unet_param_shapes
    # the port is synchronisation-bound long before 128 threads: on the GPU box (128 cores) one SDXL forward takes 9.6 s on 16 threads, 11.5 s on 32, 17.4 s
    # on 64 and 31.3 s on torch's default of all 128 (profiles/r14b_cpu_baseline_thread_count.jsonl; the REAL reference on the authoring container's 8 cores:
    # 24.8 s per forward) -- the baseline is quoted at the count that is fastest, and `cores` says which
    want = min(16, os.cpu_count() or 8)
    if os.environ.get("FMX_BENCH_CPU_THREADS"):
        want = int(os.environ["FMX_BENCH_CPU_THREADS"])
    torch.set_num_threads(want)
    nthreads = torch.get_num_threads()
    g = torch.Generator().manual_seed(0)
    if model == "flux":
        from forge_amd.backend.nn.layout import flux_param_shapes
        from oracle.flux import flux_forward
        small = dict(cfg, depth=2, depth_single_blocks=2)
        sd = {name: torch.empty(shape).normal_(0, 0.02, generator=g) for name, shape in flux_param_shapes(small).items()}
        h = latent
        x = torch.randn(1, 16, h, h)
        ctx = torch.randn(1, 256, cfg["context_in_dim"])
        y = torch.randn(1, cfg["vec_in_dim"])
        t1 = time.time()
        flux_forward(sd, small, x, torch.tensor([0.7]), ctx, y, torch.tensor([3.5]))
        dt = time.time() - t1
        scale = (cfg["depth"] * 2 + cfg["depth_single_blocks"]) / (2 * 2 + 2)   # a double block is two streams' worth of a single block
        return dt * scale, nthreads, (f"Flux-dev at 2 double + 2 single blocks of the {cfg['depth']} + {cfg['depth_single_blocks']} (full width 3072, 4096 + 256 tokens, "
                                      f"B=1, fp32, torch CPU, {nthreads} threads) took {dt:.1f} s; scaled x{scale:.1f} by block count")
    from oracle.unet import unet_forward
    sd = {name: torch.empty(shape).normal_(0, 0.02, generator=g) for name, shape in unet_param_shapes(cfg).items()}
    if model == "sd15" and sampler_name is not None:
        # a SAMPLER RUN, not one forward: the oracle's txt2img loop (CFGDenoiser -> UNet -> sampler update, same sampler as the GPU line) for
        # one image over `cpu_steps` steps; seconds per step / 2 = seconds per sample-forward, so `value` below is a measured it/s scaled
        # only by the batch (images are independent), not by an extrapolated step count
        from oracle import pipeline
        from forge_amd import synth
        c1, u1 = synth.synth_conditioning(1, cfg["context_dim"], cfg.get("adm_in_channels"), seed=1234)
        cpu_steps = 4
        t1 = time.time()
        pipeline.txt2img_latents(sd, cfg, c1, u1, [1000], latent * 8, latent * 8, cpu_steps, sampler_name=sampler_name, cfg_scale=cfg_scale)
        dt = (time.time() - t1) / cpu_steps
        return dt / 2, nthreads, (f"oracle sampler run: SD1.5 512^2, ONE image, {cpu_steps}-step {sampler_name} with CFG {cfg_scale} (2 sample-forwards per step), fp32, "
                                  f"torch CPU, {nthreads} threads: {dt:.2f} s per step = {1.0 / dt:.3f} it/s at batch 1")
    x = torch.randn(1, cfg["in_channels"], latent, latent)
    ctx = torch.randn(1, 77, cfg["context_dim"])
    y = torch.randn(1, cfg["adm_in_channels"]) if cfg.get("adm_in_channels") else None
    t1 = time.time()
    unet_forward(sd, cfg, x, torch.tensor([500.0]), ctx, y)
    dt = time.time() - t1
    return dt, nthreads, (f"one {model.upper()} UNet sample-forward (B=1, {latent}x{latent} latent, fp32, torch CPU, {nthreads} threads) took {dt:.1f} s")


def kernel_source_hash():
    """Identity of the kernel sources THE LOADED BINARY was built from (fmx_build_info(): the Makefile bakes csrc/src_hash.py's sha256 over csrc/*.hip,
    *.hpp and include/fmx.h into libfmx_gfx950.so) -- not of whatever sources lie next to it: a stale .so beside newer sources is not credited with
    measurements taken on them.  `source_tree_hash()` is the hash of the sources on disk; the bench line carries both."""
    from forge_amd import _lib
    return _lib.build_info().get("src", "unknown")


def source_tree_hash():
    import importlib.util
    spec = importlib.util.spec_from_file_location("fmx_src_hash", os.path.join(ROOT, "stable-diffusion-webui-forge_amd", "csrc", "src_hash.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.kernel_source_hash()


def _knobs(ignored):
    from forge_amd import _lib
    return _lib.active_knobs(ignored)


def pmc_traffic_per_launch():
    """HBM bytes per launch of the GEMM kernels from the committed rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE
    runs of this same workload, tools/gpu_round2.sh pmc_hbm -> profiles/*_pmc_fetch_write_summary.json; FETCH_SIZE doubled per
    MI355X_MICROARCH.md).  bench.py cannot run the profiler on itself, so this is the last committed measurement -- and ONLY if that
    summary was taken on the kernel sources being timed now (`kernel_source_hash` recorded by tools/pmc_summary.py): a summary of an
    older binary is refused (traffic null, the reason beside it) rather than quoted as this binary's traffic."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_fetch_write_summary.json")))
    if not files:
        return None, "no committed PMC summary"
    try:
        d = json.load(open(files[-1]))
        src = d.get("kernel_source_hash")
        if src != kernel_source_hash():
            return None, f"{os.path.basename(files[-1])} was measured on kernel sources {src}, this binary is {kernel_source_hash()}: stale, refused"
        n = b = 0.0
        for fam in ("gemm", "gemm256"):
            if fam in d:
                n += d[fam]["launches_FETCH_SIZE"]
                b += d[fam]["hbm_bytes_per_launch"] * d[fam]["launches_FETCH_SIZE"]
        if not n:
            return None, "summary holds no GEMM launches"
        return {"hbm_bytes_per_launch_avg": round(b / n), "source": os.path.basename(files[-1]), "kernel_source_hash": src}, None
    except Exception as e:
        return None, repr(e)


class ClockSampler:
    """Shader / memory clocks of the timed region, read from the amdgpu sysfs DPM tables (the `*` line of pp_dpm_sclk / pp_dpm_mclk) every
    20 ms on a host thread, and the socket power beside them (hwmon power1_average / power1_input against power1_cap, microwatts): puts the
    'power-limited at ~1.8 GHz' statement into the driver-visible line.  Null fields when the files are absent."""

    def __init__(self, local):
        import glob
        self.files = {}
        self.note = None
        self.power_cap_w = None
        try:   # the sysfs card of THIS HIP device: match the PCI address (the host may expose more cards than the container sees GPUs)
            import torch
            pr = torch.cuda.get_device_properties(local)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
            for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
                base = os.path.dirname(f)
                if want in os.path.realpath(base).lower():
                    self.files = {"sclk": os.path.join(base, "pp_dpm_sclk"), "mclk": os.path.join(base, "pp_dpm_mclk")}
                    self.note = "sysfs " + base + " (PCI " + want + ")"
                    for hw in sorted(glob.glob(os.path.join(base, "hwmon", "hwmon*"))):
                        for name in ("power1_average", "power1_input"):
                            if os.path.exists(os.path.join(hw, name)) and "power" not in self.files:
                                self.files["power"] = os.path.join(hw, name)
                                try:
                                    self.power_cap_w = int(open(os.path.join(hw, "power1_cap")).read()) / 1e6
                                except Exception:
                                    pass
            if not self.files:
                self.note = "no sysfs card with PCI address " + want
        except Exception as e:
            self.note = repr(e)
        self.samples = {k: [] for k in self.files}
        self._stop = False
        self._t = None

    def _read(self):
        for k, f in self.files.items():
            try:
                if k == "power":
                    self.samples[k].append(int(open(f).read()) / 1e6)
                    continue
                for line in open(f):
                    if "*" in line:
                        self.samples[k].append(int("".join(ch for ch in line.split(":")[1] if ch.isdigit())))
            except Exception:
                pass

    def __enter__(self):
        import threading

        def loop():
            while not self._stop:
                self._read()
                time.sleep(0.02)
        if self.files:
            self._t = threading.Thread(target=loop, daemon=True)
            self._t.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._t:
            self._t.join()

    def summary(self):
        out = {"source": self.note}
        for k, v in self.samples.items():
            if k == "power":
                out["power_w"] = {"min": round(min(v)), "mean": round(sum(v) / len(v)), "max": round(max(v)), "samples": len(v), "cap": self.power_cap_w} if v else None
                continue
            out[k + "_mhz"] = {"min": min(v), "mean": round(sum(v) / len(v)), "max": max(v), "samples": len(v)} if v else None
        return out


def stub_main(a, rank, world):
    """Launcher self-test: the rank / barrier / max-over-ranks / JSON contract with a trivial CPU 'step' over gloo."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
        assert dist.get_world_size() == a.gpus, f"process group has {dist.get_world_size()} ranks, --gpus asked for {a.gpus}"
    seen = torch.ones(1)
    if world > 1:
        dist.all_reduce(seen)

    def step():
        return float((torch.ones(64, 64) @ torch.ones(64, 64)).sum())
    for _ in range(a.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t0])
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "sampler it/s (UNet steps/sec)", "value": round(world * a.steps / float(el), 3), "unit": "it/s", "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(float(el) / a.steps * 1e3, 4), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "stub", "ranks_in_collective": int(seen.item()),
                          "launcher": "self" if os.environ.get("FMX_BENCH_SELF_LAUNCHED") else "external",
                          "config": {"workload": "STUB (launcher self-test: no kernels ran, not a measurement)"}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def rccl_selfcheck():
    """RCCL on ONE GPU (a process group of one rank is a real RCCL communicator: init, kernels, streams): the collectives a sharded job issues --
    the all-reduce of the rank check, `broadcast_conditioning` of an SDXL batch-8 conditioning pair, the `gather_batch` of fp32 latents and of uint8
    1024^2 images -- run on the device and are compared with their inputs.  Printed as ONE JSON line; the 1-GPU bench run embeds it as `rccl_world1`
    so that the driver's own record shows the RCCL code path executing (a scaling curve needs the 8-GPU node; this shows the path is not dead code)."""
    import torch
    import torch.distributed as dist
    import forge_amd  # noqa: F401
    from forge_amd import distributed as fdist, synth
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(_free_port()))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    t0 = time.perf_counter()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    one = torch.ones(1, device=dev)
    dist.all_reduce(one)
    torch.cuda.synchronize()
    out["init_plus_first_allreduce_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    out["ranks_in_collective"] = int(one.item())
    cfg = synth.SDXL_UNET_CONFIG
    c, uc = synth.synth_conditioning(8, cfg["context_dim"], cfg["adm_in_channels"], seed=1234)
    c = {k: v.half() for k, v in c.items()}
    uc = {k: v.half() for k, v in uc.items()}
    t0 = time.perf_counter()
    c2, uc2 = fdist.broadcast_conditioning(c, uc, dev)
    torch.cuda.synchronize()
    out["broadcast_cond_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    ok = all(torch.equal(c2[k].cpu(), c[k]) and torch.equal(uc2[k].cpu(), uc[k]) and c2[k].is_cuda for k in c)
    lat = torch.randn(8, 4, 128, 128, device=dev)
    img = torch.randint(0, 255, (8, 1024, 1024, 3), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lat2 = fdist.gather_batch(lat, 8)
    img2 = fdist.gather_batch(img, 8)
    torch.cuda.synchronize()
    out["gather_latents_and_images_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    ok = ok and torch.equal(lat2, lat) and torch.equal(img2, img) and lat2.data_ptr() != lat.data_ptr()
    out["results_equal_inputs"] = bool(ok)
    try:
        out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:  # noqa: BLE001
        out["rccl_version"] = None
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_SELFCHECK " + json.dumps(out), flush=True)


def rccl_world1_leg(timeout=120):
    """Run rccl_selfcheck() in a CHILD process with a timeout (an RCCL init that hangs must not take the bench line down) -> dict for the line."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    try:
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "--rccl-selfcheck"], capture_output=True, text=True, timeout=timeout, env=env)
        for ln in res.stdout.splitlines():
            if ln.startswith("RCCL_SELFCHECK "):
                return dict(json.loads(ln[len("RCCL_SELFCHECK "):]), ok=True)
        return {"ok": False, "returncode": res.returncode, "stderr_tail": res.stderr[-400:]}
    except subprocess.TimeoutExpired:
        return {"ok": False, "error": f"timed out after {timeout} s"}
    except Exception as e:  # noqa: BLE001
        return {"ok": False, "error": repr(e)}


def torch_rocm_baseline(batch=16, latent=128, warm=3, timed=5):
    """The SAME-GPU other side of "matching or beating" (VERDICT r5 item 5): what the reference itself executes on this part -- stock PyTorch-ROCm
    kernels (MIOpen / hipBLASLt / ATen), fp16 storage and compute as `memory_management` picks for a GPU -- on the default workload's network step:
    the SDXL UNet at batch 16 (8 images x {cond, uncond}) on the 128 x 128 latent.  The network is the oracle's walk (oracle/unet.py: the pinned
    restatement of backend/nn/unet.py:696-763 whose leaves are exactly the ATen calls ForgeOperations forwards to -- F.conv2d / F.linear /
    F.group_norm / F.layer_norm, operations.py:153-176) moved to cuda in fp16, with attention through F.scaled_dot_product_attention as
    `attention_pytorch` does (backend/attention.py:324-339).  A BASELINE leg like `cpu_baseline` (the only other place bench.py touches oracle/):
    timed, reported beside the native figure, never the thing measured as `value`.  It times the network forward only (no sampler arithmetic, no
    CFG combine: those are < 0.1 % of a step), `warm` untimed forwards (MIOpen's kernel search) then `timed` forwards between synchronisations.
    Printed as ONE JSON line by a child process."""
    import math
    import torch
    import torch.nn.functional as F
    import forge_amd  # noqa: F401
    from forge_amd import synth
    from forge_amd.backend.nn.layout import unet_param_shapes
    from oracle import unet as ou
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    cfg = synth.SDXL_UNET_CONFIG
    sd = synth.synth_state_dict_device(unet_param_shapes(cfg), 0, dev)          # fp16 on the device, the weights the native leg used
    sd = {k: v.half() for k, v in sd.items()}

    def sdpa(q, k, v, heads):                                                    # attention.py:324-339 attention_pytorch
        b, nq, c = q.shape
        d = c // heads
        q, k, v = (t.reshape(b, -1, heads, d).transpose(1, 2) for t in (q, k, v))
        return F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False).transpose(1, 2).reshape(b, nq, c)

    def temb(t, dim, max_period=10000):                                          # unet.py:55-67, on the timesteps' device
        half = dim // 2
        freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
        args = t[:, None].float() * freqs[None]
        return torch.cat([torch.cos(args), torch.sin(args)], dim=-1).half()
    ou.attention, ou.timestep_embedding = sdpa, temb
    g = torch.Generator().manual_seed(5)
    x = torch.randn(batch, cfg["in_channels"], latent, latent, generator=g).to(dev).half()
    ctx = torch.randn(batch, 77, cfg["context_dim"], generator=g).to(dev).half()
    y = torch.randn(batch, cfg["adm_in_channels"], generator=g).to(dev).half()
    t = torch.full((batch,), 500.0, device=dev)
    out = {"torch": torch.__version__, "hip": torch.version.hip, "batch": batch, "latent": latent, "dtype": "f16"}
    with torch.inference_mode():
        t0 = time.perf_counter()
        for _ in range(warm):
            eps = ou.unet_forward(sd, cfg, x, t, ctx, y)
        torch.cuda.synchronize()
        out["warmup_s"] = round(time.perf_counter() - t0, 1)
        t0 = time.perf_counter()
        for _ in range(timed):
            eps = ou.unet_forward(sd, cfg, x, t, ctx, y)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / timed
    out.update({"ms_per_forward": round(dt * 1e3, 2), "it_per_s_equivalent": round(1.0 / dt, 4), "forwards_timed": timed, "finite": bool(torch.isfinite(eps.float()).all()),
                "tflops": round(batch * FLOPS_PER_SAMPLE_FWD["sdxl"] / dt / 1e12, 1), "frac_of_mfma_peak": round(batch * FLOPS_PER_SAMPLE_FWD["sdxl"] / dt / MFMA_PEAK, 4)})
    print("TORCH_ROCM_BASELINE " + json.dumps(out), flush=True)


def torch_rocm_baseline_leg(timeout=170):   # ~120 s on a fresh box (MIOpen builds its kernels on first use); bounded so that the default run stays under five minutes
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    try:
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "--torch-rocm-baseline"], capture_output=True, text=True, timeout=timeout, env=env)
        for ln in res.stdout.splitlines():
            if ln.startswith("TORCH_ROCM_BASELINE "):
                return dict(json.loads(ln[len("TORCH_ROCM_BASELINE "):]), ok=True, wall_s_of_this_leg=round(time.time() - t0, 1))
        return {"ok": False, "returncode": res.returncode, "stderr_tail": res.stderr[-400:]}
    except subprocess.TimeoutExpired:
        return {"ok": False, "error": f"timed out after {timeout} s"}
    except Exception as e:  # noqa: BLE001
        return {"ok": False, "error": repr(e)}


def other_configs_leg(timeout=90):
    """The small-batch regime in the driver's own record: after the timed region of the default workload, BASELINE config 2
    (SD1.5 512^2, batch 4, Euler a) and SDXL 1024^2 at batch 1 (how the reference is used: modules/processing.py:139 `batch_size: int = 1`) are each
    measured by a child process running THIS file with that workload (10 timed steps after 2 warm-up steps, same contract, graph replay on) -- the
    default line's own numbers are untouched by them; and Flux.1-dev at 1024^2, batch 2 in bfloat16 (BASELINE config 5's network; 5 timed steps).  -> {name: {it_per_s, ms_per_step, step_frac_of_mfma_peak, gemm_frac_of_mfma_peak, ...}}"""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    legs = {"sd15-512-b4-eulera (BASELINE configs[1])": ["--config", "sd15-b4-eulera"],
            "sdxl-1024-b1-euler (the reference's default batch size)": ["--config", "sdxl-b8-euler20", "--batch", "1"],
            # BASELINE config 5's network in the reference's own compute type: Flux.1-dev (11.9 B parameters, random init drawn on the device), 1024^2, batch 2
            "flux-dev-1024-b2-bf16 (BASELINE configs[4]'s network)": ["--config", "flux-b2-bf16"]}
    out = {}
    for name, flags in legs.items():
        t0 = time.time()
        try:
            steps = "5" if "flux" in name else "10"
            res = subprocess.run([sys.executable, os.path.abspath(__file__), *flags, "--steps", steps, "--warmup", "2", "--no-cpu-baseline", "--no-vae",
                                  "--no-rccl-selfcheck", "--no-other-configs", "--no-torch-rocm-baseline"], capture_output=True, text=True, timeout=2 * timeout if "flux" in name else timeout, env=env)
            line = next((ln for ln in reversed(res.stdout.splitlines()) if ln.startswith("{")), None)
            if line is None:
                out[name] = {"ok": False, "returncode": res.returncode, "stderr_tail": res.stderr[-300:]}
                continue
            d = json.loads(line)
            out[name] = {"it_per_s": d["value"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "workload": d["config"]["workload"],
                         "step_tflops_per_gpu": d.get("step_tflops_per_gpu"), "step_frac_of_mfma_peak": d.get("step_frac_of_mfma_peak"),
                         "gemm_frac_of_mfma_peak": (d.get("roofline") or {}).get("frac"), "gemm_launches_per_forward": (d.get("roofline") or {}).get("launches_per_forward"),
                         "attention_frac_of_mfma_peak": (d.get("roofline_attention") or {}).get("frac"),
                         "groupnorm_ms_per_forward": (d.get("roofline_groupnorm") or {}).get("kernel_time_per_forward_ms"),
                         "sclk_mhz_mean": ((d.get("clocks_during_timed_steps") or {}).get("sclk_mhz") or {}).get("mean"),
                         "wall_s_of_this_leg": round(time.time() - t0, 1)}
        except subprocess.TimeoutExpired:
            out[name] = {"ok": False, "error": f"timed out after {timeout} s"}
        except Exception as e:  # noqa: BLE001 -- an extra leg must never take the bench line down
            out[name] = {"ok": False, "error": repr(e)}
    return out


def main():
    a = parse()
    if a.rccl_selfcheck:
        return rccl_selfcheck()
    if a.torch_rocm_baseline:
        return torch_rocm_baseline()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        self_launch(a)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to report a number for the wrong N")
    if a.stub_engine:
        return stub_main(a, rank, world)

    import torch
    import forge_amd  # noqa: F401
    from forge_amd import distributed as fdist
    from forge_amd import hipops, synth
    from forge_amd.backend.diffusion_engine.base import build_engine, build_flux_engine
    from forge_amd.backend.nn.layout import flux_param_shapes, unet_param_shapes, vae_decoder_param_shapes
    from forge_amd.backend.nn.unet import IntegratedUNet2DConditionModel
    from forge_amd.modules import processing, rng, sd_samplers, shared
    from forge_amd.modules.prompt_parser import DictWithShape

    if torch.cuda.device_count() <= local:
        raise SystemExit(f"bench.py: rank {rank} (local {local}) has no GPU: {torch.cuda.device_count()} visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ranks_seen = 1
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm
        assert dist.get_world_size() == a.gpus, f"process group has {dist.get_world_size()} ranks, --gpus asked for {a.gpus}"
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)                              # a real RCCL collective: every rank must have contributed
        ranks_seen = int(one.item())
        if ranks_seen != a.gpus:
            raise SystemExit(f"bench.py: all-reduce saw {ranks_seen} ranks, expected {a.gpus}")
    wl = dict(CONFIGS[a.config])
    model = a.model or wl["model"]
    res = a.res or wl["res"]
    bpg = a.batch or wl["batch"]
    sampler_name = a.sampler or wl["sampler"]
    cfg_scale = wl["cfg"] if a.cfg is None else a.cfg
    nominal = wl["nominal_steps"]
    is_flux = model == "flux"
    dt16 = torch.bfloat16 if (wl["dtype"] == "bf16" and is_flux) else torch.float16
    latent = res // 8

    # ---- model: random-init weights drawn on the device ---------------------------------------------------------
    t0 = time.time()
    if is_flux:
        ucfg = synth.FLUX_DEV_CONFIG
        vcfg = None
        a.no_vae = True   # the 16-channel VAE decode is the same decoder kernels; this workload times the transformer
        sd = synth.synth_state_dict_device(flux_param_shapes(ucfg), 2, dev, dtype=dt16)
        eng = build_flux_engine(ucfg, sd, device=dev, dtype=dt16, seq_len=(latent // 2) ** 2)
        del sd
    else:
        ucfg = synth.SDXL_UNET_CONFIG if model == "sdxl" else synth.SD15_UNET_CONFIG
        vcfg = synth.SDXL_VAE_CONFIG if model == "sdxl" else synth.SD15_VAE_CONFIG
        usd = synth.synth_state_dict_device(unet_param_shapes(ucfg), 0, dev)
        vsd = None if a.no_vae else synth.synth_state_dict_device(vae_decoder_param_shapes(vcfg), 1, dev)
        IntegratedUNet2DConditionModel.RETAIN_TRUNK_WEIGHTS = False  # no Control-LoRA in this run: do not keep a second copy of the encoder weights
        eng = build_engine(ucfg, usd, None if a.no_vae else vcfg, vsd, device=dev)
        del usd, vsd
    km = eng.forge_objects.unet.model
    if not is_flux:
        km.use_graph = not a.no_graph
    torch.cuda.synchronize()
    t_build = time.time() - t0

    # ---- conditioning: rank 0 owns the global batch, RCCL broadcast, each rank keeps its shard -------------------
    total = bpg * world
    if rank == 0:
        if is_flux:
            g = torch.Generator().manual_seed(1234)
            c = {"crossattn": torch.randn(total, 256, ucfg["context_in_dim"], generator=g).to(dev, dt16),
                 "vector": torch.randn(total, ucfg["vec_in_dim"], generator=g).to(dev, dt16),
                 "guidance": torch.full((total,), 3.5, device=dev)}
            uc = {k: v.clone() for k, v in c.items()}
        else:
            c, uc = synth.synth_conditioning(total, ucfg["context_dim"], ucfg.get("adm_in_channels"), seed=1234)
            to_dev = lambda t: t.to(dev).half()
            c = {k: to_dev(v) for k, v in c.items()} if isinstance(c, dict) else to_dev(c)
            uc = {k: to_dev(v) for k, v in uc.items()} if isinstance(uc, dict) else to_dev(uc)
    else:
        c = uc = None
    # (until round 4 this interval also held the host-side synthesis of the conditioning and its upload: 22-28 ms "broadcast" at world 1)
    torch.cuda.synchronize()
    t0 = time.time()
    c, uc = fdist.broadcast_conditioning(c, uc, dev)
    torch.cuda.synchronize()
    t_bcast = time.time() - t0
    lo, hi = fdist.shard_range(total, rank, world)
    c, uc = fdist.slice_conditioning(c, lo, hi), fdist.slice_conditioning(uc, lo, hi)
    if isinstance(c, dict):
        c, uc = DictWithShape(c), DictWithShape(uc)
    torch.cuda.synchronize()

    shared.opts.randn_source = "CPU"
    seeds = [1000 + i for i in range(lo, hi)]
    lat_ch = 16 if is_flux else 4

    def make_p(steps):
        p = processing.StableDiffusionProcessingTxt2Img(sd_model=eng, c=c, uc=uc, seed=seeds[0], sampler_name=sampler_name, scheduler=wl["scheduler"],
                                                        batch_size=bpg, steps=steps, cfg_scale=cfg_scale, width=res, height=res)
        p.seeds = seeds
        p.all_seeds = seeds
        p.rng = rng.ImageRNG((lat_ch, latent, latent), seeds, device=dev)
        return p

    def run_sampler(steps):
        p = make_p(steps)
        sampler = sd_samplers.create_sampler(sampler_name, eng)
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
        clocks = ClockSampler(local)
        with clocks:
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
            if a.vae_breakdown and rank == 0:
                with hipops.KernelProfiler() as vprof:
                    eng.decode_first_stage(lat)
                    torch.cuda.synchronize()
                    vprof.summary()
                write_breakdown(a.vae_breakdown, vprof)
        t0 = time.perf_counter()
        gathered = fdist.gather_latents(lat, total)
        torch.cuda.synchronize()
        t_gather = time.perf_counter() - t0
        del gathered

        # ---- roofline of the dominant kernels: HIP events around every launch in one eager network forward ----------
        roof = attn_roof = xattn_roof = gn_roof = None
        if rank == 0 and not a.no_roofline:
            if is_flux:
                net = km.diffusion_model
                x = torch.randn(bpg, 16, latent, latent, device=dev)
                ts = torch.full((bpg,), 0.7, device=dev)
                call = lambda: net.forward(x, ts, c["crossattn"], c["vector"], c["guidance"])
            else:
                km.use_graph = False
                x = torch.randn(bpg, 4, latent, latent, device=dev)
                sig = torch.full((bpg,), 5.0, device=dev)
                uctx = (uc["crossattn"], uc["vector"]) if isinstance(uc, dict) else (uc, None)
                cctx = (c["crossattn"], c["vector"]) if isinstance(c, dict) else (c, None)
                call = lambda: km.denoise_cfg(x, sig, uctx, cctx, cfg_scale)
            call()
            torch.cuda.synchronize()
            with hipops.KernelProfiler() as prof:
                call()
                torch.cuda.synchronize()
                summ = prof.summary()
            if a.breakdown:
                write_breakdown(a.breakdown, prof)
            if not is_flux:
                km.use_graph = not a.no_graph
            g = summ.get("gemm_conv")
            if g:
                ach = g["flops"] / g["seconds"]
                # the committed counters were taken on the default workload's GEMM launches (tools/gpu_round2.sh pmc_hbm): quoted for that workload only
                if a.config == "sdxl-b8-euler20" and not (a.model or a.res or a.batch):
                    traffic, traffic_note = pmc_traffic_per_launch()
                else:
                    traffic, traffic_note = None, "the committed PMC passes profile the default workload (sdxl-b8-euler20), not this one"
                roof = {"kernel": "gemm256p_kernel<256x320 | 320x256 | 256x256 | 512x128> + gemm_kernel (fmx_gemm_conv: MFMA implicit-GEMM conv3x3/1x1 + linear, fused epilogues)",
                        "bound": "mfma", "achieved": round(ach / 1e12, 1), "peak": MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                        "frac": round(ach / MFMA_PEAK, 4), "traffic": traffic, "traffic_note": traffic_note, "launches_per_forward": g["launches"],
                        "flop_per_launch_avg": round(g["flops"] / g["launches"] / 1e9, 2), "flop_unit": "GFLOP",
                        "us_per_launch_avg": round(g["seconds"] / g["launches"] * 1e6, 1),
                        "kernel_time_per_forward_ms": round(g["seconds"] * 1e3, 2)}
            at = summ.get("attention")
            if at:
                ach = at["flops"] / at["seconds"]
                attn_roof = {"kernel": "attn_q64v2_kernel / attn_ws_kernel (d_head 128) / attn_kernel<d> (fmx_attention: fused QK^T-softmax-PV), launches with more than 128 keys", "bound": "mfma",
                             "achieved": round(ach / 1e12, 1), "peak": MFMA_PEAK / 1e12, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK, 4),
                             "launches_per_forward": at["launches"], "kernel_time_per_forward_ms": round(at["seconds"] * 1e3, 2)}
            xa = summ.get("attention_short_keys")
            if xa and xa.get("bytes"):
                ach = xa["bytes"] / xa["seconds"]
                xattn_roof = {"kernel": "attn_q64v3_kernel / attn_kernel<d> on the 77-token text context (two key tiles)", "bound": "hbm",
                              "achieved": round(ach / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(ach / HBM_PEAK, 4),
                              "algorithmic_bytes": "1 read of Q, K, V + 1 write of O", "launches_per_forward": xa["launches"],
                              "kernel_time_per_forward_ms": round(xa["seconds"] * 1e3, 2)}
            gn = summ.get("groupnorm")
            if gn and gn.get("bytes"):
                ach = gn["bytes"] / gn["seconds"]
                gn_roof = {"kernel": "gn_apply (+ gn_stats where the producer GEMM did not emit the statistics) (fmx_groupnorm: GroupNorm + SiLU, NHWC fp16)",
                           "bound": "hbm", "achieved": round(ach / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(ach / HBM_PEAK, 4),
                           "algorithmic_bytes": "1 read + 1 write of the tensor per GroupNorm", "launches_per_forward": gn["launches"],
                           "kernel_time_per_forward_ms": round(gn["seconds"] * 1e3, 2)}

    ms_per_step = elapsed / a.steps * 1e3
    value = world * a.steps / elapsed
    fwd_per_image = 1 if (is_flux or cfg_scale == 1.0) else 2
    fl = FLOPS_PER_SAMPLE_FWD[model] * fwd_per_image * bpg  # per GPU per step (CFG: 2 sample-forwards per image)
    if res != wl["res"]:
        fl = None
    names = {"sdxl": "SDXL UNet", "sd15": "SD1.5 UNet", "flux": "Flux-dev DiT (4096 + 256 tokens)"}
    out = {
        "metric": "sampler it/s (UNet steps/sec)", "value": round(value, 4), "unit": "it/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if dt16 == torch.bfloat16 else "f16",
        "data": "synthetic (random-init weights, N(0,1) conditioning, per-image seeded CPU noise)",
        "config": {"workload": f"{names[model]} {res}x{res} batch={bpg}/GPU {'bf16' if dt16 == torch.bfloat16 else 'fp16'}, {sampler_name} sampler, CFG {cfg_scale} "
                               f"(network batch {fwd_per_image * bpg}), HIP-graph replay={'on' if not a.no_graph else 'off'}",
                   "name": a.config, "baseline_config": wl["baseline"], "global_batch": total, "latent": [latent, latent],
                   "parallelism": f"batch-shard x{world} (RCCL broadcast cond + gather latents)"},
        "ranks_in_collective": ranks_seen,
        "launcher": "self (bench.py re-executed under torch.distributed.run)" if os.environ.get("FMX_BENCH_SELF_LAUNCHED") else
                    ("external (torch.distributed.run)" if world > 1 else "single process"),
        "step_tflops_per_gpu": None if fl is None else round(fl / 1e12, 2),
        "achieved_tflops_per_gpu": None if fl is None else round(fl / (ms_per_step * 1e-3) / 1e12, 1),
        "step_frac_of_mfma_peak": None if fl is None else round(fl / (ms_per_step * 1e-3) / MFMA_PEAK, 4),
        "vae_decode_ms_per_batch": None if vae_ms is None else round(vae_ms, 1),
        f"ms_per_image_{nominal}_steps_plus_vae": None if vae_ms is None else round((nominal * ms_per_step + vae_ms) / bpg, 1),
        "comm_ms": {"broadcast_cond": round(t_bcast * 1e3, 2), "gather_latents": round(t_gather * 1e3, 2)},
        "build_s": round(t_build, 1),
        "clocks_during_timed_steps": clocks.summary(),
        "kernel_source_hash": kernel_source_hash(),          # of the loaded binary (fmx_build_info)
        "source_tree_hash": source_tree_hash(),              # of csrc/ on disk: differs from the line above when the .so is stale
        "knobs": {"active": _knobs(False), "set_but_ignored": _knobs(True)},   # development A/B knobs (need FMX_ALLOW_KNOBS=1); {} = the default kernels
    }
    if roof:
        out["roofline"] = roof
    if attn_roof:
        out["roofline_attention"] = attn_roof
    if xattn_roof:
        out["roofline_attention_short_keys"] = xattn_roof
    if gn_roof:
        out["roofline_groupnorm"] = gn_roof
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            dt, nthreads, note = cpu_baseline(model, ucfg, latent, sampler_name, cfg_scale)
            out["cpu_baseline"] = {"value": round(1.0 / (fwd_per_image * bpg * dt), 6), "unit": "it/s", "cores": nthreads, "kind": "port",
                                   "sample": note + f"; a step of this workload is {fwd_per_image * bpg} such forwards"}
            ref_box = reference_on_authoring_box(model)
            if ref_box:
                out["cpu_baseline"]["reference_on_authoring_box"] = ref_box
        except Exception as e:  # the baseline must never take the bench line down
            out["cpu_baseline"] = {"value": None, "error": repr(e)}
    default_workload = a.config == "sdxl-b8-euler20" and not (a.model or a.res or a.batch or a.sampler)
    if rank == 0 and world == 1 and not a.no_torch_rocm_baseline and default_workload:
        # the reference's own op stack on THIS GPU, same network step (child process, after the timed region).  `vs_baseline` = native / that: context --
        # BASELINE.md publishes no number for this metric, and the target is the kernel roofline, not this ratio
        tb = torch_rocm_baseline_leg()
        out["torch_rocm_baseline"] = dict(tb, kind="the reference's op stack (stock PyTorch-ROCm ATen / MIOpen / hipBLASLt kernels, fp16, F.scaled_dot_product_attention) "
                                                   "on the same GPU: SDXL UNet forward at batch 16, 128 x 128 latent")
        if tb.get("ok") and tb.get("it_per_s_equivalent"):
            out["vs_baseline"] = round(value / tb["it_per_s_equivalent"], 3)
            out["vs_baseline_note"] = "native it/s / same-GPU PyTorch-ROCm it/s-equivalent (torch_rocm_baseline); context only -- BASELINE.md holds no published number for this metric"
    if rank == 0 and world == 1 and not a.no_rccl_selfcheck:
        out["rccl_world1"] = rccl_world1_leg()     # after the timed region, in a child process
    if rank == 0 and world == 1 and not a.no_other_configs and a.config == "sdxl-b8-euler20" and not (a.model or a.res or a.batch or a.sampler):
        out["other_configs"] = other_configs_leg()  # after the timed region, child processes
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
