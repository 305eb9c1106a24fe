"""fp16 vs bf16 VAE decode time at SDXL 1024^2 (one log printed "fp16 14.69 ms, bf16 61.71 ms" where every earlier one had the two
level).  Builds each executor once, decodes the same latent `reps` times (wall clock around a synchronised call, as tests/test_gpu_vae_bf16.py does), then
writes the per-shape kernel-time table of one decode per element type.  usage: python tools/vae_dtype_check.py [batch] [reps] [out.jsonl]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import forge_amd  # noqa
from forge_amd import hipops, synth
from forge_amd.backend.nn.vae import IntegratedAutoencoderKL

DEV = "cuda"


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    out = sys.argv[3] if len(sys.argv) > 3 else None
    sd = synth.synth_vae_decoder_state_dict(synth.SDXL_VAE_CONFIG, seed=1)
    z = (torch.randn(batch, 4, 128, 128, generator=torch.Generator().manual_seed(3)) * 0.9).to(DEV)
    rows = []
    for order in (("f16", "bf16"), ("bf16", "f16")):          # both construction orders: does the second executor inherit something from the first?
        for name in order:
            dt = torch.float16 if name == "f16" else torch.bfloat16
            vae = IntegratedAutoencoderKL(synth.SDXL_VAE_CONFIG, sd, device=DEV, dtype=dt)
            zz = vae.process_out(z)
            times = []
            for _ in range(reps + 1):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                vae.decode(zz)
                torch.cuda.synchronize()
                times.append(round((time.perf_counter() - t0) * 1e3, 2))
            with hipops.KernelProfiler() as prof:
                vae.decode(zz)
                torch.cuda.synchronize()
                summ = prof.summary()
            kern_ms = {k: round(v["seconds"] * 1e3, 3) for k, v in summ.items()}
            top = sorted(prof.by_tag.items(), key=lambda kv: -kv[1]["seconds"])[:4]
            rec = {"dtype": name, "order": "+".join(order), "batch": batch, "wall_ms_first_then_reps": times, "fallbacks": vae.fallbacks,
                   "kernel_ms_by_family": kern_ms, "kernel_ms_total": round(sum(kern_ms.values()), 3),
                   "slowest": [{"kind": k, "shape": t, "ms": round(d["seconds"] * 1e3, 3), "launches": d["launches"]} for (k, t), d in top]}
            print(json.dumps(rec), flush=True)
            rows.append(rec)
            del vae
    if out:
        with open(out, "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
