"""RCCL on the one GPU a lease has: a process group of ONE rank over backend "nccl" (= RCCL on ROCm) is a real communicator, so the collectives of
the sharded job -- `forge_amd.distributed.broadcast_conditioning` / `gather_batch` and the product entry
`forge_amd.modules.processing.process_images_sharded` (the split of /root/reference/modules/processing.py:924-1012) -- run ON THE DEVICE over RCCL and
must reproduce the single-process job bit for bit.  (World sizes 2, 3 and 8 are covered over gloo on the CPU: tests/test_sharded_processing.py.)
Each case runs in a child process: a default process group must not leak into the rest of the GPU suite."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CHILD = r'''
import json, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
import forge_amd
from forge_amd import distributed as fdist, synth
from forge_amd.backend.diffusion_engine.base import build_engine
from forge_amd.modules import processing, shared
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
cfg, vcfg = synth.TINY_SD15_UNET_CONFIG, synth.TINY_VAE_CONFIG
eng = build_engine(cfg, synth.synth_unet_state_dict(cfg, seed=0), vcfg, synth.synth_vae_decoder_state_dict(vcfg, seed=1), device=dev)
shared.opts.randn_source = "CPU"
def job():
    c, uc = synth.synth_conditioning(6, cfg["context_dim"], None, seed=1234)
    return processing.StableDiffusionProcessingTxt2Img(sd_model=eng, c=c.to(dev).half(), uc=uc.to(dev).half(), seed=77, sampler_name="Euler a", batch_size=3, n_iter=2,
                                                       steps=4, cfg_scale=7.0, width=128, height=128)
want = processing.process_images(job())                       # no process group yet: the plain job
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(%(port)d)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
one = torch.ones(1, device=dev); dist.all_reduce(one)
got = processing.process_images_sharded(job())                # every collective of the sharded entry over RCCL
c, uc = synth.synth_conditioning(4, 2048, 2816, seed=5)
c2, uc2 = fdist.broadcast_conditioning({k: v.half() for k, v in c.items()}, None, dev)
lat = torch.randn(5, 4, 16, 16, device=dev)
lat2 = fdist.gather_batch(lat, 5)
out = {"backend": dist.get_backend(), "ranks": int(one.item()), "group_active": bool(fdist.group_active()),
       "latents_equal": bool(torch.equal(got.latents.cpu(), want.latents.cpu())), "seeds_equal": got.seeds == want.seeds,
       "images_equal": bool(np.array_equal(np.stack(got.images), np.stack(want.images))), "n_images": len(got.images),
       "latents_on": str(got.latents.device), "bcast_ok": bool(uc2 is None and all(torch.equal(c2[k].cpu(), c[k].half()) and c2[k].is_cuda for k in c)),
       "gather_ok": bool(torch.equal(lat2, lat))}
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
'''


def test_sharded_job_and_collectives_over_rccl_at_world_size_one():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "port": port}], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("RESULT ")]
    assert line, res.stdout[-2000:]
    out = json.loads(line[-1][7:])
    print("[rccl world 1]", out)
    assert out["backend"] == "nccl" and out["ranks"] == 1 and out["group_active"]
    assert out["latents_equal"] and out["seeds_equal"] and out["images_equal"] and out["n_images"] == 6
    assert out["bcast_ok"] and out["gather_ok"]


def test_bench_rccl_selfcheck_leg():
    """bench.py's `rccl_world1` leg (what the driver's 1-GPU bench line carries): child process, RCCL init + all-reduce + the job's collectives at SDXL sizes."""
    sys.path.insert(0, ROOT)
    import bench
    out = bench.rccl_world1_leg(timeout=300)
    print("[rccl selfcheck]", out)
    assert out.get("ok"), out
    assert out["backend"] == "nccl" and out["ranks_in_collective"] == 1 and out["results_equal_inputs"]
