"""CPU, world sizes 2 and 3 over gloo: `forge_amd.modules.processing.process_images_sharded` -- the product entry that splits the reference's
batch loop (modules/processing.py:924-1012) across ranks -- against the single-process `process_images` of the same job, bit for bit.

The job goes through the REAL processing code on every rank (seed plan, ImageRNG with the reference's CPU noise source, conditioning slicing,
decode, clamp / *255 / uint8 truncation, the gathers); only the two things that need the GPU are stand-ins: the sampler (`sample` = a
deterministic per-image function of the noise and the conditioning) and the VAE (`decode_first_stage` = a fixed upsample + channel mix).  On a
GPU node the same function runs over RCCL (backend "nccl")."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_job(batch, n_iter, with_dict, cfg1=False, give_conds=True):
    import forge_amd  # noqa: F401
    from forge_amd.modules import processing, shared
    from forge_amd.modules.prompt_parser import DictWithShape

    class _Vae:
        latent_channels = 4

    class _Objs:
        vae = _Vae()

        def shallow_copy(self):
            return self

    class _Engine:
        device = torch.device("cpu")
        forge_objects = _Objs()
        forge_objects_after_applying_lora = _Objs()
        is_sdxl = False

        def decode_first_stage(self, x):
            up = torch.nn.functional.interpolate(x, scale_factor=8, mode="nearest")
            mix = torch.tensor([[0.3, -0.2, 0.1, 0.4], [0.1, 0.5, -0.3, 0.2], [-0.4, 0.2, 0.6, 0.1]])
            return torch.einsum("oc,bchw->bohw", mix, up) * 0.5

    class _Job(processing.StableDiffusionProcessingTxt2Img):
        def sample(self, conditioning, unconditional_conditioning, seeds, subseeds=None, subseed_strength=0.0, prompts=None):
            x = self.rng.next()                                   # the reference's per-image CPU noise: seed + global index
            c = conditioning["crossattn"] if isinstance(conditioning, dict) else conditioning
            s = c.float().mean(dim=(1, 2)).view(-1, 1, 1, 1)
            if isinstance(conditioning, dict):
                s = s + conditioning["vector"].float().sum(dim=1).view(-1, 1, 1, 1)
            if unconditional_conditioning is not None:
                u = unconditional_conditioning["crossattn"] if isinstance(unconditional_conditioning, dict) else unconditional_conditioning
                s = s - 0.5 * u.float().std(dim=(1, 2)).view(-1, 1, 1, 1)
            return torch.tanh(x * 0.7 + s) + 0.01 * torch.tensor([float(v % 97) for v in seeds]).view(-1, 1, 1, 1)

    total = batch * n_iter
    g = torch.Generator().manual_seed(11)
    c = torch.randn(total, 154, 8, generator=g).half()
    uc = torch.randn(total, 77, 8, generator=g).half()
    if with_dict:
        c = DictWithShape({"crossattn": c, "vector": torch.randn(total, 6, generator=g)})
        uc = DictWithShape({"crossattn": uc, "vector": torch.zeros(total, 6)})
    if cfg1:
        uc = None
    shared.opts.randn_source = "CPU"
    return _Job(sd_model=_Engine(), c=c if give_conds else None, uc=uc if give_conds else None, seed=4242, batch_size=batch, n_iter=n_iter,
                steps=4, cfg_scale=1.0 if cfg1 else 7.0, width=32, height=24)


def _worker(rank, world, port, batch, n_iter, with_dict, cfg1, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from forge_amd.modules import processing
        p = _make_job(batch, n_iter, with_dict, cfg1, give_conds=(rank == 0))      # only the owner has the conditioning
        res = processing.process_images_sharded(p)
        if rank == 0:
            q.put((rank, res.latents.cpu().numpy().copy(), np.stack(res.images), list(res.seeds)))   # by value: a tensor would travel as an fd of a process that may be gone
        else:
            q.put((rank, int(res.latents.shape[0]), len(res.images), list(res.seeds)))
    finally:
        dist.destroy_process_group()


# (1, ...): a process group of ONE rank still runs every collective (what tests/test_gpu_rccl.py does over RCCL on a one-GPU box);
# (8, 64, ...): BASELINE config 4's shape -- 64 images over 8 ranks, 8 per rank
@pytest.mark.parametrize("world,batch,n_iter,with_dict,cfg1", [(2, 5, 1, False, False), (3, 5, 2, True, False), (3, 2, 1, False, True),
                                                               (2, 4, 2, True, True), (1, 3, 2, True, False), (8, 64, 1, True, False)])
def test_sharded_job_equals_the_single_process_job_bit_for_bit(world, batch, n_iter, with_dict, cfg1):
    from forge_amd import distributed as fdist
    from forge_amd.modules import processing
    want = processing.process_images(_make_job(batch, n_iter, with_dict, cfg1))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, batch, n_iter, with_dict, cfg1, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        r = q.get(timeout=180)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lat, imgs, seeds = res[0]
    lat = torch.from_numpy(lat)
    assert seeds == want.seeds == [4242 + i for i in range(batch * n_iter)]
    assert lat.shape == want.latents.shape and torch.equal(lat, want.latents), "gathered latents differ from the single-process job"
    assert imgs.shape == np.stack(want.images).shape and np.array_equal(imgs, np.stack(want.images))
    for r in range(1, world):                                   # the other ranks keep only their own share (a true gather, not all_gather)
        lo, hi = fdist.shard_range(batch, r, world)
        assert res[r][0] == (hi - lo) * n_iter and res[r][1] == (hi - lo) * n_iter


def test_sharded_entry_is_process_images_without_a_process_group():
    from forge_amd.modules import processing
    a = processing.process_images_sharded(_make_job(3, 1, False))
    b = processing.process_images(_make_job(3, 1, False))
    assert torch.equal(a.latents, b.latents) and a.seeds == b.seeds


def _failing_worker(rank, world, port, mode, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from forge_amd.modules import processing
        p = _make_job(4, 1, False, give_conds=(rank == 0))
        if mode == "owner" and rank == 0:
            p.c = ["a host-side prompt-editing schedule is not a tensor"]      # refused by the owner BEFORE the first collective
        if mode == "worker" and rank == 1:
            def boom(*a, **k):
                raise MemoryError("simulated out-of-memory on one rank")
            p.sample = boom                                                       # fails INSIDE this rank's share of the job
        try:
            processing.process_images_sharded(p)
            q.put((rank, "returned", ""))
        except Exception as e:  # noqa: BLE001
            q.put((rank, type(e).__name__, str(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["owner", "worker"])
def test_a_failure_on_one_rank_raises_on_every_rank_instead_of_hanging(mode):
    """the owner raising before the first collective (a conditioning the sharded entry does not broadcast) or one rank failing inside its
    share left the other ranks blocked in broadcast_object_list / all_gather_object.  Now the error travels in the header / the shape exchange."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in procs:
        r = q.get(timeout=120)      # a hang shows up here
        got[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        kind, msg = got[r]
        assert kind in ("NotImplementedError", "RuntimeError"), got
        assert ("prompt-editing" in msg) if mode == "owner" else ("simulated out-of-memory" in msg and "rank 1" in msg), got


# ---- round 5: the conditionings the reference normally has (schedule objects), and img2img jobs sharded by init image ---------------------------------
def _make_schedule_job(batch, n_iter, give_conds=True):
    """p.c = MulticondLearnedConditioning (two AND parts with weights, the first a prompt-editing schedule that switches at step 2; some images share ONE
    tensor object, as a repeated prompt does), p.uc = per-image schedule lists -- what modules/prompt_parser.py:294-365 builds from prompt strings"""
    import forge_amd  # noqa: F401
    from forge_amd.modules import processing, prompt_parser as pp, shared
    base = _make_job(batch, n_iter, False)
    total = batch * n_iter
    g = torch.Generator().manual_seed(23)
    shared_t = torch.randn(77, 8, generator=g).half()

    class _Job(type(base)):
        def sample(self, conditioning, unconditional_conditioning, seeds, subseeds=None, subseed_strength=0.0, prompts=None):
            x = self.rng.next()
            acc = torch.zeros(len(seeds), 1, 1, 1)
            for step in (0, 2, 3):                         # the schedule is read per step, as CFGDenoiser does (sd_samplers_cfg_denoiser.py:156-170)
                conds_list, tensor = pp.reconstruct_multicond_batch(conditioning, step)
                un = pp.reconstruct_cond_batch(unconditional_conditioning, step)
                for i, parts in enumerate(conds_list):
                    for idx, w in parts:
                        acc[i] += w * tensor[idx].float().mean() * (step + 1)
                acc -= 0.25 * un.float().std(dim=(1, 2)).view(-1, 1, 1, 1)
            return torch.tanh(x * 0.7 + acc) + 0.01 * torch.tensor([float(v % 97) for v in seeds]).view(-1, 1, 1, 1)

    def sched(i):
        a = shared_t if i % 2 == 0 else torch.randn(77, 8, generator=g).half()
        b = torch.randn(77, 8, generator=g).half()
        return [pp.ScheduledPromptConditioning(2, a), pp.ScheduledPromptConditioning(4, b)]
    c = pp.MulticondLearnedConditioning((total,), [[pp.ComposableScheduledPromptConditioning(sched(i), 1.0),
                                                      pp.ComposableScheduledPromptConditioning([pp.ScheduledPromptConditioning(4, torch.randn(77, 8, generator=g).half())], 0.5 + 0.1 * i)]
                                                     for i in range(total)])
    uc = [[pp.ScheduledPromptConditioning(4, shared_t)] for _ in range(total)]
    p = _Job(sd_model=base.sd_model, c=c if give_conds else None, uc=uc if give_conds else None, seed=4242, batch_size=batch, n_iter=n_iter, steps=4, cfg_scale=7.0,
             width=32, height=24)
    return p


def _make_img2img_job(batch, n_iter, give=True, shared_fill2=False):
    import forge_amd  # noqa: F401
    from forge_amd.modules import processing
    base = _make_job(batch, n_iter, False)
    total = batch * n_iter
    g = torch.Generator().manual_seed(31)

    class _Job(processing.StableDiffusionProcessingImg2Img):
        def sample(self, conditioning, unconditional_conditioning, seeds, subseeds=None, subseed_strength=0.0, prompts=None):
            x = self.rng.next()
            lo = self.iteration * self.batch_size
            init = self.init_latent[lo:lo + self.batch_size] if self.init_latent.shape[0] > self.batch_size else self.init_latent
            out = torch.tanh(0.5 * init + 0.3 * x + conditioning.float().mean(dim=(1, 2)).view(-1, 1, 1, 1))
            if self.mask is not None:
                m = self.mask[lo:lo + self.batch_size] if self.mask.shape[0] > self.batch_size else self.mask
                out = out * (1 - m) + init * m
            return out
    nmask = torch.zeros(1, 1, 3, 4)
    nmask[..., 1:, :2] = 1.0
    rows = 1 if shared_fill2 else total     # shared_fill2: ONE init latent for the whole job + 'latent noise' fill -- every image draws its own fill noise
    return _Job(sd_model=base.sd_model, c=base.c if give else None, uc=base.uc if give else None, seed=4242, batch_size=batch, n_iter=n_iter, steps=4, cfg_scale=7.0,
                width=32, height=24, init_latent=torch.randn(rows, 4, 3, 4, generator=g) if give else None, latent_mask=nmask if give else None,
                denoising_strength=0.6, inpainting_fill=2 if shared_fill2 else 3)


def _make_img2img_fill2_job(batch, n_iter, give=True):
    return _make_img2img_job(batch, n_iter, give, shared_fill2=True)


def _worker2(rank, world, port, kind, batch, n_iter, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from forge_amd.modules import processing
        make = {"schedules": _make_schedule_job, "img2img": _make_img2img_job, "img2img_fill2": _make_img2img_fill2_job}[kind]
        res = processing.process_images_sharded(make(batch, n_iter, rank == 0))          # only the owner holds conditionings / init latents / masks
        q.put((rank, res.latents.cpu().numpy().copy() if rank == 0 else int(res.latents.shape[0]), list(res.seeds)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind,world,batch,n_iter", [("schedules", 2, 5, 1), ("schedules", 3, 4, 2), ("img2img", 2, 5, 1), ("img2img", 3, 4, 2), ("img2img_fill2", 2, 5, 1),
                                                      ("img2img_fill2", 3, 4, 2)])
def test_sharded_prompt_editing_and_img2img_jobs_equal_the_single_process_job(kind, world, batch, n_iter):
    """`p.c` as the reference has it (MulticondLearnedConditioning with AND parts and a prompt-editing schedule, schedule lists for the
    negative prompt; tensors shared between images travel once) and an img2img job (per-image init latents, a shared latent mask, 'latent nothing' fill)
    split over 2 and 3 ranks: latents bit for bit those of the single-process job.  `img2img_fill2`: ONE shared init latent with the 'latent
    noise' fill -- the reference repeats the image to batch_size first (processing.py:1797-1799), so image i's fill comes from all_seeds[i], on whichever rank."""
    from forge_amd.modules import processing
    make = {"schedules": _make_schedule_job, "img2img": _make_img2img_job, "img2img_fill2": _make_img2img_fill2_job}[kind]
    want = processing.process_images(make(batch, n_iter))
    if kind == "img2img_fill2":
        first = want.latents[:batch].reshape(batch, -1)
        assert len({tuple(r.tolist()) for r in first[:, :]}) == batch
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker2, args=(r, world, port, kind, batch, n_iter, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = {}
    for _ in procs:
        r = q.get(timeout=180)
        res[r[0]] = r[1:]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    lat = torch.from_numpy(res[0][0])
    assert res[0][1] == want.seeds
    assert lat.shape == want.latents.shape and torch.equal(lat, want.latents)
    rows = want.latents.reshape(want.latents.shape[0], -1)
    assert len({tuple(r.tolist()) for r in rows}) == rows.shape[0], "every image of the job is its own"


def test_shared_tensors_of_a_schedule_travel_once():
    from forge_amd import distributed as fdist
    p = _make_schedule_job(6, 1)
    skel, tensors = fdist.flatten_tree(p.uc)
    assert len(tensors) == 1                                   # six images, one negative-prompt tensor
    skel, tensors = fdist.flatten_tree(p.c)
    assert len(tensors) == 1 + 3 + 6 + 6                       # the shared first segment, three own ones, six second segments, six AND parts
    sub = fdist.take_images(p.c, [4, 1])
    assert sub.shape == (2,) and sub.batch[0] is p.c.batch[4] and sub.batch[1] is p.c.batch[1]
