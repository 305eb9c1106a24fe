"""CPU, world_size 2 over gloo: the batch-shard helpers of the N > 1 path (forge_amd/distributed.py).  On the GPU node the
same code runs over RCCL (backend "nccl"); there is no data-path collective inside the step loop, only the per-job
broadcast of the conditioning and the gather of the latents."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import forge_amd  # noqa: F401
from forge_amd import distributed as fdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, with_dict, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import forge_amd  # noqa: F401
    from forge_amd import distributed as fd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7)
        cond_full = torch.randn(total, 5, 8, generator=g)
        unc_full = torch.randn(total, 5, 8, generator=g)
        vec_full = torch.randn(total, 6, generator=g)
        if with_dict:
            src_c = {"crossattn": cond_full, "vector": vec_full}
            src_u = {"crossattn": unc_full, "vector": vec_full * 0}
        else:
            src_c, src_u = cond_full, unc_full
        c, uc = fd.broadcast_conditioning(src_c if rank == 0 else None, src_u if rank == 0 else None, torch.device("cpu"))
        lo, hi = fd.shard_range(total, rank, world)
        cs, us = fd.slice_conditioning(c, lo, hi), fd.slice_conditioning(uc, lo, hi)
        ck = cs["crossattn"] if with_dict else cs
        ok_b = torch.equal(ck, cond_full[lo:hi])
        if with_dict:
            ok_b = ok_b and torch.equal(cs["vector"], vec_full[lo:hi]) and torch.equal(us["crossattn"], unc_full[lo:hi])
        # "sampling": a per-image function of (global index, cond) so the gathered result is sharding-independent
        local = torch.stack([ck[i].sum() * torch.ones(4, 3, 3) + (lo + i) for i in range(hi - lo)]) if hi > lo else torch.zeros(0, 4, 3, 3)
        got = fd.gather_latents(local, total)
        if rank == 0:
            want = torch.stack([cond_full[i].sum() * torch.ones(4, 3, 3) + i for i in range(total)])
            q.put((ok_b, bool(torch.equal(got, want))))
        else:
            q.put((ok_b, got is None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total,with_dict", [(4, False), (5, True), (1, False)])
def test_broadcast_shard_gather_world2(total, with_dict):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, with_dict, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(a and b for a, b in res), res


def _worker_flux_config5(rank, world, port, q):
    """BASELINE config 5's job shape: Flux.1-dev, global batch 16 -- T5 context [16, 256, 4096] + pooled [16, 768] in bf16, guidance [16] fp32, no uncond
    (distilled guidance, cfg scale 1), latents [16, 16, 128, 128] fp32 and decoded uint8 images [16, 1024, 1024, 3]"""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import forge_amd  # noqa: F401
    from forge_amd import distributed as fd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        total = 16
        g = torch.Generator().manual_seed(11)
        full = {"crossattn": torch.randn(total, 256, 4096, generator=g).bfloat16(), "vector": torch.randn(total, 768, generator=g).bfloat16(),
                "guidance": torch.full((total,), 3.5)}
        c, uc = fd.broadcast_conditioning(full if rank == 0 else None, None, torch.device("cpu"))
        assert uc is None
        lo, hi = fd.shard_range(total, rank, world)
        mine = fd.slice_conditioning(c, lo, hi)
        del c
        per = total // world
        ok = all(mine[k].shape[0] == per and mine[k].dtype == full[k].dtype and torch.equal(mine[k], full[k][lo:hi]) for k in full)
        # a rank keeps ITS images' conditioning only: the slices own their storage, the 16-image broadcast buffers are gone with `c`
        ok = ok and mine["crossattn"].untyped_storage().nbytes() == per * 256 * 4096 * 2
        lat = torch.stack([torch.full((16, 128, 128), float(lo + i)) + mine["crossattn"][i].float().mean() for i in range(per)])
        img = torch.stack([torch.full((1024, 1024, 3), (lo + i) * 7 % 251, dtype=torch.uint8) for i in range(per)])

        class Counting:          # what gather_batch allocates, in bytes (every tensor-creating call it makes goes through its module's `torch`)
            def __init__(self):
                self.bytes = 0

            def __getattr__(self, name):
                f = getattr(torch, name)
                if name not in ("empty", "zeros", "empty_like", "zeros_like", "cat", "stack"):
                    return f

                def counted(*a, **k):
                    r = f(*a, **k)
                    self.bytes += r.numel() * r.element_size()
                    return r
                return counted
        ct = Counting()
        fd.torch = ct
        try:
            got_lat = fd.gather_batch(lat, total)
            got_img = fd.gather_batch(img, total)
        finally:
            fd.torch = torch
        if rank == 0:
            want = torch.stack([torch.full((16, 128, 128), float(i)) + full["crossattn"][i].float().mean() for i in range(total)])
            ok = ok and torch.equal(got_lat, want) and all(int(got_img[i, 0, 0, 0]) == i * 7 % 251 for i in range(total))
            # the owner allocates the gathered batch and nothing else (no staging copy of its shard, no second copy for the reorder)
            ok = ok and ct.bytes == got_lat.numel() * 4 + got_img.numel()
        else:
            ok = ok and got_lat is None and got_img is None and ct.bytes == 0        # a non-owner's memory stays at its own shard
        q.put((rank, bool(ok), ct.bytes))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_flux_config5_job_shapes_shard_and_gather(world):
    """BASELINE config 5 (Flux.1-dev, batch 16 over 8 GPUs = 2 images per rank beside a 23.8 GB bf16 replica) through the broadcast /
    slice / gather helpers at world 2 and at the configuration's own world size 8 (gloo): every rank ends up with its own images' conditioning only,
    the gather allocates the result on the owner and nothing anywhere else."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_flux_config5, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), sorted(res)


def _worker_ragged(rank, world, port, q):
    """cond and uncond of different token counts and dtypes (a >75-token prompt with a short negative: [B,154,D] vs [B,77,D]; an SDXL dict
    cond with an fp32 vector next to fp16 cross-attention), and an absent uncond (cfg_scale 1)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import forge_amd  # noqa: F401
    from forge_amd import distributed as fd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(3)
        c_full = {"crossattn": torch.randn(4, 154, 8, generator=g).half(), "vector": torch.randn(4, 6, generator=g)}
        u_full = {"crossattn": torch.randn(4, 77, 8, generator=g).half(), "vector": torch.zeros(4, 6)}
        c, uc = fd.broadcast_conditioning(c_full if rank == 0 else None, u_full if rank == 0 else None, torch.device("cpu"))
        ok = all(torch.equal(c[k], c_full[k]) and c[k].dtype == c_full[k].dtype for k in c_full)
        ok = ok and all(torch.equal(uc[k], u_full[k]) and uc[k].dtype == u_full[k].dtype for k in u_full)
        c2, uc2 = fd.broadcast_conditioning(c_full["crossattn"] if rank == 0 else None, None, torch.device("cpu"))
        ok = ok and torch.equal(c2, c_full["crossattn"]) and uc2 is None and fd.slice_conditioning(uc2, 0, 2) is None
        q.put(bool(ok))
    finally:
        dist.destroy_process_group()


def test_broadcast_of_conds_that_differ_in_length_dtype_or_presence_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ragged, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(res), res


def test_shard_range_partitions():
    for total in (0, 1, 7, 8, 64, 65):
        for ws in (1, 2, 3, 8):
            spans = [fdist.shard_range(total, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_passthrough():
    c = torch.randn(2, 3)
    assert fdist.broadcast_conditioning(c, c, torch.device("cpu"))[0] is c
    assert fdist.gather_latents(c, 2) is c
