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
