"""Multi-GPU batch sharding for the sampling path: one process per GPU, `torch.distributed` (backend "nccl" == RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  The reference is single-process single-GPU; images in a batch
are independent (no cross-sample op anywhere on the path), so the only communication per JOB is
  1. broadcast of the conditioning from rank 0 (text-cond [B,77,Dc] (+ pooled vector) for cond and uncond), and
  2. gather of the final latents and of the decoded uint8 images on rank 0 (`modules.processing.process_images_sharded` is the product entry
     that does 1 - 2 around the ordinary single-device job);
nothing is exchanged inside the step loop.  Seeds are seed + global image index, so results do not depend on the sharding.
"""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def group_active():
    """True when a default process group exists -- also one of a single rank: the collectives below then RUN (a world-1 broadcast / gather is a
    valid collective), which is how the RCCL code path is exercised on a one-GPU box (bench.py's `rccl_world1` leg, tests/test_gpu_rccl.py).  Without a
    process group they are the identity."""
    return dist.is_available() and dist.is_initialized()


def shard_range(total, rank, world_size):
    """Contiguous shard [lo, hi) of `total` images for `rank`; earlier ranks take the remainder."""
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _flatten(cond):
    if isinstance(cond, dict):
        keys = sorted(cond.keys())
        return keys, [cond[k] for k in keys]
    return None, [cond]


def _describe(cond):
    """Picklable description of one conditioning (tensor, dict of tensors, or None): per-tensor shape AND dtype -- cond and uncond
    differ in token count when only one of the prompts is longer than 75 tokens ([B,154,D] vs [B,77,D]; the reference does not pad
    them, backend/sampling/condition.py pads per step by lcm), and dict conds mix dtypes."""
    if cond is None:
        return None
    keys, tensors = _flatten(cond)
    return {"keys": keys, "shapes": [list(t.shape) for t in tensors], "dtypes": [t.dtype for t in tensors]}


def broadcast_conditioning(cond, uncond, device, src=0):
    """Rank `src` holds (cond, uncond) for the GLOBAL batch; every rank returns them with the same structure (a tensor, a dict of
    tensors, or None for an absent uncond at cfg_scale 1).  Each of the two carries its own metadata."""
    rank, ws = world()
    if not group_active():
        return cond, uncond
    meta = [None]
    if rank == src:
        meta = [(_describe(cond), _describe(uncond))]
    dist.broadcast_object_list(meta, src=src)
    out = []
    for which, m in zip((cond, uncond), meta[0]):
        if m is None:
            out.append(None)
            continue
        if rank == src:
            _, tensors = _flatten(which)
            tensors = [t.to(device).contiguous() for t in tensors]
        else:
            tensors = [torch.empty(s, dtype=d, device=device) for s, d in zip(m["shapes"], m["dtypes"])]
        for t in tensors:
            dist.broadcast(t, src=src)
        out.append(dict(zip(m["keys"], tensors)) if m["keys"] is not None else tensors[0])
    return out[0], out[1]


# ---- conditioning OBJECTS (round 5): what `p.c` / `p.uc` normally are in the reference -- a MulticondLearnedConditioning (AND parts with weights, each a
#      prompt-editing schedule) and a list of per-image schedules (modules/prompt_parser.py:129-147, 233-242, 294-365) -- travel as a picklable skeleton plus
#      the distinct tensors they hold; a tensor that many images share (the same prompt repeated over a batch: get_learned_conditioning caches per prompt
#      text) is broadcast ONCE.  Tensors and dicts of tensors are the degenerate trees.
def _tree_flatten(obj, tensors, seen):
    from .modules.prompt_parser import ComposableScheduledPromptConditioning, DictWithShape, MulticondLearnedConditioning, ScheduledPromptConditioning
    if obj is None or isinstance(obj, (bool, int, float)):
        return ("v", obj)
    if torch.is_tensor(obj):
        if id(obj) not in seen:
            seen[id(obj)] = len(tensors)
            tensors.append(obj)
        return ("t", seen[id(obj)])
    if isinstance(obj, ScheduledPromptConditioning):
        return ("sched", obj.end_at_step, _tree_flatten(obj.cond, tensors, seen))
    if isinstance(obj, ComposableScheduledPromptConditioning):
        return ("comp", _tree_flatten(obj.schedules, tensors, seen), float(obj.weight))
    if isinstance(obj, MulticondLearnedConditioning):
        return ("multi", tuple(obj.shape), _tree_flatten(obj.batch, tensors, seen))
    if isinstance(obj, dict):
        return ("dws" if isinstance(obj, DictWithShape) else "dict", {k: _tree_flatten(v, tensors, seen) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return ("list" if isinstance(obj, list) else "tuple", [_tree_flatten(v, tensors, seen) for v in obj])
    raise NotImplementedError(f"a {type(obj).__name__} inside a conditioning: the sharded entry broadcasts tensors, dicts of tensors and prompt-editing "
                              f"schedules (ScheduledPromptConditioning / MulticondLearnedConditioning)")


def _tree_build(node, tensors):
    from .modules.prompt_parser import ComposableScheduledPromptConditioning, DictWithShape, MulticondLearnedConditioning, ScheduledPromptConditioning
    kind = node[0]
    if kind == "v":
        return node[1]
    if kind == "t":
        return tensors[node[1]]
    if kind == "sched":
        return ScheduledPromptConditioning(node[1], _tree_build(node[2], tensors))
    if kind == "comp":
        return ComposableScheduledPromptConditioning(_tree_build(node[1], tensors), node[2])
    if kind == "multi":
        return MulticondLearnedConditioning(node[1], _tree_build(node[2], tensors))
    if kind in ("dict", "dws"):
        d = {k: _tree_build(v, tensors) for k, v in node[1].items()}
        return DictWithShape(d) if kind == "dws" else d
    seq = [_tree_build(v, tensors) for v in node[1]]
    return seq if kind == "list" else tuple(seq)


def flatten_tree(obj):
    """-> (skeleton, [distinct tensors]) ; raises NotImplementedError for a leaf that is none of the conditioning types (before any collective)"""
    tensors = []
    return _tree_flatten(obj, tensors, {}), tensors


def broadcast_tree(obj, device, src=0, flat=None):
    """Rank `src` holds `obj` (any nesting of the conditioning types above, or None); every rank returns an equal object whose tensors live on
    `device`.  flat: the owner's flatten_tree(obj), when it has already been taken (to validate before the first collective)."""
    rank, ws = world()
    if not group_active():
        return obj
    meta = [None]
    tensors = []
    if rank == src:
        skel, tensors = flat if flat is not None else flatten_tree(obj)
        meta = [(skel, [(list(t.shape), t.dtype) for t in tensors])]
    dist.broadcast_object_list(meta, src=src)
    skel, descr = meta[0]
    if rank == src:
        tensors = [t.to(device).contiguous() for t in tensors]
    else:
        tensors = [torch.empty(shape, dtype=dtype, device=device) for shape, dtype in descr]
    for t in tensors:
        if t.numel():
            dist.broadcast(t, src=src)
    return _tree_build(skel, tensors)


def take_images(cond, indices, device=None):
    """The per-image entries `indices` (global image numbers) of a conditioning for a whole job: rows of a tensor / of every tensor of a dict, the
    listed images' schedules of a schedule list or a MulticondLearnedConditioning."""
    from .modules.prompt_parser import MulticondLearnedConditioning
    if cond is None:
        return None
    if torch.is_tensor(cond):
        return cond.index_select(0, torch.as_tensor(indices, dtype=torch.long, device=cond.device)).contiguous()
    if isinstance(cond, dict):
        return type(cond)({k: take_images(v, indices) for k, v in cond.items()})
    if isinstance(cond, MulticondLearnedConditioning):
        return MulticondLearnedConditioning((len(indices),), [cond.batch[i] for i in indices])
    if isinstance(cond, (list, tuple)):
        return [cond[i] for i in indices]
    raise NotImplementedError(type(cond).__name__)


def slice_conditioning(cond, lo, hi):
    """rows [lo, hi) as tensors of their OWN (a row range of a contiguous tensor is already contiguous, so `.contiguous()` would hand back a view that
    keeps the whole global-batch buffer alive on every rank)"""
    if cond is None:
        return None
    own = lambda v: v[lo:hi].clone() if (hi - lo) < v.shape[0] else v     # noqa: E731
    if isinstance(cond, dict):
        return type(cond)({k: own(v) for k, v in cond.items()})
    return own(cond)


def gather_latents(local, total, dst=0):
    """All ranks pass their [b_local, ...] tensor (rank r holds images shard_range(total, r, world)); rank `dst` gets the [total, ...] batch in
    global order, every other rank None.  A true gather -- only `dst` receives (RCCL / gloo `gather` of equal-size padded shards; there is no
    native gatherv) -- not the all_gather of round 2, which delivered the whole batch to every rank."""
    return gather_batch(local, total, 1, dst)


def gather_batch(local, batch, n_iter=1, dst=0):
    """Gather of a job of `n_iter` iterations of `batch` images: rank r holds, iteration-major, its share shard_range(batch, r, world) of every
    iteration ([n_iter * b_r, ...]); rank `dst` returns [n_iter * batch, ...] in the order of the single-process job, other ranks None."""
    rank, ws = world()
    if not group_active():
        return local
    per = -(-batch // ws) * n_iter
    if local.shape[0] == per:
        pad = local.contiguous()         # an even split: the shard itself is what travels, no staging copy
    else:
        pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[:local.shape[0]] = local
    out = bufs = None
    if rank == dst:
        # the owner receives straight into the result's rows: one allocation of the gathered batch, nothing beside it; every other rank
        # allocates nothing (its memory stays at its own shard -- Flux config 5: 2 of 16 images per rank beside a 23.8 GB replica)
        out = torch.empty((ws * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        bufs = list(out.chunk(ws))
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    if n_iter == 1 and batch % ws == 0:
        return out                        # rank-major IS the job's order
    parts = []
    for n in range(n_iter):
        for r in range(ws):
            lo, hi = shard_range(batch, r, ws)
            b_r = hi - lo
            parts.append(bufs[r][n * b_r:(n + 1) * b_r])
    return torch.cat(parts)
