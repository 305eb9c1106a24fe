"""Conditioning containers -- mirror of backend/sampling/condition.py:17-142 for the txt2img path.

`compile_conditions` / `compile_weighted_conditions` keep the reference's output structure (list of dicts with
`cross_attn`, `model_conds{c_crossattn, y}`, `strength`) so callers written against Forge see the same objects.
"""
import math

import torch


def repeat_to_batch_size(tensor, batch_size):
    if tensor.shape[0] > batch_size:
        return tensor[:batch_size]
    if tensor.shape[0] < batch_size:
        return tensor.repeat([math.ceil(batch_size / tensor.shape[0])] + [1] * (tensor.dim() - 1))[:batch_size]
    return tensor


class Condition:
    def __init__(self, cond):
        self.cond = cond

    def process_cond(self, batch_size, device, **kwargs):
        return self.__class__(repeat_to_batch_size(self.cond, batch_size).to(device))

    def can_concat(self, other):
        return self.cond.shape == other.cond.shape

    def concat(self, others):
        return torch.cat([self.cond] + [o.cond for o in others])


class ConditionCrossAttn(Condition):
    def can_concat(self, other):
        s1, s2 = self.cond.shape, other.cond.shape
        if s1 != s2:
            if s1[0] != s2[0] or s1[2] != s2[2]:
                return False
            mult_min = s1[1] * s2[1] // math.gcd(s1[1], s2[1])
            if mult_min // min(s1[1], s2[1]) > 4:
                return False
        return True

    def concat(self, others):
        conds = [self.cond] + [o.cond for o in others]
        max_len = 1
        for c in conds:
            max_len = max_len * c.shape[1] // math.gcd(max_len, c.shape[1])
        return torch.cat([c.repeat(1, max_len // c.shape[1], 1) if c.shape[1] < max_len else c for c in conds])


def compile_conditions(cond):
    if cond is None:
        return None
    if isinstance(cond, torch.Tensor):
        return [dict(cross_attn=cond, model_conds=dict(c_crossattn=ConditionCrossAttn(cond)))]
    cross_attn, pooled = cond["crossattn"], cond["vector"]
    mc = dict(c_crossattn=ConditionCrossAttn(cross_attn), y=Condition(pooled))
    if "guidance" in cond:  # Flux distilled guidance (condition.py:104-119 with diffusion_engine/flux.py:92)
        mc["guidance"] = Condition(cond["guidance"])
    return [dict(cross_attn=cross_attn, pooled_output=pooled, model_conds=mc)]


_gather_cache = {}  # (crossattn data_ptr, shape, rows) -> gathered cond: AND-composed prompts gather the same rows every step; handing out the
                    # SAME tensors keeps the executor's per-conditioning K/V cache and captured graph valid from step to step


def _gather(cond, idx):
    ca = cond["crossattn"] if isinstance(cond, dict) else cond
    key = (ca.data_ptr(), tuple(ca.shape), tuple(idx))
    hit = _gather_cache.get(key)
    if hit is not None and hit[0] is ca:
        return hit[1]
    if len(_gather_cache) > 32:
        _gather_cache.clear()
    feed = cond.advanced_indexing(idx) if hasattr(cond, "advanced_indexing") else (
        {k: v[idx] for k, v in cond.items()} if isinstance(cond, dict) else cond[idx])
    _gather_cache[key] = (ca, feed)
    return feed


def compile_weighted_conditions(cond, weights):
    transposed = list(map(list, zip(*weights)))
    results = []
    for cond_pre in transposed:
        idx, weight = [], 0
        for i, w in cond_pre:
            idx.append(i)
            weight = w
        nb = (cond["crossattn"] if isinstance(cond, dict) else cond).shape[0]
        if idx == list(range(nb)):
            # plain prompts: the composition selects every image once, in order.  The reference still gathers
            # (`cond[current_indices]`, condition.py:133-136), which only copies; keeping the tensor itself keeps its
            # identity stable across steps, which is what the per-conditioning K/V cache and the HIP graph key on.
            feed = cond
        else:
            feed = _gather(cond, idx)
        h = compile_conditions(feed)
        h[0]["strength"] = weight
        results += h
    return results
