"""`ImageRNG` -- mirror of modules/rng.py:113-177: one generator per image seeded seed+i, `next()` stacks one tensor
per image, so results do not depend on how a batch is sharded across GPUs.  Sources (rng.py:6-33): "CPU"
(torch.Generator('cpu'), bit-reproducible across devices), "NV" (Philox, generated on the MI355X), "GPU" (device
generator; not reproducible across vendors)."""
import torch

from . import rng_philox, shared


def create_generator(seed, device):
    src = shared.opts.randn_source
    if src == "NV":
        return rng_philox.Generator(seed, device)
    dev = torch.device("cpu") if src == "CPU" else device
    return torch.Generator(dev).manual_seed(int(seed))


class ImageRNG:
    def __init__(self, shape, seeds, subseeds=None, subseed_strength=0.0, seed_resize_from_h=0, seed_resize_from_w=0, device="cuda"):
        if (subseeds is not None and subseed_strength != 0) or seed_resize_from_h > 0 or seed_resize_from_w > 0:
            raise NotImplementedError("subseed / seed-resize variations (rng.py:128-158) are outside the hot path")
        self.shape = tuple(map(int, shape))
        self.seeds = list(seeds)
        self.device = torch.device(device)
        self.generators = [create_generator(s, self.device) for s in self.seeds]
        self.is_first = True

    def _one(self, g):
        src = shared.opts.randn_source
        if src == "NV":
            return g.randn(self.shape)
        if src == "CPU":
            return torch.randn(self.shape, generator=g, device="cpu").to(self.device, non_blocking=True)
        return torch.randn(self.shape, generator=g, device=self.device)

    def first(self):
        xs = [self._one(g) for g in self.generators]
        delta = shared.opts.eta_noise_seed_delta or 0
        if delta:
            self.generators = [create_generator(s + delta, self.device) for s in self.seeds]
        return torch.stack(xs).to(self.device)

    def next(self):
        if self.is_first:
            self.is_first = False
            return self.first()
        return torch.stack([self._one(g) for g in self.generators]).to(self.device)
