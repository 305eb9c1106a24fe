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


def slerp(val, low, high):
    """rng.py:100-110 as ImageRNG uses it: on ONE image's [C, H, W] noise, so `dim=1` is the H axis (mirrored, not corrected)."""
    low_norm = low / torch.norm(low, dim=1, keepdim=True)
    high_norm = high / torch.norm(high, dim=1, keepdim=True)
    dot = (low_norm * high_norm).sum(1)
    if dot.mean() > 0.9995:
        return low * val + high * (1 - val)
    omega = torch.acos(dot)
    so = torch.sin(omega)
    return (torch.sin((1.0 - val) * omega) / so).unsqueeze(1) * low + (torch.sin(val * omega) / so).unsqueeze(1) * high


class ImageRNG:
    def __init__(self, shape, seeds, subseeds=None, subseed_strength=0.0, seed_resize_from_h=0, seed_resize_from_w=0, device="cuda"):
        self.shape = tuple(map(int, shape))
        self.seeds = list(seeds)
        self.subseeds = subseeds
        self.subseed_strength = subseed_strength
        self.seed_resize_from_h = seed_resize_from_h
        self.seed_resize_from_w = seed_resize_from_w
        self.device = torch.device(device)
        self.generators = [create_generator(s, self.device) for s in self.seeds]
        self.is_first = True

    def _one(self, g, shape=None):
        shape = self.shape if shape is None else shape
        src = shared.opts.randn_source
        if src == "NV":
            return g.randn(shape)
        if src == "CPU":
            return torch.randn(shape, generator=g, device="cpu").to(self.device, non_blocking=True)
        return torch.randn(shape, generator=g, device=self.device)

    def _seeded(self, seed, shape):
        """`randn(seed, shape)` without a generator (rng.py:13-33) seeds the GLOBAL generator and draws from it: the values are the first
        tensor of a fresh generator with that seed, which is what is drawn here -- without touching the process-wide RNG state."""
        return self._one(create_generator(seed, self.device), shape)

    def first(self):
        """rng.py:130-167: variation seeds (slerp towards the subseed's noise) and seed resize (noise of the original size centred into noise
        of the new size) are job set-up arithmetic on latent-sized tensors, done with torch ops like the reference."""
        shape = self.shape
        resize = self.seed_resize_from_h > 0 and self.seed_resize_from_w > 0
        noise_shape = (shape[0], int(self.seed_resize_from_h) // 8, int(self.seed_resize_from_w // 8)) if resize else shape
        xs = []
        for i, (seed, g) in enumerate(zip(self.seeds, self.generators)):
            subnoise = None
            if self.subseeds is not None and self.subseed_strength != 0:
                subnoise = self._seeded(0 if i >= len(self.subseeds) else self.subseeds[i], noise_shape)
            noise = self._seeded(seed, noise_shape) if noise_shape != shape else self._one(g)
            if subnoise is not None:
                noise = slerp(self.subseed_strength, noise, subnoise)
            if noise_shape != shape:
                x = self._one(g)
                dx, dy = (shape[2] - noise_shape[2]) // 2, (shape[1] - noise_shape[1]) // 2
                w = noise_shape[2] if dx >= 0 else noise_shape[2] + 2 * dx
                h = noise_shape[1] if dy >= 0 else noise_shape[1] + 2 * dy
                tx, ty = max(dx, 0), max(dy, 0)
                dx, dy = max(-dx, 0), max(-dy, 0)
                x[:, ty:ty + h, tx:tx + w] = noise[:, dy:dy + h, dx:dx + w]
                noise = x
            xs.append(noise)
        delta = shared.opts.eta_noise_seed_delta or 0
        if delta:
            self.generators = [create_generator(s + delta, self.device) for s in self.seeds]
        return torch.stack(xs).to(self.device)

    def next(self):
        if self.is_first:
            self.is_first = False
            return self.first()
        return torch.stack([self._one(g) for g in self.generators]).to(self.device)
