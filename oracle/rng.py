"""TEST INFRASTRUCTURE ONLY (CPU oracle).

Restates the noise sources of modules/rng.py:113-177 (ImageRNG: one generator per image seeded
seed+i; `next()` = one tensor per image stacked) for the two device-independent sources:
  * "CPU": torch.Generator('cpu').manual_seed(seed) + torch.randn (rng.py:30-31, 89-95)
  * "NV" : Philox4x32-10 keyed by the seed, counter = (offset, 0, element index, 0), first output
           lane pair -> Box-Muller sine branch (modules/rng_philox.py:32-102).  Known-answer vector:
           the reference docstring (rng_philox.py:10-15), checked in tests/test_oracle_golden.py.
"""
import numpy as np
import torch

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10 over uint32 arrays; returns the four output words."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32).copy() for c in (c0, c1, c2, c3))
    k0 = np.asarray(k0, dtype=np.uint32).copy()
    k1 = np.asarray(k1, dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        for r in range(10):
            p0 = c0.astype(np.uint64) * _M0
            p1 = c2.astype(np.uint64) * _M1
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            if r != 9:
                k0 = k0 + _W0
                k1 = k1 + _W1
    return c0, c1, c2, c3


def philox_randn(seed, offset, n):
    idx = np.arange(n, dtype=np.uint32)
    z = np.zeros(n, dtype=np.uint32)
    k0 = np.full(n, seed & 0xFFFFFFFF, dtype=np.uint32)
    k1 = np.full(n, (seed >> 32) & 0xFFFFFFFF, dtype=np.uint32)
    g0, g1, _, _ = philox4x32_10(np.full(n, offset, dtype=np.uint32), z, idx, z, k0, k1)
    inv = np.array([2.3283064e-10], dtype=np.float32)
    inv2pi = np.array([2.3283064e-10 * 6.2831855], dtype=np.float32)
    u = g0 * inv + inv / 2
    v = g1 * inv2pi + inv2pi / 2
    return (np.sqrt(-2.0 * np.log(u)) * np.sin(v)).astype(np.float32)


class PhiloxGenerator:
    def __init__(self, seed):
        self.seed, self.offset = int(seed), 0

    def randn(self, shape):
        n = int(np.prod(shape))
        out = philox_randn(self.seed, self.offset, n).reshape(shape)
        self.offset += 1
        return out


def slerp(val, low, high):
    """modules/rng.py:100-110, applied by ImageRNG to ONE image's [C, H, W] noise: `dim=1` is therefore the H axis (the function was written
    for [B, D] batches; the reference's behaviour is what is restated, quirk included)."""
    low_n = low / torch.norm(low, dim=1, keepdim=True)
    high_n = high / torch.norm(high, dim=1, keepdim=True)
    dot = (low_n * high_n).sum(1)
    if dot.mean() > 0.9995:
        return low * val + high * (1 - val)
    omega = torch.acos(dot)
    so = torch.sin(omega)
    return (torch.sin((1.0 - val) * omega) / so).unsqueeze(1) * low + (torch.sin(val * omega) / so).unsqueeze(1) * high


class ImageRNG:
    """modules/rng.py:113-177 incl. variation seeds (subseeds + slerp, :133-146) and seed resize (:131, :148-160: the noise of the ORIGINAL
    size is centred into noise of the new size).  A tensor drawn "with seed s" and no generator (rng.py:13-33) is the first tensor of a fresh
    generator seeded s; with a generator it is the generator's next tensor."""

    def __init__(self, shape, seeds, source="CPU", subseeds=None, subseed_strength=0.0, seed_resize_from_h=0, seed_resize_from_w=0, eta_noise_seed_delta=0):
        self.shape = tuple(int(s) for s in shape)
        self.source, self.seeds = source, [int(s) for s in seeds]
        self.subseeds, self.strength = subseeds, subseed_strength
        self.rh, self.rw, self.delta = seed_resize_from_h, seed_resize_from_w, eta_noise_seed_delta
        self.generators = [self._gen(s) for s in self.seeds]
        self.is_first = True

    def _gen(self, seed):
        return PhiloxGenerator(seed) if self.source == "NV" else torch.Generator("cpu").manual_seed(int(seed))

    def _draw(self, g, shape):
        return torch.from_numpy(g.randn(shape)) if self.source == "NV" else torch.randn(shape, generator=g)

    def first(self):
        shape = self.shape
        nshape = shape if self.rh <= 0 or self.rw <= 0 else (shape[0], int(self.rh) // 8, int(self.rw // 8))
        xs = []
        for i, (seed, g) in enumerate(zip(self.seeds, self.generators)):
            sub = None
            if self.subseeds is not None and self.strength != 0:
                sub = self._draw(self._gen(0 if i >= len(self.subseeds) else self.subseeds[i]), nshape)
            noise = self._draw(self._gen(seed), nshape) if nshape != shape else self._draw(g, shape)
            if sub is not None:
                noise = slerp(self.strength, noise, sub)
            if nshape != shape:
                x = self._draw(g, shape)
                dx, dy = (shape[2] - nshape[2]) // 2, (shape[1] - nshape[1]) // 2
                w = nshape[2] if dx >= 0 else nshape[2] + 2 * dx
                h = nshape[1] if dy >= 0 else nshape[1] + 2 * dy
                tx, ty = max(dx, 0), max(dy, 0)
                dx, dy = max(-dx, 0), max(-dy, 0)
                x[:, ty:ty + h, tx:tx + w] = noise[:, dy:dy + h, dx:dx + w]
                noise = x
            xs.append(noise)
        if self.delta:
            self.generators = [self._gen(s + self.delta) for s in self.seeds]
        return torch.stack(xs)

    def next(self):
        if self.is_first:
            self.is_first = False
            return self.first()
        return torch.stack([self._draw(g, self.shape) for g in self.generators])
