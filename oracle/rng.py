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


class ImageRNG:
    def __init__(self, shape, seeds, source="CPU"):
        self.shape = tuple(int(s) for s in shape)
        self.source = source
        if source == "NV":
            self.generators = [PhiloxGenerator(s) for s in seeds]
        else:
            self.generators = [torch.Generator("cpu").manual_seed(int(s)) for s in seeds]

    def next(self):
        if self.source == "NV":
            return torch.stack([torch.from_numpy(g.randn(self.shape)) for g in self.generators])
        return torch.stack([torch.randn(self.shape, generator=g) for g in self.generators])
