""""NV" noise source: Philox4x32-10 + Box-Muller, mirror of modules/rng_philox.py:32-102 (same `Generator` API), but the
counter-based generation runs on the MI355X (fmx_philox_randn) -- it is embarrassingly parallel integer work.  The four
raw Philox words are bit-exact against the reference algorithm (tests/test_gpu_kernels.py::test_philox_bit_exact); the
fp32 Box-Muller tail uses device libm (<= 2e-6 abs from numpy)."""
import torch

from .. import hipops as ops


class Generator:
    def __init__(self, seed, device="cuda"):
        self.seed = int(seed)
        self.offset = 0
        self.device = torch.device(device)

    def randn(self, shape):
        n = 1
        for s in shape:
            n *= int(s)
        out = ops.philox_randn(self.seed, self.offset, n, self.device)
        self.offset += 1
        return out.view(*shape)
