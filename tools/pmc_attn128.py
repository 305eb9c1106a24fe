"""A few launches of the Flux-shaped attention (2 x 24 heads x 4352 tokens, d_head 128) for rocprofv3 --pmc passes: what bounds the
wave-specialised kernel (matrix-pipe busy, LDS activity / waits)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import forge_amd  # noqa
from forge_amd import hipops as ops
from tools.bench_kernels import rnd

b, h, n, d = 2, 24, 4352, 128
q, k = rnd(b, n, h, d), rnd(b, n, h, d)
vt = rnd(h, d, b, n)
out = torch.empty(b * n, h * d, dtype=torch.float16, device="cuda")
for _ in range(3):
    ops.attention(q, k, vt, batch=b, heads=h, nq=n, nk=n, nk_pad=n, dpad=d, scale=d ** -0.5, q_bs=n * h * d, q_rs=h * d, k_bs=n * h * d, k_rs=h * d,
                  vt_bs=n, vt_hs=d * b * n, vt_ds=b * n, out=out)
torch.cuda.synchronize()
print("pmc_attn128 done")
