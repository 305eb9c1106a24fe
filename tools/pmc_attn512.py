"""A few launches of the VAE's single-head 512-wide attention (8 x 16384 tokens) for rocprofv3 --pmc passes: LDS activity / bank conflicts and
matrix-pipe busy, to show what bounds fmx_attention_single_head512_f16."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import forge_amd  # noqa
from forge_amd import hipops as ops
from tools.bench_kernels import rnd

b, n, c = 8, 16384, 512
q, k = rnd(b * n, c), rnd(b * n, c)
vt = rnd(c, b * n)
out = torch.empty(b * n, c, dtype=torch.float16, device="cuda")
for _ in range(3):
    ops.attention_single_head512(q, k, vt, out, batch=b, nq=n, nk=n, nk_pad=n, q_bs=n * c, q_rs=c, k_bs=n * c, k_rs=c, vt_bs=n, vt_ds=b * n, scale=c ** -0.5)
torch.cuda.synchronize()
print("pmc_attn512 done")
